set -x
O=gpurun_out/s3j; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "qbytes and (skinny or multi or decode or streaming or batched) or test_qbytes or int8" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cp optimum_quanto_amd/lib/libquanto_hip.so /tmp/cur.so
for rep in 1 2; do
for v in prev cur; do
  if [ $v = prev ]; then cp scripts/probes/libquanto_hip_prev.so optimum_quanto_amd/lib/libquanto_hip.so; else cp /tmp/cur.so optimum_quanto_amd/lib/libquanto_hip.so; fi
  python scripts/ab.py --workloads int8_decode32 int8_qkv_fused32 int8_gateup_fused32 --env QUANTO_HIP_GROUP_M=- --rounds 9 > $O/ab_${v}_$rep.jsonl 2>$O/ab_${v}_$rep.err
done
done
cp /tmp/cur.so optimum_quanto_amd/lib/libquanto_hip.so
for f in $O/ab_*.jsonl; do echo $f; cut -c1-150 $f; done
