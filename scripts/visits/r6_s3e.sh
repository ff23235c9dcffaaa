set -x
O=gpurun_out/s3g; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "native8 or w8a8 or fp8a8 or a8 or activations or prefill or dense or dequant or quantized_act" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cp optimum_quanto_amd/lib/libquanto_hip.so /tmp/cur.so
for rep in 1 2; do
for v in prev cur; do
  if [ $v = prev ]; then cp scripts/probes/libquanto_hip_prev.so optimum_quanto_amd/lib/libquanto_hip.so; else cp /tmp/cur.so optimum_quanto_amd/lib/libquanto_hip.so; fi
  python scripts/ab.py --workloads w8a8 fp8a8 cfg4_fp8a8 cfg4_w8a8 w8a8_down512 int4_prefill cfg2 --env QUANTO_HIP_GROUP_M=- --sequential --rounds 9 > $O/ab_${v}_$rep.jsonl 2>$O/ab_${v}_$rep.err
done
done
cp /tmp/cur.so optimum_quanto_amd/lib/libquanto_hip.so
for f in $O/ab_*.jsonl; do echo $f; cut -c1-130 $f; done
