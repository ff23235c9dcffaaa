set -x
O=gpurun_out/s3f; mkdir -p $O
./scripts/probes/large_tile_timing.bin random > $O/timeline_random_halves.txt 2>&1
grep -A8 "K=4096" $O/timeline_random_halves.txt | head -9
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "large or cfg2 or cfg4 or qbytes_mm or bias" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cp optimum_quanto_amd/lib/libquanto_hip.so /tmp/cur.so
for rep in 1 2; do
for v in prev cur; do
  if [ $v = prev ]; then cp scripts/probes/libquanto_hip_prev.so optimum_quanto_amd/lib/libquanto_hip.so; else cp /tmp/cur.so optimum_quanto_amd/lib/libquanto_hip.so; fi
  python scripts/ab.py --workloads cfg2 cfg4 fp8_4k --env QUANTO_HIP_GROUP_M=- --sequential --rounds 9 > $O/ab_${v}_$rep.jsonl 2>$O/ab_${v}_$rep.err
done
done
cp /tmp/cur.so optimum_quanto_amd/lib/libquanto_hip.so
for f in $O/ab_*.jsonl; do echo $f; cut -c1-130 $f; done
