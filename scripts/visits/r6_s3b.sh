set -x
O=gpurun_out/s3b; mkdir -p $O
cp optimum_quanto_amd/lib/libquanto_hip.so /tmp/cur.so
for rep in 1 2; do
for v in prev pk_a pk_b; do
  cp scripts/probes/libquanto_hip_$v.so optimum_quanto_amd/lib/libquanto_hip.so
  python scripts/ab_prefill.py --shapes 4096x4096 --ms 256 512 1024 --fused-env "BM=64,SPLIT=1" "BM=128,SPLIT=1" --variants mfma_fused4 > $O/fused4_${v}_$rep.jsonl 2>$O/err_${v}_$rep.txt
done
done
cp /tmp/cur.so optimum_quanto_amd/lib/libquanto_hip.so
python scripts/probes/fused4_hash.py > $O/hash_pk_b.jsonl 2>$O/hash.err
diff $O/hash_pk_b.jsonl gpurun_out/s3a/hash_prev.jsonl > /dev/null; echo "hash diff rc=$?"
cat $O/fused4_*.jsonl | head -50
