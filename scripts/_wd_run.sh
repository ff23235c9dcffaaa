cd /root/repo
timeout 300 python -m pytest tests/test_hip_parity.py -q -x -k "large_tile or cfg4 or cfg2 or llama or split" 2>&1 | tail -2
for sm in 0 1; do
echo "== SMALL=$sm"
QUANTO_HIP_NATIVE8_SMALL=$sm timeout 200 python scripts/microbench_qbytes.py --pairs i8:i8 f8:f8 --iters 50 --shapes 512x14336x4096 1024x8192x4096 2048x4096x4096 1280x8192x4096 1536x8192x4096 768x8192x4096 2>&1 | grep us
done
