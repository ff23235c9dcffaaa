#!/bin/bash
# round 2, GPU call F: full parity suite after the GEMV generalisation, raster / native8 knobs
set -u
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python scripts/ab.py --workloads cfg2 --env QUANTO_HIP_GROUP_M=-,1,2,4,8 --rounds 7 > $O/ab.jsonl 2>$O/err.txt
python scripts/ab.py --workloads w8a8 fp8a8 --env QUANTO_HIP_GROUP_M=-,1,2,4,8 --rounds 5 >> $O/ab.jsonl 2>>$O/err.txt
python scripts/ab.py --workloads w8a8 fp8a8 --env QUANTO_HIP_PAIRED=-,0,1 --rounds 5 >> $O/ab.jsonl 2>>$O/err.txt
python scripts/ab.py --workloads int4_prefill int4_prefill512 --env QUANTO_HIP_DENSE_WD=1,0 --rounds 5 >> $O/ab.jsonl 2>>$O/err.txt
cat $O/ab.jsonl
