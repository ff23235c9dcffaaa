#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3s; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
for k in skinny mfma_large; do echo "kernel=$k"; QUANTO_HIP_EXPERIMENT=1 timeout 300 python scripts/microbench_qbytes.py --shapes 72x4096x4096 96x4096x4096 72x1024x4096 96x1024x4096 96x14336x4096 --pairs bf16:i8 --graph --kernel $k 2>/dev/null | cut -c1-200; done > $O/qbytes_7296.txt; cat $O/qbytes_7296.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
