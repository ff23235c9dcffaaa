#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2z; O=gpurun_out/r2z
timeout 600 python -m pytest tests/test_reference_style_gpu.py -x -q -m gpu > $O/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -n 3 $O/pytest2.log
for f in "" "--fuse"; do
timeout 400 python scripts/bench_generate.py --batch 1 32 --drivers graph $f --new 128 > $O/gen_ab$f.log 2>&1
echo "== fuse[$f]"; grep decode_tokens $O/gen_ab$f.log | grep -o '"batch": [0-9]*\|"ms_per_token": [0-9.]*\|"decode_tokens_per_s": [0-9.]*' | paste - - -
done
