#!/bin/bash
# closing run of round 3, part b: the reference's own tests in plug-in mode and the cfg5 generation numbers at the final kernel state
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3final_b; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_backward_and_workspace.py tests/test_reference_integration.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
bash scripts/run_reference_tests_gpu.sh > $O/g4_tail.txt 2>&1; tail -6 $O/g4_tail.txt
timeout 1500 python scripts/bench_generate.py --batch 1 32 --prompt 512 --new 512 --drivers reference graph --iterations 2 --fuse > $O/cfg5_generation.jsonl 2> $O/cfg5_generation.err; cat $O/cfg5_generation.jsonl | cut -c1-400; tail -2 $O/cfg5_generation.err
