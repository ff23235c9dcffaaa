#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3r; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python scripts/ab_prefill.py --shapes 4096x4096 4096x14336 14336x4096 1024x4096 --ms 72 128 192 256 384 512 --variants mfma_fused4 --fused-env "" "BM=64,SPLIT=2" "BM=64,SPLIT=4" "BM=128,SPLIT=4" > $O/plan.jsonl 2> $O/plan.err; cat $O/plan.jsonl; tail -2 $O/plan.err
timeout 900 python scripts/ab.py --workloads int4_decode32 int4_decode64 int4_decode32_up int4_decode32_down qkv_fused32 gateup_fused32 --env QUANTO_HIP_SKINNY_SPLIT=1,2,4,8 --rounds 5 > $O/ab_split.jsonl 2> $O/ab_split.err; cut -c1-180 $O/ab_split.jsonl; tail -2 $O/ab_split.err
for s in 1 2 4; do echo "LARGE_SPLIT=$s"; QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_LARGE_SPLIT=$s timeout 300 python scripts/microbench_qbytes.py --shapes 128x4096x4096 256x4096x4096 512x4096x4096 256x8192x8192 512x4096x14336 1024x4096x4096 --pairs bf16:i8 --graph 2>/dev/null | cut -c1-200; done > $O/large_split.txt; cat $O/large_split.txt
