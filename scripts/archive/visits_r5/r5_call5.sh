#!/bin/bash
# round 5, visit 5: streaming (batched decode) kernels with the activation fragments hoisted - parity, ring depth x split re-sweep, timeline, host cost
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c5; mkdir -p $OUT; export TMPDIR=/tmp
export QUANTO_HIP_EXPERIMENT=1
echo "== parity"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_multi_linear.py tests/test_dispatch_fuzz_gpu.py tests/test_backward_and_workspace.py -m gpu -q -p no:cacheprovider -x --timeout 300 \
  -k "skinny or batched_decode or multi or fuzz or e4m3fnuz or golden or auto_picks or workspace or qlinear" 2>&1 | tail -4 | tee $OUT/parity_tail.txt
echo "== sweep"
for SPLIT in 2 4 8; do
  QUANTO_HIP_SKINNY_SPLIT=$SPLIT timeout 200 python scripts/ab.py --rounds 5 --workloads int4_decode32 --env QUANTO_HIP_SKINNY_LDS_KB=50,76,100,150 2>&1 | grep -v Warning | sed "s/^{/{\"split\": $SPLIT, /" | tee -a $OUT/ring_depth_x_split.jsonl
done
timeout 300 python scripts/ab.py --rounds 5 --workloads int4_decode32 int4_decode8 int4_decode16 int4_decode64 int4_decode32_down int4_decode32_up qkv_fused32 gateup_fused32 int8_decode32 int8_gateup_fused32 --env QUANTO_HIP_SKINNY_LDS_KB=50,100 2>&1 | grep -v Warning | tee $OUT/ring_depth_shapes.jsonl
for A in 0 1 3 31; do
  QUANTO_HIP_SKINNY_ABLATE=$A timeout 100 python scripts/ab.py --rounds 5 --workloads int4_decode32 --env QUANTO_HIP_SKINNY_LDS_KB=50,100 2>&1 | grep -v Warning | sed "s/^{/{\"ablate\": $A, /" | tee -a $OUT/ablate.jsonl
done
timeout 200 python scripts/skinny_timeline.py 2>&1 | grep -v Warning | tail -16 | tee $OUT/timeline_lds50.txt
QUANTO_HIP_SKINNY_LDS_KB=100 timeout 200 python scripts/skinny_timeline.py 2>&1 | grep -v Warning | tail -16 | tee $OUT/timeline_lds100.txt
echo "== host overhead"
QUANTO_HIP_EXPERIMENT=0 timeout 120 python scripts/host_overhead.py 2>&1 | grep -v Warning | tee $OUT/host_overhead.jsonl
