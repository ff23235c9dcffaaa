set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v6; mkdir -p $OUT
export TMPDIR=/tmp QUANTO_HIP_EXPERIMENT=1
timeout 900 python -m pytest tests/test_multi_linear.py -q -x -m gpu -p no:cacheprovider --timeout 300 > $OUT/multi.log 2>&1; echo "multi exit=$?"; tail -8 $OUT/multi.log
timeout 600 python scripts/ab.py --workloads gateup_fused32 --env QUANTO_HIP_SKINNY_WIDE_MIN_BLOCKS=0,200 --rounds 5 > $OUT/ab_wide.jsonl 2>&1; tail -3 $OUT/ab_wide.jsonl
timeout 600 python scripts/ab.py --workloads gateup_fused32 --env QUANTO_HIP_SKINNY_LDS_KB=50,70,90,120,150 --rounds 5 > $OUT/ab_wide_lds.jsonl 2>&1; tail -6 $OUT/ab_wide_lds.jsonl
