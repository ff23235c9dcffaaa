set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v7; mkdir -p $OUT
export TMPDIR=/tmp QUANTO_HIP_EXPERIMENT=1
for bm in 64 128; do
QUANTO_HIP_A8_BM=$bm timeout 600 python scripts/ab.py --workloads w4a8 w4afp8 --env QUANTO_HIP_A8_ABLATE=0,1,2,3,4,7,8,15 --rounds 3 > $OUT/ab_a8_ablate_bm$bm.jsonl 2>&1; grep '^{' $OUT/ab_a8_ablate_bm$bm.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('bm$bm', d['workload'], 'ablate', d['QUANTO_HIP_A8_ABLATE'], d['us_median'])"
done
