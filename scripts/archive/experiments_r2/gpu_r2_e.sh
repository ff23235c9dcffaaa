#!/bin/bash
# round 2, GPU call E: steady-state (sequential) comparison of the large-tile configurations; skinny 8-wave blocks
set -u
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
python scripts/ab.py --workloads cfg2 --env QUANTO_HIP_LARGE_CFG=0,4,0,4 --rounds 9 --sequential --ramp-ms 400 > $O/ab.jsonl 2>$O/err.txt
python scripts/ab.py --workloads int8_8k fp8_4k --env QUANTO_HIP_LARGE_CFG=0,4 --rounds 7 --sequential --ramp-ms 400 >> $O/ab.jsonl 2>>$O/err.txt
python scripts/ab.py --workloads int4_decode32 int4_decode32_up int4_decode64 --env QUANTO_HIP_SKINNY_WAVES=4,8 --rounds 7 >> $O/ab.jsonl 2>>$O/err.txt
for S in 2 4 8; do QUANTO_HIP_SKINNY_WAVES=8 python scripts/ab.py --workloads int4_decode32 int4_decode32_up --env QUANTO_HIP_SKINNY_SPLIT=$S --rounds 5 2>>$O/err.txt | sed 's/^{/{"waves": 8, /' >> $O/ab.jsonl; done
cat $O/ab.jsonl
for CFG in 0 4; do QUANTO_HIP_LARGE_CFG=$CFG python bench.py --no-sub --no-cpu-baseline --steps 50 2>>$O/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench cfg', $CFG, d['value'], d['roofline']['launch_us'])"; done
