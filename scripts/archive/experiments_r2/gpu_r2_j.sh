#!/bin/bash
# where do the 10.6 us of the M=32 streaming kernel go: ablations (timing only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2j; O=gpurun_out/r2j
timeout 600 python scripts/ab.py --workloads int4_decode32 int4_decode32_up int4_decode8 --env QUANTO_HIP_SKINNY_ABLATE=0,1,2,3,6,14,30,31 --rounds 5 > $O/ablate.txt 2>&1
QUANTO_HIP_SKINNY_LDS_KB=150 timeout 600 python scripts/ab.py --workloads int4_decode32 --env QUANTO_HIP_SKINNY_ABLATE=0,1,2,3,31 --rounds 5 > $O/ablate_deep.txt 2>&1
cat $O/ablate.txt $O/ablate_deep.txt
