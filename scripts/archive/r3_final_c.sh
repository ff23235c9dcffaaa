#!/bin/bash
# closing run of round 3, part c: HBM counters of the remaining default sub-results, SQ counters of the fused int4 GEMM and the streaming kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3final_c; mkdir -p $O; export TMPDIR=/tmp
timeout 120 python scripts/stress_splitk.py > $O/stress.log 2>&1; echo "stress rc=$?"; tail -2 $O/stress.log
PMC_FOR="cfg3 qkv_fused cfg4 int8_gateup_fused" timeout 900 bash scripts/collect_profiles.sh r03 cfg3 qkv_fused cfg4 int8_gateup_fused > $O/collect.log 2>&1; echo "collect rc=$?"
for W in int4_prefill512 int4_decode32; do
  bash scripts/pmc.sh $W sq1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU > $O/sq_${W}_1.txt 2>&1
  bash scripts/pmc.sh $W sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE > $O/sq_${W}_2.txt 2>&1
  tail -2 $O/sq_${W}_1.txt $O/sq_${W}_2.txt
done
