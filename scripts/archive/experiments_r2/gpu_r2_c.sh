#!/bin/bash
# round 2, GPU call C: skinny knobs (nt DMA, K split, blocks per CU), fp8 single-op conversion, counters 16x16x32 vs 32x32x16
set -u
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
( timeout 600 python -m pytest tests/test_hip_parity.py tests/test_dispatch_fuzz_gpu.py -m gpu -x -q -k "skinny or large or cfg4 or fp8 or e4m3 or e5m2 or fuzz" 2>&1 | tail -8 ) > $O/pytest.log
tail -4 $O/pytest.log
python scripts/ab.py --workloads int4_decode32 int4_decode32_up --env QUANTO_HIP_SKINNY_NT=0,1 --rounds 7 > $O/ab_skinny.jsonl 2>$O/err.txt
for S in 2 4 8; do for KB in 150 76 50; do
  QUANTO_HIP_SKINNY_LDS_KB=$KB python scripts/ab.py --workloads int4_decode32 int4_decode32_up --env QUANTO_HIP_SKINNY_SPLIT=$S --rounds 5 2>>$O/err.txt | sed "s/^{/{\"lds_kb\": $KB, /" >> $O/ab_skinny.jsonl
done; done
python scripts/ab.py --workloads cfg4 fp8_4k --env QUANTO_HIP_LARGE_WD=1,0 --rounds 7 >> $O/ab_skinny.jsonl 2>>$O/err.txt
cat $O/ab_skinny.jsonl
for CFG in 0 4; do
QUANTO_HIP_LARGE_CFG=$CFG bash scripts/pmc.sh cfg2 sq_$CFG SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT > $O/pmc_cfg$CFG.txt 2>&1
QUANTO_HIP_LARGE_CFG=$CFG bash scripts/pmc.sh cfg2 sq2_$CFG SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES > $O/pmc2_cfg$CFG.txt 2>&1
tail -2 $O/pmc_cfg$CFG.txt $O/pmc2_cfg$CFG.txt
done
