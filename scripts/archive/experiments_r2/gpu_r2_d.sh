#!/bin/bash
# round 2, GPU call D: retuned streaming kernels (parity + timings), PMC 16x16x32 vs 32x32x16, generation bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
( timeout 600 python -m pytest tests/test_hip_parity.py tests/test_dispatch_fuzz_gpu.py tests/test_backward_and_workspace.py -m gpu -x -q -k "skinny or fuzz or workspace or batched or split" 2>&1 | tail -5 ) > $O/pytest.log
tail -3 $O/pytest.log
python scripts/ab.py --workloads int4_decode8 int4_decode32 int4_decode64 int4_decode32_up int4_decode32_down int8_decode32 --env QUANTO_HIP_SKINNY_LDS_KB=50,150 --rounds 7 > $O/ab_skinny.jsonl 2>$O/err.txt
python scripts/ab.py --workloads int4_decode32_down --env QUANTO_HIP_SKINNY_SPLIT=2,4,8 --rounds 7 >> $O/ab_skinny.jsonl 2>>$O/err.txt
cat $O/ab_skinny.jsonl
for CFG in 0 4; do
QUANTO_HIP_LARGE_CFG=$CFG bash scripts/pmc.sh cfg2 sq_$CFG SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT > $O/pmc_cfg$CFG.txt 2>&1
QUANTO_HIP_LARGE_CFG=$CFG bash scripts/pmc.sh cfg2 sq2_$CFG SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES > $O/pmc2_cfg$CFG.txt 2>&1
tail -n 2 $O/pmc_cfg$CFG.txt; tail -n 2 $O/pmc2_cfg$CFG.txt
done
rm -rf gpurun_out/pmc_cfg2_sq*   # raw counter csvs: only the summaries above travel back
( timeout 900 python scripts/bench_generate.py --batch 1 32 --drivers graph --new 128 > $O/gen_graph.jsonl 2> $O/gen_graph.err ); cat $O/gen_graph.jsonl
( timeout 900 python scripts/bench_generate.py --batch 1 32 --drivers graph --new 128 --fuse > $O/gen_graph_fused.jsonl 2> $O/gen_graph_fused.err ); cat $O/gen_graph_fused.jsonl
