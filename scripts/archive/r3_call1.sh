#!/bin/bash
# round 3, GPU visit 1: parity of the new 4-wave AGPR layouts, their A/B against the default, G4 (reference tests, plug-in mode),
# new hygiene tests, the default bench line, power / clock readings, vendor-kernel counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3a; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "large_tile_configurations or four_wave" -p no:cacheprovider -x > $O/parity_large.log 2>&1; echo "parity_large rc=$?"; tail -3 $O/parity_large.log
timeout 600 python scripts/ab.py --workloads cfg2 --env QUANTO_HIP_LARGE_CFG=0,1,5 --sequential --rounds 7 > $O/ab_cfg2_seq.jsonl 2> $O/ab_cfg2_seq.err; cat $O/ab_cfg2_seq.jsonl
timeout 600 python scripts/ab.py --workloads cfg2 fp8_4k int8_8k --env QUANTO_HIP_LARGE_CFG=0,1,5 --rounds 7 > $O/ab_inter.jsonl 2> $O/ab_inter.err; cat $O/ab_inter.jsonl
timeout 600 python scripts/power_probe.py --cfgs 0 1 5 --matmul --seconds 3 > $O/power.jsonl 2> $O/power.err; cat $O/power.jsonl; tail -2 $O/power.err
bash scripts/run_reference_tests_gpu.sh > $O/g4_tail.txt 2>&1; tail -8 $O/g4_tail.txt
timeout 900 python -m pytest tests/test_reference_integration.py tests/test_multi_linear.py tests/test_backward_and_workspace.py -q -m gpu -p no:cacheprovider -k "plugin_mode_runs or sibling or bias_gradient or c_api_multi" > $O/new_tests.log 2>&1; echo "new_tests rc=$?"; tail -5 $O/new_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_default.json)"; tail -2 $O/bench_default.err
C="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
bash scripts/pmc_matmul.sh sq $C > $O/pmc_matmul_sq.txt 2>&1; tail -4 $O/pmc_matmul_sq.txt
bash scripts/pmc_matmul.sh grbm GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > $O/pmc_matmul_grbm.txt 2>&1; tail -4 $O/pmc_matmul_grbm.txt
for CFG in 0 1 5; do QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_LARGE_CFG=$CFG bash scripts/pmc.sh cfg2 sq$CFG $C > $O/pmc_cfg2_sq$CFG.txt 2>&1; tail -2 $O/pmc_cfg2_sq$CFG.txt; done
QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_LARGE_CFG=5 bash scripts/pmc.sh cfg2 grbm5 GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > $O/pmc_cfg2_grbm5.txt 2>&1; tail -2 $O/pmc_cfg2_grbm5.txt
QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_LARGE_CFG=0 bash scripts/pmc.sh cfg2 grbm0 GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > $O/pmc_cfg2_grbm0.txt 2>&1; tail -2 $O/pmc_cfg2_grbm0.txt
