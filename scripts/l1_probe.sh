#!/bin/bash
# Next-round probe (not yet run): which stage of the TA -> vector L1 -> LDS path saturates in the lone 128-tile workgroup?
# three kernels with different instruction mixes all stream ~27 B/clk per CU.  Runs ON THE GPU BOX:
#   gpurun --timeout 600 -- 'bash scripts/l1_probe.sh 256x4096x4096 512x8192x8192'
# One rocprofv3 --pmc pass per counter group (never mixed with trace domains), the weights-direct 128-tile kernel
# (QUANTO_HIP_LARGE_WD=5 forces it) vs the LDS-weight loop (QUANTO_HIP_LARGE_WD=0), same shapes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/l1_probe; mkdir -p $OUT
SHAPES="${@:-256x4096x4096}"
GROUPS_=(
  "TA_BUSY_avr TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum"
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
  "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD"
)
for WD in 0 5; do
  for i in "${!GROUPS_[@]}"; do
    D=$OUT/wd${WD}_g$i
    (cd /tmp && QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_LARGE_CFG=2 QUANTO_HIP_LARGE_WD=$WD timeout 300 rocprofv3 --pmc ${GROUPS_[$i]} --output-format csv -d $D -o pmc -- \
        python $REPO/scripts/microbench_qbytes.py --kernel mfma_large --pairs bf16:i8 --iters 4 --ramp-ms 0 --shapes $SHAPES > $D.log 2>&1)
    python - "$D" "$WD" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counters in", sys.argv[1]); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "qbytes_mfma_large_kernel" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:64], r["Grid_Size"] if "Grid_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("WD=" + sys.argv[2], k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
  done
done
