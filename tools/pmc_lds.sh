#!/bin/bash
# LDS / matrix-pipe / memory-path counters of one recurrence launch run alone (tools/bench_bwd.py, or the program
# named by BENCH, e.g. BENCH=tools/bench_fwd.py); run on the GPU box from the repo root:
# tools/pmc_lds.sh [lds|waves|vmem|ta].  One counter pass per call, no traces; the per-kernel sums land in
# gpurun_out/pmc_lds.txt.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_lds
SET=${1:-lds}
if [ "$SET" = lds ]; then
  CTRS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
elif [ "$SET" = vmem ]; then  # issue of vector memory instructions
  CTRS="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
elif [ "$SET" = ta ]; then  # the address / data path behind them
  CTRS="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE"
else  # where the waves' cycles go
  CTRS="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
fi
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/pmc_lds -- python3 $R/${BENCH:-tools/bench_bwd.py} 3 > $R/gpurun_out/pmc_lds.log 2>&1 || exit 1
cd $R && python3 - <<'PY' > gpurun_out/pmc_lds.txt
import collections, csv, glob
rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            calls[k] += 1
for k, c in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:12]:
    if "fused" not in k and "pp_kernel" not in k:
        continue
    n = max(calls[k], 1)
    print(k, "launches", n)
    for name, v in sorted(c.items()):
        print("   %-28s %.4g per launch" % (name, v / n))
PY
cat $R/gpurun_out/pmc_lds.txt
