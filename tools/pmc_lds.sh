#!/bin/bash
# LDS / matrix-pipe counters of the fused backward step launched alone (tools/bench_bwd.py); run on the GPU box
# from the repo root: tools/pmc_lds.sh [lds|waves].  One counter pass, no traces; the per-kernel sums land in
# gpurun_out/pmc_lds.txt.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_lds
SET=${1:-lds}
if [ "$SET" = lds ]; then
  CTRS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
else  # where the waves' cycles go
  CTRS="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
fi
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/pmc_lds -- python3 $R/tools/bench_bwd.py 3 > $R/gpurun_out/pmc_lds.log 2>&1 || exit 1
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
