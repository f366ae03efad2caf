#!/bin/bash
# Counter passes and a kernel trace over `bench.py --kernels-only`; run on the GPU box from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_trace
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --kernels-only > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --kernels-only > $R/gpurun_out/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_trace -- python3 $R/bench.py --kernels-only > $R/gpurun_out/pmc_trace.log 2>&1
cd $R && python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_traffic.json
