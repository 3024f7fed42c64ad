#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + HBM traffic counters of bench.py; results under gpurun_out/.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=${1:-r01}
LANES=${2:-2}
mkdir -p gpurun_out/$R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --lanes $LANES > gpurun_out/$R/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --fields 256 --lanes 1 > gpurun_out/$R/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --fields 256 --lanes 1 > gpurun_out/$R/bench_write.log 2>&1
python3 tools/summarize_profiles.py $R
