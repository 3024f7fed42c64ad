#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats of the default bench (stream pipeline), HBM traffic counters and
# SQ issue counters of the LM-fit kernels (separate --pmc passes, MI355X_MICROARCH.md); results under gpurun_out/<tag>/.
# usage: tools/collect_profiles.sh <tag>      then copy gpurun_out/<tag>/{summary.md,summary.json,fit_counters.json} into profiles/
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=${1:-r03}
rm -rf gpurun_out/$R; mkdir -p gpurun_out/$R
B="python3 bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- $B --steps 6 --warmup 1 > gpurun_out/$R/bench_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats_q1 -- $B --steps 6 --warmup 1 --queues 1 > gpurun_out/$R/bench_stats_q1.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats_cfg3 -- $B --config 3 --steps 10 --warmup 2 > gpurun_out/$R/bench_stats_cfg3.log 2>&1
echo "cfg3 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats_f32 -- python3 tools/bench_f32.py 1024 4 > gpurun_out/$R/bench_stats_f32.log 2>&1
echo "f32 stats done"
# counters with ONE fit queue: kernels of a single stream do not overlap, so per-kernel durations add up to busy time
S="--steps 2 --warmup 0 --fields 256 --queues 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmc_fetch -- $B $S > gpurun_out/$R/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/pmc_write -- $B $S > gpurun_out/$R/bench_write.log 2>&1
echo "traffic done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/$R/pmc_sq1 -- $B $S > gpurun_out/$R/bench_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS_F64 --output-format csv -d gpurun_out/$R/pmc_sq2 -- $B $S > gpurun_out/$R/bench_sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU --output-format csv -d gpurun_out/$R/pmc_sq3 -- $B $S > gpurun_out/$R/bench_sq3.log 2>&1
echo "sq done"
python3 tools/summarize_profiles.py $R
