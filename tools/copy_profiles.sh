#!/bin/bash
# copy what tools/collect_profiles.sh <tag> left under gpurun_out/<tag>/ into profiles/ (run here, after gpurun merged it back)
# usage: tools/copy_profiles.sh <tag>
set -e
cd "$(dirname "$0")/.."
R=${1:-r03}
cp gpurun_out/$R/summary.md profiles/${R}_summary.md
cp gpurun_out/$R/summary.json profiles/${R}_summary.json
cp gpurun_out/$R/fit_counters.json profiles/${R}_fit_counters.json
cp gpurun_out/$R/fit_counters.json profiles/fit_counters_latest.json
for t in stats stats_q1 stats_cfg3 stats_f32; do
  f=$(ls -t gpurun_out/$R/$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] || continue
  n=$(echo $t | sed 's/^stats$/bench/; s/^stats_q1$/bench_queues1/; s/^stats_cfg3$/bench_cfg3/; s/^stats_f32$/bench_f32/')
  cp "$f" profiles/${R}_${n}_kernel_stats.csv
done
python3 -c "
import json, sys
sys.path.insert(0, '.')
from fluorosequencingimageanalysis_amd import _native as N
p = json.load(open('profiles/fit_counters_latest.json'))
print('counters taken on sources', p['source_sha16'], '- tree now', N.source_sha16(), '-', 'MATCH' if p['source_sha16'] == N.source_sha16() else 'STALE')"
