#!/bin/bash
# bench.py under different pipeline settings (no profiler): tools/r03_sweep.sh "--queues 1" "--queues 2 --inject-below 500000" ...
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sweep; mkdir -p $O
i=0
for args in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 2 $args > $O/$i.json 2> $O/$i.err || { tail -3 $O/$i.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/$i.json'));print('%-44s %.4g fits/s  %.1f ms/step' % ('$args', d['value'], d['ms_per_step']))"
done
