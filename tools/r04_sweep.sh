#!/bin/bash
# bench lines under different pipeline settings (queues, depth)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_sweep; mkdir -p $O
for args in "--queues 2" "--queues 3" "--queues 4" "--queues 2 --depth 24" "--queues 3 --depth 12"; do
  n=$(echo $args | tr -d ' -')
  timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras $args > $O/$n.json 2> $O/$n.err || { tail -5 $O/$n.err; exit 1; }
  echo "$args: $(python3 -c "import json;d=json.load(open('$O/$n.json'));print(d['value'],d['ms_per_step'])")"
done
