#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/queues; mkdir -p $O
for Q in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras --queues $Q > $O/q$Q.json 2> $O/q$Q.err || { tail -5 $O/q$Q.err; exit 1; }
  echo "queues $Q: $(python3 -c "import json;d=json.load(open('$O/q$Q.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])")"
done
