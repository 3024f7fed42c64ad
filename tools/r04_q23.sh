#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_sweep; mkdir -p $O
for rep in 1 2 3; do for q in 2 3; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --queues $q > $O/q${q}_$rep.json 2> $O/q${q}_$rep.err || { tail -5 $O/q${q}_$rep.err; exit 1; }
  echo "queues $q #$rep: $(python3 -c "import json;d=json.load(open('$O/q${q}_$rep.json'));print(d['value'],d['ms_per_step'])")"
done; done
