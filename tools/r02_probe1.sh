#!/bin/bash
# r02 probe: lane scaling of the bench and the alive-fits curve of one batch (FSQ_DEBUG_TRACE)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_probe1; mkdir -p $O
for L in 1 2 3 4; do
  timeout -k 10 240 python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --lanes $L > $O/bench_l$L.json 2> $O/bench_l$L.err || exit 1
  echo "lanes $L: $(python3 -c "import json;d=json.load(open('$O/bench_l$L.json'));print(d['value'],d['ms_per_step'])")"
done
FSQ_DEBUG_TRACE=1 FSQ_SYNC_EVERY=1 timeout -k 10 240 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 > $O/trace.json 2> $O/trace.err || exit 1
grep -c round $O/trace.err
