#!/bin/bash
# one bench line with the given arguments: tools/one_bench.sh --pipeline lanes --lanes 2
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/one_bench
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/one_bench/bench.json 2> gpurun_out/one_bench/bench.err || { tail -8 gpurun_out/one_bench/bench.err; exit 1; }
python3 -c "import json;d=json.load(open('gpurun_out/one_bench/bench.json'));print(d['metric'], d['value'],d['ms_per_step'],d['roofline']['frac'])"
