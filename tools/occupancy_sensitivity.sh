#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/occupancy; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  echo "$name: $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])")"; }
run base A=1
run ka_1wave FSQ_DEBUG_KA_LDS_PAD=20480
run kb_1wave FSQ_DEBUG_KB_LDS_PAD=24000
