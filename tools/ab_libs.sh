#!/bin/bash
# A/B of experimental library builds: tools/r02_exp.sh <lib.so> ...
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab_libs; mkdir -p $O
for lib in "$@"; do
  name=$(basename $lib .so)
  FSQ_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  echo "$name: $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])")"
done
