#!/bin/bash
# A/B of library builds with the driver's bench command (20 steps, warm-up 5), each build twice, interleaved
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_ab20; mkdir -p $O
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = prod ]; then unset FSQ_HIP_LIB; else export FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/variants/libfsq_$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/${v}_$rep.json 2> $O/${v}_$rep.err || { tail -5 $O/${v}_$rep.err; exit 1; }
  echo "$v #$rep: $(python3 -c "import json;d=json.load(open('$O/${v}_$rep.json'));print(d['value'],d['ms_per_step'])")"
done
done
