#!/bin/bash
# HIP runtime knobs against the launch- / latency-bound parts: single-image latency (a pass is ~400 small kernel launches) and the bench
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_knobs; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout -k 10 200 python3 tools/bench_single.py 30 > $O/single_$tag.log 2>&1 || { tail -3 $O/single_$tag.log; exit 1; }
  env "$@" timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$tag.json 2> $O/bench_$tag.err || { tail -3 $O/bench_$tag.err; exit 1; }
  echo "$tag: $(grep 'find_peptides(image)' $O/single_$tag.log | cut -c1-60) | bench $(python3 -c "import json;d=json.load(open('$O/bench_$tag.json'));print(round(d['ms_per_step'],2))") ms/step"
}
run default FSQ_NOOP=1
run devkernarg HIP_FORCE_DEV_KERNARG=1
run nointerrupt HSA_ENABLE_INTERRUPT=0
run both HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0
