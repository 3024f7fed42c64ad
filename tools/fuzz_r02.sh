#!/bin/bash
# the one-off fuzz (tools/fuzz_r02.py) for the seeds given (default: round 3's): tools/fuzz_r02.sh [seed ...]
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fuzz
for seed in ${@:-3 31 314}; do
  timeout -k 10 500 python3 tools/fuzz_r02.py $seed > gpurun_out/fuzz/fuzz_$seed.log 2>&1 || { tail -20 gpurun_out/fuzz/fuzz_$seed.log; exit 1; }
  echo "seed $seed"; tail -4 gpurun_out/fuzz/fuzz_$seed.log
done
