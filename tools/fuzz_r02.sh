#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fuzz
for seed in 2026 7 99; do timeout -k 10 500 python3 tools/fuzz_r02.py $seed > gpurun_out/fuzz/fuzz_$seed.log 2>&1; echo "seed $seed rc=$?"; tail -4 gpurun_out/fuzz/fuzz_$seed.log; done
