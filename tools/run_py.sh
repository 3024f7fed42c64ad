#!/bin/bash
# run a tool script on the GPU box: tools/run_py.sh tools/bench_tracking.py [args]
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/run_py
timeout -k 10 600 python3 "$@" > gpurun_out/run_py/out.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/run_py/out.log
