#!/bin/bash
# run selected GPU tests: tools/one_test.sh <pytest args>
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/one_test
timeout -k 10 900 python3 -m pytest "$@" -q -m gpu -x > gpurun_out/one_test/pytest.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/one_test/pytest.log
