#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run4; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1; echo "all gpu tests rc=$?"; tail -12 $O/pytest_all.log
