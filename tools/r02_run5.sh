#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run5; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_tracking.py -q -m gpu -x > $O/pytest_track.log 2>&1; echo "tracking tests rc=$?"; tail -15 $O/pytest_track.log
