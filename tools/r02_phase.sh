#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_phase; mkdir -p $O
FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/prof/libfsq_hip_prof.so timeout -k 10 300 python3 tools/time_fit.py 256 0 > $O/phase.log 2>&1; tail -22 $O/phase.log
