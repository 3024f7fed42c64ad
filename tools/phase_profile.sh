#!/bin/bash
# per-phase wall-cycle split of kA / kB (needs csrc/prof/libfsq_hip_prof.so built with -DFSQ_PHASE_PROFILE)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/phase
FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/prof/libfsq_hip_prof.so timeout -k 10 300 python3 tools/time_fit.py 256 0 > gpurun_out/phase/phase.log 2>&1; tail -20 gpurun_out/phase/phase.log
