#!/bin/bash
# per-phase cycle split of kA / kB (needs csrc/prof/libfsq_hip_prof.so built with -DFSQ_PHASE_PROFILE).
# With FSQ_DEBUG_KA_LDS_PAD / FSQ_DEBUG_KB_LDS_PAD = 24000 only one wave fits a SIMD and the split is each phase's own
# (solo) latency; at the default two waves per SIMD a phase is also charged the time the other wave holds the issue port.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/phase
export FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/prof/libfsq_hip_prof.so
timeout -k 10 300 python3 tools/time_fit.py 256 0 > gpurun_out/phase/phase.log 2>&1 && tail -24 gpurun_out/phase/phase.log && \
FSQ_DEBUG_KA_LDS_PAD=24000 FSQ_DEBUG_KB_LDS_PAD=24000 timeout -k 10 300 python3 tools/time_fit.py 256 0 > gpurun_out/phase/phase_solo.log 2>&1 && tail -24 gpurun_out/phase/phase_solo.log
