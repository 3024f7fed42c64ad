#!/bin/bash
# after a change of the fit kernels: adversarial ROIs (plain, forced slow path, forced norm recomputation) + a few field-fuzz seeds
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_fuzz_fit; rm -rf $O; mkdir -p $O gpurun_out/fuzz
timeout -k 10 900 python3 tools/fuzz_rois.py 91 30000 > $O/rois.log 2>&1 || { tail -20 $O/rois.log; exit 1; }
tail -1 $O/rois.log
FSQ_DEBUG_FORCE_SLOW=3 timeout -k 10 900 python3 tools/fuzz_rois.py 92 15000 > $O/rois_slow.log 2>&1 || { tail -20 $O/rois_slow.log; exit 1; }
tail -1 $O/rois_slow.log
FSQ_DEBUG_FORCE_NORM_RECOMPUTE=1 timeout -k 10 900 python3 tools/fuzz_rois.py 93 15000 > $O/rois_redo.log 2>&1 || { tail -20 $O/rois_redo.log; exit 1; }
tail -1 $O/rois_redo.log
for seed in 601 602 603 604 605 606; do
  timeout -k 10 500 python3 tools/fuzz_r02.py $seed > gpurun_out/fuzz/fuzz_$seed.log 2>&1 || { tail -20 gpurun_out/fuzz/fuzz_$seed.log; exit 1; }
  echo "seed $seed: $(grep -c identical gpurun_out/fuzz/fuzz_$seed.log) checks identical"
done
