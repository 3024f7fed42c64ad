#!/bin/bash
# round 4, after the 32-bit pixel instantiations went into the fit kernels: the wide-pixel fuzz, then the long checks of
# tools/r04_fuzz.sh on fresh seeds (progress lines keep the run alive; a failed or killed step stops the run)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_final; rm -rf $O; mkdir -p $O
for seed in 2031 2032 2033; do
  timeout -k 10 900 python3 tools/fuzz_wide.py $seed 10 40000 > $O/wide_$seed.log 2>&1 || { tail -20 $O/wide_$seed.log; exit 1; }
  grep "^fields:\|FUZZ WIDE OK" $O/wide_$seed.log
done
for seed in $(seq 600 ${1:-611}); do
  timeout -k 10 500 python3 tools/fuzz_r02.py $seed > $O/fuzz_$seed.log 2>&1 || { tail -20 $O/fuzz_$seed.log; exit 1; }
  echo "seed $seed: $(grep -c identical $O/fuzz_$seed.log) checks identical"
done
timeout -k 10 900 python3 tools/fuzz_rois.py 79 40000 > $O/rois.log 2>&1 || { tail -20 $O/rois.log; exit 1; }
tail -2 $O/rois.log
FSQ_DEBUG_FORCE_SLOW=2 timeout -k 10 900 python3 tools/fuzz_rois.py 80 20000 > $O/rois_slow.log 2>&1 || { tail -20 $O/rois_slow.log; exit 1; }
tail -1 $O/rois_slow.log
timeout -k 10 900 python3 tools/fuzz_batch.py 10 40 > $O/batch.log 2>&1 || { tail -20 $O/batch.log; exit 1; }
tail -1 $O/batch.log
