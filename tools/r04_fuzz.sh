#!/bin/bash
# round 4: the long checks on the final kernels - the whole GPU suite on dirty device memory, the field / tracker fuzz for many
# seeds, the adversarial stand-alone ROIs, the batch surface and the detection / registration fuzz (progress lines keep the run alive)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_fuzz; rm -rf $O; mkdir -p $O gpurun_out/fuzz
FSQ_TEST_DIRTY_ALLOC=1 timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > $O/pytest_dirty.log 2>&1 || { tail -30 $O/pytest_dirty.log; exit 1; }
echo "dirty-memory suite: $(tail -1 $O/pytest_dirty.log)"
for seed in $(seq 500 ${1:-524}); do
  timeout -k 10 500 python3 tools/fuzz_r02.py $seed > gpurun_out/fuzz/fuzz_$seed.log 2>&1 || { tail -20 gpurun_out/fuzz/fuzz_$seed.log; exit 1; }
  echo "seed $seed: $(grep -c identical gpurun_out/fuzz/fuzz_$seed.log) checks identical"
done
timeout -k 10 900 python3 tools/fuzz_rois.py 77 40000 > $O/rois.log 2>&1 || { tail -20 $O/rois.log; exit 1; }
tail -2 $O/rois.log
FSQ_DEBUG_FORCE_SLOW=2 timeout -k 10 900 python3 tools/fuzz_rois.py 78 20000 > $O/rois_slow.log 2>&1 || { tail -20 $O/rois_slow.log; exit 1; }
tail -1 $O/rois_slow.log
timeout -k 10 900 python3 tools/fuzz_batch.py 9 60 > $O/batch.log 2>&1 || { tail -20 $O/batch.log; exit 1; }
tail -1 $O/batch.log
timeout -k 10 600 python3 tools/fuzz_register.py 5 100 > $O/register.log 2>&1 || { tail -20 $O/register.log; exit 1; }
tail -1 $O/register.log
