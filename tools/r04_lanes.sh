#!/bin/bash
# find_peptides_batch: lanes x ramp sweep (FSQ_BATCH_LANES, FSQ_BATCH_RAMP), fields/s of a warm 1024-field call
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_lanes; mkdir -p $O
for lanes in 3 4 5; do
  for ramp in 2 4 4,2 8,4,2; do
    FSQ_BATCH_LANES=$lanes FSQ_BATCH_RAMP=$ramp timeout -k 10 120 python3 tools/batch_timeline.py 1024 > $O/l${lanes}_r${ramp}.log 2>&1 || { tail -5 $O/l${lanes}_r${ramp}.log; exit 1; }
    echo "lanes $lanes ramp $ramp: $(grep 'fields in' $O/l${lanes}_r${ramp}.log) $(grep "gpu call" $O/l${lanes}_r${ramp}.log | tail -1)"
  done
done
