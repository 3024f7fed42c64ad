#!/bin/bash
# find_peptides_batch: lanes x hardware queues (GPU_MAX_HW_QUEUES: HIP maps streams onto that many hardware queues, 4 by default;
# streams that share one run their kernels one after the other), fields/s of a warm 1024-field call
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_lanes2; mkdir -p $O
for hq in 4 8 16; do
  for lanes in 3 4 5 6; do
    GPU_MAX_HW_QUEUES=$hq FSQ_BATCH_LANES=$lanes timeout -k 10 120 python3 tools/batch_timeline.py 1024 > $O/q${hq}_l${lanes}.log 2>&1 || { tail -5 $O/q${hq}_l${lanes}.log; exit 1; }
    echo "hw queues $hq lanes $lanes: $(grep 'fields in' $O/q${hq}_l${lanes}.log) $(grep "gpu call" $O/q${hq}_l${lanes}.log | tail -1)"
  done
done
