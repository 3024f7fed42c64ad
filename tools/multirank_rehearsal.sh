#!/bin/bash
# rehearsal of the N > 1 bench path on the one-GPU box: 2 ranks sharing the GPU, gloo for the exchange
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/multirank; mkdir -p $O
FSQ_DIST_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 1 --fields 256 --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err; echo "rc=$?"; tail -c 1500 $O/bench2.json; tail -5 $O/bench2.err
