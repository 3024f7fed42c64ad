#!/bin/bash
# command line, 256 and 1024 TIFFs, 16 workers, three runs in ONE process each (the first includes `import torch` and the first use of
# the GPU, the later ones start their workers cold in a warm process - what bench.py's extras measure)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_cli3; mkdir -p $O
timeout -k 10 600 python3 tools/bench_cli.py 256 16 16 16 > $O/cli256.log 2>&1 || { tail -5 $O/cli256.log; exit 1; }
grep "io workers" $O/cli256.log
timeout -k 10 600 python3 tools/bench_cli.py 1024 16 16 > $O/cli1024.log 2>&1 || { tail -5 $O/cli1024.log; exit 1; }
grep "io workers" $O/cli1024.log
