#!/bin/bash
# r02: new stream-pipeline tests, the whole GPU suite, bench in stream mode at a few injection thresholds
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run2; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_stream.py tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $O/pytest_new.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "all gpu tests rc=$?"; tail -3 $O/pytest_all.log
for IB in 500000 1000000 2000000; do
  timeout -k 10 300 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras --inject-below $IB > $O/bench_ib$IB.json 2> $O/bench_ib$IB.err || { echo "bench ib=$IB failed"; tail -5 $O/bench_ib$IB.err; exit 1; }
  echo "inject_below $IB: $(python3 -c "import json;d=json.load(open('$O/bench_ib$IB.json'));print(d['value'],d['ms_per_step'],d['roofline']['launches'], d['roofline']['frac'])")"
done
