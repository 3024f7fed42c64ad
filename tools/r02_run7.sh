#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run7; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_pipeline.py tests/test_gpu_stream.py -q -m gpu -x > $O/pytest_fit.log 2>&1; echo "fit tests rc=$?"; tail -5 $O/pytest_fit.log
for i in 1 2; do
timeout -k 10 300 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$i.json 2> $O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
echo "bench $i: $(python3 -c "import json;d=json.load(open('$O/bench_$i.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])")"
done
FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/prof/libfsq_hip_prof.so timeout -k 10 300 python3 tools/time_fit.py 256 0 > $O/phase.log 2>&1; tail -22 $O/phase.log
