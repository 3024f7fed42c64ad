#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_ka8; mkdir -p $O
FSQ_KA_LANES=8 timeout -k 10 900 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_pipeline.py tests/test_gpu_stream.py -q -m gpu -x > $O/pytest_fit.log 2>&1; echo "fit tests (8 lanes) rc=$?"; tail -5 $O/pytest_fit.log
for L in 4 8; do
FSQ_KA_LANES=$L timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$L.json 2> $O/bench_$L.err || { tail -5 $O/bench_$L.err; exit 1; }
echo "lanes $L: $(python3 -c "import json;d=json.load(open('$O/bench_$L.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])")"
done
