#!/bin/bash
# fit parity tests + one bench line (after a kernel change)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/quick; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_pipeline.py tests/test_gpu_stream.py -q -m gpu -x > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench: $(python3 -c "import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])")"
