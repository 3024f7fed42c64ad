#!/bin/bash
# acceptance run: whole GPU suite, smoke(), the driver's bench command (steps are chained: a failed or killed step stops the run)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/acceptance; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench.json'));print('bench', d['value'], d['ms_per_step'], d['fields_per_sec'], d['peaks_per_sec'], d['roofline']['frac'], d.get('extras'), d['cpu_baseline']['value'])"
