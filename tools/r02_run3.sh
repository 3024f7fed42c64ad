#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run3; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "all gpu tests rc=$?"; tail -8 $O/pytest_all.log
for IB in 3000000 4000000; do
  timeout -k 10 300 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras --inject-below $IB > $O/bench_ib$IB.json 2> $O/bench_ib$IB.err || { echo "bench ib=$IB failed"; tail -5 $O/bench_ib$IB.err; exit 1; }
  echo "inject_below $IB: $(python3 -c "import json;d=json.load(open('$O/bench_ib$IB.json'));print(d['value'],d['ms_per_step'],d['roofline']['launches'], d['roofline']['frac'])")"
done
for LF in 2 4; do
  FSQ_LMPAR_FIRST_ITERS=$LF timeout -k 10 300 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras --inject-below 2000000 > $O/bench_lf$LF.json 2> $O/bench_lf$LF.err || exit 1
  echo "lmpar_first $LF: $(python3 -c "import json;d=json.load(open('$O/bench_lf$LF.json'));print(d['value'],d['ms_per_step'])")"
done
timeout -k 10 300 python3 bench.py --steps 6 --warmup 1 > $O/bench_full.json 2> $O/bench_full.err || { tail -20 $O/bench_full.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_full.json'));print(d['value'], d.get('extras'), d['cpu_baseline']['value'])"
