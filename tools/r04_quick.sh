#!/bin/bash
# round 4: fit / pipeline / stream / bench-field parity tests + one-queue kernel stats + one bench line (after a kernel change)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_quick; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_stream.py tests/test_gpu_pipeline.py tests/test_gpu_bench_fields.py -q -m gpu -x > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 1 --queues 1 > $O/bench_q1.log 2>&1 || { tail -5 $O/bench_q1.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys, json
O = sys.argv[1]
f = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = {r["Name"][:48]: float(r["TotalDurationNs"]) / 1e6 / 8 for r in csv.DictReader(open(f))}
ka = sum(t for n, t in rows.items() if "kA_jacobian" in n); kb = sum(t for n, t in rows.items() if "kB_step" in n)
d = json.loads([l for l in open(O + "/bench_q1.log").read().splitlines() if l.startswith("{")][-1])
print("kA %.1f ms/step  kB %.1f ms/step  (q1 bench %.4g fits/s, %.1f ms/step)" % (ka, kb, d["value"], d["ms_per_step"]))
PY
timeout -k 10 400 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench.json'));print('bench:', d['value'],d['ms_per_step'],d['roofline']['frac'])"
