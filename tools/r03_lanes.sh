#!/bin/bash
# parity tests + one-queue kernel stats for FSQ_KA_LANES = 4 and 8
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for L in ${@:-4 8}; do
  export FSQ_KA_LANES=$L
  O=gpurun_out/lanes_$L; rm -rf $O; mkdir -p $O
  timeout -k 10 600 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_stream.py -q -m gpu -x > $O/pytest.log 2>&1; echo "L=$L fit tests rc=$?"; tail -2 $O/pytest.log
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 1 --queues 1 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
  python3 - "$O" "$L" <<'PY'
import csv, glob, sys, os, json
O, v = sys.argv[1:3]
f = max(glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = {r["Name"][:48]: float(r["TotalDurationNs"]) / 1e6 / 8 for r in csv.DictReader(open(f))}
ka = sum(t for n, t in rows.items() if "kA_jacobian" in n); kb = sum(t for n, t in rows.items() if "kB_step" in n)
d = json.loads([l for l in open(O + "/bench.log").read().splitlines() if l.startswith("{")][-1])
print("L=%s kA %.1f ms/step  kB %.1f ms/step  (q1 bench %.4g fits/s, %.1f ms/step)" % (v, ka, kb, d["value"], d["ms_per_step"]))
PY
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 2 > $O/bench2.json 2> $O/bench2.err || { tail -5 $O/bench2.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/bench2.json'));print('L=$L default bench', d['value'], d['ms_per_step'])"
done
