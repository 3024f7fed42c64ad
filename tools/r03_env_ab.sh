#!/bin/bash
# one-queue kernel stats under different environment settings: tools/r03_env_ab.sh "NAME=VAL ..." "NAME2=VAL2" ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for e in "$@"; do
  i=$((i+1)); O=gpurun_out/envab_$i; rm -rf $O; mkdir -p $O
  ( [ "$e" != "-" ] && export $e; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 1 --queues 1 > $O/bench.log 2>&1 ) || { tail -5 $O/bench.log; exit 1; }
  python3 - "$O" "$e" <<'PY'
import csv, glob, sys, os, json
O, v = sys.argv[1:3]
f = max(glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
ka = sum(float(r["TotalDurationNs"]) for r in rows if "kA_jacobian" in r["Name"]) / 8e6; kb = sum(float(r["TotalDurationNs"]) for r in rows if "kB_step" in r["Name"]) / 8e6
calls = [r["Calls"] for r in rows if "kB_step" in r["Name"]]
d = json.loads([l for l in open(O + "/bench.log").read().splitlines() if l.startswith("{")][-1])
print("%-40s kA %.1f  kB %.1f ms/step  kB calls %s (q1 bench %.4g fits/s, %.1f ms/step)" % (v, ka, kb, calls, d["value"], d["ms_per_step"]))
PY
done
