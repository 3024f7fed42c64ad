#!/bin/bash
# A/B of library variants by one-queue kernel stats: tools/r03_ab.sh <variant-name> ...   ("prod" = the production library)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = prod ]; then unset FSQ_HIP_LIB; else export FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/variants/libfsq_$v.so; fi
  O=gpurun_out/ab_$v; rm -rf $O; mkdir -p $O
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 1 --queues 1 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
  python3 - "$O" "$v" <<'PY'
import csv, glob, sys, json
O, v = sys.argv[1], sys.argv[2]
f = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = {r["Name"][:48]: float(r["TotalDurationNs"]) / 1e6 / 8 for r in csv.DictReader(open(f))}
ka = sum(t for n, t in rows.items() if "kA_jacobian" in n); kb = sum(t for n, t in rows.items() if "kB_step" in n)
d = json.loads([l for l in open(O + "/bench.log").read().splitlines() if l.startswith("{")][-1])
print("%-12s kA %.1f ms/step  kB %.1f ms/step  (q1 bench %.4g fits/s, %.1f ms/step)" % (v, ka, kb, d["value"], d["ms_per_step"]))
PY
done
