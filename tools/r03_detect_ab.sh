#!/bin/bash
# detection kernel times (one-queue kernel stats) for library variants: tools/r03_detect_ab.sh prod nomed
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = prod ]; then unset FSQ_HIP_LIB; else export FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/variants/libfsq_$v.so; fi
  O=gpurun_out/det_$v; rm -rf $O; mkdir -p $O
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --queues 1 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
  python3 - "$O" "$v" <<'PY'
import csv, glob, sys, os
O, v = sys.argv[1:3]
f = max(glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k1_", "k2_")):
        print("%-8s %-50s avg %8.1f us" % (v, r["Name"][:50], float(r["AverageNs"]) / 1e3))
PY
done
