#!/bin/bash
# kA / kB time per step at two waves per SIMD (default) and at one (LDS pad), for library variants: tools/r03_occ.sh prod head ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = prod ]; then unset FSQ_HIP_LIB; else export FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/variants/libfsq_$v.so; fi
  for pad in 0 24000; do
    O=gpurun_out/occ_${v}_$pad; rm -rf $O; mkdir -p $O
    FSQ_DEBUG_KA_LDS_PAD=$pad FSQ_DEBUG_KB_LDS_PAD=$pad timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --queues 1 --fields 512 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
    python3 - "$O" "$v" "$pad" <<'PY'
import csv, glob, sys, os
O, v, pad = sys.argv[1:4]
f = max(glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = {r["Name"][:48]: float(r["TotalDurationNs"]) / 1e6 / 6 for r in csv.DictReader(open(f))}
ka = sum(t for n, t in rows.items() if "kA_jacobian" in n); kb = sum(t for n, t in rows.items() if "kB_step" in n)
print("%-10s pad %-6s kA %.1f ms/step  kB %.1f ms/step (512 fields)" % (v, pad, ka, kb))
PY
  done
done
