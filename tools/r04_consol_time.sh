#!/bin/bash
# per-call durations of k5_consolidate in tools/bench_f32.py (1 024 x 512^2, 64 x 512^2, 32 x 2 048^2 fields) for the given library variants
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = prod ]; then unset FSQ_HIP_LIB; else export FSQ_HIP_LIB=$PWD/fluorosequencingimageanalysis_amd/csrc/variants/libfsq_$v.so; fi
  O=gpurun_out/r04_ctime_$v; rm -rf $O; mkdir -p $O
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/stats -- python3 tools/bench_f32.py 1024 2 > $O/bench_f32.log 2>&1 || { tail -5 $O/bench_f32.log; exit 1; }
  python3 - "$O" "$v" <<'PY'
import csv, glob, sys, collections
O, v = sys.argv[1], sys.argv[2]
f = glob.glob(O + "/stats/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k5_consolidate" in r["Kernel_Name"]:
        d[(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(v, {k: "%.0f us (min of %d)" % (min(x), len(x)) for k, x in d.items()})
PY
done
