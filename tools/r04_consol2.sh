#!/bin/bash
# round 4: the component-based consolidation - adversarial test, pipeline / config tests, field fuzz, timing of both forms
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_consol2; rm -rf $O; mkdir -p $O gpurun_out/fuzz
timeout -k 10 900 python3 -m pytest tests/test_gpu_consolidate.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_stream.py -q -m gpu -x > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for seed in 3 31 401 402 403 404; do
  timeout -k 10 500 python3 tools/fuzz_r02.py $seed > gpurun_out/fuzz/fuzz_$seed.log 2>&1 || { tail -20 gpurun_out/fuzz/fuzz_$seed.log; exit 1; }
  echo "seed $seed: $(grep -c identical gpurun_out/fuzz/fuzz_$seed.log) checks identical"
done
for v in components blocks; do
  [ $v = blocks ] && export FSQ_CONSOLIDATE_BLOCKS=1
  P=$O/stats_$v; mkdir -p $P
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $P -- python3 tools/bench_f32.py 1024 2 > $O/bench_f32_$v.log 2>&1 || { tail -5 $O/bench_f32_$v.log; exit 1; }
  python3 - "$P" "$v" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f)) if "k5" in r["Kernel_Name"] and "total" not in r["Kernel_Name"] and "kept" not in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
# group consecutive k5 kernels of one fsq_consolidate call: a call ends with k5c_finish / k5_consolidate
calls, cur = [], []
for r in rows:
    cur.append(r)
    if "finish" in r["Kernel_Name"] or "k5_consolidate" in r["Kernel_Name"]:
        calls.append(cur); cur = []
d = collections.defaultdict(list)
for c in calls:
    fields = int(c[-1]["Grid_Size_X"]) // int(c[-1]["Workgroup_Size_X"])
    d[fields].append((sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in c) / 1e3, {r["Kernel_Name"].split("::")[-1].split("(")[0]: round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in c}))
for k, v in d.items():
    best = min(v, key=lambda t: t[0])
    print(sys.argv[2], "fields", k, "kernel time %.0f us" % best[0], best[1])
PY
done
