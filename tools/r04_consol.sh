#!/bin/bash
# round 4: the multi-wave consolidation - pipeline / config / stream tests, the field fuzz for a few seeds, and its timing
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_consol; rm -rf $O; mkdir -p $O gpurun_out/fuzz
timeout -k 10 900 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_fit.py -q -m gpu -x > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for seed in 3 31 401 402; do
  timeout -k 10 500 python3 tools/fuzz_r02.py $seed > gpurun_out/fuzz/fuzz_$seed.log 2>&1 || { tail -20 gpurun_out/fuzz/fuzz_$seed.log; exit 1; }
  echo "seed $seed: $(tail -1 gpurun_out/fuzz/fuzz_$seed.log)"
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/bench_f32.py 1024 4 > $O/bench_f32.log 2>&1 || { tail -5 $O/bench_f32.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k5_" in r["Name"] or "k1_" in r["Name"]:
        print("%-40s calls %5s avg %9.1f us  max %9.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
grep '^{' $O/bench_f32.log | head -3 | cut -c1-300
