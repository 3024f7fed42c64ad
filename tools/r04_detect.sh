#!/bin/bash
# round 4: the shared-column-sort detection kernel - candidate tests, the detection fuzz, and k1's time next to round 3's kernel (FSQ_DETECT_R03=1)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_detect; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_bench_fields.py -q -m gpu -x > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for seed in 1 2 3; do timeout -k 10 600 python3 tools/fuzz_detect.py $seed 300 > $O/fuzz_$seed.log 2>&1 || { tail -20 $O/fuzz_$seed.log; exit 1; }; tail -1 $O/fuzz_$seed.log; done
for v in new r03; do
  [ $v = r03 ] && export FSQ_DETECT_R03=1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --queues 1 > $O/bench_$v.log 2>&1 || { tail -5 $O/bench_$v.log; exit 1; }
  python3 - "$O/stats_$v" "$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k1_response" in r["Name"]:
        print(sys.argv[2], r["Name"][:50], "calls", r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
