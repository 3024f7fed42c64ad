#!/bin/bash
# one-queue rocprofv3 kernel stats of the bench (kernel durations add up) + one default bench line.
# usage: tools/r03_kstats.sh <tag> [lib.so]      results under gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
T=${1:-r03_kstats}
[ -n "$2" ] && export FSQ_HIP_LIB=$PWD/$2
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-extras"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_q1 -- $B --steps 6 --warmup 1 --queues 1 > $O/bench_q1.log 2>&1 || { tail -5 $O/bench_q1.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/stats_q1/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(O + "/kernel_stats_q1.txt", "w") as out:
    for r in rows[:14]:
        line = "%-60s calls %7s  total %10.2f ms  avg %9.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot)
        print(line); out.write(line + "\n")
import shutil; shutil.copy(f, O + "/kernel_stats_q1.csv")
PY
grep '^{' $O/bench_q1.log | tail -1 | python3 -c "import json,sys;d=json.loads(sys.stdin.readline());print('q1 bench', d['value'], d['ms_per_step'])"
timeout -k 10 300 $B --steps 12 --warmup 2 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench.json'));print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
