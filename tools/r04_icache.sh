#!/bin/bash
# Instruction-cache counters of the LM-fit kernels (kA is 31 KB of code, kB 47 KB; the I-cache is 64 KB per two CUs), with one
# fit queue (the kernels alternate) and with three (kA and kB of different queues run on the same CUs at the same time).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_icache; rm -rf $O; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*" $O/avail.txt | sort -u > $O/names.txt || true
cat $O/names.txt
B="python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 0 --fields 256"
C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"
rocprofv3 --pmc $C --output-format csv -d $O/q1 -- $B --queues 1 > $O/q1.log 2>&1 || { tail -5 $O/q1.log; exit 1; }
rocprofv3 --pmc $C --output-format csv -d $O/q3 -- $B --queues 3 > $O/q3.log 2>&1 || { tail -5 $O/q3.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
for tag in ("q1", "q3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("gpurun_out/r04_icache/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = "kA" if "kA_jacobian<true" in k else "kB" if "kB_step" in k else None
            if k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in sorted(acc.items()):
        req = d.get("SQC_ICACHE_REQ", 0)
        print(tag, k, {n: int(v) for n, v in d.items()}, "miss rate %.4f" % (d.get("SQC_ICACHE_MISSES", 0) / req if req else -1))
PY
