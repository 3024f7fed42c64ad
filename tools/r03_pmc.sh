#!/bin/bash
# SQ / cache counters of the two LM-fit kernels on a 256-field one-queue bench run, one rocprofv3 --pmc pass per counter
# group: tools/r03_pmc.sh <tag> [lib.so]   ->  gpurun_out/<tag>/pmc_summary.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
T=${1:-r03_pmc}
[ -n "$2" ] && export FSQ_HIP_LIB=$PWD/$2
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 0 --fields 256 --queues 1"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_IFETCH" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$i -- $B > $O/pmc_$i.log 2>&1 || { echo "group $i ($grp) failed"; tail -3 $O/pmc_$i.log; }
done
python3 - "$O" <<'PY'
import collections, csv, glob, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "kA" if "kA_jacobian<true" in n else "kAslow" if "kA_jacobian<false" in n else "kB" if "kB_step<true" in n else "kBres" if "kB_step<false" in n else None
        if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(O + "/pmc_summary.txt", "w") as out:
    names = sorted({c for k in acc for c in acc[k]})
    hdr = "%-32s" % "counter" + "".join("%14s" % k for k in ("kA", "kAslow", "kB", "kBres"))
    print(hdr); out.write(hdr + "\n")
    for c in names:
        line = "%-32s" % c + "".join("%14.4g" % acc[k].get(c, 0) for k in ("kA", "kAslow", "kB", "kBres"))
        print(line); out.write(line + "\n")
    for k in ("kA", "kB", "kBres"):
        w = acc[k].get("SQ_WAVES", 0)
        if w:
            line = "%s per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f  wave_cycles(x4?) %.0f busy %.3g" % (
                k, acc[k]["SQ_INSTS_VALU"] / w, acc[k]["SQ_INSTS_SALU"] / w, acc[k]["SQ_INSTS_LDS"] / w, acc[k]["SQ_INSTS_VMEM_RD"] / w,
                acc[k]["SQ_INSTS_VMEM_WR"] / w, acc[k]["SQ_WAVE_CYCLES"] / w, acc[k]["SQ_BUSY_CYCLES"])
            print(line); out.write(line + "\n")
PY
