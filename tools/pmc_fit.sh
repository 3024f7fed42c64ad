#!/bin/bash
# Run on the GPU box: SQ counters of the fit kernels (one rocprofv3 --pmc pass per counter group) -> gpurun_out/pmc_fit/
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_fit
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python3 tools/time_fit.py ${1:-32} 0 > $O/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_fit/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "kA" if "kA_jacobian" in n else "kB" if "kB_step" in n else None
        if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open("gpurun_out/pmc_fit/summary.txt", "w") as o:
    for k in acc:
        for c in sorted(acc[k]):
            line = "%s %-28s %.4g" % (k, c, acc[k][c]); print(line); o.write(line + "\n")
PY
