#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/collect_profiles.sh into gpurun_out/<round>/summary.{md,json}."""
import collections
import csv
import glob
import json
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
base = "gpurun_out/%s" % R
out = {"round": R}
md = ["# rocprofv3 summary (%s): python3 bench.py --steps 3 --warmup 1" % R, ""]
f = glob.glob(base + "/stats/*/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    md += ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    ks = []
    for r in rows[:14]:
        ks.append({"name": r["Name"], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                   "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])})
        md.append("| `%s` | %s | %.2f | %.1f | %s |" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, r["Percentage"]))
    out["kernel_stats"] = ks
    # the "kernel" bench.py prices in its roofline is the whole LM fit: kinit + rounds of kA / kB + kfinish
    allk = [{"name": r["Name"], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6} for r in rows]
    fitk = [k for k in allk if any(t in k["name"] for t in ("kinit", "kA_jacobian", "kB_step", "kfinish"))]
    launches = sum(k["calls"] for k in allk if "kinit" in k["name"])
    if launches:
        per = sum(k["total_ms"] for k in fitk) / launches
        out["fit_kernels_ms_per_launch"] = per
        out["fit_launches"] = launches
        md += ["", "LM fit as one unit (kinit + every kA_jacobian / kB_step round + kfinish): %.1f ms of kernel time per launch over "
               "%d launches - bench.py's `roofline.launch_ms` (HIP events around the same launches) adds the host gaps "
               "between rounds and, with two lanes, the time a lane's kernels wait for the other lane's." % (per, launches)]
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(base + "/%s/*/*counter_collection.csv" % tag)
    if not f:
        continue
    acc = collections.defaultdict(float)
    passes = 0
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == ctr:
            name = r["Kernel_Name"]
            passes += ("kinit" in name)          # one kinit per LM-fit launch = per pass over the fields
            key = "kA_jacobian" if "kA_jacobian" in name else "kB_step" if "kB_step" in name else \
                  "k1_response" if "k1_response" in name else "other"
            acc[key] += float(r["Counter_Value"])
    out[ctr + "_KiB_256_fields_1_step"] = {k: v / max(passes, 1) for k, v in acc.items()}
    out[ctr + "_passes_in_run"] = passes
if "FETCH_SIZE_KiB_256_fields_1_step" in out and "WRITE_SIZE_KiB_256_fields_1_step" in out:
    fe, wr = out["FETCH_SIZE_KiB_256_fields_1_step"], out["WRITE_SIZE_KiB_256_fields_1_step"]
    # MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; WRITE_SIZE is exact. Units KiB.
    fit_bytes = (2 * (fe.get("kA_jacobian", 0) + fe.get("kB_step", 0)) + wr.get("kA_jacobian", 0) + wr.get("kB_step", 0)) * 1024
    out["fit_kernel_hbm_bytes_per_1024_field_pass"] = fit_bytes * 4       # scaled from 256 to the 1024 fields of one bench step
    md += ["", "HBM traffic of the LM fit (kA_jacobian + kB_step, all rounds of one step, 256-field run scaled x4 to the "
           "1024-field step; FETCH_SIZE doubled per MI355X_MICROARCH.md): %.1f GB" % (out["fit_kernel_hbm_bytes_per_1024_field_pass"] / 1e9)]
json.dump(out, open(base + "/summary.json", "w"), indent=1)
open(base + "/summary.md", "w").write("\n".join(md) + "\n")
print("\n".join(md))
