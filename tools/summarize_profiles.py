#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/collect_profiles.sh into gpurun_out/<tag>/summary.{md,json} and fit_counters.json
(what bench.py reads from profiles/fit_counters_latest.json; it carries the hash of the kernel sources the counters were taken on)."""
import collections
import csv
import glob
import json
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
base = "gpurun_out/%s" % R
CLOCK_GHZ, N_SIMD = 2.4, 1024
out = {"round": R}
md = ["# rocprofv3 summary (%s)" % R, ""]


def kind(name):
    for k in ("kA_jacobian", "kB_step", "kinit", "kfinish", "k1_response", "k5_consolidate", "k2_", "k6_", "k5_"):
        if k in name:
            return k
    return "other"


for tag, title, steps in (("stats", "python3 bench.py --steps 6 --warmup 1 (the default: three fit queues side by side; 1 + 1 + 6 = 8 steps run). "
                           "Kernels of the queues run CONCURRENTLY: their durations overlap and add up to more than wall time", 8),
                          ("stats_q1", "python3 bench.py --steps 6 --warmup 1 --queues 1 (one fit queue: kernels run one after the other, "
                           "durations add up to busy time)", 8),
                          ("stats_cfg3", "python3 bench.py --config 3 --steps 10 --warmup 2 (1 + 2 + 10 = 13 registration calls)", 13),
                          ("stats_f32", "python3 tools/bench_f32.py 1024 4 (the opt-in single-precision solver, FSQ_MODE_TEXTBOOK_F32: kfit_f32 is "
                           "launched once per batch of candidates - 1 024-field batches of 4.36 M fits, then 64- and 32-field batches; kA / kB rows "
                           "are the fp64 textbook runs it is compared with)", 1)):
    f = sorted(glob.glob(base + "/%s/*/*kernel_stats.csv" % tag))
    if not f:
        continue
    rows = list(csv.DictReader(open(f[0])))
    md += ["## %s" % title, "", "| kernel | calls | total ms | avg us | ms per step | % |", "|---|---|---|---|---|---|"]
    ks = []
    for r in rows[:16]:
        tot = float(r["TotalDurationNs"]) / 1e6
        ks.append({"name": r["Name"], "calls": int(r["Calls"]), "total_ms": tot, "avg_us": float(r["AverageNs"]) / 1e3,
                   "ms_per_step": tot / steps, "pct": float(r["Percentage"])})
        md.append("| `%s` | %s | %.2f | %.1f | %.2f | %s |" % (r["Name"][:90], r["Calls"], tot, float(r["AverageNs"]) / 1e3, tot / steps, r["Percentage"]))
    out["kernel_stats_" + tag] = ks
    if tag in ("stats", "stats_q1"):
        fit = sum(float(r["TotalDurationNs"]) for r in rows if kind(r["Name"]) in ("kA_jacobian", "kB_step", "kinit", "kfinish")) / 1e6
        allk = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
        out["fit_kernels_ms_per_step_" + tag] = fit / steps
        out["all_kernels_ms_per_step_" + tag] = allk / steps
        md += ["", "LM fit (kinit + every kA_jacobian / kB_step round + kfinish): %.1f ms of kernel time per step; all kernels %.1f ms per step "
               "(detection and consolidation run on a second stream, concurrently with the rounds)." % (fit / steps, allk / steps), ""]
    elif tag == "stats_f32":
        for r in rows:
            if "kfit_f32" in r["Name"]:
                out["kfit_f32_max_ms"] = float(r["MaxNs"]) / 1e6
                md += ["", "kfit_f32: longest launch %.2f ms (a 1 024-field batch of 4.36 M fits), %s launches in all." % (float(r["MaxNs"]) / 1e6, r["Calls"]), ""]
    else:
        tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
        out["registration_kernels_ms_per_call"] = tot / steps
        md += ["", "fsq_phase_correlate: %.2f ms of kernel time per call of 224 pairs." % (tot / steps), ""]

acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
steps_pmc = None
for d in sorted(glob.glob(base + "/pmc_*")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f:
        continue
    seen = set()
    ninit = 0
    for r in csv.DictReader(open(f[0])):
        k = kind(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), d)
        if d.endswith("pmc_sq1") and key not in seen and "Start_Timestamp" in r:
            seen.add(key)
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if r["Counter_Name"] in ("FETCH_SIZE", "SQ_WAVES") and "kinit" in r["Kernel_Name"]:
            ninit += 1
    if ninit:
        steps_pmc = ninit
out["pmc_steps_256_fields"] = steps_pmc
if steps_pmc and "FETCH_SIZE" in acc["kA_jacobian"] and "WRITE_SIZE" in acc["kA_jacobian"]:
    # MI355X_MICROARCH.md: FETCH_SIZE (KiB) under-reports wide coalesced reads by 2x on gfx950; WRITE_SIZE (KiB) is exact
    fit_bytes = sum((2 * acc[k]["FETCH_SIZE"] + acc[k]["WRITE_SIZE"]) * 1024 for k in ("kA_jacobian", "kB_step", "kinit", "kfinish"))
    per_step = fit_bytes / steps_pmc * 4           # 256 -> 1024 fields
    out["fit_kernel_hbm_bytes_per_1024_field_step"] = per_step
    md += ["## HBM traffic of the LM fit", "",
           "kinit + kA_jacobian + kB_step + kfinish, all rounds, per 1 024-field step (256-field run scaled x4, FETCH_SIZE doubled "
           "per MI355X_MICROARCH.md): %.1f GB" % (per_step / 1e9), ""]
md += ["## SQ counters of the fit kernels (256-field run, %s steps)" % steps_pmc, "", "| counter | kA_jacobian | kB_step |", "|---|---|---|"]
names = sorted(set(acc["kA_jacobian"]) | set(acc["kB_step"]))
for c in names:
    md.append("| %s | %.4g | %.4g |" % (c, acc["kA_jacobian"].get(c, 0), acc["kB_step"].get(c, 0)))
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluorosequencingimageanalysis_amd import _native as _N  # noqa: E402
fc = {"source": "profiles/%s_summary.json (rocprofv3 --pmc, tools/collect_profiles.sh)" % R, "source_sha16": _N.source_sha16(),
      "fields_per_step": 256}
if "fit_kernel_hbm_bytes_per_1024_field_step" in out:
    fc["fit_kernel_hbm_bytes_per_1024_field_step"] = out["fit_kernel_hbm_bytes_per_1024_field_step"]
va = sum(acc[k].get("SQ_ACTIVE_INST_VALU", 0) for k in ("kA_jacobian", "kB_step"))
wc = sum(acc[k].get("SQ_WAVE_CYCLES", 0) for k in ("kA_jacobian", "kB_step"))
iv = sum(acc[k].get("SQ_INSTS_VALU", 0) for k in ("kA_jacobian", "kB_step"))
if va and dur["kA_jacobian"] + dur["kB_step"] > 0:
    # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md); the denominator is the cycles every SIMD of the chip had
    # while those kernels ran (dispatch durations of the same pass)
    cyc = (dur["kA_jacobian"] + dur["kB_step"]) * CLOCK_GHZ * N_SIMD
    fc["valu_issue_frac"] = 4.0 * va / cyc
    for k in ("kA_jacobian", "kB_step"):
        if dur[k] > 0:
            fc["valu_issue_frac_" + k] = 4.0 * acc[k].get("SQ_ACTIVE_INST_VALU", 0) / (dur[k] * CLOCK_GHZ * N_SIMD)
    md += ["", "VALU issue: SQ_ACTIVE_INST_VALU x 4 cycles / (kernel time x 2.4 GHz x 1 024 SIMDs) = %.3f (kA %.3f, kB %.3f)"
           % (fc["valu_issue_frac"], fc.get("valu_issue_frac_kA_jacobian", 0), fc.get("valu_issue_frac_kB_step", 0))]
if va and wc:
    fc["valu_active_per_wave_cycle"] = va / wc
if iv and steps_pmc:
    # wave-level VALU instructions per fit: a kA wave holds 16 fits, a kB wave 64; per-fit count over all its rounds
    fits = None
    try:
        log = open(base + "/bench_sq1.log").read()
        line = [l for l in log.splitlines() if l.startswith("{")][-1]
        fits = json.loads(line)["config"]["candidates_per_gpu"]
    except Exception:       # noqa: BLE001
        pass
    if fits:
        fc["executed_valu_per_fit"] = 64.0 * iv / (fits * steps_pmc)     # thread-level instructions per fit
        md += ["", "Executed VALU instructions: %.3g wave-instructions over %d steps x %d fits = %.0f lane-instructions per fit "
               "(algorithmic: 2.0e5 flop per fit)" % (iv, steps_pmc, fits, fc["executed_valu_per_fit"])]
out["fit_counters"] = fc
json.dump(out, open(base + "/summary.json", "w"), indent=1)
json.dump(fc, open(base + "/fit_counters.json", "w"), indent=1)
open(base + "/summary.md", "w").write("\n".join(md) + "\n")
print("\n".join(md))
