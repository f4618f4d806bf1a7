"""Condenses a tools/profile_round.sh output directory into the tracked summaries under profiles/ (python3
tools/collect_profiles.py gpurun_out/r03 r03): <round>_summary.json (per LOD kernel: rocprof average duration, HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes with
the gfx950 x2 correction on FETCH_SIZE, matrix-pipe busy fraction), the kernel-stats CSVs, the bench lines."""
import csv
import glob
import json
import os
import re
import shutil
import sys

out = sys.argv[1]
RND = sys.argv[2] if len(sys.argv) > 2 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
# substrings of the kernel names that make up a configuration's LOD scan: null-exact = k_scan_lr (rank-R class) + the table
# kernel in permuted-column mode (shared-weights class), each launched once per panel region
KERN = {"exact": ("k_scan_lr", "k_scan<0, 2, 4, true, 2, true"), "grid": ("k_scan<",), "alt": ("k_scan_alt",), "perm32": ("k_scan_f32",)}   # (k_rotate_f32 of the perm32 run: in the kernel-stats CSV)
ARGS = {"exact": "(default)", "grid": "--method null-grid", "alt": "--method alt-grid",
        "perm32": "--method perms --perm-dtype f32 --n 1000 --p 100000 --m 1250"}


def counter_avgs(d, kern):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in kern):
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


summary = {}
for tag, kern in KERN.items():
    st = glob.glob(os.path.join(out, tag, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    shutil.copy(st[0], os.path.join(prof, f"{RND}_kernel_stats_{tag}.csv"))
    rows = list(csv.DictReader(open(st[0])))
    mine = [r for r in rows if any(k in r["Name"] for k in kern)]
    # launches per library call: the null-exact scan runs two kernels per panel region (two regions when the h2 search is
    # split); k_design runs once per call.  Durations and counters below are PER CALL (sums over the launches of a call).
    ncall = next((int(r["Calls"]) for r in rows if "k_design" in r["Name"]), 0)
    nlaunch = sum(int(r["Calls"]) for r in mine)
    per_call = (nlaunch / ncall) if (mine and ncall) else 1.0
    tot_ns = sum(float(r["TotalDurationNs"]) for r in mine)
    s = {"kernel": " + ".join(r["Name"].split("(")[0] for r in mine) if mine else str(kern), "bench_args": ARGS[tag],
         "rocprof_avg_ms": (tot_ns / (ncall or nlaunch)) / 1e6 if mine else None, "calls": nlaunch,
         "launches_per_call": per_call}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"):
        s.update({k: v * per_call for k, v in counter_avgs(os.path.join(out, tag, "pmc_" + c), kern).items()})
    if "FETCH_SIZE" in s and "WRITE_SIZE" in s:   # rocprofv3 reports KB; FETCH_SIZE counts half of wide coalesced reads on gfx950
        s["hbm_fetch_bytes"] = s["FETCH_SIZE"] * 1024 * 2
        s["hbm_write_bytes"] = s["WRITE_SIZE"] * 1024
        s["hbm_bytes_per_launch"] = s["hbm_fetch_bytes"] + s["hbm_write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "GRBM_GUI_ACTIVE" in s:
        s["mfma_busy_frac"] = (s["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (s["GRBM_GUI_ACTIVE"] / 8)   # 1024 SIMDs; GUI_ACTIVE sums 8 XCDs
        if s["rocprof_avg_ms"]:
            s["effective_clock_GHz"] = s["GRBM_GUI_ACTIVE"] / 8 / (s["rocprof_avg_ms"] * 1e-3) / 1e9
    summary[tag] = s
json.dump(summary, open(os.path.join(prof, RND + "_summary.json"), "w"), indent=1)
for tag in ("n500_shard", "n1000_perm_shard"):        # trace-only runs: the kernel statistics as they are
    st = glob.glob(os.path.join(out, tag, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(prof, f"{RND}_kernel_stats_{tag}.csv"))
if os.path.exists(os.path.join(out, "timeline_bxd_step.txt")):
    shutil.copy(os.path.join(out, "timeline_bxd_step.txt"), os.path.join(prof, RND + "_timeline_bxd_step.txt"))
for name in ("bench.json", "bench_reduced.json", "configs.jsonl", "mb3_f64.log", "mb_f64.log", "mb4_rcp.log", "mb_lod.log", "power_null_exact.log",
             "power_null_exact_reduced.log"):
    src = os.path.join(out, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(prof, RND + "_" + name))
if "exact" in summary and summary["exact"].get("hbm_bytes_per_launch"):
    e = summary["exact"]
    json.dump({"method": "null-exact", "n": 79, "m": 35554, "p": 7321, "round": RND, "hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
               "fetch_bytes": e["hbm_fetch_bytes"], "write_bytes": e["hbm_write_bytes"], "rocprof_kernel_avg_ms": e["rocprof_avg_ms"],
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) and --kernel-trace --stats, per call = k_scan_lr<1,2,4> + "
                         "k_scan<0,2,4,table,perm> over both panel regions, "
                         "profiles/" + RND + "_summary.json; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 wide coalesced reads); "
                         "Infinity-Cache hits are counted"}, open(os.path.join(prof, "traffic_latest.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
