#!/usr/bin/env python3
"""Copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked).
usage: scripts/make_profiles.py <round tag, e.g. r01> <pmc dir under gpurun_out, e.g. pmc4>"""
import collections, csv, json, os, shutil, sys
tag, pmc = sys.argv[1], sys.argv[2]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
os.makedirs(P, exist_ok=True)
shutil.copy(os.path.join(G, "bench_%s.log" % tag), os.path.join(P, "%s_bench_line.json" % tag))
shutil.copy(os.path.join(G, "prof_bench", "bench_kernel_stats.csv"), os.path.join(P, "%s_bench_kernel_stats.csv" % tag))
out = {}
base = os.path.join(G, pmc)
for grp in ("sq1", "tcc", "sq2"):
    rows = list(csv.DictReader(open(os.path.join(base, grp, "p_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(base, grp, "p_kernel_trace.csv"))):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        if k.startswith("__amd") or "corpus" in k:
            continue
        d = out.setdefault(k, {})
        d.setdefault("avg_duration_us", {})[grp] = round(sum(dur[k]) / len(dur[k]), 2)
        d["launches"] = len(dur[k])
        for c, vals in v.items():
            d[c] = round(sum(vals) / len(vals))
    os.makedirs(os.path.join(P, "%s_pmc_raw" % tag), exist_ok=True)
    shutil.copy(os.path.join(base, grp, "p_counter_collection.csv"),
                os.path.join(P, "%s_pmc_raw" % tag, "%s_counter_collection.csv" % grp))
sw = [k for k in out if "k_sweep<4" in k][0]
d = out[sw]
fetch = d["FETCH_SIZE"]
dur_s = d["avg_duration_us"]["sq1"] * 1e-6
clock = d["GRBM_GUI_ACTIVE"] / 8 / dur_s
simd_quad = 1024 * dur_s * clock / 4
basis = 4 * 2 ** 30
summary = {
    "source": "rocprofv3 --pmc, three separate passes with --kernel-trace only (scripts/pmc_passes.sh %s 4 2 lean); averages per launch" % pmc,
    "command": "python scripts/prof_k2.py 4 2 lean  (6 count-only scans of the 4 GiB C2 corpus resident in HBM)",
    "kernels": out,
    "k_sweep_derived": {
        "clock_GHz": round(clock / 1e9, 3),
        "valu_insts_per_4KiB_supertile": round(d["SQ_INSTS_VALU"] / (basis / 4096), 1),
        "valu_busy_frac": round(d["SQ_ACTIVE_INST_VALU"] / simd_quad, 3),
        "wave_time_waiting_on_memory_frac": round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3),
        "lds_bank_conflict_frac_of_lds_active": round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 3)},
    "k_sweep_traffic": {
        "FETCH_SIZE_raw_KiB": fetch,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 128-B requests at 64 B -> x2; unit KiB",
        "hbm_read_bytes_per_launch": int(fetch * 1024 * 2), "algorithmic_bytes_per_launch": basis,
        "ratio": round(fetch * 1024 * 2 / basis, 4)}}
json.dump(summary, open(os.path.join(P, "%s_pmc_summary.json" % tag), "w"), indent=1)
json.dump({"bytes_per_launch_basis": basis, "hbm_read_bytes_per_launch": int(fetch * 1024 * 2), "kernel": sw,
           "note": "FETCH_SIZE x 1024 x 2 (gfx950 correction), see %s_pmc_summary.json" % tag},
          open(os.path.join(P, "%s_pmc_traffic.json" % tag), "w"), indent=1)
print(json.dumps(summary["k_sweep_derived"]), json.dumps(summary["k_sweep_traffic"]))
for k, v in out.items():
    print(k, v.get("avg_duration_us"))
