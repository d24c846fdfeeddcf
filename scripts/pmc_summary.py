#!/usr/bin/env python3
"""Per-kernel summary of the three PMC passes scripts/pmc_passes.sh leaves under gpurun_out/<dir>
(separate rocprofv3 --pmc runs with --kernel-trace only).
usage: scripts/pmc_summary.py <dir under gpurun_out> <kernel name substring> <bytes per launch> <out.json> [note]"""
import collections, csv, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base, want, nbytes, outp = os.path.join(R, "gpurun_out", sys.argv[1]), sys.argv[2], float(sys.argv[3]), sys.argv[4]
note = sys.argv[5] if len(sys.argv) > 5 else ""
out = {}
for grp in ("sq1", "tcc", "sq2", "tcc2"):
    if not os.path.exists(os.path.join(base, grp, "p_counter_collection.csv")):
        continue
    rows = list(csv.DictReader(open(os.path.join(base, grp, "p_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(base, grp, "p_kernel_trace.csv"))):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        if want not in k:
            continue
        d = out.setdefault(k, {})
        d.setdefault("avg_duration_us", {})[grp] = round(sum(dur[k]) / len(dur[k]), 2)
        d["launches"] = len(dur[k])
        for c, vals in v.items():
            d[c] = round(sum(vals) / len(vals))
res = {"source": "rocprofv3 --pmc, three separate passes with --kernel-trace only (scripts/pmc_passes.sh); averages per launch",
       "note": note, "bytes_per_launch": nbytes, "kernels": {}}
for k, d in out.items():
    dur_s = d["avg_duration_us"]["sq1"] * 1e-6
    clock = d["GRBM_GUI_ACTIVE"] / 8 / dur_s
    simd_quad = 1024 * dur_s * clock / 4           # quad-cycles of all 1024 SIMDs during the launch
    res["kernels"][k] = {
        "counters": d,
        "derived": {
            "GBps": round(nbytes / 1e3 / d["avg_duration_us"]["sq1"], 1),
            "clock_GHz": round(clock / 1e9, 3),
            "valu_insts_per_byte": round(d["SQ_INSTS_VALU"] * 64 / nbytes, 2),
            "valu_wave_insts_per_simd_clock": round(d["SQ_INSTS_VALU"] / (1024 * dur_s * clock), 3),
            "valu_active_quadcycles_over_simd_quadcycles": round(d["SQ_ACTIVE_INST_VALU"] / simd_quad, 3),
            "wave_time_waiting_on_memory_frac": round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3),
            "wave_time_waiting_to_issue_frac": round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 3),
            "lds_insts_per_byte": round(d["SQ_INSTS_LDS"] * 64 / nbytes, 3) if "SQ_INSTS_LDS" in d else None,
            "lds_bank_conflict_frac_of_lds_active": (round(d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1), 3)
                                                     if "SQ_LDS_BANK_CONFLICT" in d else None),
            "FETCH_SIZE_KiB_x1024_x2_over_bytes": round(d.get("FETCH_SIZE", 0) * 1024 * 2 / nbytes, 3),
            "ea_read_requests": d.get("TCC_EA0_RDREQ_sum"), "ea_read_requests_32B": d.get("TCC_EA0_RDREQ_32B_sum"),
            "ea_read_bytes_over_bytes_if_others_are_64B": (round((d["TCC_EA0_RDREQ_32B_sum"] * 32 + (d["TCC_EA0_RDREQ_sum"] - d["TCC_EA0_RDREQ_32B_sum"]) * 64) / nbytes, 3)
                                                             if "TCC_EA0_RDREQ_sum" in d and "TCC_EA0_RDREQ_32B_sum" in d else None),
            "l2_hit_frac": (round(d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1), 3) if "TCC_HIT_sum" in d else None),
            "fetch_note": "x2 is the guide's gfx950 correction for 128-byte requests; a kernel that reads 64-byte "
                          "segments (the ring feeder) may issue 64-byte requests, then the true ratio is half of this"}}
json.dump(res, open(os.path.join(R, outp), "w"), indent=1)
for k, v in res["kernels"].items():
    print(k[:70], json.dumps(v["derived"]))
