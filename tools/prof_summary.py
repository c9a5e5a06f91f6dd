#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel trace stats / counter collection) into small text summaries
that can be committed under profiles/.  Usage: prof_summary.py <rocprof_out_dir> <summary.txt>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("gms::", "").replace("void ", "")[:60]


def main(d, out):
    lines = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        lines.append(f"# {os.path.relpath(f, d)}  (rocprofv3 --kernel-trace --stats)")
        lines.append(f"{'kernel':62s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for r in csv.DictReader(open(f)):
            lines.append(f"{short(r['Name']):62s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e3:12.1f} "
                         f"{float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        regs = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
            regs[k] = (r.get("VGPR_Count", "?"), r.get("SGPR_Count", "?"), r.get("LDS_Block_Size", "?"))
        lines.append(f"# {os.path.relpath(f, d)}  (rocprofv3 --pmc; mean per dispatch)")
        for k in sorted(acc):
            vals = "  ".join(f"{c}={v[0]/v[1]:.4g}" for c, v in sorted(acc[k].items()))
            lines.append(f"{k:62s} n={next(iter(acc[k].values()))[1]:<5d} vgpr={regs[k][0]} sgpr={regs[k][1]} lds={regs[k][2]}  {vals}")
    # HBM traffic per launch from the PMC passes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
    # counts wide coalesced reads at half their size (MI355X_MICROARCH.md, HBM section): both the raw and
    # the corrected (2x fetch) totals are reported.
    traffic = {}
    sq = {}
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]).split("<")[0].replace("_kernel", "")
            if r["Counter_Name"].startswith("SQ_"):
                a = sq.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
                a[0] += float(r["Counter_Value"]); a[1] += 1
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            t = traffic.setdefault(k, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
            t[r["Counter_Name"]][0] += float(r["Counter_Value"]); t[r["Counter_Name"]][1] += 1
    if traffic:
        import json
        js = {}
        for k, t in traffic.items():
            fe = t["FETCH_SIZE"][0] / max(t["FETCH_SIZE"][1], 1) * 1024
            wr = t["WRITE_SIZE"][0] / max(t["WRITE_SIZE"][1], 1) * 1024
            js[k] = {"fetch_bytes_raw": round(fe), "write_bytes": round(wr), "hbm_bytes_corrected": round(2 * fe + wr)}
            if k in sq:
                js[k]["sq"] = {c: v[0] / max(v[1], 1) for c, v in sq[k].items()}      # mean per dispatch
        open(os.path.splitext(out)[0] + "_traffic.json", "w").write(json.dumps(js, indent=1, sort_keys=True))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:80]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
