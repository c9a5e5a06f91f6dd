#!/usr/bin/env python
"""profiles/pmc_traffic.json from a collect_profiles.sh traffic summary.
usage: make_pmc_traffic.py <summary_traffic.json> <workload/state> [profiles/pmc_traffic.json] [blend_stats.json]
The optional blend_stats.json (tools/blend_stats.py, CPU) adds the active (pixel, splat) pair count of the workload."""
import json
import os
import sys

src, key = sys.argv[1], sys.argv[2]
stats_path = sys.argv[4] if len(sys.argv) > 4 else None
dst = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
t = json.load(open(src))
names = {"preprocess_fwd": ["preprocess_fwd_dma", "preprocess_fwd"], "tile_scan": ["tile_scan"], "emit_instances": ["emit_instances"],
         "tile_sort": ["tile_presort_reg", "tile_presort", "tile_merge"], "blend_head": ["micro_head", "blend_head"], "blend_fwd": ["micro_fwd", "blend_fwd"],
         "blend_finalize": ["micro_finalize", "blend_finalize"], "blend_bwd": ["micro_bwd", "blend_bwd"], "micro_filter": ["micro_filter"],
         "preprocess_bwd": ["preprocess_bwd"],
         "mesh_fwd": ["mesh_fwd"], "mesh_bwd_splat": ["mesh_bwd_splat"], "mesh_bwd_face": ["mesh_bwd_face_thread", "mesh_bwd_face_wave", "mesh_bwd_fused"]}
out = {}
for k, srcs in names.items():
    present = [s for s in srcs if s in t]
    if k != "tile_sort":
        present = present[:1]                    # (the first name that ran: micro_* kernels or the quadrant kernels)
    vals = [t[s]["hbm_bytes_corrected"] for s in present]
    if vals:
        out[k] = int(sum(vals) / len(vals))      # per launch (tile_sort = mean of its two launches)
# SQ counters (wave-instruction counts, mean per dispatch) for the VALU roofline of the compositing kernels
sqo = {}
for k, srcs in names.items():
    for s_ in srcs:
        if s_ in t and "sq" in t[s_]:
            sqo[k] = {c: round(v) for c, v in t[s_]["sq"].items()}
            break
if stats_path and os.path.exists(stats_path):
    st = json.load(open(stats_path))
    pairs = st["active_pairs_per_instance"] * st["N_total"]
    for k in ("blend_bwd", "blend_head", "blend_fwd"):
        if k in sqo:
            sqo[k]["active_pairs"] = round(pairs)
            if sqo[k].get("SQ_INSTS_VALU"):
                sqo[k]["active_lane_frac"] = round(st["blocks"]["4x4"]["active_lane_frac_exact"], 3)      # micro-tile rows
out["_sq"] = sqo
# the build the counters belong to: bench.py withholds them from any other build (same hash function as bench.py)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import srchash
out["_source_hash"] = srchash.kernel_source_hash()
allj = json.load(open(dst)) if os.path.exists(dst) else {}
allj[key] = out
allj["_note"] = ("HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB->bytes) from separate rocprofv3 --pmc passes; "
                 "FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md (HBM section)")
json.dump(allj, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
