"""Offline: what a better block order would buy the micro-tile forward launches (round 6).  Input: the per-block records tools/micro_fwd_phases.py
dumps with GMS_PHASES_DUMP (start / staged / end of every block, its unit's kind, segment, entries, its tile's list length).  The launch is
replayed as greedy list scheduling on `slots` block slots (the dispatcher starts the next block of the grid whenever a slot frees) with the
measured block durations, under: the grid's own order, longest-first (the ideal a static order can reach), and candidate keys that are
known when the unit table is written (the tile's list length, the segment index)."""
import heapq
import sys

import numpy as np


def makespan(dur, order, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for i in order:
        t = heapq.heappop(h) + dur[i]
        end = max(end, t)
        heapq.heappush(h, t)
    return end


for path in sys.argv[1:]:
    z = np.load(path)
    dur = z["end"] - z["start"]
    walk = z["end"] - z["staged"]
    n = len(dur)
    order0 = np.argsort(z["block"])
    slots = 2048
    print(f"{path}: {n} blocks, measured span {z['end'].max():.1f} us, sum of durations {dur.sum():.0f} block-us = {dur.sum() / slots:.1f} us on {slots} slots, longest block {dur.max():.1f}")
    print(f"  replay, grid order            : {makespan(dur, order0, slots):5.1f} us")
    print(f"  replay, longest first (ideal) : {makespan(dur, np.argsort(-dur), slots):5.1f} us")
    te, seg, nseg, ent, kind = z["tile_entries"].astype(float), z["seg"], z["nseg"], z["entries"], z["kind"]
    keys = {
        "shallow tiles first (ascending tile list length among non-empty)": np.where(ent > 0, te, 1e12),
        "first segments of shallow tiles, then by segment index descending": np.where(ent > 0, te * 1e-3 - (seg == 0) * 1e6 - seg, 1e12),
        "full units first, shallow tiles first": np.where(ent > 0, -(ent >= 256).astype(float) * 1e9 + te, 1e12),
        "full units first, deep tiles first (descending)": np.where(ent > 0, -(ent >= 256).astype(float) * 1e9 - te, 1e12),
    }
    for name, k in keys.items():
        print(f"  replay, {name:68s}: {makespan(dur, np.argsort(k, kind='stable'), slots):5.1f} us")
    # what the walk time depends on
    for kd in sorted(set(kind.tolist())):
        m = (kind == kd) & (ent >= 256)
        if m.sum() < 20:
            continue
        print(f"  kind {kd}: full units {int(m.sum())}: walk vs tile list length  corr {np.corrcoef(te[m], walk[m])[0, 1]:+.2f};  by tile-length quartile: "
              + "  ".join(f"{np.mean(walk[m][(te[m] >= lo) & (te[m] <= hi)]):.1f}" for lo, hi in zip(np.quantile(te[m], [0, .25, .5, .75]), np.quantile(te[m], [.25, .5, .75, 1]))))
        if kd in (2, 4):
            print(f"          walk vs segment index corr {np.corrcoef(seg[m], walk[m])[0, 1]:+.2f}; vs (seg / nseg) {np.corrcoef(seg[m] / np.maximum(nseg[m], 1), walk[m])[0, 1]:+.2f}")
