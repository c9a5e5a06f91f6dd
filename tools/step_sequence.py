"""Per-launch view of one bench step from a rocprofv3 kernel trace: the launches of a step in stream order with their average
durations over the last steps of the trace (the --stats table merges launches of the same kernel, e.g. the two phases of
`blend_head` on deep frames).

    python tools/step_sequence.py <dir holding *_kernel_trace.csv> <out.txt> [first kernel of a step = mesh_fwd_kernel]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    root, out = sys.argv[1], sys.argv[2]
    first = sys.argv[3] if len(sys.argv) > 3 else "mesh_fwd_kernel"
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no kernel trace under {root}")
    rows = []
    with open(files[0]) as f:
        rd = csv.DictReader(f)
        col = lambda part: next(c for c in rd.fieldnames if part in c.lower())
        c_start, c_end, c_name = col("start"), col("end"), col("kernel_name")
        for r in rd:
            rows.append((int(r[c_start]), int(r[c_end]), r[c_name]))
    rows.sort()
    short = lambda n: re.sub(r"\(.*", "", n).replace("gms::", "").replace("void ", "")[:64]
    steps, cur = [], None
    for s, e, n in rows:
        if short(n).startswith(first):
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append((short(n), (e - s) / 1000.0, s, e))
    steps = [st for st in steps[:-1]]                     # (the last one may be cut)
    shape = defaultdict(int)
    for st in steps:
        shape[tuple(k for k, *_ in st)] += 1
    seq = max(shape, key=shape.get)
    use = [st for st in steps if tuple(k for k, *_ in st) == seq][-20:]
    with open(out, "w") as f:
        f.write(f"# {files[0].split('/')[-1]}: {len(steps)} steps, {len(use)} averaged (most frequent launch sequence, {len(seq)} launches)\n")
        f.write(f"{'#':>3} {'kernel':64s} {'avg_us':>9} {'gap_before_us':>14}\n")
        tot = 0.0
        for i, k in enumerate(seq):
            d = sum(st[i][1] for st in use) / len(use)
            gap = sum((st[i][2] - st[i - 1][3]) / 1000.0 for st in use) / len(use) if i else 0.0
            tot += d
            f.write(f"{i:3d} {k:64s} {d:9.2f} {gap:14.2f}\n")
        span = sum((st[-1][3] - st[0][2]) / 1000.0 for st in use) / len(use)
        f.write(f"# sum of kernels {tot:.1f} us, first start to last end {span:.1f} us\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
