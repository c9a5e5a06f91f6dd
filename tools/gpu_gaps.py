"""GPU busy / idle analysis of a rocprofv3 kernel trace (run on the GPU box).
usage: gpu_gaps.py <kernel_trace.csv> [skip_fraction]   -- prints busy %, the idle gaps by the kernel that FOLLOWS them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(len(ev) * skip):]
span = ev[-1][1] - ev[0][0]
busy = sum(e - s for s, e, _ in ev)
print(f"kernels {len(ev)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us ({100*busy/span:.1f} %)  idle {(span-busy)/1e3:.1f} us")
gaps = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end = ev[0][1]
for s, e, n in ev[1:]:
    g = max(0, s - prev_end)
    k = n.split("(")[0][:60]
    gaps[k][0] += 1; gaps[k][1] += g; gaps[k][2] = max(gaps[k][2], g)
    prev_end = max(prev_end, e)
print(f"{'gap BEFORE kernel':62s} {'n':>5s} {'mean_us':>8s} {'max_us':>8s} {'total_us':>9s}")
for k, (n, tot, mx) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:62s} {n:5d} {tot/n/1e3:8.2f} {mx/1e3:8.1f} {tot/1e3:9.1f}")
