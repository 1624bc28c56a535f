#!/usr/bin/env python3
"""Per-stage kernel summary of a rocprofv3 --kernel-trace CSV of tools/stage_trace.py: the calls are separated by host syncs (gaps > 30 us); prints, for the last
`--last` calls, every kernel (name, grid) with its launch count, average duration and average gap to the previous kernel's end.
python tools/trace_stages.py gpurun_out/.../kernel_trace.csv [--last 2]"""
import argparse, collections, csv, re
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--last", type=int, default=2); ap.add_argument("--min-kernels", type=int, default=60)
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.csv)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("sdvar::", ""))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in rows]
segs, cur = [], [ev[0]]
for p, q in zip(ev, ev[1:]):
    if q[0] - p[1] > 30000:
        segs.append(cur); cur = []
    cur.append(q)
segs.append(cur)
segs = [s for s in segs if len(s) >= a.min_kernels]
for si, s in enumerate(segs[-a.last:]):
    agg, prev = collections.OrderedDict(), None
    for st, en, nm, g in s:
        d = agg.setdefault((nm, g), [0, 0.0, 0.0]); d[0] += 1; d[1] += (en - st) / 1e3
        if prev is not None:
            d[2] += (st - prev) / 1e3
        prev = en
    print(f"--- call {si}: {len(s)} kernels, span {(s[-1][1] - s[0][0]) / 1e3:.1f} us")
    for (nm, g), (c, d, gp) in agg.items():
        print(f"   {nm[:60]:60s} grid {g:>7} n {c:3d} avg dur {d / c:6.2f} gap {gp / c:5.2f}")
