#!/usr/bin/env python3
"""Timeline of the LAST `--count` kernels whose grid matches a stage from a rocprofv3 kernel-trace CSV: name, duration, gap to the previous kernel's end.
python tools/trace_timeline.py <kernel_trace.csv> [--skip-last N] [--count 40]"""
import argparse, csv, re
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--count", type=int, default=40); ap.add_argument("--skip-last", type=int, default=0)
ap.add_argument("--summary", action="store_true", help="per-kernel-name totals over the selected window instead of the timeline")
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = len(rows) - a.skip_last
sel = rows[max(0, end - a.count):end]
def short(n):
    n = re.sub(r"^void ", "", n); n = n.replace("sdvar::", ""); n = re.sub(r"\(.*", "", n); return n[:44]
prev = None; t0 = int(sel[0]["Start_Timestamp"]); agg = {}
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    nm = short(r["Kernel_Name"])
    if a.summary:
        d = agg.setdefault(nm, [0, 0.0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3; d[2] += max(gap, 0.0)
    else:
        print(f"{(s - t0) / 1e3:9.1f} us  {nm:44s} dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))} lds {r.get('LDS_Block_Size', '?')}")
    prev = e
if a.summary:
    for nm, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{nm:44s} n {c:4d}  dur {d:9.1f} us (avg {d / c:6.2f})  gaps before {g:8.1f} us")
    print(f"window {(int(sel[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, {len(sel)} kernels")
