#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of bench.py: over the last `--window-ms` of GPU activity, the share of wall time in which at least one kernel runs (union of the
kernel intervals), the share in which at least two run, and the mean number of kernels in flight.   python tools/busy_fraction.py <kernel_trace.csv> [--window-ms 400] [--skip-ms 0]"""
import argparse, csv
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--window-ms", type=float, default=400.0); ap.add_argument("--skip-ms", type=float, default=0.0)
a = ap.parse_args()
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(a.csv)))
t_end = max(e for _, e in iv) - int(a.skip_ms * 1e6)
t_beg = t_end - int(a.window_ms * 1e6)
ev = []
for s, e in iv:
    s, e = max(s, t_beg), min(e, t_end)
    if e > s:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, busy1, busy2, area = 0, t_beg, 0, 0, 0
for t, d in ev:
    dt = t - last
    if depth >= 1: busy1 += dt
    if depth >= 2: busy2 += dt
    area += depth * dt
    depth += d; last = t
W = t_end - t_beg
print(f"window {W / 1e6:.1f} ms: >= 1 kernel running {busy1 / W:.3f}, >= 2 running {busy2 / W:.3f}, mean kernels in flight {area / W:.2f}, kernels {len(ev) // 2}")
