#!/usr/bin/env python3
"""Matrix-pipe occupancy per kernel class from one rocprofv3 SQ counter pass (its own run: --kernel-trace --pmc only):
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \\
              -d gpurun_out/sq --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-modes --serial-decode --no-run-ahead
    python tools/sq_counters.py gpurun_out/sq profiles/r02_sq_counters.json
Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_*); SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES is summed over the 32 shader engines (32 SIMDs each), so the matrix-pipe
occupancy of a class is  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)  and the clock the chip held is SQ_BUSY_CYCLES / 32 / time.
(GRBM_GUI_ACTIVE / 8 / time reads high on the sub-0.3-ms dispatches this workload consists of - the guide warns of it - and is kept only as a raw value.)"""
import collections, csv, glob, json, sys

from pmc_traffic import CLASSES, cls  # noqa: F401  (same kernel -> class map)

d, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        c = cls(r["Kernel_Name"])
        if not c:
            continue
        acc[c][r["Counter_Name"]] += float(r["Counter_Value"])
        key = r["Dispatch_Id"]
        if key not in n[c]:
            n[c].add(key)
            dur[c] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {"_note": "rocprofv3 --kernel-trace --pmc <SQ counters> GRBM_GUI_ACTIVE over python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-modes --serial-decode "
                "--no-run-ahead (every kernel alone on the GPU); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES) (32 shader engines x 32 SIMDs); clock_ghz = SQ_BUSY_CYCLES / 32 / time; "
                "wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of wave lifetime parked in s_waitcnt / barriers); profiled passes run at a lower clock than unprofiled ones"}
for c, _ in CLASSES:
    if c not in acc:
        continue
    a = acc[c]
    e = {"launches": len(n[c]), "time_ms": dur[c] * 1e-6, "counters": {k: v for k, v in sorted(a.items())}}
    if a.get("SQ_BUSY_CYCLES"):
        e["clock_ghz"] = a["SQ_BUSY_CYCLES"] / 32.0 / dur[c] if dur[c] else None
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            e["mfma_busy"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * a["SQ_BUSY_CYCLES"])
    if a.get("SQ_WAVE_CYCLES"):
        e["wait_frac"] = a.get("SQ_WAIT_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
        e["issue_stall_frac"] = a.get("SQ_WAIT_INST_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
    res[c] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in res.items() if k != "_note"}, indent=1))
