#!/usr/bin/env python3
"""Summarise the counter passes of tools/runs/attn_counters.sh: per configuration (busy (row, head) pairs) the attention kernel's launches, average duration,
clock the chip held (SQ_BUSY_CYCLES / 32 / time - the units of tools/sq_counters.py), matrix-pipe occupancy, wait share, and per-launch instruction counts.
python tools/attn_counters.py gpurun_out/<tag> out.json"""
import collections, csv, glob, json, os, re, sys
pre, out = sys.argv[1:3]
res = {"_note": "attention_f16x2_pp_kernel alone on the GPU, l = 256 queries over K = 680 keys per (row, head) unless named P4 (l = 1024, K = 2240, one fp16 plane); clock_ghz = SQ_BUSY_CYCLES / 32 / time; "
                "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 SQ_BUSY_CYCLES); per-launch instruction counts are summed over all waves of a launch; us_unprofiled from the same shape timed with HIP events"}
for d in sorted(glob.glob(pre + "_A_*") + glob.glob(pre + "_B_*")):
    if not os.path.isdir(d):
        continue
    m = re.search(r"_([AB])_(\w+)$", d)
    key = m.group(2)
    acc, disp, dur = collections.defaultdict(float), set(), 0.0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attention_f16x2" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp:
                disp.add(r["Dispatch_Id"]); dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if not disp:
        continue
    e = res.setdefault(key, {})
    n = len(disp)
    e.setdefault("launches_" + m.group(1), n)
    if m.group(1) == "A":
        e["avg_us_profiled"] = dur / n * 1e-3
        if acc.get("SQ_BUSY_CYCLES"):
            e["clock_ghz"] = acc["SQ_BUSY_CYCLES"] / 32.0 / dur
            e["mfma_busy"] = acc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * acc["SQ_BUSY_CYCLES"])
        if acc.get("SQ_WAVE_CYCLES"):
            e["wait_frac"] = acc.get("SQ_WAIT_ANY", 0.0) / acc["SQ_WAVE_CYCLES"]
            e["valu_active_frac"] = acc.get("SQ_ACTIVE_INST_VALU", 0.0) / acc["SQ_WAVE_CYCLES"]
        e["grbm_gui_active_per_launch"] = acc.get("GRBM_GUI_ACTIVE", 0.0) / n
    else:
        for k, v in acc.items():
            e[k + "_per_launch"] = v / n
    t = pre + "_time_" + key + ".log"
    if os.path.exists(t):
        mm = re.search(r": ([0-9.]+) us/launch", open(t).read())
        if mm:
            e["us_unprofiled"] = float(mm.group(1))
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
