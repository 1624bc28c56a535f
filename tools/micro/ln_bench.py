#!/usr/bin/env python3
"""ln_modulate alone (planes output, as inside stage_forward): us per launch and algorithmic GB/s (8 M C bytes: x read + two fp16 planes written).
python tools/micro/ln_bench.py [C]
Round 3 measured a two-rows-per-wave variant with it: 14.6 us against 8.3 us at 4096 rows (profiles/r03_i_ln.log) - dropped; one row per wave reaches 4.0 TB/s."""
import ctypes as C_, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
Cw = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = E.load_library(); dev = torch.device("cuda:0"); st = C_.c_void_p(torch.cuda.current_stream().cuda_stream); P = lambda t: C_.c_void_p(t.data_ptr()) if t is not None else None
for rows, l in ((1024, 64), (1600, 100), (2704, 169), (4096, 256), (6800, 425)):
    x = torch.randn(rows, Cw, device=dev); mod = torch.randn(16, 6 * Cw, device=dev)
    op = torch.empty(2, Cw // 32, rows, 32, dtype=torch.int16, device=dev)
    sc, sh = C_.c_void_p(mod.data_ptr() + 4 * 2 * Cw), C_.c_void_p(mod.data_ptr() + 4 * 4 * Cw)
    run = lambda: E._check(lib.sdvar_op_ln_modulate(P(x), sc, sh, None, P(op), rows * Cw, 2, rows, Cw, l, 6 * Cw, st))
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print(f"rows {rows:5d} C {Cw}: {us:6.2f} us  {8.0 * rows * Cw / us * 1e-3:7.0f} GB/s", flush=True)
