#!/usr/bin/env python3
"""HBM-side bytes per launch and kernel class from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f --output-format csv -- python3 bench.py --steps 1 --warmup 0 ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w --output-format csv -- python3 bench.py --steps 1 --warmup 0 ...
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/pmc_traffic.json
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide streaming reads at 64 bytes and is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv, glob, json, sys, collections

# "gemm_small" = the 32- / 64-row kernels (ring and K-split ping-pong) and the skinny (M <= 80) kernel (the launches with M < 1024 rows, bench.py's gemm_small class); "gemm" = the 128 x 128, 256 x 128 and 256 x 256 kernels
CLASSES = [("gemm_small", ("gemm_f16x2_small_kernel", "gemm_f16x2_small_pp_kernel", "gemm_f16x2_skinny_kernel", "gemm_f16x2_rowblk_kernel")), ("gemm", ("gemm_f16x2", "gemm_bf16x3", "gemm_f32_nt")), ("splitk_reduce", ("splitk_reduce",)), ("ln_modulate", ("ln_modulate",)),
           ("qk_norm_append", ("qk_norm_append",)), ("attention", ("attention_f16x2", "attention_bf16x3", "attention_f32")), ("sampler", ("cfg_sample",)),
           ("decoder_conv", ("conv_f16x2", "conv_bf16x3", "conv_reduce")), ("decoder_rows", ("prep_planes", "gn_partial", "gn_finalize", "vae_attn", "convout", "rows_from_nchw"))]


def cls(name):
    for c, keys in CLASSES:
        if any(k in name for k in keys):
            return c
    return None


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            c = cls(r["Kernel_Name"])
            if c:
                acc[c][0] += 1; acc[c][1] += float(r["Counter_Value"])
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                    "--no-extra-modes --serial-decode; KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)"}
    for c, _ in CLASSES:
        if c in F and c in W and F[c][0]:
            n = F[c][0]
            fb, wb = F[c][1] * 1024 * 2 / n, W[c][1] * 1024 / max(W[c][0], 1)
            res[c] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
