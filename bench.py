#!/usr/bin/env python3
"""Headline benchmark: images/s of the speculative draft->verify sampler, VAR-d16 256^2, B=8 per GPU, d12 draft
(BASELINE.json configs[1]), random-init weights, synthetic labels.  One process per GPU (torchrun env), weak scaling:
every rank samples its own B images; the only collective is the RCCL all-gather of the per-rank counters.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant kernel class of the step (by HIP-event time on the launch stream), algorithmic
                flops / bytes per launch from DESIGN.md section 4 vs the gfx950 peak
  cpu_baseline  the CPU oracle (oracle/var_oracle.py, the pinned restatement of the reference's
                VAR.autoregressive_infer_cfg, target model only, no speculation) timed on this host's cores, N=1 only
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Usable host threads: the affinity mask, capped by the cgroup CPU quota when there is one (a 1-GPU box gets a
    16-CPU share of a much larger host; sizing OpenMP to os.cpu_count() there oversubscribes by an order of magnitude)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("SDVAR_CPU_THREADS", 16))))


# BASELINE.json configs -> (draft depth, target depth, ladder, images per GPU, cfg, fp16 KV cache)
CONFIGS = {
    "P1": dict(dd=12, dt=16, ladder="256", B=8, cfg=1.5, kv_fp16=False, name="VAR-d16 256^2 B=8, d12 draft + d16 verify"),
    "P2": dict(dd=16, dt=24, ladder="256", B=16, cfg=1.5, kv_fp16=False, name="VAR-d24 256^2 B=16, d16 draft + d24 verify"),
    "P3": dict(dd=16, dt=30, ladder="256", B=8, cfg=1.5, kv_fp16=False, name="VAR-d30 256^2 B=64 over 8 GPUs = B=8 per GPU (draft unspecified in BASELINE.json: d16), batch split, counters gathered over RCCL"),
    "P4": dict(dd=16, dt=30, ladder="512", B=8, cfg=3.0, kv_fp16=True, name="VAR-d30 512^2 B=8, cfg 3.0, fp16 KV cache (draft unspecified in BASELINE.json: d16)"),
}

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide, dense bf16 MFMA; the bf16x3 GEMM spends 6 bf16 products per fp32 product
PEAK_HBM_GBS = 8000.0             # same guide, HBM3E spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="P1", choices=sorted(CONFIGS), help="BASELINE.json configuration (P1 = configs[1], the one `metric` is quoted on; "
                    "P2 / P3 / P4 = configs[2] / configs[3] (one rank's share: run with --gpus 8 for the whole config) / configs[4], parity cases with a bench row)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the configuration's)")
    ap.add_argument("--depth-draft", type=int, default=None)
    ap.add_argument("--depth-target", type=int, default=None)
    ap.add_argument("--gamma", type=int, default=2)
    ap.add_argument("--mode", default="natural", choices=["natural", "accept_all", "reject_all"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-modes", action="store_true")
    ap.add_argument("--sampler-high-prio", action="store_true", help="experiment: run the sampler's streams at the higher HIP priority, the decode at the normal one")
    ap.add_argument("--decode-high-prio", action="store_true", help="experiment: give the decode stream the higher HIP priority")
    ap.add_argument("--torch-decode", action="store_true", help="A/B: decode with the PyTorch/MIOpen reference decoder instead of the HIP decoder")
    ap.add_argument("--no-run-ahead", action="store_true", help="keep the draft waiting for the verifier at gamma == 1 (every kernel alone on the GPU: profiling runs)")
    ap.add_argument("--serial-decode", action="store_true", help="decode on the sampling stream instead of overlapping it with the next batch")
    ap.add_argument("--gemm-mode", default=None, choices=["f32", "bf16x3", "f16x2"], help="default: sdvar_amd.engine.DEFAULT_GEMM_MODE")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark for the VQVAE decoder convs")
    args = ap.parse_args()

    from sdvar_amd import dist as D
    from sdvar_amd import engine as E
    from sdvar_amd.ladder import LADDER_256, LADDER_512, as_ladder
    from sdvar_amd.vqvae import VQVAE
    from sdvar_amd.weights import var_state_dict_device, vae_state_dict

    rank, world, local = D.init_from_env("cuda")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    conf = CONFIGS[args.config]
    args.batch = args.batch or conf["B"]
    args.depth_draft, args.depth_target = args.depth_draft or conf["dd"], args.depth_target or conf["dt"]
    pns = LADDER_256 if conf["ladder"] == "256" else LADDER_512
    B, lad, CFG = args.batch, as_ladder(pns), conf["cfg"]
    thr = {"natural": 0.5, "accept_all": 0.0, "reject_all": 2.0}

    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(dev)}; host cpu_count={os.cpu_count()} affinity={len(os.sched_getaffinity(0))} usable={host_cores()}")
    sd_d = var_state_dict_device(args.depth_draft, pns, dev, seed=1234)
    sd_t = var_state_dict_device(args.depth_target, pns, dev, seed=1234)
    sd_v = vae_state_dict(pns, "perf", 1234, with_encoder=False)
    vae = VQVAE(vocab_size=4096, z_channels=32, ch=160, v_patch_nums=pns, with_encoder=False)
    vae.load_state_dict(sd_v); vae = vae.to(dev)
    dc = E.ModelCtx(sd_d, args.depth_draft, pns, B, 1, dev, gemm_mode=args.gemm_mode, kv_fp16=conf["kv_fp16"])
    tc = E.ModelCtx(sd_t, args.depth_target, pns, B, max(args.gamma, 1), dev, gemm_mode=args.gemm_mode, kv_fp16=conf["kv_fp16"])
    qc = E.QuantCtx(sd_v, pns, B, dev)
    smp = E.Sampler(tc, qc, dc)
    log("models bound, buffers allocated")
    lo, _ = D.shard_range(B * world, rank, world)
    labels = ((torch.arange(B) + lo) % 1000).to(dev)

    # The VQVAE decode of batch i runs on a second HIP stream and overlaps the sampling loop of batch i+1 (whose early
    # stages leave most CUs idle); f_hat is double-buffered and every decode is finished inside the timed region.
    decode = vae.fhat_to_img
    if args.torch_decode:          # A/B only: the PyTorch/MIOpen decoder of tests/torch_ref.py (test infrastructure, not the product path)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from torch_ref import fhat_to_img_torch
        decode = lambda f: fhat_to_img_torch(vae, f)
    if args.sampler_high_prio:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    main_stream = torch.cuda.current_stream()
    dec_stream = torch.cuda.Stream(device=dev, priority=0 if not args.decode_high_prio else -1)
    fh_buf = [torch.zeros(B, 32, lad.HW, lad.HW, device=dev) for _ in range(2)]
    dec_done = [None, None]
    state = {"i": 0, "img": None}

    def step(mode, seed, run_ahead=None):
        ra = (not args.no_run_ahead) if run_ahead is None else run_ahead
        res = smp.spec_decode(labels, CFG, args.gamma, 900, 0.96, E.Noise("device", seed, image_offset=lo), thr=thr[mode], run_ahead=ra)
        st = dict(res.stats); st["images"] = B
        if args.serial_decode:
            state["img"] = decode(res.f_hat).add_(1).mul_(0.5)                   # (B,3,256,256) in [0,1]  (var.py:215)
            return state["img"], st
        j = state["i"] & 1; state["i"] += 1
        if dec_done[j] is not None:
            main_stream.wait_event(dec_done[j])                     # the decode that last read this buffer
        fh_buf[j].copy_(res.f_hat)
        ready = torch.cuda.Event(); ready.record(main_stream)
        dec_stream.wait_event(ready)
        with torch.cuda.stream(dec_stream):
            state["img"] = decode(fh_buf[j]).add_(1).mul_(0.5)
            dec_done[j] = torch.cuda.Event(); dec_done[j].record(dec_stream)
        return state["img"], st

    def drain():
        dec_stream.synchronize()

    def timed(mode, steps, warmup):
        for i in range(warmup):
            step(mode, i)
        D.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        tot = {k: 0 for k in D.COUNTER_KEYS}
        for i in range(steps):
            _, st = step(mode, 1000 + i)
            for k in tot: tot[k] += int(st.get(k, 0))
        agg = D.gather_counters(tot, dev)                            # the one collective: counters only
        drain(); torch.cuda.synchronize(); D.barrier()
        dt = D.max_over_ranks(time.perf_counter() - t0, dev)
        return dt, agg

    dt, agg = timed(args.mode, args.steps, args.warmup)
    value = agg["images"] / dt
    log(f"timed region: {dt:.3f}s for {args.steps} steps -> {value:.2f} images/s")

    # no-decode rate (sampler only), same mode, shorter
    def timed_nodecode(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            smp.spec_decode(labels, CFG, args.gamma, 900, 0.96, E.Noise("device", 2000 + i, image_offset=lo), thr=thr[args.mode], run_ahead=not args.no_run_ahead)
        torch.cuda.synchronize()
        return D.max_over_ranks(time.perf_counter() - t0, dev)
    nd_steps = max(2, args.steps // 2)
    dt_nd = timed_nodecode(nd_steps)
    log(f"no-decode: {B * world * nd_steps / dt_nd:.2f} images/s")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        decode(fh_buf[0])
    torch.cuda.synchronize(); dec_ms = (time.perf_counter() - t0) / 5 * 1e3
    log(f"decode alone: {dec_ms:.2f} ms per batch of {B} ({'PyTorch/MIOpen' if args.torch_decode else 'HIP decoder'})")

    def decoder_flops(ch=160, mult=(1, 1, 2, 2, 4), nrb=2, z=32, h0=lad.HW):
        """Multiply-adds x 2 of decoder(post_quant_conv(f_hat)) per image (basic_vae.py:163-226), direct 3x3 convolutions."""
        conv = lambda cin, cout, hw, k: 2.0 * hw * hw * cin * cout * k * k
        ctop = ch * mult[-1]
        res = lambda cin, cout, hw: conv(cin, cout, hw, 3) + conv(cout, cout, hw, 3) + (conv(cin, cout, hw, 1) if cin != cout else 0.0)
        att = lambda c, hw: conv(c, 3 * c, hw, 1) + conv(c, c, hw, 1) + 4.0 * (hw * hw) ** 2 * c
        f = conv(z, z, h0, 3) + conv(z, ctop, h0, 3) + 2 * res(ctop, ctop, h0) + att(ctop, h0)
        cprev, hw = ctop, h0
        for lv in reversed(range(len(mult))):
            c = ch * mult[lv]
            for _ in range(nrb + 1):
                f += res(cprev, c, hw); cprev = c
                if lv == len(mult) - 1:
                    f += att(c, hw)
            if lv != 0:
                hw *= 2; f += conv(c, c, hw, 3)
        return f + conv(cprev, 3, hw, 3)
    dec_tflops = decoder_flops() * B / (dec_ms * 1e-3) / 1e12

    extra = {}
    if not args.no_extra_modes and rank == 0 and world == 1 and args.config == "P1":
        # Two sampler pipelines per GPU (an extra, NOT `value`): a second set of model objects driven by a second host thread on its
        # own HIP streams; two B=8 batches are in flight, so one batch's launch-bound early stages run under the other's GEMMs.
        import threading
        dc2 = E.ModelCtx(sd_d, args.depth_draft, pns, B, 1, dev, gemm_mode=args.gemm_mode)
        tc2 = E.ModelCtx(sd_t, args.depth_target, pns, B, max(args.gamma, 1), dev, gemm_mode=args.gemm_mode)
        qc2 = E.QuantCtx(sd_v, pns, B, dev)
        pipes = [dict(smp=smp, dec=E.VaeCtx(sd_v, B, dev)), dict(smp=E.Sampler(tc2, qc2, dc2), dec=E.VaeCtx(sd_v, B, dev))]
        for pi, pp in enumerate(pipes):
            pp["s"], pp["d"] = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            pp["fh"] = [torch.zeros(B, 32, lad.HW, lad.HW, device=dev) for _ in range(2)]

        def pipe_run(pp, pi, steps):
            done = [None, None]
            with torch.cuda.stream(pp["s"]):
                for i in range(steps):
                    res = pp["smp"].spec_decode(labels, CFG, args.gamma, 900, 0.96, E.Noise("device", 3000 + 2 * i + pi, image_offset=lo), thr=thr[args.mode])
                    j = i & 1
                    if done[j] is not None:
                        pp["s"].wait_event(done[j])
                    pp["fh"][j].copy_(res.f_hat)
                    ready = torch.cuda.Event(); ready.record(pp["s"])
                    pp["d"].wait_event(ready)
                    with torch.cuda.stream(pp["d"]):
                        pp["dec"].decode(pp["fh"][j]).add_(1).mul_(0.5)
                        done[j] = torch.cuda.Event(); done[j].record(pp["d"])
            pp["d"].synchronize(); pp["s"].synchronize()

        def two_pipes(steps):
            ts = [threading.Thread(target=pipe_run, args=(pp, pi, steps)) for pi, pp in enumerate(pipes)]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for t in ts: t.start()
            for t in ts: t.join()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        two_pipes(2)
        n2 = max(2, args.steps // 2)
        d2p = two_pipes(n2)
        log(f"two pipelines: {2 * n2 * B / d2p:.2f} images/s ({2 * n2} steps of B={B}, 2 in flight)")
        extra["two_pipelines_per_gpu"] = dict(images_per_s=2 * n2 * B / d2p, note="2 host threads x (sampler stream + decode stream); not the headline value")
        for o in (dc2, tc2, qc2, pipes[0]["dec"], pipes[1]["dec"]):
            o.close()
    if not args.no_extra_modes:
        for m in ("accept_all", "reject_all", "natural"):
            if m == args.mode:
                continue
            d2, a2 = timed(m, max(2, args.steps // 3), 1)
            log(f"mode {m}: {a2['images'] / d2:.2f} images/s")
            extra[m] = dict(images_per_s=a2["images"] / d2, mean_accepted_tokens_per_step=a2["mean_accepted_tokens_per_step"],
                            target_calls_per_image_batch=a2["target_calls"] / max(1, a2["images"] // B))
        # ---- what a caller of the reference API gets, and what speculation has to beat --------------------------------------------------
        n3 = max(2, args.steps // 3)
        def rate(fn, steps=n3):
            fn(0); drain(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(steps):
                fn(100 + i)
            drain(); torch.cuda.synchronize()
            return D.max_over_ranks(time.perf_counter() - t0, dev), steps
        # (a) plain AR of the TARGET on the GPU (VAR.autoregressive_infer_cfg, 10 target calls, no draft), decode overlapped as in `value`
        smp_t = E.Sampler(tc, qc)
        def plain_step(seed):
            res = smp_t.plain_ar(labels, CFG, 900, 0.96, E.Noise("device", seed, image_offset=lo))
            j = state["i"] & 1; state["i"] += 1
            if dec_done[j] is not None:
                main_stream.wait_event(dec_done[j])
            fh_buf[j].copy_(res.f_hat)
            ready = torch.cuda.Event(); ready.record(main_stream); dec_stream.wait_event(ready)
            with torch.cuda.stream(dec_stream):
                decode(fh_buf[j]).add_(1).mul_(0.5)
                dec_done[j] = torch.cuda.Event(); dec_done[j].record(dec_stream)
        dtp, n = rate(plain_step)
        extra["plain_target_ar"] = dict(images_per_s=B * world * n / dtp, note=f"GPU autoregressive_infer_cfg of the d{args.depth_target} target alone (10 target calls, no draft), decode overlapped")
        log(f"plain target AR: {extra['plain_target_ar']['images_per_s']:.2f} images/s")
        # (b) the PUBLIC API: sdvar_amd.SDVAR.sdvar_autoregressive_infer_cfg_parallel_v1 on module objects holding the same weights: one
        # synchronous call per batch (sampling loop with run-ahead, then the decode on the same stream), as a reference user would call it
        if rank == 0 and world == 1:
            from sdvar_amd.var import SDVAR, VAR
            mods = []
            for depth, sdx in ((args.depth_draft, sd_d), (args.depth_target, sd_t)):
                with torch.device("meta"):
                    m = VAR(vae_local=vae, depth=depth, embed_dim=64 * depth, num_heads=depth, attn_l2_norm=True, patch_nums=pns)
                m = m.to_empty(device=dev)
                m.load_state_dict(sdx, strict=False)                  # buffers (lvl_1L, mask) are rebuilt below; parameters come from the bench's tensors
                m.lvl_1L.copy_(torch.cat([torch.full((n_,), i_, dtype=torch.int64) for i_, n_ in enumerate(lad.lens)]).view(1, lad.L))
                m.rng = torch.Generator(device="cpu")
                mods.append(m)
            sdv = SDVAR(mods[0], mods[1])
            sdv.match_threshold = thr[args.mode]
            def api_step(seed):
                state["img"] = sdv.sdvar_autoregressive_infer_cfg_parallel_v1(B=B, label_B=labels, g_seed=seed, cfg=CFG, gamma=args.gamma, top_k=900, top_p=0.96)
            dta, n = rate(api_step)
            extra["api_path"] = dict(images_per_s=B * n / dta, note="SDVAR.sdvar_autoregressive_infer_cfg_parallel_v1 (the reference's entry point), one synchronous call per batch incl. decode; "
                                     "`value` differs only by overlapping the decode of batch i with the sampling of batch i+1")
            log(f"public API path: {extra['api_path']['images_per_s']:.2f} images/s")
            for m in mods:
                m.invalidate_engine()
            del sdv, mods
        # (c) speculation against not speculating, per acceptance mode (>= 1 means the draft -> verify loop pays on this hardware at this batch)
        plain = extra["plain_target_ar"]["images_per_s"]
        extra["speculation_gain"] = dict({m: extra[m]["images_per_s"] / plain for m in ("accept_all", "reject_all", "natural") if m in extra}, **{args.mode: value / plain},
                                         note="images/s of the speculative sampler / images/s of plain_target_ar, same GPU, same decode overlap; with a d12 draft in front of a d16 target "
                                              "(43 % of its flops) and the large stages matrix-pipe bound at B=8, chunked verification cannot recover the draft's cost: see DESIGN.md section 5")
        if rank == 0 and world == 1:
            # (d) SURVEY App. C.2: draft == target weights, top_k = 1 - the one random-weight setting whose NATURAL acceptance is not degenerate (I1: everything accepted)
            dc_same = E.ModelCtx(sd_t, args.depth_target, pns, B, 1, dev, gemm_mode=args.gemm_mode, kv_fp16=conf["kv_fp16"])
            smp_same = E.Sampler(tc, qc, dc_same)
            def same_step(seed):
                res = smp_same.spec_decode(labels, CFG, args.gamma, 1, 0.0, E.Noise("device", seed, image_offset=lo), thr=0.5)
                state["st"] = dict(res.stats)
                j = state["i"] & 1; state["i"] += 1
                if dec_done[j] is not None:
                    main_stream.wait_event(dec_done[j])
                fh_buf[j].copy_(res.f_hat)
                ready = torch.cuda.Event(); ready.record(main_stream); dec_stream.wait_event(ready)
                with torch.cuda.stream(dec_stream):
                    decode(fh_buf[j]).add_(1).mul_(0.5)
                    dec_done[j] = torch.cuda.Event(); dec_done[j].record(dec_stream)
            dts, n = rate(same_step)
            stx = state["st"]
            extra["identical_draft_top1"] = dict(images_per_s=B * n / dts, mean_accepted_tokens_per_step=stx["accepted_tokens"] / max(1, stx["target_calls"]), target_calls=stx["target_calls"],
                                                 forced_accepts=stx["forced_accepts"], note=f"draft = the d{args.depth_target} target's own weights, top_k=1 (greedy), natural threshold 0.5: every stage is accepted (invariant I1)")
            log(f"identical draft, top_k=1: {extra['identical_draft_top1']['images_per_s']:.2f} images/s, {extra['identical_draft_top1']['mean_accepted_tokens_per_step']:.1f} accepted tokens/step")
            # (d') where speculation pays: the speculative sampler against plain target AR at B = 1, 2, 4, 8 (the reference's own harness measures at B = 4,
            # sdvar_colab_test.py:129,193,267-329, and expects 1.3-1.7x, PROJECT_STATUS_SUMMARY.md:33).  Sampler only (the decode is the same work on both sides and is
            # left out), same model objects, gamma = 2 and 3 (a target context with room for 3-stage chunks shares nothing but the weights' source tensors).
            tc3 = E.ModelCtx(sd_t, args.depth_target, pns, B, 3, dev, gemm_mode=args.gemm_mode, kv_fp16=conf["kv_fp16"])
            smp3, smp_same3 = E.Sampler(tc3, qc, dc), E.Sampler(tc3, qc, dc_same)
            byb = {}
            for Bx in [b_ for b_ in (1, 2, 4, 8, 16) if b_ <= B]:
                lb = labels[:Bx].contiguous()
                def srate(fn, steps=max(3, args.steps // 3)):
                    fn(0); torch.cuda.synchronize(); t0 = time.perf_counter()
                    for i in range(steps):
                        fn(200 + i)
                    torch.cuda.synchronize()
                    return Bx * steps / (time.perf_counter() - t0)
                nz = lambda sd_: E.Noise("device", sd_, image_offset=lo)
                row = dict(plain_target_ar=srate(lambda sd_: smp_t.plain_ar(lb, CFG, 900, 0.96, nz(sd_))))
                for gmm, sa, ss in ((2, smp, smp_same), (3, smp3, smp_same3)):
                    row[f"accept_all_gamma{gmm}"] = srate(lambda sd_: sa.spec_decode(lb, CFG, gmm, 900, 0.96, nz(sd_), thr=0.0, run_ahead=not args.no_run_ahead))
                    row[f"identical_draft_top1_gamma{gmm}"] = srate(lambda sd_: ss.spec_decode(lb, CFG, gmm, 1, 0.0, nz(sd_), thr=0.5))
                row["gain"] = {k: v / row["plain_target_ar"] for k, v in row.items() if k != "plain_target_ar"}
                byb[str(Bx)] = row
                log(f"B={Bx}: plain {row['plain_target_ar']:.1f} images/s; gain " + ", ".join(f"{k} {v:.2f}" for k, v in row["gain"].items()))
            best = max(((g_, k, bx) for bx, r in byb.items() for k, g_ in r["gain"].items()), key=lambda t_: t_[0])
            extra["speculation_gain_by_batch"] = dict(rows=byb, best=dict(gain=best[0], mode=best[1], batch=int(best[2])),
                note=f"sampler-only images/s (no decode on either side) of d{args.depth_draft} -> d{args.depth_target} speculation / plain d{args.depth_target} AR at each batch size; accept_all = every "
                     f"round accepted at threshold 0 with the d{args.depth_draft} draft (the loop's best case), identical_draft_top1 = a d{args.depth_target} draft that the target accepts naturally; "
                     "gain >= 1 means speculation pays at that batch")
            tc3.close(); del smp3, smp_same3
            dc_same.close(); del smp_same
            # (e) the exact split-operand mode (bf16x3: no range limit on the activations; f16x2 saturates at +-65504), same loop
            if tc.gemm_mode != "bf16x3":
                dcx = E.ModelCtx(sd_d, args.depth_draft, pns, B, 1, dev, gemm_mode="bf16x3", kv_fp16=conf["kv_fp16"])
                tcx = E.ModelCtx(sd_t, args.depth_target, pns, B, max(args.gamma, 1), dev, gemm_mode="bf16x3", kv_fp16=conf["kv_fp16"])
                smpx = E.Sampler(tcx, qc, dcx)
                def exact_step(seed):
                    res = smpx.spec_decode(labels, CFG, args.gamma, 900, 0.96, E.Noise("device", seed, image_offset=lo), thr=thr[args.mode])
                    j = state["i"] & 1; state["i"] += 1
                    if dec_done[j] is not None:
                        main_stream.wait_event(dec_done[j])
                    fh_buf[j].copy_(res.f_hat)
                    ready = torch.cuda.Event(); ready.record(main_stream); dec_stream.wait_event(ready)
                    with torch.cuda.stream(dec_stream):
                        decode(fh_buf[j]).add_(1).mul_(0.5)
                        dec_done[j] = torch.cuda.Event(); dec_done[j].record(dec_stream)
                dtx, n = rate(exact_step)
                extra["bf16x3"] = dict(images_per_s=B * n / dtx, note="the same loop with gemm_mode bf16x3 (operands split EXACTLY into 3 bf16 planes, 6 MFMA products per fp32 product): "
                                                                      "the mode without f16x2's activation range limit")
                log(f"bf16x3 (exact split) mode: {extra['bf16x3']['images_per_s']:.2f} images/s")
                dcx.close(); tcx.close(); del smpx

    # ---- roofline leg: HIP events around every launch of one step (launch stream = torch's current stream)
    drain(); torch.cuda.synchronize()
    E.prof_enable(True)
    step(args.mode, 4242, run_ahead=False)       # per-kernel durations with each kernel alone on the GPU (no draft/verify overlap)
    drain(); torch.cuda.synchronize()
    prof = E.prof_collect()
    E.prof_enable(False)
    classes = {k: v for k, v in prof.items() if v["launches"]}
    dom = max(classes, key=lambda k: classes[k]["ms"])
    pmc = {}
    try:   # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        pass

    def roof(name):
        c = classes[name]
        sec = c["ms"] * 1e-3
        if name in ("gemm", "gemm_small"):
            ach = c["flops"] / sec / 1e12            # algorithmic 2*M*N*K per launch / measured duration
            if tc.gemm_mode == "bf16x3":
                peak = PEAK_BF16_MFMA_TFLOPS / 6.0   # 6 bf16 MFMA products per algorithmic fp32 product
                kname, note = "gemm_bf16x3_{v3,v2,}_kernel", "peak = dense bf16 MFMA peak / 6 (split-operand products per fp32 product)"
            elif tc.gemm_mode == "f16x2":
                peak = PEAK_BF16_MFMA_TFLOPS / 3.0   # 3 fp16 MFMA products per algorithmic fp32 product (f16 MFMA = bf16 MFMA rate)
                kname, note = ("gemm_f16x2_{v5,v3,v2}_kernel" if name == "gemm" else "gemm_f16x2_{small_pp,small,skinny,v2}_kernel"), "peak = dense f16 MFMA peak / 3 (split-operand products per fp32 product)"
            else:
                peak, kname, note = PEAK_F32_MFMA_TFLOPS, "gemm_f32_nt_kernel", "peak = fp32-in MFMA"
            regime = "launches with M >= 1024 rows (stages 6-9, large verify chunks): matrix-pipe regime" if name == "gemm" else \
                     "launches with M < 1024 rows (stages 0-5, adaLN hoist): weight-streaming / launch-latency regime, priced against the same MFMA peak"
            return dict(kernel=kname, bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, note=note, regime=regime,
                        vs_fp32_mfma_peak=ach / PEAK_F32_MFMA_TFLOPS,
                        traffic=pmc.get(name, {}).get("hbm_bytes_per_launch"), algorithmic_bytes=c["bytes"] / c["launches"],
                        launches=c["launches"], avg_us=c["ms"] * 1e3 / c["launches"],
                        weight_stream_gbs=(c["bytes"] / (c["ms"] * 1e-3) / 1e9))
        ach = c["bytes"] / sec / 1e9
        return dict(kernel=name, bound="hbm", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                    traffic=pmc.get(name, {}).get("hbm_bytes_per_launch"), algorithmic_bytes=c["bytes"] / c["launches"],
                    launches=c["launches"], avg_us=c["ms"] * 1e3 / c["launches"], tflops=c["flops"] / sec / 1e12)
    roofline = roof(dom)
    roofline_gemm_small = roof("gemm_small") if "gemm_small" in classes and dom != "gemm_small" else None
    if "gemm" in classes and "gemm_small" in classes:          # the whole GEMM class as round 2 reported it (one average over every launch of the step)
        ca, cb = classes["gemm"], classes["gemm_small"]
        all_tf = (ca["flops"] + cb["flops"]) / ((ca["ms"] + cb["ms"]) * 1e-3) / 1e12
        roofline["all_gemm_launches"] = dict(achieved=all_tf, frac=all_tf / roofline["peak"], launches=ca["launches"] + cb["launches"], ms=ca["ms"] + cb["ms"])
    # verify-attention by regime: launches with more than 36 queries per (row, head) are matrix/vector-pipe bound (fp32-accurate arithmetic:
    # 6 bf16 MFMA products per fp32 product), the short stages stream the cache once and are HBM / latency bound
    roofline_attn = None
    if "attention" in classes:
        c = classes["attention"]
        roofline_attn = roof("attention")
        roofline_attn["regime"] = "l > 36 queries per (row, head): matrix-pipe bound"
        # matrix-core products per algorithmic fp32 product in the attention kernel of this configuration: bf16x3 planes 6, f16x2 planes 3, one fp16 plane
        # (fp16 KV cache) 2, fp32 MFMA (gemm mode f32): the fp32 MFMA peak
        if tc.gemm_mode == "f32":
            apeak, anote = PEAK_F32_MFMA_TFLOPS, "fp32-in MFMA"
        else:
            nprod = 2 if conf["kv_fp16"] else (3 if tc.gemm_mode == "f16x2" else 6)
            apeak, anote = PEAK_BF16_MFMA_TFLOPS / nprod, f"dense 16-bit MFMA peak / {nprod} plane products per fp32 product"
        roofline_attn["mfma"] = dict(achieved=c["flops"] / (c["ms"] * 1e-3) / 1e12, peak=apeak, unit="TFLOP/s", note=anote,
                                     frac=c["flops"] / (c["ms"] * 1e-3) / 1e12 / apeak)
        # measured HBM traffic is per launch over ALL attention launches of the step (both regimes): compare like with like
        n_all = c["launches"] + (classes["attention_small"]["launches"] if "attention_small" in classes else 0)
        b_all = c["bytes"] + (classes["attention_small"]["bytes"] if "attention_small" in classes else 0.0)
        roofline_attn["algorithmic_bytes_all_attention_launches"] = b_all / n_all
    roofline_attn_small = None
    if "attention_small" in classes:
        roofline_attn_small = roof("attention_small")
        roofline_attn_small["regime"] = "l <= 36 queries per (row, head) (stages 0-5): HBM / launch-latency bound"
    # ---- verify-attention at config P4's shape (d30, 512^2, fp16 KV cache: ONE fp16 plane per operand, the last stage: l = 1024 queries over K = 2240 keys, R = 16 rows,
    # H = 30 heads), timed alone with HIP events over four rotating caches (keys from HBM, not from the Infinity Cache): the fraction round 3 reported as 0.11
    def p4_attention(iters=20):
        import ctypes as Cc, math
        lib = E.load_library(); P_ = lambda t: Cc.c_void_p(t.data_ptr()); st_ = Cc.c_void_p(torch.cuda.current_stream().cuda_stream)
        R_, H_, l_, K_ = 16, 30, 1024, 2240
        Lp_ = (K_ + 63) // 64 * 64
        sm_ = torch.full((H_,), math.log(4.0), device=dev)
        caches = []
        for _ in range(4):
            kc = torch.zeros(R_, H_, 1, Lp_, 64, device=dev, dtype=torch.int16); vc = torch.zeros_like(kc)
            for n_, pos0 in ((K_ - l_, 0), (l_, K_ - l_)):
                qkv = torch.randn(R_ * n_, 3 * 64 * H_, device=dev); qo = torch.zeros(R_, H_, n_, 64, device=dev)
                E._check(lib.sdvar_op_qk_norm_append(P_(qkv), P_(sm_), P_(qo), P_(kc), P_(vc), 4, R_, n_, H_, Lp_, pos0, st_))
            caches.append((kc, vc))
        outp = torch.empty(2, H_ * 2, R_ * l_, 32, dtype=torch.int16, device=dev)
        qb, vs = (Cc.c_int32 * 1)(0), (Cc.c_int32 * 1)(K_)
        def run(i):
            kc, vc = caches[i % 4]
            E._check(lib.sdvar_op_attention(P_(qo), P_(kc), P_(vc), 4, None, P_(outp), R_ * l_ * H_ * 64, 2, R_, H_, l_, Lp_, K_, 1, qb, vs, st_))
        for i in range(3): run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): run(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        by = R_ * H_ * 64.0 * (2 * K_ * 2 + 2 * l_ * 4)                      # fp16 K and V rows once, fp32 q in, fp32-equivalent planes out
        fl = 4.0 * R_ * H_ * l_ * K_ * 64
        del caches
        return dict(kernel="attention_f16x2_pp_kernel<1>", shape=f"R={R_} H={H_} l={l_} K={K_}, one fp16 plane per cache operand", bound="hbm", avg_us=us, achieved=by / us * 1e-3, peak=PEAK_HBM_GBS,
                    unit="GB/s", frac=by / us * 1e-3 / PEAK_HBM_GBS, algorithmic_bytes=by, tflops=fl / us * 1e-6,
                    mfma=dict(achieved=fl / us * 1e-6, peak=PEAK_BF16_MFMA_TFLOPS / 2.0, frac=fl / us * 1e-6 / (PEAK_BF16_MFMA_TFLOPS / 2.0), note="2 plane products per fp32 product (exact fp16 K / V, two-plane Q / P)"),
                    note="config P4's largest verify-attention launch, alone on the GPU; 51 flop per algorithmic byte: matrix / vector pipe bound, priced against HBM as the north-star asks")
    roofline_attn_p4 = None
    if rank == 0 and tc.gemm_mode == "f16x2":
        try:
            roofline_attn_p4 = p4_attention()
        except Exception as ex:          # never lose the bench line to the extra
            roofline_attn_p4 = dict(error=str(ex))
    class_ms = {k: round(v["ms"], 3) for k, v in classes.items()}
    log(f"profiled step: {class_ms}")

    out = {
        "metric": f"images/s (+ mean accepted tokens/step), {conf['name']}, per GPU",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16x3": "f32 (GEMM operands split exactly into 3 bf16 planes, 6 bf16 MFMA products, fp32 accumulate)",
                  "f16x2": "f32 (GEMM operands split into 2 fp16 planes to 2^-22, 3 fp16 MFMA products, fp32 accumulate)"}[tc.gemm_mode], "data": "synthetic (random-init weights, labels arange(B)%1000, device Philox noise)",
        "config": {"workload": f"BASELINE.json {args.config}: VAR-d{args.depth_target} {conf['ladder']}^2 B={B}/GPU, d{args.depth_draft} draft + d{args.depth_target} verify, gamma={args.gamma}, "
                               f"cfg={CFG} top_k=900 top_p=0.96, {'fp16 KV cache, ' if conf['kv_fp16'] else ''}acceptance={args.mode}, incl. VQVAE decode ({'serial' if args.serial_decode else 'overlapped with the next batch on a 2nd stream'}); "
                               f"verifier {'in lock-step with' if args.no_run_ahead else 'one round behind'} the draft once gamma = 1", "parallelism": f"{world} independent batch shards"},
        "value_bf16x3": extra.get("bf16x3", {}).get("images_per_s"),          # the same loop in the exact-split mode (no activation range limit): quote it beside `value`
        "mean_accepted_tokens_per_step": agg["mean_accepted_tokens_per_step"],
        "target_calls": agg["target_calls"], "draft_stage_calls": agg["draft_stage_calls"], "forced_accepts": agg["forced_accepts"],
        "per_rank_counters": dict(keys=list(D.COUNTER_KEYS), rows=agg["per_rank"], note="all-gathered over the process group (RCCL on the GPU box): one row per rank"),
        "images_per_s_no_decode": B * world * nd_steps / dt_nd,
        "decode_ms_per_batch": dec_ms, "decoder_tflops_algorithmic": dec_tflops, "decoder": "pytorch-miopen" if args.torch_decode else "hip (csrc/conv.hip, csrc/vae.hip)",
        "modes": extra, "roofline_note": "roofline / kernel_class_ms_per_step come from one step with every kernel alone on the GPU (no draft/verify or decode overlap)", "roofline": roofline, "roofline_gemm_small_m": roofline_gemm_small, "roofline_verify_attention": roofline_attn, "roofline_verify_attention_short_stages": roofline_attn_small, "roofline_verify_attention_P4_fp16_kv": roofline_attn_p4, "kernel_class_ms_per_step": class_ms,
    }

    # ---- CPU baseline (rank 0, N=1): the oracle's plain AR of the TARGET model on the host cores
    if world == 1 and not args.no_cpu_baseline:
        from oracle import var_oracle as orc
        cores = host_cores()
        torch.set_num_threads(cores)
        log(f"cpu baseline on {cores} threads ...")
        sd_cpu = {k: v.cpu() for k, v in sd_t.items()}
        model, quant = orc.OracleVAR(sd_cpu, args.depth_target, pns, kv_fp16=conf["kv_fp16"]), orc.OracleQuant(sd_v, pns)
        g = torch.Generator(); g.manual_seed(0)
        Bc = B if args.config == "P1" else 1                      # bounded sample: the large configurations time one image per call
        n_timed = 3 if args.config == "P1" else 1
        t_ar, t_all = [], []
        for it in range(1 + n_timed):                             # warm-up, then timed calls, median
            t0 = time.perf_counter()
            tr = orc.plain_ar(model, quant, labels.cpu()[:Bc], CFG, 900, 0.96, orc.torch_noise(g), keep=False)
            ta = time.perf_counter() - t0
            orc.decode_image(sd_v, tr.f_hat)
            tb = time.perf_counter() - t0
            log(f"cpu baseline call {it}{' (warm-up)' if it == 0 else ''}: {tb:.1f}s ({ta:.1f}s without decode)")
            if it > 0 or n_timed == 0:
                t_ar.append(ta); t_all.append(tb)
        med = lambda v: sorted(v)[len(v) // 2]
        out["cpu_baseline"] = {"value": Bc / med(t_all), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"plain autoregressive_infer_cfg of the CPU oracle (target d{args.depth_target} only, no speculation), B={Bc}, incl. decode: 1 warm-up + "
                                         f"{n_timed} timed call(s), median {med(t_all):.1f}s ({med(t_ar):.1f}s without decode)", "images_per_s_no_decode": Bc / med(t_ar)}
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
