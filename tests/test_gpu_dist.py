"""GPU, world_size 2: the N > 1 host path with the HIP sampler on every rank (tests/test_dist_gloo.py runs the same sdvar_amd.dist calls with the CPU
oracle as the per-rank sampler).  Both ranks share the one GPU of the test box, so the process group is gloo (RCCL needs one device per rank); on the
8-GPU node bench.py runs the identical code path over RCCL (backend "nccl").  Checked:
  * I5 shard invariance on the device: the two shards' ids are the ids of the same images in ONE process sampling the whole batch;
  * the one collective of the design (all-gather of the per-rank counters) and the max-over-ranks timing reduction;
  * accept_scope="global" (one all-reduce of the match counts per round): both ranks take the whole-batch decision - ids, rounds and summed counters
    equal the one-process whole-batch run at the natural threshold, also with token-level acceptance (per-rank statistics must stay per rank)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import state_dicts

pytestmark = pytest.mark.gpu
PNS = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
B_TOTAL = 4
SEED = 9


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _sampler(dev, B):
    from sdvar_amd import engine as E
    sd_d, sd_v = state_dicts(2, PNS); sd_t, _ = state_dicts(4, PNS)
    dc, tc, qc = E.ModelCtx(sd_d, 2, PNS, B, 1, dev), E.ModelCtx(sd_t, 4, PNS, B, 2, dev), E.QuantCtx(sd_v, PNS, B, dev)
    return E.Sampler(tc, qc, dc), (dc, tc, qc)


def _run(smp, dev, lo, hi, thr, scope, token_level=False):
    from sdvar_amd import engine as E
    labels = ((torch.arange(lo, hi) * 37) % 1000).to(dev)
    rule = E.MatchRule("top1", token_level=True) if token_level else None
    res = smp.spec_decode(labels, 1.5, 2, 900, 0.96, E.Noise("host", SEED, image_offset=lo), thr=thr, accept_scope=scope, match=rule)
    st = {k: v for k, v in res.stats.items()}
    return res.ids.cpu().numpy().copy(), st


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SDVAR_DIST_BACKEND="gloo")
    torch.set_grad_enabled(False)
    from sdvar_amd import dist as D
    r, w, local = D.init_from_env("cuda")
    assert (r, w) == (rank, world)
    dev = torch.device("cuda", local)
    lo, hi = D.shard_range(B_TOTAL, r, w)
    smp, objs = _sampler(dev, hi - lo)
    out = {}
    ids, st = _run(smp, dev, lo, hi, 0.0, "shard")                               # accept_all: per-image results, no coupling between ranks
    st["images"] = hi - lo
    agg = D.gather_counters(st, dev)
    out.update(ids_shard=ids, images=agg["images"], accepted=agg["accepted_tokens"], target_calls=agg["target_calls"], per_rank=np.array(agg["per_rank"]),
               tmax=D.max_over_ranks(float(rank + 1), dev))
    for name, tl in (("glob", False), ("globtl", True)):
        ids, st = _run(smp, dev, lo, hi, 0.5, "global", token_level=tl)          # natural threshold, batch-wide decision across the ranks
        st["images"] = hi - lo
        agg = D.gather_counters(st, dev)
        out[f"ids_{name}"] = ids
        out[f"nacc_{name}"] = np.array([rr["n_accept"] for rr in st["rounds"]])
        out[f"matched_{name}"] = np.array([sum(rr["matched"]) for rr in st["rounds"]])
        out[f"accepted_{name}"] = agg["accepted_tokens"]; out[f"calls_{name}"] = agg["target_calls"]
        out[f"corrected_{name}"] = int(st.get("corrected_tokens", 0))
    D.barrier()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, **out)
    for o in objs:
        o.close()
    torch.distributed.destroy_process_group()


def test_two_ranks_hip_sampler_shard_and_global_scope(dev, tmp_path):
    torch.set_grad_enabled(False)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 2, 2, 4)
    # the collective: identical totals on both ranks = the sums of the per-rank rows
    for k in ("images", "accepted", "target_calls"):
        assert int(r0[k]) == int(r1[k])
    assert int(r0["images"]) == B_TOTAL and np.array_equal(r0["per_rank"], r1["per_rank"]) and r0["per_rank"].shape[0] == 2
    assert float(r0["tmax"]) == float(r1["tmax"]) == 2.0
    L = sum(p * p for p in PNS)
    assert int(r0["accepted"]) == 2 * L and int(r0["target_calls"]) == 2 * 5            # accept_all on both shards: every stage accepted, 5 chunk verifies each
    # one process, whole batch, same global-index noise
    smp, objs = _sampler(dev, B_TOTAL)
    ids_full, _ = _run(smp, dev, 0, B_TOTAL, 0.0, "shard")
    assert np.array_equal(np.concatenate([r0["ids_shard"], r1["ids_shard"]], 0), ids_full)        # I5 on the device
    for name, tl in (("glob", False), ("globtl", True)):
        ids_g, st_g = _run(smp, dev, 0, B_TOTAL, 0.5, "shard", token_level=tl)                   # one process: shard scope IS the whole batch
        assert np.array_equal(np.concatenate([r0[f"ids_{name}"], r1[f"ids_{name}"]], 0), ids_g), name
        want_nacc = np.array([rr["n_accept"] for rr in st_g["rounds"]])
        assert np.array_equal(r0[f"nacc_{name}"], want_nacc) and np.array_equal(r1[f"nacc_{name}"], want_nacc), name      # the same decision everywhere
        # per-rank match counts add up to the whole batch's (they are NOT the all-reduced numbers on each rank)
        assert np.array_equal(r0[f"matched_{name}"] + r1[f"matched_{name}"], np.array([sum(rr["matched"]) for rr in st_g["rounds"]])), name
        assert int(r0[f"calls_{name}"]) == 2 * st_g["target_calls"]
        assert int(r0[f"accepted_{name}"]) == int(r1[f"accepted_{name}"])
        if tl:
            # corrected tokens are counted per token with the rank's own batch: the shards' counts add up to the whole batch's and are never negative
            assert int(r0[f"corrected_{name}"]) >= 0 and int(r1[f"corrected_{name}"]) >= 0
            assert int(r0[f"corrected_{name}"]) + int(r1[f"corrected_{name}"]) == int(st_g.get("corrected_tokens", 0))
    for o in objs:
        o.close()
