"""CPU, world_size 2 (gloo): the N>1 host path - contiguous batch shards, noise keyed by the GLOBAL image index, and
the one collective of the design (all-gather of per-rank counters).  The per-rank sampler here is the CPU oracle (this
is a test); on the GPU box the same sdvar_amd.dist calls run over RCCL (backend 'nccl') in bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import state_dicts

PNS = (1, 2, 3, 4)
B_TOTAL = 4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_shard(lo, hi, thr, gamma=2):
    from oracle import var_oracle as orc
    from sdvar_amd.noise import exponential_noise
    sd_d, sd_v = state_dicts(2, PNS); sd_t, _ = state_dicts(4, PNS)
    od, ot, oq = orc.OracleVAR(sd_d, 2, PNS), orc.OracleVAR(sd_t, 4, PNS), orc.OracleQuant(sd_v, PNS)
    labels = (torch.arange(lo, hi) * 37) % 1000
    noise = orc.array_noise(lambda d, B, l, V: exponential_noise(9, d, B, l, V, image_offset=lo))
    return orc.spec_decode(od, ot, oq, labels, 1.5, gamma, 900, 0.96, noise, thr=thr)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from sdvar_amd import dist as D
    r, w, _ = D.init_from_env("cpu")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_range(B_TOTAL, r, w)
    tr = _run_shard(lo, hi, thr=0.0)
    st = dict(tr.stats); st["images"] = hi - lo
    D.barrier()
    agg = D.gather_counters(st, "cpu")
    t = D.max_over_ranks(float(rank + 1), "cpu")
    # accept_scope="global": rank 0 matched 3/8 and 8/8, rank 1 5/8 and 0/8 -> global rates 0.5 and 0.5 -> both accepted everywhere
    n_glob, gm, gt = D.global_accept([3, 8] if rank == 0 else [5, 0], [8, 8], 0.5, "cpu")
    assert (n_glob, list(gm), list(gt)) == (2, [8, 8], [16, 16]), (n_glob, gm, gt)
    assert D.leading_accepted([3, 8] if rank == 0 else [5, 0], [8, 8], 0.5) == (0 if rank == 0 else 1)        # the shard-scope decisions differ
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ids=torch.cat(tr.ids, 1).numpy(), lo=lo, hi=hi, images=agg["images"], accepted=agg["accepted_tokens"],
             target_calls=agg["target_calls"], per_rank=np.array(agg["per_rank"]), tmax=t, mean_acc=agg["mean_accepted_tokens_per_step"])
    torch.distributed.destroy_process_group()


def test_leading_accepted_float32_mean_semantics():
    from sdvar_amd.dist import leading_accepted
    assert leading_accepted([8, 4, 3], [8, 8, 8], 0.5) == 2
    assert leading_accepted([0], [5], 0.0) == 1 and leading_accepted([5], [5], 2.0) == 0
    # float32(1)/float32(3) = 0.3333333432674408 >= 0.33333334 (python float) -> accepted, as the reference's .item() compare
    assert leading_accepted([1], [3], 0.33333334) == 1 and leading_accepted([1], [3], 0.333333344) == 0


def test_shard_ranges():
    from sdvar_amd.dist import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_range(64, r, 8) for r in range(8)][-1] == (56, 64)
    assert shard_range(1, 1, 2) == (1, 1)                                   # an empty shard is legal


def test_two_rank_gloo_shards_match_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 2, 2, 4)
    # the collective: both ranks hold the same totals, which are the sums of the per-rank rows
    for k in ("images", "accepted", "target_calls"):
        assert int(r0[k]) == int(r1[k])
    assert int(r0["images"]) == B_TOTAL and np.array_equal(r0["per_rank"], r1["per_rank"])
    assert np.array_equal(r0["per_rank"].sum(0)[:3], [int(r0["images"]), int(r0["accepted"]), int(r0["target_calls"])])
    assert float(r0["tmax"]) == float(r1["tmax"]) == 2.0                    # max-over-ranks timing reduction
    L = sum(p * p for p in PNS)
    assert float(r0["mean_acc"]) == pytest.approx(2 * L / int(r0["target_calls"]))          # thr = 0: every stage accepted on both ranks
    # shard invariance (I5): with per-image decisions (accept_all) the sharded ids are the ids of the same images in ONE
    # process sampling the whole batch, because the noise is keyed by the global image index
    full = _run_shard(0, B_TOTAL, thr=0.0)
    ids_full = torch.cat(full.ids, 1).numpy()
    assert np.array_equal(np.concatenate([r0["ids"], r1["ids"]], 0), ids_full)
