"""The N > 1 path on CPU: world_size-2 gloo run of the per-tick detection exchange.  Each rank builds
its shard's packed union list with the oracle (same format zrk_compact writes) and the gathered,
rank-ordered result must equal the single-process list."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(n, R, seed):
    from zrk_modulation_amd import scenario as S
    ids, sp, vel, t0 = S.synthetic_targets(n, seed)
    radars = S.synthetic_radars(R)
    return sp, radars


def _masks(sp, radars):
    """vis_mask of every entity via the oracle (noise off, positions as given)."""
    from oracle import oracle as O
    L = O.lib()
    n = len(sp)
    pos = np.ascontiguousarray(sp.T).reshape(-1).copy()
    alive = np.ones(n, np.uint8)
    vis = np.zeros(n, np.uint32)
    arr = O.radar_array([(r["position"][0], r["position"][1], r["position"][2], r["max_distance"],
                          r["azimuth_start"], r["azimuth_range"], r["elevation_start"], r["elevation_range"])
                         for r in radars])
    L.zo_radar_phase_fused(n, n, O.dptr(pos), O.u8ptr(alive), len(radars), arr, 0, None, 0, 0, 0, O.u32ptr(vis), 1)
    return vis


def _pack(vis, gid0, cap):
    seen = np.nonzero(vis)[0]
    out = np.zeros(cap + 1, np.int64)
    out[0] = len(seen)
    out[1:1 + len(seen)] = ((seen + gid0).astype(np.int64) << 32) | vis[seen].astype(np.int64)
    return out


def _worker(rank, world, port, n, R, q, fmt="pairs"):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zrk_modulation_amd.exchange import DetectionExchange
    sp, radars = _scene(n, R, 77)
    shard = n // world
    lo, hi = rank * shard, (rank + 1) * shard
    vis = _masks(sp[lo:hi], radars)
    cap = shard
    if fmt == "pairs":
        ex = DetectionExchange(cap, torch.device("cpu"))
        ex.all_gather(torch.from_numpy(_pack(vis, lo, cap)))
    else:
        from zrk_modulation_amd.exchange import encode_union_bits, union_bits_words
        union_only = fmt == "union"                  # the bitmap alone: no room for masks, none sent
        words = union_bits_words(shard, R, 0 if union_only else cap)
        ex = DetectionExchange(words, torch.device("cpu"), fmt="bits", offsets=[g * shard for g in range(world)], R=R, union_only=union_only)
        ex.all_gather(torch.from_numpy(encode_union_bits(vis, R, words)))
    idx, mask = ex.merged()
    lists = [ex.radar_list(r).numpy() for r in range(R)] if mask is not None else None
    if rank == 0:
        q.put((idx.numpy(), None if mask is None else mask.numpy(), lists, ex.counts(), ex.overflowed()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fmt", ["pairs", "bits", "union"])
def test_two_rank_exchange_reproduces_single_process_order(fmt):
    n, R, world = 4000, 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, R, q, fmt)) for r in range(world)]
    for p in procs:
        p.start()
    idx, mask, lists, counts, overflow = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sp, radars = _scene(n, R, 77)
    vis = _masks(sp, radars)
    seen = np.nonzero(vis)[0]
    assert not overflow and sum(counts) == len(seen)
    assert np.array_equal(idx, seen)
    if fmt == "union":                               # who was seen by any radar, in list order; nothing else travels
        assert mask is None and lists is None
        return
    assert np.array_equal(mask.astype(np.uint32), vis[seen])
    for r in range(R):
        assert np.array_equal(lists[r], np.nonzero((vis >> r) & 1)[0]), f"radar {r}: order differs from single process"


def test_single_process_exchange_is_identity():
    from zrk_modulation_amd.exchange import DetectionExchange
    ex = DetectionExchange(8, torch.device("cpu"))
    packed = torch.tensor([3, (5 << 32) | 1, (9 << 32) | 3, (12 << 32) | 2, 0, 0, 0, 0, 0], dtype=torch.int64)
    ex.all_gather(packed)
    idx, mask = ex.merged()
    assert idx.tolist() == [5, 9, 12] and mask.tolist() == [1, 3, 2]
    assert ex.radar_list(1).tolist() == [9, 12] and not ex.overflowed()
