"""Two ranks on the one GPU of the test box (gloo rendezvous, CUDA tensors): the sharded engine +
DetectionExchange path that bench.py --gpus N runs, checked against a single-table engine."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, R, ticks, q, fmt="pairs"):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    from zrk_modulation_amd.exchange import DetectionExchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ids, sp, vel, t0 = S.synthetic_targets(n, 515)
    radars = S.synthetic_radars(R)
    per = n // world
    lo, hi = rank * per, (rank + 1) * per
    eng = HotPathEngine(device=dev, dt_ms=100, seed=21, noise="philox", gid0=lo)
    eng.load(ids[lo:hi], sp[lo:hi], vel[lo:hi], t0[lo:hi], radars, union_capacity=per, union_format=fmt)
    words = eng.packed.numel()
    if fmt == "pairs":
        ex = [DetectionExchange(per, dev) for _ in range(2)]
    else:
        ex = [DetectionExchange(words, dev, fmt="bits", offsets=[g * per for g in range(world)], R=R) for _ in range(2)]
    bufs = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(2)]
    work = [None, None]
    for k in range(ticks):                     # the double-buffered, asynchronous loop of bench.py
        b = k & 1
        if work[b] is not None:
            work[b].wait()
        eng.packed = bufs[b]
        eng.run(1)
        work[b] = ex[b].all_gather(eng.packed, async_op=True)
    for w in work:
        if w is not None:
            w.wait()
    last = ex[(ticks - 1) & 1]
    idx, mask = last.merged()
    if rank == 0:
        q.put((idx.cpu().numpy(), mask.cpu().numpy(), last.counts(), last.overflowed()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fmt", ["pairs", "bits"])
def test_two_ranks_on_one_gpu_match_single_table(fmt):
    import torch.multiprocessing as mp
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    n, R, world, ticks = 40_000, 5, 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, R, ticks, q, fmt)) for r in range(world)]
    for p in procs:
        p.start()
    idx, mask, counts, overflow = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ids, sp, vel, t0 = S.synthetic_targets(n, 515)
    whole = HotPathEngine(device="cuda:0", dt_ms=100, seed=21, noise="philox", gid0=0)
    whole.load(ids, sp, vel, t0, S.synthetic_radars(R), union_capacity=n)
    whole.run(ticks)
    vis = whole.store.vis()[:n].cpu().numpy().view(np.uint32)
    seen = np.nonzero(vis)[0]
    assert not overflow and sum(counts) == len(seen)
    assert np.array_equal(idx, seen) and np.array_equal(mask.astype(np.uint32), vis[seen])
