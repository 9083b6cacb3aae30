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


def _sharded_worker(rank, world, port, n, R, m, ticks, dt, q):
    """One rank of a sharded run WITH missiles, through the rehearsal loop of bench.py (gloo, wire format "bits"); rank 0
    then replays the WHOLE population on the oracle and compares every tick's merged wire with it."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from tests.test_gpu_c4 import build_shards, varied_radars
    from tests.test_gpu_engine import OracleMirror
    from zrk_modulation_amd.exchange import DetectionExchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    radars = varied_radars(R, 5)
    # (every rank builds only its own shard: build_shards with `only`)
    engines, stride, _ = build_shards(n, R, m, world, 321, dt, radars, only=rank)
    eng = engines[0]
    part = OracleMirror(eng, radars)
    st = eng.store
    words = int(st.lib.zrk_union_bits_words(st.cap, R, st.cap))
    caps = [None] * world
    dist.all_gather_object(caps, words)
    words = max(caps)
    offsets = [g * stride for g in range(world)]
    ex = [DetectionExchange(words, dev, fmt="bits", offsets=offsets, R=R) for _ in range(2)]
    bufs = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(2)]
    eng.loop.flags |= 8                                   # ZRK_F_UNION_BITS
    lidx = st.d_lidx.cpu().numpy() if st.d_lidx is not None else None
    glob = (lambda row: eng.gid0 + (int(lidx[row]) if lidx is not None else int(row)))
    per_tick = []
    for k in range(ticks):
        b = k & 1
        eng.packed = bufs[b]
        eng.run(1)
        ex[b].all_gather(eng.packed)
        idx, mask = ex[b].merged()
        ne = int(st.dm_evn.item()) if st.m else 0
        ev = [(glob(a), -1 if t < 0 else glob(t)) for a, t in zip(st.dm_evm[:ne].cpu().tolist(), st.dm_evt[:ne].cpu().tolist())]
        all_ev = [None] * world
        dist.all_gather_object(all_ev, ev)                # rank after rank, each rank's in list order
        per_tick.append((idx.cpu().numpy(), mask.cpu().numpy(), [e for part_ev in all_ev for e in part_ev]))
    mine = {k: v for k, v in part.__dict__.items() if k not in ("O", "L")}
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    if rank == 0:
        from tests.test_gpu_c4 import WholeOracle
        mirrors = []
        for d in parts:
            p = OracleMirror.__new__(OracleMirror)
            p.__dict__.update(d); p.O, p.L = part.O, part.L
            mirrors.append(p)
        ora = WholeOracle.__new__(WholeOracle)
        ora.parts, ora.whole, ora.stride, ora.R = mirrors, OracleMirror.from_parts(mirrors), stride, R
        ora.engines = None
        table = torch.zeros(R, ora.whole.n, 3, dtype=torch.float64, device=dev)
        seen_events = seen = 0
        try:
            for k in range(ticks):
                for r in range(R):                        # the noise every rank drew: keyed by the GLOBAL index
                    for g, p in enumerate(mirrors):
                        lo = int(ora.whole.base[g])
                        st.ctx.check(st.lib.zrk_selftest_noise(st.ctx.handle, eng.seed, k, r, g * stride, table[r, lo:lo + p.n].data_ptr(), p.n, None), "noise")
                events = ora.whole.tick(k * dt, dt, 2, np.ascontiguousarray(table.cpu().numpy()).reshape(-1), threads=8)
                idx, mask, ev = per_tick[k]
                ora.check(torch.from_numpy(idx), torch.from_numpy(mask), ev, (events, ora.whole.vis.copy(), ora.whole.lists()), f"tick {k}")
                seen_events += len(events); seen += int(np.count_nonzero(ora.whole.vis))
            q.put(("ok", seen_events, seen))
        except AssertionError as exc:
            q.put(("failed", str(exc), 0))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharded_with_missiles_match_the_oracle_of_the_whole_population():
    """configs[3] in the rehearsal loop: two ranks (gloo, one GPU), 200 000 rows and 400 missiles cut in two, Philox noise,
    eight ticks; the merged wire of every tick and the gathered detonations against the oracle of the whole population."""
    import torch.multiprocessing as mp
    n, R, m, world, ticks, dt = 200_000, 6, 400, 2, 8, 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, n, R, m, ticks, dt, q)) for r in range(world)]
    for p in procs:
        p.start()
    verdict = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert verdict[0] == "ok", verdict[1]
    assert verdict[1] > 10 and verdict[2] > 10_000
