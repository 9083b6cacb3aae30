"""The exchange issued from the C side (zrk_run_ticks_x + zrk_exchange_*): on the one GPU of the test box the
communicator has one rank, which exercises everything but the wire -- RCCL bound at run time, its stream and the
events both ways, the list and the detonation events in wire format -- against the plain loop."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engines(n, R, m, seed):
    from tests.test_gpu_engine import _engine
    return _engine(n, R, m, seed=seed, noise="philox")


def test_c_side_exchange_carries_the_same_list_and_events_as_the_plain_loop():
    from zrk_modulation_amd.exchange import RcclExchange, union_bits_words
    n, R, m = 30_000, 6, 400
    eng_a, _, launched = _engines(n, R, m, 11)
    eng_b, _, _ = _engines(n, R, m, 11)
    assert launched > 100
    eng_b.gid0 = eng_b.loop.gid0 = 0
    x = RcclExchange(union_bits_words(eng_b.store.cap, R, eng_b.store.cap), eng_b.store.device, R, offsets=[0], ev_capacity=512)
    total_events = 0
    for k in range(40):
        eng_a.run(1)
        eng_b.run(1, exchange=x)
        slot = k % x.slots
        idx, msk = x.merged(slot)
        st = eng_a.store
        vis = st.vis()[:st.n_uploaded].cpu().numpy().view(np.uint32)
        want = np.nonzero(vis)[0]
        assert np.array_equal(idx.cpu().numpy(), want), f"tick {k}: union list differs"
        assert np.array_equal(msk.cpu().numpy().astype(np.uint32), vis[want]), f"tick {k}: masks differ"
        ev = x.events(slot)
        ne = int(st.dm_evn.item())
        rows_m, rows_t = st.dm_evm[:ne].cpu().numpy(), st.dm_evt[:ne].cpu().numpy()
        lidx = st.d_lidx.cpu().numpy() if st.d_lidx is not None else None
        to_list = lambda r: int(lidx[r]) if lidx is not None else int(r)      # noqa: E731
        assert ev == [(to_list(a), -1 if b < 0 else to_list(b)) for a, b in zip(rows_m, rows_t)], f"tick {k}: events differ"
        total_events += len(ev)
        # and the two loops stay identical (same masks every tick)
        vis_b = eng_b.store.vis()[:st.n_uploaded].cpu().numpy().view(np.uint32)
        assert np.array_equal(vis, vis_b)
    assert total_events > 5
    assert not x.overflowed()
    x.close()


@pytest.mark.parametrize("env", [{}, {"ZRK_EXCHANGE_THREAD": "0"}, {"ZRK_EXCHANGE_EVENTS": "1"}, {"ZRK_EXCHANGE_WAIT_IN_STREAM": "1"}],
                         ids=["flag+thread", "flag", "events", "wait-in-stream"])
def test_k_ticks_per_call_with_the_exchange(env, monkeypatch):
    """Inside one call the lists of all ticks but the last are handed over by the flag the next sweep raises (and
    issued by the exchange's own thread), the last one by an event: both slots must hold what the plain loop found."""
    from zrk_modulation_amd.exchange import RcclExchange, union_bits_words
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, R, m = 20_000, 4, 100
    eng_a, _, _ = _engines(n, R, m, 5)
    eng_b, _, _ = _engines(n, R, m, 5)
    x = RcclExchange(union_bits_words(eng_b.store.cap, R, eng_b.store.cap), eng_b.store.device, R, offsets=[0], ev_capacity=128)
    st = eng_a.store
    for calls in range(3):
        eng_b.run(9, exchange=x)       # (a count that walks the last tick of the call through the slots)
        x.sync()
        last = (9 * (calls + 1) - 1) % x.slots
        eng_a.run(8)
        vis = st.vis()[:st.n_uploaded].cpu().numpy().view(np.uint32)
        idx, msk = x.merged((last - 1) % x.slots)  # the second last tick: handed over by the flag
        assert np.array_equal(idx.cpu().numpy(), np.nonzero(vis)[0]), f"call {calls}: list of the second last tick differs"
        assert np.array_equal(msk.cpu().numpy().astype(np.uint32), vis[np.nonzero(vis)[0]])
        eng_a.run(1)
        vis = st.vis()[:st.n_uploaded].cpu().numpy().view(np.uint32)
        idx, msk = x.merged(last)      # the last tick: by the event
        assert np.array_equal(idx.cpu().numpy(), np.nonzero(vis)[0]), f"call {calls}: list of the last tick differs"
    x.close()


def test_a_collective_whose_list_never_came_goes_out_poisoned_and_fails_the_call(monkeypatch):
    """k_wait_flag gives up (here at once: ZRK_WAIT_FLAG_SPINS=0) instead of holding the device for ever.  The collective
    behind it still runs -- the peers are in it -- but nobody may take what it carried for a tick's list: the count on the
    wire is -1 from the tick after at the latest, every decoder rejects it, and zrk_run_ticks_x / zrk_exchange_sync fail."""
    from zrk_modulation_amd._lib import ZrkError
    from zrk_modulation_amd.exchange import PoisonedList, RcclExchange, decode_events, decode_union_bits, union_bits_words
    monkeypatch.setenv("ZRK_WAIT_FLAG_SPINS", "0")
    n, R, m = 20_000, 4, 100
    eng, _, _ = _engines(n, R, m, 5)
    x = RcclExchange(union_bits_words(eng.store.cap, R, eng.store.cap), eng.store.device, R, offsets=[0], ev_capacity=128)
    with pytest.raises((ZrkError, RuntimeError)):
        eng.run(9, exchange=x)                           # (the host may learn of it only when it synchronises)
        x.sync()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError):
        x.sync()
    with pytest.raises((ZrkError, RuntimeError)):        # and it stays failed: no later list of this exchange passes
        eng.run(2, exchange=x)
        x.sync()
    torch.cuda.synchronize()
    # from the tick after the give-up on every list is poisoned behind its compaction (the first one races with its own):
    # after nine ticks each slot's last list carries the poison
    counts = [int(x.recv[k][0, 0].item()) for k in range(x.slots)]
    assert counts == [-1] * x.slots, counts
    slot = 1
    with pytest.raises(PoisonedList):
        decode_union_bits(x.recv[slot], R, [0], 128)
    with pytest.raises(PoisonedList):
        decode_events(x.recv[slot], 128)
    x.close()


@pytest.mark.parametrize("algo", ["allgather", "direct", "allgather-ungrouped", "direct-ungrouped"])
def test_union_only_wire_and_the_direct_pattern_on_one_rank(algo, monkeypatch):
    """wire "union": the bitmap of the slots seen by any radar and the events, nothing else (a fifth of the bytes); and
    ZRK_EXCHANGE_ALGO=direct: grouped send / receive to every peer instead of ncclAllGather -- with one rank that is the local
    copy, which is all a one-GPU box can run of it.  The two collectives of a two-tick launch go out as one RCCL group
    (ZRK_EXCHANGE_GROUP=0: one after the other).  zrk_exchange_info says what the communicator thinks of itself."""
    from zrk_modulation_amd.exchange import RcclExchange, union_bits_words
    grouped = not algo.endswith("-ungrouped")
    algo = algo.split("-")[0]
    monkeypatch.setenv("ZRK_EXCHANGE_ALGO", algo)
    monkeypatch.setenv("ZRK_EXCHANGE_GROUP", "1" if grouped else "0")
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
    n, R, m = 30_000, 6, 400
    eng_a, _, _ = _engines(n, R, m, 11)
    eng_b, _, _ = _engines(n, R, m, 11)
    eng_b.gid0 = eng_b.loop.gid0 = 0
    words = union_bits_words(eng_b.store.cap, R, 0)
    assert words == 2 + (eng_b.store.cap + 63) // 64
    x = RcclExchange(words, eng_b.store.device, R, offsets=[0], ev_capacity=512, wire="union")
    st = eng_a.store
    tick, events = 0, 0
    for K in (1, 6, 1, 9):                          # plain loop, overlapped pair launches, an odd tail
        eng_a.run(K)
        eng_b.run(K, exchange=x)
        tick += K
        slot = (tick - 1) % x.slots
        idx, msk = x.merged(slot)
        vis = st.vis()[:st.n_uploaded].cpu().numpy().view(np.uint32)
        assert msk is None and np.array_equal(idx.cpu().numpy(), np.nonzero(vis)[0]), f"after {tick} ticks: union list differs"
        ne = int(st.dm_evn.item())
        lidx = st.d_lidx.cpu().numpy() if st.d_lidx is not None else None
        to_list = lambda r: int(lidx[r]) if lidx is not None else int(r)      # noqa: E731
        want = [(to_list(a), -1 if b < 0 else to_list(b)) for a, b in zip(st.dm_evm[:ne].cpu().numpy(), st.dm_evt[:ne].cpu().numpy())]
        assert x.events(slot) == want
        events += ne
    assert not x.overflowed()
    info = x.info()
    assert info["world"] == 1 and info["rccl_ranks_seen"] in (1, -1) and info["collectives"] == tick
    assert info["pattern"] == ("direct send/recv" if algo == "direct" else "ncclAllGather") and info["grouped_pairs"] == grouped
    x.close()


@pytest.mark.parametrize("wire", ["masks", "union"])
def test_radars_of_interest_shape_the_exchanged_list(wire):
    """zrk_exchange_io::interest: the list that travels is the union list of the radars a remote consumer listens to (the
    reference's command post reads one FoundObjectsMessage per radar of its radar_ids, modules/CCP.py:409-417) -- a slot is on it
    when one of THEM saw it, its mask carries their bits; the loop's own mask buffers are still cleared by what ANY radar saw
    (the next ticks' lists would otherwise carry stale bits), and events travel as before.  Overlapped two-tick launches and
    calls of one tick."""
    from zrk_modulation_amd.exchange import RcclExchange, union_bits_words
    n, R, m = 60_000, 6, 300
    eng_a, _, launched = _engines(n, R, m, 21)
    eng_b, _, _ = _engines(n, R, m, 21)
    for eng in (eng_a, eng_b):                       # (the rank's own per-radar lists are not available beside an interest mask)
        eng.det_idx = None
    eng_b.gid0 = eng_b.loop.gid0 = 0
    interest = [1, 4]
    sel = sum(1 << r for r in interest)
    x = RcclExchange(union_bits_words(eng_b.store.cap, R, 0 if wire == "union" else eng_b.store.cap), eng_b.store.device, R, offsets=[0],
                     ev_capacity=256, wire=wire, interest=interest)
    st = eng_a.store
    tick = 0
    for K in (1, 8, 1, 9, 6):
        eng_b.run(K, exchange=x)
        x.sync()
        for j in range(K):                           # the plain loop tick by tick: what every tick's masks were
            eng_a.run(1)
            vis = st.vis()[:st.n_uploaded].cpu().numpy().view(np.uint32)
            slot = (tick + j) % x.slots
            if K - j <= x.slots:                     # (the slots hold the call's last ticks)
                idx, msk = x.merged(slot)
                want = np.nonzero(vis & sel)[0]
                assert np.array_equal(idx.cpu().numpy(), want), f"tick {tick + j}: the list of the radars of interest differs"
                if wire == "masks":
                    assert np.array_equal(msk.cpu().numpy().astype(np.uint32), vis[want] & sel), f"tick {tick + j}: masks differ"
        tick += K
        assert np.array_equal(st.vis()[:st.n_uploaded].cpu().numpy(), eng_b.store.vis()[:st.n_uploaded].cpu().numpy())
    assert not x.overflowed()
    x.close()
