"""Overlap mode of zrk_run_ticks (ZRK_OVERLAP=1): the lists of tick t are compacted on a side stream beside the sweep of
tick t + 1, the tombstones / dispatch order / radar records stay on the compute stream as a small launch of their own.
The tick-by-tick loop is checked against the oracle elsewhere (test_gpu_engine.py); here the overlapped loop, in calls
of several ticks, must leave exactly what that loop leaves: position bits of both buffers, flags, masks of the last
tick, every list, the ordered events, the missile table."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state(eng):
    st = eng.store
    n, m = st.n_uploaded, st.m
    torch.cuda.synchronize()
    eng.store.compact_status()
    ne = int(st.dm_evn.item())
    out = dict(pos_cur=st.host_pos("cur").view(np.uint64), pos_prev=st.host_pos("prev").view(np.uint64),
               alive=st.d_alive[:n].cpu().numpy(), vis=st.vis()[:n].cpu().numpy(),
               period=st.dm_period[:m].cpu().numpy().view(np.uint64), status=st.dm_status[:m].cpu().numpy(),
               ev_m=st.dm_evm[:ne].cpu().numpy(), ev_t=st.dm_evt[:ne].cpu().numpy(), ne=np.array([ne]),
               radar=np.array(eng.radar_state()))
    for r, lst in enumerate(eng.detections()):
        out[f"list{r}"] = lst
    return out


def _same(a, b, what):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{what}: {k} differs"


@pytest.mark.parametrize("n,R,m,noise", [(40_000, 6, 300, "philox"), (40_000, 6, 300, "off"), (3_000, 16, 0, "philox"),
                                         (700_000, 5, 2_000, "philox"), (2_600_000, 4, 1_000, "philox")],
                         ids=["mid", "mid-no-noise", "small-no-missiles", "multi-round-grid", "pair-compaction-of-635-workgroups"])
def test_overlapped_loop_leaves_what_the_tick_by_tick_loop_leaves(n, R, m, noise, monkeypatch):
    from tests.test_gpu_engine import _engine
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")           # (by default only tables of 4e5 rows or more overlap)
    monkeypatch.setenv("ZRK_OVERLAP", "0")
    ref, _, launched = _engine(n, R, m, seed=21, noise=noise)
    monkeypatch.setenv("ZRK_OVERLAP", "1")
    ovl, _, _ = _engine(n, R, m, seed=21, noise=noise)
    assert m == 0 or launched > 50
    total_events = 0
    for calls, K in enumerate([5, 4, 9, 1, 6, 2, 7]):            # (calls of fewer than four ticks take the plain loop)
        for _ in range(K):
            ref.run(1)
        ovl.run(K)
        assert ovl.store.lib.zrk_last_run_overlapped(ovl.store.ctx.handle) == (1 if K >= 4 else 0)
        assert ref.store.lib.zrk_last_run_overlapped(ref.store.ctx.handle) == 0
        a, b = _state(ref), _state(ovl)
        _same(a, b, f"after call {calls} of {K} ticks")
        total_events += int(a["ne"][0])
    assert m == 0 or total_events > 0
    assert np.count_nonzero(a["vis"]) > 0


def test_overlap_with_the_exchange_on_one_rank(monkeypatch):
    """The same with the list going through the C-side all-gather (one rank): the compaction (with the events tail) on the
    side stream, the collective on the exchange's own, released by the side stream's next launch; the merged list and
    the events of the second last and the last tick of a call are the plain loop's."""
    from tests.test_gpu_engine import _engine
    from zrk_modulation_amd.exchange import RcclExchange, union_bits_words
    n, R, m = 30_000, 6, 400
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
    monkeypatch.setenv("ZRK_OVERLAP", "0")
    ref, _, _ = _engine(n, R, m, seed=11, noise="philox")
    monkeypatch.setenv("ZRK_OVERLAP", "1")
    ovl, _, _ = _engine(n, R, m, seed=11, noise="philox")
    ovl.gid0 = ovl.loop.gid0 = 0
    x = RcclExchange(union_bits_words(ovl.store.cap, R, ovl.store.cap), ovl.store.device, R, offsets=[0], ev_capacity=512)
    st = ref.store
    lidx = st.d_lidx.cpu().numpy() if st.d_lidx is not None else None
    to_list = (lambda r: int(lidx[r])) if lidx is not None else (lambda r: int(r))
    tick = 0
    seen_events = 0
    for K in (7, 5, 8):
        ovl.run(K, exchange=x)
        x.sync()
        for back in (1, 0):                                      # second last tick, then the last one
            ref.run(K - 1 if back else 1)
            slot = (tick + (K - 2 if back else K - 1)) % x.slots
            vis = st.vis()[:st.n_uploaded].cpu().numpy().view(np.uint32)
            idx, msk = x.merged(slot)
            want = np.nonzero(vis)[0]
            assert np.array_equal(idx.cpu().numpy(), want), f"K={K} back={back}: union list differs"
            assert np.array_equal(msk.cpu().numpy().astype(np.uint32), vis[want])
            ne = int(st.dm_evn.item())
            rows_m, rows_t = st.dm_evm[:ne].cpu().numpy(), st.dm_evt[:ne].cpu().numpy()
            assert x.events(slot) == [(to_list(a), -1 if b < 0 else to_list(b)) for a, b in zip(rows_m, rows_t)]
            seen_events += ne
        tick += K
    assert seen_events > 0
    _same(_state(ref), _state(ovl), "after the exchanged calls")
    x.close()


def test_several_missiles_on_one_target(monkeypatch):
    """Removals travel as marks in the overlapped loop (first mark stands, never cleared inside a call).  The case that
    needs exactly that: several missiles on the same target with different fuse radii -- the later ones arrive when the
    target is gone, detonate on its frozen position (or run out of time) and mark it again, in the very tick in which
    its own thread carries the first removal out.  State after every call must be the two-launch loop's."""
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    n, R, m = 60_000, 5, 1200
    ids, sp, vel, t0 = S.synthetic_targets(n, 77)
    sp *= 0.25                                                   # a compact swarm: short flights, hits within the test
    vel *= 0.2
    radars = S.synthetic_radars(R)
    tgt = (np.arange(m) // 6 * 97 % n).astype(np.int32)          # six missiles per target
    radius = np.tile(np.array([2500.0, 1500.0, 900.0, 500.0, 250.0, 120.0]), m // 6)
    engines = []
    for ov in ("0", "1"):
        monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
        monkeypatch.setenv("ZRK_OVERLAP", ov)
        eng = HotPathEngine(device="cuda:0", dt_ms=200, seed=5, noise="philox", gid0=0)
        eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
        assert eng.launch_missiles(tgt, launcher_pos=(0.0, 0.0, 0.0), speed=2500.0, radius=radius, period=30.0) > 600
        engines.append(eng)
    ref, ovl = engines
    removed_late = 0
    prev_alive = None
    for calls, K in enumerate([6, 9, 4, 12, 7, 5, 20, 8, 16]):
        ref.run(K)
        ovl.run(K)
        assert ovl.store.lib.zrk_last_run_overlapped(ovl.store.ctx.handle) == 1
        a, b = _state(ref), _state(ovl)
        _same(a, b, f"after call {calls} of {K} ticks")
        if prev_alive is not None:
            removed_late += int(prev_alive.sum() - a["alive"].sum())
        prev_alive = a["alive"]
    st = ref.store
    status = st.dm_status[:st.m].cpu().numpy()
    done_per_target = np.bincount(np.asarray(st.hm_tgt[:st.m])[status == 2], minlength=1)
    assert done_per_target.max() >= 3, "several missiles of one target should have detonated or timed out"
    assert removed_late > 0


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_gather_records_follow_the_table(overlap, monkeypatch):
    """The missile phase reads its targets from 64-byte records that zrk_run_ticks keeps per row (built for the rows
    appended since the last call).  A second and a third salvo between calls append rows: same state as an engine that
    reads the columns (ZRK_GATHER_RECORDS=0), after every call."""
    from tests.test_gpu_engine import _engine
    from zrk_modulation_amd import scenario as S
    n, R, m = 50_000, 6, 900
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
    monkeypatch.setenv("ZRK_OVERLAP", overlap)
    engines = []
    for rec in ("1", "0"):
        monkeypatch.setenv("ZRK_GATHER_RECORDS", rec)
        eng, _, launched = _engine(n, R, 300, seed=9, noise="philox")
        assert launched > 50
        engines.append(eng)
    a_eng, b_eng = engines
    later = S.missile_targets(n, 300) + 57                       # other targets for the later salvos
    for calls, K in enumerate([6, 5, 9, 4, 12, 7]):
        for eng in engines:
            eng.run(K)
        _same(_state(a_eng), _state(b_eng), f"after call {calls} of {K} ticks")
        if calls in (0, 2):
            got = [eng.launch_missiles((later + 1000 * calls) % n, launcher_pos=(500.0, -300.0, 0.0), speed=2800.0, radius=900.0, period=20.0)
                   for eng in engines]
            assert got[0] == got[1] and got[0] > 50
    assert a_eng.store.m > 450


def test_overlapped_calls_match_the_oracle_directly():
    """The overlapped loop against the ORACLE, without the two-launch loop in between: a table large enough for the
    default thresholds (no environment set), calls of 8 and 6 ticks; the oracle is ticked the same number of times on
    the host (with the noise table the device will draw), then masks, position bits, detonation events and lists of
    the call's last tick, the flags and the radars' scan state must be the oracle's."""
    from tests.test_gpu_engine import OracleMirror, _compare_tick, _device_noise_table
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    n, R, m = 640_000, 7, 2000
    ids, sp, vel, t0 = S.synthetic_targets(n, 99)
    radars = S.synthetic_radars(R)
    g = np.random.Generator(np.random.PCG64(100))
    for k, rd in enumerate(radars):
        rd["max_distance"] = float(g.uniform(2e4, 6e4)); rd["azimuth_start"] = float(g.uniform(0, 360))
        rd["azimuth_range"] = float(g.uniform(20, 200)); rd["elevation_range"] = float(g.uniform(10, 90))
        rd["azimuth_speed"] = float(g.uniform(1, 30)); rd["elevation_speed"] = float(g.uniform(0, 5))
        rd["position"] = [float(v) for v in g.normal(0, 8e3, 3) * [1, 1, 0.05]]
        if k == 4:
            rd["scan_mode"] = "vertical"
    eng = HotPathEngine(device="cuda:0", dt_ms=250, seed=31337, noise="philox")
    eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
    assert eng.launch_missiles(S.missile_targets(n, m), speed=2500.0, radius=500.0, period=45.0) > 100
    mir = OracleMirror(eng, radars)
    tick = 0
    seen_events = 0
    for K in (8, 6):
        events = None
        for _ in range(K):
            table = _device_noise_table(eng, tick, R, mir.n)
            events = mir.tick(tick * 250, 250, 2, table, threads=16)
            seen_events += len(events)
            tick += 1
        eng.run(K)
        assert eng.store.lib.zrk_last_run_overlapped(eng.store.ctx.handle) == 1
        vis, alive = _compare_tick(eng, mir, events, f"after {tick} ticks")
        lists = eng.detections()
        for r, want in enumerate(mir.lists()):
            assert np.array_equal(lists[r], want), f"after {tick} ticks: radar {r}"
        assert eng.radar_state() == [(r["caz"], r["cel"]) for r in mir.rs]
        assert np.count_nonzero(vis) > 1000
    assert seen_events > 0


def test_a_missile_whose_target_is_a_missile(monkeypatch):
    """A missile row may name another missile's row as its target (the C ABI takes any row).
    * B hits A in tick t: A is out of the air from tick t + 1 on (AirEnv.py:33-40) -- in the overlapped loop that removal is
      a mark which A's own row thread carries out somewhere inside a later grid, the very grid whose leading workgroups step
      the missiles: A's missile thread must read the mark, not the flag.  Half of these A's would time out in exactly that
      tick if they were allowed to step once more: an extra event, an extra removal.
    * A chases C, which stands BEHIND it in the list: A compares with what C held after the tick before -- radar noise
      included -- which in the second tick of a pair launch no thread may have written yet (the missile phase then replays
      C's radar phase of the first tick).  The chases end in hits at ticks of either parity.
    Against the two-launch loop after every call, and against the oracle tick by tick."""
    from tests.test_gpu_engine import OracleMirror, _compare_tick, _device_noise_table
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    n, R, pairs, dt = 50_000, 4, 700, 200
    ids, sp, vel, t0 = S.synthetic_targets(n, 31)
    sp[:, :2] *= 0.3
    radars = S.synthetic_radars(R)
    tgt = (np.arange(pairs) * 61 % n).astype(np.int32)
    engines = []
    for ov in ("0", "1"):
        monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
        monkeypatch.setenv("ZRK_OVERLAP", ov)
        eng = HotPathEngine(device="cuda:0", dt_ms=dt, seed=8, noise="philox", gid0=0)
        eng.load(ids, sp, vel, t0, radars, missile_capacity=3 * pairs).enable_lists()
        # the A's: slow, with different fuse radii (the chases below end at different times)
        radius_a = 120.0 + 35.0 * (np.arange(pairs) % 10)
        k = eng.launch_missiles(tgt, launcher_pos=(0.0, 0.0, 600.0), speed=2350.0, radius=radius_a, period=40.0)
        ok_a = eng.launch_results["rc"] == 0
        assert k > pairs // 2
        # the B's fly the same lines (same launcher, same target, same speed: the same solves succeed); the C's leave at the
        # same time from 600 m below, faster, for the same targets: the two lines converge, and what an A sees of its C --
        # where C stood a tick ago, C being behind A in the list -- comes within A's fuse radius somewhere on the way
        assert eng.launch_missiles(tgt, launcher_pos=(0.0, 0.0, 600.0), speed=2350.0, radius=50.0, period=40.0) == k
        kc = eng.launch_missiles(tgt, speed=2500.0, radius=50.0, period=40.0)
        ok_c = eng.launch_results["rc"] == 0
        st = eng.store
        rows_m = np.asarray(st.hm_slot[:st.m], np.int32)
        row_a = np.full(pairs, -1, np.int32); row_a[ok_a] = rows_m[:k]
        row_b = np.full(pairs, -1, np.int32); row_b[ok_a] = rows_m[k:2 * k]
        row_c = np.full(pairs, -1, np.int32); row_c[ok_c] = rows_m[2 * k:2 * k + kc]
        mrow_of = {int(r): i for i, r in enumerate(rows_m)}              # table row -> missile row
        new_tgt = np.asarray(st.hm_tgt[:st.m], np.int32).copy()
        who = np.arange(pairs)
        hit_by_b = ok_a & (who % 3 == 0)
        chasing = ok_a & ok_c & (who % 3 == 1)
        for i in np.nonzero(hit_by_b)[0]:
            new_tgt[mrow_of[int(row_b[i])]] = row_a[i]                   # B_i -> A_i
        for i in np.nonzero(chasing)[0]:
            new_tgt[mrow_of[int(row_a[i])]] = row_c[i]                   # A_i -> C_i (behind it in the list)
        st.dm_tgt[:st.m] = torch.as_tensor(new_tgt, device=st.device)
        st.hm_tgt = new_tgt
        rows_a, rows_c = row_a, row_c
        # every sixth A would run out of time in its second tick in the air (a launch with so short a fuse would be
        # cancelled, Missile.py:98-99: set afterwards)
        st.dm_period[0:k:6] = 1.5 * dt / 1000
        engines.append(eng)
    ref, ovl = engines
    mir = OracleMirror(ref, radars)
    tick = 0
    seen = []
    for calls, K in enumerate([6, 4, 9, 5, 8]):
        events = None
        for _ in range(K):
            events = mir.tick(tick * dt, dt, 2, _device_noise_table(ref, tick, R, mir.n))
            seen.append(events)
            tick += 1
        ref.run(K)
        ovl.run(K)
        assert ovl.store.lib.zrk_last_run_overlapped(ovl.store.ctx.handle) == 1
        _same(_state(ref), _state(ovl), f"after call {calls} of {K} ticks")
        _compare_tick(ovl, mir, events, f"overlapped loop after {tick} ticks")
    # tick 0: every third B hits its A (distance 0); tick 1: none of those A's detonates, although half of them were due
    list_a = set(int(x) for x in mir.lidx[rows_a[hit_by_b]])
    assert sum(1 for ms, ts in seen[0] if ts in list_a) == len(list_a)
    assert not any(ms in list_a for ms, ts in seen[1])
    # the chases A_i -> C_i: hits in ticks of both parities
    list_c = set(int(x) for x in mir.lidx[rows_c[chasing]])
    hit_ticks = [t for t, evs in enumerate(seen) for ms, ts in evs if ts in list_c]
    assert len(hit_ticks) > 20 and len({t % 2 for t in hit_ticks}) == 2, hit_ticks[:40]


def test_a_failed_side_stream_is_not_sticky(monkeypatch):
    """The side stream's thread gives up when the compute stream does not reach the next sweep within the host wait limit
    (here: 40 ms, behind ~0.3 s of unrelated work queued on the stream).  THAT call fails with ZRK_E_STATE; the table
    stands as after the ticks that were swept; the next call runs overlapped again and ends where an engine that never
    failed ends."""
    from tests.test_gpu_engine import _engine
    from zrk_modulation_amd._lib import ZrkError
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
    monkeypatch.setenv("ZRK_OVERLAP", "0")
    ref, _, _ = _engine(40_000, 5, 300, seed=13, noise="philox")
    monkeypatch.setenv("ZRK_OVERLAP", "1")
    ovl, _, _ = _engine(40_000, 5, 300, seed=13, noise="philox")
    st = ovl.store
    ovl.run(6); ref.run(6)
    _same(_state(ref), _state(ovl), "before the failure")
    monkeypatch.setenv("ZRK_HOST_WAIT_MS", "40")
    st.lib.zrk_ctx_reload_env(st.ctx.handle)
    a = torch.ones(6144, 6144, dtype=torch.float64, device=st.device)
    b = torch.empty_like(a)
    torch.mm(a, a, out=b)
    torch.cuda.synchronize()
    for _ in range(40):                                  # a few hundred milliseconds of unrelated work in front of the loop
        torch.mm(a, a, out=b)
    tick0 = int(ovl.loop.tick)
    with pytest.raises(ZrkError):
        ovl.run(40)
    torch.cuda.synchronize()
    done = int(ovl.loop.tick) - tick0
    assert 0 < done < 40, "the call should have stopped at the ring slot whose compaction never came"
    monkeypatch.setenv("ZRK_HOST_WAIT_MS", "30000")
    st.lib.zrk_ctx_reload_env(st.ctx.handle)
    ref.run(done)
    a, b = _state_table(ref), _state_table(ovl)
    _same(a, b, f"table after the failed call ({done} ticks swept)")
    ovl.run(7); ref.run(7)
    assert st.lib.zrk_last_run_overlapped(st.ctx.handle) == 1
    _same(_state(ref), _state(ovl), "after the call behind the failure")


def _state_table(eng):
    """What a failed call must still leave intact: the table and the missile rows (not the lists, not the events)."""
    st = eng.store
    n, m = st.n_uploaded, st.m
    torch.cuda.synchronize()
    return dict(pos_cur=st.host_pos("cur").view(np.uint64), pos_prev=st.host_pos("prev").view(np.uint64),
                alive=st.d_alive[:n].cpu().numpy(), period=st.dm_period[:m].cpu().numpy().view(np.uint64),
                status=st.dm_status[:m].cpu().numpy(), radar=np.array(eng.radar_state()))


@pytest.mark.parametrize("tail", ["compute", "side", "compute-behind-the-side-stream"])
def test_calls_on_alternating_streams(tail, monkeypatch):
    """A call's last compaction runs on the caller's stream (ZRK_TAIL_COMPUTE=0: on the side stream, taken in through an
    event), and the next call may come on ANOTHER stream that the caller has ordered behind the first: short calls (two
    launches each) and long ones on two streams in turn must leave what the same calls on one stream leave."""
    from tests.test_gpu_engine import _engine
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
    monkeypatch.setenv("ZRK_OVERLAP", "1")
    monkeypatch.setenv("ZRK_TAIL_COMPUTE", "0" if tail == "side" else "1")
    # (ZRK_TAIL_FREE=0: the call's last compaction, on the compute stream, takes the side stream's launch before it in through an
    # event instead of standing alone with control words of its own -- calls of two launches (K = 4) and of more)
    monkeypatch.setenv("ZRK_TAIL_FREE", "0" if tail == "compute-behind-the-side-stream" else "1")
    one, _, launched = _engine(120_000, 6, 400, seed=33, noise="philox")
    two, _, _ = _engine(120_000, 6, 400, seed=33, noise="philox")
    assert launched > 50
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for calls, K in enumerate([4, 4, 5, 4, 11, 4, 4, 6, 4, 4, 4, 4, 9, 4]):
        one.run(K)
        s = streams[calls % 2]
        s.wait_stream(streams[(calls + 1) % 2])                 # (the caller's ordering of its own streams)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            two.run(K)
        if calls % 4 == 3:
            torch.cuda.current_stream().wait_stream(s)
            _same(_state(one), _state(two), f"after call {calls} of {K} ticks on stream {calls % 2}")
    torch.cuda.synchronize()
    _same(_state(one), _state(two), "at the end")


def test_marks_carried_out_by_the_last_compaction_or_by_a_launch_of_their_own(monkeypatch):
    """A call's removal marks become tombstones in extra workgroups of the call's last compaction launch (MarksArgs); with
    ZRK_MARKS_IN_TAIL=0 in a launch of their own between the last sweep and that compaction, as before round 5.  Calls of even
    and odd length with missiles that hit inside them must leave the same table, masks, lists and events either way -- and the
    first launch of a call zeroes the caller's mask buffer that its last tick writes (SweepParams::vis_clear): the masks read
    after every call are compared too."""
    from tests.test_gpu_engine import _engine
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
    monkeypatch.setenv("ZRK_OVERLAP", "1")
    monkeypatch.setenv("ZRK_MARKS_IN_TAIL", "0")
    one, _, launched = _engine(90_000, 6, 500, seed=41, noise="philox")
    monkeypatch.setenv("ZRK_MARKS_IN_TAIL", "1")
    two, _, _ = _engine(90_000, 6, 500, seed=41, noise="philox")
    assert launched > 50
    dead0 = int((one.store.d_alive[:one.store.n_uploaded] == 0).sum().item())
    for calls, K in enumerate([4, 7, 12, 5, 4, 20, 9, 6]):
        one.run(K)
        two.run(K)
        assert two.store.lib.zrk_last_run_overlapped(two.store.ctx.handle) == 1
        _same(_state(one), _state(two), f"after call {calls} of {K} ticks")
    assert int((one.store.d_alive[:one.store.n_uploaded] == 0).sum().item()) > dead0, "no removal inside the calls: the test is empty"
