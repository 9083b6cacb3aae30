"""Parity of the loop variants bench.py TIMES, at the sizes it times them.

* C3 at full size (1e6 x 16 radars, 1e4 missiles): overlapped calls, two ticks per launch, one compaction launch per
  pair, Philox noise, against the oracle ticked the same number of times -- masks, position bits, lists, ordered events,
  flags, missile rows, scan state.
* One overlapped call longer than the removal marks' period (a mark is 2 + 2 * (tick % 126) + b and is never cleared
  inside a call): detonations before and after the wrap, missiles that arrive at a target removed more than 126 ticks
  earlier, a missile whose target is a missile -- against the tick-by-tick loop AND the oracle.
* C5 at its stated size (128 scenarios x 1e4 targets x 4 radars x 100 missiles in one table): every scenario against
  its own oracle replay, tick by tick and through an overlapped call.

Reference lines at stake: modules/AirEnv.py:33-48 (removal effective from the next tick, list order), modules/Missile.py:186-193
(fuse), modules/Radar.py:53-71,138-142 (gate in list order, in-place noise), main.py:151-174 (one run per scenario)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _alive_after(mir):
    """The oracle applies a tick's removals at the start of the next one (AirEnv.py:33-40); the device's table has them
    as tombstones when a call returns."""
    a = mir.alive.copy()
    for ms, ts in mir.pending:
        a[ms] = 0
        if ts >= 0:
            a[ts] = 0
    return a


def _compare_call_end(eng, mir, events, tag):
    from tests.test_gpu_engine import _compare_tick
    st = eng.store
    n, m = st.n_uploaded, st.m
    vis, alive = _compare_tick(eng, mir, events, tag)
    lists = eng.detections()
    for r, want in enumerate(mir.lists()):
        assert np.array_equal(lists[r], want), f"{tag}: radar {r} list differs"
    assert eng.radar_state() == [(r["caz"], r["cel"]) for r in mir.rs], f"{tag}: scan state differs"
    assert np.array_equal(eng.list_view(alive), _alive_after(mir)), f"{tag}: flags differ"
    # missile rows: status (1 in the air, 2 detonated) and the fuse timer's bits
    assert np.array_equal(st.dm_status[:m].cpu().numpy() == 1, mir.m_status[:m] == 1), f"{tag}: missile status differs"
    assert np.array_equal(st.dm_period[:m].cpu().numpy().view(np.uint64), mir.m_period[:m].view(np.uint64)), f"{tag}: fuse timers differ"
    return vis


def test_c3_full_size_overlapped_pair_philox_matches_the_oracle():
    """BASELINE configs[2] exactly as bench.py builds and times it (scenario.WORKLOADS['C3'], seed, dt = 10 ms, noise on,
    1e4 missiles in flight, no environment overrides: overlapped, two ticks per launch, k_compact_pair) -- except that some
    missiles get fuse radii / life times that end inside the test, so that removals, marks and the ordered event lists are
    exercised at this size too.  Calls of 8, 9 (an odd tick at the end) and 12 ticks."""
    from tests.test_gpu_engine import OracleMirror, _device_noise_table
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    n, R, m = S.WORKLOADS["C3"]
    ids, sp, vel, t0 = S.synthetic_targets(n, S.SEEDS["C3"])
    radars = S.synthetic_radars(R)
    eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=S.SEEDS["C3"], noise="philox")
    eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
    # every fifth missile's fuse radius ends a few tens of metres short of where its target starts: the gap closes at
    # ~10 m per tick, so these hit in different ticks of the test; the others keep the stock 150 m.  Launched as bench.py
    # launches (fill_missiles): until m are in flight, the failed solves' neighbours next
    want = S.missile_targets(n, m).astype(np.int64)
    launched, shift = 0, 0
    while launched < m and shift < 16 and len(want):
        tg = want % n
        radius = np.full(len(tg), 150.0)
        short = (tg // 100) % 5 == 0
        radius[short] = np.linalg.norm(sp[tg[short]], axis=1) - 8.0 * (1 + (tg[short] // 500) % 24)
        launched += eng.launch_missiles(tg, radius=radius)
        shift += 1
        want = (want[eng.launch_results["rc"] != 0] + shift)[: m - launched]
    assert launched == m
    st = eng.store
    # ... and every seventh runs out of time within the test (a launch with so short a life would be cancelled,
    # Missile.py:98-99: set afterwards)
    life = torch.arange(launched, device=st.device)
    st.dm_period[:launched] = torch.where(life % 7 == 3, 0.005 + 0.01 * ((life // 7) % 27).double(), st.dm_period[:launched])
    mir = OracleMirror(eng, radars)
    tick, event_ticks, n_events = 0, set(), 0
    for K in (8, 9, 12):
        events = None
        for _ in range(K):
            table = _device_noise_table(eng, tick, R, mir.n)
            events = mir.tick(tick * 10, 10, 2, table, threads=16)
            if events:
                event_ticks.add(tick)
                n_events += len(events)
            tick += 1
            del table
        eng.run(K)
        assert st.lib.zrk_last_run_overlapped(st.ctx.handle) == 1
        assert st.lib.zrk_last_run_ticks_per_launch(st.ctx.handle) == 2
        vis = _compare_call_end(eng, mir, events, f"C3 after {tick} ticks")
        assert np.count_nonzero(vis) > 100_000
    assert n_events > 500 and len(event_ticks) >= 12, (n_events, sorted(event_ticks))
    hits = int((~_alive_after(mir).astype(bool)).sum())
    assert hits > n_events                      # hits removed targets as well as missiles


def _long_call_engine(monkeypatch, overlap):
    """5e4 targets, 4 radars, dt = 200 ms; four salvos whose flights end all over the first 330 ticks: fast and slow
    missiles (hits from tick ~3 to tick ~300), life times spread over the whole call (timeouts), two or three missiles per
    target with different speeds (the late ones arrive at a target that was removed long before), and a few missiles whose
    target is another missile."""
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    n, R, dt = 50_000, 4, 200
    ids, sp, vel, t0 = S.synthetic_targets(n, 515)
    sp[:, :2] *= 0.3                                               # within 18 km (x 1.4) of the launcher
    vel *= 0.35                                                    # slow enough for the slow missiles' solves to succeed
    radars = S.synthetic_radars(R)
    for k, rd in enumerate(radars):
        rd["azimuth_start"] = 45.0 * k
        rd["elevation_speed"] = 0.0 if k % 2 else 5.0
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
    monkeypatch.setenv("ZRK_OVERLAP", overlap)
    eng = HotPathEngine(device="cuda:0", dt_ms=dt, seed=77, noise="philox", gid0=0)
    eng.load(ids, sp, vel, t0, radars, missile_capacity=4000).enable_lists()
    tgt = (np.arange(700) * 67 % n).astype(np.int32)
    got = []
    # the same 700 targets four times over: 2500 m/s (arrive within ~40 ticks), 500, 250 and 130 m/s (arrive, if their
    # target is still there, hundreds of ticks later; most find it removed by the fast ones and fly to where it froze)
    for speed, radius, period in ((2500.0, 300.0, 70.0), (500.0, 400.0, 70.0), (250.0, 500.0, 70.0), (130.0, 600.0, 70.0)):
        got.append(eng.launch_missiles(tgt, launcher_pos=(0.0, 0.0, 300.0), speed=speed, radius=radius, period=period))
    st = eng.store
    mtot = st.m
    assert got[0] > 500 and mtot > 1400, got
    # life times all over the call for a third of the rows (timeouts before, at and after the wrap)
    rows = torch.arange(mtot, device=st.device)
    st.dm_period[:mtot] = torch.where(rows % 3 == 1, 0.2 * (2 + (rows * 37) % 330).double() - 0.1, st.dm_period[:mtot])
    # missiles chasing missiles: a few rows of the slowest salvo take a row of the fastest as their target
    rows_m = np.asarray(st.hm_slot[:mtot], np.int32)
    new_tgt = np.asarray(st.hm_tgt[:mtot], np.int32).copy()
    last0 = mtot - got[3]
    for i in range(0, min(got[3], got[0]), 9):
        new_tgt[last0 + i] = rows_m[i]
    st.dm_tgt[:mtot] = torch.as_tensor(new_tgt, device=st.device)
    st.hm_tgt = new_tgt
    return eng, radars, dt, R


def test_one_overlapped_call_longer_than_the_mark_period(monkeypatch):
    """K = 330 ticks in ONE call (pair launches, marks wrapping after 126 and 252 ticks), then a call of 131: the same state
    as 330 + 131 calls of one tick on the plain loop, and as the oracle's."""
    from tests.test_gpu_engine import OracleMirror, _device_noise_table
    from tests.test_gpu_overlap import _same, _state
    ref, radars, dt, R = _long_call_engine(monkeypatch, "0")
    ovl, _, _, _ = _long_call_engine(monkeypatch, "1")
    mir = OracleMirror(ref, radars)
    removed_at = {}                                                # list index of a removed target -> tick
    tick = 0
    per_tick = []
    late_arrivals = 0
    tgt_of = {int(mir.lidx[s]): int(mir.m_tgt[k]) for k, s in enumerate(ref.store.hm_slot[:ref.store.m])}
    for K in (330, 131):
        events = None
        for _ in range(K):
            events = mir.tick(tick * dt, dt, 2, _device_noise_table(ref, tick, R, mir.n))
            per_tick.append(len(events))
            for ms, ts in events:
                t_of_ms = tgt_of.get(ms)
                if t_of_ms in removed_at and tick - removed_at[t_of_ms] > 126:
                    late_arrivals += 1                            # this missile's target had been gone for more than a mark period
                if ts >= 0:
                    removed_at.setdefault(ts, tick)
                removed_at.setdefault(ms, tick)
            ref.run(1)
            tick += 1
        ovl.run(K)
        assert ovl.store.lib.zrk_last_run_overlapped(ovl.store.ctx.handle) == 1
        assert ovl.store.lib.zrk_last_run_ticks_per_launch(ovl.store.ctx.handle) == 2
        assert ref.store.lib.zrk_last_run_overlapped(ref.store.ctx.handle) == 0
        _same(_state(ref), _state(ovl), f"after {tick} ticks (one call of {K})")
        _compare_call_end(ovl, mir, events, f"overlapped call of {K} ticks, {tick} in all")
    ev = np.array(per_tick)
    assert ev[:126].sum() >= 100 and ev[126:252].sum() >= 100 and ev[252:330].sum() >= 30, (ev[:126].sum(), ev[126:252].sum(), ev[252:330].sum())
    assert late_arrivals >= 10, late_arrivals


def _check_ensemble(eng, ticks_single, call_ticks, dt_ms):
    """Every scenario of a loaded EnsembleEngine against its own OracleMirror: `ticks_single` calls of one tick compared
    after each, then one call of `call_ticks` ticks (the overlapped ensemble loop) compared at its end."""
    from tests.test_gpu_engine import OracleMirror, _device_noise_table
    S_, R, P = eng.S, eng.R, eng.P
    st = eng.store
    views = [eng.scenario_view(s) for s in range(S_)]
    mirrors = [OracleMirror(v, eng.radars[s]) for s, v in enumerate(views)]
    total_events, seen = 0, 0

    def compare(tag, want_events, full):
        nonlocal total_events, seen
        vis_all = st.vis()[:S_ * P].cpu().numpy().view(np.uint32)
        pos_all = st.host_pos("cur")
        ne = int(st.dm_evn.item())
        evm, evt = st.dm_evm[:ne].cpu().numpy(), st.dm_evt[:ne].cpu().numpy()
        det_cnt = eng.det_cnt.cpu().numpy() if full else None
        det_idx = eng.det_idx.cpu().numpy() if full else None
        if full:
            st.compact_status()
        for s, (v, mir) in enumerate(zip(views, mirrors)):
            lo = s * P
            assert np.array_equal(vis_all[lo:lo + mir.n], mir.vis), f"{tag} scenario {s}: masks differ"
            assert not vis_all[lo + mir.n:lo + P].any(), f"{tag} scenario {s}: padding rows detected"
            pl = v.list_view(pos_all[lo:lo + P])
            assert np.array_equal(np.ascontiguousarray(pl.T).reshape(-1).view(np.uint64), mir.pos.view(np.uint64)), \
                f"{tag} scenario {s}: position bits differ"
            sel = (evm >= lo) & (evm < lo + P)
            got = [(int(mir.lidx[a - lo]), int(mir.lidx[b - lo]) if b >= 0 else -1) for a, b in zip(evm[sel], evt[sel])]
            assert got == want_events[s], f"{tag} scenario {s}: events differ: {got} vs {want_events[s]}"
            total_events += len(got)
            seen += int(np.count_nonzero(mir.vis))
            if full:
                for r, want in enumerate(mir.lists()):
                    k = int(det_cnt[s * (R + 1) + r])
                    a = (s * R + r) * eng.det_stride
                    assert np.array_equal(det_idx[a:a + k], want), f"{tag} scenario {s} radar {r}: list differs"
                assert eng.radar_state(s) == [(r["caz"], r["cel"]) for r in mir.rs], f"{tag} scenario {s}: scan state differs"

    tick = 0
    for k in range(ticks_single):
        want = [mir.tick(tick * dt_ms, dt_ms, 2, _device_noise_table(v, tick, R, mir.n)) for v, mir in zip(views, mirrors)]
        eng.run(1)
        compare(f"tick {tick}", want, full=(k == ticks_single - 1))
        tick += 1
    if call_ticks:
        want = None
        for _ in range(call_ticks):
            want = [mir.tick(tick * dt_ms, dt_ms, 2, _device_noise_table(v, tick, R, mir.n)) for v, mir in zip(views, mirrors)]
            tick += 1
        eng.run(call_ticks)
        compare(f"call of {call_ticks} ticks ending at tick {tick - 1}", want, full=True)
    return total_events, seen


def test_c5_at_its_stated_size_every_scenario_matches_its_own_oracle():
    """BASELINE configs[4] per GPU as bench.py builds it: 128 scenarios x 1e4 targets x 4 radars x 100 missiles, dt = 10 ms,
    Philox noise, the bench's seeds.  Some life times are shortened so that events occur inside the test."""
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.ensemble import EnsembleEngine
    scen, n, R, m = S.ENSEMBLES["C5"]
    assert (scen, n, R, m) == (128, 10_000, 4, 100)
    eng = EnsembleEngine(device="cuda:0", dt_ms=10, noise="philox")
    eng.load_synthetic(scen, n, R, m, seed=S.SEEDS["C5"], first_scenario=0)
    st = eng.store
    assert eng.launched > 0.6 * scen * m             # (as bench.py builds it: the solves that fail are not retried)
    rows = torch.arange(st.m, device=st.device)
    st.dm_period[:st.m] = torch.where(rows % 5 == 2, 0.005 + 0.01 * ((rows // 5) % 9).double(), st.dm_period[:st.m])
    events, seen = _check_ensemble(eng, ticks_single=3, call_ticks=6, dt_ms=10)
    assert events > 200 and seen > 128 * 1000


def test_sweeps_time_themselves(monkeypatch):
    """zrk_sweep_stamps: the sweep launches of a call write the wall clock as their first waves start and as every wave ends;
    zrk_read_sweep_stamps gives one duration per launch and the ticks it swept -- pair launches of the overlapped loop, one-tick
    launches of the plain loop, nothing when off -- and the durations agree with an event pair riding on the same dispatch
    (which also holds the dispatch in front and the release behind: a few microseconds more, never less)."""
    from tests.test_gpu_engine import _engine
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
    eng, _, _ = _engine(120_000, 6, 300, seed=3, noise="philox")
    us, ticks = eng.read_sweep_stamps()
    assert len(us) == 0                                            # off by default
    eng.sweep_stamps(True)
    eng.run(13)                                                    # six pair launches and an odd tick
    us, ticks = eng.read_sweep_stamps()
    assert ticks.tolist() == [2] * 6 + [1] and ((us > 1.0) & (us < 2000.0)).all(), (us, ticks)
    # ... and when they ran (zrk_last_sweep_stamp_times): one after the other on their stream, from the first one's first wave on
    b, e = eng.sweep_stamp_times()
    assert len(b) == len(us) and b[0] == 0.0 and np.allclose(e - b, us, atol=0.02)
    assert (b[1:] >= e[:-1] - 0.02).all() and e[-1] < 5000.0, (b, e)
    for _ in range(3):                                             # the plain loop, one event pair per launch beside the stamps
        ms = np.zeros(1, np.float32)
        eng.run(1, sweep_ms=ms, prof_stride=1)
        us1, t1 = eng.read_sweep_stamps()
        assert t1.tolist() == [1] and 1.0 < us1[0] <= ms[0] * 1e3 + 0.5 and us1[0] > 0.3 * ms[0] * 1e3, (us1, ms)
    eng.run(150)                                                   # more launches than slots: every k-th is kept
    us, ticks = eng.read_sweep_stamps()
    assert 32 <= len(us) <= 64 and set(ticks.tolist()) == {2}
    eng.sweep_stamps(False)
    eng.run(6)
    assert len(eng.read_sweep_stamps()[0]) == 0
