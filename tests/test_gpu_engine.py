"""The headless fused path (zrk_run_ticks: device-resident events, fused advance + multi-radar
sweep with Philox noise, compaction) against the CPU oracle, and size-independent properties at
the full BASELINE sizes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(n, R, m, seed, noise, lists=True, union=False, sort=True):
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    ids, sp, vel, t0 = S.synthetic_targets(n, seed)
    radars = S.synthetic_radars(R)
    # make the scene less uniform than the bench one: mixed ranges, a vertical scanner, elevation steps
    for k, rd in enumerate(radars):
        rd["max_distance"] = 50e3 if k % 2 == 0 else 30e3
        rd["elevation_speed"] = 0.0 if k % 3 else 5.0
        if k % 4 == 3:
            rd["scan_mode"] = "vertical"
        rd["azimuth_start"] = 20.0 * k
    eng = HotPathEngine(device="cuda:0", dt_ms=500, seed=4242, noise=noise, gid0=0)
    eng.load(ids, sp, vel, t0, radars, missile_capacity=m, union_capacity=(n + m) if union else None, sort=sort)
    if lists:
        eng.enable_lists()
    # closer targets so that missiles arrive within the test
    launched = eng.launch_missiles(S.missile_targets(n, m), launcher_pos=(0.0, 0.0, 0.0), speed=3000.0, radius=1000.0,
                                   period=25.0)
    return eng, (ids, sp, vel, t0, radars), launched


class OracleMirror:
    """Host copy of the engine's table driven by the oracle's C functions."""

    def __init__(self, eng, radars):
        from oracle import oracle as O
        self.O, self.L = O, O.lib()
        st = eng.store
        n = self.n = st.n_uploaded
        L = eng.list_view                                 # table rows -> AirEnv list order
        self.lidx = st.h_lidx[:n].astype(np.int64) if st.h_lidx is not None else np.arange(n)
        self.sp = np.ascontiguousarray(L(st.h_sp[:n]).T).reshape(-1); self.vel = np.ascontiguousarray(L(st.h_vel[:n]).T).reshape(-1)
        self.t0 = L(st.h_t0[:n]).copy(); self.pos = np.ascontiguousarray(L(st.h_pos0[:n]).T).reshape(-1).copy()
        self.prev = self.pos.copy(); self.pv = np.zeros(n, np.uint8); self.alive = np.ones(n, np.uint8)
        self.kind = L(st.h_kind[:n]).copy(); self.mrow = np.full(n, -1, np.int32)
        m = self.m = st.m
        self.mrow[self.lidx[st.hm_slot[:m]]] = np.arange(m, dtype=np.int32)
        self.m_tgt = self.lidx[st.hm_tgt[:m]].astype(np.int32)
        self.m_radius = st.dm_radius[:m].cpu().numpy().copy(); self.m_period = st.dm_period[:m].cpu().numpy().copy()
        self.m_status = np.ones(max(m, 1), np.uint8)
        self.ev = (np.zeros(max(m, 1), np.int32), np.zeros(max(m, 1), np.int32), np.zeros(max(m, 1), np.uint8))
        self.vis = np.zeros(n, np.uint32)
        self.rs = [dict(r, caz=r["azimuth_start"], cel=r["elevation_start"]) for r in radars]
        self.pending = []

    def tick(self, time_ms, dt_ms, mode, table=None, threads=8):
        from zrk_modulation_amd.engine import scan_mode_code, scan_next
        O, L, n = self.O, self.L, self.n
        for ms, ts in self.pending:
            self.alive[ms] = 0
            if ts >= 0:
                self.alive[ts] = 0
        evm, evt, evs = self.ev
        nev = L.zo_airenv_step(n, n, time_ms, dt_ms, O.dptr(self.sp), O.dptr(self.vel), O.dptr(self.t0),
                               O.u8ptr(self.alive), O.u8ptr(self.kind), O.i32ptr(self.mrow), O.dptr(self.pos),
                               O.dptr(self.prev), O.u8ptr(self.pv), O.i32ptr(self.m_tgt), O.dptr(self.m_radius),
                               O.dptr(self.m_period), O.u8ptr(self.m_status), O.i32ptr(evm), O.i32ptr(evt), O.u8ptr(evs))
        events = [(int(evm[k]), int(evt[k])) for k in range(nev)]
        self.pending = events
        arr = O.radar_array([(r["position"][0], r["position"][1], r["position"][2], r["max_distance"], r["caz"],
                              r["azimuth_range"], r["cel"], r["elevation_range"]) for r in self.rs])
        L.zo_radar_phase_fused(n, n, O.dptr(self.pos), O.u8ptr(self.alive), len(self.rs), arr, mode,
                               O.dptr(table) if table is not None else None, 0, 0, 0, O.u32ptr(self.vis), threads)
        for r in self.rs:
            r["caz"], r["cel"] = scan_next(scan_mode_code(r["scan_mode"]), r["azimuth_range"], r["azimuth_speed"],
                                           r["elevation_speed"], r["elevation_start"], r["caz"], r["cel"])
        return events

    @classmethod
    def from_parts(cls, parts):
        """ONE population out of the mirrors of its shards (BASELINE config 4): the lists of the shards one after the
        other -- shard g's targets, then its missiles -- which is the order the rank-ordered concatenation of the shards'
        wire lists must reproduce.  Dense index of shard g's list element i: base[g] + i."""
        self = cls.__new__(cls)
        self.O, self.L = parts[0].O, parts[0].L
        self.base = np.cumsum([0] + [p.n for p in parts])
        mbase = np.cumsum([0] + [p.m for p in parts])
        n = self.n = int(self.base[-1])
        m = self.m = int(mbase[-1])
        planes = lambda name: np.concatenate([getattr(p, name).reshape(3, p.n) for p in parts], axis=1).reshape(-1).copy()   # noqa: E731
        self.sp, self.vel, self.pos, self.prev = planes("sp"), planes("vel"), planes("pos"), planes("prev")
        self.t0 = np.concatenate([p.t0 for p in parts])
        self.pv = np.zeros(n, np.uint8); self.alive = np.ones(n, np.uint8)
        self.kind = np.concatenate([p.kind for p in parts])
        self.mrow = np.concatenate([np.where(p.mrow >= 0, p.mrow + mbase[g], -1) for g, p in enumerate(parts)]).astype(np.int32)
        self.m_tgt = np.concatenate([p.m_tgt[:p.m] + self.base[g] for g, p in enumerate(parts)] + [np.zeros(0, np.int64)]).astype(np.int32)
        self.m_radius = np.concatenate([p.m_radius[:p.m] for p in parts] + [np.zeros(0)]).copy()
        self.m_period = np.concatenate([p.m_period[:p.m] for p in parts] + [np.zeros(0)]).copy()
        if m == 0:
            self.m_tgt, self.m_radius, self.m_period = np.zeros(1, np.int32), np.zeros(1), np.zeros(1)
        self.m_status = np.ones(max(m, 1), np.uint8)
        self.ev = (np.zeros(max(m, 1), np.int32), np.zeros(max(m, 1), np.int32), np.zeros(max(m, 1), np.uint8))
        self.vis = np.zeros(n, np.uint32)
        self.rs = [dict(r) for r in parts[0].rs]
        self.pending = []
        self.lidx = np.arange(n)
        return self

    def lists(self):
        out = []
        buf = np.zeros(self.n, np.int32)
        for r in range(len(self.rs)):
            k = self.L.zo_compact_bit(self.n, self.O.u32ptr(self.vis), r, 0, self.O.i32ptr(buf))
            out.append(buf[:k].copy())
        return out


def _device_noise_table(eng, tick, R, n):
    """The noise triples the device will draw this tick (k-th detection of each slot), dumped by the
    device itself."""
    import torch
    st = eng.store
    tab = torch.zeros(R, n, 3, dtype=torch.float64, device=st.device)
    for r in range(R):            # entity key = gid0 + LIST index, so entity0 + i enumerates the list
        st.ctx.check(st.lib.zrk_selftest_noise(st.ctx.handle, eng.seed, tick, r, eng.gid0, tab[r].data_ptr(), n, None), "noise")
    return np.ascontiguousarray(tab.cpu().numpy()).reshape(-1)


def _compare_tick(eng, mir, events, tag):
    st = eng.store
    n = st.n_uploaded
    vis = st.vis()[:n].cpu().numpy().view(np.uint32)               # already indexed by list index
    assert np.array_equal(vis, mir.vis), f"{tag}: visibility masks differ"
    P = eng.list_view(st.host_pos("cur"))
    assert np.array_equal(np.ascontiguousarray(P.T).reshape(-1).view(np.uint64), mir.pos.view(np.uint64)), \
        f"{tag}: position bits differ"
    k = int(st.dm_evn.item()) if st.m else 0
    to_list = lambda rows: [int(mir.lidx[r]) if r >= 0 else -1 for r in rows]      # noqa: E731  event rows -> list indices
    got = list(zip(to_list(st.dm_evm[:k].cpu().tolist()), to_list(st.dm_evt[:k].cpu().tolist())))
    assert got == events, f"{tag}: detonation events differ: {got} vs {events}"
    alive = st.d_alive[:n].cpu().numpy()
    # device tombstones lag one tick exactly like the reference's (applied at the start of the next tick)
    return vis, alive


@pytest.mark.parametrize("noise", ["off", "philox"])
def test_fused_engine_matches_oracle_tick_by_tick(noise):
    n, R, m = 20000, 6, 300
    eng, scene, launched = _engine(n, R, m, seed=77, noise=noise)
    assert launched > 50
    mir = OracleMirror(eng, scene[4])
    total_events = 0
    for k in range(60):
        t_ms = k * 500
        table = _device_noise_table(eng, k, R, mir.n) if noise == "philox" else None
        events = mir.tick(t_ms, 500, 2 if noise == "philox" else 0, table)
        eng.run(1)
        _compare_tick(eng, mir, events, f"{noise} tick {k}")
        lists = eng.detections()
        for r, want in enumerate(mir.lists()):
            assert np.array_equal(lists[r], want), f"{noise} tick {k}: radar {r} list differs"
        assert eng.radar_state() == [(r["caz"], r["cel"]) for r in mir.rs]
        total_events += len(events)
    assert total_events > 20, "scene too quiet to exercise the missile path"


def test_run_k_ticks_equals_k_single_ticks():
    """zrk_run_ticks(K) == K x zrk_run_ticks(1): nothing in the loop depends on returning to the host."""
    a, _, _ = _engine(5000, 4, 100, seed=5, noise="philox")
    b, _, _ = _engine(5000, 4, 100, seed=5, noise="philox")
    a.run(40)
    for _ in range(40):
        b.run(1)
    assert np.array_equal(a.store.host_pos("cur"), b.store.host_pos("cur"))
    c = _engine(5000, 4, 100, seed=5, noise="philox", sort=False)[0]      # rows in list order: same observables
    c.run(40)
    assert np.array_equal(a.list_view(a.store.host_pos("cur")), c.store.host_pos("cur"))
    assert np.array_equal(a.store.vis()[:a.store.n_uploaded].cpu().numpy(), c.store.vis()[:c.store.n_uploaded].cpu().numpy())
    for x, y in zip(a.detections(), c.detections()):
        assert np.array_equal(x, y)
    assert np.array_equal(a.store.d_alive.cpu().numpy(), b.store.d_alive.cpu().numpy())
    for x, y in zip(a.detections(), b.detections()):
        assert np.array_equal(x, y)


def test_union_list_is_consistent_with_masks_and_lists():
    eng, _, _ = _engine(30000, 8, 0, seed=9, noise="philox", union=True)
    eng.run(3)
    st = eng.store
    n = st.n_uploaded
    vis = st.vis()[:n].cpu().numpy().view(np.uint32)
    packed = eng.packed.cpu().numpy()
    cnt = int(packed[0])
    seen = np.nonzero(vis)[0]
    assert cnt == len(seen)
    assert np.array_equal(packed[1:1 + cnt] >> 32, seen + eng.gid0)
    assert np.array_equal((packed[1:1 + cnt] & 0xFFFFFFFF).astype(np.uint32), vis[seen])
    for r, lst in enumerate(eng.detections()):
        assert np.array_equal(lst, np.nonzero((vis >> r) & 1)[0])


# ---- full BASELINE sizes: size-independent properties ------------------------------------------------
@pytest.mark.parametrize("workload", ["C2", "C3"])
def test_full_size_properties(workload):
    """At BASELINE sizes: (1) one tick against the oracle (OpenMP, noise off) bit for bit,
    (2) lists are sorted, duplicate-free and equal to the mask bits, (3) fused multi-radar sweep ==
    R single-radar sweeps when nothing perturbs positions, (4) alive count only ever decreases."""
    import torch
    from oracle import oracle as O
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    from zrk_modulation_amd._lib import F_ADVANCE
    n, R, m = S.WORKLOADS[workload]
    ids, sp, vel, t0 = S.synthetic_targets(n, S.SEEDS[workload])
    radars = S.synthetic_radars(R)
    eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise="off")
    eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
    eng.launch_missiles(S.missile_targets(n, m))
    mir = OracleMirror(eng, radars)
    for k in range(2):
        events = mir.tick(k * 10, 10, 0, None, threads=16)
        eng.run(1)
        vis, alive = _compare_tick(eng, mir, events, f"{workload} tick {k}")
    lists = eng.detections()
    for r, lst in enumerate(lists):
        assert np.all(np.diff(lst) > 0)
        assert np.array_equal(lst, np.nonzero((vis >> r) & 1)[0])
    # (3) tick 0 again on a fresh table, one radar at a time, OR-ing the masks
    eng2 = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise="off")
    eng2.load(ids, sp, vel, t0, radars, missile_capacity=m)
    eng2.launch_missiles(S.missile_targets(n, m))
    st2 = eng2.store
    st2.begin_tick(0)
    acc = np.zeros(st2.n_uploaded, np.uint32)
    params = [(r["position"][0], r["position"][1], r["position"][2], r["max_distance"], r["azimuth_start"],
               r["azimuth_range"], r["elevation_start"], r["elevation_range"]) for r in radars]
    st2.sweep([], F_ADVANCE)
    for r in range(R):
        st2.sweep([params[r]], 0)
        acc |= (st2.d_vis[:st2.n_uploaded].cpu().numpy().view(np.uint32) & 1) << r
    eng3 = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise="off")
    eng3.load(ids, sp, vel, t0, radars, missile_capacity=m)
    eng3.launch_missiles(S.missile_targets(n, m))
    eng3.run(1)
    assert np.array_equal(eng3.store.vis()[:st2.n_uploaded].cpu().numpy().view(np.uint32), acc)
    # (4) run on for a while: tombstones are monotone and positions stay finite
    before = eng.alive_count()
    eng.run(200)
    after = eng.alive_count()
    assert after <= before
    assert np.isfinite(eng.store.host_pos("cur")).all()


def test_sharded_population_equals_single_table():
    """BASELINE config 4 in miniature: the population cut into contiguous index
    ranges, one engine per shard with its global offset, gives -- concatenated in rank order -- exactly
    the single-table result: masks, positions, per-radar lists, union list.  The noise stream is keyed by
    the global list index, so the cut does not show in the noise either."""
    import torch
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    from zrk_modulation_amd.exchange import DetectionExchange
    n, R, shards, ticks = 48_000, 6, 3, 12
    ids, sp, vel, t0 = S.synthetic_targets(n, 404)
    radars = S.synthetic_radars(R)
    whole = HotPathEngine(device="cuda:0", dt_ms=100, seed=9, noise="philox", gid0=0)
    whole.load(ids, sp, vel, t0, radars, union_capacity=n).enable_lists()
    parts = []
    per = n // shards
    for g in range(shards):
        lo, hi = g * per, (g + 1) * per
        e = HotPathEngine(device="cuda:0", dt_ms=100, seed=9, noise="philox", gid0=lo)
        e.load(ids[lo:hi], sp[lo:hi], vel[lo:hi], t0[lo:hi], radars, union_capacity=per).enable_lists()
        parts.append(e)
    for _ in range(ticks):
        whole.run(1)
        for e in parts:
            e.run(1)
    wst = whole.store
    vis_whole = wst.vis()[:n].cpu().numpy()
    vis_parts = np.concatenate([e.store.vis()[:per].cpu().numpy() for e in parts])
    assert np.array_equal(vis_whole, vis_parts)
    pos_whole = whole.list_view(wst.host_pos("cur"))
    pos_parts = np.concatenate([e.list_view(e.store.host_pos("cur")) for e in parts])
    assert np.array_equal(pos_whole, pos_parts)
    # rank-ordered concatenation of the shards' packed union lists == the single table's
    packed_whole = whole.packed.cpu().numpy()
    cnt = int(packed_whole[0])
    merged = np.concatenate([e.packed.cpu().numpy()[1:1 + int(e.packed[0].item())] for e in parts])
    assert np.array_equal(merged, packed_whole[1:1 + cnt])
    # and through the exchange's own decoding (single process: world 1 per shard)
    for r, lst in enumerate(whole.detections()):
        per_shard = [e.detections()[r] + g * per for g, e in enumerate(parts)]
        assert np.array_equal(np.concatenate(per_shard), lst)
    ex = DetectionExchange(n, torch.device("cuda", 0))
    ex.all_gather(whole.packed)
    assert np.array_equal(ex.radar_list(2).cpu().numpy(), whole.detections()[2])


def test_ensemble_replicas_are_independent_and_match_oracle():
    """BASELINE config 5 in miniature: independent scenarios (own seed, own targets, own missiles) as
    replicas on one GPU -- no collective, no shared state: each replica matches its own oracle replay and
    is unaffected by the others running beside it."""
    n, R, m = 10_000, 4, 100
    solo, _, _ = _engine(n, R, m, seed=900, noise="off")
    for _ in range(10):
        solo.run(1)
    solo_pos = solo.list_view(solo.store.host_pos("cur")).copy()
    engines = [_engine(n, R, m, seed=900 + k, noise="off") for k in range(4)]
    mirrors = [OracleMirror(e, scene[4]) for e, scene, _ in engines]
    for k in range(10):
        for (e, _, _), mir in zip(engines, mirrors):
            events = mir.tick(k * 500, 500, 0, None)
            e.run(1)
            _compare_tick(e, mir, events, f"replica tick {k}")
    assert np.array_equal(engines[0][0].list_view(engines[0][0].store.host_pos("cur")), solo_pos)


def test_multi_round_grid_with_noise_matches_oracle():
    """A table large enough that the sweep grid is not resident at once (so the cost-ordered dispatch is on,
    with the wave-level cull deciding most waves): masks, position bits, lists and events against the oracle
    for a few ticks of varied, moving sectors with Philox noise."""
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    n, R, m, ticks = 640_000, 7, 2000, 5
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
    seen = 0
    for k in range(ticks):
        table = _device_noise_table(eng, k, R, mir.n)
        events = mir.tick(k * 250, 250, 2, table, threads=16)
        eng.run(1)
        vis, _ = _compare_tick(eng, mir, events, f"tick {k}")
        lists = eng.detections()
        for r, want in enumerate(mir.lists()):
            assert np.array_equal(lists[r], want), f"tick {k} radar {r}"
        seen += int(np.count_nonzero(vis))
    eng.store.compact_status()
    assert seen > 10_000


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_match_oracle(seed):
    """Random radars (ranges, sector widths past 180 degrees, tilted and vertical scanners, radars far from and in
    the middle of the swarm), random swarm shapes, sorted storage, Philox noise, missiles: every mask, position
    bit, list and event against the oracle, tick by tick.  Small enough to run many; the cull and the float32
    pre-classification see every kind of sector face here."""
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    g = np.random.Generator(np.random.PCG64(4000 + seed))
    n, R, m, ticks = int(g.integers(20_000, 60_000)), int(g.integers(1, 12)), 400, 8
    ids, sp, vel, t0 = S.synthetic_targets(n, 900 + seed)
    sp[:, :2] *= g.uniform(0.2, 1.5)                               # tight or wide swarm
    sp[:, 2] = g.uniform(-2e3, 2e4, n) if seed % 2 else sp[:, 2]   # also below the radars
    t0[:] = g.uniform(-50.0, 0.0, n)                               # trajectories that started at different times
    radars = S.synthetic_radars(R)
    for k, rd in enumerate(radars):
        rd["max_distance"] = float(g.choice([g.uniform(5e3, 9e4), g.uniform(5e2, 5e3)]))
        rd["azimuth_start"] = float(g.uniform(0, 360)); rd["azimuth_range"] = float(g.choice([g.uniform(5, 180), g.uniform(180, 360)]))
        rd["elevation_start"] = float(g.uniform(0, 60)); rd["elevation_range"] = float(g.uniform(5, 120))
        rd["azimuth_speed"] = float(g.uniform(0, 40)); rd["elevation_speed"] = float(g.uniform(0, 10))
        rd["position"] = [float(v) for v in g.normal(0, 3e4, 3) * [1, 1, 0.05]]
        rd["scan_mode"] = ["horizontal", "vertical", "horizontal"][k % 3]
    eng = HotPathEngine(device="cuda:0", dt_ms=200, seed=seed, noise="philox")
    eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
    eng.launch_missiles(S.missile_targets(n, m), speed=2500.0, radius=600.0, period=30.0)
    mir = OracleMirror(eng, radars)
    for k in range(ticks):
        table = _device_noise_table(eng, k, R, mir.n)
        events = mir.tick(k * 200, 200, 2, table)
        eng.run(1)
        _compare_tick(eng, mir, events, f"seed {seed} tick {k}")
        lists = eng.detections()
        for r, want in enumerate(mir.lists()):
            assert np.array_equal(lists[r], want), f"seed {seed} tick {k} radar {r}"
        assert eng.radar_state() == [(r["caz"], r["cel"]) for r in mir.rs]


def test_replay_frame_from_a_headless_tick():
    """ReplayLog.record_store after a fused tick on a spatially sorted table: live objects in list order, their
    positions, seen = any radar's bit -- against the oracle mirror of the same tick."""
    from zrk_modulation_amd.replay import ReplayLog
    eng, scene, launched = _engine(9000, 5, 120, seed=5, noise="off")
    mir = OracleMirror(eng, scene[4])
    log = ReplayLog(max_steps=2)
    for k in range(12):
        events = mir.tick(k * 500, 500, 0, None)
        eng.run(1)
        log.record_store(k * 500, eng.store)
    ids, kinds, pos, seen = log.frame(11 * 500)
    live = mir.alive.astype(bool)
    for ms, ts in mir.pending:                         # the oracle applies the last tick's removals lazily
        live[ms] = False
        if ts >= 0:
            live[ts] = False
    all_ids = eng.list_view(eng.store.h_ids[:mir.n])
    assert np.array_equal(ids, all_ids[live]) and len(log.steps()) == 2
    assert np.array_equal(pos, mir.pos.reshape(3, mir.n).T[live])
    assert np.array_equal(seen, (mir.vis != 0)[live])
    assert set(kinds.tolist()) <= {0, 1} and (kinds == 1).sum() <= launched
