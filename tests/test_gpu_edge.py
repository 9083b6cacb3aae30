"""Edge cases of the device path: degenerate radar parameters, pathological positions, empty and
ragged inputs, the size limits of the C ABI, the long-missile-table launch path."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_masks(pos, alive, radars, threads=8):
    from oracle import oracle as O
    L = O.lib()
    n = len(alive)
    p = np.ascontiguousarray(pos.T).reshape(-1).copy()
    vis = np.zeros(n, np.uint32)
    arr = O.radar_array(radars)
    with np.errstate(all="ignore"):
        L.zo_radar_phase_fused(n, n, O.dptr(p), O.u8ptr(alive), len(radars), arr, 0, None, 0, 0, 0, O.u32ptr(vis), threads, None)
    return vis


def _device_masks(pos, alive, radars, flags=0):
    from zrk_modulation_amd._lib import F_ADVANCE
    from zrk_modulation_amd.store import EntityStore
    n = len(alive)
    st = EntityStore("cuda:0", capacity=max(n, 1))
    st.add_entities(np.arange(n), pos, np.zeros((n, 3)), 0.0)
    st.flush()
    dead = np.nonzero(alive == 0)[0]
    st.kill(dead)
    st.begin_tick(0)
    st.sweep(radars, F_ADVANCE | flags)           # zero velocity: the advance leaves every position as given
    return st.d_vis[:n].cpu().numpy().view(np.uint32), st


def _weird_radars(g, R):
    pick = lambda *xs: xs[g.integers(len(xs))]      # noqa: E731
    out = []
    for _ in range(R):
        pos = g.normal(0, 3000, 3)
        max_d = pick(g.uniform(1e3, 6e4), g.uniform(1e3, 6e4), g.uniform(1e3, 6e4), 0.0, -5.0, np.inf, np.nan, 1e-3, 1e20)
        az0 = pick(g.uniform(0, 360), g.uniform(0, 360), g.uniform(-90, 0), g.uniform(360, 500), 0.0, 180.0, np.nan)
        azr = pick(g.uniform(5, 200), g.uniform(5, 200), g.uniform(200, 420), 0.0, 180.0, 360.0, -10.0)
        el0 = pick(g.uniform(0, 60), g.uniform(0, 60), g.uniform(-40, 0), g.uniform(90, 185), 0.0, 90.0)
        elr = pick(g.uniform(5, 90), g.uniform(5, 90), g.uniform(90, 200), 0.0, 180.0, -3.0)
        out.append((pos[0], pos[1], pos[2], max_d, az0, azr, el0, elr))
    return out


def _weird_positions(g, n, radars):
    pos = np.stack([g.uniform(-5e4, 5e4, n), g.uniform(-5e4, 5e4, n), g.uniform(-8e3, 1.5e4, n)], 1)
    k = n // 16
    rp = np.array([r[:3] for r in radars])
    pos[0:k] = rp[g.integers(len(rp), size=k)]                                  # exactly on a radar
    pos[k:2 * k] = rp[g.integers(len(rp), size=k)] + g.normal(0, 1e-9, (k, 3))    # a hair away
    pos[2 * k:3 * k, 2] = rp[g.integers(len(rp), size=k), 2]                     # same height: dz == 0 exactly
    pos[3 * k:4 * k, 1] = rp[g.integers(len(rp), size=k), 1]                     # due east / west: dy == 0
    pos[4 * k:4 * k + 8] = [[np.nan, 0, 0], [0, np.inf, 0], [1e25, 1e25, 0], [0, 0, -np.inf], [1e-300, 0, 0],
                            [0, 0, 1e-310], [-1e19, 3e18, 5e18], [np.nan, np.nan, np.nan]]
    return pos


@pytest.mark.parametrize("seed", range(6))
def test_degenerate_radars_and_positions_match_oracle(seed):
    """The host-side folding of sector conventions into float32 thresholds (derive_radar) against the
    reference formula, over parameters nobody sane would configure: empty, inverted, wrapped, infinite,
    NaN sectors and ranges; objects on top of radars, on exact axes, at infinity."""
    from zrk_modulation_amd._lib import F_EXACT_ONLY
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    R = [1, 7, 16, 32, 32, 3][seed]
    radars = _weird_radars(g, R)
    n = 40_000
    pos = _weird_positions(g, n, radars)
    alive = (g.uniform(size=n) > 0.05).astype(np.uint8)
    want = _oracle_masks(pos, alive, radars)
    got, _ = _device_masks(pos, alive, radars)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"{len(bad)} masks differ, first at {bad[:5]}: got {got[bad[:5]]}, want {want[bad[:5]]}, pos {pos[bad[:3]]}"
    got_exact, _ = _device_masks(pos, alive, radars, F_EXACT_ONLY)
    assert np.array_equal(got_exact, want)
    assert (got[alive == 0] == 0).all()


def test_empty_ragged_and_limits():
    from zrk_modulation_amd import _lib
    from zrk_modulation_amd._lib import F_ADVANCE, ZrkError
    from zrk_modulation_amd.store import EntityStore
    radar = (0.0, 0.0, 0.0, 5e4, 0.0, 360.0, 0.0, 180.0)
    # no entities at all: every call is a no-op that still leaves well-defined outputs
    st = EntityStore("cuda:0", capacity=16)
    st.begin_tick(0)
    st.sweep([radar], F_ADVANCE)
    det, off = st.compact(1)
    assert off[:2].cpu().tolist() == [0, 0]
    assert st.missile_step(10) == []
    # entities but no radars: advance only, masks cleared
    st.add_entities([1, 2, 3], [[1, 2, 3], [4, 5, 6], [7, 8, 9]], [[1, 0, 0]] * 3, 0.0)
    st.begin_tick(1000)
    st.sweep([], F_ADVANCE)
    assert np.array_equal(st.host_pos("cur"), [[2, 2, 3], [5, 5, 6], [8, 8, 9]])
    assert st.d_vis[:3].cpu().tolist() == [0, 0, 0]
    # every entity dead: nothing seen, positions frozen
    st.kill([0, 1, 2])
    st.begin_tick(2000)
    st.sweep([radar], F_ADVANCE)
    assert st.d_vis[:3].cpu().tolist() == [0, 0, 0]
    assert np.array_equal(st.host_pos("cur"), [[2, 2, 3], [5, 5, 6], [8, 8, 9]])
    # growth past the initial capacity keeps earlier rows and the double buffer intact
    st2 = EntityStore("cuda:0", capacity=256)
    g = np.random.Generator(np.random.PCG64(5))
    a = g.normal(0, 1e4, (300, 3))
    st2.add_entities(np.arange(300), a, np.zeros((300, 3)), 0.0)
    st2.begin_tick(0)
    st2.sweep([radar], F_ADVANCE)
    b = g.normal(0, 1e4, (5000, 3))
    st2.add_entities(1000 + np.arange(5000), b, np.zeros((5000, 3)), 0.0)
    st2.begin_tick(10)
    st2.sweep([radar], F_ADVANCE)
    assert st2.cap >= 5300 and np.array_equal(st2.host_pos("cur"), np.concatenate([a, b]))
    assert np.array_equal(st2.host_pos("prev")[:300], a)
    # more radars than mask bits: refused with an error code and message, nothing launched
    with pytest.raises(ZrkError, match="radar count"):
        st2.sweep([radar] * (_lib.ZRK_MAX_RADARS + 1), 0)
    # exactly the maximum works and bit 31 is usable
    st2.sweep([radar] * _lib.ZRK_MAX_RADARS, 0)
    near = np.linalg.norm(np.concatenate([a, b]), axis=1) <= 5e4
    assert np.array_equal(st2.d_vis[:5300].cpu().numpy().view(np.uint32) == 0xFFFFFFFF, near)
    # detection segment too small: counts stay exact, entries beyond the stride are dropped
    det = st2.det_buffer(16)
    st2.sweep([radar], 0)
    small = st2.det_buffer(1)[:64]
    st2.ctx.check(st2.lib.zrk_compact(st2.ctx.handle, st2.d_vis.data_ptr(), 5300, 1, 0, st2.workspace().data_ptr(),
                                      small.data_ptr(), 64, st2._det_cnt.data_ptr(), None, 0, 0, None), "compact")
    assert int(st2._det_cnt[0].item()) == int(near.sum()) > 64
    assert np.array_equal(small.cpu().numpy(), np.nonzero(near)[0][:64])
    st2.compact_status()


def test_long_missile_table_takes_the_multi_launch_path():
    """More than 16384 missiles in flight: the ordered event list is built by the looping kernel and the
    tombstones by zrk_apply_events; same events and survivors as the oracle."""
    from tests.test_gpu_engine import OracleMirror, _compare_tick
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    n, R, m = 60_000, 2, 20_000
    ids, sp, vel, t0 = S.synthetic_targets(n, 31)
    sp[:, :2] *= 0.25                                   # closer in, so that most launches succeed
    radars = S.synthetic_radars(R)
    eng = HotPathEngine(device="cuda:0", dt_ms=500, seed=3, noise="off")
    eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
    launched = eng.launch_missiles(S.missile_targets(n, m), speed=2500.0, radius=800.0, period=40.0)
    assert launched > 16384
    mir = OracleMirror(eng, radars)
    total = 0
    for k in range(16):
        events = mir.tick(k * 500, 500, 0, None)
        eng.run(1)
        _compare_tick(eng, mir, events, f"tick {k}")
        total += len(events)
    assert total > 2000
    alive = eng.list_view(eng.store.d_alive[:mir.n].cpu().numpy())
    for ms, ts in mir.pending:                           # the oracle applies the last tick's removals lazily
        mir.alive[ms] = 0
        if ts >= 0:
            mir.alive[ts] = 0
    assert np.array_equal(alive, mir.alive)


@pytest.mark.parametrize("path", ["records-in-device-memory", "records-in-the-kernel-arguments"])
def test_every_pair_through_the_exact_tier_on_both_record_paths(path, monkeypatch):
    """ZRK_F_EXACT_ONLY sends every in-range (radar, row) pair through visible_exact, which reads the radar's cold record
    by ADDRESS: in the two-launch loop the records sit in device memory (put there by the previous tick's compaction
    launch), in the overlapped loop and in a stand-alone sweep in the kernel-argument segment.  Round 2 lost a GPU to a
    record address formed where the kernel-argument pointer is null; both paths against the oracle,
    and the device fault word (zrk_compact_status) must stay clear."""
    from tests.test_gpu_c4 import varied_radars
    from tests.test_gpu_engine import OracleMirror, _compare_tick
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd._lib import F_EXACT_ONLY
    from zrk_modulation_amd.engine import HotPathEngine
    n, R, m = 30_000, 7, 150
    monkeypatch.setenv("ZRK_OVERLAP_MIN_ROWS", "0")
    monkeypatch.setenv("ZRK_OVERLAP", "0" if path == "records-in-device-memory" else "1")
    ids, sp, vel, t0 = S.synthetic_targets(n, 17)
    radars = varied_radars(R, 9)
    eng = HotPathEngine(device="cuda:0", dt_ms=300, seed=2, noise="off")
    eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
    eng.launch_missiles(S.missile_targets(n, m), speed=2500.0, radius=600.0, period=30.0)
    eng.loop.flags |= F_EXACT_ONLY
    mir = OracleMirror(eng, radars)
    tick = 0
    for K in (4, 5):
        events = None
        for _ in range(K):
            events = mir.tick(tick * 300, 300, 0, None)
            tick += 1
        eng.run(K)
        assert eng.store.lib.zrk_last_run_overlapped(eng.store.ctx.handle) == (0 if path == "records-in-device-memory" else 1)
        vis, _ = _compare_tick(eng, mir, events, f"{path} after {tick} ticks")
        for r, (got, want) in enumerate(zip(eng.detections(), mir.lists())):       # (detections() checks the fault word)
            assert np.array_equal(got, want), f"{path} after {tick} ticks: radar {r}"
        assert np.count_nonzero(vis) > 500
