"""CPU-only checks: host-side logic of the drop-in modules, the C-ABI surface, fail-loud behaviour
without a GPU.  No compute kernels run here."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_symbol_the_header_declares():
    from zrk_modulation_amd import _lib
    _lib.build()
    header = (ROOT / "include" / "zrk_hot.h").read_text()
    declared = set(re.findall(r"\b(zrk_[a-z_0-9]+)\s*\(", header))
    lib = C.CDLL(str(_lib.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/zrk_hot.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), "ctypes binding and header disagree on the surface"
    assert lib.zrk_abi_version() == _lib.ZRK_ABI_VERSION


def test_ctypes_structs_match_the_header_layout():
    from zrk_modulation_amd import _lib
    assert C.sizeof(_lib.ZrkRadar) == 8 * 8
    assert C.sizeof(_lib.ZrkEntities) == 8 * 11
    assert C.sizeof(_lib.ZrkMissiles) == 8 * 10
    assert C.sizeof(_lib.ZrkLaunchReq) == 56 and C.sizeof(_lib.ZrkLaunchRes) == 40
    assert C.sizeof(_lib.ZrkScan) == 32 and C.sizeof(_lib.ZrkLoop) == 64
    assert C.sizeof(_lib.ZrkRcclId) == 128 and C.sizeof(_lib.ZrkExchangeIo) == 8 + 2 * 8 * _lib.EXCHANGE_SLOTS + 16 and C.sizeof(_lib.ZrkEnsemble) == 56
    _lib.launch_dtypes()                      # numpy views of the launch records agree with the structs


def test_no_gpu_means_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from zrk_modulation_amd import HotPathUnavailable
    from zrk_modulation_amd.store import EntityStore
    with pytest.raises(HotPathUnavailable):
        EntityStore()
    from zrk_modulation_amd.modules.AirEnv import AirEnv
    from zrk_modulation_amd.modules.Manager import Manager
    from zrk_modulation_amd.modules.AirObject import Trajectory
    from zrk_modulation_amd.modules.utils import Target
    m = Manager()
    ae = AirEnv(m, 1, np.zeros(3))           # constructing is GPU-free ...
    with pytest.raises(HotPathUnavailable):   # ... touching the table is not
        ae.add_target(Target(m, 7, np.zeros(3), Trajectory((1, 0, 0), (0, 0, 0), 0.0)))


def test_product_package_never_imports_the_oracle():
    for path in (ROOT / "zrk_modulation_amd").rglob("*.py"):
        text = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{path} imports the oracle"
        assert "libzrk_oracle" not in text and "zrk_oracle" not in text, f"{path} loads the oracle library"
    for path in (ROOT / "zrk_modulation_amd").rglob("*.hip"):
        assert "zrk_oracle" not in path.read_text()


def test_manager_schedules_by_class_name_and_sorts_by_relevance():
    """SURVEY.md 5.9-1 and 5.9-12 (reference modules/Manager.py:58, :123-127)."""
    from zrk_modulation_amd.modules.BaseMessage import BaseMessage
    from zrk_modulation_amd.modules.constants import MessageType
    from zrk_modulation_amd.modules.Manager import Manager
    order = []

    def make(name):
        return type(name, (), {"id": name, "step": lambda self: order.append(type(self).__name__)})()

    m = Manager()
    for n in ("Zeta", "CombatControlPoint", "SectorRadar", "MissileLauncher", "AirEnv", "Alpha"):
        m.add_module(make(n))
    second_radar = make("SectorRadar")
    m.add_module(second_radar)
    m.time.set_dt(10)
    m.run_simulation(10)
    assert order == ["AirEnv", "SectorRadar", "SectorRadar", "MissileLauncher", "CombatControlPoint", "Zeta", "Alpha"]
    assert m.time.get_time() == 10
    a = BaseMessage(MessageType.MISSILE_POS, 1)
    b = BaseMessage(MessageType.MISSILE_COUNT_REQUEST, 2, relevance=3)
    c = BaseMessage(MessageType.MISSILE_POS, 3)
    for x in (a, b, c):
        m.add_message(x, step_time=50)
    assert m.give_messages(50) == [b, a, c] and a.send_time == 50
    assert m.give_messages_by_type(MessageType.MISSILE_POS, step_time=50) == [a, c]
    assert m.give_messages_by_id(None, step_time=50) == [b, a, c]


def test_message_shapes_and_quirks():
    from zrk_modulation_amd.modules import Messages as M
    from zrk_modulation_amd.modules.constants import MessageType
    d = M.DestroyedMissileId(sender_id=5, missile_id=3005, receiver_id=0, self_detonation=True)
    assert d.missile_id == (3005,) and d.type == MessageType.DESTROYED_MISSILE
    det = M.MissileDetonateMessage(sender_id=9, target_id=None, self_detonation=True)
    assert (det.missile_id, det.target_id, det.receiver_id) == (9, None, None)
    assert M.MissileCountRequestMessage(sender_id=0).relevance == 3
    assert len(list(MessageType)) == 18


def test_scan_state_machine_matches_the_oracle_and_keeps_the_quirks():
    """reference modules/Radar.py:96-117: azimuth wraps to elevation_start, unknown modes never move."""
    from oracle import oracle as O
    from zrk_modulation_amd.engine import scan_mode_code, scan_next
    L = O.lib()
    g = np.random.Generator(np.random.PCG64(3))
    for _ in range(300):
        mode = ["horizontal", "vertical", "spiral"][g.integers(3)]
        azr, azs, els, el0 = g.uniform(10, 360), g.uniform(0.5, 200), g.uniform(0, 40), g.uniform(0, 30)
        caz, cel = g.uniform(0, 360), g.uniform(0, 90)
        a, e = caz, cel
        ca, ce = C.c_double(caz), C.c_double(cel)
        for _ in range(50):
            a, e = scan_next(scan_mode_code(mode), azr, azs, els, el0, a, e)
            L.zo_scan_next(O.SCAN_MODES.get(mode, 2), azr, azs, els, el0, C.byref(ca), C.byref(ce))
            assert (a, e) == (ca.value, ce.value)
    # the stock radar flips 0 <-> 180
    a, e = 0.0, 0.0
    seq = []
    for _ in range(4):
        a, e = scan_next(0, 180.0, 180.0, 0.0, 0.0, a, e)
        seq.append(a)
    assert seq == [180.0, 0.0, 180.0, 0.0]


def test_c_scan_advance_matches_python(tmp_path):
    """zrk_scan_advance (host function of the library) against the Python state machine."""
    from zrk_modulation_amd import _lib
    from zrk_modulation_amd.engine import scan_next
    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(11))
    R = 8
    rad = (_lib.ZrkRadar * R)()
    sc = (_lib.ZrkScan * R)()
    py = []
    for r in range(R):
        rad[r].cur_azimuth, rad[r].azimuth_range = g.uniform(0, 360), g.uniform(10, 360)
        rad[r].cur_elevation, rad[r].elevation_range = g.uniform(0, 90), g.uniform(5, 180)
        sc[r].azimuth_speed, sc[r].elevation_speed, sc[r].elevation_start = g.uniform(1, 200), g.uniform(0, 30), g.uniform(0, 20)
        sc[r].mode = r % 3
        py.append([rad[r].cur_azimuth, rad[r].cur_elevation])
    for _ in range(100):
        assert lib.zrk_scan_advance(rad, sc, R) == 0
        for r in range(R):
            py[r] = list(scan_next(sc[r].mode, rad[r].azimuth_range, sc[r].azimuth_speed, sc[r].elevation_speed,
                                   sc[r].elevation_start, py[r][0], py[r][1]))
            assert py[r] == [rad[r].cur_azimuth, rad[r].cur_elevation]


def test_synthetic_scenario_is_seeded_and_matches_the_survey_definition():
    from zrk_modulation_amd import scenario as S
    ids, sp, vel, t0 = S.synthetic_targets(1000, 1236)
    ids2, sp2, _, _ = S.synthetic_targets(1000, 1236)
    assert np.array_equal(sp, sp2) and ids[0] == 1000 and (t0 == 0).all()
    assert (np.abs(sp[:, :2]) <= 60e3).all() and (sp[:, 2] >= 100).all() and (sp[:, 2] <= 12e3).all()
    rs = S.synthetic_radars(4)
    assert rs[2]["position"] == [2000.0, 0.0, 0.0] and rs[0]["azimuth_range"] == 90.0
    assert S.missile_targets(100000, 1000)[:3].tolist() == [0, 100, 200]
    assert S.WORKLOADS["C2"] == (100_000, 4, 1_000) and S.WORKLOADS["C3"] == (1_000_000, 16, 10_000)


def test_replay_log_round_trip_and_bound():
    """ReplayLog alone (no device): frames come back in message order with the reference's field values, old steps
    fall off the end, and the log is columnar (bytes, not objects)."""
    from tests.helpers import Fixture
    from zrk_modulation_amd.modules.Messages import CPPDrawerObjectsMessage
    from zrk_modulation_amd.replay import ReplayLog
    fx = Fixture("stock_config_seed0")
    log = ReplayLog(max_steps=5)
    sent = []
    for T in range(fx.n_ticks):
        ids, types, pos, vis = fx.draw(T)
        t = int(fx.tick_ms[T])
        for k in range(len(ids)):
            p = pos[k].copy()
            log.add_message(t, CPPDrawerObjectsMessage(sender_id=1, obj_id=int(ids[k]), target_type=types[k], coordinates=p,
                                                       is_visible_by_radar=bool(vis[k]), time=t, receiver_id=0))
            p += 1e9                                   # the sender's array moves on; the log holds values
        if len(ids):
            sent.append(T)
    assert log.steps() == [int(fx.tick_ms[T]) for T in sent[-5:]] and log.dropped_steps == len(sent) - 5
    for T in sent[-5:]:
        ids, types, pos, vis = fx.draw(T)
        msgs = log.messages(int(fx.tick_ms[T]))
        assert [m.obj_id for m in msgs] == ids.tolist() and [m.target_type for m in msgs] == types
        assert [m.is_visible_by_radar for m in msgs] == vis.tolist()
        assert np.array_equal(np.array([m.coordinates for m in msgs]).reshape(-1, 3), pos)
        assert all(m.send_time == int(fx.tick_ms[T]) and m.sender_id == 1 and m.receiver_id == 0 for m in msgs)
    assert log.messages(int(fx.tick_ms[sent[0]])) == [] and log.nbytes() < 64 * 1024


def test_synthetic_config_is_in_the_gui_schema():
    """scenario.synthetic_config speaks the schema the reference's GUI saves and its loader reads: same sections
    in the same order as a GUI-saved stock file, same keys per entry, YAML round trip intact."""
    import yaml
    from tests.helpers import Fixture
    from zrk_modulation_amd import scenario as S
    cfg = S.synthetic_config(50, 3, seed=4, launchers=2, missiles_per_launcher=3)
    stock = Fixture("stock_config_seed0").cfg                      # a file the GUI wrote
    assert list(cfg) == list(stock)
    for section in ("simulation", "air_environment", "combat_control_point"):
        assert list(cfg[section]) == list(stock[section])
    assert list(cfg["air_environment"]["targets"][0]) == ["id", "type", "position", "velocity"]      # the GUI's order
    assert set(cfg["air_environment"]["targets"][0]) == set(stock["air_environment"]["targets"][0])
    assert list(cfg["radars"][0]) == list(stock["radars"][0])
    assert list(cfg["missile_launchers"][0]) == list(stock["missile_launchers"][0])
    assert set(stock["missile_launchers"][0]["missiles"][0]) <= set(cfg["missile_launchers"][0]["missiles"][0])
    assert yaml.safe_load(yaml.dump(cfg, allow_unicode=True, sort_keys=False)) == cfg
    ids, sp, vel, _ = S.synthetic_targets(50, 4)
    assert [t["id"] for t in cfg["air_environment"]["targets"]] == ids.tolist()
    assert np.array_equal(np.array([t["position"] for t in cfg["air_environment"]["targets"]]), sp)


def test_trajectory_get_pos_and_unbound_airobject_step_follow_the_reference_arithmetic():
    """reference modules/AirObject.py:23-25, :39-42: three roundings per axis; prev_pos aliases the old pos, None on
    the tick where t equals start_time."""
    from zrk_modulation_amd.modules.AirObject import AirObject, Trajectory
    from zrk_modulation_amd.modules.Manager import Manager
    from zrk_modulation_amd.modules.Timer import Timer
    tr = Trajectory(velocity=(0.1, -0.2, 0.3), start_pos=(1e5, 2e5, 3e3), start_time=0.5)
    t = 1234 / 1000
    want = np.array([1e5, 2e5, 3e3]) + np.array([0.1, -0.2, 0.3]) * (t - 0.5)
    assert np.array_equal(tr.get_pos(t).view(np.uint64), want.view(np.uint64))
    m = Manager()
    m.time = Timer(); m.time.set_dt(10); m.time.set_time(500)
    obj = AirObject(m, 7, np.array([1e5, 2e5, 3e3]), tr)
    obj.step()
    assert obj.prev_pos is None and np.array_equal(obj.pos, tr.get_pos(0.5))
    first = obj.pos
    m.time.set_time(510)
    obj.step()
    assert np.array_equal(obj.prev_pos, first) and np.array_equal(obj.pos, tr.get_pos(0.51))


def test_host_mirror_append_is_amortised():
    """EntityStore's host mirrors grow by doubling: appending objects one at a time does not re-copy the table."""
    from zrk_modulation_amd.store import EntityStore
    st = EntityStore.__new__(EntityStore)
    st.h_ids = np.zeros(0, np.int64)
    copies = 0
    last = None
    for k in range(5000):
        st._happend("h_ids", np.asarray([k], np.int64))
        base = st._hbuf["h_ids"]
        if base is not last:
            copies += 1
            last = base
    assert copies <= 5 and np.array_equal(st.h_ids, np.arange(5000))


def test_host_waits_are_bounded_and_end_in_a_state_error():
    """Every host-side wait of the library (helper threads, collectives posted ticks ago, side-stream compactions) goes
    through one bounded spin: a condition that never comes -- a stalled flag, a ring nobody empties -- ends in
    ZRK_E_STATE after the limit instead of a process that spins for ever (include/zrk_hot.h, ZRK_HOST_WAIT_MS)."""
    import time
    from zrk_modulation_amd import _lib
    L = _lib.load()
    for what in (0, 1):
        t0 = time.perf_counter()
        assert L.zrk_selftest_host_wait(what, 40) == _lib.ZRK_E_STATE
        took = time.perf_counter() - t0
        assert 0.03 < took < 2.0, f"a 40 ms limit took {took:.3f} s"


def test_poisoned_lists_are_rejected_by_every_decoder():
    """A rank whose list was never handed over sends count -1 (k_wait_flag's give-up): nothing downstream may read such
    a buffer as a tick's detections."""
    import torch
    from zrk_modulation_amd import exchange as X
    R, n, ev = 5, 300, 8
    vis = np.zeros(n, np.uint32)
    vis[[3, 64, 65, 200]] = [1, 3, 16, 31]
    words = X.union_bits_words(n, R, 16)
    good = np.concatenate([X.encode_union_bits(vis, R, words), np.zeros(1 + ev, np.int64)])
    bad = good.copy()
    bad[0] = -1
    ok = torch.from_numpy(np.stack([good, good]))
    idx, msk = X.decode_union_bits(ok, R, [0, 1000], ev)
    assert idx.tolist() == [3, 64, 65, 200, 1003, 1064, 1065, 1200] and msk.tolist() == [1, 3, 16, 31] * 2
    assert X.decode_events(ok, ev) == []
    for rows in ([bad, good], [good, bad]):
        g = torch.from_numpy(np.stack(rows))
        with pytest.raises(X.PoisonedList):
            X.decode_union_bits(g, R, [0, 1000], ev)
        with pytest.raises(X.PoisonedList):
            X.decode_events(g, ev)


def test_helper_threads_planned_from_the_cores_a_rank_has_to_itself():
    """zrk_exchange_plan_helpers (ABI 11): two helper threads per rank where the rank has three usable cores or more to itself,
    one below that (the side stream's thread then issues the collectives too) -- eight ranks under a 16-thread quota take the
    two-thread-per-rank path.  Checked in a child process whose affinity mask this test sets; no GPU is touched."""
    import os
    import subprocess
    import sys
    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from zrk_modulation_amd import _lib\n"
        "lib = _lib.load()\n"
        "cores = sorted(os.sched_getaffinity(0))\n"
        "out = []\n"
        "for mask, world in ((cores[:8], 8), (cores[:8], 2), (cores[:6], 2), (cores[:5], 2), (cores[:3], 1), (cores[:2], 1)):\n"
        "    os.sched_setaffinity(0, mask)\n"
        "    out.append((len(mask), world, lib.zrk_exchange_plan_helpers(world)))\n"
        "os.environ['ZRK_HELPERS'] = '2'\n"
        "out.append((len(mask), 8, lib.zrk_exchange_plan_helpers(8)))\n"
        "print(out)\n" % str(ROOT))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = eval(res.stdout.strip().splitlines()[-1])
    for cores, world, helpers in got[:-1]:
        # (the cgroup quota may cap the cores below the mask: then fewer helpers, never more)
        assert helpers in (1, 2) and (helpers == 1 or cores // world >= 3), (cores, world, helpers)
    by = {(c, w): h for c, w, h in got[:-1]}
    if len(os.sched_getaffinity(0)) >= 8:
        assert by[(8, 8)] == 1 and by[(5, 2)] == 1 and by[(2, 1)] == 1
    assert got[-1][2] == 2                                   # ZRK_HELPERS forces
