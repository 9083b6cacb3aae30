"""The drop-in module stack (Manager + AirEnv + SectorRadar + Missile on the device, MissileLauncher
and CombatControlPoint on the host) run exactly as the reference's main.py runs a scenario, compared
tick by tick with what the reference produced (tests/golden, captured by gen_golden.py)."""
import json

import numpy as np
import pytest

from tests.helpers import ALL_FIXTURES, Fixture

pytestmark = pytest.mark.gpu


def _scripted_commander(mod):
    """Test stand-in for the command post in scripted scenes; the class NAME gives it the CCP's slot in
    the schedule (same trick as tests/golden/gen_golden.py uses against the reference)."""
    from zrk_modulation_amd.modules.BaseModel import BaseModel
    from zrk_modulation_amd.modules.constants import MessageType
    from zrk_modulation_amd.modules.Messages import CPPLaunchMissileRequestMessage

    class CombatControlPoint(BaseModel):
        def __init__(self, manager, id, script):
            super().__init__(manager, id, np.zeros(3))
            self.script = script
            self.known = {}

        def step(self):
            now = self._manager.time.get_time()
            for msg in self._manager.give_messages_by_type(MessageType.ACTIVE_OBJECTS):
                for obj in msg.active_objects:
                    self.known.setdefault(obj.id, obj)
            for t_ms, launcher_id, target_id in self.script:
                if t_ms == now and target_id in self.known:
                    obj = self.known[target_id]
                    self._manager.add_message(CPPLaunchMissileRequestMessage(
                        time=now, sender_id=self.id, receiver_id=launcher_id, target=obj, target_position=obj.pos,
                        radar_id=0))
    return CombatControlPoint


def build(fx, replay=None):
    import zrk_modulation_amd.main as M
    from zrk_modulation_amd.modules.Radar import SectorRadar
    cfg = fx.cfg
    if fx.scene.get("script") is None:
        manager, objs = M.create_objects_from_config(cfg, replay=replay)
        return manager, [o for o in objs.values() if isinstance(o, SectorRadar)]
    cfg2 = dict(cfg)
    cfg2["combat_control_point"] = {}
    # same order as the reference builder: AirEnv, radars, launchers, commander, targets
    targets = cfg2["air_environment"].get("targets", [])
    cfg2["air_environment"] = dict(cfg2["air_environment"], targets=[])
    manager, objs = M.create_objects_from_config(cfg2)
    manager.add_module(_scripted_commander(M)(manager, 0, [tuple(s) for s in fx.scene["script"]]))
    from zrk_modulation_amd.modules.AirObject import Trajectory
    from zrk_modulation_amd.modules.utils import Target, TargetType
    ae = objs[cfg["air_environment"]["id"]]
    for tc in targets:
        pos, vel = np.array(tc["position"]), np.array(tc["velocity"])
        ae.add_target(Target(manager, tc["id"], pos, Trajectory(vel, pos, 0.0), getattr(TargetType, tc["type"])))
    return manager, [o for o in objs.values() if isinstance(o, SectorRadar)]


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_module_stack_reproduces_reference_run(name):
    from zrk_modulation_amd.modules.constants import MessageType
    fx = Fixture(name)
    np.random.seed(fx.scene["seed"])
    real_normal = np.random.normal
    if fx.scene.get("zero_noise"):
        np.random.normal = lambda loc, scale, size=None: np.zeros(size)
    try:
        manager, radars = build(fx)
        hist = {}
        for T in range(fx.n_ticks):
            t = int(fx.tick_ms[T])
            manager.run_simulation(t + fx.dt)
            msgs = manager.messages.get(t, [])
            act = [m for m in msgs if m.type == MessageType.ACTIVE_OBJECTS][0].active_objects
            assert [o.id for o in act] == fx.active_ids(T).tolist(), f"live ids differ at t={t}"
            P = np.array([o.pos for o in act]).reshape(len(act), 3)
            dig = int(np.bitwise_xor.reduce(np.ascontiguousarray(P).view(np.uint64).ravel())) if len(act) else 0
            assert dig == int(fx.pos_digest[T]), f"position bits differ at t={t}"
            found = [m for m in msgs if m.type == MessageType.FOUND_OBJECTS]
            assert [m.sender_id for m in found] == [r.id for r in radars]
            for r, m in enumerate(found):
                assert [o.id for o in m.visible_objects] == fx.found(T, r).tolist(), f"radar {r} at t={t}"
            state = np.array([[r.current_azimuth, r.current_elevation] for r in radars], np.float64).reshape(-1, 2)
            assert np.array_equal(state, fx.radar_state[T])
            if T in fx.samp_index:
                k = fx.samp_index[T]
                lo, hi = fx.samp_off[k], fx.samp_off[k + 1]
                assert np.array_equal(P, fx.pos[lo:hi])
                pv = np.array([o.prev_pos is not None for o in act], dtype=bool)
                assert np.array_equal(pv, fx.prev_valid[lo:hi].astype(bool)), f"prev_pos None-ness at t={t}"
                PP = np.array([o.prev_pos if o.prev_pos is not None else np.zeros(3) for o in act]).reshape(len(act), 3)
                assert np.array_equal(PP[pv], fx.prev[lo:hi][pv]), f"prev_pos differs at t={t}"
            det = [[t, m.missile_id, -1 if m.target_id is None else m.target_id, int(m.self_detonation)]
                   for m in msgs if m.type == MessageType.MISSILE_DETONATE]
            assert det == fx.rows_at(fx.detonations, t).tolist(), f"detonations differ at t={t}"
            draw = [m for m in msgs if m.type == MessageType.DRAW_OBJECTS]
            ids, types, pos, vis = fx.draw(T)
            assert [m.obj_id for m in draw] == ids.tolist(), f"objects sent to the GUI differ at t={t}"
            assert [getattr(m.target_type, "name", None) or str(m.target_type) for m in draw] == types
            assert [bool(m.is_visible_by_radar) for m in draw] == vis.tolist(), f"visibility flags differ at t={t}"
            assert np.array_equal(np.array([m.coordinates for m in draw]).reshape(len(draw), 3), pos), f"GUI coordinates differ at t={t}"
            for m in msgs:
                hist[m.type.name] = hist.get(m.type.name, 0) + 1
                if m.type == MessageType.LAUNCH_SUCCESSFUL:
                    row = [k for k, r in enumerate(fx.launch_ok) if r[0] == t and r[1] == m.missile.id]
                    assert row, f"unexpected launch of {m.missile.id} at t={t}"
                    tr = m.missile.trajectory
                    want = fx.launch_traj[row[0]]
                    assert np.array_equal(np.concatenate([tr.velocity, tr.start_pos, [tr.start_time]]), want)
                elif m.type == MessageType.LAUNCH_CANCELLED:
                    row = [k for k, r in enumerate(fx.launch_cancel) if r[0] == t and r[1] == m.missile.id]
                    assert row and fx.reasons[row[0]] == m.reason
        assert hist == fx.histogram, "message-type histogram of the whole run differs from the reference's"
    finally:
        np.random.normal = real_normal


def test_handles_keep_identity_and_freeze_after_removal():
    """CCP-style consumers keep references across ticks: a removed object's pos / prev_pos stay at
    their last values, and missiles chasing it see the same frozen position on the device."""
    from zrk_modulation_amd.modules.constants import MessageType
    fx = Fixture("missiles")
    np.random.seed(fx.scene["seed"])
    manager, _ = build(fx)
    seen = {}
    frozen = {}
    for T in range(fx.n_ticks):
        t = int(fx.tick_ms[T])
        manager.run_simulation(t + fx.dt)
        act = [m for m in manager.messages[t] if m.type == MessageType.ACTIVE_OBJECTS][0].active_objects
        live = {o.id for o in act}
        for o in act:
            assert seen.setdefault(o.id, o) is o          # same Python object every tick
        for oid, o in seen.items():
            if oid not in live:
                if oid not in frozen:
                    frozen[oid] = (o.pos.copy(), None if o.prev_pos is None else o.prev_pos.copy())
                assert np.array_equal(o.pos, frozen[oid][0])
    assert frozen, "scene should remove at least one object"


@pytest.mark.parametrize("name", ["stock_config_seed0", "stock_simulation_config_seed1"])
def test_columnar_replay_log_holds_what_the_gui_replays(name):
    """The same run with Manager(replay=ReplayLog(max_steps=...)): DRAW_OBJECTS messages never enter the per-tick
    lists, the log answers the GUI's query with the reference's contents for the steps it still holds, and it
    holds no more than it was told to."""
    from zrk_modulation_amd.modules.constants import MessageType
    from zrk_modulation_amd.replay import ReplayLog
    fx = Fixture(name)
    np.random.seed(fx.scene["seed"])
    keep = 7
    log = ReplayLog(max_steps=keep)
    manager, _ = build(fx, replay=log)
    for T in range(fx.n_ticks):
        t = int(fx.tick_ms[T])
        manager.run_simulation(t + fx.dt)
        assert not [m for m in manager.messages.get(t, []) if m.type == MessageType.DRAW_OBJECTS]
        assert len(log.steps()) <= keep
    sent = [T for T in range(fx.n_ticks) if fx.draw_off[T + 1] > fx.draw_off[T]]
    assert log.steps() == [int(fx.tick_ms[T]) for T in sent[-keep:]]
    for T in sent[-keep:]:
        t = int(fx.tick_ms[T])
        ids, types, pos, vis = fx.draw(T)
        got = manager.give_messages_by_type(MessageType.DRAW_OBJECTS, step_time=t)      # what UI/PolygonEditor.py:631 calls
        assert [m.obj_id for m in got] == ids.tolist()
        assert [getattr(m.target_type, "name", None) or str(m.target_type) for m in got] == types
        assert [m.is_visible_by_radar for m in got] == vis.tolist()
        assert np.array_equal(np.array([m.coordinates for m in got]).reshape(len(got), 3), pos)
        f_ids, f_types, f_pos, f_vis = log.frame(t)
        assert np.array_equal(f_ids, ids) and np.array_equal(f_pos, pos) and np.array_equal(f_vis, vis)
    assert log.dropped_steps == len(sent) - min(keep, len(sent))


def test_gui_schema_scene_runs_the_same_through_modules_and_headless():
    """One synthetic scene, twice: as a configuration in the GUI's schema through create_objects_from_config and the
    module stack (noise draws patched to zero), and as arrays through the headless engine with noise off -- same
    detections per radar, every tick."""
    import zrk_modulation_amd.main as M
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.engine import HotPathEngine
    from zrk_modulation_amd.modules.constants import MessageType
    from zrk_modulation_amd.modules.Radar import SectorRadar
    n, R, seed, dt, ticks = 400, 3, 12, 500, 12
    cfg = S.synthetic_config(n, R, seed, launchers=0, time_step=dt, duration=dt * ticks)
    cfg["combat_control_point"] = {}                         # no command post: nothing launches, nothing is drawn
    real_normal = np.random.normal
    np.random.normal = lambda loc, scale, size=None: np.zeros(size)
    try:
        manager, objs = M.create_objects_from_config(cfg)
        radars = [o for o in objs.values() if isinstance(o, SectorRadar)]
        ids, sp, vel, t0 = S.synthetic_targets(n, seed)
        eng = HotPathEngine(device="cuda:0", dt_ms=dt, seed=0, noise="off")
        eng.load(ids, sp, vel, t0, S.synthetic_radars(R)).enable_lists()
        total = 0
        for T in range(ticks):
            manager.run_simulation((T + 1) * dt)
            eng.run(1)
            found = [m for m in manager.messages.get(T * dt, []) if m.type == MessageType.FOUND_OBJECTS]
            assert [m.sender_id for m in found] == [r.id for r in radars]
            lists = eng.detections()
            for r, m in enumerate(found):
                assert [o.id for o in m.visible_objects] == ids[lists[r]].tolist(), f"radar {r} tick {T}"
                total += len(lists[r])
        assert total > 50
    finally:
        np.random.normal = real_normal


def test_bulk_scenario_load_is_linear_and_equals_one_by_one():
    """A GUI-schema scenario of 1e5 targets loads through create_objects_from_config in seconds (one table append),
    and a bulk-loaded AirEnv is the same table as one filled by add_target calls."""
    import time
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.main import create_objects_from_config
    cfg = S.synthetic_config(100_000, 2, seed=3, duration=30)
    cfg.pop("combat_control_point")          # the command post is host code, quadratic like the reference's: not the loader
    cfg["missile_launchers"] = []
    t = time.perf_counter()
    mgr, by_id = create_objects_from_config(cfg, device="cuda:0")
    mgr.run_simulation(30)
    took = time.perf_counter() - t
    assert took < 60, f"loading and 3 ticks of 1e5 targets took {took:.1f} s"
    st = by_id[1].store
    assert st.n_uploaded == 100_000 and int(st.d_alive[:st.n_uploaded].sum().item()) == 100_000
    small = S.synthetic_config(300, 2, seed=4, duration=50)
    ma, a = create_objects_from_config(small, device="cuda:0")
    mb, b = create_objects_from_config(dict(small, air_environment=dict(small["air_environment"], targets=[])), device="cuda:0")
    from zrk_modulation_amd.modules.AirObject import Trajectory
    from zrk_modulation_amd.modules.utils import Target, TargetType
    for tc in small["air_environment"]["targets"]:
        pos, vel = np.array(tc["position"]), np.array(tc["velocity"])
        b[1].add_target(Target(mb, tc["id"], pos, Trajectory(velocity=vel, start_pos=pos, start_time=0.0), TargetType.AIR_PLANE))
    np.random.seed(1); ma.run_simulation(50)
    np.random.seed(1); mb.run_simulation(50)
    sa, sb = a[1].store, b[1].store
    assert np.array_equal(sa.host_pos("cur").view(np.uint64), sb.host_pos("cur").view(np.uint64))
    assert np.array_equal(sa.h_ids[:300], sb.h_ids[:300])


def test_smooth_objects_and_get_pos_work_on_device_backed_objects():
    """SectorRadar.smooth_objects (reference modules/Radar.py:138-142) on a list of device-backed objects: the same
    draws from numpy's global stream as the reference's loop, added in place; SectorRadar.start (reference :207-218)."""
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.main import create_objects_from_config
    cfg = S.synthetic_config(64, 1, seed=8, duration=10)
    mgr, by_id = create_objects_from_config(cfg, device="cuda:0")
    mgr.run_simulation(10)
    env, radar = by_id[1], by_id[10_000]
    objs = [env._handles[k] for k in (5, 9, 2)]
    before = [o.pos.copy() for o in objs]
    np.random.seed(33)
    want = [b + np.random.normal(0, 5, 3) for b in before]
    np.random.seed(33)
    radar.smooth_objects(objs)
    for o, w in zip(objs, want):
        assert np.array_equal(o.pos.view(np.uint64), w.view(np.uint64))
    o = objs[0]
    assert np.array_equal(o.trajectory.get_pos(0.0), o.trajectory.start_pos)
    from zrk_modulation_amd.modules.constants import MessageType
    active = mgr.give_messages_by_type(MessageType.ACTIVE_OBJECTS, step_time=0)[0].active_objects
    radar.elevation_speed = 15.0
    seen = radar.start(active)
    assert len(seen) == int(radar.azimuth_range / radar.azimuth_speed * radar.elevation_range / radar.elevation_speed)
    assert all(isinstance(lst, list) for lst in seen)
