#!/usr/bin/env python3
"""Capture golden vectors from the reference itself (Ollegorii/ZRK_modulation).

Runs ONLY in the build container, where the reference is mounted read-only at
/root/reference; the GPU box never sees the reference.  The script imports the
reference's modules, drives `Manager.run_simulation` one tick at a time and dumps
inputs + observed outputs into small .npz fixtures next to this file.  Nothing of the
reference's source text is stored: scenes are described by the YAML *schema* the
reference loads (main.py:35-149) as plain data, outputs are ids / float64 arrays.

    python tests/golden/gen_golden.py            # regenerate every fixture

Per fixture (see `capture`):
    scene            JSON: config dict (YAML schema) + script + seed + zero_noise
    tick_ms[T]
    act_ids / act_off            live ids per tick (ACTIVE_OBJECTS), ragged
    pos / prev / prev_valid      float64 state of those objects after the tick (sampled ticks)
    pos_digest[T]                xor of the uint64 bit patterns of all live positions, every tick
    found_ids / found_off        ids per (tick, radar) in FoundObjectsMessage order, ragged
    radar_state[T,R,2]           (current_azimuth, current_elevation) after the tick
    detonations[k,4]             (t_ms, missile_id, target_id | -1, self_detonation)
    launch_req[k,4]              (t_ms, launcher_id, target_id, radar_id)         LAUNCH_COMMAND (the command post's requests, in order)
    launch_cmd[k,4]              (t_ms, launcher_id, missile_id, target_id)      LAUNCH_MISSILE
    launch_ok[k,3] + launch_traj[k,7]   (t_ms, missile_id, target_id), (V, start_pos, start_time)
    launch_cancel[k,2] + reasons (t_ms, missile_id), str
    new_missile[k,2]             (t_ms, missile_id)                              NEW_MISSILE
    draw_ids / draw_off / draw_vis / draw_pos / draw_type   DRAW_OBJECTS per tick, in message order (ragged):
                                 obj_id, is_visible_by_radar, coordinates (float64), index into draw_types
    histogram                    JSON {MessageType.name: count}
"""
import json
import os
import sys
import warnings
from pathlib import Path

import numpy as np
import yaml

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))
warnings.filterwarnings("ignore", category=RuntimeWarning)

from modules.Manager import Manager                      # noqa: E402
from modules.Timer import Timer                          # noqa: E402
from modules.AirEnv import AirEnv                        # noqa: E402
from modules.Radar import SectorRadar                    # noqa: E402
from modules.utils import Target, TargetType             # noqa: E402
from modules.AirObject import Trajectory                 # noqa: E402
from modules.BaseModel import BaseModel                  # noqa: E402
from modules.MissileLauncher import MissileLauncher      # noqa: E402
from modules.Missile import Missile                      # noqa: E402
from modules.Messages import CPPLaunchMissileRequestMessage  # noqa: E402
from modules.constants import MessageType                # noqa: E402
import main as ref_main                                  # noqa: E402


class CombatControlPoint(BaseModel):
    """Scripted stand-in for the command post in synthetic scenes: the class NAME puts it in the
    reference's CCP scheduling slot (modules/Manager.py:123-127).  At scripted ticks it sends the
    same launch request the real one would (modules/CCP.py:306-314)."""

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
                    time=now, sender_id=self.id, receiver_id=launcher_id, target=obj,
                    target_position=obj.pos, radar_id=0))


def build_scripted(scene):
    """Same construction order as main.create_objects_from_config (main.py:35-149), with the
    scripted commander in place of the real one."""
    cfg = scene["config"]
    manager = Manager()
    timer = Timer(); timer.set_dt(cfg["simulation"]["time_step"]); manager.time = timer
    ae = AirEnv(manager, cfg["air_environment"]["id"], np.array(cfg["air_environment"]["position"]))
    manager.add_module(ae)
    radars = []
    for rc in cfg.get("radars", []):
        r = SectorRadar(manager, rc["id"], np.array(rc["position"]), rc["azimuth_start"], rc["elevation_start"],
                        rc["max_distance"], rc["azimuth_range"], rc["elevation_range"], rc["azimuth_speed"],
                        rc["elevation_speed"], rc["scan_mode"])
        manager.add_module(r); radars.append(r)
    for lc in cfg.get("missile_launchers", []):
        ml = MissileLauncher(manager, lc["id"], np.array(lc["position"]), lc.get("max_missiles", 5))
        for mc in lc.get("missiles", []):
            ml.add_missile(Missile(manager, mc["id"], np.array(lc["position"]), mc.get("velocity", 1000),
                                   mc.get("explosion_radius", 50), mc.get("life_time", 60)))
        manager.add_module(ml)
    manager.add_module(CombatControlPoint(manager, 0, [tuple(s) for s in scene.get("script", [])]))
    for tc in cfg["air_environment"].get("targets", []):
        pos = np.array(tc["position"]); vel = np.array(tc["velocity"])
        ae.add_target(Target(manager, tc["id"], pos, Trajectory(vel, pos, 0.0), getattr(TargetType, tc["type"])))
    return manager, radars


def capture(scene, sample_every=1):
    cfg = scene["config"]
    np.random.seed(scene["seed"])
    real_normal = np.random.normal
    if scene.get("zero_noise"):
        np.random.normal = lambda loc, scale, size=None: np.zeros(size)   # numpy patched, not the reference
    try:
        if scene.get("script") is None:
            manager, objs = ref_main.create_objects_from_config(cfg)
            radars = [o for o in objs.values() if isinstance(o, SectorRadar)]
        else:
            manager, radars = build_scripted(scene)
        dt = cfg["simulation"]["time_step"]; duration = cfg["simulation"]["duration"]
        T = 0
        rec = dict(tick_ms=[], act_ids=[], act_off=[0], pos=[], prev=[], prev_valid=[], samp_tick=[],
                   samp_off=[0], pos_digest=[], found_ids=[], found_off=[0], radar_state=[], detonations=[],
                   launch_req=[], launch_cmd=[], launch_ok=[], launch_traj=[], launch_cancel=[], reasons=[], new_missile=[],
                   draw_ids=[], draw_off=[0], draw_vis=[], draw_pos=[], draw_type=[])
        draw_types = []
        hist = {}
        t = 0
        while t < duration:
            manager.run_simulation(t + dt)          # exactly one tick (modules/Manager.py:116)
            msgs = manager.messages.get(t, [])
            rec["tick_ms"].append(t)
            act = [m for m in msgs if m.type == MessageType.ACTIVE_OBJECTS][0].active_objects
            ids = [o.id for o in act]
            rec["act_ids"] += ids; rec["act_off"].append(len(rec["act_ids"]))
            P = np.array([np.asarray(o.pos, dtype=np.float64) for o in act]).reshape(len(act), 3)
            rec["pos_digest"].append(int(np.bitwise_xor.reduce(P.view(np.uint64).ravel())) if len(act) else 0)
            if T % sample_every == 0 or t + dt >= duration:
                rec["samp_tick"].append(T)
                rec["pos"].append(P)
                pv = np.array([0 if o.prev_pos is None else 1 for o in act], np.uint8)
                PP = np.array([np.zeros(3) if o.prev_pos is None else np.asarray(o.prev_pos, np.float64)
                               for o in act]).reshape(len(act), 3)
                rec["prev"].append(PP); rec["prev_valid"].append(pv)
                rec["samp_off"].append(rec["samp_off"][-1] + len(act))
            found = [m for m in msgs if m.type == MessageType.FOUND_OBJECTS]
            assert [m.sender_id for m in found] == [r.id for r in radars]
            for m in found:
                rec["found_ids"] += [o.id for o in m.visible_objects]
                rec["found_off"].append(len(rec["found_ids"]))
            rec["radar_state"].append([[r.current_azimuth, r.current_elevation] for r in radars])
            for m in msgs:
                hist[m.type.name] = hist.get(m.type.name, 0) + 1
                if m.type == MessageType.MISSILE_DETONATE:
                    rec["detonations"].append([t, m.missile_id, -1 if m.target_id is None else m.target_id,
                                               int(m.self_detonation)])
                elif m.type == MessageType.LAUNCH_COMMAND:
                    rec["launch_req"].append([t, m.receiver_id, m.target.id, m.radar_id])
                elif m.type == MessageType.LAUNCH_MISSILE:
                    rec["launch_cmd"].append([t, m.sender_id, m.receiver_id, m.target.id])
                elif m.type == MessageType.LAUNCH_SUCCESSFUL:
                    tr = m.missile.trajectory
                    rec["launch_ok"].append([t, m.missile.id, m.target_id])
                    rec["launch_traj"].append(list(tr.velocity) + list(tr.start_pos) + [tr.start_time])
                elif m.type == MessageType.LAUNCH_CANCELLED:
                    rec["launch_cancel"].append([t, m.missile.id]); rec["reasons"].append(m.reason)
                elif m.type == MessageType.NEW_MISSILE:
                    rec["new_missile"].append([t, m.missile.id])
                elif m.type == MessageType.DRAW_OBJECTS:
                    name = getattr(m.target_type, "name", None) or str(m.target_type)
                    if name not in draw_types:
                        draw_types.append(name)
                    rec["draw_ids"].append(m.obj_id); rec["draw_vis"].append(int(bool(m.is_visible_by_radar)))
                    rec["draw_pos"].append(np.asarray(m.coordinates, dtype=np.float64).copy())
                    rec["draw_type"].append(draw_types.index(name))
            rec["draw_off"].append(len(rec["draw_ids"]))
            t += dt; T += 1
    finally:
        np.random.normal = real_normal
    R = len(radars)
    out = dict(
        scene=json.dumps(scene), histogram=json.dumps(hist), reasons=json.dumps(rec["reasons"]),
        tick_ms=np.array(rec["tick_ms"], np.int64),
        act_ids=np.array(rec["act_ids"], np.int64), act_off=np.array(rec["act_off"], np.int64),
        samp_tick=np.array(rec["samp_tick"], np.int64), samp_off=np.array(rec["samp_off"], np.int64),
        pos=np.concatenate(rec["pos"]) if rec["pos"] else np.zeros((0, 3)),
        prev=np.concatenate(rec["prev"]) if rec["prev"] else np.zeros((0, 3)),
        prev_valid=np.concatenate(rec["prev_valid"]) if rec["prev_valid"] else np.zeros(0, np.uint8),
        pos_digest=np.array(rec["pos_digest"], np.uint64),
        found_ids=np.array(rec["found_ids"], np.int64), found_off=np.array(rec["found_off"], np.int64),
        radar_state=np.array(rec["radar_state"], np.float64).reshape(T, R, 2),
        detonations=np.array(rec["detonations"], np.int64).reshape(-1, 4),
        launch_req=np.array(rec["launch_req"], np.int64).reshape(-1, 4),
        launch_cmd=np.array(rec["launch_cmd"], np.int64).reshape(-1, 4),
        launch_ok=np.array(rec["launch_ok"], np.int64).reshape(-1, 3),
        launch_traj=np.array(rec["launch_traj"], np.float64).reshape(-1, 7),
        launch_cancel=np.array(rec["launch_cancel"], np.int64).reshape(-1, 2),
        new_missile=np.array(rec["new_missile"], np.int64).reshape(-1, 2),
        draw_ids=np.array(rec["draw_ids"], np.int64), draw_off=np.array(rec["draw_off"], np.int64),
        draw_vis=np.array(rec["draw_vis"], np.uint8), draw_type=np.array(rec["draw_type"], np.int32),
        draw_pos=np.array(rec["draw_pos"], np.float64).reshape(-1, 3), draw_types=json.dumps(draw_types),
    )
    return out


def yaml_scene(name, fname, seed):
    with open(REF / fname) as f:
        cfg = yaml.safe_load(f)
    return dict(name=name, config=cfg, script=None, seed=seed, zero_noise=False)


def radar(id, pos, az0, el0, rng, azr, elr, azs, els, mode="horizontal"):
    return dict(id=id, position=list(map(float, pos)), azimuth_start=float(az0), elevation_start=float(el0),
                max_distance=float(rng), azimuth_range=float(azr), elevation_range=float(elr),
                azimuth_speed=float(azs), elevation_speed=float(els), scan_mode=mode)


def target(id, pos, vel, type="AIR_PLANE"):
    return dict(id=id, type=type, position=list(map(float, pos)), velocity=list(map(float, vel)))


def launcher(id, pos, missiles):
    return dict(id=id, position=list(map(float, pos)), max_missiles=max(5, len(missiles)),
                missiles=[dict(id=m[0], velocity=m[1], explosion_radius=m[2], life_time=m[3]) for m in missiles])


def edge_scene(zero_noise, seed):
    """Targets exactly on range / azimuth / elevation edges, at the radar itself, below it, a sector
    whose upper edge passes 360, a vertical-mode radar, an unknown scan mode, two radars seeing one
    target."""
    T = [
        target(1, (1000, 0, 0), (0, 0, 0)),             # dist == max_distance, az 0, el 0
        target(2, (0, 1000, 0), (0, 0, 0)),             # az 90 exactly
        target(3, (-1000, 0, 0), (0, 0, 0)),            # az 180
        target(4, (0, -1000, 0), (0, 0, 0)),            # az 270
        target(5, (0, 0, 0), (0, 0, 0)),                # at the radar: dist 0 -> NaN elevation
        target(6, (500, 500, 0), (0, 0, 0)),            # az 45
        target(7, (300, 0, 300.5), (0, 0, 0)),          # el just above 45 (el == 45 exactly is libm-dependent:
                                                        #   numpy/SVML asin gives 45.0, glibc 44.99999999999999)
        target(8, (0, 0, 700), (0, 0, 0)),              # zenith, el 90, az 0
        target(9, (300, 300, -200), (0, 0, 0)),         # below the radar: el in (90,180)
        target(10, (1000.0000000001, 0, 0), (0, 0, 0)), # just outside range
        target(11, (600, 10, 50), (-40, 0, 0)),         # walks through the origin region
        target(12, (-900, -5, 20), (75, 3, 1)),         # crosses az 180 -> 0 side
        target(13, (100, -400, 30), (0, 90, 0)),        # crosses the 0/360 seam
        target(14, (2000, 2000, 100), (-130, -120, 5)), # enters range later
        target(15, (999.5, 0, 31.6), (0, 0, 0)),        # within a hair of the range sphere
        target(16, (700, 700, 141.4), (0.5, -0.5, 0)),  # rides along the range edge of radar 20
        target(17, (0, 0, -300), (0, 0, 0)),            # nadir: el 90 via the %180 wrap (asin -> -90)
        target(18, (250, 0, -1e-13), (0, 0, 0)),        # tiny negative dz: el -> 180.0 after %180
    ]
    R = [
        radar(20, (0, 0, 0), 0, 0, 1000, 90, 90, 45, 15),                # stepping horizontal
        radar(21, (0, 0, 0), 350, 0, 5000, 30, 180, 10, 0),              # hi edge beyond 360: no wrap
        radar(22, (0, 0, 0), 0, 0, 5000, 360, 180, 10, 30, "vertical"),  # vertical mode
        radar(23, (10, -20, 5), 180, 0, 3000, 180, 90, 180, 0),          # flips 180<->0 like the stock one
        radar(24, (0, 0, 0), 100, 20, 2500, 45, 30, 7, 3, "spiral"),     # unknown mode: never moves
    ]
    cfg = dict(simulation=dict(time_step=500, duration=30000),
               air_environment=dict(id=999, position=[0.0, 0.0, 0.0], targets=T),
               radars=R, missile_launchers=[], combat_control_point=dict(id=0))
    return dict(name="edges_zero_noise" if zero_noise else "edges", config=cfg, script=[], seed=seed,
                zero_noise=zero_noise)


def missile_scene(seed):
    """A launch cancelled then retried (the launcher re-appends a cancelled missile,
    modules/MissileLauncher.py:126-129), two missiles on one target (the second keeps chasing a
    removed object until its timer self-detonates it), a launch against a removed target, a launch
    against a missile (always cancelled), a static target (NaN unit velocity)."""
    T = [
        target(101, (8000, 6000, 2000), (-120, 0, 0)),
        target(102, (-7000, 9000, 1500), (90, -60, 0)),
        target(103, (20000, 0, 3000), (300, 0, 0)),           # fast, receding: discriminant < 0 for a slow missile
        target(104, (4000, -3000, 1000), (0, 0, 0)),          # static target (NaN unit velocity)
        target(105, (30000, 30000, 5000), (-50, -50, 0)),     # beyond a short life_time
        target(106, (1500, 1500, 500), (-200, -200, 0)),
        target(107, (5000, 0, 1000), (0, 300, 0)),            # crossing: b ~ 0, discriminant < 0 for a slow missile
    ]
    # launchers pop from the END of their list
    L = [launcher(3, (0, 0, 0), [(3005, 1000, 150, 60), (3004, 1000, 150, 60), (3003, 1000, 150, 2.0),
                                 (3002, 200, 150, 60), (3001, 1000, 150, 60)]),
         launcher(4, (500, -500, 0), [(4003, 900, 100, 50), (4002, 900, 100, 50), (4001, 900, 100, 50)])]
    R = [radar(5, (0, 0, 10), 0, 0, 60000, 360, 180, 10, 0),
         radar(6, (1000, 1000, 10), 0, 0, 40000, 180, 90, 180, 0)]
    script = [
        (0, 3, 101), (0, 4, 101),          # 3001 and 4001 on one target
        (200, 3, 107),                     # 3002 (200 m/s) vs 300 m/s crossing: discriminant < 0, re-appended
        (400, 3, 103),                     # 3002 again vs 300 m/s receding: both roots negative, re-appended
        (600, 3, 106),                     # 3002 again, closing target: launches
        (800, 3, 105),                     # 3003 (2 s): too far, re-appended
        (1000, 3, 104),                    # 3003 vs static target: NaN -> 'not positive'
        (1200, 3, 106),                    # 3003 again, close target
        (1400, 4, 3001),                   # 4002 against a missile: cancelled
        (1600, 3, 102),                    # 3004
        (14000, 4, 101),                   # 4002 against an already removed target
    ]
    cfg = dict(simulation=dict(time_step=200, duration=80000),
               air_environment=dict(id=999, position=[0.0, 0.0, 0.0], targets=T),
               radars=R, missile_launchers=L, combat_control_point=dict(id=0))
    return dict(name="missiles", config=cfg, script=[list(s) for s in script], seed=seed, zero_noise=False)


def solve_branch_scene(seed):
    """Noise patched to zero so the degenerate |a| < 1e-6 branch of the launch solve
    (modules/Missile.py:70-78) is reachable: target speed == missile speed == 300."""
    T = [
        target(201, (-300, 5000, 0), (300, 0, 0)),       # at t = 1.0 s: d perpendicular to v_t -> |b| < 1e-6
        target(202, (1000, 0, 1000), (300, 0, 0)),       # receding: t = -c/b <= 0
        target(203, (9000, 100, 500), (-300, 0, 0)),     # approaching: linear branch succeeds
    ]
    L = [launcher(3, (0, 0, 0), [(3003, 300, 150, 60), (3002, 300, 150, 60), (3001, 300, 150, 60)])]
    R = [radar(5, (0, 0, 10), 0, 0, 60000, 360, 180, 10, 0)]
    script = [(500, 3, 201), (1000, 3, 202), (1500, 3, 203)]
    cfg = dict(simulation=dict(time_step=500, duration=40000),
               air_environment=dict(id=999, position=[0.0, 0.0, 0.0], targets=T),
               radars=R, missile_launchers=L, combat_control_point=dict(id=0))
    return dict(name="solve_branches_zero_noise", config=cfg, script=[list(s) for s in script], seed=seed,
                zero_noise=True)


def bulk_scene(seed, n=1000, n_radars=4, ticks=300, dt=100):
    g = np.random.Generator(np.random.PCG64(seed))
    P = np.stack([g.uniform(-30e3, 30e3, n), g.uniform(-30e3, 30e3, n), g.uniform(100, 12e3, n)], 1)
    V = g.normal(0, 150, (n, 3))
    T = [target(1000 + i, P[i], V[i]) for i in range(n)]
    R = [radar(10 + r, (1000.0 * r, 0, 0), 0, 0, 50e3 if r % 2 == 0 else 25e3, 90, 45, 10, 0 if r < 2 else 5,
               "horizontal" if r != 3 else "vertical") for r in range(n_radars)]
    L = [launcher(3, (0, 0, 0), [(30000 + k, 1000, 150, 60 if k != 20 else 8.0) for k in range(40)]),
         launcher(4, (2000, -1000, 0), [(40000 + k, 1200, 120, 45) for k in range(20)])]
    script = []
    picks = g.choice(n, 60, replace=False)
    for k in range(5, 60, 10):
        picks[k] = picks[k - 1]            # second missile (other launcher) on the same target
    for k, tgt in enumerate(picks):
        script.append([int(200 * k), 3 if k % 2 else 4, int(1000 + tgt)])
    cfg = dict(simulation=dict(time_step=dt, duration=ticks * dt),
               air_environment=dict(id=999, position=[0.0, 0.0, 0.0], targets=T),
               radars=R, missile_launchers=L, combat_control_point=dict(id=0))
    return dict(name="bulk_n1000_r4", config=cfg, script=script, seed=seed, zero_noise=False)


def battery_scene(zero_noise, seed, n=160, ticks=150, dt=200):
    """The CLOSED loop with the reference's own command post and launchers (no script): a spread-out raid flies at a site with
    two all-round radars and three launchers; the command post asks for the missile count on tick 0, learns it on tick 1,
    requests launches as it links detections to tracks; launchers solve a tick later, missiles enter the air two ticks
    after that, hit, miss, time out, and some launches are cancelled (a short-lived magazine) and their missiles re-used.
    Pins the latencies and the order of everything the device-side battery loop composes."""
    g = np.random.Generator(np.random.PCG64(seed))
    ang = g.uniform(0, 2 * np.pi, n)
    rad = g.uniform(9e3, 26e3, n)
    P = np.stack([rad * np.cos(ang), rad * np.sin(ang), g.uniform(800, 6000, n)], 1)
    spd = g.uniform(140, 320, n)
    heading = ang + np.pi + g.normal(0, 0.35, n)                   # roughly at the site
    V = np.stack([spd * np.cos(heading), spd * np.sin(heading), g.normal(0, 8, n)], 1)
    T = [target(1000 + i, P[i], V[i]) for i in range(n)]
    R = [radar(10, (0, 0, 10), 0, 0, 30e3, 360, 180, 0, 0), radar(11, (1000, 1000, 10), 0, 0, 18e3, 360, 180, 0, 0)]
    L = [launcher(3, (0, 0, 0), [(30000 + k, 1000, 150, 60) for k in range(10)]),
         launcher(4, (2000, -1000, 0), [(40000 + k, 1200, 120, 45 if k % 3 else 6.0) for k in range(8)]),
         # (no tight fuses here: a missile that flies past its target is linked to the target's track, and the reference's
         # command post then raises in send_objects_to_GUI -- modules/CCP.py:259 reads `.type` of a Missile; timeouts and chases of
         # removed targets are pinned at the L1 boundary by the `missiles` fixture)
         launcher(5, (-1500, 2500, 5), [(50000 + k, 900, 200, 40) for k in range(12)])]
    cfg = dict(simulation=dict(time_step=dt, duration=ticks * dt),
               air_environment=dict(id=999, position=[0.0, 0.0, 0.0], targets=T),
               radars=R, missile_launchers=L,
               combat_control_point=dict(id=0, missile_launcher_ids=[3, 4, 5], radar_ids=[10, 11]))
    return dict(name="battery_zero_noise" if zero_noise else "battery", config=cfg, script=None, seed=seed, zero_noise=zero_noise)


def main():
    os.chdir(REF)        # the reference opens its YAML by relative path
    jobs = []
    for seed in (0, 1, 2):
        jobs.append((yaml_scene(f"stock_simulation_config_seed{seed}", "simulation_config.yaml", seed), 1))
    jobs.append((yaml_scene("stock_config_seed0", "config.yaml", 0), 1))
    jobs.append((yaml_scene("stock_simulation_config_copy_seed0", "simulation_config copy.yaml", 0), 1))
    jobs.append((edge_scene(True, 7), 1))
    jobs.append((edge_scene(False, 7), 1))
    jobs.append((missile_scene(11), 1))
    jobs.append((solve_branch_scene(5), 1))
    jobs.append((bulk_scene(1239, ticks=400), 50))
    jobs.append((battery_scene(True, 21), 25))
    jobs.append((battery_scene(False, 21), 25))
    for scene, every in jobs:
        out = capture(scene, sample_every=every)
        path = OUT / f"{scene['name']}.npz"
        np.savez_compressed(path, **out)
        h = json.loads(out["histogram"])
        print(f"{scene['name']:40s} ticks={len(out['tick_ms']):4d} msgs={sum(h.values()):6d} "
              f"found={len(out['found_ids']):6d} det={out['detonations'].tolist()} "
              f"ok={len(out['launch_ok'])} cancel={len(out['launch_cancel'])} {path.stat().st_size // 1024} KiB")


if __name__ == "__main__":
    main()
