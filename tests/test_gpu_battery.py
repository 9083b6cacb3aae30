"""The battery's CLOSED loop on the device (zrk_modulation_amd.battery.DeviceBattery: sweep -> lists -> zrk_ccp_step -> launchers
-> rows in the air, nothing read back inside the loop) against (i) a full run of the REFERENCE with its own command post and
launchers (tests/golden/battery_zero_noise.npz: the noise-free capture, since the headless loop draws its noise from its own
counter-based stream) and (ii) the oracle's closed loop (oracle/battery.py, itself pinned on the reference's runs with and
without noise: tests/test_oracle_battery.py) at configs[1] scale with Philox noise."""
import numpy as np
import pytest
import torch

from tests.helpers import REASON_CODE, Fixture

pytestmark = pytest.mark.gpu


def _engine_from_cfg(cfg, noise, seed=0, sort=True):
    from zrk_modulation_amd.battery import DeviceBattery
    from zrk_modulation_amd.engine import HotPathEngine
    T = cfg["air_environment"]["targets"]
    ids = np.array([t["id"] for t in T], np.int64)
    sp = np.array([t["position"] for t in T], np.float64)
    vel = np.array([t["velocity"] for t in T], np.float64)
    L = cfg["missile_launchers"]
    nm = sum(min(len(l["missiles"]), l.get("max_missiles", 5)) for l in L)
    eng = HotPathEngine(device="cuda:0", dt_ms=cfg["simulation"]["time_step"], seed=seed, noise=noise)
    eng.load(ids, sp, vel, 0.0, cfg["radars"], missile_capacity=nm, sort=sort).enable_lists()
    bat = DeviceBattery(eng, L, ccp_launcher_ids=cfg["combat_control_point"]["missile_launcher_ids"])
    return eng, bat


@pytest.mark.parametrize("sort", [False, True])
def test_closed_loop_equals_the_reference_run(sort):
    """150 ticks of the reference's own closed loop (real CombatControlPoint, real launchers, noise patched to zero): the
    device's launches -- tick, launcher, missile id (the launchers' LIFO magazines, cancelled missiles re-used), target, V bits,
    cancel reasons --, the ticks the missiles enter the air and every detonation, from ONE enqueued sequence with nothing read
    back until the end; at sampled ticks also the per-radar lists."""
    fx = Fixture("battery_zero_noise")
    eng, bat = _engine_from_cfg(fx.cfg, "off", sort=sort)
    ids_of_list = np.concatenate([np.array([t["id"] for t in fx.cfg["air_environment"]["targets"]], np.int64), np.array([m["id"] for m in bat.missiles])])
    done = 0
    for upto in (1, 2, 3, 4, 26, 76, fx.n_ticks):
        bat.run(upto - done)
        done = upto
        T = upto - 1
        lists = eng.detections()
        for r in range(fx.R):
            want = fx.found(T, r)
            got = lists[r]
            # (list indices; the magazine's rows enter the list in the order the missiles took the air)
            air = bat.air_missile.cpu().numpy()
            ids = np.array([ids_of_list[i] if i < bat.n_targets else bat.missiles[air[i - bat.n_targets]]["id"] for i in got], np.int64)
            assert np.array_equal(ids, want), f"tick {T}: radar {r} sees other objects"
    res = bat.results()
    got_cmd = [[t, l, m, tg] for t, l, m, tg, rc, V in res["solves"]]
    assert got_cmd == fx.launch_cmd.tolist()
    ok = [(t, m, tg, V) for t, l, m, tg, rc, V in res["solves"] if rc == 0]
    assert [[t, m, tg] for t, m, tg, V in ok] == fx.launch_ok.tolist()
    assert np.array_equal(np.array([V for *_, V in ok]).view(np.uint64), fx.launch_traj[:, 0:3].view(np.uint64)), "launch velocity bits"
    bad = [(t, m, rc) for t, l, m, tg, rc, V in res["solves"] if rc != 0]
    assert [[t, m] for t, m, rc in bad] == fx.launch_cancel.tolist() and [rc for *_, rc in bad] == [REASON_CODE[r] for r in fx.reasons]
    assert [list(x) for x in res["new_missile"]] == fx.new_missile.tolist()
    assert [list(x) for x in res["detonations"]] == fx.detonations.tolist()
    assert len(res["detonations"]) == 27 and len(bad) == 3
    # what is left in the air, and the launchers' lists (three missiles came back)
    assert eng.alive_count() == int(len(fx.active_ids(fx.n_ticks - 1))) - len(fx.rows_at(fx.detonations, int(fx.tick_ms[-1]))) * 2
    assert bat.top.cpu().tolist() == [0, 3, 0][:bat.L] or sum(bat.top.cpu().tolist()) == 3
    bat.close()


def test_closed_loop_at_configs1_scale_against_the_oracle():
    """configs[1]'s table (1e5 AirObjects, 4 rotating sector radars) with a battery of four launchers and 2400 missiles, Philox
    noise, 90 ticks: the device's closed loop (nothing read back until the end) against the oracle's (oracle/battery.py, pinned
    on the reference's own closed-loop runs), which is fed the noise triples the device will draw.  Launches in more than five
    different ticks, > 1000 of them, cancelled ones, > 100 kills: launcher, missile id, target id, return code and V bits of
    every solve, the tick every missile enters the air, every detonation in order.  (The radars reach 15 km instead of 50: the
    oracle's command post is the reference's sequential loop, detections x tracks per tick.)"""
    from oracle.battery import OracleBattery
    from tests.test_gpu_engine import _device_noise_table
    from zrk_modulation_amd import scenario as S
    n, R, ticks, dt = 100_000, 4, 90, 200
    ids, sp, vel, t0 = S.synthetic_targets(n, 1235)
    radars = S.synthetic_radars(R)
    for rd in radars:
        rd["max_distance"] = 15e3
        rd["id"] = 10 + radars.index(rd)
    launchers = [dict(id=3 + l, position=[float(x), float(y), 0.0], max_missiles=600,
                      missiles=[dict(id=10_000_000 * (3 + l) + k, velocity=1000.0 + 50.0 * l, explosion_radius=150.0, life_time=60.0 if k % 7 else 5.0)
                                for k in range(600)])
                 for l, (x, y) in enumerate([(0, 0), (3000, 1500), (-2000, 4000), (5000, -500)])]
    cfg = dict(simulation=dict(time_step=dt, duration=ticks * dt),
               air_environment=dict(id=999, position=[0.0, 0.0, 0.0],
                                    targets=[dict(id=int(ids[i]), type="AIR_PLANE", position=sp[i].tolist(), velocity=vel[i].tolist()) for i in range(n)]),
               radars=radars, missile_launchers=launchers, combat_control_point=dict(id=0, missile_launcher_ids=[3, 4, 5, 6], radar_ids=[10, 11, 12, 13]))
    eng, bat = _engine_from_cfg(cfg, "philox", seed=777)
    n_list = eng.n_list
    # the oracle, fed the device's draws: entity = list index, ordinal = how many radars have seen it this tick
    state = dict(table=None, seen=None)

    def noise(found, r):
        ordn = state["seen"][found]
        state["seen"][found] += 1
        return state["table"][ordn, found]
    noise.by_slot = True
    ora = OracleBattery(cfg, noise)
    want = dict(cmd=[], ok=[], V=[], bad=[], new=[], det=[])
    for T in range(ticks):
        state["table"] = _device_noise_table(eng, T, R, n_list).reshape(R, n_list, 3)
        state["seen"] = np.zeros(n_list, np.int64)
        log = ora.tick()
        want["cmd"] += log["launch_cmd"]; want["ok"] += log["launch_ok"]; want["V"] += [tr[0:3] for tr in log["launch_traj"]]
        want["bad"] += log["launch_cancel"]; want["new"] += log["new_missile"]; want["det"] += log["detonations"]
    bat.run(ticks)
    res = bat.results()
    eng.store.compact_status()
    assert [[t, l, m, tg] for t, l, m, tg, rc, V in res["solves"]] == want["cmd"]
    ok = [(t, m, tg, V) for t, l, m, tg, rc, V in res["solves"] if rc == 0]
    assert [[t, m, tg] for t, m, tg, V in ok] == want["ok"]
    assert np.array_equal(np.array([V for *_, V in ok]).view(np.uint64), np.array(want["V"], np.float64).view(np.uint64)), "launch velocity bits"
    assert [[t, m, rc] for t, l, m, tg, rc, V in res["solves"] if rc != 0] == want["bad"]
    assert [list(x) for x in res["new_missile"]] == want["new"]
    assert [list(x) for x in res["detonations"]] == want["det"]
    solve_ticks = {t for t, *_ in res["solves"]}
    assert len(solve_ticks) > 5 and len(ok) > 1000 and len(want["bad"]) > 10 and len(want["det"]) > 100, \
        (len(solve_ticks), len(ok), len(want["bad"]), len(want["det"]))
    # the table at the end: who is in the air, bit for bit where they are
    sim = ora.sim
    alive_dev = eng.list_view(eng.store.d_alive[:eng.store.n_uploaded].cpu().numpy())
    # (the device lowers the flags of the last tick's detonations behind that tick, AirEnv at the start of the next one)
    gone = {s for pair in sim._pending_kill for s in pair if s >= 0}
    live = np.array([s for s in sim.active_slots() if s not in gone], np.int64)
    assert np.array_equal(np.nonzero(alive_dev)[0], live)
    P = eng.list_view(eng.store.host_pos("cur"))[live]
    assert np.array_equal(np.ascontiguousarray(P).view(np.uint64), np.ascontiguousarray(sim.pos_of(live)).view(np.uint64))
    bat.close()
