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
