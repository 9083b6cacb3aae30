"""The oracle's CLOSED loop (oracle/battery.py: AirEnv + radars + launchers + command post composed with the reference's
message latencies) against full runs of the reference itself with its OWN command post and launchers
(tests/golden/gen_golden.py::battery_scene -> battery.npz, battery_zero_noise.npz).  Nothing of the run is fed in but the
scene and the noise stream's seed: every tick's live ids, detection lists, scan state, position bits, the command post's
launch requests (order, launcher, target), every launch solve (missile, target, V bits or the reason it was cancelled),
when each missile was announced to AirEnv, and every detonation must come out as the reference produced them."""
import numpy as np
import pytest

from tests.helpers import REASON_CODE, Fixture


def replay_battery(make_tick, fx):
    """make_tick() -> per-tick log dictionaries in oracle.battery.OracleBattery.tick()'s shape, compared with the fixture."""
    seen = dict(found=0, requests=0, ok=0, cancel=0, new=0, detonations=0)
    for T in range(fx.n_ticks):
        t = int(fx.tick_ms[T])
        log = make_tick()
        assert log["t"] == t
        assert np.array_equal(log["active"], fx.active_ids(T)), f"live ids differ at t={t}"
        assert log["detonations"] == fx.rows_at(fx.detonations, t).tolist(), f"detonations differ at t={t}"
        ids = log["ids"]
        for r in range(fx.R):
            assert np.array_equal(ids[log["found"][r]], fx.found(T, r)), f"radar {r} detections differ at t={t}"
            seen["found"] += len(log["found"][r])
        assert np.array_equal(log["radar_state"], fx.radar_state[T]), f"scan state differs at t={t}"
        assert log["launch_req"] == fx.rows_at(fx.launch_req, t).tolist(), f"launch requests differ at t={t}"
        assert log["launch_cmd"] == fx.rows_at(fx.launch_cmd, t).tolist(), f"launches differ at t={t}"
        want_ok = fx.rows_at(fx.launch_ok, t)
        assert log["launch_ok"] == want_ok.tolist(), f"successful launches differ at t={t}"
        traj = fx.launch_traj[fx.launch_ok[:, 0] == t]
        assert np.array_equal(np.array(log["launch_traj"], np.float64).reshape(-1, 7).view(np.uint64), traj.view(np.uint64)), f"launch V bits differ at t={t}"
        bad = [k for k, r in enumerate(fx.launch_cancel) if r[0] == t]
        assert [[a, b] for a, b, _ in log["launch_cancel"]] == [fx.launch_cancel[k].tolist() for k in bad], f"cancelled launches differ at t={t}"
        assert [c for _, _, c in log["launch_cancel"]] == [REASON_CODE[fx.reasons[k]] for k in bad]
        assert log["new_missile"] == fx.rows_at(fx.new_missile, t).tolist(), f"NEW_MISSILE differs at t={t}"
        assert log["pos_digest"] == int(fx.pos_digest[T]), f"position bits differ at t={t}"
        seen["requests"] += len(log["launch_req"]); seen["ok"] += len(log["launch_ok"]); seen["cancel"] += len(log["launch_cancel"])
        seen["new"] += len(log["new_missile"]); seen["detonations"] += len(log["detonations"])
    return seen


def oracle_ticker(fx):
    from oracle.battery import OracleBattery
    bat = OracleBattery(fx.cfg, fx.noise_fn())

    def tick():
        log = bat.tick()
        sim = bat.sim
        act = sim.active_slots()
        P = np.ascontiguousarray(sim.pos_of(act))
        log["pos_digest"] = int(np.bitwise_xor.reduce(P.view(np.uint64).ravel())) if len(act) else 0
        log["ids"] = sim.ids
        return log
    return tick


@pytest.mark.parametrize("name", ["battery_zero_noise", "battery"])
def test_oracle_closed_loop_equals_the_reference_run(name):
    fx = Fixture(name)
    seen = replay_battery(oracle_ticker(fx), fx)
    # what the fixture holds (known answers of the capture): 30 requests on tick 1 (the counts arrive then), 27 launches, 3
    # cancelled (a 6-second magazine), 27 hits
    assert seen == dict(found=len(fx.found_ids), requests=30, ok=27, cancel=3, new=27, detonations=27)
    assert (fx.launch_req[:, 0] == fx.dt).all() and (fx.launch_cmd[:, 0] == 2 * fx.dt).all() and (fx.new_missile[:, 0] == 3 * fx.dt).all()
    first_step = {int(m): int(t) for t, m in fx.new_missile}
    # (a missile is live from the tick after its NEW_MISSILE)
    for T in range(fx.n_ticks):
        for mid in set(first_step) & set(int(i) for i in fx.active_ids(T)):
            assert fx.tick_ms[T] >= first_step[mid] + fx.dt
