"""Shared test plumbing: golden-fixture loading and the L1 replay that drives either the
CPU oracle (oracle.OracleSim) or the device engine through the same duck-typed surface."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"

ALL_FIXTURES = [
    "stock_simulation_config_seed0", "stock_simulation_config_seed1", "stock_simulation_config_seed2",
    "stock_config_seed0", "stock_simulation_config_copy_seed0", "edges_zero_noise", "edges", "missiles",
    "solve_branches_zero_noise", "bulk_n1000_r4",
]

REASON_CODE = {
    "No interception possible: target and interceptor are stationary relative or parallel.": 1,
    "Interception impossible in the future: computed time t <= 0.": 2,
    "No real interception time: target is too fast or out of range.": 3,
    "Interception times are not positive; interception not possible in future.": 4,
    "Target is too far for this rocket (detonation_period over limited)": 5,
}


class Fixture:
    def __init__(self, name):
        z = np.load(GOLDEN / f"{name}.npz")
        self.name = name
        self.scene = json.loads(str(z["scene"]))
        self.cfg = self.scene["config"]
        self.histogram = json.loads(str(z["histogram"]))
        self.reasons = json.loads(str(z["reasons"]))
        for k in z.files:
            if k not in ("scene", "histogram", "reasons", "draw_types"):
                setattr(self, k, z[k])
        self.draw_types = json.loads(str(z["draw_types"]))
        self.dt = self.cfg["simulation"]["time_step"]
        self.n_ticks = len(self.tick_ms)
        self.R = self.radar_state.shape[1]
        self.samp_index = {int(t): k for k, t in enumerate(self.samp_tick)}

    def draw(self, T):
        """DRAW_OBJECTS messages of tick T, message order: (ids, type names, positions, visible)."""
        lo, hi = self.draw_off[T], self.draw_off[T + 1]
        return (self.draw_ids[lo:hi], [self.draw_types[c] for c in self.draw_type[lo:hi]], self.draw_pos[lo:hi],
                self.draw_vis[lo:hi].astype(bool))

    def active_ids(self, T):
        return self.act_ids[self.act_off[T]:self.act_off[T + 1]]

    def found(self, T, r):
        k = T * self.R + r
        return self.found_ids[self.found_off[k]:self.found_off[k + 1]]

    def rows_at(self, arr, t_ms):
        return arr[arr[:, 0] == t_ms]

    def noise_fn(self):
        """The draws SectorRadar.smooth_objects makes, from the same global legacy stream
        (reference modules/Radar.py:138-142), seeded as the capture was."""
        np.random.seed(self.scene["seed"])
        if self.scene.get("zero_noise"):
            return lambda k: np.zeros((k, 3))
        return lambda k: np.random.normal(0, 5, (k, 3))

    def missiles(self):
        out = []
        for lc in self.cfg.get("missile_launchers", []) or []:
            for mc in lc.get("missiles", []) or []:
                out.append(dict(id=mc["id"], pos=lc["position"], velocity_module=mc.get("velocity", 1000),
                                detonate_radius=mc.get("explosion_radius", 50),
                                detonate_period=mc.get("life_time", 60)))
        return out


def populate(sim, fx):
    for tc in fx.cfg["air_environment"].get("targets", []) or []:
        sim.add_target(tc["id"], tc["position"], tc["velocity"], 0.0)
    for rc in fx.cfg.get("radars", []) or []:
        sim.add_radar(rc["id"], rc["position"], rc["azimuth_start"], rc["elevation_start"], rc["max_distance"],
                      rc["azimuth_range"], rc["elevation_range"], rc["azimuth_speed"], rc["elevation_speed"],
                      rc["scan_mode"])
    for m in fx.missiles():
        sim.add_missile(m["id"], m["pos"], m["velocity_module"], m["detonate_radius"], m["detonate_period"])


def replay_l1(sim, fx, check_pos="bits", max_ticks=None):
    """Drive `sim` with the L2 events recorded in the fixture and compare every observable the
    reference produced at the L1 boundary.  Returns a small stats dict."""
    noise = fx.noise_fn()
    stats = dict(found=0, detonations=0, launches=0)
    n_ticks = fx.n_ticks if max_ticks is None else min(fx.n_ticks, max_ticks)
    for T in range(n_ticks):
        t = int(fx.tick_ms[T])
        events = sim.airenv_step()
        ids = sim.ids
        act = sim.active_slots()
        assert np.array_equal(ids[act], fx.active_ids(T)), f"{fx.name}: live ids differ at t={t}"
        want = fx.rows_at(fx.detonations, t)
        got = [[t, int(ids[m]), -1 if tg < 0 else int(ids[tg]), int(s)] for m, tg, s in events]
        assert got == want.tolist(), f"{fx.name}: detonations differ at t={t}: {got} vs {want.tolist()}"
        stats["detonations"] += len(got)
        for r in range(fx.R):
            found = sim.radar_step(r, noise)
            assert np.array_equal(ids[found], fx.found(T, r)), f"{fx.name}: radar {r} detections differ at t={t}"
            stats["found"] += len(found)
        state = np.array([[rd["caz"], rd["cel"]] for rd in sim.radars])
        assert np.array_equal(state, fx.radar_state[T]), f"{fx.name}: scan state differs at t={t}"
        ok_rows = {int(r[1]): k for k, r in enumerate(fx.launch_ok) if r[0] == t}
        bad_rows = {int(r[1]): k for k, r in enumerate(fx.launch_cancel) if r[0] == t}
        for _, _launcher, mid, tid in fx.rows_at(fx.launch_cmd, t):
            slot = sim.slot_of_id[int(tid)][0]
            rc, V, _tt = sim.launch(int(mid), slot)
            stats["launches"] += 1
            if rc == 0:
                assert int(mid) in ok_rows, f"{fx.name}: launch of {mid} at t={t} should have been cancelled"
                traj = fx.launch_traj[ok_rows[int(mid)]]
                assert np.array_equal(V, traj[0:3]), f"{fx.name}: launch V differs at t={t}: {V} vs {traj[0:3]}"
                assert traj[6] == t / 1000
            else:
                assert int(mid) in bad_rows, f"{fx.name}: launch of {mid} at t={t} should have succeeded"
                assert REASON_CODE[fx.reasons[bad_rows[int(mid)]]] == rc
        for _, mid in fx.rows_at(fx.new_missile, t):
            sim.announce_missile(int(mid))
        P = sim.pos_of(act)
        if check_pos == "bits":
            dig = int(np.bitwise_xor.reduce(np.ascontiguousarray(P).view(np.uint64).ravel())) if len(act) else 0
            assert dig == int(fx.pos_digest[T]), f"{fx.name}: position bits differ at t={t}"
        if T in fx.samp_index:
            k = fx.samp_index[T]
            lo, hi = fx.samp_off[k], fx.samp_off[k + 1]
            if check_pos == "bits":
                assert np.array_equal(P, fx.pos[lo:hi])
            else:
                np.testing.assert_allclose(P, fx.pos[lo:hi], rtol=1e-6, atol=0)
            pv = fx.prev_valid[lo:hi].astype(bool)
            assert np.array_equal(sim.prev_valid_of(act).astype(bool), pv), f"{fx.name}: prev_pos None-ness at t={t}"
            PP = sim.prev_of(act)
            assert np.array_equal(PP[pv], fx.prev[lo:hi][pv]), f"{fx.name}: prev_pos differs at t={t}"
        sim.end_tick()
    return stats
