"""The CPU oracle (oracle/zrk_oracle.c driven by oracle.OracleSim) against the vectors captured
from the reference itself (tests/golden/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import ALL_FIXTURES, Fixture, populate, replay_l1


def make_sim(fx):
    n_t = len(fx.cfg["air_environment"].get("targets", []) or [])
    n_m = len(fx.missiles())
    sim = O.OracleSim(fx.dt, n_t + n_m + 4, n_m + 1)
    sim.prev_valid_of = lambda slots: sim.prev_valid[np.asarray(slots)]
    populate(sim, fx)
    return sim


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_oracle_replays_reference(name):
    fx = Fixture(name)
    stats = replay_l1(make_sim(fx), fx, check_pos="bits")
    assert stats["found"] == len(fx.found_ids)
    assert stats["detonations"] == len(fx.detonations)
    assert stats["launches"] == len(fx.launch_cmd)


def test_known_answers_of_stock_runs():
    """SURVEY.md 8c / BASELINE.md section 2: seed-independent anchors of the shipped YAMLs."""
    fx = Fixture("stock_simulation_config_seed0")
    assert fx.n_ticks == 200 and sum(fx.histogram.values()) == 685 and len(fx.found_ids) == 20
    assert fx.detonations.tolist() == [[2200, 3005, 40, 0], [2600, 4005, 39, 0]]
    fx = Fixture("stock_config_seed0")
    assert fx.n_ticks == 400 and sum(fx.histogram.values()) == 2888 and len(fx.found_ids) == 872
    assert fx.detonations.tolist() == [[14800, 10, 101, 0], [29200, 11, 100, 0]]
    fx = Fixture("stock_simulation_config_copy_seed0")
    assert fx.n_ticks == 200 and sum(fx.histogram.values()) == 800 and len(fx.found_ids) == 100
    assert len(fx.detonations) == 0


def test_oracle_association_equals_the_reference_loop_in_numpy():
    """oracle/zrk_oracle.c::zo_ccp_link is the reference's link_object loop (modules/CCP.py:171-219, :414-429): pinned
    here against that loop written out with numpy exactly as the reference writes it (np.linalg.norm, max(0, .),
    strict <), on this host's numpy -- the one the golden fixtures come from."""
    from oracle import oracle as O
    g = np.random.Generator(np.random.PCG64(1))
    T, D = 70, 60
    trk = g.uniform(-300, 300, (T, 3))
    det = np.where(g.uniform(size=(D, 1)) < 0.8, trk[g.integers(0, T, D)] + g.normal(0, 30, (D, 3)), g.uniform(-300, 300, (D, 3)))
    sp = g.uniform(50, 900, D)
    now, slack = 12.34, 1.0
    upd = now - g.choice([0.01, 0.02, 0.5, 3.0], T)
    upd[g.uniform(size=T) < 0.1] = now
    got = O.ccp_link(det, sp, trk, upd, now, slack)
    u = upd.copy()
    want = np.full(D, -1, np.int32)
    for d in range(D):
        best, m = float("inf"), -1
        for t in range(T):
            if u[t] == now:
                continue
            age = now - u[t]
            lo, hi = max(0, sp[d] * (age - slack)), max(0, sp[d] * (age + slack))
            dist = np.linalg.norm(trk[t] - det[d])
            if dist < best and lo <= dist <= hi:
                best, m = dist, t
        want[d] = m
        if m >= 0:
            u[m] = now
    assert np.array_equal(got, want) and (want >= 0).sum() > 20


def test_multithreaded_airenv_step_equals_the_literal_loop():
    """zo_airenv_step_mt (targets in parallel, missiles in list order -- the all-core CPU baseline of bench.py) against the
    literal loop zo_airenv_step, which the fixtures pin: positions, prev positions, prev_valid, missile table, ordered
    events, tick by tick -- with missiles that chase missiles in front of and behind them in the list, targets behind their
    missiles, removals and time-outs on the way."""
    L = O.lib()
    g = np.random.Generator(np.random.PCG64(77))
    n_t, n_m, ticks, dt = 3000, 400, 60, 100
    n = n_t + n_m
    kind = np.zeros(n, np.uint8)
    slots_m = np.sort(g.choice(n, n_m, replace=False))           # missiles scattered through the list
    kind[slots_m] = 1
    mrow = np.full(n, -1, np.int32); mrow[slots_m] = np.arange(n_m, dtype=np.int32)
    sp = g.uniform(-3e3, 3e3, (3, n)); vel = g.normal(0, 200, (3, n)); t0 = np.where(g.uniform(size=n) < 0.1, 0.3, 0.0)
    m_tgt = g.integers(0, n, n_m).astype(np.int32)               # anything: targets, missiles, in front or behind
    m_radius = g.uniform(200, 900, n_m); period0 = g.uniform(0.5, 8.0, n_m)

    def state():
        return dict(pos=np.ascontiguousarray(sp).reshape(-1).copy(), prev=np.ascontiguousarray(sp).reshape(-1).copy(),
                    pv=np.zeros(n, np.uint8), alive=np.ones(n, np.uint8), period=period0.copy(), status=np.ones(n_m, np.uint8),
                    ev=(np.zeros(n_m, np.int32), np.zeros(n_m, np.int32), np.zeros(n_m, np.uint8)))
    a, b = state(), state()
    spf, velf = np.ascontiguousarray(sp).reshape(-1), np.ascontiguousarray(vel).reshape(-1)
    total = 0
    for k in range(ticks):
        out = []
        for S, fn, extra in ((a, L.zo_airenv_step, ()), (b, L.zo_airenv_step_mt, (4,))):
            nev = fn(n, n, k * dt, dt, O.dptr(spf), O.dptr(velf), O.dptr(t0), O.u8ptr(S["alive"]), O.u8ptr(kind), O.i32ptr(mrow),
                     O.dptr(S["pos"]), O.dptr(S["prev"]), O.u8ptr(S["pv"]), O.i32ptr(m_tgt), O.dptr(m_radius), O.dptr(S["period"]),
                     O.u8ptr(S["status"]), O.i32ptr(S["ev"][0]), O.i32ptr(S["ev"][1]), O.u8ptr(S["ev"][2]), *extra)
            out.append([(int(S["ev"][0][j]), int(S["ev"][1][j]), int(S["ev"][2][j])) for j in range(nev)])
        assert out[0] == out[1], f"tick {k}: events differ"
        for key in ("pos", "prev", "pv", "period", "status"):
            assert np.array_equal(a[key].view(np.uint8), b[key].view(np.uint8)), f"tick {k}: {key} differs"
        for S in (a, b):                                          # AirEnv.py:33-40: removals take effect on the next tick
            for ms, ts, _ in out[0]:
                S["alive"][ms] = 0
                if ts >= 0:
                    S["alive"][ts] = 0
        total += len(out[0])
    assert total > 100


def test_oracle_command_post_step_equals_the_reference():
    """oracle/zrk_oracle.c::zo_ccp_step is CombatControlPoint.step's detection loop (modules/CCP.py:406-429: link_object,
    new_target / old_target / old_rocket, try_to_launch_missile): pinned against the decisions of the reference's own
    class, recorded over 36 ticks of a tight swarm with two radars, three launchers and missiles entering the air
    (tests/golden/gen_ccp_golden.py -> ccp_step.npz): every processed detection's verdict, matched key and launcher."""
    import json
    from pathlib import Path
    fx = np.load(Path(__file__).parent / "golden" / "ccp_step.npz")
    meta = json.loads(str(fx["meta"]))
    N = len(fx["obj_id"])
    st = O.CcpState(N, meta["launcher_pos"], np.zeros(len(meta["capacity"]), np.int32))
    dt = meta["dt_ms"] / 1000
    slack = meta["slack_steps"] * dt
    ticks = len(fx["seq_off"]) - 1
    seen = np.zeros(3, np.int64)
    launches = 0
    for k in range(ticks):
        now = k * meta["dt_ms"] / 1000
        if k == 1:                                   # MissileCountResponse arrives in tick 1
            st.l_cap[:] = meta["capacity"]
        for mi, _tgt in fx["new_missile"][fx["new_missile_off"][k]:fx["new_missile_off"][k + 1]]:
            st.add_missile(int(mi), now)
        seq = fx["seq_obj"][fx["seq_off"][k]:fx["seq_off"][k + 1]]
        rows, verdict, match, launcher = st.step(seq, fx["pos"][k], fx["prev"][k], fx["prev_none"][k], fx["speed"], now, slack)
        a, b = fx["out_off"][k], fx["out_off"][k + 1]
        assert np.array_equal(rows, fx["out_obj"][a:b]), f"tick {k}: processed objects differ"
        assert np.array_equal(verdict, fx["out_verdict"][a:b]), f"tick {k}: verdicts differ"
        # the reference names the matched dict KEY, the oracle the track: compare through the key arrays
        key = np.where(verdict == 1, st.tt_key[np.maximum(match, 0)], np.where(verdict == 2, st.tm_key[np.maximum(match, 0)], -1))
        assert np.array_equal(key, fx["out_match"][a:b]), f"tick {k}: matched keys differ"
        lid = np.array([meta["launcher_ids"][l] if l >= 0 else -1 for l in launcher])
        assert np.array_equal(lid, fx["out_launcher"][a:b]), f"tick {k}: launchers differ"
        for v in range(3):
            seen[v] += int((verdict == v).sum())
        launches += int((launcher >= 0).sum())
    assert seen.min() > 100 and launches == sum(meta["capacity"])
