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
