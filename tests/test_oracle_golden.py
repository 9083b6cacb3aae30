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
