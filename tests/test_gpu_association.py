"""zrk_ccp_link against the reference's sequential loop as the oracle restates it (modules/CCP.py:171-219 applied in the
order of :414-429): same verdict for every detection, also where detections compete for tracks, where a detection's
in-gate candidates outnumber the list the kernel keeps, and with tracks updated "now"."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sequential_link(det_pos, det_speed, trk_ref, trk_upd, now_s, slack_s):
    """The reference's loop, restated in the oracle (oracle/zrk_oracle.c::zo_ccp_link)."""
    from oracle import oracle as O
    return O.ccp_link(det_pos, det_speed, trk_ref, trk_upd, now_s, slack_s)


def _ctx():
    from zrk_modulation_amd._lib import Context
    return Context(0)


@pytest.mark.parametrize("seed,D,T,spread,dup", [(1, 300, 400, 3000.0, 0.0), (2, 500, 500, 300.0, 0.3), (3, 64, 2000, 40.0, 0.5),
                                                 (4, 1200, 900, 800.0, 0.2), (5, 5, 0, 10.0, 0.0), (6, 0, 7, 10.0, 0.0)])
def test_device_association_equals_the_sequential_loop(seed, D, T, spread, dup):
    from zrk_modulation_amd.association import link_all
    g = np.random.Generator(np.random.PCG64(seed))
    trk_ref = g.uniform(-spread, spread, (T, 3))
    # detections near tracks (dense: many tracks in gate, detections competing), some far away, some exact duplicates
    det_pos = g.uniform(-spread, spread, (D, 3))
    if T and D:
        near = g.integers(0, T, D)
        det_pos = np.where(g.uniform(size=(D, 1)) < 0.8, trk_ref[near] + g.normal(0, 30.0, (D, 3)), det_pos)
        k = int(dup * D)
        if k:
            det_pos[g.integers(0, D, k)] = det_pos[g.integers(0, D, k)]          # ties between detections
    det_speed = g.uniform(50.0, 900.0, D)
    now_s, slack_s = 12.34, 100 * 0.01
    trk_upd = now_s - g.choice([0.01, 0.02, 0.5, 3.0], T)
    if T:
        trk_upd[g.uniform(size=T) < 0.1] = now_s                                  # updated this tick: skipped
    want = sequential_link(det_pos, det_speed, trk_ref, trk_upd, now_s, slack_s)
    got = link_all(_ctx(), "cuda:0", det_pos, det_speed, trk_ref, trk_upd, now_s, slack_s)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {D} verdicts differ"
    if D and T:
        assert (want >= 0).sum() > 0 or spread > 1000


def test_more_candidates_in_gate_than_the_kernel_keeps():
    """Forty tracks inside every detection's annulus, thirty detections competing for them in order: the kept prefix of
    eight runs out and the re-scan path decides."""
    from zrk_modulation_amd.association import link_all
    g = np.random.Generator(np.random.PCG64(9))
    T, D = 40, 30
    trk_ref = g.normal(0, 5.0, (T, 3))
    det_pos = g.normal(0, 5.0, (D, 3))
    det_speed = np.full(D, 500.0)
    now_s, slack_s = 5.0, 1.0
    trk_upd = np.full(T, now_s - 0.01)
    want = sequential_link(det_pos, det_speed, trk_ref, trk_upd, now_s, slack_s)
    got = link_all(_ctx(), "cuda:0", det_pos, det_speed, trk_ref, trk_upd, now_s, slack_s)
    assert np.array_equal(got, want)
    assert len(set(want[want >= 0])) == (want >= 0).sum() == D


def test_command_post_with_device_association_sends_what_the_host_loop_sends(monkeypatch):
    """The module stack on a scene with a few hundred targets, twice: the command post's link_object calls on the device
    (all detections of a tick at once) and as the reference's host loop.  Same messages, tick by tick."""
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.main import create_objects_from_config

    def run(host):
        if host:
            monkeypatch.setenv("ZRK_CCP_HOST", "1")
        else:
            monkeypatch.delenv("ZRK_CCP_HOST", raising=False)
        cfg = S.synthetic_config(300, 2, seed=21, launchers=2, missiles_per_launcher=12, time_step=100, duration=1500)
        for t in cfg["air_environment"]["targets"]:
            t["position"] = [p * 0.3 for p in t["position"][:2]] + [t["position"][2]]
        np.random.seed(5)
        mgr, _ = create_objects_from_config(cfg, device="cuda:0")
        mgr.run_simulation(1500)
        post = [m for m in mgr.modules if type(m).__name__ == "CombatControlPoint"][0]
        assert (post.device_ticks == 0) if host else (post.device_ticks >= 10), (post.device_ticks, post.host_ticks)
        out = []
        for step in sorted(mgr.messages):
            for m in mgr.messages[step]:
                rec = [step, type(m).__name__, getattr(m, "sender_id", None), getattr(m, "receiver_id", None)]
                for attr in ("obj_id", "missile_id", "target_id", "is_visible_by_radar"):
                    if hasattr(m, attr):
                        rec.append((attr, getattr(m, attr)))
                if hasattr(m, "target") and m.target is not None:
                    rec.append(("target", m.target.id))
                if hasattr(m, "coordinates"):
                    rec.append(("xyz", tuple(np.asarray(m.coordinates, np.float64).view(np.uint64).tolist())))
                out.append(tuple(map(str, rec)))
        return out

    a, b = run(host=False), run(host=True)
    assert len(a) == len(b) and len(a) > 1000
    assert a == b
    assert any("CPPLaunchMissileRequestMessage" in r[1] for r in a)
