"""BASELINE configs[4] in one table: a batch of independent scenarios swept and compacted by one launch each per tick
(zrk_run_ticks_ensemble), every scenario bit-exact against its OWN oracle replay -- masks, position bits, per-radar
lists, detonation events, scan state, every tick -- and independent of its neighbours."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scenarios(S, n, R, seed0):
    from zrk_modulation_amd import scenario as SC
    out = []
    for k in range(S):
        ids, sp, vel, t0 = SC.synthetic_targets(n - (k % 3) * 17, seed0 + k)          # ragged: not every scenario is full
        sp[:, :2] *= 0.4
        radars = SC.synthetic_radars(R)
        for j, rd in enumerate(radars):
            rd["azimuth_start"] = float((37 * k + 90 * j) % 270)
            rd["max_distance"] = 50e3 if (k + j) % 2 == 0 else 25e3
            rd["elevation_speed"] = 0.0 if (k + j) % 3 else 5.0
            rd["position"] = [700.0 * j - 300.0 * (k % 5), 200.0 * (k % 7), 0.0]
            if (k + j) % 5 == 4:
                rd["scan_mode"] = "vertical"
        out.append(dict(ids=ids, start_pos=sp, velocity=vel, start_time=t0, radars=radars))
    return out


def _run_and_check(S, n, R, m, ticks, dt_ms, noise):
    from tests.test_gpu_engine import OracleMirror, _device_noise_table
    from zrk_modulation_amd.ensemble import EnsembleEngine
    from zrk_modulation_amd import scenario as SC
    scs = _scenarios(S, n, R, 500)
    eng = EnsembleEngine(device="cuda:0", dt_ms=dt_ms, noise=noise)
    eng.load(scs, missile_capacity=m, seeds=[900 + 3 * k for k in range(S)])
    tg = [SC.missile_targets(len(sc["ids"]), m) for sc in scs]
    launched = eng.launch_missiles(tg, launcher_pos=(0.0, 0.0, 0.0), speed=3000.0, radius=1000.0, period=25.0)
    assert launched > S * m // 3
    views = [eng.scenario_view(s) for s in range(S)]
    mirrors = [OracleMirror(v, scs[s]["radars"]) for s, v in enumerate(views)]
    st = eng.store
    P = eng.P
    total_events = 0
    for k in range(ticks):
        want_events = []
        for s, (v, mir) in enumerate(zip(views, mirrors)):
            table = _device_noise_table(v, k, R, mir.n) if noise == "philox" else None
            want_events.append(mir.tick(k * dt_ms, dt_ms, 2 if noise == "philox" else 0, table))
        eng.run(1)
        vis_all = st.vis()[:eng.S * P].cpu().numpy().view(np.uint32)
        pos_all = st.host_pos("cur")
        ne = int(st.dm_evn.item())
        evm, evt = st.dm_evm[:ne].cpu().numpy(), st.dm_evt[:ne].cpu().numpy()
        for s, (v, mir) in enumerate(zip(views, mirrors)):
            lo = s * P
            assert np.array_equal(vis_all[lo:lo + mir.n], mir.vis), f"tick {k} scenario {s}: masks differ"
            assert not vis_all[lo + mir.n:lo + P].any(), f"tick {k} scenario {s}: padding rows detected"
            pl = v.list_view(pos_all[lo:lo + P])
            assert np.array_equal(np.ascontiguousarray(pl.T).reshape(-1).view(np.uint64), mir.pos.view(np.uint64)), \
                f"tick {k} scenario {s}: position bits differ"
            sel = (evm >= lo) & (evm < lo + P)
            got = [(int(mir.lidx[a - lo]), int(mir.lidx[b - lo]) if b >= 0 else -1) for a, b in zip(evm[sel], evt[sel])]
            assert got == want_events[s], f"tick {k} scenario {s}: events differ: {got} vs {want_events[s]}"
            total_events += len(got)
            if k % 5 == 4 or k == ticks - 1:
                lists = eng.detections(s)
                for r, want in enumerate(mir.lists()):
                    assert np.array_equal(lists[r], want), f"tick {k} scenario {s} radar {r}: list differs"
                assert eng.radar_state(s) == [(r["caz"], r["cel"]) for r in mir.rs]
    return total_events


@pytest.mark.parametrize("noise", ["philox", "off"])
def test_every_scenario_of_a_batch_matches_its_own_oracle_replay(noise):
    events = _run_and_check(S=130, n=900, R=3, m=6, ticks=12, dt_ms=500, noise=noise)
    assert events > 20


def test_small_batch_many_ticks():
    """Fewer scenarios, longer: records of the box cache age and are retaken, missiles die, scans wrap."""
    events = _run_and_check(S=5, n=2500, R=4, m=20, ticks=45, dt_ms=250, noise="philox")
    assert events > 10
