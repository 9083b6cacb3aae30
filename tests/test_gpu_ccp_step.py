"""zrk_ccp_step -- the command post's detection loop of one tick on the device -- against the
oracle's literal loop (oracle/zrk_oracle.c::zo_ccp_step, itself pinned on the reference's own decisions:
tests/test_oracle_golden.py, tests/golden/ccp_step.npz), and on that fixture directly."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ctx():
    from zrk_modulation_amd._lib import Context
    return Context(0)


class Table:
    """A bare entity table on the device: only what zrk_ccp_step reads (positions of this tick and the one before,
    start_time -- prev_pos is None where it equals `now` --, speed_mod)."""

    def __init__(self, cap):
        from zrk_modulation_amd import _lib
        self.cap = cap
        self.pos = [torch.zeros(3, cap, dtype=torch.float64, device="cuda:0") for _ in range(2)]
        self.t0 = torch.full((cap,), -1.0, dtype=torch.float64, device="cuda:0")
        self.speed = torch.zeros(cap, dtype=torch.float64, device="cuda:0")
        e = self.ents = _lib.ZrkEntities()
        e.capacity = cap
        e.start_time = self.t0.data_ptr()
        e.pos[0], e.pos[1] = self.pos[0].data_ptr(), self.pos[1].data_ptr()

    def set_tick(self, pos, prev, prev_none, speed, now_s):
        self.pos[0].copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(pos).T)))
        self.pos[1].copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(prev).T)))
        self.t0.copy_(torch.from_numpy(np.where(prev_none, now_s, -1.0)))
        self.speed.copy_(torch.from_numpy(np.ascontiguousarray(speed, np.float64)))


def _first_occurrences(seq):
    _, first = np.unique(seq, return_index=True)
    return np.asarray(seq)[np.sort(first)].astype(np.int32)


@pytest.mark.parametrize("rounds,grid", [(1, "0"), (3, "0"), (12, "0"), (1024, "0"), (3, "1"), (12, "1")])
def test_device_command_post_reproduces_the_reference_fixture(rounds, grid, monkeypatch):
    """The 36 ticks the reference's CombatControlPoint decided (ccp_step.npz): the device, fed the same air picture, names
    the same verdict, the same matched key and the same launcher for every processed detection, tick after tick (its
    dictionaries carry over on the device).  Whatever the bound on the parallel rounds: with 1 the workgroup that walks the
    leftovers in order decides nearly everything, with 1024 nothing."""
    from zrk_modulation_amd.association import DeviceCommandPost
    monkeypatch.setenv("ZRK_CCP_GRID", grid)                      # (the candidate pass: tiled all-pairs, or through the spatial index)
    fx = np.load(Path(__file__).parent / "golden" / "ccp_step.npz")
    meta = json.loads(str(fx["meta"]))
    N = len(fx["obj_id"])
    tab = Table(N)
    # (a tight swarm under the reference's hundred-step gates: every track is in every detection's gate, detections displace
    # one another down long chains, and each link of a chain costs a round)
    post = DeviceCommandPost(_ctx(), "cuda:0", N, N, meta["launcher_pos"], np.zeros(len(meta["capacity"]), np.int32), dmax=N, rounds=rounds)
    dt = meta["dt_ms"] / 1000
    slack = meta["slack_steps"] * dt
    launches = 0
    for k in range(len(fx["seq_off"]) - 1):
        now = k * meta["dt_ms"] / 1000
        if k == 1:
            post.l_cap.copy_(torch.tensor(meta["capacity"], dtype=torch.int32))
        for mi, _ in fx["new_missile"][fx["new_missile_off"][k]:fx["new_missile_off"][k + 1]]:
            post.add_missile(int(mi), now)
        tab.set_tick(fx["pos"][k], fx["prev"][k], fx["prev_none"][k], fx["speed"], now)
        seq = _first_occurrences(fx["seq_obj"][fx["seq_off"][k]:fx["seq_off"][k + 1]])      # (the reference skips processed ids, :414)
        d_seq = torch.from_numpy(seq).cuda()
        d_cnt = torch.tensor([len(seq)], dtype=torch.int32, device="cuda:0")
        post.step(tab.ents, 0, tab.speed, d_seq, d_cnt, now, slack)
        rows, verdict, match, launcher = post.results()
        a, b = fx["out_off"][k], fx["out_off"][k + 1]
        assert np.array_equal(rows, fx["out_obj"][a:b]) and np.array_equal(verdict, fx["out_verdict"][a:b]), f"tick {k}: verdicts differ"
        tt_key, tm_key = post.tt_key.cpu().numpy(), post.tm_key.cpu().numpy()
        key = np.where(verdict == 1, tt_key[np.maximum(match, 0)], np.where(verdict == 2, tm_key[np.maximum(match, 0)], -1))
        assert np.array_equal(key, fx["out_match"][a:b]), f"tick {k}: matched keys differ"
        lid = np.array([meta["launcher_ids"][l] if l >= 0 else -1 for l in launcher])
        assert np.array_equal(lid, fx["out_launcher"][a:b]), f"tick {k}: launchers differ"
        launches += int((launcher >= 0).sum())
    assert launches == sum(meta["capacity"])


@pytest.mark.parametrize("grid", ["0", "1"], ids=["all-pairs", "spatial-index"])
@pytest.mark.parametrize("seed,n,D,L,cap_per,ticks", [(1, 3000, 1200, 3, 40, 6), (2, 100_000, 10_000, 5, 700, 3), (3, 500, 500, 1, 3, 6)])
def test_device_command_post_equals_the_oracle_over_ticks(seed, n, D, L, cap_per, ticks, grid, monkeypatch):
    """Random air pictures over several ticks (the dictionaries carry over, new targets keep appearing, tracks are matched,
    lost and found again under another key, launchers run dry in detection order): the device against the oracle's literal
    loop.  The largest case: 10 000 detections a tick against 100 000 tracks."""
    from oracle import oracle as O
    from zrk_modulation_amd.association import DeviceCommandPost
    monkeypatch.setenv("ZRK_CCP_GRID", grid)
    g = np.random.Generator(np.random.PCG64(seed))
    p0 = g.uniform(-6e4, 6e4, (n, 3)) * [1, 1, 0.1]
    vel = g.normal(0, 250, (n, 3)) * [1, 1, 0.2]
    speed = np.linalg.norm(vel, axis=1)
    lpos = g.uniform(-2e4, 2e4, (L, 3)) * [1, 1, 0]
    caps = np.full(L, cap_per, np.int32)
    tab = Table(n)
    post = DeviceCommandPost(_ctx(), "cuda:0", n, n, lpos, caps, dmax=n, rounds=2 if seed == 3 else 12)
    ora = O.CcpState(n, lpos, caps)
    dt, slack_steps = 0.5, 4
    # the first tick sees everything (the dictionaries fill up), the later ones a random subset, some of it moved so far
    # that its old track is out of gate (a new target whose id is a key already: replaced in place)
    pos = p0.copy()
    total = np.zeros(3, np.int64)
    launches = 0
    for k in range(ticks):
        now = k * dt
        prev = pos.copy()
        pos = p0 + vel * now + g.normal(0, 5, (n, 3))
        if k >= 2:
            jump = g.choice(n, max(1, n // 50), replace=False)
            pos[jump] += g.normal(0, 3000, (len(jump), 3))
        none = np.zeros(n, bool) if k else np.ones(n, bool)
        seq = (np.arange(n) if k == 0 else g.permutation(n)[:D]).astype(np.int32)
        tab.set_tick(pos, prev, none, speed, now)
        post.step(tab.ents, 0, tab.speed, torch.from_numpy(seq).cuda(), torch.tensor([len(seq)], dtype=torch.int32, device="cuda:0"), now,
                  slack_steps * dt)
        got = post.results()
        want = ora.step(seq, pos, prev, none, speed, now, slack_steps * dt)
        for name, a, b in zip(("rows", "verdicts", "matches", "launchers"), got, want):
            assert np.array_equal(a, b), f"tick {k}: {name} differ in {int((a != b).sum())} of {len(a)} places"
        for v in range(3):
            total[v] += int((want[1] == v).sum())
        launches += int((want[3] >= 0).sum())
        # the dictionaries themselves
        ntt = int(ora.n_tt[0])
        assert post.counts.cpu().tolist() == [ntt, int(ora.n_tm[0])]
        for name in ("tt_key", "tt_obj", "tt_follow"):
            assert np.array_equal(getattr(post, name)[:ntt].cpu().numpy(), getattr(ora, name)[:ntt]), f"tick {k}: {name}"
        assert np.array_equal(post.tt_upd[:ntt].cpu().numpy(), ora.tt_upd[:ntt])
        assert np.array_equal(post.l_launched.cpu().numpy(), ora.l_launched)
    assert total[0] >= n and total[1] > D // 2 and launches == min(L * cap_per, int(total[0] + total[1]))


def test_spatial_index_names_the_candidates_the_all_pairs_pass_names(monkeypatch):
    """3*10^5 tracks, 3*10^4 detections a tick, with everything the index has to get right at once: tracks of very different
    ages (annuli from metres to the whole scene), tracks and detections at the same point, positions that are not numbers,
    a detection faster than anything else, speeds of zero.  Two command posts, one per candidate pass, fed the same ticks:
    same verdicts, matches, launchers and dictionaries -- and the index is what makes 10^5 x 10^6 a matter of milliseconds."""
    import time
    from zrk_modulation_amd.association import DeviceCommandPost
    n, D, L = 300_000, 30_000, 4
    g = np.random.Generator(np.random.PCG64(77))
    p0 = g.uniform(-1.5e5, 1.5e5, (n, 3)) * [1, 1, 0.05]
    vel = g.normal(0, 250, (n, 3)) * [1, 1, 0.2]
    speed = np.linalg.norm(vel, axis=1)
    speed[g.choice(n, 50, replace=False)] = 0.0
    speed[7] = 40_000.0                                           # reaches across hundreds of cells
    lpos = g.uniform(-2e4, 2e4, (L, 3)) * [1, 1, 0]
    caps = np.full(L, 2000, np.int32)
    tab = Table(n)
    posts = {}
    for mode in ("0", "1"):
        posts[mode] = DeviceCommandPost(_ctx(), "cuda:0", n, n, lpos, caps, dmax=n, rounds=8)
    dt, slack_steps = 0.5, 4
    pos = p0.copy()
    spent = {"0": 0.0, "1": 0.0}
    for k in range(5):
        now = k * dt
        prev = pos.copy()
        pos = p0 + vel * now + g.normal(0, 5, (n, 3))
        if k >= 1:
            same = g.choice(n, 200, replace=False)
            pos[same[:100]] = prev[same[100:]]                    # exactly on another track's reference position
            pos[g.choice(n, 20, replace=False)] = np.nan
        none = np.zeros(n, bool) if k else np.ones(n, bool)
        # every tick a different share of the table is seen: the tracks' ages spread out over the ticks
        seq = (np.arange(n) if k == 0 else g.permutation(n)[:D]).astype(np.int32)
        tab.set_tick(pos, prev, none, speed, now)
        got = {}
        for mode, post in posts.items():
            monkeypatch.setenv("ZRK_CCP_GRID", mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            post.step(tab.ents, 0, tab.speed, torch.from_numpy(seq).cuda(), torch.tensor([len(seq)], dtype=torch.int32, device="cuda:0"), now,
                      slack_steps * dt)
            torch.cuda.synchronize()
            spent[mode] += time.perf_counter() - t0
            got[mode] = post.results()
        for name, a, b in zip(("rows", "verdicts", "matches", "launchers"), got["0"], got["1"]):
            assert np.array_equal(a, b), f"tick {k}: {name} differ in {int((a != b).sum())} of {len(a)} places"
        assert posts["0"].counts.cpu().tolist() == posts["1"].counts.cpu().tolist()
        ntt = int(posts["0"].counts[0])
        for name in ("tt_key", "tt_obj", "tt_follow", "tt_upd"):
            assert torch.equal(getattr(posts["0"], name)[:ntt], getattr(posts["1"], name)[:ntt]), f"tick {k}: {name}"
    v = got["0"][1]
    assert (v == 1).sum() > D // 2 and (v == 0).sum() > 0
    print(f"\n5 ticks of the command post at 3e5 tracks: all pairs {spent['0'] * 1e3:.1f} ms, spatial index {spent['1'] * 1e3:.1f} ms")
    # (no assertion on the times: tools/ccp_scale.py measures the passes at scale, 1.5 against 241 ms a tick)


def test_launch_requests_feed_the_salvo_without_the_host():
    """The chain sweep -> lists -> command post -> launch requests -> salvo with nothing read back in between: the requests the
    device builds from the step's decisions (zrk_ccp_requests) are, byte for byte, the ones a host would build from the step's
    results, in request order, padded with requests for no row; the salvo launched from them for the padded count leaves the
    tables a salvo of exactly those requests leaves, the padding's rows dead behind; the loop carries on."""
    import ctypes as C
    from zrk_modulation_amd import _lib, scenario as S
    from zrk_modulation_amd.association import DeviceCommandPost
    from zrk_modulation_amd.engine import HotPathEngine
    n, k_max = 60_000, 1500
    lpos = np.array([[0.0, 0.0, 0.0], [4000.0, -1500.0, 10.0], [-2500.0, 3000.0, 5.0]])
    caps = np.array([400, 300, 500], np.int32)                    # 1200 missiles in all: fewer than the detections that ask
    params = np.array([[1000.0, 60.0, 150.0], [900.0, 45.0, 120.0], [1100.0, 30.0, 200.0]])
    engines, outs = [], []
    for twin in range(2):
        ids, sp, vel, t0 = S.synthetic_targets(n, 31)
        eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise="off")
        eng.load(ids, sp, vel, t0, S.synthetic_radars(4), missile_capacity=k_max, sort=False).enable_lists()
        eng.run(3)
        engines.append(eng)
    eng, twin = engines
    st = eng.store
    lists = eng.detections()
    seq = _first_occurrences(np.concatenate(lists))               # FoundObjectsMessage order, every object once (rows: unsorted table)
    assert len(seq) > 2 * caps.sum()
    speed = torch.linalg.vector_norm(st.d_vel[:, :st.cap], dim=0).contiguous()
    post = DeviceCommandPost(st.ctx, "cuda:0", st.cap, st.cap, lpos, caps, dmax=len(seq))
    now = eng.loop.time_ms / 1000
    post.step(st.ents, st.cur, speed, torch.from_numpy(seq).cuda(), torch.tensor([len(seq)], dtype=torch.int32, device="cuda:0"), now, 1.0)
    d_req, d_count = post.requests(params, k_max)
    rows, verdicts, matches, launchers = post.results()
    asked = np.nonzero(launchers >= 0)[0]
    assert len(asked) == caps.sum() and int(d_count.item()) == len(asked)     # every missile handed out, in detection order
    req_t, res_t = _lib.launch_dtypes()
    want = np.zeros(k_max, req_t)
    want["target_slot"] = -1
    want["target_slot"][:len(asked)] = rows[asked]
    want["missile_pos"][:len(asked)] = lpos[launchers[asked]]
    want["speed"][:len(asked)], want["period"][:len(asked)], want["radius"][:len(asked)] = params[launchers[asked]].T
    got = d_req.cpu().numpy().view(req_t)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    # the salvo from the device's padded list against a salvo of exactly the requests, on a twin table
    n0, m0 = st.n_uploaded, st.m
    count = eng.launch_requests_on_device(d_req, k_max)
    exact = torch.from_numpy(want[:len(asked)].view(np.uint8).reshape(-1).copy()).cuda()
    count_twin = twin.launch_requests_on_device(exact, len(asked))
    c = int(count.item())
    assert c == int(count_twin.item()) and 0 < c <= len(asked)
    res = eng._last_device_results.cpu().numpy().view(res_t)
    assert (res["rc"][len(asked):] == 6).all() and set(np.unique(res["rc"][:len(asked)])) <= {0, 1, 2, 3, 4, 5}
    a, b = st, twin.store
    live = slice(n0, n0 + c)
    for name in ("d_sp", "d_vel"):
        assert torch.equal(getattr(a, name)[:, live], getattr(b, name)[:, live]), name
    assert torch.equal(a.d_t0[live], b.d_t0[live]) and a.d_alive[live].all() and b.d_alive[live].all()
    assert not a.d_alive[n0 + c:n0 + k_max].any()
    for name in ("dm_slot", "dm_tgt", "dm_period", "dm_status"):
        assert torch.equal(getattr(a, name)[m0:m0 + c], getattr(b, name)[m0:m0 + c]), name
    assert not a.dm_status[m0 + c:m0 + k_max].any()
    eng.run(6)
    twin.run(6)
    assert eng.alive_count() == twin.alive_count()
    assert np.array_equal(eng.store.host_pos("cur")[n0:n0 + c].view(np.uint64), twin.store.host_pos("cur")[n0:n0 + c].view(np.uint64))
    # the rows behind the successes are given back before the next salvo: the tables do not grow by k_max dead rows per call
    assert st.n_uploaded == n0 + k_max and st.m == m0 + k_max
    assert eng.settle_device_launches() == k_max - c and twin.settle_device_launches() == len(asked) - c
    assert st.n_uploaded == n0 + c == twin.store.n_uploaded and st.m == m0 + c == twin.store.m
    again = torch.from_numpy(want[:200].view(np.uint8).reshape(-1).copy()).cuda()
    c2 = [int(e.launch_requests_on_device(again, 200).item()) for e in (eng, twin)]
    assert c2[0] == c2[1] > 0 and eng.store.n_uploaded == n0 + c + 200
    eng.run(5)
    twin.run(5)
    assert eng.alive_count() == twin.alive_count()
    assert np.array_equal(eng.store.host_pos("cur")[:n0 + c + c2[0]].view(np.uint64), twin.store.host_pos("cur")[:n0 + c + c2[0]].view(np.uint64))
    for r, (x, y) in enumerate(zip(eng.detections(), twin.detections())):
        assert np.array_equal(x, y), f"radar {r}"
