"""GPU parity proper: the HIP path, called through the C ABI, against (a) the golden vectors captured
from the reference and (b) the CPU oracle on seeded random scenes."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import ALL_FIXTURES, Fixture, populate, replay_l1

pytestmark = pytest.mark.gpu


def make_device_sim(fx, **kw):
    from zrk_modulation_amd.engine import DeviceSim
    sim = DeviceSim(fx.dt, **kw)
    populate(sim, fx)
    return sim


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_device_replays_reference(name):
    """Detected ids, detonations, launch solves (rc and V bits), scan state and position bits of
    every tick of every fixture, with the reference's own noise stream."""
    fx = Fixture(name)
    stats = replay_l1(make_device_sim(fx), fx, check_pos="bits")
    assert stats["found"] == len(fx.found_ids)
    assert stats["detonations"] == len(fx.detonations)
    assert stats["launches"] == len(fx.launch_cmd)


@pytest.mark.parametrize("name", ["edges_zero_noise", "edges", "bulk_n1000_r4"])
def test_exact_only_path_agrees(name):
    """ZRK_F_EXACT_ONLY (every in-range pair decided in binary64) gives the same answers as the
    float32 pre-classification + binary64 fallback."""
    fx = Fixture(name)
    replay_l1(make_device_sim(fx, exact_only=True), fx, check_pos="bits", max_ticks=120)


def _ctx_and_buffers(n):
    import torch
    from zrk_modulation_amd import _lib
    ctx = _lib.Context(0)
    return ctx, torch


@pytest.mark.parametrize("op,name", [(0, "sqrt"), (1, "div"), (4, "norm")])
def test_device_binary64_ops_are_correctly_rounded(op, name):
    """sqrt, divide and the fma-chain norm must match IEEE/numpy bit for bit: they feed positions,
    launch velocities and the fuse distance."""
    ctx, torch = _ctx_and_buffers(0)
    g = np.random.Generator(np.random.PCG64(99 + op))
    n = 1 << 20
    a = np.abs(g.normal(0, 1e4, n)) * 10.0 ** g.integers(-8, 8, n)
    b = g.normal(0, 1e3, n) * 10.0 ** g.integers(-6, 6, n)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    dy = torch.zeros(n, dtype=torch.float64, device="cuda")
    ctx.check(ctx.lib.zrk_selftest_math(ctx.handle, op, da.data_ptr(), db.data_ptr(), dy.data_ptr(), n, None), "selftest")
    y = dy.cpu().numpy()
    if op == 0:
        want = np.sqrt(a)
    elif op == 1:
        want = a / b
    else:
        from oracle import oracle as O
        L = O.lib()
        want = np.array([L.zo_norm3(float(x), float(z), 0.0) for x, z in zip(a[:20000], b[:20000])])
        y = y[:20000]
    assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), f"{name}: device differs from IEEE result"


@pytest.mark.parametrize("op,name,fn", [(2, "atan2", np.arctan2), (3, "asin", np.arcsin)])
def test_device_angles_within_2ulp(op, name, fn):
    """Angles only feed comparisons; they need to be accurate, not bit-identical."""
    ctx, torch = _ctx_and_buffers(0)
    g = np.random.Generator(np.random.PCG64(7 + op))
    n = 1 << 18
    a = g.normal(0, 1e4, n) if op == 2 else g.uniform(-1, 1, n)
    b = g.normal(0, 1e4, n)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    dy = torch.zeros(n, dtype=torch.float64, device="cuda")
    ctx.check(ctx.lib.zrk_selftest_math(ctx.handle, op, da.data_ptr(), db.data_ptr(), dy.data_ptr(), n, None), "selftest")
    y = dy.cpu().numpy()
    want = fn(a, b) if op == 2 else fn(a)
    ulp = np.abs(y - want) / np.spacing(np.abs(want))
    assert ulp.max() <= 2.0, f"{name}: max error {ulp.max()} ulp"
    # the exactly representable cases sector edges are made of
    ea = np.array([0.0, 1.0, 0.0, -1.0, 1.0, -0.0, 5.0]) if op == 2 else np.array([0.0, 1.0, -1.0, -0.0, 0.5, 0.0, 0.0])
    eb = np.array([1.0, 0.0, -1.0, 0.0, 1.0, 1.0, 5.0])
    da, db = torch.from_numpy(ea).cuda(), torch.from_numpy(eb).cuda()
    dy = torch.zeros(len(ea), dtype=torch.float64, device="cuda")
    ctx.check(ctx.lib.zrk_selftest_math(ctx.handle, op, da.data_ptr(), db.data_ptr(), dy.data_ptr(), len(ea), None), "selftest")
    got = dy.cpu().numpy()
    want = fn(ea, eb) if op == 2 else fn(ea)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), f"{name}: exact cases differ: {got} vs {want}"


def test_philox_noise_matches_oracle_stream():
    """Same Philox / xoshiro integers on both sides; the float32 Box-Muller differs only by the
    hardware log2/sin/cos approximations."""
    from oracle import oracle as O
    ctx, torch = _ctx_and_buffers(0)
    n = 4096
    for ordinal in (0, 3):
        out = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        ctx.check(ctx.lib.zrk_selftest_noise(ctx.handle, 1234, 77, ordinal, 1000, out.data_ptr(), n, None), "noise")
        got = out.cpu().numpy()
        want = np.array([O.philox_noise(1234, 77, ordinal, 1000 + i) for i in range(n)])
        assert np.abs(got - want).max() < 5e-4
        assert abs(got.std() - 5.0) < 0.15 and abs(got.mean()) < 0.15


def test_throughput_noise_is_gaussian():
    """Distribution of the device noise: N(0, 5^2) per axis, axes and detections uncorrelated
    (what the reference's np.random.normal(0, 5, 3) provides, modules/Radar.py:138-142)."""
    from scipy import stats
    ctx, torch = _ctx_and_buffers(0)
    n = 1 << 18
    outs = []
    for ordinal in (0, 1):
        out = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        ctx.check(ctx.lib.zrk_selftest_noise(ctx.handle, 99, 5, ordinal, 0, out.data_ptr(), n, None), "noise")
        outs.append(out.cpu().numpy())
    z = np.concatenate(outs, axis=1)                       # [n, 6]
    for c in range(6):
        assert stats.kstest(z[:, c] / 5.0, "norm").pvalue > 1e-4
    corr = np.corrcoef(z.T)
    assert np.abs(corr - np.eye(6)).max() < 0.01
