"""zrk_compact alone: the single-launch path (tickets + published counts) and the three-launch path
against numpy, over sizes around the workgroup boundaries, every radar count class and mask density."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expected(vis, R):
    lists = [np.nonzero((vis >> r) & 1)[0].astype(np.int32) for r in range(R)]
    seen = np.nonzero(vis)[0]
    return lists, seen


def _run(store_factory, vis, R, base_index=0, gid0=0, stride=None, packed_cap=None):
    import torch
    st = store_factory()
    n = len(vis)
    dvis = torch.as_tensor(vis.view(np.int32), device=st.device)
    stride = n if stride is None else stride
    det = torch.full((max(stride * R, 1),), -7, dtype=torch.int32, device=st.device)
    cnt = torch.full((R + 1,), -7, dtype=torch.int32, device=st.device)
    pcap = n + 1 if packed_cap is None else packed_cap
    packed = torch.full((pcap,), -7, dtype=torch.int64, device=st.device)
    st.ctx.check(st.lib.zrk_compact(st.ctx.handle, dvis.data_ptr(), n, R, base_index, st.workspace().data_ptr(),
                                    det.data_ptr(), stride, cnt.data_ptr(), packed.data_ptr(), pcap, gid0, None), "compact")
    st.compact_status()
    return det.cpu().numpy(), cnt.cpu().numpy(), packed.cpu().numpy()


def _check(det, cnt, packed, vis, R, base_index=0, gid0=0, stride=None):
    n = len(vis)
    stride = n if stride is None else stride
    lists, seen = _expected(vis, R)
    assert cnt.tolist() == [len(x) for x in lists] + [len(seen)]
    for r in range(R):
        k = min(len(lists[r]), stride)
        assert np.array_equal(det[r * stride:r * stride + k], lists[r][:k] + base_index), f"radar {r}"
        assert (det[r * stride + k:(r + 1) * stride] == -7).all(), f"radar {r} wrote past its list"
    assert packed[0] == len(seen)
    k = min(len(seen), len(packed) - 1)
    want = ((seen[:k].astype(np.int64) + gid0) << 32) | vis[seen[:k]].astype(np.int64)
    assert np.array_equal(packed[1:1 + k], want)
    assert (packed[1 + k:] == -7).all()


def _masks(g, n, R, density):
    full = np.uint32((1 << R) - 1) if R < 32 else np.uint32(0xFFFFFFFF)
    vis = g.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) & full
    if density < 1.0:
        vis[g.uniform(size=n) >= density] = 0
    return vis


_STORES = {}


@pytest.fixture
def factory(monkeypatch):
    def make(max_blocks=None, items=None):
        from zrk_modulation_amd.store import EntityStore
        if items is not None:
            monkeypatch.setenv("ZRK_COMPACT_ITEMS", str(items))
        else:
            monkeypatch.delenv("ZRK_COMPACT_ITEMS", raising=False)

        def get():
            # the tuning variables are read when a context is created and by zrk_ctx_reload_env
            if max_blocks is not None:
                monkeypatch.setenv("ZRK_COMPACT_FUSED_MAX_BLOCKS", str(max_blocks))
            else:
                monkeypatch.delenv("ZRK_COMPACT_FUSED_MAX_BLOCKS", raising=False)
            if max_blocks not in _STORES:
                _STORES[max_blocks] = EntityStore("cuda:0", capacity=1 << 21)
            st = _STORES[max_blocks]
            st.lib.zrk_ctx_reload_env(st.ctx.handle)
            return st
        return get
    return make


@pytest.mark.parametrize("n", [1, 63, 1023, 1024, 1025, 8191, 8193, 70_001, 1_048_577])
@pytest.mark.parametrize("R,density", [(1, 0.5), (5, 0.15), (16, 0.15), (17, 1.0), (32, 0.02), (32, 1.0), (7, 0.0)])
def test_single_launch_matches_numpy(factory, n, R, density):
    g = np.random.Generator(np.random.PCG64(n * 131 + R))
    vis = _masks(g, n, R, density)
    for items in (None, 1, 3, 8):
        if items == 1 and n > 1_000_000:
            continue                                        # 1025 workgroups: that is the three-launch case below
        det, cnt, packed = _run(factory(items=items), vis, R, base_index=1000, gid0=5_000_000_000)
        _check(det, cnt, packed, vis, R, base_index=1000, gid0=5_000_000_000)


@pytest.mark.parametrize("n,R", [(1, 1), (1025, 16), (300_000, 32), (1_048_577, 16)])
def test_three_launch_path_matches_numpy(factory, n, R):
    g = np.random.Generator(np.random.PCG64(n + R))
    vis = _masks(g, n, R, 0.2)
    det, cnt, packed = _run(factory(max_blocks=0), vis, R, base_index=-3, gid0=77)
    _check(det, cnt, packed, vis, R, base_index=-3, gid0=77)


def test_truncation_keeps_counts_exact(factory):
    g = np.random.Generator(np.random.PCG64(9))
    n, R = 50_000, 6
    vis = _masks(g, n, R, 0.5)
    for mb in (None, 0):
        det, cnt, packed = _run(factory(max_blocks=mb), vis, R, stride=1000, packed_cap=501)
        _check(det, cnt, packed, vis, R, stride=1000)


def test_many_launches_on_one_workspace(factory):
    """Tickets rearm and epochs advance: 200 back-to-back compactions of changing sizes on one store, no
    host synchronisation in between, the last few checked."""
    import torch
    st = factory()()
    g = np.random.Generator(np.random.PCG64(4))
    keep = []
    for k in range(200):
        n = int(g.integers(1, 400_000))
        R = int(g.integers(1, 33))
        vis = _masks(g, n, R, float(g.uniform(0, 0.6)))
        dvis = torch.as_tensor(vis.view(np.int32), device=st.device)
        det = torch.full((n * R,), -7, dtype=torch.int32, device=st.device)
        cnt = torch.full((R + 1,), -7, dtype=torch.int32, device=st.device)
        packed = torch.full((n + 1,), -7, dtype=torch.int64, device=st.device)
        st.ctx.check(st.lib.zrk_compact(st.ctx.handle, dvis.data_ptr(), n, R, 0, st.workspace().data_ptr(), det.data_ptr(),
                                        n, cnt.data_ptr(), packed.data_ptr(), n + 1, 0, None), "compact")
        keep.append((vis, R, det, cnt, packed, dvis))
        keep = keep[-8:]
    st.compact_status()
    for vis, R, det, cnt, packed, _ in keep:
        _check(det.cpu().numpy(), cnt.cpu().numpy(), packed.cpu().numpy(), vis, R)


@pytest.mark.parametrize("order", ["ticket", "block"])
def test_both_waiting_orders(factory, monkeypatch, order):
    """Workgroups ordered by atomic tickets (the rule beyond two workgroups per compute unit) and by blockIdx
    give the same lists; the ticket counter rearms itself between launches."""
    monkeypatch.setenv("ZRK_COMPACT_ORDER", order)
    g = np.random.Generator(np.random.PCG64(77))
    for n, R in [(300_000, 16), (5_000, 3), (1_000_000, 32)]:
        vis = _masks(g, n, R, 0.2)
        for items in (1, 2):
            if n // (1024 * items) >= 1024:
                continue
            det, cnt, packed = _run(factory(items=items), vis, R)
            _check(det, cnt, packed, vis, R)


def test_status_reports_a_foreign_workspace(factory):
    """Control words that are not as the library left them: the launch refuses to scribble, the status call says so,
    and the next use starts from a cleared control area."""
    import torch
    from zrk_modulation_amd._lib import ZrkError
    st = factory()()
    g = np.random.Generator(np.random.PCG64(5))
    vis = _masks(g, 10_000, 4, 0.3)
    det, cnt, packed = _run(lambda: st, vis, 4)
    _check(det, cnt, packed, vis, 4)
    ws = st.workspace()
    ws[:4] = torch.tensor([0x40, 0x42, 0x0F, 0x7F], dtype=torch.uint8, device=ws.device)    # a huge "ticket"
    import os
    os.environ["ZRK_COMPACT_ORDER"] = "ticket"
    st.lib.zrk_ctx_reload_env(st.ctx.handle)
    try:
        with pytest.raises(ZrkError):
            _run(lambda: st, vis, 4)
    finally:
        os.environ.pop("ZRK_COMPACT_ORDER", None)
        st.lib.zrk_ctx_reload_env(st.ctx.handle)
    det, cnt, packed = _run(lambda: st, vis, 4)               # cleared again on the next use
    _check(det, cnt, packed, vis, 4)


@pytest.mark.parametrize("n,R,density", [(1, 1, 1.0), (63, 3, 0.5), (1025, 16, 0.2), (70_001, 17, 0.3), (1_048_577, 16, 0.17),
                                         (300_000, 32, 1.0), (5_000, 8, 0.0)])
def test_union_in_the_wire_format(factory, n, R, density):
    """zrk_compact_bits: count, n, one bit per slot, then the masks of the seen slots (16 or 32 bits wide) -- against
    the host encoder, on the single-launch and the three-launch path, with room for all masks and for too few."""
    import torch
    from zrk_modulation_amd.exchange import encode_union_bits, union_bits_words
    g = np.random.Generator(np.random.PCG64(n + 7 * R))
    vis = _masks(g, n, R, density)
    seen = int(np.count_nonzero(vis))
    for mb in (None, 0):
        st = factory(max_blocks=mb)()
        for entries in (seen, seen // 2):
            words = union_bits_words(n, R, entries)
            assert words == st.lib.zrk_union_bits_words(n, R, entries)
            dvis = torch.as_tensor(vis.view(np.int32), device=st.device)
            out = torch.full((words,), -7, dtype=torch.int64, device=st.device)
            det = torch.zeros(n * R, dtype=torch.int32, device=st.device)
            cnt = torch.zeros(R + 1, dtype=torch.int32, device=st.device)
            st.ctx.check(st.lib.zrk_compact_bits(st.ctx.handle, dvis.data_ptr(), n, R, 0, st.workspace().data_ptr(), det.data_ptr(),
                                                 n, cnt.data_ptr(), out.data_ptr(), words, None), "compact_bits")
            st.compact_status()
            got, want = out.cpu().numpy(), encode_union_bits(vis, R, words)
            nb = (n + 63) // 64
            assert got[0] == seen and got[1] == n and np.array_equal(got[2:2 + nb], want[2:2 + nb])
            dt = np.uint16 if R <= 16 else np.uint32
            k = min(seen, len(want[2 + nb:].view(dt)))
            assert np.array_equal(got[2 + nb:].view(dt)[:k], want[2 + nb:].view(dt)[:k])
            assert cnt.cpu().tolist()[R] == seen
