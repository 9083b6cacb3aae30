"""Is the dispatcher's placement of a sweep's workgroups the same from launch to launch?  Two pair launches inside the
overlapped loop (diagnostics build): for every place in the dispatch (blockIdx), the compute unit its workgroup ran on.
usage: placement_repeat.py [n] [R] [m]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from zrk_modulation_amd import _lib  # noqa: E402

_lib.LIB_PATH = _lib.CSRC / "libzrk_hot_probe.so"
from zrk_modulation_amd import scenario as S  # noqa: E402
from zrk_modulation_amd.engine import HotPathEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise="philox")
eng.load(ids, sp, vel, t0, S.synthetic_radars(R), missile_capacity=m).enable_lists()
if m:
    eng.launch_missiles(S.missile_targets(n, m))
eng.run(40)
torch.cuda.synchronize()
nw = (eng.store.n_uploaded + 63) // 64
lib = _lib.load()
lib.zrk_debug_wave_probe.argtypes = [C.c_void_p]


def capture():
    buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda:0")
    assert lib.zrk_debug_wave_probe(buf.data_ptr()) == 0
    eng.run(8)
    torch.cuda.synchronize()
    assert lib.zrk_debug_wave_probe(None) == 0
    t = buf.cpu().numpy().reshape(nw, 8)
    t = t[(t[:, 0] > 0)]
    t = t[np.arange(len(t)) % 4 == 0] if False else t
    hw = t[:, 6]
    bid = (hw >> 40) & 0xFFFFFF
    hwid, xcc = hw & 0xFFFFFFFF, (hw >> 32) & 0xF
    cu = ((xcc * 8 + ((hwid >> 13) & 7)) * 2 + ((hwid >> 12) & 1)) * 16 + ((hwid >> 8) & 15)
    start = (t[:, 0] - t[:, 0].min()) * 0.01
    out = np.full(int(bid.max()) + 1, -1, np.int64)
    out[bid] = cu                      # (the four waves of a workgroup agree)
    st = np.zeros(int(bid.max()) + 1)
    st[bid] = start
    return out, st, (t[:, 7] & 0xFF)


a, sa, _ = capture()
b, sb, _ = capture()
k = min(len(a), len(b))
a, b, sa = a[:k], b[:k], sa[:k]
ok = (a >= 0) & (b >= 0)
print(f"{int(ok.sum())} places seen in both launches")
for lo, hi in ((0, 256), (256, 1024), (1024, 1792), (1792, 2600), (2600, k)):
    s = ok[lo:hi]
    same = (a[lo:hi] == b[lo:hi]) & s
    print(f"places {lo:5d}..{hi:5d}: same compute unit in both launches {100 * same.sum() / max(1, s.sum()):5.1f} %   "
          f"(same XCD {100 * ((a[lo:hi] // 256 == b[lo:hi] // 256) & s).sum() / max(1, s.sum()):5.1f} %), started at {np.median(sa[lo:hi]):.1f} us (median)")
