"""Per-wave timeline of the sweep kernel (diagnostics build): when each wave starts, how long it waits for
its row, how long the radar loop takes, when it ends.  usage: sweep_phases.py [n] [R]"""
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
ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise="philox")
eng.load(ids, sp, vel, t0, S.synthetic_radars(R)).enable_lists()
eng.run(int(sys.argv[3]) if len(sys.argv) > 3 else 20)
torch.cuda.synchronize()
nw = (n + 63) // 64
buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda:0")
lib = _lib.load()
lib.zrk_debug_wave_probe.argtypes = [C.c_void_p]
assert lib.zrk_debug_wave_probe(buf.data_ptr()) == 0
eng.run(1)
torch.cuda.synchronize()
assert lib.zrk_debug_wave_probe(None) == 0
t = buf.cpu().numpy().reshape(nw, 8)
t0_ = t[:, 0].min()
start, loaded, swept, end = [(t[:, k] - t0_) * 0.01 for k in range(4)]
det, deep = t[:, 4], t[:, 5]


def q(x):
    return " ".join(f"{v:7.2f}" for v in np.percentile(x, [0, 10, 50, 90, 100]))


print(f"n={n} R={R}: {nw} waves; percentiles 0 10 50 90 100 [us]")
print("start              ", q(start))
print("row arrived - start", q(loaded - start))
print("radar loop         ", q(swept - loaded))
print("stores issued      ", q(end - swept))
print("end                ", q(end))
print("waves with a detection: %.1f %%; radars past both early-outs per wave: mean %.2f" % (100 * (det > 0).mean(), deep.mean()))
for lo, hi in [(0, 1), (1, 3), (3, 6), (6, 17)]:
    sel = (deep >= lo) & (deep < hi)
    if sel.any():
        print(f"  waves with {lo}..{hi - 1} deep radars: {100 * sel.mean():5.1f} %, radar loop median {np.median((swept - loaded)[sel]):6.2f} us")
order = np.argsort(start)
k = nw // 8
print("radar-loop median by start-time octile:", " ".join(f"{np.median((swept - loaded)[order[j * k:(j + 1) * k]]):6.2f}" for j in range(8)))
print("row-wait   median by start-time octile:", " ".join(f"{np.median((loaded - start)[order[j * k:(j + 1) * k]]):6.2f}" for j in range(8)))
print("start      median by start-time octile:", " ".join(f"{np.median(start[order[j * k:(j + 1) * k]]):6.2f}" for j in range(8)))
print("t [us]   resident waves   rows arriving /us (x64)   waves in the radar loop   finishing /us")
for tt in range(0, int(end.max()) + 2):
    res = int(((start <= tt) & (end > tt)).sum())
    arr = int(((loaded >= tt) & (loaded < tt + 1)).sum())
    inloop = int(((loaded <= tt) & (swept > tt)).sum())
    fin = int(((end >= tt) & (end < tt + 1)).sum())
    print(f"{tt:5d} {res:12d} {arr:18d} {inloop:24d} {fin:18d}")
# the slowest waves: who are they?
st = eng.store
slow = np.argsort(swept - loaded)[-8:]
hp = st.host_pos("cur")
for w in slow:
    rows = np.arange(w * 64, min(w * 64 + 64, st.n_uploaded))
    p = hp[rows]
    print(f"slow wave {w}: loop {float((swept - loaded)[w]):6.2f} us start {float(start[w]):5.2f} deep {int(deep[w])} det lanes {int(det[w])} probe7 {int(t[w,7]) & 0xFF} in {(int(t[w,7])>>8)&0xFF} pl {(int(t[w,7])>>16)&0xFF} "
          f"box x [{p[:,0].min():9.1f},{p[:,0].max():9.1f}] y [{p[:,1].min():9.1f},{p[:,1].max():9.1f}] kinds {np.bincount(st.h_kind[rows], minlength=2).tolist()}")
