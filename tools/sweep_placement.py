"""Where the sweep's waves run (diagnostics build): per-SIMD totals of radar-loop time and of expensive waves,
so that "the kernel ends when the most loaded SIMD ends" can be told from "every SIMD is equally busy".
usage: sweep_placement.py [n] [R] [ticks_before]"""
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
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise="philox")
eng.load(ids, sp, vel, t0, S.synthetic_radars(R)).enable_lists()
eng.run(warm)
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
deep = t[:, 5]
hw = t[:, 6] & 0xFFFFFFFF
xcc = (t[:, 6] >> 32) & 0xF
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
cukey = key >> 2
uniq, inv = np.unique(key, return_inverse=True)
print(f"n={n} R={R}: {nw} waves on {len(uniq)} SIMDs of {len(np.unique(cukey))} CUs, {len(np.unique(xcc))} XCDs; kernel {end.max():.2f} us")
loop = swept - loaded
heavy = deep >= 6
per_loop = np.bincount(inv, weights=loop)
per_heavy = np.bincount(inv, weights=heavy.astype(float))
per_waves = np.bincount(inv)
per_end = np.zeros(len(uniq)); np.maximum.at(per_end, inv, end)
per_first = np.full(len(uniq), 1e9); np.minimum.at(per_first, inv, start)


def q(x):
    return " ".join(f"{v:7.2f}" for v in np.percentile(x, [0, 10, 50, 90, 100]))


print("per SIMD, percentiles 0 10 50 90 100")
print("waves               ", q(per_waves))
print("heavy waves (>=6 deep)", q(per_heavy))
print("sum radar-loop [us] ", q(per_loop))
print("last wave ends [us] ", q(per_end))
print("first wave starts   ", q(per_first))
print("corr(end, heavy) = %.2f   corr(end, sum loop) = %.2f" % (np.corrcoef(per_end, per_heavy)[0, 1], np.corrcoef(per_end, per_loop)[0, 1]))
# per XCD
for x in np.unique(xcc):
    sel = xcc == x
    print(f"  xcd {x}: waves {sel.sum():6d} heavy {int(heavy[sel].sum()):5d} end {end[sel].max():6.2f}")
# the SIMDs that end last
worst = np.argsort(per_end)[-8:]
for w in worst:
    sel = inv == w
    print(f"  late SIMD {uniq[w]:6d}: waves {per_waves[w]:3d} heavy {int(per_heavy[w]):2d} sum loop {per_loop[w]:6.2f} end {per_end[w]:6.2f}  "
          f"starts of its heavy waves: {np.sort(start[sel & heavy]).round(1).tolist()}")
# heavy waves: duration against how many heavy waves share the SIMD at the time
hs = np.nonzero(heavy)[0]
print("heavy wave radar-loop duration percentiles:", q(loop[hs]), " light:", q(loop[~heavy]))
print("heavy wave start percentiles:", q(start[hs]), " end:", q(end[hs]))

walked = t[:, 7] & 0xFF; n_in = (t[:, 7] >> 8) & 0xFF; n_pl = (t[:, 7] >> 16) & 0xFF
print("NOTE: probe slot 7 is indexed by dispatch position, not by wave; totals only")
print("radars walked per wave: mean %.2f; of which certainly inside %.2f, inside but for the plane %.2f" % (walked.mean(), n_in.mean(), n_pl.mean()))
hv = walked >= 6
print("waves walking >= 6 radars: %.1f %%; among them inside %.2f plane %.2f of %.2f walked" % (100 * hv.mean(), n_in[hv].mean(), n_pl[hv].mean(), walked[hv].mean()))
