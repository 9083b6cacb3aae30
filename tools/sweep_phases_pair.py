"""Per-wave timeline of a PAIR sweep launch inside the overlapped loop (diagnostics build, `make -C zrk_modulation_amd/csrc
probe`): when each wave starts, how long it waits for its rows, how long the two radar loops take, when it ends -- with
the previous pair's compaction running beside it, as in the loop bench.py times.  usage: sweep_phases_pair.py [n] [R] [m]"""
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
nrow = eng.store.n_uploaded
nw = (nrow + 63) // 64
buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda:0")
lib = _lib.load()
lib.zrk_debug_wave_probe.argtypes = [C.c_void_p]
assert lib.zrk_debug_wave_probe(buf.data_ptr()) == 0
eng.run(8)                                   # four pair launches: the stamps of the last one stand
torch.cuda.synchronize()
assert lib.zrk_debug_wave_probe(None) == 0
t = buf.cpu().numpy().reshape(nw, 8)
ok = t[:, 0] > 0
t = t[ok]
t0_ = t[:, 0].min()
start, loaded, loop1, end, loop2 = [(t[:, k] - t0_) * 0.01 for k in range(5)]
walked = t[:, 7] & 0xFF


def q(x):
    return " ".join(f"{v:7.2f}" for v in np.percentile(x, [0, 10, 50, 90, 100]))


print(f"n={nrow} R={R} m={m}: {len(t)} waves stamped; percentiles 0 10 50 90 100 [us]")
print("start                   ", q(start))
print("rows arrived - start    ", q(loaded - start))
print("tick t radar loop       ", q(loop1 - loaded))
print("tick t stores + t+1 loop", q(loop2 - loop1))
print("tail (stores)           ", q(end - loop2))
print("end                     ", q(end))
light = walked == 0
print(f"light waves (no radar to walk in tick t): {100 * light.mean():.1f} %; their lifetime {q((end - start)[light])}")
print(f"heavy waves: lifetime {q((end - start)[~light])}")
print(f"wave-slot time: light {float((end - start)[light].sum()):.0f} us, heavy {float((end - start)[~light].sum()):.0f} us, "
      f"kernel {float(end.max()):.1f} us x 7168 slots = {float(end.max()) * 7168:.0f}")
print("t [us]   resident waves   rows arriving /us   waves in a radar loop   finishing /us")
for tt in range(0, int(end.max()) + 2):
    res = int(((start <= tt) & (end > tt)).sum())
    arr = int(((loaded >= tt) & (loaded < tt + 1)).sum())
    inloop = int(((loaded <= tt) & (loop2 > tt)).sum())
    fin = int(((end >= tt) & (end < tt + 1)).sum())
    print(f"{tt:5d} {res:12d} {arr:18d} {inloop:24d} {fin:18d}")
