"""Per-wave timeline of a PAIR sweep launch inside the overlapped loop (diagnostics build, `make -C zrk_modulation_amd/csrc
probe`): when each wave starts, how long it waits for its rows, how long the two radar loops take, when it ends -- with
the previous pair's compaction running beside it, as in the loop bench.py times.  usage: sweep_phases_pair.py [n] [R] [m]"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")        # (as bench.py runs: the sweep's 20 KB of arguments in device memory)
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

# where the heavy waves ran: HW_REG_HW_ID (simd 5:4, cu 11:8, sh 12, se 15:13) and HW_REG_XCC_ID (3:0)
hw = t[:, 6]
hwid, xcc = (hw & 0xFFFFFFFF).astype(np.int64), ((hw >> 32) & 0xF).astype(np.int64)
simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
unit = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
cost = (loop1 - loaded) + (loop2 - loop1)
nu = len(np.unique(unit))
heavy_per = np.bincount(unit[~light], minlength=int(unit.max()) + 1)
cost_per = np.bincount(unit, weights=np.where(light, 0.0, cost), minlength=int(unit.max()) + 1)
used = np.bincount(unit, minlength=int(unit.max()) + 1) > 0
print(f"{nu} SIMDs seen; heavy waves per SIMD: percentiles 0 10 50 90 100  {q(heavy_per[used])}")
print(f"radar-loop time of the heavy waves summed per SIMD [us]:            {q(cost_per[used])}")
# which SIMD a workgroup's wave w lands on (wave index within the workgroup = table wave % 4): a fixed assignment means that a
# scheme which makes "wave w of every workgroup" the heavy one piles all of them onto one SIMD in four
widx = np.arange(len(ok))[ok] % 4                  # the wave's place in its workgroup (a workgroup sweeps four consecutive table waves)
tab = np.zeros((4, 4), dtype=np.int64)
np.add.at(tab, (widx, simd), 1)
print("wave-in-workgroup x SIMD (rows: wave 0..3, columns: SIMD 0..3):", " | ".join(" ".join(str(int(v)) for v in row) for row in tab))
# ... and whether the first wave's SIMD is a property of the compute unit (a rotor that moves on by four per workgroup) or changes from
# workgroup to workgroup: per unit, the share of its workgroups that start on its commonest SIMD
cu_all = ((xcc * 8 + se) * 2 + sh) * 16 + cu
base = (simd - widx) % 4
share = []
for u in np.unique(cu_all):
    b = np.bincount(base[(cu_all == u) & (widx == 0)], minlength=4)
    if b.sum() >= 8:
        share.append(b.max() / b.sum())
print(f"per compute unit, share of workgroups whose wave 0 sits on the unit's commonest SIMD: percentiles {q(np.array(share))}")
same = 0; tot = 0
wv_ids = np.arange(len(ok))[ok]
for u in np.unique(cu_all)[:64]:
    sel_u = np.nonzero((cu_all == u) & (widx == 0))[0]
    sel_u = sel_u[np.argsort(start[sel_u], kind="stable")]
    bs = base[sel_u]
    same += int((bs[1:] == bs[:-1]).sum()); tot += max(0, len(bs) - 1)
print(f"consecutive workgroups of a unit (start order) with the same first SIMD: {same} of {tot}")
wg = np.arange(len(ok))[ok] // 4                   # (the row block)
# A finished wave's slot is not given out again before its whole workgroup has ended (tools/dispatch_rate_probe.hip): what the
# waves of a workgroup hold beyond their own end
wg_end = np.zeros(int(wg.max()) + 1); np.maximum.at(wg_end, wg, end)
held = wg_end[wg] - end
print(f"slot time held by finished waves until their workgroup ends: {float(held.sum()):.0f} us of {float((end - start).sum()):.0f} us lived "
      f"(light {float(held[light].sum()):.0f}, heavy {float(held[~light].sum()):.0f}); per wave percentiles {q(held)}")
hv_per_wg = np.bincount(wg[~light], minlength=int(wg.max()) + 1)
print("workgroups by number of heavy waves (0..4):", np.bincount(hv_per_wg, minlength=5)[:5].tolist(),
      "; heavy waves by place in the workgroup (0..3):", np.bincount(widx[~light], minlength=4).tolist())
first_wg = np.zeros(int(wg.max()) + 1); np.maximum.at(first_wg, wg, -start); first_wg = -first_wg
early = first_wg[wg] < 3.0
print(f"workgroups that start in the first 3 us: {int((first_wg < 3.0).sum())}; their heavy waves: {int((~light & early).sum())}")
# the placement of three compute units' workgroups, in start order: the SIMD of each of the four waves (heavy = *), and how many waves
# the unit's SIMDs held when the workgroup's first wave started
for u in np.unique(cu_all)[[3, 77, 200]]:
    on_u = cu_all == u
    wgs = np.unique(wg[on_u])
    wgs = wgs[np.argsort([start[on_u & (wg == g)].min() for g in wgs])]
    print(f"unit {int(u)}: workgroups in start order  [start us | SIMD of wave 0..3 | resident per SIMD before]")
    for g in wgs[:24]:
        sel_g = np.nonzero(on_u & (wg == g))[0]
        sel_g = sel_g[np.argsort(widx[sel_g])]
        t_g = start[sel_g].min()
        res = [int(((start < t_g) & (end > t_g) & on_u & (simd == k)).sum()) for k in range(4)]
        hv_res = [int(((start < t_g) & (end > t_g) & on_u & (simd == k) & ~light).sum()) for k in range(4)]
        print(f"  {t_g:6.2f} | " + " ".join(f"{int(simd[i])}{'*' if not light[i] else ' '}" for i in sel_g) + f" | {res} heavy {hv_res}")
first = start < 3.0
print(f"of the waves that start in the first 3 us ({int(first.sum())}): {100 * float((~light)[first].mean()):.0f} % heavy; "
      f"heavy waves per SIMD among them {q(np.bincount(unit[first & ~light], minlength=int(unit.max()) + 1)[used])}")

# the dispatcher's placement: the first workgroups of XCD 0 in start order (wave 0 of each workgroup), as (se, sh, cu)
w0 = widx == 0
sel = np.nonzero(w0 & (xcc == 0))[0]
sel = sel[np.argsort(start[sel], kind="stable")][:96]
print("XCD 0, workgroups in start order: se.sh.cu (heavy = *)")
print(" ".join(f"{int(se[i])}.{int(sh[i])}.{int(cu[i]):02d}{'*' if not light[i] else ' '}" for i in sel))
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
per_cu = np.bincount(cuid[w0], minlength=int(cuid.max()) + 1)
print("workgroups per CU over the launch: percentiles", q(per_cu[per_cu > 0]), " CUs used:", int((per_cu > 0).sum()))

# the same in units of work: radars walked (tick t) per SIMD, and when each SIMD's last heavy wave ended
walk_per = np.bincount(unit, weights=walked.astype(np.float64), minlength=int(unit.max()) + 1)
print(f"radars walked in tick t, summed per SIMD: mean {walk_per[used].mean():.1f}; percentiles {q(walk_per[used])}")
last_heavy = np.zeros(int(unit.max()) + 1)
np.maximum.at(last_heavy, unit[~light], end[~light])
print(f"end of the last heavy wave per SIMD [us]: {q(last_heavy[used])}")
r = np.corrcoef(walk_per[used], last_heavy[used])[0, 1]
print(f"correlation of a SIMD's walked sum with the end of its last heavy wave: {r:.2f}")

# what the heavy waves walk: radars that are candidates / see every row of the block (inside) / every row above their plane,
# per wave of tick t (probe slot 7), the radars whose walk got to the full classification (slot 5), detected rows (slot 4 is
# overwritten by the pair's second loop stamp)
inside_n, plane_n = (t[:, 7] >> 8) & 0xFF, (t[:, 7] >> 16) & 0xFF
deep, deep_all, deep_none = t[:, 5] & 0xFF, (t[:, 5] >> 8) & 0xFF, (t[:, 5] >> 16) & 0xFF   # (of the pair's SECOND tick: the slot is written twice)
hv = ~light
print(f"heavy waves ({int(hv.sum())}): candidates per wave {q(walked[hv])}; inside {q(inside_n[hv])}; plane {q(plane_n[hv])}; "
      f"to the full classification {q(deep[hv])}")
print(f"sums over the launch (tick t): candidates {int(walked.sum())}, inside {int(inside_n.sum())}, plane {int(plane_n.sum())}, "
      f"full classification (tick t + 1) {int(deep.sum())}, of which every live row seen {int(deep_all.sum())}, none {int(deep_none.sum())}")
