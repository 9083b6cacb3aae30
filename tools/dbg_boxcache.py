import sys, numpy as np
sys.path.insert(0, "/root/repo")
from tests.test_gpu_engine import OracleMirror
from zrk_modulation_amd import scenario as S
from zrk_modulation_amd.engine import HotPathEngine
n, R, m = 60_000, 2, 20_000
ids, sp, vel, t0 = S.synthetic_targets(n, 31)
sp[:, :2] *= 0.25
radars = S.synthetic_radars(R)
eng = HotPathEngine(device="cuda:0", dt_ms=500, seed=3, noise="off")
eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
launched = eng.launch_missiles(S.missile_targets(n, m), speed=2500.0, radius=800.0, period=40.0)
mir = OracleMirror(eng, radars)
for k in range(4):
    events = mir.tick(k * 500, 500, 0, None)
    eng.run(1)
    st = eng.store
    nn = st.n_uploaded
    vis = st.vis()[:nn].cpu().numpy().view(np.uint32)
    bad = np.nonzero(vis != mir.vis)[0]
    print("tick", k, "differing list slots:", len(bad), bad[:10], "dev", vis[bad[:10]], "oracle", mir.vis[bad[:10]])
    if len(bad):
        rows = eng.row_of_list[bad[:10]] if eng.row_of_list is not None else bad[:10]
        print("  rows", rows, "waves", rows // 64, "alive", st.d_alive[:nn].cpu().numpy()[rows], "kind", st.h_kind[rows])
        P = st.host_pos("cur")[rows]
        print("  pos", P)
        break

# dump the box records of the first failing waves
import torch
cap = st.cap
comp_blocks = (cap + 1023) // 1024 + 1
order_ints = ((cap + 255) // 256 + 64) & ~63
off = 327936 + 4 * (64 + ((2 * 33 * comp_blocks + 63) & ~63)) + 4 * 2 * order_ints
ws = st.workspace().cpu().numpy()
rec = ws[off:off + 48 * ((cap + 63) // 64)].view(np.float32).reshape(-1, 12)
recu = rec.view(np.uint32)
allpos = st.host_pos("cur")
alive = st.d_alive[:nn].cpu().numpy()
for w in (rows // 64)[:4]:
    r = rec[w]
    tref = recu[w, 10:12].copy().view(np.float64)[0]
    sel = np.arange(w * 64, min(w * 64 + 64, nn))
    sel = sel[alive[sel] != 0]
    p = allpos[sel]
    print("wave", w, "rec lo", r[0:3], "hi", r[3:6], "vmax", r[6:9], "state", recu[w, 9], "t_ref", tref)
    print("      true lo", p.min(0), "hi", p.max(0), "vel max", np.abs(st.h_vel[w * 64:w * 64 + 64]).max(0))
