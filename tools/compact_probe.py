"""Time zrk_compact alone on masks of the density the C3 sweep produces.  usage: compact_probe.py [n] [R] [density]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zrk_modulation_amd.store import EntityStore  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
density = float(sys.argv[3]) if len(sys.argv) > 3 else 0.17
g = np.random.Generator(np.random.PCG64(1))
vis = (g.integers(0, 1 << R, n, dtype=np.uint64).astype(np.uint32) & g.integers(0, 1 << R, n, dtype=np.uint64).astype(np.uint32)
       & g.integers(0, 1 << R, n, dtype=np.uint64).astype(np.uint32))
blocks = np.repeat(g.uniform(size=(n + 4095) // 4096) < density * 2.2, 4096)[:n]      # clustered, as sorted storage gives
vis[~blocks] = 0
print("nonzero fraction", (vis != 0).mean(), "bits", np.unpackbits(vis.view(np.uint8)).sum())
for label, env in [("3-launch", {"ZRK_COMPACT_FUSED_MAX_BLOCKS": "0"})] + [
        (f"fused items={i} order={o}", {"ZRK_COMPACT_ITEMS": str(i), "ZRK_COMPACT_ORDER": o})
        for o in ("ticket", "block") for i in (1, 2, 4, 8)]:
    for k in ("ZRK_COMPACT_FUSED_MAX_BLOCKS", "ZRK_COMPACT_ITEMS", "ZRK_COMPACT_ORDER"):
        os.environ.pop(k, None)
    os.environ.update(env)
    st = EntityStore("cuda:0", capacity=n)
    dvis = torch.as_tensor(vis.view(np.int32), device=st.device)
    det = torch.zeros(n * R, dtype=torch.int32, device=st.device)
    cnt = torch.zeros(R + 1, dtype=torch.int32, device=st.device)
    packed = torch.zeros(n + 1, dtype=torch.int64, device=st.device)
    zero = torch.zeros(n, dtype=torch.int32, device=st.device)

    def go(reps):
        for _ in range(reps):
            st.ctx.check(st.lib.zrk_compact(st.ctx.handle, dvis.data_ptr(), n, R, 0, st.workspace().data_ptr(), det.data_ptr(), n,
                                            cnt.data_ptr(), packed.data_ptr(), n + 1, 0, None), "compact")
    go(20)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go(500)
    e1.record()
    torch.cuda.synchronize()
    st.compact_status()
    print(f"{label:32s} {e0.elapsed_time(e1) / 500 * 1000:7.2f} us per compaction")
