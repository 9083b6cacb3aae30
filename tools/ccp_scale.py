#!/usr/bin/env python3
"""The command post's step on the device at scale: 10^6 tracks, 10^5 detections a tick, the candidate pass as tiled all pairs
and through the spatial index (ZRK_CCP_GRID).  Prints the time of a step in either mode and checks that they agree.

    python tools/ccp_scale.py [tracks] [detections]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_ccp_step import Table, _ctx                     # noqa: E402
from zrk_modulation_amd.association import DeviceCommandPost        # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    D = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000
    g = np.random.Generator(np.random.PCG64(5))
    p0 = g.uniform(-3e5, 3e5, (n, 3)) * [1, 1, 0.03]
    vel = g.normal(0, 250, (n, 3)) * [1, 1, 0.2]
    speed = np.linalg.norm(vel, axis=1)
    lpos = g.uniform(-2e4, 2e4, (8, 3)) * [1, 1, 0]
    caps = np.full(8, 5000, np.int32)
    tab = Table(n)
    posts = {m: DeviceCommandPost(_ctx(), "cuda:0", n, n, lpos, caps, dmax=n, rounds=int(os.environ.get("ZRK_CCP_ROUNDS", "8"))) for m in ("1", "0")}
    dt, slack = 0.5, 2.0
    pos = p0.copy()
    for k in range(4):
        now = k * dt
        prev = pos.copy()
        pos = p0 + vel * now + g.normal(0, 5, (n, 3))
        none = np.zeros(n, bool) if k else np.ones(n, bool)
        seq = (np.arange(n) if k == 0 else g.permutation(n)[:D]).astype(np.int32)
        tab.set_tick(pos, prev, none, speed, now)
        seq_d = torch.from_numpy(seq).cuda()
        cnt = torch.tensor([len(seq)], dtype=torch.int32, device="cuda:0")
        got = {}
        for mode, post in posts.items():
            os.environ["ZRK_CCP_GRID"] = mode
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            post.step(tab.ents, 0, tab.speed, seq_d, cnt, now, slack)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            got[mode] = post.results()
            print(f"tick {k}: {len(seq):>8} detections x {int(post.counts.sum()) if k else 0:>8} tracks after, "
                  f"{'spatial index' if mode == '1' else 'all pairs    '} {ms:9.2f} ms", flush=True)
        for a, b in zip(got["0"], got["1"]):
            assert np.array_equal(a, b)
    print("both passes agree")


if __name__ == "__main__":
    main()
