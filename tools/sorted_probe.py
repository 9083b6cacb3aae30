#!/usr/bin/env python3
"""Probe: how much does spatial coherence of the slot order buy the sweep kernel?"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zrk_modulation_amd import scenario as S
from zrk_modulation_amd.engine import HotPathEngine
import torch


def morton(sp, bits=10, use_z=True):
    lo = sp.min(0); hi = sp.max(0)
    q = ((sp - lo) / (hi - lo + 1e-9) * ((1 << bits) - 1)).astype(np.uint64)
    def spread(v):
        out = np.zeros_like(v)
        for b in range(bits):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out
    code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))
    if use_z:
        code |= spread(q[:, 2]) << np.uint64(2)
    return np.argsort(code, kind="stable")


def run(order_name, n=1_000_000, R=16, ticks=200):
    ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
    if order_name == "morton3d":
        o = morton(sp)
    elif order_name == "morton2d":
        o = morton(sp, use_z=False)
    elif order_name == "azimuth":
        o = np.argsort(np.arctan2(sp[:, 1], sp[:, 0]), kind="stable")
    else:
        o = np.arange(n)
    ids, sp, vel = ids[o], sp[o], vel[o]
    for noise in ("off", "philox"):
        eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise=noise)
        eng.load(ids, sp, vel, t0, S.synthetic_radars(R), missile_capacity=0)
        eng.run(20)
        ms = np.zeros(ticks, np.float32)
        eng.run(ticks, sweep_ms=ms, prof_stride=1)
        print(f"{order_name:10s} noise={noise:6s} sweep={ms.mean()*1e3:6.1f} us", flush=True)


for name in ("random", "azimuth", "morton2d", "morton3d"):
    run(name)
