#!/usr/bin/env python3
"""Sweep-kernel time against population size (fixed overhead vs slope)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zrk_modulation_amd import scenario as S
from zrk_modulation_amd.engine import HotPathEngine

for R, noise in ((1, "off"), (16, "off"), (16, "philox")):
    for n in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000):
        ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
        eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise=noise)
        eng.load(ids, sp, vel, t0, S.synthetic_radars(R)).enable_lists()
        eng.run(20)
        ms = np.zeros(100, np.float32)
        eng.run(100, sweep_ms=ms, prof_stride=1)
        us = ms.mean() * 1e3
        print(f"R={R:2d} noise={noise:6s} n={n:8d} sweep={us:7.1f} us  {85.0*n/us/1e6:6.2f} TB/s  {us/n*1e3:6.2f} ns/kentity", flush=True)
        del eng
