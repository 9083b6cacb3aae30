#!/usr/bin/env python3
"""Sweep-kernel time (HIP events, every tick) for a few (R, noise, lists, missiles) combinations at n rows."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
from sweep_ablate import run

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for noise, R, lists, m in [("off", 0, False, 0), ("off", 16, False, 0), ("philox", 16, False, 0), ("off", 16, True, 0),
                           ("philox", 16, True, 0), ("philox", 16, True, 10000), ("philox", 4, True, 0), ("philox", 32, True, 0)]:
    k, w = run(n, R, noise, lists, m=m)
    print(f"n={n} noise={noise:6s} R={R:2d} lists={int(lists)} m={m:5d}  sweep={k:7.2f} us   tick wall={w:7.2f} us", flush=True)
