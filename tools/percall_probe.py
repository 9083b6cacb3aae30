#!/usr/bin/env python3
"""Host overhead of driving the loop tick by tick (the multi-GPU bench does) vs K ticks per call."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import build_engine
for wl in ("C3", "C2"):
    eng, info = build_engine(wl, 0, 1, torch.device("cuda", 0))
    eng.run(50); torch.cuda.synchronize()
    t = time.perf_counter(); eng.run(500); torch.cuda.synchronize(); a = (time.perf_counter() - t) / 500 * 1e6
    t = time.perf_counter()
    for _ in range(500):
        eng.run(1)
    torch.cuda.synchronize(); b = (time.perf_counter() - t) / 500 * 1e6
    # union-only compaction, as the multi-GPU path uses
    eng2, _ = build_engine(wl, 0, 2, torch.device("cuda", 0))
    eng2.run(50); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(500):
        eng2.run(1)
    torch.cuda.synchronize(); c = (time.perf_counter() - t) / 500 * 1e6
    print(f"{wl}: run(500) {a:.1f} us/tick   500 x run(1) {b:.1f} us/tick   500 x run(1) union-only {c:.1f} us/tick")
