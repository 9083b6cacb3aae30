#!/usr/bin/env python3
"""The bench's C3 tick loop (lists + missiles + Philox noise) for profiling:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/prof_run.py [ticks] [n] [R] [m]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zrk_modulation_amd import _lib
if os.environ.get("ZRK_LIB"):               # a differently built library (A/B of a build-time switch)
    _lib.LIB_PATH = _lib.CSRC / os.environ["ZRK_LIB"]
from zrk_modulation_amd import scenario as S
from zrk_modulation_amd.engine import HotPathEngine
import torch

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("PROF_TICKS", 300))
n = int(sys.argv[2]) if len(sys.argv) > 2 else int(os.environ.get("PROF_N", 1_000_000))
R = int(sys.argv[3]) if len(sys.argv) > 3 else int(os.environ.get("PROF_R", 16))
m = int(sys.argv[4]) if len(sys.argv) > 4 else int(os.environ.get("PROF_M", 10_000))
ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1237, noise="philox")
radars = S.synthetic_radars(R)
if os.environ.get("PROF_RANGE"):          # e.g. 1: nobody is ever in range, every wave takes the light path
    for r in radars:
        r["max_distance"] = float(os.environ["PROF_RANGE"])
eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
if m:
    tg = S.missile_targets(n, m)
    if os.environ.get("PROF_TGT_NEAR"):     # all missiles against a few neighbouring rows: their gathers share cache lines
        tg = (tg % 64).astype(tg.dtype)
    eng.launch_missiles(tg)
eng.run(ticks)
torch.cuda.synchronize()
eng.store.compact_status()
