#!/usr/bin/env python3
"""Run a few ticks of the C3 scene for counter collection:
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out -- python3 tools/pmc_run.py [noise] [R] [n] [sort]
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from zrk_modulation_amd import scenario as S
from zrk_modulation_amd.engine import HotPathEngine
import torch

noise = sys.argv[1] if len(sys.argv) > 1 else "philox"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
sort = (sys.argv[4] != "nosort") if len(sys.argv) > 4 else True
ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1, noise=noise)
eng.load(ids, sp, vel, t0, S.synthetic_radars(R), missile_capacity=0, sort=sort).enable_lists()
eng.run(20)
torch.cuda.synchronize()
