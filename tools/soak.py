#!/usr/bin/env python3
"""Long randomized parity run of the fused headless path against the CPU oracle (development aid):
    python tools/soak.py [n] [R] [m] [ticks] [seed]
Every tick: visibility masks, position bits and detonation events must match bit for bit."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests.test_gpu_engine import OracleMirror, _compare_tick, _device_noise_table
from zrk_modulation_amd import scenario as S
from zrk_modulation_amd.engine import HotPathEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
ticks = int(sys.argv[4]) if len(sys.argv) > 4 else 200
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 2024
ids, sp, vel, t0 = S.synthetic_targets(n, seed)
radars = S.synthetic_radars(R)
g = np.random.Generator(np.random.PCG64(seed + 1))
for k, rd in enumerate(radars):                      # varied radars: ranges, sectors, scan speeds, one vertical
    rd["max_distance"] = float(g.uniform(2e4, 6e4)); rd["azimuth_start"] = float(g.uniform(0, 360))
    rd["azimuth_range"] = float(g.uniform(20, 200)); rd["elevation_range"] = float(g.uniform(10, 90))
    rd["azimuth_speed"] = float(g.uniform(1, 30)); rd["elevation_speed"] = float(g.uniform(0, 5))
    rd["position"] = [float(v) for v in g.normal(0, 8e3, 3) * [1, 1, 0.05]]
    if k % 5 == 4:
        rd["scan_mode"] = "vertical"
eng = HotPathEngine(device="cuda:0", dt_ms=250, seed=seed, noise="philox")
eng.load(ids, sp, vel, t0, radars, missile_capacity=m).enable_lists()
launched = eng.launch_missiles(S.missile_targets(n, m), speed=2500.0, radius=500.0, period=45.0)
mir = OracleMirror(eng, radars)
t_start = time.time(); ev_total = 0; det_total = 0
for k in range(ticks):
    table = _device_noise_table(eng, k, R, mir.n)
    events = mir.tick(k * 250, 250, 2, table, threads=16)
    eng.run(1)
    vis, _ = _compare_tick(eng, mir, events, f"soak tick {k}")
    lists = eng.detections()
    for r, want in enumerate(mir.lists()):
        assert np.array_equal(lists[r], want), f"tick {k} radar {r}"
    ev_total += len(events); det_total += int(np.count_nonzero(vis))
    if k % 20 == 0:
        print(f"tick {k:4d} ok  events so far {ev_total}  detected now {np.count_nonzero(vis)}  {time.time()-t_start:.0f}s", flush=True)
print(f"SOAK OK: n={n} R={R} launched={launched} ticks={ticks} events={ev_total} detections={det_total}")
