#!/usr/bin/env python3
"""What torch.cuda.synchronize() costs behind a 20-tick call: on an idle device (everything long over), and right behind the
call (the device's remaining work + the wake-up), with the library's streams in existence."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
eng = bench.build_engine("C3", 0, 1, "cuda:0")
eng = eng[0] if isinstance(eng, tuple) else eng
eng.run(40)
torch.cuda.synchronize()
idle = []
for _ in range(50):
    t0 = time.perf_counter(); torch.cuda.synchronize(); idle.append((time.perf_counter() - t0) * 1e6)
print("idle device: synchronize %.1f us median, %.1f min" % (np.median(idle), np.min(idle)))
s = torch.cuda.current_stream()
for mode in ("device", "stream", "event-spin"):
    tot, call, sync = [], [], []
    for _ in range(15):
        torch.cuda.synchronize(); time.sleep(0.002)
        t0 = time.perf_counter()
        eng.run(20)
        t1 = time.perf_counter()
        if mode == "device":
            torch.cuda.synchronize()
        elif mode == "stream":
            s.synchronize()
        else:
            ev = torch.cuda.Event(); ev.record(s)
            while not ev.query():
                pass
        t2 = time.perf_counter()
        tot.append((t2 - t0) * 1e6); call.append((t1 - t0) * 1e6); sync.append((t2 - t1) * 1e6)
    print("%-10s total %.1f us (min %.1f)  call %.1f  wait %.1f" % (mode, np.median(tot), np.min(tot), np.median(call), np.median(sync)))
