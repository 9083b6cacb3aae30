import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
for wl in ("C3", "C2", "C3x4"):
    info = bench.WORKLOADS[wl] if hasattr(bench, "WORKLOADS") else None
    eng = bench.build_engine(wl, 0, 1, "cuda:0")
    eng = eng[0] if isinstance(eng, tuple) else eng
    eng.run(20)
    d = eng.detections()
    tot = sum(len(x) for x in d)
    vis = eng.store.vis()[:eng.store.n_uploaded]
    print(wl, "rows", eng.store.n_uploaded, "detections per radar", [len(x) for x in d][:32], "sum", tot, "rows seen by any radar", int((vis != 0).sum()))
