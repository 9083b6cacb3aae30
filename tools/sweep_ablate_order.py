"""The pair sweep (events) at 10^6 rows with 4 / 16 radars under the loop's schedule switches: TAG=base | ZRK_SWEEP_ORDER=0 (table order) |
ZRK_DIAG=2 (no box records) | HIP_FORCE_DEV_KERNARG=0 (arguments in host memory).  Development aid; profiles/r05_sweep_ablate.txt."""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from sweep_ablate import run
for R, noise, m in [(0, "off", 0), (4, "philox", 0), (16, "philox", 10000)]:
    k, w = run(1_000_000, R, noise, True, m=m)
    print(f"{os.environ.get('TAG','')}: R={R} noise={noise} m={m}: sweep (pair, events) {k:.2f} us, tick wall {w:.2f}", flush=True)
