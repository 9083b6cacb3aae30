#!/usr/bin/env python3
"""Host time per tick of the loop's three ways of being driven, at the C3 shard size: K ticks per C call (plain),
K ticks per C call with the exchange issued from the C side (one-rank communicator: every call and event of the
N > 1 path but the wire), and tick by tick from Python with torch.distributed (what round 1 did).
Writes profiles/r02_exchange_host_time.json when run with --record."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from zrk_modulation_amd import scenario as S  # noqa: E402
from zrk_modulation_amd.engine import HotPathEngine  # noqa: E402
from zrk_modulation_amd.exchange import DetectionExchange, RcclExchange, union_bits_words  # noqa: E402

n, R, m = 1_000_000, 16, 10_000


def engine():
    ids, sp, vel, t0 = S.synthetic_targets(n, 1237)
    eng = HotPathEngine(device="cuda:0", dt_ms=10, seed=1237, noise="philox")
    eng.load(ids, sp, vel, t0, S.synthetic_radars(R), missile_capacity=m)
    eng.launch_missiles(S.missile_targets(n, m))
    return eng


def timed(fn, ticks, reps=10):
    host = wall = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter(); fn(); host += time.perf_counter() - t
        torch.cuda.synchronize()
        wall += time.perf_counter() - t
    return host / (ticks * reps) * 1e6, wall / (ticks * reps) * 1e6


res = {}
K = 100            # short bursts: the launch queue never fills, so `host` is the host's own cost
eng = engine(); eng.enable_lists(); eng.run(100)
res["plain_loop"] = dict(zip(("host_us_per_tick", "wall_us_per_tick"), timed(lambda: eng.run(K), K)))
eng = engine()
words = union_bits_words(eng.store.cap, R, 300_000)
x = RcclExchange(words, eng.store.device, R, offsets=[0], ev_capacity=m)
eng.run(100, exchange=x)
res["c_side_exchange_one_rank"] = dict(zip(("host_us_per_tick", "wall_us_per_tick"), timed(lambda: eng.run(K, exchange=x), K)))
x.sync(); x.close()
eng = engine()
ex = [DetectionExchange(words, eng.store.device, fmt="bits", offsets=[0], R=R) for _ in range(2)]
buf = [torch.zeros(words, dtype=torch.int64, device=eng.store.device) for _ in range(2)]
eng.loop.flags |= 8


def py_loop(k):
    for j in range(k):
        eng.packed = buf[j & 1]
        eng.run(1)
        ex[j & 1].all_gather(eng.packed, async_op=True)


py_loop(100)
res["python_tick_by_tick_one_rank"] = dict(zip(("host_us_per_tick", "wall_us_per_tick"), timed(lambda: py_loop(K), K)))
res["note"] = ("C3 shard (1e6 AirObjects, 16 radars, up to 10 000 missiles), one MI355X; host = time the enqueueing call takes, "
               "wall = until the device is idle; the exchange is a one-rank communicator (RCCL runs, nothing crosses xGMI)")
print(json.dumps(res, indent=1))
if "--record" in sys.argv:
    out = ROOT / "gpurun_out" / "r02_exchange_host_time.json"
    out.parent.mkdir(exist_ok=True)
    out.write_text(json.dumps(res, indent=1) + "\n")
