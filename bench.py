#!/usr/bin/env python3
"""Benchmark of the hot path: entity-timesteps/sec of the L1 tick loop on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one simulation tick over one batch of synthetic input: apply last
tick's detonations, step every in-flight missile, advance every live air object, sweep every radar
over them with measurement noise (Philox mode), compact the detection lists, advance the scan.
Inputs are resident in HBM when the timed region starts.  Default workload: BASELINE.json configs[2]
(1e6 targets, 16 radars, 1e4 missiles per GPU), the configuration north_star quotes the HBM-roofline
target on; `--workload C2` runs configs[1].  With N > 1 every rank owns a contiguous shard of the
population (weak scaling: the per-GPU shard is the single-GPU workload) and each tick ends with an
RCCL all-gather of the packed detection list.

Prints ONE JSON line (rank 0).  `roofline` is the fused advance+sweep kernel: algorithmic bytes
per launch (85 B per live entity, 1 B per tombstone) over its average duration, measured with HIP
events on the launch stream inside the timed region.  `cpu_baseline` is the oracle (C restatement of
the reference, oracle/) timed on this host on a bounded number of ticks of the same scene.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
PROF_STRIDE = 8                # sweep kernel timed with HIP events every 8th tick (one call covers all ticks) ...
PROF_STRIDE_RANKS = 64         # ... every 64th when ticks are driven one call at a time: reading the events drains the stream


def build_engine(workload, rank, world, device, seed_off=0):
    from zrk_modulation_amd.engine import HotPathEngine
    from zrk_modulation_amd import scenario as S
    n, R, m = S.WORKLOADS[workload]
    ids, sp, vel, t0 = S.synthetic_targets(n, S.SEEDS[workload] + 1000 * rank + seed_off, first_id=1000 + rank * n)
    radars = S.synthetic_radars(R)
    stride = n + m                                   # global index space: shard g starts at g * stride
    eng = HotPathEngine(device=device, dt_ms=10, seed=S.SEEDS[workload], noise="philox", gid0=rank * stride)
    eng.load(ids, sp, vel, t0, radars, missile_capacity=m, union_capacity=(n + m) if world > 1 else None,
             union_format="bits")
    # (the exchange buffers are re-sized to the observed detection count after warm-up, see main)
    if world == 1:
        eng.enable_lists()
    launched = eng.launch_missiles(S.missile_targets(n, m))
    return eng, dict(n=n, R=R, m=m, launched=launched, scene=(ids, sp, vel, t0, radars))


def pmc_traffic(workload, world):
    """HBM bytes per sweep launch from the committed PMC passes (profiles/r01_pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc runs of this workload, corrected as
    MI355X_MICROARCH.md prescribes and calibrated on a pure-streaming launch).  Counters cannot be read
    from inside this process, so the figure is the recorded one for the same workload, else null."""
    try:
        rec = json.load(open(ROOT / "profiles" / "r01_pmc_traffic.json"))
        if rec["workload"] == workload and world == 1:
            return rec["traffic_bytes"]
    except Exception:
        pass
    return None


def profiled_kernel_us(workload, world):
    """Average duration of the sweep kernel in the committed rocprofv3 --kernel-trace --stats summary of this
    command (profiles/r01_g_final_kernel_stats.csv), for cross-reference with the live HIP-event figure: the
    event pair also brackets the dispatch latency on both sides of the kernel (about 3 us).  Recorded, not live."""
    try:
        if workload != "C3" or world != 1:
            return None
        import csv
        for row in csv.DictReader(open(ROOT / "profiles" / "r01_g_final_kernel_stats.csv")):
            if "k_tick_sweep" in row["Name"]:
                return float(row["AverageNs"]) / 1e3
    except Exception:
        pass
    return None


def usable_cores():
    """Threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(info, eng, budget_s=15.0):
    """Time the oracle's L1 tick on the same scene (test infrastructure used as the reported CPU
    baseline only).  Single-threaded advance/missile loop + OpenMP radar phase on all host cores."""
    import ctypes as C
    from oracle import oracle as O
    L = O.lib()
    ids, sp, vel, t0, radars = info["scene"]
    st = eng.store
    n_t = info["n"]
    # rebuild the initial table (targets + the launched missiles) on the host
    n = st.n_uploaded
    cap = n
    hsp = np.ascontiguousarray(st.h_sp[:n].T).reshape(-1); hvel = np.ascontiguousarray(st.h_vel[:n].T).reshape(-1)
    ht0 = st.h_t0[:n].copy(); pos = np.ascontiguousarray(st.h_pos0[:n].T).reshape(-1).copy()
    prev = pos.copy(); pv = np.zeros(n, np.uint8); alive = np.ones(n, np.uint8)
    kind = st.h_kind[:n].copy(); mrow = np.full(n, -1, np.int32)
    m = st.m
    mrow[st.hm_slot[:m]] = np.arange(m, dtype=np.int32)
    m_tgt = st.hm_tgt[:m].copy(); m_radius = np.full(m, 150.0); m_period = np.full(m, 60.0)
    m_status = np.ones(max(m, 1), np.uint8)
    evm = np.zeros(max(m, 1), np.int32); evt = np.zeros(max(m, 1), np.int32); evs = np.zeros(max(m, 1), np.uint8)
    vis = np.zeros(n, np.uint32)
    from zrk_modulation_amd.engine import scan_next, scan_mode_code
    rs = [dict(r, caz=r["azimuth_start"], cel=r["elevation_start"]) for r in radars]
    cores = usable_cores()

    def tick(k, threads):
        nev = L.zo_airenv_step(n, cap, 10 * k, 10, O.dptr(hsp), O.dptr(hvel), O.dptr(ht0), O.u8ptr(alive),
                               O.u8ptr(kind), O.i32ptr(mrow), O.dptr(pos), O.dptr(prev), O.u8ptr(pv),
                               O.i32ptr(m_tgt), O.dptr(m_radius), O.dptr(m_period), O.u8ptr(m_status),
                               O.i32ptr(evm), O.i32ptr(evt), O.u8ptr(evs))
        arr = O.radar_array([(r["position"][0], r["position"][1], r["position"][2], r["max_distance"], r["caz"],
                              r["azimuth_range"], r["cel"], r["elevation_range"]) for r in rs])
        L.zo_radar_phase_fused(n, cap, O.dptr(pos), O.u8ptr(alive), len(rs), arr, 1, None, 1237, k, 0,
                               O.u32ptr(vis), threads)
        out = np.zeros(n, np.int32)
        for r in range(len(rs)):
            L.zo_compact_bit(n, O.u32ptr(vis), r, 0, O.i32ptr(out))
        for r in rs:
            r["caz"], r["cel"] = scan_next(scan_mode_code(r["scan_mode"]), r["azimuth_range"], r["azimuth_speed"],
                                           r["elevation_speed"], r["elevation_start"], r["caz"], r["cel"])
        for j in range(nev):            # tombstones take effect on the next tick (AirEnv.py:33-40)
            alive[evm[j]] = 0
            if evt[j] >= 0:
                alive[evt[j]] = 0

    t_a = time.perf_counter(); tick(0, cores); t_one = time.perf_counter() - t_a
    ticks = int(max(2, min(200, budget_s / max(t_one, 1e-6))))
    t_a = time.perf_counter()
    for k in range(1, 1 + ticks):
        tick(k, cores)
    t_all = time.perf_counter() - t_a
    live = int(alive.sum())
    t_b = time.perf_counter(); tick(1 + ticks, 1); t_1core = time.perf_counter() - t_b
    return {"value": live * ticks / t_all, "unit": "entity-timesteps/s", "cores": cores, "kind": "port",
            "sample": f"same scene, {ticks} ticks, oracle/zrk_oracle.c: 1-thread AirEnv+missile loop, "
                      f"OpenMP radar phase on {cores} threads",
            "value_1core": live / t_1core,
            "reference_python_note": "reference itself (pure Python) measured in the survey container: "
                                     "1.6 us/entity + 4.3 us/(radar x entity), 1 thread (BASELINE.md section 2)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ZRK_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks
    backend = os.environ.get("ZRK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    eng, info = build_engine(args.workload, rank, world, device)
    # N > 1: two packed buffers / exchanges in flight, so the all-gather of tick t (RCCL's own stream)
    # overlaps the sweep of tick t+1; a buffer is reused only after its collective has been waited for
    xchg = {"ex": [], "buf": [], "work": [None, None], "tick": 0}

    def size_exchange(entries):
        """Buffers for up to `entries` seen objects per rank, in the wire format of zrk_compact_bits (count, n, one
        bit per slot, 16-bit masks): a quarter of the bytes of (index, mask) pairs."""
        from zrk_modulation_amd.exchange import DetectionExchange, union_bits_words
        n_slots = torch.tensor([int(eng.loop.n)], dtype=torch.int64, device=device)
        dist.all_reduce(n_slots, op=dist.ReduceOp.MAX)                 # launch counts differ a little between ranks
        words = union_bits_words(int(n_slots.item()), info["R"], entries)
        stride = info["n"] + info["m"]
        xchg["ex"] = [DetectionExchange(words, device, fmt="bits", offsets=[g * stride for g in range(world)], R=info["R"])
                      for _ in range(2)]
        xchg["buf"] = [torch.zeros(words, dtype=torch.int64, device=device) for _ in range(2)]
        xchg["work"] = [None, None]
        xchg["entries"], xchg["words"] = int(entries), int(words)

    def drain_exchange():
        for k, w in enumerate(xchg["work"]):
            if w is not None:
                w.wait()
                xchg["work"][k] = None

    stride = PROF_STRIDE if world == 1 else PROF_STRIDE_RANKS

    def run_ticks(k, sweep_ms=None):
        if world == 1:
            eng.run(k, sweep_ms=sweep_ms, prof_stride=PROF_STRIDE)
            return
        for j in range(k):
            b = xchg["tick"] & 1
            if xchg["work"][b] is not None:
                xchg["work"][b].wait()
            eng.packed = xchg["buf"][b]
            one = np.zeros(1, np.float32) if (sweep_ms is not None and j % stride == 0) else None
            eng.run(1, sweep_ms=one, prof_stride=1)
            xchg["work"][b] = xchg["ex"][b].all_gather(eng.packed, async_op=True)
            xchg["tick"] += 1
            if one is not None:
                sweep_ms[j // stride] = one[0]

    if world > 1:
        size_exchange(info["n"] + info["m"])

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    run_ticks(args.warmup)
    overflow = False
    if world > 1:
        # size the fixed exchange buffers from what the warm-up saw (1.25x the largest per-rank count: the count
        # follows the sectors round their scan period, which the warm-up covers; an overflow is reported)
        drain_exchange()
        seen = torch.tensor([max(max(e.counts()) for e in xchg["ex"])], dtype=torch.int64, device=device)
        dist.all_reduce(seen, op=dist.ReduceOp.MAX)
        size_exchange(int(seen.item() * 1.25) + 1024)
    barrier()
    live0 = eng.alive_count()
    sweep_ms = np.zeros((args.steps + stride - 1) // stride, np.float32)
    barrier()
    t0 = time.perf_counter()
    run_ticks(args.steps, sweep_ms)
    if world > 1:
        drain_exchange()
    barrier()
    elapsed = time.perf_counter() - t0
    live1 = eng.alive_count()
    if world > 1:
        overflow = any(e.overflowed() for e in xchg["ex"])

    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    units = torch.tensor([float(min(live0, live1)) * args.steps], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    elapsed = float(el.item()); total_units = float(units.item())

    if rank == 0:
        n_slots = eng.store.n_uploaded
        live_avg = 0.5 * (live0 + live1)
        alg_bytes = 85.0 * live_avg + 1.0 * (n_slots - live_avg)
        good = sweep_ms[sweep_ms > 0]
        sweep_avg_ms = float(good.mean()) if len(good) else float("nan")
        achieved = alg_bytes / (sweep_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "entity-timesteps/sec (targets+missiles)", "value": total_units / elapsed,
            "unit": "entity-timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {info['n']} AirObjects, {info['R']} SectorRadars, "
                                   f"{info['launched']}/{info['m']} missiles in flight per GPU, dt=10 ms, "
                                   f"Philox measurement noise, "
                                   + ("union compaction in the bitmap wire format + per-tick RCCL all-gather of the detection list, "
                                      "overlapped with the next sweep" if world > 1 else "per-radar compaction"),
                       "entities_per_gpu": n_slots, "live_per_gpu": int(live1), "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.workload, world),
                         "kernel": "k_tick_sweep",
                         "avg_kernel_us": sweep_avg_ms * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
                         "profiled_kernel_us": profiled_kernel_us(args.workload, world)},
        }
        if world > 1:
            out["config"]["exchange_entries_per_rank"] = xchg["entries"]
            out["config"]["exchange_bytes_per_rank"] = 8 * xchg["words"]
            out["config"]["exchange_overflow"] = bool(overflow)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(info, eng, args.cpu_budget)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
