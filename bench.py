#!/usr/bin/env python3
"""Benchmark of the hot path: entity-timesteps/sec of the L1 tick loop on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3|C2|C4|C5|tiny|tiny4|tiny5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one simulation tick over one batch of synthetic input: apply last tick's
detonations, step every in-flight missile, advance every live air object, sweep every radar over them with
measurement noise (Philox mode), compact the detection lists, advance the scan.  Inputs are resident in HBM
when the timed region starts.

Workloads (BASELINE.json `configs`):
  C3 (default)  configs[2]: 1e6 AirObjects, 16 radars, 1e4 missiles in flight per GPU -- the configuration
                north_star quotes the HBM-roofline target on.  With N > 1 every rank holds such a shard (weak scaling).
  C2            configs[1]: 1e5 / 4 / 1e3 (launch-bound; weak scaling).
  C3x4          C3's scene with 4e6 AirObjects: past the 256 MiB Infinity Cache, for the HBM reading of the roofline.
  C4            configs[3]: ONE seeded population of 1e7, rank g owns [g*1e7/N, (g+1)*1e7/N) (strong scaling).
  C5            configs[4]: Monte-Carlo ensemble, 128 independent scenarios x 1e4 targets per GPU in one batched table
                (no exchange; weak scaling).
  C2-battery    configs[1]'s table with the battery's CLOSED loop around it (zrk_modulation_amd.battery.DeviceBattery): four
                launchers with 250 missiles each and the command post on the device -- every tick: sweep + missile step +
                lists, the launchers' step (magazines, launch solves), zrk_ccp_step over the tick's detections, the next
                salvo's requests -- with the reference's latencies and nothing read back inside the loop (one GPU).
  tiny, tiny4, tiny5, tiny-battery: the same mechanics at test size.
With N > 1 (C2 / C3 / C4) each tick ends with an RCCL all-gather of the rank's compacted detection list (bitmap wire
format) and its detonation events, issued from the C side (zrk_run_ticks_x) on RCCL's own stream so that it
overlaps the next tick's sweep.  ZRK_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs
than ranks (the exchange then goes through torch.distributed from Python, tick by tick).

`--gpus N` without a torch.distributed environment starts the N ranks itself (children of this process, spawned
before anything here touches the GPU) and relays rank 0's line; under torch.distributed.run it must equal WORLD_SIZE.

The timed steps go through ONE zrk_run_ticks call, which (for calls of four ticks or more, ZRK_OVERLAP=0 turns it off)
compacts tick t's lists on a side stream beside tick t+1's sweep (`config.loop`); the kernels and their results are
the two-launch loop's (tests/test_gpu_overlap.py).

Prints ONE JSON line (rank 0).  `roofline` is the fused advance+sweep kernel.  Its duration is measured live, inside the
timed region, by the kernel itself: with zrk_sweep_stamps on, every sweep launch of the timed call (every k-th in runs of
more than 64 launches) writes the wall clock (s_memrealtime, 100 MHz) as its first waves start and as each of its waves
ends (zrk_read_sweep_stamps, read after the region).  That puts no event, signal or barrier on the stream: the timed
launches run exactly as untimed ones do, and ALL of them are samples (per launch they spread from 28 to 44 us at C3,
depending on what the compaction beside them is doing; the two event pairs of round 3's 20-step line happened to sit on
slow ones, and cost the run 2-3 us per tick).  A launch's time on its stream as a profiler stamps it also holds the
dispatch in front of its first wave and the release behind its last; that part (`dispatch_overhead_us`, 1.5-2 us) is
measured behind the region on stand-alone launches timed both ways -- stamps and an event pair riding on the dispatch --
and added: `avg_kernel_us` = first wave in to last wave out + dispatch overhead, which is what rocprofv3 reports for
the same command (profiles/).  `achieved` = the bytes the launch MUST move -- a launch that sweeps two ticks reads the 57 B
of trajectory columns once and writes 28 B per tick: 113 B per live entity, 1 B per tombstone -- over that duration;
`effective` puts SURVEY section 8d's per-tick figure (85 B per live entity and tick swept, i.e. 170 B for a two-tick
launch) over the same duration, for comparison with one-tick-per-launch figures of earlier rounds.  In the overlapped
loop the sweep shares the device with the previous launch's compaction; the rocprofv3 figures of the same command are
kept under profiles/ and echoed as `profiled_kernel_us_recorded`.  `traffic_recorded` is the PMC figure of the committed
counter passes -- counters cannot be read from inside this process.  `cpu_baseline` is the oracle (C restatement of the
reference, oracle/) timed on this host on a bounded number of ticks of the same scene.  Before the warm-up steps the
device runs ~100 ms of an unrelated self-test kernel so that a 20-step run is not measured while the clocks are still
ramping (`setup.clock_spinup_ms`; none of it counts as a step).
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBS = 6300.0    # what a pure streaming kernel reaches on this part (the same guide, HBM section)
SPINUP_MS = float(os.environ.get("ZRK_BENCH_SPINUP_MS", "100"))   # (0: none -- what the spin-up is worth: profiles/r04_bench_c3_driver_no_spinup.json)
PROFILE_TAG = "r05"            # profiles/<tag>_* hold the recorded figures echoed in the line


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C3x4", "C4", "C5", "tiny", "tiny4", "tiny5", "C2-battery", "tiny-battery"])
    ap.add_argument("--wire", default="masks", choices=["union", "masks"],
                    help="N > 1: what a rank's per-tick list carries -- the bitmap of the slots seen by any radar and the radar masks of "
                         "the seen slots (default: what a replicated command post needs, modules/CCP.py:409-417 keeps radar_id per "
                         "message), or the bitmap alone (a fifth of the bytes).  A default multi-rank run times BOTH: the line's "
                         "value is the `masks` figure, the bitmap-only figure stands beside it as `union_wire`")
    ap.add_argument("--interest", default="",
                    help="N > 1, C-side exchange: comma-separated radar indices the consumers on other ranks listen to (the reference's "
                         "command post reads one FoundObjectsMessage per radar of its radar_ids): the exchanged list is then the union "
                         "list of these radars alone.  Default: all radars")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4", action="store_true",
                    help="skip the strong-scaling sub-record (configs[3], one population of 1e7) a default C3 run adds to its line")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--selfcheck-launch", action="store_true",
                    help="rank start-up and rendezvous only (gloo, no GPU): what tests/ use to cover --gpus N on a CPU box")
    return ap.parse_args()


def launch_ranks(args):
    """Parent of an N > 1 run started as plain `python bench.py --gpus N`: N ranks through torch.distributed.run,
    as children.  Nothing in this process has touched HIP (torch is not even imported here)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def recorded(name):
    """A figure recorded under profiles/ for cross-reference (never measured by this run)."""
    try:
        return json.load(open(ROOT / "profiles" / f"{PROFILE_TAG}_recorded.json")).get(name)
    except Exception:
        return None


def fill_missiles(eng, n_targets, m, S):
    """Launch until m missiles are in flight (some of the synthetic lead-collision solves fail: the next
    candidates are the failed targets' neighbours)."""
    if m <= 0:
        return 0
    launched, shift = 0, 0
    base = S.missile_targets(n_targets, m).astype("int64")
    want = base
    while launched < m and shift < 16 and len(want):
        k = eng.launch_missiles(want % n_targets)
        failed = want[eng.launch_results["rc"] != 0]
        launched += k
        shift += 1
        want = (failed + shift)[: m - launched]
    return launched


def build_engine(workload, rank, world, device):
    import numpy as np
    from zrk_modulation_amd.engine import HotPathEngine
    from zrk_modulation_amd import scenario as S
    n_total, R, m_total = S.WORKLOADS[workload]
    radars = S.synthetic_radars(R)
    if workload in S.STRONG:
        lo, hi = S.shard_bounds(n_total, world, rank)
        ids, sp, vel, t0 = S.population_slice(S.SEEDS[workload], lo, hi)
        n = hi - lo
        m = (m_total * (rank + 1)) // world - (m_total * rank) // world
        stride = (n_total + world - 1) // world + (m_total + world - 1) // world
    else:
        n, m = n_total, m_total
        ids, sp, vel, t0 = S.synthetic_targets(n, S.SEEDS[workload] + 1000 * rank, first_id=1000 + rank * n)
        stride = n + m
    # global index space: shard g starts at g * stride (targets, then the missiles the rank launches)
    eng = HotPathEngine(device=device, dt_ms=10, seed=S.SEEDS[workload], noise="philox", gid0=rank * stride)
    eng.load(ids, sp, vel, t0, radars, missile_capacity=m)
    if world == 1:
        eng.enable_lists()
    launched = fill_missiles(eng, n, m, S)
    return eng, dict(n=n, R=R, m=m, launched=launched, stride=stride, scene=(ids, sp, vel, t0, radars),
                     n_total=n_total if workload in S.STRONG else n * world)


def build_ensemble(workload, rank, world, device):
    from zrk_modulation_amd.ensemble import EnsembleEngine
    from zrk_modulation_amd import scenario as S
    scen, n, R, m = S.ENSEMBLES[workload]
    eng = EnsembleEngine(device=device, dt_ms=10, noise="philox")
    first = rank * scen
    eng.load_synthetic(scen, n, R, m, seed=S.SEEDS[workload], first_scenario=first)
    return eng, dict(n=scen * n, R=R, m=scen * m, launched=eng.launched, scenarios=scen, per_scenario=n,
                     n_total=scen * n * world)


def cpu_baseline(info, eng, budget_s=15.0):
    """Time the oracle's L1 tick on the same scene (test infrastructure used as the reported CPU
    baseline only), on all host cores (OpenMP: AirEnv step and radar phase) and on one."""
    import numpy as np
    from oracle import oracle as O
    L = O.lib()
    ids, sp, vel, t0, radars = info["scene"]
    st = eng.store
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
        # (AirEnv.step on `threads` cores: targets in parallel, missiles in list order; identical results to the literal loop)
        nev = L.zo_airenv_step_mt(n, cap, 10 * k, 10, O.dptr(hsp), O.dptr(hvel), O.dptr(ht0), O.u8ptr(alive),
                                  O.u8ptr(kind), O.i32ptr(mrow), O.dptr(pos), O.dptr(prev), O.u8ptr(pv),
                                  O.i32ptr(m_tgt), O.dptr(m_radius), O.dptr(m_period), O.u8ptr(m_status),
                                  O.i32ptr(evm), O.i32ptr(evt), O.u8ptr(evs), threads)
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
    return {"value": live * ticks / t_all, "unit": "entity-timesteps/s", "cores": cores, "cpu_model": cpu_model(),
            "kind": "port",
            "sample": f"same scene, {ticks} ticks, oracle/zrk_oracle.c on {cores} OpenMP threads: AirEnv step (targets in parallel, "
                      f"the {m} missiles in list order), radar phase, per-radar compaction (one thread)",
            "value_1core": live / t_1core,
            "reference_python_note": "reference itself (pure Python) measured in the survey container: "
                                     "1.6 us/entity + 4.3 us/(radar x entity), 1 thread (BASELINE.md section 2)"}


def battery_config(workload, S):
    """The closed-loop workloads' scene as the YAML-schema dictionary the reference loads (main.py:35-149): configs[1]'s targets
    and radars, four launchers with a quarter of its missiles each, the command post over all of them."""
    n, R, m = S.WORKLOADS[workload.replace("-battery", "")]
    ids, sp, vel, _t0 = S.synthetic_targets(n, S.SEEDS[workload.replace("-battery", "")])
    radars = S.synthetic_radars(R)
    for k, rd in enumerate(radars):
        rd["id"] = 10 + k
    per = max(1, m // 4)
    launchers = [dict(id=100 + l, position=[float(x), float(y), 0.0], max_missiles=per,
                      missiles=[dict(id=10_000_000 * (1 + l) + k, velocity=1000.0, explosion_radius=150.0, life_time=60.0) for k in range(per)])
                 for l, (x, y) in enumerate([(0.0, 0.0), (3000.0, 1500.0), (-2000.0, 4000.0), (5000.0, -500.0)])]
    return dict(simulation=dict(time_step=10, duration=0),
                air_environment=dict(id=999, position=[0.0, 0.0, 0.0],
                                     targets=[dict(id=int(ids[i]), type="AIR_PLANE", position=sp[i].tolist(), velocity=vel[i].tolist()) for i in range(n)]),
                radars=radars, missile_launchers=launchers,
                combat_control_point=dict(id=0, missile_launcher_ids=[l["id"] for l in launchers], radar_ids=[r["id"] for r in radars]))


def measure_battery(args, device):
    """--workload C2-battery: a step is one tick of the CLOSED loop.  The sweep's duration: HIP events on every 16th tick of the
    timed region (that tick synchronises; the others are enqueued back to back)."""
    import numpy as np
    import torch
    from zrk_modulation_amd import scenario as S
    from zrk_modulation_amd.battery import DeviceBattery
    from zrk_modulation_amd.engine import HotPathEngine
    cfg = battery_config(args.workload, S)
    T = cfg["air_environment"]["targets"]
    n = len(T)
    ids = np.array([t["id"] for t in T], np.int64); sp = np.array([t["position"] for t in T]); vel = np.array([t["velocity"] for t in T])
    nm = sum(len(l["missiles"]) for l in cfg["missile_launchers"])
    eng = HotPathEngine(device=device, dt_ms=10, seed=S.SEEDS[args.workload.replace("-battery", "")], noise="philox")
    eng.load(ids, sp, vel, 0.0, cfg["radars"], missile_capacity=nm).enable_lists()
    bat = DeviceBattery(eng, cfg["missile_launchers"], rounds=int(os.environ.get("ZRK_BATTERY_ROUNDS", "2")))
    steps, warmup = args.steps, args.warmup
    bat.run(warmup)
    torch.cuda.synchronize(device)
    live0 = eng.alive_count()
    samples = []
    t0 = time.perf_counter()
    for k in range(steps):
        if k % 16 == 15:
            one = np.zeros(1, np.float32)
            bat.run(1, sweep_ms=one)
            samples.append(float(one[0]) * 1e3)
        else:
            bat.run(1)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    live1 = eng.alive_count()
    res = bat.results()                      # (raises if the command post's step did not go through)
    eng.store.compact_status()
    sweep_us = float(np.mean(samples)) if samples else float("nan")
    live_avg = 0.5 * (live0 + live1)
    alg_bytes = 85.0 * live_avg + 1.0 * (eng.store.n_uploaded - live_avg)
    achieved = alg_bytes / (sweep_us * 1e-6) / 1e9
    R = len(cfg["radars"])
    out = {
        "metric": "entity-timesteps/sec (targets+missiles)", "value": float(min(live0, live1)) * steps / elapsed, "unit": "entity-timesteps/s",
        "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n} AirObjects, {R} SectorRadars, {len(cfg['missile_launchers'])} launchers with {nm} missiles, the command "
                               "post and the launchers on the device (closed loop: sweep + missile step + lists, launchers' step, zrk_ccp_step, next salvo), "
                               "dt=10 ms, Philox measurement noise, the reference's message latencies, nothing read back inside the loop",
                   "entities_per_gpu": eng.store.n_uploaded, "live_per_gpu": int(live1), "parallelism": "shard1",
                   "loop": "closed loop, one tick per call of the plain two-launch loop + nine event-rate launches",
                   "launch_solves": len(res["solves"]), "launches": len(res["new_missile"]), "detonations": len(res["detonations"]),
                   "ticks_run": bat.tick},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "achievable_peak": HBM_ACHIEVABLE_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS, "kernel": "k_tick_sweep",
                     "avg_kernel_us": sweep_us, "samples": len(samples), "timed_by": "HIP events riding on the dispatch, every 16th tick of the timed region",
                     "algorithmic_bytes_per_launch": alg_bytes, "ticks_per_launch": 1,
                     "whole_tick": {"achieved": alg_bytes / (elapsed / steps) / 1e9, "frac": alg_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS}},
        "cpu_baseline": None,
    }
    if not args.no_cpu_baseline:
        from oracle.battery import OracleBattery
        ob = OracleBattery(cfg, None)
        t_a = time.perf_counter()
        ticks = 0
        while ticks < 3 or (time.perf_counter() - t_a < args.cpu_budget and ticks < 200):
            ob.tick()
            ticks += 1
        t_cpu = time.perf_counter() - t_a
        out["cpu_baseline"] = {"value": float(len(ob.sim.active_slots())) * ticks / t_cpu, "unit": "entity-timesteps/s", "cores": 1, "cpu_model": cpu_model(),
                               "kind": "port", "sample": f"the same scene, the first {ticks} ticks of the closed loop through oracle/battery.py (AirEnv step, "
                                                          "radars one after the other, launchers, the command post's sequential loop; no measurement noise) on one thread"}
    bat.close()
    return out


def cpu_baseline_ensemble(self, budget_s, cores, cpu_model):
    """The oracle (oracle/, test infrastructure) timed on scenario 0 of the batch: the reported CPU baseline."""
    import time
    from oracle import oracle as O
    from zrk_modulation_amd.engine import scan_mode_code, scan_next
    import numpy as np
    L = O.lib()
    v = self.scenario_view(0).store
    n = v.n_uploaded
    hsp = np.ascontiguousarray(v.h_sp.T).reshape(-1); hvel = np.ascontiguousarray(v.h_vel.T).reshape(-1)
    ht0 = v.h_t0.copy(); pos = np.ascontiguousarray(v.h_pos0.T).reshape(-1).copy()
    prev = pos.copy(); pv = np.zeros(n, np.uint8); alive = np.ones(n, np.uint8)
    kind = v.h_kind.copy(); mrow = np.full(n, -1, np.int32)
    m = v.m
    mrow[v.hm_slot] = np.arange(m, dtype=np.int32)
    m_tgt = v.hm_tgt.astype(np.int32).copy(); m_radius = np.full(max(m, 1), 150.0); m_period = np.full(max(m, 1), 60.0)
    m_status = np.ones(max(m, 1), np.uint8)
    evm = np.zeros(max(m, 1), np.int32); evt = np.zeros(max(m, 1), np.int32); evs = np.zeros(max(m, 1), np.uint8)
    vis = np.zeros(n, np.uint32)
    rs = [dict(r, caz=r["azimuth_start"], cel=r["elevation_start"]) for r in self.radars[0]]
    out = np.zeros(n, np.int32)

    def tick(k):
        L.zo_airenv_step(n, n, self.dt_ms * k, self.dt_ms, O.dptr(hsp), O.dptr(hvel), O.dptr(ht0), O.u8ptr(alive),
                         O.u8ptr(kind), O.i32ptr(mrow), O.dptr(pos), O.dptr(prev), O.u8ptr(pv), O.i32ptr(m_tgt),
                         O.dptr(m_radius), O.dptr(m_period), O.u8ptr(m_status), O.i32ptr(evm), O.i32ptr(evt), O.u8ptr(evs))
        arr = O.radar_array([(r["position"][0], r["position"][1], r["position"][2], r["max_distance"], r["caz"],
                              r["azimuth_range"], r["cel"], r["elevation_range"]) for r in rs])
        L.zo_radar_phase_fused(n, n, O.dptr(pos), O.u8ptr(alive), len(rs), arr, 1, None, int(self.seeds[0]), k, 0,
                               O.u32ptr(vis), cores)
        for r in range(len(rs)):
            L.zo_compact_bit(n, O.u32ptr(vis), r, 0, O.i32ptr(out))
        for r in rs:
            r["caz"], r["cel"] = scan_next(scan_mode_code(r.get("scan_mode", "horizontal")), r["azimuth_range"],
                                           r["azimuth_speed"], r["elevation_speed"], r["elevation_start"], r["caz"], r["cel"])

    t_a = time.perf_counter(); tick(0); t_one = time.perf_counter() - t_a
    ticks = int(max(2, min(2000, budget_s / max(t_one, 1e-6))))
    t_a = time.perf_counter()
    for k in range(1, 1 + ticks):
        tick(k)
    t_all = time.perf_counter() - t_a
    return {"value": n * ticks / t_all, "unit": "entity-timesteps/s", "cores": cores, "cpu_model": cpu_model,
            "kind": "port", "sample": f"scenario 0 of the batch ({n} entities, {len(rs)} radars), {ticks} ticks, "
                                      f"oracle/zrk_oracle.c with the radar phase on {cores} OpenMP threads"}


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or under "
                 f"torch.distributed.run with --nproc-per-node equal to --gpus")
    if args.selfcheck_launch:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([int(os.environ.get("RANK", "0"))], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"selfcheck": True, "n_gpus": world, "rank_sum": int(t.item())}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")       # kernel arguments in device memory (PyTorch's default too)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the library's helper threads stay runnable for this long after their last item (default 5 ms: an embedding application
    # should not find cores held between its calls); here the warm-up call, the spin-up and the timed call are tens of
    # milliseconds apart, and a helper woken from its sleep at the timed call's entry can cost that call milliseconds
    os.environ.setdefault("ZRK_HELPER_YIELD_MS", "250")
    # ... and with ONE rank on the node they do not even give up their core in that time (a thread that yields on a host busy
    # with other tenants' work may not be back for a scheduler's slice: one of forty runs of the driver's command waited 1.1 ms
    # for the side stream's thread, profiles/r05_driver_regime_distribution.txt).  With several ranks the node's cores are the
    # ranks' to share: the library's default (1 ms) stands
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1:
        os.environ.setdefault("ZRK_HELPER_IDLE_MS", "300")

    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ZRK_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks
    backend = os.environ.get("ZRK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    from zrk_modulation_amd import scenario as S

    if args.workload.endswith("-battery"):
        if world != 1:
            sys.exit("bench.py: the closed-loop workloads run on one GPU")
        print(json.dumps(measure_battery(args, device)), flush=True)
        return

    def measure(workload, steps, warmup, cpu_base, wire=None):
        """One workload, set up, warmed up and timed as the docstring says; rank 0 gets the line's dictionary."""
        wire = wire or args.wire
        from zrk_modulation_amd import scenario as S
        ensemble = workload in S.ENSEMBLES
        strong = workload in S.STRONG
        if ensemble:
            eng, info = build_ensemble(workload, rank, world, device)
        else:
            eng, info = build_engine(workload, rank, world, device)
        # ZRK_BENCH_FORCE_EXCHANGE=1: the N > 1 control flow (C-side exchange, list sizing, overflow report) on ONE rank, for
        # boxes with one GPU: RCCL runs with a one-rank communicator, nothing crosses a link
        exchanging = (world > 1 or bool(os.environ.get("ZRK_BENCH_FORCE_EXCHANGE"))) and not ensemble
        state = {"c_side": exchanging and backend == "nccl"}     # may fall back to the Python exchange at set-up
        xchg = {"x": None, "ex": [], "buf": [], "work": [None, None], "tick": 0, "entries": 0, "words": 0}
        # room for a tick's detonations behind the list: 1024 of them (8 KB) -- a tick that has more is reported as an overflow
        # (room for all 10^4 missiles at once would be 40 % of what a rank sends per tick, for a list that is nearly always empty)
        ev_cap = max(64, min(info["m"], 1024)) if exchanging else 0

        def coll_device(t):
            return t.to(device) if backend == "nccl" else t.cpu()

        def all_reduce(t, op):
            if world > 1:
                dist.all_reduce(t, op=op)

        def size_exchange(entries):
            """Lists for up to `entries` seen objects per rank, in the wire format of zrk_compact_bits (count, n, one
            bit per slot, 16-bit masks): a quarter of the bytes of (index, mask) pairs."""
            from zrk_modulation_amd.exchange import DetectionExchange, RcclExchange, union_bits_words
            n_slots = coll_device(torch.tensor([int(eng.store.cap)], dtype=torch.int64))
            all_reduce(n_slots, dist.ReduceOp.MAX)
            words = union_bits_words(int(n_slots.item()), info["R"], 0 if wire == "union" else entries)
            offsets = [g * info["stride"] for g in range(world)]
            made = False
            if state["c_side"] and xchg["x"] is not None:
                torch.cuda.synchronize(device)
                xchg["x"].resize(words)                      # same communicator, new buffers
                made = True
            elif state["c_side"]:
                try:
                    if args.interest:
                        eng.det_idx = None           # (the exchanged list is shaped for the remote consumers; no local per-radar lists beside it)
                    xchg["x"] = RcclExchange(words, device, info["R"], offsets=offsets, ev_capacity=ev_cap, wire=wire,
                                             interest=[int(r) for r in args.interest.split(",") if r.strip()] or None)
                    made = True
                except Exception as exc:                     # the library's own communicator could not be set up here
                    print(f"[bench rank {rank}] C-side exchange unavailable ({exc}); falling back to torch.distributed", file=sys.stderr)
            if state["c_side"]:
                ok = coll_device(torch.tensor([1 if made else 0], dtype=torch.int64))
                all_reduce(ok, dist.ReduceOp.MIN)               # every rank takes the same path
                if int(ok.item()) == 0:
                    if xchg["x"] is not None:
                        xchg["x"].close()
                        xchg["x"] = None
                    state["c_side"] = False
            if not state["c_side"]:
                xchg["ex"] = [DetectionExchange(words, device, fmt="bits", offsets=offsets, R=info["R"], union_only=wire == "union")
                              for _ in range(2)]
                xchg["buf"] = [torch.zeros(words, dtype=torch.int64, device=device) for _ in range(2)]
                xchg["work"] = [None, None]
            xchg["entries"], xchg["words"] = int(entries), int(words)

        def drain_exchange():
            if state["c_side"] and xchg["x"] is not None:
                xchg["x"].sync()
            for k, w in enumerate(xchg["work"]):
                if w is not None:
                    w.wait()
                    xchg["work"][k] = None

        def max_count():
            if state["c_side"]:
                return max(max(xchg["x"].counts(k)) for k in range(xchg["x"].slots))
            return max(max(e.counts()) for e in xchg["ex"])

        if exchanging:
            size_exchange(eng.store.cap)     # (may settle for the Python exchange: decided before anything depends on it)

        # every 10th tick's sweep is timed when the run is short (the last tick of each window of ten: two samples in the
        # driver's 20-step run), every 8th otherwise: a timed launch costs the stream 5-13 us of idle device around it, whether
        # the events ride on the dispatch or are recorded around it (profiles/r03_timeline_20steps.txt: five samples cost the
        # 20-step run 3 us per tick); the events are read after the timed region
        stride = int(os.environ.get("ZRK_BENCH_STRIDE", 0)) or (10 if steps <= 64 else 8)
        if exchanging and not state["c_side"]:
            stride = 64                      # ticks driven one call at a time: reading the events drains the stream
        deferred = not (exchanging and not state["c_side"])

        def run_ticks(k, sweep_ms=None, every=None):
            every = every or stride
            ps = (-every if deferred else every) if sweep_ms is not None else 1
            if not exchanging:
                eng.run(k, sweep_ms=None if deferred else sweep_ms, prof_stride=ps)
                return
            if state["c_side"]:
                eng.run(k, sweep_ms=None if deferred else sweep_ms, prof_stride=ps, exchange=xchg["x"])
                return
            for j in range(k):               # rehearsal path: the exchange goes through torch.distributed, tick by tick
                b = xchg["tick"] & 1
                if xchg["work"][b] is not None:
                    xchg["work"][b].wait()
                eng.packed = xchg["buf"][b]
                eng.loop.flags |= 8          # ZRK_F_UNION_BITS
                one = np.zeros(1, np.float32) if (sweep_ms is not None and j % stride == 0) else None
                eng.run(1, sweep_ms=one, prof_stride=1)
                xchg["work"][b] = xchg["ex"][b].all_gather(eng.packed, async_op=True)
                xchg["tick"] += 1
                if one is not None:
                    sweep_ms[j // stride] = one[0]

        def barrier():
            torch.cuda.synchronize(device)
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize(device)

        def spin_up():
            """~100 ms of an unrelated kernel (the noise self-test) so that the clocks are up when the timed steps start, as
            they are inside any run longer than a few milliseconds (see the docstring)."""
            buf = torch.zeros(3 << 20, dtype=torch.float64, device=device)
            st0 = eng.store
            t_spin = time.perf_counter()
            while (time.perf_counter() - t_spin) * 1e3 < SPINUP_MS:
                for _ in range(8):
                    st0.ctx.check(st0.lib.zrk_selftest_noise(st0.ctx.handle, 1, 1, 3, 0, buf.data_ptr(), 1 << 20, None), "spin-up")
                torch.cuda.synchronize(device)

        # the warm-up steps go through the same code as the timed ones, sweep timing included: the library creates its timing
        # events on first use, and a first hipExtLaunchKernel is slow -- neither belongs into the timed region
        # (every warm-up sweep is timed, so that the timed region's events exist already)
        stamping = (hasattr(eng, "sweep_stamps") and not (exchanging and not state["c_side"])
                    and os.environ.get("ZRK_BENCH_STAMPS", "1") != "0")       # (=0: event pairs as in round 3, for A/B runs)
        if stamping:
            eng.sweep_stamps(True)           # (the ring of stamps is allocated by the first call that uses it: a warm-up call)
        warm_ms = np.zeros(warmup, np.float32) if (warmup > 0 and deferred) else None
        run_ticks(warmup, warm_ms, every=1)
        if warm_ms is not None and deferred:
            eng.read_sweep_ms(len(warm_ms))
        overflow = False
        if exchanging and warmup > 0:
            # size the fixed lists from what the warm-up saw (1.25x the largest per-rank count: the count follows the
            # sectors round their scan period, which a full warm-up covers; an overflow is reported)
            drain_exchange()
            seen = coll_device(torch.tensor([max_count()], dtype=torch.int64))
            all_reduce(seen, dist.ReduceOp.MAX)
            size_exchange(int(seen.item() * 1.25) + 1024)
        barrier()
        live0 = eng.alive_count()
        # with the sweeps timing themselves nothing else times them inside the region (two event pairs cost a 20-step run
        # 2-3 us per tick); the rehearsal backends, which have no stamps, keep the event pairs
        sweep_ms = np.zeros((steps + stride - 1) // stride if not stamping else 0, np.float32)
        spin_up()
        barrier()
        t0 = time.perf_counter()
        run_ticks(steps, sweep_ms if len(sweep_ms) else None)
        t_issued = time.perf_counter()
        if exchanging:
            drain_exchange()
        barrier()
        elapsed = time.perf_counter() - t0
        # (asked here: the calibration launches behind the region are calls of one tick, which take the plain loop)
        timed_call_overlapped = eng.store.lib.zrk_last_run_overlapped(eng.store.ctx.handle) == 1
        sweep_ticks = np.ones(len(sweep_ms), np.int32)
        if deferred and len(sweep_ms):
            sweep_ms[:] = eng.read_sweep_ms(len(sweep_ms))
            if hasattr(eng, "read_sweep_ticks"):
                sweep_ticks[:] = eng.read_sweep_ticks(len(sweep_ms))
        stamp_us, stamp_ticks = eng.read_sweep_stamps() if stamping else (np.zeros(0, np.float32), np.zeros(0, np.int32))
        # when the sampled launches ran, by the device's clock: with every launch of the call sampled (up to 64), the span of the
        # call's sweeps and the idle time between them -- what is left of the region is the call's start and its drain
        span_us = gaps_us = None
        if stamping and hasattr(eng, "sweep_stamp_times") and len(stamp_us) and int(stamp_ticks.sum()) == steps:
            sb, se = eng.sweep_stamp_times()
            if len(sb) == len(stamp_us):
                span_us = float(se[-1] - sb[0])
                gaps_us = float(np.maximum(sb[1:] - se[:-1], 0.0).sum())
        live1 = eng.alive_count()
        eng.store.compact_status()           # outside the timing: a compaction that did not run to completion raises here
        if exchanging:
            overflow = xchg["x"].overflowed() if state["c_side"] else any(e.overflowed() for e in xchg["ex"])
        # What a launch costs its stream outside "first wave in to last wave out" -- the dispatch in front (the packet, 10-20 KB of
        # arguments) and the release behind -- is what a profiler's begin / end stamps of the same launch add to the launch's own:
        # measured behind the region on stand-alone launches (calls of one tick, device otherwise idle), each timed both ways
        overhead_us = []
        # (the exchange's own counters as the timed call left them: the calibration calls below post collectives of their own)
        xinfo_timed = xchg["x"].info() if (exchanging and state["c_side"] and xchg["x"] is not None) else None
        if stamping and deferred:
            for _ in range(10):
                run_ticks(1, np.zeros(1, np.float32), every=1)
                ev_us = float(eng.read_sweep_ms(1)[0]) * 1e3
                st_us, _tk = eng.read_sweep_stamps()
                if len(st_us) and ev_us > 0:
                    overhead_us.append(ev_us - float(st_us[0]))
            if exchanging:
                drain_exchange()
            eng.store.compact_status()       # (the calibration calls' compactions as well)
            overhead_us = sorted(overhead_us)[1:-1]             # (without the two extremes)
        dispatch_overhead_us = float(np.mean(overhead_us)) if len(overhead_us) else 0.0

        elapsed_rank0 = elapsed
        el = coll_device(torch.tensor([elapsed], dtype=torch.float64))
        units = coll_device(torch.tensor([float(min(live0, live1)) * steps], dtype=torch.float64))
        all_reduce(el, dist.ReduceOp.MAX)
        all_reduce(units, dist.ReduceOp.SUM)
        elapsed = float(el.item()); total_units = float(units.item())

        if rank == 0:
            n_slots = eng.store.n_uploaded
            live_avg = 0.5 * (live0 + live1)
            dead = n_slots - live_avg
            # a launch of the overlapped loop sweeps two consecutive ticks in one pass (zrk_hot.h: zrk_read_sweep_ticks): the
            # samples are launches; only those of the prevailing kind are averaged (an odd tick at a call's end is a launch of one)
            ev_ok = sweep_ms > 0
            if len(stamp_us):
                tpl = int(np.bincount(stamp_ticks).argmax())
                wave_us = stamp_us[stamp_ticks == tpl].astype(np.float64)
                good = wave_us + dispatch_overhead_us
                timing = ("the launches' own wall-clock stamps (zrk_sweep_stamps: first wave in to last wave out) + the dispatch / release "
                          "overhead of a launch measured behind the region (dispatch_overhead_us)")
            else:                                                  # (rehearsal backends: the event pairs are all there is)
                tpl = int(np.bincount(sweep_ticks[ev_ok]).argmax()) if ev_ok.any() else 1
                good = wave_us = sweep_ms[ev_ok & (sweep_ticks == tpl)].astype(np.float64) * 1e3
                timing = "HIP events riding on the dispatch"
            sweep_avg_us = float(good.mean()) if len(good) else float("nan")
            # bytes the launch MUST move: the trajectory columns and the flag once (57 B), position and mask per tick swept
            # (28 B), one flag byte per tombstone; and SURVEY 8d's per-tick figure times the ticks swept
            alg_bytes_tick = 85.0 * live_avg + 1.0 * dead
            alg_bytes = (57.0 + 28.0 * tpl) * live_avg + 1.0 * dead
            eff_bytes = alg_bytes_tick * tpl
            achieved = alg_bytes / (sweep_avg_us * 1e-6) / 1e9
            eff_achieved = eff_bytes / (sweep_avg_us * 1e-6) / 1e9
            if ensemble:
                what = (f"{workload}: {info['scenarios']} independent scenarios x {info['per_scenario']} AirObjects, "
                        f"{info['R']} SectorRadars and {info['m'] // info['scenarios']} missiles each, per GPU, one batched table")
            else:
                what = (f"{workload}: {info['n']} AirObjects, {info['R']} SectorRadars, "
                        f"{info['launched']}/{info['m']} missiles in flight per GPU")
                if strong:
                    what += f" (rank 0's shard of ONE population of {info['n_total']})"
            what += ", dt=10 ms, Philox measurement noise, "
            if exchanging:
                what += ("union compaction in the bitmap wire format + per-tick RCCL all-gather of the detection list and the "
                         "detonation events, " + ("issued from the C side, overlapped with the next sweep" if state["c_side"]
                                                  else "through torch.distributed (rehearsal backend)"))
            else:
                what += "per-radar compaction"
            overlapped = timed_call_overlapped
            loop_mode = ("two launches per tick on one stream" if not overlapped else
                         "overlapped: tick t's compaction on a side stream beside tick t+1's sweep" if tpl == 1 else
                         "overlapped, two ticks per sweep launch: the trajectory columns are read once for ticks t and t+1, their "
                         "compactions run on a side stream beside the next launch")
            out = {
                "metric": "entity-timesteps/sec (targets+missiles)", "value": total_units / elapsed,
                "unit": "entity-timesteps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
                "scaling": "strong" if strong else "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": what, "entities_per_gpu": n_slots, "live_per_gpu": int(live1),
                           "parallelism": f"shard{world}" if not ensemble else f"replicas{world}",
                           "loop": loop_mode},
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                             # (what a pure streaming kernel reaches on this part, and the launch's bytes against that)
                             "achievable_peak": HBM_ACHIEVABLE_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                             "traffic_recorded": recorded(f"traffic_bytes_{workload}") if world == 1 else None,
                             "kernel": "k_tick_sweep", "avg_kernel_us": sweep_avg_us, "samples": int(len(good)),
                             "min_kernel_us": float(good.min()) if len(good) else None,
                             "max_kernel_us": float(good.max()) if len(good) else None, "timed_by": timing,
                             # the MEASURED quantity: every sweep launch of the timed call, by its own stamps.  avg_kernel_us adds
                             # dispatch_overhead_us to it, which is measured on another launch shape (one tick, device idle, behind the region)
                             "first_wave_in_to_last_wave_out_us": float(wave_us.mean()) if len(wave_us) else None,
                             "dispatch_overhead_us": dispatch_overhead_us,
                             "first_wave_in_to_last_wave_out_us_per_launch": [round(float(v), 2) for v in stamp_us] if 0 < len(stamp_us) <= 16 else None,
                             "algorithmic_bytes_per_launch": alg_bytes, "ticks_per_launch": tpl,
                             "effective": {"bytes_per_launch": eff_bytes, "achieved": eff_achieved, "frac": eff_achieved / HBM_PEAK_GBS,
                                           "what": "85 B per live entity and tick swept (SURVEY 8d) x ticks per launch, over the same duration"},
                             "profiled_kernel_us_recorded": recorded(f"sweep_us_{workload}") if world == 1 else None,
                             "profiled_kernel_us_alone_recorded": recorded(f"sweep_us_{workload}_plain_loop") if world == 1 else None,
                             "recorded_from": f"profiles/{PROFILE_TAG}_recorded.json (rocprofv3 passes of this command, committed)",
                             # the same algorithmic bytes over the whole tick (this rank's): what the loop around the kernel leaves of it
                             "whole_tick": {"achieved": alg_bytes_tick / (elapsed / steps) / 1e9,
                                            "frac": alg_bytes_tick / (elapsed / steps) / 1e9 / HBM_PEAK_GBS}},
                "setup": {"clock_spinup_ms": SPINUP_MS, "sweep_timing_stride": stride,
                          # of the timed region: until the one zrk_run_ticks call returned (everything issued, the side stream's
                          # work handed over), and the synchronisation behind it
                          "call_returned_after_us": (t_issued - t0) * 1e6, "sync_us": (elapsed_rank0 - (t_issued - t0)) * 1e6,
                          # by the launches' own stamps (all launches of the call sampled): first wave of the first sweep to last
                          # wave of the last one, and the idle time between consecutive sweeps within that span
                          "sweeps_span_us": span_us, "sweep_gaps_us": gaps_us},
            }
            if exchanging:
                out["config"]["exchange"] = "rccl, C side" if state["c_side"] else f"torch.distributed {backend}"
                out["config"]["exchange_entries_per_rank"] = xchg["entries"]
                out["config"]["exchange_bytes_per_rank"] = 8 * (xchg["words"] + (1 + ev_cap if state["c_side"] else 0))
                out["config"]["exchange_overflow"] = bool(overflow)
                out["config"]["exchange_wire"] = ("bitmap of the slots seen by any radar" if wire == "union" else
                                                  "bitmap + radar masks of the seen slots") + " + detonation events"
                # host threads a rank keeps busy inside a call: the caller + the library's helpers (zrk_exchange_plan_helpers: one where
                # the rank has fewer than three usable cores to itself -- the side stream's thread then issues the collectives too)
                helpers = xchg["x"].info()["helper_threads"] if state["c_side"] else int(eng.store.lib.zrk_exchange_plan_helpers(world))
                out["config"]["helpers_mode"] = (("one helper: the side stream's thread issues the collectives as well" if helpers == 1 else
                                                  "two helpers: the side stream's thread and a poster of the collectives") +
                                                 ("" if state["c_side"] else " (planned for this world size on this host; the rehearsal backend posts from Python)"))
                out["config"]["host_threads_per_rank"] = 1 + helpers
                if args.interest and state["c_side"]:
                    out["config"]["exchange_radars_of_interest"] = [int(r) for r in args.interest.split(",") if r.strip()]
                out["config"]["usable_host_cores"] = usable_cores()
                if state["c_side"]:
                    xi = xinfo_timed or xchg["x"].info()
                    # the first multi-rank record checks itself: RCCL's own count of the communicator's ranks, and how long
                    # the calling thread waited for collectives (per tick of the timed call and the calibration ticks behind it)
                    out["config"]["rccl_ranks_seen"] = xi["rccl_ranks_seen"]
                    out["config"]["exchange_pattern"] = xi["pattern"] + (", the two ticks of a launch as one group" if xi["grouped_pairs"] else "")
                    out["config"]["exchange_host_wait_us_per_tick"] = xi["host_wait_us"] / max(1, xi["collectives"])
                    out["config"]["exchange_host_waits"] = xi["host_waits"]
            if world == 1 and cpu_base and not ensemble:
                out["cpu_baseline"] = cpu_baseline(info, eng, args.cpu_budget)
            elif world == 1 and cpu_base and ensemble:
                out["cpu_baseline"] = cpu_baseline_ensemble(eng, args.cpu_budget, usable_cores(), cpu_model())
            else:
                out["cpu_baseline"] = None
        if xchg["x"] is not None:
            xchg["x"].close()
        return out if rank == 0 else None

    out = measure(args.workload, args.steps, args.warmup, not args.no_cpu_baseline)
    # N > 1: the line's value is measured with the wire the command line names (default: bitmap + radar masks, what a replicated
    # command post consumes); the other wire is timed behind it on the same ranks, so that both stand side by side in one record
    if world > 1 and args.workload not in S.ENSEMBLES:
        other = "union" if args.wire == "masks" else "masks"
        sub = measure(args.workload, min(args.steps, 200), min(max(args.warmup, 4), 50), False, wire=other)
        if rank == 0:
            out[f"{other}_wire"] = {k: sub[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")}
            out[f"{other}_wire"]["exchange_wire"] = sub["config"]["exchange_wire"]
            out[f"{other}_wire"]["exchange_bytes_per_rank"] = sub["config"]["exchange_bytes_per_rank"]
            out["config"]["headline_wire"] = args.wire
    # The scaling curve north_star names is on the ONE-population 1e7-target scenario (configs[3], strong scaling): the
    # default run (C3 per GPU, weak scaling -- the roofline configuration) measures it as well, behind the main timing, so
    # that `python bench.py --gpus N` for N = 1, 2, 4, 8 yields both curves (--no-c4 skips it).
    if args.workload == "C3" and not args.no_c4 and not os.environ.get("ZRK_BENCH_FORCE_EXCHANGE"):
        sub = measure("C4", min(args.steps, 60), min(max(args.warmup, 4), 12), False)
        if rank == 0:
            out["c4_strong"] = {k: sub[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling")}
            out["c4_strong"]["config"] = sub["config"]
            out["c4_strong"]["roofline"] = {k: sub["roofline"][k] for k in ("achieved", "frac", "avg_kernel_us", "ticks_per_launch", "samples")}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
