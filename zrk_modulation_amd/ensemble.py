"""Batched Monte-Carlo ensemble (BASELINE.json configs[4]): S independent scenarios in ONE device table, swept and
compacted by one launch each per tick (zrk_run_ticks_ensemble, include/zrk_hot.h).

The reference runs one scenario per process (main.py:151-174); an ensemble of them has no shared state, so the
only thing batching changes is how many launches a tick costs.  Scenario s owns rows [s * P, (s + 1) * P) of the
table (P = rows_per_scenario, a multiple of 1024): its targets in Morton order of their start position, then the
rows its missiles take as they are launched, then padding (alive = 0).  Radars, scan state and the noise key are
per scenario and live on the device; detection lists come back per (scenario, radar) in scenario-local list
indices.  Every scenario is bit-for-bit what a single-scenario HotPathEngine computes for it
(tests/test_gpu_ensemble.py checks each against its own oracle replay).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from ._lib import F_PHILOX
from .engine import morton_order_xy, scan_mode_code
from .store import EntityStore


def _round_up(n, q):
    return ((int(n) + q - 1) // q) * q


class EnsembleEngine:
    def __init__(self, device=None, dt_ms=10, noise="philox"):
        self.device = device
        self.dt_ms = int(dt_ms)
        self.noise = noise
        self.store = None
        self.launched = 0

    # ---- construction ---------------------------------------------------------------------------------------
    def load(self, scenarios, missile_capacity=0, seeds=None):
        """scenarios: list of dicts with ids, start_pos, velocity, start_time (list order) and radars (list of
        SectorRadar constructor dicts, the same count in every scenario)."""
        S = self.S = len(scenarios)
        R = self.R = len(scenarios[0]["radars"])
        assert all(len(sc["radars"]) == R for sc in scenarios), "every scenario of a batch has the same number of radars"
        n_max = max(len(sc["ids"]) for sc in scenarios)
        self.mcap = int(missile_capacity)
        P = self.P = _round_up(n_max + self.mcap, 1024)
        st = self.store = EntityStore(self.device, S * P, max(S * self.mcap, 64))
        st.two_vis = True
        st._alloc_entities(st.cap)
        ids = np.zeros(S * P, np.int64); sp = np.zeros((S * P, 3)); vel = np.zeros((S * P, 3)); t0 = np.zeros(S * P)
        lidx = np.tile(np.arange(P, dtype=np.int32), S)          # padding rows: their own place in the block
        alive = np.zeros(S * P, np.uint8)
        self.n_s = np.zeros(S, np.int64)                        # list length of each scenario so far
        self.row_of_list = []                                   # per scenario: table row of list element k
        for s, sc in enumerate(scenarios):
            n = len(sc["ids"])
            spos = np.asarray(sc["start_pos"], np.float64).reshape(n, 3)
            order = morton_order_xy(spos) if n > 1 else np.arange(n)
            rows = s * P + np.arange(n)
            ids[rows] = np.asarray(sc["ids"])[order]
            sp[rows] = spos[order]
            vel[rows] = np.asarray(sc["velocity"], np.float64).reshape(n, 3)[order]
            t0[rows] = np.broadcast_to(np.asarray(sc["start_time"], np.float64), (n,))[order]
            lidx[rows] = order.astype(np.int32)
            alive[rows] = 1
            rol = np.empty(n, np.int64); rol[order] = rows
            self.row_of_list.append(rol)
            self.n_s[s] = n
        st.add_entities(ids, sp, vel, t0, kind=0, list_index=lidx)
        st.flush()
        st.h_alive[:] = alive
        st.d_alive[:S * P] = torch.from_numpy(alive).to(st.device)
        # radars on the device
        lib = st.lib
        rad = (_lib.ZrkRadar * (S * max(R, 1)))()
        scan = (_lib.ZrkScan * (S * max(R, 1)))()
        d2 = np.zeros(S * max(R, 1))
        self.radars = [[dict(r) for r in sc["radars"]] for sc in scenarios]
        for s, sc in enumerate(scenarios):
            for k, rd in enumerate(sc["radars"]):
                cr, cs = rad[s * R + k], scan[s * R + k]
                cr.pos[0], cr.pos[1], cr.pos[2] = (float(v) for v in rd["position"])
                cr.max_distance = float(rd["max_distance"])
                cr.cur_azimuth, cr.azimuth_range = float(rd["azimuth_start"]), float(rd["azimuth_range"])
                cr.cur_elevation, cr.elevation_range = float(rd["elevation_start"]), float(rd["elevation_range"])
                cs.azimuth_speed, cs.elevation_speed = float(rd["azimuth_speed"]), float(rd["elevation_speed"])
                cs.elevation_start = float(rd["elevation_start"])
                cs.mode = scan_mode_code(rd.get("scan_mode", "horizontal"))
                d2[s * R + k] = lib.zrk_d2_threshold(float(rd["max_distance"]))
        dev = st.device
        self.d_radars = torch.frombuffer(bytearray(bytes(rad)), dtype=torch.uint8).clone().to(dev)
        self.d_scan = torch.frombuffer(bytearray(bytes(scan)), dtype=torch.uint8).clone().to(dev)
        self.d_d2 = torch.from_numpy(d2).to(dev)
        self.seeds = np.asarray(seeds if seeds is not None else np.arange(S), np.uint64)
        self.d_seeds = torch.from_numpy(self.seeds.view(np.int64)).to(dev)
        self.d_tables = torch.zeros(int(lib.zrk_ensemble_table_bytes(S)), dtype=torch.uint8, device=dev)
        ens = self.ens = _lib.ZrkEnsemble()
        ens.scenarios, ens.radars, ens.rows_per_scenario = S, R, P
        ens.radar_state, ens.scan, ens.d2_max = self.d_radars.data_ptr(), self.d_scan.data_ptr(), self.d_d2.data_ptr()
        ens.seeds, ens.tables = self.d_seeds.data_ptr(), self.d_tables.data_ptr()
        loop = self.loop = _lib.ZrkLoop()
        loop.n = S * P
        loop.time_ms, loop.dt_ms = 0, self.dt_ms
        loop.gid0, loop.seed, loop.tick = 0, 0, 0
        loop.cur, loop.base_index = st.cur, 0
        loop.flags = F_PHILOX if self.noise == "philox" else 0
        self.det_stride = P
        self.det_idx = torch.zeros(S * max(R, 1) * P, dtype=torch.int32, device=dev)
        self.det_cnt = torch.zeros(S * (R + 1), dtype=torch.int32, device=dev)
        return self

    def load_synthetic(self, scenarios, n, R, m, seed, first_scenario=0):
        """`scenarios` synthetic scenes of n targets, R radars and (up to) m missiles each: scenario k draws its
        targets from seed * 1000 + first_scenario + k, starts its sectors 10 degrees further round than k - 1 and
        uses noise key seed + first_scenario + k."""
        from . import scenario as SC
        scs = []
        for k in range(scenarios):
            g = first_scenario + k
            ids, sp, vel, t0 = SC.synthetic_targets(n, seed * 1000 + g)
            radars = SC.synthetic_radars(R)
            for rd in radars:
                rd["azimuth_start"] = float((10 * g) % 270)
            scs.append(dict(ids=ids, start_pos=sp, velocity=vel, start_time=t0, radars=radars))
        self.load(scs, missile_capacity=m, seeds=[seed + first_scenario + k for k in range(scenarios)])
        if m:
            tg = SC.missile_targets(n, m)
            self.launch_missiles([tg] * scenarios)
        return self

    def launch_missiles(self, target_lists, launcher_pos=(0.0, 0.0, 0.0), speed=1000.0, radius=150.0, period=60.0):
        """Batched Missile._launch at the current time: target_lists[s] = list indices in scenario s.  One launch-solve
        launch for all scenarios; the successful ones take the next free rows of their scenario."""
        st = self.store
        S, P = self.S, self.P
        rows = np.concatenate([self.row_of_list[s][np.asarray(t, np.int64)] for s, t in enumerate(target_lists)])
        scen = np.concatenate([np.full(len(t), s, np.int64) for s, t in enumerate(target_lists)])
        k = len(rows)
        if k == 0:
            return 0
        req_t, res_t = _lib.launch_dtypes()
        req = np.zeros(k, dtype=req_t)
        req["target_slot"] = rows.astype(np.int32)
        req["missile_pos"] = np.asarray(launcher_pos, np.float64)
        req["speed"], req["period"] = speed, period
        d_req = torch.from_numpy(req.view(np.uint8).reshape(-1)).to(st.device)
        d_res = torch.zeros(k * C.sizeof(_lib.ZrkLaunchRes), dtype=torch.uint8, device=st.device)
        st.ctx.check(st.lib.zrk_launch_solve(st.ctx.handle, C.byref(st.ents), st.cur, d_req.data_ptr(), d_res.data_ptr(), k,
                                             st._stream()), "zrk_launch_solve")
        res = d_res.cpu().numpy().view(res_t)
        self.launch_results = res
        ok = np.nonzero(res["rc"] == 0)[0]
        if len(ok) == 0:
            return 0
        # the k-th success of scenario s becomes list element n_s + k, in row s * P + n_s + k
        new_rows = np.empty(len(ok), np.int64)
        for s in range(S):
            mine = np.nonzero(scen[ok] == s)[0]
            assert self.n_s[s] + len(mine) <= P, "scenario out of rows: raise missile_capacity"
            new_rows[mine] = s * P + self.n_s[s] + np.arange(len(mine))
            self.row_of_list[s] = np.concatenate([self.row_of_list[s], new_rows[mine]])
            self.n_s[s] += len(mine)
        t0 = self.loop.time_ms / 1000
        lp = np.broadcast_to(np.asarray(launcher_pos, np.float64), (len(ok), 3))
        st.overwrite_rows(new_rows, lp, res["velocity"][ok], t0, kind=1)
        st.lib.zrk_ctx_invalidate_boxes(st.ctx.handle)
        st.add_missile_rows(new_rows.astype(np.int32), rows[ok].astype(np.int32), radius, period)
        self.launched += len(ok)
        return len(ok)

    # ---- ticks ----------------------------------------------------------------------------------------------
    def run(self, K, sweep_ms=None, prof_stride=1):
        st = self.store
        self.loop.cur = st.cur
        ms_ptr = sweep_ms.ctypes.data_as(C.POINTER(C.c_float)) if sweep_ms is not None else None
        st.ctx.check(st.lib.zrk_run_ticks_ensemble(
            st.ctx.handle, C.byref(st.ents), C.byref(st.mis), st.m, C.byref(self.loop), C.byref(self.ens),
            st.workspace().data_ptr(), self.det_idx.data_ptr(), self.det_stride, self.det_cnt.data_ptr(), int(K), ms_ptr,
            int(prof_stride), st._stream()), "zrk_run_ticks_ensemble")
        st.cur = int(self.loop.cur)
        st.vis_cur = int(self.loop.vis_cur)
        st.time_ms = int(self.loop.time_ms) - self.dt_ms
        st.n_stepped = st.n_uploaded
        st._bump()

    def read_sweep_ms(self, n):
        """Durations [ms] of the sweeps a run(..., prof_stride < 0) call recorded events around (waits for them)."""
        out = np.zeros(int(n), np.float32)
        st = self.store
        st.ctx.check(st.lib.zrk_read_sweep_ms(st.ctx.handle, out.ctypes.data_as(C.POINTER(C.c_float)), int(n)), "zrk_read_sweep_ms")
        return out

    def sweep_stamps(self, on=True):
        st = self.store
        st.ctx.check(st.lib.zrk_sweep_stamps(st.ctx.handle, 1 if on else 0), "zrk_sweep_stamps")

    def read_sweep_stamps(self, cap=64):
        us, ticks = np.zeros(int(cap), np.float32), np.zeros(int(cap), np.int32)
        st = self.store
        k = st.lib.zrk_read_sweep_stamps(st.ctx.handle, us.ctypes.data_as(C.POINTER(C.c_float)),
                                         ticks.ctypes.data_as(C.POINTER(C.c_int32)), int(cap), st._stream())
        if k < 0:
            st.ctx.check(k, "zrk_read_sweep_stamps")
        return us[:k], ticks[:k]

    # ---- results (each synchronises) ------------------------------------------------------------------------
    def alive_count(self):
        return int(self.store.d_alive[:self.S * self.P].sum().item())

    def detections(self, s):
        """Per-radar detection lists of scenario s (scenario-local list indices, ascending)."""
        self.store.compact_status()
        R, P = self.R, self.P
        cnt = self.det_cnt[s * (R + 1):s * (R + 1) + R].cpu().numpy()
        out = []
        for r in range(R):
            lo = (s * R + r) * self.det_stride
            out.append(self.det_idx[lo:lo + min(int(cnt[r]), self.det_stride)].cpu().numpy())
        return out

    def radar_state(self, s):
        raw = self.d_radars.cpu().numpy().view(np.float64).reshape(self.S, max(self.R, 1), 8)
        return [(float(raw[s, r, 4]), float(raw[s, r, 6])) for r in range(self.R)]

    def scenario_view(self, s):
        """What tests' OracleMirror reads of a single-scenario engine, for scenario s of the batch (rows renumbered
        from the scenario's first row)."""
        st = self.store
        P, lo = self.P, s * self.P
        n = int(self.n_s[s])
        mine = np.nonzero((st.hm_slot[:st.m] >= lo) & (st.hm_slot[:st.m] < lo + P))[0]
        rows = np.arange(lo, lo + n)
        store = SimpleNamespace(
            n_uploaded=n, h_lidx=st.h_lidx[rows], h_sp=st.h_sp[rows], h_vel=st.h_vel[rows], h_t0=st.h_t0[rows],
            h_pos0=st.h_pos0[rows], h_kind=st.h_kind[rows], m=len(mine), hm_slot=st.hm_slot[mine] - lo,
            hm_tgt=st.hm_tgt[mine] - lo, dm_radius=st.dm_radius[torch.as_tensor(mine, device=st.device)],
            dm_period=st.dm_period[torch.as_tensor(mine, device=st.device)], ctx=st.ctx, lib=st.lib, device=st.device)
        rol = self.row_of_list[s] - lo
        return SimpleNamespace(store=store, seed=int(self.seeds[s]), gid0=0, missile_rows=mine,
                               list_view=lambda a, rol=rol: a[rol])
