"""Engine-level driver of the hot path: the reference's L1 loop without Python objects per entity.

`DeviceSim` exposes the L1 boundary the way the reference's modules use it (one call per
AirEnv.step / SectorRadar.step, launch and new-missile events in between) and is what the
parity tests replay golden fixtures through.  The drop-in `AirEnv` / `SectorRadar` / `Missile`
classes in zrk_modulation_amd.modules are thin message-bus adapters over the same store.
"""
from __future__ import annotations

import numpy as np
import torch

from ._lib import F_ADVANCE, F_EXACT_ONLY, F_PHILOX
from .store import EntityStore

SCAN_HORIZONTAL, SCAN_VERTICAL, SCAN_OTHER = 0, 1, 2
_SCAN_CODES = {"horizontal": SCAN_HORIZONTAL, "vertical": SCAN_VERTICAL}


def scan_mode_code(name):
    return _SCAN_CODES.get(name, SCAN_OTHER)


def scan_next(mode, az_range, az_speed, el_speed, el_start, caz, cel):
    """SectorRadar.move_to_next_sector_circular (reference modules/Radar.py:96-117) as a pure
    function of the scan state; `%` is Python's float floor-mod, as there.  Quirks kept: on wrap the
    azimuth restarts from elevation_start (:105); an unknown mode string never moves."""
    if mode == SCAN_HORIZONTAL:
        caz = (caz + az_speed) % 360 if caz + az_range < 360 else el_start
        if caz < az_speed:
            cel = (cel + el_speed) % 90 if cel + el_speed < 90 else el_start
    elif mode == SCAN_VERTICAL:
        cel = (cel + el_speed) % 90
        if cel < el_speed:
            caz = (caz + az_speed) % 360
    return caz, cel


class DeviceSim:
    """L1 boundary as the reference's modules use it; the surface tests/helpers.replay_l1 drives."""

    def __init__(self, dt_ms, device=None, capacity=1024, missile_capacity=64, exact_only=False):
        self.store = EntityStore(device, capacity, missile_capacity)
        self.dt_ms = int(dt_ms)
        self.time_ms = 0
        self.radars = []
        self.mdef = {}                 # missile id -> definition / launch state (host; event-rate)
        self._pending_kill = []
        self._pending_new = []
        self._xflag = F_EXACT_ONLY if exact_only else 0

    # construction ---------------------------------------------------------------------------
    def add_target(self, id, start_pos, velocity, start_time=0.0, pos=None):
        return self.store.add_entities([id], [start_pos], [velocity], start_time, kind=0,
                                       pos0=None if pos is None else [pos])

    def add_radar(self, id, pos, azimuth_start, elevation_start, max_distance, azimuth_range, elevation_range,
                  azimuth_speed, elevation_speed, scan_mode="horizontal"):
        self.radars.append(dict(id=id, pos=np.asarray(pos, np.float64), az_start=azimuth_start,
                                el_start=elevation_start, max_distance=max_distance, az_range=azimuth_range,
                                el_range=elevation_range, az_speed=azimuth_speed, el_speed=elevation_speed,
                                mode=scan_mode_code(scan_mode), caz=azimuth_start, cel=elevation_start))

    def add_missile(self, id, pos, velocity_module, detonate_radius, detonate_period):
        self.mdef[int(id)] = dict(id=int(id), pos=np.asarray(pos, np.float64), v0=velocity_module,
                                  radius=detonate_radius, period=detonate_period, traj=None, target=None)

    # views ------------------------------------------------------------------------------------
    @property
    def ids(self):
        return self.store.h_ids

    @property
    def slot_of_id(self):
        return self.store.slots_of_id

    def active_slots(self):
        return np.nonzero(self.store.h_alive[:self.store.n])[0]

    def pos_of(self, slots):
        return self.store.host_pos("cur")[np.asarray(slots, np.int64)]

    def prev_of(self, slots):
        return self.store.host_pos("prev")[np.asarray(slots, np.int64)]

    def prev_valid_of(self, slots):
        st = self.store
        slots = np.asarray(slots, np.int64)
        return (slots < st.n_stepped) & (st.h_t0[slots] != st.time_ms / 1000)

    # L2 events ----------------------------------------------------------------------------------
    def launch(self, missile_id, target_slot):
        m = self.mdef[int(missile_id)]
        rc, V, t = self.store.launch_solve(target_slot, m["pos"], m["v0"], m["period"])
        if rc == 0:
            m["traj"] = (V, m["pos"].copy(), self.time_ms / 1000)
            m["target"] = int(target_slot)
        return rc, V, t

    def announce_missile(self, missile_id):
        self._pending_new.append(int(missile_id))

    # ticks --------------------------------------------------------------------------------------
    def airenv_step(self):
        st = self.store
        st.flush()
        kill = []
        for ms, ts in self._pending_kill:
            for s in (ms, ts):
                if s >= 0:
                    kill += st.slots_for_id(st.h_ids[s])
        st.kill(kill)
        self._pending_kill = []
        for mid in self._pending_new:
            m = self.mdef[mid]
            V, sp, t0 = m["traj"]
            slot = st.add_entities([mid], [sp], [V], t0, kind=1, pos0=[m["pos"]])
            st.add_missile_row(slot, m["target"], m["radius"], m["period"])
        self._pending_new = []
        st.begin_tick(self.time_ms)
        events = st.missile_step(self.dt_ms)
        st.sweep([], F_ADVANCE)
        self._pending_kill = list(events)
        return [(a, b, b < 0) for a, b in events]

    def radar_params(self, rd):
        return (rd["pos"][0], rd["pos"][1], rd["pos"][2], rd["max_distance"], rd["caz"], rd["az_range"], rd["cel"],
                rd["el_range"])

    def radar_step(self, k, noise_fn=None):
        st = self.store
        rd = self.radars[k]
        st.sweep([self.radar_params(rd)], self._xflag)
        det, cnts = st.compact(1)
        cnt = int(cnts[0].item())
        found = det[:cnt].cpu().numpy() if cnt else np.zeros(0, np.int32)
        if noise_fn is not None and cnt:
            st.noise_apply(det, cnt, noise_fn(cnt))
        rd["caz"], rd["cel"] = scan_next(rd["mode"], rd["az_range"], rd["az_speed"], rd["el_speed"], rd["el_start"],
                                         rd["caz"], rd["cel"])
        return found

    def end_tick(self):
        self.time_ms += self.dt_ms


def morton_order_xy(start_pos, bits=12):
    """Row order that keeps neighbours in (x, y) together: argsort of the interleaved cell bits."""
    xy = np.asarray(start_pos, np.float64)[:, :2]
    lo, hi = xy.min(0), xy.max(0)
    q = ((xy - lo) / np.maximum(hi - lo, 1e-9) * ((1 << bits) - 1)).astype(np.uint64)
    code = np.zeros(len(xy), np.uint64)
    for b in range(bits):
        code |= ((q[:, 0] >> np.uint64(b)) & np.uint64(1)) << np.uint64(2 * b)
        code |= ((q[:, 1] >> np.uint64(b)) & np.uint64(1)) << np.uint64(2 * b + 1)
    return np.argsort(code, kind="stable")


class HotPathEngine:
    """Headless L1 loop for large scenes: the whole tick stays on the device (events, tombstones,
    compaction) and `run(K)` enqueues K ticks through one C call (zrk_run_ticks).

    noise: "philox" = counter-based measurement noise inside the sweep (throughput mode; same
    distribution as the reference's np.random.normal(0, 5, 3), not the same stream), "off" = none.
    """

    def __init__(self, device=None, dt_ms=10, seed=0, noise="philox", gid0=0):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        self.store = None
        self.device = device
        self.dt_ms = int(dt_ms)
        self.seed = int(seed)
        self.noise = noise
        self.gid0 = int(gid0)
        self.loop = None
        self.R = 0

    def load(self, ids, start_pos, velocity, start_time, radars, missile_capacity=0, det_stride=None,
             union_capacity=None, sort=True, union_format="pairs"):
        """ids/start_pos/velocity/start_time: target columns in LIST order; radars: list of dicts with
        the SectorRadar constructor fields (reference modules/Radar.py:13-42).

        sort=True stores the rows in Morton order of their (x, y) start position: consecutive rows
        are neighbours in space, so the 64 lanes of a wave mostly agree on whether a radar sees them
        and the divergent parts of the sweep (noise draw, binary64 fallback) run for few waves.
        Every observable stays in list order (detection lists, noise keys, event order): the table
        carries a list_index column (include/zrk_hot.h)."""
        C, _lib = self._C, self._lib
        n = len(ids)
        ids = np.asarray(ids); start_pos = np.asarray(start_pos, np.float64).reshape(n, 3)
        velocity = np.asarray(velocity, np.float64).reshape(n, 3)
        start_time = np.broadcast_to(np.asarray(start_time, np.float64), (n,))
        st = self.store = EntityStore(self.device, n + missile_capacity, max(missile_capacity, 64))
        st.two_vis = True
        st._alloc_entities(st.cap)
        self.n_list = n                                  # length of AirEnv's list so far
        if sort and n > 1:
            order = morton_order_xy(start_pos)
            st.add_entities(ids[order], start_pos[order], velocity[order], start_time[order], kind=0,
                            list_index=order.astype(np.int32))
            self.row_of_list = np.empty(n, np.int64)
            self.row_of_list[order] = np.arange(n)
        else:
            st.add_entities(ids, start_pos, velocity, start_time, kind=0)
            self.row_of_list = None
        st.flush()
        self.R = R = len(radars)
        self.c_radars = (_lib.ZrkRadar * max(R, 1))()
        self.c_scan = (_lib.ZrkScan * max(R, 1))()
        for k, rd in enumerate(radars):
            cr, cs = self.c_radars[k], self.c_scan[k]
            cr.pos[0], cr.pos[1], cr.pos[2] = (float(v) for v in rd["position"])
            cr.max_distance = float(rd["max_distance"])
            cr.cur_azimuth, cr.azimuth_range = float(rd["azimuth_start"]), float(rd["azimuth_range"])
            cr.cur_elevation, cr.elevation_range = float(rd["elevation_start"]), float(rd["elevation_range"])
            cs.azimuth_speed, cs.elevation_speed = float(rd["azimuth_speed"]), float(rd["elevation_speed"])
            cs.elevation_start = float(rd["elevation_start"])
            cs.mode = scan_mode_code(rd.get("scan_mode", "horizontal"))
        self.loop = _lib.ZrkLoop()
        self.loop.n = st.n_uploaded
        self.loop.time_ms, self.loop.dt_ms = 0, self.dt_ms
        self.loop.gid0, self.loop.seed, self.loop.tick = self.gid0, self.seed, 0
        self.loop.cur, self.loop.base_index = st.cur, 0
        self.loop.flags = F_PHILOX if self.noise == "philox" else 0
        self.det_stride = int(det_stride or 0)
        self.det_idx = None
        self.det_cnt = torch.zeros(_lib.ZRK_MAX_RADARS + 1, dtype=torch.int32, device=st.device)
        self.packed = None
        self.union_format = union_format
        if union_capacity:
            # "pairs": [count, (global index << 32 | mask) ...]; "bits": the wire format of zrk_compact_bits
            words = (int(union_capacity) + 1 if union_format == "pairs"
                     else int(st.lib.zrk_union_bits_words(st.cap, self.R, int(union_capacity))))
            self.packed = torch.zeros(words, dtype=torch.int64, device=st.device)
            if union_format == "bits":
                self.loop.flags |= _lib.F_UNION_BITS
        return self

    def enable_lists(self, det_stride=None):
        """Per-radar detection lists: R segments of `det_stride` entries (default: the table capacity)."""
        st = self.store
        self.det_stride = int(det_stride or self.det_stride or st.cap)
        self.det_idx = torch.zeros(self.det_stride * max(self.R, 1), dtype=torch.int32, device=st.device)
        return self

    def launch_missiles(self, target_slots, launcher_pos=(0.0, 0.0, 0.0), speed=1000.0, radius=150.0, period=60.0,
                        id0=10_000_000, mirror=True):
        """Batched Missile._launch at the current time against the given targets (list indices); the ones whose
        solve succeeds enter the table as active missiles, in request order.  Solve AND append run on the device
        (zrk_launch_salvo): nothing is uploaded but the requests.  With mirror=True (default) the results are read
        back once for the host-side mirrors (ids, trajectories, row map) and the table grows by the number of
        successes, which is returned; with mirror=False nothing is read back, the table grows by len(target_slots)
        (the failed requests' rows sit dead behind the successes) and None is returned."""
        C, _lib = self._C, self._lib
        st = self.store
        k = len(target_slots)
        if k == 0:
            return 0
        target_slots = np.asarray(target_slots, np.int64)
        if self.row_of_list is not None:                 # callers speak list indices, the table speaks rows
            target_slots = self.row_of_list[target_slots]
        req_t, res_t = _lib.launch_dtypes()
        req = np.zeros(k, dtype=req_t)
        req["target_slot"] = np.asarray(target_slots, np.int32)
        req["missile_pos"] = np.asarray(launcher_pos, np.float64)
        req["speed"], req["period"], req["radius"] = speed, period, radius
        st.flush()
        if st.n + k > st.cap:
            st._alloc_entities(max(st.n + k, 2 * st.cap))
            self.loop.n = st.n_uploaded
        if st.m + k > st.mcap:
            st._alloc_missiles(max(2 * st.mcap, st.m + k))
        d_req = torch.from_numpy(req.view(np.uint8).reshape(-1)).to(st.device)
        d_res = torch.zeros(k * C.sizeof(_lib.ZrkLaunchRes), dtype=torch.uint8, device=st.device)
        n0, m0, list_base = st.n, st.m, self.n_list
        st.ctx.check(st.lib.zrk_launch_salvo(st.ctx.handle, C.byref(st.ents), st.cur, C.byref(st.mis), n0, m0, d_req.data_ptr(),
                                             d_res.data_ptr(), k, int(self.loop.time_ms), list_base, None, st._stream()),
                     "zrk_launch_salvo")
        st._bump()
        if not mirror:
            st.adopt_device_rows(k, kind=1)
            st.m += k
            self.n_list += k
            if self.row_of_list is not None:
                self.row_of_list = np.concatenate([self.row_of_list, n0 + np.arange(k)])
            self.loop.n = st.n_uploaded
            self.launch_results = None
            return None
        res = d_res.cpu().numpy().view(res_t)
        self.launch_results = res
        ok = np.nonzero(res["rc"] == 0)[0]
        c = len(ok)
        if c == 0:
            return 0
        t0 = self.loop.time_ms / 1000
        li = None
        if self.row_of_list is not None:
            li = np.arange(list_base, list_base + c, dtype=np.int32)
            self.row_of_list = np.concatenate([self.row_of_list, n0 + np.arange(c)])
        self.n_list += c
        st.adopt_device_rows(c, kind=1, ids=id0 + ok, start_pos=np.broadcast_to(np.asarray(launcher_pos, np.float64), (c, 3)),
                             velocity=res["velocity"][ok], start_time=t0, list_index=li)
        st.adopt_device_missile_rows(np.arange(n0, n0 + c, dtype=np.int32), np.asarray(target_slots, np.int32)[ok])
        self.loop.n = st.n_uploaded
        return c

    def launch_requests_on_device(self, d_req, k):
        """zrk_launch_salvo for k requests that already are on the device (a uint8 tensor of zrk_launch_req, e.g. from
        association.DeviceCommandPost.requests: target ROWS, padded with requests for no row): nothing is uploaded, nothing read
        back; the table grows by k rows (the failed and the padding requests' rows sit dead behind the successes).  Returns
        the int32 device tensor [1] that receives the number of missiles that entered the air.

        The dead rows do not pile up from call to call: the NEXT call (or settle_device_launches()) reads this call's count
        -- long written by then, so nothing waits -- and gives the rows behind the successes back, provided nothing else
        was appended in between."""
        C, _lib = self._C, self._lib
        st = self.store
        k = int(k)
        self.settle_device_launches()
        count = torch.zeros(1, dtype=torch.int32, device=st.device)
        if k == 0:
            return count
        assert d_req.numel() >= k * C.sizeof(_lib.ZrkLaunchReq)
        st.flush()
        if st.n + k > st.cap:
            st._alloc_entities(max(st.n + k, 2 * st.cap))
            self.loop.n = st.n_uploaded
        if st.m + k > st.mcap:
            st._alloc_missiles(max(2 * st.mcap, st.m + k))
        d_res = torch.zeros(k * C.sizeof(_lib.ZrkLaunchRes), dtype=torch.uint8, device=st.device)
        n0, list_base = st.n, self.n_list
        st.ctx.check(st.lib.zrk_launch_salvo(st.ctx.handle, C.byref(st.ents), st.cur, C.byref(st.mis), n0, st.m, d_req.data_ptr(),
                                             d_res.data_ptr(), k, int(self.loop.time_ms), list_base, count.data_ptr(), st._stream()),
                     "zrk_launch_salvo")
        st._bump()
        st.adopt_device_rows(k, kind=1)
        st.m += k
        self.n_list += k
        if self.row_of_list is not None:
            self.row_of_list = np.concatenate([self.row_of_list, n0 + np.arange(k)])
        self.loop.n = st.n_uploaded
        self.launch_results = None
        self._last_device_results = d_res
        self._pending_launch = (n0, st.m - k, list_base, k, count)
        return count

    def settle_device_launches(self):
        """The rows the last launch_requests_on_device call took for requests that did not put a missile in the air (failed
        solves, padding) are given back: its count is read (one int32) and table, missile table and list shrink to the
        successes.  Nothing happens when other rows were appended behind that call's."""
        pend, self._pending_launch = getattr(self, "_pending_launch", None), None
        if pend is None:
            return 0
        st = self.store
        n0, m0, list_base, k, count = pend
        if st.n != n0 + k or st.n_uploaded != st.n or st.m != m0 + k or self.n_list != list_base + k:
            return 0
        j = k - int(count.item())
        if j <= 0:
            return 0
        st.drop_tail_rows(j)
        st.m -= j
        self.n_list -= j
        if self.row_of_list is not None:
            self.row_of_list = self.row_of_list[:-j]
        self.loop.n = st.n_uploaded
        return j

    def run(self, K, sweep_ms=None, prof_stride=1, exchange=None):
        """Enqueue K ticks.  With `sweep_ms` (a float32 numpy array of ceil(K/prof_stride)) the call
        also times the sweep kernel with HIP events and synchronises the stream.  `exchange` (an
        exchange.RcclExchange): every tick's union list goes through its all-gather, issued from the C side."""
        C = self._C
        st = self.store
        self.loop.cur = st.cur
        ms_ptr = sweep_ms.ctypes.data_as(C.POINTER(C.c_float)) if sweep_ms is not None else None
        det = (self.det_idx.data_ptr() if self.det_idx is not None else None,
               self.det_stride if self.det_idx is not None else 0, self.det_cnt.data_ptr())
        try:
            if exchange is not None:
                self.loop.flags |= self._lib.F_UNION_BITS
                st.ctx.check(st.lib.zrk_run_ticks_x(
                    st.ctx.handle, C.byref(st.ents), C.byref(st.mis), st.m, C.byref(self.loop), self.c_radars, self.c_scan,
                    self.R, st.workspace().data_ptr(), det[0], det[1], det[2], None, 0, C.byref(exchange.io), int(K), ms_ptr,
                    int(prof_stride), st._stream()), "zrk_run_ticks_x")
            else:
                st.ctx.check(st.lib.zrk_run_ticks(
                    st.ctx.handle, C.byref(st.ents), C.byref(st.mis), st.m, C.byref(self.loop), self.c_radars, self.c_scan,
                    self.R, st.workspace().data_ptr(), det[0], det[1], det[2],
                    self.packed.data_ptr() if self.packed is not None else None,
                    self.packed.numel() if self.packed is not None else 0, int(K), ms_ptr, int(prof_stride), st._stream()),
                    "zrk_run_ticks")
        finally:
            # (also when the call failed: the loop state says how many ticks were swept, and the table stands as after those)
            st.cur = int(self.loop.cur)
            st.vis_cur = int(self.loop.vis_cur)
            st.time_ms = int(self.loop.time_ms) - self.dt_ms
            st.n_stepped = st.n_uploaded
            st._bump()

    def read_sweep_ms(self, n):
        """Durations [ms] of the sweeps a run(..., prof_stride < 0) call recorded events around (waits for them)."""
        C = self._C if hasattr(self, "_C") else __import__("ctypes")
        out = np.zeros(int(n), np.float32)
        st = self.store
        st.ctx.check(st.lib.zrk_read_sweep_ms(st.ctx.handle, out.ctypes.data_as(C.POINTER(C.c_float)), int(n)), "zrk_read_sweep_ms")
        return out

    def sweep_stamps(self, on=True):
        """The sweeps of the following run() calls time themselves (zrk_sweep_stamps, include/zrk_hot.h)."""
        st = self.store
        st.ctx.check(st.lib.zrk_sweep_stamps(st.ctx.handle, 1 if on else 0), "zrk_sweep_stamps")

    def read_sweep_stamps(self, cap=64):
        """(durations [us], ticks swept) of the sampled sweep launches of the last run() call, from the launches' own
        wall-clock stamps (synchronises the stream)."""
        C = self._C
        us, ticks = np.zeros(int(cap), np.float32), np.zeros(int(cap), np.int32)
        st = self.store
        k = st.lib.zrk_read_sweep_stamps(st.ctx.handle, us.ctypes.data_as(C.POINTER(C.c_float)),
                                         ticks.ctypes.data_as(C.POINTER(C.c_int32)), int(cap), st._stream())
        if k < 0:
            st.ctx.check(k, "zrk_read_sweep_stamps")
        return us[:k], ticks[:k]

    def sweep_stamp_times(self, cap=64):
        """(first wave in [us], last wave out [us]) of the launches the last read_sweep_stamps() sampled, from the first one's
        first wave on, by the device's clock: the span of a call's sweeps and the gaps between them."""
        C = self._C
        b, e = np.zeros(int(cap), np.float64), np.zeros(int(cap), np.float64)
        st = self.store
        k = st.lib.zrk_last_sweep_stamp_times(st.ctx.handle, b.ctypes.data_as(C.POINTER(C.c_double)), e.ctypes.data_as(C.POINTER(C.c_double)), int(cap))
        if k < 0:
            st.ctx.check(k, "zrk_last_sweep_stamp_times")
        return b[:k], e[:k]

    def read_sweep_ticks(self, n):
        """Ticks swept by the launch behind each timing sample of the last run (1, or 2 for a pair launch)."""
        C = self._C
        out = np.zeros(int(n), np.int32)
        st = self.store
        st.ctx.check(st.lib.zrk_read_sweep_ticks(st.ctx.handle, out.ctypes.data_as(C.POINTER(C.c_int32)), int(n)), "zrk_read_sweep_ticks")
        return out

    # results (each synchronises) ---------------------------------------------------------------
    def alive_count(self):
        return int(self.store.d_alive[:self.store.n_uploaded].sum().item())

    def detections(self):
        self.store.compact_status()                # a compaction that did not run to completion left wrong lists: raise
        cnt = self.det_cnt[:self.R].cpu().numpy()
        idx = self.det_idx.cpu().numpy() if cnt.max(initial=0) * 4 > self.det_stride else None
        out = []
        for r in range(self.R):
            lo, k = r * self.det_stride, min(int(cnt[r]), self.det_stride)
            out.append(idx[lo:lo + k] if idx is not None else self.det_idx[lo:lo + k].cpu().numpy())
        return out

    def list_view(self, rows_first):
        """Reorder a per-row array (first axis = table rows) into list order."""
        return rows_first if self.row_of_list is None else rows_first[self.row_of_list]

    def radar_state(self):
        return [(self.c_radars[r].cur_azimuth, self.c_radars[r].cur_elevation) for r in range(self.R)]
