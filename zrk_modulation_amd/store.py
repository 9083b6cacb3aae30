"""EntityStore: AirEnv's object list as structure-of-arrays float64 tensors in HBM, plus the
thin calls into libzrk_hot.so that act on it.

Layout (see DESIGN.md "Data layout in HBM"):
    start_pos, velocity          float64 [3][cap]   Trajectory columns (reference modules/AirObject.py:19-21)
    start_time                   float64 [cap]
    alive, kind                  uint8   [cap]      tombstones (modules/AirEnv.py:39-40); 0 target / 1 missile
    pos                          float64 [2][3][cap] double buffer: pos[cur] is this tick's obj.pos,
                                                     pos[cur^1] what obj.prev_pos aliases (AirObject.py:41)
    vis_mask                     uint32  [cap]      bit r = detected by radar r
    missile table                rows in the order missiles entered the air (= slot order)
Slots are never reused, so slot order == the order of the reference's `self.__objects`.

PyTorch is only the allocator / stream owner here; every computation is a HIP kernel behind the
C ABI.  Constructing a store without a usable GPU + libzrk_hot.so raises HotPathUnavailable.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import (F_ADVANCE, F_EXACT_ONLY, F_PHILOX, HotPathUnavailable, ZrkEntities, ZrkLaunchReq,
                   ZrkLaunchRes, ZrkMissiles, ZrkRadar)

LAUNCH_ERRORS = {
    1: "No interception possible: target and interceptor are stationary relative or parallel.",
    2: "Interception impossible in the future: computed time t <= 0.",
    3: "No real interception time: target is too fast or out of range.",
    4: "Interception times are not positive; interception not possible in future.",
    5: "Target is too far for this rocket (detonation_period over limited)",
}


def _round_up(n, q):
    return ((int(n) + q - 1) // q) * q


def radar_struct_array(params):
    """params: iterable of (px,py,pz,max_distance,cur_az,az_range,cur_el,el_range) -> ZrkRadar[]"""
    params = list(params)
    arr = (ZrkRadar * max(len(params), 1))()
    for k, p in enumerate(params):
        arr[k].pos[0], arr[k].pos[1], arr[k].pos[2] = float(p[0]), float(p[1]), float(p[2])
        arr[k].max_distance = float(p[3])
        arr[k].cur_azimuth, arr[k].azimuth_range = float(p[4]), float(p[5])
        arr[k].cur_elevation, arr[k].elevation_range = float(p[6]), float(p[7])
    return arr


class EntityStore:
    def __init__(self, device=None, capacity: int = 1024, missile_capacity: int = 64):
        if not torch.cuda.is_available():
            raise HotPathUnavailable("no HIP device visible: the hot path has no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise HotPathUnavailable(f"EntityStore needs a HIP device, got {self.device}")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.ctx = _lib.Context(idx)
        self.lib = self.ctx.lib
        self.cap = 0
        self.mcap = 0
        self.n = 0                 # slots in use
        self.n_uploaded = 0
        self.n_stepped = 0         # slots that existed at the last advance
        self.m = 0                 # in-flight missile rows
        self.cur = 0
        self.time_ms = None        # time of the last advance
        self.version = 0           # bumped whenever device positions change
        self.rows_version = 0      # bumped whenever table rows are given out again (drop_tail_rows, overwrite_rows): per-row caches start afresh
        self._snap = {}
        # host mirrors of the immutable columns (what the handles expose as .trajectory etc.)
        self.h_ids = np.zeros(0, np.int64)
        self.h_kind = np.zeros(0, np.uint8)
        self.h_sp = np.zeros((0, 3)); self.h_vel = np.zeros((0, 3)); self.h_t0 = np.zeros(0)
        self.h_pos0 = np.zeros((0, 3))
        self.h_alive = np.zeros(0, np.uint8)
        self.two_vis = False       # hand zrk_run_ticks both mask buffers (HotPathEngine turns this on)
        self.vis_cur = 0
        self.h_lidx = None         # list index of each row when rows are not stored in list order
        self.d_lidx = None
        self.slots_of_id = {}
        # host mirror of the missile table's static columns
        self.hm_slot = np.zeros(0, np.int32); self.hm_tgt = np.zeros(0, np.int32)
        self._alloc_entities(max(int(capacity), 256))
        self._alloc_missiles(max(int(missile_capacity), 64))
        self._ws = None
        self._det_idx = None
        self._det_cnt = torch.zeros(_lib.ZRK_MAX_RADARS + 1, dtype=torch.int32, device=self.device)

    # -- allocation -------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _alloc_entities(self, cap):
        cap = _round_up(cap, 256)
        dev, f64 = self.device, torch.float64
        new = dict(sp=torch.zeros(3, cap, dtype=f64, device=dev), vel=torch.zeros(3, cap, dtype=f64, device=dev),
                   t0=torch.zeros(cap, dtype=f64, device=dev), alive=torch.zeros(cap, dtype=torch.uint8, device=dev),
                   kind=torch.zeros(cap, dtype=torch.uint8, device=dev),
                   pos=torch.zeros(2, 3, cap, dtype=f64, device=dev),
                   vis=torch.zeros(cap, dtype=torch.int32, device=dev),
                   vis_alt=torch.zeros(cap, dtype=torch.int32, device=dev))
        if self.cap:
            k = self.n_uploaded
            for name, t in new.items():
                old = getattr(self, "d_" + name)
                t[..., :k] = old[..., :k]
        for name, t in new.items():
            setattr(self, "d_" + name, t)
        self.cap = cap
        e = ZrkEntities()
        e.capacity = cap
        e.start_pos, e.velocity, e.start_time = self.d_sp.data_ptr(), self.d_vel.data_ptr(), self.d_t0.data_ptr()
        e.alive, e.kind = self.d_alive.data_ptr(), self.d_kind.data_ptr()
        e.pos[0], e.pos[1] = self.d_pos[0].data_ptr(), self.d_pos[1].data_ptr()
        e.vis_mask = self.d_vis.data_ptr()
        self.vis_cur = 0
        if self.two_vis:
            e.vis_mask_alt = self.d_vis_alt.data_ptr()
        if self.h_lidx is not None:
            old = self.d_lidx
            self.d_lidx = torch.zeros(cap, dtype=torch.int32, device=dev)
            if old is not None:
                self.d_lidx[:self.n_uploaded] = old[:self.n_uploaded]
            e.list_index = self.d_lidx.data_ptr()
        self.ents = e
        self._ws = None
        self._det_idx = None

    def _alloc_missiles(self, mcap):
        mcap = _round_up(mcap, 64)
        dev = self.device
        new = dict(slot=torch.zeros(mcap, dtype=torch.int32, device=dev),
                   tgt=torch.zeros(mcap, dtype=torch.int32, device=dev),
                   radius=torch.zeros(mcap, dtype=torch.float64, device=dev),
                   period=torch.zeros(mcap, dtype=torch.float64, device=dev),
                   status=torch.zeros(mcap, dtype=torch.uint8, device=dev),
                   evcode=torch.zeros(mcap, dtype=torch.uint8, device=dev),
                   evm=torch.zeros(mcap, dtype=torch.int32, device=dev),
                   evt=torch.zeros(mcap, dtype=torch.int32, device=dev),
                   evn=torch.zeros(1, dtype=torch.int32, device=dev))
        if self.mcap:
            for name, t in new.items():
                if name != "evn":
                    t[:self.m] = getattr(self, "dm_" + name)[:self.m]
        for name, t in new.items():
            setattr(self, "dm_" + name, t)
        self.mcap = mcap
        ms = ZrkMissiles()
        ms.capacity = mcap
        ms.slot, ms.target = self.dm_slot.data_ptr(), self.dm_tgt.data_ptr()
        ms.radius, ms.period, ms.status = self.dm_radius.data_ptr(), self.dm_period.data_ptr(), self.dm_status.data_ptr()
        ms.ev_code, ms.ev_missile, ms.ev_target = self.dm_evcode.data_ptr(), self.dm_evm.data_ptr(), self.dm_evt.data_ptr()
        ms.ev_count = self.dm_evn.data_ptr()
        self.mis = ms

    def workspace(self):
        need = int(self.lib.zrk_workspace_bytes(self.cap))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def det_buffer(self, entries):
        if self._det_idx is None or self._det_idx.numel() < entries:
            self._det_idx = torch.zeros(max(int(entries), 256), dtype=torch.int32, device=self.device)
        return self._det_idx

    # -- population -------------------------------------------------------------------------
    def _happend(self, name, vals):
        """Append to a host mirror column in amortised O(1) per row (the public attribute is a view of a buffer that
        doubles): a scenario loaded one object at a time stays linear in the number of objects."""
        vals = np.asarray(vals)
        bufs = self.__dict__.setdefault("_hbuf", {})
        cur = getattr(self, name)
        n = len(cur)
        buf = bufs.get(name)
        if buf is None or buf.dtype != cur.dtype or not (n == 0 or np.shares_memory(buf, cur)):
            buf = cur                                   # somebody rebound the attribute: start from what it holds
        need = n + len(vals)
        if need > len(buf):
            grown = np.empty((max(need, 2 * len(buf), 1024),) + cur.shape[1:], cur.dtype)
            grown[:n] = cur
            buf = grown
        buf[n:need] = vals
        bufs[name] = buf
        setattr(self, name, buf[:need])

    def add_entities(self, ids, start_pos, velocity, start_time, kind=0, pos0=None, list_index=None):
        """Append rows (host side); they reach the device at the next flush().  `list_index` (all rows
        or none, for the whole life of the store) says where each row sits in AirEnv's list when the
        rows are stored in another order."""
        sp = np.asarray(start_pos, np.float64).reshape(-1, 3)
        k = sp.shape[0]
        vel = np.asarray(velocity, np.float64).reshape(k, 3)
        t0 = np.broadcast_to(np.asarray(start_time, np.float64), (k,))
        ids = np.broadcast_to(np.asarray(ids, np.int64), (k,))
        p0 = sp if pos0 is None else np.asarray(pos0, np.float64).reshape(k, 3)
        first = self.n
        self._happend("h_ids", ids); self._happend("h_kind", np.full(k, kind, np.uint8))
        self._happend("h_sp", sp); self._happend("h_vel", vel); self._happend("h_t0", t0); self._happend("h_pos0", p0)
        self._happend("h_alive", np.ones(k, np.uint8))
        if list_index is not None:
            assert self.h_lidx is not None or first == 0, "list_index must be given for every row or for none"
            li = np.asarray(list_index, np.int32).reshape(k)
            if self.h_lidx is None:
                self.h_lidx = np.zeros(0, np.int32)
            self._happend("h_lidx", li)
        else:
            assert self.h_lidx is None, "list_index must be given for every row or for none"
        if self.slots_of_id is not None and k <= 4096:
            for j in range(k):
                self.slots_of_id.setdefault(int(ids[j]), []).append(first + j)
        else:
            self.slots_of_id = None        # bulk load: id lookups go through np.nonzero(h_ids == id)
        self.n += k
        return first

    def slots_for_id(self, id):
        if self.slots_of_id is not None:
            return self.slots_of_id.get(int(id), [])
        return np.nonzero(self.h_ids[:self.n] == id)[0].tolist()

    def flush(self):
        """Upload rows added since the last flush."""
        a, b = self.n_uploaded, self.n
        if a == b:
            return
        if b > self.cap or (self.h_lidx is not None and self.d_lidx is None):
            self._alloc_entities(max(b, 2 * self.cap if b > self.cap else self.cap))
        dev = self.device
        if self.h_lidx is not None:
            self.d_lidx[a:b] = torch.from_numpy(np.ascontiguousarray(self.h_lidx[a:b])).to(dev)
        self.d_sp[:, a:b] = torch.from_numpy(np.ascontiguousarray(self.h_sp[a:b].T)).to(dev)
        self.d_vel[:, a:b] = torch.from_numpy(np.ascontiguousarray(self.h_vel[a:b].T)).to(dev)
        self.d_t0[a:b] = torch.from_numpy(np.ascontiguousarray(self.h_t0[a:b])).to(dev)
        self.d_kind[a:b] = torch.from_numpy(np.ascontiguousarray(self.h_kind[a:b])).to(dev)
        self.d_alive[a:b] = 1
        p0 = torch.from_numpy(np.ascontiguousarray(self.h_pos0[a:b].T)).to(dev)
        self.d_pos[0][:, a:b] = p0
        self.d_pos[1][:, a:b] = p0          # both buffers: whichever is "previous" at its first step
        self.n_uploaded = b
        self._bump()

    def adopt_device_rows(self, k, kind=0, ids=None, start_pos=None, velocity=None, start_time=0.0, list_index=None):
        """The next k table rows were written on the device (zrk_launch_salvo): count them in; the host mirrors
        take what the caller knows of them (nothing is uploaded)."""
        z3 = np.zeros((k, 3))
        self._happend("h_ids", np.asarray(ids, np.int64) if ids is not None else np.full(k, -1, np.int64))
        self._happend("h_kind", np.full(k, kind, np.uint8))
        sp = np.asarray(start_pos, np.float64).reshape(k, 3) if start_pos is not None else z3
        self._happend("h_sp", sp)
        self._happend("h_vel", np.asarray(velocity, np.float64).reshape(k, 3) if velocity is not None else z3)
        self._happend("h_t0", np.broadcast_to(np.asarray(start_time, np.float64), (k,)))
        self._happend("h_pos0", sp)
        self._happend("h_alive", np.ones(k, np.uint8))
        if self.h_lidx is not None:
            li = np.asarray(list_index, np.int32) if list_index is not None else np.arange(self.n, self.n + k, dtype=np.int32)
            self._happend("h_lidx", li)
        self.slots_of_id = None
        self.n += k
        self.n_uploaded += k
        self._bump()

    def drop_tail_rows(self, j):
        """Forget the table's last j rows (dead rows behind a device-side salvo, engine.launch_requests_on_device): the next
        rows appended take their place.  The loop's per-row records of this table start afresh."""
        j = int(j)
        if j <= 0:
            return
        assert self.n == self.n_uploaded and j <= self.n
        for name in ("h_ids", "h_kind", "h_sp", "h_vel", "h_t0", "h_pos0", "h_alive") + (("h_lidx",) if self.h_lidx is not None else ()):
            setattr(self, name, getattr(self, name)[:-j])
        self.slots_of_id = None
        self.n -= j
        self.n_uploaded -= j
        self.n_stepped = min(self.n_stepped, self.n_uploaded)
        self.lib.zrk_ctx_invalidate_boxes(self.ctx.handle)
        self.rows_version += 1
        self._bump()

    def adopt_device_missile_rows(self, slots, target_slots):
        """Missile-table rows written on the device (zrk_launch_salvo): host mirrors of their static columns."""
        self.hm_slot = np.concatenate([self.hm_slot, np.asarray(slots, np.int32)])
        self.hm_tgt = np.concatenate([self.hm_tgt, np.asarray(target_slots, np.int32)])
        self.m += len(slots)

    def overwrite_rows(self, rows, start_pos, velocity, start_time, kind=0):
        """Give existing (padding) rows a trajectory and bring them to life, in place: what a batched ensemble does
        when a missile enters the air in a scenario's block of rows (the rows keep their list index).  Tells the loop
        that rows changed under it (zrk_ctx_invalidate_boxes)."""
        rows = np.asarray(rows, np.int64)
        k = len(rows)
        sp = np.asarray(start_pos, np.float64).reshape(k, 3); vel = np.asarray(velocity, np.float64).reshape(k, 3)
        t0 = np.broadcast_to(np.asarray(start_time, np.float64), (k,))
        self.flush()
        self.h_sp[rows] = sp; self.h_vel[rows] = vel; self.h_t0[rows] = t0; self.h_pos0[rows] = sp
        self.h_kind[rows] = kind; self.h_alive[rows] = 1
        self.rows_version += 1
        dev = self.device
        r = torch.as_tensor(rows, device=dev)
        dsp = torch.from_numpy(np.ascontiguousarray(sp.T)).to(dev)
        self.d_sp[:, r] = dsp
        self.d_vel[:, r] = torch.from_numpy(np.ascontiguousarray(vel.T)).to(dev)
        self.d_t0[r] = torch.from_numpy(np.ascontiguousarray(t0)).to(dev)
        self.d_kind[r] = int(kind)
        self.d_pos[0][:, r] = dsp
        self.d_pos[1][:, r] = dsp
        self.d_alive[r] = 1
        # rows changed under the loop: its box records and gather records of them are stale (the call is cheap, and a
        # caller that forgets it would get wrong detections, not an error)
        self.lib.zrk_ctx_invalidate_boxes(self.ctx.handle)
        self._bump()

    def add_missile_row(self, slot, target_slot, radius, period):
        if self.m + 1 > self.mcap:
            self._alloc_missiles(2 * self.mcap)
        r = self.m
        self.dm_slot[r] = int(slot); self.dm_tgt[r] = int(target_slot)
        self.dm_radius[r] = float(radius); self.dm_period[r] = float(period)
        self.dm_status[r] = 1
        self.hm_slot = np.append(self.hm_slot, np.int32(slot)); self.hm_tgt = np.append(self.hm_tgt, np.int32(target_slot))
        self.m += 1
        return r

    def add_missile_rows(self, slots, target_slots, radius, period):
        """Bulk form of add_missile_row (synthetic scenarios)."""
        k = len(slots)
        if self.m + k > self.mcap:
            self._alloc_missiles(max(2 * self.mcap, self.m + k))
        a, b = self.m, self.m + k
        dev = self.device
        self.dm_slot[a:b] = torch.as_tensor(np.asarray(slots, np.int32), device=dev)
        self.dm_tgt[a:b] = torch.as_tensor(np.asarray(target_slots, np.int32), device=dev)
        self.dm_radius[a:b] = torch.as_tensor(np.broadcast_to(np.asarray(radius, np.float64), (k,)).copy(), device=dev)
        self.dm_period[a:b] = torch.as_tensor(np.broadcast_to(np.asarray(period, np.float64), (k,)).copy(), device=dev)
        self.dm_status[a:b] = 1
        self.hm_slot = np.concatenate([self.hm_slot, np.asarray(slots, np.int32)])
        self.hm_tgt = np.concatenate([self.hm_tgt, np.asarray(target_slots, np.int32)])
        self.m = b

    # -- snapshots for host-side readers ----------------------------------------------------------
    def _bump(self):
        self.version += 1
        self._snap.clear()

    def vis(self):
        """The mask buffer holding the last tick's visibility (int32 view of the uint32 masks)."""
        return self.d_vis_alt if self.vis_cur else self.d_vis

    def host_pos(self, which="cur"):
        """(n,3) float64 copy of pos[cur] ('cur') or pos[cur^1] ('prev'); cached until the next kernel."""
        key = which
        if key not in self._snap:
            buf = self.cur if which == "cur" else self.cur ^ 1
            self._snap[key] = self.d_pos[buf][:, :self.n_uploaded].T.contiguous().cpu().numpy()
        return self._snap[key]

    def write_pos(self, slot, value, which="cur"):
        v = torch.as_tensor(np.asarray(value, np.float64).reshape(3), device=self.device)
        self.d_pos[self.cur if which == "cur" else self.cur ^ 1][:, int(slot)] = v
        self._bump()

    # -- kernels ------------------------------------------------------------------------------
    def kill(self, slots):
        """AirEnv tombstoning; `slots` are frozen at their current (pos[cur]) values."""
        slots = [int(s) for s in slots if self.h_alive[int(s)]]
        if not slots:
            return
        for s in slots:
            self.h_alive[s] = 0
        d = torch.as_tensor(np.asarray(slots, np.int32), device=self.device)
        self.ctx.check(self.lib.zrk_kill_slots(self.ctx.handle, C.byref(self.ents), self.cur, d.data_ptr(), len(slots),
                                               self._stream()), "zrk_kill_slots")
        self._bump()

    def begin_tick(self, time_ms):
        """Flip the position double buffer: last tick's final positions become 'previous'."""
        self.flush()
        self.cur ^= 1
        self.time_ms = int(time_ms)
        self._bump()

    def missile_step(self, dt_ms):
        """Missile.step for all in-flight rows.  Returns [(missile slot, target slot | -1)] in list order."""
        self.ctx.check(self.lib.zrk_missile_step(self.ctx.handle, C.byref(self.ents), self.cur, C.byref(self.mis), self.m,
                                                 self.time_ms, int(dt_ms), 0, self._stream()), "zrk_missile_step")
        if self.m == 0:
            return []
        k = int(self.dm_evn.item())
        if k == 0:
            return []
        em = self.dm_evm[:k].cpu().numpy(); et = self.dm_evt[:k].cpu().numpy()
        return [(int(a), int(b)) for a, b in zip(em, et)]

    def sweep(self, radars, flags, seed=0, tick=0, gid0=0, n=None):
        """zrk_tick_sweep over slots [0, n) for the given radar parameter tuples."""
        n = self.n_uploaded if n is None else n
        arr = radars if isinstance(radars, C.Array) else radar_struct_array(radars)
        R = len(radars)
        self.ctx.check(self.lib.zrk_tick_sweep(self.ctx.handle, C.byref(self.ents), n, self.cur, self.time_ms, arr, R,
                                               flags, seed, tick, gid0, self.workspace().data_ptr(),
                                               self._stream()), "zrk_tick_sweep")
        if flags & F_ADVANCE:
            self.n_stepped = n
        if flags & (F_ADVANCE | F_PHILOX):
            self._bump()
        return R

    def compact(self, R, base_index=0, n=None, det_stride=None):
        """zrk_compact after sweep(); returns (det_idx tensor [R][stride], det_cnt tensor [R+1]) on the device.
        Radar r's list is det_idx[r*stride : r*stride + det_cnt[r]] with stride = `det_stride` or n."""
        n = self.n_uploaded if n is None else n
        stride = n if det_stride is None else det_stride
        det = self.det_buffer(stride * max(R, 1))
        self.ctx.check(self.lib.zrk_compact(self.ctx.handle, self.d_vis.data_ptr(), n, R, base_index,
                                            self.workspace().data_ptr(), det.data_ptr(), stride,
                                            self._det_cnt.data_ptr(), None, 0, 0, self._stream()), "zrk_compact")
        return det, self._det_cnt

    def compact_status(self):
        """Synchronise and raise ZrkError if a compaction on this store's workspace did not run to completion."""
        self.ctx.check(self.lib.zrk_compact_status(self.ctx.handle, self.workspace().data_ptr(), self._stream()),
                       "zrk_compact_status")

    def noise_apply(self, det_idx, k, noise, idx_base=0):
        nz = torch.as_tensor(np.ascontiguousarray(noise, np.float64).reshape(k, 3), device=self.device)
        self.ctx.check(self.lib.zrk_noise_apply(self.ctx.handle, self.d_pos[self.cur].data_ptr(), self.cap,
                                                det_idx.data_ptr(), idx_base, nz.data_ptr(), k, self._stream()),
                       "zrk_noise_apply")
        self._bump()

    def launch_solve(self, target_slot, missile_pos, speed, period):
        """Missile._calculate_trajectory_params on the device.  Returns (rc, V[3], t_hit)."""
        req_t, res_t = _lib.launch_dtypes()
        req = np.zeros(1, dtype=req_t)
        req["target_slot"] = int(target_slot); req["missile_pos"] = np.asarray(missile_pos, np.float64)
        req["speed"] = float(speed); req["period"] = float(period)
        d_req = torch.from_numpy(req.view(np.uint8)).to(self.device)
        d_res = torch.zeros(C.sizeof(ZrkLaunchRes), dtype=torch.uint8, device=self.device)
        self.flush()
        self.ctx.check(self.lib.zrk_launch_solve(self.ctx.handle, C.byref(self.ents), self.cur, d_req.data_ptr(),
                                                 d_res.data_ptr(), 1, self._stream()), "zrk_launch_solve")
        res = d_res.cpu().numpy().view(res_t)[0]
        return int(res["rc"]), np.array(res["velocity"], np.float64), float(res["t_hit"])
