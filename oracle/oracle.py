"""ctypes front end of the CPU oracle (oracle/zrk_oracle.c) plus `OracleSim`, a
structure-of-arrays replay of the reference's L1 loop.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

`OracleSim` follows, tick by tick, what `Manager.run_simulation` makes the L1
modules do (reference modules/Manager.py:111-140):
    AirEnv.step           modules/AirEnv.py:26-53
    SectorRadar.step      modules/Radar.py:144-205   (one call per radar, in module order)
and accepts the two inbound events L2 produces for L1:
    launch command        modules/MissileLauncher.py:58-80 -> modules/Missile.py:153-160, :104-133
    new missile in air    modules/MissileLauncher.py:117-124 -> modules/AirEnv.py:42-43
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libzrk_oracle.so"


def build(force: bool = False) -> Path:
    """Compile the C restatement with the committed Makefile (gcc)."""
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < (_HERE / "zrk_oracle.c").stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B" if force else "-s", "libzrk_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class ZoRadar(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("px", "py", "pz", "max_distance", "caz", "az_range", "cel", "el_range")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        build()
    L = C.CDLL(str(_LIB_PATH))
    dp, u8p, i32p, u32p = (C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                           C.POINTER(C.c_uint32))
    i64 = C.c_int64
    L.zo_norm3.restype = C.c_double
    L.zo_norm3.argtypes = [C.c_double] * 3
    L.zo_floormod.restype = C.c_double
    L.zo_floormod.argtypes = [C.c_double] * 2
    L.zo_unit_velocity.restype = None
    L.zo_unit_velocity.argtypes = [dp, dp, dp]
    L.zo_airenv_step.restype = i64
    L.zo_airenv_step.argtypes = [i64, i64, i64, i64, dp, dp, dp, u8p, u8p, i32p, dp, dp, u8p,
                                 i32p, dp, dp, u8p, i32p, i32p, u8p]
    L.zo_airenv_step_mt.restype = i64
    L.zo_airenv_step_mt.argtypes = [i64, i64, i64, i64, dp, dp, dp, u8p, u8p, i32p, dp, dp, u8p,
                                    i32p, dp, dp, u8p, i32p, i32p, u8p, C.c_int]
    L.zo_radar_sweep.restype = i64
    L.zo_radar_sweep.argtypes = [i64, i64, dp, u8p, C.POINTER(ZoRadar), i32p]
    L.zo_noise_apply.restype = None
    L.zo_noise_apply.argtypes = [i64, i32p, dp, i64, dp]
    L.zo_scan_next.restype = None
    L.zo_scan_next.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]
    L.zo_ccp_link.restype = None
    L.zo_ccp_link.argtypes = [i64, dp, dp, i64, dp, dp, C.c_double, C.c_double, i32p, dp]
    L.zo_launch_solve.restype = C.c_int
    L.zo_launch_solve.argtypes = [dp, dp, dp, C.c_double, C.c_double, C.c_double, dp, dp]
    L.zo_philox_noise.restype = None
    L.zo_philox_noise.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, dp]
    L.zo_radar_phase_fused.restype = None
    L.zo_radar_phase_fused.argtypes = [i64, i64, dp, u8p, C.c_int, C.POINTER(ZoRadar), C.c_int, dp,
                                       C.c_uint64, C.c_uint64, i64, u32p, C.c_int]
    L.zo_advance_all.restype = None
    L.zo_advance_all.argtypes = [i64, i64, i64, dp, dp, dp, u8p, dp, dp, C.c_int]
    L.zo_compact_bit.restype = i64
    L.zo_compact_bit.argtypes = [i64, u32p, C.c_int, C.c_int32, i32p]
    _lib = L
    return L


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def dptr(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return _p(a, C.c_double)


def u8ptr(a):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return _p(a, C.c_uint8)


def i32ptr(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return _p(a, C.c_int32)


def u32ptr(a):
    assert a.dtype == np.uint32 and a.flags.c_contiguous
    return _p(a, C.c_uint32)


SCAN_MODES = {"horizontal": 0, "vertical": 1}
LAUNCH_ERRORS = {
    1: "No interception possible: target and interceptor are stationary relative or parallel.",
    2: "Interception impossible in the future: computed time t <= 0.",
    3: "No real interception time: target is too fast or out of range.",
    4: "Interception times are not positive; interception not possible in future.",
    5: "Target is too far for this rocket (detonation_period over limited)",
}


def radar_array(radars):
    """list of dicts / tuples (px,py,pz,max_distance,caz,az_range,cel,el_range) -> ZoRadar[]"""
    arr = (ZoRadar * len(radars))()
    for k, r in enumerate(radars):
        for f, v in zip(("px", "py", "pz", "max_distance", "caz", "az_range", "cel", "el_range"), r):
            setattr(arr[k], f, float(v))
    return arr


def launch_solve(tp, mp, tvu, tsm, v0, period):
    tp = np.ascontiguousarray(tp, dtype=np.float64)
    mp = np.ascontiguousarray(mp, dtype=np.float64)
    tvu = np.ascontiguousarray(tvu, dtype=np.float64)
    V = np.zeros(3)
    t = C.c_double(0.0)
    rc = lib().zo_launch_solve(dptr(tp), dptr(mp), dptr(tvu), float(tsm), float(v0), float(period),
                               dptr(V), C.byref(t))
    return rc, V, t.value


def ccp_link(det_pos, det_speed, trk_ref, trk_upd, now_s, slack_s):
    """CombatControlPoint.link_object for all detections of a tick, sequentially (reference modules/CCP.py:171-219)."""
    det_pos = np.ascontiguousarray(det_pos, np.float64).reshape(-1, 3)
    trk_ref = np.ascontiguousarray(trk_ref, np.float64).reshape(-1, 3)
    D, T = len(det_pos), len(trk_ref)
    det_speed = np.ascontiguousarray(det_speed, np.float64).reshape(D)
    trk_upd = np.ascontiguousarray(trk_upd, np.float64).reshape(T)
    match = np.full(max(D, 1), -1, np.int32)
    scratch = np.zeros(max(T, 1))
    pad = lambda a: a if a.size else np.zeros(3)          # noqa: E731  (ctypes wants a buffer)
    lib().zo_ccp_link(D, dptr(pad(det_pos)), dptr(pad(det_speed)), T, dptr(pad(trk_ref)), dptr(pad(trk_upd)), float(now_s),
                      float(slack_s), i32ptr(match), dptr(scratch))
    return match[:D]


class CcpState:
    """The command post's dictionaries as arrays (zo_ccp_step): target tracks and missile tracks in dict order."""

    def __init__(self, cap, launcher_pos, capacity):
        self.cap = int(cap)
        self.tt_key = np.full(self.cap, -1, np.int32); self.tt_obj = np.full(self.cap, -1, np.int32)
        self.tt_upd = np.zeros(self.cap); self.tt_follow = np.zeros(self.cap, np.uint8)
        self.tm_key = np.full(self.cap, -1, np.int32); self.tm_obj = np.full(self.cap, -1, np.int32); self.tm_upd = np.zeros(self.cap)
        self.n_tt = np.zeros(1, np.int64); self.n_tm = np.zeros(1, np.int64)
        self.key_tt = np.full(self.cap, -1, np.int32)
        self.l_pos = np.ascontiguousarray(launcher_pos, np.float64).reshape(-1, 3)
        self.l_cap = np.ascontiguousarray(capacity, np.int32)
        self.l_launched = np.zeros(len(self.l_cap), np.int32)

    def add_missile(self, row, now_s):
        i64p = C.POINTER(C.c_int64)
        lib().zo_ccp_add_missile(C.c_int32(int(row)), C.c_double(float(now_s)), i32ptr(self.tm_key), i32ptr(self.tm_obj), dptr(self.tm_upd),
                                 self.n_tm.ctypes.data_as(i64p))

    def step(self, seq, pos, prev, prev_none, speed, now_s, slack_s):
        """seq: rows in FoundObjectsMessage order (duplicates allowed); pos / prev (cap, 3).  Returns (rows, verdicts, matched
        track index, launcher index) of the processed detections, in order."""
        i64p = C.POINTER(C.c_int64)
        seq = np.ascontiguousarray(seq, np.int32)
        D = len(seq)
        P = np.ascontiguousarray(np.asarray(pos, np.float64).T).reshape(-1)
        Q = np.ascontiguousarray(np.asarray(prev, np.float64).T).reshape(-1)
        none = np.ascontiguousarray(prev_none, np.uint8); speed = np.ascontiguousarray(speed, np.float64)
        out = [np.zeros(max(D, 1), np.int32) for _ in range(4)]
        scratch = np.zeros(self.cap, np.uint8)
        L = lib()
        L.zo_ccp_step.restype = C.c_int64
        n = L.zo_ccp_step(C.c_int64(D), i32ptr(seq if D else np.zeros(1, np.int32)), C.c_int64(self.cap), dptr(P), dptr(Q), u8ptr(none),
                          dptr(speed), C.c_double(now_s), C.c_double(slack_s), i32ptr(self.tt_key), i32ptr(self.tt_obj), dptr(self.tt_upd),
                          u8ptr(self.tt_follow), self.n_tt.ctypes.data_as(i64p), i32ptr(self.tm_key), i32ptr(self.tm_obj),
                          dptr(self.tm_upd), self.n_tm.ctypes.data_as(i64p), i32ptr(self.key_tt), C.c_int64(len(self.l_cap)),
                          dptr(self.l_pos.reshape(-1)), i32ptr(self.l_cap), i32ptr(self.l_launched), u8ptr(scratch),
                          i32ptr(out[0]), i32ptr(out[1]), i32ptr(out[2]), i32ptr(out[3]))
        if n < 0:
            raise RuntimeError(f"the reference would raise at sequence element {-1 - n}: a target track whose handle has prev_pos None")
        return tuple(o[:n].copy() for o in out)


def philox_noise(seed, tick, ordinal, entity):
    """Noise triple of the `ordinal`-th detection of `entity` in `tick` (throughput mode)."""
    out = np.zeros(3)
    lib().zo_philox_noise(seed, tick, ordinal, entity, dptr(out))
    return out


class OracleSim:
    """SoA replay of the reference L1 loop.  Slots are never reused; slot order is the
    order of AirEnv's object list (targets in add_target order, missiles as appended)."""

    def __init__(self, dt_ms: int, capacity: int, missile_capacity: int):
        self.L = lib()
        self.dt_ms = int(dt_ms)
        self.time_ms = 0
        cap = self.cap = int(capacity)
        mcap = self.mcap = int(missile_capacity)
        self.n = 0                                   # slots in use (AirEnv list length)
        self.sp = np.zeros(3 * cap); self.vel = np.zeros(3 * cap); self.t0 = np.zeros(cap)
        self.pos = np.zeros(3 * cap); self.prev = np.zeros(3 * cap)
        self.prev_valid = np.zeros(cap, np.uint8)    # 0 = prev_pos is None
        self.alive = np.zeros(cap, np.uint8); self.kind = np.zeros(cap, np.uint8)
        self.mrow = np.full(cap, -1, np.int32)
        self.ids = np.zeros(cap, np.int64)
        self.vunit = np.full(3 * cap, np.nan); self.speed_mod = np.zeros(cap)
        # missile table (rows exist from construction; a row gets a slot when it enters AirEnv)
        self.m = 0
        self.m_id = np.zeros(mcap, np.int64); self.m_slot = np.full(mcap, -1, np.int32)
        self.m_tgt = np.full(mcap, -1, np.int32)
        self.m_pos0 = np.zeros((mcap, 3)); self.m_v0 = np.zeros(mcap)
        self.m_radius = np.zeros(mcap); self.m_period = np.zeros(mcap)
        self.m_status = np.zeros(mcap, np.uint8)     # 0 ready, 1 active, 2 detonated
        self.m_traj = np.zeros((mcap, 7))            # V[3], start_pos[3], start_time of the launch solve
        self.radars = []                             # dicts, module order
        self.slot_of_id = {}
        self.row_of_missile = {}
        self._pending_kill = []                      # (missile slot, target slot|-1) from tick t-dt
        self._pending_new = []                       # missile rows announced at tick t-dt
        self._ev = (np.zeros(mcap, np.int32), np.zeros(mcap, np.int32), np.zeros(mcap, np.uint8))

    # -- construction ---------------------------------------------------------
    def add_target(self, id, start_pos, velocity, start_time=0.0, pos=None):
        i = self.n; self.n += 1
        cap = self.cap
        v = np.asarray(velocity, np.float64); s = np.asarray(start_pos, np.float64)
        for c in range(3):
            self.sp[c * cap + i] = s[c]; self.vel[c * cap + i] = v[c]
            self.pos[c * cap + i] = (s if pos is None else np.asarray(pos, np.float64))[c]
        self.t0[i] = start_time; self.alive[i] = 1; self.kind[i] = 0; self.ids[i] = id
        u = np.zeros(3); sm = C.c_double()
        with np.errstate(all="ignore"):
            self.L.zo_unit_velocity(dptr(v.copy()), dptr(u), C.byref(sm))
        for c in range(3):
            self.vunit[c * cap + i] = u[c]
        self.speed_mod[i] = sm.value
        self.slot_of_id.setdefault(int(id), []).append(i)
        return i

    def add_radar(self, id, pos, azimuth_start, elevation_start, max_distance, azimuth_range,
                  elevation_range, azimuth_speed, elevation_speed, scan_mode="horizontal"):
        self.radars.append(dict(id=id, pos=np.asarray(pos, np.float64), az_start=azimuth_start,
                                el_start=elevation_start, max_distance=max_distance,
                                az_range=azimuth_range, el_range=elevation_range,
                                az_speed=azimuth_speed, el_speed=elevation_speed,
                                mode=SCAN_MODES.get(scan_mode, 2),
                                caz=float(azimuth_start), cel=float(elevation_start)))

    def add_missile(self, id, pos, velocity_module, detonate_radius, detonate_period):
        r = self.m; self.m += 1
        self.m_id[r] = id; self.m_pos0[r] = np.asarray(pos, np.float64)
        self.m_v0[r] = velocity_module; self.m_radius[r] = detonate_radius
        self.m_period[r] = detonate_period
        self.row_of_missile[int(id)] = r
        return r

    # -- inbound L2 events ----------------------------------------------------
    def launch(self, missile_id, target_slot):
        """Missile._launch at the current tick (after the radar phase).  Returns (rc, V, t)."""
        r = self.row_of_missile[int(missile_id)]
        cap = self.cap; j = int(target_slot)
        tp = np.array([self.pos[c * cap + j] for c in range(3)])
        tvu = np.array([self.vunit[c * cap + j] for c in range(3)])
        with np.errstate(all="ignore"):
            rc, V, t = launch_solve(tp, self.m_pos0[r], tvu, self.speed_mod[j], self.m_v0[r],
                                    self.m_period[r])
        if rc == 0:
            self.m_tgt[r] = j
            self.m_traj[r, 0:3] = V; self.m_traj[r, 3:6] = self.m_pos0[r]
            self.m_traj[r, 6] = self.time_ms / 1000
            self.m_status[r] = 1
        return rc, V, t

    def announce_missile(self, missile_id):
        """NEW_MISSILE posted at the current tick; AirEnv appends it on the next one."""
        self._pending_new.append(self.row_of_missile[int(missile_id)])

    # -- one tick ---------------------------------------------------------------
    def airenv_step(self):
        cap = self.cap
        # modules/AirEnv.py:33-40 tombstones (by id, hence every slot holding that id)
        for ms, ts in self._pending_kill:
            for s in (ms, ts):
                if s >= 0:
                    for q in self.slot_of_id.get(int(self.ids[s]), [s]):
                        self.alive[q] = 0
        self._pending_kill = []
        # modules/AirEnv.py:42-43 append
        for r in self._pending_new:
            i = self.n; self.n += 1
            for c in range(3):
                self.vel[c * cap + i] = self.m_traj[r, c]
                self.sp[c * cap + i] = self.m_traj[r, 3 + c]
                self.pos[c * cap + i] = self.m_pos0[r, c]
            self.t0[i] = self.m_traj[r, 6]
            self.alive[i] = 1; self.kind[i] = 1; self.mrow[i] = r; self.ids[i] = self.m_id[r]
            self.speed_mod[i] = self.m_v0[r]       # Missile.py:28; velocity stays NaN
            self.m_slot[r] = i
            self.slot_of_id.setdefault(int(self.m_id[r]), []).append(i)
        self._pending_new = []
        evm, evt, evs = self._ev
        nev = self.L.zo_airenv_step(self.n, cap, self.time_ms, self.dt_ms, dptr(self.sp), dptr(self.vel),
                                    dptr(self.t0), u8ptr(self.alive), u8ptr(self.kind), i32ptr(self.mrow),
                                    dptr(self.pos), dptr(self.prev), u8ptr(self.prev_valid),
                                    i32ptr(self.m_tgt), dptr(self.m_radius), dptr(self.m_period),
                                    u8ptr(self.m_status), i32ptr(evm), i32ptr(evt), u8ptr(evs))
        events = [(int(evm[k]), int(evt[k]), bool(evs[k])) for k in range(nev)]
        self._pending_kill = [(m, t) for m, t, _ in events]
        return events

    def radar_params(self, rd):
        return (rd["pos"][0], rd["pos"][1], rd["pos"][2], rd["max_distance"], rd["caz"], rd["az_range"],
                rd["cel"], rd["el_range"])

    def radar_step(self, k, noise_fn=None):
        """One SectorRadar.step: sweep, in-place noise, scan advance.  noise_fn(count) -> (count,3)
        array (reference: np.random.normal(0, 5, 3) per object, modules/Radar.py:138-142)."""
        rd = self.radars[k]
        arr = radar_array([self.radar_params(rd)])
        out = np.zeros(max(self.n, 1), np.int32)
        cnt = self.L.zo_radar_sweep(self.n, self.cap, dptr(self.pos), u8ptr(self.alive), arr, i32ptr(out))
        found = out[:cnt].copy()
        if noise_fn is not None and cnt:
            # (a stream keyed by entity -- the device's counter-based one -- wants to know whose draws these are)
            nz = noise_fn(found, k) if getattr(noise_fn, "by_slot", False) else noise_fn(cnt)
            nz = np.ascontiguousarray(nz, dtype=np.float64).reshape(cnt, 3)
            self.L.zo_noise_apply(cnt, i32ptr(found), dptr(nz), self.cap, dptr(self.pos))
        caz = C.c_double(rd["caz"]); cel = C.c_double(rd["cel"])
        self.L.zo_scan_next(rd["mode"], rd["az_range"], rd["az_speed"], rd["el_speed"], rd["el_start"],
                            C.byref(caz), C.byref(cel))
        rd["caz"], rd["cel"] = caz.value, cel.value
        return found

    def end_tick(self):
        self.time_ms += self.dt_ms

    # -- views ------------------------------------------------------------------
    def active_slots(self):
        return np.nonzero(self.alive[:self.n])[0]

    def pos_of(self, slots):
        return np.stack([self.pos[c * self.cap + np.asarray(slots)] for c in range(3)], axis=-1)

    def prev_of(self, slots):
        return np.stack([self.prev[c * self.cap + np.asarray(slots)] for c in range(3)], axis=-1)
