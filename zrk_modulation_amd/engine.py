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
    """Replay surface shared with oracle.OracleSim (tests/helpers.replay_l1)."""

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
        det, off = st.compact(1)
        cnt = int(off[1].item())
        found = det[:cnt].cpu().numpy() if cnt else np.zeros(0, np.int32)
        if noise_fn is not None and cnt:
            st.noise_apply(det, cnt, noise_fn(cnt))
        rd["caz"], rd["cel"] = scan_next(rd["mode"], rd["az_range"], rd["az_speed"], rd["el_speed"], rd["el_start"],
                                         rd["caz"], rd["cel"])
        return found

    def end_tick(self):
        self.time_ms += self.dt_ms
