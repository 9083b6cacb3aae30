"""The battery's closed loop on the device: sweep -> lists -> command post -> launchers -> missiles in the air, tick after
tick, with nothing read back in between (include/zrk_hot.h: zrk_battery).

What the reference does through its message bus with four kinds of modules (reference modules/Manager.py:111-140) --
AirEnv and the radars (the hot path: HotPathEngine), MissileLauncher (modules/MissileLauncher.py:82-138), Missile._launch
(modules/Missile.py:104-133) and CombatControlPoint (modules/CCP.py:368-431) -- runs here as one sequence of launches per
tick on one stream, with the reference's latencies: a request of tick b is solved in tick b + 1 against the target's position
after that tick's radars, announced in tick b + 2 (the command post takes the missile into its dictionary; a cancelled
missile is back on its launcher's list) and flies from tick b + 3; the command post learns the launchers' missile counts in
tick 1.  The magazine's rows are part of the table from the start (dead until their missile flies), so the table never
grows and no count has to come back to the host.  No CPU fallback: without the library and a GPU this raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .association import DeviceCommandPost


class DeviceBattery:
    """launchers: [{"id", "position", "max_missiles", "missiles": [{"id", "velocity", "explosion_radius", "life_time"}]}] in module
    order (the YAML schema of the reference's main.py:74-101); ccp_launcher_ids: the launchers the command post knows, in its
    order (main.py:106-110; default: all).  `engine`: a HotPathEngine that has been loaded with the targets and has its
    per-radar lists enabled; the battery appends the magazine's rows to its table."""

    def __init__(self, engine, launchers, ccp_launcher_ids=None, slack_steps=100, rounds=2, log_capacity=None):
        self.eng = eng = engine
        st = self.st = eng.store
        assert eng.det_idx is not None, "enable_lists() first: the command post reads the per-radar lists"
        dev = st.device
        f64, i32 = torch.float64, torch.int32
        self.n_targets = st.n
        # the magazine: launcher after launcher, each in its list's order (MissileLauncher.add_missile keeps max_missiles of them)
        mi, stacks = [], []
        for lc in launchers:
            own = []
            for mc in (lc.get("missiles") or []):
                if len(own) < lc.get("max_missiles", 5):
                    own.append(len(mi))
                    mi.append(dict(id=int(mc["id"]), pos=np.asarray(lc["position"], np.float64), speed=float(mc.get("velocity", 1000)),
                                   period=float(mc.get("life_time", 60)), radius=float(mc.get("explosion_radius", 50)), launcher=len(stacks)))
            stacks.append(own)
        self.missiles, self.launchers = mi, launchers
        self.L, self.nm = len(launchers), len(mi)
        ids = [lc["id"] for lc in launchers]
        known = ids if ccp_launcher_ids is None else [i for i in ccp_launcher_ids if i in ids]
        assert known == ids[:len(known)] and len(known) == len(ids), "the command post's launchers: all of them, in module order"
        nm = self.nm
        # reserved rows: dead, kind 1, at the end of the table and of AirEnv's list
        st.flush()
        row0 = self.row0 = st.n
        if nm:
            li = None if st.h_lidx is None else np.arange(eng.n_list, eng.n_list + nm, dtype=np.int32)
            st.add_entities([m["id"] for m in mi], [m["pos"] for m in mi], np.zeros((nm, 3)), 0.0, kind=1, list_index=li)
            st.flush()
            st.d_alive[row0:row0 + nm] = 0
            st.h_alive[row0:row0 + nm] = 0
            if eng.row_of_list is not None:
                eng.row_of_list = np.concatenate([eng.row_of_list, row0 + np.arange(nm)])
            eng.n_list += nm
            if st.mcap < nm:
                st._alloc_missiles(nm)
            st.dm_slot[:nm] = torch.arange(row0, row0 + nm, dtype=i32, device=dev)
            st.dm_tgt[:nm] = 0
            st.dm_status[:nm] = 0
            st.m = nm
            st.hm_slot = np.arange(row0, row0 + nm, dtype=np.int32); st.hm_tgt = np.zeros(nm, np.int32)
        eng.loop.n = st.n_uploaded
        # (the lists must hold every row)
        if eng.det_stride < st.cap:
            eng.enable_lists(st.cap)
        self.k_max = k_max = max(64, nm)
        z = lambda n, dt, fill=0: torch.full((max(int(n), 1),), fill, dtype=dt, device=dev)     # noqa: E731
        self.mi_pos = torch.tensor(np.array([m["pos"] for m in mi]).reshape(-1, 3), dtype=f64, device=dev) if nm else z(3, f64)
        self.mi_speed = torch.tensor([m["speed"] for m in mi], dtype=f64, device=dev) if nm else z(1, f64)
        self.mi_period = torch.tensor([m["period"] for m in mi], dtype=f64, device=dev) if nm else z(1, f64)
        self.mi_radius = torch.tensor([m["radius"] for m in mi], dtype=f64, device=dev) if nm else z(1, f64)
        stack = np.full((max(self.L, 1), max(nm, 1)), -1, np.int32)
        for l, own in enumerate(stacks):
            stack[l, :len(own)] = own
        self.stack = torch.from_numpy(stack).to(dev)
        self.stock0 = [len(own) for own in stacks]
        self.top = torch.tensor(self.stock0 or [0], dtype=i32, device=dev)
        self.sal_row, self.sal_launcher, self.sal_missile, self.sal_rc, self.sal_air = (z(3 * k_max, i32, -1) for _ in range(5))
        self.sal_V = z(9 * k_max, f64)
        self.sal_count, self.air_count, self.air_missile = z(6, i32), z(1, i32), z(nm, i32, -1)
        self.speed = torch.zeros(st.cap, dtype=f64, device=dev)
        self.log_cap = int(log_capacity or max(4 * nm, 4096))
        self.log_solve, self.log_V, self.log_event, self.log_count = z(5 * self.log_cap, i32), z(3 * self.log_cap, f64), z(3 * self.log_cap, i32), z(2, i32)
        b = self.bat = _lib.ZrkBattery()
        b.L, b.k_max, b.n_missiles, b.row0 = self.L, k_max, nm, row0
        for name in ("mi_pos", "mi_speed", "mi_period", "mi_radius", "stack", "top", "sal_row", "sal_launcher", "sal_missile", "sal_rc", "sal_air",
                     "sal_V", "sal_count", "air_count", "air_missile", "log_solve", "log_V", "log_event", "log_count"):
            setattr(b, name, getattr(self, name).data_ptr())
        b.speed_mod = self.speed.data_ptr()
        b.log_cap = self.log_cap
        self._cap = st.cap                       # (the table must not be re-allocated behind the descriptors)
        st.ctx.check(st.lib.zrk_battery_speed_column(st.ctx.handle, C.byref(st.ents), self.n_targets, self.speed.data_ptr(), st._stream()),
                     "zrk_battery_speed_column")
        # the command post: every row may become a track; its capacities start at 0 (modules/CCP.py:122-136)
        self.post = DeviceCommandPost(st.ctx, dev, st.cap, st.cap, [lc["position"] for lc in launchers], [0] * self.L, dmax=st.cap, rounds=rounds)
        self.frozen_prev = torch.full((st.cap, 3), float("nan"), dtype=f64, device=dev)
        st.ctx.check(st.lib.zrk_ctx_keep_prev(st.ctx.handle, self.frozen_prev.data_ptr()), "zrk_ctx_keep_prev")
        self.post.tracks.row_ref_fixed = self.frozen_prev.data_ptr()
        self.seq = torch.zeros(st.cap, dtype=i32, device=dev)
        self.seq_count = torch.zeros(1, dtype=i32, device=dev)
        self.row_of_list_dev = (torch.from_numpy(np.ascontiguousarray(eng.row_of_list, np.int32)).to(dev) if eng.row_of_list is not None else None)
        self.slack_s = slack_steps * eng.dt_ms / 1000
        self.tick = 0

    def close(self):
        st = self.st
        st.ctx.check(st.lib.zrk_ctx_keep_prev(st.ctx.handle, None), "zrk_ctx_keep_prev")

    def run(self, K, sweep_ms=None):
        """K ticks of the closed loop, enqueued; nothing is read back (sweep_ms: a float32 array of K, to time every sweep --
        that synchronises)."""
        eng, st, lib, h = self.eng, self.st, self.st.lib, self.st.ctx.handle
        check = st.ctx.check
        assert st.cap == self._cap, "the table was re-allocated behind the battery"
        bat, ents, mis, post = C.byref(self.bat), C.byref(st.ents), C.byref(st.mis), self.post
        for k in range(int(K)):
            t, time_ms = self.tick, int(eng.loop.time_ms)
            now_s = time_ms / 1000
            s = st._stream()
            check(lib.zrk_battery_activate(h, bat, ents, mis, t, st.workspace().data_ptr(), s), "zrk_battery_activate")
            eng.run(1, sweep_ms=None if sweep_ms is None else sweep_ms[k:k + 1], prof_stride=1)
            check(lib.zrk_battery_launchers(h, bat, ents, st.cur, mis, t, time_ms, s), "zrk_battery_launchers")
            if t == 1:                               # MissileCountResponse arrives (modules/CCP.py:138-146)
                post.l_cap.copy_(torch.tensor(self.stock0 or [0], dtype=torch.int32))
            check(lib.zrk_battery_announce(h, bat, C.byref(post.tracks), t, now_s, s), "zrk_battery_announce")
            vis = st.d_vis_alt if st.vis_cur else st.d_vis
            check(lib.zrk_battery_sequence(h, eng.det_idx.data_ptr(), eng.det_stride, eng.det_cnt.data_ptr(), eng.R, 0, vis.data_ptr(),
                                           self.row_of_list_dev.data_ptr() if self.row_of_list_dev is not None else None,
                                           self.seq.data_ptr(), self.seq_count.data_ptr(), self.seq.numel(), s), "zrk_battery_sequence")
            post.step(st.ents, st.cur, self.speed, self.seq, self.seq_count, now_s, self.slack_s)
            check(lib.zrk_battery_requests(h, bat, C.byref(post.out), post.dmax, t, s), "zrk_battery_requests")
            check(lib.zrk_battery_log_events(h, bat, mis, t, s), "zrk_battery_log_events")
            self.tick += 1

    def results(self):
        """Reads the logs (synchronises): every launch solve and every detonation since the start, in order, with ids --
        {"solves": [(t_ms, launcher_id, missile_id, target_id, rc, V[3])], "new_missile": [(t_ms, missile_id)],
         "detonations": [(t_ms, missile_id, target_id | -1, self_detonation)]}."""
        st, dt = self.st, self.eng.dt_ms
        post_status = int(self.post.o_status.item())
        if post_status:
            raise _lib.ZrkError(f"zrk_ccp_step: status {post_status}")
        ns, ne = (int(v) for v in self.log_count.cpu().tolist())
        assert ns <= self.log_cap and ne <= self.log_cap, "log capacity exceeded"
        sol = self.log_solve.cpu().numpy().reshape(-1, 5)[:ns]
        V = self.log_V.cpu().numpy().reshape(-1, 3)[:ns]
        ev = self.log_event.cpu().numpy().reshape(-1, 3)[:ne]
        air = self.air_missile.cpu().numpy()
        ids = st.h_ids

        def row_id(row):
            if row < 0:
                return -1
            return int(ids[row]) if row < self.row0 else self.missiles[int(air[row - self.row0])]["id"]
        solves, new = [], []
        for (tick, m, trow, rc, j), v in zip(sol, V):
            if m < 0:
                continue                               # (the launcher had no missile left: nothing was sent, MissileLauncher.py:65-67)
            mm = self.missiles[int(m)]
            solves.append((int(tick) * dt, self.launchers[mm["launcher"]]["id"], mm["id"], row_id(int(trow)), int(rc), v.copy()))
            if rc == 0:
                new.append(((int(tick) + 1) * dt, mm["id"]))
        dets = [(int(tick) * dt, row_id(int(mrow)), row_id(int(trow)), int(trow < 0)) for tick, mrow, trow in ev]
        return dict(solves=solves, new_missile=new, detonations=dets)
