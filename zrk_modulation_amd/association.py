"""Track association of the command post on the device.

`link_all` answers, for every detection of a tick at once, what the reference's sequential
`for obj in visible_objects: link_object(obj)` loop answers one call at a time (reference modules/CCP.py:171-219,
:414-429): which existing track -- target or missile -- the detection continues, or that it is a new target.  The
pairwise distance work (every detection against every track, the part that is quadratic in the reference) runs
LDS-tiled on the device; the order dependence of the reference's loop (a track matched by an earlier detection of the
tick is gone for the later ones) is resolved on the device in rounds that reproduce the sequential result exactly
(include/zrk_hot.h, zrk_ccp_link).  No CPU fallback: without the library and a GPU this raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def link_all(ctx, device, det_pos, det_speed, trk_ref, trk_upd, now_s, slack_s):
    """det_pos (D,3), det_speed (D,), trk_ref (T,3), trk_upd (T,) -> int32 (D,): track index or -1 (new target).
    Tracks in the order the command post scans them: target tracks, then missile tracks."""
    det_pos = np.ascontiguousarray(det_pos, np.float64).reshape(-1, 3)
    D = det_pos.shape[0]
    if D == 0:
        return np.zeros(0, np.int32)
    trk_ref = np.ascontiguousarray(trk_ref, np.float64).reshape(-1, 3)
    T = trk_ref.shape[0]
    dev = torch.device(device)
    d_pos = torch.from_numpy(det_pos).to(dev)
    d_speed = torch.from_numpy(np.ascontiguousarray(det_speed, np.float64).reshape(D)).to(dev)
    d_ref = torch.from_numpy(trk_ref).to(dev) if T else torch.zeros(3, dtype=torch.float64, device=dev)
    d_upd = torch.from_numpy(np.ascontiguousarray(trk_upd, np.float64).reshape(T)).to(dev) if T else torch.zeros(1, dtype=torch.float64, device=dev)
    match = torch.full((D,), -2, dtype=torch.int32, device=dev)
    scratch = torch.zeros(int(ctx.lib.zrk_ccp_scratch_bytes(D, T)), dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.zrk_ccp_link(ctx.handle, d_pos.data_ptr(), d_speed.data_ptr(), D, d_ref.data_ptr(), d_upd.data_ptr(), T,
                                   float(now_s), float(slack_s), match.data_ptr(), scratch.data_ptr(), stream), "zrk_ccp_link")
    return match.cpu().numpy()


class DeviceCommandPost:
    """The command post's dictionaries and one tick of its detection loop on the device (include/zrk_hot.h: zrk_ccp_step):
    what CombatControlPoint.step does per detection -- link_object, new_target / old_target / old_rocket,
    try_to_launch_missile (reference modules/CCP.py:171-219, :287-366, :406-429) -- for all detections of a tick, with the
    result of the reference's sequential loop and without a read-back.  Tracks and launchers live in device arrays; the
    detections are rows of the entity table."""

    def __init__(self, ctx, device, rows_capacity, track_capacity, launcher_pos, launcher_capacity, dmax, rounds=12):
        self.ctx, self.dev = ctx, torch.device(device)
        dev = self.dev
        i32, f64, u8 = torch.int32, torch.float64, torch.uint8
        self.tcap, self.dmax, self.rounds = int(track_capacity), int(dmax), int(rounds)
        z = lambda n, dt, fill=0: torch.full((max(int(n), 1),), fill, dtype=dt, device=dev)     # noqa: E731
        self.tt_key, self.tt_obj, self.tt_upd, self.tt_follow = z(self.tcap, i32, -1), z(self.tcap, i32, -1), z(self.tcap, f64), z(self.tcap, u8)
        self.tm_key, self.tm_obj, self.tm_upd = z(self.tcap, i32, -1), z(self.tcap, i32, -1), z(self.tcap, f64)
        self.counts = z(2, i32)
        self.key_tt = z(rows_capacity, i32, -1)
        lp = np.ascontiguousarray(launcher_pos, np.float64).reshape(-1, 3)
        self.L = len(lp)
        self.l_pos = torch.from_numpy(lp).to(dev) if self.L else z(3, f64)
        self.l_cap = torch.from_numpy(np.ascontiguousarray(launcher_capacity, np.int32)).to(dev) if self.L else z(1, i32)
        self.l_launched = z(self.L, i32)
        self.o_obj, self.o_verdict, self.o_match, self.o_launcher = (z(self.dmax, i32, -1) for _ in range(4))
        self.o_count, self.o_status = z(2, i32), z(1, i32)
        self.scratch = torch.zeros(int(ctx.lib.zrk_ccp_step_scratch_bytes(self.dmax, self.tcap)), dtype=u8, device=dev)
        t = self.tracks = _lib.ZrkCcpTracks()
        t.capacity = self.tcap
        t.tt_key, t.tt_obj, t.tt_upd, t.tt_follow = (x.data_ptr() for x in (self.tt_key, self.tt_obj, self.tt_upd, self.tt_follow))
        t.tm_key, t.tm_obj, t.tm_upd = (x.data_ptr() for x in (self.tm_key, self.tm_obj, self.tm_upd))
        t.counts, t.key_tt = self.counts.data_ptr(), self.key_tt.data_ptr()
        # what link_object compares with for tracks whose object has left the air (NaN: read the table)
        self.tt_ref_fixed = torch.full((max(self.tcap, 1), 3), float("nan"), dtype=f64, device=dev)
        self.tm_ref_fixed = torch.full((max(self.tcap, 1), 3), float("nan"), dtype=f64, device=dev)
        t.tt_ref_fixed, t.tm_ref_fixed = self.tt_ref_fixed.data_ptr(), self.tm_ref_fixed.data_ptr()
        l = self.launchers = _lib.ZrkCcpLaunchers()
        l.L, l.pos, l.capacity, l.launched = self.L, self.l_pos.data_ptr(), self.l_cap.data_ptr(), self.l_launched.data_ptr()
        o = self.out = _lib.ZrkCcpOut()
        o.obj, o.verdict, o.match, o.launcher = (x.data_ptr() for x in (self.o_obj, self.o_verdict, self.o_match, self.o_launcher))
        o.count, o.status = self.o_count.data_ptr(), self.o_status.data_ptr()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def add_missile(self, row, now_s):
        self.ctx.check(self.ctx.lib.zrk_ccp_add_missile(self.ctx.handle, C.byref(self.tracks), int(row), float(now_s), self._stream()),
                       "zrk_ccp_add_missile")

    def step(self, ents, cur, speed_mod, seq, seq_count, now_s, slack_s):
        """ents: a ctypes ZrkEntities; speed_mod: float64 tensor [capacity]; seq: int32 tensor of rows in processing order;
        seq_count: int32 tensor [1] on the device.  Enqueues the tick; results stay on the device (`results()` reads them)."""
        self.ctx.check(self.ctx.lib.zrk_ccp_step(self.ctx.handle, C.byref(ents), int(cur), speed_mod.data_ptr(), seq.data_ptr(),
                                                 seq_count.data_ptr(), min(self.dmax, seq.numel()), C.byref(self.tracks),
                                                 C.byref(self.launchers), C.byref(self.out), float(now_s), float(slack_s), self.rounds,
                                                 self.scratch.data_ptr(), self._stream()), "zrk_ccp_step")

    def requests(self, missile_params, k_max):
        """The last tick's launch decisions as a device array of zrk_launch_req in request order, padded to k_max with requests
        for no row, and their number (int32 tensor [1]) -- what zrk_launch_salvo takes, with nothing read back
        (include/zrk_hot.h: zrk_ccp_requests).  missile_params: [L][3] velocity_module, detonate_period, detonate_radius."""
        dev = self.dev
        par = torch.as_tensor(np.ascontiguousarray(missile_params, np.float64).reshape(-1, 3)).to(dev)
        assert par.shape[0] == self.L
        req = torch.zeros(int(k_max) * C.sizeof(_lib.ZrkLaunchReq), dtype=torch.uint8, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ctx.check(self.ctx.lib.zrk_ccp_requests(self.ctx.handle, C.byref(self.out), self.dmax, C.byref(self.launchers), par.data_ptr(),
                                                     req.data_ptr(), int(k_max), count.data_ptr(), self._stream()), "zrk_ccp_requests")
        return req, count

    def results(self):
        """(rows, verdicts, matched track index, launcher index) of the last tick's detections; raises if the device says the
        tick did not go through (zrk_hot.h: status)."""
        status = int(self.o_status.item())
        if status:
            raise _lib.ZrkError(f"zrk_ccp_step: status {status} ("
                                + {1: "the order-dependent part did not settle within the round limit",
                                   2: "a target track's handle has no prev_pos: the reference raises there",
                                   3: "track capacity exhausted"}.get(status, "?") + ")")
        n = int(self.o_count[0].item())
        return tuple(x[:n].cpu().numpy() for x in (self.o_obj, self.o_verdict, self.o_match, self.o_launcher))
