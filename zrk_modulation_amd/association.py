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
