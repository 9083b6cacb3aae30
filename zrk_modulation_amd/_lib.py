"""ctypes binding of libzrk_hot.so (include/zrk_hot.h).

There is no fallback: if the shared library is missing or does not load, importing
anything that computes raises `HotPathUnavailable`.  The library is built in-tree by
`__graft_entry__.build()` / `make -C zrk_modulation_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
# (ZRK_HOT_LIB: another build of the same ABI, for A/B runs on one box)
LIB_PATH = Path(os.environ["ZRK_HOT_LIB"]) if os.environ.get("ZRK_HOT_LIB") else CSRC / "libzrk_hot.so"

ZRK_ABI_VERSION = 11
EXCHANGE_SLOTS = 8          # ZRK_EXCHANGE_SLOTS
ZRK_MAX_RADARS = 32
ZRK_BLOCK = 256
F_ADVANCE, F_PHILOX, F_EXACT_ONLY, F_UNION_BITS = 1, 2, 4, 8
ZRK_E_INVALID, ZRK_E_HIP, ZRK_E_CAPACITY, ZRK_E_STATE = -1, -2, -3, -4


class HotPathUnavailable(RuntimeError):
    pass


class ZrkCcpTracks(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("tt_key", C.c_void_p), ("tt_obj", C.c_void_p), ("tt_upd", C.c_void_p),
                ("tt_follow", C.c_void_p), ("tm_key", C.c_void_p), ("tm_obj", C.c_void_p), ("tm_upd", C.c_void_p),
                ("counts", C.c_void_p), ("key_tt", C.c_void_p), ("tt_ref_fixed", C.c_void_p), ("tm_ref_fixed", C.c_void_p),
                ("row_ref_fixed", C.c_void_p)]


class ZrkBattery(C.Structure):
    """include/zrk_hot.h: zrk_battery"""
    _fields_ = [("L", C.c_int32), ("k_max", C.c_int32), ("n_missiles", C.c_int32), ("row0", C.c_int32),
                ("mi_pos", C.c_void_p), ("mi_speed", C.c_void_p), ("mi_period", C.c_void_p), ("mi_radius", C.c_void_p),
                ("stack", C.c_void_p), ("top", C.c_void_p),
                ("sal_row", C.c_void_p), ("sal_launcher", C.c_void_p), ("sal_missile", C.c_void_p), ("sal_rc", C.c_void_p), ("sal_air", C.c_void_p),
                ("sal_V", C.c_void_p), ("sal_count", C.c_void_p), ("air_count", C.c_void_p), ("air_missile", C.c_void_p), ("speed_mod", C.c_void_p),
                ("log_solve", C.c_void_p), ("log_V", C.c_void_p), ("log_event", C.c_void_p), ("log_count", C.c_void_p),
                ("log_cap", C.c_int32), ("_pad", C.c_int32)]


class ZrkCcpLaunchers(C.Structure):
    _fields_ = [("L", C.c_int32), ("_pad", C.c_int32), ("pos", C.c_void_p), ("capacity", C.c_void_p), ("launched", C.c_void_p)]


class ZrkCcpOut(C.Structure):
    _fields_ = [("obj", C.c_void_p), ("verdict", C.c_void_p), ("match", C.c_void_p), ("launcher", C.c_void_p),
                ("count", C.c_void_p), ("status", C.c_void_p)]


class ZrkEntities(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64),
        ("start_pos", C.c_void_p),
        ("velocity", C.c_void_p),
        ("start_time", C.c_void_p),
        ("alive", C.c_void_p),
        ("kind", C.c_void_p),
        ("pos", C.c_void_p * 2),
        ("vis_mask", C.c_void_p),
        ("vis_mask_alt", C.c_void_p),
        ("list_index", C.c_void_p),
    ]


class ZrkRadar(C.Structure):
    _fields_ = [
        ("pos", C.c_double * 3),
        ("max_distance", C.c_double),
        ("cur_azimuth", C.c_double),
        ("azimuth_range", C.c_double),
        ("cur_elevation", C.c_double),
        ("elevation_range", C.c_double),
    ]


class ZrkMissiles(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64),
        ("slot", C.c_void_p),
        ("target", C.c_void_p),
        ("radius", C.c_void_p),
        ("period", C.c_void_p),
        ("status", C.c_void_p),
        ("ev_code", C.c_void_p),
        ("ev_missile", C.c_void_p),
        ("ev_target", C.c_void_p),
        ("ev_count", C.c_void_p),
    ]


class ZrkLaunchReq(C.Structure):
    _fields_ = [
        ("target_slot", C.c_int32),
        ("_pad", C.c_int32),
        ("missile_pos", C.c_double * 3),
        ("speed", C.c_double),
        ("period", C.c_double),
        ("radius", C.c_double),
    ]


class ZrkLaunchRes(C.Structure):
    _fields_ = [
        ("rc", C.c_int32),
        ("_pad", C.c_int32),
        ("velocity", C.c_double * 3),
        ("t_hit", C.c_double),
    ]


class ZrkExchangeStats(C.Structure):
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("comm_ranks", C.c_int32), ("direct", C.c_int32),
                ("helper_threads", C.c_int32), ("grouped_pairs", C.c_int32), ("collectives", C.c_int64), ("host_waits", C.c_int64),
                ("host_wait_us", C.c_double)]


def launch_dtypes():
    """numpy views of zrk_launch_req / zrk_launch_res arrays."""
    import numpy as np
    req = np.dtype([("target_slot", "<i4"), ("_pad", "<i4"), ("missile_pos", "<f8", 3), ("speed", "<f8"),
                    ("period", "<f8"), ("radius", "<f8")])
    res = np.dtype([("rc", "<i4"), ("_pad", "<i4"), ("velocity", "<f8", 3), ("t_hit", "<f8")])
    assert req.itemsize == C.sizeof(ZrkLaunchReq) and res.itemsize == C.sizeof(ZrkLaunchRes)
    return req, res


class ZrkScan(C.Structure):
    _fields_ = [
        ("azimuth_speed", C.c_double),
        ("elevation_speed", C.c_double),
        ("elevation_start", C.c_double),
        ("mode", C.c_int32),
        ("_pad", C.c_int32),
    ]


class ZrkLoop(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("time_ms", C.c_int64),
        ("dt_ms", C.c_int64),
        ("gid0", C.c_int64),
        ("seed", C.c_uint64),
        ("tick", C.c_uint64),
        ("cur", C.c_int32),
        ("base_index", C.c_int32),
        ("flags", C.c_uint32),
        ("vis_cur", C.c_int32),
    ]


class ZrkRcclId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class ZrkExchangeIo(C.Structure):
    _fields_ = [
        ("x", C.c_void_p),
        ("send", C.c_void_p * EXCHANGE_SLOTS),
        ("recv", C.c_void_p * EXCHANGE_SLOTS),
        ("words", C.c_int64),
        ("ev_capacity", C.c_int32),
        ("interest", C.c_uint32),
    ]


class ZrkEnsemble(C.Structure):
    _fields_ = [
        ("scenarios", C.c_int32),
        ("radars", C.c_int32),
        ("rows_per_scenario", C.c_int64),
        ("radar_state", C.c_void_p),
        ("scan", C.c_void_p),
        ("d2_max", C.c_void_p),
        ("seeds", C.c_void_p),
        ("tables", C.c_void_p),
    ]


# name -> (restype, argtypes); the exported surface of include/zrk_hot.h
_PROTOTYPES = {
    "zrk_abi_version": (C.c_int, []),
    "zrk_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "zrk_ctx_destroy": (None, [C.c_void_p]),
    "zrk_ctx_reload_env": (None, [C.c_void_p]),
    "zrk_ctx_invalidate_boxes": (None, [C.c_void_p]),
    "zrk_last_error": (C.c_char_p, [C.c_void_p]),
    "zrk_workspace_bytes": (C.c_int64, [C.c_int64]),
    "zrk_tick_sweep": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.c_int64, C.c_int, C.c_int64,
                                 C.POINTER(ZrkRadar), C.c_int, C.c_uint32, C.c_uint64, C.c_uint64,
                                 C.c_int64, C.c_void_p, C.c_void_p]),
    "zrk_compact": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int32, C.c_void_p, C.c_void_p,
                              C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "zrk_compact_bits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "zrk_union_bits_words": (C.c_int64, [C.c_int64, C.c_int, C.c_int64]),
    "zrk_compact_status": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "zrk_scan_advance": (C.c_int, [C.POINTER(ZrkRadar), C.POINTER(ZrkScan), C.c_int]),
    "zrk_run_ticks": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.POINTER(ZrkMissiles), C.c_int64,
                                C.POINTER(ZrkLoop), C.POINTER(ZrkRadar), C.POINTER(ZrkScan), C.c_int, C.c_void_p,
                                C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "zrk_run_ticks_x": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.POINTER(ZrkMissiles), C.c_int64,
                                  C.POINTER(ZrkLoop), C.POINTER(ZrkRadar), C.POINTER(ZrkScan), C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(ZrkExchangeIo),
                                  C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "zrk_run_ticks_ensemble": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.POINTER(ZrkMissiles), C.c_int64,
                                         C.POINTER(ZrkLoop), C.POINTER(ZrkEnsemble), C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "zrk_ensemble_table_bytes": (C.c_int64, [C.c_int]),
    "zrk_d2_threshold": (C.c_double, [C.c_double]),
    "zrk_exchange_unique_id": (C.c_int, [C.c_char_p, C.POINTER(ZrkRcclId)]),
    "zrk_exchange_create": (C.c_int, [C.c_char_p, C.POINTER(ZrkRcclId), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "zrk_exchange_destroy": (None, [C.c_void_p]),
    "zrk_exchange_last_error": (C.c_char_p, [C.c_void_p]),
    "zrk_exchange_all_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "zrk_exchange_wait": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "zrk_exchange_sync": (C.c_int, [C.c_void_p]),
    "zrk_exchange_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "zrk_exchange_plan_helpers": (C.c_int, [C.c_int]),
    "zrk_ctx_keep_prev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "zrk_battery_speed_column": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.c_int64, C.c_void_p, C.c_void_p]),
    "zrk_battery_activate": (C.c_int, [C.c_void_p, C.POINTER(ZrkBattery), C.POINTER(ZrkEntities), C.POINTER(ZrkMissiles), C.c_int64, C.c_void_p,
                                       C.c_void_p]),
    "zrk_battery_launchers": (C.c_int, [C.c_void_p, C.POINTER(ZrkBattery), C.POINTER(ZrkEntities), C.c_int, C.POINTER(ZrkMissiles), C.c_int64,
                                        C.c_int64, C.c_void_p]),
    "zrk_battery_announce": (C.c_int, [C.c_void_p, C.POINTER(ZrkBattery), C.POINTER(ZrkCcpTracks), C.c_int64, C.c_double, C.c_void_p]),
    "zrk_battery_sequence": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p]),
    "zrk_battery_requests": (C.c_int, [C.c_void_p, C.POINTER(ZrkBattery), C.POINTER(ZrkCcpOut), C.c_int64, C.c_int64, C.c_void_p]),
    "zrk_battery_log_events": (C.c_int, [C.c_void_p, C.POINTER(ZrkBattery), C.POINTER(ZrkMissiles), C.c_int64, C.c_void_p]),
    "zrk_last_run_overlapped": (C.c_int, [C.c_void_p]),
    "zrk_read_sweep_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "zrk_noise_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_int64, C.c_void_p]),
    "zrk_missile_step": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.c_int, C.POINTER(ZrkMissiles),
                                   C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "zrk_kill_slots": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.c_int, C.c_void_p, C.c_int64,
                                 C.c_void_p]),
    "zrk_apply_events": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.c_int, C.POINTER(ZrkMissiles),
                                   C.c_void_p]),
    "zrk_launch_solve": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_int64, C.c_void_p]),
    "zrk_launch_salvo": (C.c_int, [C.c_void_p, C.POINTER(ZrkEntities), C.c_int, C.POINTER(ZrkMissiles), C.c_int64,
                                   C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                                   C.c_void_p]),
    "zrk_ccp_link": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zrk_ccp_scratch_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "zrk_selftest_math": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p]),
    "zrk_selftest_noise": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int64, C.c_void_p,
                                     C.c_int64, C.c_void_p]),
    "zrk_selftest_host_wait": (C.c_int, [C.c_int, C.c_int]),
    "zrk_ccp_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "zrk_ccp_step_scratch_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "zrk_ccp_add_missile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p]),
    "zrk_ccp_requests": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p]),
    "zrk_read_sweep_ticks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "zrk_last_run_ticks_per_launch": (C.c_int, [C.c_void_p]),
    "zrk_sweep_stamps": (C.c_int, [C.c_void_p, C.c_int]),
    "zrk_read_sweep_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "zrk_last_sweep_stamp_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


def build(force: bool = False) -> Path:
    """Compile csrc/zrk_hot.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    src = CSRC / "zrk_hot.hip"
    hdr = CSRC.parent.parent / "include" / "zrk_hot.h"
    stale = (not LIB_PATH.exists()
             or LIB_PATH.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime))
    if force or stale:
        subprocess.check_call(["make", "-C", str(CSRC)] + (["-B"] if force else []) + ["libzrk_hot.so"])
    return LIB_PATH


def load():
    """dlopen libzrk_hot.so and attach prototypes.  Raises HotPathUnavailable, never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own HIP runtime; whichever copy enters the process first is the one everybody binds
    # to, and a process where this library pulled in the system's copy before torch initialised its own has
    # been seen to end up with no usable device.  The device memory handed across the ABI is torch's anyway.
    import torch  # noqa: F401
    if not LIB_PATH.exists():
        raise HotPathUnavailable(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {CSRC}`; there is no CPU fallback for the hot path")
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as e:
        raise HotPathUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HotPathUnavailable(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.zrk_abi_version() != ZRK_ABI_VERSION:
        raise HotPathUnavailable(f"ABI mismatch: library {lib.zrk_abi_version()}, binding {ZRK_ABI_VERSION}")
    _lib = lib
    return lib


class ZrkError(RuntimeError):
    pass


class Context:
    """Owns one zrk_ctx on one device."""

    def __init__(self, device_index: int):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.zrk_ctx_create(int(device_index), C.byref(h))
        if rc != 0 or not h.value:
            raise HotPathUnavailable(f"zrk_ctx_create(device={device_index}) failed with {rc}: no usable HIP device")
        self.handle = h

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.zrk_last_error(self.handle)
            raise ZrkError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.zrk_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
