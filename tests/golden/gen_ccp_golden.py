#!/usr/bin/env python3
"""Golden vectors for the command post's per-tick loop, captured from the reference itself.

Runs ONLY in the build container (the reference is mounted read-only at /root/reference).  Drives the reference's own
`CombatControlPoint.step` (modules/CCP.py:368-431) with synthetic radar messages over a few dozen ticks and records what
it decided: for every detection, in the order the reference processed them, the verdict of `link_object` (:171-219), the
track it matched, and the launcher `try_to_launch_missile` (:287-320) sent the request to.  No source text is stored:
inputs are object tables (ids, positions, speeds) and detection sequences, outputs are integers.

    python tests/golden/gen_ccp_golden.py        # writes ccp_step.npz next to this file

Fixture layout (ticks concatenated, ragged arrays with offsets):
    meta              JSON: dt_ms, launcher ids / positions / capacities, POSSIBLE_TARGET_RADIUS
    obj_id[N], obj_kind[N]                     the object table (kind 0 target, 1 missile of our own)
    pos[T,N,3], prev[T,N,3], prev_none[T,N]    obj.pos / obj.prev_pos as the command post saw them in tick T
    speed[N]                                   obj.speed_mod
    alive_from[N]                              first tick an object exists (missiles appear when "launched")
    seq_off[T+1], seq_obj[...], seq_radar[...] objects in FoundObjectsMessage order, radar after radar (duplicates kept:
                                               the command post skips ids it has processed in this tick)
    new_missile_off[T+1], new_missile[...,2]   (object index of the missile, object index of its target): LAUNCHED_MISSILE
                                               messages delivered in that tick
    out_off[T+1], out_obj[...], out_verdict[...], out_match[...], out_launcher[...]
                                               per PROCESSED detection: object index, 0 new / 1 old target / 2 old missile,
                                               matched dict key as object index (-1), launcher id the request went to (-1)
"""
import json
import sys
import types
import warnings
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))
warnings.filterwarnings("ignore", category=RuntimeWarning)

from modules.Manager import Manager                              # noqa: E402
from modules import CCP as ref_ccp                               # noqa: E402
from modules.Messages import (AllObjectsMessage, FoundObjectsMessage, LaunchedMissileMessage,   # noqa: E402
                              MissileCountResponseMessage)
from modules.constants import CCP_ID, POSSIBLE_TARGET_RADIUS, MessageType   # noqa: E402


def main(seed=11, n_targets=140, ticks=36, dt_ms=200):
    g = np.random.Generator(np.random.PCG64(seed))
    mgr = Manager()
    mgr.time.set_dt(dt_ms)
    launchers = {501: np.array([0.0, 0.0, 0.0]), 502: np.array([4000.0, -1500.0, 0.0]), 503: np.array([-2500.0, 3000.0, 10.0])}
    capacity = {501: 7, 502: 5, 503: 9}
    radars = {601: np.zeros(3), 602: np.array([1000.0, 0.0, 0.0])}
    ccp = ref_ccp.CombatControlPoint(mgr, CCP_ID, launchers, radars, np.zeros(3))
    n_missiles_max = sum(capacity.values())
    N = n_targets + n_missiles_max
    ids = np.concatenate([1000 + np.arange(n_targets), 9000 + np.arange(n_missiles_max)])
    kind = np.concatenate([np.zeros(n_targets, np.uint8), np.ones(n_missiles_max, np.uint8)])
    # a tight swarm: gates overlap, detections compete for tracks
    p0 = np.zeros((N, 3)); vel = np.zeros((N, 3))
    p0[:n_targets] = g.uniform(-4000, 4000, (n_targets, 3)) * [1, 1, 0.3] + [0, 0, 3000]
    vel[:n_targets] = g.normal(0, 220, (n_targets, 3)) * [1, 1, 0.2]
    speed = np.zeros(N)
    speed[:n_targets] = np.linalg.norm(vel[:n_targets], axis=1)
    objs = [types.SimpleNamespace(id=int(ids[i]), pos=p0[i].copy(), prev_pos=None, speed_mod=float(speed[i]), type="AIR_PLANE")
            for i in range(N)]
    alive_from = np.concatenate([np.zeros(n_targets, np.int64), np.full(n_missiles_max, 1 << 30)])
    index_of = {int(ids[i]): i for i in range(N)}

    # record what the reference decides
    log = []
    real_link = ccp.link_object
    real_try = ccp.try_to_launch_missile

    def link_spy(obj):
        verdict, key = real_link(obj)
        log.append([index_of[obj.id], {ref_ccp.NEW_TARGET: 0, ref_ccp.OLD_TARGET: 1, ref_ccp.OLD_ROCKET: 2}[verdict],
                    -1 if key is None else index_of[key], -1])
        return verdict, key

    def try_spy(obj, radar_id):
        before = dict(ccp.missile_launcher_launched)
        ok = real_try(obj, radar_id)
        if ok:
            chosen = [k for k in before if ccp.missile_launcher_launched[k] != before[k]]
            log[-1][3] = chosen[0]
        return ok

    ccp.link_object = link_spy
    ccp.try_to_launch_missile = try_spy

    pos_t, prev_t, none_t = [], [], []
    seq_off, seq_obj, seq_radar = [0], [], []
    nm_off, nm = [0], []
    out_off, out = [0], []
    pending_launch = []          # (due tick, launcher id, target object index)
    next_missile = n_targets
    for k in range(ticks):
        t_ms = k * dt_ms
        t = t_ms / 1000
        # the air picture of this tick: straight lines plus radar noise on what is seen (in place, like the reference's radar)
        for i in range(N):
            if alive_from[i] > k:
                continue
            o = objs[i]
            o.prev_pos = o.pos if k > alive_from[i] else None
            o.pos = p0[i] + vel[i] * (t - alive_from[i] * dt_ms / 1000)
        live = [i for i in range(N) if alive_from[i] <= k]
        per_radar = []
        for r, rid in enumerate(radars):
            seen = [i for i in live if g.uniform() < (0.55 if r == 0 else 0.35)]
            for i in seen:
                objs[i].pos = objs[i].pos + g.normal(0, 5, 3)
            per_radar.append((rid, seen))
        # messages of this tick
        if k == 1:
            for lid, c in capacity.items():
                mgr.add_message(MissileCountResponseMessage(time=t_ms, sender_id=lid, receiver_id=CCP_ID, count=c))
        launched_now = []
        for due, lid, tgt in [p for p in pending_launch if p[0] == k]:
            i = next_missile; next_missile += 1
            alive_from[i] = k
            p0[i] = launchers[lid]
            d = objs[tgt].pos - p0[i]
            vel[i] = d / np.linalg.norm(d) * 900.0
            speed[i] = 900.0
            objs[i].pos = p0[i].copy(); objs[i].prev_pos = None; objs[i].speed_mod = 900.0
            objs[i].target = objs[tgt]
            mgr.add_message(LaunchedMissileMessage(time=t_ms, sender_id=lid, receiver_id=CCP_ID, missile=objs[i], target_id=objs[tgt].id))
            launched_now.append((i, tgt))
        pending_launch = [p for p in pending_launch if p[0] != k]
        for rid, seen in per_radar:
            mgr.add_message(AllObjectsMessage(time=t_ms, sender_id=rid, receiver_id=CCP_ID, objects=[objs[i] for i in live]))
            mgr.add_message(FoundObjectsMessage(time=t_ms, sender_id=rid, receiver_id=CCP_ID, visible_objects=[objs[i] for i in seen]))
            seq_obj += seen; seq_radar += [rid] * len(seen)
        seq_off.append(len(seq_obj))
        nm += launched_now; nm_off.append(len(nm))
        pos_t.append(np.array([o.pos for o in objs])); none_t.append(np.array([o.prev_pos is None for o in objs]))
        prev_t.append(np.array([o.pos if o.prev_pos is None else o.prev_pos for o in objs]))
        log.clear()
        ccp.step()
        out += [list(x) for x in log]; out_off.append(len(out))
        # the launchers answer two ticks later, like the reference's (command -> launch -> in the air)
        # (the missile is aimed at the object the track is KEYED by: the reference's add_missile looks the key up, :107)
        for rec in log:
            if rec[3] >= 0:
                pending_launch.append((k + 2, rec[3], rec[0] if rec[1] == 0 else rec[2]))
        mgr.time.update_time()
    out = np.array(out, np.int64).reshape(-1, 4)
    meta = dict(dt_ms=dt_ms, launcher_ids=list(launchers), launcher_pos=[v.tolist() for v in launchers.values()],
                capacity=[capacity[k] for k in launchers], slack_steps=POSSIBLE_TARGET_RADIUS, seed=seed)
    np.savez_compressed(OUT / "ccp_step.npz", meta=json.dumps(meta), obj_id=ids, obj_kind=kind, pos=np.array(pos_t), prev=np.array(prev_t),
                        prev_none=np.array(none_t), speed=speed, alive_from=alive_from, seq_off=np.array(seq_off), seq_obj=np.array(seq_obj),
                        seq_radar=np.array(seq_radar), new_missile_off=np.array(nm_off), new_missile=np.array(nm, np.int64).reshape(-1, 2),
                        out_off=np.array(out_off), out_obj=out[:, 0], out_verdict=out[:, 1], out_match=out[:, 2], out_launcher=out[:, 3])
    v = out[:, 1]
    print(f"ccp_step.npz: {ticks} ticks, {len(out)} processed detections: {int((v == 0).sum())} new, {int((v == 1).sum())} old targets, "
          f"{int((v == 2).sum())} old missiles, {int((out[:, 3] >= 0).sum())} launch requests, {next_missile - n_targets} missiles in the air")


if __name__ == "__main__":
    main()
