"""Combat control point: track association and launch decisions (behavioural counterpart of
reference modules/CCP.py:15-431).  The bookkeeping and the messages are host-side scalar logic as in the reference;
the part that is quadratic there -- `link_object` for every detection against every track (reference :171-219) -- runs
on the device for all detections of a tick at once when the objects live in a device table
(zrk_modulation_amd/association.py, zrk_ccp_link), with the result of the reference's sequential loop.

What it consumes from the hot path every tick: AllObjectsMessage / FoundObjectsMessage per radar and
the entity handles inside them (.id .pos .prev_pos .speed_mod .type).  Observable quirks of the
reference are kept: a DestroyedMissileId never matches because its missile_id is a 1-tuple
(SURVEY.md 5.9-5), so tracked targets and missiles are never forgotten.
"""
import logging

import numpy as np

from .BaseModel import BaseModel
from .constants import MANAGER_ID, POSSIBLE_TARGET_RADIUS, MessageType
from .Messages import (CPPDrawerObjectsMessage, CPPLaunchMissileRequestMessage, CPPUpdateTargetRadarMessage,
                       MissileCountRequestMessage)
from .utils import Target, to_seconds

OLD_TARGET = "старая цель"
NEW_TARGET = "новая цель"
OLD_ROCKET = "старая ЗУР"
MISSILE_LABEL = "ЗУР"

logger = logging.getLogger(__name__)


class TargetCCP:
    """A tracked target: last handle seen, when, and whether a missile is already after it."""

    def __init__(self, target, time, following) -> None:
        self.target = target
        self.upd_time = time
        self.following = following
        self.missile_id = None

    def upd_target_ccp(self, target, time, upd_follow) -> None:
        self.target, self.upd_time, self.following = target, time, upd_follow

    def upd_missile_id(self, missile_id) -> None:
        self.missile_id = missile_id


class MissileCCP:
    """A tracked missile of our own."""

    def __init__(self, missile, time):
        self.missile = missile
        self.upd_time = time
        self.target_id = missile.target.id

    def upd_missile_ccp(self, missile, time) -> None:
        self.missile, self.upd_time = missile, time


class CombatControlPoint(BaseModel):
    def __init__(self, manager, id: int, missile_launcher_coords: dict, radars_coords: dict, position: np.ndarray):
        super().__init__(manager, id, position)
        self._target_dict = {}
        self._missile_dict = {}
        self.radars_coords = radars_coords
        self.missile_launcher_coords = missile_launcher_coords
        self.missile_launcher_launched = {}
        self.missile_launcher_capacity = {}
        self.initialized = False
        self.device_ticks = 0           # ticks whose detection loop ran on the device (zrk_ccp_step)
        self.host_ticks = {}            # ... and why the others went through the host loop: reason -> ticks
        self._post = None               # the dictionaries' mirror on the device (association.DeviceCommandPost), made on first use
        self._post_stale = True         # the host changed a dictionary behind its back: rebuilt from the dictionaries
        self._post_store, self._post_rows_version = None, -1    # the table the mirror was built from, and its rows_version then

    # bookkeeping -------------------------------------------------------------------------------
    def add_target(self, target_ccp: TargetCCP):
        self._target_dict[target_ccp.target.id] = target_ccp

    def delete_target(self, target_id):
        self._target_dict.pop(target_id, None)
        self._post_stale = True

    def add_missile(self, missile_ccp: MissileCCP):
        m = missile_ccp.missile
        self._missile_dict[m.id] = missile_ccp
        self._target_dict[m.target.id].upd_missile_id(m.id)

    def delete_missile(self, missile_id: int, self_detonation: bool):
        if not self_detonation:
            self.delete_target(self._missile_dict[missile_id].missile.target.id)
        self._missile_dict.pop(missile_id, None)
        self._post_stale = True

    def _now_s(self):
        return to_seconds(self._manager.time.get_time())

    # inbound -----------------------------------------------------------------------------------
    def send_request_msg_to_ml_capacity(self):
        now = self._manager.time.get_time()
        for ml_id in self.missile_launcher_coords:
            self.missile_launcher_capacity[ml_id] = 0
            self.missile_launcher_launched[ml_id] = 0
            self._manager.add_message(MissileCountRequestMessage(time=now, sender_id=self.id, receiver_id=ml_id))

    def get_current_missile_launcher_capacity(self):
        for msg in self._manager.give_messages_by_type(MessageType.MISSILE_COUNT_RESPONSE):
            self.missile_launcher_capacity[msg.sender_id] = msg.count

    def check_if_missile_get_hit(self):
        for msg in self._manager.give_messages_by_type(MessageType.DESTROYED_MISSILE):
            if msg.missile_id in self._missile_dict:          # a 1-tuple is never a key: see module docstring
                self.delete_missile(msg.missile_id, msg.self_detonation)

    def check_if_missiles_launched(self):
        for msg in self._manager.give_messages_by_type(MessageType.LAUNCHED_MISSILE):
            self.add_missile(MissileCCP(msg.missile, self._now_s()))
            # the device mirror takes the one new entry (zrk_ccp_add_missile) when the missile already is a row of its table;
            # a missile that has not entered the air yet (the launcher announces it a tick before AirEnv appends it,
            # MissileLauncher.py:103-124) has no row: the mirror is rebuilt from the dictionaries then
            post, m = self._post, msg.missile
            if (post is not None and not self._post_stale and getattr(m, "_store", None) is not None and m._slot >= 0
                    and getattr(m, "_frozen", None) is None and m._store is getattr(self, "_post_store", None)
                    and m._store.rows_version == getattr(self, "_post_rows_version", -1)):
                post.add_missile(m._slot, self._now_s())
            else:
                self._post_stale = True

    # track association ---------------------------------------------------------------------------
    def _gate(self, detected, ref_pos, ref_time):
        """(min range, max range, distance) of `detected` from a track last updated at ref_time."""
        speed = detected.speed_mod
        slack = POSSIBLE_TARGET_RADIUS * to_seconds(self._manager.time.get_dt())
        age = self._now_s() - ref_time
        dist = np.linalg.norm(ref_pos - detected.pos)
        return max(0, speed * (age - slack)), max(0, speed * (age + slack)), dist

    def link_object(self, detected_object):
        best, verdict, match = float("inf"), NEW_TARGET, None
        now_s = self._now_s()
        for tid, track in self._target_dict.items():
            if track.upd_time == now_s:
                continue
            lo, hi, dist = self._gate(detected_object, track.target.prev_pos, track.upd_time)
            if dist < best and lo <= dist <= hi:
                best, verdict, match = dist, OLD_TARGET, tid
        for mid, track in self._missile_dict.items():
            if track.upd_time == now_s:
                continue
            ref = track.missile.prev_pos
            if ref is None:
                ref = track.missile.pos
            lo, hi, dist = self._gate(detected_object, ref, track.upd_time)
            if dist < best and lo <= dist <= hi:
                best, verdict, match = dist, OLD_ROCKET, mid
        return verdict, match

    def _link_all(self, objs):
        """link_object for every detection of the tick, in order: [(verdict, matched id | None)].  On the device when the
        detections are rows of a device table (one call: the pairwise distances and the order-dependent resolution,
        zrk_ccp_link); otherwise -- plain host objects -- the reference's loop, one link_object at a time as the caller
        applies the results."""
        import os
        store = getattr(objs[0], "_store", None) if objs else None
        if store is None or os.environ.get("ZRK_CCP_HOST") or any(getattr(o, "_store", None) is None for o in objs):
            return None
        now_s = self._now_s()
        t_keys, m_keys = list(self._target_dict), list(self._missile_dict)
        refs, upds = [], []
        for k in t_keys:
            tr = self._target_dict[k]
            ref = tr.target.prev_pos
            if ref is None and tr.upd_time != now_s:
                return None                       # the reference would fail on `None - pos` here: let the host loop do so
            refs.append(ref if ref is not None else np.zeros(3)); upds.append(tr.upd_time)
        for k in m_keys:
            tr = self._missile_dict[k]
            ref = tr.missile.prev_pos
            refs.append(tr.missile.pos if ref is None else ref); upds.append(tr.upd_time)
        from ..association import link_all
        slack = POSSIBLE_TARGET_RADIUS * to_seconds(self._manager.time.get_dt())
        match = link_all(store.ctx, store.device, np.asarray([o.pos for o in objs], np.float64).reshape(-1, 3),
                         np.asarray([o.speed_mod for o in objs], np.float64),
                         np.asarray(refs, np.float64).reshape(-1, 3), np.asarray(upds, np.float64), now_s, slack)
        # a NEW_TARGET verdict for an id that is a key already replaces that entry on the spot (add_target, reference :88-93):
        # the old track is then updated "now" and gone for every LATER detection of the tick, which this batched link does
        # not know -- if a later detection matched it, the tick goes through the sequential loop
        index_of_key = {k: i for i, k in enumerate(t_keys)}
        for d, m in enumerate(match):
            if m < 0 and objs[d].id in index_of_key and (match[d + 1:] == index_of_key[objs[d].id]).any():
                return None
        out = []
        for m in match:
            if m < 0:
                out.append((NEW_TARGET, None))
            elif m < len(t_keys):
                out.append((OLD_TARGET, t_keys[m]))
            else:
                out.append((OLD_ROCKET, m_keys[m - len(t_keys)]))
        return out

    # the detection loop on the device -------------------------------------------------------------
    def _device_tick(self, found_msgs, dets):
        """CombatControlPoint.step's detection loop for this tick on the device (zrk_ccp_step: link_object, the launcher
        choice and the dictionaries' updates with the sequential loop's result); the host then only walks the verdicts to
        keep its own dictionaries (they hold the handles the messages carry) and to send the messages.  Returns False when
        the tick has to go through the host loop instead (objects that are not rows of one device table)."""
        import os
        import torch
        from ..association import DeviceCommandPost
        def host(why):
            self.host_ticks[why] = self.host_ticks.get(why, 0) + 1
            return False
        if os.environ.get("ZRK_CCP_HOST") or not dets:
            return host("forced" if dets else "no detections")
        store = getattr(found_msgs[0], "device_store", None)
        if store is None or any(getattr(m, "device_store", None) is not store for m in found_msgs):
            return host("no device table")
        if any(getattr(o, "_store", None) is not store or o._frozen is not None for o, _ in dets):
            return host("object outside the table")
        now_s = self._now_s()
        ml_ids = list(self.missile_launcher_coords)
        if len(ml_ids) > 64:
            return host("launchers")
        # the detections of the tick in processing order, each row once (:414), from the radars' device lists
        seq_all = torch.cat([m.device_rows for m in found_msgs]).to(torch.int64)
        pos_in_seq = torch.arange(seq_all.numel(), device=seq_all.device)
        first = torch.full((store.cap,), seq_all.numel(), dtype=torch.int64, device=seq_all.device)
        first.scatter_reduce_(0, seq_all, pos_in_seq, "amin")
        seq = seq_all[first[seq_all] == pos_in_seq].to(torch.int32).contiguous()
        if seq.numel() != len(dets):
            return host("row lists")
        cnt = torch.tensor([seq.numel()], dtype=torch.int32, device=seq.device)
        # the mirror of the dictionaries: made, or rebuilt when the host changed them (missiles entering, deletions)
        cap = max(2 * store.cap, 64)
        if self._post is None or self._post.dev != store.device or self._post.key_tt.numel() < store.cap or self._post.tcap < cap:
            self._post = DeviceCommandPost(store.ctx, store.device, store.cap, cap, list(self.missile_launcher_coords.values()),
                                           [0] * len(ml_ids), dmax=store.cap)
            self._post_speed_dev = torch.zeros(store.cap, dtype=torch.float64, device=store.device)
            self._post_speed_known = np.zeros(store.cap, bool)
            self._post_stale = True
            self._post_store, self._post_rows_version = store, store.rows_version
        if self._post_store is not store or self._post_rows_version != store.rows_version:
            # rows were given out again (a salvo's dead rows dropped, padding rows revived): what the device column holds
            # for them is the old occupant's speed
            self._post_speed_known[:] = False
            self._post_stale = True
            self._post_store, self._post_rows_version = store, store.rows_version
        post = self._post
        # speed_mod is a column of the device post, written once per row when the row is first detected
        fresh = [o for o, _ in dets if not self._post_speed_known[o._slot]]
        if fresh:
            rows_new = torch.tensor([o._slot for o in fresh], dtype=torch.int64, device=store.device)
            self._post_speed_dev[rows_new] = torch.tensor([o.speed_mod for o in fresh], dtype=torch.float64, device=store.device)
            for o in fresh:
                self._post_speed_known[o._slot] = True
        if self._post_stale and not self._mirror_dictionaries(post, store, now_s):
            return host("mirror")
        if ml_ids:
            post.l_cap.copy_(torch.tensor([self.missile_launcher_capacity[k] for k in ml_ids], dtype=torch.int32))
            post.l_launched.copy_(torch.tensor([self.missile_launcher_launched[k] for k in ml_ids], dtype=torch.int32))
        speed = self._post_speed_dev
        slack = POSSIBLE_TARGET_RADIUS * to_seconds(self._manager.time.get_dt())
        post.step(store.ents, store.cur, speed, seq, cnt, now_s, slack)
        try:
            rows, verdict, match, launcher = post.results()
        except Exception as e:
            self._post_stale = True          # (nothing applied on the host: the host loop takes the tick)
            return host(f"step: {e}")
        t_keys, m_keys = list(self._target_dict), list(self._missile_dict)
        for d, (obj, radar_id) in enumerate(dets):
            assert int(rows[d]) == obj._slot
            launched = launcher[d] >= 0
            if launched:
                ml = ml_ids[int(launcher[d])]
                self.missile_launcher_launched[ml] += 1
                self._manager.add_message(CPPLaunchMissileRequestMessage(
                    time=self._manager.time.get_time(), sender_id=self.id, receiver_id=ml, target=obj, target_position=obj.pos,
                    radar_id=radar_id))
            if verdict[d] == 0:
                self.add_target(TargetCCP(obj, now_s, bool(launched)))
            elif verdict[d] == 1:
                track = self._target_dict[t_keys[int(match[d])]]
                if not track.following:
                    track.upd_target_ccp(obj, now_s, bool(launched))
                else:
                    track.upd_target_ccp(obj, now_s, track.following)
                    self.send_update_msg_to_radar(obj, track.missile_id, radar_id)
            else:
                self._missile_dict[m_keys[int(match[d])]].upd_missile_ccp(obj, now_s)
        self.device_ticks += 1
        return True

    def _mirror_dictionaries(self, post, store, now_s):
        """The dictionaries as the device arrays of zrk_ccp_step, in dictionary order.  False: some handle is not a row of
        this table (the host loop then)."""
        import torch
        t_tracks, m_tracks = list(self._target_dict.values()), list(self._missile_dict.values())
        handles = [tr.target for tr in t_tracks] + [tr.missile for tr in m_tracks]
        if any(getattr(h, "_store", None) is not store for h in handles) or max(len(t_tracks), len(m_tracks)) > post.tcap:
            return False
        dev = post.dev
        i32 = lambda xs: torch.tensor(list(xs), dtype=torch.int32, device=dev)         # noqa: E731
        nt, nm = len(t_tracks), len(m_tracks)
        # (a key may name an object no track holds any more -- replaced in place, :88-93 -- or one that has left the air: the
        # table still knows its row, and rows are never given out twice)
        def slot_of(obj_id):
            rows = store.slots_for_id(obj_id)
            return int(rows[0]) if len(rows) else -1
        post.key_tt.fill_(-1)
        post.tt_ref_fixed.fill_(float("nan")); post.tm_ref_fixed.fill_(float("nan"))
        if nt:
            keys = [slot_of(k) for k in self._target_dict]
            if min(keys) < 0:
                return False
            post.tt_key[:nt] = i32(keys); post.tt_obj[:nt] = i32(tr.target._slot for tr in t_tracks)
            post.tt_upd[:nt] = torch.tensor([tr.upd_time for tr in t_tracks], dtype=torch.float64, device=dev)
            post.tt_follow[:nt] = torch.tensor([1 if tr.following else 0 for tr in t_tracks], dtype=torch.uint8, device=dev)
            post.key_tt[torch.tensor(keys, dtype=torch.int64, device=dev)] = torch.arange(nt, dtype=torch.int32, device=dev)
        if nm:
            post.tm_key[:nm] = i32(slot_of(k) for k in self._missile_dict)
            post.tm_obj[:nm] = i32(tr.missile._slot for tr in m_tracks)
            post.tm_upd[:nm] = torch.tensor([tr.upd_time for tr in m_tracks], dtype=torch.float64, device=dev)
        # tracks whose object has left the air: what the handle still holds as prev_pos (the table has lost it)
        for k, tr in enumerate(t_tracks):
            if tr.target._frozen is not None:
                ref = tr.target.prev_pos
                if ref is None:
                    return False
                post.tt_ref_fixed[k] = torch.tensor(np.asarray(ref, np.float64), device=dev)
        for k, tr in enumerate(m_tracks):
            if tr.missile._frozen is not None:
                ref = tr.missile.prev_pos
                post.tm_ref_fixed[k] = torch.tensor(np.asarray(tr.missile.pos if ref is None else ref, np.float64), device=dev)
        post.counts.copy_(torch.tensor([nt, nm], dtype=torch.int32))
        self._post_stale = False
        self._post_frozen = {id(h) for h in handles if h._frozen is not None}
        return True

    # outbound ----------------------------------------------------------------------------------
    def send_update_msg_to_radar(self, target, missile_id, radar_id):
        self._manager.add_message(CPPUpdateTargetRadarMessage(
            time=self._manager.time.get_time(), sender_id=self.id, receiver_id=radar_id, target=target,
            missile_id=missile_id))

    def _draw(self, obj_id, kind, coordinates, visible):
        self._manager.add_message(CPPDrawerObjectsMessage(
            time=self._manager.time.get_time(), sender_id=self.id, receiver_id=MANAGER_ID, obj_id=obj_id,
            target_type=kind, coordinates=coordinates, is_visible_by_radar=visible))

    def send_objects_to_GUI(self, all, visible):
        for track in self._missile_dict.values():
            if track.missile.id in visible:
                self._draw(track.missile.id, MISSILE_LABEL, track.missile.pos, True)
        for track in self._target_dict.values():
            if track.target.id in visible:
                self._draw(track.target.id, track.target.type, track.target.pos, True)
        for obj_id, kind, coord in all:
            if obj_id not in visible:
                self._draw(obj_id, kind, coord, False)

    def try_to_launch_missile(self, obj, radar_id):
        best, chosen = float("inf"), None
        for ml_id, ml_pos in self.missile_launcher_coords.items():
            if self.missile_launcher_launched[ml_id] < self.missile_launcher_capacity[ml_id]:
                d = (np.sum((ml_pos - obj.pos) ** 2)) ** 0.5
                if d < best:
                    best, chosen = d, ml_id
        if chosen is None:
            return False
        self.missile_launcher_launched[chosen] += 1
        self._manager.add_message(CPPLaunchMissileRequestMessage(
            time=self._manager.time.get_time(), sender_id=self.id, receiver_id=chosen, target=obj,
            target_position=obj.pos, radar_id=radar_id))
        return True

    def new_target(self, obj, radar_id):
        self.add_target(TargetCCP(obj, self._now_s(), self.try_to_launch_missile(obj, radar_id)))

    def old_target(self, obj, old_obj_id, radar_id):
        track = self._target_dict[old_obj_id]
        if not track.following:
            track.upd_target_ccp(obj, self._now_s(), self.try_to_launch_missile(obj, radar_id))
        else:
            track.upd_target_ccp(obj, self._now_s(), track.following)
            self.send_update_msg_to_radar(obj, track.missile_id, radar_id)

    def old_rocket(self, obj, old_obj_id):
        self._missile_dict[old_obj_id].upd_missile_ccp(obj, self._now_s())

    # tick --------------------------------------------------------------------------------------
    def step(self) -> None:
        mgr = self._manager
        if not self.initialized:
            self.send_request_msg_to_ml_capacity()
            self.initialized = True
        self.get_current_missile_launcher_capacity()
        self.check_if_missile_get_hit()
        self.check_if_missiles_launched()

        to_draw, seen_ids = [], []
        for msg in mgr.give_messages_by_type(MessageType.ALL_OBJECTS):
            for obj in msg.objects:
                if obj.id not in seen_ids:
                    kind = obj.type if isinstance(obj, Target) else MISSILE_LABEL
                    to_draw.append([obj.id, kind, obj.pos])
                    seen_ids.append(obj.id)

        processed, seen, dets = [], set(), []
        found_msgs = mgr.give_messages_by_type(MessageType.FOUND_OBJECTS)
        for msg in found_msgs:
            radar_id = msg.sender_id
            for obj in msg.visible_objects:
                if obj.id in seen:
                    continue
                seen.add(obj.id)
                processed.append(obj.id)
                dets.append((obj, radar_id))
        # handles that have left the air since the mirror was made hold a prev_pos the table has lost: mirror again
        if self._post is not None and not self._post_stale:
            frozen = {id(tr.target) for tr in self._target_dict.values() if getattr(tr.target, "_frozen", None) is not None}
            frozen |= {id(tr.missile) for tr in self._missile_dict.values() if getattr(tr.missile, "_frozen", None) is not None}
            if frozen != getattr(self, "_post_frozen", set()):
                self._post_stale = True
        # the whole loop on the device where the objects are rows of one device table (zrk_ccp_step) ...
        if self._device_tick(found_msgs, dets):
            self.send_objects_to_GUI(to_draw, processed)
            return
        if dets:
            self._post_stale = True          # (the host loop below changes the dictionaries behind the mirror's back)
        # ... otherwise every verdict of the tick at once on the device (zrk_ccp_link: a verdict depends on earlier ones only
        # through tracks they took -- updated "now", hence skipped --, which the device resolution reproduces), or the
        # reference's loop
        verdicts = self._link_all([o for o, _ in dets]) if dets else []
        for k, (obj, radar_id) in enumerate(dets):
            verdict, old_id = verdicts[k] if verdicts is not None else self.link_object(obj)
            if verdict == NEW_TARGET:
                self.new_target(obj, radar_id)
            elif verdict == OLD_TARGET:
                self.old_target(obj, old_id, radar_id)
            elif verdict == OLD_ROCKET:
                self.old_rocket(obj, old_id)

        self.send_objects_to_GUI(to_draw, processed)
