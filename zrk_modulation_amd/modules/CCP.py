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

    # bookkeeping -------------------------------------------------------------------------------
    def add_target(self, target_ccp: TargetCCP):
        self._target_dict[target_ccp.target.id] = target_ccp

    def delete_target(self, target_id):
        self._target_dict.pop(target_id, None)

    def add_missile(self, missile_ccp: MissileCCP):
        m = missile_ccp.missile
        self._missile_dict[m.id] = missile_ccp
        self._target_dict[m.target.id].upd_missile_id(m.id)

    def delete_missile(self, missile_id: int, self_detonation: bool):
        if not self_detonation:
            self.delete_target(self._missile_dict[missile_id].missile.target.id)
        self._missile_dict.pop(missile_id, None)

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
        out = []
        for m in match:
            if m < 0:
                out.append((NEW_TARGET, None))
            elif m < len(t_keys):
                out.append((OLD_TARGET, t_keys[m]))
            else:
                out.append((OLD_ROCKET, m_keys[m - len(t_keys)]))
        return out

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
        for msg in mgr.give_messages_by_type(MessageType.FOUND_OBJECTS):
            radar_id = msg.sender_id
            for obj in msg.visible_objects:
                if obj.id in seen:
                    continue
                seen.add(obj.id)
                processed.append(obj.id)
                dets.append((obj, radar_id))
        # every verdict of the tick at once on the device; a verdict depends on earlier ones only through tracks they
        # took (updated "now", hence skipped), which the device resolution reproduces
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
