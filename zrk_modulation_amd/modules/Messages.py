"""The seventeen message shapes of the bus (reference modules/Messages.py:14-256).

Constructor signatures and attribute names are the boundary contract: the hot-path modules emit
and consume eight of them, the command post, launchers and the GUI replay
the rest.  Attribute quirks that are observable are kept, notably DestroyedMissileId.missile_id
being a 1-tuple (reference Messages.py:160).
"""
from .BaseMessage import BaseMessage
from .constants import MessageType


def _ids(objs):
    return [o.id for o in objs] if objs else []


# launcher -> missile / command post -> launcher ---------------------------------------------------
class LaunchMissileMessage(BaseMessage):
    """MissileLauncher -> Missile: fire at `target`."""

    def __init__(self, sender_id, receiver_id=None, time=None, target=None):
        super().__init__(MessageType.LAUNCH_MISSILE, sender_id, receiver_id, send_time=time)
        self.target = target


class CPPLaunchMissileRequestMessage(BaseMessage):
    """CCP -> MissileLauncher: launch one missile at `target`."""

    def __init__(self, sender_id, target, target_position, radar_id, time=None, receiver_id=None, relevance=1):
        super().__init__(MessageType.LAUNCH_COMMAND, sender_id, receiver_id, send_time=time, relevance=relevance)
        self.target = target
        self.target_position = target_position
        self.radar_id = radar_id

    def __repr__(self):
        return (f"{super().__repr__()}, target={self.target}, target_position={self.target_position}, "
                f"radar_id={self.radar_id}")


class LaunchedMissileMessage(BaseMessage):
    """MissileLauncher -> CCP: a missile left the rail."""

    def __init__(self, sender_id, missile, target_id, time=None, receiver_id=None):
        super().__init__(MessageType.LAUNCHED_MISSILE, sender_id, receiver_id, send_time=time)
        self.missile = missile
        self.target_id = target_id

    def __repr__(self):
        mid = self.missile.id if self.missile else None
        return f"{super().__repr__()}, missile.id={mid}, target_id={self.target_id}"


class MissileCountRequestMessage(BaseMessage):
    """CCP -> MissileLauncher: how many missiles are left?  (the only relevance-3 message)"""

    def __init__(self, sender_id, time=None, receiver_id=None, relevance=3):
        super().__init__(MessageType.MISSILE_COUNT_REQUEST, sender_id, receiver_id, send_time=time, relevance=relevance)


class MissileCountResponseMessage(BaseMessage):
    """MissileLauncher -> CCP."""

    def __init__(self, sender_id, count, time=None, receiver_id=None):
        super().__init__(MessageType.MISSILE_COUNT_RESPONSE, sender_id, receiver_id, send_time=time)
        self.count = count

    def __repr__(self):
        return f"{super().__repr__()}, count={self.count}"


# radar -> command post ---------------------------------------------------------------------------
class AllObjectsMessage(BaseMessage):
    """Radar -> CCP: every live object (for drawing)."""

    def __init__(self, sender_id, objects, time=None, receiver_id=None):
        super().__init__(MessageType.ALL_OBJECTS, sender_id, receiver_id, send_time=time)
        self.objects = objects


class FoundObjectsMessage(BaseMessage):
    """Radar -> CCP: the objects inside the current sector, in list order."""

    def __init__(self, sender_id, visible_objects, time=None, receiver_id=None):
        super().__init__(MessageType.FOUND_OBJECTS, sender_id, receiver_id, send_time=time)
        self.visible_objects = visible_objects

    def __repr__(self):
        ids = _ids(self.visible_objects)
        return f"{super().__repr__()}, visible_objects.ids={ids}, visible_objects.count={len(ids)}"


class CPPUpdateTargetRadarMessage(BaseMessage):
    """CCP -> Radar: new target data for a missile in flight (never relayed on)."""

    def __init__(self, sender_id, target, missile_id, time=None, receiver_id=None):
        super().__init__(MessageType.CCP_UPDATE_TARGET, sender_id, receiver_id, send_time=time)
        self.target = target
        self.missile_id = missile_id

    def __repr__(self):
        tid = self.target.id if self.target else None
        return f"{super().__repr__()}, target.id={tid}, missile_id={self.missile_id}"


class ActiveObjectsMessage(BaseMessage):
    """AirEnv -> Radar: the live objects of this tick."""

    def __init__(self, sender_id, active_objects, time=None, receiver_id=None):
        super().__init__(MessageType.ACTIVE_OBJECTS, sender_id, receiver_id, send_time=time)
        self.active_objects = active_objects

    def __repr__(self):
        n = len(self.active_objects) if self.active_objects is not None else 0
        return f"{super().__repr__()}, active_objects.count={n}"


class UpdateTargetPosition(BaseMessage):
    """Radar -> Missile."""

    def __init__(self, sender_id, upd_object, time=None, receiver_id=None):
        super().__init__(MessageType.UPDATE_TARGET, sender_id, receiver_id, send_time=time)
        self.upd_object = upd_object

    def __repr__(self):
        o = self.upd_object
        return (f"{super().__repr__()}, upd_object.id={o.id if o else None}, "
                f"upd_object.pos={o.pos if o else None}")


class DestroyedMissileId(BaseMessage):
    """Radar -> CCP: a missile is gone."""

    def __init__(self, sender_id, missile_id, time=None, receiver_id=None, self_detonation=False):
        super().__init__(MessageType.DESTROYED_MISSILE, sender_id, receiver_id, send_time=time)
        self.missile_id = (missile_id,)          # 1-tuple, as the reference stores it
        self.self_detonation = self_detonation

    def __repr__(self):
        return f"{super().__repr__()}, missile_id={self.missile_id}"


# missile life cycle ------------------------------------------------------------------------------
class MissileDetonateMessage(BaseMessage):
    """Missile -> AirEnv, Radar: hit (target_id set, self_detonation False) or timeout."""

    def __init__(self, sender_id, target_id=None, self_detonation=False):
        super().__init__(MessageType.MISSILE_DETONATE, sender_id)
        self.missile_id = sender_id
        self.target_id = target_id
        self.self_detonation = self_detonation

    def __repr__(self):
        return f"{super().__repr__()}, missile_id={self.missile_id}, target_id={self.target_id}"


class MissilePosMessage(BaseMessage):
    """Missile -> AirEnv: still flying."""

    def __init__(self, sender_id):
        super().__init__(MessageType.MISSILE_POS, sender_id)
        self.missile_id = sender_id

    def __repr__(self):
        return f"{super().__repr__()}, missile_id={self.missile_id}"


class MissileSuccessfulLaunchMessage(BaseMessage):
    """Missile -> MissileLauncher."""

    def __init__(self, sender_id, launch_time, target, missile, receiver_id):
        super().__init__(MessageType.LAUNCH_SUCCESSFUL, sender_id, receiver_id)
        self.missile = missile
        self.launch_time = launch_time
        self.target_id = target.id

    def __repr__(self):
        return (f"{super().__repr__()}, missile_id={self.missile.id}, target_id={self.target_id}, "
                f"launch_time={self.launch_time}")


class MissileLaunchCancelledMessage(BaseMessage):
    """Missile -> MissileLauncher: the intercept solve failed, `reason` says why."""

    def __init__(self, sender_id, reason, missile, receiver_id):
        super().__init__(MessageType.LAUNCH_CANCELLED, sender_id, receiver_id)
        self.missile = missile
        self.reason = reason

    def __repr__(self):
        return f'{super().__repr__()}, missile_id={self.missile.id}, reason="{self.reason}"'


class CPPDrawerObjectsMessage(BaseMessage):
    """CCP -> GUI: one object to draw this tick."""

    def __init__(self, sender_id, obj_id, target_type, coordinates, is_visible_by_radar, time=None, receiver_id=None):
        super().__init__(MessageType.DRAW_OBJECTS, sender_id, receiver_id, send_time=time)
        self.obj_id = obj_id
        self.target_type = target_type
        self.coordinates = coordinates
        self.is_visible_by_radar = is_visible_by_radar

    def __repr__(self):
        return (f"{super().__repr__()}, obj_id={self.obj_id}, target_type={self.target_type}, "
                f"coordinates={self.coordinates}, is_visible_by_radar={self.is_visible_by_radar}")


class MissileToAirEnvMessage(BaseMessage):
    """MissileLauncher -> AirEnv: append this missile to the air picture next tick."""

    def __init__(self, sender_id, missile, time=None, receiver_id=None):
        super().__init__(MessageType.NEW_MISSILE, sender_id, receiver_id, send_time=time)
        self.missile = missile

    def __repr__(self):
        return f"{super().__repr__()}, missile.id={self.missile.id if self.missile else None}"
