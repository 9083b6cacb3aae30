"""Missile: ready -> active -> detonated (reference modules/Missile.py:15-196).

The flight itself ('active' branch, :162-193) is one row of the device missile table stepped by
AirEnv; this class is its Python identity plus the launch transition.  `_launch` runs the
lead-collision solve (`_calculate_trajectory_params`, :35-102) on the device against the target's
current table row, so the velocity it flies with is the one the fuse kernel integrates.
"""
from typing import Optional, Tuple

import numpy as np

from ..store import LAUNCH_ERRORS
from .AirObject import AirObject, Trajectory
from .constants import (MISSILE_DETONATE_PERIOD, MISSILE_DETONATE_RADIUS, MISSILE_VELOCITY_MODULE, MessageType)
from .utils import to_seconds


class InterceptionError(Exception):
    pass


class Missile(AirObject):
    def __init__(self, manager, id: int, pos: tuple = (0, 0, 0), velocity_module: float = MISSILE_VELOCITY_MODULE,
                 detonate_radius: float = MISSILE_DETONATE_RADIUS, detonate_period: float = MISSILE_DETONATE_PERIOD):
        super().__init__(manager, id, np.array(pos), Trajectory(start_pos=pos))   # zero velocity -> NaN unit vector
        self.speed_mod = velocity_module
        self.detonate_radius = detonate_radius
        self._detonate_period = detonate_period
        self.status = "ready"
        self.launch_time: Optional[float] = None
        self.target: Optional[AirObject] = None
        self._row = -1                   # row in the device missile table once in the air

    # the life timer is decremented on the device while the missile flies
    @property
    def detonate_period(self):
        if self._row >= 0 and self._store is not None:
            return float(self._store.dm_period[self._row].item())
        return self._detonate_period

    @detonate_period.setter
    def detonate_period(self, value):
        self._detonate_period = value
        if self._row >= 0 and self._store is not None:
            self._store.dm_period[self._row] = float(value)

    def _period_at_launch(self):
        return self._detonate_period

    def _calculate_trajectory_params(self, target: AirObject) -> Tuple[np.ndarray, float]:
        store = getattr(target, "_store", None)
        if store is None:
            raise RuntimeError("the intercept solve runs on the device: the target must belong to an AirEnv")
        rc, V, t = store.launch_solve(target._slot, np.asarray(self.pos, np.float64), self.speed_mod,
                                      self._detonate_period)
        if rc != 0:
            if rc == 5:
                print("wtf")             # the reference prints this before raising (Missile.py:93)
            raise ValueError(LAUNCH_ERRORS[rc])
        return V, t

    def _launch(self, target: AirObject, launcher_id):
        from .Messages import MissileLaunchCancelledMessage, MissileSuccessfulLaunchMessage
        mgr = self._manager
        try:
            V, _t = self._calculate_trajectory_params(target)
        except (InterceptionError, ValueError) as e:
            mgr.add_message(MissileLaunchCancelledMessage(sender_id=self.id, reason=str(e), missile=self,
                                                          receiver_id=launcher_id))
            return
        self.target = target
        now_s = to_seconds(mgr.time.get_time())
        self._set_trajectory(Trajectory(velocity=tuple(V), start_pos=tuple(self.pos), start_time=now_s))
        self.launch_time = now_s
        mgr.add_message(MissileSuccessfulLaunchMessage(sender_id=self.id, launch_time=now_s, target=target,
                                                       missile=self, receiver_id=launcher_id))
        self.status = "active"

    def _set_trajectory(self, new_trajectory: Trajectory):
        self.trajectory = new_trajectory          # velocity / speed_mod deliberately not refreshed

    def _detonate(self, target_id: int = None, self_detonation: bool = True):
        from .Messages import MissileDetonateMessage
        self._manager.add_message(MissileDetonateMessage(sender_id=self.id, target_id=target_id,
                                                         self_detonation=self_detonation))
        self.status = "detonated"

    def step(self):
        if self.status == "ready":
            mgr = self._manager
            orders = mgr.give_messages_by_type(MessageType.LAUNCH_MISSILE, self.id, step_time=mgr.time.get_time())
            if orders:
                self._launch(orders[-1].target, orders[-1].sender_id)
        elif self.status == "active":
            raise RuntimeError("an active missile is stepped on the device by AirEnv.step()")
