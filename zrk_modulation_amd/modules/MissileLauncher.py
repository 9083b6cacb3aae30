"""Missile launcher: magazine + launch relay (behavioural counterpart of reference
modules/MissileLauncher.py:15-172; host-side, event-rate, outside the accelerated path).

Kept message-for-message compatible so the stock scenarios run end to end on top of the device
AirEnv: a launch request pops the LAST missile of the magazine, posts LaunchMissileMessage and steps
that missile once (reference :58-80); a cancelled launch puts the missile back at the end (:126-129).
"""
import logging
from typing import List, Optional

import numpy as np

from .BaseModel import BaseModel
from .constants import CCP_ID
from .Messages import (CPPLaunchMissileRequestMessage, LaunchedMissileMessage, LaunchMissileMessage,
                       MissileCountRequestMessage, MissileCountResponseMessage, MissileLaunchCancelledMessage,
                       MissileSuccessfulLaunchMessage, MissileToAirEnvMessage)
from .Missile import Missile

logger = logging.getLogger(__name__)


class MissileLauncher(BaseModel):
    def __init__(self, manager, id: int, pos: np.ndarray, max_missiles: int = 5, air_env=None) -> None:
        super().__init__(manager, id, pos)
        self.max_missiles = max_missiles
        self.missiles: List[Missile] = []             # magazine
        self.launched_missiles: List[Missile] = []
        self.air_env = air_env

    # magazine ----------------------------------------------------------------------------------
    def add_missile(self, missile: Missile) -> bool:
        if len(self.missiles) >= self.max_missiles:
            return False
        self.missiles.append(missile)
        return True

    def count_missiles(self) -> int:
        return len(self.missiles)

    get_missile_count = count_missiles

    def launch_missile(self, target, target_id: Optional[int] = None, radar_id: Optional[int] = None):
        if not self.missiles:
            logger.info("launcher %s: magazine empty", self.id)
            return None
        missile = self.missiles.pop()
        self._manager.add_message(LaunchMissileMessage(receiver_id=missile.id, sender_id=self.id, target=target))
        missile.step()          # 'ready' branch: reads the order just posted and solves the intercept
        return None             # the reference returns None on every path

    # tick --------------------------------------------------------------------------------------
    def step(self) -> None:
        mgr = self._manager
        now, dt = mgr.time.get_time(), mgr.time.get_dt()
        for msg in mgr.give_messages_by_id(self.id, step_time=now - dt):
            if isinstance(msg, CPPLaunchMissileRequestMessage):
                self.launch_missile(target=msg.target, radar_id=msg.radar_id)
            elif isinstance(msg, MissileSuccessfulLaunchMessage):
                missile = msg.missile
                self.launched_missiles.append(missile)
                mgr.add_message(LaunchedMissileMessage(sender_id=self.id, receiver_id=CCP_ID, missile=missile,
                                                       target_id=msg.target_id))
                mgr.add_message(MissileToAirEnvMessage(sender_id=self.id, missile=missile))
            elif isinstance(msg, MissileLaunchCancelledMessage):
                self.missiles.append(msg.missile)
            elif isinstance(msg, MissileCountRequestMessage):
                mgr.add_message(MissileCountResponseMessage(time=now, sender_id=self.id, receiver_id=msg.sender_id,
                                                            count=self.count_missiles()))

    def get_status(self) -> dict:
        return {
            "id": self.id,
            "position": np.asarray(self.pos).tolist(),
            "available_missiles": len(self.missiles),
            "launched_missiles": len(self.launched_missiles),
            "max_missiles": self.max_missiles,
            "missiles": [m.id for m in self.missiles],
            "active_missiles": [m.id for m in self.launched_missiles if m.status == "active"],
        }
