"""Tick loop and message bus (reference modules/Manager.py:9-140).

Scalar orchestration: nothing here is data-parallel, so it stays on the host.  The contract the
hot-path modules rely on is reproduced exactly: per-tick message lists keyed by time, stable
relevance ordering on insert, filtering by type / receiver, and module scheduling by CLASS NAME
(SURVEY.md 5.9-1): AirEnv, SectorRadar, MissileLauncher, CombatControlPoint, then everything else.
"""
import logging
from typing import Dict, List, Optional

from .BaseMessage import BaseMessage
from .constants import MessageType
from .Timer import Timer

logger = logging.getLogger(__name__)

_SCHEDULE = ("AirEnv", "SectorRadar", "MissileLauncher", "CombatControlPoint")


def _rank(module) -> int:
    name = type(module).__name__
    return _SCHEDULE.index(name) if name in _SCHEDULE else len(_SCHEDULE)


class Manager:
    def __init__(self, replay=None):
        """replay: an optional zrk_modulation_amd.replay.ReplayLog.  With one, DRAW_OBJECTS messages (the only
        thing the GUI's replay reads, one per object per tick) are kept there as bounded columnar frames and
        `give_messages_by_type(DRAW_OBJECTS, step_time=...)` is answered from it; without, everything is kept
        as the reference does."""
        self.time = Timer()
        self.messages: Dict[int, List[BaseMessage]] = {}     # tick time -> messages (never pruned)
        self.modules: List = []
        self.replay = replay

    # modules -----------------------------------------------------------------------------------
    def add_module(self, module) -> None:
        if module in self.modules:
            logger.warning("module %s is already registered", getattr(module, "id", "unknown"))
            return
        self.modules.append(module)
        logger.info("module %s registered", getattr(module, "id", "unknown"))

    def remove_module(self, module_id: int) -> bool:
        for k, module in enumerate(self.modules):
            if getattr(module, "id", None) == module_id:
                del self.modules[k]
                return True
        return False

    def get_module_by_id(self, module_id: int):
        for module in self.modules:
            if getattr(module, "id", None) == module_id:
                return module
        return None

    # messages ----------------------------------------------------------------------------------
    def add_message(self, msg: BaseMessage, step_time: Optional[int] = None) -> None:
        when = self.time.get_time() if step_time is None else step_time
        if msg.send_time is None:
            msg.send_time = when
        if self.replay is not None and getattr(msg, "type", None) == MessageType.DRAW_OBJECTS:
            self.replay.add_message(when, msg)
            return
        bucket = self.messages.setdefault(when, [])
        bucket.append(msg)
        bucket.sort(key=lambda m: -m.relevance)               # stable: insertion order within a relevance

    def give_messages(self, step_time: Optional[int] = None) -> List[BaseMessage]:
        when = self.time.get_time() if step_time is None else step_time
        return self.messages.get(when, [])

    def give_messages_by_id(self, receiver_id: int, step_time: Optional[int] = None) -> List[BaseMessage]:
        return [m for m in self.give_messages(step_time) if m.receiver_id == receiver_id]

    def give_messages_by_type(self, msg_type: MessageType, receiver_id: Optional[int] = None,
                              step_time: Optional[int] = None) -> List[BaseMessage]:
        if self.replay is not None and msg_type == MessageType.DRAW_OBJECTS:
            when = self.time.get_time() if step_time is None else step_time
            return [m for m in self.replay.messages(when) if receiver_id is None or m.receiver_id == receiver_id]
        out = []
        for m in self.give_messages(step_time):
            if getattr(m, "type", None) == msg_type and (receiver_id is None or m.receiver_id == receiver_id):
                out.append(m)
        return out

    # loop --------------------------------------------------------------------------------------
    def run_simulation(self, end_time: int) -> None:
        while self.time.get_time() < end_time:
            now = self.time.get_time()
            for module in sorted(self.modules, key=_rank):     # stable: ties keep add_module order
                module.step()
            if logger.isEnabledFor(logging.INFO):
                batch = self.give_messages(now)
                logger.info("t=%d: %d messages", now, len(batch))
            self.time.update_time()
