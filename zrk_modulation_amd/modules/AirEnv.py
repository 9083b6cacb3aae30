"""AirEnv: the air picture, stepped on the device (reference modules/AirEnv.py:11-61).

Per tick (same order as the reference's step(), :26-53):
  1. MISSILE_DETONATE of the previous tick -> tombstone the missile and, for a hit, its target
     (by id, like `objects.id in objects_to_remove`)            zrk_kill_slots
  2. NEW_MISSILE of the previous tick -> append the missile        table append (event-rate)
  3. every live object steps in list order                        zrk_missile_step + zrk_tick_sweep(ADVANCE)
     - Target.step / AirObject.step: prev_pos, pos = trajectory(t)
     - Missile.step 'active': MissilePosMessage, proximity fuse, life timer
  4. ActiveObjectsMessage with the live objects
The object list lives in an EntityStore (structure-of-arrays tensors in HBM); the Python objects the
caller handed in stay the identities other modules hold on to.
"""
from collections.abc import Sequence
from typing import List

import numpy as np

from .._lib import F_ADVANCE
from ..store import EntityStore
from .AirObject import AirObject
from .BaseModel import BaseModel
from .constants import MessageType
from .Messages import ActiveObjectsMessage, MissileDetonateMessage, MissilePosMessage
from .utils import Target


class ActiveView(Sequence):
    """Read-only sequence of the live objects of one tick, in list order.  Cheap to retain (the bus
    never prunes messages): it shares the handle list and an index array instead of copying objects."""

    def __init__(self, store, handles, slots):
        self.store = store
        self._handles = handles
        self.slots = slots              # int64 array of live slots, ascending

    def __len__(self):
        return len(self.slots)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self._handles[s] for s in self.slots[k]]
        return self._handles[self.slots[k]]

    def __iter__(self):
        h = self._handles
        return (h[s] for s in self.slots)


class AirEnv(BaseModel):
    def __init__(self, manager, id: int, pos: np.ndarray, device=None) -> None:
        super().__init__(manager, id, pos)
        self._device = device
        self._store = None
        self._handles: List[AirObject] = []          # slot -> object
        self._missile_rows = []                      # in-flight missiles in table-row order
        self._live_slots = np.zeros(0, np.int64)
        self._live_dirty = True

    # the device table is created on first use so that importing / constructing stays GPU-free
    @property
    def store(self) -> EntityStore:
        if self._store is None:
            self._store = EntityStore(self._device)
        return self._store

    def add_target(self, target: Target) -> None:
        self._append(target, kind=0)

    def add_targets(self, targets) -> None:
        """Bulk form of add_target: one table append for the whole list (a scenario of 1e5 targets loads in time
        linear in their number).  List order = the order given, exactly as repeated add_target calls."""
        targets = list(targets)
        if not targets:
            return
        first = self.store.add_entities(
            [t.id for t in targets], np.asarray([t.trajectory.start_pos for t in targets], np.float64),
            np.asarray([t.trajectory.velocity for t in targets], np.float64),
            np.asarray([t.trajectory.start_time for t in targets], np.float64), kind=0,
            pos0=np.asarray([np.asarray(t.pos, np.float64) for t in targets], np.float64))
        for k, t in enumerate(targets):
            t._bind(self.store, first + k)
        self._handles.extend(targets)
        self._live_dirty = True

    def _append(self, obj: AirObject, kind: int) -> int:
        tr = obj.trajectory
        pos0 = np.asarray(obj.pos, dtype=np.float64)      # int-typed YAML positions are coerced
        slot = self.store.add_entities([obj.id], [tr.start_pos], [tr.velocity], tr.start_time, kind=kind, pos0=[pos0])
        obj._bind(self.store, slot)
        self._handles.append(obj)
        self._live_dirty = True
        return slot

    def step(self) -> None:
        mgr = self._manager
        now, dt = mgr.time.get_time(), mgr.time.get_dt()
        st = self.store
        st.flush()

        # 1. removals announced last tick
        doomed_ids = []
        for msg in mgr.give_messages_by_type(MessageType.MISSILE_DETONATE, step_time=now - dt):
            doomed_ids.append(msg.missile_id)
            if msg.target_id is not None:
                doomed_ids.append(msg.target_id)
        if doomed_ids:
            slots = [s for i in doomed_ids for s in st.slots_for_id(i) if st.h_alive[s]]
            for s in slots:
                self._handles[s]._freeze()
            st.kill(slots)
            self._live_dirty = True

        # 2. missiles that left their launcher last tick
        for msg in mgr.give_messages_by_type(MessageType.NEW_MISSILE, step_time=now - dt):
            m = msg.missile
            slot = self._append(m, kind=1)
            tgt = m.target
            if tgt is None or getattr(tgt, "_store", None) is not st:
                raise RuntimeError(f"missile {m.id} enters the air without a target in this AirEnv")
            m._row = st.add_missile_row(slot, tgt._slot, m.detonate_radius, m._period_at_launch())
            self._missile_rows.append(m)

        # 3. everything steps
        st.begin_tick(now)
        events = dict(st.missile_step(dt))            # missile slot -> target slot | -1
        st.sweep([], F_ADVANCE)
        for m in self._missile_rows:
            if m.status != "active" or not st.h_alive[m._slot]:
                continue
            mgr.add_message(MissilePosMessage(sender_id=m.id))                   # Missile.py:183-184
            if m._slot in events:
                ts = events[m._slot]
                m.status = "detonated"
                mgr.add_message(MissileDetonateMessage(                          # Missile.py:138-146
                    sender_id=m.id, target_id=None if ts < 0 else self._handles[ts].id, self_detonation=ts < 0))

        # 4. the live list
        if self._live_dirty:
            self._live_slots = np.nonzero(st.h_alive[:st.n])[0]
            self._live_dirty = False
        mgr.add_message(ActiveObjectsMessage(sender_id=self.id,
                                             active_objects=ActiveView(st, self._handles, self._live_slots)))
