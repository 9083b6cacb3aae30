"""Bounded, columnar retention of what the GUI replays.

The reference keeps every message object of every tick in `Manager.messages` (modules/Manager.py:13,
:41-58) and the GUI's replay walks it afterwards, reading only the DRAW_OBJECTS messages of each step:
obj_id, coordinates[:2], is_visible_by_radar and, for colour, target_type (UI/PolygonEditor.py:612-637;
message: modules/Messages.py:229-243).  A run of 1e4 ticks x 1e5 objects would hold 1e9 Python objects.

ReplayLog keeps, for the last `max_steps` steps, one frame per step as four arrays -- ids (int64),
type codes (int32 into a shared name table), positions (float64 [k, 3]), visibility (bool) -- in the order
the messages were sent, and hands them back either as arrays (`frame`) or as the message objects the GUI
expects (`messages`).  `Manager(replay=ReplayLog(...))` routes DRAW_OBJECTS messages here instead of into the
per-tick lists; everything else about the bus is unchanged.  `record_store` fills a frame straight from a
device table (headless runs have no command post to send the messages).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


class _Frame:
    __slots__ = ("ids", "types", "pos", "vis", "n", "sender", "receiver")

    def __init__(self, sender=None, receiver=None):
        self.ids = np.empty(16, np.int64)
        self.types = np.empty(16, np.int32)
        self.pos = np.empty((16, 3), np.float64)
        self.vis = np.empty(16, np.bool_)
        self.n = 0
        self.sender, self.receiver = sender, receiver

    def _grow(self, need):
        cap = len(self.ids)
        if need <= cap:
            return
        cap = max(need, 2 * cap)
        for name in ("ids", "types", "vis"):
            old = getattr(self, name)
            new = np.empty(cap, old.dtype)
            new[:self.n] = old[:self.n]
            setattr(self, name, new)
        new = np.empty((cap, 3), np.float64)
        new[:self.n] = self.pos[:self.n]
        self.pos = new

    def trim(self):
        k = self.n
        return self.ids[:k], self.types[:k], self.pos[:k], self.vis[:k]


class ReplayLog:
    def __init__(self, max_steps=None):
        """max_steps: how many of the most recent steps to keep (None: all of them -- still columnar)."""
        self.max_steps = max_steps
        self.type_names = []                 # code -> what the GUI sees as target_type (enum member or str)
        self._code = {}
        self._frames = OrderedDict()         # step time -> _Frame
        self.dropped_steps = 0

    # writing ----------------------------------------------------------------------------------------
    def _frame(self, step_time, sender=None, receiver=None):
        fr = self._frames.get(step_time)
        if fr is None:
            fr = self._frames[step_time] = _Frame(sender, receiver)
            while self.max_steps is not None and len(self._frames) > self.max_steps:
                self._frames.popitem(last=False)
                self.dropped_steps += 1
        return fr

    def _type_code(self, target_type):
        key = getattr(target_type, "name", None) or str(target_type)
        code = self._code.get(key)
        if code is None:
            code = self._code[key] = len(self.type_names)
            self.type_names.append(target_type)
        return code

    def add_message(self, step_time, msg):
        """One CPPDrawerObjectsMessage (reference field names)."""
        fr = self._frame(step_time, msg.sender_id, msg.receiver_id)
        fr._grow(fr.n + 1)
        k = fr.n
        fr.ids[k] = msg.obj_id
        fr.types[k] = self._type_code(msg.target_type)
        fr.pos[k] = np.asarray(msg.coordinates, np.float64)      # by value: the object's array moves on
        fr.vis[k] = bool(msg.is_visible_by_radar)
        fr.n = k + 1

    def add_arrays(self, step_time, ids, type_codes, pos, visible, sender=None, receiver=None):
        fr = self._frame(step_time, sender, receiver)
        k, m = fr.n, len(ids)
        fr._grow(k + m)
        fr.ids[k:k + m] = ids
        fr.types[k:k + m] = type_codes
        fr.pos[k:k + m] = pos
        fr.vis[k:k + m] = visible
        fr.n = k + m

    def record_store(self, step_time, store, type_codes=None, sender=None, receiver=None):
        """A frame from a device table after a fused multi-radar tick (headless runs have no command post to send
        the messages): live objects in list order, their current positions, seen = any radar bit set.
        type_codes: per ROW (default: the table's kind column, 0 target / 1 missile)."""
        n = store.n_uploaded
        alive = store.d_alive[:n].cpu().numpy().astype(bool)
        pos = store.host_pos("cur")
        lidx = store.h_lidx[:n] if store.h_lidx is not None else np.arange(n)    # list position of each row
        seen_by_list = store.vis()[:n].cpu().numpy().view(np.uint32) != 0        # masks are indexed by list position
        rows = np.argsort(lidx, kind="stable")                                    # rows in list order ...
        rows = rows[alive[rows]]                                                  # ... that are alive
        codes = store.h_kind[:n] if type_codes is None else np.asarray(type_codes)
        self.add_arrays(step_time, store.h_ids[:n][rows], codes[rows], pos[rows], seen_by_list[lidx[rows]], sender, receiver)

    # reading ----------------------------------------------------------------------------------------
    def steps(self):
        return list(self._frames)

    def frame(self, step_time):
        """(ids, type_codes, pos, visible) of a step, message order; empty arrays if the step sent nothing."""
        fr = self._frames.get(step_time)
        if fr is None:
            return np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros((0, 3)), np.zeros(0, bool)
        return fr.trim()

    def messages(self, step_time):
        """The step's DRAW_OBJECTS messages as the GUI reads them (modules/Messages.py:229-243)."""
        from .modules.Messages import CPPDrawerObjectsMessage
        fr = self._frames.get(step_time)
        if fr is None:
            return []
        ids, types, pos, vis = fr.trim()
        return [CPPDrawerObjectsMessage(sender_id=fr.sender, obj_id=int(ids[k]), target_type=self.type_names[types[k]],
                                        coordinates=pos[k].copy(), is_visible_by_radar=bool(vis[k]), time=step_time,
                                        receiver_id=fr.receiver) for k in range(len(ids))]

    def nbytes(self):
        return sum(fr.ids.nbytes + fr.types.nbytes + fr.pos.nbytes + fr.vis.nbytes for fr in self._frames.values())
