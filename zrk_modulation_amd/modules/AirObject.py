"""Trajectory and the air-object handle (reference modules/AirObject.py:5-42).

An `AirObject` is the Python identity of one row of the device-resident entity table: the command
post and the launchers keep references to these objects across ticks, exactly as they do in the
reference, but `pos` / `prev_pos` read through to the table (EntityStore) once the object has been
added to an `AirEnv`.  The per-tick arithmetic of `AirObject.step` (reference :39-42) is not done
here: AirEnv.step() advances every row in one kernel launch.
"""
import numpy as np

from .BaseModel import BaseModel


def to_seconds(time: int) -> float:
    return time / 1000


class Trajectory:
    """S(t) = start_pos + velocity * (t - start_time)   (reference modules/AirObject.py:9-25)."""

    def __init__(self, velocity=(0.0, 0.0, 0.0), start_pos=(0.0, 0.0, 0.0), start_time: float = 0.0):
        self.velocity = np.array(velocity, dtype=np.float64)
        self.start_pos = np.array(start_pos, dtype=np.float64)
        self.start_time = start_time

    def get_pos(self, t: float) -> np.ndarray:
        """Host evaluation with the reference's three roundings per axis (reference :23-25).  AirEnv.step() does
        the same arithmetic for every live row on the device; this is for callers that ask about one object."""
        return self.start_pos + self.velocity * (t - self.start_time)


class AirObject(BaseModel):
    def __init__(self, manager, id: int, pos: np.ndarray, trajectory: Trajectory, prev_pos: np.ndarray = None):
        super().__init__(manager, id, pos)
        self.trajectory = trajectory
        with np.errstate(divide="ignore", invalid="ignore"):
            norm = np.linalg.norm(trajectory.velocity)
            self.velocity = trajectory.velocity / norm          # unit vector (NaN for a zero velocity)
        self.speed_mod = norm
        self._initial_prev = prev_pos
        self._store = None          # EntityStore once bound
        self._slot = -1
        self._frozen = None         # (pos, prev_pos) snapshot taken when the object is removed

    # binding ---------------------------------------------------------------------------------
    def _bind(self, store, slot):
        self._store, self._slot = store, int(slot)

    def _freeze(self):
        """Called when AirEnv tombstones the object: later readers keep seeing its last state."""
        if self._store is not None and self._frozen is None:
            self._frozen = (self.pos, self.prev_pos)

    # state -----------------------------------------------------------------------------------
    @property
    def pos(self) -> np.ndarray:
        if self._frozen is not None:
            return self._frozen[0]
        if self._store is None:
            return self._model_pos
        return self._store.host_pos("cur")[self._slot].copy()

    @pos.setter
    def pos(self, new_pos) -> None:
        if self._store is None or self._frozen is not None:
            self._model_pos = new_pos
            if self._frozen is not None:
                self._frozen = (new_pos, self._frozen[1])
        else:
            self._store.write_pos(self._slot, new_pos)

    @property
    def prev_pos(self):
        if self._frozen is not None:
            return self._frozen[1]
        st = self._store
        if st is None or self._slot >= st.n_stepped:
            return self._initial_prev                       # never stepped yet
        if self.trajectory.start_time == st.time_ms / 1000:  # reference AirObject.py:41
            return None
        return st.host_pos("prev")[self._slot].copy()

    @prev_pos.setter
    def prev_pos(self, value):
        self._initial_prev = value

    def step(self):
        """One object stepped on its own (reference :39-42).  Inside an AirEnv every live row is advanced by one kernel
        launch and nobody calls this; an object that is not in any AirEnv (or a caller that insists) gets the
        reference's two assignments, written through to the table when the object has a row."""
        t = to_seconds(self._manager.time.get_time())
        new_prev = self.pos if self.trajectory.start_time != t else None
        new_pos = self.trajectory.get_pos(t)
        if self._store is None or self._frozen is not None:
            self._initial_prev = new_prev
            self.pos = new_pos
        else:
            st = self._store
            if new_prev is not None:
                st.write_pos(self._slot, new_prev, which="prev")
            st.write_pos(self._slot, new_pos)
