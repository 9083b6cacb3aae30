"""Integer-millisecond simulation clock (reference modules/Timer.py:5-30)."""
from .constants import SIMULATION_STEP


class Timer:
    def __init__(self):
        self._now = 0
        self._step = SIMULATION_STEP

    def get_time(self) -> int:
        return self._now

    def set_time(self, time: int) -> None:
        self._now = time

    def get_dt(self):
        return self._step

    def set_dt(self, dt: int) -> None:
        self._step = dt

    def update_time(self) -> None:
        """One tick forward."""
        self._now += self._step
