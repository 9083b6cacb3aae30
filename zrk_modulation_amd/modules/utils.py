"""Target and its type tag (reference modules/utils.py:6-59)."""
from enum import Enum

import numpy as np

from .AirObject import AirObject, to_seconds  # noqa: F401  (re-exported like the reference does)


class TargetType(Enum):
    AIR_PLANE = "самолет"
    HELICOPTER = "вертолет"
    ANOTHER = "другое"


class Target(AirObject):
    def __init__(self, manager, id: int, pos: np.ndarray = None, trajectory=None,
                 type: TargetType = TargetType.ANOTHER):
        super().__init__(manager, id, pos, trajectory)
        self._target_type = type

    @property
    def type(self) -> TargetType:
        return self._target_type

    def __repr__(self) -> str:
        fmt = lambda v: "[" + ", ".join(f"{c:.2f}" for c in v) + "]"      # noqa: E731
        vel = fmt(self.trajectory.velocity) if getattr(self, "trajectory", None) is not None else "[unknown]"
        return (f"Target(id={self.id}, type={self._target_type.name}, pos={fmt(self.pos)}, vel={vel}, "
                f"prev_pos={self.prev_pos})")
