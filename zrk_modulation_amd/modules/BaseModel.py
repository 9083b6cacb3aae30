"""Identity + position + step(): the module interface (reference modules/BaseModel.py:7-44)."""
from abc import ABCMeta, abstractmethod

import numpy as np


class BaseModel(metaclass=ABCMeta):
    @abstractmethod
    def __init__(self, manager, id: int, pos: np.ndarray) -> None:
        self._manager = manager
        self._model_id = id
        self._model_pos = pos

    @property
    def id(self) -> int:
        return self._model_id

    @property
    def pos(self) -> np.ndarray:
        return self._model_pos

    @pos.setter
    def pos(self, new_pos: np.ndarray) -> None:
        self._model_pos = new_pos

    @abstractmethod
    def step(self) -> None:
        """One simulation tick of this module."""
