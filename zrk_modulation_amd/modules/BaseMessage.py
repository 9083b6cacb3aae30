"""Common header of every bus message (reference modules/BaseMessage.py:4-22)."""
from .constants import MessageType


class BaseMessage:
    def __init__(self, type: MessageType, sender_id: int, receiver_id: int = None, send_time: int = None,
                 relevance: int = 1) -> None:
        self.type = type
        self.sender_id = sender_id
        self.receiver_id = receiver_id
        self.send_time = send_time          # filled in by Manager.add_message when None
        self.relevance = relevance          # higher sorts first inside a tick

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}: send_time={self.send_time}, type={self.type}, "
                f"sender_id={self.sender_id}, receiver_id={self.receiver_id}, relevance={self.relevance}")
