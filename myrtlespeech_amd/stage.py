"""Mirror of myrtlespeech/run/stage.py:1-9 (values of protos/stage.proto)."""
from enum import IntEnum


class Stage(IntEnum):
    TRAIN = 0
    EVAL = 1
    TRAIN_AND_EVAL = 2
