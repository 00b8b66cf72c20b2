"""Mirror of myrtlespeech/builders/language_model.py:8-38 (only ``no_lm`` exists upstream)."""
from typing import Callable, Optional, Tuple


def build(lm_cfg) -> Optional[Callable[[Tuple[int, ...]], float]]:
    supported_lm = lm_cfg.WhichOneof("supported_lms")
    if supported_lm == "no_lm":
        return None
    raise ValueError(f"{supported_lm} not supported")
