"""Checkpoint compatibility.  The reference's ``Saver`` callback writes
``state_dict_{epoch}.pt`` = ``SeqToSeq.state_dict()`` (run/run.py:172-185: keys ``model.<module path>``), while
its export script loads a file of bare encoder keys straight into ``stt.model`` (scripts/export_ds1_onnx.py:49-50).
The accelerated modules keep the reference's parameter names and shapes, so both flavours load unchanged; the
device-side packed copies of the weights are rebuilt lazily on the next forward (they are keyed on each
parameter's version counter)."""
import os
from pathlib import Path
from typing import Dict, Union

import torch

_PREFIX = "model."


def encoder_state_dict(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Bare encoder keys from either flavour (``model.``-prefixed SeqToSeq dump or bare)."""
    if state_dict and all(k.startswith(_PREFIX) for k in state_dict):
        return {k[len(_PREFIX):]: v for k, v in state_dict.items()}
    return dict(state_dict)


def load(target: torch.nn.Module, path: Union[str, os.PathLike], strict: bool = True):
    """Load a reference checkpoint into a ``SeqToSeq`` / ``SpeechToText`` or directly into its encoder.
    Returns what ``load_state_dict`` returns; with ``strict`` any missing / unexpected key raises ``RuntimeError``."""
    state_dict = torch.load(str(path), map_location=torch.device("cpu"))
    encoder = target.model if _is_seq_to_seq(target) else target
    return encoder.load_state_dict(encoder_state_dict(state_dict), strict=strict)


def save(seq_to_seq: torch.nn.Module, log_dir: Union[str, os.PathLike], epoch: int) -> Path:
    """Write ``log_dir/state_dict_{epoch}.pt`` with the reference Saver's key layout."""
    path = Path(log_dir).joinpath(f"state_dict_{epoch}.pt")
    torch.save({k: v.detach().cpu() for k, v in seq_to_seq.state_dict().items()}, str(path))
    return path


def _is_seq_to_seq(module: torch.nn.Module) -> bool:
    from myrtlespeech_amd.model.seq_to_seq import SeqToSeq
    return isinstance(module, SeqToSeq)
