"""``PreProcessStep`` config -> ``(callable, Stage)`` (myrtlespeech/builders/pre_process_step.py:13-61), built
onto the device front-end of ``myrtlespeech_amd.data.preprocess``."""
from typing import Callable, Tuple

from myrtlespeech_amd.data.preprocess import AddContextFrames, MFCC, MFCCLegacy, SpecAugment, Standardize
from myrtlespeech_amd.stage import Stage


def build(pre_process_step_cfg) -> Tuple[Callable, Stage]:
    """Raises ``ValueError`` when the ``pre_process_step`` oneof is unset or unknown."""
    kind = pre_process_step_cfg.WhichOneof("pre_process_step")
    if kind == "mfcc":
        cfg = pre_process_step_cfg.mfcc
        cls = MFCCLegacy if cfg.legacy else MFCC
        step: Callable = cls(n_mfcc=cfg.n_mfcc, melkwargs={"win_length": cfg.win_length, "hop_length": cfg.hop_length})
    elif kind == "spec_augment":
        cfg = pre_process_step_cfg.spec_augment
        step = SpecAugment(feature_mask=cfg.feature_mask, time_mask=cfg.time_mask,
                           n_feature_masks=cfg.n_feature_masks, n_time_masks=cfg.n_time_masks)
    elif kind == "standardize":
        step = Standardize()
    elif kind == "context_frames":
        step = AddContextFrames(n_context=pre_process_step_cfg.context_frames.n_context)
    else:
        raise ValueError(f"unknown pre_process_step '{kind}'")
    return step, Stage(pre_process_step_cfg.stage)
