"""Mirror of myrtlespeech/builders/ctc_loss.py:5-44."""
from myrtlespeech_amd.loss.ctc_loss import CTCLoss

_REDUCTION = {0: "none", 1: "mean", 2: "sum"}


def build(ctc_loss_cfg) -> CTCLoss:
    if ctc_loss_cfg.reduction not in _REDUCTION:
        raise ValueError(f"reduction={ctc_loss_cfg.reduction} not supported")
    return CTCLoss(blank=ctc_loss_cfg.blank_index, reduction=_REDUCTION[ctc_loss_cfg.reduction])
