"""Mirror of myrtlespeech/model/seq_to_seq.py:10-61."""
from typing import Callable, Optional, Sequence, Tuple

import torch

from myrtlespeech_amd.stage import Stage


class SeqToSeq(torch.nn.Module):
    """Container of a sequence-to-sequence model: ``model``, ``loss``, stage-tagged
    ``pre_process_steps`` and an optional optimiser."""

    def __init__(self, model: torch.nn.Module, loss: torch.nn.Module,
                 pre_process_steps: Sequence[Tuple[Callable, Stage]], optim: Optional[torch.optim.Optimizer] = None):
        super().__init__()
        self.model = model
        self.loss = loss
        self.pre_process_steps = pre_process_steps
        self.optim = optim
        self.use_cuda = torch.cuda.is_available()
        if self.use_cuda:
            self.model = self.model.cuda()

    @property
    def pre_process(self) -> Callable:
        """Applies every step whose stage matches ``self.training``."""

        def process(x):
            for step, stage in self.pre_process_steps:
                if (stage is Stage.TRAIN and not self.training) or (stage is Stage.EVAL and self.training):
                    continue
                x = step(x)
            return x

        return process
