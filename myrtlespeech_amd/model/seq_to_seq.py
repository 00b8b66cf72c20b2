"""Model containers of the reference's training/eval API (model/seq_to_seq.py:10-61 and
model/speech_to_text.py:10-33): they own no arithmetic, they only bundle the encoder, the loss,
the decoder, the alphabet and the stage-tagged pre-processing steps.  Both classes live here;
``model/speech_to_text.py`` re-exports ``SpeechToText`` under the reference's module path."""
from typing import Any, Callable, Optional, Sequence, Tuple

import torch

from myrtlespeech_amd.stage import Stage


class _StagePipeline:
    """Callable that runs, in order, the steps whose stage tag matches the owner's mode
    (TRAIN steps only while training, EVAL steps only while evaluating, TRAIN_AND_EVAL always)."""

    def __init__(self, owner: torch.nn.Module):
        self._owner = owner

    def active_steps(self):
        training = self._owner.training
        for step, stage in self._owner.pre_process_steps:
            skip = (stage is Stage.EVAL) if training else (stage is Stage.TRAIN)
            if not skip:
                yield step

    def __call__(self, x: Any) -> Any:
        for step in self.active_steps():
            x = step(x)
        return x

    def batch(self, x: torch.Tensor, lens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """The same steps over a zero-padded ragged batch: ``(waves [N, samples], sample counts)`` in,
        ``(features [N, C, F, T], frame counts)`` out -- one launch set per step instead of one per utterance."""
        for step in self.active_steps():
            if hasattr(step, "batch"):
                x, lens = step.batch(x, lens)
            else:
                x = step(x)
        return x, lens


class SeqToSeq(torch.nn.Module):
    """``model`` + ``loss`` + ``pre_process_steps`` (+ optional optimiser); ``pre_process`` is the
    stage-filtered composition of the steps."""

    def __init__(self, model: torch.nn.Module, loss: torch.nn.Module,
                 pre_process_steps: Sequence[Tuple[Callable, Stage]], optim: Optional[torch.optim.Optimizer] = None):
        super().__init__()
        self.model = model.cuda() if torch.cuda.is_available() else model
        self.loss = loss
        self.pre_process_steps = pre_process_steps
        self.optim = optim
        self.use_cuda = torch.cuda.is_available()

    @property
    def pre_process(self) -> Callable:
        return _StagePipeline(self)

    def pre_process_batch(self, x: torch.Tensor, lens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return _StagePipeline(self).batch(x, lens)


class SpeechToText(SeqToSeq):
    """Speech recognition flavour: additionally holds the ``alphabet`` (symbols <-> indices) and the
    ``post_process`` decoder (greedy, beam or None)."""

    def __init__(self, alphabet, post_process, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.alphabet = alphabet
        self.post_process = post_process

    def extra_repr(self) -> str:
        return f"(alphabet): {self.alphabet}"

    @torch.no_grad()
    def transcribe(self, waves: torch.Tensor, sample_counts: torch.Tensor):
        """Waveforms to text, entirely on the device until the label lists come back: ``waves [N, samples]``
        zero-padded and sorted longest first, ``sample_counts [N]`` -> ``(sentences, label lists)``.  Runs the
        active pre-processing steps batched (``pre_process_batch``), the encoder, the decoder and the alphabet."""
        if self.post_process is None:
            raise ValueError("this SpeechToText has no post_process decoder")
        x, lens = self.pre_process_batch(waves, sample_counts)
        (logits, out_lens), _ = self.model((x, lens))
        labels = self.post_process(logits, out_lens)
        return ["".join(self.alphabet.get_symbols(seq)) for seq in labels], labels
