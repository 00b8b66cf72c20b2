"""Mirror of myrtlespeech/model/speech_to_text.py:10-33."""
from typing import Union

from myrtlespeech_amd.data.alphabet import Alphabet
from myrtlespeech_amd.model.seq_to_seq import SeqToSeq
from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder


class SpeechToText(SeqToSeq):
    """A :py:class:`SeqToSeq` for speech recognition: adds the ``alphabet`` and the decoder."""

    def __init__(self, alphabet: Alphabet, post_process: Union[None, CTCGreedyDecoder, CTCBeamDecoder], *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.alphabet = alphabet
        self.post_process = post_process

    def extra_repr(self) -> str:
        return f"(alphabet): {self.alphabet}"
