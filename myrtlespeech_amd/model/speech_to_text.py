"""``SpeechToText`` under the reference's module path (myrtlespeech/model/speech_to_text.py);
the class itself is defined next to ``SeqToSeq``."""
from myrtlespeech_amd.model.seq_to_seq import SpeechToText  # noqa: F401

__all__ = ["SpeechToText"]
