"""Decode -> transcripts -> word error rate: the arithmetic of the reference's
``ReportCTCDecoder`` callback (run/run.py:29-109) without the callback plumbing."""
from typing import Iterable, List, Sequence, Tuple

from myrtlespeech_amd.post_process.utils import levenshtein


class WordSegmentor:
    """Groups a sequence of symbols into words at ``separator`` (run/run.py:29-48);
    empty words are dropped."""

    def __init__(self, separator: str):
        self.separator = separator

    def __call__(self, sentence: List[str]) -> List[str]:
        words, word = [], []
        for symbol in sentence:
            if symbol == self.separator:
                if word:
                    words.append("".join(word))
                    word = []
            else:
                word.append(symbol)
        if word:
            words.append("".join(word))
        return words


class WordErrorRate:
    """Accumulates (hypothesis, reference) index sequences over batches and reports
    ``100 * sum(edit distances) / sum(reference lengths)`` in words (run/run.py:84-109)."""

    def __init__(self, alphabet, word_segmentor: WordSegmentor):
        self.alphabet = alphabet
        self.word_segmentor = word_segmentor
        self.transcripts: List[Tuple[List[str], List[str]]] = []
        self.distances: List[int] = []
        self.lengths: List[int] = []

    def _words(self, indices: Iterable[int]) -> List[str]:
        return self.word_segmentor(self.alphabet.get_symbols(list(indices)))

    def update(self, hypotheses: Sequence[Sequence[int]], targets, target_lens) -> None:
        for hyp, target, n in zip(hypotheses, targets, target_lens):
            act = self._words(hyp)
            exp = self._words(int(e) for e in target[:int(n)])
            self.transcripts.append((act, exp))
            self.distances.append(levenshtein(act, exp))
            self.lengths.append(len(exp))

    def value(self) -> float:
        return float(sum(self.distances)) / sum(self.lengths) * 100
