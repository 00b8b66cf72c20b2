"""Mirror of myrtlespeech/post_process/utils.py:4-60."""
from typing import Sequence


def levenshtein(a: Sequence, b: Sequence) -> int:
    """Minimum number of single-element insertions, deletions and substitutions
    that turn ``a`` into ``b`` (two-row dynamic programme)."""
    if len(a) > len(b):
        a, b = b, a
    row = list(range(len(a) + 1))
    for i, bi in enumerate(b, start=1):
        diag, row[0] = row[0], i
        for j, aj in enumerate(a, start=1):
            diag, row[j] = row[j], min(row[j] + 1, row[j - 1] + 1, diag + (aj != bi))
    return row[len(a)]
