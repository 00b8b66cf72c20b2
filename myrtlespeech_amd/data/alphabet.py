"""Mirror of myrtlespeech/data/alphabet.py:5-78: symbol <-> index maps."""
from typing import Dict, List, Optional


class Alphabet:
    """An ordered set of unique symbols; a symbol's index is its position."""

    def __init__(self, symbols: List[str]):
        if len(set(symbols)) != len(symbols):
            raise ValueError("Duplicate symbol in symbols.")
        self.symbols = symbols
        self._to_symbol: Dict[int, str] = dict(enumerate(symbols))
        self._to_index: Dict[str, int] = {s: i for i, s in enumerate(symbols)}

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(symbols={self.symbols})"

    def __len__(self) -> int:
        return len(self.symbols)

    def __getitem__(self, index: int) -> str:
        symbol = self.get_symbol(index)
        if symbol is None:
            raise IndexError(f"Index {index} is out of range")
        return symbol

    def get_symbol(self, index: int) -> Optional[str]:
        return self._to_symbol.get(index)

    def get_index(self, symbol: str) -> Optional[int]:
        return self._to_index.get(symbol)

    def get_symbols(self, indices: List[int]) -> List[str]:
        """Indices without a symbol are skipped (the result may be shorter)."""
        return [self._to_symbol[i] for i in indices if i in self._to_symbol]

    def get_indices(self, sentence: List[str]) -> List[int]:
        """Symbols outside the alphabet are skipped."""
        return [self._to_index[s] for s in sentence if s in self._to_index]
