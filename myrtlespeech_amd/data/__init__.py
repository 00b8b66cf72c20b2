"""Alphabet (symbol <-> index maps)."""
