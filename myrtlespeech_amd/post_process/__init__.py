"""Decoders: CTC greedy / prefix beam search, RNN-T greedy / beam."""
