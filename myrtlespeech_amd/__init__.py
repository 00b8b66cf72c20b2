"""MI355X-native implementation of the myrtlespeech acoustic-encoder / CTC hot path.

Mirrors the ``myrtlespeech.model`` / ``myrtlespeech.loss`` /
``myrtlespeech.post_process`` class surface (same constructor arguments, same
``(tensor, seq_lens)`` call convention, same ``state_dict`` keys) so it drops in
behind the reference's builders; the arithmetic runs in hand-written HIP kernels
for gfx950 behind the C ABI declared in ``include/ms_hotpath.h``.
"""
__all__ = ["model", "loss", "post_process"]
