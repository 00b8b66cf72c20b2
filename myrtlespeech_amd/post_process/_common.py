"""Argument checks shared by the CTC decoders (ctc_greedy_decoder.py:49-72 ==
ctc_beam_decoder.py:148-171)."""
import torch

SUPPORTED_LENGTH_DTYPES = [torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64]


def check_decoder_args(x: torch.Tensor, lengths: torch.Tensor):
    if lengths.dtype not in SUPPORTED_LENGTH_DTYPES:
        raise ValueError(f"lengths.dtype={lengths.dtype} must be in {SUPPORTED_LENGTH_DTYPES}")
    seq_len, x_batch, symbols = x.size()
    l_batch = len(lengths)
    if x_batch != l_batch:
        raise ValueError(f"batch size of x ({x_batch}) and lengths {l_batch} must be equal")
    from myrtlespeech_amd import _lib
    if not bool((_lib.host_lens(lengths) <= seq_len).all()):   # host values: no read-back when a module attached them
        raise ValueError("length values must be less than or equal to x seq_len")
    return seq_len, x_batch, symbols


def ragged_to_lists(idx: torch.Tensor, lens: torch.Tensor):
    """[N, T] int32 + [N] int32 (device) -> List[List[int]] with ONE read-back (lengths ride in column 0) and numpy
    slicing on the host (a torch slice + tolist per utterance costs ~3 us each)."""
    packed = torch.cat([lens.to(idx.dtype).reshape(-1, 1), idx], dim=1).cpu().numpy()
    return [packed[n, 1:1 + int(packed[n, 0])].tolist() for n in range(packed.shape[0])]
