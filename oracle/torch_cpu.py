"""CPU baseline on stock PyTorch -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by the product path).

The reference's CPU path IS PyTorch: masked ``F.pad`` + ``Conv2d``, ``pack_padded_sequence`` +
``torch.nn.LSTM/GRU/RNN``, ``Linear`` (model/cnn.py:445-483, model/rnn.py:133-185,
model/fully_connected.py:133-166, model/deep_speech_2.py:123-172).  The reference package cannot travel to
the GPU box, so this module re-assembles the same stock torch CPU operators in the reference's order from a
reference-keyed state dict; it is the ``cpu_baseline`` leg of bench.py and is pinned
against the reference's golden fixtures in tests/test_oracle_golden.py like the numpy oracle.
"""
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ds_oracle as O

_RNN_CLS = {O.LSTM: torch.nn.LSTM, O.GRU: torch.nn.GRU, O.BASIC_RNN: torch.nn.RNN}


def _out_lens(lens: torch.Tensor, kernel: int, stride: int, pad_total: int) -> torch.Tensor:
    # cnn.py:191-197: float32 arithmetic, then back to the integer dtype
    return (((lens.float() + pad_total - (kernel - 1) - 1) / stride) + 1).floor().to(lens.dtype)


@torch.no_grad()
def deep_speech_2_forward(x, lens, cfg: dict, sd: Dict[str, np.ndarray]):
    """Same contract as ``ds_oracle.deep_speech_2_forward`` (conv2d blocks, no lookahead): returns
    (logits [T, N, V] numpy, out_lens numpy)."""
    h = torch.as_tensor(np.asarray(x, dtype=np.float32)).clone()
    lens = torch.as_tensor(np.asarray(lens, dtype=np.int64))
    for c in cfg["convs"]:
        assert c["kind"] == "conv2d"
        w = torch.as_tensor(sd[f"cnn.{c['idx']}.weight"])
        b = sd.get(f"cnn.{c['idx']}.bias")
        b = None if b is None else torch.as_tensor(b)
        T = h.shape[-1]
        mask = torch.arange(T)[None, :] >= lens[:, None]                      # cnn.py:425-443
        h.masked_fill_(mask[:, None, None, :], 0.0)
        kf, kt = w.shape[2], w.shape[3]
        sf, st = c["stride"]
        pad_t = (0, 0)
        if c["same"]:
            pad_f = O.pad_same(h.shape[2], kf, sf)
            pad_t = O.pad_same(T, kt, st)
            h = F.pad(h, (pad_t[0], pad_t[1], pad_f[0], pad_f[1]))            # cnn.py:412
        h = F.conv2d(h, w, b, stride=(sf, st))
        lens = _out_lens(lens, kt, st, pad_t[0] + pad_t[1])
        if c["act"] is not None:
            h = F.hardtanh(h, c["act"][0], c["act"][1])
    n, ch, f, t = h.shape
    h = h.view(n, ch * f, t).permute(2, 0, 1).contiguous()                    # deep_speech_2.py:114-117
    r = cfg["rnn"]
    rnn = _RNN_CLS[r["kind"]](h.shape[2], r["hidden"], num_layers=r["layers"], bidirectional=r["bidirectional"])
    rnn.load_state_dict({k[len("rnn.rnn."):]: torch.as_tensor(v) for k, v in sd.items() if k.startswith("rnn.rnn.")})
    packed = torch.nn.utils.rnn.pack_padded_sequence(h, lens, enforce_sorted=True)   # rnn.py:170-183
    out, _ = rnn(packed)
    h, _ = torch.nn.utils.rnn.pad_packed_sequence(out, total_length=h.shape[0])
    assert cfg.get("lookahead") is None
    h = h.transpose(0, 1)                                                      # deep_speech_2.py:167
    fc = cfg["fc"]
    if fc["n_hidden"] == 0:
        h = F.linear(h, torch.as_tensor(sd["fully_connected.fully_connected.weight"]),
                     torch.as_tensor(sd["fully_connected.fully_connected.bias"]))
    else:
        keys = sorted({int(k.split(".")[2]) for k in sd if k.startswith("fully_connected.fully_connected.")
                       and k.endswith(".weight")})
        for i, key in enumerate(keys):
            h = F.linear(h, torch.as_tensor(sd[f"fully_connected.fully_connected.{key}.weight"]),
                         torch.as_tensor(sd[f"fully_connected.fully_connected.{key}.bias"]))
            if i + 1 < len(keys) and fc["act"] is not None:
                h = F.hardtanh(h, fc["act"][0], fc["act"][1])
    return h.transpose(0, 1).contiguous().numpy(), lens.numpy()


@torch.no_grad()
def deep_speech_1_forward(x, lens, sd: Dict[str, np.ndarray], n_hidden: int, relu_clip: float = 20.0):
    """model/deep_speech_1.py:138-188 in eval mode on stock torch CPU operators (torch.nn.LSTM flavour, batch_first):
    [N, C, F, T] -> view / permute [N, T, C*F] (:175-176) -> 3 x (Linear + Hardtanh(0, clip)) -> packed BiLSTM -> Linear +
    Hardtanh -> Linear; returns (logits [T, N, V] numpy, lens numpy)."""
    h = torch.as_tensor(np.asarray(x, dtype=np.float32))
    lens = torch.as_tensor(np.asarray(lens, dtype=np.int64))
    n, c, f, t = h.shape
    h = h.view(n, c * f, t).permute(0, 2, 1)
    for k in (1, 2, 3):
        h = F.hardtanh(F.linear(h, torch.as_tensor(sd[f"fc{k}.0.weight"]), torch.as_tensor(sd[f"fc{k}.0.bias"])), 0.0, relu_clip)
    rnn = torch.nn.LSTM(2 * n_hidden, n_hidden, num_layers=1, bidirectional=True, batch_first=True)   # deep_speech_1.py:101-108
    rnn.load_state_dict({k[len("bi_lstm.rnn."):]: torch.as_tensor(v) for k, v in sd.items() if k.startswith("bi_lstm.rnn.")})
    packed = torch.nn.utils.rnn.pack_padded_sequence(h, lens, batch_first=True, enforce_sorted=True)     # rnn.py:170-183
    out, _ = rnn(packed)
    h, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=t)
    h = F.hardtanh(F.linear(h, torch.as_tensor(sd["fc4.0.weight"]), torch.as_tensor(sd["fc4.0.bias"])), 0.0, relu_clip)
    h = F.linear(h, torch.as_tensor(sd["out.weight"]), torch.as_tensor(sd["out.bias"]))
    return h.transpose(0, 1).contiguous().numpy(), lens.numpy()


@torch.no_grad()
def ctc_greedy_decode(x: np.ndarray, lens: np.ndarray, blank: int):
    """ctc_greedy_decoder.py:74-92 with one argmax over the batch and a host loop."""
    best = torch.as_tensor(x).argmax(dim=2).numpy()
    out = []
    for n in range(best.shape[1]):
        seq, prev = [], None
        for t in range(int(lens[n])):
            sym = int(best[t, n])
            if sym != blank and (prev is None or prev == blank or sym != prev):
                seq.append(sym)
            prev = sym
        out.append(seq)
    return out
