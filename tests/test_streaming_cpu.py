"""CPU checks of the carried-context stream arithmetic (``myrtlespeech_amd/streaming.py::_ConvStage``): which frames a
convolution holds back, where the SAME padding goes and how the stride phase is kept across chunks.  The convolution kernel
itself is replaced by stock torch CPU operators here (the HIP kernel is exercised on the GPU, tests/test_gpu_parity.py::
test_streaming_with_carried_context_*); what is checked is that a chain of stream operators fed chunk by chunk produces the
frames the reference's full-length masked SAME convolutions produce (the numpy oracle, pinned to the reference's fixtures)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ds_oracle as O


def _torch_conv_forward(x4, seq_lens, weight4, bias, packed, stride, dilation, groups, same, act, time_pads=None):
    """Stand-in for model/cnn.py::_conv_forward on the CPU: mask (cnn.py:425-443), SAME feature padding, explicit time pads."""
    from myrtlespeech_amd.model.cnn import pad_same
    assert time_pads == (0, 0)
    x = x4.clone()
    t = x.shape[-1]
    mask = torch.arange(t)[None, :] >= seq_lens.to(torch.int64)[:, None]
    x.masked_fill_(mask[:, None, None, :], 0.0)
    kf = weight4.shape[2]
    pf = pad_same(x.shape[2], kf, stride[0], dilation[0]) if same else (0, 0)
    x = F.pad(x, (0, 0, pf[0], pf[1]))
    y = F.conv2d(x, weight4, bias, stride=stride, dilation=dilation, groups=groups)
    if act is not None:
        y = y.clamp(act[0], act[1])
    return y, None


@pytest.mark.parametrize("seed", range(60))
def test_conv_stream_operators_reproduce_full_length_same_convolutions(monkeypatch, seed):
    from myrtlespeech_amd import streaming
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    monkeypatch.setattr(streaming, "_conv_forward", _torch_conv_forward)
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    n, feats = int(rng.integers(1, 5)), 12
    total = int(rng.integers(9, 80))
    lens = np.sort(rng.integers(1, total + 1, size=n))[::-1].copy()
    lens[0] = total
    geo = [(int(rng.integers(1, 7)), int(rng.integers(1, 4)), int(rng.integers(1, 3))) for _ in range(int(rng.integers(1, 4)))]
    with torch.no_grad():
        convs, cin = [], 1
        for kt, st, dt in geo:
            c = MaskConv2d(cin, 3, [3, kt], [1, st], PaddingMode.SAME, dilation=[1, dt])
            convs.append(c)
            cin = 3
    x = torch.randn(n, 1, feats, total)
    # full length: the oracle's masked SAME convolutions, layer by layer (lens travel like cnn.py:191-197)
    h, hl = x.numpy().copy(), lens.copy()
    for c, (kt, st, dt) in zip(convs, geo):
        h, hl = O.mask_conv2d(h, hl, c.weight.detach().numpy(), c.bias.detach().numpy(), stride=(1, st), padding_same=True,
                              dilation=(1, dt))
        h = np.clip(h, 0.0, 20.0)
    # streamed, with a chunk size drawn per call
    stages, tot, cl = [], total, torch.as_tensor(lens.copy())
    for c in convs:
        s = streaming._ConvStage(c, (0.0, 20.0), tot, cl)
        stages.append(s)
        tot, cl = s.total_out, s.lens_out
    assert tot == h.shape[-1] and cl.tolist() == [int(v) for v in hl]
    outs, t0 = [], 0
    while t0 < total:
        t1 = min(total, t0 + int(rng.integers(1, 12)))
        cur = x[..., t0:t1]
        with torch.no_grad():
            for s in stages:
                cur = s.push(cur, final=(t1 == total))
        if cur is not None:
            outs.append(cur)
        t0 = t1
    got = torch.cat(outs, -1).numpy()
    assert got.shape == h.shape
    # frames an utterance owns must agree; frames past its output length hold bias / edge values in the reference too, and
    # agree as well because both sides zero exactly the same input frames
    np.testing.assert_allclose(got, h, rtol=1e-4, atol=1e-4)


def test_latency_of_the_shipped_stack():
    """conv 11 / stride 2, conv 11 / stride 1 (SAME), lookahead 80: 174 input frames before logit row 0."""
    from myrtlespeech_amd import streaming
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    c1 = streaming._ConvStage(MaskConv2d(1, 2, [3, 11], [2, 2], PaddingMode.SAME), None, 1001, torch.tensor([1001]))
    c2 = streaming._ConvStage(MaskConv2d(2, 2, [3, 11], [2, 1], PaddingMode.SAME), None, c1.total_out, c1.lens_out)
    need = 80
    for st in (c2, c1):
        need = (need - 1) * st.stride + st.span - st.left
    assert (c1.left, c1.right, c2.left, c2.right) == (5, 6, 5, 5) and need == 174
