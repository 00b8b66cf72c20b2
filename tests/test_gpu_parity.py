"""Parity of the HIP path (through the C ABI) against the golden vectors generated
from the reference and against the CPU oracle.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest
import torch

from oracle import ds_oracle as O
from util import Golden, golden_names, unragged

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-4, atol=1e-4)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_sd(module, sd):
    module.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    return module.eval()


def cpu(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- linear
@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (7, 5, 3), (33, 17, 29), (130, 64, 129), (257, 100, 64), (300, 640, 200),
                                   (64, 36, 32)])
@pytest.mark.parametrize("act", [None, (0.0, 20.0)])
def test_linear(lib, M, K, N, act):
    from myrtlespeech_amd import _lib
    rng = np.random.default_rng(M * 1000 + K * 10 + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = rng.normal(size=(N, K)).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32)
    xd, wd, bd = T(x).cuda(), T(w).cuda(), T(b).cuda()
    y = torch.empty((M, N), dtype=torch.float32, device="cuda")
    a, lo, hi = (0, 0.0, 0.0) if act is None else (1, act[0], act[1])
    _lib.check(lib.ms_linear_forward(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), M, K, N, a, lo, hi,
                                     _lib.stream_ptr()), "linear")
    want = x.astype(np.float64) @ w.T.astype(np.float64) + b
    if act is not None:
        want = np.clip(want, *act)
    np.testing.assert_allclose(cpu(y), want, rtol=1e-4, atol=1e-4)


# ----------------------------------------------------------------------------- rnn
@pytest.mark.parametrize("name", golden_names("rnn_"))
def test_rnn_golden(name):
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    g = Golden(name)
    c = g.cfg
    m = RNN(RNNType(c["rnn_type"]), c["input_size"], c["hidden_size"], num_layers=c["num_layers"],
            bidirectional=c["bidirectional"], forget_gate_bias=c["forget_gate_bias"], batch_first=c["batch_first"])
    load_sd(m, g.sd())
    hx = None
    if g.has("in/h0"):
        hx = (T(g["in/h0"]), T(g["in/c0"])) if c["rnn_type"] == 0 else T(g["in/h0"])
    (out, lens), hid = m((T(g["in/x"]), T(g["in/lens"])), hx)
    np.testing.assert_allclose(cpu(out), g["out/y"], **TOL)
    np.testing.assert_array_equal(cpu(lens), g["out/lens"])
    if c["rnn_type"] == 0:
        np.testing.assert_allclose(cpu(hid[0]), g["out/hn"], **TOL)
        np.testing.assert_allclose(cpu(hid[1]), g["out/cn"], **TOL)
    else:
        np.testing.assert_allclose(cpu(hid), g["out/hn"], **TOL)


def test_rnn_rejects_unsorted():
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    m = RNN(RNNType.LSTM, 4, 8)
    with pytest.raises(RuntimeError):
        m((torch.randn(5, 2, 4), torch.tensor([3, 5])))


@pytest.mark.parametrize("H,N,T_,bidir", [(256, 32, 9, True), (512, 7, 5, True), (1024, 32, 6, True), (256, 64, 4, False),
                                          (96, 40, 7, True)])
def test_lstm_persistent_vs_oracle(H, N, T_, bidir):
    """Shapes that take the persistent kernel (incl. its pipelined H%256==0 loop, two batch
    tiles and the config-2 width) against the numpy oracle, ragged lengths, random hx."""
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(H + N)
    In = 48
    m = RNN(RNNType.LSTM, In, H, num_layers=2, bidirectional=bidir, forget_gate_bias=1.0).eval()
    rng = np.random.default_rng(H * 7 + N)
    lens = np.sort(rng.integers(1, T_ + 1, size=N))[::-1].copy()
    lens[0] = T_
    x = rng.normal(size=(T_, N, In)).astype(np.float32)
    D = 2 if bidir else 1
    h0 = (rng.normal(size=(2 * D, N, H)) * 0.3).astype(np.float32)
    c0 = (rng.normal(size=(2 * D, N, H)) * 0.3).astype(np.float32)
    (out, _), (hn, cn) = m((T(x), T(lens)), (T(h0), T(c0)))
    sd = {k[len("rnn."):]: cpu(v) for k, v in m.state_dict().items()}
    want, (whn, wcn) = O.rnn_forward(O.LSTM, x, lens, sd, H, 2, bidir, (h0, c0))
    np.testing.assert_allclose(cpu(out), want, **TOL)
    np.testing.assert_allclose(cpu(hn), whn, **TOL)
    np.testing.assert_allclose(cpu(cn), wcn, **TOL)


@pytest.mark.parametrize("kind,H,N,T_,bidir,In", [("GRU", 64, 32, 7, False, 32), ("GRU", 128, 17, 5, True, 48),
                                                  ("GRU", 256, 70, 4, False, 64), ("BASIC_RNN", 64, 5, 6, True, 20),
                                                  ("BASIC_RNN", 192, 33, 3, False, 32), ("LSTM", 1280, 9, 3, False, 32),
                                                  ("GRU", 2560, 32, 3, False, 96), ("GRU", 1280, 40, 5, True, 64),
                                                  ("GRU", 2560, 7, 4, False, 32),
                                                  # round 4: the persistent GRU at the other multiples of 128 (8 units per workgroup)
                                                  ("GRU", 1024, 32, 6, True, 64), ("GRU", 1024, 45, 4, False, 32),
                                                  ("GRU", 512, 32, 7, True, 48), ("GRU", 768, 20, 5, True, 64),
                                                  ("GRU", 1536, 33, 4, False, 32), ("GRU", 2048, 32, 3, False, 64),
                                                  ("GRU", 2048, 9, 3, True, 32),
                                                  # round 6: bidirectional layers whose two directions do not fit the CUs together
                                                  # (persistent, one launch per direction; they used to take a launch per step)
                                                  ("GRU", 1536, 40, 4, True, 32), ("GRU", 2560, 33, 3, True, 64),
                                                  # ... and tanh-RNN layers on the persistent GRU kernel (reset gate held at 1,
                                                  # update gate at 0: model/rnn.py tanh_rnn_as_gru), also padded widths and 2 layers
                                                  ("BASIC_RNN", 1024, 32, 6, True, 64), ("BASIC_RNN", 700, 40, 4, False, 48),
                                                  ("BASIC_RNN", 256, 20, 9, True, 32),
                                                  # ... and the two-stream LSTM beyond 1024 (a bidirectional layer's directions in two launches)
                                                  ("LSTM", 1280, 32, 5, True, 64), ("LSTM", 1536, 20, 4, False, 32),
                                                  ("LSTM", 2048, 32, 4, True, 64), ("LSTM", 2048, 40, 3, False, 32),
                                                  # round 5: any hidden size -- widths without a persistent kernel are padded
                                                  # to the next one that has one (two layers at H <= 256: padded input columns)
                                                  ("LSTM", 200, 20, 9, True, 40), ("LSTM", 1000, 32, 5, True, 64),
                                                  ("LSTM", 800, 33, 4, False, 32), ("LSTM", 1100, 8, 3, True, 32),
                                                  ("GRU", 200, 20, 9, True, 40), ("GRU", 800, 32, 5, True, 64),
                                                  ("GRU", 1000, 40, 4, False, 32), ("GRU", 2300, 8, 3, False, 32)])
def test_streamed_weights_step_kernel_vs_oracle(kind, H, N, T_, bidir, In):
    """Cells / sizes outside the persistent LSTM: the MFMA step kernel that streams W_hh per step (H % 64 == 0: GRU,
    tanh-RNN, wide LSTM; 1, 2 and 4 batch tiles, a batch beyond 64) and the persistent register-resident GRU
    (H = 1280 / 2560, the shipped DS2 config's width: one and two batch groups, a partial group, both directions);
    ragged lengths, random initial state, split-bf16 input projection."""
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(H + N)
    layers = 2 if H <= 256 else 1
    m = RNN(getattr(RNNType, kind), In, H, num_layers=layers, bidirectional=bidir).eval()
    rng = np.random.default_rng(H * 3 + N)
    lens = np.sort(rng.integers(1, T_ + 1, size=N))[::-1].copy()
    lens[0] = T_
    x = rng.normal(size=(T_, N, In)).astype(np.float32)
    D = 2 if bidir else 1
    h0 = (rng.normal(size=(layers * D, N, H)) * 0.3).astype(np.float32)
    sd = {k[len("rnn."):]: cpu(v) for k, v in m.state_dict().items()}
    if kind == "LSTM":
        c0 = (rng.normal(size=(layers * D, N, H)) * 0.3).astype(np.float32)
        (out, _), (hn, cn) = m((T(x), T(lens)), (T(h0), T(c0)))
        want, (whn, wcn) = O.rnn_forward(O.LSTM, x, lens, sd, H, layers, bidir, (h0, c0))
        np.testing.assert_allclose(cpu(cn), wcn, **TOL)
    else:
        (out, _), hn = m((T(x), T(lens)), T(h0))
        want, whn = O.rnn_forward(getattr(O, kind), x, lens, sd, H, layers, bidir, h0)
    np.testing.assert_allclose(cpu(out), want, **TOL)
    np.testing.assert_allclose(cpu(hn), whn, **TOL)


@pytest.mark.parametrize("kind,H,Hp", [("LSTM", 200, 256), ("LSTM", 1000, 1024), ("GRU", 800, 1024), ("GRU", 200, 512)])
def test_any_hidden_size_runs_on_a_persistent_kernel(kind, H, Hp, tmp_path):
    """hidden sizes without a persistent kernel (LSTM not a multiple of 64, GRU not one of the persistent
    widths) are padded with zero weights to the next width that has one.  The padded units stay EXACTLY zero (checked on the
    raw padded run), the padded run equals the per-step kernels within float32 rounding (they add the same products in another
    order) and the oracle within the usual tolerance, and it takes one recurrence launch per layer instead of one per step."""
    import ctypes
    import os
    import subprocess
    import sys
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model import rnn as R
    lib = _lib.load()
    cell = R._CELL[getattr(R.RNNType, kind)]
    assert lib.ms_rnn_padded_hidden(cell, H, 2) == Hp and lib.ms_rnn_padded_hidden(cell, Hp, 2) == Hp
    torch.manual_seed(H)
    T_, N, In = 40, 32, 96
    m = R.RNN(getattr(R.RNNType, kind), In, H, num_layers=2, bidirectional=True).eval()
    rng = np.random.default_rng(H)
    lens = np.sort(rng.integers(1, T_ + 1, size=N))[::-1].copy()
    lens[0] = T_
    x = rng.normal(size=(T_, N, In)).astype(np.float32)
    h0 = (rng.normal(size=(4, N, H)) * 0.3).astype(np.float32)
    hx = (T(h0), T(h0 * 0.5)) if kind == "LSTM" else T(h0)
    ms, cnt = (ctypes.c_float * _lib.PROF_KINDS)(), (ctypes.c_int * _lib.PROF_KINDS)()
    lib.ms_prof_enable(1)
    lib.ms_prof_read(ms, cnt)
    (out, _), hid = m((T(x), T(lens)), hx)
    torch.cuda.synchronize()
    lib.ms_prof_read(ms, cnt)
    lib.ms_prof_enable(0)
    assert cnt[1] == 2, f"{cnt[1]} recurrence spans for two layers"        # MS_PROF_RECURRENCE: one per layer
    sd = {k[len("rnn."):]: cpu(v) for k, v in m.state_dict().items()}
    want, whid = O.rnn_forward(getattr(O, kind), x, lens, sd, H, 2, True, (h0, h0 * 0.5) if kind == "LSTM" else h0)
    np.testing.assert_allclose(cpu(out), want, **TOL)
    for got, w in zip(hid if kind == "LSTM" else (hid,), whid if kind == "LSTM" else (whid,)):
        assert got.shape == (4, N, H)
        np.testing.assert_allclose(cpu(got), w, **TOL)
    # the raw padded run: units H .. Hp of every direction are exact zeros, in the output and in the final states
    params = m._layer_params()
    raw, rhn, rcn = R.run_layers(cell, T(x).cuda(), _lib.lens_i32(torch.as_tensor(lens)), T_, params, [R.PackedLayer(), R.PackedLayer()], H,
                                 T(h0).cuda(), T(h0 * 0.5).cuda() if kind == "LSTM" else None, _lib.Workspace(), ragged=True,
                                 keep_padding=True)
    assert raw.shape == (T_, N, 2 * Hp)
    assert float(raw.view(T_, N, 2, Hp)[..., H:].abs().max()) == 0.0 and float(rhn[..., H:].abs().max()) == 0.0
    assert rcn is None or float(rcn[..., H:].abs().max()) == 0.0
    assert torch.equal(raw.view(T_, N, 2, Hp)[..., :H].reshape(T_, N, 2 * H), out)
    # against the per-step kernels (MS_RNN_PAD_HIDDEN=0 is read once per process: a child)
    f_in, f_out = str(tmp_path / "pad_in.npz"), str(tmp_path / "pad_out.npy")      # (per test case: parametrisations share H)
    np.savez(f_in, x=x, lens=lens, h0=h0)
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from myrtlespeech_amd.model.rnn import RNN, RNNType\n"
        "torch.manual_seed(%d); m = RNN(RNNType.%s, %d, %d, num_layers=2, bidirectional=True).eval()\n"
        "d = np.load(sys.argv[1]); hx = torch.tensor(d['h0']).cuda()\n"
        "hx = (hx, hx * 0.5) if %r == 'LSTM' else hx\n"
        "(o, _), _ = m((torch.tensor(d['x']).cuda(), torch.tensor(d['lens'])), hx); np.save(sys.argv[2], o.cpu().numpy())\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), H, kind, In, H, kind)
    child = subprocess.run([sys.executable, "-c", code, f_in, f_out],
                           env=dict(os.environ, MS_RNN_PAD_HIDDEN="0"), capture_output=True, text=True, timeout=600)
    assert child.returncode == 0, child.stderr[-2000:]
    np.testing.assert_allclose(cpu(out), np.load(f_out), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cin,cout,k,s,d,T_", [(80, 512, 11, 2, 1, 1001), (512, 256, 11, 1, 1, 501), (96, 64, 5, 3, 2, 700),
                                               (33, 128, 7, 1, 1, 900)])
def test_maskconv1d_gemm_lowering_vs_oracle(cin, cout, k, s, d, T_):
    """Conv1d stacks with many input channels take the im2col + split-bf16 GEMM path (conv1d_gemm.hip): SAME padding,
    stride, dilation, ragged lengths (masked input frames), bias + clamp epilogue, odd channel counts (K padding)."""
    from myrtlespeech_amd.model.cnn import MaskConv1d, PaddingMode
    torch.manual_seed(cin + k)
    m = MaskConv1d(cin, cout, k, s, PaddingMode.SAME, dilation=d).eval()
    rng = np.random.default_rng(cout)
    N = 24
    x = rng.normal(size=(N, cin, T_)).astype(np.float32)
    lens = np.sort(rng.integers(T_ // 3, T_ + 1, size=N))[::-1].copy()
    lens[0] = T_
    y, ol = m((T(x), T(lens)))
    want, wl = O.mask_conv1d(x.copy(), lens, cpu(m.weight), cpu(m.bias), s, True, d)
    np.testing.assert_array_equal(cpu(ol), wl)
    scale = float(np.abs(want).max())
    np.testing.assert_allclose(cpu(y), want, rtol=1e-4, atol=1e-5 * max(1.0, scale))


@pytest.mark.parametrize("name", golden_names("hard_lstm_"))
def test_hard_lstm_golden(name):
    from myrtlespeech_amd.model.hard_lstm import HardLSTM
    g = Golden(name)
    c = g.cfg
    m = HardLSTM(c["input_size"], c["hidden_size"], num_layers=c["num_layers"], bidirectional=c["bidirectional"],
                 batch_first=c["batch_first"], forget_gate_bias=c["forget_gate_bias"])
    load_sd(m, g.sd())
    (out, _), (hn, cn) = m((T(g["in/x"]), T(g["in/lens"])), (T(g["in/h0"]), T(g["in/c0"])))
    np.testing.assert_allclose(cpu(out), g["out/y"], **TOL)
    np.testing.assert_allclose(cpu(hn), g["out/hn"], **TOL)
    np.testing.assert_allclose(cpu(cn), g["out/cn"], **TOL)


def test_hard_lstm_persistent_vs_oracle():
    from myrtlespeech_amd.model.hard_lstm import HardLSTM
    torch.manual_seed(5)
    m = HardLSTM(20, 64, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
    rng = np.random.default_rng(5)
    x = (rng.normal(size=(6, 9, 20)) * 2).astype(np.float32)
    (out, _), (hn, cn) = m((T(x), torch.tensor([6] * 9)))
    sd = {k[len("rnn."):]: cpu(v) for k, v in m.state_dict().items()}
    want, (whn, wcn) = O.hard_lstm_forward(x, sd, 64, 1, True)
    np.testing.assert_allclose(cpu(out), want, **TOL)
    np.testing.assert_allclose(cpu(cn), wcn, **TOL)


@pytest.mark.parametrize("H,bidir,N", [(1280, True, 9), (2048, False, 32), (1536, True, 40)])
def test_hard_lstm_beyond_1024_runs_on_the_two_stream_kernel_vs_oracle(H, bidir, N):
    """Round 6: HardLSTM (hard_lstm.py: the ONNX-exportable cell) at H = 1280 / 1536 / 2048 on the persistent two-stream kernel
    (it took a launch per step there; the plain LSTM has had these widths since round 4): against the oracle, with more than
    one batch group and -- bidirectional -- one launch per direction."""
    from myrtlespeech_amd.model.hard_lstm import HardLSTM
    torch.manual_seed(H + N)
    m = HardLSTM(24, H, num_layers=1, bidirectional=bidir, forget_gate_bias=1.0).eval()
    rng = np.random.default_rng(H + N)
    x = (rng.normal(size=(5, N, 24)) * 2).astype(np.float32)
    (out, _), (hn, cn) = m((T(x), torch.tensor([5] * N)))
    sd = {k[len("rnn."):]: cpu(v) for k, v in m.state_dict().items()}
    want, (whn, wcn) = O.hard_lstm_forward(x, sd, H, 1, bidir)
    np.testing.assert_allclose(cpu(out), want, **TOL)
    np.testing.assert_allclose(cpu(hn), whn, **TOL)
    np.testing.assert_allclose(cpu(cn), wcn, **TOL)


@pytest.mark.parametrize("hard,bidir,N", [(False, True, 40), (False, False, 64), (True, True, 37), (False, True, 33)])
def test_wide_workgroup_lstm_two_groups_vs_oracle(hard, bidir, N):
    """``lstm_persistent_wide2_kernel`` (round 3; H = 1024, 33 .. 64 sequences = two batch groups side by side in one launch):
    ragged lengths that end inside both groups, a second group that is not full, both cells, one and two directions, an
    initial state -- against the numpy oracle, and utterance by utterance against the 8-unit kernel (the same rows as two
    calls of <= 32 sequences) within float32 rounding."""
    from myrtlespeech_amd.model.hard_lstm import HardLSTM
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    H, In, steps = 1024, 32, 9
    torch.manual_seed(N)
    if hard:
        m = HardLSTM(In, H, num_layers=1, bidirectional=bidir, forget_gate_bias=1.0).eval()
    else:
        m = RNN(RNNType.LSTM, In, H, num_layers=1, bidirectional=bidir, forget_gate_bias=1.0).eval()
    rng = np.random.default_rng(N)
    x = rng.normal(size=(steps, N, In)).astype(np.float32)
    lens = np.sort(rng.integers(1, steps + 1, size=N))[::-1].copy()
    lens[0] = steps
    D = 2 if bidir else 1
    h0 = (rng.normal(size=(D, N, H)) * 0.3).astype(np.float32)
    c0 = (rng.normal(size=(D, N, H)) * 0.3).astype(np.float32)
    hx = None if hard else (T(h0).cuda(), T(c0).cuda())
    (out, _), (hn, cn) = m((T(x), T(lens)), hx=hx) if not hard else m((T(x), T(lens)))
    sd = {k[len("rnn."):]: cpu(v) for k, v in m.state_dict().items()}
    if hard:
        want, (whn, wcn) = O.hard_lstm_forward(x, sd, H, 1, bidir)
    else:
        want, (whn, wcn) = O.rnn_forward(O.LSTM, x, lens, sd, H, 1, bidir, hx=(h0, c0))
    np.testing.assert_allclose(cpu(out), want, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(cpu(hn), whn, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(cpu(cn), wcn, rtol=1e-4, atol=2e-5)
    if not hard:
        # the same utterances through the 8-unit kernel (two calls of <= 32 sequences)
        for lo, hi in ((0, 32), (32, N)):
            hx_part = (T(h0[:, lo:hi]).cuda().contiguous(), T(c0[:, lo:hi]).cuda().contiguous())
            (o2, _), (hn2, cn2) = m((T(x[:int(lens[lo]), lo:hi]).contiguous(), T(lens[lo:hi])), hx=hx_part)
            torch.testing.assert_close(out[:int(lens[lo]), lo:hi], o2, rtol=0, atol=2e-6)
            torch.testing.assert_close(cn[:, lo:hi], cn2, rtol=0, atol=2e-6)


@pytest.mark.parametrize("bidir,N,steps", [(True, 1, 23), (True, 9, 14), (False, 16, 11), (True, 17, 9), (True, 41, 9), (False, 48, 7)])
def test_wide_workgroup_lstm_small_groups_vs_oracle(bidir, N, steps):
    """``lstm_persistent_wide2_kernel`` on small batch groups: single clips (N = 1: BASELINE configs[0]'s recurrence), N = 16
    (configs[3]'s encoder), 17 rows, and two groups of which the SECOND is small (41 = 32 + 9, 48 = 32 + 16) -- two chained
    layers, ragged lengths and an initial state, against the oracle; and the rows of a small batch equal, bit for bit, the same
    rows computed inside a batch of 32 (an utterance's result does not depend on its batch).  (Round 4 built a
    one-row-stream form for groups of <= 16 rows -- the second stream is all padding there -- and measured 3.19 against 3.24 us
    per step: one stream's step IS the publish -> visible -> pull -> cell chain that two interleaved streams hide in each
    other; not kept, EXPERIMENTS.md.)"""
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    H, In = 1024, 32
    torch.manual_seed(100 + N)
    m = RNN(RNNType.LSTM, In, H, num_layers=2, bidirectional=bidir, forget_gate_bias=1.0).eval()
    rng = np.random.default_rng(100 + N)
    x = rng.normal(size=(steps, N, In)).astype(np.float32)
    lens = np.sort(rng.integers(1, steps + 1, size=N))[::-1].copy()
    lens[0] = steps
    D = 2 if bidir else 1
    h0 = (rng.normal(size=(2 * D, N, H)) * 0.3).astype(np.float32)
    c0 = (rng.normal(size=(2 * D, N, H)) * 0.3).astype(np.float32)
    (out, _), (hn, cn) = m((T(x), T(lens)), hx=(T(h0).cuda(), T(c0).cuda()))
    sd = {k[len("rnn."):]: cpu(v) for k, v in m.state_dict().items()}
    want, (whn, wcn) = O.rnn_forward(O.LSTM, x, lens, sd, H, 2, bidir, hx=(h0, c0))
    np.testing.assert_allclose(cpu(out), want, rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(cpu(hn), whn, rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(cpu(cn), wcn, rtol=1e-4, atol=3e-5)
    if N <= 16:
        # the same utterances as rows 0 .. N-1 of a batch of 32 (two row streams): identical bits
        xb = np.concatenate([x, rng.normal(size=(steps, 32 - N, In)).astype(np.float32)], 1)
        lb = np.concatenate([lens, np.ones(32 - N, dtype=lens.dtype)])
        hb = np.concatenate([h0, np.zeros((2 * D, 32 - N, H), np.float32)], 1)
        cb = np.concatenate([c0, np.zeros((2 * D, 32 - N, H), np.float32)], 1)
        (ob, _), (hnb, cnb) = m((T(xb), T(lb)), hx=(T(hb).cuda(), T(cb).cuda()))
        assert torch.equal(ob[:, :N], out) and torch.equal(hnb[:, :N], hn) and torch.equal(cnb[:, :N], cn)


# ----------------------------------------------------------------------------- conv
@pytest.mark.parametrize("name", golden_names("conv2d_"))
def test_conv2d_golden(name):
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    g = Golden(name)
    c = g.cfg
    m = MaskConv2d(c["in_channels"], c["out_channels"], c["kernel_size"], c["stride"],
                   PaddingMode.SAME if c["same"] else PaddingMode.NONE)
    load_sd(m, g.sd())
    x = T(g["in/x"]).cuda()
    y, lens = m((x, T(g["in/lens"])))
    np.testing.assert_allclose(cpu(y), g["out/y"], **TOL)
    np.testing.assert_array_equal(cpu(lens), g["out/lens"])
    assert cpu(lens).dtype == g["out/lens"].dtype
    np.testing.assert_array_equal(cpu(x), g["out/x_after"])  # input masked in place like the reference


@pytest.mark.parametrize("name", golden_names("conv1d_"))
def test_conv1d_golden(name):
    from myrtlespeech_amd.model.cnn import MaskConv1d, PaddingMode
    g = Golden(name)
    c = g.cfg
    m = MaskConv1d(c["in_channels"], c["out_channels"], c["kernel_size"], c["stride"],
                   PaddingMode.SAME if c["same"] else PaddingMode.NONE)
    load_sd(m, g.sd())
    y, lens = m((T(g["in/x"]), T(g["in/lens"])))
    np.testing.assert_allclose(cpu(y), g["out/y"], **TOL)
    np.testing.assert_array_equal(cpu(lens), g["out/lens"])


def test_conv2d_groups_dilation_vs_oracle():
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    torch.manual_seed(3)
    m = MaskConv2d(4, 6, [3, 4], [1, 2], PaddingMode.SAME, groups=2).eval()
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 4, 7, 21)).astype(np.float32)
    lens = np.array([21, 13])
    y, nl = m((T(x), T(lens)))
    want, wl = O.mask_conv2d(x, lens, cpu(m.weight), cpu(m.bias), (1, 2), True, (1, 1), groups=2)
    np.testing.assert_allclose(cpu(y), want, **TOL)
    np.testing.assert_array_equal(cpu(nl), wl)


# ----------------------------------------------------------------------------- fc / lookahead
@pytest.mark.parametrize("name", golden_names("fc_"))
def test_fc_golden(name):
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    g = Golden(name)
    c = g.cfg
    act = None
    if c["act"] == "relu":
        act = torch.nn.ReLU()
    elif c["act"] is not None:
        act = torch.nn.Hardtanh(*c["act"])
    m = FullyConnected(c["in_features"], c["out_features"], c["num_hidden_layers"], c["hidden_size"], act)
    load_sd(m, g.sd())
    y, _ = m((T(g["in/x"]), T(g["in/lens"])))
    np.testing.assert_allclose(cpu(y), g["out/y"], **TOL)


@pytest.mark.parametrize("name", golden_names("lookahead_"))
def test_lookahead_golden(name):
    from myrtlespeech_amd.model.lookahead import Lookahead
    g = Golden(name)
    m = Lookahead(g.cfg["in_features"], g.cfg["context"])
    load_sd(m, g.sd())
    y, _ = m((T(g["in/x"]), T(g["in/lens"])))
    np.testing.assert_allclose(cpu(y), g["out/y"], **TOL)
    # strided (feature-contiguous) variant: hand over a permuted view of a [T,N,F] tensor
    x_tnf = T(g["in/x"]).permute(2, 0, 1).contiguous().cuda()
    y2, _ = m((x_tnf.permute(1, 2, 0), T(g["in/lens"])))
    np.testing.assert_allclose(cpu(y2), g["out/y"], **TOL)


# ----------------------------------------------------------------------------- DS2 / DS1
def build_ds2(cfg):
    from myrtlespeech_amd.model.cnn import Conv1dTo2d, Conv2dTo1d, MaskConv1d, MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    from myrtlespeech_amd.model.lookahead import Lookahead
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper

    def act(a):
        return torch.nn.Identity() if a is None else torch.nn.Hardtanh(*a)

    layers, dims = [], 4
    for c in cfg["convs"]:
        pm = PaddingMode.SAME if c["same"] else PaddingMode.NONE
        if c["kind"] == "conv2d":
            if dims == 3:
                layers.append(Conv1dTo2d())
                dims = 4
            layers.append(MaskConv2d(c["in_channels"], c["out_channels"], c["kernel"], c["stride"], pm))
        else:
            if dims == 4:
                layers.append(Conv2dTo1d())
                dims = 3
            layers.append(MaskConv1d(c["in_channels"], c["out_channels"], c["kernel"], c["stride"], pm))
        layers.append(SeqLenWrapper(act(c["act"]), torch.nn.Identity()))
    if dims == 3:
        layers.append(Conv1dTo2d())
    r = cfg["rnn"]
    rnn = RNN(RNNType(r["kind"]), r["input"], r["hidden"], num_layers=r["layers"], bidirectional=r["bidirectional"],
              forget_gate_bias=r["forget_gate_bias"])
    la = None
    if cfg["lookahead"] is not None:
        la = torch.nn.Sequential(Lookahead(r["hidden"] * (2 if r["bidirectional"] else 1), cfg["lookahead"]["context"]),
                                 SeqLenWrapper(act(cfg["lookahead"]["act"]), torch.nn.Identity()))
    f = cfg["fc"]
    fc = FullyConnected(f["in_features"], f["out_features"], f["n_hidden"], f["hidden"],
                        None if f["act"] is None else torch.nn.Hardtanh(*f["act"]))
    return DeepSpeech2(torch.nn.Sequential(*layers), rnn, la, fc)


@pytest.mark.parametrize("name", golden_names("ds2_tiny"))
def test_ds2_tiny_golden(name):
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    g = Golden(name)
    m = load_sd(build_ds2(g.cfg), g.sd())
    hx = T(g["in/h0"]) if g.has("in/h0") else None
    (y, lens), hid = m((T(g["in/x"]), T(g["in/lens"])), hx)
    np.testing.assert_allclose(cpu(y), g["out/y"], **TOL)
    np.testing.assert_array_equal(cpu(lens), g["out/lens"])
    hn = hid[0] if isinstance(hid, tuple) else hid
    np.testing.assert_allclose(cpu(hn), g["out/hn"], **TOL)
    if g.has("out/greedy_flat"):
        dec = CTCGreedyDecoder(g.cfg["blank"])(y, lens)
        assert dec == unragged(g["out/greedy_flat"], g["out/greedy_lens"])


@pytest.mark.parametrize("name", golden_names("ds1_tiny"))
def test_ds1_tiny_golden(name):
    from myrtlespeech_amd.model.deep_speech_1 import DeepSpeech1
    g = Golden(name)
    c = g.cfg
    m = DeepSpeech1(c["input_features"], c["input_channels"], c["n_hidden"], c["out_features"], drop_prob=0.25,
                    relu_clip=c["relu_clip"], hard_lstm=c["hard_lstm"])
    load_sd(m, g.sd())
    (y, lens), hid = m((T(g["in/x"]), T(g["in/lens"])))
    np.testing.assert_allclose(cpu(y), g["out/y"], **TOL)
    np.testing.assert_allclose(cpu(hid[0]), g["out/hn"], **TOL)
    np.testing.assert_allclose(cpu(hid[1]), g["out/cn"], **TOL)


@pytest.mark.parametrize("name", ["ds1_cfg1_summary", "ds1_cfg1_hard_summary", "ds1_cfg1_trained_summary",
                                  "ds1_cfg1_hard_trained_summary"])
def test_ds1_shipped_width_vs_reference_summary(name):
    """BASELINE.json configs[0]: DS1 at the shipped width (n_hidden 1024, input [1, 19, 26, 201]) and a ragged batch of
    3, torch-LSTM and HardLSTM flavours; weights and inputs regenerated from the generator's seeds (checksums pinned),
    logits / states on the stored sub-grids within 1e-3 of the reference, greedy transcripts bit-exact."""
    from myrtlespeech_amd.model.deep_speech_1 import DeepSpeech1
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    g = Golden(name)
    hard = g.cfg["hard_lstm"]
    torch.manual_seed(g.cfg["seed_weights"])
    m = DeepSpeech1(input_features=26, input_channels=19, n_hidden=1024, out_features=29, drop_prob=0.25,
                    relu_clip=20.0, forget_gate_bias=1.0, hard_lstm=hard).eval()
    if "gains" in g.cfg:     # the *_trained_* twins: logits of mean ~3 / max ~14, ~45 % of the BiLSTM's gates beyond |4|
        from util import apply_ds1_trained_gains
        with torch.no_grad():
            apply_ds1_trained_gains(m, g.cfg["gains"])
    for k, v in m.state_dict().items():
        assert abs(float(v.double().abs().sum()) - g.cfg["weight_abs_sums"][k]) <= 1e-6 * max(1.0, g.cfg["weight_abs_sums"][k]), k
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    x1 = torch.randn(1, 19, 26, 201, generator=gen)
    x3 = torch.randn(3, 19, 26, 120, generator=gen)
    l3 = T(g["in/l3"])
    (y1, o1), h1 = m((x1, torch.tensor([201])))
    (y3, o3), h3 = m((x3, l3))
    np.testing.assert_allclose(cpu(y1[::5]), g["out/y1_sub"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(cpu(y3[::4]), g["out/y3_sub"], rtol=0, atol=1e-3)
    np.testing.assert_array_equal(cpu(o3), g["out/o3"])
    np.testing.assert_allclose(cpu(h1[0][:, :, ::32]), g["out/hn1_sub"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(cpu(h3[1][:, :, ::32]), g["out/cn3_sub"], rtol=0, atol=1e-3)
    dec = CTCGreedyDecoder(28)
    assert dec(y1, o1) == unragged(g["out/g1_flat"], g["out/g1_lens"])
    assert dec(y3, o3) == unragged(g["out/g3_flat"], g["out/g3_lens"])
    if g.has("out/am1"):
        assert np.array_equal(cpu(y1.argmax(-1))[:, 0], g["out/am1"][:, 0].astype(np.int64))
    print(f"{name}: max |logit err| {float(np.abs(cpu(y1[::5]) - g['out/y1_sub']).max()):.3e}"
          + (f" (vs the float64 twin {float(np.abs(cpu(y1[::5]) - g['out/y1d_sub']).max()):.3e}; the reference itself "
             f"{g.cfg['stats_clip']['ref_f32_vs_f64_max_abs']:.3e})" if g.has("out/y1d_sub") else ""))


# ----------------------------------------------------------------------------- CTC loss / greedy
def test_ctc_loss_golden():
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    g = Golden("ctc_loss_small")
    b = g.cfg["blank"]
    for red in ("none", "mean", "sum"):
        for zi in (0, 1):
            got = CTCLoss(blank=b, reduction=red, zero_infinity=bool(zi))(
                (T(g["in/x"]), T(g["in/x_lens"])), (T(g["in/y"]), T(g["in/y_lens"])))
            np.testing.assert_allclose(cpu(got), g[f"out/{red}_{zi}"], rtol=1e-4, atol=1e-4)
    got = CTCLoss(blank=b, reduction="none")((T(g["in/x"]), T(g["in/x_lens"])), (T(g["in/y_flat"]), T(g["in/y_lens"])))
    np.testing.assert_allclose(cpu(got), g["out/none_flat"], rtol=1e-4, atol=1e-4)
    g = Golden("ctc_loss_v29")
    for red in ("none", "mean", "sum"):
        got = CTCLoss(blank=28, reduction=red)((T(g["in/x"]), T(g["in/x_lens"])), (T(g["in/y"]), T(g["in/y_lens"])))
        np.testing.assert_allclose(cpu(got), g[f"out/{red}_0"], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("V", [29, 200])      # 200: phase 1 as the chip-wide launch, its flags handed to the pipeline kernel
def test_ctc_loss_nan_logits_give_nan_like_torch_and_no_timeout_status(V):
    """the four-wave pipeline clamps normalised log-probabilities to a finite "log zero"; a NaN logit row used to
    come out as nll = +inf, which ``zero_infinity`` silently turned into 0 (loss AND gradient) where torch.nn.CTCLoss gives
    NaN.  A non-finite normaliser now poisons that utterance's loss -- and only that one's -- and is NOT a time-out."""
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    rng = np.random.default_rng(11)
    x = rng.normal(size=(60, 4, V)).astype(np.float32)
    xl = np.array([60, 55, 50, 40], dtype=np.int32)
    y = rng.integers(0, V - 1, size=(4, 12)).astype(np.int32)
    yl = np.array([12, 10, 8, 5], dtype=np.int32)
    bad = x.copy()
    bad[17, 1, 3] = np.nan           # utterance 1, inside its length
    bad[45, 3, :] = 7.0              # utterance 3, past its length (40): ignored
    bad[45, 3, 2] = np.nan
    bad[20, 2, 5] = np.inf           # utterance 2: log_softmax gives NaN at that symbol
    want = torch.nn.CTCLoss(blank=V - 1, reduction="none", zero_infinity=True)(
        torch.log_softmax(torch.from_numpy(bad), -1), torch.from_numpy(y).long(), torch.from_numpy(xl).long(), torch.from_numpy(yl).long()).numpy()
    assert np.isnan(want[1]) and np.isfinite(want[0]) and np.isfinite(want[3])
    for zi in (False, True):
        loss = CTCLoss(blank=V - 1, reduction="none", zero_infinity=zi)      # check_status on: a time-out would raise here
        got = cpu(loss((T(bad), T(xl)), (T(y), T(yl))))
        assert np.isnan(got[1]) and np.isnan(got[2])
        np.testing.assert_allclose(got[[0, 3]], want[[0, 3]], rtol=1e-4, atol=1e-4)
        loss.status()                                                      # nothing sticky was left behind
    # the same through autograd: the poisoned utterances' gradients are NaN, the others' match torch
    xt = T(bad).requires_grad_(True)
    loss = CTCLoss(blank=V - 1, reduction="sum", zero_infinity=True)
    loss((xt, T(xl)), (T(y), T(yl))).backward()
    g = cpu(xt.grad)
    assert np.isnan(g[:, 1]).any() and np.isfinite(g[:, 0]).all() and np.isfinite(g[:, 3]).all()
    # C ABI: ms_ctc_status on a clean workspace is MS_OK; a set word is reported once as MS_ERR_TIMEOUT and cleared
    lib = _lib.load()
    ws = torch.zeros(lib.ms_ctc_loss_workspace_bytes(60, 4, V, 25), dtype=torch.uint8, device="cuda")
    assert lib.ms_ctc_status(_lib.ptr(ws), _lib.stream_ptr()) == 0
    ws[0] = 1
    assert lib.ms_ctc_status(_lib.ptr(ws), _lib.stream_ptr()) == 4          # MS_ERR_TIMEOUT
    assert lib.ms_ctc_status(_lib.ptr(ws), _lib.stream_ptr()) == 0


def test_ctc_loss_dim_golden():
    """VERDICT r2 missing 5: ``CTCLoss(dim != -1)`` runs like the reference (ctc_loss.py:37-45 forwards any dim): values
    normalised over time or over the batch go to the alpha recursion as they are.  Fixture made by the reference."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    g = Golden("ctc_loss_dim")
    for dim in g.cfg["dims"]:
        for red in ("none", "mean", "sum"):
            got = CTCLoss(blank=g.cfg["blank"], reduction=red, dim=dim)(
                (T(g["in/x"]), T(g["in/x_lens"])), (T(g["in/y"]), T(g["in/y_lens"])))
            np.testing.assert_allclose(cpu(got), g[f"out/dim{dim}_{red}"], rtol=1e-4, atol=1e-4)
    # larger, against the oracle, batch axis
    rng = np.random.default_rng(3)
    x = rng.normal(size=(200, 9, 29)).astype(np.float32)
    xl = np.sort(rng.integers(100, 201, size=9))[::-1].astype(np.int32)
    yl = rng.integers(1, 31, size=9).astype(np.int32)
    y = rng.integers(0, 28, size=(9, 30)).astype(np.int32)
    for dim in (0, 1):
        got = CTCLoss(blank=28, reduction="none", dim=dim)((T(x), T(xl)), (T(y), T(yl)))
        np.testing.assert_allclose(cpu(got), O.ctc_loss(x, xl, y, yl, 28, "none", dim=dim), rtol=2e-4, atol=1e-2)
    with pytest.raises(IndexError):
        CTCLoss(blank=28, dim=3)((T(x), T(xl)), (T(y), T(yl)))


def test_ctc_loss_dim_backward_matches_reference_autograd():
    """VERDICT r3 missing 5: ``loss.backward()`` through ``CTCLoss(dim != -1)`` -- the alpha-beta kernel on the values normalised
    over time / over the batch, LogSoftmax(dim)'s backward behind it -- against x.grad from the reference (ctc_loss.py:37-45,
    95-101), every reduction, with and without zero_infinity; and at a larger shape against the float64 oracle."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    g = Golden("ctc_grad_dim")
    w = T(g["in/w"]).cuda()
    for key in [k for k in g.a if k.startswith("grad/")]:
        dim, red, zi = key[len("grad/dim"):].split("_")
        x = T(g["in/x"]).cuda().requires_grad_(True)
        out = CTCLoss(blank=g.cfg["blank"], reduction=red, zero_infinity=bool(int(zi)), dim=int(dim))(
            (x, T(g["in/x_lens"])), (T(g["in/y"]), T(g["in/y_lens"])))
        np.testing.assert_allclose(cpu(out), g["out/" + key[len("grad/"):]], rtol=1e-4, atol=1e-4)
        ((out * w).sum() if red == "none" else out * g.cfg["scale"]).backward()
        np.testing.assert_allclose(cpu(x.grad), g[key], rtol=1e-4, atol=2e-5)
    rng = np.random.default_rng(5)
    x = rng.normal(size=(120, 6, 29)).astype(np.float32)
    xl = np.sort(rng.integers(60, 121, size=6))[::-1].astype(np.int32)
    yl = rng.integers(1, 21, size=6).astype(np.int32)
    y = rng.integers(0, 28, size=(6, 20)).astype(np.int32)
    for dim in (0, 1):
        xg = T(x).cuda().requires_grad_(True)
        CTCLoss(blank=28, reduction="sum", dim=dim)((xg, T(xl)), (T(y), T(yl))).backward()
        want = O.ctc_grad(x, xl, y, yl, np.ones(6, np.float32), 28, dim=dim)
        np.testing.assert_allclose(cpu(xg.grad), want, rtol=2e-3, atol=2e-4)


def test_ctc_loss_full_size_vs_oracle():
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    rng = np.random.default_rng(9)
    Tn, N, V, S = 501, 32, 29, 120
    x = rng.normal(size=(Tn, N, V)).astype(np.float32)
    xl = np.sort(rng.integers(300, Tn + 1, size=N))[::-1].astype(np.int32)
    yl = rng.integers(1, S + 1, size=N).astype(np.int32)
    y = rng.integers(0, 28, size=(N, S)).astype(np.int32)
    got = CTCLoss(blank=28, reduction="none")((T(x), T(xl)), (T(y), T(yl)))
    want = O.ctc_loss(x, xl, y, yl, 28, "none")
    np.testing.assert_allclose(cpu(got), want, rtol=2e-4, atol=1e-2)


@pytest.mark.parametrize("Tn,N,V,L", [(501, 32, 29, 120), (64, 9, 29, 20), (17, 4, 5, 3), (16, 3, 7, 8), (15, 3, 7, 7),
                                      (33, 5, 29, 16), (2, 3, 4, 1), (1, 2, 4, 1), (700, 6, 40, 255), (300, 5, 29, 128),
                                      (400, 4, 29, 200), (900, 3, 33, 400), (1200, 2, 29, 511), (1300, 2, 29, 600),
                                      (120, 3, 100, 30), (90, 2, 300, 25), (60, 2, 65, 12),      # wide alphabets: a wave per frame
                                      (70, 3, 1500, 20), (40, 2, 5000, 9)])                       # ... rows beyond one 1 024-symbol slab
def test_ctc_alpha_wave_pipeline_vs_the_lds_row_kernel_and_the_oracle(Tn, N, V, L, monkeypatch):
    """The four-wave pipeline (alphas in registers, DPP + a per-frame LDS mailbox; csrc/ctc.hip) against the LDS-row kernel it
    replaced (MS_CTC_WAVE=0, read per call) and against the oracle.  (Its first form shared lse3 with that kernel and was
    bit-identical to it on these shapes: the exchange logic was checked that way before the arithmetic was changed.)
    Shapes walk the states-per-lane instantiations (S <= 256 / 512 / 1024 and the fallback beyond), ring boundaries
    (T = 15, 16, 17, 33), ragged and empty inputs, empty targets and repeated labels."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    rng = np.random.default_rng(Tn * 7 + L)
    x = (rng.normal(size=(Tn, N, V)) * 3).astype(np.float32)
    xl = rng.integers(max(1, Tn // 2), Tn + 1, size=N).astype(np.int32)
    xl[0] = Tn
    if N > 2:
        xl[-1] = 0
    yl = rng.integers(0, L + 1, size=N).astype(np.int32)
    yl[0] = L
    if N > 1:
        yl[1] = 0
    y = rng.integers(0, V - 1, size=(N, max(L, 1))).astype(np.int32)
    for i in range(1, y.shape[1]):                                # repeated labels: the skip transition is off there
        y[:, i] = np.where(rng.random(N) < 0.2, y[:, i - 1], y[:, i])
    loss = CTCLoss(blank=V - 1, reduction="none")
    monkeypatch.setenv("MS_CTC_WAVE", "0")
    old = cpu(loss((T(x), T(xl)), (T(y), T(yl))))
    monkeypatch.setenv("MS_CTC_WAVE", "1")
    got = cpu(loss((T(x), T(xl)), (T(y), T(yl))))
    # the pipeline works in the log2 domain on the hardware exp2 / log2 and a finite "log zero": rounding differs, nothing else
    fin_old = np.isfinite(old)
    np.testing.assert_array_equal(np.isfinite(got), fin_old)
    np.testing.assert_array_equal(got[~fin_old], old[~fin_old])
    np.testing.assert_allclose(got[fin_old], old[fin_old], rtol=2e-6, atol=2e-5)
    if Tn * L <= 501 * 130:
        want = O.ctc_loss(x, xl, y, yl, V - 1, "none")
        fin = np.isfinite(want)
        assert (np.isfinite(got) == fin).all()
        np.testing.assert_allclose(got[fin], want[fin], rtol=2e-4, atol=1e-2)


@pytest.mark.parametrize("Tn,N,V,L,zi", [(501, 8, 29, 120, False), (64, 9, 29, 20, True), (17, 4, 5, 3, False), (16, 3, 7, 8, True),
                                         (33, 5, 29, 16, False), (2, 3, 4, 1, True), (1, 2, 4, 1, False), (300, 5, 29, 128, False),
                                         (400, 4, 29, 200, True), (700, 3, 33, 400, False), (100, 3, 130, 20, False),
                                         (80, 2, 70, 15, True), (50, 2, 1200, 10, False), (30, 2, 5000, 6, True)])
def test_ctc_gradient_pipeline_vs_the_lds_row_kernel_and_the_oracle(Tn, N, V, L, zi, monkeypatch):
    """The backward of the loss on the pipeline path (alpha rows, beta rows = the same kernel on the reversed utterance,
    then one wave per frame for the gradient row; csrc/ctc.hip) against ``ctc_grad_kernel`` (MS_CTC_WAVE=0, read per call) and
    against the oracle: ragged and empty inputs, empty and impossible targets (with and without ``zero_infinity``), repeated
    labels, every states-per-lane form."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    rng = np.random.default_rng(Tn * 11 + L)
    x = (rng.normal(size=(Tn, N, V)) * 2).astype(np.float32)
    xl = rng.integers(max(1, Tn // 2), Tn + 1, size=N).astype(np.int32)
    xl[0] = Tn
    if N > 2:
        xl[-1] = 0
    yl = rng.integers(0, L + 1, size=N).astype(np.int32)
    yl[0] = L
    if N > 1:
        yl[1] = 0
    if N > 3:
        xl[2], yl[2] = max(1, L // 2), L            # more labels than frames: no path
    y = rng.integers(0, V - 1, size=(N, max(L, 1))).astype(np.int32)
    for i in range(1, y.shape[1]):
        y[:, i] = np.where(rng.random(N) < 0.2, y[:, i - 1], y[:, i])
    wts = rng.uniform(0.5, 1.5, size=N).astype(np.float32)
    loss = CTCLoss(blank=V - 1, reduction="none", zero_infinity=zi)
    grads = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MS_CTC_WAVE", flag)
        xt = T(x).cuda().requires_grad_(True)
        out = loss((xt, T(xl)), (T(y), T(yl)))
        fin = torch.isfinite(out)
        (out[fin] * T(wts).cuda()[fin]).sum().backward()
        grads.append(cpu(xt.grad))
    old, got = grads
    assert np.isfinite(got).all() == np.isfinite(old).all()
    # float32 log-domain values of magnitude ~T carry ~T 2^-24 of absolute error into a posterior (both kernels, and the
    # reference's float32 path): 1e-4 at a few hundred frames, 1e-3 at 501 (the tolerance of the config-size tests)
    # (measured against the float64 oracle, tools/ctc_grad_err.py: both kernels 6e-4 .. 2.6e-3 at 501 frames, 1e-4 at 64)
    tol = 3e-4 if Tn <= 128 else 4e-3
    np.testing.assert_allclose(got, old, rtol=1e-4, atol=2 * tol)
    if Tn * L <= 501 * 130:
        nll = O.ctc_loss(x, xl, y, yl, V - 1, "none")
        ok = np.isfinite(nll)                      # utterances with a path (the others: zeros or softmax rows, checked above)
        want = O.ctc_grad(x[:, ok], xl[ok], y[ok], yl[ok], wts[ok], V - 1, zi)
        np.testing.assert_allclose(got[:, ok], want, rtol=1e-3, atol=tol)


def test_greedy_golden():
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    g = Golden("greedy_ties")
    for b in g.cfg["blanks"]:
        got = CTCGreedyDecoder(b)(T(g["in/x"]), T(g["in/lens"]))
        assert got == unragged(g[f"out/flat_b{b}"], g[f"out/lens_b{b}"])


@pytest.mark.parametrize("Tn,N,V", [(501, 32, 29), (1000, 3, 5), (257, 5, 64), (1, 1, 2),
                                    # alphabets beyond 64 symbols: a wave per frame finds the arg max (round 6)
                                    (300, 7, 65), (257, 5, 200), (130, 3, 5000)])
def test_greedy_vs_oracle(Tn, N, V):
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    rng = np.random.default_rng(Tn + N + V)
    x = (rng.normal(size=(Tn, N, V)) * 2).round().astype(np.float32) / 2  # many ties and repeats
    lens = rng.integers(0, Tn + 1, size=N).astype(np.int64)
    lens[0] = Tn
    got = CTCGreedyDecoder(V - 1)(T(x), T(lens))
    assert got == O.ctc_greedy_decode(x, lens, V - 1)


@pytest.mark.parametrize("V", [29, 100])
def test_greedy_counts_nan_as_the_maximum_like_torch_argmax(V):
    """A NaN logit is the frame's arg max, the first one if there are several (torch.argmax, which the reference calls:
    ctc_greedy_decoder.py:73) -- on the thread-per-frame path and on the wave-per-frame one (V > 64)."""
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    rng = np.random.default_rng(V)
    x = rng.normal(size=(90, 4, V)).astype(np.float32)
    for t, n, v in [(3, 0, V - 2), (3, 0, 5), (10, 1, 0), (40, 2, V - 1), (41, 2, 70 % V), (41, 2, 7), (89, 3, 1)]:
        x[t, n, v] = np.nan
    lens = np.array([90, 80, 90, 90], dtype=np.int64)
    want = [torch.unique_consecutive(torch.from_numpy(x[:l, n]).argmax(-1)).tolist() for n, l in enumerate(lens)]
    want = [[v for v in w if v != V - 1] for w in want]
    assert CTCGreedyDecoder(V - 1)(T(x), T(lens)) == want


# ----------------------------------------------------------------------------- config-2, full size
def test_ds2_cfg2_full_size_vs_reference_summary():
    """BASELINE.json configs[1] at full size (32 x 1001 frames, 5xBiLSTM-1024): weights and
    inputs are regenerated from the seeds the golden generator used (weight checksums
    pinned), logits compared on the stored sub-grid within the north-star's 1e-3, greedy
    transcripts bit-exact (default f16x3 mode; the f32 twin runs in a child: test_gpu_configs.py)."""
    import cfg_checks
    err = cfg_checks.cfg2_full(atol=1e-3)
    assert err < 1e-5   # measured 2.5e-7


@pytest.mark.parametrize("name", ["ds2_shipped_summary", "ds2_shipped_trained_summary"])
def test_ds2_shipped_architecture_full_width_vs_reference_summary(name):
    """The reference's SHIPPED config shape (2 x conv2d, 3 x GRU-2560 unidirectional, lookahead 80, FC 1 x 1024) at full
    width: weights and inputs regenerated from the golden generator's seeds (checksums pinned), logits on the stored
    sub-grid and the final hidden state within 1e-3 of the reference, greedy transcripts bit-exact.  Exercises the
    persistent register-resident GRU and the register-tiled lookahead against the reference itself."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    from myrtlespeech_amd.model.lookahead import Lookahead
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    g = Golden(name)

    def act():
        return SeqLenWrapper(torch.nn.Hardtanh(0.0, 20.0), torch.nn.Identity())
    torch.manual_seed(g.cfg["seed_weights"])
    cnn = torch.nn.Sequential(MaskConv2d(1, 32, [41, 11], [2, 2], PaddingMode.SAME), act(),
                              MaskConv2d(32, 32, [21, 11], [2, 1], PaddingMode.SAME), act())
    rnn = RNN(RNNType.GRU, 640, 2560, num_layers=3, bidirectional=False)
    la = torch.nn.Sequential(Lookahead(2560, 80), SeqLenWrapper(torch.nn.Identity(), torch.nn.Identity()))
    fc = FullyConnected(2560, 29, 1, 1024, torch.nn.Hardtanh(0.0, 20.0))
    model = DeepSpeech2(cnn, rnn, la, fc).eval()
    if "gains" in g.cfg:     # the trained-scale twin: GRU weight_ih x 8, weight_hh x 2, FC x 6 -> logits of mean 2.4, max 13
        from util import apply_trained_gains
        with torch.no_grad():
            apply_trained_gains(model, g.cfg["gains"])
    for k, v in model.state_dict().items():
        assert abs(float(v.double().abs().sum()) - g.cfg["weight_abs_sums"][k]) <= 1e-6 * max(1.0, g.cfg["weight_abs_sums"][k]), k
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    N, Tn = g.cfg["N"], g.cfg["T"]
    x = torch.randn(N, 1, 80, Tn, generator=gen)
    lens = torch.sort(torch.randint(150, Tn + 1, (N,), generator=gen), descending=True).values
    lens[0] = Tn
    assert abs(float(x.double().abs().sum()) - float(g["in/x_abs_sum"])) < 1e-3
    np.testing.assert_array_equal(lens.numpy(), g["in/lens"])
    (y, ol), hn = model((x, lens))
    np.testing.assert_array_equal(cpu(ol), g["out/lens"])
    np.testing.assert_allclose(cpu(y[::10, ::2, :]), g["out/y_sub"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(cpu(hn[:, :, ::64]), g["out/hn_sub"], rtol=0, atol=1e-3)
    assert CTCGreedyDecoder(28)(y, ol) == unragged(g["out/greedy_flat"], g["out/greedy_lens"])
    err = float(np.abs(cpu(y[::10, ::2, :]) - g["out/y_sub"]).max())
    if g.has("out/argmax"):
        am, ref = cpu(y.argmax(-1)), g["out/argmax"].astype(np.int64)
        for n_ in range(N):
            assert np.array_equal(am[:int(g["out/lens"][n_]), n_], ref[:int(g["out/lens"][n_]), n_]), n_
        print(f"{name}: max |logit err| {err:.3e} (vs the float64 twin {float(np.abs(cpu(y[::10, ::2, :]) - g['out/yd_sub']).max()):.3e}; "
              f"the reference itself {g.cfg['stats']['ref_f32_vs_f64_max_abs']:.3e}; mean |logit| {g.cfg['stats']['logit_abs_mean']:.2f})")
    else:
        print(f"shipped-architecture max |logit err| on the sub-grid: {err:.3e} (mean |logit| {float(g['out/y_abs_mean']):.3e})")


# ----------------------------------------------------------------------------- streaming with carried context (a16)
@pytest.mark.parametrize("name", ["ds2_tiny_gru_lookahead", "ds2_tiny_ctx_lstm_even_kernel", "ds2_tiny_ctx_gru_lookahead_act"])
def test_streaming_with_carried_context_equals_the_full_utterance_reference(name):
    """``ChunkedDeepSpeech2(carry_context=True)``: chunk outputs concatenated == the reference's FULL-utterance
    ``DeepSpeech2.forward`` (deep_speech_2.py:123-172 on the whole clip; fixtures made by the reference): logits, output
    lengths, final state, greedy transcripts -- for chunk sizes from one frame to the whole clip.  The stacks cover a conv1d
    block, an even time kernel with stride 3 (the SAME split then depends on the padded length, cnn.py:148-163), stride-2
    layers, ragged lengths down to 5 frames, a lookahead with and without an activation, an initial state."""
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    g = Golden(name)
    m = load_sd(build_ds2(g.cfg), g.sd()).eval()
    hx = T(g["in/h0"]) if g.has("in/h0") else None
    total = g["in/x"].shape[-1]
    for chunk in (1, 2, 5, 8, 13, 32, total):
        st = ChunkedDeepSpeech2(m, chunk, carry_context=True)
        (y, lens), hid = st(T(g["in/x"]), T(g["in/lens"]), hx)
        np.testing.assert_allclose(cpu(y), g["out/y"], **TOL, err_msg=f"chunk {chunk}")
        np.testing.assert_array_equal(cpu(lens), g["out/lens"])
        hn = hid[0] if isinstance(hid, tuple) else hid
        np.testing.assert_allclose(cpu(hn), g["out/hn"], **TOL)
        if g.has("out/cn"):
            np.testing.assert_allclose(cpu(hid[1]), g["out/cn"], **TOL)
        if g.has("out/greedy_flat"):
            assert CTCGreedyDecoder(g.cfg["blank"])(y, lens) == unragged(g["out/greedy_flat"], g["out/greedy_lens"])
    # the explicit stream interface: rows come out as soon as their context has arrived, never earlier
    st = ChunkedDeepSpeech2(m, 4, carry_context=True)
    st.begin(T(g["in/lens"]), total, hx)
    lat = st.latency_frames(total)
    x = T(g["in/x"]).cuda()
    rows, t0 = 0, 0
    while t0 < total:
        out = st.push(x[..., t0:t0 + 4], final=(t0 + 4 >= total))
        t0 += 4
        if t0 < lat and t0 < total:
            assert out is None, (t0, lat)
        rows += 0 if out is None else out.shape[0]
    assert rows == g["out/y"].shape[0]
    with pytest.raises(ValueError):
        ChunkedDeepSpeech2(load_sd(build_ds2(Golden("ds2_tiny_bilstm").cfg), Golden("ds2_tiny_bilstm").sd()), 8,
                           carry_context=True)(T(Golden("ds2_tiny_bilstm")["in/x"]), T(Golden("ds2_tiny_bilstm")["in/lens"]))


@pytest.mark.parametrize("name", ["ds2_tiny_gru_lookahead", "ds2_tiny_ctx_lstm_even_kernel", "ds2_tiny_ctx_gru_lookahead_act"])
def test_streaming_with_carried_context_hip_graph_replay_equals_the_eager_pushes(name):
    """Steady-state pushes of the carried-context mode replay a captured HIP graph (``streaming._ContextGraph``: the
    convolutions' cached frames, the lookahead's held-back rows and the recurrent state in static tensors, the stream's
    counters advanced on the host): the same launches on the same buffers as the eager push, so ``torch.equal`` logits, lengths
    and states -- on a long clip (the fixture's input tiled in time), ragged lengths (utterances end inside the stream: the
    tail runs eagerly), several chunk sizes, and with a second batch re-using the first one's graph."""
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    g = Golden(name)
    m = load_sd(build_ds2(g.cfg), g.sd()).eval()
    x0 = T(g["in/x"])
    reps = 6
    x = torch.cat([x0] * reps, dim=-1)
    total = x.shape[-1]
    n = x0.shape[0]
    lens = torch.clamp(total - torch.arange(n) * 11, min=total // 2).to(torch.int64)     # utterances end inside the last chunks
    replays = 0
    for chunk in (6, 12, 16, 24):
        eager = ChunkedDeepSpeech2(m, chunk, carry_context=True, use_graph=False)
        (ye, le), he = eager(x.clone(), lens)
        graph = ChunkedDeepSpeech2(m, chunk, carry_context=True, use_graph=True)
        for batch in range(2):                       # the second batch attaches to the first one's graph
            (yg, lg), hg = graph(x.clone(), lens)
            assert graph.graph_error is None, graph.graph_error
            assert torch.equal(yg, ye), f"chunk {chunk} batch {batch}: max diff {float((yg - ye).abs().max())}"
            assert torch.equal(lg, le)
            for a, b in zip(hg if isinstance(hg, tuple) else (hg,), he if isinstance(he, tuple) else (he,)):
                assert torch.equal(a, b)
        replays += graph.graph_replays
    assert replays > 0, "no chunk size reached a steady state: the graph path was not exercised"


@pytest.mark.parametrize("kt,st,chunk", [(1, 3, 5), (1, 3, 6), (2, 3, 5), (3, 2, 7), (5, 1, 4)])
def test_streaming_carried_context_graph_with_strides_wider_than_kernels(kt, st, chunk):
    """A time stride wider than the kernel leaves gaps between a convolution's windows: a push that STARTED inside a gap
    consumed fewer frames than the next push of the same size will, and must not be taken for the steady state a graph is
    captured from (tests/soak.py seed 94000234: kernel 1, stride 3, chunks of 5).  Graph replay == eager pushes == the oracle's
    full-utterance forward."""
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    cfg = dict(convs=[dict(kind="conv2d", idx=0, in_channels=1, out_channels=6, kernel=[3, kt], stride=[2, st], same=True,
                           act=(0.0, 20.0)),
                      dict(kind="conv2d", idx=2, in_channels=6, out_channels=7, kernel=[5, 1], stride=[1, 1], same=True,
                           act=(0.0, 20.0))],
               rnn=dict(kind=0, input=84, hidden=64, layers=1, bidirectional=False, forget_gate_bias=1.0),
               lookahead=dict(context=4, act=None),
               fc=dict(in_features=64, out_features=5, n_hidden=1, hidden=19, act=(0.0, 20.0)))
    torch.manual_seed(kt * 100 + st * 10 + chunk)
    m = build_ds2(cfg).eval()
    rng = np.random.default_rng(chunk)
    N, Tn = 3, 97
    x = rng.normal(size=(N, 1, 24, Tn)).astype(np.float32)
    lens = np.array([Tn, Tn - 9, Tn - 30])
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want, wl, _ = O.deep_speech_2_forward(x, lens, cfg, sd)
    (ye, le), _ = ChunkedDeepSpeech2(m, chunk, carry_context=True, use_graph=False)(T(x.copy()), T(lens))
    g = ChunkedDeepSpeech2(m, chunk, carry_context=True, use_graph=True)
    (yg, lg), _ = g(T(x.copy()), T(lens))
    assert g.graph_error is None, g.graph_error
    assert torch.equal(yg, ye) and torch.equal(lg, le)
    for n in range(N):
        np.testing.assert_allclose(cpu(yg)[:wl[n], n], want[:wl[n], n], rtol=2e-4, atol=2e-4)


def test_streaming_with_carried_context_shipped_architecture_vs_full_utterance_reference():
    """The reference's SHIPPED config shape (2 x conv2d, 3 x GRU-2560 unidirectional, lookahead 80, FC 1 x 1024;
    configs/deep_speech_2_en.config:19-93) streamed in 320 ms chunks with carried context against the reference's
    full-utterance run of ``tests/golden/ds2_shipped_summary.npz``: logits on the stored sub-grid and the final state within
    1e-3, greedy transcripts bit-exact; 174 input frames of algorithmic latency (1.58 s of it the lookahead's 79 frames)."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    from myrtlespeech_amd.model.lookahead import Lookahead
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    g = Golden("ds2_shipped_summary")

    def act():
        return SeqLenWrapper(torch.nn.Hardtanh(0.0, 20.0), torch.nn.Identity())
    torch.manual_seed(g.cfg["seed_weights"])
    cnn = torch.nn.Sequential(MaskConv2d(1, 32, [41, 11], [2, 2], PaddingMode.SAME), act(),
                              MaskConv2d(32, 32, [21, 11], [2, 1], PaddingMode.SAME), act())
    rnn = RNN(RNNType.GRU, 640, 2560, num_layers=3, bidirectional=False)
    la = torch.nn.Sequential(Lookahead(2560, 80), SeqLenWrapper(torch.nn.Identity(), torch.nn.Identity()))
    fc = FullyConnected(2560, 29, 1, 1024, torch.nn.Hardtanh(0.0, 20.0))
    model = DeepSpeech2(cnn, rnn, la, fc).eval()
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    N, Tn = g.cfg["N"], g.cfg["T"]
    x = torch.randn(N, 1, 80, Tn, generator=gen)
    lens = torch.sort(torch.randint(150, Tn + 1, (N,), generator=gen), descending=True).values
    lens[0] = Tn
    np.testing.assert_array_equal(lens.numpy(), g["in/lens"])
    st = ChunkedDeepSpeech2(model, 32, carry_context=True)
    assert st.latency_frames() == 174
    (y, ol), hn = st(x, lens)
    np.testing.assert_array_equal(cpu(ol), g["out/lens"])
    np.testing.assert_allclose(cpu(y[::10, ::2, :]), g["out/y_sub"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(cpu(hn[:, :, ::64]), g["out/hn_sub"], rtol=0, atol=1e-3)
    assert CTCGreedyDecoder(28)(y, ol) == unragged(g["out/greedy_flat"], g["out/greedy_lens"])
    err = float(np.abs(cpu(y[::10, ::2, :]) - g["out/y_sub"]).max())
    print(f"shipped architecture streamed with carried context: max |logit err| vs the full-utterance reference {err:.3e}")


# ----------------------------------------------------------------------------- CTC beam search
def test_beam_config_size_vs_reference():
    """The reference CTCBeamDecoder at the BASELINE decode size (T = 501, V = 29, beam 8, prune 1e-3; 4 ragged utterances
    of peaky posteriors, regenerated from the generator's seed): transcripts bit-exact, plain and with
    separator / word_weight; the oracle agrees as well."""
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    g = Golden("beam_cfg2")
    c = g.cfg
    torch.manual_seed(c["seed"])
    x = torch.softmax(torch.randn(c["T"], c["N"], c["V"]) * c["scale"], dim=2)
    assert abs(float(x.double().sum()) - float(g["in/x_abs_sum"])) < 1e-6 * c["T"] * c["N"]
    np.testing.assert_array_equal(x[::50, :, ::7].numpy(), g["in/x_probe"])
    lens = T(g["in/lens"])
    plain = CTCBeamDecoder(blank_index=28, beam_width=c["beam_width"], prune_threshold=c["prune"])(x, lens)
    assert plain == unragged(g["out/plain_flat"], g["out/plain_lens"])
    words = CTCBeamDecoder(blank_index=28, beam_width=c["beam_width"], prune_threshold=c["prune"],
                           separator_index=c["sep"], word_weight=c["word_weight"])(x, lens)
    assert words == unragged(g["out/words_flat"], g["out/words_lens"])
    assert sum(map(len, plain)) > 100   # the search survived (no float32 underflow to empty beams)


def test_beam_reference_kats():
    """tests/post_process/test_ctc_beam_decoder.py:17-102 (the reference's own known answers)."""
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    g = Golden("beam_kats")
    dec = CTCBeamDecoder(blank_index=1, beam_width=2, prune_threshold=0.0)
    assert dec(T(g["kat2x2/x"]), torch.tensor([2], dtype=torch.int8)) == [[0]]
    al = dict(zip("deouw_ ", range(7)))
    x, ln = T(g["katlm/x"]), torch.tensor([4], dtype=torch.int8)
    assert CTCBeamDecoder(blank_index=al["_"], beam_width=20)(x, ln) == [[al[c] for c in "do"]]
    for target in ("dew", "due"):
        tt = tuple(al[c] for c in target) + (al[" "],)
        dec = CTCBeamDecoder(blank_index=al["_"], beam_width=20, separator_index=al[" "],
                             language_model=lambda w, tt=tt: 2.0 if w == tt else 0.0, lm_weight=10.0, word_weight=2.0)
        assert dec(x, ln) == [[al[c] for c in target + " "]]


def test_beam_with_host_language_model_config_size_vs_oracle_and_the_reference_call_pattern():
    """The host-LM path at T = 501: 4 ragged utterances of peaky posteriors, beam 8, separator 0 with a
    language model and a word weight -- transcripts equal the oracle's (ctc_beam_decoder.py:175-258 restated), and the model is
    called exactly for the (frame, utterance, beam entry) triples the reference calls it for: those whose separator extension
    passes the float32 pruning test (ctc_beam_decoder.py:198, 214-230) -- counted by running the oracle with a counting model."""
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    torch.manual_seed(5)
    probs = torch.softmax(torch.randn(501, 4, 29) * 12, dim=2)
    lens = torch.tensor([501, 300, 120, 40], dtype=torch.int64)
    calls = {"n": 0}

    def lm(prefix):
        calls["n"] += 1
        return O.toy_language_model(prefix)
    dec = CTCBeamDecoder(28, 8, 0.001, language_model=lm, lm_weight=0.8, separator_index=0, word_weight=1.2)
    got = dec(T(probs.numpy()), lens)
    gpu_calls = calls["n"]
    calls["n"] = 0
    want = O.ctc_beam_decode(probs.numpy(), lens.numpy(), 28, 8, 0.001, language_model=lm, lm_weight=0.8, separator_index=0,
                             word_weight=1.2)
    assert got == want
    assert gpu_calls == calls["n"] == dec.lm_calls, (gpu_calls, calls["n"], dec.lm_calls)
    assert 0 < dec.lm_frames < 501            # the frames in between ran as runs, one launch each


def test_beam_golden_random():
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    g = Golden("beam_random")
    for c in g.cfg["cases"]:
        s = c["set"]
        key = f"{s}/out{c['idx']}"
        want = unragged(g[key + "_flat"], g[key + "_lens"])
        dec = CTCBeamDecoder(c["blank"], c["beam_width"], c["prune"],
                             language_model=O.toy_language_model if c["lm"] else None, lm_weight=c.get("lm_weight"),
                             separator_index=c["sep"], word_weight=c["word_weight"])
        assert dec(T(g[f"{s}/x"]), T(g[f"{s}/lens"])) == want, c


@pytest.mark.parametrize("Tn,N,V,W,thr,sep,temp", [(60, 6, 29, 8, 0.001, None, 6.0), (40, 4, 12, 20, 0.0, 3, 2.0),
                                                    (80, 3, 6, 3, 0.05, 0, 1.0), (30, 5, 40, 16, 0.001, None, 3.0),
                                                    # low temperature, no pruning: prefixes leave the beam and come back (their
                                                    # child rows are restored from the workspace)
                                                    (120, 4, 5, 4, 0.0, None, 0.7), (90, 3, 8, 6, 0.0, 2, 0.5),
                                                    # widths whose candidate tables do not fit the LDS (the round-4
                                                    # layout refused beam_width >= 84 at V = 29): 76 is the last LDS width, 100
                                                    # and 128 run with the working arrays in the workspace
                                                    (24, 3, 29, 76, 0.0, None, 1.5), (24, 3, 29, 100, 0.0, None, 1.5),
                                                    (20, 2, 29, 128, 0.0005, 7, 1.0), (12, 2, 8, 256, 0.0, None, 0.8),
                                                    # round 6: the constant-shape instantiations at 29 symbols x widths 4 / 16 / 32
                                                    (60, 4, 29, 4, 0.001, None, 6.0), (50, 3, 29, 16, 0.0, 5, 3.0),
                                                    (40, 3, 29, 32, 0.0005, None, 2.0), (70, 2, 29, 32, 0.0, 28 - 3, 0.6)])
def test_beam_vs_oracle(Tn, N, V, W, thr, sep, temp):
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    rng = np.random.default_rng(Tn * 31 + V)
    z = rng.normal(size=(Tn, N, V)) * temp
    x = np.exp(z - z.max(-1, keepdims=True))
    x = (x / x.sum(-1, keepdims=True)).astype(np.float32)
    lens = rng.integers(0, Tn + 1, size=N).astype(np.int64)
    lens[0] = Tn
    got = CTCBeamDecoder(V - 1, W, thr, separator_index=sep, word_weight=1.7)(T(x), T(lens))
    assert got == O.ctc_beam_decode(x, lens, V - 1, W, thr, separator_index=sep, word_weight=1.7)


def test_beam_call_split_between_frames_equals_one_call():
    """The state a call leaves in the workspace (beam, last frame's tables, the live beam's child rows, the trie) carries the
    search across calls: every split of the frame range gives the single call's beams -- the path the host language model
    uses one frame at a time (ctc_beam_decoder.py:222-228)."""
    from myrtlespeech_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    Tn, N, V, W = 50, 5, 9, 5
    z = rng.normal(size=(Tn, N, V)) * 0.8
    x = np.exp(z - z.max(-1, keepdims=True))
    x = (x / x.sum(-1, keepdims=True)).astype(np.float32)
    lens = np.array([50, 50, 37, 20, 0], dtype=np.int32)
    xd, ld = T(x).cuda(), T(lens).cuda()

    def run(cuts):
        ws = torch.zeros(lib.ms_ctc_beam_workspace_bytes(Tn, N, V, W), dtype=torch.uint8, device="cuda")
        oi = torch.zeros((N, Tn), dtype=torch.int32, device="cuda")
        ol = torch.zeros(N, dtype=torch.int32, device="cuda")
        bl = torch.zeros(N, dtype=torch.int32, device="cuda")
        bi = torch.zeros((N, W, Tn), dtype=torch.int32, device="cuda")
        bp = torch.zeros((N, W), dtype=torch.int32, device="cuda")
        edges = [0] + list(cuts) + [Tn]
        for k in range(len(edges) - 1):
            _lib.check(lib.ms_ctc_beam_decode(_lib.ptr(xd), _lib.ptr(ld), _lib.ptr(oi), _lib.ptr(ol), Tn, N, V, V - 1, W, 0.0, -1,
                                              None, edges[k], edges[k + 1], None, 1 if k == len(edges) - 2 else 0, _lib.ptr(bl),
                                              _lib.ptr(bi), _lib.ptr(bp), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "beam")
        return cpu(oi), cpu(ol), cpu(bl), cpu(bi), cpu(bp)

    whole = run([])
    want = O.ctc_beam_decode(x, lens.astype(np.int64), V - 1, W, 0.0)
    assert [list(whole[0][n, :whole[1][n]]) for n in range(N)] == want
    for cuts in ([1], [25], [49], [7, 8, 9, 30], list(range(1, Tn)), [0, 0, 20, 20]):
        got = run(cuts)
        for a, b in zip(whole[:3], got[:3]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(whole[4], got[4])
        for n in range(N):
            for w in range(int(whole[2][n])):
                np.testing.assert_array_equal(whole[3][n, w, :whole[4][n, w]], got[3][n, w, :got[4][n, w]])


def test_beam_one_output_per_batch_element_on_garbage():
    """tests/post_process/test_ctc_beam_decoder.py:105-114: uninitialised input, only the shape is pinned."""
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    x = torch.empty(7, 5, 4)
    x[0, 0, 0] = float("nan")
    out = CTCBeamDecoder(0, 3)(x, torch.tensor([7, 7, 3, 1, 0]))
    assert len(out) == 5 and all(isinstance(o, list) for o in out)


@pytest.mark.parametrize("M,K,N", [(300, 64, 200), (257, 2048, 129), (1000, 640, 8192)])
@pytest.mark.parametrize("act", [None, (0.0, 20.0)])
def test_linear_split_bf16x3(lib, M, K, N, act):
    """x.w as x_hi.w_hi + x_lo.w_hi + x_hi.w_lo: within 1e-5 of |x|.|w| per output."""
    from myrtlespeech_amd import _lib
    rng = np.random.default_rng(M + K + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32)
    xd, wd, bd = T(x).cuda(), T(w).cuda(), T(b).cuda()
    y = torch.empty((M, N), dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
    a, lo, hi = (0, 0.0, 0.0) if act is None else (1, act[0], act[1])
    _lib.check(lib.ms_linear_split_forward(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), M, K, N, a, lo, hi,
                                           _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "linear_split")
    want = x.astype(np.float64) @ w.T.astype(np.float64) + b
    if act is not None:
        want = np.clip(want, *act)
    bound = 1e-5 * (np.abs(x).astype(np.float64) @ np.abs(w.T).astype(np.float64)).max() + 1e-6
    assert np.abs(cpu(y) - want).max() <= bound


@pytest.mark.parametrize("M,K,N,act", [(512, 2560, 1024, (0.0, 20.0)),    # a chunk's hidden FC layer, shipped architecture, 32 streams
                                       (1024, 2048, 1024, (0.0, 20.0)),   # the same at configs[4] (64 streams)
                                       (1, 32, 1, None), (63, 64, 65, None), (130, 96, 200, (-0.5, 0.5)), (700, 640, 70, None),
                                       (16, 1024, 2048, None), (2048, 352, 128, None)])
def test_split_gemm_64x64_tile_kernel_is_bit_identical_to_the_256x128_kernel(lib, M, K, N, act, monkeypatch):
    """Small outputs run on 64 x 64 tiles (``gemm_nt_bf16x3_tile64_kernel``: a four-block register ring, two LDS stages)
    instead of 256 x 128 ones that leave most CUs without a workgroup: the same k-ordered sums, so ``torch.equal`` to the
    kernels it replaces (``MS_GEMM_TILE64=0``, read per call), ragged edges and the guard row included; and within the split's
    error of float64."""
    from myrtlespeech_amd import _lib
    rng = np.random.default_rng(M * 7 + K + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32)
    xd, wd, bd = T(x).cuda(), T(w).cuda(), T(b).cuda()
    ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
    a, lo, hi = (0, 0.0, 0.0) if act is None else (1, act[0], act[1])
    ys = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MS_GEMM_TILE64", flag)
        y = torch.full((M + 1, N), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(lib.ms_linear_split_forward(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), M, K, N, a, lo, hi,
                                               _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "linear_split")
        assert bool(torch.isnan(y[M]).all())
        ys.append(y[:M])
    assert torch.equal(ys[0], ys[1])
    want = x.astype(np.float64) @ w.T.astype(np.float64) + b
    if act is not None:
        want = np.clip(want, *act)
    bound = 1e-5 * (np.abs(x).astype(np.float64) @ np.abs(w.T).astype(np.float64)).max() + 1e-6
    assert np.abs(cpu(ys[1]) - want).max() <= bound


@pytest.mark.parametrize("M,K,N,act", [(2100, 96, 2052, None),        # ragged M edge, N % 4 == 0 but not a tile multiple
                                       (4100, 32, 1026, (0.0, 20.0)),  # one K-block; N % 4 != 0 -> scalar epilogue tail
                                       (16032, 640, 8192, None),       # the first projection of config 2
                                       (2048, 2048, 2048, (-1.0, 1.0))])
def test_split_gemm_lds_dma_kernel_vs_float64_and_register_staged_kernel(lib, M, K, N, act):
    """The shipped split-operand GEMM (kernel4: operands staged by LDS-DMA with a swizzled source address, double-buffered
    fragments, W as the MFMA's A operand) against float64 AND against the round-1 register-staged kernel2
    (``ms_gemm_set_variant(2)``): every output is the same k-ordered sum of the same products, so the two must agree bit
    for bit -- a swizzle, a DMA landing order or an epilogue index that is wrong anywhere shows up as a difference."""
    from myrtlespeech_amd import _lib
    rng = np.random.default_rng(M + K + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32)
    xd, wd, bd = T(x).cuda(), T(w).cuda(), T(b).cuda()
    ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
    a, lo, hi = (0, 0.0, 0.0) if act is None else (1, act[0], act[1])
    ys = []
    try:
        for variant in (0, 2):
            lib.ms_gemm_set_variant(variant)
            y = torch.full((M + 1, N), float("nan"), dtype=torch.float32, device="cuda")   # guard row: nothing past M is written
            _lib.check(lib.ms_linear_split_forward(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), M, K, N, a, lo, hi,
                                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "linear_split")
            assert bool(torch.isnan(y[M]).all())
            ys.append(y[:M])
    finally:
        lib.ms_gemm_set_variant(0)
    assert torch.equal(ys[0], ys[1])
    want = x.astype(np.float64) @ w.T.astype(np.float64) + b
    if act is not None:
        want = np.clip(want, *act)
    bound = 1e-5 * (np.abs(x).astype(np.float64) @ np.abs(w.T).astype(np.float64)).max() + 1e-6
    assert np.abs(cpu(ys[0]) - want).max() <= bound


@pytest.mark.parametrize("name", ["ctc_grad_small", "ctc_grad_v29"])
def test_ctc_loss_backward_matches_reference_autograd(name):
    """loss.backward() through the accelerated CTCLoss fills x.grad like the reference module (golden x.grad from the
    reference's LogSoftmax + torch.nn.CTCLoss under autograd): every reduction, per-utterance upstream weights,
    ragged inputs, repeated labels, an empty target, an infeasible utterance under zero_infinity."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    g = Golden(name)
    w = T(g["in/w"]).cuda()
    for key in [k for k in g.a if k.startswith("grad/")]:
        red, zi = key[len("grad/"):].rsplit("_", 1)
        x = T(g["in/x"]).cuda().requires_grad_(True)
        out = CTCLoss(blank=g.cfg["blank"], reduction=red, zero_infinity=bool(int(zi)))((x, T(g["in/x_lens"])),
                                                                                       (T(g["in/y"]), T(g["in/y_lens"])))
        ((out * w).sum() if red == "none" else out * g.cfg["scale"]).backward()
        np.testing.assert_allclose(cpu(x.grad), g[key], rtol=1e-4, atol=2e-5)


def test_ctc_loss_and_gradient_config_size_vs_reference():
    """loss/ctc_loss.py at the config-2 logits shape [501, 32, 29] (ragged input lengths, 60..120 labels), inputs
    regenerated from the generator's seed: per-utterance losses and both reductions against the reference, x.grad
    (reduction sum) on the stored sub-grid."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    g = Golden("ctc_cfg2")
    torch.manual_seed(g.cfg["seed"])
    x0 = torch.randn(501, 32, 29)
    xl = torch.sort(torch.randint(300, 502, (32,)), descending=True).values.to(torch.int32)
    np.testing.assert_array_equal(xl.numpy(), g["in/x_lens"])
    np.testing.assert_array_equal(x0[::100, ::8, ::7].numpy(), g["in/x_probe"])
    y, yl = T(g["in/y"]), T(g["in/y_lens"])
    for red in ("none", "mean", "sum"):
        got = CTCLoss(blank=28, reduction=red)((x0, xl), (y, yl))
        np.testing.assert_allclose(cpu(got), g[f"out/{red}"], rtol=2e-5, atol=1e-3)
    x = x0.clone().cuda().requires_grad_(True)
    CTCLoss(blank=28, reduction="sum")((x, xl), (y, yl)).backward()
    np.testing.assert_allclose(cpu(x.grad)[::25, ::4, :], g["grad/sum_sub"], rtol=2e-3, atol=1e-3)


def test_ctc_loss_backward_full_size_vs_oracle_rows_and_properties():
    """Config-2 logits shape [501, 32, 29], targets of 120: gradient rows of two utterances against the float64 oracle,
    and for all of them the CTC posterior identity: every valid frame's gradient sums to zero (softmax mass 1 minus
    occupation mass 1), padding frames are exactly zero."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    rng = np.random.default_rng(3)
    t_, n_, v_ = 501, 32, 29
    x_np = rng.normal(size=(t_, n_, v_)).astype(np.float32)
    xl = np.sort(rng.integers(300, t_ + 1, size=n_))[::-1].copy()
    y = rng.integers(0, 28, size=(n_, 120)).astype(np.int32)
    yl = rng.integers(60, 121, size=n_).astype(np.int32)
    x = T(x_np).cuda().requires_grad_(True)
    CTCLoss(blank=28, reduction="sum")((x, T(xl)), (T(y), T(yl))).backward()
    gr = cpu(x.grad)
    for n in range(n_):
        assert np.all(gr[xl[n]:, n] == 0)
        assert np.abs(gr[:xl[n], n].sum(axis=1)).max() < 2e-3  # float32 log-space sums of magnitude ~800 (ulp 6e-5)
    sel = [0, 17]
    want = O.ctc_grad(x_np[:, sel], xl[sel], y[sel], yl[sel], np.ones(2, np.float32), 28)
    np.testing.assert_allclose(gr[:, sel], want, rtol=2e-3, atol=1e-3)


# ----------------------------------------------------------------------------- protobuf builders
def test_builder_built_ds2_matches_reference_golden():
    """A text-format config -> builders -> modules; with the reference's weights loaded it must
    reproduce the reference's output (same state_dict keys, same arithmetic)."""
    from myrtlespeech_amd import protos as P
    from myrtlespeech_amd.builders.speech_to_text import build as build_stt
    from myrtlespeech_amd.wer import WordErrorRate, WordSegmentor
    g = Golden("ds2_tiny_bilstm")
    cfg = P.parse('''
    alphabet: " abcdefghi_";
    pre_process_step { stage: TRAIN_AND_EVAL; mfcc { n_mfcc: 16; win_length: 400; hop_length: 160; } }
    deep_speech_2 {
      conv_block { conv2d { output_channels: 4; kernel_feature: 5; kernel_time: 3; stride_feature: 2; stride_time: 2;
                            padding_mode: SAME; bias: true; } activation { hardtanh { min_val: 0.0; max_val: 20.0; } } }
      conv_block { conv2d { output_channels: 4; kernel_feature: 3; kernel_time: 3; stride_feature: 2; stride_time: 1;
                            padding_mode: SAME; bias: true; } activation { hardtanh { min_val: 0.0; max_val: 20.0; } } }
      rnn { rnn_type: LSTM; hidden_size: 16; num_layers: 2; bias: true; bidirectional: true; forget_gate_bias { value: 1.0 } }
      lookahead_block { no_lookahead {} activation { identity {} } }
      fully_connected { num_hidden_layers: 1; hidden_size: 24; activation { hardtanh { min_val: 0.0; max_val: 20.0; } } }
    }
    ctc_loss { blank_index: 10; reduction: SUM; }
    ctc_greedy_decoder { blank_index: 10; }
    ''', P.SpeechToText)
    stt = build_stt(cfg).eval()
    load_sd(stt.model, g.sd())
    (y, lens), _ = stt.model((T(g["in/x"]), T(g["in/lens"])))
    np.testing.assert_allclose(cpu(y), g["out/y"], **TOL)
    hyp = stt.post_process(y, lens)
    assert hyp == unragged(g["out/greedy_flat"], g["out/greedy_lens"])
    # loss + WER plumbing on the same outputs
    tgt = torch.tensor([[1, 2, 0, 3], [4, 0, 5, 0], [6, 0, 0, 0]], dtype=torch.int32)
    tl = torch.tensor([4, 3, 1], dtype=torch.int32)
    loss = stt.loss((y, lens), (tgt, tl))
    want = O.ctc_loss(g["out/y"], g["out/lens"], tgt.numpy(), tl.numpy(), 10, "sum")
    np.testing.assert_allclose(cpu(loss), want, rtol=1e-4, atol=1e-3)
    # WER of the decoded batch: run/run.py:84-109's arithmetic with the ORACLE's levenshtein (pinned to the reference's by
    # tests/golden/wer.npz, test_oracle_golden.py::test_wer_fixture_pins_levenshtein_alphabet_and_wer) over words cut by
    # str.split from the reference fixture's transcripts
    wer = WordErrorRate(stt.alphabet, WordSegmentor(" "))
    wer.update(hyp, tgt, tl)
    symbols = " abcdefghi_"
    want_hyp = [[w for w in "".join(symbols[i] for i in h).split(" ") if w] for h in unragged(g["out/greedy_flat"], g["out/greedy_lens"])]
    want_tgt = [[w for w in "".join(symbols[int(i)] for i in t[:int(n)]).split(" ") if w] for t, n in zip(tgt, tl)]
    dists = [O.levenshtein(a, b) for a, b in zip(want_hyp, want_tgt)]
    assert wer.transcripts == list(zip(want_hyp, want_tgt)) and wer.distances == dists
    assert wer.value() == float(sum(dists)) / sum(len(t) for t in want_tgt) * 100


# ----------------------------------------------------------------------------- chunked streaming (a16)
@pytest.mark.parametrize("name", golden_names("stream_"))
def test_chunked_streaming_matches_reference_hx_threading(name):
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    g = Golden(name)
    m = load_sd(build_ds2(g.cfg), g.sd())
    (y, lens), (hn, cn) = ChunkedDeepSpeech2(m, g.cfg["chunk_frames"])(T(g["in/x"]), T(g["in/lens"]))
    np.testing.assert_allclose(cpu(y), g["out/y"], **TOL)
    np.testing.assert_array_equal(cpu(lens), g["out/lens"])
    alive_last = g["out/hn_last"].shape[1]
    np.testing.assert_allclose(cpu(hn)[:, :alive_last], g["out/hn_last"], **TOL)
    np.testing.assert_allclose(cpu(cn)[:, :alive_last], g["out/cn_last"], **TOL)


def test_chunked_streaming_full_width_vs_reference_summary():
    """BASELINE configs[4] shape: the config-2 network (5 x BiLSTM-1024, bench.build_model's seed-0 weights, checksums
    pinned) on 32-frame chunks with the state carried, 4 ragged utterances -- against the reference run chunk by chunk
    with hx threaded (tests/golden/cfg5_stream_summary.npz): logits on the stored sub-grid and the last chunk's states."""
    import bench
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    g = Golden("cfg5_stream_summary")
    model = bench.build_model()
    for k, v in model.state_dict().items():
        assert abs(float(v.double().abs().sum()) - g.cfg["weight_abs_sums"][k]) <= 1e-6 * max(1.0, g.cfg["weight_abs_sums"][k]), k
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    x = torch.randn(g.cfg["N"], 1, 80, g.cfg["T"], generator=gen)
    (y, lens), (hn, cn) = ChunkedDeepSpeech2(model, g.cfg["chunk_frames"])(x, T(g["in/lens"]))
    np.testing.assert_array_equal(cpu(lens), g["out/lens"])
    np.testing.assert_allclose(cpu(y)[::3, :, ::2], g["out/y_sub"], rtol=0, atol=1e-3)
    alive_last = g["out/hn_last_sub"].shape[1]
    np.testing.assert_allclose(cpu(hn)[:, :alive_last, ::64], g["out/hn_last_sub"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(cpu(cn)[:, :alive_last, ::64], g["out/cn_last_sub"], rtol=0, atol=1e-3)


# ----------------------------------------------------------------------------- RNN-T (own spec, a15)
def _rnnt_parts(V=11, E=40, D=16, P=96, J=48, seed=3):
    from myrtlespeech_amd.model.rnnt import RNNTJoint, RNNTPredictor
    torch.manual_seed(seed)
    pred = RNNTPredictor(V, D, P, num_layers=2).eval()
    joint = RNNTJoint(E, P, J, V).eval()
    psd = {k: cpu(v) for k, v in pred.state_dict().items()}
    jsd = {k: cpu(v) for k, v in joint.state_dict().items()}
    return pred, joint, psd, jsd


def test_rnnt_kernels_vs_numpy(lib):
    from myrtlespeech_amd import _lib
    rng = np.random.default_rng(0)
    R, J, V1, C = 37, 200, 30, 240
    enc = rng.normal(size=(50, J)).astype(np.float32)
    rows = rng.integers(0, 50, size=R).astype(np.int32)
    pp = rng.normal(size=(R, J)).astype(np.float32)
    w = (rng.normal(size=(V1, J)) * 0.2).astype(np.float32)
    b = rng.normal(size=(V1,)).astype(np.float32)
    logp = torch.empty((R, V1), device="cuda")
    a = [T(v).cuda() for v in (enc, rows, pp, w, b)]
    _lib.check(lib.ms_rnnt_joint_forward(*[_lib.ptr(v) for v in a], _lib.ptr(logp), R, J, V1, _lib.stream_ptr()), "joint")
    want = O.log_softmax((np.tanh(enc[rows] + pp) @ w.T + b).astype(np.float32))
    np.testing.assert_allclose(cpu(logp), want, rtol=1e-4, atol=1e-4)
    sc = (rng.normal(size=(5, C)) * 3).round().astype(np.float32)  # many ties
    sc[2, 10:] = -np.inf
    idx = torch.empty((5, 8), dtype=torch.int32, device="cuda")
    val = torch.empty((5, 8), device="cuda")
    scd = T(sc).cuda()
    _lib.check(lib.ms_rnnt_topk(_lib.ptr(scd), _lib.ptr(idx), _lib.ptr(val), 5, C, 8, _lib.stream_ptr()), "topk")
    for r in range(5):
        order = np.argsort(-sc[r], kind="stable")[:8]
        np.testing.assert_array_equal(cpu(idx)[r], order)
        np.testing.assert_array_equal(cpu(val)[r], sc[r][order])


def test_rnnt_greedy_and_beam_vs_oracle():
    """Own-spec transducer decode (parity unpinned by the reference: it has none): HIP path vs oracle."""
    from myrtlespeech_amd.post_process.rnnt_decoder import RNNTBeamDecoder, RNNTGreedyDecoder
    from oracle import rnnt_oracle as RO
    V, E, P = 11, 40, 96
    pred, joint, psd, jsd = _rnnt_parts(V=V, E=E, P=P)
    rng = np.random.default_rng(1)
    enc = (rng.normal(size=(14, 3, E)) * 1.5).astype(np.float32)
    lens = np.array([14, 9, 4])
    got = RNNTGreedyDecoder(pred, joint, max_symbols=3)(T(enc), T(lens))
    assert got == RO.greedy_decode(enc, lens, psd, jsd, P, 2, V, 3)
    dec = RNNTBeamDecoder(pred, joint, beam_width=4, max_symbols=3)
    got = dec(T(enc), T(lens))
    want, want_scores = RO.beam_decode(enc, lens, psd, jsd, P, 2, V, 4, 3)
    assert got == want
    np.testing.assert_allclose(dec.last_scores, want_scores, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("ms,blank_bias,seed", [(1, 0.0, 0), (3, 0.5, 2), (4, 0.75, 3), (2, 1.0, 4), (3, 1.5, 5), (3, 30.0, 6)])
def test_rnnt_greedy_event_driven_decode_vs_oracle(ms, blank_bias, seed):
    """The greedy decode evaluates the joint for 32 consecutive frames of every utterance against the current prediction and
    steps the predictor only after a label (round 3).  Sequences longer than several chunks, blank runs that span chunk
    boundaries (raised blank logit; +30 = no label at all), labels on consecutive frames, the per-frame quota ``max_symbols``
    and utterances that end inside a chunk, an empty one included: transcripts equal to the oracle's frame-by-frame loop."""
    from myrtlespeech_amd.post_process.rnnt_decoder import RNNTGreedyDecoder
    from oracle import rnnt_oracle as RO
    V, E, P = 9, 24, 64
    pred, joint, psd, jsd = _rnnt_parts(V=V, E=E, D=8, P=P, J=32, seed=10 + seed)
    with torch.no_grad():
        joint.out.bias[V] += blank_bias
    jsd = {k: cpu(v) for k, v in joint.state_dict().items()}
    rng = np.random.default_rng(seed)
    steps, N = 150, 5
    enc = (rng.normal(size=(steps, N, E)) * 2.0).astype(np.float32)
    lens = np.array([150, 97, 64, 33, 0])
    want = RO.greedy_decode(enc, lens, psd, jsd, P, 2, V, ms)
    dec = RNNTGreedyDecoder(pred, joint, max_symbols=ms)
    got = dec(T(enc), T(lens))
    assert got == want
    # (dense transcripts at bias 0 .. 0.75: 1 .. 3 labels per frame; 52 labels over 150 frames at 1.0; 0 .. 4 at 1.5; none at 30)
    assert sum(map(len, want)) > 0 if blank_bias < 30.0 else all(h == [] for h in want)
    assert dec(T(enc), T(lens)) == got                      # a second call re-uses the workspace


@pytest.mark.parametrize("V,w,ms,N,steps,seed", [(4, 8, 3, 4, 24, 0), (11, 1, 2, 2, 12, 1), (6, 5, 4, 3, 10, 2),
                                                 (28, 8, 3, 2, 8, 3), (3, 16, 2, 2, 16, 4),
                                                 # vocabularies around the re-cut sequence's LDS gate (V + 1 <= 182): the last
                                                 # one it serves, the first one that takes the round-4 sequence, and a BPE-sized one
                                                 (181, 4, 2, 2, 8, 5), (182, 4, 2, 2, 8, 6), (255, 4, 2, 2, 6, 7)])
def test_rnnt_device_decode_sweep(V, w, ms, N, steps, seed):
    """Small vocabularies force same-prefix merges of blank transitions (the trie / logaddexp path); ragged lengths
    including an empty utterance; every width / round count against the oracle, greedy as well."""
    from myrtlespeech_amd.post_process.rnnt_decoder import RNNTBeamDecoder, RNNTGreedyDecoder
    from oracle import rnnt_oracle as RO
    E, P = 24, 64
    pred, joint, psd, jsd = _rnnt_parts(V=V, E=E, D=8, P=P, J=32, seed=seed)
    rng = np.random.default_rng(seed)
    enc = (rng.normal(size=(steps, N, E)) * 2.0).astype(np.float32)
    lens = np.sort(rng.integers(1, steps + 1, size=N))[::-1].copy()
    lens[0] = steps
    lens[-1] = 0 if seed % 2 == 0 else lens[-1]
    dec = RNNTBeamDecoder(pred, joint, beam_width=w, max_symbols=ms)
    got = dec(T(enc), T(lens))
    want, want_scores = RO.beam_decode(enc, lens, psd, jsd, P, 2, V, w, ms)
    assert got == want
    np.testing.assert_allclose(dec.last_scores, want_scores, rtol=1e-4, atol=1e-4)
    assert RNNTGreedyDecoder(pred, joint, max_symbols=ms)(T(enc), T(lens)) == RO.greedy_decode(enc, lens, psd, jsd, P, 2, V, ms)


def test_rnnt_decode_validation():
    from myrtlespeech_amd.post_process.rnnt_decoder import RNNTBeamDecoder, RNNTGreedyDecoder
    pred, joint, _, _ = _rnnt_parts()
    with pytest.raises(ValueError):
        RNNTBeamDecoder(pred, joint, beam_width=0)
    with pytest.raises(ValueError):
        RNNTGreedyDecoder(pred, joint, max_symbols=0)
    dec = RNNTBeamDecoder(pred, joint, beam_width=2)
    with pytest.raises(ValueError):
        dec(torch.randn(5, 2, 40), torch.tensor([5, 6]))
    with pytest.raises(ValueError):
        dec(torch.randn(5, 2, 40), torch.tensor([5]))
    assert dec(torch.randn(5, 2, 40), torch.tensor([0, 0])) == [[], []]


def test_exact_f32_mode_in_subprocess():
    """MS_PRECISION=f32 is read once per process, so the exact-f32 kernels are exercised in one child process: the
    two-stream register-resident kernel on each of its shapes (H = 1024 / 768 / 512 / 256; ragged, 2 layers, batch groups
    of 32 + remainder, 37 steps through both packet slots, a given initial state, the hard cell) and the one-stream
    kernel on a shape outside that set (H = 96)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from myrtlespeech_amd.model.rnn import RNN, RNNType
from oracle import ds_oracle as O
from myrtlespeech_amd.model.hard_lstm import HardLSTM
for H, N, T_, bidir, with_hx in ((1024, 32, 5, True, False), (256, 40, 6, True, False), (512, 33, 9, False, True),
                                 (768, 16, 37, True, True), (96, 7, 5, True, False)):
    torch.manual_seed(H)
    m = RNN(RNNType.LSTM, 64, H, num_layers=2, bidirectional=bidir, forget_gate_bias=1.0).eval()
    rng = np.random.default_rng(H)
    lens = np.sort(rng.integers(1, T_ + 1, size=N))[::-1].copy(); lens[0] = T_
    x = rng.normal(size=(T_, N, 64)).astype(np.float32)
    D = 2 if bidir else 1
    hx = None
    if with_hx:
        hx = tuple((rng.normal(size=(2 * D, N, H)) * 0.5).astype(np.float32) for _ in range(2))
    (out, _), (hn, cn) = m((torch.from_numpy(x), torch.from_numpy(lens)),
                           None if hx is None else tuple(torch.from_numpy(a) for a in hx))
    sd = {k[4:]: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want, (whn, wcn) = O.rnn_forward(O.LSTM, x, lens, sd, H, 2, bidir, hx)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(hn.cpu().numpy(), whn, rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(cn.cpu().numpy(), wcn, rtol=1e-5, atol=3e-6)
    if hx is None and (lens < T_).any():
        # an utterance shard (shorter longest sequence => other launch step indices) reproduces the whole batch bit for bit
        k0 = int(np.argmax(lens < T_)); t0 = int(lens[k0])
        (out2, _), _ = m((torch.from_numpy(x[:t0, k0:].copy()), torch.from_numpy(lens[k0:].copy())))
        assert np.array_equal(out.cpu().numpy()[:t0, k0:], out2.cpu().numpy()), ("shard", H)
torch.manual_seed(9)
m = HardLSTM(20, 256, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
x = (np.random.default_rng(9).normal(size=(11, 35, 20)) * 2).astype(np.float32)
(out, _), (hn, cn) = m((torch.from_numpy(x), torch.tensor([11] * 35)))
sd = {k[len("rnn."):]: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
want, (whn, wcn) = O.hard_lstm_forward(x, sd, 256, 1, True)
np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=3e-6)
np.testing.assert_allclose(cn.cpu().numpy(), wcn, rtol=1e-5, atol=3e-6)
print("f32 mode ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MS_PRECISION="f32")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "f32 mode ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("cin,cout,k,s,same,F,Tn,N", [(32, 32, [21, 11], [2, 1], True, 40, 140, 2),
                                                      (16, 40, [3, 5], [1, 2], False, 9, 300, 3),
                                                      (48, 33, [5, 4], [2, 1], True, 12, 131, 2)])
def test_conv2d_channels_last_split_vs_oracle(cin, cout, k, s, same, F, Tn, N):
    """Multi-channel convolutions take the split-bf16 channels-last kernel (conv_cl.hip)."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    torch.manual_seed(cin + cout)
    m = MaskConv2d(cin, cout, k, s, PaddingMode.SAME if same else PaddingMode.NONE).eval()
    rng = np.random.default_rng(cin)
    x = rng.normal(size=(N, cin, F, Tn)).astype(np.float32)
    lens = np.sort(rng.integers(Tn // 2, Tn + 1, size=N))[::-1].copy()
    y, nl = m((T(x), T(lens)), fused_activation=(0.0, 20.0))
    want, wl = O.mask_conv2d(x, lens, cpu(m.weight), cpu(m.bias), tuple(s), same)
    want = np.clip(want, 0.0, 20.0)
    np.testing.assert_allclose(cpu(y), want, rtol=1e-4, atol=2e-4)
    np.testing.assert_array_equal(cpu(nl), wl)


@pytest.mark.parametrize("cout,k,s,same,F,Tn,N", [(32, [41, 11], [2, 2], True, 80, 700, 3),     # DS2 conv1
                                                  (40, [21, 5], [2, 1], False, 64, 700, 8),    # two cout tiles, no padding
                                                  (32, [16, 11], [4, 2], True, 30, 2800, 8)])  # window = one k-step
def test_conv2d_single_channel_feature_window_vs_oracle(cout, k, s, same, F, Tn, N):
    """Single-channel convolutions with a tall filter take the feature-window form of the split-bf16 kernel
    (conv_cl.hip, ms_maskconv_fwin_*): ragged lengths, SAME and no padding, bias + clamp epilogue."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    torch.manual_seed(cout + k[0])
    m = MaskConv2d(1, cout, k, s, PaddingMode.SAME if same else PaddingMode.NONE).eval()
    rng = np.random.default_rng(cout)
    x = rng.normal(size=(N, 1, F, Tn)).astype(np.float32)
    lens = np.sort(rng.integers(Tn // 2, Tn + 1, size=N))[::-1].copy()
    y, nl = m((T(x), T(lens)), fused_activation=(0.0, 20.0))
    assert 2.0 * y.numel() * k[0] * k[1] >= 1e9   # large enough to be routed to the MFMA path
    want, wl = O.mask_conv2d(x, lens, cpu(m.weight), cpu(m.bias), tuple(s), same)
    want = np.clip(want, 0.0, 20.0)
    np.testing.assert_allclose(cpu(y), want, rtol=1e-4, atol=2e-4)
    np.testing.assert_array_equal(cpu(nl), wl)


def test_conv_mfma_paths_small_shape_sweep(monkeypatch):
    """Both split-bf16 convolution kernels over a grid of small shapes (MS_CONV_MFMA_MIN_FLOPS=0 routes them there):
    every (taps, stride, padding, parity of T) combination has its own staging extent, tile tail and SAME offsets.
    504 cases; written after a randomized soak case exposed a staging fault in an interim build of this kernel."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    monkeypatch.setenv("MS_CONV_MFMA_MIN_FLOPS", "0")
    rng = np.random.default_rng(0)
    cases = []
    for cin in (16, 32):                                    # channels-last kernel
        for kt in range(1, 8):
            for st in (1, 2, 3):
                for same in (True, False):
                    for tn in (300, 301):
                        cases.append((cin, [1 + (kt % 2), kt], [1, st], same, 4, tn, 1))
    for kf in (16, 41):                                     # feature-window kernel
        for kt in (1, 3, 4, 7, 11):
            for sf in (2, 4):
                for st in (1, 2, 3):
                    for same in (True, False):
                        for tn, F, N in ((300, 50, 1), (301, 81, 2)):
                            cases.append((1, [kf, kt], [sf, st], same, F, tn, N))
    for cin in (16, 32):                                    # short inputs: the 32-frame x 8-row tile of both kernels
        for kt in (1, 3, 5, 11):
            for st in (1, 2):
                for same in (True, False):
                    for tn in (17, 33):
                        cases.append((cin, [3, kt], [1, st], same, 19, tn, 2))
    for kf in (16, 41):
        for kt in (3, 11):
            for st in (1, 2):
                for same in (True, False):
                    for tn, F, N in ((32, 80, 3), (19, 50, 2)):
                        cases.append((1, [kf, kt], [2, st], same, F, tn, N))
    for cin, k, s, same, F, tn, N in cases:
        torch.manual_seed(1)
        m = MaskConv2d(cin, 32, k, s, PaddingMode.SAME if same else PaddingMode.NONE).eval()
        x = rng.normal(size=(N, cin, F, tn)).astype(np.float32)
        lens = np.array([tn, max(tn - 37, k[1]), max(tn - 5, k[1])][:N])
        lens = np.sort(lens)[::-1].copy()
        y, nl = m((T(x), T(lens)))
        want, wl = O.mask_conv2d(x, lens, cpu(m.weight), cpu(m.bias), tuple(s), same)
        np.testing.assert_allclose(cpu(y), want, rtol=1e-4, atol=3e-4, err_msg=str((cin, k, s, same, F, tn, N)))
        np.testing.assert_array_equal(cpu(nl), wl)
    assert len(cases) == 408 + 64 + 32


def conv_cl_lanemask_trigger_shapes():
    """(cin, kt, st, dt, tile) for which the round-1 interim build of maskconv_cl_kernel staged zeros (root cause, from its
    ISA: the runtime feature-window flag was compared by the VALU into an SGPR lane mask inside the FIRST output row's
    staging loop, under that loop's shrinking EXEC, and re-used in the SECOND row's loop; tools/micro/convflag/).  The mask
    held only the lanes that were active in a wave's LAST staging iteration, tid < m = KG * PW mod 256 (0 < m < 64); if those
    lanes are all left-padding frames at the start of the second row (m <= KG * pad_left) and other lanes are not, the wave
    took the wrong load path.  tile = output frames per workgroup (128, or 32 for short inputs)."""
    from myrtlespeech_amd.model.cnn import pad_same
    out = []
    for tile in (128, 32):
        for cin in (16, 32, 48, 64):
            kg = cin // 8
            for kt in range(1, 12):
                for st in (1, 2, 3):
                    for dt in (1, 2, 3):
                        pw = (tile - 1) * st + (kt - 1) * dt + 1
                        m = kg * pw % 256
                        pad_left = pad_same(300, kt, st, dt)[0]
                        rows = 2 if tile == 128 else 8
                        lds = 2 * rows * kg * pw * 16 + 2 * ((kt + 1) // 2) * kg * 32 * 16
                        if 0 < m < 64 and m <= kg * pad_left < 64 and lds <= 160 * 1024:
                            out.append((cin, kt, st, dt, tile))
    return out


def test_conv_cl_second_row_staging_when_the_last_iteration_is_partial(monkeypatch):
    """Regression for the staging fault of the interim runtime-flag build (VERDICT r1 item 3 / ADVICE r1): every (channels,
    taps, stride, dilation) whose staging extent leaves wave 0 a partial last iteration made only of frames that are SAME
    padding at the start of the next row -- the exact condition under which that build staged zeros -- on both tile shapes,
    against the oracle.  Kernel 3 / stride 2 / SAME at 16 channels is the smallest member."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    monkeypatch.setenv("MS_CONV_MFMA_MIN_FLOPS", "0")
    shapes = conv_cl_lanemask_trigger_shapes()
    assert (16, 3, 2, 1, 128) in shapes and len(shapes) >= 20
    rng = np.random.default_rng(5)
    for cin, kt, st, dt, tile in shapes:
        tn = 300 if tile == 128 else 40           # Tout > 48 selects the 128-frame tile, Tout <= 48 the 32-frame one
        if tile == 32 and (tn + st - 1) // st > 48:
            continue
        torch.manual_seed(2)
        m = MaskConv2d(cin, 32, [3, kt], [1, st], PaddingMode.SAME, dilation=[1, dt]).eval()
        x = (rng.normal(size=(2, cin, 5, tn)) + 2.0).astype(np.float32)     # offset: a zero granule cannot hide
        lens = np.array([tn, tn - 7])
        y, nl = m((T(x), T(lens)))
        want, wl = O.mask_conv2d(x, lens, cpu(m.weight), cpu(m.bias), (1, st), True, dilation=(1, dt))
        np.testing.assert_allclose(cpu(y), want, rtol=1e-4, atol=1e-3, err_msg=str((cin, kt, st, dt, tile)))
        np.testing.assert_array_equal(cpu(nl), wl)


def test_conv_cl_short_input_kernel_vs_oracle(monkeypatch):
    """``maskconv_cl_short_kernel`` (round 4: 32 input channels, at most 16 output frames, time stride 1 -- a streaming chunk at
    DS2's second convolution): 16-frame x 16-cout MFMA tiles, every input row of a block staged once, filters streamed from L2.
    Kernel heights / widths, feature strides and dilations, SAME and no padding, output channels that are not a multiple of
    16, ragged lengths (the mask, cnn.py:425-443), 1 .. 70 utterances (different row-block plans), against the oracle; the
    tiled kernel (MS_CONV_SHORT=0 is read once per process, so it is compared through the oracle, not in-process)."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    monkeypatch.setenv("MS_CONV_MFMA_MIN_FLOPS", "0")
    rng = np.random.default_rng(11)
    cases = [  # (cout, kf, kt, sf, df, same, F, T, N)
        (32, 21, 11, 2, 1, True, 40, 16, 64), (32, 21, 11, 2, 1, True, 40, 16, 3), (16, 3, 3, 1, 1, True, 7, 16, 5),
        (29, 5, 4, 2, 1, True, 23, 13, 2), (48, 4, 5, 1, 2, True, 12, 9, 6), (40, 7, 1, 3, 1, False, 50, 16, 1),
        (32, 1, 7, 1, 1, False, 5, 22, 4), (33, 9, 2, 2, 2, False, 41, 10, 70), (32, 21, 11, 2, 1, True, 40, 1, 2),
        # ONE feature row, feature stride 3, SAME: the reference's padding puts a zero row in front of the only real one
        # (cnn.py:148-163), so every output is the bias -- this shape used to be routed to the conv1d lowering (soak 81000307)
        (33, 1, 5, 3, 2, True, 1, 14, 7), (16, 1, 3, 2, 1, True, 1, 9, 3),
    ]
    for cout, kf, kt, sf, df, same, F, Tn, N in cases:
        torch.manual_seed(3)
        m = MaskConv2d(32, cout, [kf, kt], [sf, 1], PaddingMode.SAME if same else PaddingMode.NONE, dilation=[df, 1]).eval()
        x = (rng.normal(size=(N, 32, F, Tn)) + 1.0).astype(np.float32)
        lens = np.sort(rng.integers(1, Tn + 1, size=N))[::-1].copy()
        lens[0] = Tn
        y, nl = m((T(x), T(lens)), fused_activation=(0.0, 20.0))
        want, wl = O.mask_conv2d(x, lens, cpu(m.weight), cpu(m.bias), (sf, 1), same, dilation=(df, 1))
        want = np.clip(want, 0.0, 20.0)
        assert y.shape[-1] <= 16
        np.testing.assert_allclose(cpu(y), want, rtol=1e-4, atol=2e-3, err_msg=str((cout, kf, kt, sf, df, same, F, Tn, N)))
        np.testing.assert_array_equal(cpu(nl), wl)


@pytest.mark.parametrize("H,bidir,N,Tn", [(256, True, 37, 9), (512, False, 5, 12), (1024, True, 32, 6), (1280, True, 20, 5),
                                          (2048, False, 32, 4)])
def test_lstm_stack_plane_chaining_is_bit_identical_to_single_layers(H, bidir, N, Tn):
    """A two-stream LSTM stack hands each layer's output to the next layer as GEMM operand planes inside the workspace
    (MS_RNN_OUT_PLANES_TO_WS / MS_RNN_X_PLANES_IN_WS); the same weights run as single-layer modules (float32 hand-over,
    split afterwards) must give the same bits."""
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(H)
    stack = RNN(RNNType.LSTM, 64, H, num_layers=3, bidirectional=bidir, forget_gate_bias=1.0).eval()
    D = 2 if bidir else 1
    singles = []
    for layer in range(3):
        m = RNN(RNNType.LSTM, 64 if layer == 0 else D * H, H, num_layers=1, bidirectional=bidir).eval()
        sd = {}
        for k, v in stack.state_dict().items():
            if f"_l{layer}" in k:
                sd[k.replace(f"_l{layer}", "_l0")] = v
        m.load_state_dict(sd)
        singles.append(m)
    rng = np.random.default_rng(H)
    x = T(rng.normal(size=(Tn, N, 64)).astype(np.float32))
    lens = np.sort(rng.integers(1, Tn + 1, size=N))[::-1].copy()
    lens[0] = Tn
    lens = T(lens)
    (out, _), (hn, cn) = stack((x, lens))
    h = x
    hs, cs = [], []
    for m in singles:
        (h, _), (a, b) = m((h, lens))
        hs.append(a)
        cs.append(b)
    assert torch.equal(out, h)
    assert torch.equal(hn, torch.cat(hs, 0)) and torch.equal(cn, torch.cat(cs, 0))


def test_gru_stack_plane_chaining_is_bit_identical_to_single_layers():
    """The same hand-over in the persistent GRU (H = 1280, the half-width sibling of the shipped config's GRU-2560)."""
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    H, N, Tn = 1280, 9, 7
    torch.manual_seed(3)
    stack = RNN(RNNType.GRU, 64, H, num_layers=2, bidirectional=False).eval()
    singles = []
    for layer in range(2):
        m = RNN(RNNType.GRU, 64 if layer == 0 else H, H, num_layers=1, bidirectional=False).eval()
        m.load_state_dict({k.replace(f"_l{layer}", "_l0"): v for k, v in stack.state_dict().items() if f"_l{layer}" in k})
        singles.append(m)
    rng = np.random.default_rng(3)
    x = T(rng.normal(size=(Tn, N, 64)).astype(np.float32))
    lens = np.sort(rng.integers(1, Tn + 1, size=N))[::-1].copy()
    lens[0] = Tn
    lens = T(lens)
    (out, _), hn = stack((x, lens))
    h, hs = x, []
    for m in singles:
        (h, _), a = m((h, lens))
        hs.append(a)
    assert torch.equal(out, h) and torch.equal(hn, torch.cat(hs, 0))


def test_rnn_status_word_is_sticky_and_reported_once():
    """ms_rnn_status reads the sticky time-out word at the head of the workspace (set by the kernels, never cleared by a
    layer call), reports it once and clears it; a layer call in between leaves it alone."""
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    lib = _lib.load()
    m = RNN(RNNType.LSTM, 16, 64, num_layers=2, bidirectional=True).eval()
    m.check_status = False
    x, lens = torch.randn(5, 3, 16), torch.tensor([5, 4, 2])
    m((x, lens))
    ws = m._workspace.buf
    assert lib.ms_rnn_status(_lib.ptr(ws), _lib.stream_ptr()) == 0
    ws[:4] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8)           # what a timed-out kernel leaves behind
    m((x, lens))                                                     # later layer calls do not clear it
    assert lib.ms_rnn_status(_lib.ptr(ws), _lib.stream_ptr()) == 4   # MS_ERR_TIMEOUT
    assert lib.ms_rnn_status(_lib.ptr(ws), _lib.stream_ptr()) == 0   # reported once
    m.check_status = True
    ws[:4] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="TIMEOUT"):
        m((x, lens))


# ----------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("kind,H,bidir", [(0, 1024, True), (0, 64, True), (0, 48, False), (1, 256, True), (2, 200, True)])
def test_rnn_all_lengths_shorter_than_the_buffer(kind, H, bidir):
    """total_length > max(lengths) (rnn.py:179-183): frames past every sequence are zero, on the
    two-stream / one-stream / exact-f32 persistent kernels and on the generic GRU / tanh path."""
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(kind * 10 + H)
    In, N, Tn = 24, 3, 9
    m = RNN(RNNType(kind), In, H, num_layers=1, bidirectional=bidir, forget_gate_bias=1.0 if kind == 0 else None).eval()
    rng = np.random.default_rng(H)
    x = rng.normal(size=(Tn, N, In)).astype(np.float32)
    lens = np.array([6, 6, 1])
    (out, _), hid = m((T(x), T(lens)))
    sd = {k[len("rnn."):]: cpu(v) for k, v in m.state_dict().items()}
    want, whid = O.rnn_forward(kind, x, lens, sd, H, 1, bidir)
    np.testing.assert_allclose(cpu(out), want, **TOL)
    assert float(cpu(out)[6:].__abs__().max()) == 0.0
    hn = hid[0] if kind == 0 else hid
    np.testing.assert_allclose(cpu(hn), whid[0] if kind == 0 else whid, **TOL)


def test_single_frame_single_utterance():
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    torch.manual_seed(1)
    m = RNN(RNNType.LSTM, 8, 256, bidirectional=True).eval()
    x = np.random.default_rng(1).normal(size=(1, 1, 8)).astype(np.float32)
    (out, _), _ = m((T(x), torch.tensor([1])))
    sd = {k[len("rnn."):]: cpu(v) for k, v in m.state_dict().items()}
    want, _ = O.rnn_forward(O.LSTM, x, np.array([1]), sd, 256, 1, True)
    np.testing.assert_allclose(cpu(out), want, **TOL)
    p = torch.tensor([[[0.2, 0.5, 0.3]]])
    assert CTCGreedyDecoder(2)(p, torch.tensor([1])) == [[1]]
    assert CTCBeamDecoder(2, 4)(p, torch.tensor([1])) == O.ctc_beam_decode(p.numpy(), np.array([1]), 2, 4)


def test_greedy_and_beam_max_sizes():
    """Config-2 decode shapes (T=501, N=32, V=29) and a wide beam, against the oracle on a slice."""
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    rng = np.random.default_rng(77)
    z = rng.normal(size=(501, 32, 29)) * 7
    x = np.exp(z - z.max(-1, keepdims=True))
    x = (x / x.sum(-1, keepdims=True)).astype(np.float32)
    lens = np.sort(rng.integers(100, 502, size=32))[::-1].astype(np.int64)
    lens[0] = 501
    got = CTCBeamDecoder(28, 8)(T(x), T(lens))
    sel = [0, 13, 31]
    want = O.ctc_beam_decode(x[:, sel], lens[sel], 28, 8)
    assert [got[i] for i in sel] == want
    got = CTCBeamDecoder(28, 64, 0.0)(T(x[:60, :2]), torch.tensor([60, 41]))
    assert got == O.ctc_beam_decode(x[:60, :2], np.array([60, 41]), 28, 64, 0.0)


def test_fp16_mode_in_subprocess():
    """MS_PRECISION=fp16 (optional fast mode for chunked streaming, BASELINE.json configs[4]): single-pass
    fp16 operands in the two-stream LSTM, the projection GEMM, large Linears and the channels-last conv.
    Checked at fp16-level tolerances against the oracle in one child process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from myrtlespeech_amd.model.rnn import RNN, RNNType
from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
from oracle import ds_oracle as O
for H, N, T_ in ((1024, 32, 6), (256, 20, 7)):
    torch.manual_seed(H)
    m = RNN(RNNType.LSTM, 64, H, num_layers=2, bidirectional=True, forget_gate_bias=1.0).eval()
    rng = np.random.default_rng(H)
    lens = np.sort(rng.integers(1, T_ + 1, size=N))[::-1].copy(); lens[0] = T_
    x = rng.normal(size=(T_, N, 64)).astype(np.float32)
    (out, _), (hn, cn) = m((torch.from_numpy(x), torch.from_numpy(lens)))
    sd = {k[4:]: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want, (whn, wcn) = O.rnn_forward(O.LSTM, x, lens, sd, H, 2, True)
    err = float(np.abs(out.cpu().numpy() - want).max())
    assert err < 3e-3, err
    assert float(np.abs(out.cpu().numpy()[int(lens[-1]):, -1]).max() if int(lens[-1]) < T_ else 0.0) == 0.0
    print("lstm fp16 err", H, err)
torch.manual_seed(1)
c = MaskConv2d(32, 32, [5, 5], [2, 1], PaddingMode.SAME).eval()
x = np.random.default_rng(2).normal(size=(2, 32, 12, 300)).astype(np.float32)
y, _ = c((torch.from_numpy(x), torch.tensor([300, 200])), fused_activation=(0.0, 20.0))
want, _ = O.mask_conv2d(x, np.array([300, 200]), c.weight.detach().cpu().numpy(), c.bias.detach().cpu().numpy(), (2, 1), True)
err = float(np.abs(y.cpu().numpy() - np.clip(want, 0, 20)).max())
assert err < 2e-2, err
print("conv fp16 err", err)
print("fp16 mode ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MS_PRECISION="fp16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fp16 mode ok" in r.stdout, r.stdout + r.stderr


# ----------------------------------------------------------------------------- randomized sweeps
def test_random_conv_configs_vs_oracle():
    """20 random MaskConv1d/2d configurations (channels, kernels, strides, padding mode, ragged lengths)."""
    from myrtlespeech_amd.model.cnn import MaskConv1d, MaskConv2d, PaddingMode
    rng = np.random.default_rng(2024)
    for case in range(20):
        two_d = bool(rng.integers(0, 2))
        cin, cout = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        same = bool(rng.integers(0, 2))
        N, Tn = int(rng.integers(1, 4)), int(rng.integers(12, 200))
        lens = np.sort(rng.integers(1, Tn + 1, size=N))[::-1].copy()
        torch.manual_seed(case)
        if two_d:
            k = [int(rng.integers(1, 6)), int(rng.integers(1, 8))]
            s = [int(rng.integers(1, 3)), int(rng.integers(1, 4))]
            F = int(rng.integers(k[0], 20))
            if not same and Tn < k[1]:
                continue
            m = MaskConv2d(cin, cout, k, s, PaddingMode.SAME if same else PaddingMode.NONE).eval()
            x = rng.normal(size=(N, cin, F, Tn)).astype(np.float32)
            want, wl = O.mask_conv2d(x, lens, cpu(m.weight), cpu(m.bias), tuple(s), same)
        else:
            k, s = int(rng.integers(1, 9)), int(rng.integers(1, 4))
            if not same and Tn < k:
                continue
            m = MaskConv1d(cin, cout, k, s, PaddingMode.SAME if same else PaddingMode.NONE).eval()
            x = rng.normal(size=(N, cin, Tn)).astype(np.float32)
            want, wl = O.mask_conv1d(x, lens, cpu(m.weight), cpu(m.bias), s, same)
        y, nl = m((T(x), T(lens)))
        np.testing.assert_allclose(cpu(y), want, rtol=2e-4, atol=3e-4, err_msg=f"case {case}")
        np.testing.assert_array_equal(cpu(nl), wl)


def test_random_ctc_loss_vs_oracle():
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    rng = np.random.default_rng(7)
    for case in range(12):
        Tn, N, V = int(rng.integers(1, 80)), int(rng.integers(1, 6)), int(rng.integers(2, 40))
        S = int(rng.integers(1, 12))
        x = (rng.normal(size=(Tn, N, V)) * 3).astype(np.float32)
        xl = rng.integers(1, Tn + 1, size=N).astype(np.int64)
        yl = rng.integers(0, S + 1, size=N).astype(np.int64)
        blank = int(rng.integers(0, V))
        y = rng.integers(0, V - 1, size=(N, S)).astype(np.int64)
        y[y >= blank] += 1  # labels never equal blank
        for red in ("none", "sum", "mean"):
            got = cpu(CTCLoss(blank=blank, reduction=red, zero_infinity=True)((T(x), T(xl)), (T(y), T(yl))))
            want = O.ctc_loss(x, xl, y, yl, blank, red, True)
            np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-3, err_msg=f"case {case} {red}")
