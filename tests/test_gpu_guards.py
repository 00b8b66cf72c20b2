"""Guards around the persistent recurrent launches and the host-length side channel (VERDICT r1 items 6 / ADVICE r1).
Needs a real MI355X: -m gpu.

The persistent LSTM / GRU kernels fill every CU with workgroups that spin on their peers, so (i) the whole grid has to fit
the device (occupancy API x CU count, decided once per device at pack time; otherwise the per-step kernels run), (ii) two
such launches must never be resident together (launches of this process are chained across streams), (iii) a foreign
kernel that holds CUs for a while only delays the launch.  The lengths side channel must not outlive an in-place edit."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _lstm(seed, H=1024, layers=2):
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(seed)
    return RNN(RNNType.LSTM, 128, H, num_layers=layers, bidirectional=True, forget_gate_bias=1.0).eval()


def _inputs(seed, T_=40, N=32):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T_, N, 128, generator=g).cuda()
    lens = torch.sort(torch.randint(5, T_ + 1, (N,), generator=g), descending=True).values
    lens[0] = T_
    return x, lens


def test_two_persistent_layers_on_two_streams_do_not_wait_for_each_other():
    """Two 256-workgroup persistent stacks issued back to back on two streams: without the cross-stream chain both grids
    can be half resident and spin until the 2 s limit (MS_ERR_TIMEOUT + garbage); with it they run one after the other."""
    a, b = _lstm(1), _lstm(2)
    xa, la = _inputs(3)
    xb, lb = _inputs(4)
    (ya, _), (hna, _) = a((xa, la))        # also packs the weights; quiet, sequential answers
    (yb, _), (hnb, _) = b((xb, lb))
    torch.cuda.synchronize()
    a.check_status = b.check_status = False          # no host sync between the two issues
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(6):
        with torch.cuda.stream(sa):
            (y1, _), (h1, _) = a((xa, la))
        with torch.cuda.stream(sb):
            (y2, _), (h2, _) = b((xb, lb))
        outs.append((y1, h1, y2, h2))
    torch.cuda.synchronize()
    from myrtlespeech_amd import _lib
    lib = _lib.load()
    for m in (a, b):
        _lib.check(lib.ms_rnn_status(_lib.ptr(m._workspace.buf), _lib.stream_ptr()), "persistent LSTM on two streams")
    for y1, h1, y2, h2 in outs:
        assert torch.equal(y1, ya) and torch.equal(h1, hna)
        assert torch.equal(y2, yb) and torch.equal(h2, hnb)


def test_persistent_layer_behind_a_second_stream_occupant():
    """A foreign kernel sequence that keeps every CU busy on another stream (large f32 GEMMs, ~100 ms in all) while the
    persistent stack is launched: its workgroups trickle in as CUs free up, the early ones spin meanwhile; the answer is
    the quiet run's, bit for bit, and no time-out is recorded."""
    m = _lstm(5)
    x, lens = _inputs(6)
    (y0, _), (hn0, cn0) = m((x, lens))
    torch.cuda.synchronize()
    big = torch.randn(8192, 8192, device="cuda")
    side = torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(8):
                big2 = big @ big
        (y, _), (hn, cn) = m((x, lens))     # check_status=True: raises on a time-out
        assert torch.equal(y, y0) and torch.equal(hn, hn0) and torch.equal(cn, cn0)
        torch.cuda.synchronize()
    del big2


def test_grid_that_cannot_be_resident_takes_the_step_kernels_in_subprocess():
    """MS_RNN_FAKE_OCCUPANCY=0 makes the occupancy probe answer 'no block fits': LSTM-1024 and GRU-2560 must then be packed
    for and run by the per-step kernels (no spinning grid), with the same results against the oracle."""
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from myrtlespeech_amd import _lib
from myrtlespeech_amd.model.rnn import RNN, RNNType
from oracle import ds_oracle as O
lib = _lib.load()
assert lib.ms_rnn_layer_chains_planes(_lib.CELL_LSTM, 1024, 2) == 0
assert lib.ms_rnn_layer_chains_planes(_lib.CELL_GRU, 2560, 1) == 0
for kind, okind, H, bi in ((RNNType.LSTM, O.LSTM, 1024, True), (RNNType.GRU, O.GRU, 1280, False)):
    torch.manual_seed(H)
    m = RNN(kind, 64, H, num_layers=2, bidirectional=bi, forget_gate_bias=1.0 if kind == RNNType.LSTM else None).eval()
    rng = np.random.default_rng(H)
    T_, N = 7, 9
    lens = np.sort(rng.integers(1, T_ + 1, size=N))[::-1].copy(); lens[0] = T_
    x = rng.normal(size=(T_, N, 64)).astype(np.float32)
    (out, _), hid = m((torch.from_numpy(x), torch.from_numpy(lens)))
    sd = {k[4:]: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want, whid = O.rnn_forward(okind, x, lens, sd, H, 2, bi)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    hn = hid[0] if isinstance(hid, tuple) else hid
    np.testing.assert_allclose(hn.cpu().numpy(), whid[0] if isinstance(whid, tuple) else whid, rtol=1e-4, atol=1e-4)
print("step-kernel fallback ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MS_RNN_FAKE_OCCUPANCY="0"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "step-kernel fallback ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_in_place_edit_of_the_lengths_between_two_modules_is_seen():
    """conv -> (caller edits the returned device lengths in place) -> conv -> RNN: the second module must take its output
    lengths, its mask and its step count from the edited values, as it does for a fresh tensor holding the same values."""
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(0)
    c1 = MaskConv2d(1, 4, [5, 3], [2, 1], PaddingMode.SAME).eval()
    c2 = MaskConv2d(4, 4, [3, 3], [1, 2], PaddingMode.SAME).eval()
    rnn = RNN(RNNType.LSTM, 4 * 8, 64, bidirectional=True).eval()
    x = torch.randn(3, 1, 16, 30)
    lens = torch.tensor([30, 22, 9])

    def tail(h, l):
        y, l2 = c2((h.clone(), l))
        n, c, f, t = y.shape
        (o, l3), (hn, _) = rnn((y.permute(3, 0, 1, 2).reshape(t, n, c * f).contiguous(), l2))
        return y, l2, o, hn

    h, l1 = c1((x.clone(), lens))
    assert l1.is_cuda and _lib.cached_host(l1) is not None
    edited = l1.clone()                       # plain tensor with the edited values: the expected behaviour
    edited -= 3
    l1.sub_(3)                                # in place, behind the library's back
    assert _lib.cached_host(l1) is None
    got = tail(h, l1)
    want = tail(h, edited)
    assert torch.equal(got[1].cpu(), want[1].cpu())
    from myrtlespeech_amd.model.cnn import out_lens, pad_same
    assert torch.equal(got[1].cpu(), out_lens(edited.cpu(), 3, 2, 1, sum(pad_same(30, 3, 2, 1))))
    assert [int(v) for v in got[1].cpu()] == [14, 10, 3]       # from the edited (27, 19, 6), not the stale (30, 22, 9)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # and the masked update idiom
    h, l1 = c1((x.clone(), lens))
    l1[l1 > 25] = 25
    y, l2 = c2((h.clone(), l1))
    assert [int(v) for v in l2.cpu()] == [13, 11, 5]


def test_ds2_forward_under_inference_mode_equals_no_grad():
    """a forward under ``torch.inference_mode()`` (lengths and activations are inference tensors without
    a version counter) must run and give the bits of the ``no_grad`` run, through the pipeline too."""
    import __graft_entry__  # noqa: F401  (sys.path)
    from myrtlespeech_amd.pipeline import TwoBatchesInFlight
    from test_gpu_pipeline import _batches, _small_ds2
    model = _small_ds2(256)
    batches = _batches(3, 6, 80, 40, 21)
    with torch.no_grad():
        want = [model((x.clone(), lens)) for x, lens in batches]
    with torch.inference_mode():
        got = [model((x.clone(), lens)) for x, lens in batches]
        piped = TwoBatchesInFlight(model)([(x.clone(), lens) for x, lens in batches])
    for ((y, ol), (hn, cn)), ((wy, wol), (whn, wcn)), ((py, _), _) in zip(got, want, piped):
        assert torch.equal(y, wy) and torch.equal(ol.cpu(), wol.cpu()) and torch.equal(hn, whn) and torch.equal(cn, wcn)
        assert torch.equal(py, wy)
