"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and
exports exactly what include/ms_hotpath.h declares; host-side argument validation mirrors
the reference's ValueErrors.  No compute entry point is called."""
import numpy as np
import pytest
import torch

from myrtlespeech_amd import _lib


def test_library_exports_every_declared_symbol(lib):
    declared = _lib.header_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.SIGNATURES) == declared
    # the binding, the header and the library agree on the ABI version (``_lib.load`` refuses a library that differs)
    import re
    with open(_lib.HEADER_PATH) as f:
        header_version = int(re.search(r"#define MS_ABI_VERSION (\d+)", f.read()).group(1))
    assert lib.ms_abi_version() == header_version == _lib.ABI_VERSION == 4


def test_size_queries_without_gpu(lib):
    # pure host arithmetic: safe on a CPU-only box
    assert lib.ms_maskconv_packed_bytes(32, 1, 41, 11, 1) == 1 * 41 * 12 * 32 * 4
    assert lib.ms_maskconv_packed_bytes(0, 1, 1, 1, 1) == 0
    assert lib.ms_rnn_packed_bytes(0, 640, 1024, 2) > 2 * 4096 * (640 + 1024) * 4
    assert lib.ms_rnn_packed_bytes(9, 1, 1, 1) == 0
    assert lib.ms_ctc_loss_workspace_bytes(501, 32, 29, 241) >= 501 * 32 * 4


def test_pad_same_and_out_lens_match_reference_values():
    from myrtlespeech_amd.model.cnn import out_lens, pad_same
    assert pad_same(80, 41, 2) == (20, 20)
    assert pad_same(1001, 11, 2) == (5, 6)
    assert pad_same(40, 21, 2) == (10, 10)
    assert pad_same(501, 11, 1) == (5, 5)
    for bad in [(0, 1, 1, 1), (1, 0, 1, 1), (1, 1, 0, 1), (1, 1, 1, 0)]:
        with pytest.raises(ValueError):
            pad_same(*bad)
    for dt in (torch.int32, torch.int64, torch.float32):
        l = torch.tensor([1001, 700, 11], dtype=dt)
        got = out_lens(l, 11, 2, 1, 11)
        assert got.dtype == dt
        assert got.tolist() == [501, 351, 6]
    # property from tests/model/test_cnn.py:185-256: SAME => ceil(L / stride)
    rng = np.random.default_rng(0)
    for _ in range(200):
        L, k, s, d = int(rng.integers(1, 300)), int(rng.integers(1, 12)), int(rng.integers(1, 5)), int(rng.integers(1, 3))
        pl, pr = pad_same(L, k, s, d)
        n_out = (L + pl + pr - (d * (k - 1) + 1)) // s + 1
        assert n_out == -(-L // s)
        assert out_lens(torch.tensor([L]), k, s, d, pl + pr).item() == n_out


def test_decoder_argument_validation():
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    x = torch.zeros(5, 2, 4)
    for dec in (CTCGreedyDecoder(0), CTCBeamDecoder(0, 2)):
        with pytest.raises(ValueError):
            dec(x, torch.tensor([5.0, 3.0]))          # float lengths
        with pytest.raises(ValueError):
            dec(x, torch.tensor([5]))                 # batch mismatch
        with pytest.raises(ValueError):
            dec(x, torch.tensor([6, 1]))              # length > seq_len
    # ctor contracts, tests/post_process/test_ctc_beam_decoder.py:117-207
    with pytest.raises(ValueError):
        CTCBeamDecoder(-1, 2)
    with pytest.raises(ValueError):
        CTCBeamDecoder(0, 0)
    with pytest.raises(ValueError):
        CTCBeamDecoder(0, 2, prune_threshold=-0.1)
    with pytest.raises(ValueError):
        CTCBeamDecoder(0, 2, prune_threshold=1.1)
    with pytest.raises(ValueError):
        CTCBeamDecoder(0, 2, language_model=lambda p: 1.0)
    with pytest.raises(ValueError):
        CTCBeamDecoder(0, 2, separator_index=-1)


def test_fully_connected_ctor_contracts():
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    with pytest.raises(ValueError):
        FullyConnected(3, 2, -1, None, None)
    with pytest.raises(ValueError):
        FullyConnected(3, 2, 0, 4, None)
    with pytest.raises(ValueError):
        FullyConnected(3, 2, 0, None, torch.nn.ReLU())
    with pytest.raises(ValueError):
        FullyConnected(3, 2, 0, None, None, dropout=0.5)
    with pytest.raises(ValueError):
        FullyConnected(3, 2, 1, 4, None, dropout=1.5)
    m = FullyConnected(3, 2, 2, 4, torch.nn.Hardtanh(0, 20))
    assert sorted(m.state_dict()) == ["fully_connected.0.bias", "fully_connected.0.weight", "fully_connected.2.bias",
                                      "fully_connected.2.weight", "fully_connected.4.bias", "fully_connected.4.weight"]


def test_state_dict_keys_and_init_match_reference_layout():
    """SURVEY 8b weight-format contract + 8g.8 forget-gate quirk."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.deep_speech_1 import DeepSpeech1
    from myrtlespeech_amd.model.hard_lstm import HardLSTM
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    if torch.cuda.is_available():
        pytest.skip("CPU-side structural test")
    r = RNN(RNNType.LSTM, 6, 4, num_layers=2, bidirectional=True, forget_gate_bias=1.0)
    keys = set(r.state_dict())
    assert {"rnn.weight_ih_l0", "rnn.weight_hh_l1_reverse", "rnn.bias_ih_l0_reverse", "rnn.bias_hh_l1"} <= keys
    assert isinstance(r.rnn, torch.nn.LSTM)
    assert float(r.rnn.bias_ih_l0[4:8].sum()) == 4.0 and float(r.rnn.bias_hh_l1[4:8].abs().sum()) == 0.0
    assert float(r.rnn.bias_ih_l0_reverse[4:8].sum()) != 4.0      # reverse biases keep default init
    assert RNN(RNNType.GRU, 3, 2).rnn.__class__ is torch.nn.GRU
    with pytest.raises(ValueError):
        RNN(7, 3, 2)
    h = HardLSTM(5, 3, num_layers=2, bidirectional=True, forget_gate_bias=1.0)
    assert "rnn.layers.1.bwd.cell.weight_hh" in h.state_dict()
    assert float(h.rnn.layers[0].bwd.cell.bias_ih[3:6].sum()) == 3.0  # HardLSTM sets both directions
    d = DeepSpeech1(5, 3, 8, 6, 0.1)
    assert {"fc1.0.weight", "fc4.0.bias", "bi_lstm.rnn.weight_ih_l0_reverse", "out.weight"} <= set(d.state_dict())
    c = MaskConv2d(1, 4, [5, 3], [2, 2], PaddingMode.SAME)
    assert sorted(c.state_dict()) == ["bias", "weight"] and "padding_mode=PaddingMode.SAME" in repr(c)


def test_no_cpu_fallback():
    """The product path must fail loudly without a HIP device."""
    if torch.cuda.is_available():
        pytest.skip("needs a CPU-only box")
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    with pytest.raises(RuntimeError, match="HIP device"):
        RNN(RNNType.LSTM, 4, 8)((torch.randn(3, 2, 4), torch.tensor([3, 2])))
    with pytest.raises(RuntimeError, match="HIP device"):
        CTCGreedyDecoder(0)(torch.zeros(3, 1, 2), torch.tensor([3]))


def test_device_pointer_arguments_refuse_host_and_strided_tensors():
    """``_lib.ptr`` turned any tensor's ``data_ptr()`` into a kernel argument -- a CPU tensor reached a
    kernel as a device pointer (GPU page fault, process abort).  It now raises instead; ``host_ptr`` is the explicit door for
    the ``_host`` arguments."""
    from myrtlespeech_amd import _lib
    assert _lib.ptr(None).value in (None, 0)
    with pytest.raises(ValueError, match="cannot be a device-pointer argument"):
        _lib.ptr(torch.zeros(1))
    with pytest.raises(TypeError):
        _lib.ptr(np.zeros(1))
    assert _lib.host_ptr(torch.zeros(4)).value
    with pytest.raises(ValueError):
        _lib.host_ptr(torch.zeros(4, 4)[:, 1])
    if torch.cuda.is_available():
        with pytest.raises(ValueError, match="not contiguous"):
            _lib.ptr(torch.zeros(4, 4, device="cuda")[:, 1])
        with pytest.raises(ValueError):
            _lib.host_ptr(torch.zeros(4, device="cuda"))


def test_levenshtein():
    from myrtlespeech_amd.post_process.utils import levenshtein
    from oracle.ds_oracle import levenshtein as ref
    rng = np.random.default_rng(1)
    assert levenshtein("kitten", "sitting") == 3
    for _ in range(100):
        a = rng.integers(0, 4, size=int(rng.integers(0, 9))).tolist()
        b = rng.integers(0, 4, size=int(rng.integers(0, 9))).tolist()
        assert levenshtein(a, b) == ref(a, b)


def test_host_length_side_channel_drops_stale_values():
    """the host copy attached to a lengths tensor must not outlive an in-place edit of that tensor
    (``lens -= k`` between two modules, the reference's float in-place ``out_lens``, ``cnn.py:191-197``)."""
    from myrtlespeech_amd import _lib
    host = torch.tensor([9, 7, 4])
    dev = _lib.attach_host(host.clone(), host)          # a CPU stand-in for the device tensor: same bookkeeping
    assert torch.equal(_lib.cached_host(dev), host)
    host[0] = 100                                       # the caller's own tensor is not aliased
    assert _lib.cached_host(dev)[0] == 9
    dev.sub_(1)                                         # in-place edit -> version bump -> stale
    assert _lib.cached_host(dev) is None
    assert torch.equal(_lib.host_lens(dev), torch.tensor([8, 6, 3]))
    dev2 = _lib.attach_host(torch.tensor([5, 5]), torch.tensor([5, 5]))
    dev2.data = torch.tensor([3, 2])                    # re-pointed storage -> address differs -> stale
    assert _lib.cached_host(dev2) is None
    dev3 = _lib.attach_host(torch.tensor([5, 5]), torch.tensor([5, 5]))
    dev3[1] = 2                                         # masked / indexed update
    assert _lib.cached_host(dev3) is None


def test_host_length_side_channel_under_inference_mode():
    """inference tensors track no version counter (reading ``_version`` raises); they cannot be edited in
    place outside inference mode either, so the side channel keys them on their address."""
    from myrtlespeech_amd import _lib
    with torch.inference_mode():
        host = torch.tensor([9, 7, 4])
        dev = _lib.attach_host(host.clone(), host)
        assert dev.is_inference()
        assert torch.equal(_lib.cached_host(dev), host)
        assert torch.equal(_lib.host_lens(dev), host)
    assert torch.equal(_lib.cached_host(dev), host)         # read back outside the mode: still keyed on the address
    with torch.inference_mode():
        p = torch.nn.Parameter(torch.zeros(2, 2))
        assert _lib.version_of(p) == -1
    assert _lib.version_of(torch.zeros(2)) == 0
