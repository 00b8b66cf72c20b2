"""Size-independent properties at BASELINE.json's full sizes, on the GPU: what the utterance-sharded
multi-GPU path relies on, checked on one card.  DS2 config 2 (2 x conv2d, 5 x BiLSTM-1024, FC), 80 features x 1001
frames, ragged sorted batch of 32."""
import numpy as np
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    model = bench.build_model()
    g = torch.Generator().manual_seed(99)
    x = torch.randn(32, 1, 80, 1001, generator=g)
    lens = torch.sort(torch.randint(501, 1002, (32,), generator=g), descending=True).values
    lens[0] = 1001
    (y, ol), _ = model((x.clone(), lens))
    return model, CTCGreedyDecoder(28), x, lens, y, ol


def test_shards_reproduce_the_whole_batch_bit_for_bit(setup):
    """Contiguous shards of the sorted batch (what each rank of an N-GPU job runs) give exactly the rows of the
    single-GPU result -- logits, lengths and transcripts -- for 2, 4 and 8 ranks: no utterance sees another."""
    from myrtlespeech_amd.parallel import shard_batch
    model, dec, x, lens, y, ol = setup
    whole = dec(y, ol)
    for world in (2, 4, 8):
        hyps, row = [], 0
        for rank in range(world):
            xs, ls = shard_batch(x, lens, world, rank)
            (ys, ols), _ = model((xs.clone(), ls))
            t = ys.shape[0]
            diff = (ys - y[:t, row:row + ls.numel()]).abs()
            assert torch.equal(ys, y[:t, row:row + ls.numel()]), (world, rank, float(diff.max()), (diff.amax(dim=(0, 2)) > 0).nonzero().flatten().tolist(), (diff.amax(dim=(1, 2)) > 0).nonzero().flatten().tolist()[:5])
            assert torch.equal(ols.cpu(), ol[row:row + ls.numel()].cpu())
            hyps += dec(ys, ols)
            row += ls.numel()
        assert hyps == whole


def test_a_batch_of_96_gives_every_utterance_the_bits_of_its_batch_of_32(setup):
    """Batches beyond the 64 rows of one wide-workgroup recurrence launch run as launches of 64 rows of the SAME kernel
    (round 6; they used to fall to the 8-unit kernel in groups of 32, whose K shares add up in another order): an
    utterance's logits do not depend on how many others share its batch -- here three interleaved copies of the ragged
    batch, sorted by length as the reference requires (rnn.py:170-183)."""
    model, dec, x, lens, y, ol = setup
    lens3 = torch.cat([lens, lens, lens])
    order = torch.sort(lens3, descending=True, stable=True).indices
    x3 = torch.cat([x, x, x])[order]
    (y3, ol3), _ = model((x3.clone(), lens3[order]))
    src = (order % 32).to(y.device)
    assert y3.shape == (y.shape[0], 96, y.shape[2])
    assert torch.equal(y3, y.index_select(1, src))
    assert torch.equal(ol3.cpu(), ol.cpu()[order % 32])
    assert dec(y3, ol3) == [dec(y, ol)[k] for k in (order % 32).tolist()]


def test_padding_content_is_ignored(setup):
    """Whatever lies in the padded frames of the input (t >= len) cannot reach the output: MaskConv zeroes it
    (cnn.py:425-443) and the packed recurrence never reads it."""
    model, dec, x, lens, y, ol = setup
    noisy = x.clone()
    for n, l in enumerate(lens.tolist()):
        noisy[n, :, :, l:] = 1e3 * torch.randn(1, 80, 1001 - l)
    (y2, ol2), _ = model((noisy, lens))
    assert torch.equal(y2, y) and torch.equal(ol2.cpu(), ol.cpu())


def test_output_rows_past_each_length_are_zero_before_the_fc_bias(setup):
    """Packed-sequence semantics (rnn.py:170-183): recurrent outputs at t >= len are exactly zero, so past its length
    every utterance's logits equal the constant FC response to a zero vector."""
    model, dec, x, lens, y, ol = setup
    const = y[ol[-1].item():, -1]            # the shortest utterance's padded tail
    assert const.shape[0] > 0 and torch.equal(const, const[:1].expand_as(const))
    for n in (0, 7, 20, 31):
        tail = y[ol[n].item():, n]
        if tail.shape[0]:
            assert torch.equal(tail, const[:1].expand_as(tail))


def test_decode_is_deterministic_and_idempotent(setup):
    model, dec, x, lens, y, ol = setup
    a, b = dec(y, ol), dec(y.clone(), ol.clone())
    assert a == b
    (y2, ol2), _ = model((x.clone(), lens))
    assert torch.equal(y2, y)


@pytest.mark.parametrize("kind,H,bidir", [("GRU", 2560, False), ("GRU", 1280, True), ("GRU", 256, True), ("LSTM", 512, True)])
def test_recurrent_layers_do_not_mix_utterances(kind, H, bidir):
    """The persistent GRU (H = 2560 / 1280), the streamed-weights step kernel (GRU-256) and another persistent LSTM
    width: rows of a ragged batch equal the same rows run as 2 and 4 contiguous shards, bit for bit (outputs and the
    final states), including the backward direction whose step index depends on the batch's longest sequence."""
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(H)
    m = RNN(getattr(RNNType, kind), 64, H, num_layers=2, bidirectional=bidir).eval()
    g = torch.Generator().manual_seed(H + 1)
    T_, N = 40, 32
    x = torch.randn(T_, N, 64, generator=g)
    lens = torch.sort(torch.randint(5, T_ + 1, (N,), generator=g), descending=True).values
    lens[0] = T_
    (yw, _), hw = m((x, lens))
    hw = hw[0] if isinstance(hw, tuple) else hw
    for world in (2, 4):
        per = N // world
        for r in range(world):
            sl = slice(r * per, (r + 1) * per)
            (ys, _), hs = m((x[:, sl].contiguous(), lens[sl]))
            hs = hs[0] if isinstance(hs, tuple) else hs
            assert torch.equal(ys, yw[:, sl]), (kind, H, world, r, float((ys - yw[:, sl]).abs().max()))
            assert torch.equal(hs, hw[:, sl])


@pytest.mark.parametrize("kind,H,bidir,N", [("LSTM", 512, True, 64), ("LSTM", 256, True, 50), ("LSTM", 768, False, 64),
                                            ("LSTM", 512, False, 45), ("LSTM", 768, True, 64), ("LSTM", 512, True, 100),
                                            ("GRU", 512, True, 64), ("GRU", 1024, False, 40), ("GRU", 1280, False, 64),
                                            ("GRU", 768, True, 64), ("GRU", 1280, True, 50)])
def test_two_batch_groups_in_one_launch_give_the_bits_of_their_own_launches(kind, H, bidir, N):
    """Round 6: where two batch groups' workgroups of the 8-unit two-stream LSTM kernel fit the CUs together (H <= 512
    bidirectional, H <= 1024 unidirectional) a batch of 33 .. 64 rows runs both groups in ONE launch (and 100 rows as 64 + 36);
    768 bidirectional does not fit and keeps one group per launch.  Either way every row gets the bits it gets when its group
    of 32 runs alone, initial state and ragged lengths included."""
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    torch.manual_seed(H + N)
    m = RNN(getattr(RNNType, kind), 96, H, num_layers=2, bidirectional=bidir, forget_gate_bias=1.0 if kind == "LSTM" else None).eval()
    g = torch.Generator().manual_seed(H + N + 1)
    T_ = 37
    D = 2 if bidir else 1
    x = torch.randn(T_, N, 96, generator=g)
    lens = torch.sort(torch.randint(3, T_ + 1, (N,), generator=g), descending=True).values
    lens[0] = T_
    h0, c0 = torch.randn(2 * D, N, H, generator=g) * 0.3, torch.randn(2 * D, N, H, generator=g) * 0.3
    lstm = kind == "LSTM"
    (yw, _), hid = m((x, lens), (h0, c0) if lstm else h0)
    hw, cw = hid if lstm else (hid, None)
    for a in range(0, N, 32):
        sl = slice(a, min(a + 32, N))
        # (a slice's longest sequence may be shorter than the batch's: the module returns max(lens) frames' worth of time)
        hx = (h0[:, sl].contiguous(), c0[:, sl].contiguous()) if lstm else h0[:, sl].contiguous()
        (ys, _), hid = m((x[:, sl].contiguous(), lens[sl]), hx)
        hs, cs = hid if lstm else (hid, None)
        t = ys.shape[0]
        assert torch.equal(ys, yw[:t, sl]), (kind, H, N, a, float((ys - yw[:t, sl]).abs().max()))
        assert float(yw[t:, sl].abs().max()) == 0.0 if t < yw.shape[0] else True
        assert torch.equal(hs, hw[:, sl]) and (not lstm or torch.equal(cs, cw[:, sl]))
