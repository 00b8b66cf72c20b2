"""Config-size checks shared by the in-process GPU tests and their child processes (``MS_PRECISION`` is read once per
process, so the fp16 legs run in a child: ``python -c "import cfg_checks; cfg_checks.stream64(...)"``).  Needs a GPU."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from util import Golden, unragged  # noqa: E402


def cfg2_full(atol: float = 1e-3) -> float:
    """BASELINE.json configs[1] at full size (32 x 1001 frames, 2 x conv2d + 5 x BiLSTM-1024 + FC) in THIS process's precision
    mode against the reference's own outputs (tests/golden/ds2_cfg2_summary.npz, made by gen_golden.py::gen_cfg2_summary
    from the imported reference, model/deep_speech_2.py:123-172 in fp32): weights and inputs regenerated from the
    generator's seeds (weight checksums pinned), logits on the stored sub-grid and (h_n, c_n) within `atol`, output
    lengths equal, greedy transcripts bit-exact.  Returns the max |logit error| on the sub-grid."""
    import bench
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    g = Golden("ds2_cfg2_summary")
    model = bench.build_model()
    for k, v in model.state_dict().items():
        want = g.cfg["weight_abs_sums"][k]
        assert abs(float(v.double().abs().sum()) - want) <= 1e-6 * max(1.0, want), k
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    N, Tn = g.cfg["N"], g.cfg["T"]
    x = torch.randn(N, 1, 80, Tn, generator=gen)
    lens = torch.sort(torch.randint(501, 1002, (N,), generator=gen), descending=True).values
    lens[0] = Tn
    assert abs(float(x.double().abs().sum()) - float(g["in/x_abs_sum"])) < 1e-3
    np.testing.assert_array_equal(lens.numpy(), g["in/lens"])
    (y, ol), (hn, cn) = model((x, lens))
    np.testing.assert_array_equal(ol.cpu().numpy(), g["out/lens"])
    y_sub = y[::25, ::4, :].cpu().numpy()
    np.testing.assert_allclose(y_sub, g["out/y_sub"], rtol=0, atol=atol)
    np.testing.assert_allclose(hn[:, ::8, ::64].cpu().numpy(), g["out/hn_sub"], rtol=0, atol=atol)
    np.testing.assert_allclose(cn[:, ::8, ::64].cpu().numpy(), g["out/cn_sub"], rtol=0, atol=atol)
    dec = CTCGreedyDecoder(28)(y, ol)
    assert dec == unragged(g["out/greedy_flat"], g["out/greedy_lens"])
    err = float(np.abs(y_sub - g["out/y_sub"]).max())
    print(f"cfg2 full-size [{os.environ.get('MS_PRECISION', 'f16x3')}] max |logit err| on the sub-grid: {err:.3e} "
          f"(mean |logit| {float(g['out/y_abs_mean']):.3e})")
    return err


def trained_model():
    """``bench.build_model()`` (seed-0 weights of BASELINE configs[1]) with the trained-scale gains of
    tests/golden/ds2_cfg2_trained_summary.npz applied and the weight checksums of the REFERENCE model verified."""
    import bench
    from util import apply_trained_gains
    g = Golden("ds2_cfg2_trained_summary")
    model = bench.build_model()
    with torch.no_grad():
        apply_trained_gains(model, g.cfg["gains"])
    for k, v in model.state_dict().items():
        want = g.cfg["weight_abs_sums"][k]
        assert abs(float(v.double().abs().sum()) - want) <= 1e-6 * max(1.0, want), k
    return model, g


def trained_batch(g):
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    N, Tn = g.cfg["N"], g.cfg["T"]
    x = torch.randn(N, 1, 80, Tn, generator=gen)
    lens = torch.sort(torch.randint(501, 1002, (N,), generator=gen), descending=True).values
    lens[0] = Tn
    assert abs(float(x.double().abs().sum()) - float(g["in/x_abs_sum"])) < 1e-3
    np.testing.assert_array_equal(lens.numpy(), g["in/lens"])
    return x, lens


def cfg2_trained(atol: float = 1e-3, strict_transcripts: bool = True) -> dict:
    """BASELINE configs[1] at full size in the regime a TRAINED model lives in: the reference's
    ``DeepSpeech2`` (model/deep_speech_2.py:123-172, model/rnn.py:112-127) with weight_ih x 16, weight_hh x 2, FC weights x 6
    -- mean |logit| 2.8, max 17, 37 % of the LSTM gate pre-activations beyond |4|, greedy transcripts over 28 symbols --
    against THIS process's precision mode: logits on the stored sub-grid and (h_n, c_n) within ``atol`` ABSOLUTE of the
    reference's float32 outputs, greedy transcripts bit-exact, CTCLoss within 1e-4 relative, and the reference's
    ``CTCBeamDecoder(beam 8, prune 1e-3)`` transcripts (post_process/ctc_beam_decoder.py:175-258) bit-exact on
    softmax(OUR logits).  The fixture also holds the reference's float64 twin on the same grid: the reference's own
    float32 rounding sits 7e-4 from it, so the errors are reported against both."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    model, g = trained_model()
    x, lens = trained_batch(g)
    (y, ol), (hn, cn) = model((x, lens))
    np.testing.assert_array_equal(ol.cpu().numpy(), g["out/lens"])
    y_np = y.cpu().numpy()
    y_sub = y_np[::25, ::4, :]
    valid = (np.arange(y_np.shape[0])[:, None] < g["out/lens"][None, :])
    vs = valid[::25, ::4]
    err32 = float(np.abs(y_sub - g["out/y_sub"])[vs].max())
    err64 = float(np.abs(y_sub - g["out/y64_sub"])[vs].max())
    ref3264 = float(np.abs(g["out/y_sub"].astype(np.float64) - g["out/y64_sub"])[vs].max())
    errh = float(np.abs(hn[:, ::8, ::64].cpu().numpy() - g["out/hn_sub"]).max())
    errc = float(np.abs(cn[:, ::8, ::64].cpu().numpy() - g["out/cn_sub"]).max())
    # arg max of every frame (what the greedy decoder collapses) against the reference's
    am = y_np.argmax(-1)
    diff = (am != g["out/argmax"].astype(np.int64)) & valid
    top2 = np.sort(y_np, axis=-1)[..., -2:]
    margin = top2[..., 1] - top2[..., 0]
    flips = int(diff.sum())
    flip_margin = float(margin[diff].max()) if flips else 0.0
    dec = CTCGreedyDecoder(28)(y, ol)
    want_dec = unragged(g["out/greedy_flat"], g["out/greedy_lens"])
    n_bad = sum(a != b for a, b in zip(dec, want_dec))
    # CTC loss of the fixture's targets on our logits
    xl = ol.to(torch.int32)
    tg = torch.from_numpy(g["ctc/y"]).cuda()
    yl = torch.from_numpy(g["ctc/y_lens"])
    loss_none = CTCLoss(blank=28, reduction="none")((y, xl), (tg, yl)).cpu().numpy()
    loss_sum = float(CTCLoss(blank=28, reduction="sum")((y, xl), (tg, yl)))
    ctc_rel = float(np.abs(loss_none / g["ctc/none"] - 1).max())
    # the prefix beam search on the encoder's own posteriors
    probs = torch.softmax(y, dim=2)[:, torch.from_numpy(g["beam/utts"]).cuda(), :].contiguous()
    beam = CTCBeamDecoder(blank_index=28, beam_width=8, prune_threshold=0.001)(probs, torch.from_numpy(g["beam/lens"]))
    want_beam = unragged(g["beam/flat"], g["beam/out_lens"])
    rec = dict(mode=os.environ.get("MS_PRECISION", "f16x3"), logit_err_vs_ref_f32=err32, logit_err_vs_ref_f64=err64,
               ref_f32_vs_ref_f64=ref3264, hn_err=errh, cn_err=errc, argmax_flips=flips, argmax_flip_margin_max=flip_margin,
               greedy_transcripts_differing=n_bad, ctc_none_rel_err=ctc_rel,
               ctc_sum=loss_sum, ctc_sum_ref=float(g["ctc/sum"]), beam_equal=beam == want_beam,
               logit_abs_mean=g.cfg["stats"]["logit_abs_mean"], logit_abs_max=g.cfg["stats"]["logit_abs_max"],
               gate_share_beyond_4=g.cfg["stats"]["gate_share_beyond_4"])
    print("cfg2 trained-scale:", rec)
    assert err32 <= atol, rec
    assert errh <= atol and errc <= 4 * atol, rec       # c_n is unbounded (|c| up to ~20 here): 4e-3 absolute
    assert atol > 1 or (ctc_rel <= 1e-4 and abs(loss_sum / float(g["ctc/sum"]) - 1) <= 1e-5), rec
    if strict_transcripts:
        assert flips == 0 and n_bad == 0, rec
        assert beam == want_beam, (beam, want_beam)
    return rec


def trained_modes_equal() -> None:
    """The throughput modes on the TRAINED-SCALE fixture (saturated gates: a rounding difference would show): ``PairedBatches``,
    ``TwoBatchesInFlight`` and 2 / 4 / 8 contiguous shards give ``torch.equal`` logits, lengths and greedy transcripts to the
    one-batch run of the same utterances."""
    from myrtlespeech_amd.parallel import shard_batch
    from myrtlespeech_amd.pipeline import PairedBatches, TwoBatchesInFlight
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    model, g = trained_model()
    x, lens = trained_batch(g)
    x = x.cuda()
    dec = CTCGreedyDecoder(28)
    (y, ol), (hn, cn) = model((x.clone(), lens))
    want = dec(y, ol)
    assert want == unragged(g["out/greedy_flat"], g["out/greedy_lens"])
    gen2 = torch.Generator().manual_seed(99)
    x2 = torch.randn(32, 1, 80, 1001, generator=gen2).cuda()
    lens2 = torch.sort(torch.randint(501, 1002, (32,), generator=gen2), descending=True).values
    (y2, ol2), _ = model((x2.clone(), lens2))
    got = PairedBatches(model)([(x2.clone(), lens2), (x.clone(), lens)])
    assert torch.equal(got[1][0][0], y) and torch.equal(got[0][0][0], y2), "PairedBatches differs from the one-batch run"
    assert torch.equal(got[1][1][0], hn) and torch.equal(got[1][1][1], cn)
    pipe = TwoBatchesInFlight(model)
    outs = pipe([(x.clone(), lens), (x2.clone(), lens2), (x.clone(), lens)])
    pipe.check_status()
    assert torch.equal(outs[0][0][0], y) and torch.equal(outs[1][0][0], y2) and torch.equal(outs[2][0][0], y)
    for world in (2, 4, 8):
        hyps, row = [], 0
        for rank in range(world):
            xs, ls = shard_batch(x, lens, world, rank)
            (ys, ols), _ = model((xs.clone(), ls))
            assert torch.equal(ys, y[:ys.shape[0], row:row + ls.numel()]), (world, rank)
            hyps += dec(ys, ols)
            row += ls.numel()
        assert hyps == want
    print(f"trained-scale fixture: paired / two-in-flight / 2-4-8 shards == one batch, mode {os.environ.get('MS_PRECISION', 'f16x3')}")


def paired_full(atol: float = 1e-3) -> float:
    """``PairedBatches`` at full size: the golden config-2 batch AND a second seeded batch in one forward of 64 (the two
    32-row groups side by side in the wide-workgroup recurrence).  The golden batch's logits / states / transcripts against
    the reference's summary exactly as in ``cfg2_full``; the other batch against the one-batch path."""
    import bench
    from myrtlespeech_amd.pipeline import PairedBatches
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    g = Golden("ds2_cfg2_summary")
    model = bench.build_model()
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    N, Tn = g.cfg["N"], g.cfg["T"]
    x = torch.randn(N, 1, 80, Tn, generator=gen)
    lens = torch.sort(torch.randint(501, 1002, (N,), generator=gen), descending=True).values
    lens[0] = Tn
    np.testing.assert_array_equal(lens.numpy(), g["in/lens"])
    gen2 = torch.Generator().manual_seed(99)
    x2 = torch.randn(N, 1, 80, Tn, generator=gen2)
    lens2 = torch.sort(torch.randint(501, 1002, (N,), generator=gen2), descending=True).values
    dec = CTCGreedyDecoder(28)
    (y2w, ol2w), _ = model((x2.clone().cuda(), lens2))
    want2 = dec(y2w, ol2w)
    got = PairedBatches(model)([(x2.cuda(), lens2), (x.cuda(), lens)])
    (y2, ol2), _ = got[0]
    ((y, ol), (hn, cn)) = got[1]
    np.testing.assert_array_equal(ol.cpu().numpy(), g["out/lens"])
    y_sub = y[::25, ::4, :].cpu().numpy()
    np.testing.assert_allclose(y_sub, g["out/y_sub"], rtol=0, atol=atol)
    np.testing.assert_allclose(hn[:, ::8, ::64].cpu().numpy(), g["out/hn_sub"], rtol=0, atol=atol)
    np.testing.assert_allclose(cn[:, ::8, ::64].cpu().numpy(), g["out/cn_sub"], rtol=0, atol=atol)
    assert dec(y, ol) == unragged(g["out/greedy_flat"], g["out/greedy_lens"])
    assert dec(y2, ol2) == want2
    assert torch.equal(y2, y2w), "a batch of a pair must equal the one-batch path bit for bit (same kernels, rows independent)"
    err = float(np.abs(y_sub - g["out/y_sub"]).max())
    print(f"cfg2 full-size through PairedBatches (wide-workgroup recurrence): max |logit err| on the sub-grid {err:.3e}; "
          f"other batch vs the one-batch path max |diff| {float((y2 - y2w).abs().max()):.3e}")
    return err


def pipeline_full_equal(n_batches: int = 4) -> None:
    """``TwoBatchesInFlight`` on full-size config-2 batches in THIS process's precision mode: logits, lengths, final
    states and greedy transcripts ``torch.equal`` to the same batches run one after the other on one stream."""
    import bench
    from myrtlespeech_amd.pipeline import TwoBatchesInFlight
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    model = bench.build_model()
    dec = CTCGreedyDecoder(28)
    g = torch.Generator().manual_seed(7)
    batches = []
    for _ in range(n_batches):
        x = torch.randn(32, 1, 80, 1001, generator=g).cuda()
        lens = torch.sort(torch.randint(501, 1002, (32,), generator=g), descending=True).values
        lens[0] = 1001
        batches.append((x, lens))
    want = []
    for x, lens in batches:
        (y, ol), (hn, cn) = model((x.clone(), lens))
        want.append((y, ol, hn, cn, dec(y, ol)))
    pipe = TwoBatchesInFlight(model, post=lambda out: (out, dec.launch(out[0][0], out[0][1])))
    got = pipe([(x.clone(), lens) for x, lens in batches])
    pipe.check_status()
    for (((y, ol), (hn, cn)), pending), (wy, wol, whn, wcn, wdec) in zip(got, want):
        assert torch.equal(y, wy) and torch.equal(ol.cpu(), wol.cpu())
        assert torch.equal(hn, whn) and torch.equal(cn, wcn)
        assert pending.result() == wdec
    print(f"two batches in flight == sequential run, {n_batches} full-size batches, mode "
          f"{os.environ.get('MS_PRECISION', 'f16x3')}")


def stream64(atol: float, check_argmax: bool, name: str = "cfg5_stream_n64_summary") -> float:
    """BASELINE.json configs[4] at its stated batch: the config-2 network on 32-frame (320 ms) chunks, state carried, 64
    ragged utterances (two 32-row batch groups per recurrent layer call, utterances leaving the batch inside and across
    the groups) against the reference run chunk by chunk with ``hx`` threaded (tests/golden/cfg5_stream_n64_summary.npz,
    made by tests/golden/gen_golden.py::gen_streaming_n64).  Returns the max |logit error| on the stored sub-grid."""
    import bench
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    g = Golden(name)
    model = bench.build_model()
    if "gains" in g.cfg:     # cfg5_stream_n64_trained_summary: config 2's trained-scale gains (logits of mean 2.8, max 15)
        from util import apply_trained_gains
        with torch.no_grad():
            apply_trained_gains(model, g.cfg["gains"])
    for k, v in model.state_dict().items():
        want = g.cfg["weight_abs_sums"][k]
        assert abs(float(v.double().abs().sum()) - want) <= 1e-6 * max(1.0, want), k
    gen = torch.Generator().manual_seed(g.cfg["seed_input"])
    N, Tn = g.cfg["N"], g.cfg["T"]
    x = torch.randn(N, 1, 80, Tn, generator=gen)
    lens = torch.sort(torch.randint(20, Tn + 1, (N,), generator=gen), descending=True).values
    lens[0] = Tn
    np.testing.assert_array_equal(lens.numpy(), g["in/lens"])
    (y, out_lens), (hn, cn) = ChunkedDeepSpeech2(model, g.cfg["chunk_frames"])(x, lens)
    y, hn, cn = y.cpu().numpy(), hn.cpu().numpy(), cn.cpu().numpy()
    np.testing.assert_array_equal(out_lens.cpu().numpy(), g["out/lens"])
    np.testing.assert_allclose(y[::3, ::3, ::2], g["out/y_sub"], rtol=0, atol=atol)
    np.testing.assert_allclose(hn[:, ::3, ::64], g["out/hn_sub"], rtol=0, atol=atol)
    np.testing.assert_allclose(cn[:, ::3, ::64], g["out/cn_sub"], rtol=0, atol=atol)
    if check_argmax:
        # every frame's arg max (what the greedy decoder collapses) equals the reference's, except where the two best
        # logits of the frame are closer than the tolerance allows one to tell apart
        am = y.argmax(-1)
        diff = am != g["out/argmax"].astype(np.int64)
        if diff.any():
            top2 = np.sort(y, axis=-1)[..., -2:]
            margin = top2[..., 1] - top2[..., 0]
            # (padded frames past an utterance's end hold the FC's response to zeros: the fixture's arg max there is of exact ties)
            assert float(margin[diff].max()) < 1e-5, f"{int(diff.sum())} arg-max mismatches, margins up to {margin[diff].max()}"
    err = float(np.abs(y[::3, ::3, ::2] - g["out/y_sub"]).max())
    print(f"cfg5 N=64 chunked streaming: max |logit err| {err:.3e} (mean |logit| {g.cfg['y_abs_mean']:.3e})")
    return err


def rccl_one_rank() -> None:
    """BASELINE.json configs[2]'s exchange step on the one GPU this box has: a ONE-rank RCCL group with the collective
    path forced (MS_FORCE_COLLECTIVE=1): init, the device all-gather of the padded logits block, the host-side shape
    exchange over the gloo twin group, and a persistent recurrent launch right behind the collective."""
    import torch.distributed as dist
    assert os.environ.get("MS_FORCE_COLLECTIVE") == "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from myrtlespeech_amd import _lib
        from myrtlespeech_amd import parallel as P
        from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
        g = torch.Generator().manual_seed(7)
        logits = torch.randn(501, 32, 29, generator=g).cuda()
        lens = torch.sort(torch.randint(250, 502, (32,), generator=g), descending=True).values
        lens_dev = _lib.lens_to_device(lens)
        full, full_lens = P.gather_logits(logits, lens_dev)
        assert full.data_ptr() != logits.data_ptr(), "the collective path was not taken"
        assert torch.equal(full, logits)
        assert torch.equal(full_lens.cpu(), lens)
        assert torch.equal(_lib.host_lens(full_lens), lens)          # host values ride along: no read-back in the decoder
        # the real pipeline: encoder on this rank's shard -> all-gather -> batched decode == per-shard decode
        from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
        from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
        from myrtlespeech_amd.model.fully_connected import FullyConnected
        from myrtlespeech_amd.model.rnn import RNN, RNNType
        from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper
        torch.manual_seed(0)

        def act():
            return SeqLenWrapper(torch.nn.Hardtanh(0.0, 20.0), torch.nn.Identity())
        cnn = torch.nn.Sequential(MaskConv2d(1, 8, [11, 5], [2, 2], PaddingMode.SAME), act(),
                                  MaskConv2d(8, 8, [5, 5], [2, 1], PaddingMode.SAME), act())
        rnn = RNN(RNNType.LSTM, 8 * 10, 256, num_layers=2, bidirectional=True, forget_gate_bias=1.0)   # persistent kernel
        fc = FullyConnected(512, 29, 1, 96, torch.nn.Hardtanh(0.0, 20.0))
        model = DeepSpeech2(cnn, rnn, None, fc).eval()
        x = torch.randn(6, 1, 40, 90, generator=g)
        xl = torch.tensor([90, 77, 61, 50, 33, 20])
        dec = CTCGreedyDecoder(28)
        a = P.sharded_forward_decode(model, dec, x.clone(), xl, batched_decode=True)
        b = P.sharded_forward_decode(model, dec, x.clone(), xl, batched_decode=False)
        (y, ol), _ = model((x.clone(), xl))
        assert a == b == dec(y, ol)
        for _ in range(3):   # collective and persistent launches back to back
            (y2, ol2), _ = model((x.clone(), xl))
            f2, _ = P.gather_logits(y2, ol2)
            assert torch.equal(f2, y)
        dist.barrier()
        torch.cuda.synchronize()
        print("rccl one-rank ok: backend", dist.get_backend(), "world", dist.get_world_size())
    finally:
        dist.destroy_process_group()


def gemm_variants_equal() -> None:
    """The split-operand GEMM's kernels (0 = LDS-DMA 8-wave, 7 / 8 / 10 / 11 = LDS-DMA 4-wave co-tenant forms that differ in how the
    DMA pieces are issued -- 10 is the one the pipeline uses --, 2 = register-staged) on the
    same operands in THIS process's precision mode (run in a child with MS_PRECISION=fp16 for the single-pass instantiations):
    bit-identical outputs, ragged edges included."""
    from myrtlespeech_amd import _lib
    lib = _lib.load()
    # (the first three shapes leave CUs without a 256 x 256 tile: variant 0 is then the 256 x 128 eight-wave kernel of round 4;
    # (4100, 64, 4100) fills the chip: variant 0 = the 256 x 256 kernel; (1024, 2048, 8192) is a streaming chunk's projection)
    for M, K, N, act in ((2100, 96, 2052, 0), (4100, 32, 1026, 1), (3000, 640, 2048, 0), (4100, 64, 4100, 1), (1024, 2048, 8192, 0),
                         (700, 640, 1500, 1)):
        g = torch.Generator().manual_seed(M + K + N)
        x = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
        ys = []
        try:
            for variant in (0, 7, 8, 10, 11, 2):
                lib.ms_gemm_set_variant(variant)
                y = torch.full((M + 1, N), float("nan"), device="cuda")
                _lib.check(lib.ms_linear_split_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, K, N, act, 0.0, 1.5,
                                                       _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "linear_split")
                assert bool(torch.isnan(y[M]).all())
                ys.append(y[:M])
        finally:
            lib.ms_gemm_set_variant(0)
        assert all(torch.equal(ys[0], y) for y in ys[1:]), (M, K, N)
        want = x.double() @ w.double().T + b.double()
        if act:
            want = want.clamp(0.0, 1.5)
        tol = 2e-3 if os.environ.get("MS_PRECISION") == "fp16" else 1e-4
        assert float((ys[0].double() - want).abs().max()) < tol * float(want.abs().max() + 1.0)
    print("gemm variants equal in mode", os.environ.get("MS_PRECISION", "f16x3"))


def wide_layer(path: str, n: int = 32, steps: int = 37) -> None:
    """One BiLSTM-1024 layer (the wide-workgroup kernel's shape) on seeded inputs with ragged lengths; outputs and final states are
    saved to ``path``.  Run once in the test process and once in a child with ``MS_LSTM_STAMPS=1`` (the kernel's diagnostic
    instantiation, which tools/wide_stamps.py reads): the two files must hold the same bits.  With the stamps on, the phase sums of
    every workgroup must have been written and add up to the same span for both wave classes."""
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    lib = _lib.load()
    torch.manual_seed(11)
    m = RNN(RNNType.LSTM, 96, 1024, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
    g = torch.Generator().manual_seed(12)
    x = torch.randn(steps, n, 96, generator=g).cuda()
    lens = torch.sort(torch.randint(steps // 2, steps + 1, (n,), generator=g), descending=True).values
    lens[0] = steps
    with torch.no_grad():
        (y, _), (hn, cn) = m((x, lens))
    torch.cuda.synchronize()
    np.savez(path, y=y.cpu().numpy(), hn=hn.cpu().numpy(), cn=cn.cpu().numpy())
    if os.environ.get("MS_LSTM_STAMPS") == "1":
        assert lib.ms_rnn_layer_is_wide(0, 1024, 2, n) == 1
        nwg = 128 * ((n + 31) // 32)
        off = lib.ms_rnn_debug_offset(0, steps, n, 96, 1024, 2)
        dbg = m._workspace.buf[off:off + nwg * 16 * 8].view(torch.int64).reshape(nwg, 16).cpu().double()
        span = {}
        for base in (0, 8):
            tot = dbg[:, base] + dbg[:, base + 2] + dbg[:, base + 3] + dbg[:, base + 4]      # slot 1 is a count
            assert float(tot.min()) > 0.0, "a workgroup wrote no stamps"
            span[base] = tot
        # both wave classes of a workgroup stamp the same loop: their sums agree to the skew of the last barrier
        assert float(((span[0] - span[8]).abs() / span[0]).max()) < 0.02
        print("stamps ok: %.0f ns per stream-step" % (float(span[0].mean()) / 4.0 / (2 * steps) * 10.0))


def wide_fp16_vs_oracle() -> float:
    """``MS_PRECISION=fp16`` (run in a child): the wide-workgroup recurrence's fp16 form -- one fp16 plane of ``W_hh`` and of the
    exchanged ``h``, one MFMA pass -- on one and on two batch groups of ragged rows, two chained BiLSTM-1024 layers, against
    the numpy oracle.  Tolerance: fp16 operands carry 2^-11 relative error per product (|h| <= 1, weights
    ~U(-1/32, 1/32), K = 1024 + In): 3e-3 absolute on outputs and states is ~10x the observed error."""
    from oracle import ds_oracle as O
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    assert os.environ.get("MS_PRECISION") == "fp16"
    lib = _lib.load()
    worst = 0.0
    for n, steps in ((20, 33), (40, 29), (64, 17)):
        assert lib.ms_rnn_layer_is_wide(0, 1024, 2, n) == 1
        torch.manual_seed(31 + n)
        m = RNN(RNNType.LSTM, 96, 1024, num_layers=2, bidirectional=True, forget_gate_bias=1.0).eval()
        g = torch.Generator().manual_seed(32 + n)
        x = torch.randn(steps, n, 96, generator=g)
        lens = torch.sort(torch.randint(1, steps + 1, (n,), generator=g), descending=True).values
        lens[0] = steps
        h0 = torch.randn(4, n, 1024, generator=g) * 0.3
        c0 = torch.randn(4, n, 1024, generator=g) * 0.3
        with torch.no_grad():
            (y, _), (hn, cn) = m((x.cuda(), lens), (h0.cuda(), c0.cuda()))
        torch.cuda.synchronize()
        sd = {k[len("rnn."):]: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        want, (wh, wc) = O.rnn_forward(O.LSTM, x.numpy(), lens.numpy(), sd, 1024, 2, True, (h0.numpy(), c0.numpy()))
        err = max(float(np.abs(y.cpu().numpy() - want).max()), float(np.abs(hn.cpu().numpy() - wh).max()),
                  float(np.abs(cn.cpu().numpy() - wc).max()))
        yy = y.cpu().numpy()
        for i, L in enumerate(lens.tolist()):
            assert not yy[L:, i].any()
        print(f"wide fp16 recurrence, {n} rows x {steps} steps: max err vs oracle {err:.3e}")
        worst = max(worst, err)
    assert worst < 3e-3, worst
    print("wide fp16 ok %.3e" % worst)
    return worst


def ragged_stack(path: str, n: int = 40, steps: int = 48, layers: int = 2) -> float:
    """A chained stack of BiLSTM-1024 layers (the wide-workgroup kernel's shape, two batch groups) on a batch of RAGGED lengths:
    by default the layers work on the rows that exist only (``MS_RNN_PACKED_ROWS``: projection over sum(lens) rows, packed planes
    between the layers), with ``MS_RNN_PACKED=0`` in the environment on all ``steps * n`` rows.  Saves outputs and final states
    to ``path`` (the two modes must give the same bits) and returns the largest error against the numpy oracle."""
    from oracle import ds_oracle as O
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    lib = _lib.load()
    torch.manual_seed(21)
    m = RNN(RNNType.LSTM, 96, 1024, num_layers=layers, bidirectional=True, forget_gate_bias=1.0).eval()
    g = torch.Generator().manual_seed(22)
    x = torch.randn(steps, n, 96, generator=g)
    lens = torch.sort(torch.randint(1, steps + 1, (n,), generator=g), descending=True).values
    lens[0] = steps
    expect_packed = os.environ.get("MS_RNN_PACKED") != "0"
    assert bool(lib.ms_rnn_layer_packs_rows(0, steps, n, 96, 1024, 2)) == expect_packed
    with torch.no_grad():
        (y, _), (hn, cn) = m((x.cuda(), lens))
    torch.cuda.synchronize()
    np.savez(path, y=y.cpu().numpy(), hn=hn.cpu().numpy(), cn=cn.cpu().numpy())
    sd = {k[len("rnn."):]: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want, (wh, wc) = O.rnn_forward(O.LSTM, x.numpy(), lens.numpy(), sd, 1024, layers, True, None)
    err = max(float(np.abs(y.cpu().numpy() - want).max()), float(np.abs(hn.cpu().numpy() - wh).max()),
              float(np.abs(cn.cpu().numpy() - wc).max()))
    # rows past a sequence's end are exactly zero (rnn.py:181 pad_packed_sequence)
    yy = y.cpu().numpy()
    for i, L in enumerate(lens.tolist()):
        assert not yy[L:, i].any()
    print("ragged stack: packed rows" if expect_packed else "ragged stack: all rows", "max err vs oracle %.3e" % err)
    return err


def ds2_ragged_logits(path: str) -> None:
    """The config-2 network on one full-size batch of ragged lengths (~U[501, 1001] frames); logits, output lengths and final
    states saved to ``path``: the packed-rows path (default) and ``MS_RNN_PACKED=0`` must write the same bits."""
    import bench
    model = bench.build_model()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(32, 1, 80, 1001, generator=g)
    lens = torch.sort(torch.randint(501, 1002, (32,), generator=g), descending=True).values
    with torch.no_grad():
        (y, ol), (hn, cn) = model((x.cuda(), lens))
    torch.cuda.synchronize()
    np.savez(path, y=y.cpu().numpy(), ol=ol.cpu().numpy(), hn=hn.cpu().numpy(), cn=cn.cpu().numpy())
    print("ds2 ragged logits saved:", "all rows" if os.environ.get("MS_RNN_PACKED") == "0" else "packed rows")


def tile64_matches_other_kernels(shapes=((512, 2560, 1024), (1024, 2048, 1024), (130, 96, 200))):
    """The 64 x 64-tile split GEMM against the kernels it replaces (``MS_GEMM_TILE64=0``) in the PROCESS's precision mode
    (run with ``MS_PRECISION=fp16`` for the single-pass fp16 form): the same k-ordered sums, ``torch.equal``."""
    import os
    import numpy as np
    import torch
    from myrtlespeech_amd import _lib
    lib = _lib.load()
    for M, K, N in shapes:
        rng = np.random.default_rng(M + K + N)
        x = torch.from_numpy(rng.normal(size=(M, K)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)).cuda()
        b = torch.from_numpy(rng.normal(size=(N,)).astype(np.float32)).cuda()
        ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
        ys = []
        for flag in ("0", "1"):
            os.environ["MS_GEMM_TILE64"] = flag
            y = torch.full((M + 1, N), float("nan"), dtype=torch.float32, device="cuda")
            _lib.check(lib.ms_linear_split_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, K, N, 1, 0.0, 20.0,
                                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "linear_split")
            assert bool(torch.isnan(y[M]).all())
            ys.append(y[:M])
        assert torch.equal(ys[0], ys[1]), (M, K, N, float((ys[0] - ys[1]).abs().max()))
        want = (x.double() @ w.double().T + b.double()).clamp(0.0, 20.0)
        print(f"tile64 {M}x{K}x{N}: max err vs float64 {float((ys[1].double() - want).abs().max()):.3e}")
