#!/usr/bin/env python3
"""The BASELINE.json configs other than the headline one, and BASELINE.md section 3's separately reported legs, as
functions ``bench.py`` calls at N = 1 (so the numbers are in the line the driver records) and as a CLI:

    python tools/bench_configs.py [ds1 ctc beam rnnt stream frontend] > gpurun_out/configs.json

Every leg: inputs resident in HBM, HIP events (``torch.cuda.Event`` on the stream the library launches on) around each of
>= 10 iterations after a warm-up, ``ms`` = their mean (the two streaming legs: the median, see there); ``floor_ms`` = the leg's algorithmic floor with the arithmetic it
comes from stated in ``floor``; ``frac_of_floor`` = floor_ms / ms (1 = at the floor); ``cpu_baseline`` = the reference's
operator sequence (stock torch CPU operators, ``oracle/torch_cpu.py``) or the numpy oracle on a stated, bounded sample.

Floors are built from two figures measured in the same process: ``barrier_step_us`` (``ms_barrier_chain_probe``: one
LDS-write / barrier / LDS-read phase of a 256-thread workgroup -- what the scan kernels' serial chains are made of) and
``lstm_step_us`` (the persistent recurrence's time per sequential step, one layer, both directions, from a 501-step
launch) -- plus HBM at 8 TB/s for bytes that have to be streamed once.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_GBS = 8000.0
F32_MFMA_TF = 157.3
CLOCK_GHZ = 2.4            # MI355X peak engine clock (MI355X_MICROARCH.md)
# The instruction-chain floor of the CTC alpha pipeline: a frame of the wave that never waits is ~40 mostly
# DEPENDENT instructions (max3 / sub / exp2 / add / log2 / add per state group + the DPP shifts and the ring load); a wave64
# VALU instruction issues over 4 cycles on a 16-lane SIMD, the two transcendentals over 16
CTC_FRAME_CYCLES = 38 * 4 + 2 * 16
CTC_CHAIN_NOTE = ("501 frames x (38 dependent VALU ops x 4 cycles + 2 transcendentals x 16) = 184 cycles per frame at 2.4 GHz on one "
                  "wave per SIMD: what the kernel's frame is made of (measured 0.105 us = 250 cycles on the wave that never waits); "
                  "the barrier-phase floor beside it does not bind this kernel (no barrier in its frame)")


def host_threads():
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, share)))
    return int(torch.get_num_threads())


LAST_SAMPLES = []      # the per-iteration milliseconds of the most recent ev_timed call (legs that report a median read it)


def ev_timed(fn, warmup=2, iters=10):
    """Mean / min device milliseconds of ``fn()`` over ``iters`` runs (HIP events on the current stream)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ms.append(a.elapsed_time(b))
    LAST_SAMPLES[:] = ms
    return float(np.mean(ms)), float(np.min(ms))


def wall_timed(fn, warmup=1, iters=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def barrier_step_us(threads=256, steps=20000):
    """One barrier-separated phase of a workgroup's serial chain, measured: ticks of the 100 MHz wall clock over
    2 * steps phases inside ONE launch (no launch overhead in the figure)."""
    from myrtlespeech_amd import _lib
    lib = _lib.load()
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(2):
        _lib.check(lib.ms_barrier_chain_probe(_lib.ptr(out), steps, threads, _lib.stream_ptr()), "ms_barrier_chain_probe")
    torch.cuda.synchronize()
    return float(out[0].item()) * 0.01 / (2 * steps)


def lstm_step_us(n=32, steps=501):
    """Time per sequential step of the persistent BiLSTM-1024 recurrence (one layer, both directions) at batch ``n``:
    the library's own HIP-event span of the recurrence launch / steps."""
    import ctypes
    from myrtlespeech_amd import _lib
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    lib = _lib.load()
    torch.manual_seed(3)
    m = RNN(RNNType.LSTM, 2048, 1024, num_layers=1, bidirectional=True, forget_gate_bias=1.0).eval()
    x = torch.randn(steps, n, 2048, device="cuda")
    lens = torch.full((n,), steps, dtype=torch.int64)
    kinds = 9
    ms, cnt = (ctypes.c_float * kinds)(), (ctypes.c_int * kinds)()
    with torch.no_grad():
        m((x, lens))
        lib.ms_prof_enable(1)
        lib.ms_prof_read(ms, cnt)
        for _ in range(3):
            m((x, lens))
        torch.cuda.synchronize()
        lib.ms_prof_read(ms, cnt)
        lib.ms_prof_enable(0)
    return ms[1] / max(cnt[1], 1) / steps * 1e3


# ------------------------------------------------------------------------------------------------------------------ legs
def leg_ds1(ctx):
    """BASELINE configs[0]: DeepSpeech1 (scripts/export_ds1_onnx.py:30-41: n_hidden 1024, input [1, 19, 26, 201]) + greedy."""
    from myrtlespeech_amd.model.deep_speech_1 import DeepSpeech1
    from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
    torch.manual_seed(0)
    m = DeepSpeech1(26, 19, 1024, 29, 0.25).eval()
    x = torch.randn(1, 19, 26, 201).cuda()
    lens = torch.tensor([201])
    dec = CTCGreedyDecoder(28)

    from myrtlespeech_amd.streaming import GraphedForward
    gm = GraphedForward(m)       # the forward of this one input shape as a captured HIP graph (same launches, the eager call's bits)

    def run():
        with torch.no_grad():
            (y, ol), _ = gm(x, lens)
        return dec(y, ol)

    def run_eager():
        with torch.no_grad():
            (y, ol), _ = m((x, lens))
        return dec(y, ol)
    ms_eager, _ = ev_timed(run_eager, 3, 20)
    ms, ms_min = ev_timed(run, 3, 20)
    # single-clip serving switch: the wide exact-f32 layers in K slices too (MS_LINEAR_FEW_ROWS; the clip's rounding then
    # differs from the same clip inside a batch, so it is opt-in and `ms` above is the default)
    m.few_rows = True
    gm_few = GraphedForward(m)

    def run_few():
        with torch.no_grad():
            (y, ol), _ = gm_few(x, lens)
        return dec(y, ol)
    ms_few, _ = ev_timed(run_few, 3, 20)
    m.few_rows = False
    params = sum(p.numel() for p in m.parameters())
    step_us = ctx["lstm_step_us_n1"]
    floor = params * 4 / (HBM_GBS * 1e6) + 201 * step_us * 1e-3
    out = {"workload": "cfg[0] DS1 n_hidden 1024, 1 x 4 s clip [1,19,26,201], forward + greedy (export_ds1_onnx.py:30-41)",
           "ms": round(ms, 4), "ms_min": round(ms_min, 4), "ms_eager": round(ms_eager, 4), "ms_few_rows": round(ms_few, 4), "hip_graph_replays": gm.replays,
           "hip_graph_error": gm.graph_error, "audio_sec_per_s": round(4.0 / ms * 1e3, 1),
           "floor_ms": round(floor, 4), "frac_of_floor": round(floor / ms, 3),
           "floor": f"{params * 4 / 1e6:.0f} MB of weights once at 8 TB/s + 201 sequential BiLSTM-1024 steps x {step_us:.2f} us "
                    "(this library's persistent recurrence at batch 1, measured in this run)"}
    if ctx.get("cpu"):
        from oracle import torch_cpu as TC
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        xc = x.cpu().numpy()
        TC.deep_speech_1_forward(xc, np.array([201]), sd, 1024)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            y, yl = TC.deep_speech_1_forward(xc, np.array([201]), sd, 1024)
            TC.ctc_greedy_decode(y, yl, 28)
        dt = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = {"value": round(4.0 / dt, 1), "unit": "audio-sec/s", "ms": round(dt * 1e3, 2), "cores": ctx["cores"],
                               "kind": "port", "sample": f"{reps} passes over the same clip, stock torch CPU operators in the reference's order"}
    return out


def leg_ctc_loss(ctx):
    """CTC loss forward, [501, 32, 29], targets 32 x 120, reduction sum (loss/ctc_loss.py:51-101)."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(501, 32, 29, generator=g).cuda()
    lens = torch.full((32,), 501, dtype=torch.int32)
    tgt = torch.randint(0, 28, (32, 120), dtype=torch.int32, generator=g)
    tl = torch.full((32,), 120, dtype=torch.int32)
    loss = CTCLoss(blank=28, reduction="sum")
    loss.check_status = False          # no read-back + synchronisation per call inside the timed region; asked once below
    ms, ms_min = ev_timed(lambda: loss((logits, lens), (tgt, tl)), 3, 20)
    loss.status()
    b_us = ctx["barrier_step_us"]
    floor = 501 * b_us * 1e-3 + 501 * 32 * 29 * 4 / (HBM_GBS * 1e6)
    chain = 501 * CTC_FRAME_CYCLES / (CLOCK_GHZ * 1e6)
    out = {"workload": "CTC loss forward, logits [501,32,29], targets 32 x 120, reduction sum (BASELINE.md 3)",
           "ms": round(ms, 4), "ms_min": round(ms_min, 4), "floor_ms": round(floor, 4), "frac_of_floor": round(floor / ms, 3),
           "chain_floor_ms": round(chain, 4), "frac_of_chain_floor": round(chain / ms, 3), "chain_floor": CTC_CHAIN_NOTE,
           "floor": f"501 frames x 1 barrier-separated alpha row x {b_us:.3f} us (measured barrier phase) + 1.86 MB at 8 TB/s; "
                    "one workgroup per utterance, 32 of 256 CUs busy.  The kernel (four-wave pipeline, alphas in registers, "
                    "csrc/ctc.hip) has no barrier in its frame: 0.105 us per frame on the wave that never waits, 0.15 us on "
                    "the waves behind it -- one wave per SIMD issues ~40 dependent instructions per frame"}
    if ctx.get("cpu"):
        x = logits.cpu()
        f = torch.nn.CTCLoss(blank=28, reduction="sum")
        lsm = torch.nn.LogSoftmax(dim=-1)
        with torch.no_grad():
            f(lsm(x), tgt, lens, tl)
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                f(lsm(x), tgt, lens, tl)
            dt = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = {"value": round(dt * 1e3, 3), "unit": "ms", "cores": ctx["cores"], "kind": "port",
                               "sample": f"{reps} calls of LogSoftmax + torch.nn.CTCLoss on the same tensors (ctc_loss.py:95-101)"}
    return out


def leg_ctc_grad(ctx):
    """CTC loss forward + backward (the alpha-beta posteriors = the gradient with respect to the logits; loss/ctc_loss.py:95-101
    under autograd), [501, 32, 29], targets 32 x 120, reduction sum."""
    from myrtlespeech_amd.loss.ctc_loss import CTCLoss
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(501, 32, 29, generator=g).cuda().requires_grad_(True)
    lens = torch.full((32,), 501, dtype=torch.int32)
    tgt = torch.randint(0, 28, (32, 120), dtype=torch.int32, generator=g)
    tl = torch.full((32,), 120, dtype=torch.int32)
    loss = CTCLoss(blank=28, reduction="sum")

    loss.check_status = False

    def step():
        logits.grad = None
        loss((logits, lens), (tgt, tl)).backward()
    ms, ms_min = ev_timed(step, 3, 20)
    loss.status()
    b_us = ctx["barrier_step_us"]
    rows = 501 * 32 * 241 * 4
    floor = 2 * 501 * b_us * 1e-3 + (4 * rows + 3 * 501 * 32 * 29 * 4) / (HBM_GBS * 1e6)
    # forward's chain, then alpha and beta side by side (one more chain), then the gradient rows at HBM rate
    chain = 2 * 501 * CTC_FRAME_CYCLES / (CLOCK_GHZ * 1e6) + (4 * rows + 3 * 501 * 32 * 29 * 4) / (HBM_GBS * 1e6)
    out = {"workload": "CTC loss forward + backward, logits [501,32,29], targets 32 x 120, reduction sum",
           "ms": round(ms, 4), "ms_min": round(ms_min, 4), "floor_ms": round(floor, 4), "frac_of_floor": round(floor / ms, 3),
           "chain_floor_ms": round(chain, 4), "frac_of_chain_floor": round(chain / ms, 3), "chain_floor": CTC_CHAIN_NOTE,
           "floor": f"two dependent chains (alpha, beta) of 501 frames x {b_us:.3f} us (measured barrier phase) + alpha / beta rows "
                    "written and read once + logits read, gradient written, at 8 TB/s.  Kernels: the forward pipeline, the same "
                    "kernel twice more with row stores (alpha, and reversed = beta), one wave per frame for the gradient rows "
                    "(csrc/ctc.hip); the LDS-row kernel this replaced took 26 ms"}
    if ctx.get("cpu"):
        x = logits.detach().cpu().requires_grad_(True)
        f = torch.nn.CTCLoss(blank=28, reduction="sum")
        lsm = torch.nn.LogSoftmax(dim=-1)

        def cpu_step():
            x.grad = None
            f(lsm(x), tgt, lens, tl).backward()
        cpu_step()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            cpu_step()
        dt = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = {"value": round(dt * 1e3, 3), "unit": "ms", "cores": ctx["cores"], "kind": "port",
                               "sample": f"{reps} forward + backward passes of LogSoftmax + torch.nn.CTCLoss on the same tensors"}
    return out


def leg_ctc_beam(ctx):
    """CTC prefix beam search, softmax(randn(501, 32, 29) * 12), beam 8 (ctc_beam_decoder.py:175-258)."""
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    g = torch.Generator().manual_seed(0)
    probs = torch.softmax(torch.randn(501, 32, 29, generator=g) * 12, dim=2)
    probs_d = probs.cuda()
    lens = torch.full((32,), 501, dtype=torch.int64)
    dec = CTCBeamDecoder(blank_index=28, beam_width=8)
    ms, ms_min = ev_timed(lambda: dec(probs_d, lens), 2, 10)
    b_us = ctx["barrier_step_us"]
    phases = 7
    floor = 501 * phases * b_us * 1e-3
    # the instruction-chain floor: since round 5 a frame reads nothing from global memory (the live beam's
    # child rows are in LDS); what it is made of is waits for LDS data -- hipcc's gfx950 code of the frame loop has 95
    # `s_waitcnt lgkmcnt` (tools: hipcc -S), a wave passes ~55 of them per frame, each >= the 50-cycle ds_read latency
    chain = 501 * (55 * 50 / (CLOCK_GHZ * 1e3) + phases * b_us) * 1e-3
    out = {"workload": "CTC beam decode, softmax(randn(501,32,29)*12), beam 8, prune 1e-3, no LM (BASELINE.md 3)",
           "ms": round(ms, 3), "ms_min": round(ms_min, 3), "utterances_per_s": round(32 / ms * 1e3, 1),
           "floor_ms": round(floor, 4), "frac_of_floor": round(floor / ms, 3),
           "chain_floor_ms": round(chain, 4), "frac_of_chain_floor": round(chain / ms, 3),
           "chain_floor": "501 frames x (~55 exposed LDS round trips x 50 cycles at 2.4 GHz + 7 barrier phases): one workgroup per "
                          "utterance, one wave per SIMD, every phase a chain of dependent LDS reads (a frame measures 4.6 us = 11 000 cycles)",
           "floor": f"501 frames x {phases} barrier-separated phases per frame (csrc/beam.hip) x {b_us:.3f} us; the dependent LDS "
                    "reads between them are not in this floor (see chain_floor)"}
    if ctx.get("cpu"):
        from oracle import ds_oracle as O
        t0 = time.perf_counter()
        want = O.ctc_beam_decode(probs[:, :1].numpy(), np.array([501]), 28, 8, 0.001)
        dt = time.perf_counter() - t0
        got = dec(probs_d[:, :1].contiguous(), lens[:1])
        out["cpu_baseline"] = {"value": round(1.0 / dt, 3), "unit": "utterances/s", "s_per_utterance": round(dt, 2), "cores": 1,
                               "kind": "port", "sample": "utterance 0 of the 32 through the numpy restatement of the reference's "
                                                         "pure-Python loop (sequential over the batch, ctc_beam_decoder.py:175)",
                               "same_transcript_as_gpu": bool(got == want)}
    return out


def leg_ctc_beam_lm(ctx):
    """The prefix beam search WITH a host language model (ctc_beam_decoder.py:214-230: the model is a Python callable consulted
    for the separator extension of every beam entry that survives pruning): T = 501, 4 utterances, beam 8, separator 0, a
    deterministic stand-in model.  The kernel advances over the runs of frames in which no utterance's separator survives in one
    launch and stops where the model is needed."""
    from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
    g = torch.Generator().manual_seed(1)
    probs = torch.softmax(torch.randn(501, 4, 29, generator=g) * 12, dim=2)
    probs_d = probs.cuda()
    lens = torch.tensor([501, 433, 250, 77], dtype=torch.int64)

    def lm(prefix):          # a pure function of the prefix in [0.05, 0.85] (the shape of oracle.ds_oracle.toy_language_model)
        h = 0
        for s_ in prefix:
            h = (h * 31 + int(s_) + 1) % 1009
        return 0.05 + (h % 17) / 20.0
    dec = CTCBeamDecoder(blank_index=28, beam_width=8, language_model=lm, lm_weight=0.7, separator_index=0, word_weight=1.3)
    ms, ms_min = ev_timed(lambda: dec(probs_d, lens), 1, 4)
    plain = CTCBeamDecoder(blank_index=28, beam_width=8, separator_index=0, word_weight=1.3)
    ms_plain, _ = ev_timed(lambda: plain(probs_d, lens), 1, 4)
    out = {"workload": "CTC beam decode with a host language model: softmax(randn(501,4,29)*12), beam 8, separator 0, lm_weight 0.7",
           "ms": round(ms, 3), "ms_min": round(ms_min, 3), "utterances_per_s": round(4 / ms * 1e3, 1),
           "lm_calls": dec.lm_calls, "lm_frames": dec.lm_frames, "frames": 501,
           "ms_same_search_without_the_model": round(ms_plain, 3),
           "note": "host-bound by construction: every frame whose separator survives pruning costs a read-back of the live beam, "
                   "the model's Python calls and a launch; the frames in between run as one launch per run"}
    if ctx.get("cpu"):
        from oracle import ds_oracle as O
        t0 = time.perf_counter()
        want = O.ctc_beam_decode(probs[:, 3:4].numpy(), np.array([77]), 28, 8, 0.001, language_model=lm, lm_weight=0.7,
                                 separator_index=0, word_weight=1.3)
        dt = time.perf_counter() - t0
        got = dec(probs_d[:, 3:4].contiguous(), lens[3:4])
        out["cpu_baseline"] = {"value": round(77.0 / 501 / dt, 3), "unit": "utterances/s (a 501-frame utterance, from 77 frames)",
                               "cores": 1, "kind": "port", "sample": "utterance 3 (77 frames) through the numpy restatement",
                               "same_transcript_as_gpu": bool(got == want)}
    return out


def leg_rnnt(ctx):
    """BASELINE configs[3] (own specification, parity unpinned): DS2 encoder at batch 16 + 2 x LSTM-1024 predictor + joint 512,
    beam 8, on the inputs and weights of tests/golden/gen_rnnt_cfg4.py (transcripts checked against its stored answers)."""
    import bench
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    from myrtlespeech_amd.model.rnnt import RNNTJoint, RNNTPredictor
    from myrtlespeech_amd.post_process.rnnt_decoder import RNNTBeamDecoder, RNNTGreedyDecoder

    class G:     # the construction of tests/golden/gen_rnnt_cfg4.py (same seeds: its stored answers apply)
        V, D, P, J, E, L = 28, 256, 1024, 512, 1024, 2
        N, T, W, MS = 16, 501, 8, 3

        @staticmethod
        def parts():
            torch.manual_seed(4)
            pred = RNNTPredictor(G.V, G.D, G.P, num_layers=G.L).eval()
            joint = RNNTJoint(G.E, G.P, G.J, G.V).eval()
            with torch.no_grad():
                joint.out.weight.mul_(16.0)
                joint.out.bias[G.V] += 8.0
            return pred, joint

        @staticmethod
        def inputs():
            g = torch.Generator().manual_seed(44)
            enc = torch.randn(G.T, G.N, G.E, generator=g)
            lens = torch.sort(torch.randint(301, G.T + 1, (G.N,), generator=g), descending=True).values
            lens[0] = G.T
            return enc, lens
    enc_model = bench.build_model()
    enc_model.fully_connected = FullyConnected(2048, 1024, 0, None, None)   # encoder projection 2048 -> 1024
    enc_model.eval()
    x = torch.randn(16, 1, 80, 1001).cuda()
    xl = torch.full((16,), 1001, dtype=torch.int64)
    with torch.no_grad():
        ms_enc, _ = ev_timed(lambda: enc_model((x, xl)), 2, 10)
    pred, joint = G.parts()
    enc, lens = G.inputs()
    enc_d = enc.cuda()
    gdec = RNNTGreedyDecoder(pred, joint, max_symbols=G.MS)
    bdec = RNNTBeamDecoder(pred, joint, beam_width=G.W, max_symbols=G.MS)
    ms_b, ms_b_min = ev_timed(lambda: bdec(enc_d, lens), 1, 10)
    hyp = bdec(enc_d, lens)
    ms_g, _ = ev_timed(lambda: gdec(enc_d, lens), 1, 5)
    gold = None
    try:
        with np.load(os.path.join(ROOT, "tests", "golden", "rnnt_cfg4.npz")) as z:
            gold = all(hyp[n] == z[f"out/beam_{n}"].tolist() for n in range(G.N))
    except Exception:
        pass
    frames = int(lens.sum())
    rows = G.N * G.W
    # floor of the round-5 sequence (csrc/rnnt_decode.hip, beam2_*): per predictor step the operand planes of W_hh0, W_ih1,
    # W_hh1 (4H x H) and W_pred (J x H) -- two fp16 planes per weight since round 6, three bf16 before -- are streamed once; a frame is 10 dependent launches
    # (joint, round, cell 0, layer 1 | joint + G, round, cell 0, layer 1 | joint + G, frame end) at the chip's dependent-launch
    # boundary (MI355X_MICROARCH.md price list, "boundary": 1.45 us)
    planes = 3 if os.environ.get("MS_RNNT_PLANES") == "3" else 2          # round 6: two fp16 planes (4 B per weight) by default
    plane_bytes = (3 * 4 * G.P * G.P + G.J * G.P) * 2 * planes
    launches = 3 * G.MS + 2 * (G.MS - 1) - 1 + 2 * (G.MS - 1) * (G.L - 2)      # 10 at MS = 3, L = 2
    per_frame_us = (G.MS - 1) * plane_bytes / (HBM_GBS * 1e3) + launches * 1.45
    floor = G.T * per_frame_us * 1e-3
    audio_s = G.N * 10.0
    out = {"workload": "cfg[3] RNN-T (own spec, parity unpinned): batch 16 x 501 frames, 2xLSTM-1024 predictor, joint 512, beam 8, "
                       "max_symbols 3; encoder = DS2 5xBiLSTM-1024 at batch 16 x 10 s",
           "encoder_ms": round(ms_enc, 3), "beam8_decode_ms": round(ms_b, 2), "beam8_decode_ms_min": round(ms_b_min, 2),
           "greedy_decode_ms": round(ms_g, 2), "ms": round(ms_enc + ms_b, 2),
           "audio_sec_per_s": round(audio_s / (ms_enc + ms_b) * 1e3, 1), "us_per_frame": round(ms_b / G.T * 1e3, 1),
           "launches_per_frame": launches,
           "transcripts_equal_oracle_fixture": gold,
           "floor_ms": round(floor, 2), "frac_of_floor": round(floor / ms_b, 3),
           "floor": f"decode only: 501 frames x ({G.MS - 1} predictor steps x {plane_bytes / 1e6:.0f} MB of operand planes ({planes} x 16 bit per weight) "
                    f"(W_hh0, W_ih1, W_hh1, W_pred) at 8 TB/s + {launches} dependent launches x 1.45 us); the kernels' own latency "
                    "chains (2 .. 4 dependent L2 round trips each) are not in the floor"}
    if ctx.get("cpu"):
        from oracle import rnnt_oracle as RO
        psd = {k: v.detach().cpu().numpy() for k, v in pred.state_dict().items()}
        jsd = {k: v.detach().cpu().numpy() for k, v in joint.state_dict().items()}
        sample = 48
        t0 = time.perf_counter()
        RO.beam_decode(enc[:sample, :1].numpy(), np.array([sample]), psd, jsd, G.P, G.L, G.V, G.W, G.MS)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(sample / dt, 2), "unit": "frames/s (one utterance)", "cores": ctx["cores"], "kind": "port",
                               "gpu_frames_per_s": round(frames / ms_b * 1e3, 0),
                               "sample": f"beam-8 decode of the first {sample} frames of utterance 0 with the numpy oracle of the own "
                                         f"specification (oracle/rnnt_oracle.py), {dt:.1f} s"}
    return out


def build_stream_model():
    import bench
    return bench.build_model()


def leg_stream(ctx, chunks=16):
    """BASELINE configs[4]: chunked DS2 (320 ms = 32-frame chunks, state carried), batch 64, in the process's precision mode."""
    import bench
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2
    m = build_stream_model()
    N = 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, 1, 80, 32 * chunks, generator=g).cuda()
    lens = torch.full((N,), 32 * chunks, dtype=torch.int64)
    stream = ChunkedDeepSpeech2(m, 32)
    with torch.no_grad():
        ms_mean, ms_min = ev_timed(lambda: stream(x, lens), 4, 12)
        samples = sorted(LAST_SAMPLES)
        wall = wall_timed(lambda: stream(x, lens), 1, 5)
    # `ms` = the MEDIAN of 12 clips of 16 chunks: about one clip in fifteen takes 40 .. 80 ms longer (the caching allocator
    # growing its pool under the host's feet: a hipMalloc stall, as in bench.py's pipeline warm-up), which moved the mean of ten
    # by up to 30 % from run to run; the mean is reported beside it
    ms = 0.5 * (samples[len(samples) // 2 - 1] + samples[len(samples) // 2])
    per_chunk, per_chunk_wall = ms / chunks, wall / chunks
    mode = bench.precision_mode()
    bpe = {"f16x3": 4, "bf16x3": 4, "fp16": 2, "f32": 4}[mode]            # bytes per weight element as the kernels read it
    step_us = ctx.get("lstm_step_us_n64") or lstm_step_us(64)
    w_ih = [2 * 4096 * 640] + [2 * 4096 * 2048] * 4
    w_hh = 2 * 4096 * 1024
    w_bytes = sum((a + w_hh) * bpe for a in w_ih) + (2048 * 1024 + 1024 * 29) * 4
    floor = w_bytes / (HBM_GBS * 1e6) + 5 * 16 * step_us * 1e-3
    out = {"workload": f"cfg[4] chunked DS2 5xBiLSTM-1024, batch {N}, {chunks} chunks of 32 frames (320 ms), state carried",
           "dtype": mode, "ms_per_chunk": round(per_chunk, 4), "ms_per_chunk_min": round(ms_min / chunks, 4),
           "ms_per_chunk_mean": round(ms_mean / chunks, 4), "statistic": "median of 12 clips (mean and min beside it)",
           "ms_per_chunk_wall": round(per_chunk_wall, 4), "realtime_factor": round(N * 0.32 / per_chunk * 1e3, 1),
           "ms": round(per_chunk, 4), "floor_ms": round(floor, 4), "frac_of_floor": round(floor / per_chunk, 3),
           "hip_graph_replays": stream.graph_replays, "hip_graph_error": stream.graph_error,
           "floor": f"per chunk: {w_bytes / 1e6:.0f} MB of weights streamed once at 8 TB/s + 5 layers x 16 sequential steps x "
                    f"{step_us:.2f} us (the persistent recurrence's step at batch 64 in this mode, measured in this run); a BiLSTM "
                    "layer's projection needs the whole layer below, so the five recurrences are in series"}
    if ctx.get("cpu"):
        from oracle import ds_oracle as O
        from oracle import torch_cpu as TC
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        cfg = dict(convs=[dict(kind="conv2d", idx=0, stride=(2, 2), same=True, act=(0.0, 20.0)),
                          dict(kind="conv2d", idx=2, stride=(2, 1), same=True, act=(0.0, 20.0))],
                   rnn=dict(kind=O.LSTM, hidden=1024, layers=5, bidirectional=True), lookahead=None,
                   fc=dict(n_hidden=1, act=(0.0, 20.0)))
        xc = x[:, :, :, :32].cpu().numpy()
        lc = np.full(N, 32, dtype=np.int64)
        TC.deep_speech_2_forward(xc, lc, cfg, sd)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            TC.deep_speech_2_forward(xc, lc, cfg, sd)
        dt = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = {"value": round(dt * 1e3, 1), "unit": "ms per chunk", "cores": ctx["cores"], "kind": "port",
                               "realtime_factor": round(N * 0.32 / dt, 1),
                               "sample": f"{reps} passes of ONE 32-frame chunk of the 64 streams, stock torch CPU operators in the "
                                         "reference's order (zero initial state: the same arithmetic as a carried one)"}
    return out


def leg_stream_context(ctx, chunks=24):
    """The reference's SHIPPED architecture (configs/deep_speech_2_en.config:19-93: 2 x conv2d, 3 x GRU-2560 unidirectional, lookahead
    80, FC 1024) streamed in 320 ms chunks WITH carried convolution / lookahead context (``carry_context=True``: the chunks'
    outputs concatenated are the full-utterance logits), 32 streams.  Steady-state chunks only (the first six hold back the
    174-frame latency): HIP-graph replays of the captured push (``streaming._ContextGraph``); the last push flushes the
    held-back context eagerly."""
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    from myrtlespeech_amd.model.lookahead import Lookahead
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper
    from myrtlespeech_amd.streaming import ChunkedDeepSpeech2

    def act():
        return SeqLenWrapper(torch.nn.Hardtanh(0.0, 20.0), torch.nn.Identity())
    torch.manual_seed(7)
    cnn = torch.nn.Sequential(MaskConv2d(1, 32, [41, 11], [2, 2], PaddingMode.SAME), act(),
                              MaskConv2d(32, 32, [21, 11], [2, 1], PaddingMode.SAME), act())
    rnn = RNN(RNNType.GRU, 640, 2560, num_layers=3, bidirectional=False)
    la = torch.nn.Sequential(Lookahead(2560, 80), SeqLenWrapper(torch.nn.Identity(), torch.nn.Identity()))
    fc = FullyConnected(2560, 29, 1, 1024, torch.nn.Hardtanh(0.0, 20.0))
    m = DeepSpeech2(cnn, rnn, la, fc).eval()
    m.rnn.check_status = False
    N, chunk = 32, 32
    total = chunk * (chunks + 8)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(N, 1, 80, total, generator=g).cuda()
    lens = torch.full((N,), total, dtype=torch.int64)
    st = ChunkedDeepSpeech2(m, chunk, carry_context=True)
    lat = st.latency_frames()
    passes = []
    with torch.no_grad():
        for _ in range(3):       # the clip three times over: `ms` = the median pass (one hipMalloc stall in a 24-chunk pass is +3 ms per chunk)
            st.begin(lens, total)
            t0 = 0
            for _ in range(8):                                   # warm-up: fills the held-back context, reaches steady state
                st.push(x[..., t0:t0 + chunk])
                t0 += chunk
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0 = time.perf_counter()
            a.record()
            rows = 0
            for k in range(chunks):
                out = st.push(x[..., t0:t0 + chunk], final=(k == chunks - 1))
                rows += 0 if out is None else out.shape[0]
                t0 += chunk
            b.record()
            b.synchronize()
            passes.append((a.elapsed_time(b) / chunks, (time.perf_counter() - w0) / chunks * 1e3))
    ms, wall = sorted(passes)[1]
    step_us = 7.8                                               # persistent GRU-2560 at 32 rows (profiles/r01d)
    w_bytes = (3 * 3 * 2560 * (2560 + 2560) - 3 * 2560 * (2560 - 640)) * 4 + (2560 * 1024 + 1024 * 29) * 4
    floor = w_bytes / (HBM_GBS * 1e6) + 3 * 16 * step_us * 1e-3
    return {"workload": f"shipped DS2 (3xGRU-2560 + lookahead 80) streamed with carried context, {N} streams, {chunk}-frame chunks",
            "ms_per_chunk": round(ms, 4), "ms_per_chunk_wall": round(wall, 4), "ms": round(ms, 4),
            "ms_per_chunk_passes": [round(p_[0], 4) for p_ in passes], "statistic": "median of three passes over the clip",
            "realtime_factor": round(N * 0.32 / ms * 1e3, 1), "latency_frames": lat, "rows_per_chunk": round(rows / chunks, 2),
            "hip_graph_replays": st.graph_replays, "hip_graph_error": st.graph_error,
            "floor_ms": round(floor, 4), "frac_of_floor": round(floor / ms, 3),
            "floor": f"per chunk: {w_bytes / 1e6:.0f} MB of weights once at 8 TB/s + 3 layers x 16 dependent GRU steps x {step_us} us"}


def leg_frontend(ctx):
    from myrtlespeech_amd.data.preprocess import MFCC, MFCCLegacy, Standardize
    w = (torch.randn(32, 160000) * 0.1).cuda()
    wl = torch.full((32,), 160000)
    mf, sd = MFCC(n_mfcc=80, melkwargs={"win_length": 400, "hop_length": 160}), Standardize()

    def fe():
        y, fl = mf.batch(w, wl)
        return sd.batch(y, fl)
    ms, _ = ev_timed(fe, 3, 10)
    leg = MFCCLegacy(26, {"win_length": 400, "hop_length": 320})
    w4 = w[:, :64000].contiguous().clamp(-1, 1)
    ms_l, _ = ev_timed(lambda: leg.batch(w4, torch.full((32,), 64000)), 2, 5)
    return {"workload": "32 x 10 s @ 16 kHz -> MFCC(80, 400, 160) -> Standardize = [32,1,80,1001]",
            "ms": round(ms, 3), "audio_sec_per_s": round(320.0 / ms * 1e3, 0), "legacy_mfcc26_32x4s_ms": round(ms_l, 3)}


LEGS = {"ds1": ("cfg1_ds1", leg_ds1), "ctc": ("ctc_loss", leg_ctc_loss), "ctcgrad": ("ctc_loss_backward", leg_ctc_grad),
        "beam": ("ctc_beam_decode", leg_ctc_beam), "beamlm": ("ctc_beam_decode_lm", leg_ctc_beam_lm),
        "rnnt": ("cfg4_rnnt", leg_rnnt), "stream": ("cfg5_streaming", leg_stream), "streamctx": ("stream_carried_context", leg_stream_context),
        "frontend": ("frontend", leg_frontend)}


def shader_clock_under_projection_ghz():
    """The shader clock the chip sustains under the projection GEMM (16 032 x 8 192 x 2 048, the process's split mode), sampled
    by one wave on a stream of its own (ms_clock_probe: {100 MHz wall ticks, shader cycles} every 20 us; tools/clock_probe.py):
    the mean over 0.5 .. 4.5 ms of ~6 ms of back-to-back launches.  A box calibration figure: the GEMM is clock-capped, so a
    slow box shows here (and in lstm_step_us_*) while a regression of the kernels does not."""
    from myrtlespeech_amd import _lib
    lib = _lib.load()
    M, K, NN = 501 * 32, 2048, 8192
    xa = torch.randn(M, K, device="cuda")
    w = torch.randn(NN, K, device="cuda") * 0.02
    y = torch.empty(M, NN, device="cuda")
    ws = torch.empty(lib.ms_linear_split_workspace_bytes(M, K, NN), dtype=torch.uint8, device="cuda")
    samples, spacing = 300, 20
    buf = torch.zeros(2 * samples, dtype=torch.int64, device="cuda")
    side, probe = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(2):
        torch.cuda.synchronize()
        go = torch.cuda.Event()
        go.record()
        with torch.cuda.stream(probe):
            probe.wait_event(go)
            _lib.check(lib.ms_clock_probe(_lib.ptr(buf), samples, spacing, _lib.stream_ptr()), "ms_clock_probe")
        with torch.cuda.stream(side):
            side.wait_event(go)
            for _ in range(5):
                _lib.check(lib.ms_linear_split_forward(_lib.ptr(xa), _lib.ptr(w), None, _lib.ptr(y), M, K, NN, 0, 0.0, 0.0, _lib.ptr(ws),
                                                       ws.numel(), _lib.stream_ptr()), "ms_linear_split_forward")
        torch.cuda.synchronize()
    b = buf.cpu().view(samples, 2).double()
    dt = b[1:, 0] - b[:-1, 0]
    clk = (b[1:, 1] - b[:-1, 1]) / dt * 0.1
    t_ms = (b[1:, 0] - b[0, 0]) / 1e5
    busy = clk[(t_ms > 0.5) & (t_ms < 4.5)]
    return float(busy.mean()) if busy.numel() else float("nan")


def context(cpu=True, which=None):
    """The per-process calibration figures the floors are built from."""
    ctx = {"cpu": cpu, "cores": host_threads() if cpu else 0}
    ctx["barrier_step_us"] = barrier_step_us()
    try:
        ctx["shader_clock_under_gemm_ghz_us_"] = shader_clock_under_projection_ghz()     # ("_us_": kept by run_legs' calibration filter)
    except Exception as e:  # noqa: BLE001 -- a calibration figure, never a reason to lose the legs
        sys.stderr.write(f"bench_configs: shader clock calibration failed: {type(e).__name__}: {e}\n")
    which = which or set(LEGS)
    if "ds1" in which:
        ctx["lstm_step_us_n1"] = lstm_step_us(1, 201)
    if "stream" in which:
        ctx["lstm_step_us_n64"] = lstm_step_us(64, 501)
    return ctx


def run_legs(which, cpu=True):
    """{leg name: record}; a leg that raises is reported as {"error": ...} instead of taking the others with it."""
    out = {}
    try:
        ctx = context(cpu, set(which))
    except Exception as e:  # noqa: BLE001
        return {"error": f"calibration failed: {type(e).__name__}: {e}"[:300]}
    out["calibration"] = {k.replace("_ghz_us_", "_ghz"): round(v, 4) for k, v in ctx.items() if k.endswith("_us") or "_us_" in k}
    for key in which:
        name, fn = LEGS[key]
        try:
            out[name] = fn(ctx)
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.synchronize()
    return out


def main():
    import bench
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    which = args or list(LEGS)
    out = {"precision": bench.precision_label()}
    out.update(run_legs(which, cpu="--no-cpu" not in sys.argv))
    print(json.dumps(out, indent=None if "--line" in sys.argv else 1))


if __name__ == "__main__":
    main()
