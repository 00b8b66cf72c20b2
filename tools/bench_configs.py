#!/usr/bin/env python3
"""Timings for the BASELINE.json configs other than the headline one (which bench.py owns) and for the path
functions of SURVEY 8a that the headline does not exercise: DS1 single clip (cfg1), CTC loss, CTC beam decode,
RNN-T decode (cfg4), fp16 chunked streaming (cfg5, run with MS_PRECISION=fp16) and the feature front-end.
Run on the GPU box:  python tools/bench_configs.py > gpurun_out/configs.json
Prints one JSON object; each entry carries its workload, ms and the derived rate."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def timed(fn, warmup=2, iters=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    which = set(sys.argv[1:]) or {"ds1", "ctc", "beam", "rnnt", "stream", "frontend"}
    out = {"precision": bench.precision_label()}
    torch.manual_seed(0)

    if "ds1" in which:
        from myrtlespeech_amd.model.deep_speech_1 import DeepSpeech1
        from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
        m = DeepSpeech1(26, 19, 1024, 29, 0.25).eval()
        x = torch.randn(1, 19, 26, 201).cuda()
        lens = torch.tensor([201])
        dec = CTCGreedyDecoder(28)

        def run():
            (y, ol), _ = m((x, lens))
            return dec(y, ol)
        ms = timed(run)
        out["cfg1_ds1_single_clip"] = {"workload": "DS1 n_hidden 1024, 1 x 4 s clip [1,19,26,201], forward + greedy",
                                       "ms": round(ms, 3), "audio_sec_per_s": round(4.0 / ms * 1e3, 1)}

    if "ctc" in which:
        from myrtlespeech_amd.loss.ctc_loss import CTCLoss
        logits = torch.randn(501, 32, 29).cuda()
        lens = torch.full((32,), 501, dtype=torch.int32)
        tgt = torch.randint(0, 28, (32, 120), dtype=torch.int32)
        tl = torch.full((32,), 120, dtype=torch.int32)
        loss = CTCLoss(blank=28, reduction="sum")
        ms = timed(lambda: loss((logits, lens), (tgt, tl)))
        out["ctc_loss_forward"] = {"workload": "[501,32,29] logits, targets 32 x 120, reduction sum", "ms": round(ms, 3)}

    if "beam" in which:
        from myrtlespeech_amd.post_process.ctc_beam_decoder import CTCBeamDecoder
        probs = torch.softmax(torch.randn(501, 32, 29) * 12, dim=2).cuda()
        lens = torch.full((32,), 501, dtype=torch.int64)
        dec = CTCBeamDecoder(blank_index=28, beam_width=8)
        ms = timed(lambda: dec(probs, lens), warmup=1, iters=3)
        out["ctc_beam_w8"] = {"workload": "softmax(randn(501,32,29)*12), beam 8, prune 1e-3, no LM", "ms": round(ms, 2),
                              "utterances_per_s": round(32 / ms * 1e3, 1)}

    if "rnnt" in which:
        from myrtlespeech_amd.model.rnnt import RNNTJoint, RNNTPredictor
        from myrtlespeech_amd.post_process.rnnt_decoder import RNNTBeamDecoder, RNNTGreedyDecoder
        V = 28
        enc_model = bench.build_model()
        from myrtlespeech_amd.model.fully_connected import FullyConnected
        enc_model.fully_connected = FullyConnected(2048, 1024, 0, None, None)   # encoder projection 2048 -> 1024
        enc_model.eval()
        pred = RNNTPredictor(V, 256, 1024, num_layers=2).eval()
        joint = RNNTJoint(1024, 1024, 512, V).eval()
        x = torch.randn(16, 1, 80, 1001).cuda()
        lens = torch.full((16,), 1001, dtype=torch.int64)
        (enc, el), _ = enc_model((x, lens))
        g = RNNTGreedyDecoder(pred, joint, max_symbols=3)
        b = RNNTBeamDecoder(pred, joint, beam_width=8, max_symbols=3)
        ms_enc = timed(lambda: enc_model((x, lens)))
        ms_g = timed(lambda: g(enc, el), warmup=1, iters=2)
        labels_dense = [len(h) for h in g(enc, el)]
        ms_b = timed(lambda: b(enc, el), warmup=1, iters=2)
        # (a random-initialised joint makes the GREEDY search emit max_symbols labels on every frame: the worst case of the
        # event-driven greedy decode; transcripts as sparse as a trained transducer's: tools/rnnt_cfg4_time.py)
        out["cfg4_rnnt"] = {"workload": "DS2 encoder (batch 16 x 10 s) + 2-layer LSTM-1024 predictor + joint 512, 501 frames",
                            "encoder_ms": round(ms_enc, 2), "greedy_decode_ms": round(ms_g, 1),
                            "greedy_labels_per_utterance": [min(labels_dense), max(labels_dense)],
                            "beam8_decode_ms": round(ms_b, 1),
                            "audio_sec_per_s_beam8": round(160.0 / (ms_enc + ms_b) * 1e3, 1)}

    if "stream" in which:
        from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
        from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
        from myrtlespeech_amd.model.fully_connected import FullyConnected
        from myrtlespeech_amd.model.rnn import RNN, RNNType
        from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper
        from myrtlespeech_amd.streaming import ChunkedDeepSpeech2

        def act():
            return SeqLenWrapper(torch.nn.Hardtanh(0.0, 20.0), torch.nn.Identity())
        cnn = torch.nn.Sequential(MaskConv2d(1, 32, [41, 11], [2, 2], PaddingMode.SAME), act(),
                                  MaskConv2d(32, 32, [21, 11], [2, 1], PaddingMode.SAME), act())
        rnn = RNN(RNNType.LSTM, 640, 1024, num_layers=5, bidirectional=True, forget_gate_bias=1.0)
        fc = FullyConnected(2048, 29, 1, 1024, torch.nn.Hardtanh(0.0, 20.0))
        m = DeepSpeech2(cnn, rnn, None, fc).eval()
        N, chunks = 64, 8
        x = torch.randn(N, 1, 80, 32 * chunks).cuda()
        lens = torch.full((N,), 32 * chunks, dtype=torch.int64)
        stream = ChunkedDeepSpeech2(m, 32)
        ms = timed(lambda: stream(x, lens), warmup=1, iters=3)
        per_chunk = ms / chunks
        out["cfg5_streaming"] = {"workload": f"DS2 5xBiLSTM-1024, batch {N}, {chunks} chunks of 32 frames (320 ms), state carried",
                                 "ms_per_chunk": round(per_chunk, 3),
                                 "realtime_factor": round(N * 0.32 / per_chunk * 1e3, 1)}

    if "frontend" in which:
        from myrtlespeech_amd.data.preprocess import MFCC, MFCCLegacy, Standardize
        w = (torch.randn(32, 160000) * 0.1).cuda()
        wl = torch.full((32,), 160000)
        mf, sd = MFCC(n_mfcc=80, melkwargs={"win_length": 400, "hop_length": 160}), Standardize()

        def fe():
            y, fl = mf.batch(w, wl)
            return sd.batch(y, fl)
        ms = timed(fe)
        leg = MFCCLegacy(26, {"win_length": 400, "hop_length": 320})
        w4 = w[:, :64000].contiguous().clamp(-1, 1)
        ms_l = timed(lambda: leg.batch(w4, torch.full((32,), 64000)))
        out["frontend"] = {"workload": "32 x 10 s @ 16 kHz -> MFCC(80, 400, 160) -> Standardize = [32,1,80,1001]",
                           "ms": round(ms, 3), "audio_sec_per_s": round(320.0 / ms * 1e3, 0),
                           "legacy_mfcc26_32x4s_ms": round(ms_l, 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
