"""Host-side logic added in round 2 that needs no GPU: the baton of the batches-in-flight pipeline, the staleness check of
the PMC profile that bench.py quotes, the PMC summary's per-K classification of the projection GEMM."""
import csv
import importlib.util
import json
import os
import subprocess
import sys
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_baton_alternates_round_robin_and_skips_finished_threads():
    from myrtlespeech_amd.pipeline import _Baton
    for n, segments in ((2, (3, 5)), (3, (4, 1, 6)), (2, (1, 1))):
        baton = _Baton(n)
        order = []

        def worker(me, count):
            baton.wait_turn(me)
            for s in range(count):
                order.append((me, s))          # only the thread that holds the baton appends
                baton.pass_on(me)
            baton.leave(me)

        threads = [threading.Thread(target=worker, args=(m, segments[m])) for m in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=30)
            assert not t.is_alive()
        # every segment ran exactly once, each thread's segments in order, and while several threads are active they alternate
        assert sorted(order) == sorted((m, s) for m in range(n) for s in range(segments[m]))
        left = list(segments)
        expect, turn = [], 0
        done = [0] * n
        while any(l > 0 for l in left):
            if left[turn] > 0:
                expect.append((turn, done[turn]))
                done[turn] += 1
                left[turn] -= 1
            turn = (turn + 1) % n
        assert order == expect


def test_pipeline_argument_checks_need_no_gpu():
    import torch
    from myrtlespeech_amd.pipeline import BatchesInFlight
    with pytest.raises(RuntimeError):          # no HIP device here: there is no CPU fallback
        BatchesInFlight(torch.nn.Identity())


def test_replica_shares_parameters_and_owns_its_caches():
    """``pipeline.replica``: the second stream's model holds the SAME Parameter objects (a checkpoint loaded into the caller's
    model afterwards reaches it) and its own workspaces / packed-weight caches."""
    import torch
    import bench
    from myrtlespeech_amd.pipeline import replica
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    cnn = torch.nn.Sequential(MaskConv2d(1, 4, [5, 3], [2, 2], PaddingMode.SAME))
    rnn = RNN(RNNType.LSTM, 4 * 10, 32, num_layers=2, bidirectional=True, forget_gate_bias=1.0)
    model = DeepSpeech2(cnn, rnn, None, FullyConnected(64, 29, 1, 16, torch.nn.Hardtanh(0.0, 20.0))).eval()
    rep = replica(model)
    a, b = dict(model.named_parameters()), dict(rep.named_parameters())
    assert a.keys() == b.keys() and all(a[k] is b[k] for k in a)
    assert rep is not model and rep.rnn is not model.rnn
    assert rep.rnn._workspace is not model.rnn._workspace
    assert all(p is not q for p, q in zip(rep.rnn._packed, model.rnn._packed))
    # the parameter container's flat weight list points at the shared objects too
    assert all(p is q for p, q in zip(rep.rnn.rnn._flat_weights, model.rnn.rnn._flat_weights))
    sd = {k: torch.full_like(v, 0.25) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    assert all(float(p.min()) == 0.25 == float(p.max()) for p in rep.parameters())
    assert bench.BATCH_PER_GPU == 32


def test_bench_refuses_a_stale_counter_profile(tmp_path, monkeypatch):
    import bench
    digests = bench.source_sha16()
    assert set(digests) >= {"rnn.hip", "gemm_split.hip", "common.h"}
    prof = {"precision": "f16x3", "source_sha16": dict(digests),
            "kernels": {"lstm_persistent_split2_kernel": {"hbm_bytes": 1, "mfma_flop": 2.0},
                        "gemm_nt_bf16x3_kernel4@K2048": {"hbm_bytes": 3, "mfma_flop": 4.0}}}
    path = tmp_path / "pmc.json"
    path.write_text(json.dumps(prof))
    monkeypatch.setattr(bench, "PMC_PROFILE", str(path))
    monkeypatch.delenv("MS_PRECISION", raising=False)
    rec, why = bench.pmc_record("lstm_persistent_split2_kernel")
    assert why is None and rec["hbm_bytes"] == 1
    assert bench.pmc_record("gemm_nt_bf16x3_kernel4@K2048")[0]["hbm_bytes"] == 3
    # another kernel's source changing does not invalidate this kernel's counters; its own source does
    prof["source_sha16"]["conv_cl.hip"] = "0" * 16
    prof["source_sha16"]["gemm_split.hip"] = "0" * 16
    path.write_text(json.dumps(prof))
    assert bench.pmc_record("lstm_persistent_split2_kernel")[1] is None
    rec, why = bench.pmc_record("gemm_nt_bf16x3_kernel4@K2048")
    assert rec is None and "gemm_split.hip" in why
    # counters of another precision mode are not quoted
    monkeypatch.setenv("MS_PRECISION", "f32")
    assert bench.pmc_record("lstm_persistent_split2_kernel")[0] is None
    monkeypatch.delenv("MS_PRECISION")
    # the committed profile describes the committed sources
    monkeypatch.setattr(bench, "PMC_PROFILE", os.path.join(ROOT, "profiles", "r06_pmc_bench.json"))
    assert bench.pmc_record("lstm_persistent_wide2_kernel@1group")[1] is None, "profiles/r06_pmc_bench.json is stale: re-run tools/pmc_bench.sh"
    assert bench.pmc_record("gemm_nt_bf16x3_kernel4@K2048")[1] is None


def test_pmc_summary_splits_the_projection_gemm_by_dispatch_order(tmp_path):
    d = tmp_path / "p1"
    d.mkdir()
    rows = []
    disp = 0
    for step in range(2):
        for layer in range(5):                     # full-grid launches: layer 1 (K = 640) then layers 2-5 (K = 2048)
            disp += 1
            rows.append(("void ms::gemm_nt_bf16x3_kernel4<false>(...)", disp, 1032192, "SQ_INSTS_MFMA", 100.0 if layer == 0 else 300.0))
        disp += 1
        rows.append(("void ms::gemm_nt_bf16x3_kernel4<false>(...)", disp, 129024, "SQ_INSTS_MFMA", 40.0))   # FC-sized launch
        disp += 1
        rows.append(("void (anonymous namespace)::lstm_persistent_split2_kernel<8, false, false, false>(...)", disp, 65536, "SQ_INSTS_MFMA", 7.0))
    with open(d / "p1_counter_collection.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatch_Id", "Grid_Size", "Counter_Name", "Counter_Value"])
        w.writerows(rows)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_bench_summary.py"), str(tmp_path), "bf16x3"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout)
    k = out["kernels"]
    assert k["gemm_nt_bf16x3_kernel4@K640"]["counters"]["SQ_INSTS_MFMA"] == 100.0
    assert k["gemm_nt_bf16x3_kernel4@K2048"]["counters"]["SQ_INSTS_MFMA"] == 300.0
    assert k["gemm_nt_bf16x3_kernel4@other"]["counters"]["SQ_INSTS_MFMA"] == 40.0
    assert k["gemm_nt_bf16x3_kernel4@K2048"]["mfma_flop"] == 300.0 * 2 * 32 * 32 * 16
    assert k["lstm_persistent_split2_kernel"]["mfma_flop"] == 7.0 * 2 * 16 * 16 * 32
    assert out["precision"] == "bf16x3" and "rnn.hip" in out["source_sha16"]
