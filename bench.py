#!/usr/bin/env python3
"""Headline benchmark: DS2 (2x masked conv2d -> 5xBiLSTM-1024 -> FC 1024 -> 29) forward +
CTC greedy decode on synthetic 80-feature 10 s clips, batch 32 per GPU (BASELINE.json
configs[1]); utterances are sharded across GPUs (weak scaling, one process per GPU).

    python bench.py --gpus 1 --steps 100 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see the task contract): metric = audio-sec/s with inputs resident in HBM, plus

* ``roofline`` for the dominant kernel (the persistent LSTM recurrence): ``achieved`` / ``frac`` = SURVEY 8d's ALGORITHMIC
  bytes per launch / the launch duration measured live with HIP events on the launch stream / 8 TB/s (the north-star's
  definition; it re-reads ``W_hh`` every step although the kernel keeps it in registers, so it can exceed 1 and says
  nothing about headroom) AND what physically binds: ``hbm_frac_measured`` (PMC FETCH/WRITE bytes of this binary, from
  profiles/, / the live duration / 8 TB/s), ``mfma_frac`` (PMC MFMA instruction count x FLOP per instruction / live
  duration / 2.5 PFLOP/s) and ``bound`` -- for this kernel the cross-CU exchange of ``h`` each step;
* ``projection_gemm``: the same three figures for the second-largest kernel (``gemm_nt_bf16x3_kernel4`` at K = 2048);
* ``ragged_lengths``: the same step on lengths ~U[501, 1001] (sorted), BASELINE.md 3's second case;
* ``precision_f32``: the whole bench repeated by a child process in ``MS_PRECISION=f32`` (float32 MFMA everywhere -- the
  reference's own arithmetic width), started before this process touches the GPU (N = 1 only);
* ``cpu_baseline``: the reference's operator sequence on stock torch CPU operators on this box's host cores (N = 1 only).
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH_PER_GPU = 32
FRAMES = 1001          # 10 s @ 16 kHz, hop 160 (configs/deep_speech_2_en.config:5-11)
FEATURES = 80
CLIP_SECONDS = 10.0
HIDDEN = 1024
LAYERS = 5
VOCAB = 29
BLANK = 28
T_OUT = 501            # RNN steps after the two convolutions (time strides 2 and 1, SAME)
# algorithmic bytes of ONE LSTM layer-direction-timestep at H=1024, N=32, fp32:
#   W_hh 4H*H*4 + x-gates N*4H*4 + h read/write 2*N*H*4 + c read/write 2*N*H*4
LSTM_STEP_BYTES = 4 * HIDDEN * HIDDEN * 4 + BATCH_PER_GPU * 4 * HIDDEN * 4 + 4 * BATCH_PER_GPU * HIDDEN * 4
LSTM_STEP_FLOP = 2 * BATCH_PER_GPU * HIDDEN * 4 * HIDDEN            # 268.4 MFLOP per layer-direction-timestep
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 / fp16 MFMA
MFMA_F32_PEAK_TF = 157.3    # float32-input MFMA
PMC_PROFILE = os.path.join(ROOT, "profiles", "r06_pmc_bench.json")


def precision_mode():
    m = os.environ.get("MS_PRECISION")
    return m if m in ("f32", "fp16", "bf16x3") else "f16x3"


def split2(mode=None):
    """True in the two-plane split modes (three MFMA passes): f16x3 (default) and bf16x3."""
    return (mode or precision_mode()) in ("f16x3", "bf16x3")


def precision_label():
    """Arithmetic of the step, kept under 120 characters (the driver's record cuts strings there).  Default mode: every f32
    operand of the large contractions -- LSTM recurrence, input-projection GEMMs, conv1 (feature-window form), conv2
    (channels-last) and the hidden FC layer -- is split into fp16 hi + lo (22 mantissa bits; bf16 pairs, 16 bits, with
    MS_PRECISION=bf16x3: the default of rounds 1-5, which misses the 1e-3 gate on trained-scale weights) and multiplied as
    hi*hi + lo*hi + hi*lo with f32 accumulation (model/cnn.py, model/fully_connected.py route by size); the output layer
    (29 columns), CTC and the decoders are exact f32.  MS_PRECISION=f32 selects float32 MFMA everywhere."""
    mode = precision_mode()
    if mode == "f32":
        return "f32 (float32 MFMA everywhere; recurrent h crosses workgroups with its mantissa LSB as epoch tag)"
    if mode == "fp16":
        return "fp16 (one fp16 MFMA pass, f32 accumulate: LSTM, projections, conv2, FC1; optional mode outside the 1e-3 gate)"
    if mode == "bf16x3":
        return "bf16x3 (f32 split in bf16 hi+lo, 3 MFMA passes, f32 acc: LSTM, projections, conv1/2, FC1; FC2/CTC/decode f32)"
    return "f16x3 (f32 split in fp16 hi+lo, 3 fp16 MFMA passes, f32 acc: LSTM, projections, conv1/2, FC1; FC2/CTC/decode f32)"


KERNEL_SOURCES = {"lstm": ("rnn.hip", "common.h"), "gemm_nt_bf16x3": ("gemm_split.hip", "common.h"),
                  "gemm_nt_f32": ("gemm.hip", "common.h")}


def source_sha16(names=None):
    """Digests of the kernel source files the timed binary was built from ({file: sha16}); profiles/r06_pmc_bench.json
    records the digests its counters were taken on, so a stale counter file is detected instead of being quoted."""
    csrc = os.path.join(ROOT, "myrtlespeech_amd", "csrc")
    out = {}
    for f in sorted(names or (n for n in os.listdir(csrc) if n.endswith((".hip", ".h", ".cpp")))):
        with open(os.path.join(csrc, f), "rb") as fh:
            out[f] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


def pmc_record(kernel_key):
    """Per-launch PMC figures of `kernel_key` from the committed counter passes over THIS bench (tools/pmc_bench.sh), or
    (None, reason) when the file is missing, was taken in another precision mode or on other sources of that kernel."""
    try:
        with open(PMC_PROFILE) as f:
            prof = json.load(f)
    except Exception:
        return None, "profiles/r06_pmc_bench.json not found"
    if prof.get("precision") != precision_mode():
        return None, f"counters were taken in {prof.get('precision')} mode"
    files = next((v for k, v in KERNEL_SOURCES.items() if kernel_key.startswith(k)), None)
    now = source_sha16(files)
    if any(prof.get("source_sha16", {}).get(f) != d for f, d in now.items()):
        return None, f"{', '.join(now)} changed since the counter passes (profiles/r06_pmc_bench.json is stale for this kernel)"
    rec = prof.get("kernels", {}).get(kernel_key)
    return (rec, None) if rec else (None, f"no counters for {kernel_key}")


def build_model():
    from myrtlespeech_amd.model.cnn import MaskConv2d, PaddingMode
    from myrtlespeech_amd.model.deep_speech_2 import DeepSpeech2
    from myrtlespeech_amd.model.fully_connected import FullyConnected
    from myrtlespeech_amd.model.rnn import RNN, RNNType
    from myrtlespeech_amd.model.seq_len_wrapper import SeqLenWrapper

    def act():
        return SeqLenWrapper(torch.nn.Hardtanh(0.0, 20.0), torch.nn.Identity())

    torch.manual_seed(0)
    cnn = torch.nn.Sequential(MaskConv2d(1, 32, [41, 11], [2, 2], PaddingMode.SAME), act(),
                              MaskConv2d(32, 32, [21, 11], [2, 1], PaddingMode.SAME), act())
    rnn = RNN(RNNType.LSTM, 640, HIDDEN, num_layers=LAYERS, bidirectional=True, forget_gate_bias=1.0)
    fc = FullyConnected(2 * HIDDEN, VOCAB, 1, 1024, torch.nn.Hardtanh(0.0, 20.0))
    return DeepSpeech2(cnn, rnn, None, fc).eval()


def cpu_baseline(model, sample_batch=BATCH_PER_GPU):
    """The reference's CPU path is stock PyTorch; the reference package cannot travel, so the same torch CPU
    operators are re-assembled in the reference's order (oracle/torch_cpu.py, pinned against the reference's golden
    fixtures) and timed on this box's host cores on one pass over the same workload."""
    from oracle import ds_oracle as O
    from oracle import torch_cpu as TC
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    cfg = dict(convs=[dict(kind="conv2d", idx=0, stride=(2, 2), same=True, act=(0.0, 20.0)),
                      dict(kind="conv2d", idx=2, stride=(2, 1), same=True, act=(0.0, 20.0))],
               rnn=dict(kind=O.LSTM, hidden=HIDDEN, layers=LAYERS, bidirectional=True), lookahead=None,
               fc=dict(n_hidden=1, act=(0.0, 20.0)))
    rng = np.random.default_rng(0)
    x = rng.standard_normal((sample_batch, 1, FEATURES, FRAMES), dtype=np.float32)
    lens = np.full(sample_batch, FRAMES, dtype=np.int64)
    # the GPU box gives one GPU's job a share of 16 host cores: more torch threads than that only oversubscribe
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, share)))
    t0 = time.perf_counter()
    y, yl = TC.deep_speech_2_forward(x, lens, cfg, sd)
    TC.ctc_greedy_decode(y, yl, BLANK)
    dt = time.perf_counter() - t0
    return {"value": round(sample_batch * CLIP_SECONDS / dt, 2), "unit": "audio-sec/s", "cores": int(torch.get_num_threads()),
            "kind": "port", "sample": f"1 pass, {sample_batch} of 32 clips (10 s, full network, fp32), stock torch CPU ops, {dt:.1f} s wall",
            "sample_note": "the reference's operator sequence re-assembled from stock torch CPU operators (oracle/torch_cpu.py)"}


class GpuRuntime:
    """Everything ``main`` touches outside its own control flow (device, events, library handle, model, decoder, pipeline).
    The product runtime is this class; tests/test_bench_flow_cpu.py drives the same control flow -- legs, barriers, the
    MAX all-reduce, the ``--gather-logits`` branch -- on two gloo ranks with a CPU stand-in, so that the first real
    N > 1 run is not also the first execution of that code."""
    dist_backend = "nccl"
    device = "cuda"

    def set_device(self, local_rank):
        torch.cuda.set_device(local_rank)

    def init_process_group(self, dist, rank, world, local_rank):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(self.dist_backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def to_device(self, t):
        return t.cuda()

    def synchronize(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)

    def lib(self):
        from myrtlespeech_amd import _lib
        return _lib.load()

    def batch_per_rank(self, rank):
        return BATCH_PER_GPU

    def build_model(self, rank):
        model = build_model()
        model.rnn.check_status = False  # no per-layer host sync inside the timed region (checked once afterwards)
        return model

    def decoder(self):
        from myrtlespeech_amd.post_process.ctc_greedy_decoder import CTCGreedyDecoder
        return CTCGreedyDecoder(BLANK)

    def pipe(self, model, post, pre):
        from myrtlespeech_amd.pipeline import TwoBatchesInFlight
        return TwoBatchesInFlight(model, post=post, pre=pre)

    def paired(self, model, post, pre):
        from myrtlespeech_amd.pipeline import PairedBatches
        return PairedBatches(model, post=post, pre=pre)

    def check_status(self, models):
        from myrtlespeech_amd import _lib
        lib = _lib.load()
        for m in models:
            _lib.check(lib.ms_rnn_status(_lib.ptr(m.rnn._workspace.buf), _lib.stream_ptr()), "persistent LSTM")


def f32_child(args):
    """The fp32-arithmetic figure, timed by the driver's own run: a child process in MS_PRECISION=f32 (the mode is read
    once per process), started BEFORE this process makes its first GPU call, run to completion, its JSON line kept."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(max(10, args.steps // 2)), "--warmup",
           str(args.warmup), "--precision", "f32", "--no-cpu-baseline", "--no-f32-child", "--no-frontend", "--no-legs", "--detail-path", os.devnull]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        keep = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "latency_ms_per_batch", "encoder_ms", "decode_ms", "kernel_ms",
                "roofline", "projection_gemm", "stages", "timing")
        out = {k: d[k] for k in keep if k in d}
        # (the child prints the compact line: its one-batch figure is a flat scalar of `config`)
        out["one_batch_in_flight"] = {"ms_per_step": d.get("config", {}).get("one_batch_ms_per_step"),
                                      "value": d.get("config", {}).get("one_batch_value")}
        out["headline_mode"] = d.get("config", {}).get("headline_mode")
        return out
    except Exception as e:  # the headline must not die with the side measurement
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def fp16_stream_child():
    """BASELINE configs[4] names "fp16 MFMA": the streaming leg once more in a child with MS_PRECISION=fp16 (the mode is read
    once per process), started before this process makes its first GPU call."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), "stream", "--line", "--no-cpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MS_PRECISION"] = "fp16"
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return d.get("cfg5_streaming", {"error": "no cfg5_streaming record"})
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def gather_one_rank_child(args):
    """the collective path's cost on hardware in every round's record.  A child process -- started before
    this process touches the GPU -- runs the same bench as ONE rank with the collective path forced (MS_FORCE_COLLECTIVE=1:
    RCCL init, the all-gather of the padded logits block, batched decode of the gathered batch) and `--gather-logits`.
    One rank is all this pool can give: no statement about scaling."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(max(10, args.steps // 2)), "--warmup",
           str(args.warmup), "--gather-logits", "--no-cpu-baseline", "--no-f32-child", "--no-frontend", "--no-legs", "--no-ragged",
           "--reps", "0", "--detail-path", os.devnull]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               MS_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        c = d.get("config", {})
        return {"ms": d.get("ms_per_step"), "value": d.get("value"), "one_batch_ms": c.get("one_batch_ms_per_step"),
                "headline_mode": c.get("headline_mode"), "decode": c.get("decode"), "backend": "nccl (RCCL), world 1, collective forced",
                "steps": d.get("steps")}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def compact(rec, keys):
    return {k: rec[k] for k in keys if isinstance(rec, dict) and k in rec}


LINE_LIMIT = 8192          # bytes: the driver's parser lost a 22.9 KB line in round 4
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")
# keys of the full record that are prose or nested detail: they live in bench_detail.json (and on stderr), not in the line
LINE_DROP = ("legs_detail", "one_batch_in_flight", "two_batches_in_flight", "two_batches_per_forward", "precision_f32",
             "ragged_lengths", "parity", "what_binds", "definition", "launch_ms_source", "note", "latency_note", "floor",
             "two_in_flight_note", "pmc_source", "pipeline", "kernel_note", "sample_note", "chain_floor")
# dropped, in this order, only if the line would still exceed LINE_LIMIT (it does not today: ~4 KB)
LINE_OPTIONAL = ("stages", "projection_gemm", "kernel_ms", "timing")


def shrink(obj, maxlen=120):
    """The record without its prose: LINE_DROP keys removed at every depth, every remaining string cut to `maxlen`."""
    if isinstance(obj, dict):
        return {k: shrink(v, maxlen) for k, v in obj.items() if k not in LINE_DROP}
    if isinstance(obj, (list, tuple)):
        return [shrink(v, maxlen) for v in obj]
    if isinstance(obj, str) and len(obj) > maxlen:
        return obj[:maxlen - 1] + "~"
    if isinstance(obj, float) and (obj != obj or obj in (float("inf"), float("-inf"))):
        return None          # strict JSON: no NaN / Infinity tokens in the line
    return obj


def bench_line(full, detail_name):
    """The ONE stdout line: standard keys, flat `config`, numeric `roofline` / `projection_gemm` / `cpu_baseline` / `legs`;
    strict JSON and under LINE_LIMIT bytes whatever the detail record grows to."""
    line = shrink(full)
    line["detail"] = detail_name
    legs = line.pop("legs", None)
    if legs is not None:
        line["legs"] = legs          # stays the LAST key: a record that keeps only the tail of stdout still has the numbers
    for k in ("",) + LINE_OPTIONAL:
        line.pop(k, None)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
        if len(text.encode()) < LINE_LIMIT:
            return text
    for k in [k for k in line if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                          "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")][::-1]:
        line.pop(k)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
        if len(text.encode()) < LINE_LIMIT:
            return text
    raise RuntimeError(f"the standard keys alone exceed {LINE_LIMIT} bytes")


def spawn_ranks(argv, n):
    """`python bench.py --gpus N` without torchrun: start the N ranks as ONE child --
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` -- BEFORE this process makes any GPU call, relay rank 0's
    line, exit with the child's code.  Never a re-exec: this process stays the parent and does not touch the device."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return r.returncode if (r.returncode or lines) else 1


def main(argv=None, runtime=None, json_fd=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-child", action="store_true", help="skip the MS_PRECISION=f32 child run")
    ap.add_argument("--no-frontend", action="store_true", help="skip the waveform -> MFCC side measurement")
    ap.add_argument("--no-ragged", action="store_true", help="skip the ragged-length leg")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the other BASELINE configs / BASELINE.md 3 legs (DS1, CTC loss, CTC beam, RNN-T, streaming)")
    ap.add_argument("--in-flight", type=int, choices=[1, 2], default=2,
                    help="batches in flight per GPU for the headline figure: 2 (default) = myrtlespeech_amd.pipeline."
                         "TwoBatchesInFlight (two streams, the next batch's projection GEMMs run beside this batch's "
                         "persistent recurrence; bit-identical results, per-batch latency of two batches); 1 = one batch "
                         "at a time.  The one-batch leg is always timed too (kernel durations / rooflines come from it)")
    ap.add_argument("--precision", choices=["f16x3", "bf16x3", "f32", "fp16"], default=None,
                    help="operand mode of the recurrence / projection kernels (default: MS_PRECISION or f16x3); "
                         "f32 = float32 MFMA everywhere")
    ap.add_argument("--gather-logits", action="store_true",
                    help="batched-decode path: all-gather every shard's logits (RCCL over xGMI) and decode the whole "
                         "global batch on every rank instead of decoding per shard")
    ap.add_argument("--detail-path", default=DETAIL_FILE,
                    help="where rank 0 writes the full record (prose, per-stage tables, every leg's detail); the stdout line "
                         "names it in `detail` and carries numbers only")
    ap.add_argument("--runtime", default=None,
                    help="module:factory of a stand-in for GpuRuntime (factory(rank) -> runtime); the CPU tests of the "
                         "self-spawned multi-rank path use it, nothing else should")
    ap.add_argument("--reps", type=int, default=None,
                    help="extra timed regions of exactly --steps steps per headline leg, reported as min / median / max beside "
                         "the first region's mean (default: 4 more when a region lasts under 1 s, else none)")
    args = ap.parse_args(argv)
    if args.precision is not None:   # read once by the library at its first launch
        os.environ["MS_PRECISION"] = args.precision
    if args.gpus > 1 and "RANK" not in os.environ and runtime is None:
        sys.exit(spawn_ranks(sys.argv[1:] if argv is None else argv, args.gpus))
    if runtime is None and args.runtime:
        import importlib
        mod, _, fac = args.runtime.partition(":")
        runtime = getattr(importlib.import_module(mod), fac)(int(os.environ.get("RANK", "0")))
    rt = runtime or GpuRuntime()

    # Anything the runtime libraries print (RCCL's banner goes to stdout) is sent to stderr, so that stdout carries
    # the ONE JSON line and nothing else.
    if json_fd is None:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    f32 = stream_fp16 = None
    if world == 1 and "RANK" not in os.environ and precision_mode() == "f16x3" and not args.no_f32_child:
        f32 = f32_child(args)          # before the first GPU call of this process
    gather1 = None
    if world == 1 and "RANK" not in os.environ and precision_mode() == "f16x3" and not args.no_legs:
        stream_fp16 = fp16_stream_child()
        gather1 = gather_one_rank_child(args)

    dist = None
    if world > 1 or "RANK" in os.environ:  # under torchrun (also with one rank) the collective path is exercised
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rt.set_device(local_rank)
        rt.init_process_group(dist, rank, world, local_rank)
        from myrtlespeech_amd import parallel
        parallel.init_host_group()      # gloo twin for host metadata: collective over the world, so at set-up time
    else:
        rt.set_device(0)

    lib = rt.lib()
    model = rt.build_model(rank)
    decoder = rt.decoder()

    # this rank's shard of the global batch: 32 utterances, resident in HBM before timing starts
    g = torch.Generator().manual_seed(1234 + rank)
    n_rank = rt.batch_per_rank(rank)
    x = rt.to_device(torch.randn(n_rank, 1, FEATURES, FRAMES, generator=g))
    lens_full = torch.full((n_rank,), FRAMES, dtype=torch.int64)
    lens_ragged = torch.sort(torch.randint(501, FRAMES + 1, (n_rank,), generator=g), descending=True).values

    def step_launch(lens, ev=None):
        """Enqueue one whole step (encoder forward, greedy decode, copy of the transcripts to pinned host memory) on the
        current stream; returns the pending transcripts.  The masked convolution zeroes its input past each length IN PLACE
        (cnn.py:442), so the ragged leg gets its own copy of the batch (made outside the timed region) and the full-length
        leg's input stays intact."""
        e0 = e1 = e2 = None
        if ev is not None:
            e0, e1, e2 = (rt.event() for _ in range(3))
            e0.record()
        (logits, out_lens), _ = model((x if lens is lens_full else x_ragged, lens))
        if ev is not None:
            e1.record()
        if args.gather_logits and dist is not None:
            from myrtlespeech_amd.parallel import gather_logits
            logits, out_lens = gather_logits(logits, out_lens)
        pending = decoder.launch(logits, out_lens)
        if ev is not None:
            e2.record()
            ev.append((e0, e1, e2))
        return pending

    def step(lens, ev=None):
        return step_launch(lens, ev).result()

    def set_overlap(on):
        """Switch the recurrent stack between its two schedules (same bits): layer by layer, and the overlapped one the library
        uses by default for one batch of <= 32 utterances (myrtlespeech_amd/model/rnn.py, ms_rnn_stack_forward).  Returns whether
        the overlapped schedule exists for this model in this mode."""
        try:
            from myrtlespeech_amd.model import rnn as _R
        except Exception:  # noqa: BLE001
            return False
        if os.environ.get("MS_RNN_OVERLAP") == "0" or precision_mode() == "f32":
            return False
        _R._OVERLAP = bool(on)
        return True

    def barrier():
        if dist is not None:
            dist.barrier()
        rt.synchronize()

    # Throughput modes.  With --gather-logits (batched decode behind an RCCL all-gather) the two-batches-per-forward mode
    # stays on: its `post` runs on the calling thread in batch order, so every rank issues the collectives in the same order.
    # The threaded two-in-flight pipeline calls `post` from two worker threads, whose interleaving may differ between ranks
    # (a collective order mismatch = a hang), so it is not timed under --gather-logits.
    gathering = args.gather_logits and dist is not None
    pipelined = args.in_flight == 2
    pipe = paired = None
    if pipelined:
        starts = {}

        def pre(k):
            starts[k] = rt.event()
            starts[k].record()

        def post(out):
            logits, out_lens = out[0][0], out[0][1]
            if gathering:        # this batch's shard of logits from every rank, then the batched decode on every rank
                from myrtlespeech_amd.parallel import gather_logits
                logits, out_lens = gather_logits(logits, out_lens)
            pending = decoder.launch(logits, out_lens)
            end = rt.event()
            end.record()
            return pending, end

        pipe = None if gathering else rt.pipe(model, post, pre)
        paired = rt.paired(model, post, pre)

    STAGES = ("projection", "recurrence", "gemm_k_large", "gemm_k_small", "conv", "layout", "linear", "greedy", "other")

    def timed(lens, steps, two, run_ahead=True, prof=True):
        """`steps` passes bracketed by barrier + synchronize on both sides; MAX over ranks; in-library HIP-event spans of every
        kernel family.  One batch at a time (`two` False): the host enqueues step k+1 BEFORE it collects step k's transcripts
        (`run_ahead`; one stream, so the device still runs one batch after the other and never idles while the host builds
        the lists) -- every step's transcripts are on the host before the clock stops either way."""
        ev = []
        ms = (ctypes.c_float * len(STAGES))()
        cnt = (ctypes.c_int * len(STAGES))()
        # (prof=False: no in-library event spans in this region -- a span is two event records on the stream, and the overlapped
        # one-batch schedule has ~75 of them per step where the layer-by-layer one has ~30)
        lib.ms_prof_enable(0 if (not prof or (two and os.environ.get('BENCH_PROF_TWO') == '0')) else 1)
        lib.ms_prof_read(ms, cnt)       # drop spans recorded so far
        barrier()
        t0 = time.perf_counter()
        if two:
            runner = two if callable(two) else pipe          # `two` = the throughput mode's runner (True: the threaded pipeline)
            pend = runner([(x if lens is lens_full else x_ragged, lens)] * steps)
            for pd, _ in pend:
                pd.result()                      # every step's transcripts are on the host before the clock stops
        else:
            prev = None
            for _ in range(steps):
                cur = step_launch(lens, ev)
                if not run_ahead:
                    cur.result()
                elif prev is not None:
                    prev.result()
                prev = cur
            prev.result()
        rt.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        lib.ms_prof_read(ms, cnt)
        lib.ms_prof_enable(0)
        rt.check_status(list(pipe.models) if (pipe is not None and (two is True or two is pipe)) else [model])
        t_max = torch.tensor([elapsed], dtype=torch.float64, device=rt.device)
        if dist is not None:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        r = {"elapsed": float(t_max.item()), "steps": steps,
             "per_launch": {k: ms[i] / max(cnt[i], 1) for i, k in enumerate(STAGES)},
             "per_step": {k: ms[i] / steps for i, k in enumerate(STAGES)},
             "launches_per_step": {k: cnt[i] / steps for i, k in enumerate(STAGES)}}
        if ev:
            r["enc"] = sum(a.elapsed_time(b) for a, b, _ in ev) / len(ev)
            r["dec"] = sum(b.elapsed_time(c) for _, b, c in ev) / len(ev)
            r["span"] = sum(a.elapsed_time(c) for a, _, c in ev) / len(ev)          # first kernel of a step -> its copy enqueued
            # device time between one step's last event and the next step's first: the device waiting for the host
            r["between"] = (sum(ev[i][2].elapsed_time(ev[i + 1][0]) for i in range(len(ev) - 1)) / (len(ev) - 1)) if len(ev) > 1 else 0.0
        if two:   # per-batch latency: first launch of the batch -> its transcripts' copy enqueued and done
            r["latency"] = sum(starts[k].elapsed_time(end) for k, (_, end) in enumerate(pend)) / len(pend)
        return r

    def stage_report(r, overlapped):
        """Per-step device time by kernel family (the nested GEMM-only spans are inside `projection`), what of the step's
        device span is NOT inside any family's span (`gpu_idle_ms_per_step`: launch gaps between dependent kernels, torch's
        own small kernels for the transcripts' packing) and the device time between two steps (`between_steps_ms`)."""
        top = ("conv", "layout", "projection", "recurrence", "linear", "greedy", "other")
        out = {"stage_ms": {k: round(r["per_step"][k], 4) for k in top},
               "stage_launches_per_step": {k: round(r["launches_per_step"][k], 2) for k in top},
               "stage_sum_ms": round(sum(r["per_step"][k] for k in top), 4)}
        if overlapped:
            out["note"] = ("two batches in flight: the spans of the two streams overlap in time (a projection GEMM runs beside "
                           "the other batch's recurrence), so the sum exceeds the wall time per step by design")
            out["wall_ms_per_step"] = round(r["elapsed"] / r["steps"] * 1e3, 4)
        else:
            out["event_span_ms"] = round(r["span"], 4)
            out["gpu_idle_ms_per_step"] = round(r["span"] - out["stage_sum_ms"], 4)
            out["between_steps_ms"] = round(r["between"], 4)
            out["wall_ms_per_step"] = round(r["elapsed"] / r["steps"] * 1e3, 4)
        return out

    x_ragged = x.clone()
    for _ in range(args.warmup):
        step(lens_full)
    # leg 1: one batch at a time, LAYER BY LAYER (projection, then recurrence: every kernel alone on the device -- kernel
    # durations, rooflines, the stage table, the encoder / decode split)
    overlap_available = set_overlap(False)
    one = timed(lens_full, args.steps, False)
    one_serial_elapsed, enc_ms, dec_ms = one["elapsed"], one["enc"], one["dec"]
    set_overlap(True)
    # leg 1a: the literal batch-32 step as the library runs it by default: the overlapped stack schedule (round 6:
    # ms_rnn_stack_forward -- layer l+1's projection beside layer l's recurrence; same bits), spans off
    one_ov = None
    if overlap_available:
        try:
            for _ in range(2):
                step(lens_full)
            one_ov = timed(lens_full, args.steps, False, prof=False)
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"bench: overlapped one-batch leg failed: {type(e).__name__}: {e}\n")
            rt.synchronize()
    one_elapsed = one_ov["elapsed"] if one_ov is not None else one_serial_elapsed
    # leg 1b (short): the same with the host collecting every step's transcripts before it issues the next step -- the
    # difference to leg 1 is the device time the host's read-back and list building would otherwise expose
    sync_each = timed(lens_full, max(5, args.steps // 4), False, run_ahead=False)
    # leg 2 (headline when --in-flight 2): two batches in flight.  A failure here must not cost the run its headline: the
    # one-batch figure (already measured) is reported instead, with the reason.
    pipeline_error = None
    two = None
    elapsed = latency_ms = None
    if pipelined and pipe is not None:
        try:
            # untimed: the pipeline's own warm-up.  A call keeps every batch's outputs until the caller has collected them, so
            # the caching allocator only reaches its steady state after a call of the timed call's length: with a shorter
            # warm-up the first timed call still grows the pool (a 40 .. 66 ms stall of hipMalloc inside it, tools/pipe_timeline.py)
            pipe([(x, lens_full)] * max(2, args.warmup, min(args.steps, 24)))
            two = timed(lens_full, args.steps, True)
            elapsed, latency_ms = two["elapsed"], two["latency"]
        except Exception as e:  # noqa: BLE001
            pipeline_error = f"{type(e).__name__}: {e}"[:300]
            pipe = None
            two = None
            lib.ms_gemm_set_variant(0)
            rt.synchronize()
    two_in_flight = None
    if two is not None:
        two_in_flight = {"value": round(world * BATCH_PER_GPU * CLIP_SECONDS * args.steps / elapsed, 1),
                         "ms_per_step": round(elapsed / args.steps * 1e3, 3), "latency_ms_per_batch": round(latency_ms, 3)}
        two_in_flight.update(stage_report(two, True))
    # leg 3 (the other throughput mode, round 3): two batches per forward (pipeline.PairedBatches -> one stream, the two
    # batches' recurrences side by side in one launch of the wide-workgroup kernel)
    pair = per_forward = None
    headline = pipe if two is not None else None          # the runner whose figure is `value`
    pipelined = headline is not None
    if paired is not None:
        try:
            paired([(x, lens_full)] * max(2, args.warmup, min(args.steps, 24)))
            pair = timed(lens_full, args.steps, paired)
            per_forward = {"value": round(world * BATCH_PER_GPU * CLIP_SECONDS * args.steps / pair["elapsed"], 1),
                           "ms_per_step": round(pair["elapsed"] / args.steps * 1e3, 3),
                           "latency_ms_per_batch": round(pair["latency"], 3)}
            rep = stage_report(pair, True)
            rep["note"] = ("two batches per forward on ONE stream: every kernel of a step runs once on 64 utterances, the "
                           "recurrence as one launch of the wide-workgroup kernel for both batches; stage_ms is per batch of 32")
            per_forward.update(rep)
            if headline is None or pair["elapsed"] < elapsed:
                headline, elapsed, latency_ms, pipelined = paired, pair["elapsed"], pair["latency"], True
        except Exception as e:  # noqa: BLE001
            per_forward = {"error": f"{type(e).__name__}: {e}"[:300]}
            rt.synchronize()
    if pipelined and elapsed >= one_elapsed:   # few steps: fill and drain outweigh what a pipeline hides; the headline is the faster leg
        pipelined, headline = False, None
    if not pipelined:
        elapsed, latency_ms = one_elapsed, None

    # more regions of exactly K steps for the two figures people quote (20 steps = 0.24 s cannot tell a slow
    # box from a regression): `value` stays the FIRST region's mean; min / median / max over all regions go beside it
    reps = args.reps if args.reps is not None else (4 if one_elapsed < 1.0 else 0)
    spread = {"regions": 1 + reps, "steps_per_region": args.steps}

    def stats(first_ms, more):
        v = sorted([first_ms] + more)
        return {"min": round(v[0], 3), "median": round(v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]), 3),
                "max": round(v[-1], 3)}

    def more_regions(runner):
        """`reps` further K-step regions of a leg; a failure in one of them degrades the spread to the regions that ran (the
        first region -- `value` -- is already in hand) instead of losing the whole line."""
        out = []
        for _ in range(reps):
            try:
                out.append(timed(lens_full, args.steps, runner, prof=bool(runner))["elapsed"] / args.steps * 1e3)
            except Exception as e:  # noqa: BLE001
                spread["error"] = f"{type(e).__name__}: {e}"[:200]
                rt.synchronize()
                break
        return out

    # both legs carry both statistics under parallel keys: the first region's mean (`value` / ms_per_step for the headline,
    # config.one_batch_first_region_ms_per_step for the one-batch leg) and min / median / max over all regions
    spread["one_batch_ms_per_step"] = stats(one_elapsed / args.steps * 1e3, more_regions(False))
    if pipelined:
        spread["headline_ms_per_step"] = stats(elapsed / args.steps * 1e3, more_regions(headline))
    else:
        spread["headline_ms_per_step"] = spread["one_batch_ms_per_step"]

    ragged = None
    if not args.no_ragged:
        rsteps = max(6, args.steps // 2)
        for _ in range(2):
            step(lens_ragged)
        # both throughput modes (the recurrence takes max(lens) steps whatever the lengths; what shrinks with them is the
        # projection, which works on sum(lens) rows -- MS_RNN_PACKED_ROWS -- so the mode that runs projection and
        # recurrence one after the other gains, the one that hides the projection behind a recurrence does not)
        r_modes = {}
        if pipelined:
            for name, runner in (("two_batches_in_flight", pipe if two_in_flight is not None else None),
                                 ("two_batches_per_forward", paired if pair is not None else None)):
                if runner is None:
                    continue
                runner([(x_ragged, lens_ragged)] * 2)
                r_modes[name] = timed(lens_ragged, rsteps, runner)["elapsed"]
        r_elapsed = min(r_modes.values()) if r_modes else timed(lens_ragged, rsteps, False)["elapsed"]
        audio_s = float(lens_ragged.sum()) * CLIP_SECONDS / (FRAMES - 1) * world   # hop 10 ms
        ragged = {"workload": "same batch, lengths ~U[501, 1001] frames sorted in decreasing order (BASELINE.md 3 (ii)); "
                              "the backward direction of every utterance starts at its own last frame",
                  "in_flight": 2 if pipelined else 1,
                  "ms_per_step_by_mode": {k: round(v / rsteps * 1e3, 3) for k, v in r_modes.items()},
                  "steps": rsteps, "ms_per_step": round(r_elapsed / rsteps * 1e3, 3),
                  "audio_seconds_per_step": round(audio_s, 1),
                  "value": round(audio_s * rsteps / r_elapsed, 1), "unit": "audio-sec/s (real, unpadded audio)",
                  "padded_value": round(world * BATCH_PER_GPU * CLIP_SECONDS * rsteps / r_elapsed, 1)}

    frontend_ms = None
    if rank == 0 and not args.no_frontend:
        # side measurement, NOT part of `value` (the metric starts at feature tensors resident in HBM): the same batch
        # from 16 kHz waveforms -- MFCC(80, win 400, hop 160) + Standardize of the shipped DS2 config on the device
        from myrtlespeech_amd.data.preprocess import MFCC, Standardize
        waves = rt.to_device(torch.randn(BATCH_PER_GPU, 160000, generator=g) * 0.1)
        wl = torch.full((BATCH_PER_GPU,), 160000)
        mfcc, std = MFCC(n_mfcc=FEATURES, melkwargs={"win_length": 400, "hop_length": 160}), Standardize()
        for i in range(8):
            if i == 3:
                rt.synchronize()
                tf0 = time.perf_counter()
            std.batch(*mfcc.batch(waves, wl))
        rt.synchronize()
        frontend_ms = (time.perf_counter() - tf0) / 5 * 1e3

    if rank == 0:
        mode = precision_mode()
        ms_per_step = elapsed / args.steps * 1e3
        value = world * BATCH_PER_GPU * CLIP_SECONDS * args.steps / elapsed
        one_ms = one_elapsed / args.steps * 1e3
        pl = one["per_launch"]
        proj_ms, rec_ms, gemm_k2048_ms, gemm_k640_ms = pl["projection"], pl["recurrence"], pl["gemm_k_large"], pl["gemm_k_small"]
        two_spans = None
        if two is not None:
            tl = two["per_launch"]
            two_spans = (tl["projection"], tl["recurrence"], tl["gemm_k_large"], tl["gemm_k_small"])
        # ---- dominant kernel: the persistent recurrence (one launch = one layer, both directions, 501 steps)
        launch_bytes = T_OUT * 2 * LSTM_STEP_BYTES
        achieved = launch_bytes / (rec_ms * 1e-3) / 1e9 if rec_ms > 0 else 0.0
        wide = split2(mode) and os.environ.get("MS_LSTM_WIDE") != "0"
        kname = {"f32": "lstm_persistent_kernel" if os.environ.get("MS_LSTM_F32_ONE_STREAM") == "1"
                 else "lstm_persistent_f32x2_kernel"}.get(mode, "lstm_persistent_wide2_kernel" if wide else "lstm_persistent_split2_kernel")
        pmc_key = kname + ("@1group" if wide else "")
        mfma_peak = MFMA_F32_PEAK_TF if mode == "f32" else MFMA_BF16_PEAK_TF
        roof = {"bound": "hbm",   # the north-star's roofline for the LSTM step; what physically binds: next key
                "what_binds": "the cross-workgroup exchange of h in series with the step's arithmetic: every stream-step (16 rows) "
                              "each workgroup publishes its 16 hidden units and pulls the whole h of its direction (64 KB) -- "
                              "0.95 us at the ~67 GB/s a CU ingests when every CU of its XCD pulls the same rows, plus ~0.75 us of "
                              "MFMA tail, barrier, cell and publish that cannot overlap it (a publish is answered fresh only ~0.8 us "
                              "later; asking earlier, more row streams, a prefetch a stream-step ahead were all measured: "
                              "profiles/r03ag..r03ao, tools/micro/exchange_wide.hip reproduces the 1.7 us with sleeps for the "
                              "arithmetic).  16 units per workgroup leave half of the CUs to a second batch "
                              "(two_batches_per_forward) or to the other batch's projection GEMM (two_batches_in_flight); "
                              "neither HBM nor MFMA",
                "binds": "cross-workgroup h exchange (64 KB pull per stream-step) + serial MFMA tail, cell, publish; neither HBM nor MFMA",
                "kernel": kname,
                "kernel_note": ("one launch = 1 layer x 2 directions x 501 steps of one batch: 128 workgroups of 16 hidden units on "
                                "half of the CUs" if wide else "one launch = 1 layer x 2 directions x 501 steps"),
                "launch_ms": round(rec_ms, 4),
                "launch_ms_source": "HIP events on the launch stream, mean over the timed steps of the one-batch-in-flight leg "
                                    "(the kernel alone on the device)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "algorithmic_frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": launch_bytes,
                "definition": "frac = SURVEY 8d algorithmic bytes (17 825 792 B x 1002 layer-direction-steps, W_hh counted "
                              "once per step although it stays in registers) / launch_ms / 8 TB/s: the north-star's "
                              "roofline figure, not a physical utilisation"}
        if two_spans is not None:
            roof["launch_ms_two_in_flight"] = round(two_spans[1], 4)
            roof["frac_two_in_flight"] = round(launch_bytes / (two_spans[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if two_spans[1] > 0 else None
        if pair is not None and pair["per_launch"]["recurrence"] > 0:
            # one launch of lstm_persistent_wide2_kernel = 1 layer x 2 directions x 501 steps of TWO batches of 32
            pr = pair["per_launch"]["recurrence"]
            roof["two_batches_per_forward"] = {
                "kernel": ("lstm_persistent_wide2_kernel (16 hidden units per workgroup, two 32-row batch groups side by side)" if wide
                           else kname + " (this mode has no wide-workgroup form: two launches, one per 32-row batch group)"),
                "launch_ms": round(pr, 4), "algorithmic_bytes_per_launch": 2 * launch_bytes,
                "achieved": round(2 * launch_bytes / (pr * 1e-3) / 1e9, 1), "frac": round(2 * launch_bytes / (pr * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            wrec, wwhy = pmc_record("lstm_persistent_wide2_kernel@2groups")
            if wrec is not None:
                roof["two_batches_per_forward"].update({
                    "traffic": int(wrec["hbm_bytes"]),
                    "hbm_frac_measured": round(wrec["hbm_bytes"] / (pr * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "mfma_frac": round(wrec["mfma_flop"] / (pr * 1e-3) / 1e12 / mfma_peak, 4),
                    "mfma_busy_frac_pmc": wrec.get("mfma_busy_frac"), "l2_hit_rate_pmc": wrec.get("l2_hit_rate")})
            else:
                roof["two_batches_per_forward"]["pmc_note"] = wwhy
        rec, why = pmc_record(pmc_key)
        if rec is not None and rec_ms > 0:
            roof["traffic"] = int(rec["hbm_bytes"])
            roof["hbm_frac_measured"] = round(rec["hbm_bytes"] / (rec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roof["mfma_frac"] = round(rec["mfma_flop"] / (rec_ms * 1e-3) / 1e12 / mfma_peak, 4)
            roof["mfma_busy_frac_pmc"] = rec.get("mfma_busy_frac")
            roof["l2_hit_rate_pmc"] = rec.get("l2_hit_rate")
            roof["pmc_source"] = "profiles/r06_pmc_bench.json (rocprofv3 --pmc passes over this bench, same kernel sources)"
        else:
            roof["traffic"] = None
            roof["pmc_note"] = why
        # ---- second kernel: the input-projection GEMM at K = 2048 (layers 1..4): 3 bf16 MFMA passes in bf16x3 mode
        M, K, N = T_OUT * BATCH_PER_GPU, 2 * HIDDEN, 2 * 4 * HIDDEN
        passes = {"f16x3": 3, "bf16x3": 3, "fp16": 1, "f32": 1}[mode]
        gname = "gemm_nt_f32_kernel" if mode == "f32" else "gemm_nt_bf16x3_kernel4"
        gemm = {"kernel": gname, "shape": f"M {M} x N {N} x K {K}", "kernel_note": "layers 2-5 of the stack", "bound": "mfma"}
        g_ms = gemm_k2048_ms
        if g_ms > 0:
            flop = passes * 2.0 * M * N * K
            gemm.update({"launch_ms": round(g_ms, 4), "executed_flop_per_launch": flop,
                         "achieved": round(flop / (g_ms * 1e-3) / 1e12, 1), "peak": mfma_peak, "unit": "TFLOP/s",
                         "frac": round(flop / (g_ms * 1e-3) / 1e12 / mfma_peak, 4),
                         "first_layer_k640_launch_ms": round(gemm_k640_ms, 4) if gemm_k640_ms > 0 else None})
            if gemm_k640_ms > 0:
                gemm["first_layer_k640_frac"] = round(passes * 2.0 * M * N * 640 / (gemm_k640_ms * 1e-3) / 1e12 / mfma_peak, 4)
            if two_spans is not None and two_spans[2] > 0:
                gemm["launch_ms_two_in_flight"] = round(two_spans[2], 4)
                gemm["two_in_flight_note"] = (
                    "the same float32 GEMM time-slicing with the other batch's recurrence (it cannot share a CU with it)"
                    if mode == "f32" else
                    "gemm_nt_bf16x3_kernel4n: 256 x 128 tiles, one wave per SIMD, resident beside the recurrence of the other "
                    "batch; its duration under that co-tenant")
            if pair is not None and pair["per_launch"]["gemm_k_large"] > 0:
                pg = pair["per_launch"]["gemm_k_large"]          # one launch over the 64 utterances of a pair: M = 32 064
                gemm["two_batches_per_forward"] = {"launch_ms": round(pg, 4), "M": 2 * M,
                                                   "achieved": round(2 * flop / (pg * 1e-3) / 1e12, 1),
                                                   "frac": round(2 * flop / (pg * 1e-3) / 1e12 / mfma_peak, 4)}
            grec, gwhy = pmc_record(gname + "@K2048")
            if grec is not None:
                gemm["traffic"] = int(grec["hbm_bytes"])
                gemm["hbm_frac_measured"] = round(grec["hbm_bytes"] / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                gemm["mfma_frac"] = round(grec["mfma_flop"] / (g_ms * 1e-3) / 1e12 / mfma_peak, 4)
                gemm["mfma_busy_frac_pmc"] = grec.get("mfma_busy_frac")
            else:
                gemm["traffic"] = None
                gemm["pmc_note"] = gwhy
        out = {
            "metric": "audio-sec/s (RTF), DS2 5xBiLSTM-1024 encoder forward + CTC greedy, 80-feature 10 s clips @ batch 32/GPU",
            "value": round(value, 1), "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": precision_label(), "data": "synthetic",
            "config": {"workload": "configs[1]: DS2 2xconv2d + 5xBiLSTM-1024 + FC, 80 feat x 1001 frames (10 s), batch 32/GPU, CTC greedy",
                       "global_batch": world * BATCH_PER_GPU, "frames": FRAMES, "parallelism": f"utterance-shard x{world}",
                       "in_flight": 2 if pipelined else 1,
                       "pipeline": ("two batches per forward (myrtlespeech_amd.pipeline.PairedBatches): consecutive batches of 32 go "
                                    "through the network two at a time on one stream; the two batches' recurrences run side by side "
                                    "in one launch of the wide-workgroup kernel, every other kernel once on 64 utterances; every "
                                    "step's full work incl. decode completes inside the timed region; outputs bit-identical to "
                                    "the one-batch path") if (pipelined and headline is paired) else
                                   ("two batches in flight per GPU on two HIP streams (myrtlespeech_amd.pipeline.TwoBatchesInFlight): "
                                    "this batch's wide-workgroup recurrence takes half of the CUs, the next batch's projection GEMMs the rest; "
                                    "every step's full work incl. decode completes inside the timed region; outputs "
                                    "bit-identical to the one-batch path") if pipelined else "one batch at a time",
                       "decode": "all-gather logits, batched decode on every rank" if args.gather_logits else
                                 "per-shard decode (no data-path collective)",
                       # flat scalars (the driver's record keeps scalars of `config`, not nested objects): the literal
                       # "batch 32, one batch at a time" step and the strict-f32 figure beside the headline
                       "headline_mode": ("two_batches_per_forward" if (pipelined and headline is paired) else
                                         "two_batches_in_flight" if pipelined else "one_batch_in_flight"),
                       "headline_latency_ms_per_batch": round(latency_ms if pipelined else one["span"], 3),
                       # the one-batch leg is the FIRST thing timed after W warm-up steps and its first region runs 4 .. 12 % slower
                       # than the four that follow (allocator / clock settling): the side figure is the median region, the first
                       # region is kept beside it; `value` (the headline) stays its first region, as the contract says
                       "one_batch_ms_per_step": spread["one_batch_ms_per_step"]["median"],
                       "one_batch_first_region_ms_per_step": round(one_ms, 3),
                       # the literal step's two schedules: layer by layer (every kernel alone on the device: what `stages`,
                       # `roofline` and `projection_gemm` are measured on) and overlapped (the default: one_batch_ms_per_step)
                       "one_batch_layer_by_layer_ms_per_step": round(one_serial_elapsed / args.steps * 1e3, 3),
                       "one_batch_schedule": "overlapped_stack" if one_ov is not None else "layer_by_layer",
                       "one_batch_value": round(world * BATCH_PER_GPU * CLIP_SECONDS / (spread["one_batch_ms_per_step"]["median"] * 1e-3), 1),
                       "f32_value": (f32 or {}).get("value"), "f32_ms_per_step": (f32 or {}).get("ms_per_step"),
                       "f32_one_batch_ms_per_step": ((f32 or {}).get("one_batch_in_flight") or {}).get("ms_per_step"),
                       "ragged_value": (ragged or {}).get("value"), "ragged_ms_per_step": (ragged or {}).get("ms_per_step"),
                       "ms_per_step_min": spread["headline_ms_per_step"]["min"],
                       "ms_per_step_median": spread["headline_ms_per_step"]["median"],
                       "ms_per_step_max": spread["headline_ms_per_step"]["max"],
                       "one_batch_ms_per_step_min": spread["one_batch_ms_per_step"]["min"],
                       # the second kernel's figures as scalars too (the nested `projection_gemm` object may be cut from the line)
                       "projection_gemm_launch_ms": gemm.get("launch_ms"), "projection_gemm_frac": gemm.get("frac"),
                       "recurrence_launch_ms": roof.get("launch_ms")},
            "timing": spread,
            "stages": {"one_batch": {k: round(one["per_step"][k], 3) for k in ("conv", "layout", "projection", "recurrence", "linear", "greedy")},
                       "headline": ({k: round((pair if headline is paired else two)["per_step"][k], 3)
                                     for k in ("conv", "layout", "projection", "recurrence", "linear", "greedy")} if pipelined else None)},
            "one_batch_in_flight": dict(
                {"value": round(world * BATCH_PER_GPU * CLIP_SECONDS * args.steps / one_elapsed, 1),
                 "ms_per_step": round(one_ms, 3),
                 "latency_ms_per_batch": round(one["span"], 3),
                 "latency_note": "device time from a batch's first kernel to the copy of its transcripts (HIP events); the host "
                                 "enqueues batch k+1 before it collects batch k's transcripts, the device runs one batch at a time",
                 "encoder_ms": round(enc_ms, 3), "decode_ms": round(dec_ms, 3),
                 "sync_each_step": {"steps": sync_each["steps"],
                                    "ms_per_step": round(sync_each["elapsed"] / sync_each["steps"] * 1e3, 3),
                                    "between_steps_ms": round(sync_each["between"], 4),
                                    "note": "the same leg with every step's transcripts collected before the next step is "
                                            "issued (round 2's protocol): the difference is the device idling while the host "
                                            "reads back, builds the lists and issues the next step's first launches"},
                 "note": "same run, same process: K steps one after the other on one stream; the kernel "
                         "durations and rooflines below are taken here"}, **stage_report(one, False)),
            "latency_ms_per_batch": round(latency_ms if pipelined else one["span"], 3),
            "encoder_ms": round(enc_ms, 3), "decode_ms": round(dec_ms, 3),
            "encoder_ms_per_rnn_step": round(enc_ms / T_OUT, 4),
            "parity": {"tolerance": "logits within 1e-3 of the reference (fp32), CTC indices bit-exact",
                       "measured": "full-size config 2 against the reference's own outputs, default f16x3 mode: TRAINED-SCALE weights "
                                   "(tests/golden/ds2_cfg2_trained_summary.npz: logits of mean 2.8 / max 17, 37 % of the LSTM gates "
                                   "saturated) max |logit error| 8.8e-4 (5.8e-4 against the reference's float64 twin; the reference's "
                                   "own float32 rounding 5.1e-4), every arg max, greedy and beam transcript equal "
                                   "(tests/test_gpu_configs.py::test_cfg2_trained_scale_default_mode_vs_reference); default-init "
                                   "weights (logits of 0.02) 7.8e-8 (tests/test_gpu_parity.py::test_ds2_cfg2_full_size_vs_reference_"
                                   "summary).  MS_PRECISION=f32: 6.3e-4 / 4.7e-8.  bf16x3 (the default of rounds 1-5): 7.7e-3 on the "
                                   "trained-scale fixture, 4 of 32 greedy transcripts differ -- outside the gate; fp16: 0.5",
                       "measured_trained_scale_max_abs_logit_error": 8.8e-4, "measured_default_init_max_abs_logit_error": 7.8e-8},
            "kernel_ms": {"lstm_recurrent_per_layer": round(rec_ms, 3), "lstm_input_projection_per_layer": round(proj_ms, 3)},
            "roofline": roof, "projection_gemm": gemm,
        }
        if two_in_flight is not None:
            two_in_flight["what_binds"] = (
                "each layer slot = one batch's wide-workgroup recurrence on half of the CUs beside the other batch's projection "
                "GEMM on the rest (all of them once the recurrence ends); the slot is the recurrence's ~2.1 ms (1.7 ms alone: "
                "the GEMM shares its L2s and the chip holds a lower clock under it, profiles/r03m_clock_probe.txt), five slots "
                "per batch plus the convolutions and output layers")
            out["two_batches_in_flight"] = two_in_flight
        if per_forward is not None:
            out["two_batches_per_forward"] = per_forward
        if pipeline_error is not None:
            out["pipeline_error"] = pipeline_error
        if frontend_ms is not None:
            out["frontend_ms_not_in_value"] = round(frontend_ms, 3)
        if ragged is not None:
            out["ragged_lengths"] = ragged
        if f32 is not None:
            out["precision_f32"] = f32
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model)
        # ---- the other BASELINE configs and BASELINE.md 3's separately reported legs (N = 1 only): every record carries its
        # floor and its CPU baseline (tools/bench_configs.py); `legs` -- the LAST key of the line, so that it survives a
        # record that keeps only the tail of stdout -- repeats the numbers without the prose
        if world == 1 and dist is None and not args.no_legs:
            from tools import bench_configs
            del pipe, paired
            torch.cuda.empty_cache() if torch.cuda.is_available() else None
            detail = bench_configs.run_legs(["ctc", "ctcgrad", "beam", "beamlm", "ds1", "rnnt", "stream", "streamctx"], cpu=not args.no_cpu_baseline)
            if stream_fp16 is not None:
                detail["cfg5_streaming_fp16"] = stream_fp16
            if gather1 is not None:
                detail["gather_logits_one_rank"] = gather1
            out["legs_detail"] = detail
            num = ("ms", "ms_min", "ms_few_rows", "ms_per_chunk", "ms_per_chunk_wall", "floor_ms", "frac_of_floor", "chain_floor_ms", "frac_of_chain_floor", "audio_sec_per_s", "utterances_per_s",
                   "encoder_ms", "beam8_decode_ms", "greedy_decode_ms", "us_per_frame", "realtime_factor", "dtype", "latency_frames",
                   "transcripts_equal_oracle_fixture", "error", "value", "one_batch_ms", "headline_mode", "steps", "lm_calls", "lm_frames")
            legs = {"calibration": detail.get("calibration")}
            for name, rec in detail.items():
                if name == "calibration" or not isinstance(rec, dict):
                    continue
                legs[name] = compact(rec, num)
                cb = rec.get("cpu_baseline")
                if cb:
                    legs[name]["cpu"] = compact(cb, ("value", "unit", "cores"))
            legs["encoder_greedy"] = {"ms": round(ms_per_step, 3), "value": round(value, 1), "one_batch_ms": spread["one_batch_ms_per_step"]["median"],
                                      "f32_ms": (f32 or {}).get("ms_per_step"), "roofline_frac": roof["frac"],
                                      "cpu": compact(out.get("cpu_baseline", {}), ("value", "unit", "cores"))}
            out["legs"] = legs
            # the box calibration as FLAT config scalars (the driver's record keeps scalars of `config`): a reader of
            # BENCH_rNN.json can tell a slow box (these move with it) from a regression of the kernels (these do not)
            cal = detail.get("calibration") or {}
            for k_, v_ in cal.items():
                out["config"]["calib_" + k_] = v_
        # the full record: a side file + stderr; stdout gets ONE line of numbers under LINE_LIMIT bytes
        full = json.dumps(out, indent=1, default=str)
        try:
            with open(args.detail_path, "w") as f:
                f.write(full + "\n")
        except OSError as e:
            sys.stderr.write(f"bench: could not write {args.detail_path}: {e}\n")
        sys.stderr.write(full + "\n")
        sys.stderr.flush()
        sys.stdout.flush()
        os.write(json_fd, (bench_line(out, os.path.basename(args.detail_path)) + "\n").encode())
    if dist is not None:
        parallel.drop_host_groups()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
