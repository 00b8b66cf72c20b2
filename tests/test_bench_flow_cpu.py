"""bench.py's control flow on two gloo ranks with a CPU stand-in for the device side: the legs, the
barriers, the MAX all-reduce of the elapsed time, the pipelined leg and the ``--gather-logits`` branch with UNEQUAL shard
shapes per rank (``T_r`` and ``N_r`` differ, so the pad and ``cat`` path of ``parallel.gather_logits`` runs).  No scaling
number is expected from it: the first real N > 1 run on hardware must not also be the first execution of this code."""
import json
import os
import socket
import time

import torch
import torch.multiprocessing as mp

V = 29
N_RANK = (5, 3)           # utterances per rank (unequal)
T_RANK = (47, 40)         # output frames per rank (unequal)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Event:
    def __init__(self):
        self.t = None

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _Lib:
    def __init__(self, log):
        self.log = log

    def ms_prof_enable(self, on):
        self.log.append(("prof_enable", int(on)))
        return 0

    def ms_prof_read(self, ms, cnt):
        for i in range(len(ms)):
            ms[i] = 1.0
            cnt[i] = 2
        return 0

    def ms_gemm_set_variant(self, v):
        self.log.append(("gemm_variant", int(v)))
        return 0


class _Pending:
    def __init__(self, n):
        self.n = n

    def result(self):
        return [[1, 2]] * self.n


class _Decoder:
    """Checks what it is handed: per-shard logits of this rank, or -- behind ``gather_logits`` -- the whole batch, shards in
    rank order, every shard's frames past its own ``T_r`` zero."""

    def __init__(self, rank, log):
        self.rank, self.log = rank, log

    def launch(self, logits, lens):
        self.log.append(("decode", tuple(logits.shape), [int(v) for v in lens]))
        if logits.shape[1] == sum(N_RANK):
            n0 = 0
            for r, (nr, tr) in enumerate(zip(N_RANK, T_RANK)):
                blk = logits[:, n0:n0 + nr]
                assert torch.equal(blk[:tr], _shard_logits(r)), f"shard {r} arrived changed"
                assert float(blk[tr:].abs().sum()) == 0.0
                n0 += nr
        else:
            assert torch.equal(logits, _shard_logits(self.rank))
        return _Pending(logits.shape[1])


def _shard_logits(rank):
    t = torch.arange(T_RANK[rank], dtype=torch.float32).view(-1, 1, 1)
    n = torch.arange(N_RANK[rank], dtype=torch.float32).view(1, -1, 1)
    v = torch.arange(V, dtype=torch.float32).view(1, 1, -1)
    return 1000.0 * (rank + 1) + 10.0 * n + 0.01 * t + 1e-4 * v


class _Model:
    def __init__(self, rank):
        self.rank = rank

    def __call__(self, batch):
        x, lens = batch
        assert x.shape[0] == N_RANK[self.rank] == lens.numel()
        out_lens = torch.clamp((lens + 1) // 2, max=T_RANK[self.rank])
        return (_shard_logits(self.rank), out_lens), None


class _Pipe:
    def __init__(self, model, post, pre, log):
        self.models, self.post, self.pre, self.log = (model, model), post, pre, log

    def __call__(self, batches):
        self.log.append(("pipe", len(batches)))
        out = []
        for k, b in enumerate(batches):
            self.pre(k)
            out.append(self.post(self.models[k % 2](b)))
        return out


class _CpuRuntime:
    dist_backend = "gloo"
    device = "cpu"

    def __init__(self, rank):
        self.rank, self.log = rank, []

    def set_device(self, local_rank):
        self.log.append(("set_device", local_rank))

    def init_process_group(self, dist, rank, world, local_rank):
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def to_device(self, t):
        return t

    def synchronize(self):
        self.log.append(("sync",))

    def event(self):
        return _Event()

    def lib(self):
        return _Lib(self.log)

    def batch_per_rank(self, rank):
        return N_RANK[rank]

    def build_model(self, rank):
        return _Model(rank)

    def decoder(self):
        return _Decoder(self.rank, self.log)

    def pipe(self, model, post, pre):
        return _Pipe(model, post, pre, self.log)

    def paired(self, model, post, pre):
        log = self.log

        class _Paired:
            def __call__(self, batches):
                log.append(("paired", len(batches)))
                out = []
                for k, b in enumerate(batches):
                    pre(k)
                    out.append(post(model(b)))
                return out
        return _Paired()

    def check_status(self, models):
        self.log.append(("check_status", len(models)))


def _worker(rank, world, port, out_dir, extra):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    rt = _CpuRuntime(rank)
    path = os.path.join(out_dir, f"rank{rank}.json")
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    try:
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--no-frontend", "--detail-path",
                    os.path.join(out_dir, "detail.json")] + list(extra), runtime=rt, json_fd=fd)
    finally:
        os.close(fd)
    with open(os.path.join(out_dir, f"log{rank}.json"), "w") as f:
        json.dump(rt.log, f)


def _run(tmp_path, extra):
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), extra)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    with open(tmp_path / "rank0.json") as f:
        lines = [l for l in f.read().splitlines() if l.strip()]
    assert (tmp_path / "rank1.json").read_text().strip() == ""        # ONE JSON line, from rank 0
    assert len(lines) == 1
    logs = [json.loads((tmp_path / f"log{r}.json").read_text()) for r in range(world)]
    line = _check_line(lines[0])
    assert line["detail"] == "detail.json"
    # the nested per-mode records live in the side file; the line carries the standard keys and numbers
    detail = json.loads((tmp_path / "detail.json").read_text())
    for k in ("metric", "value", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype"):
        assert detail[k] == line[k]
    assert "one_batch_in_flight" not in line and "legs_detail" not in line
    return detail, logs


def _reject_constant(name):
    raise ValueError(f"non-strict JSON token {name}")


def _check_line(text):
    """What the driver's parser needs of the ONE stdout line: under 8 KB, strict JSON, the contract's keys."""
    assert "\n" not in text.strip() and len(text.encode()) < 8192, len(text.encode())
    line = json.loads(text, parse_constant=_reject_constant)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    assert all(not isinstance(v, (dict, list)) for v in line["config"].values())       # flat scalars
    assert isinstance(line["roofline"]["frac"], float) and line["roofline"]["bound"] in ("hbm", "mfma")
    assert {"achieved", "peak", "unit", "traffic", "kernel"} <= set(line["roofline"])
    assert isinstance(line["dtype"], str) and line["ms_per_step"] > 0
    assert all(len(v) <= 120 for v in _strings(line))
    return line


def _strings(o):
    if isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)
    elif isinstance(o, str):
        yield o


def test_bench_control_flow_two_ranks_per_shard_decode(tmp_path):
    out, logs = _run(tmp_path, [])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "audio-sec/s" and out["higher_is_better"] is True and out["value"] > 0
    assert out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "utterance-shard x2"
    assert "cpu_baseline" not in out and "precision_f32" not in out          # N = 1 only
    assert out["two_batches_in_flight"]["ms_per_step"] > 0 and out["one_batch_in_flight"]["ms_per_step"] > 0
    assert out["two_batches_per_forward"]["ms_per_step"] > 0
    assert set(out["one_batch_in_flight"]["stage_ms"]) == {"conv", "layout", "projection", "recurrence", "linear", "greedy", "other"}
    assert out["ragged_lengths"]["steps"] >= 6
    for rank, log in enumerate(logs):
        kinds = [e[0] for e in log]
        assert kinds.count("pipe") >= 2                    # warm-up + timed leg of the pipeline (+ ragged)
        assert kinds.count("paired") >= 2                  # the same for the two-batches-per-forward mode
        assert ["check_status", 2] in log and ["check_status", 1] in log
        shapes = {tuple(e[1]) for e in log if e[0] == "decode"}
        assert shapes == {(T_RANK[rank], N_RANK[rank], V)}  # per-shard decode: no collective on the data path


def test_bench_control_flow_two_ranks_gather_logits_unequal_shards(tmp_path):
    """--gather-logits keeps a throughput mode: the two-batches-per-forward runner is timed with the
    all-gather inside its per-batch ``post`` (one thread, batch order: the same collective order on every rank), the threaded
    two-in-flight pipeline is not (its two worker threads could order the collectives differently on different ranks)."""
    out, logs = _run(tmp_path, ["--gather-logits"])
    assert out["n_gpus"] == 2
    assert out["config"]["decode"].startswith("all-gather logits")
    assert "two_batches_in_flight" not in out and out["two_batches_per_forward"]["ms_per_step"] > 0
    assert out["config"]["headline_mode"] in ("two_batches_per_forward", "one_batch_in_flight")
    for log in logs:
        kinds = [e[0] for e in log]
        assert "pipe" not in kinds and kinds.count("paired") >= 2
        dec = [e for e in log if e[0] == "decode"]
        # every decode of the run -- the one-batch legs AND the paired legs -- saw the gathered global batch
        assert dec and all(tuple(e[1]) == (max(T_RANK), sum(N_RANK), V) for e in dec)
        # lengths of the whole batch, shards in rank order
        assert all(len(e[2]) == sum(N_RANK) for e in dec)
        full = dec[0][2]
        assert all(v <= T_RANK[0] for v in full[:N_RANK[0]]) and all(v <= T_RANK[1] for v in full[N_RANK[0]:])


def test_bench_line_stays_parseable_whatever_the_detail_grows_to():
    """Round 4's own 22.9 KB record (the one the driver could not parse) goes through `bench_line`: under 8 KB, strict JSON,
    `roofline.frac`, `cpu_baseline.value`, `config.workload`, `dtype`, `ms_per_step` and the compact `legs` still there."""
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r04s7_bench_line_20steps.json")) as f:
        old = json.loads(f.read())
    assert len(json.dumps(old)) > 20000
    old["legs_detail"]["ctc_loss"]["ms"] = float("nan")          # a NaN anywhere must not reach the line as a bare token
    old["legs"]["ctc_loss"]["ms"] = float("nan")
    line = _check_line(bench.bench_line(old, "bench_detail.json"))
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["kind"] == "port"
    assert list(line)[-1] == "legs" and line["legs"]["cfg4_rnnt"]["beam8_decode_ms"] > 0 and line["legs"]["ctc_loss"]["ms"] is None
    # and a record bloated far beyond anything seen still yields a line with the standard keys
    old["legs"] = {f"leg{i}": {"ms": 1.0, "text": "x" * 100} for i in range(200)}
    line = _check_line(bench.bench_line(old, "bench_detail.json"))
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_the_driver_record_carries_the_box_calibration_as_flat_config_scalars():
    """a reader of BENCH_rNN.json must be able to tell a slow box from a regression.  The round's own record
    (profiles/r06_bench_detail_20steps.json, the full output of `python bench.py --steps 20 --warmup 5`) goes through
    `bench_line`: the calibration figures, the second kernel's launch time / fraction and the literal step's two schedules are
    flat scalars of `config` (the driver keeps those), and the line stays under the limit."""
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r06_bench_detail_20steps.json")) as f:
        rec = json.loads(f.read())
    line = _check_line(bench.bench_line(rec, "bench_detail.json"))
    cfg = line["config"]
    for key in ("calib_lstm_step_us_n1", "calib_lstm_step_us_n64", "calib_barrier_step_us", "projection_gemm_launch_ms",
                "projection_gemm_frac", "recurrence_launch_ms", "one_batch_ms_per_step", "one_batch_layer_by_layer_ms_per_step"):
        assert isinstance(cfg.get(key), (int, float)) and cfg[key] > 0, key
    assert cfg["one_batch_schedule"] == "overlapped_stack"
    assert all(not isinstance(v, (dict, list)) for v in cfg.values())
    assert line["dtype"].startswith("f16x3") and line["roofline"]["traffic"] > 0


def test_bench_spawns_its_own_ranks_without_torchrun(tmp_path):
    """`python bench.py --gpus 2` with no RANK in the environment: the parent starts the ranks under
    torch.distributed.run before it touches a device, relays rank 0's ONE line and the exit code."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-frontend",
           "--runtime", "tests.test_bench_flow_cpu:_CpuRuntime", "--detail-path", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = _check_line(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["scaling"] == "weak"
    assert (tmp_path / "detail.json").exists()
    # a rank that dies takes the parent's exit code with it
    bad = subprocess.run(cmd + ["--runtime", "tests.test_bench_flow_cpu:_no_such_factory"], env=env, capture_output=True, text=True,
                         timeout=300, cwd=root)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]
