"""Utterance sharding across the GPUs of one node (new design; the reference has no
distributed code at all -- SURVEY 0.2, only the DataParallel remark at rnn.py:167-168).

Every stage of the hot path is independent per utterance, so the batch is cut
into contiguous shards *after* the length-descending sort the reference's collate applies
(data/batch.py:97-100): each shard stays sorted, as ``enforce_sorted=True`` (rnn.py:174)
requires, weights are replicated, and the encoder needs no collective.  The single
exchange step is for a *batched* decode on every rank: an all-gather (RCCL over xGMI on
the GPU, gloo in the CPU tests) of the per-shard logits padded to the global frame count,
plus the lengths.  At 1.86 MB per shard this is latency-bound, so it is one direct
all-gather, not a ring of smaller pieces.  When decoding per shard, only the ragged index
lists are gathered (host objects) and no device collective runs.
"""
import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from myrtlespeech_amd import _lib


def shard_bounds(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of rank's contiguous shard; earlier ranks take the remainder."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} not in [0, {world_size})")
    base, extra = divmod(batch, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_batch(x: torch.Tensor, lens: torch.Tensor, world_size: int, rank: int, batch_dim: int = 0
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """This rank's utterances of a length-sorted batch ``(x, lens)``."""
    if lens.numel() > 1 and bool((lens[:-1] < lens[1:]).any()):
        raise ValueError("shard_batch expects lengths sorted in decreasing order (data/batch.py:97-100)")
    b, e = shard_bounds(lens.numel(), world_size, rank)
    return x.narrow(batch_dim, b, e - b), lens[b:e]


def _force_collective() -> bool:
    """MS_FORCE_COLLECTIVE=1: take the collective path at world size 1 too (a one-rank RCCL run on a one-GPU box exercises
    init, the device all-gathers and the stream hand-over of the real N > 1 path; tests/test_gpu_parity.py)."""
    return os.environ.get("MS_FORCE_COLLECTIVE") == "1"


_HOST_GROUPS = {}      # ranks tuple of a device group -> its gloo twin


def _ranks_key(group: Optional[dist.ProcessGroup]) -> Tuple[int, ...]:
    if group is None or group is dist.group.WORLD:
        return tuple(range(dist.get_world_size()))
    return tuple(dist.get_process_group_ranks(group))


def init_host_group(group: Optional[dist.ProcessGroup] = None):
    """Create the gloo twin of ``group`` (default: the world) that carries small host metadata (shard shapes, lengths,
    transcripts).  ``torch.distributed.new_group`` must be entered by EVERY rank of the default group, members of ``group`` or
    not, so this is a set-up call for all world ranks, once, right after ``init_process_group`` / ``new_group(...)`` -- not
    something ``gather_logits`` may do lazily from inside a sub-group.  Returns the twin (``group`` itself when it already is
    a gloo group)."""
    if dist.get_backend(group) == "gloo":
        return group
    key = _ranks_key(group)
    if key not in _HOST_GROUPS:
        _HOST_GROUPS[key] = dist.new_group(ranks=list(key), backend="gloo")
    return _HOST_GROUPS[key]


def drop_host_groups() -> None:
    """Forget the gloo twins (call before ``destroy_process_group``; a later ``init_process_group`` starts afresh)."""
    _HOST_GROUPS.clear()


def _host_group(group: Optional[dist.ProcessGroup]):
    """Process group for host objects: with the RCCL backend ``all_gather_object`` would stage the pickled bytes through
    device tensors and read their sizes back (two host syncs in the middle of a step), so small host metadata travels over
    a gloo twin of ``group`` (``init_host_group``).  For the WORLD group every rank reaches the first gather together, so
    the twin may still be created there; for a sub-group it must have been created at set-up time by all world ranks."""
    if dist.get_backend(group) == "gloo":
        return group
    key = _ranks_key(group)
    twin = _HOST_GROUPS.get(key)
    if twin is None:
        if len(key) != dist.get_world_size():
            raise RuntimeError("myrtlespeech_amd.parallel: call init_host_group(group) on EVERY world rank right after creating "
                               "the sub-group (torch.distributed.new_group is collective over the whole world)")
        twin = init_host_group(group)
    return twin


def gather_logits(logits: torch.Tensor, lens: torch.Tensor, group: Optional[dist.ProcessGroup] = None
                  ) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather ``(logits[T_r, N_r, V], lens[N_r])`` of every rank into the full batch
    ``(logits[T_max, sum N_r, V], lens[sum N_r])`` on every rank, shards in rank order.
    Frames past a shard's own T_r are zero (they lie beyond every length of that shard).

    The shard shapes travel as host integers (``all_gather_object``), never as device tensors that would have to be read
    back in the middle of a step; the payload is ONE ``all_gather_into_tensor`` of equal-sized padded blocks plus one for the
    lengths (RCCL on the device, gloo in the CPU tests).  The lengths returned carry their host values (``_lib.attach_host``),
    so the decoder that follows does not read them back either."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _force_collective()):
        return logits, lens
    world = dist.get_world_size(group)
    dev = logits.device
    lens_host = _lib.host_lens(lens)
    shapes: List[Optional[Tuple[int, int, List[int]]]] = [None] * world
    dist.all_gather_object(shapes, (int(logits.shape[0]), int(logits.shape[1]), lens_host.tolist()),
                           group=_host_group(group))
    t_all = [s[0] for s in shapes]
    n_all = [s[1] for s in shapes]
    t_max, n_max, v = max(t_all), max(n_all), logits.shape[2]
    # equal-sized payloads so the exchange is ONE all-gather
    if logits.shape[0] == t_max and logits.shape[1] == n_max:
        pad = logits.contiguous()
    else:
        pad = torch.zeros((t_max, n_max, v), dtype=logits.dtype, device=dev)
        pad[:logits.shape[0], :logits.shape[1]] = logits
    out = torch.empty((world * t_max, n_max, v), dtype=logits.dtype, device=dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view(world, t_max, n_max, v)
    full = out[0, :, :n_all[0]] if world == 1 else torch.cat([out[r, :, :n_all[r]] for r in range(world)], dim=1)
    full_host = torch.tensor([l for s in shapes for l in s[2]], dtype=torch.int64)
    if lens.is_cuda:
        full_lens = _lib.attach_host(_lib.upload(full_host.to(lens.dtype)), full_host)
    else:
        full_lens = full_host.to(lens.dtype)
    return full.contiguous(), full_lens


def gather_transcripts(local: Sequence[List[int]], group: Optional[dist.ProcessGroup] = None) -> List[List[int]]:
    """Per-shard decode results -> the full batch's ``List[List[int]]`` on every rank
    (host objects only; no device collective)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [list(s) for s in local]
    parts: List[Optional[List[List[int]]]] = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, [list(s) for s in local], group=_host_group(group))
    return [s for part in parts for s in part]


def sharded_forward_decode(model, decoder, x: torch.Tensor, lens: torch.Tensor, batched_decode: bool = False,
                           group: Optional[dist.ProcessGroup] = None):
    """Run ``model`` on this rank's shard of the sorted global batch ``(x[N,...], lens[N])``
    and return the whole batch's transcripts on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    xs, ls = shard_batch(x, lens, world, rank)
    (logits, out_lens), _ = model((xs, ls))
    if batched_decode:
        full, full_lens = gather_logits(logits, out_lens, group)
        return decoder(full, full_lens)
    return gather_transcripts(decoder(logits, out_lens), group)
