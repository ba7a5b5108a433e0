"""Sharding of self-play over ranks (one process per GPU) and the one exchange step of a learn iteration.

Games are independent units: rank r plays its own G games with disjoint seed streams and there is NO
data-path collective during self-play (reference: one self-play thread per GPU, alphazero_trainer.cpp:41-57).
The only exchange is the gather of finished (s, pi, z) records (reference: trainStorage.extend per GPU,
alphazero_trainer.cpp:59-62) — here an all_gather over torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).
"""
import os

import numpy as np

RECORD_BYTES = 265
SEED_STRIDE = 1 << 24  # rank r draws seeds base + r * 2^24 + game_no * G + g  (game_no * G + g < 2^24)


def rank_base_seed(base_seed, rank):
    """disjoint per-rank seed ranges for azr_selfplay_start"""
    return (int(base_seed) + int(rank) * SEED_STRIDE) & 0xFFFFFFFF


def selfplay_seed(base_seed, rank, games_started):
    """first seed of a rank's next batch of self-play games: base + rank * 2^24 + (games this rank has started so far).
    With azr_selfplay_start_games (game i of the batch plays seed + i) every game of every (iteration, rank) pair has
    its own seed as long as a rank starts fewer than 2^24 games in all."""
    if games_started >= SEED_STRIDE:
        raise ValueError("a rank's seed range (2^24 games) is exhausted")
    return (int(base_seed) + int(rank) * SEED_STRIDE + int(games_started)) & 0xFFFFFFFF


def force_dist():
    """AZR_FORCE_DIST=1: a world of ONE rank still goes through every collective (all_gather, all_reduce, broadcast) instead
    of the single-process shortcuts — how the RCCL code path is exercised on a box with one GPU."""
    return os.environ.get("AZR_FORCE_DIST", "0") not in ("", "0")


def _active(dist):
    return dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_dist())


def gather_records(records, dist=None, device=None):
    """all ranks contribute a uint8 tensor [n_r, 265]; every rank gets the concatenation in rank order.
    Padded all_gather: counts first, then buffers padded to the maximum count."""
    import torch

    if not _active(dist):
        return records
    world = dist.get_world_size()
    dev = records.device if device is None else device
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    buf = torch.zeros((cap, RECORD_BYTES), dtype=torch.uint8, device=dev)
    buf[:records.shape[0]] = records.to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def reduce_counters(counters, dist=None, device="cpu"):
    """sum the per-rank counter dicts (reference: GameResults::add, game.cpp:303-307)"""
    import torch

    keys = sorted(counters)
    t = torch.tensor([counters[k] for k in keys], dtype=torch.int64, device=device)
    if _active(dist):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def split_count(total, world, rank):
    """rank's share of `total` units (games, pairs): total // world, the first total % world ranks one more"""
    return total // world + (1 if rank < total % world else 0)


def broadcast_flat(flat, dist=None, src=0, device="cpu"):
    """the trained AZRW vector from the training rank to every rank (RCCL broadcast over xGMI; the reference hands the
    weights to the other GPUs through checkpoints/temp.bin, alphazero_gpu_cluster.cpp:221-231).  flat: float32 ndarray,
    same size on every rank; returns the received ndarray."""
    import torch

    if not _active(dist):
        return flat
    t = torch.from_numpy(np.ascontiguousarray(flat, np.float32)).to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def native_dp_init(eng, dist, rank, world, device):
    """give `eng` its own RCCL communicator over the ranks of `dist` (azr_dp_init): rank 0 draws the id, torch.distributed carries
    its 128 bytes to the others, then every rank joins.  Afterwards eng.train_dp(..., allreduce=None, ...) sums in stream order on
    the engine's own stream — no host hand-over per all-reduce."""
    import importlib

    import torch

    P = importlib.import_module("alphazero-risk_amd")
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(P.dp_unique_id()), dtype=torch.uint8).clone()
    buf = buf.to(device)
    if _active(dist) or (dist is not None and dist.is_initialized() and dist.get_world_size() > 1):
        dist.broadcast(buf, src=0)
    eng.dp_init(rank, world, bytes(buf.cpu().numpy().tobytes()))


class _DevicePtr:
    """a raw device allocation seen through the CUDA array interface (torch.as_tensor aliases it, no copy)"""

    def __init__(self, ptr, count, dtype):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4" if dtype == 0 else "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def make_allreduce(dist, on_device, device_index=None):
    """the callback azr_nn_train_dp wants: sum a device buffer of the engine over the ranks, in place.
    on_device (backend "nccl" = RCCL over xGMI): the collective runs on the buffer itself.  Otherwise (gloo rehearsal, ranks
    sharing a GPU): staged through a host tensor.  device_index = the ENGINE's HIP device (default: torch's current device);
    the tensor must alias the engine's buffer there — checked, because a silent copy to another device would leave the
    engine with its un-reduced sums.  Returns when the result is in place."""
    import torch

    _one_hip_runtime()
    idx = torch.cuda.current_device() if device_index is None else int(device_index)
    dev = torch.device("cuda", idx)

    def allreduce(ptr, count, dtype):
        with torch.cuda.device(dev):
            t = torch.as_tensor(_DevicePtr(ptr, count, dtype), device=dev)
            if t.data_ptr() != int(ptr) or t.device.index != idx:
                raise RuntimeError(f"make_allreduce: tensor at {t.data_ptr():#x} on {t.device} does not alias the engine buffer {int(ptr):#x} on cuda:{idx}")
            if on_device:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            else:
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.SUM)
                t.copy_(c)
            torch.cuda.synchronize(dev)

    return allreduce


def device_records_to_torch(eng, device):
    """the engine's finished records as a torch uint8 tensor [n, 265] on `device` (the GPU the engine runs on): one
    device-to-device copy on the engine's own stream (azr_samples_copy_device) — the send buffer of the record gather.
    The records stay buffered in the engine."""
    import torch

    _one_hip_runtime()
    n = eng.samples_device_view()[1]
    out = torch.empty((n, RECORD_BYTES), dtype=torch.uint8, device=device)
    if n:
        got = eng.samples_copy_device(out.data_ptr(), n)
        if got != n:
            raise RuntimeError(f"azr_samples_copy_device copied {got} of {n} records")
    return out


def _one_hip_runtime():
    """torch ships its own libamdhip64 (same soname as /opt/rocm's).  Imported FIRST, the C-ABI library binds to it too
    and both sides share one runtime; loaded the other way round the process holds two runtimes and a pointer of one
    means nothing to the other.  Refuse that state instead of copying through it."""
    libs = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    libs.add(line.split()[-1])
    except OSError:
        return
    if len(libs) > 1:
        raise RuntimeError("two HIP runtimes are mapped (%s): import torch BEFORE creating the first Engine" % ", ".join(sorted(libs)))
