"""One process per GPU: environment sharding and the few collectives the path needs.

The observation path has no exchange step (environments are independent), so ranks only meet to
agree on timing (max over ranks) and, for the learner, to average one flat gradient buffer.
backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, device=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (no-op for a single process).
    FLM_DIST_BACKEND overrides the backend (gloo lets several ranks share one GPU in rehearsals)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        backend = os.environ.get("FLM_DIST_BACKEND") or backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def _scalar_device(device):
    """Where the small reduction tensors live: on the GPU for RCCL, on the host for gloo."""
    return "cpu" if dist.get_backend() == "gloo" else device


def shard_range(n_total, rank, world):
    """Contiguous block of environments owned by a rank: sizes differ by at most one, blocks tile [0, n)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, device="cpu"):
    """Largest value of a python float over all ranks (the step time every rank waited for)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_scalar_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(value, device="cpu"):
    """Smallest value of a python int over all ranks (how many collective steps every rank can take)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=_scalar_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_scalar_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_mean_(flat):
    """Average one flat buffer over all ranks in place (the data-parallel gradient exchange: one bucket)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if flat.is_cuda and dist.get_backend() == "gloo":      # rehearsal without RCCL: stage through the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())
    return flat


def distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class GradientExchange:
    """The data-parallel gradient exchange of one learn step, started in two parts so that the first travels while
    the backward pass is still computing the second: begin(part) starts the sum of a slice of the flat gradient
    buffer over all ranks without waiting for it, finish() waits for every part and divides by the number of ranks.
    With RCCL the sums run on the communicator's stream, ordered behind the kernels enqueued so far; with gloo
    (rehearsal on the CPU, or several ranks on one GPU) a part is staged through the host and summed by the
    backend's own thread.  `parts` records what was reduced (tests)."""

    def __init__(self):
        self._pending = []
        self.parts = []

    def begin(self, part):
        if not distributed():
            return
        self.parts.append((part.data_ptr(), part.numel()))
        if part.is_cuda and dist.get_backend() == "gloo":
            host = part.cpu()
            self._pending.append((dist.all_reduce(host, op=dist.ReduceOp.SUM, async_op=True), part, host))
        else:
            self._pending.append((dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True), part, None))

    def finish(self):
        world = dist.get_world_size() if distributed() else 1
        for work, part, host in self._pending:
            work.wait()
            if host is not None:
                part.copy_(host)
            part.div_(world)
        self._pending = []
