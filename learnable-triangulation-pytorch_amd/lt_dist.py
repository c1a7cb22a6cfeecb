"""Multi-GPU plumbing: one process per GPU, launched by ``torch.distributed.run``; backend "nccl" (= RCCL over xGMI
on ROCm) on GPUs, "gloo" in the CPU tests.

The forward path shards by SAMPLE (one sample = NV views -> one skeleton): ranks own disjoint
slices of the batch and exchange nothing on the data path.  The only collectives are the barrier that brackets the
timed region and the MAX-reduce of the per-rank elapsed time.

Training adds ONE exchange, the gradient all-reduce of data parallelism (the reference: DistributedDataParallel,
train.py:450-453).  ``GradReducer`` is that exchange laid out for this backward: the tape (lt_train.TrainTape) hands over every
parameter gradient the moment its layer's backward has been launched, the reducer packs them into large flat buckets and
starts an asynchronous all-reduce per bucket -- RCCL runs it on its own stream behind the kernels that produced the bucket,
so the ring over xGMI overlaps the rest of the backward (the V2V gradients travel while the backbone's are computed).
Buckets are large (64 MiB): xGMI rings are per-link bound and a step has only ~320 MB of gradients.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for a single process).
    Returns (world, rank, local_rank)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        # the image exports this already (its host driver only supports dmabuf IPC; without it RCCL's hipIpcGetMemHandle fails):
        # kept for environments built by hand
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)
    return world, rank, local


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX of a python float over all ranks (the step time the job is judged on is the slowest rank's)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(value, device="cpu"):
    """[value of rank 0, value of rank 1, ...] on every rank (all_gather of one fp64; a list of one for a single process)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def shard_samples(n_samples, rank, world):
    """Indices of the samples rank ``rank`` owns: r, r+world, r+2*world, ... (the reference's DistributedSampler
    striding, train.py:68).  Shards are disjoint and cover range(n_samples)."""
    return list(range(rank, n_samples, world))


def job_throughput(samples_this_rank, elapsed_this_rank, device="cpu"):
    """Whole-job samples/s = (samples processed by ALL ranks) / (MAX elapsed over ranks)."""
    total = sum_over_ranks(samples_this_rank, device)
    t = max_over_ranks(elapsed_this_rank, device)
    return total / t, total, t


class GradReducer:
    """Bucketed, overlapped mean of parameter gradients over the ranks.

        reducer.push(param, grad)   as soon as a gradient has been launched (any order, the same on every rank)
        reducer.finish()            -> {param: averaged gradient}; waits (stream-wise on GPUs) for the buckets in flight
    """

    def __init__(self, bucket_bytes=64 << 20, group=None):
        self.bucket_bytes, self.group = int(bucket_bytes), group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.items, self.size, self.pending, self.inplace = [], 0, [], []
        self.buckets_sent = 0

    def push(self, param, grad):
        self.items.append((param, grad))
        self.size += grad.numel() * grad.element_size()
        if self.size >= self.bucket_bytes:
            self._flush()

    def _flush(self):
        if not self.items:
            return
        items, self.items, self.size = self.items, [], 0
        flat = torch.cat([g.reshape(-1) for _, g in items]) if len(items) > 1 else items[0][1].reshape(-1).clone()
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.world > 1 else None
        self.pending.append((work, flat, items))
        self.buckets_sent += 1

    def reduce_inplace(self, flat):
        """Starts the (asynchronous) sum of a contiguous gradient range over the ranks, in place; ``wait_all`` turns sums into means."""
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.world > 1 else None
        self.inplace.append((work, flat))
        self.buckets_sent += 1

    def wait_all(self):
        for work, flat in self.inplace:
            if work is not None:
                work.wait()
                flat.div_(self.world)
        self.inplace = []

    def finish(self):
        self._flush()
        out = {}
        for work, flat, items in self.pending:
            if work is not None:
                work.wait()
                flat.div_(self.world)
            o = 0
            for p, g in items:
                out[p] = flat[o:o + g.numel()].view(g.shape)
                o += g.numel()
        self.pending = []
        return out


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
