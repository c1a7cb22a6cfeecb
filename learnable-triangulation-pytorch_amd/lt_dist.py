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
import time

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


def _flat_broadcast(tensors, src, group):
    """Broadcast ``tensors`` from rank ``src`` in as few collectives as possible: one flat buffer per (dtype, device), copied back in
    place.  What DistributedDataParallel does with ``_sync_module_states`` / ``_broadcast_coalesced`` (the reference: train.py:453)."""
    by = {}
    for t in tensors:
        by.setdefault((t.dtype, t.device), []).append(t)
    n = 0
    for (dtype, device), ts in by.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        with torch.no_grad():
            # in place (optimiser state and plan fingerprints follow), ONE multi-tensor copy per flat buffer: a copy_ per tensor was ~630 tiny
            # kernels per training forward on the N-GPU path (ResNet-152 + V2V buffers)
            views, o = [], 0
            for t in ts:
                k = t.numel()
                views.append(flat[o:o + k].view(t.shape))
                o += k
            torch._foreach_copy_(ts, views)
        n += 1
    return n


def _checksum(tensors):
    """fp64 (sum, sum of squares, count) over ``tensors`` -- cheap fingerprint used to PROVE that replicas are identical."""
    s = torch.zeros(3, dtype=torch.float64, device=tensors[0].device if tensors else "cpu")
    for t in tensors:
        d = t.detach().double()
        s[0] += d.sum(); s[1] += (d * d).sum(); s[2] += d.numel()
    return s


class GradReducer:
    """The data-parallel exchange of the training step, with DistributedDataParallel's semantics (reference: train.py:450-453) laid out for
    the recorded backward:

        reducer.attach(model)        construction-time broadcast of every parameter AND buffer from rank 0 (DDP's _sync_module_states): ranks
                                     that built their model from different seeds start identical; remembers the buffers for sync_buffers()
        reducer.sync_buffers()       DDP(broadcast_buffers=True): rank 0's buffers (BatchNorm running statistics, num_batches_tracked) to every
                                     rank at the start of a training forward -- called by VolumetricTriangulationNet._forward_train
        reducer.reduce_inplace(flat) asynchronous all-reduce of a contiguous range of the gradient arena, recorded by the tape right behind the
                                     kernels that complete the bucket (the ring over xGMI overlaps the rest of the backward)
        reducer.wait_all()           end of the backward: sums -> means.  On RCCL the mean is the collective's own (ReduceOp.AVG): no extra
                                     kernel per bucket; gloo has no AVG, there the division is one in-place op per bucket
        reducer.push / finish        the packed (non-arena) form, any gradient order that is the same on every rank
        reducer.replicas_identical(model)   all-gathers a checksum of parameters + buffers: True iff every rank holds the same model
    """

    def __init__(self, bucket_bytes=64 << 20, group=None, broadcast_buffers=True):
        self.bucket_bytes, self.group = int(bucket_bytes), group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.backend = dist.get_backend(group) if self.on else None
        # RCCL / NCCL average inside the collective; gloo (CPU tests) sums and the division is ours
        self.avg = self.backend == "nccl"
        self.op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        self.items, self.size, self.pending, self.inplace = [], 0, [], []
        self.buckets_sent = 0
        self.broadcast_buffers = bool(broadcast_buffers)
        self._buffers, self.attached = [], False
        self.n_broadcasts = 0
        # self-diagnosis of the exchange (bench.py --gpus N --train prints it, VERDICT r3 "next" 7): bytes and buckets handed to the collective, host
        # time spent issuing / waiting, and -- on a GPU -- the window from the first bucket's all-reduce to the completion of the last one
        self.bytes_sent, self.t_issue, self.t_wait, self.steps_done = 0, 0.0, 0.0, 0
        self._ev_first = self._ev_last = None
        self._window_ms, self._window_n = 0.0, 0
        # event pairs around the exchange + a synchronize() per step to read them: diagnostics for bench.py's N-GPU line, OFF in production training
        # (they cap how far the host can run ahead).  LT_TRAIN_TIMING=1 (bench.py sets it) or ``reducer.timing = True`` turns them on.
        self.timing = os.environ.get("LT_TRAIN_TIMING") == "1"

    # ---- DistributedDataParallel's construction-time / per-forward synchronisation -------------------------------------------------------
    def attach(self, model, src=0):
        """Every parameter and buffer of ``model`` becomes rank ``src``'s (in place).  Returns self."""
        params = [p for p in model.parameters()]
        self._buffers = [b for b in model.buffers()]
        if self.world > 1:
            self.n_broadcasts += _flat_broadcast(params + self._buffers, src, self.group)
            for p in params:          # cached inference plans / weight fingerprints see the new values
                torch.autograd.graph.increment_version(p)
        self.attached = True
        return self

    def sync_buffers(self, src=0):
        if self.world > 1 and self.broadcast_buffers and self._buffers:
            self.n_broadcasts += _flat_broadcast(self._buffers, src, self.group)

    def replicas_identical(self, model, buffers=True):
        """buffers=False: parameters only (BatchNorm statistics are per rank between two sync_buffers(), as under DDP)."""
        ts = list(model.parameters()) + (list(model.buffers()) if buffers else [])
        mine = _checksum(ts)
        if self.world == 1:
            return True
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.group)
        return all(bool(torch.equal(o, out[0])) for o in out[1:])

    # ---- gradients ------------------------------------------------------------------------------------------------------------------
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
        work = dist.all_reduce(flat, op=self.op, group=self.group, async_op=True) if self.world > 1 else None
        self.pending.append((work, flat, items))
        self.buckets_sent += 1

    def reduce_inplace(self, flat):
        """Starts the (asynchronous) mean of a contiguous gradient range over the ranks, in place; complete after ``wait_all``."""
        if self.timing and flat.is_cuda and not self.inplace:          # first bucket of this backward: the exchange window opens on the stream the collective is issued on
            self._collect_window()
            self._ev_first = torch.cuda.Event(enable_timing=True)
            self._ev_first.record(torch.cuda.current_stream(flat.device))
        t0 = time.perf_counter()
        work = dist.all_reduce(flat, op=self.op, group=self.group, async_op=True) if self.world > 1 else None
        self.t_issue += time.perf_counter() - t0
        self.inplace.append((work, flat))
        self.buckets_sent += 1
        self.bytes_sent += flat.numel() * flat.element_size()

    def wait_all(self):
        t0 = time.perf_counter()
        cuda_dev = None
        for work, flat in self.inplace:
            if flat.is_cuda:
                cuda_dev = flat.device
            if work is not None:
                work.wait()
                if not self.avg:
                    flat.div_(self.world)
        self.t_wait += time.perf_counter() - t0
        if self.inplace:
            self.steps_done += 1
            if self.timing and cuda_dev is not None and self._ev_first is not None:          # ... and closes behind the last bucket (the waits are stream waits on a GPU)
                self._ev_last = torch.cuda.Event(enable_timing=True)
                self._ev_last.record(torch.cuda.current_stream(cuda_dev))
        self.inplace = []

    def _collect_window(self):
        if self._ev_first is not None and self._ev_last is not None:
            self._ev_last.synchronize()
            self._window_ms += self._ev_first.elapsed_time(self._ev_last)
            self._window_n += 1
        self._ev_first = self._ev_last = None

    def stats(self):
        """Per-step averages of the exchange since construction: what an N-GPU training line needs to diagnose itself."""
        self._collect_window()
        n = max(1, self.steps_done)
        return {"gradient_buckets_per_step": self.buckets_sent / n, "gradient_bytes_per_step": self.bytes_sent / n,
                "allreduce_host_issue_ms_per_step": 1e3 * self.t_issue / n, "allreduce_host_wait_ms_per_step": 1e3 * self.t_wait / n,
                "allreduce_window_ms_per_step": (self._window_ms / self._window_n) if self._window_n else None,
                "steps_with_exchange": self.steps_done, "world": self.world, "backend": self.backend, "mean_inside_collective": bool(self.avg)}

    def finish(self):
        self._flush()
        out = {}
        for work, flat, items in self.pending:
            if work is not None:
                work.wait()
                if not self.avg:
                    flat.div_(self.world)
            o = 0
            for p, g in items:
                out[p] = flat[o:o + g.numel()].view(g.shape)
                o += g.numel()
        self.pending = []
        return out


def comm_info(device="cpu"):
    """What the COMMUNICATOR says about the job (not the environment): the number of ranks that took part in an all-reduce of ones, the
    backend, the RCCL version it was built against, and every rank's device -- bench.py prints it so that an N-GPU line proves N ranks."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"nranks": 1, "backend": None}
    one = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    info = {"nranks": int(round(float(one.item()))), "backend": dist.get_backend(), "world_size": dist.get_world_size()}
    dev = torch.device(device)
    if dev.type == "cuda":
        try:
            v = torch.cuda.nccl.version()
            info["version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception as e:          # noqa: BLE001 -- informational only
            info["version"] = "unavailable (%s)" % type(e).__name__
        pr = torch.cuda.get_device_properties(dev)
        ident = "%s|%s" % (pr.name, getattr(pr, "pci_bus_id", getattr(pr, "uuid", dev.index)))
    else:
        ident = "cpu|pid %d" % os.getpid()
    # every rank's device, gathered through the communicator (fixed-size byte strings)
    raw = ident.encode()[:64]
    code = torch.zeros(64, dtype=torch.uint8)
    code[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    code = code.to(dev)
    out = [torch.zeros_like(code) for _ in range(dist.get_world_size())]
    dist.all_gather(out, code)
    info["devices"] = [bytes(o.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for o in out]
    return info


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
