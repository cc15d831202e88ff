"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The RM2 job shards its TARGET USERS across ranks (a work-balanced range of the cluster-major user order, cut inside
the native library identically on every rank).  Scores of different users are independent given the global item
statistics, so the only exchange step is the all-gather of every rank's partial per-item rating sums plus its partial
floor-sum total (jobs RM2-1 / RM2-2 and the Hadoop counter of the reference, M/rm/RM2Job.java:130-149, 184-198):
(n_items + 1) doubles per rank, summed in rank order inside the library so the result is bit-reproducible.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run sets them)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def all_gather_stats(local):
    """local: 1-D float64 tensor (CPU with gloo, GPU with nccl/RCCL) -> (world * n) tensor, rank-major."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local.clone()
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, local.contiguous())       # RCCL over xGMI
    elif local.is_cuda:                                              # gloo rehearsal on a one-GPU box: stage through the host
        host = torch.empty(world * local.numel(), dtype=local.dtype)
        dist.all_gather(list(host.view(world, -1).unbind(0)), local.cpu().contiguous())
        out.copy_(host)
    else:
        dist.all_gather(list(out.view(world, -1).unbind(0)), local.contiguous())
    return out


def combine_in_rank_order(gathered, world):
    """What fy_rm2_set_global_stats does on the GPU, restated for host-side tests: a fixed-order sum over ranks."""
    g = gathered.view(world, -1)
    total = g[0].clone()
    for r in range(1, world):
        total += g[r]
    return total


class _DevicePointer:
    """A raw HBM pointer dressed as a CUDA-array-interface object so torch can view it without a copy."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class StatsExchange:
    """Callable handed to RM2Job.run(exchange=...): all-gathers the library's HBM buffer over RCCL."""

    def __init__(self, device_index):
        self.device = torch.device("cuda", device_index)
        self._keep = None

    def __call__(self, ptr, n):
        local = torch.as_tensor(_DevicePointer(ptr, n, "<f8"), device=self.device)
        gathered = all_gather_stats(local)
        torch.cuda.current_stream(self.device).synchronize()   # the library continues on its own stream
        self._keep = gathered
        return gathered.data_ptr()


def _view(ptr, n, typestr, device):
    """n elements of `typestr` at raw address `ptr` as a torch tensor without a copy (HBM, or host memory when the
    device is the CPU: the gloo tests of the collective contract)."""
    if device.type == "cpu":
        import ctypes
        import numpy as np
        dt = np.dtype(typestr)
        buf = (ctypes.c_char * (n * dt.itemsize)).from_address(int(ptr))
        return torch.from_numpy(np.frombuffer(buf, dtype=dt))
    return torch.as_tensor(_DevicePointer(ptr, n, typestr), device=device)


class TorchCollectives:
    """fy_collectives (include/filmyou.h) played by torch.distributed: handed to RM2Job.run(collectives=...).

    Backend "nccl" = RCCL over xGMI: the library's hipStream_t is made torch's current stream (ExternalStream), so the
    collective is ordered behind the kernels already queued on it and the kernels queued afterwards wait for it -- no
    host synchronisation.  Backend "gloo" (rehearsal on a one-GPU box / CPU tests of the plumbing) stages through host
    memory with full synchronisation on both sides."""

    def __init__(self, device_index, group=None):
        self.device = torch.device("cuda", device_index) if device_index is not None else torch.device("cpu")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.nccl = dist.get_backend(group) == "nccl"
        self.calls = {"all_gather": 0, "reduce_scatter_f32": 0, "bytes": 0}

    def _stream(self, stream_ptr):
        return torch.cuda.ExternalStream(int(stream_ptr), device=self.device)

    def _before_host(self, stream_ptr):
        if self.device.type == "cuda":
            self._stream(stream_ptr).synchronize()

    def _after_host(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def all_gather(self, send, recv, nbytes, stream):
        s = _view(send, nbytes, "|u1", self.device)
        r = _view(recv, nbytes * self.world, "|u1", self.device)
        self.calls["all_gather"] += 1
        self.calls["bytes"] += nbytes * self.world
        if self.nccl:
            with torch.cuda.stream(self._stream(stream)):
                dist.all_gather_into_tensor(r, s, group=self.group)
            return
        self._before_host(stream)
        host = s.cpu()
        outs = [torch.empty_like(host) for _ in range(self.world)]
        dist.all_gather(outs, host, group=self.group)
        r.copy_(torch.cat(outs))
        self._after_host()

    def reduce_scatter_f32(self, send, recv, count, stream):
        s = _view(send, count * self.world, "<f4", self.device)
        r = _view(recv, count, "<f4", self.device)
        self.calls["reduce_scatter_f32"] += 1
        self.calls["bytes"] += 4 * count * self.world
        if self.nccl:
            with torch.cuda.stream(self._stream(stream)):
                dist.reduce_scatter_tensor(r, s, op=dist.ReduceOp.SUM, group=self.group)
            return
        self._before_host(stream)
        host = s.cpu().clone()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)   # gloo has no reduce-scatter
        r.copy_(host.view(self.world, -1)[self.rank])
        self._after_host()


class RcclCollectives:
    """fy_collectives played by the library's own compiled RCCL transport (fy_rccl_*, csrc/fy_rccl.hip): ncclAllGather /
    ncclReduceScatter on the job's stream, no Python and no host synchronisation in the path.  torch.distributed is used
    once, to hand rank 0's 128-byte ncclUniqueId to the other ranks (a C++ / JNI host uses its own channel for that)."""

    def __init__(self, ctx, rank, world, group=None):
        import ctypes as C
        from . import _native
        self._lib = _native.load()
        self._ctx = ctx
        self.rank, self.world = rank, world
        idbuf = C.create_string_buffer(128)
        why = None
        if rank == 0:
            rc = self._lib.fy_rccl_unique_id(idbuf)
            if rc:
                why = "fy_rccl_unique_id: %s" % self._lib.fy_last_error().decode()
        if world > 1:
            # rank 0's failure travels in place of the id: every rank raises, nobody is left waiting in the broadcast
            box = [(idbuf.raw, why) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            raw, why = box[0]
            idbuf = C.create_string_buffer(raw, 128)
        if why:
            raise RuntimeError(why)
        h = C.c_void_p()
        rc = self._lib.fy_rccl_create(ctx._h, rank, world, idbuf, C.byref(h))
        if rc:
            raise RuntimeError("fy_rccl_create: %s" % self._lib.fy_last_error().decode())
        self._h = h
        self.native_struct = _native.Collectives()
        rc = self._lib.fy_rccl_collectives(self._h, C.byref(self.native_struct))
        if rc:
            raise RuntimeError("fy_rccl_collectives: %s" % self._lib.fy_last_error().decode())

    @property
    def calls(self):
        import ctypes as C
        a, r, b = C.c_int64(), C.c_int64(), C.c_int64()
        self._lib.fy_rccl_counters(self._h, C.byref(a), C.byref(r), C.byref(b))
        return {"all_gather": a.value, "reduce_scatter_f32": r.value, "bytes": b.value}

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self._ctx, "_h", None):       # the context went first (interpreter teardown): do not touch its stream
                self._lib.fy_rccl_detach_context(self._h)
            self._lib.fy_rccl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ThreadGroup:
    """In-process stand-in for a process group: `world` threads of ONE process, each with its own fy context on the
    same GPU, meet at a barrier.  Test harness for the cooperative multi-rank path on a one-GPU box.

    serialize=True: only one rank computes at a time (a lock handed over at every collective), so each rank's HIP-event
    and wall-clock phase times are those of a GPU it has to itself; `busy_s[rank]` adds up the time a rank held the GPU,
    i.e. its compute critical path without communication (tools/coop_rehearsal.py)."""

    def __init__(self, world, serialize=False):
        import threading
        torch.cuda.init()      # torch's lazy CUDA initialisation is not safe to race from the rank threads
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.lock = threading.Lock() if serialize else None
        self.busy_s = [0.0] * world
        self.busy_log = [[] for _ in range(world)]


class ThreadCollectives:
    def __init__(self, group, rank, device_index=0):
        self.g, self.rank, self.world = group, rank, group.world
        self.device = torch.device("cuda", device_index)
        self.calls = {"all_gather": 0, "reduce_scatter_f32": 0, "bytes": 0}
        self._t0 = None

    # -- serialised rehearsal: the rank owns the GPU between enter() / a collective / the next collective / leave()
    def enter(self):
        if self.g.lock is not None:
            import time
            self.g.lock.acquire()
            self._t0 = time.perf_counter()

    def leave(self, what="end"):
        if self.g.lock is not None:
            import time
            torch.cuda.synchronize(self.device)
            dt = time.perf_counter() - self._t0
            self.g.busy_s[self.rank] += dt
            self.g.busy_log[self.rank].append((what, dt))
            self.g.lock.release()

    def _meet(self, mine, stream, what):
        torch.cuda.ExternalStream(int(stream), device=self.device).synchronize()
        self.leave(what)
        self.g.slots[self.rank] = mine.clone()
        torch.cuda.synchronize(self.device)
        self.g.barrier.wait()
        return list(self.g.slots)

    def _leave(self):
        torch.cuda.synchronize(self.device)
        self.g.barrier.wait()
        self.enter()

    def all_gather(self, send, recv, nbytes, stream):
        parts = self._meet(_view(send, nbytes, "|u1", self.device), stream, "all_gather %d B" % nbytes)
        _view(recv, nbytes * self.world, "|u1", self.device).copy_(torch.cat(parts))
        self.calls["all_gather"] += 1
        self.calls["bytes"] += nbytes * self.world
        self._leave()

    def reduce_scatter_f32(self, send, recv, count, stream):
        parts = self._meet(_view(send, count * self.world, "<f4", self.device), stream, "reduce_scatter %d B" % (4 * count * self.world))
        total = parts[0].view(self.world, -1)[self.rank].clone()
        for k in range(1, self.world):            # fixed rank order
            total += parts[k].view(self.world, -1)[self.rank]
        _view(recv, count, "<f4", self.device).copy_(total)
        self.calls["reduce_scatter_f32"] += 1
        self.calls["bytes"] += 4 * count * self.world
        self._leave()
