"""Tensor-parallel bring-up for one-process-per-GPU runs (the reference's vLLM flavour:
web_demo/vllm_tools/vllm_file/mixtral.py:375-414,441-476 — RowParallel o_proj and FusedMoE reduce_results).

`setup_tensor_parallel(engine, rank, world, device)` makes the engine's per-layer all-reduces work: it binds the
process to its GPU, joins (or creates) the torch.distributed group used for bootstrap, and installs the collective
in this order of preference, every rank AGREEING on the outcome (a rank-local fallback would hang the job):

  "ipc"   the library's own one-shot / two-shot all-reduce over IPC-mapped peer buffers (vh_comm_*): latency-optimal
          for the 16 KB decode messages; verified at bring-up against torch.distributed on random vectors
  "rccl"  ncclAllReduce enqueued straight from the C layer loop (librccl resolved by dlopen)
  "torch" torch.distributed.all_reduce called back from the C layer loop (gloo on CPU-side tests)
"""
import ctypes
import os
import sys
import threading

import torch


def _agree(dist, ok, device, backend):
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def native_rccl(eng, rank, dist, device, backend, timeout_s=180):
    """Bring up the engine's own RCCL communicator, bounded in time.  Returns True when it is active on ALL ranks;
    otherwise every rank has cancelled its attempt (a late ncclCommInitRank then discards its communicator)."""
    from . import _lib
    uid = ctypes.create_string_buffer(128)
    ok = [0]
    try:
        if rank == 0:
            _lib.check(_lib.load().vh_rccl_unique_id(uid), "vh_rccl_unique_id")
        obj = [bytes(uid.raw)]
        dist.broadcast_object_list(obj, src=0)

        def run():
            try:
                torch.cuda.set_device(device)      # HIP's current device is per thread; ncclCommInitRank binds to it
                eng.use_rccl(obj[0])
                ok[0] = 1
            except Exception as e:
                print(f"[vita_amd.parallel] rank {rank}: native RCCL init failed: {e}", file=sys.stderr)

        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            print(f"[vita_amd.parallel] rank {rank}: native RCCL init still not done after {timeout_s}s", file=sys.stderr)
            ok[0] = 0
    except Exception as e:
        print(f"[vita_amd.parallel] rank {rank}: native RCCL setup failed: {e}", file=sys.stderr)
        ok[0] = 0
    agreed = _agree(dist, ok[0], device, backend)
    if not agreed:
        eng.cancel_rccl()          # whoever is still inside ncclCommInitRank must not install its communicator later
    return agreed


class IpcComm:
    """The library's own all-reduce over IPC-mapped peer buffers (include/vita_hip.h vh_comm_*): create on every rank,
    exchange the 64-byte handles through any bootstrap channel, connect, then allreduce(tensor) in lock step."""

    def __init__(self, rank, world, cap_elems, same_device=False):
        """same_device: every rank runs on ONE GPU (tests): only then may the library fall back to a coarse-grained receive
        buffer when fine-grained (peer-coherent) memory cannot be allocated; across devices it refuses instead."""
        from . import _lib
        self.lib = _lib.load()
        _lib.tune("comm_allow_coarse", 1 if same_device else 0)
        self._handle = ctypes.create_string_buffer(64)
        self.rank, self.world = int(rank), int(world)
        self.ptr = self.lib.vh_comm_create(self.rank, self.world, int(cap_elems), self._handle)
        if not self.ptr:
            raise _lib.VitaHipError("vh_comm_create failed: " + (self.lib.vh_comm_last_error() or b"").decode())

    @property
    def handle(self):
        return bytes(self._handle.raw)

    def connect(self, handles):
        from . import _lib
        buf = ctypes.create_string_buffer(b"".join(handles), 64 * len(handles))
        if self.lib.vh_comm_connect(self.ptr, buf) != 0:
            raise _lib.VitaHipError("vh_comm_connect failed: " + (self.lib.vh_comm_last_error() or b"").decode())

    @property
    def capacity(self):
        return int(self.lib.vh_comm_capacity(self.ptr))

    def allreduce(self, t):
        """in-place sum of a contiguous float32 device tensor across the ranks, on torch's current stream."""
        from . import _lib
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("IpcComm.allreduce wants a contiguous float32 device tensor")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self.lib.vh_comm_allreduce(self.ptr, t.data_ptr(), t.numel(), st) != 0:
            raise _lib.VitaHipError("vh_comm_allreduce failed: " + (self.lib.vh_comm_last_error() or b"").decode())
        return t

    def status(self):
        return int(self.lib.vh_comm_status(self.ptr))

    @property
    def fine_grained(self):
        return bool(self.lib.vh_comm_is_fine_grained(self.ptr))

    def destroy(self):
        if self.ptr:
            self.lib.vh_comm_destroy(self.ptr)
            self.ptr = None


def ranks_share_one_device(dist, device, world):
    """True when every rank of the group drives the same physical GPU (same host, same visible-device list, same index)."""
    import socket
    me = (socket.gethostname(), os.environ.get("HIP_VISIBLE_DEVICES", ""), os.environ.get("ROCR_VISIBLE_DEVICES", ""),
          os.environ.get("CUDA_VISIBLE_DEVICES", ""), torch.device(device).index or 0)
    all_ = [None] * world
    dist.all_gather_object(all_, me)
    return all(a == all_[0] for a in all_)


def ipc_allreduce(eng, rank, world, dist, device, backend):
    """Bring up IpcComm for the engine and self-test it against torch.distributed before use.  Returns True when every
    rank's self-test passed (the engine then routes its all-reduces through it)."""
    comm, ok = None, 0
    try:
        same = ranks_share_one_device(dist, device, world)
        comm = IpcComm(rank, world, eng.max_prefill * eng.c.hidden, same_device=same)
        # The decode exchange can be FUSED into the kernels around it (tp_fuse = 1, vh_api.hip:decode_one_step / VhXchg).  It is
        # correct (tests/test_comm_gpu.py runs it at world 2 / 4 / 8) but measured slower than one small all-reduce kernel per
        # exchange wherever it could be measured (several ranks on one GPU: profiles/r03_tp_fuse_latency_*.json), and on a
        # shared device at the released geometry its 384-1024 waiting consumer blocks starve the other ranks' kernels.
        # Default off; VITA_AMD_TP_FUSE=1 selects it (e.g. to try it on real xGMI links).
        from . import _lib
        _lib.tune("tp_fuse", int(os.environ.get("VITA_AMD_TP_FUSE", "0")))
        handles = [None] * world
        dist.all_gather_object(handles, comm.handle)
        comm.connect(handles)
        ok = 1
    except Exception as e:
        print(f"[vita_amd.parallel] rank {rank}: IPC all-reduce bring-up failed: {e}", file=sys.stderr)
    if not _agree(dist, ok, device, backend):
        if comm is not None:
            comm.destroy()
        return False
    good = 1
    try:
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        for n in (eng.c.hidden, 1000, min(comm.capacity, 300 * eng.c.hidden)):   # decode message, odd size, prefill-sized
            x = torch.randn(n, generator=g)
            ref = x.clone()
            if backend == "nccl":
                ref = ref.to(device)
                dist.all_reduce(ref)
            else:
                dist.all_reduce(ref)
                ref = ref.to(device)
            y = comm.allreduce(x.to(device))
            torch.cuda.synchronize()
            if comm.status() != 0 or not torch.allclose(y, ref, rtol=1e-5, atol=1e-5):
                good = 0
                break
    except Exception as e:
        print(f"[vita_amd.parallel] rank {rank}: IPC all-reduce self-test failed: {e}", file=sys.stderr)
        good = 0
    if not _agree(dist, good, device, backend):
        comm.destroy()
        return False
    eng.attach_comm(comm)
    return True


def setup_tensor_parallel(engine, rank, world, device, backend="nccl", collective="auto", rccl_timeout_s=180):
    """Returns the name of the collective in use ("none" for world 1)."""
    if world <= 1:
        return "none"
    import torch.distributed as dist
    device = torch.device(device)
    torch.cuda.set_device(device)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if collective in ("auto", "ipc") and hasattr(engine, "attach_comm"):
        if ipc_allreduce(engine, rank, world, dist, device, backend):
            return "ipc"
        if collective == "ipc":
            collective = "auto"
    if collective in ("auto", "rccl") and backend == "nccl":
        if native_rccl(engine, rank, dist, device, backend, rccl_timeout_s):
            return "rccl"
    engine.use_torch_allreduce()
    return "torch"
