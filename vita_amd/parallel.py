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


def ipc_allreduce(eng, rank, world, dist, device, backend):
    """The library's own all-reduce over IPC-mapped peer buffers; self-tested against torch.distributed before use."""
    ok = 0
    try:
        eng.comm_create(rank, world)
        handles = [None] * world
        dist.all_gather_object(handles, eng.comm_handle())
        eng.comm_connect(handles)
        ok = 1
    except Exception as e:
        print(f"[vita_amd.parallel] rank {rank}: IPC all-reduce bring-up failed: {e}", file=sys.stderr)
    if not _agree(dist, ok, device, backend):
        eng.comm_destroy()
        return False
    # self-test: three sizes (decode message, odd size, a prefill-sized message) against torch.distributed
    good = 1
    try:
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        for n in (eng.c.hidden, 1000, min(eng.comm_capacity(), 300 * eng.c.hidden)):
            x = torch.randn(n, generator=g).to(device)
            ref = x.clone()
            dist.all_reduce(ref) if backend == "nccl" else None
            if backend != "nccl":
                c = x.cpu()
                dist.all_reduce(c)
                ref = c.to(device)
            eng.comm_allreduce(x)
            torch.cuda.synchronize()
            if eng.comm_status() != 0 or not torch.allclose(x, ref, rtol=1e-5, atol=1e-5):
                good = 0
                break
    except Exception as e:
        print(f"[vita_amd.parallel] rank {rank}: IPC all-reduce self-test failed: {e}", file=sys.stderr)
        good = 0
    if not _agree(dist, good, device, backend):
        eng.comm_destroy()
        return False
    eng.use_comm()
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
    if collective in ("auto", "ipc") and hasattr(engine, "comm_create"):
        if ipc_allreduce(engine, rank, world, dist, device, backend):
            return "ipc"
        if collective == "ipc":
            collective = "auto"
    if collective in ("auto", "rccl") and backend == "nccl":
        if native_rccl(engine, rank, dist, device, backend, rccl_timeout_s):
            return "rccl"
    engine.use_torch_allreduce()
    return "torch"
