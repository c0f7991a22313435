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
import time

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

    def __init__(self, rank, world, cap_elems, same_device=False, ranks_per_device=None, loopback=False):
        """same_device: every rank runs on ONE GPU (tests): only then may the library fall back to a coarse-grained receive
        buffer when fine-grained (peer-coherent) memory cannot be allocated; across devices it refuses instead.
        ranks_per_device: how many ranks drive THIS rank's GPU (default: world when same_device, else 1) — the bulk all-reduce
        divides its resident-block cap by it.  loopback: a single-process communicator in which this rank plays all `world`
        ranks into its own receive slots (vh_comm_create_loopback: protocol cost of the decode exchanges without a link)."""
        from . import _lib
        self.lib = _lib.load()
        _lib.tune("comm_allow_coarse", 1 if same_device else 0)
        if ranks_per_device is None:
            ranks_per_device = world if same_device else 1
        _lib.tune("comm_ranks_per_device", 1 if loopback else max(1, int(ranks_per_device)))
        self._handle = ctypes.create_string_buffer(64)
        self.rank, self.world, self.loopback = int(rank), int(world), bool(loopback)
        self.ranks_per_device = 1 if loopback else max(1, int(ranks_per_device))
        if loopback:
            self.ptr = self.lib.vh_comm_create_loopback(self.rank, self.world, int(cap_elems))
        else:
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


def device_identity(device):
    """(host, physical GPU) of a torch device: the PCI bus id when the runtime gives one, else the visible-device lists and
    the RESOLVED index (an un-indexed "cuda" means torch.cuda.current_device(), not device 0: ADVICE r03)."""
    import socket
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pci = None
    try:
        p = torch.cuda.get_device_properties(idx)
        if getattr(p, "pci_bus_id", None) is not None:
            pci = (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    except Exception:
        pci = None
    if pci is not None:
        return (socket.gethostname(), "pci", pci)
    return (socket.gethostname(), os.environ.get("HIP_VISIBLE_DEVICES", ""), os.environ.get("ROCR_VISIBLE_DEVICES", ""),
            os.environ.get("CUDA_VISIBLE_DEVICES", ""), idx)


def ranks_on_my_device(dist, device, world):
    """(n, max_n): how many ranks of the group drive THIS rank's physical GPU, and the largest such count over the group
    (collective).  n == world: every rank shares one device (tests); max_n == 1: a GPU per rank (a node)."""
    me = device_identity(device)
    all_ = [None] * world
    dist.all_gather_object(all_, me)
    counts = [sum(1 for b in all_ if b == a) for a in all_]
    return counts[all_.index(me)], max(counts)


def ranks_share_one_device(dist, device, world):
    """True when every rank of the group drives the same physical GPU."""
    return ranks_on_my_device(dist, device, world)[0] == world


def vote_decode_exchange(dist, same_device, t_kernel_ms, t_fused_ms, fused_ok, device="cpu", backend="gloo", margin=0.02):
    """All ranks agree on the form of the batch-1 decode exchange (collective call: every rank passes ITS measurements).
    "fused" (VhXchg: the exchange inside the producer / consumer kernels, no all-reduce launch) only when
      * no two ranks share a device (a waiting consumer block then never holds a CU another rank's producer needs),
      * the fused trial finished without a device-side time-out on EVERY rank, and
      * the slowest rank's fused time beats the slowest rank's kernel time by more than `margin`;
    "kernel" (one small all-reduce kernel per exchange) otherwise.  The decision uses MAX-reduced times and a MIN-reduced
    ok flag, so every rank computes it from the same numbers."""
    t = torch.tensor([float(t_kernel_ms), float(t_fused_ms), 0.0 if fused_ok else 1.0, 1.0 if same_device else 0.0],
                     dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tk, tf, bad, shared = (float(v) for v in t.tolist())
    if shared > 0 or bad > 0 or not (tf > 0 and tk > 0):
        return "kernel", tk, tf
    return ("fused" if tf < (1.0 - margin) * tk else "kernel"), tk, tf


def _device_sync():
    torch.cuda.synchronize()


def _minmax(dist, value, device, backend):
    """(min, max) of an integer over the ranks (collective)."""
    t = torch.tensor([-int(value), int(value)], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lo, hi = t.tolist()
    return -int(lo), int(hi)


def choose_decode_exchange(eng, comm, rank, world, dist, device, backend, same_device, steps=32):
    """Times `steps` greedy decode steps of THIS engine under both exchange forms (tp_fuse 0 / 1) and lets the ranks vote
    (no environment variable on a real node).  Ranks sharing a device skip the timing — the fused form's
    waiting blocks starve the other ranks there (profiles/r03_tp_fuse_latency_*.json) — and get "kernel".
    VITA_AMD_TP_FUSE=0/1 still forces a form (debugging) — the ranks CHECK that they were all given the same value: the two
    forms are different kernels on the two sides of one exchange; VITA_AMD_TP_TRIAL=1 runs the timed trial even when the ranks
    share a device (tests: the trial is the code a real node executes at bring-up).

    Every step of the trial that contains a collective (the engine's per-layer exchanges) is entered by ALL ranks or by none:
    the ranks all-reduce an "ok so far" flag in front of each leg and behind it (a rank that failed used to walk
    to the vote while its peers sat in a barrier or inside a device-side exchange until the spin time-out).  Leaves the engine
    reset (vh_mixtral_reset) for the caller's first prefill."""
    from . import _lib
    forced = {"0": 0, "1": 1}.get(os.environ.get("VITA_AMD_TP_FUSE", ""), -1)
    lo, hi = _minmax(dist, forced, device, backend)
    if lo != hi:
        raise _lib.VitaHipError(f"VITA_AMD_TP_FUSE differs between the ranks (min {lo}, max {hi}; -1 = unset): every rank "
                                "must run the same form of the decode exchange")
    if forced >= 0:
        _lib.tune("tp_fuse", forced)
        return "fused" if forced == 1 else "kernel"
    tk = tf = 0.0
    ok = True
    steps = min(steps, max(0, (eng.max_new - 4) // 2))
    force_trial = os.environ.get("VITA_AMD_TP_TRIAL", "") == "1"
    can_time = (force_trial or not same_device) and steps >= 4 and eng.max_prefill >= 8 and eng.max_ctx > 8 + 2 * steps + 4
    can_time = _agree(dist, can_time, device, backend)          # engines are built alike, but the trial is all-or-none
    if can_time:
        emb = None
        try:
            emb = eng.packed["embed"][:8].float().contiguous()          # any 8 rows: only the timing matters
        except Exception as e:
            print(f"[vita_amd.parallel] rank {rank}: decode-exchange trial could not start: {e}", file=sys.stderr)
            ok = False
        for fuse in (0, 1):
            if not _agree(dist, ok, device, backend):                    # nobody enters a leg unless every rank can
                ok = False
                break
            ms = 0.0
            try:
                _lib.tune("tp_fuse", fuse)
                eng.prefill(emb)
                eng.decode(2)                                            # first launches of these kernels
                _device_sync()
                t0 = time.perf_counter()                                 # 32 steps are tens of ms: wall time between two syncs
                eng.decode(steps)
                _device_sync()
                ms = (time.perf_counter() - t0) * 1e3
                if comm.status() != 0 or int(eng.counters[3].item()) != 0:
                    ok = False
            except Exception as e:
                # the peers' kernels of this leg run into their bounded spins (sticky error word) and arrive at the flag below
                print(f"[vita_amd.parallel] rank {rank}: decode-exchange trial failed: {e}", file=sys.stderr)
                ok = False
            if fuse:
                tf = ms
            else:
                tk = ms
        ok = _agree(dist, ok, device, backend)                          # behind the last leg: one verdict for all ranks
    choice, tk_all, tf_all = vote_decode_exchange(dist, (same_device and not force_trial) or not can_time, tk, tf, ok, device, backend)
    _lib.tune("tp_fuse", 1 if choice == "fused" else 0)
    try:
        eng.reset()                                                      # host_pos / n_gen back to zero: the trial leaves no state
    except Exception as e:
        print(f"[vita_amd.parallel] rank {rank}: engine reset after the trial failed: {e}", file=sys.stderr)
    if rank == 0 and can_time:
        print(f"[vita_amd.parallel] decode exchange: {choice} (kernel {tk_all / max(steps, 1):.3f} ms/token, fused "
              f"{tf_all / max(steps, 1):.3f} ms/token over {steps} steps, slowest rank)", file=sys.stderr)
    return choice


def ipc_allreduce(eng, rank, world, dist, device, backend):
    """Bring up IpcComm for the engine and self-test it against torch.distributed before use.  Returns True when every
    rank's self-test passed (the engine then routes its all-reduces through it)."""
    comm, ok = None, 0
    same = False
    shared = False
    try:
        mine, most = ranks_on_my_device(dist, device, world)
        same = mine == world                  # every rank on ONE device: a coarse-grained receive buffer would still be correct
        shared = most > 1                     # SOME ranks share a device (e.g. 8 ranks on 4 GPUs): no exchange fused into the kernels
        comm = IpcComm(rank, world, eng.max_prefill * eng.c.hidden, same_device=same, ranks_per_device=mine)
        from . import _lib
        _lib.tune("tp_fuse", 0)                      # the self-test below and the trial start from the kernel form
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
    # which form the batch-1 decode exchange takes is measured here, on the ranks' own devices, and agreed by all ranks
    eng.decode_exchange = choose_decode_exchange(eng, comm, rank, world, dist, device, backend, same or shared)
    if not _agree(dist, comm.status() == 0, device, backend):
        # a trial that timed out leaves the communicator's error word set (sticky): no IPC collective for this job
        eng.attach_comm(None)
        comm.destroy()
        eng.decode_exchange = "kernel"
        from . import _lib
        _lib.tune("tp_fuse", 0)
        return False
    return True


def collective_label(engine, name):
    """what bench.py prints as config.collective: the transport plus the decode exchange form ("ipc+fused" / "ipc+kernel")."""
    return f"{name}+{getattr(engine, 'decode_exchange', 'kernel')}" if name == "ipc" else name


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
    # how many ranks drive THIS rank's GPU: known to the library before any collective is chosen (ranks that share a device keep the
    # decode attention block as three launches and never fuse the exchange into the kernels, whatever collective wins below)
    try:
        from . import _lib
        mine, _most = ranks_on_my_device(dist, device, world)
        _lib.tune("comm_ranks_per_device", mine)
    except Exception as e:
        print(f"[vita_amd.parallel] rank {rank}: could not count the ranks on this device: {e}", file=sys.stderr)
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
