"""MixtralEngine — host handle of the C-side Mixtral engine (vh_mixtral_* in include/vita_hip.h).

Owns the packed weights, the caller-provided workspace (KV cache + scratch) and the RoPE
tables; `prefill()` and `decode()` each enqueue a whole forward with one C call."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import MixtralCfg, MixtralLayer, check
from .config import VitaConfig


def rope_tables(max_pos, head_dim, theta):
    """cos/sin of pos * inv_freq in fp32, inv_freq = theta^(-2i/d) — HF MixtralRotaryEmbedding
    (modeling_mixtral.py:154-201), computed once on the host."""
    inv_freq = (1.0 / (np.float32(theta) ** (np.arange(0, head_dim, 2, dtype=np.float32) / np.float32(head_dim))))
    inv_freq = inv_freq.astype(np.float32)
    freqs = np.arange(max_pos, dtype=np.float32)[:, None] * inv_freq[None, :]
    return np.cos(freqs).astype(np.float32), np.sin(freqs).astype(np.float32)


class MixtralEngine:
    def __init__(self, cfg: VitaConfig, packed, device, max_ctx=None, max_prefill=None, max_new=1024, rank=0,
                 world=1, nsplit=0, logit_rows=0, max_seqs=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.VitaHipError("MixtralEngine needs a GPU (no CPU fallback)")
        t = cfg.text
        self.cfg, self.device, self.packed = cfg, device, packed
        max_prefill = max_prefill or cfg.tokenizer_model_max_length
        max_ctx = max_ctx or (max_prefill + max_new + 1)
        if max_seqs > 0:
            max_ctx = -(-max_ctx // 64) * 64     # paged KV cache: the pool is whole 64-token pages
        self.max_ctx, self.max_prefill, self.max_new = max_ctx, max_prefill, max_new
        lay0 = packed["layers"][0]
        nq = (lay0["wqkv"].shape[0] // t.head_dim) * t.num_attention_heads // (
            t.num_attention_heads + 2 * t.num_key_value_heads)
        nkv = nq * t.num_key_value_heads // t.num_attention_heads
        c = MixtralCfg()
        c.hidden, c.n_layers, c.n_q_heads, c.n_kv_heads = t.hidden_size, t.num_hidden_layers, nq, nkv
        c.head_dim, c.inter, c.n_experts, c.top_k = t.head_dim, lay0["w1"].shape[1], t.num_local_experts, t.num_experts_per_tok
        c.vocab, c.rms_eps = t.vocab_size, t.rms_norm_eps
        c.max_ctx, c.max_prefill, c.max_new = max_ctx, max_prefill, max_new
        c.tp_rank, c.tp_world, c.nsplit, c.logit_rows = rank, world, nsplit, logit_rows
        c.vocab_lo, c.vocab_n = int(packed.get("vocab_lo", 0)), int(packed.get("vocab_n", 0))   # vocab-sharded LM head
        c.max_seqs = int(max_seqs)
        if c.vocab_n and packed["lm_head"].shape[0] != c.vocab_n:
            raise ValueError("packed lm_head does not match its vocab shard")
        self.c = c
        nbytes = self.lib.vh_mixtral_workspace_bytes(C.byref(c))
        if nbytes == 0:
            check(-1, "vh_mixtral_workspace_bytes")
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        cos, sin = rope_tables(max_ctx, t.head_dim, t.rope_theta)
        self.rope_cos = torch.from_numpy(cos).to(device)
        self.rope_sin = torch.from_numpy(sin).to(device)
        layers = (MixtralLayer * t.num_hidden_layers)()
        for i, L in enumerate(packed["layers"]):
            for f, _ in MixtralLayer._fields_:
                setattr(layers[i], f, L[f].data_ptr())
        self._layers = layers
        self.h = self.lib.vh_mixtral_create(C.byref(c), layers, packed["embed"].data_ptr(),
                                            packed["final_norm"].data_ptr(), packed["lm_head"].data_ptr(),
                                            self.rope_cos.data_ptr(), self.rope_sin.data_ptr(),
                                            self.workspace.data_ptr(), nbytes)
        if not self.h:
            check(-1, "vh_mixtral_create")
        self._ar_cb = None
        self._comm = None
        self.decode_exchange = "kernel"     # form of the batch-1 TP exchange under the IPC transport (vita_amd.parallel votes)
        self._tok_ptr = self.lib.vh_mixtral_tokens(self.h)
        self._cnt_ptr = self.lib.vh_mixtral_counters(self.h)
        self._logit_ptr = self.lib.vh_mixtral_logits(self.h)
        # zero-copy views of the engine state living inside the workspace
        base = self.workspace.data_ptr()
        self.tokens = self.workspace[self._tok_ptr - base: self._tok_ptr - base + 4 * max(max_new, 1)].view(torch.int32)
        self.counters = self.workspace[self._cnt_ptr - base: self._cnt_ptr - base + 16].view(torch.int32)
        self.logit_rows = max(1, logit_rows)
        self.logits_all = self.workspace[self._logit_ptr - base: self._logit_ptr - base +
                                         4 * t.vocab_size * self.logit_rows].view(torch.float32).view(
            self.logit_rows, t.vocab_size)  # row i = the scores that produced generated token i
        self.n_gen = 0

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def close(self):
        if getattr(self, "_comm", None) is not None and getattr(self, "h", None):
            self.lib.vh_mixtral_use_comm(self.h, None)
            self._comm = None
        if getattr(self, "h", None):
            self.lib.vh_mixtral_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tensor parallel ---------------------------------------------------------------------
    def use_rccl(self, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        check(self.lib.vh_mixtral_init_rccl(self.h, buf), "vh_mixtral_init_rccl")

    def attach_comm(self, comm):
        """route the per-layer all-reduces through the library's IPC all-reduce (vita_amd.parallel.IpcComm); None detaches."""
        check(self.lib.vh_mixtral_use_comm(self.h, comm.ptr if comm is not None else None), "vh_mixtral_use_comm")
        self._comm = comm

    def cancel_rccl(self):
        """give up on a use_rccl() still running in another thread: it then discards its communicator."""
        check(self.lib.vh_mixtral_cancel_rccl(self.h), "vh_mixtral_cancel_rccl")

    def use_torch_allreduce(self, group=None):
        """Fallback collective: torch.distributed.all_reduce (RCCL under backend 'nccl', gloo on CPU
        tests) called back from the C layer loop."""
        import torch.distributed as dist
        base, ws = self.workspace.data_ptr(), self.workspace

        def cb(_user, ptr, count, _stream):
            t = ws[ptr - base: ptr - base + 4 * count].view(torch.float32)
            dist.all_reduce(t, group=group)
            return 0

        self._ar_cb = _lib.ALLREDUCE_FN(cb)
        check(self.lib.vh_mixtral_set_allreduce(self.h, self._ar_cb, None), "vh_mixtral_set_allreduce")

    # ---- forward -----------------------------------------------------------------------------
    @property
    def vocab_sharded(self):
        return bool(self.c.vocab_n) and self.c.tp_world > 1

    def gather_vocab(self, rows):
        """Full-vocabulary score rows under a vocab-sharded LM head (ParallelLMHead + logits gather,
        web_demo/vllm_tools/vllm_file/mixtral.py:939-951): a row kept by the engine holds THIS rank's slice and zeros
        elsewhere, so the sum over ranks is the full row.  Collective: every rank must call it.  Returns a new tensor
        (the input when the head is not sharded)."""
        if not self.vocab_sharded:
            return rows
        import torch.distributed as dist
        if not dist.is_initialized():
            raise _lib.VitaHipError("vocab-sharded LM head: full logits need torch.distributed (or build the model with "
                                    "shard_vocab=False)")
        out = rows.clone()
        if dist.get_backend() == "nccl":
            dist.all_reduce(out)
        else:
            r = out.cpu()
            dist.all_reduce(r)
            out = r.to(rows.device)
        return out

    def prefill(self, embeds, pos0=0, want_hidden=False, want_route=False, gather_logits=False):
        """embeds fp32 [S, hidden] on device.  Returns (logits_of_last_pos, hidden_dbg or None); with
        want_route the per-layer top-2 expert ids [layers, S, 2] are left in self.route_ids.  Under a vocab-sharded
        head the returned row is this rank's raw row (its vocabulary slice, zeros elsewhere) and the call has NO
        collective in it — generate() and the bench only need the token the engine already selected; gather_logits=True
        all-reduces the row over the ranks (then EVERY rank must make the same call; needs torch.distributed)."""
        if embeds.dtype != torch.float32 or not embeds.is_cuda:
            raise TypeError("embeds must be a float32 GPU tensor")
        embeds = embeds.contiguous()
        S = embeds.shape[0]
        hid = None
        if want_hidden:
            hid = torch.empty((self.c.n_layers, S, self.c.hidden), dtype=torch.float32, device=self.device)
        self.route_ids = None
        if want_route:
            self.route_ids = torch.empty((self.c.n_layers, S, 2), dtype=torch.int32, device=self.device)
            check(self.lib.vh_mixtral_route_debug(self.h, self.route_ids.data_ptr()), "vh_mixtral_route_debug")
        try:
            check(self.lib.vh_mixtral_prefill(self.h, embeds.data_ptr(), S, pos0, None,
                                              hid.data_ptr() if hid is not None else None, self._stream()),
                  "vh_mixtral_prefill")
        finally:
            if want_route:
                check(self.lib.vh_mixtral_route_debug(self.h, None), "vh_mixtral_route_debug")
        self.n_gen = 1
        return (self.gather_vocab(self.logits_all[0]) if gather_logits else self.logits_all[0]), hid

    def decode(self, n_steps):
        check(self.lib.vh_mixtral_decode(self.h, int(n_steps), self._stream()), "vh_mixtral_decode")
        self.n_gen += int(n_steps)

    def decode_schedule(self):
        """how the attention block of the last decode call ran: "fused-attention-block" (ONE launch per layer: fused QKV ->
        split-KV attention -> O projection as blocks of one grid with granule hand-offs, the default), "three-launches"
        (vh_tune("dec_fused", 0) or widths without an instantiation) or "none" before the first decode call."""
        return {1: "fused-attention-block", 0: "three-launches"}.get(int(self.lib.vh_mixtral_decode_schedule(self.h)), "none")

    def reset(self):
        """forget the current request: position and generated-token counters to zero, every KV page back to the pool
        (vh_mixtral_reset); weights, workspace and collective stay."""
        check(self.lib.vh_mixtral_reset(self.h, self._stream()), "vh_mixtral_reset")
        self.n_gen = 0

    @property
    def logits(self):
        """scores of the most recent step (row n_gen-1 of the history, or the single row) as the engine keeps them: under a
        vocab-sharded head THIS rank's slice and zeros elsewhere — gather_vocab() assembles the full row."""
        return self.logits_all[min(self.n_gen - 1, self.logit_rows - 1) if self.logit_rows > 1 else 0]

    def profile(self, stride, max_samples=2048):
        check(self.lib.vh_mixtral_profile(self.h, int(stride), int(max_samples)), "vh_mixtral_profile")

    def profile_read(self):
        """(total_ms, n_samples) of the sampled gate|up GEMV launches since the last read."""
        tot, n = C.c_double(0.0), C.c_int(0)
        check(self.lib.vh_mixtral_profile_read(self.h, C.byref(tot), C.byref(n)), "vh_mixtral_profile_read")
        return tot.value, n.value

    def check_device_flag(self, counters=None):
        """raise if a kernel reported a device-side error (counters[3]: hand-off / all-reduce spin time-out)."""
        c = self.counters.tolist() if counters is None else counters
        comm = getattr(self, "_comm", None)
        if comm is not None and comm.status() != 0:
            raise _lib.VitaHipError(f"all-reduce spin time-out (phase {comm.status()}): a peer rank is not responding")
        if c[3] != 0:
            raise _lib.VitaHipError("device-side time-out (fused decode hand-off or all-reduce): error flag set, "
                                    "the tokens of this request are not trustworthy")
        return c

    # ---- concurrent sequences over the paged KV cache (vh_mixtral_seq_*) -----------------------------------------
    def _view_i32(self, ptr, n):
        base = self.workspace.data_ptr()
        return self.workspace[ptr - base: ptr - base + 4 * n].view(torch.int32)

    def seq_alloc(self):
        """-> sequence slot id (raises VitaHipError, code VH_E_FULL, when every slot is taken)."""
        s = self.lib.vh_mixtral_seq_alloc(self.h)
        if s < 0:
            check(s, "vh_mixtral_seq_alloc")
        return s

    def seq_free(self, s):
        check(self.lib.vh_mixtral_seq_free(self.h, int(s)), "vh_mixtral_seq_free")

    def seq_prefill(self, s, embeds, want_logits=False):
        """append embeds [S, hidden] to sequence s; its next greedy token lands in seq_tokens(s)[0].  want_logits: the
        full-vocabulary row (all-reduced over the ranks under a vocab-sharded head: collective)."""
        if embeds.dtype != torch.float32 or not embeds.is_cuda:
            raise TypeError("embeds must be a float32 GPU tensor")
        embeds = embeds.contiguous()
        out = torch.empty(self.c.vocab, dtype=torch.float32, device=self.device) if want_logits else None
        check(self.lib.vh_mixtral_seq_prefill(self.h, int(s), embeds.data_ptr(), embeds.shape[0],
                                              out.data_ptr() if out is not None else None, self._stream()),
              "vh_mixtral_seq_prefill")
        return self.gather_vocab(out) if out is not None else None

    def seq_decode(self, seqs):
        """one greedy step for every listed sequence (one continuous-batching iteration)."""
        arr = (C.c_int * len(seqs))(*[int(x) for x in seqs])
        check(self.lib.vh_mixtral_seq_decode(self.h, arr, len(seqs), self._stream()), "vh_mixtral_seq_decode")

    def seq_tokens(self, s):
        return self._view_i32(self.lib.vh_mixtral_seq_tokens(self.h, int(s)), max(self.max_new, 1))

    def seq_counters(self, s):
        return self._view_i32(self.lib.vh_mixtral_seq_counters(self.h, int(s)), 4)

    def seq_pos(self, s):
        return self.lib.vh_mixtral_seq_pos(self.h, int(s))

    def seq_pages(self, s):
        buf = (C.c_int * (self.max_ctx // 64 + 1))()
        n = self.lib.vh_mixtral_seq_table(self.h, int(s), buf, len(buf))
        return list(buf[:max(n, 0)])

    def pages_free(self):
        return self.lib.vh_mixtral_pages_free(self.h)

    def generated(self):
        """(synchronising) list of token ids generated so far."""
        c = self.check_device_flag()
        return self.tokens[:min(c[1], self.max_new)].tolist()
