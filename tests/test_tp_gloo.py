"""World-size-2 check of the tensor-parallel partition on CPU (gloo).

The device engine shards the backbone exactly as the reference's vLLM flavour does
(web_demo/vllm_tools/vllm_file/mixtral.py:375-414,441-476): q/kv heads column-sharded, o_proj
row-sharded, every expert on every rank with intermediate/world columns, one all-reduce(sum) of
the [tokens, hidden] partial after o_proj and one after the MoE down projection.  The slices
come from vita_amd.checkpoint.tp_slices — the same function pack_mixtral() uses for the HIP
engine — so this test proves the partition + collective placement reproduces the unsharded
oracle; the per-rank arithmetic here is the oracle's (no GPU in this container)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mixtral as om
from vita_amd.checkpoint import synth_state_dict, tp_slices, vocab_shard
from vita_amd.config import VitaConfig

F32 = np.float32


def _allreduce(x):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=F32))
    dist.all_reduce(t)
    return t.numpy()


def _tp_forward(sd, t, x, rank, world, n_allreduce):
    """One prefill forward of the sharded backbone on this rank; returns last-row logits."""
    qs, kvs, ff = tp_slices(t, rank, world)
    d = t.head_dim
    nq_r, nkv_r = (qs.stop - qs.start) // d, (kvs.stop - kvs.start) // d
    S = x.shape[0]
    cos, sin = om.rope_cos_sin(np.arange(S), d, t.rope_theta)
    g = lambda k: np.asarray(sd[k], F32)
    x = x.astype(F32)
    for l in range(t.num_hidden_layers):
        p = f"model.layers.{l}."
        xn = om.rmsnorm(x, g(p + "input_layernorm.weight"), t.rms_norm_eps)
        q = (xn @ g(p + "self_attn.q_proj.weight")[qs].T).astype(F32).reshape(S, nq_r, d).transpose(1, 0, 2)
        k = (xn @ g(p + "self_attn.k_proj.weight")[kvs].T).astype(F32).reshape(S, nkv_r, d).transpose(1, 0, 2)
        v = (xn @ g(p + "self_attn.v_proj.weight")[kvs].T).astype(F32).reshape(S, nkv_r, d).transpose(1, 0, 2)
        a = om.attention(om.apply_rope(q, cos, sin), om.apply_rope(k, cos, sin), v, 0)
        part = (a @ g(p + "self_attn.o_proj.weight")[:, qs].T).astype(F32)
        x = (x + _allreduce(part)).astype(F32); n_allreduce[0] += 1
        xn = om.rmsnorm(x, g(p + "post_attention_layernorm.weight"), t.rms_norm_eps)
        E = t.num_local_experts
        ex = lambda nm, sl: np.stack([g(p + f"block_sparse_moe.experts.{e}.{nm}.weight")[sl] for e in range(E)])
        lw = {"gate": g(p + "block_sparse_moe.gate.weight"), "w1": ex("w1", ff), "w3": ex("w3", ff),
              "w2": ex("w2", (slice(None), ff))}
        y, _, _ = om.moe(xn, lw, t.num_experts_per_tok)          # router replicated, experts I-sliced
        x = (x + _allreduce(y)).astype(F32); n_allreduce[0] += 1
    return (om.rmsnorm(x[-1:], g("model.norm.weight"), t.rms_norm_eps) @ g("lm_head.weight").T).astype(F32)[0]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = VitaConfig.tiny()
        sd = synth_state_dict(cfg, seed=0, rich=True, parts=("text",))
        rng = np.random.default_rng(7)
        ids = rng.integers(3, cfg.text.vocab_size, size=19)
        x = np.asarray(sd["model.embed_tokens.weight"], F32)[ids]
        n_ar = [0]
        logits = _tp_forward(sd, cfg.text, x, rank, world, n_ar)
        gathered = [torch.zeros(logits.shape[0]) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(logits))
        # vocab-sharded LM head (ParallelLMHead + logits gather, vllm_file/mixtral.py:939-951) as the engine does it: every
        # rank scores ITS rows (vita_amd.checkpoint.vocab_shard), keeps one (max, global index) candidate, writes it into
        # its slot of a zeroed [world][2] vector, the vector is all-reduced (adding zeros is exact; indices < 2^24 are
        # exact in fp32) and every rank takes the same argmax with the lowest-index tie rule
        lo, nrows = vocab_shard(cfg.text.vocab_size, rank, world)
        mine = logits[lo:lo + nrows]
        cand = np.zeros((world, 2), F32)
        j = int(np.argmax(mine))                                  # first maximum = lowest index
        cand[rank] = (mine[j], lo + j)
        cand = _allreduce(cand)
        best = max(range(world), key=lambda r: (cand[r, 0], -cand[r, 1]))
        tok_sharded = int(cand[best, 1])
        toks = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(toks, torch.tensor([tok_sharded]))
        if rank == 0:
            full, _ = om.MixtralOracle(sd, cfg.text).forward(x)
            ret["err"] = float(np.max(np.abs(full[-1] - logits)))
            ret["argmax_same"] = bool(int(np.argmax(full[-1])) == int(np.argmax(logits)))
            ret["ranks_agree"] = bool(all(torch.equal(gathered[0], gi) for gi in gathered))
            ret["n_allreduce"] = n_ar[0]
            ret["sharded_token_same"] = bool(tok_sharded == int(np.argmax(logits)) and all(int(t[0]) == tok_sharded for t in toks))
            ret["layers"] = cfg.text.num_hidden_layers
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_tp2_partition_matches_unsharded_oracle():
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert ret["ranks_agree"]                        # every rank ends with identical logits
        assert ret["n_allreduce"] == 2 * ret["layers"]   # the reference's 2 collectives per layer
        assert ret["err"] < 1e-4, ret["err"]             # fp32 summation-order noise only
        assert ret["argmax_same"]
        assert ret["sharded_token_same"]                 # the (max, index) exchange of the vocab-sharded head picks the same token


def test_tp_slices_cover_real_geometry():
    t = VitaConfig().text
    for world in (1, 2, 4, 8):
        qcov, kvcov, ffcov = [], [], []
        for r in range(world):
            q, kv, ff = tp_slices(t, r, world)
            qcov += list(range(q.start, q.stop, t.head_dim)); kvcov += list(range(kv.start, kv.stop, t.head_dim))
            ffcov.append((ff.start, ff.stop))
            # each rank's q heads map onto exactly its own kv heads (GQA group stays local)
            g = t.num_attention_heads // t.num_key_value_heads
            assert q.start // t.head_dim // g == kv.start // t.head_dim
        assert qcov == list(range(0, t.num_attention_heads * t.head_dim, t.head_dim))
        assert kvcov == list(range(0, t.num_key_value_heads * t.head_dim, t.head_dim))
        assert ffcov[0][0] == 0 and ffcov[-1][1] == t.intermediate_size
        assert all(ffcov[i][1] == ffcov[i + 1][0] for i in range(world - 1))


def test_vocab_shards_cover_the_table():
    V = VitaConfig().text.vocab_size
    for world in (1, 2, 3, 4, 8):
        rows = [vocab_shard(V, r, world) for r in range(world)]
        assert rows[0][0] == 0 and sum(n for _, n in rows) == V
        assert all(rows[r][0] + rows[r][1] == rows[r + 1][0] for r in range(world - 1))


def _vote_worker(rank, world, port, ret):
    """each rank brings DIFFERENT local measurements; the vote must come out the same on both (MAX-reduced times, any bad
    trial or a shared device forces "kernel")."""
    from vita_amd.parallel import collective_label, vote_decode_exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cases = {
            # name: (same_device, t_kernel per rank, t_fused per rank, fused_ok per rank)
            "fused_wins": (False, (40.0, 41.0), (33.0, 34.0), (True, True)),
            "fused_slower_on_one_rank": (False, (40.0, 40.0), (30.0, 45.0), (True, True)),   # slowest rank decides
            "within_margin": (False, (40.0, 40.0), (39.5, 39.7), (True, True)),              # < 2 % better: keep the kernel form
            "one_rank_timed_out": (False, (40.0, 40.0), (20.0, 20.0), (True, False)),
            "shared_device": (True, (40.0, 40.0), (10.0, 10.0), (True, True)),
            "shared_on_one_rank_only": (rank == 1, (40.0, 40.0), (10.0, 10.0), (True, True)),  # a disagreeing flag is still "shared"
            "no_timing": (False, (0.0, 0.0), (0.0, 0.0), (True, True)),
        }
        out = {}
        for name, (same, tk, tf, ok) in cases.items():
            same_flag = same if isinstance(same, bool) else bool(same)
            choice, tk_all, tf_all = vote_decode_exchange(dist, same_flag, tk[rank], tf[rank], ok[rank])
            out[name] = (choice, tk_all, tf_all)
        ret[rank] = out

        class E:
            decode_exchange = out["fused_wins"][0]
        ret[f"label{rank}"] = (collective_label(E(), "ipc"), collective_label(E(), "rccl"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_decode_exchange_vote_is_rank_consistent():
    """the ranks agree on the batch-1 decode exchange form from their own timings; ranks on one device always
    get the kernel form."""
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_vote_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        a, b = ret[0], ret[1]
        assert a == b                                            # same choice AND same reduced numbers on every rank
        assert a["fused_wins"] == ("fused", 41.0, 34.0)
        assert a["fused_slower_on_one_rank"][0] == "kernel"
        assert a["within_margin"][0] == "kernel"
        assert a["one_rank_timed_out"][0] == "kernel"
        assert a["shared_device"][0] == "kernel" and a["shared_on_one_rank_only"][0] == "kernel"
        assert a["no_timing"][0] == "kernel"
        assert ret["label0"] == ret["label1"] == ("ipc+fused", "rccl")


class _ScriptedEngine:
    """host-only stand-in for MixtralEngine inside choose_decode_exchange: counts calls, can fail a chosen leg on a chosen rank."""

    def __init__(self, rank, fail_at=None):
        self.max_new, self.max_prefill, self.max_ctx = 80, 64, 256
        self.packed = {"embed": torch.zeros(16, 8)}
        self.counters = torch.zeros(4, dtype=torch.int32)
        self.rank, self.fail_at = rank, fail_at          # fail_at = (rank, leg): prefill of that leg raises there
        self.legs, self.resets, self.decodes = 0, 0, 0

    def prefill(self, emb):
        leg = self.legs
        self.legs += 1
        if self.fail_at == (self.rank, leg):
            raise RuntimeError("scripted failure")
        dist.barrier()                                   # the real engine's per-layer exchange: a collective

    def decode(self, n):
        self.decodes += 1
        dist.barrier()

    def reset(self):
        self.resets += 1


class _ScriptedComm:
    def status(self):
        return 0


def _trial_worker(rank, world, port, ret):
    import vita_amd.parallel as par
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par._device_sync = lambda: None
    tuned = []
    import vita_amd._lib as lib
    lib_tune, lib.tune = lib.tune, (lambda k, v: tuned.append((k, v)))
    try:
        out = {}
        # 1. a healthy trial: both legs run on both ranks, the engine is reset, the ranks agree
        os.environ.pop("VITA_AMD_TP_FUSE", None)
        os.environ["VITA_AMD_TP_TRIAL"] = "1"
        e = _ScriptedEngine(rank)
        out["healthy"] = (par.choose_decode_exchange(e, _ScriptedComm(), rank, world, dist, "cpu", "gloo", True), e.legs, e.resets)
        # 2. rank 1 fails in the FIRST leg: rank 0 must not be left inside a collective; nobody runs the second leg
        e = _ScriptedEngine(rank, fail_at=(1, 0))
        if rank == 1:
            orig = e.prefill

            def failing(emb, orig=orig):
                try:
                    orig(emb)
                except RuntimeError:
                    for _ in range(3):                   # rank 0's prefill / decode(2) / decode(steps) of this leg: on the device its
                        dist.barrier()                   # bounded spins time out and the calls return; here: join its barriers
                    raise
            e.prefill = failing
        out["one_rank_fails"] = (par.choose_decode_exchange(e, _ScriptedComm(), rank, world, dist, "cpu", "gloo", True), e.legs, e.resets)
        # 3. the forced form must be the same on every rank
        os.environ["VITA_AMD_TP_FUSE"] = "1" if rank == 0 else "0"
        try:
            par.choose_decode_exchange(_ScriptedEngine(rank), _ScriptedComm(), rank, world, dist, "cpu", "gloo", True)
            out["mismatch"] = "no error"
        except Exception as ex:
            out["mismatch"] = type(ex).__name__
        os.environ["VITA_AMD_TP_FUSE"] = "1"
        out["forced"] = par.choose_decode_exchange(_ScriptedEngine(rank), _ScriptedComm(), rank, world, dist, "cpu", "gloo", True)
        ret[rank] = out
    finally:
        lib.tune = lib_tune
        os.environ.pop("VITA_AMD_TP_FUSE", None)
        os.environ.pop("VITA_AMD_TP_TRIAL", None)
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_decode_exchange_trial_is_all_or_none():
    """every leg of the timed trial is entered by all ranks or by none, a failing rank takes its peers to the vote
    instead of leaving them in a collective, a forced form is checked across the ranks, and the engine is reset afterwards."""
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_trial_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        a, b = ret[0], ret[1]
        assert a["healthy"][0] == b["healthy"][0] and a["healthy"][0] in ("kernel", "fused")
        assert a["healthy"][1:] == b["healthy"][1:] == (2, 1)                 # two legs, one reset
        assert a["one_rank_fails"][0] == b["one_rank_fails"][0] == "kernel"
        assert a["one_rank_fails"][1] == b["one_rank_fails"][1] == 1          # the second leg was entered by nobody
        assert a["one_rank_fails"][2] == b["one_rank_fails"][2] == 1
        assert a["mismatch"] == b["mismatch"] == "VitaHipError"
        assert a["forced"] == b["forced"] == "fused"
