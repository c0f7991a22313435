"""The library's IPC all-reduce (vh_comm_*) with TWO processes sharing ONE GPU (same-device IPC mapping works; no
multi-GPU box is available to this repo's tests): one-shot and two-shot paths against the plain sum, bit-identical
results on both ranks, several epochs back to back (double-buffer reuse), and the engine's tensor-parallel decode
through it against the unsharded run."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    from vita_amd.parallel import IpcComm
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = IpcComm(rank, world, 1 << 21, same_device=True)
        handles = [None] * world
        dist.all_gather_object(handles, comm.handle)
        comm.connect(handles)
        out = {}
        for it, n in enumerate([4096, 1, 1000, 32768, 40000, 1 << 20, 4096, 4096, 300 * 4096 + 7]):
            g = torch.Generator(device="cpu").manual_seed(100 * it + rank)
            x = torch.randn(n, generator=g)
            parts = [torch.randn(n, generator=torch.Generator(device="cpu").manual_seed(100 * it + r)) for r in range(world)]
            ref = parts[0].clone()
            for p in parts[1:]:
                ref += p                                # rank order, fp32: the kernel's order
            y = comm.allreduce(x.to(dev))
            torch.cuda.synchronize()
            assert comm.status() == 0
            out[(it, n)] = (float((y.cpu() - ref).abs().max()), y.cpu().numpy().tobytes())
        # the 32-bit granule tag wraps after 2^32 calls: jump every rank to just below it and cross it with mixed sizes
        # (parity tracked separately from the tag, barrier + re-zero + barrier at the wrap)
        assert comm.lib.vh_comm_debug_set_calls(comm.ptr, (1 << 32) - 3) == 0
        dist.barrier()
        for it, n in enumerate([4096, 70000, 4096, 4096, 1000, 4096]):
            parts = [torch.randn(n, generator=torch.Generator(device="cpu").manual_seed(9000 + 100 * it + r)) for r in range(world)]
            ref = parts[0].clone()
            for p in parts[1:]:
                ref += p
            y = comm.allreduce(parts[rank].to(dev))
            torch.cuda.synchronize()
            assert comm.status() == 0
            out[("wrap", it, n)] = (float((y.cpu() - ref).abs().max()), y.cpu().numpy().tobytes())
        out[("fine_grained",)] = (0.0, bytes([int(comm.fine_grained)]))
        ret[rank] = out
        dist.barrier()
        comm.destroy()
    finally:
        dist.destroy_process_group()


def test_ipc_allreduce_two_processes_one_gpu():
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    for key in r0:
        assert r0[key][0] == 0.0 and r1[key][0] == 0.0, (key, r0[key][0], r1[key][0])   # exact: same order of fp32 adds
        assert r0[key][1] == r1[key][1], key                                             # bit-identical across ranks


def _tp_cfg(world):
    """world 2: the tiny geometry; world 4 / 8: 16 q heads over 8 KV heads (one KV head + its GQA group of 2 per rank at 8,
    as the released 32 / 8 geometry has one KV head + 4 q heads per rank: vllm_file/mixtral.py:441-470), intermediate 1024
    (128 columns of every expert per rank at 8)."""
    from vita_amd.config import TextConfig, VitaConfig
    if world <= 2:
        return VitaConfig.tiny()
    cfg = VitaConfig.tiny()
    cfg.text = TextConfig(hidden_size=2048, num_hidden_layers=2, num_attention_heads=16, num_key_value_heads=8,
                          intermediate_size=1024, num_local_experts=4, vocab_size=1000)
    return cfg


def _tp_worker(rank, world, port, overlap, ret, fuse=1):
    import torch.distributed as dist
    from vita_amd import _lib
    from vita_amd.checkpoint import pack_mixtral, synth_state_dict
    from vita_amd.config import VitaConfig
    from vita_amd.engine import MixtralEngine
    from vita_amd.parallel import setup_tensor_parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["VITA_AMD_TP_FUSE"] = str(int(fuse))   # 1: the exchange fused into the decode kernels (what a GPU per rank runs)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = _tp_cfg(world)
        sd = synth_state_dict(cfg, seed=3, parts=("text",))
        packed = pack_mixtral(sd, cfg, dev, rank=rank, world=world)
        eng = MixtralEngine(cfg, packed, dev, max_ctx=128, max_prefill=64, max_new=16, rank=rank, world=world, logit_rows=16)
        t = cfg.text
        assert eng.c.n_q_heads == t.num_attention_heads // world and eng.c.n_kv_heads == t.num_key_value_heads // world
        assert eng.c.inter == t.intermediate_size // world
        _lib.tune("tp_overlap", overlap)
        name = setup_tensor_parallel(eng, rank, world, dev, backend="gloo", collective="ipc")
        rng = np.random.default_rng(5)
        ids = rng.integers(3, cfg.text.vocab_size, size=37).tolist()
        emb = packed["embed"][torch.as_tensor(ids, device=dev)].float()
        row0, _ = eng.prefill(emb, gather_logits=True)   # sharded head: the full row (collective inside)
        row0 = row0.cpu()
        eng.decode(9)
        torch.cuda.synchronize()
        lg = eng.logits_all[:10].cpu()
        dist.all_reduce(lg)                      # vocab-sharded head: rows hold this rank's slice, zeros elsewhere
        assert eng.vocab_sharded and torch.equal(row0, lg[0]), "prefill() must return the full-vocabulary row under TP"
        ret[rank] = (name, eng.generated(), lg.numpy(), (eng.c.vocab_lo, eng.c.vocab_n), comm_status(eng))
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


def comm_status(eng):
    c = getattr(eng, "_comm", None)
    return c.status() if c is not None else None


@pytest.mark.parametrize("world", [4, 8])
def test_tp_engine_world_4_and_8_match_oracle(dev, world):
    """the sharded ENGINE beyond world 2 — `world` engine processes on ONE GPU over the IPC all-reduce
    (q heads 16 / world, KV heads 8 / world: one KV head per rank at 8, every expert's intermediate columns 1024 / world,
    vocabulary rows 1000 / world): greedy ids == the unsharded fp32 oracle, logits within 1e-3, all ranks bit-identical,
    no spin time-out."""
    import torch.multiprocessing as mp
    from oracle import mixtral as om
    from vita_amd.checkpoint import synth_state_dict
    ret = mp.Manager().dict()
    mp.spawn(_tp_worker, args=(world, _free_port(), 1, ret), nprocs=world, join=True)
    cfg = _tp_cfg(world)
    sd = synth_state_dict(cfg, seed=3, parts=("text",))
    rng = np.random.default_rng(5)
    ids = rng.integers(3, cfg.text.vocab_size, size=37).tolist()
    ref_ids, ref_lg = om.MixtralOracle(sd, cfg.text).greedy(sd["model.embed_tokens.weight"][ids], 10)
    # every rank agreed on ONE collective.  With `world` processes time-slicing a single GPU the IPC bring-up self-test can lose
    # its time-out race and the ranks then agree on torch.distributed (gloo) instead: the sharded engine is what this test is
    # about (the IPC transport itself is pinned at world 2 above and by profiles/comm_world_check.py at 4 and 8)
    names = {ret[r][0] for r in range(world)}
    assert len(names) == 1 and names <= {"ipc", "torch"}, {r: ret[r][0] for r in range(world)}
    assert all(ret[r][4] in (0, None) for r in range(world)), {r: ret[r][4] for r in range(world)}
    print("collective at world", world, ":", names)
    shards = [ret[r][3] for r in range(world)]
    assert shards[0][0] == 0 and sum(n for _, n in shards) == cfg.text.vocab_size
    assert all(shards[r][0] == shards[r - 1][0] + shards[r - 1][1] for r in range(1, world))      # the shards tile the table
    for r in range(world):
        assert ret[r][1] == ref_ids, f"rank {r} tokens differ from the unsharded oracle: {ret[r][1]} vs {ref_ids}"
        assert np.array_equal(ret[r][2], ret[0][2]), f"rank {r} logits differ from rank 0's"
    assert float(np.abs(ret[0][2] - ref_lg).max()) < 1e-3


@pytest.mark.parametrize("overlap,fuse", [(1, 1), (0, 1), (1, 0)])
def test_tp2_engine_over_ipc_allreduce_matches_oracle(dev, overlap, fuse):
    """two engine processes (TP = 2, one GPU) with the IPC all-reduce installed by setup_tensor_parallel: greedy ids
    equal the unsharded fp32 oracle's, logits within 1e-3, both ranks identical.  overlap = 1: the prefill's
    o_proj / MoE-down GEMMs run as column halves with the all-reduce of one half on the comm stream under the GEMM of
    the other (the default); 0: one all-reduce per sub-block on the compute stream.  fuse = 1: the DECODE exchanges are fused
    into the producer / consumer kernels (VhXchg: pushes from the O / down projections, reduction by the first blocks of the
    next kernel — the form a GPU per rank runs); 0: one all-reduce kernel per exchange (r02, the same-device default)."""
    import torch.multiprocessing as mp
    from oracle import mixtral as om
    from vita_amd.checkpoint import synth_state_dict
    from vita_amd.config import VitaConfig
    ret = mp.Manager().dict()
    mp.spawn(_tp_worker, args=(2, _free_port(), overlap, ret, fuse), nprocs=2, join=True)
    cfg = VitaConfig.tiny()
    sd = synth_state_dict(cfg, seed=3, parts=("text",))
    rng = np.random.default_rng(5)
    ids = rng.integers(3, cfg.text.vocab_size, size=37).tolist()
    ref_ids, ref_lg = om.MixtralOracle(sd, cfg.text).greedy(sd["model.embed_tokens.weight"][ids], 10)
    assert ret[0][0] == ret[1][0] == "ipc"
    V = cfg.text.vocab_size
    assert ret[0][3] == (0, (V + 1) // 2) and ret[1][3] == ((V + 1) // 2, V - (V + 1) // 2)    # the head IS sharded
    assert ret[0][1] == ret[1][1] == ref_ids
    assert np.array_equal(ret[0][2], ret[1][2])
    assert np.abs(ret[0][2] - ref_lg).max() < 1e-3
