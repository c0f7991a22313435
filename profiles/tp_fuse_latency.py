"""Protocol cost of the tensor-parallel decode exchange, fused into the kernels (tp_fuse = 1, VhXchg) against one
all-reduce kernel per exchange (tp_fuse = 0, r02): `world` engine processes on ONE GPU over the IPC transport, at a
geometry whose kernels are short (H = 1024, 32 layers, I = 1024, 4 experts, V = 2048) so that a token is mostly its
65 exchanges and BOTH ranks' launches fit the device side by side (at the released geometry the waiting blocks of one
rank starve the other rank's kernels on a shared device: vita_amd/parallel.py).  The knob is read per decode step, so
one engine alternates the two forms; ms per token and their difference / 65 = cost per exchange.
    python profiles/tp_fuse_latency.py [world]  -> one JSON line"""
import json
import os
import socket
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, ret):
    import torch.distributed as dist
    from vita_amd import _lib
    from vita_amd.checkpoint import pack_mixtral, synth_state_dict
    from vita_amd.config import TextConfig, VitaConfig
    from vita_amd.engine import MixtralEngine
    from vita_amd.parallel import setup_tensor_parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["VITA_AMD_TP_FUSE"] = "1"
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = VitaConfig.tiny()
    cfg.text = TextConfig(hidden_size=1024, num_hidden_layers=32, num_attention_heads=8, num_key_value_heads=8 if world > 2 else 2,
                          intermediate_size=1024, num_local_experts=4, vocab_size=2048)
    sd = synth_state_dict(cfg, seed=3, parts=("text",))
    packed = pack_mixtral(sd, cfg, dev, rank=rank, world=world)
    eng = MixtralEngine(cfg, packed, dev, max_ctx=1024, max_prefill=64, max_new=600, rank=rank, world=world)
    name = setup_tensor_parallel(eng, rank, world, dev, backend="gloo", collective="ipc")
    ids = np.random.default_rng(5).integers(3, cfg.text.vocab_size, size=37).tolist()
    emb = packed["embed"][torch.as_tensor(ids, device=dev)].float()
    eng.prefill(emb)
    out = {"collective": name, "fused": [], "kernel_per_exchange": []}
    steps = 48
    toks = {}
    for rnd in range(4):
        for fuse in (1, 0):
            _lib.tune("tp_fuse", fuse)
            eng.decode(4)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            eng.decode(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["fused" if fuse else "kernel_per_exchange"].append(round(dt / steps * 1e3, 4))
    out["tokens"] = eng.generated()[:24]
    out["status"] = eng._comm.status() if eng._comm is not None else None
    ret[rank] = out
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r]["tokens"] == ret[0]["tokens"] and ret[r]["status"] == 0 for r in range(world)), dict(ret)
    f = float(np.median([max(ret[r]["fused"][i] for r in range(world)) for i in range(4)]))
    k = float(np.median([max(ret[r]["kernel_per_exchange"][i] for r in range(world)) for i in range(4)]))
    print(json.dumps({"what": f"decode ms per token, TP = {world} engine processes sharing ONE MI355X, H 1024 x 32 layers (65 exchanges per token), "
                              "4 interleaved rounds of 48 steps, median of the per-round max over ranks",
                      "world": world, "collective": ret[0]["collective"], "ms_per_token_fused": round(f, 4), "ms_per_token_kernel_per_exchange": round(k, 4),
                      "saved_us_per_exchange": round((k - f) * 1e3 / 65, 2), "rounds_fused": ret[0]["fused"], "rounds_kernel": ret[0]["kernel_per_exchange"]}))
