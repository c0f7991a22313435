"""Latency of the library's IPC all-reduce (vh_comm_*) with N processes on ONE GPU: measures the protocol (launch + push +
poll + sum), not xGMI links — both ranks share the device.  python profiles/comm_latency.py [world] -> one JSON line."""
import json
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, ret):
    import torch.distributed as dist
    from vita_amd.parallel import IpcComm
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = IpcComm(rank, world, 1024 * 4096, same_device=True)
    handles = [None] * world
    dist.all_gather_object(handles, comm.handle)
    comm.connect(handles)
    out = {}
    for name, n, iters in (("decode_16KB", 4096, 2000), ("candidates_64B", 2 * world, 2000), ("oneshot_max_128KB", 32768, 500),
                           ("prefill_half_4.5MB", 552 * 2048, 200), ("prefill_9MB", 552 * 4096, 200)):
        x = torch.ones(n, device=dev)
        for _ in range(20):
            comm.allreduce(x)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            comm.allreduce(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert comm.status() == 0
        out[name] = round(dt / iters * 1e6, 2)
    ret[rank] = out
    dist.barrier()
    comm.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(world, port, ret), nprocs=world, join=True)
    print(json.dumps({"what": "vh_comm all-reduce, us per call back to back on one stream (max over ranks), "
                              f"{world} processes sharing ONE MI355X", "world": world,
                      "us_per_call": {k: max(ret[r][k] for r in range(world)) for k in ret[0]}}))
