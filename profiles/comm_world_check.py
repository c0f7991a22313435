"""Exactness of the library's IPC all-reduce at world 4 and 8 (processes sharing ONE GPU): every rank must hold the
rank-ordered fp32 sum, bit-identical across ranks, at one-shot and two-shot sizes including ragged ones — the same
worker as tests/test_comm_gpu.py (which runs world 2).  python profiles/comm_world_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.multiprocessing as mp
    from tests.test_comm_gpu import _free_port, _worker
    for world in (4, 8):
        ret = mp.Manager().dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        r0 = ret[0]
        bad = [(key, r, ret[r][key][0]) for key in r0 for r in range(world)
               if ret[r][key][0] != 0.0 or ret[r][key][1] != r0[key][1]]
        print("world", world, "sizes", len(r0), "mismatches", bad[:5], flush=True)


if __name__ == "__main__":
    main()
