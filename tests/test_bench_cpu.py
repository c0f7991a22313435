"""bench.py's CPU leg (the only part of the bench that may touch oracle/): runs on the tiny geometry and
returns the fields the bench line's `cpu_baseline` object carries."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_cpu_baseline_fields():
    import bench
    from vita_amd.config import VitaConfig
    from vita_amd.host.synthetic import make_request
    cfg = VitaConfig.tiny()
    r = bench.cpu_baseline(cfg, n_layers=2, ctx=16, n_tok=3, request=make_request(cfg, seconds=1.0))
    assert r["kind"] == "port" and r["unit"] == "tokens/s" and r["value"] > 0
    assert r["prefill_ms"] > 0 and r["vit_projector_ms"] > 0 and r["audio_encoder_ms"] > 0
    assert str(r["cores"]) in r["tokens_per_s_by_threads"]
    assert r["value"] == max(r["tokens_per_s_by_threads"].values())


def test_pmc_traffic_reads_committed_profile():
    import bench
    v = bench.pmc_traffic("k_dec_gateup")
    assert v is not None and 0.9 < v / 469827584 < 1.2     # HBM bytes per launch ~ the algorithmic bytes
    assert bench.pmc_traffic("no_such_kernel") is None


def test_prefill_kernel_roofline_field():
    """the bench line carries a kernel-level PREFILL roofline (`roofline_prefill`) that can be recomputed from
    its own fields: achieved = bytes_per_launch / avg_launch_us, frac = achieved / peak, traffic from the committed PMC pass."""
    import bench
    S, E, I, H, L = 552, 8, 14336, 4096, 32
    r = bench.prefill_kernel_roofline(S, E, I, H, L, total_ms=96 * 0.6, samples=96)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBPS and "k_gemm_sp" in r["kernel"]
    assert r["bytes_per_launch"] == 2 * E * I * H * 2 + S * H * 4 + 2 * S * I * 4 and r["launches_per_prefill"] == L
    assert abs(r["avg_launch_us"] - 600.0) < 1e-6 and r["samples"] == 96
    assert abs(r["achieved"] - r["bytes_per_launch"] / 600e-6 / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert 0.2 < r["frac"] < 1.0 and 150 < r["mfma_floor_us"] < 250
    assert r["traffic"] is not None and 0.9 < r["traffic"] / (2 * E * I * H * 2) < 1.4 and r["traffic_source"].startswith("profiles/")
    assert bench.prefill_kernel_roofline(S, E, I // 8, H, L, 0.0, 0, world=8)["traffic"] is None      # TP: no static traffic figure
    assert bench.prefill_kernel_roofline(S, E, I, H, L, 0.0, 0)["achieved"] is None                   # no samples -> no claim


# ---- N>1 bring-up: every rank must reach the SAME decision about the native RCCL communicator -----------------
class _FakeEngine:
    def __init__(self, fail):
        self.fail, self.called = fail, False

    def use_rccl(self, uid):
        self.called = True
        if self.fail:
            raise RuntimeError("simulated ncclCommInitRank failure")

    def cancel_rccl(self):
        self.cancelled = True


def _bringup_worker(rank, world, port, fail_rank, ret):
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = _FakeEngine(fail=(rank == fail_rank))
        ok = bench.native_rccl_or_fallback(eng, rank, dist, torch.device("cpu"), "gloo", timeout_s=20)
        ret[rank] = (bool(ok), eng.called, getattr(eng, "cancelled", False))
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bringup(fail_rank):
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_bringup_worker, args=(2, _free_port(), fail_rank, ret), nprocs=2, join=True)
    return dict(ret)


def test_rccl_bringup_consensus_world2():
    """One rank failing its native init (or rank 0 being unable to mint an id on a GPU-less host) must send BOTH
    ranks to the torch fallback; nobody may be left believing the native communicator is active alone."""
    r = _run_bringup(fail_rank=1)
    assert r[0][0] is False and r[1][0] is False, r
    assert r[0][2] and r[1][2], r              # ... and every rank cancelled its attempt (a late init must not install)
    r2 = _run_bringup(fail_rank=-1)
    assert r2[0][0] == r2[1][0], r2            # agree either way (True only where librccl could mint an id)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it spawns its own ranks (torch.distributed.run, 127.0.0.1)
    and rank 0 prints ONE JSON line — the driver's multi-GPU command shape.  --launch-check stops after the process
    group + one all-reduce, so this runs on CPU (gloo)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--backend", "gloo"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["launch_check"] and j["n_gpus"] == 2 and j["allreduce_sum"] == j["expected"] == 3.0


def test_gpu_state_sampler_and_generate_leg_helpers(tmp_path):
    """r03 bench additions: the clock / power sampler degrades to {"available": False} without the amdgpu hwmon files and
    summarises fake ones; the synthetic tokenizer drives the reference's KeywordsStoppingCriteria (tail ids and text)."""
    import torch
    import bench
    from vita_amd.host.prompt import KeywordsStoppingCriteria
    s = bench.GpuStateSampler(index=0)
    if not s.files:
        assert s.summary() == {"available": False}
    hw = tmp_path / "hwmon0"
    hw.mkdir()
    (hw / "freq1_input").write_text("2100000000\n")
    (hw / "power1_average").write_text("750000000\n")
    s.hw, s.files = str(hw), {"sclk_mhz": str(hw / "freq1_input"), "power_w": str(hw / "power1_average")}
    s.start("phase")
    import time
    time.sleep(0.1)
    s.stop()
    out = s.summary()
    assert out["available"] and out["phase"]["samples"] >= 2
    assert out["phase"]["sclk_mhz"]["median"] == 2100.0 and out["phase"]["power_w"]["median"] == 750.0
    ids = torch.tensor([[1, 7, 9]])
    crit = KeywordsStoppingCriteria(["</s>"], bench._BenchTokenizer(), ids)
    assert not crit(torch.tensor([[1, 7, 9, 11, 12]]), None)
    assert crit(torch.tensor([[1, 7, 9, 11, 2]]), None)                 # tail ids == ids("</s>")
