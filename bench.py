#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: greedy-decode tokens/s (+ prefill / encoder ms) of the
VITA-Mixtral-8x7B geometry on a 1 image + 10 s audio + text prompt, tensor-parallel over N GPUs.

  python bench.py --gpus 1 --steps 64 --warmup 8
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one greedy decode step (one pass of the decode hot path over the whole model for one
token).  Warm-up steps are untimed; exactly K steps are timed between barrier + synchronize pairs,
inputs (weights, KV cache, prompt) resident in HBM; value = K / max-over-ranks time.  Rank 0 prints
ONE JSON line.  Data and weights are synthetic (no checkpoint offline): weights of the released geometry from the
counter-based hash generator (vh_fill_hash_bf16 / oracle/hashw.py: sums of four 6-bit fields, ~N(0, 0.018^2), exact in
bf16, the same bits on the device and in the CPU oracle), a random 448x448 image, a seeded 10 s waveform, random prompt
ids.  Next to `value` (the engine's decode loop, no host work inside the timed region) the line carries
`generate_tokens_per_s`: the reference demo's own call — model.generate(..., stopping_criteria=[KeywordsStoppingCriteria])
(video_audio_demo.py:257-270) — timed end to end, and `gpu_state`: shader clock / socket power sampled while the prefill
and decode phases run (prefill time moves by up to 30 % from box to box at identical code: profiles/r03_layout_box_*.jsonl)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


class GpuStateSampler:
    """Shader clock (MHz) and socket power (W) of one GPU, sampled from sysfs by a background thread while a phase runs.
    Read-only files of the amdgpu driver (hwmon freq1_input / power1_average|power1_input); everything is optional: a box
    without them yields {"available": False}."""

    def __init__(self, index=0, period_s=0.02):
        import glob
        self.period = period_s
        self.hw = None
        try:   # the hwmon directory of THIS device: torch index -> PCI address -> sysfs (a node exposes every GPU's card*)
            p = torch.cuda.get_device_properties(index)
            bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            hw = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*"))
            self.hw = hw[0] if hw else None
        except Exception:
            pass
        if self.hw is None:
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
            self.hw = cards[0] if len(cards) == 1 else None          # ambiguous with several cards: report nothing
        self.files = {}
        if self.hw:
            for key, names in (("sclk_mhz", ("freq1_input",)), ("power_w", ("power1_average", "power1_input")),
                               ("temp_c", ("temp2_input", "temp1_input"))):
                for n in names:
                    f = os.path.join(self.hw, n)
                    if os.path.exists(f):
                        self.files[key] = f
                        break
        self.samples, self._stop, self._thr = {}, False, None

    def _read(self):
        out = {}
        for k, f in self.files.items():
            try:
                v = float(open(f).read().strip())
                out[k] = v / 1e6 if k in ("sclk_mhz", "power_w") else v / 1e3
            except Exception:
                pass
        return out

    def start(self, tag):
        import threading
        self._stop = False
        buf = self.samples.setdefault(tag, [])

        def loop():
            while not self._stop:
                r = self._read()
                if r:
                    buf.append(r)
                time.sleep(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        self._stop = True
        if self._thr is not None:
            self._thr.join()
            self._thr = None

    def summary(self):
        if not self.files:
            return {"available": False}
        out = {"available": True, "source": self.hw}
        for tag, buf in self.samples.items():
            d = {"samples": len(buf)}
            for k in ("sclk_mhz", "power_w", "temp_c"):
                v = [b[k] for b in buf if k in b]
                if v:
                    d[k] = {"median": round(float(np.median(v)), 1), "min": round(min(v), 1), "max": round(max(v), 1)}
            out[tag] = d
        return out


class _BenchTokenizer:
    """Just enough tokenizer for KeywordsStoppingCriteria on synthetic ids (no checkpoint / vocabulary offline): "</s>" is
    id 2, every other id decodes to a word — the per-token host work of the demo's criterion (tail compare + decode of the
    last few ids, mm_utils.py:136-155) is what the generate() leg is meant to include."""
    bos_token_id = 1

    def __call__(self, text):
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=[1, 2] if text == "</s>" else [1, 3])

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join("</s>" if int(t) == 2 else f"w{int(t)}" for t in row) for row in ids.tolist()]


TRAFFIC_FILES = ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json")


def traffic_source(kernel=None):
    for name in TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        if kernel is not None:
            try:
                with open(path) as f:
                    if kernel not in json.load(f):
                        continue
            except Exception:
                continue
        return "profiles/" + name
    return None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC pass (FETCH_SIZE, corrected as
    MI355X_MICROARCH.md prescribes; collected by profiles/pmc_pass.sh in its own run — counters cannot be
    read from inside this process).  TP=1 only: the launch moves 1/N of the bytes at TP=N."""
    for name in TRAFFIC_FILES:                       # newest committed pass first
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return int(json.load(f)[kernel]["hbm_read_bytes_per_launch"])
        except Exception:
            continue
    return None


def kernel_trace_us(kernel):
    """average duration (us) of `kernel` in the committed `rocprofv3 --kernel-trace --stats` run of this same command (TP=1; written by
    profiles/r06_measure.sh -> profiles/r06_kernel_times.json), or None.  Kernel time proper: the live HIP events of this process also
    contain the launch gap in front of the kernel (~2-4 % longer)."""
    for name in ("r06_kernel_times.json",):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return float(json.load(f)[kernel]["avg_us"]), "profiles/" + name
        except Exception:
            continue
    return None, None


PREFILL_KERNEL_KEY = "k_gemm_ps_moe_gateup (prefill S=552)"     # key of the PMC traffic files (profiles/make_traffic_json.py)


def prefill_kernel_roofline(S, E, I_r, H, layers, total_ms, samples, world=1):
    """`roofline_prefill` of the bench line: the dominant PREFILL kernel (MoE gate|up grouped GEMM + SiLU*up) against the HBM
    roofline, from live HIP events around its launches (vh_mixtral_profile with a negative stride).  Algorithmic bytes of one
    launch = the gate and up weights of every expert once (all experts are touched at S >> 8) + the bf16 hi / lo planes of the
    S normed rows once + the hi / lo planes of h (2 S rows x I) written once; re-reads through L2 are not algorithmic."""
    nbytes = int(2 * E * I_r * H * 2 + S * H * 4 + 2 * S * I_r * 4)
    us = total_ms * 1e3 / samples if samples else None
    ach = nbytes / (us * 1e-6) / 1e9 if us else None
    one_gpu = world == 1
    return {"bound": "hbm", "kernel": "k_gemm_sp<GLU> (vh_gemm_sp.hip: MoE gate|up grouped GEMM + SiLU*up, all experts; specialised form of k_gemm_ps)",
            "achieved": round(ach, 1) if ach else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 4) if ach else None,
            "frac_live_events": round(ach / HBM_PEAK_GBPS, 4) if ach else None,
            "frac_kernel_trace": (round(nbytes / (kernel_trace_us("k_gemm_sp_glu")[0] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
                                  if one_gpu and kernel_trace_us("k_gemm_sp_glu")[0] else None),
            "bytes_per_launch": nbytes, "avg_launch_us": round(us, 2) if us else None, "samples": int(samples),
            "launches_per_prefill": int(layers),
            # exact mode: 2 bf16 MFMAs per weight fragment (hi + lo planes) -> 2 x (2 S rows x 2 I x H MACs) x 2 FLOP at 2.5 PF dense
            "mfma_floor_us": round(2 * 2 * (2 * S) * (2 * I_r) * H / 2.5e15 * 1e6, 1),
            "traffic": pmc_traffic(PREFILL_KERNEL_KEY) if one_gpu else None,
            "traffic_source": traffic_source(PREFILL_KERNEL_KEY) if one_gpu else None}


def native_rccl_or_fallback(eng, rank, dist, dev, backend, timeout_s=180):
    """(kept under its round-1 name) the engine's own RCCL communicator with rank consensus: vita_amd.parallel."""
    from vita_amd.parallel import native_rccl
    return native_rccl(eng, rank, dist, dev, backend, timeout_s)


def self_launch(n):
    """`python bench.py --gpus N` started WITHOUT a launcher: re-run this command under torch.distributed.run with one
    rank per GPU (the driver's multi-GPU form is the explicit torchrun line; both end in the same main())."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(cfg, n_layers=2, ctx=None, n_tok=6, encoders=True, request=None):
    """The oracle (numpy port of the arithmetic the reference reaches: HF Mixtral, InternViT, Whale) timed on THIS box's
    host cores, on a bounded sample of the bench's own workload (SURVEY 8(d) "Reference CPU timing" a-c):
      decode   `n_layers` real-geometry decoder layers + LM head, greedy steps at the bench's real context (prompt S =
               552 rows in the KV cache), extrapolated to 32 layers; best of a few BLAS thread counts
      prefill  the same layers over the S-row prompt (the pass that fills that KV cache), extrapolated likewise
      encoders the full 24-layer InternViT + projector on one tile and the full Whale encoder + adapter on the 10 s
               clip (fp64 numpy restatements, oracle/encoders.py)
    Weights come from the counter-based generator (oracle/hashw.py) — values do not matter for timing."""
    import copy
    from oracle import hashw, mixtral as om, stream
    t = copy.deepcopy(cfg.text)
    L_full = cfg.text.num_hidden_layers
    t.num_hidden_layers = n_layers
    H = t.hidden_size
    S = int(ctx) if ctx else 552
    bufs = [stream.LayerBuffers(t) for _ in range(n_layers)]
    layers = [bufs[l].load(t, l, 0) for l in range(n_layers)]
    lm = hashw.fill((t.vocab_size, H), hashw.tensor_seed("lm_head.weight", 0))
    norm = np.ones(H, np.float32)
    x0 = hashw.fill((S, H), 12345)
    tok = hashw.fill((1, H), 54321)
    d, nq, nkv = t.head_dim, t.num_attention_heads, t.num_key_value_heads

    def forward(x, pos0, kc, vc):
        n = x.shape[0]
        cos, sin = om.rope_cos_sin(np.arange(pos0, pos0 + n), d, t.rope_theta)
        for l, Lw in enumerate(layers):
            xn = om.rmsnorm(x, Lw["ln1"], t.rms_norm_eps)
            q = om.apply_rope((xn @ Lw["q"].T).reshape(n, nq, d).transpose(1, 0, 2), cos, sin)
            k = om.apply_rope((xn @ Lw["k"].T).reshape(n, nkv, d).transpose(1, 0, 2), cos, sin)
            v = (xn @ Lw["v"].T).reshape(n, nkv, d).transpose(1, 0, 2)
            kc[l] = k if kc[l] is None else np.concatenate([kc[l], k], 1)
            vc[l] = v if vc[l] is None else np.concatenate([vc[l], v], 1)
            x = x + om.attention(q, kc[l], vc[l], pos0) @ Lw["o"].T
            y, _, _ = om.moe(om.rmsnorm(x, Lw["ln2"], t.rms_norm_eps), Lw, t.num_experts_per_tok)
            x = x + y
        return x

    def timed():
        kc, vc = [None] * n_layers, [None] * n_layers
        t0 = time.perf_counter()
        forward(x0, 0, kc, vc)                                  # prefill of the sample layers (fills the KV cache)
        t_pre = time.perf_counter() - t0
        t_full, t_head = [], []
        for i in range(n_tok):
            kk, vv = [a.copy() for a in kc], [a.copy() for a in vc]   # every step at the same context
            t0 = time.perf_counter()
            xo = forward(tok, S, kk, vv)
            t_full.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            _ = om.rmsnorm(xo, norm, t.rms_norm_eps) @ lm.T
            t_head.append(time.perf_counter() - t0)
        full, head = float(np.median(t_full[1:])), float(np.median(t_head[1:]))
        per_layer = full / n_layers
        return 1.0 / (per_layer * L_full + head), per_layer, head, t_pre

    # batch-1 GEMVs are memory-bound: the BLAS pool's default (all cores) is not the fastest setting on a
    # many-core host, so a few thread counts are tried and the BEST one is the reported baseline
    sweep = {}
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        all_thr = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        for thr in sorted({all_thr, min(all_thr, 32), min(all_thr, 8)}, reverse=True):
            with threadpool_limits(limits=thr):
                sweep[thr] = timed()
    except ImportError:
        all_thr = os.cpu_count() or 1
        sweep[all_thr] = timed()
    threads = max(sweep, key=lambda k: sweep[k][0])
    tok_s, per_layer, head, _ = sweep[threads]
    pre_thr = min(sweep, key=lambda k: sweep[k][3])
    prefill_ms = (sweep[pre_thr][3] / n_layers * L_full + head) * 1e3
    out = {"value": round(tok_s, 4), "unit": "tokens/s", "cores": int(threads), "kind": "port",
           "sample": f"numpy fp32 oracle: {n_layers} of {L_full} real-geometry decoder layers + LM head, {n_tok - 1} timed greedy steps "
                     f"at context {S} (the bench's prompt), median, best of the BLAS thread counts tried, extrapolated x{L_full}/{n_layers} "
                     "(the full fp32 model is 187 GB)",
           "ms_per_layer_token": round(per_layer * 1e3, 3), "ms_lm_head": round(head * 1e3, 3),
           "tokens_per_s_by_threads": {str(k): round(v[0], 4) for k, v in sweep.items()},
           "prefill_ms": round(prefill_ms, 1), "prefill_cores": int(pre_thr),
           "prefill_sample": f"the same {n_layers} layers over the S={S} prompt rows, x{L_full}/{n_layers} + LM head (extrapolated)",
           # every leg with ITS thread count (each is the best of the counts tried; the top-level "cores" is the decode leg's)
           "legs": {"decode": {"value": round(tok_s, 4), "unit": "tokens/s", "cores": int(threads)},
                    "prefill": {"value": round(prefill_ms, 1), "unit": "ms", "cores": int(pre_thr)}}}
    if encoders and request is not None:
        import torch as _torch
        from oracle import encoders_torch as ot
        from vita_amd.checkpoint import synth_state_dict
        sd = synth_state_dict(cfg, seed=1, rich=False, parts=("vision", "audio"))
        # small-batch encoder GEMMs do not scale to every core of a 128-thread host: a few thread counts are tried (as for the decode leg)
        # and the best pass of each tower is the reported baseline
        all_t = _torch.get_num_threads()
        t_v, t_a = {}, {}
        with _torch.no_grad():
            ot.projector(sd, ot.internvit_tower(sd, cfg.vision, request["pixel_values"][:1]))          # untimed: pages touched, pool up
            for thr in sorted({all_t, min(all_t, 32), min(all_t, 16)}, reverse=True):
                _torch.set_num_threads(thr)
                t0 = time.perf_counter()
                ot.projector(sd, ot.internvit_tower(sd, cfg.vision, request["pixel_values"][:1]))
                t_v[thr] = time.perf_counter() - t0
                t0 = time.perf_counter()
                ot.whale_encoder(sd, cfg.audio, request["fbank"])
                t_a[thr] = time.perf_counter() - t0
            _torch.set_num_threads(all_t)
        bv, ba = min(t_v, key=t_v.get), min(t_a, key=t_a.get)
        out["legs"]["vit_projector"] = {"value": round(t_v[bv] * 1e3, 1), "unit": "ms", "cores": int(bv)}
        out["legs"]["audio_encoder"] = {"value": round(t_a[ba] * 1e3, 1), "unit": "ms", "cores": int(ba)}
        out.update({"vit_projector_ms": round(t_v[bv] * 1e3, 1), "audio_encoder_ms": round(t_a[ba] * 1e3, 1),
                    "encoder_cores": int(bv), "audio_encoder_cores": int(ba),
                    "encoder_ms_by_threads": {str(k): [round(t_v[k] * 1e3, 1), round(t_a[k] * 1e3, 1)] for k in t_v},
                    "encoder_sample": "torch CPU fp32 restatement of the towers (oracle/encoders_torch.py: the operators the reference's own "
                                      "modules run; pinned to the fp64 checker): 24-layer InternViT + projector on one 448x448 tile, Whale "
                                      "encoder + adapter on the 10 s clip, one timed pass per thread count tried, best reported",
                    "encoder_note": "the reference's OWN modules on 8 cores of the build container: 1373 ms (ViT + projector) / 349 ms (Whale) "
                                    "(profiles/r02_reference_cpu_timing.json); this restatement there: 1571 / 596 ms"})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--phase-warmup", type=int, default=2, help="untimed encoder+prefill passes")
    ap.add_argument("--phase-iters", type=int, default=7, help="timed encoder+prefill passes (median and min reported)")
    ap.add_argument("--layers", type=int, default=0, help="debug only: fewer layers (result marked invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collective", default="auto", choices=["auto", "ipc", "rccl", "torch"],
                    help="auto = the library's IPC all-reduce, else native RCCL, else torch.distributed (ranks agree)")
    ap.add_argument("--launch-check", action="store_true",
                    help="debug: only bring the ranks up (process group + one all-reduce) and print a JSON line; no GPU needed with --backend gloo")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --one-device is the 1-GPU functional check of the N>1 path")
    ap.add_argument("--try-rccl", action="store_true",
                    help="debug (with --backend gloo --one-device): attempt the native RCCL init anyway — it must "
                         "fail (duplicate GPU) and every rank must agree on the torch fallback")
    ap.add_argument("--one-device", action="store_true", help="debug: every rank uses cuda:0 (invalid as a measurement)")
    ap.add_argument("--profile-stride", type=int, default=4,
                    help="live HIP-event sampling of the decode gate|up kernel: every n-th layer inside the timed steps (0 = off: "
                         "the roofline object then has no live number)")
    ap.add_argument("--emulate-tp", type=int, default=0,
                    help="debug: run ONE rank's 1/N shard on one GPU — per-rank time of TP=N without a second device (tokens are "
                         "those of the shard alone, result marked invalid).  With --loopback the 65 exchanges of a decode step run too")
    ap.add_argument("--loopback", action="store_true",
                    help="with --emulate-tp N: attach a loop-back communicator (vh_comm_create_loopback: the rank pushes into its own N "
                         "receive slots and reduces them — the stores, polls, tags and launches of the real exchange, no link)")
    ap.add_argument("--exchange", default="fused", choices=["fused", "kernel"],
                    help="with --loopback: form of the decode exchange — fused into the producer / consumer kernels (what ranks that own "
                         "their device vote for) or one all-reduce kernel per exchange")
    ap.add_argument("--frames", type=int, default=1,
                    help="debug: number of 448x448 tiles / video frames in the prompt (default 1 = configs[2]; 4-16 is the "
                         "video shape of configs[4])")
    ap.add_argument("--text-tokens", type=int, default=32, help="debug: length of the user text (default 32 = configs[2])")
    ap.add_argument("--batch", default="", help="also measure B concurrent sequences over the paged KV cache, e.g. 2,4 "
                    "(SURVEY 8(f)#1; reported under \"concurrent\", the headline value stays batch 1)")
    ap.add_argument("--tune", default="", help="debug: kernel-variant knobs key=val[,key=val] (vh_tune)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))       # no launcher around us: spawn one rank per GPU
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group(backend=args.backend if args.backend == "gloo" or torch.cuda.is_available() else "gloo")
            v = torch.tensor([float(rank + 1)])
            if dist.get_backend() == "nccl":
                torch.cuda.set_device(local_rank)
                v = v.cuda()
            dist.all_reduce(v)
            total = float(v.item())
            dist.barrier()
            dist.destroy_process_group()
        else:
            total = 1.0
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "allreduce_sum": total,
                              "expected": world * (world + 1) / 2}), flush=True)
        return
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")
            if not args.try_rccl and args.collective == "rccl":
                args.collective = "torch"  # RCCL refuses two ranks on one device; gloo stages through the host

    from vita_amd.checkpoint import synth_mixtral_device, synth_state_dict
    from vita_amd.config import VitaConfig, audio_token_count
    from vita_amd.model.vita_mixtral import VITAMixtralForCausalLM

    if args.tune:
        from vita_amd import _lib as _l
        for kv in args.tune.split(","):
            k, v = kv.split("=")
            _l.tune(k.strip(), int(v))
    cfg = VitaConfig()
    if args.layers:
        cfg.text.num_hidden_layers = args.layers
    t = cfg.text
    K, Wm = args.steps, args.warmup

    # ---- model: random-init weights of the released geometry --------------------------------------
    t0 = time.time()
    packed = synth_mixtral_device(cfg, dev, seed=0, rank=rank, world=args.emulate_tp or world)
    sd_enc = synth_state_dict(cfg, seed=1, rich=False, parts=("vision", "audio"))
    batches = [int(b) for b in args.batch.split(",") if b.strip()]
    max_prefill = max(1024, args.text_tokens + 768 + 256 * args.frames)
    seq_kw = {}
    if batches:   # a pool of 64-token KV pages holding max(B) sequences of prompt + generated tokens
        seq_kw = dict(max_seqs=max(batches), kv_pool_tokens=max(batches) * (-(-(max_prefill + K + Wm + 72) // 64) * 64))
    model = VITAMixtralForCausalLM(cfg, sd_enc, device=dev, packed_llm=packed, max_new_tokens=K + Wm + 8,
                                   max_prefill=max_prefill, rank=rank, world=world, keep_scores=False, **seq_kw)
    model.get_vision_tower().load_model()
    eng = model.engine
    collective = "none"
    loop_comm = None
    if args.loopback:
        if not args.emulate_tp or world > 1:
            raise SystemExit("--loopback needs --emulate-tp N on one process")
        from vita_amd import _lib as _l
        from vita_amd.parallel import IpcComm
        loop_comm = IpcComm(0, args.emulate_tp, t.hidden_size, loopback=True)
        eng.attach_comm(loop_comm)
        _l.tune("tp_fuse", 1 if args.exchange == "fused" else 0)
        eng.decode_exchange = args.exchange
        collective = f"loopback+{args.exchange}"
    if world > 1:
        from vita_amd.parallel import collective_label, setup_tensor_parallel
        collective = collective_label(eng, setup_tensor_parallel(eng, rank, world, dev, backend=args.backend, collective=args.collective))
    torch.cuda.synchronize()
    t_build = time.time() - t0

    # ---- synthetic request: 1 image tile + 10 s audio + text (configs[2]) -----------------------------
    from vita_amd.host.synthetic import make_request
    req = make_request(cfg, frames=args.frames, text_tokens=args.text_tokens)   # the request tests/test_realgeom_gpu.py checks
    image = torch.from_numpy(req["pixel_values"]).to(dev)
    feats = req["fbank"]                                                      # [998, 80]
    n_aud_tok = audio_token_count(feats.shape[0])
    ids = req["input_ids"]
    input_ids = torch.tensor([ids], dtype=torch.long, device=dev)
    audios = {"audios": torch.from_numpy(feats)[None].to(dev), "lengths": torch.tensor([feats.shape[0]], device=dev)}

    def ev():
        return torch.cuda.Event(enable_timing=True)

    emb_last = None

    def encode_and_prefill():
        e = [ev() for _ in range(5)]
        e[0].record()
        img_feats = model.encode_images(image)
        e[1].record()
        aud = model.get_audio_encoder()(audios["audios"], audios["lengths"])
        e[2].record()
        _, _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(input_ids, None, None, None, None, image, audios)
        e[3].record()
        eng.prefill(emb[0])
        e[4].record()
        torch.cuda.synchronize()
        nonlocal emb_last
        emb_last = emb[0]
        return emb.shape[1], {"vit_proj_ms": e[0].elapsed_time(e[1]), "audio_ms": e[1].elapsed_time(e[2]),
                              # what a request pays in front of the prefill: both towers (the audio tower on a side stream under the
                              # vision tower since r06), the projector and the embedding splice — the reference's
                              # prepare_inputs_labels_for_multimodal
                              "encode_ms": e[2].elapsed_time(e[3]),
                              "prefill_ms": e[3].elapsed_time(e[4]), "n_audio_tokens": int(aud["inputs_embeds"].shape[1])}

    gpu_state = GpuStateSampler(local_rank)
    for _ in range(max(1, args.phase_warmup)):                         # warm-up of the encoder + prefill path
        S, _ = encode_and_prefill()
    gpu_state.start("prefill_phase")
    runs = [encode_and_prefill()[1] for _ in range(max(1, args.phase_iters))]
    gpu_state.stop()
    if gpu_state.files and not args.layers:
        # clock / power settle over hundreds of ms (and the hwmon power is a moving average): keep the prefill running for
        # ~1.5 s and keep only what was sampled in the second half — the state the timed prefill passes converge to
        gpu_state.start("prefill_steady")
        t_end = time.perf_counter() + 1.5
        while time.perf_counter() < t_end:
            eng.prefill(emb_last)
            torch.cuda.synchronize()
        gpu_state.stop()
        buf = gpu_state.samples["prefill_steady"]
        del buf[:len(buf) // 2]
    # the dominant PREFILL kernel (MoE gate|up grouped GEMM), timed live with HIP events the engine records on its own stream
    # around each of the layers' launches (the prefill half of the metric recomputable from this line alone)
    eng.profile(stride=-1, max_samples=4 * t.num_hidden_layers + 8)
    for _ in range(3):
        eng.prefill(emb_last)
    torch.cuda.synchronize()
    pf_ms, pf_n = eng.profile_read()
    eng.profile(stride=0)
    phase = {k: float(np.median([r[k] for r in runs])) for k in ("vit_proj_ms", "audio_ms", "encode_ms", "prefill_ms")}
    phase_min = {k: float(min(r[k] for r in runs)) for k in ("vit_proj_ms", "audio_ms", "encode_ms", "prefill_ms")}
    assert runs[-1]["n_audio_tokens"] == n_aud_tok                     # the last run's KV cache feeds the decode below

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- decode: W warm-up steps, then exactly K timed steps ---------------------------------------------
    eng.decode(Wm)
    eng.profile(stride=args.profile_stride, max_samples=K * 8 + 8)       # sample the gate|up GEMV of every 4th layer (every 8th would save 0.3 % but weights layer 0, the slowest, twice as much)
    barrier()
    gpu_state.start("decode_phase")
    t1 = time.perf_counter()
    eng.decode(K)
    t_host = time.perf_counter() - t1                  # the host's share: enqueueing K steps (the device runs behind it)
    barrier()
    dt = time.perf_counter() - t1
    gpu_state.stop()
    tot_ms, n_samp = eng.profile_read()
    eng.profile(stride=0)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    toks = eng.generated()
    assert len(toks) == 1 + Wm + K

    ms_step = dt * 1e3 / K
    tok_s = K / dt

    # ---- the demo's own call: model.generate() with the keyword stopping criterion, end to end ------------------------------
    gen = None
    if not args.emulate_tp:
        from vita_amd.host.prompt import KeywordsStoppingCriteria
        n_new = K + Wm                                                   # fits the engine's max_new (K + Wm + 8)
        crit = KeywordsStoppingCriteria(["</s>"], _BenchTokenizer(), input_ids)
        model.generate(input_ids, images=image, audios=audios, do_sample=False, num_beams=1, max_new_tokens=4,
                       use_cache=True, stopping_criteria=[crit], eos_token_id=-1)         # warm-up of the host path
        barrier()
        t3 = time.perf_counter()
        out_g = model.generate(input_ids, images=image, audios=audios, do_sample=False, temperature=0.01, top_p=None, num_beams=1,
                               return_dict_in_generate=True, max_new_tokens=n_new, use_cache=True, stopping_criteria=[crit],
                               eos_token_id=-1)
        barrier()
        dtg = time.perf_counter() - t3
        if world > 1:
            tt = torch.tensor([dtg], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtg = float(tt.item())
        n_gen = int(out_g.sequences.shape[1] - input_ids.shape[1])
        lt = dict(model.last_timing)
        dec_ms = dtg * 1e3 - lt["encode_ms"] - phase["prefill_ms"]       # everything after the first token
        gen = {"new_tokens": n_gen, "total_ms": round(dtg * 1e3, 2), "encode_ms": round(lt["encode_ms"], 2),
               "tokens_per_s_end_to_end": round(n_gen / dtg, 2),
               "tokens_per_s_after_first_token": round((n_gen - 1) / (dec_ms * 1e-3), 2) if dec_ms > 0 and n_gen > 1 else None,
               "lookahead": int(model.lookahead), "stopping_criteria": "KeywordsStoppingCriteria(['</s>'])"}
        assert out_g.sequences[0, input_ids.shape[1]:].tolist() == toks[:n_gen], "generate() and the engine loop disagree"

    # ---- concurrent sequences (continuous-batching iterations over the paged KV cache) ----------------------------
    concurrent = []
    for B in batches:
        eng.lib.vh_mixtral_reset(eng.h, None)
        torch.cuda.synchronize()
        seqs = []
        for i in range(B):
            ids_i = [x if x < 0 else 3 + (x + 17 * i) % (t.vocab_size - 3) for x in ids]      # B different prompts
            _, _, _, _, emb_i, _ = model.prepare_inputs_labels_for_multimodal(
                torch.tensor([ids_i], dtype=torch.long, device=dev), None, None, None, None, image, audios)
            sq = eng.seq_alloc()
            eng.seq_prefill(sq, emb_i[0])
            seqs.append(sq)
        for _ in range(Wm):
            eng.seq_decode(seqs)
        barrier()
        t2 = time.perf_counter()
        for _ in range(K):
            eng.seq_decode(seqs)
        barrier()
        dtb = time.perf_counter() - t2
        if world > 1:
            tt = torch.tensor([dtb], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtb = float(tt.item())
        for sq in seqs:
            c = eng.check_device_flag(eng.seq_counters(sq).tolist())
            assert c[1] == 1 + Wm + K
            eng.seq_free(sq)
        concurrent.append({"batch": B, "aggregate_tokens_per_s": round(B * K / dtb, 2),
                           "ms_per_iteration": round(dtb * 1e3 / K, 4), "vs_batch1": round(B * K / dtb / tok_s, 3)})
    lay0 = packed["layers"][0]
    I_r = lay0["w1"].shape[1]
    gateup_bytes = 2 * 2 * I_r * t.hidden_size * 2 + t.num_local_experts * t.hidden_size * 2   # per launch, this rank
    k_ms = tot_ms / max(n_samp, 1)
    achieved = gateup_bytes / (k_ms * 1e-3) / 1e9 if n_samp else None
    ctx_mid = S + Wm + K // 2
    heads_r = lay0["wqkv"].shape[0] // t.head_dim                      # q + 2*kv heads on this rank
    nkv_r = heads_r * t.num_key_value_heads // (t.num_attention_heads + 2 * t.num_key_value_heads)
    per_layer_w = 2 * (lay0["wqkv"].numel() + lay0["wo"].numel() + 2 * 3 * I_r * t.hidden_size + lay0["wrouter"].numel())
    step_bytes = t.num_hidden_layers * per_layer_w + 2 * packed["lm_head"].numel()   # algorithmic weight bytes / token
    kv_bytes = t.num_hidden_layers * 2 * nkv_r * t.head_dim * 4 * ctx_mid            # fp32 KV cache read / token
    eff = (step_bytes + kv_bytes) / (ms_step * 1e-3) / 1e9
    prefill_bytes = int(t.num_hidden_layers * 2 * (lay0["wqkv"].numel() + lay0["wo"].numel() + lay0["wrouter"].numel() +
                                                    t.num_local_experts * 3 * I_r * t.hidden_size)
                        + 2 * packed["lm_head"].numel())               # all experts are touched at S >> 8

    rf_prefill = prefill_kernel_roofline(int(S), t.num_local_experts, I_r, t.hidden_size, t.num_hidden_layers, pf_ms, pf_n, world)
    if rank == 0:
        out = {
            "metric": "decode_tokens_per_s", "value": round(tok_s, 3), "unit": "tokens/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": round(ms_step, 4), "host_enqueue_ms_per_step": round(t_host * 1e3 / K, 4),
            "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16 weights, f32 activations/accumulate",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: 1 image (448x448, 1 tile -> 256 tokens)" if args.frames == 1 else
                                    f"video-shaped prompt: {args.frames} frames (448x448, 1 tile each -> {256 * args.frames} tokens)")
                                   + f" + 10 s audio (998 fbank frames -> {n_aud_tok} tokens) + "
                                   f"{len(ids) - args.frames - 1} text ids, prefill S={S}, "
                                   "greedy decode, batch 1; VITA-Mixtral-8x7B geometry (32 layers, 8 experts top-2)",
                       "parallelism": f"tp{world}", "collective": collective, "prompt_tokens": int(S),
                       "layers": t.num_hidden_layers,
                       # how the timed steps ran a layer's attention block: "fused-attention-block" = ONE launch (QKV rows, attention tiles
                       # and O rows as work items with granule hand-offs), "three-launches" = QKV, attention, O projection
                       "decode_schedule": eng.decode_schedule(),
                       "launches_per_layer": 3 if eng.decode_schedule() == "fused-attention-block" else 5},
            "prefill_ms": round(phase["prefill_ms"], 3), "vit_projector_ms": round(phase["vit_proj_ms"], 3),
            "audio_encoder_ms": round(phase["audio_ms"], 3),
            # time to first token of one request: encoders (concurrent) + projector + splice, then the prefill; `ttft_serial_ms` is the
            # r01-r05 definition (each tower timed alone, summed, splice not counted)
            "encode_ms": round(phase["encode_ms"], 3),
            "ttft_ms": round(phase["prefill_ms"] + phase["encode_ms"], 3),
            "ttft_serial_ms": round(phase["prefill_ms"] + phase["vit_proj_ms"] + phase["audio_ms"], 3),
            "phase_min_ms": {k: round(v, 3) for k, v in phase_min.items()}, "phase_iters": len(runs),
            "prefill_roofline": {"bound": "hbm", "algorithmic_bytes": prefill_bytes,
                                 "floor_ms": round(prefill_bytes / HBM_PEAK_GBPS / 1e6, 3),
                                 "frac": round(prefill_bytes / HBM_PEAK_GBPS / 1e6 / phase["prefill_ms"], 4),
                                 "note": "every expert is touched at this S: all backbone weights read once per rank "
                                         "(SURVEY 8(d)); MFMA floor is lower"},
            "decode_effective_GBps_per_gpu": round(eff, 1),
            "decode_effective_frac_of_8TBps": round(eff / HBM_PEAK_GBPS, 4),
            "roofline": {"bound": "hbm", "kernel": "k_dec_gateup (router + gate|up GEMV of the 2 routed experts)",
                         "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4) if achieved else None,
                         # the same fraction under its two clocks: live HIP events around the launch inside THIS run's timed steps (they
                         # include the launch gap) and the kernel's own duration in the committed rocprofv3 --kernel-trace run of this command
                         "frac_live_events": round(achieved / HBM_PEAK_GBPS, 4) if achieved else None,
                         "frac_kernel_trace": (round(gateup_bytes / (kernel_trace_us("k_dec_gateup")[0] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
                                               if world == 1 and not args.emulate_tp and kernel_trace_us("k_dec_gateup")[0] else None),
                         "kernel_trace_source": kernel_trace_us("k_dec_gateup")[1] if world == 1 and not args.emulate_tp else None,
                         "bytes_per_launch": gateup_bytes, "avg_launch_us": round(k_ms * 1e3, 2), "samples": n_samp,
                         "traffic": pmc_traffic("k_dec_gateup") if world == 1 else None,
                         "traffic_source": (f"{traffic_source()} (static: rocprofv3 --pmc FETCH_SIZE pass of this kernel, "
                                            "counters cannot be read inside this process)") if world == 1 else None},
            "roofline_prefill": rf_prefill,
            "gpu_state": gpu_state.summary(),
            "build_s": round(t_build, 1),
        }
        if gen:
            out["generate_tokens_per_s"] = gen["tokens_per_s_after_first_token"]
            out["generate"] = gen
        if concurrent:
            out["concurrent"] = concurrent
        if args.layers:
            out["INVALID_debug_layers"] = args.layers
        if args.tune:
            out["tune"] = args.tune
        if args.emulate_tp:
            out["INVALID_emulated_tp_rank_compute_only"] = args.emulate_tp
            out["emulated_tp"] = {"world": args.emulate_tp, "exchanges": ("looped back into this rank's own receive slots: 2 per layer + the head's "
                                                                           "candidate exchange, " + args.exchange + " form") if args.loopback else "skipped",
                                  "decode_schedule": eng.decode_schedule(), "comm_status": loop_comm.status() if loop_comm else None}
        if args.text_tokens != 32:
            out["INVALID_debug_text_tokens"] = args.text_tokens
        if args.frames != 1:
            out["OTHER_WORKLOAD_frames"] = args.frames      # not the metric's configuration: video-shaped prompt
        if args.one_device or args.backend != "nccl":
            out["INVALID_debug_backend"] = f"{args.backend}, one_device={args.one_device}"
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed once, at N=1 (torchrun also pins OMP to 1 thread)
            try:
                out["cpu_baseline"] = cpu_baseline(VitaConfig(), ctx=int(S), request=req)
            except Exception as e:  # never lose the GPU line to a host-side problem
                out["cpu_baseline"] = {"value": None, "error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
