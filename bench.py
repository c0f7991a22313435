#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: greedy-decode tokens/s (+ prefill / encoder ms) of the
VITA-Mixtral-8x7B geometry on a 1 image + 10 s audio + text prompt, tensor-parallel over N GPUs.

  python bench.py --gpus 1 --steps 64 --warmup 8
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one greedy decode step (one pass of the decode hot path over the whole model for one
token).  Warm-up steps are untimed; exactly K steps are timed between barrier + synchronize pairs,
inputs (weights, KV cache, prompt) resident in HBM; value = K / max-over-ranks time.  Rank 0 prints
ONE JSON line.  Data and weights are synthetic (no checkpoint offline): seeded N(0, 0.02) weights of
the released geometry, a random 448x448 image, a seeded 10 s waveform, random prompt ids."""
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


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC pass (FETCH_SIZE, corrected as
    MI355X_MICROARCH.md prescribes; collected by profiles/pmc_pass.sh in its own run — counters cannot be
    read from inside this process).  TP=1 only: the launch moves 1/N of the bytes at TP=N."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as f:
            return int(json.load(f)[kernel]["hbm_read_bytes_per_launch"])
    except Exception:
        return None


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


def cpu_baseline(cfg, n_layers=2, ctx=64, n_tok=6):
    """The oracle (numpy port of the HF Mixtral arithmetic the reference calls) timed on this box's
    host cores: `n_layers` real-geometry decoder layers + LM head, decode steps at a short context,
    extrapolated to 32 layers.  Bounded to a few seconds of weight generation + ~10-20 s of compute."""
    import copy
    from oracle import mixtral as om
    t = copy.deepcopy(cfg.text)
    t.num_hidden_layers = n_layers
    rng = np.random.default_rng(0)
    H, I, E, hd = t.hidden_size, t.intermediate_size, t.num_local_experts, t.head_dim

    def W(*shape):  # cheap uniform init (values do not matter for timing), distinct memory per tensor
        return (rng.random(shape, dtype=np.float32) - 0.5) * np.float32(0.07)

    sd = {"model.embed_tokens.weight": W(1024, H), "model.norm.weight": np.ones(H, np.float32),
          "lm_head.weight": W(t.vocab_size, H)}
    for l in range(n_layers):
        p = f"model.layers.{l}."
        sd[p + "input_layernorm.weight"] = np.ones(H, np.float32)
        sd[p + "post_attention_layernorm.weight"] = np.ones(H, np.float32)
        sd[p + "self_attn.q_proj.weight"] = W(t.num_attention_heads * hd, H)
        sd[p + "self_attn.k_proj.weight"] = W(t.num_key_value_heads * hd, H)
        sd[p + "self_attn.v_proj.weight"] = W(t.num_key_value_heads * hd, H)
        sd[p + "self_attn.o_proj.weight"] = W(H, t.num_attention_heads * hd)
        sd[p + "block_sparse_moe.gate.weight"] = W(E, H)
        for e in range(E):
            q = p + f"block_sparse_moe.experts.{e}."
            sd[q + "w1.weight"], sd[q + "w2.weight"], sd[q + "w3.weight"] = W(I, H), W(H, I), W(I, H)
    orc = om.MixtralOracle(sd, t)
    del sd
    x = W(ctx, H)
    orc.forward(x)                                   # prefill the oracle's KV cache (untimed)
    tok = W(1, H)

    def timed():
        t_full, t_head = [], []
        for _ in range(n_tok):
            t0 = time.perf_counter()
            orc.forward(tok)
            t_full.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            _ = (om.rmsnorm(tok, orc.norm, t.rms_norm_eps) @ orc.lm_head.T)
            t_head.append(time.perf_counter() - t0)
        full, head = float(np.median(t_full[1:])), float(np.median(t_head[1:]))
        per_layer = max(full - head, 1e-9) / n_layers
        return 1.0 / (per_layer * cfg.text.num_hidden_layers + head), per_layer, head

    # batch-1 GEMVs are memory-bound: the BLAS pool's default (all cores) is not the fastest setting on a
    # many-core host, so a few thread counts are tried and the BEST one is the reported baseline
    sweep = {}
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        all_thr = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        for thr in sorted({all_thr, min(all_thr, 32), min(all_thr, 8), 1}, reverse=True):
            with threadpool_limits(limits=thr):
                sweep[thr] = timed()
    except ImportError:
        sweep[os.cpu_count() or 1] = timed()
    threads = max(sweep, key=lambda k: sweep[k][0])
    tok_s, per_layer, head = sweep[threads]
    return {"value": round(tok_s, 4), "unit": "tokens/s", "cores": int(threads), "kind": "port",
            "sample": f"numpy fp32 oracle: {n_layers} of {cfg.text.num_hidden_layers} real-geometry decoder layers + "
                      f"LM head, {n_tok - 1} timed decode steps at ctx {ctx}, median, best of the BLAS thread counts tried, extrapolated x"
                      f"{cfg.text.num_hidden_layers}/{n_layers} (the full fp32 model is 187 GB)",
            "ms_per_layer_token": round(per_layer * 1e3, 3), "ms_lm_head": round(head * 1e3, 3),
            "tokens_per_s_by_threads": {str(k): round(v[0], 4) for k, v in sweep.items()}}


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
    ap.add_argument("--emulate-tp", type=int, default=0,
                    help="debug: run ONE rank's 1/N shard on one GPU with the collective skipped — per-rank compute "
                         "time of TP=N without communication (tokens are meaningless, result marked invalid)")
    ap.add_argument("--frames", type=int, default=1,
                    help="debug: number of 448x448 tiles / video frames in the prompt (default 1 = configs[2]; 4-16 is the "
                         "video shape of configs[4])")
    ap.add_argument("--text-tokens", type=int, default=32, help="debug: length of the user text (default 32 = configs[2])")
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
    model = VITAMixtralForCausalLM(cfg, sd_enc, device=dev, packed_llm=packed, max_new_tokens=K + Wm + 8,
                                   max_prefill=max(1024, args.text_tokens + 768 + 256 * args.frames), rank=rank, world=world, keep_scores=False)
    model.get_vision_tower().load_model()
    eng = model.engine
    collective = "none"
    if world > 1:
        from vita_amd.parallel import setup_tensor_parallel
        collective = setup_tensor_parallel(eng, rank, world, dev, backend=args.backend, collective=args.collective)
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
        return emb.shape[1], {"vit_proj_ms": e[0].elapsed_time(e[1]), "audio_ms": e[1].elapsed_time(e[2]),
                              "prefill_ms": e[3].elapsed_time(e[4]), "n_audio_tokens": int(aud["inputs_embeds"].shape[1])}

    for _ in range(max(1, args.phase_warmup)):                         # warm-up of the encoder + prefill path
        S, _ = encode_and_prefill()
    runs = [encode_and_prefill()[1] for _ in range(max(1, args.phase_iters))]
    phase = {k: float(np.median([r[k] for r in runs])) for k in ("vit_proj_ms", "audio_ms", "prefill_ms")}
    phase_min = {k: float(min(r[k] for r in runs)) for k in ("vit_proj_ms", "audio_ms", "prefill_ms")}
    assert runs[-1]["n_audio_tokens"] == n_aud_tok                     # the last run's KV cache feeds the decode below

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- decode: W warm-up steps, then exactly K timed steps ---------------------------------------------
    eng.decode(Wm)
    eng.profile(stride=4, max_samples=K * 8 + 8)       # sample the gate|up GEMV of every 4th layer
    barrier()
    t1 = time.perf_counter()
    eng.decode(K)
    barrier()
    dt = time.perf_counter() - t1
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

    if rank == 0:
        out = {
            "metric": "decode_tokens_per_s", "value": round(tok_s, 3), "unit": "tokens/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16 weights, f32 activations/accumulate",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: 1 image (448x448, 1 tile -> 256 tokens)" if args.frames == 1 else
                                    f"video-shaped prompt: {args.frames} frames (448x448, 1 tile each -> {256 * args.frames} tokens)")
                                   + f" + 10 s audio (998 fbank frames -> {n_aud_tok} tokens) + "
                                   f"{len(ids) - args.frames - 1} text ids, prefill S={S}, "
                                   "greedy decode, batch 1; VITA-Mixtral-8x7B geometry (32 layers, 8 experts top-2)",
                       "parallelism": f"tp{world}", "collective": collective, "prompt_tokens": int(S),
                       "layers": t.num_hidden_layers},
            "prefill_ms": round(phase["prefill_ms"], 3), "vit_projector_ms": round(phase["vit_proj_ms"], 3),
            "audio_encoder_ms": round(phase["audio_ms"], 3),
            "ttft_ms": round(phase["prefill_ms"] + phase["vit_proj_ms"] + phase["audio_ms"], 3),
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
                         "bytes_per_launch": gateup_bytes, "avg_launch_us": round(k_ms * 1e3, 2), "samples": n_samp,
                         "traffic": pmc_traffic("k_dec_gateup") if world == 1 else None},
            "build_s": round(t_build, 1),
        }
        if args.layers:
            out["INVALID_debug_layers"] = args.layers
        if args.tune:
            out["tune"] = args.tune
        if args.emulate_tp:
            out["INVALID_emulated_tp_rank_compute_only"] = args.emulate_tp
        if args.text_tokens != 32:
            out["INVALID_debug_text_tokens"] = args.text_tokens
        if args.frames != 1:
            out["OTHER_WORKLOAD_frames"] = args.frames      # not the metric's configuration: video-shaped prompt
        if args.one_device or args.backend != "nccl":
            out["INVALID_debug_backend"] = f"{args.backend}, one_device={args.one_device}"
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed once, at N=1 (torchrun also pins OMP to 1 thread)
            try:
                out["cpu_baseline"] = cpu_baseline(VitaConfig())
            except Exception as e:  # never lose the GPU line to a host-side problem
                out["cpu_baseline"] = {"value": None, "error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
