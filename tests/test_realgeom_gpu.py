"""Parity on the HEADLINE configuration (BASELINE configs[2], the request bench.py times) at the RELEASED geometry:
24-layer InternViT + projector, 24-layer Whale + adapter, 32-layer Mixtral-8x7B with the full vocabulary, prefill
S = 552 and 16 greedy steps — device vs the layer-streamed fp32 oracle (oracle/stream.py) on identical weights
(counter-based generator, same integers on both sides).  SURVEY 8(c) golden list: encoder outputs, spliced
inputs_embeds, hidden states after layers 0 / 15 / 31, router top-2 ids of every layer, last-row logits within 1e-3,
greedy ids bit-exact (vita/model/language_model/vita_mixtral.py:158-173; video_audio_demo.py:257-270).

VITA_REALGEOM_LAYERS=n shortens the backbone for a quick run (default: all 32 layers)."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import encoders as oe, hashw, stream
from tests.util import assert_close, report, to_np
from vita_amd.checkpoint import synth_mixtral_device, synth_state_dict
from vita_amd.config import VitaConfig
from vita_amd.host.synthetic import make_request

pytestmark = pytest.mark.gpu
T_NEW = 16
SEED = 0


@pytest.fixture(scope="module")
def run(dev):
    from vita_amd.model.vita_mixtral import VITAMixtralForCausalLM
    cfg = VitaConfig()
    cfg.text.num_hidden_layers = int(os.environ.get("VITA_REALGEOM_LAYERS", cfg.text.num_hidden_layers))
    t0 = time.time()
    packed = synth_mixtral_device(cfg, dev, seed=SEED)
    sd_enc = synth_state_dict(cfg, seed=1, rich=False, parts=("vision", "audio"))
    model = VITAMixtralForCausalLM(cfg, sd_enc, device=dev, packed_llm=packed, max_new_tokens=T_NEW + 8, max_prefill=1024,
                                   keep_scores=True)
    model.get_vision_tower().load_model()
    req = make_request(cfg)
    pix = torch.from_numpy(req["pixel_values"]).to(dev)
    feats = torch.from_numpy(req["fbank"]).to(dev)
    ids = torch.tensor([req["input_ids"]], dtype=torch.long, device=dev)
    audios = {"audios": feats[None], "lengths": torch.tensor([feats.shape[0]], device=dev)}
    vit = model.get_vision_tower()(pix)
    img = model.model.mm_projector(vit)
    aud = model.get_audio_encoder()(audios["audios"], audios["lengths"])
    _, _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix, audios)
    eng = model.engine
    _, hid = eng.prefill(emb[0], want_hidden=True, want_route=True)
    route = eng.route_ids
    eng.decode(T_NEW - 1)
    torch.cuda.synchronize()
    out = dict(cfg=cfg, sd_enc=sd_enc, req=req, vit=to_np(vit), img=to_np(img), aud=to_np(aud["inputs_embeds"]),
               aud_mask=to_np(aud["attention_mask"]), emb=to_np(emb[0]), toks=eng.generated(),
               logits=to_np(eng.logits_all[:T_NEW]), route=route.cpu().numpy(),
               hidden={l: to_np(hid[l]) for l in {0, min(15, cfg.text.num_hidden_layers - 1), cfg.text.num_hidden_layers - 1}})
    print(f"[realgeom] device side done in {time.time() - t0:.1f}s: S={emb.shape[1]}, tokens {out['toks']}")
    model.engine.close()
    del model, packed, hid
    torch.cuda.empty_cache()
    return out


@pytest.fixture(scope="module")
def oracle_embeds(run):
    """the oracle's own encoders and splice -> inputs_embeds of the prompt."""
    cfg, sd, req = run["cfg"], run["sd_enc"], run["req"]
    t0 = time.time()
    vit = oe.internvit_tower(sd, cfg.vision, req["pixel_values"])
    img = oe.projector(sd, vit)
    aud, amask = oe.whale_encoder(sd, cfg.audio, req["fbank"])[:2]
    table = hashw.fill((cfg.text.vocab_size, cfg.text.hidden_size), hashw.tensor_seed("model.embed_tokens.weight", SEED))
    emb = oe.splice(np.asarray(req["input_ids"]), table, img, aud[None] if aud.ndim == 2 else aud)
    print(f"[realgeom] oracle encoders + splice in {time.time() - t0:.1f}s")
    return dict(vit=vit, img=img, aud=aud, emb=np.asarray(emb, np.float32))


def test_encoders_full_depth(run, oracle_embeds):
    """A8-A10 at full depth (24 + 24 layers) on the bench's image and 10 s clip."""
    o = oracle_embeds
    assert run["vit"].shape == (1, 256, 4096) and run["aud"].shape == (1, 124, 4096) and run["aud_mask"].all()
    assert_close("InternViT (24 layers) + pixel shuffle", run["vit"], o["vit"], atol=2e-3, rtol=1e-3)
    assert_close("projector", run["img"], o["img"], atol=2e-3, rtol=1e-3)
    assert_close("Whale (24 layers) + adapter", run["aud"][0], o["aud"].reshape(124, 4096), atol=2e-3, rtol=1e-3)
    assert_close("spliced inputs_embeds", run["emb"], o["emb"], atol=2e-3, rtol=1e-3)


@pytest.fixture(scope="module")
def oracle_ref(run, oracle_embeds):
    """ONE teacher-forced pass of the layer-streamed fp32 oracle over prompt + the device's generated tokens; shared by the
    single-GPU test and the TP = 8 test below (the same request, and — when both are right — the same tokens)."""
    cfg = run["cfg"]
    t, L = cfg.text, cfg.text.num_hidden_layers
    S = run["emb"].shape[0]
    toks = run["toks"]
    cap = sorted(run["hidden"])
    full = np.concatenate([oracle_embeds["emb"], stream.embed_rows(t, toks[:-1], SEED)], 0)
    t0 = time.time()
    ref = stream.forward(t, SEED, full, n_layers=L, capture=cap, logits_from=S - 1, verbose=True)
    print(f"[realgeom] oracle backbone ({L} layers, {full.shape[0]} rows) in {time.time() - t0:.1f}s")
    return ref


def test_backbone_32_layers_prefill_and_greedy(run, oracle_embeds, oracle_ref):
    """A11-A14: one teacher-forced oracle forward over prompt + generated tokens vs the device's prefill + 15 decode steps."""
    cfg = run["cfg"]
    t, L = cfg.text, cfg.text.num_hidden_layers
    S = run["emb"].shape[0]
    toks = run["toks"]
    assert S == 552 and len(toks) == T_NEW
    cap = sorted(run["hidden"])
    ref = oracle_ref
    # router decisions of every layer over the prompt rows
    r_dev, r_ref = np.sort(run["route"], -1), np.sort(ref["route"][:, :S], -1)
    mism = np.argwhere((r_dev != r_ref).any(-1))
    print(f"router top-2 sets: {r_dev.shape[0] * S} decisions, {len(mism)} differ", mism[:5].tolist())
    assert len(mism) == 0
    for l in cap:
        # both sides accumulate in fp32 in different orders: the tolerance follows the residual stream's scale
        # (|x| grows to ~40 by layer 31), 3e-4 of the tensor's largest magnitude + 1e-3 relative
        h_ref = ref["hidden"][l][:S]
        assert_close(f"hidden after layer {l}", run["hidden"][l], h_ref, atol=3e-4 * float(np.abs(h_ref).max()), rtol=1e-3)
    # the drift is linear in depth (one fp32 re-association per layer): its SLOPE is asserted, not only its end point
    # (the 3e-4 * max|ref| bound is 1.4e-2 absolute at layer 31 against 5.3e-3 measured)
    if L >= 32:
        errs = {l: float(np.abs(run["hidden"][l] - ref["hidden"][l][:S]).max()) for l in cap}
        print("hidden-state error by depth:", {l: f"{e:.2e}" for l, e in errs.items()})
        assert errs[L - 1] <= 2.5e-4 * L, f"layer {L - 1}: {errs[L - 1]:.2e} exceeds 2.5e-4 per layer"
        assert errs[15] <= 2.5e-4 * 16, f"layer 15: {errs[15]:.2e} exceeds 2.5e-4 per layer"
    ref_ids = ref["logits"].argmax(-1).tolist()
    print("device ids", toks)
    print("oracle ids", ref_ids)
    print(report("logits of the 16 steps", run["logits"], ref["logits"]))
    assert toks == ref_ids                                               # greedy ids bit-exact
    assert np.abs(run["logits"] - ref["logits"]).max() < 1e-3            # north-star: logits within 1e-3 (fp32)


# ---- BASELINE configs[3]: the same omni request under TP = 8 at FULL depth ------------------------------------------------------
TP8_NEW = 8


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tp8_omni_worker(rank, world, port, layers, enc_path, ret):
    """one rank of the released partition at `world` ranks (TP = 8: 4 q heads + 1 KV head, 1792 columns of every expert, 6470
    vocabulary rows; TP = 2, the degree both web demos run — web_demo/web_ability_demo.py:340-348 —: 16 q + 4 KV heads, 7168
    columns, 25880 rows; web_demo/vllm_tools/vllm_file/mixtral.py:375-414,441-476,939-951) running the WHOLE omni request:
    replicated encoders + projector, splice, sharded prefill (64 bulk all-reduces of 9 MB) and greedy decode (65 exchanges per
    token) over the library's IPC all-reduce, `world` processes on one GPU."""
    import torch.distributed as dist
    from vita_amd.model.vita_mixtral import VITAMixtralForCausalLM
    from vita_amd.parallel import setup_tensor_parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.pop("VITA_AMD_TP_FUSE", None)
    os.environ.pop("VITA_AMD_TP_TRIAL", None)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tw = [time.time()]
        lap = lambda: (tw.append(time.time()), tw[-1] - tw[-2])[1]
        cfg = VitaConfig()
        cfg.text.num_hidden_layers = layers
        packed = synth_mixtral_device(cfg, dev, seed=SEED, rank=rank, world=world)
        t_w = lap()
        # the replicated encoder weights: the parent's state dict, memory-mapped (generating 0.7 B values takes ~45 s per process)
        sd_enc = {k: v.numpy() for k, v in torch.load(enc_path, mmap=True, weights_only=True).items()}
        model = VITAMixtralForCausalLM(cfg, sd_enc, device=dev, packed_llm=packed, max_new_tokens=TP8_NEW + 8, max_prefill=640,
                                       rank=rank, world=world, keep_scores=True)
        eng = model.engine
        assert (eng.c.n_q_heads, eng.c.n_kv_heads, eng.c.inter) == (32 // world, 8 // world, 14336 // world)
        assert eng.c.vocab_n in (51760 // world, 51760 // world + 1)
        t_m = lap()
        name = setup_tensor_parallel(eng, rank, world, dev, backend="gloo", collective="ipc")
        t_tp = lap()
        model.get_vision_tower().load_model()
        req = make_request(cfg)
        pix = torch.from_numpy(req["pixel_values"]).to(dev)
        feats = torch.from_numpy(req["fbank"]).to(dev)
        ids = torch.tensor([req["input_ids"]], dtype=torch.long, device=dev)
        audios = {"audios": feats[None], "lengths": torch.tensor([feats.shape[0]], device=dev)}
        _, _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix, audios)
        torch.cuda.synchronize()
        t_enc = lap()
        eng.prefill(emb[0])
        torch.cuda.synchronize()
        t_pf = lap()
        eng.decode(TP8_NEW - 1)
        torch.cuda.synchronize()
        t_dec = lap()
        if rank == 0:
            print(f"[realgeom] TP = {world} rank 0: weights {t_w:.1f}s, model {t_m:.1f}s, collective bring-up {t_tp:.1f}s, encoders + splice "
                  f"{t_enc:.1f}s, prefill {t_pf:.1f}s, {TP8_NEW - 1} decode steps {t_dec:.1f}s", flush=True)
        lg = eng.logits_all[:TP8_NEW].cpu()
        dist.all_reduce(lg)                      # vocab-sharded head: rows hold this rank's slice, zeros elsewhere
        c = getattr(eng, "_comm", None)
        ret[rank] = (name, eng.generated(), lg.numpy() if rank == 0 else None, c.status() if c is not None else None,
                     eng.decode_exchange, int(emb.shape[1]), lg.numpy().tobytes() if rank > 0 else None, eng.decode_schedule())
        dist.barrier()
        time.sleep(0.5)                          # gloo: a rank that leaves the barrier first must not close its sockets under the others
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world", [8, 2])
def test_tp_full_depth_omni_matches_oracle(run, oracle_ref, tmp_path, world):
    """BASELINE configs[3] (world 8) and the reference's OWN deployment degree (world 2: web_demo/web_ability_demo.py:340-348,
    web_demo/web_interactive_demo.py:942-996; 2 x TP = 4 is configs[4]) — `world` engine processes on ONE GPU — at the released
    shard shapes, ALL layers, the omni request of make_request() (encoders + splice + prefill S = 552 + 8 greedy steps) over the
    IPC all-reduce: 64 all-reduces per forward of re-ordered fp32 sums in front of a discontinuous router (ranks sharing a device
    run the decode attention block as three launches; its one-launch form: tests/test_comm_gpu.py test_loopback_*).  Greedy ids == the streamed fp32
    oracle's and logits within 1e-3 (the oracle pass is the one test_backbone_32_layers_prefill_and_greedy paid for: same
    request, and the tokens must agree), every rank bit-identical, no spin time-out."""
    import torch.multiprocessing as mp
    cfg = run["cfg"]
    L = cfg.text.num_hidden_layers
    t0 = time.time()
    enc_path = str(tmp_path / "encoders.pt")
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in run["sd_enc"].items()}, enc_path)
    ret = mp.Manager().dict()
    mp.spawn(_tp8_omni_worker, args=(world, _free_port(), L, enc_path, ret), nprocs=world, join=True)
    print(f"[realgeom] TP = {world} ranks done in {time.time() - t0:.1f}s")
    names = {ret[r][0] for r in range(world)}
    assert len(names) == 1, {r: ret[r][0] for r in range(world)}
    if names != {"ipc"}:
        msg = f"the ranks agreed on {names}, not on the IPC all-reduce: configs[3] ran over gloo on this box"
        if os.environ.get("VITA_ALLOW_GLOO_FALLBACK", "") == "1":
            pytest.xfail(msg)
        pytest.fail(msg)
    assert all(ret[r][3] == 0 for r in range(world)), {r: ret[r][3] for r in range(world)}
    assert all(ret[r][5] == 552 for r in range(world))
    toks = ret[0][1]
    lg0 = ret[0][2]
    for r in range(1, world):
        assert ret[r][1] == toks, f"rank {r}: {ret[r][1]} vs rank 0 {toks}"
        assert ret[r][6] == lg0.tobytes(), f"rank {r} logits differ from rank 0's"
    ref_lg = oracle_ref["logits"][:TP8_NEW]
    ref_ids = ref_lg.argmax(-1).tolist()
    err = float(np.abs(lg0 - ref_lg).max())
    print(f"TP = {world}, {L} layers, S = 552 omni request: ids {toks}, oracle {ref_ids}, TP = 1 device {run['toks'][:TP8_NEW]}, "
          f"max |logit diff| {err:.2e}, exchange form {ret[0][4]}, decode schedule {ret[0][7]}")
    assert all(ret[r][7] == "three-launches" for r in range(world)), {r: ret[r][7] for r in range(world)}      # ranks sharing a device
    assert toks == ref_ids
    assert err < 1e-3
