"""Released-geometry parity on the REFERENCE'S OWN ASSETS (SURVEY 8(d) configs 2 / 3 variant (i); the request
video_audio_demo.py:180-226 builds from `--image_path asset/vita_log2.png --audio_path asset/q1.wav`):

  image  asset/vita_log2.png -> 4 tiles + thumbnail by the reference's dynamic_preprocess (tests/golden/assets_request.npz,
         written by oracle/make_golden_assets.py) -> the product's CLIP-style processor -> 24-layer InternViT + projector:
         5 x 256 image tokens
  audio  asset/q1.wav -> Kaldi fbank (tests/golden/q1_audio.npz: 352 frames) -> 24-layer Whale + adapter: 44 audio tokens
  text   the bench's stand-in ids (no tokenizer offline): 139 system + 32 user ids

S = 1 + 139 + 1280 + 32 + 44 = 1496 prompt rows through the Mixtral backbone at the released widths against the
layer-streamed fp32 oracle: encoder outputs, spliced embeddings, router top-2 sets of every layer and prompt row, hidden
states, the logits of 8 greedy steps (< 1e-3) and the greedy ids (==).

The backbone runs all 32 layers (r05; VITA_ASSETS_LAYERS=n shortens it for a quick run: the oracle streams 5.7 GB of fp32
weights per layer over 1500 rows, ~160 s at full depth)."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import encoders as oe, hashw, stream
from tests.util import assert_close, report, to_np
from vita_amd.checkpoint import synth_mixtral_device, synth_state_dict
from vita_amd.config import VitaConfig, audio_token_count

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
T_NEW = 8
SEED = 0
LAYERS = int(os.environ.get("VITA_ASSETS_LAYERS", "32"))


def assets_request(cfg):
    from PIL import Image
    from vita_amd.host.constants import AUDIO_TOKEN_INDEX, IMAGE_TOKEN_INDEX
    from vita_amd.host.image_processing import make_image_processor
    g = np.load(os.path.join(GOLD, "assets_request.npz"))
    tiles = [Image.fromarray(t) for t in g["tiles"]]
    pix = np.asarray(make_image_processor(cfg.vision.image_size).preprocess(tiles, return_tensors="np")["pixel_values"], np.float32)
    fbank = np.load(os.path.join(GOLD, "q1_audio.npz"))["fbank"].astype(np.float32)
    r1 = np.random.default_rng(1)
    sys_ids = r1.integers(3, 51000, size=139).tolist()
    txt_ids = r1.integers(3, 51000, size=32).tolist()
    # video_audio_demo.py:222-224: qs = DEFAULT_IMAGE_TOKEN * p_num[0] + "\n" + qs + DEFAULT_AUDIO_TOKEN
    ids = [cfg.text.bos_token_id] + sys_ids + [IMAGE_TOKEN_INDEX] * len(tiles) + txt_ids + [AUDIO_TOKEN_INDEX]
    return {"pixel_values": pix, "fbank": fbank, "input_ids": ids}


@pytest.fixture(scope="module")
def run(dev):
    from vita_amd.model.vita_mixtral import VITAMixtralForCausalLM
    cfg = VitaConfig()
    cfg.text.num_hidden_layers = LAYERS
    t0 = time.time()
    packed = synth_mixtral_device(cfg, dev, seed=SEED)
    sd_enc = synth_state_dict(cfg, seed=1, rich=False, parts=("vision", "audio"))
    model = VITAMixtralForCausalLM(cfg, sd_enc, device=dev, packed_llm=packed, max_new_tokens=T_NEW + 8, max_prefill=1536,
                                   keep_scores=True)
    model.get_vision_tower().load_model()
    req = assets_request(cfg)
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
               hidden={l: to_np(hid[l]) for l in {0, LAYERS - 1}})
    print(f"[assets] device side done in {time.time() - t0:.1f}s: S={emb.shape[1]}, tokens {out['toks']}")
    model.engine.close()
    del model, packed, hid
    torch.cuda.empty_cache()
    return out


@pytest.fixture(scope="module")
def oracle_embeds(run):
    cfg, sd, req = run["cfg"], run["sd_enc"], run["req"]
    t0 = time.time()
    vit = oe.internvit_tower(sd, cfg.vision, req["pixel_values"])
    img = oe.projector(sd, vit)
    aud, _ = oe.whale_encoder(sd, cfg.audio, req["fbank"])[:2]
    table = hashw.fill((cfg.text.vocab_size, cfg.text.hidden_size), hashw.tensor_seed("model.embed_tokens.weight", SEED))
    emb = oe.splice(np.asarray(req["input_ids"]), table, img, aud[None] if aud.ndim == 2 else aud)
    print(f"[assets] oracle encoders (5 tiles, 352 frames) + splice in {time.time() - t0:.1f}s")
    return dict(vit=vit, img=img, aud=aud, emb=np.asarray(emb, np.float32))


def test_assets_encoders_and_splice(run, oracle_embeds):
    """A8-A10 + A7 at full depth on the reference's image (5 tiles in one batch) and clip (T = 352 -> 87 -> 44)."""
    o = oracle_embeds
    n_aud = audio_token_count(run["req"]["fbank"].shape[0])
    assert run["req"]["fbank"].shape == (352, 80) and n_aud == 44
    assert run["vit"].shape == (5, 256, 4096) and run["aud"].shape == (1, 44, 4096) and run["aud_mask"].all()
    assert_close("InternViT (24 layers, 5 tiles) + pixel shuffle", run["vit"], o["vit"], atol=2e-3, rtol=1e-3)
    assert_close("projector", run["img"], o["img"], atol=2e-3, rtol=1e-3)
    assert_close("Whale (24 layers, q1.wav) + adapter", run["aud"][0], o["aud"].reshape(44, 4096), atol=2e-3, rtol=1e-3)
    assert run["emb"].shape == (1 + 139 + 5 * 256 + 32 + 44, 4096)
    assert_close("spliced inputs_embeds", run["emb"], o["emb"], atol=2e-3, rtol=1e-3)


def test_assets_backbone_prefill_and_greedy(run, oracle_embeds):
    """A11-A14 on the S = 1496 prompt: one teacher-forced oracle forward over prompt + generated tokens vs the device's
    prefill + 7 decode steps (experts see ~375 rows each: two m-tiles per expert in the streaming GEMM)."""
    cfg = run["cfg"]
    t, L = cfg.text, cfg.text.num_hidden_layers
    S = run["emb"].shape[0]
    toks = run["toks"]
    assert S == 1496 and len(toks) == T_NEW
    cap = sorted(run["hidden"])
    full = np.concatenate([oracle_embeds["emb"], stream.embed_rows(t, toks[:-1], SEED)], 0)
    t0 = time.time()
    ref = stream.forward(t, SEED, full, n_layers=L, capture=cap, logits_from=S - 1, verbose=True)
    print(f"[assets] oracle backbone ({L} layers, {full.shape[0]} rows) in {time.time() - t0:.1f}s")
    r_dev, r_ref = np.sort(run["route"], -1), np.sort(ref["route"][:, :S], -1)
    mism = np.argwhere((r_dev != r_ref).any(-1))
    print(f"router top-2 sets: {r_dev.shape[0] * S} decisions, {len(mism)} differ", mism[:5].tolist())
    assert len(mism) == 0
    for l in cap:
        h_ref = ref["hidden"][l][:S]
        assert_close(f"hidden after layer {l}", run["hidden"][l], h_ref, atol=3e-4 * float(np.abs(h_ref).max()), rtol=1e-3)
    ref_ids = ref["logits"].argmax(-1).tolist()
    print("device ids", toks)
    print("oracle ids", ref_ids)
    print(report(f"logits of the {T_NEW} steps", run["logits"], ref["logits"]))
    assert toks == ref_ids
    assert np.abs(run["logits"] - ref["logits"]).max() < 1e-3
