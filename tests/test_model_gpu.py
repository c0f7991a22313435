"""GPU parity of the drop-in model classes (vita_amd.model) through the C ABI:
 * against the golden vectors recorded from the REFERENCE'S OWN modules (tests/golden/tiny_e2e.npz),
 * against the numpy oracle on other seeds and at the released model's layer geometry.
Bars: encoder features and logits within 1e-3 (fp32), greedy ids bit-exact (BASELINE.json)."""
import os

import numpy as np
import pytest
import torch

from oracle import encoders as oe
from oracle import mixtral as om
from tests.util import assert_close, to_np
from vita_amd.checkpoint import synth_state_dict
from vita_amd.config import AudioConfig, TextConfig, VisionConfig, VitaConfig

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tiny_model(dev):
    from vita_amd.model import build_synthetic_model
    g = np.load(os.path.join(GOLD, "tiny_e2e.npz"))
    model, sd = build_synthetic_model(VitaConfig.tiny(), seed=int(g["seed"]), device=dev, max_new_tokens=32,
                                      max_prefill=512)
    return model, sd, g


def test_vision_tower_and_projector_vs_reference_golden(tiny_model, dev):
    model, sd, g = tiny_model
    tower = model.get_vision_tower()
    assert tower.is_loaded and tower.hidden_size == 512 and tower.num_patches == 16
    vit = tower(torch.from_numpy(g["pix"]).to(dev))
    assert_close("vit tower vs reference", to_np(vit), g["vit_out"], atol=2e-5)
    feats = model.encode_images(torch.from_numpy(g["pix"]).to(dev))
    assert_close("encode_images vs reference", to_np(feats), g["proj_out"], atol=2e-5)
    # list-of-images form (internvit_encoder.py:57-66)
    vit2 = tower([torch.from_numpy(g["pix"][i]).to(dev) for i in range(2)])
    assert torch.equal(vit, vit2)
    with pytest.raises(ValueError):
        tower(torch.zeros(3, 56, 56, device=dev))


def test_audio_encoder_vs_reference_golden(tiny_model, dev):
    model, sd, g = tiny_model
    enc = model.get_audio_encoder()
    out = enc(torch.from_numpy(g["feats"]).to(dev)[None], torch.tensor([123.0]).half().to(dev))
    assert_close("whale vs reference", to_np(out["inputs_embeds"][0]), g["audio_out"], atol=2e-5)
    assert out["attention_mask"].dtype == torch.bool and to_np(out["attention_mask"][0]).tolist() == g["audio_mask"].tolist()
    feats_pad = np.concatenate([g["feats"], np.zeros((37, 80), np.float32)])
    outp = enc(torch.from_numpy(feats_pad).to(dev)[None], torch.tensor([123]).to(dev))
    m = g["audio_pad_mask"].astype(bool)
    assert to_np(outp["attention_mask"][0]).astype(bool).tolist() == m.tolist()
    assert_close("whale padded vs reference (valid rows)", to_np(outp["inputs_embeds"][0])[m], g["audio_pad_out"][m],
                 atol=2e-5)


def test_audio_chunk_mask_vs_oracle(tiny_model, dev):
    """the deterministic (chunk, left) window mode of hazard H1."""
    model, sd, g = tiny_model
    enc = model.get_audio_encoder()
    enc.set_chunk_mask(5, 1)
    try:
        out = enc(torch.from_numpy(g["feats"]).to(dev)[None], torch.tensor([123]).to(dev))
    finally:
        enc.set_chunk_mask(0, -1)
    ref, _ = oe.whale_encoder(sd, VitaConfig.tiny().audio, g["feats"], chunk=5, left=1)
    assert_close("whale chunk mask vs oracle", to_np(out["inputs_embeds"][0]), ref, atol=2e-5)


def test_splice_vs_reference_golden(tiny_model, dev):
    model, sd, g = tiny_model
    ids = torch.from_numpy(g["input_ids"])[None].to(dev)
    audios = {"audios": torch.from_numpy(g["feats"])[None].to(dev), "lengths": torch.tensor([123]).to(dev)}
    out = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, torch.from_numpy(g["pix"]).to(dev), audios)
    assert out[0] is None and out[4].shape[0] == 1
    assert_close("inputs_embeds vs reference", to_np(out[4][0]), g["inputs_embeds"], atol=2e-5)
    with pytest.raises(AssertionError):  # placeholder count != number of tiles (vita_arch.py:227-231)
        model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None,
                                                   torch.from_numpy(g["pix"][:1]).to(dev), audios)


def test_generate_end_to_end_vs_reference_golden(tiny_model, dev):
    """image tiles + audio + text -> greedy ids: the whole path of video_audio_demo.py:257-276."""
    model, sd, g = tiny_model
    ids = torch.from_numpy(g["input_ids"])[None].to(dev)
    audios = {"audios": torch.from_numpy(g["feats"])[None].half().to(dev),  # the demo feeds fp16 features
              "lengths": torch.tensor([123.0]).half().to(dev)}
    audios32 = {"audios": torch.from_numpy(g["feats"])[None].to(dev), "lengths": torch.tensor([123]).to(dev)}
    n = len(g["gen_ids"])
    out = model.generate(ids, images=torch.from_numpy(g["pix"]).to(dev), audios=audios32, do_sample=False,
                         temperature=0.01, top_p=None, num_beams=1, output_scores=True, return_dict_in_generate=True,
                         max_new_tokens=n, use_cache=True, eos_token_id=-1)
    seq = out.sequences
    assert torch.equal(seq[:, :ids.shape[1]], ids)                    # video_audio_demo.py:273-276
    got = seq[0, ids.shape[1]:].tolist()
    print("reference ids:", g["gen_ids"].tolist())
    print("device ids   :", got)
    for i in range(n):
        assert_close(f"scores step {i} vs HF", to_np(out.scores[i][0]), g["gen_logits"][i], atol=1e-3)
    assert got == g["gen_ids"].tolist()
    # lookahead must not change the result
    model.lookahead = 1
    out1 = model.generate(ids, images=torch.from_numpy(g["pix"]).to(dev), audios=audios32, max_new_tokens=n,
                          eos_token_id=-1)
    model.lookahead = 8
    assert torch.equal(out1, seq)
    # fp16 audio features (what the demo passes) still run; values differ at fp16 input precision
    out16 = model.generate(ids, images=torch.from_numpy(g["pix"]).to(dev), audios=audios, max_new_tokens=4,
                           eos_token_id=-1)
    assert out16.shape[1] == ids.shape[1] + 4


def test_generate_stops_on_eos_and_criteria(tiny_model, dev):
    model, sd, g = tiny_model
    ids = torch.from_numpy(g["input_ids"])[None].to(dev)
    audios = {"audios": torch.from_numpy(g["feats"])[None].to(dev), "lengths": torch.tensor([123]).to(dev)}
    gold = g["gen_ids"].tolist()
    out = model.generate(ids, images=torch.from_numpy(g["pix"]).to(dev), audios=audios, max_new_tokens=12,
                         eos_token_id=gold[3])
    assert out[0, ids.shape[1]:].tolist() == gold[:4]

    class StopAt:
        def __call__(self, output_ids, scores, **kw):
            return output_ids[0, -1].item() == gold[6]
    out = model.generate(ids, images=torch.from_numpy(g["pix"]).to(dev), audios=audios, max_new_tokens=12,
                         eos_token_id=-1, stopping_criteria=[StopAt()])
    assert out[0, ids.shape[1]:].tolist() == gold[:7]
    with pytest.raises(NotImplementedError):
        model.generate(ids, images=torch.from_numpy(g["pix"]).to(dev), audios=audios, do_sample=True)


def test_text_only_prompt_runs_dummy_encoders(tiny_model, dev):
    """text-only still runs both encoders on dummy inputs and splices zero-length slices
    (video_audio_demo.py:188-195,227-231; vita_arch.py:240-251)."""
    model, sd, g = tiny_model
    cfg = VitaConfig.tiny()
    ids = torch.tensor([[1, 5, 9, 77, 300]], device=dev)
    img = torch.zeros((1, 3, 56, 56), device=dev)
    audios = {"audios": torch.zeros((1, 400, 80), device=dev).half(), "lengths": torch.tensor([400.0]).half().to(dev)}
    out = model.generate(ids, images=img, audios=audios, max_new_tokens=5, eos_token_id=-1)
    emb = sd["model.embed_tokens.weight"][[1, 5, 9, 77, 300]]
    ref_ids, _ = om.MixtralOracle(sd, cfg.text).greedy(emb, 5)
    assert out[0, 5:].tolist() == ref_ids


# ---- released-model layer geometry (2 layers each; oracle in fp64 numpy) -----------------------------
@pytest.mark.parametrize("per_operator", [False, True])
def test_vit_real_geometry_two_layers(dev, per_operator):
    """per_operator: one C call per operator (vh_gemm[_ln] / vh_attention) instead of one per block (vh_encoder_layer)."""
    from vita_amd.model.encoders import InternViTVisionTower
    cfg = VitaConfig()
    cfg.vision = VisionConfig(num_hidden_layers=2)
    sd = synth_state_dict(cfg, seed=21, parts=("vision",))
    tower = InternViTVisionTower("InternViT-300M-448px", vcfg=cfg.vision)
    tower.per_operator = per_operator
    tower.set_state_dict(sd, dev)
    rng = np.random.default_rng(22)
    pix = rng.standard_normal((1, 3, 448, 448)).astype(np.float32)
    out, layers = tower(torch.from_numpy(pix).to(dev), want_layers=True)
    ref, rl = oe.internvit_tower(sd, cfg.vision, pix, want_layers=True)
    for i in range(2):
        assert_close(f"vit real layer {i}", to_np(layers[i]), rl[i], atol=1e-3, rtol=1e-4)
    assert tuple(out.shape) == (1, 256, 4096)
    assert_close("vit real tower out", to_np(out), ref, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("per_operator", [False, True])
def test_whale_real_geometry_two_layers(dev, per_operator):
    from vita_amd.model.encoders import WhaleAudioEncoder
    cfg = VitaConfig()
    cfg.audio = AudioConfig(num_hidden_layers=2)
    sd = synth_state_dict(cfg, seed=23, parts=("audio",))
    enc = WhaleAudioEncoder(sd, acfg=cfg.audio, device=dev, llm_dim=4096)
    enc.per_operator = per_operator
    g = np.load(os.path.join(GOLD, "q1_audio.npz"))
    feats = g["fbank"]                                        # real fbank of asset/q1.wav: 352 frames -> 44 tokens
    out, mask, layers = enc.encode_one(torch.from_numpy(feats).to(dev), want_layers=True)
    ref, rmask, rl = oe.whale_encoder(sd, cfg.audio, feats, want_layers=True)
    assert tuple(out.shape) == (44, 4096) and mask.all()
    for i in range(2):
        assert_close(f"whale real layer {i}", to_np(layers[i]), rl[i], atol=1e-3, rtol=1e-4)
    assert_close("whale real out", to_np(out), ref, atol=1e-3, rtol=1e-4)
