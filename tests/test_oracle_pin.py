"""CPU: pins the numpy oracle (oracle/) — (1) against the golden vectors produced by the
REFERENCE'S OWN modules in the build container (tests/golden/tiny_e2e.npz, oracle/make_golden.py),
(2) live against the installed HF Mixtral (the third-party code carrying the backbone arithmetic),
(3) live against the reference modules when /root/reference is present."""
import os

import numpy as np
import pytest

from oracle import encoders as oe
from oracle import mixtral as om
from tests.util import assert_close
from vita_amd.checkpoint import synth_state_dict
from vita_amd.config import VitaConfig

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tiny():
    cfg = VitaConfig.tiny()
    g = np.load(os.path.join(GOLD, "tiny_e2e.npz"))
    return cfg, synth_state_dict(cfg, seed=int(g["seed"])), g


def test_vit_and_projector_vs_reference_golden(tiny):
    cfg, sd, g = tiny
    vit = oe.internvit_tower(sd, cfg.vision, g["pix"])
    assert_close("oracle vit vs reference", vit, g["vit_out"], atol=5e-6)
    assert_close("oracle projector vs reference", oe.projector(sd, g["vit_out"]), g["proj_out"], atol=2e-6)


def test_whale_vs_reference_golden(tiny):
    cfg, sd, g = tiny
    out, mask = oe.whale_encoder(sd, cfg.audio, g["feats"])
    assert_close("oracle whale vs reference", out, g["audio_out"], atol=5e-6)
    assert mask.tolist() == g["audio_mask"].tolist()
    feats_pad = np.concatenate([g["feats"], np.zeros((37, 80), np.float32)])
    outp, maskp = oe.whale_encoder(sd, cfg.audio, feats_pad, length=123)
    assert maskp.tolist() == g["audio_pad_mask"].tolist()
    v = maskp
    assert_close("oracle whale (padded) vs reference, valid rows", outp[v], g["audio_pad_out"][v], atol=5e-6)


def test_splice_vs_reference_golden(tiny):
    cfg, sd, g = tiny
    emb = oe.splice(g["input_ids"], sd["model.embed_tokens.weight"], g["proj_out"], g["audio_out"][None],
                    cfg.tokenizer_model_max_length)
    assert_close("oracle splice vs reference", emb, g["inputs_embeds"], atol=2e-6)


def test_mixtral_vs_hf_golden(tiny):
    cfg, sd, g = tiny
    orc = om.MixtralOracle(sd, cfg.text)
    ids, lg = orc.greedy(g["inputs_embeds"], len(g["gen_ids"]))
    assert ids == g["gen_ids"].tolist()
    assert_close("oracle logits vs HF", lg, g["gen_logits"], atol=5e-6)
    orc.reset()
    _, hid = orc.forward(g["inputs_embeds"], want_hidden=True)
    assert_close("oracle hidden vs HF", hid[:-1], g["hidden_layers"], atol=5e-6)


def test_mixtral_live_vs_installed_hf():
    """different seed / shapes than the golden; transformers is part of the image on both boxes."""
    from oracle import hf_mixtral
    cfg = VitaConfig.tiny()
    sd = synth_state_dict(cfg, seed=7, parts=("text",))
    rng = np.random.default_rng(8)
    emb = sd["model.embed_tokens.weight"][rng.integers(3, cfg.text.vocab_size, size=19)]
    m = hf_mixtral.build(cfg.text, sd)
    ids_hf, lg_hf, _ = hf_mixtral.greedy(m, emb, 6)
    ids, lg = om.MixtralOracle(sd, cfg.text).greedy(emb, 6)
    assert ids == ids_hf
    assert_close("oracle vs installed HF", lg, lg_hf, atol=5e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/vita"), reason="reference tree only exists in the build container")
def test_live_vs_reference_modules():
    import torch
    from oracle import ref_harness as rh
    cfg = VitaConfig.tiny()
    sd = synth_state_dict(cfg, seed=11)
    rng = np.random.default_rng(12)
    pix = rng.standard_normal((1, 3, 56, 56)).astype(np.float32)
    feats = (rng.standard_normal((77, 80)) * 2 + 10).astype(np.float32)
    with torch.no_grad():
        ref_v = rh.build_internvit(cfg, sd)(torch.from_numpy(pix)).numpy()
        ref_a = rh.build_whale(cfg, sd)(torch.from_numpy(feats)[None], torch.tensor([77]))["inputs_embeds"][0].numpy()
    assert_close("live vit", oe.internvit_tower(sd, cfg.vision, pix), ref_v, atol=5e-6)
    assert_close("live whale", oe.whale_encoder(sd, cfg.audio, feats)[0], ref_a, atol=5e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/vita"), reason="reference tree only exists in the build container")
def test_live_whale_padded_batch_vs_reference():
    """two clips of different lengths in ONE reference forward (pad mask in attention, zeroing before the adapter,
    adpter mask) against the oracle run per clip with `length`."""
    import torch
    from oracle import ref_harness as rh
    cfg = VitaConfig.tiny()
    sd = synth_state_dict(cfg, seed=13)
    rng = np.random.default_rng(14)
    T, lens = 140, [140, 97]
    feats = (rng.standard_normal((2, T, 80)) * 3 + 12).astype(np.float32)
    with torch.no_grad():
        ref = rh.build_whale(cfg, sd)(torch.from_numpy(feats), torch.tensor(lens))
    for b in range(2):
        z, mask = oe.whale_encoder(sd, cfg.audio, feats[b], length=lens[b])
        rmask = ref["attention_mask"][b].numpy().astype(bool)
        assert mask.tolist() == rmask.tolist()
        # rows the reference marks valid must agree; padded rows are whatever the adapter makes of zeros in both
        assert_close(f"padded whale clip {b}", z[rmask], ref["inputs_embeds"][b].numpy()[rmask], atol=5e-6)


def test_torch_encoder_baseline_matches_the_checker():
    """bench.py's cpu_baseline times oracle/encoders_torch.py (torch CPU fp32: the operators the reference's modules run) for the
    encoder legs; it must compute what the fp64 checker (oracle/encoders.py, itself pinned to the reference's modules above) computes:
    tiny towers, two images, a 101-frame clip."""
    import torch
    from oracle import encoders as oe, encoders_torch as ot
    from vita_amd.checkpoint import synth_state_dict
    from vita_amd.config import VitaConfig
    cfg = VitaConfig.tiny()
    sd = synth_state_dict(cfg, seed=1, parts=("vision", "audio"))
    rng = np.random.default_rng(0)
    img = cfg.vision.patch_size * cfg.vision.grid
    pix = rng.standard_normal((2, 3, img, img)).astype(np.float32)
    fb = rng.standard_normal((101, 80)).astype(np.float32)
    with torch.no_grad():
        got_v = ot.projector(sd, ot.internvit_tower(sd, cfg.vision, pix)).numpy()
        got_a = ot.whale_encoder(sd, cfg.audio, fb).numpy()
    assert_close("torch ViT + projector vs fp64 checker", got_v, oe.projector(sd, oe.internvit_tower(sd, cfg.vision, pix)), atol=2e-6)
    assert_close("torch Whale + adapter vs fp64 checker", got_a, oe.whale_encoder(sd, cfg.audio, fb)[0], atol=5e-6)


def test_moe_forced_pair_is_the_routers_pair_when_they_agree():
    """oracle/mixtral.py moe(force=...) (the tie-branch hook of tests/test_video_shape_gpu.py): forcing the pair the router picks anyway
    changes nothing; forcing another pair changes that row only, with that pair's renormalised probabilities as weights."""
    from oracle import mixtral as om
    rng = np.random.default_rng(3)
    H, I, E, S = 32, 48, 8, 6
    lw = dict(gate=rng.standard_normal((E, H)).astype(np.float32), w1=rng.standard_normal((E, I, H)).astype(np.float32) * 0.1,
              w3=rng.standard_normal((E, I, H)).astype(np.float32) * 0.1, w2=rng.standard_normal((E, H, I)).astype(np.float32) * 0.1)
    x = rng.standard_normal((S, H)).astype(np.float32)
    y0, idx0, val0 = om.moe(x, lw)
    y1, idx1, val1 = om.moe(x, lw, force={2: tuple(int(v) for v in idx0[2])})
    assert np.array_equal(idx0, idx1) and np.allclose(val0, val1, atol=1e-7) and np.allclose(y0, y1, atol=1e-6)
    other = tuple(e for e in range(E) if e not in idx0[4])[:2]
    y2, idx2, val2 = om.moe(x, lw, force={4: other})
    assert tuple(idx2[4]) == other and abs(float(val2[4].sum()) - 1.0) < 1e-6
    keep = np.arange(S) != 4
    assert np.allclose(y2[keep], y0[keep], atol=1e-6) and not np.allclose(y2[4], y0[4], atol=1e-3)
    p = om.softmax((x[4:5] @ lw["gate"].T).astype(np.float32))[0]
    assert np.allclose(val2[4], p[list(other)] / p[list(other)].sum(), atol=1e-6)
