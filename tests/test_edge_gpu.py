"""Edge cases of the path (the reference's own runtime asserts and limits):
padded audio clips and a batch of clips, the 12-tile + thumbnail maximum, truncation at
tokenizer_model_max_length, a one-token prompt, and KV-cache exhaustion reported as an error."""
import numpy as np
import pytest
import torch

from oracle import encoders as oe
from oracle import mixtral as om
from tests.util import assert_close, to_np
from vita_amd.checkpoint import synth_state_dict
from vita_amd.config import VitaConfig
from vita_amd.model import build_synthetic_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny(dev):
    model, sd = build_synthetic_model(VitaConfig.tiny(), seed=51, device=dev, max_new_tokens=16, max_prefill=4096)
    return model, sd


def test_audio_padded_clip_and_batch(tiny, dev):
    """speech_lengths shorter than T: padded frames are masked in attention and zeroed before the adapter
    (whale/module/encoder/encoder.py:140-147, adapter.py:115-116); two clips in one call."""
    model, sd = tiny
    cfg = VitaConfig.tiny()
    rng = np.random.default_rng(0)
    T = 140
    feats = (rng.standard_normal((2, T, 80)) * 3 + 12).astype(np.float32)
    lens = [T, 97]
    out = model.get_audio_encoder()(torch.from_numpy(feats).to(dev), torch.tensor(lens).to(dev))
    assert out["inputs_embeds"].shape[0] == 2 and out["attention_mask"].dtype == torch.bool
    for b in range(2):
        ref, mask = oe.whale_encoder(sd, cfg.audio, feats[b], length=lens[b])
        assert_close(f"clip {b} (len {lens[b]})", to_np(out["inputs_embeds"][b]), ref, atol=3e-5)
        assert to_np(out["attention_mask"][b]).tolist() == mask.tolist()
    assert int(out["attention_mask"][1].sum()) < int(out["attention_mask"][0].sum())


def test_max_tiles_through_tower(tiny, dev):
    """12 tiles + thumbnail = 13 tiles (max_dynamic_patch 12, config.json:112) in one tower call."""
    model, sd = tiny
    cfg = VitaConfig.tiny()
    pix = np.random.default_rng(1).standard_normal((13, 3, cfg.vision.image_size, cfg.vision.image_size)).astype(np.float32)
    got = model.encode_images(torch.from_numpy(pix).to(dev))
    ref = oe.projector(sd, oe.internvit_tower(sd, cfg.vision, pix))
    assert got.shape[0] == 13
    assert_close("13 tiles", to_np(got), ref, atol=3e-5)


def test_truncation_at_model_max_length(tiny, dev):
    """vita_arch.py:326-329: the spliced sequence is cut at tokenizer_model_max_length."""
    model, sd = tiny
    old = model.config.tokenizer_model_max_length
    model.config.tokenizer_model_max_length = 50
    try:
        ids = torch.randint(3, 900, (1, 80), device=dev)
        pix = torch.zeros((1, 3, 56, 56), device=dev)
        audios = {"audios": torch.zeros((1, 400, 80), device=dev), "lengths": torch.tensor([400], device=dev)}
        emb = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix, audios)[4]
        assert emb.shape[1] == 50
    finally:
        model.config.tokenizer_model_max_length = old


def test_single_token_prompt_and_cache_exhaustion(dev):
    from vita_amd import _lib
    from vita_amd.checkpoint import pack_mixtral
    from vita_amd.engine import MixtralEngine
    cfg = VitaConfig.tiny()
    sd = synth_state_dict(cfg, seed=52, parts=("text",))
    emb = sd["model.embed_tokens.weight"][[7]]
    ref_ids, ref_lg = om.MixtralOracle(sd, cfg.text).greedy(emb, 4)
    eng = MixtralEngine(cfg, pack_mixtral(sd, cfg, dev), dev, max_ctx=8, max_prefill=4, max_new=16, logit_rows=16)
    eng.prefill(torch.from_numpy(emb).to(dev))
    eng.decode(3)
    torch.cuda.synchronize()
    assert eng.generated() == ref_ids
    assert_close("S=1 logits", to_np(eng.logits_all[3]), ref_lg[3], atol=1e-3)
    with pytest.raises(_lib.VitaHipError):            # positions 1..7 exist: asking past the cache must fail loudly
        eng.decode(16)
    with pytest.raises(_lib.VitaHipError):            # prefill longer than max_prefill
        eng.prefill(torch.zeros((5, cfg.text.hidden_size), device=dev))
    # a rejected call leaves the engine consistent and usable: the steps that fitted were kept (positions 1..7), the
    # host mirror did not run ahead of the device counter, and a fresh prefill + decode reproduces the oracle
    torch.cuda.synchronize()
    c = eng.check_device_flag()
    assert c[0] == 7 and c[1] == 7, c                 # pos = max_ctx - 1; 1 prefill token + 3 + the 3 steps that fitted
    with pytest.raises(_lib.VitaHipError):
        eng.decode(1)                                 # still full: refused again, nothing enqueued
    eng.prefill(torch.from_numpy(emb).to(dev))
    eng.decode(3)
    torch.cuda.synchronize()
    assert eng.generated() == ref_ids
    eng.close()
