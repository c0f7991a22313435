"""ORACLE (test infrastructure): generates tests/golden/*.npz by running the REFERENCE'S OWN code
in this container (imported from /root/reference with oracle/ref_harness.py shims) plus the
installed HF Mixtral for the backbone, on seeded synthetic weights at the tiny geometry.
The GPU box has no /root/reference, so these committed vectors are how the reference itself
travels to the `-m gpu` tests.

    python -m oracle.make_golden          # rewrites tests/golden/
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import hf_mixtral, ref_harness as rh  # noqa: E402
from vita_amd.checkpoint import synth_state_dict  # noqa: E402
from vita_amd.config import VitaConfig  # noqa: E402

SEED = 0


def tiny_inputs(cfg):
    rng = np.random.default_rng(1234)
    pix = rng.standard_normal((2, 3, cfg.vision.image_size, cfg.vision.image_size)).astype(np.float32)
    feats = (rng.standard_normal((123, 80)) * 3 + 12).astype(np.float32)
    V = cfg.text.vocab_size
    txt = lambda n: rng.integers(3, V, size=n).tolist()
    ids = [1] + txt(9) + [-200, -200] + txt(5) + [-500] + txt(6)
    return pix, feats, np.asarray(ids, np.int64)


class FakeTokenizer:
    """Deterministic whitespace/char tokenizer with a BOS, enough to pin the placeholder logic."""
    bos_token_id = 1

    def __call__(self, text):
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=[1] + [3 + (ord(c) % 90) for c in text])


def main():
    import torch
    assert rh.available(), "needs /root/reference"
    os.makedirs(GOLD, exist_ok=True)
    cfg = VitaConfig.tiny()
    sd = synth_state_dict(cfg, seed=SEED)
    pix, feats, ids = tiny_inputs(cfg)
    rh.install()

    # ---- towers + projector + splice from the reference's own modules -----------------------
    tower, pj, aud = rh.build_internvit(cfg, sd), rh.build_projector(cfg, sd), rh.build_whale(cfg, sd)
    with torch.no_grad():
        vit = tower(torch.from_numpy(pix))
        proj = pj(vit)
        ao = aud(torch.from_numpy(feats)[None], torch.tensor([feats.shape[0]]))
    embeds = rh.reference_inputs_embeds(cfg, sd, ids, pix, feats, feats.shape[0])
    # padded-utterance variant: 123 valid frames inside 160 (exercises pad masks)
    feats_pad = np.concatenate([feats, np.zeros((37, 80), np.float32)])
    with torch.no_grad():
        ao_pad = aud(torch.from_numpy(feats_pad)[None], torch.tensor([123]))

    # ---- backbone: HF Mixtral greedy from the spliced embeddings ---------------------------------
    m = hf_mixtral.build(cfg.text, sd)
    gen_ids, gen_logits, hidden = hf_mixtral.greedy(m, embeds, 12)

    np.savez_compressed(
        os.path.join(GOLD, "tiny_e2e.npz"), seed=SEED, pix=pix, feats=feats, input_ids=ids,
        vit_out=vit.numpy(), proj_out=proj.numpy(), audio_out=ao["inputs_embeds"][0].numpy(),
        audio_mask=ao["attention_mask"][0].numpy(), audio_pad_out=ao_pad["inputs_embeds"][0].numpy(),
        audio_pad_mask=ao_pad["attention_mask"][0].numpy(), inputs_embeds=embeds, gen_ids=np.asarray(gen_ids),
        gen_logits=gen_logits, hidden_layers=hidden[:-1])

    # ---- host-side logic: prompt strings, placeholder tokenisation, tiling -------------------------
    from vita.conversation import conv_templates
    from vita.util.mm_utils import tokenizer_image_audio_token, tokenizer_image_token
    from vita.util.data_utils_video_audio_neg_patch import dynamic_preprocess
    from PIL import Image
    tok = FakeTokenizer()
    host = {}
    cases = {"image": "<image><image>\ndescribe<audio>", "video": "<image>" * 4 + "\n<audio>", "lang": "hello there"}
    for mod, q in cases.items():
        c = conv_templates["mixtral_two"].copy()
        c.append_message(c.roles[0], q)
        c.append_message(c.roles[1], None)
        p = c.get_prompt(mod)
        host[f"prompt_{mod}"] = np.frombuffer(p.encode("utf-8"), dtype=np.uint8)
        host[f"ids_ia_{mod}"] = np.asarray(tokenizer_image_audio_token(p, tok), np.int64)
        host[f"ids_i_{mod}"] = np.asarray(tokenizer_image_token(p, tok), np.int64)
    sizes = [(448, 448), (2633, 717), (717, 2633), (1000, 600), (300, 900), (1920, 1080), (640, 480), (100, 100),
             (1344, 896), (3000, 500)]
    grids = []
    for (w, h) in sizes:
        tiles, n = dynamic_preprocess(Image.new("RGB", (w, h)), min_num=1, max_num=12, image_size=448, use_thumbnail=True)
        grids.append(n[0])
    host["tile_sizes"] = np.asarray(sizes)
    host["tile_counts"] = np.asarray(grids)
    # one real crop comparison: pixel content of the tiles of the reference's asset image
    img = Image.open(os.path.join(rh.REF, "asset", "vita_log2.png")).convert("RGB")
    tiles, n = dynamic_preprocess(img, min_num=1, max_num=12, image_size=448, use_thumbnail=True)
    host["logo_n"] = np.asarray(n)
    host["logo_tile_sums"] = np.asarray([np.asarray(t, np.int64).sum() for t in tiles])
    host["logo_small"] = np.asarray(img.resize((64, 18)))  # the input travels as a small thumbnail only for docs
    np.savez_compressed(os.path.join(GOLD, "host_logic.npz"), **host)

    # ---- audio front end (A4) ---------------------------------------------------------------------------
    # fbank              : asset/q1.wav through oracle/kaldi_fbank.py — an independent scalar restatement of
    #                      torchaudio.compliance.kaldi.fbank (the HF path's dependency, init_model.py:46-56)
    # whale_numpy_*      : the SAME clip through the reference's OWN vLLM-flavour extractor, loaded from
    #                      web_demo/vllm_tools/model_weight_file/processor_whale.py (its numpy fallback: torchaudio
    #                      is not installed here), dither 0: raw log-mel and the CMVN-normalised input_features
    import importlib.util
    import json
    from oracle import kaldi_fbank as okf
    from vita_amd.audio_frontend import load_wav
    w, sr = load_wav(os.path.join(rh.REF, "asset", "q1.wav"))
    pcm16 = np.round(w * 32768).astype(np.int16)
    fb = okf.fbank(pcm16.astype(np.float64), float(sr))
    mdir = os.path.join(rh.REF, "web_demo", "vllm_tools", "model_weight_file")
    spec = importlib.util.spec_from_file_location("ref_processor_whale", os.path.join(mdir, "processor_whale.py"))
    pw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pw)
    with open(os.path.join(mdir, "feature_extractor", "preprocessor_config.json")) as f:
        pc = json.load(f)
    fe = pw.WhaleFeatureExtractor(**{**pc, "dither": 0.0})
    wf = (pcm16.astype(np.float32) / 32768.0)
    raw = fe._extract_fbank_features(wf[None])
    feats = fe(wf, sampling_rate=int(sr), return_tensors="np")["input_features"][0]
    np.savez_compressed(os.path.join(GOLD, "q1_audio.npz"), pcm16=pcm16, sr=sr,
                        fbank=np.ascontiguousarray(fb, dtype=np.float32),
                        whale_numpy_fbank=np.ascontiguousarray(raw, dtype=np.float32),
                        whale_numpy_input_features=np.ascontiguousarray(feats, dtype=np.float32))
    # ---- vLLM-flavour placeholder expansion (f#1): the reference's OWN repeat_and_pad_image_tokens and
    # get_audio_feature_size, cut out of web_demo/vllm_tools/vllm_file/mixtral.py (the module itself needs vllm),
    # with the tile counts of the reference's dynamic_preprocess --------------------------------------------
    fx = rh.extract_functions(os.path.join(rh.REF, "web_demo", "vllm_tools", "vllm_file", "mixtral.py"),
                              ["repeat_and_pad_image_tokens", "get_audio_feature_size"])
    IMG, AUD = 51000, 51001
    ex_ids = [1, 5, 6, IMG, 7, AUD, 8, IMG, 9, AUD, 10]
    ex_sizes = [(448, 448), (1000, 500)]
    ex_frames = [352, 998]
    ex_tiles = [len(dynamic_preprocess(Image.new("RGB", wh), min_num=1, max_num=12, image_size=448, use_thumbnail=True)[0])
                for wh in ex_sizes]
    _, ids1 = fx["repeat_and_pad_image_tokens"](None, None, list(ex_ids), image_token_id=IMG,
                                                repeat_count=[256 * n for n in ex_tiles])
    aud_sizes = [fx["get_audio_feature_size"](torch.zeros(n, 80)) for n in ex_frames]
    _, ids2 = fx["repeat_and_pad_image_tokens"](None, None, ids1, image_token_id=AUD, repeat_count=list(aud_sizes))
    np.savez_compressed(os.path.join(GOLD, "vllm_expand.npz"), ids=np.asarray(ex_ids), sizes=np.asarray(ex_sizes),
                        frames=np.asarray(ex_frames), tiles=np.asarray(ex_tiles), audio_sizes=np.asarray(aud_sizes),
                        new_token_ids=np.asarray(ids2))
    print("golden written:", sorted(os.listdir(GOLD)))
    print("gen ids:", gen_ids)


if __name__ == "__main__":
    main()
