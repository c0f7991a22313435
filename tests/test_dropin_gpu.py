"""Drop-in check of the `vita.model` boundary (SURVEY §8(b) flavour 1): the call sequence of the
reference's video_audio_demo.py:155-283, written against the reference's OWN import names
(`from vita.model.builder import load_pretrained_model`, ... resolved by compat/), on a tiny
checkpoint directory in the reference's format (config.json + tokenizer + safetensors shards with
HF-4.41 names).  Token ids and scores are compared with the ORACLE (oracle/encoders.py + oracle/mixtral.py, pinned to
the reference's own modules by tests/test_oracle_pin.py) run on the same state dict and the same input tensors."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from tests import tiny_ckpt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("vita_tiny_ckpt"))
    sd = tiny_ckpt.write(d, seed=21)
    wav = os.path.join(d, "q.wav")
    tiny_ckpt.write_wav(wav)
    img = os.path.join(d, "img.png")
    rng = np.random.default_rng(9)
    Image.fromarray(rng.integers(0, 255, size=(90, 150, 3), dtype=np.uint8)).save(img)
    return d, sd, wav, img


def _demo(model_path, image_path, audio_path, question, max_new_tokens=12):
    """video_audio_demo.py:155-283 with its own names; only argparse / printing removed."""
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        from vita.constants import DEFAULT_AUDIO_TOKEN, DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX
        from vita.conversation import SeparatorStyle, conv_templates
        from vita.model.builder import load_pretrained_model
        from vita.util.data_utils_video_audio_neg_patch import dynamic_preprocess
        from vita.util.mm_utils import (KeywordsStoppingCriteria, get_model_name_from_path,
                                        tokenizer_image_audio_token, tokenizer_image_token)
        from vita.util.utils import disable_torch_init
    finally:
        sys.path.pop(0)
    disable_torch_init()
    model_name = get_model_name_from_path(model_path)
    tokenizer, model, image_processor, context_len = load_pretrained_model(model_path, None, model_name, "mixtral-8x7b")
    model.resize_token_embeddings(len(tokenizer))
    vision_tower = model.get_vision_tower()
    if not vision_tower.is_loaded:
        vision_tower.load_model()
    image_processor = vision_tower.image_processor
    audio_encoder = model.get_audio_encoder()
    audio_encoder.to(dtype=torch.float16)
    audio_processor = audio_encoder.audio_processor
    model.eval()
    qs = question
    if audio_path is not None:
        audio, audio_for_llm_lens = audio_processor.process(os.path.join(audio_path))
        audio_length = audio.shape[0]
    else:
        audio = torch.zeros(400, 80)
        audio_length = audio.shape[0]
    audio = torch.unsqueeze(audio, dim=0)
    audio_length = torch.unsqueeze(torch.tensor(audio_length), dim=0)
    audios = {"audios": audio.half().cuda(), "lengths": audio_length.half().cuda()}
    if image_path is not None:
        image = Image.open(image_path).convert("RGB")
        image, p_num = dynamic_preprocess(image, min_num=1, max_num=12, image_size=image_processor.crop_size["height"],
                                          use_thumbnail=True)
        assert len(p_num) == 1
        image_tensor = model.process_images(image, model.config).to(dtype=model.dtype, device="cuda")
        qs = DEFAULT_IMAGE_TOKEN * p_num[0] + "\n" + qs + (DEFAULT_AUDIO_TOKEN if audio_path else "")
        modality = "image"
    else:
        size = image_processor.crop_size["height"]
        image_tensor = torch.zeros((1, 3, size, size)).to(dtype=model.dtype, device="cuda")
        if audio_path:
            qs = qs + DEFAULT_AUDIO_TOKEN
        modality = "lang"
    conv = conv_templates["mixtral_two"].copy()
    conv.append_message(conv.roles[0], qs)
    conv.append_message(conv.roles[1], None)
    prompt = conv.get_prompt(modality)
    if audio_path:
        input_ids = tokenizer_image_audio_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0).cuda()
    else:
        input_ids = tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0).cuda()
    stop_str = conv.sep if conv.sep_style != SeparatorStyle.TWO else conv.sep2
    stopping_criteria = KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)
    with torch.inference_mode():
        output_ids = model.generate(input_ids, images=image_tensor, audios=audios, do_sample=False, temperature=0.01,
                                    top_p=None, num_beams=1, output_scores=True, return_dict_in_generate=True,
                                    max_new_tokens=max_new_tokens, use_cache=True,
                                    stopping_criteria=[stopping_criteria])
    seqs = output_ids.sequences
    n_in = input_ids.shape[1]
    assert (input_ids != seqs[:, :n_in]).sum().item() == 0            # the demo's own self-check (:272-276)
    text = tokenizer.batch_decode(seqs[:, n_in:], skip_special_tokens=False)[0].strip()
    return model, input_ids, image_tensor, audios, seqs[0, n_in:].tolist(), output_ids.scores, text


def _expected(sd, cfg, input_ids, image_tensor, audios, n):
    """the oracle on the tensors the demo sequence handed to generate(): (ids, fp32 logits per step)."""
    from tests.oracle_e2e import oracle_generate
    ids, logits, _ = oracle_generate(sd, cfg, input_ids[0], pix=image_tensor, fbank=audios["audios"][0],
                                     fbank_len=int(float(audios["lengths"][0])), n_new=n)
    return ids, logits


def _score_err(scores, exp_logits, n):
    return max(float(np.abs(scores[i][0].float().cpu().numpy() - exp_logits[i]).max()) for i in range(n))


def test_demo_sequence_image_audio(ckpt, dev):
    d, sd, wav, img = ckpt
    from vita_amd.config import VitaConfig
    model, ids, pix, audios, got, scores, text = _demo(d, img, wav, "describe this picture")
    assert (ids == -200).sum().item() == pix.shape[0] >= 1 and (ids == -500).sum().item() == 1
    exp, exp_scores = _expected(sd, VitaConfig.tiny(), ids, pix, audios, 12)
    n = len(got)
    assert n >= 1 and got == exp[:n], (got, exp)
    assert n == 12 or got[-1] == 2                                      # stopped on </s> or ran to the cap
    assert _score_err(scores, exp_scores, n) < 1e-3                     # logits within 1e-3 of the fp32 oracle (BASELINE.json)
    assert isinstance(text, str)


def test_hf_path_audio_side_files(tmp_path, dev):
    """HF-path checkpoint layout: CMVN + fbank configuration come from <mm_audio_encoder>/global_cmvn and
    train.yaml (dither and the random chunk mask are overridden, with warnings)."""
    from vita_amd.config import VitaConfig
    d = str(tmp_path)
    sd = tiny_ckpt.write(d, seed=22, audio_side_files=True)
    wav = os.path.join(d, "q.wav")
    tiny_ckpt.write_wav(wav, seconds=0.9, seed=5)
    with pytest.warns(UserWarning, match="audio encoder"):
        model, ids, pix, audios, got, scores, _ = _demo(d, None, wav, "what is this sound", max_new_tokens=6)
    assert model.get_audio_encoder().audio_processor.dataset_conf["fbank_conf"]["dither"] == 0.0
    exp, exp_scores = _expected(sd, VitaConfig.tiny(), ids, pix, audios, 6)
    assert got == exp[:len(got)], (got, exp)
    assert _score_err(scores, exp_scores, len(got)) < 1e-3   # CMVN statistics went through the Kaldi text file (float64 -> float32)


def test_demo_sequence_text_only(ckpt, dev):
    """text-only prompt: the demo still feeds a zero image and a 400-frame zero clip (video_audio_demo.py:188-195,227-231)."""
    d, sd, _, _ = ckpt
    from vita_amd.config import VitaConfig
    model, ids, pix, audios, got, scores, _ = _demo(d, None, None, "hello what is your name", max_new_tokens=6)
    assert (ids < 0).sum().item() == 0 and 1 <= len(got) <= 6
    exp, exp_scores = _expected(sd, VitaConfig.tiny(), ids, pix, audios, 6)   # no sentinels: embeddings only
    assert got == exp[:len(got)], (got, exp)
    assert len(got) == 6 or got[-1] == 2
    assert _score_err(scores, exp_scores, len(got)) < 1e-3


def test_demo_sequence_video(ckpt, dev, tmp_path):
    """video prompt (video_audio_demo.py:199-212): frames sampled at 1 fps, clamped to [4, 16], one <image> per frame."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        from decord import VideoReader, cpu
        from vita.constants import DEFAULT_AUDIO_TOKEN, DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX
        from vita.conversation import conv_templates
        from vita.model.builder import load_pretrained_model
        from vita.util.mm_utils import tokenizer_image_audio_token
    finally:
        sys.path.pop(0)
    from vita_amd.host.video import get_rawvideo
    d, sd, wav, _ = ckpt
    path = str(tmp_path / "clip.npz")
    np.savez(path, frames=np.random.default_rng(2).integers(0, 255, size=(66, 40, 60, 3), dtype=np.uint8), fps=6.0)
    tokenizer, model, image_processor, _ = load_pretrained_model(d, None, "vita-tiny", "mixtral-8x7b")
    frames, slice_len = get_rawvideo(VideoReader(path, ctx=cpu(0)), image_processor,
                                     image_aspect_ratio=getattr(model.config, "image_aspect_ratio", None))
    assert slice_len == 11 and frames.shape[0] == 11                 # 66 frames at 6 fps -> one per second
    audio, _ = model.get_audio_encoder().audio_processor.process(wav)
    audios = {"audios": audio[None].half().cuda(), "lengths": torch.tensor([audio.shape[0]]).half().cuda()}
    qs = DEFAULT_IMAGE_TOKEN * slice_len + "\n" + "what happens in this video" + DEFAULT_AUDIO_TOKEN
    conv = conv_templates["mixtral_two"].copy()
    conv.append_message(conv.roles[0], qs)
    conv.append_message(conv.roles[1], None)
    input_ids = tokenizer_image_audio_token(conv.get_prompt("video"), tokenizer, IMAGE_TOKEN_INDEX,
                                            return_tensors="pt").unsqueeze(0).cuda()
    assert (input_ids == IMAGE_TOKEN_INDEX).sum().item() == slice_len
    out = model.generate(input_ids, images=frames.to(dtype=model.dtype, device="cuda"), audios=audios, do_sample=False,
                         num_beams=1, return_dict_in_generate=True, output_scores=True, max_new_tokens=6, use_cache=True,
                         eos_token_id=-1)
    assert (input_ids != out.sequences[:, :input_ids.shape[1]]).sum().item() == 0
    from vita_amd.config import VitaConfig
    exp, exp_scores = _expected(sd, VitaConfig.tiny(), input_ids, frames.to(dtype=model.dtype), audios, 6)
    assert out.sequences[0, input_ids.shape[1]:].tolist() == exp, (out.sequences[0, input_ids.shape[1]:].tolist(), exp)
    assert _score_err(out.scores, exp_scores, 6) < 1e-3                # 11 frames x 4 tokens + audio + text vs the oracle
    assert model.last_timing["prompt_tokens"] == input_ids.shape[1] - slice_len - 1 + slice_len * 4 + \
        model.get_audio_encoder()(audios["audios"], audios["lengths"])["inputs_embeds"].shape[1]


def test_load_pretrained_rejects_unknown_type(ckpt):
    from vita_amd.model import load_pretrained_model
    with pytest.raises(ValueError):
        load_pretrained_model(ckpt[0], None, "x", "qwen2")             # builder.py:25-26
