"""Writes a tiny checkpoint directory in the REFERENCE'S format: config.json,
tokenizer files, model.safetensors with HF-4.41 parameter names — the input of
load_pretrained_model (vita/model/builder.py:14-306).  Weights are the seeded synthetic state dict."""
import json
import os

import numpy as np
import torch

from vita_amd.checkpoint import synth_state_dict
from vita_amd.config import VitaConfig

WORDS = ("you are an ai robot and your name is vita the user asks question about image audio video please answer "
         "what is this picture describe sound hello world system user bot a of in to it that").split()


def write_tokenizer(path, vocab_size):
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for w in WORDS:
        vocab.setdefault(w, len(vocab))
    i = 0
    while len(vocab) < min(vocab_size, 400):
        vocab.setdefault(f"w{i}", len(vocab)); i += 1
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
    tok.add_special_tokens(["<unk>", "<s>", "</s>"])
    tok.save(os.path.join(path, "tokenizer.json"))
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "bos_token": "<s>", "eos_token": "</s>",
                   "unk_token": "<unk>", "model_max_length": 4600, "clean_up_tokenization_spaces": False}, f)
    with open(os.path.join(path, "special_tokens_map.json"), "w") as f:
        json.dump({"bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>"}, f)


def write(path, cfg: VitaConfig = None, seed=0, audio_side_files=False):
    from safetensors.torch import save_file
    cfg = cfg or VitaConfig.tiny()
    os.makedirs(path, exist_ok=True)
    t, v, a = cfg.text, cfg.vision, cfg.audio
    conf = {"model_type": "vita-mixtral", "architectures": ["VITAMixtralForCausalLM"],
            "hidden_size": t.hidden_size, "num_hidden_layers": t.num_hidden_layers,
            "num_attention_heads": t.num_attention_heads, "num_key_value_heads": t.num_key_value_heads,
            "head_dim": t.head_dim, "intermediate_size": t.intermediate_size,
            "num_local_experts": t.num_local_experts, "num_experts_per_tok": t.num_experts_per_tok,
            "vocab_size": t.vocab_size, "rms_norm_eps": t.rms_norm_eps, "rope_theta": t.rope_theta,
            "bos_token_id": 1, "eos_token_id": 2, "torch_dtype": "bfloat16",
            "mm_vision_tower": "InternViT-300M-448px", "mm_projector_type": "mlp2x_gelu",
            "mm_audio_encoder": "audio-encoder", "tokenizer_model_max_length": cfg.tokenizer_model_max_length,
            "vision_config": {"hidden_size": v.hidden_size, "num_hidden_layers": v.num_hidden_layers,
                              "num_attention_heads": v.num_attention_heads, "intermediate_size": v.intermediate_size,
                              "patch_size": v.patch_size, "image_size": v.image_size, "layer_norm_eps": v.layer_norm_eps},
            "audio_config": {"input_dim": a.input_dim, "hidden_size": a.hidden_size,
                             "num_hidden_layers": a.num_hidden_layers, "num_attention_heads": a.num_attention_heads,
                             "intermediate_size": a.intermediate_size, "layer_norm_eps": a.layer_norm_eps}}
    sd = synth_state_dict(cfg, seed=seed, rich=True)
    skip = set()
    if audio_side_files:
        # HF-path layout (multimodal_encoder/builder.py:44-59): CMVN statistics and the fbank configuration
        # live in <mm_audio_encoder>/{global_cmvn, train.yaml}, not in the safetensors
        conf["mm_audio_encoder"] = "audio-encoder"
        d = os.path.join(path, "audio-encoder")
        os.makedirs(d, exist_ok=True)
        mean = sd["model.audio_encoder.encoder.global_cmvn.mean"].astype(np.float64)
        istd = sd["model.audio_encoder.encoder.global_cmvn.istd"].astype(np.float64)
        n = 1.0e6
        s1, s2 = mean * n, (1.0 / (istd * istd) + mean * mean) * n
        with open(os.path.join(d, "global_cmvn"), "w") as f:
            f.write("[\n " + " ".join(repr(float(x)) for x in s1) + f" {n!r}\n " +
                    " ".join(repr(float(x)) for x in s2) + " 0 ]\n")
        with open(os.path.join(d, "train.yaml"), "w") as f:
            f.write("input_dim: 80\nis_json_cmvn: false\ndataset_conf:\n  resample_conf: {resample_rate: 16000}\n"
                    "  fbank_conf: {num_mel_bins: 80, frame_length: 25, frame_shift: 10, dither: 1.0}\n"
                    "encoder_conf:\n  transformer-dynamic-chunks: true\n")
        skip = {k for k in sd if "global_cmvn" in k}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(conf, f, indent=1)
    # two shards, as HF writes large checkpoints; bf16 is the released dtype (values are bf16-exact)
    keys = sorted(k for k in sd if k not in skip)
    half = len(keys) // 2
    from vita_amd.checkpoint import round_bf16

    def as_saved(x):  # bf16 where that is exact (all weights), fp32 otherwise (the vendored CMVN statistics)
        t = torch.from_numpy(np.ascontiguousarray(x))
        return t.to(torch.bfloat16) if np.array_equal(round_bf16(x), x) else t

    for i, ks in enumerate((keys[:half], keys[half:])):
        save_file({k: as_saved(sd[k]) for k in ks}, os.path.join(path, f"model-{i + 1:05d}-of-00002.safetensors"))
    write_tokenizer(path, t.vocab_size)
    return sd


def write_wav(path, seconds=1.3, sr=16000, seed=3):
    import wave
    x = (0.1 * np.random.default_rng(seed).standard_normal(int(seconds * sr))).clip(-1, 1)
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((x * 32767).astype("<i2").tobytes())
