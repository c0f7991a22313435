"""load_pretrained_model — drop-in for vita/model/builder.py:14-306 (same signature and return
tuple) that builds the HIP model instead of an HF/accelerate one.

Checkpoint format accepted: the reference's own — a directory with config.json,
tokenizer files and *.safetensors shards whose keys are `model.layers.N...`, `lm_head.weight`,
`model.vision_tower.vision_tower.*`, `model.mm_projector.{0,2}.*`, `model.audio_encoder.*`.
The reference's 2-GPU layer split (builder.py:177-218) is not reproduced: the 93.7 GB bf16 model
fits one 288 GB MI355X; multi-GPU is tensor parallel (see vita_amd.parallel)."""
import json
import os
import warnings

import torch

from ..config import AudioConfig, TextConfig, VisionConfig, VitaConfig
from .encoders import InternViTVisionTower, VisionProjector, WhaleAudioEncoder
from .vita_mixtral import VITAMixtralForCausalLM


def config_from_json(path):
    """Accepts either the HF-path config (flat Mixtral keys) or the vLLM-path config.json with
    text_config / vision_config / audio_config blocks (web_demo/vllm_tools/model_weight_file/config.json)."""
    with open(path) as f:
        j = json.load(f)
    t = j.get("text_config", j)
    tc = TextConfig(hidden_size=t["hidden_size"], num_hidden_layers=t["num_hidden_layers"],
                    num_attention_heads=t["num_attention_heads"], num_key_value_heads=t["num_key_value_heads"],
                    head_dim=t.get("head_dim") or t["hidden_size"] // t["num_attention_heads"],
                    intermediate_size=t["intermediate_size"], num_local_experts=t["num_local_experts"],
                    num_experts_per_tok=t.get("num_experts_per_tok", 2), vocab_size=t["vocab_size"],
                    rms_norm_eps=t.get("rms_norm_eps", 1e-5), rope_theta=t.get("rope_theta", 1e6),
                    bos_token_id=t.get("bos_token_id", 1), eos_token_id=t.get("eos_token_id", 2))
    vc, ac = VisionConfig(), AudioConfig()
    if "vision_config" in j:
        v = j["vision_config"]
        vc = VisionConfig(hidden_size=v["hidden_size"], num_hidden_layers=v["num_hidden_layers"],
                          num_attention_heads=v["num_attention_heads"], intermediate_size=v["intermediate_size"],
                          patch_size=v["patch_size"], image_size=v["image_size"],
                          layer_norm_eps=v.get("layer_norm_eps", 1e-6))
    if "audio_config" in j:
        a = j["audio_config"]
        ac = AudioConfig(input_dim=a.get("input_dim", 80), hidden_size=a["hidden_size"],
                         num_hidden_layers=a["num_hidden_layers"], num_attention_heads=a["num_attention_heads"],
                         intermediate_size=a["intermediate_size"], layer_norm_eps=a.get("layer_norm_eps", 1e-5))
    return VitaConfig(text=tc, vision=vc, audio=ac,
                      tokenizer_model_max_length=j.get("tokenizer_model_max_length", 4600),
                      max_dynamic_patch=j.get("max_dynamic_patch", 12), image_aspect_ratio=j.get("image_aspect_ratio"))


class _LazyShards(dict):
    """name -> tensor view over safetensors shards, loaded on first access (the packer touches each
    tensor once, so host memory stays at one tensor at a time)."""

    def __init__(self, model_path):
        super().__init__()
        from safetensors import safe_open
        self._files = {}
        for fn in sorted(os.listdir(model_path)):
            if fn.endswith(".safetensors"):
                f = safe_open(os.path.join(model_path, fn), framework="pt", device="cpu")
                for k in f.keys():
                    self._files[k] = f
        if not self._files:
            raise FileNotFoundError(f"no *.safetensors shards under {model_path}")

    def __contains__(self, k):
        return k in self._files

    def __iter__(self):
        return iter(self._files)

    def keys(self):
        return self._files.keys()

    def __getitem__(self, k):
        return self._files[k].get_tensor(k)


def load_pretrained_model(model_path, model_base, model_name, model_type, load_8bit=False, load_4bit=False,
                          device_map="auto", device="cuda", **kwargs):
    if model_type not in {"mixtral-8x7b"}:
        raise ValueError(f"Unknown Model Type {model_type}")          # builder.py:25-26
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes quantisation is not part of the HIP path (bf16 weights)")
    if model_base is not None or "lora" in (model_name or "").lower():
        warnings.warn("LoRA merge is a training-side feature and is not reproduced; loading merged weights only")
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=True)
    cfg = config_from_json(os.path.join(model_path, "config.json"))
    dev = "cuda:0" if device == "cuda" else device
    sd = _LazyShards(model_path)
    model = VITAMixtralForCausalLM(cfg, sd, device=dev, **kwargs)
    _apply_audio_side_files(model, os.path.join(model_path, "config.json"), model_path)
    model.resize_token_embeddings(len(tokenizer))
    tower = model.get_vision_tower()
    if not tower.is_loaded:
        tower.load_model()
    image_processor = tower.image_processor
    context_len = getattr(model.config, "max_sequence_length", 2048)     # builder.py:295-298
    if model.generation_config.pad_token_id is None:
        model.generation_config.pad_token_id = model.generation_config.eos_token_id
    return tokenizer, model, image_processor, context_len


def _apply_audio_side_files(model, config_json, model_path):
    """config.mm_audio_encoder names a directory with train.yaml + global_cmvn
    (multimodal_encoder/builder.py:44-59): take the fbank configuration from it, and its CMVN
    statistics when the checkpoint itself carries none."""
    from ..audio_config import read_audio_encoder_dir
    from ..audio_frontend import AudioEncoderProcessor
    with open(config_json) as f:
        j = json.load(f)
    d = j.get("mm_audio_encoder")
    if not d:
        return
    if not os.path.isabs(d):
        d = os.path.join(model_path, d)
    if not os.path.isfile(os.path.join(d, "train.yaml")):
        return
    side = read_audio_encoder_dir(d)
    enc = model.get_audio_encoder()
    enc.audio_processor = AudioEncoderProcessor(side["dataset_conf"])
    if side["input_dim"] != enc.acfg.input_dim:
        raise ValueError(f"train.yaml input_dim {side['input_dim']} != model {enc.acfg.input_dim}")
    if side["mean"] is not None and enc.w is not None and not getattr(enc, "cmvn_from_checkpoint", False):
        enc.w["mean"] = torch.from_numpy(side["mean"]).to(enc.device)
        enc.w["istd"] = torch.from_numpy(side["istd"]).to(enc.device)
    for msg in side["overridden"]:
        warnings.warn("audio encoder: " + msg)


def build_synthetic_model(cfg: VitaConfig = None, seed=0, device="cuda:0", rich=True, **kwargs):
    """Deterministic random-init model with the reference's parameter names (tests / smoke)."""
    from ..checkpoint import synth_state_dict
    cfg = cfg or VitaConfig.tiny()
    sd = synth_state_dict(cfg, seed=seed, rich=rich)
    model = VITAMixtralForCausalLM(cfg, sd, device=device, **kwargs)
    model.get_vision_tower().load_model()
    return model, sd


# sub-module factories with the reference's names (multimodal_encoder/builder.py:12,44;
# multimodal_projector/builder.py:154)
def build_vision_tower(vision_tower_cfg, **kwargs):
    name = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    if name is None or "internvit" not in name.lower():
        raise ValueError(f"Unknown vision tower: {name} (the released VITA checkpoint uses InternViT-300M; "
                         "CLIP/SigLIP/EVA towers are out of scope for the HIP path)")
    if getattr(vision_tower_cfg, "use_s2", False):
        raise ValueError("Currently not supporting S2 for InternViT")
    return InternViTVisionTower(name, args=vision_tower_cfg, **kwargs)


def build_audio_encoder(audio_encoder_config, **kwargs):
    return WhaleAudioEncoder(**kwargs)


def build_vision_projector(config, delay_load=False, **kwargs):
    ptype = getattr(config, "mm_projector_type", "mlp2x_gelu")
    if ptype != "mlp2x_gelu":
        raise ValueError(f"Unknown projector type: {ptype} (only mlp2x_gelu, the released checkpoint's, is built)")
    return VisionProjector()
