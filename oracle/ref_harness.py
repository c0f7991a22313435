"""ORACLE (test infrastructure): drives the REFERENCE'S OWN modules on the CPU in fp32.
Runs only where /root/reference exists (this build container), never on the GPU box.
The reference cannot be imported as-is here (torchvision / timm / torchaudio / xformers are
absent — SURVEY §8(c)); minimal in-memory stubs for those imports are installed first.  Nothing
from the reference is copied: its modules are imported from where they lie and called."""
import importlib.machinery
import os
import sys
import types

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "vita"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_done = False


def install():
    global _done
    if _done:
        return
    import torch
    import transformers  # noqa: F401  must be imported before the stubs exist
    from transformers import CLIPImageProcessor  # noqa: F401  resolve lazily-loaded pieces now

    class _Any:  # permissive placeholder for unused symbols
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, n):
            return _Any()

    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tr = _stub("torchvision.transforms", InterpolationMode=_Any(), Compose=_Any, Resize=_Any, ToTensor=_Any,
                   Normalize=_Any, CenterCrop=_Any, RandomResizedCrop=_Any, RandomHorizontalFlip=_Any, Lambda=_Any)
        _stub("torchvision.transforms.functional", InterpolationMode=_Any())
        tv.transforms = tr
        _stub("torchvision.models")
        _stub("torchvision.models.mobilenetv3", InvertedResidual=_Any, InvertedResidualConfig=_Any)
        _stub("torchvision.ops")
        _stub("torchvision.ops.misc", SqueezeExcitation=_Any)
    if "timm" not in sys.modules:
        _stub("timm")
        _stub("timm.layers", drop_path=lambda x, *a, **k: x, to_2tuple=lambda x: (x, x),
              trunc_normal_=torch.nn.init.trunc_normal_, DropPath=torch.nn.Identity)
        _stub("timm.layers.norm_act", LayerNormAct2d=_Any)
        _stub("timm.models")
        _stub("timm.models.layers", DropPath=torch.nn.Identity, drop_path=lambda x, *a, **k: x,
              to_2tuple=lambda x: (x, x), trunc_normal_=torch.nn.init.trunc_normal_)
        _stub("timm.models.registry", register_model=lambda f: f)
    if "xformers" not in sys.modules:
        _stub("xformers")
        _stub("xformers.ops")
    if "torchaudio" not in sys.modules:
        ta = _stub("torchaudio")
        _stub("torchaudio.compliance")
        _stub("torchaudio.compliance.kaldi")
        ta.transforms = _stub("torchaudio.transforms")
    if "decord" not in sys.modules:
        _stub("decord", VideoReader=_Any, cpu=_Any)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _done = True


def whale_config(a, llm_dim):
    """The yaml dict layout build_audio_encoder expects (multimodal_encoder/builder.py:44-59,
    whale/module/encoder/encoder.py:55-120, component argparse keys)."""
    C = a.hidden_size
    return {
        "cmvn_file": None, "is_json_cmvn": True, "input_dim": a.input_dim,
        "encoder_conf": {
            "overview_conf": {"encoder-input-dim": a.input_dim, "encoder-layer-config": "subsampling-transformer",
                              "encoder-output-dim": C},
            "para_conf": {
                "subsampling": {"subsampling-rate": 4, "subsampling-input-dim": a.input_dim,
                                "subsampling-output-dim": C, "subsampling-dropout-rate": 0.1},
                "transformer": {"transformer-input-dim": C, "transformer-output-dim": C,
                                "transformer-attention-dim": C, "transformer-attention-heads": a.num_attention_heads,
                                "transformer-linear-units": a.intermediate_size,
                                "transformer-num-blocks": a.num_hidden_layers, "transformer-dropout-rate": 0.1,
                                "transformer-attention-dropout-rate": 0.0,
                                "transformer-positional-dropout-rate": 0.1, "transformer-input-layer": "linear",
                                "transformer-pos-enc-class": "rel-enc", "transformer-normalize-before": True,
                                "transformer-concat-after": False, "transformer-positionwise-layer-type": "linear",
                                "transformer-chunk_size": -1, "transformer-left_chunks": -1,
                                "transformer-dynamic-chunks": False},
            },
        },
        "model_conf": {"llm_path": "", "enc_out_dim": C, "llm_embed_dim": llm_dim, "kernel_size": a.adapter_kernel,
                       "adpter_type": "subsampling", "activation_func": "gelu", "norm": "layer",
                       "freeze_encoder": True, "freeze_adpter": True},
        "dataset_conf": {"resample_conf": {"resample_rate": 16000},
                         "fbank_conf": {"num_mel_bins": 80, "frame_length": 25, "frame_shift": 10, "dither": 0.0}},
    }


def build_whale(cfg, sd):
    install()
    import torch
    from vita.model.multimodal_encoder.whale.init_model import init_model
    from vita.model.multimodal_encoder.whale.cmvn import GlobalCMVN
    argv, sys.argv = sys.argv, sys.argv[:1]  # whaleEncoder re-parses sys.argv (encoder.py:59-63)
    try:
        m = init_model(whale_config(cfg.audio, cfg.text.hidden_size))
    finally:
        sys.argv = argv
    P = "model.audio_encoder."
    m.encoder.global_cmvn = GlobalCMVN(torch.from_numpy(sd[P + "encoder.global_cmvn.mean"]),
                                       torch.from_numpy(sd[P + "encoder.global_cmvn.istd"]))
    state = {k[len(P):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(P)}
    missing, unexpected = m.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    assert all("global_cmvn" in k for k in missing), missing
    return m.float().eval()


def build_internvit(cfg, sd):
    install()
    import torch
    from vita.model.multimodal_encoder.internvit.configuration_intern_vit import InternVisionConfig
    from vita.model.multimodal_encoder.internvit.internvit_encoder import InternViTVisionTower
    from vita.model.multimodal_encoder.internvit.modeling_intern_vit import InternVisionModel
    v = cfg.vision
    ic = InternVisionConfig(hidden_size=v.hidden_size, num_hidden_layers=v.num_hidden_layers,
                            num_attention_heads=v.num_attention_heads, intermediate_size=v.intermediate_size,
                            patch_size=v.patch_size, image_size=v.image_size, layer_norm_eps=v.layer_norm_eps,
                            qkv_bias=True, qk_normalization=False, norm_type="layer_norm", use_flash_attn=False,
                            drop_path_rate=0.0, hidden_act="gelu", initializer_factor=1.0)
    model = InternVisionModel(ic)
    P = "model.vision_tower.vision_tower."
    state = {k[len(P):]: torch.from_numpy(x) for k, x in sd.items() if k.startswith(P)}
    model.load_state_dict(state, strict=True)
    tower = InternViTVisionTower.__new__(InternViTVisionTower)  # skip from_pretrained (no hub offline)
    torch.nn.Module.__init__(tower)
    tower.is_loaded, tower.select_layer, tower.scale_pix_shuffle = True, -1, 0.5
    tower.vision_tower = model.float().eval()
    return tower


def build_projector(cfg, sd):
    install()
    import torch
    from types import SimpleNamespace
    from vita.model.multimodal_projector.builder import build_vision_projector
    pj = build_vision_projector(SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=cfg.vision.out_dim,
                                                hidden_size=cfg.text.hidden_size))
    P = "model.mm_projector."
    pj.load_state_dict({k[len(P):]: torch.from_numpy(x) for k, x in sd.items() if k.startswith(P)}, strict=True)
    return pj.float().eval()


class _SpliceHost:
    """Smallest object prepare_inputs_labels_for_multimodal needs as `self` (vita_arch.py:151-407)."""

    def __init__(self, cfg, embed, tower, projector, audio):
        import torch
        from types import SimpleNamespace
        self.config = SimpleNamespace(tokenizer_model_max_length=cfg.tokenizer_model_max_length,
                                      tokenizer_padding_side="right")
        self.device = torch.device("cpu")
        emb = torch.nn.Embedding.from_pretrained(torch.from_numpy(embed), freeze=True)
        self._model = SimpleNamespace(embed_tokens=emb, get_vision_tower=lambda: tower, mm_projector=projector,
                                      get_audio_encoder=lambda: audio)

    def get_model(self):
        return self._model

    def get_vision_tower(self):
        return self._model.get_vision_tower()

    def get_audio_encoder(self):
        return self._model.get_audio_encoder()


def reference_inputs_embeds(cfg, sd, input_ids, images, audios, lengths):
    """Runs the reference's encode + splice path and returns inputs_embeds [S, H] (numpy)."""
    install()
    import torch
    from vita.model.vita_arch import VITAMetaForCausalLM
    tower, pj, aud = build_internvit(cfg, sd), build_projector(cfg, sd), build_whale(cfg, sd)
    host = _SpliceHost(cfg, sd["model.embed_tokens.weight"], tower, pj, aud)
    host.encode_images = lambda im: VITAMetaForCausalLM.encode_images(host, im)
    with torch.no_grad():
        out = VITAMetaForCausalLM.prepare_inputs_labels_for_multimodal(
            host, torch.as_tensor(input_ids)[None], None, None, None, None, torch.from_numpy(images),
            {"audios": torch.from_numpy(audios)[None], "lengths": torch.as_tensor([lengths])})
    return out[4][0].numpy()


def extract_functions(path, names):
    """Run selected `def`s of a reference file whose module cannot be imported here (its other imports are absent:
    vllm) — the FunctionDef nodes are cut out of the file's AST, nested ones included, and executed unchanged."""
    import ast
    import typing
    src = open(path).read()
    tree = ast.parse(src)
    ns = {k: getattr(typing, k) for k in ("Optional", "List", "Tuple", "Union", "TypeVar")}
    ns["_T"] = typing.TypeVar("_T", str, int)
    ns["PreTrainedTokenizerBase"] = object
    import torch
    ns["torch"] = torch
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
            found[node.name] = ns[node.name]
    missing = set(names) - set(found)
    if missing:
        raise KeyError(f"{path}: no function(s) {sorted(missing)}")
    return found
