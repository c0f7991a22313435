"""Geometry of the three sub-models.  Defaults are the released VITA-Mixtral-8x7B checkpoint as
stated in the reference's web_demo/vllm_tools/model_weight_file/config.json:16-116."""
from dataclasses import dataclass, field, asdict


@dataclass
class TextConfig:  # config.json:16-44 (text_config)
    hidden_size: int = 4096
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 14336
    num_local_experts: int = 8
    num_experts_per_tok: int = 2
    vocab_size: int = 51760
    rms_norm_eps: float = 1e-5
    rope_theta: float = 1e6
    bos_token_id: int = 1
    eos_token_id: int = 2


@dataclass
class VisionConfig:  # config.json:45-74 (vision_config) + downsample_ratio 0.5
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    patch_size: int = 14
    image_size: int = 448
    layer_norm_eps: float = 1e-6
    qkv_bias: bool = True

    @property
    def grid(self):
        return self.image_size // self.patch_size

    @property
    def num_tokens(self):
        return self.grid * self.grid + 1

    @property
    def out_tokens(self):
        return (self.grid // 2) ** 2

    @property
    def out_dim(self):
        return self.hidden_size * 4


@dataclass
class AudioConfig:  # config.json:75-109 (audio_config) + adapter (vllm_file/mixtral.py:821-859)
    input_dim: int = 80
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    layer_norm_eps: float = 1e-5
    adapter_kernel: int = 5
    adapter_norm_eps: float = 1e-3
    max_pe_len: int = 5000

    @property
    def sub_freq(self):  # frequency bins after the two stride-2 convs: ((80-1)//2-1)//2 = 19
        return ((self.input_dim - 1) // 2 - 1) // 2


@dataclass
class VitaConfig:
    text: TextConfig = field(default_factory=TextConfig)
    vision: VisionConfig = field(default_factory=VisionConfig)
    audio: AudioConfig = field(default_factory=AudioConfig)
    tokenizer_model_max_length: int = 4600  # config.json:116
    max_dynamic_patch: int = 12             # config.json:112
    image_aspect_ratio: str = None

    def to_dict(self):
        return asdict(self)

    @staticmethod
    def tiny():
        """Small geometry that still satisfies every kernel constraint (head_dim 128 / 64, K % 64);
        used by the parity tests and the golden fixtures."""
        return VitaConfig(
            text=TextConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                            intermediate_size=512, num_local_experts=4, vocab_size=1000),
            vision=VisionConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                image_size=56),
            audio=AudioConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256),
        )


def audio_token_count(n_frames: int) -> int:
    """len(ones(T)[2::2][2::2][0::2]) — whale/init_model.py:57-60."""
    t1 = len(range(2, n_frames, 2))
    t2 = len(range(2, t1, 2))
    return len(range(0, t2, 2))
