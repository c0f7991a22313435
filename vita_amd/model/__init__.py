from .builder import (build_audio_encoder, build_synthetic_model, build_vision_projector, build_vision_tower,
                      load_pretrained_model)
from .encoders import InternViTVisionTower, VisionProjector, WhaleAudioEncoder
from .vita_mixtral import VITAMixtralForCausalLM

VITAMixtralConfig = None  # HF AutoConfig registration is not needed: no from_pretrained() through HF
