"""The synthetic request of BASELINE configs[2]: one 448x448 tile, a seeded 10 s waveform, ~140 stand-in
system-prompt ids + 32 text ids.  Shared by bench.py and the released-geometry parity test, so the request the bench
times is the request the oracle checks."""
import numpy as np

from ..audio_frontend import kaldi_fbank
from .constants import AUDIO_TOKEN_INDEX, IMAGE_TOKEN_INDEX


def make_request(cfg, frames=1, text_tokens=32, seconds=10.0):
    """Returns dict(pixel_values float32 [frames,3,S,S] (ImageNet-normalised uniform noise), fbank float32 [T,80],
    input_ids list[int] with one IMAGE sentinel per frame and one AUDIO sentinel)."""
    size = cfg.vision.image_size
    rng = np.random.default_rng(2)
    mean = np.asarray([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.asarray([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    pix = ((rng.random((frames, 3, size, size), dtype=np.float32) - mean) / std).astype(np.float32)
    wav = 0.1 * np.random.default_rng(3).standard_normal(int(16000 * seconds))
    fbank = kaldi_fbank(wav * (1 << 15), 16000)                      # [998, 80] for 10 s
    r1 = np.random.default_rng(1)
    sys_ids = r1.integers(3, 51000, size=139).tolist()               # stand-in for the ~140-token system prompt
    txt_ids = r1.integers(3, 51000, size=text_tokens).tolist()
    ids = [cfg.text.bos_token_id] + sys_ids + [IMAGE_TOKEN_INDEX] * frames + txt_ids + [AUDIO_TOKEN_INDEX]
    return {"pixel_values": pix, "fbank": fbank, "input_ids": ids}
