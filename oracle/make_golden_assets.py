"""ORACLE (test infrastructure): the reference's OWN assets as a second released-geometry request (SURVEY 8(d) configs 2 / 3
variant (i); video_audio_demo.py:180-226).  Writes tests/golden/assets_request.npz:

  tiles      uint8 [5, 448, 448, 3]  asset/vita_log2.png (2633 x 717 RGBA -> RGB) through the REFERENCE'S dynamic_preprocess
                                     (vita/util/data_utils_video_audio_neg_patch.py:1197-1255): 4 tiles + thumbnail
  pix_check  float32 [5, 3, 8, 8]    the corner 8 x 8 of every tile after HF's CLIPImageProcessor with the reference's
                                     preprocessor_config.json (what model.process_images feeds the tower) — pins the host path
  (the audio half of the request is tests/golden/q1_audio.npz:fbank, asset/q1.wav: 352 frames -> 44 tokens)

    python -m oracle.make_golden_assets
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402


def main():
    from PIL import Image
    assert rh.available(), "needs /root/reference"
    rh.install()
    from vita.util.data_utils_video_audio_neg_patch import dynamic_preprocess
    img = Image.open(os.path.join(rh.REF, "asset", "vita_log2.png")).convert("RGB")
    tiles, n = dynamic_preprocess(img, min_num=1, max_num=12, image_size=448, use_thumbnail=True)
    assert n == [5] and len(tiles) == 5, (n, len(tiles))
    arr = np.stack([np.asarray(t, np.uint8) for t in tiles])
    from transformers import CLIPImageProcessor
    with open(os.path.join(rh.REF, "web_demo", "vllm_tools", "model_weight_file", "preprocessor_config.json")) as f:
        pc = json.load(f)
    proc = CLIPImageProcessor(**{k: v for k, v in pc.items() if k not in ("image_processor_type", "processor_class")})
    pix = proc.preprocess(tiles, return_tensors="np")["pixel_values"].astype(np.float32)
    assert pix.shape == (5, 3, 448, 448)
    out = os.path.join(ROOT, "tests", "golden", "assets_request.npz")
    np.savez_compressed(out, tiles=arr, pix_check=np.ascontiguousarray(pix[:, :, :8, :8]), pix_mean=pix.mean(axis=(1, 2, 3)))
    print("written", out, os.path.getsize(out), "bytes; tile sums", arr.reshape(5, -1).sum(1).tolist())


if __name__ == "__main__":
    main()
