"""The audio encoder's side files, as the reference's HF path reads them
(vita/model/multimodal_encoder/builder.py:44-59 -> whale/init_model.py:162-177):

  <mm_audio_encoder>/train.yaml   dataset_conf (resample / fbank incl. dither), is_json_cmvn,
                                  encoder_conf flags (transformer-dynamic-chunks, ...), input_dim
  <mm_audio_encoder>/global_cmvn  accumulated mean / variance statistics, JSON or Kaldi text
                                  (whale/cmvn.py:35-89)

Hazards: dither is forced to 0.0 and the random inference-time chunk mask of
`transformer-dynamic-chunks` is replaced by full attention — both are stated in the returned dict so a
caller can see what was overridden."""
import json
import math
import os

import numpy as np


def _finish(avg, var, count):
    """accumulated sums -> (mean, inverse std) — cmvn.py:43-50 / 77-83."""
    mean = np.asarray(avg, np.float64) / count
    v = np.asarray(var, np.float64) / count - mean * mean
    v = np.maximum(v, 1.0e-20)
    return mean, 1.0 / np.sqrt(v)


def load_cmvn_json(path):
    with open(path) as f:
        j = json.load(f)
    return _finish(j["mean_stat"], j["var_stat"], j["frame_num"])


def load_cmvn_kaldi(path):
    """Kaldi text matrix `[ sum_1 .. sum_D count  sq_1 .. sq_D 0 ]` (binary '\\0B' files are rejected,
    as the reference does)."""
    with open(path, "r", errors="replace") as f:
        if f.read(2) == "\0B":
            raise ValueError("kaldi cmvn binary file is not supported")
        f.seek(0)
        arr = f.read().split()
    if not (arr and arr[0] == "[" and arr[-2] == "0" and arr[-1] == "]"):
        raise ValueError(f"{path}: not a Kaldi text CMVN matrix")
    d = (len(arr) - 4) // 2
    avg = [float(x) for x in arr[1:d + 1]]
    count = float(arr[d + 1])
    var = [float(x) for x in arr[d + 2:2 * d + 2]]
    return _finish(avg, var, count)


def load_cmvn(path, is_json):
    mean, istd = load_cmvn_json(path) if is_json else load_cmvn_kaldi(path)
    return mean.astype(np.float32), istd.astype(np.float32)


def read_audio_encoder_dir(path):
    """-> dict(dataset_conf, mean, istd, input_dim, overridden=[...]) from <path>/train.yaml + global_cmvn."""
    import yaml
    with open(os.path.join(path, "train.yaml")) as f:
        conf = yaml.load(f, Loader=yaml.FullLoader)
    overridden = []
    ds = dict(conf.get("dataset_conf") or {})
    fb = dict(ds.get("fbank_conf") or {"num_mel_bins": 80, "frame_length": 25, "frame_shift": 10, "dither": 0.0})
    if float(fb.get("dither", 0.0)) != 0.0:
        overridden.append(f"fbank dither {fb['dither']} -> 0.0 (deterministic features)")
        fb["dither"] = 0.0
    ds["fbank_conf"] = fb
    ds.setdefault("resample_conf", {"resample_rate": 16000})
    enc = conf.get("encoder_conf") or {}
    if enc.get("transformer-dynamic-chunks", enc.get("transformer_dynamic_chunks", False)):
        overridden.append("transformer-dynamic-chunks: random inference-time chunk mask -> full attention")
    mean = istd = None
    cmvn_file = os.path.join(path, "global_cmvn")
    if os.path.exists(cmvn_file):
        mean, istd = load_cmvn(cmvn_file, bool(conf.get("is_json_cmvn", False)))
    return {"dataset_conf": ds, "mean": mean, "istd": istd, "input_dim": int(conf.get("input_dim", 80)),
            "overridden": overridden}
