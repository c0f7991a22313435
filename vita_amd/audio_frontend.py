"""Host-side speech front end: wav -> Kaldi-compatible log-mel filterbank features.

Mirrors the reference's audioEncoderProcessor.process (vita/model/multimodal_encoder/whale/
init_model.py:35-60), which calls torchaudio.compliance.kaldi.fbank on the CPU with
num_mel_bins=80, frame_length=25 ms, frame_shift=10 ms, energy_floor=0 and the yaml's dither.
torchaudio is not in this image, so the Kaldi recipe is written out in numpy: snip-edges framing,
DC removal, 0.97 pre-emphasis, Povey window, 512-point power spectrum, 80 triangular filters on
the Kaldi mel scale (20 Hz .. Nyquist), log with the fp32-epsilon floor.  It stays on the CPU like
the reference's (it is ~1 ms of work and not on the GPU hot path).
"""
import math
import wave

import numpy as np

from .config import audio_token_count

EPS = 1.1920928955078125e-07  # torch.finfo(float32).eps, Kaldi's log floor


def _mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def kaldi_mel_banks(num_bins=80, n_fft=512, sample_rate=16000, low=20.0, high=0.0, variant="kaldi"):
    """Triangular filters on the Kaldi mel scale, triangularised in mel space.
    variant "kaldi": torchaudio.compliance.kaldi.get_mel_banks (FFT bin width = sample_rate / n_fft) — what the
      reference's HF path computes (whale/init_model.py:46-56 -> kaldi.fbank).
    variant "hf_numpy": the filter bank of the reference's vLLM-flavour extractor when torchaudio is absent
      (processor_whale.py:127-140: transformers.audio_utils.mel_filter_bank(num_frequency_bins=256, ...,
      triangularize_in_mel_space=True)), whose FFT bin width is sample_rate / (2 * 255): the filters differ from
      Kaldi's by up to 0.117 and the log-mel features of asset/q1.wav by up to 2.95 (tests/test_host_logic.py)."""
    nyq = 0.5 * sample_rate
    if high <= 0.0:
        high += nyq
    if variant == "hf_numpy":
        fft_bin_width = sample_rate / ((n_fft // 2 - 1) * 2)
    elif variant == "kaldi":
        fft_bin_width = sample_rate / n_fft
    else:
        raise ValueError(f"unknown mel filter bank variant {variant!r}")
    mel_lo, mel_hi = _mel(low), _mel(high)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1) * delta, mel_lo + (b + 2) * delta
    mel = _mel(fft_bin_width * np.arange(n_fft // 2, dtype=np.float64))[None, :]
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    return np.maximum(0.0, np.minimum(up, down))  # [num_bins, n_fft/2]; the Nyquist bin gets weight 0


def povey_window(n):
    return (0.5 - 0.5 * np.cos(2.0 * math.pi * np.arange(n, dtype=np.float64) / (n - 1))) ** 0.85


def kaldi_fbank(waveform, sample_rate=16000, num_mel_bins=80, frame_length_ms=25.0, frame_shift_ms=10.0,
                dither=0.0, preemphasis=0.97, rng=None, mel_variant="kaldi"):
    """waveform: 1-D float array already scaled to the int16 range (the reference multiplies by
    1<<15, init_model.py:46).  Returns float32 [num_frames, num_mel_bins]."""
    x = np.asarray(waveform, dtype=np.float64).reshape(-1)
    win = int(sample_rate * frame_length_ms * 0.001)
    hop = int(sample_rate * frame_shift_ms * 0.001)
    n_fft = 1 << (win - 1).bit_length()
    if x.shape[0] < win:
        return np.zeros((0, num_mel_bins), np.float32)
    n_frames = 1 + (x.shape[0] - win) // hop
    idx = np.arange(win)[None, :] + hop * np.arange(n_frames)[:, None]
    fr = x[idx]
    if dither != 0.0:
        rng = rng or np.random.default_rng()
        fr = fr + dither * rng.standard_normal(fr.shape)
    fr = fr - fr.mean(axis=1, keepdims=True)
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)  # replicate-pad the first sample
    fr = fr - preemphasis * prev
    fr = fr * povey_window(win)[None, :]
    spec = np.fft.rfft(fr, n=n_fft, axis=1)
    power = (spec.real ** 2 + spec.imag ** 2)[:, : n_fft // 2]
    mel = power @ kaldi_mel_banks(num_mel_bins, n_fft, sample_rate, variant=mel_variant).T
    return np.log(np.maximum(mel, EPS)).astype(np.float32)


def load_wav(path):
    """PCM wav -> (float waveform in [-1,1), sample_rate), first channel."""
    with wave.open(path, "rb") as w:
        sr, nch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        d = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    elif sw == 4:
        d = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    elif sw == 1:
        d = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width {sw}")
    return d.reshape(-1, nch)[:, 0], sr


class AudioEncoderProcessor:
    """Same surface as the reference's audioEncoderProcessor: process(wav_path) -> (Tensor[T,80], n_tokens)."""

    def __init__(self, dataset_conf=None):
        self.dataset_conf = dataset_conf or {
            "resample_conf": {"resample_rate": 16000},
            "fbank_conf": {"num_mel_bins": 80, "frame_length": 25, "frame_shift": 10, "dither": 0.0}}

    def process(self, wav_path):
        import torch
        wavef, sr = load_wav(wav_path)
        target = self.dataset_conf["resample_conf"]["resample_rate"]
        if sr != target:
            from scipy.signal import resample_poly
            g = math.gcd(int(sr), int(target))
            wavef = resample_poly(wavef, target // g, sr // g)
        fb = self.dataset_conf["fbank_conf"]
        # NB the reference passes the ORIGINAL file rate as sample_frequency (init_model.py:55)
        mat = kaldi_fbank(wavef * (1 << 15), sample_rate=sr, num_mel_bins=fb["num_mel_bins"],
                          frame_length_ms=fb["frame_length"], frame_shift_ms=fb["frame_shift"], dither=fb["dither"])
        return torch.from_numpy(mat), audio_token_count(mat.shape[0])


class WhaleFeatureExtractor:
    """The vLLM-flavour extractor (web_demo/vllm_tools/model_weight_file/processor_whale.py): fbank followed by
    CMVN with the PRELOADED means / inverse stds (utterance_cmvn with cmvn_means/cmvn_istds, :211-233), so the
    audio tower receives normalised features.  Call: extractor(waveform [1, n] or [n] in [-1, 1),
    sampling_rate=16000, return_tensors="pt") -> {"input_features": [1, T, 80], "attention_mask": [1, T]}.
    dither is pinned to 0 (the shipped preprocessor_config.json says 1.0 = random noise per call).
    mel_variant: "kaldi" (default) = torchaudio's filter bank, i.e. what the reference extractor computes when
    torchaudio is installed (processor_whale.py:179-191) and what the audio tower was trained on;
    "hf_numpy" = bit-compatible with its numpy fallback (processor_whale.py:127-140,192-206), see kaldi_mel_banks."""

    def __init__(self, sampling_rate=16000, num_mel_bins=80, frame_length=25, frame_shift=10, cmvn_means=None,
                 cmvn_istds=None, mel_variant="kaldi", **_):
        from .checkpoint import vendored_cmvn
        self.sampling_rate, self.num_mel_bins = sampling_rate, num_mel_bins
        self.frame_length, self.frame_shift = frame_length, frame_shift
        self.mel_variant = mel_variant
        if cmvn_means is None or cmvn_istds is None:
            cmvn_means, cmvn_istds = vendored_cmvn(num_mel_bins)
        self.cmvn_means = np.asarray(cmvn_means, np.float32)
        self.cmvn_istds = np.asarray(cmvn_istds, np.float32)

    @classmethod
    def from_pretrained(cls, model_path, subfolder="feature_extractor", **_):
        import json
        import os
        with open(os.path.join(model_path, subfolder, "preprocessor_config.json")) as f:
            j = json.load(f)
        return cls(sampling_rate=j.get("sampling_rate", 16000), num_mel_bins=j.get("num_mel_bins", 80),
                   frame_length=j.get("frame_length", 25), frame_shift=j.get("frame_shift", 10),
                   cmvn_means=j.get("cmvn_means"), cmvn_istds=j.get("cmvn_istds"))

    def __call__(self, raw_speech, sampling_rate=None, return_tensors=None, **_):
        import torch
        x = raw_speech.detach().cpu().numpy() if hasattr(raw_speech, "detach") else np.asarray(raw_speech)
        x = np.squeeze(x).astype(np.float64)
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            from scipy.signal import resample_poly
            g = math.gcd(int(sampling_rate), int(self.sampling_rate))
            x = resample_poly(x, self.sampling_rate // g, sampling_rate // g)
        feats = kaldi_fbank(x * (1 << 15), self.sampling_rate, self.num_mel_bins, self.frame_length, self.frame_shift,
                            dither=0.0, mel_variant=self.mel_variant)
        feats = ((feats - self.cmvn_means[None]) * self.cmvn_istds[None]).astype(np.float32)
        mask = np.ones((1, feats.shape[0]), np.int32)
        out = {"input_features": feats[None], "attention_mask": mask}
        if return_tensors == "pt":
            out = {k: torch.from_numpy(v) for k, v in out.items()}
        return out
