"""ORACLE (test infrastructure; never imported by the product path): an INDEPENDENT restatement of
`torchaudio.compliance.kaldi.fbank` (torchaudio==2.3.1, the dependency the reference calls at
vita/model/multimodal_encoder/whale/init_model.py:46-56; not vendored under /root/reference and not installed here).
Written from the published algorithm (kaldi.py: _get_waveform_and_window_properties, _get_window, get_mel_banks,
fbank) with scalar loops, so every number can be checked by hand; it shares no code with vita_amd/audio_frontend.py,
which is what it pins.  Parameters as the reference passes them: num_mel_bins 80, frame_length 25 ms, frame_shift
10 ms, dither 0 (hazard H2), energy_floor 0, defaults otherwise (povey window, preemphasis 0.97, remove_dc_offset,
round_to_power_of_two, snip_edges, low_freq 20, high_freq 0 = Nyquist, use_power, use_log_fbank).
"""
import math

import numpy as np

EPSILON = float(np.finfo(np.float32).eps)   # torchaudio: EPSILON = torch.tensor(torch.finfo(torch.float).eps)


def mel_scale_scalar(freq):
    return 1127.0 * math.log(1.0 + freq / 700.0)


def get_mel_banks(num_bins, window_length_padded, sample_freq, low_freq=20.0, high_freq=0.0):
    """kaldi.py get_mel_banks (no VTLN): -> [num_bins][num_fft_bins] list of lists.  num_fft_bins = padded / 2:
    the Nyquist bin is NOT covered (fbank() pads one zero column on the right)."""
    num_fft_bins = window_length_padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / window_length_padded          # <- sr / 512 (HF's numpy variant uses sr / 510)
    mel_low, mel_high = mel_scale_scalar(low_freq), mel_scale_scalar(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    banks = []
    for b in range(num_bins):
        left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
        row = []
        for i in range(num_fft_bins):
            mel = mel_scale_scalar(fft_bin_width * i)
            up = (mel - left) / (center - left)
            down = (right - mel) / (right - center)
            row.append(max(0.0, min(up, down)))
        banks.append(row)
    return banks


def fbank(waveform, sample_frequency=16000.0, num_mel_bins=80, frame_length=25.0, frame_shift=10.0,
          preemphasis_coefficient=0.97):
    """waveform: 1-D array already in the int16 range.  Returns float32 [m, num_mel_bins]."""
    x = [float(v) for v in np.asarray(waveform, dtype=np.float64).reshape(-1)]
    window_shift = int(sample_frequency * frame_shift * 0.001)
    window_size = int(sample_frequency * frame_length * 0.001)
    padded = 1
    while padded < window_size:
        padded *= 2
    if len(x) < window_size:
        return np.zeros((0, num_mel_bins), np.float32)
    m = 1 + (len(x) - window_size) // window_shift              # snip_edges
    povey = [(0.5 - 0.5 * math.cos(2.0 * math.pi * n / (window_size - 1))) ** 0.85 for n in range(window_size)]
    banks = np.asarray(get_mel_banks(num_mel_bins, padded, sample_frequency), np.float64)
    out = np.zeros((m, num_mel_bins), np.float64)
    for t in range(m):
        fr = x[t * window_shift: t * window_shift + window_size]
        mean = sum(fr) / window_size                            # remove_dc_offset
        fr = [v - mean for v in fr]
        pre = [fr[0] - preemphasis_coefficient * fr[0]] + [fr[n] - preemphasis_coefficient * fr[n - 1]
                                                            for n in range(1, window_size)]   # replicate pad
        win = np.zeros(padded, np.float64)
        win[:window_size] = [pre[n] * povey[n] for n in range(window_size)]
        spec = np.fft.rfft(win)                                 # padded/2 + 1 bins
        power = spec.real ** 2 + spec.imag ** 2
        mel = banks @ power[: padded // 2]                      # the padded Nyquist column has weight 0
        out[t] = np.log(np.maximum(mel, EPSILON))
    return out.astype(np.float32)
