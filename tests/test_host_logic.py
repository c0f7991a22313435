"""CPU: host-side logic against golden values taken from the reference's own functions
(tests/golden/host_logic.npz, q1_audio.npz), the C-ABI surface, and config / checkpoint helpers."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


class FakeTokenizer:  # same toy tokenizer oracle/make_golden.py used
    bos_token_id = 1

    def __call__(self, text):
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=[1] + [3 + (ord(c) % 90) for c in text])

    def batch_decode(self, ids, skip_special_tokens=True):
        return ["".join(chr(int(i)) for i in row if int(i) > 2) for row in ids]


CASES = {"image": "<image><image>\ndescribe<audio>", "video": "<image>" * 4 + "\n<audio>", "lang": "hello there"}


def test_prompt_and_placeholder_tokenisation_match_reference():
    from vita_amd.host.prompt import conv_templates, tokenizer_image_audio_token, tokenizer_image_token
    g = np.load(os.path.join(GOLD, "host_logic.npz"))
    tok = FakeTokenizer()
    for mod, q in CASES.items():
        c = conv_templates["mixtral_two"].copy()
        c.append_message(c.roles[0], q)
        c.append_message(c.roles[1], None)
        p = c.get_prompt(mod)
        assert p == bytes(g[f"prompt_{mod}"]).decode("utf-8"), mod
        assert tokenizer_image_audio_token(p, tok) == g[f"ids_ia_{mod}"].tolist()
        assert tokenizer_image_token(p, tok) == g[f"ids_i_{mod}"].tolist()
    # the template object must not be mutated by get_prompt on a copy
    assert isinstance(conv_templates["mixtral_two"].system, list)
    with pytest.raises(AssertionError):
        c = conv_templates["mixtral_two"].copy()
        c.append_message("user", "text only")
        c.get_prompt("image")


def test_dynamic_preprocess_matches_reference():
    from PIL import Image
    from vita_amd.host.image_processing import dynamic_preprocess
    g = np.load(os.path.join(GOLD, "host_logic.npz"))
    for (w, h), n in zip(g["tile_sizes"].tolist(), g["tile_counts"].tolist()):
        tiles, cnt = dynamic_preprocess(Image.new("RGB", (w, h)), min_num=1, max_num=12, image_size=448,
                                        use_thumbnail=True)
        assert cnt == [n] and len(tiles) == n, (w, h)
        assert all(t.size == (448, 448) for t in tiles)


def test_stopping_criteria_and_names():
    import torch
    from vita_amd.host.prompt import KeywordsStoppingCriteria, get_model_name_from_path
    tok = FakeTokenizer()
    prompt = torch.tensor([[1, 50, 51]])
    crit = KeywordsStoppingCriteria(["</s>"], tok, prompt)
    kw = tok("</s>").input_ids[1:]
    assert not crit(torch.tensor([[1, 50, 51, 60]]), None)
    assert crit(torch.tensor([[1, 50, 51, 60] + kw]), None)
    assert get_model_name_from_path("/a/b/VITA_ckpt/") == "VITA_ckpt"
    assert get_model_name_from_path("/a/run/checkpoint-100") == "run_checkpoint-100"


def test_fbank_matches_independent_oracle_and_token_count():
    """A4 (HF path): the product's vectorised Kaldi fbank against the fixture written by oracle/kaldi_fbank.py — an
    independent scalar restatement of torchaudio.compliance.kaldi.fbank (the dependency init_model.py:46-56 calls)."""
    from vita_amd.audio_frontend import kaldi_fbank
    from vita_amd.config import audio_token_count
    g = np.load(os.path.join(GOLD, "q1_audio.npz"))
    fb = kaldi_fbank(g["pcm16"].astype(np.float64), int(g["sr"]))
    assert fb.shape == (352, 80)                       # q1.wav = 3.54 s -> 352 frames
    assert np.abs(fb - g["fbank"]).max() < 1e-5
    assert audio_token_count(352) == 44 and audio_token_count(998) == 124 and audio_token_count(400) == 50


def test_fbank_matches_third_party_kaldi_implementation():
    """A4 pinned to code the builder did not write: HF transformers' Kaldi-compatible front end
    (transformers.audio_utils: povey window, remove_dc_offset, 0.97 pre-emphasis, kaldi mel scale triangularised in mel
    space, 257 frequency bins = FFT bin width sr/512 — the parameters of torchaudio.compliance.kaldi.fbank as
    whale/init_model.py:46-56 calls it: num_mel_bins 80, 25/10 ms, energy_floor 0, dither 0) on asset/q1.wav scaled to the
    int16 range.  Both the product and the committed fixture must agree with it to fp32 rounding."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    from vita_amd.audio_frontend import kaldi_fbank
    g = np.load(os.path.join(GOLD, "q1_audio.npz"))
    wav = g["pcm16"].astype(np.float64)                # = waveform * (1 << 15), init_model.py:46
    mel = mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000,
                          sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    hf = spectrogram(wav, window_function(400, "povey", periodic=False), frame_length=400, hop_length=160, fft_length=512,
                     power=2.0, center=False, preemphasis=0.97, mel_filters=mel, log_mel="log",
                     mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    assert hf.shape == (352, 80)
    assert np.abs(hf - g["fbank"]).max() < 1e-5         # measured 1.9e-6
    assert np.abs(hf - kaldi_fbank(wav, int(g["sr"]))).max() < 1e-5
    # a shorter, offset excerpt as a second input (framing / edge handling, not only the one file)
    part = wav[777:777 + 400 + 160 * 57]
    hf2 = spectrogram(part, window_function(400, "povey", periodic=False), frame_length=400, hop_length=160, fft_length=512,
                      power=2.0, center=False, preemphasis=0.97, mel_filters=mel, log_mel="log",
                      mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    fb2 = kaldi_fbank(part, 16000)
    assert fb2.shape == hf2.shape == (58, 80) and np.abs(hf2 - fb2).max() < 1e-5


def test_oracle_fbank_live_equals_fixture():
    """the committed fixture is what oracle/kaldi_fbank.py computes (first 40 frames re-run here)."""
    from oracle import kaldi_fbank as okf
    g = np.load(os.path.join(GOLD, "q1_audio.npz"))
    n = 400 + 160 * 39
    fb = okf.fbank(g["pcm16"][:n].astype(np.float64), float(g["sr"]))
    assert fb.shape == (40, 80) and np.array_equal(fb, g["fbank"][:40])


def test_kaldi_mel_banks_hand_checked():
    """torchaudio get_mel_banks by hand: mel(f) = 1127 ln(1 + f/700), 80 triangles between mel(20) and mel(8000),
    FFT bin width 16000/512 = 31.25 Hz.  A few weights worked out from the closed form, the partition-of-unity of
    neighbouring triangles, and the zero weight of bins outside [20 Hz, Nyquist)."""
    from vita_amd.audio_frontend import kaldi_mel_banks
    from oracle import kaldi_fbank as okf
    B = kaldi_mel_banks()
    assert B.shape == (80, 256)
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    lo, hi = mel(20.0), mel(8000.0)
    d = (hi - lo) / 81.0
    # bin 0: left = mel(20), centre = left + d; FFT bin 1 (31.25 Hz) sits on its rising edge
    assert abs(B[0, 1] - (mel(31.25) - lo) / d) < 1e-12 and B[0, 0] == 0.0          # 0 Hz is below 20 Hz
    # FFT bin 100 (3125 Hz): between centres j and j+1 -> the two weights are (1 - t) and t
    j = int((mel(3125.0) - lo) / d) - 1
    t = (mel(3125.0) - (lo + (j + 1) * d)) / d
    assert abs(B[j, 100] - (1 - t)) < 1e-12 and abs(B[j + 1, 100] - t) < 1e-12 and abs(B[:, 100].sum() - 1.0) < 1e-12
    assert np.count_nonzero(B[:, 100]) == 2
    # last triangle ends exactly at Nyquist; the highest FFT bin (7968.75 Hz) is on its falling edge
    assert abs(B[79, 255] - (hi - mel(7968.75)) / d) < 1e-12
    assert np.abs(B - np.asarray(okf.get_mel_banks(80, 512, 16000.0))).max() < 1e-12   # the oracle's scalar loops


def test_whale_extractor_numpy_variant_vs_reference_and_documented_deviation():
    """f#1: WhaleFeatureExtractor(mel_variant="hf_numpy") reproduces the reference's OWN extractor
    (web_demo/vllm_tools/model_weight_file/processor_whale.py run here by oracle/make_golden.py: numpy fallback,
    dither 0), and the default Kaldi filter bank deviates from that fallback by the documented amounts
    (filters up to 0.117, q1.wav log-mel up to 2.95: the fallback uses an FFT bin width of sr/510)."""
    from vita_amd.audio_frontend import WhaleFeatureExtractor, kaldi_fbank, kaldi_mel_banks
    g = np.load(os.path.join(GOLD, "q1_audio.npz"))
    wav = g["pcm16"].astype(np.float32) / 32768.0
    raw = kaldi_fbank(g["pcm16"].astype(np.float64), int(g["sr"]), mel_variant="hf_numpy")
    assert np.abs(raw - g["whale_numpy_fbank"]).max() < 1e-5
    out = WhaleFeatureExtractor(mel_variant="hf_numpy")(wav, sampling_rate=int(g["sr"]))
    assert out["input_features"].shape == (1, 352, 80) and out["attention_mask"].sum() == 352
    assert np.abs(out["input_features"][0] - g["whale_numpy_input_features"]).max() < 1e-5
    fdiff = np.abs(kaldi_mel_banks() - kaldi_mel_banks(variant="hf_numpy")).max()
    assert 0.11 < fdiff < 0.12
    dev = np.abs(g["fbank"] - g["whale_numpy_fbank"])
    assert 2.9 < dev.max() < 3.0 and 0.08 < dev.mean() < 0.095
    with pytest.raises(ValueError):
        kaldi_mel_banks(variant="slaney")


def test_clip_image_processor_equals_hf():
    """A3: the host image processor against the installed HF CLIPImageProcessor configured as the reference's
    preprocessor_config.json (448 shortest edge, bicubic, centre crop, ImageNet mean/std): bit-identical on random
    non-square images."""
    from PIL import Image
    from transformers import CLIPImageProcessor
    from vita_amd.host.image_processing import IMAGENET_MEAN, IMAGENET_STD, make_image_processor
    ref = CLIPImageProcessor(size={"shortest_edge": 448}, crop_size={"height": 448, "width": 448}, do_resize=True,
                             do_center_crop=True, do_normalize=True, do_rescale=True, do_convert_rgb=True,
                             image_mean=list(IMAGENET_MEAN), image_std=list(IMAGENET_STD), resample=3)
    ip = make_image_processor(448)
    rng = np.random.default_rng(5)
    for (w, h) in [(448, 448), (640, 360), (301, 977), (1280, 720), (97, 53)]:
        img = Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
        a = ip.preprocess(img, return_tensors="pt")["pixel_values"].numpy()
        b = ref.preprocess(img, return_tensors="np")["pixel_values"]
        assert a.shape == b.shape == (1, 3, 448, 448)
        assert np.array_equal(a, np.asarray(b, dtype=a.dtype)), (w, h, np.abs(a - b).max())


def test_vllm_placeholder_expansion_vs_reference_function():
    """f#1: serving.expand_placeholders + the splice widths against `new_token_ids` of the reference's OWN
    repeat_and_pad_image_tokens / get_audio_feature_size (mixtral.py:100-190,283-287, executed by
    oracle/make_golden.py), and the same case evaluated by hand."""
    from PIL import Image
    from vita_amd.host.constants import AUDIO_TOKEN_INDEX, IMAGE_TOKEN_INDEX
    from vita_amd.serving import audio_feature_size, expand_placeholders, expanded_token_ids
    g = np.load(os.path.join(GOLD, "vllm_expand.npz"))
    IMG, AUD = 51000, 51001
    ids = g["ids"].tolist()
    images = [Image.new("RGB", tuple(int(v) for v in wh)) for wh in g["sizes"]]
    frames = g["frames"].tolist()
    sent, tiles = expand_placeholders(ids, images, [np.zeros((n, 80), np.float32) for n in frames],
                                      image_token_index=IMG, audio_token_index=AUD, image_size=448)
    assert len(tiles) == int(g["tiles"].sum()) == 4 and [audio_feature_size(n) for n in frames] == g["audio_sizes"].tolist()
    assert sent == [1, 5, 6, IMAGE_TOKEN_INDEX, 7, AUDIO_TOKEN_INDEX, 8] + [IMAGE_TOKEN_INDEX] * 3 + [9, AUDIO_TOKEN_INDEX, 10]
    full = expanded_token_ids(sent, frames, image_token_index=IMG, audio_token_index=AUD)
    assert full == g["new_token_ids"].tolist()
    by_hand = [1, 5, 6] + [IMG] * 256 + [7] + [AUD] * 44 + [8] + [IMG] * 768 + [9] + [AUD] * 124 + [10]
    assert full == by_hand
    with pytest.raises(ValueError):
        expand_placeholders(ids, images[:1], [None, None], image_token_index=IMG, audio_token_index=AUD, image_size=448)


def test_image_processor_normalisation():
    from PIL import Image
    from vita_amd.host.image_processing import IMAGENET_MEAN, IMAGENET_STD, make_image_processor, process_images
    ip = make_image_processor(448)
    img = Image.new("RGB", (448, 448), (255, 0, 128))
    px = ip.preprocess(img, return_tensors="pt")["pixel_values"]
    assert tuple(px.shape) == (1, 3, 448, 448)
    exp = [(v / 255.0 - m) / s for v, m, s in zip((255, 0, 128), IMAGENET_MEAN, IMAGENET_STD)]
    assert np.allclose(px[0, :, 10, 10].numpy(), exp, atol=1e-6)
    out = process_images([Image.new("RGB", (600, 300))], ip, "pad")
    assert tuple(out.shape) == (1, 3, 448, 448)


def test_c_abi_exports_every_declared_symbol():
    """the library loads without a GPU and exports exactly what include/vita_hip.h declares."""
    from vita_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vita_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vh_[a-z0-9_]+)\s*\(", hdr)) - {"vh_allreduce_fn"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    assert lib.vh_version() >= 100
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (vh_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    # argument validation happens before any launch, so it is testable without a GPU
    assert lib.vh_gemm(None, None) != 0 and b"null" in lib.vh_last_error()
    cfg = _lib.MixtralCfg()
    cfg.head_dim = 64
    assert lib.vh_mixtral_workspace_bytes(ctypes.byref(cfg)) == 0 and b"head_dim" in lib.vh_last_error()


def test_removed_knobs_are_rejected():
    """vh_tune refuses keys of variants that no longer exist (a script that still sets them must fail loudly)."""
    import pytest
    from vita_amd import _lib
    for key in ("gateup_variant", "down_grid", "dec_prefetch", "batch_moe", "fuse_attn_oproj", "prefill_moe_gemm", "attn_presplit", "gemm_tall",
                "ps_rtcap", "reduce_wave", "dec_overlap", "dec_gateup_rp"):
        with pytest.raises(_lib.VitaHipError):
            _lib.tune(key, 1)
    _lib.tune("attn_rows", 0)      # a live key is accepted (no GPU needed)
    assert len(KNOBS) <= 21
    for key in KNOBS:
        _lib.tune(key, {"batch_moe_min": 3, "batch_decode": 1, "prefill_fuse_rows": 1, "ps_cfg": -1, "ps_nt": -1, "tp_overlap": 1, "moe_ksplit": -4, "attn_fa": 1,
                        "dec_fused": -1, "comm_ranks_per_device": 1, "attn_img": 1, "attn_xcd": 1, "ps_xcd": -1}.get(key, 0))    # (every key back at its default)


# r04's 15 keys + r06's dec_fused (the attention block of a decode layer as one launch; 0 = three launches) and comm_ranks_per_device
# (r05's dec_overlap — the side-stream schedule — is gone with its machinery)
KNOBS = ("batch_moe_min", "batch_decode", "attn_impl", "attn_fa", "attn_rows", "attn_ksplit", "prefill_attn_gemm",
         "prefill_fuse_rows", "ps_cfg", "ps_nt", "tp_overlap", "moe_ksplit", "force_allreduce", "tp_fuse", "comm_allow_coarse",
         "comm_ranks_per_device", "dec_fused", "dec_gateup_grid", "attn_img", "attn_xcd", "ps_xcd")


def test_ctypes_structs_match_the_header(tmp_path):
    """every struct the Python binding mirrors has the size AND field offsets the C header gives it (compiled here with gcc):
    the binding and the header are edited by hand in two places."""
    from vita_amd import _lib
    pairs = {"vh_gemm_args": _lib.GemmArgs, "vh_gemm_ps_args": _lib.GemmPsArgs, "vh_attn_args": _lib.AttnArgs,
             "vh_encoder_layer_args": _lib.EncoderLayerArgs, "vh_mixtral_cfg": _lib.MixtralCfg, "vh_mixtral_layer": _lib.MixtralLayer,
             "vh_vit_embed_args": _lib.VitEmbedArgs}
    lines = []
    for cname, cls in pairs.items():
        lines.append(f'printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('printf("\\n");')
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vita_hip.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for line in out:
        cname, size, *offs = line.split()
        cls = pairs[cname]
        assert ctypes.sizeof(cls) == int(size), f"{cname}: ctypes {ctypes.sizeof(cls)} vs C {size}"
        got = [getattr(cls, f).offset for f, _ in cls._fields_]
        assert got == [int(o) for o in offs], f"{cname}: field offsets differ: {got} vs {offs}"


def test_workspace_size_real_geometry():
    from vita_amd import _lib
    lib = _lib.load()
    c = _lib.MixtralCfg(hidden=4096, n_layers=32, n_q_heads=32, n_kv_heads=8, head_dim=128, inter=14336, n_experts=8,
                        top_k=2, vocab=51760, rms_eps=1e-5, max_ctx=2048, max_prefill=1024, max_new=1024, tp_rank=0,
                        tp_world=1, nsplit=0, logit_rows=0)
    n = lib.vh_mixtral_workspace_bytes(ctypes.byref(c))
    kv = 2 * 32 * 8 * 2048 * 128 * 4
    assert kv < n < kv + (1 << 30)


def test_tp_slices_partition():
    from vita_amd.checkpoint import tp_slices
    from vita_amd.config import TextConfig
    t = TextConfig()
    for world in (1, 2, 4, 8):
        qs, kvs, ffs = zip(*[tp_slices(t, r, world) for r in range(world)])
        for parts, total in ((qs, 4096), (kvs, 1024), (ffs, 14336)):
            assert parts[0].start == 0 and parts[-1].stop == total
            assert all(a.stop == b.start for a, b in zip(parts, parts[1:]))
    with pytest.raises(ValueError):
        tp_slices(t, 0, 3)


def test_ops_refuse_cpu_tensors():
    import torch
    from vita_amd import _lib, ops
    with pytest.raises(_lib.VitaHipError):
        ops.gemm(torch.zeros(4, 64), torch.zeros(8, 64, dtype=torch.bfloat16))


def test_audio_side_files_match_reference_loaders(tmp_path):
    """train.yaml + global_cmvn (JSON and Kaldi text) parsed as the reference's whale/cmvn.py does."""
    import json
    import numpy as np
    from vita_amd.audio_config import load_cmvn, read_audio_encoder_dir
    rng = np.random.default_rng(4)
    n, d = 12345.0, 80
    s1 = rng.standard_normal(d) * 40 * n
    s2 = (rng.random(d) * 30 + (s1 / n) ** 2) * n
    pj, pk = tmp_path / "cmvn.json", tmp_path / "global_cmvn"
    pj.write_text(json.dumps({"mean_stat": s1.tolist(), "var_stat": s2.tolist(), "frame_num": n}))
    pk.write_text("[\n " + " ".join(repr(float(x)) for x in s1) + f" {n!r}\n " + " ".join(repr(float(x)) for x in s2) + " 0 ]\n")
    mean = s1 / n
    istd = 1.0 / np.sqrt(np.maximum(s2 / n - mean * mean, 1e-20))
    for path, is_json in ((pj, True), (pk, False)):
        m, i = load_cmvn(str(path), is_json)
        np.testing.assert_allclose(m, mean, rtol=1e-6); np.testing.assert_allclose(i, istd, rtol=1e-6)
    if os.path.isdir("/root/reference/vita"):
        from oracle import ref_harness as rh
        rh.install()
        import logging, sys, math  # noqa: F401  (the reference's kaldi loader uses names it never imports)
        from vita.model.multimodal_encoder.whale import cmvn as rc
        rc.logging, rc.sys = logging, sys
        for path, is_json in ((pj, True), (pk, False)):
            rm, ri = rc.load_cmvn(str(path), is_json)
            m, i = load_cmvn(str(path), is_json)
            np.testing.assert_allclose(m, rm, rtol=1e-6); np.testing.assert_allclose(i, ri, rtol=1e-6)
    (tmp_path / "train.yaml").write_text(
        "input_dim: 80\nis_json_cmvn: false\ndataset_conf:\n  resample_conf: {resample_rate: 16000}\n"
        "  fbank_conf: {num_mel_bins: 80, frame_length: 25, frame_shift: 10, dither: 1.0}\n"
        "encoder_conf:\n  transformer-dynamic-chunks: true\n")
    side = read_audio_encoder_dir(str(tmp_path))
    assert side["dataset_conf"]["fbank_conf"]["dither"] == 0.0 and len(side["overridden"]) == 2
    np.testing.assert_allclose(side["mean"], mean, rtol=1e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/vita"), reason="reference tree only exists in the build container")
def test_video_front_end_matches_reference_function(tmp_path):
    """The reference's OWN `_get_rawvideo_dec` (video_audio_demo.py:30-118), imported from its demo script with
    compat/ serving `vita.*` and `decord`, against vita_amd.host.video on the same synthetic frame container."""
    import importlib.util
    import sys
    import numpy as np
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "compat"))
    mine = lambda k: k in ("vita", "decord") or k.startswith(("vita.", "decord."))   # may be oracle/ref_harness stubs
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if mine(k)}
    try:
        spec = importlib.util.spec_from_file_location("ref_video_audio_demo", "/root/reference/video_audio_demo.py")
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)                      # module level only defines functions + imports
        from decord import VideoReader, cpu
        from vita_amd.host.image_processing import make_image_processor
        from vita_amd.host.video import get_rawvideo, sample_positions
        rng = np.random.default_rng(0)
        proc = make_image_processor(56)
        for T, fps, kw in ((70, 5.0, {}), (9, 3.0, {}), (400, 10.0, {}), (120, 4.0, {"s": 3, "e": 11})):
            path = str(tmp_path / f"v{T}.npz")
            np.savez(path, frames=rng.integers(0, 255, size=(T, 40, 60, 3), dtype=np.uint8), fps=fps)
            want, n_want = ref._get_rawvideo_dec(path, proc, max_frames=16, min_frames=4, video_framerate=1,
                                                 image_aspect_ratio="pad", **kw)
            got, n_got = get_rawvideo(VideoReader(path, ctx=cpu(0)), proc, **kw)
            assert n_got == n_want and 4 <= n_got <= 16
            assert torch.equal(got, want)
        assert sample_positions(3, 30.0) == [0, 0, 0, 0]                     # shorter than a second: frame 0 repeated
    finally:
        sys.path.pop(0)
        for k in [k for k in sys.modules if mine(k)]:
            sys.modules.pop(k)
        sys.modules.update(saved)


def test_asset_tiles_through_the_host_image_processor():
    """the reference's own asset (asset/vita_log2.png -> 5 tiles by ITS dynamic_preprocess, tests/golden/assets_request.npz):
    the product's CLIP-style processor reproduces HF CLIPImageProcessor's pixel tensor on them (corner patch + per-tile
    mean recorded by oracle/make_golden_assets.py) — the host half of the released-geometry assets request of
    tests/test_assets_gpu.py."""
    from PIL import Image
    from vita_amd.host.image_processing import make_image_processor
    g = np.load(os.path.join(GOLD, "assets_request.npz"))
    tiles = [Image.fromarray(t) for t in g["tiles"]]
    assert len(tiles) == 5 and tiles[0].size == (448, 448)
    pix = make_image_processor(448).preprocess(tiles, return_tensors="np")["pixel_values"]
    pix = np.asarray(pix, np.float32)
    assert pix.shape == (5, 3, 448, 448)
    np.testing.assert_allclose(pix[:, :, :8, :8], g["pix_check"], atol=1e-6)
    np.testing.assert_allclose(pix.mean(axis=(1, 2, 3)), g["pix_mean"], atol=1e-6)
