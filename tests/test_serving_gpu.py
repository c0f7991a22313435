"""vLLM-shaped surface: the app-side contract of web_demo/web_ability_demo.py:203-232 —
one placeholder id per image / clip, PIL images and CMVN-normalised features in multi_modal_data,
SamplingParams(temperature=0.01) == greedy — must produce the tokens of the ORACLE (oracle/encoders.py +
oracle/mixtral.py on the same state dict and inputs), and the same tokens as the HF-flavour boundary."""
import json
import os
import time

import numpy as np
import pytest
import torch
from PIL import Image

from tests import tiny_ckpt

pytestmark = pytest.mark.gpu
IMG_ID, AUD_ID = 990, 991      # stand-ins for 51000 / 51001 inside the tiny 1000-entry vocabulary


@pytest.fixture(scope="module")
def served(tmp_path_factory):
    from vita_amd.serving import LLM
    d = str(tmp_path_factory.mktemp("vita_tiny_serving"))
    sd = tiny_ckpt.write(d, seed=31)
    cj = os.path.join(d, "config.json")
    with open(cj) as f:
        j = json.load(f)
    j.update(image_token_index=IMG_ID, audio_token_index=AUD_ID, min_dynamic_patch=1, max_dynamic_patch=12,
             use_thumbnail=True)
    with open(cj, "w") as f:
        json.dump(j, f)
    return LLM(model=d, dtype="float16", tensor_parallel_size=1, limit_mm_per_prompt={"image": 256, "audio": 50},
               max_new_tokens=32), sd, d


def test_generate_image_audio_request(served, dev):
    from vita_amd.audio_frontend import WhaleFeatureExtractor, kaldi_fbank
    from vita_amd.config import VitaConfig
    from vita_amd.host.image_processing import dynamic_preprocess
    from vita_amd.model.vita_mixtral import VITAMixtralForCausalLM
    from vita_amd.serving import SamplingParams, audio_feature_size
    llm, sd, d = served
    rng = np.random.default_rng(3)
    img = Image.fromarray(rng.integers(0, 255, size=(70, 170, 3), dtype=np.uint8))     # wide: several tiles + thumbnail
    wav = 0.1 * rng.standard_normal(16000)
    feats = WhaleFeatureExtractor()(wav[None], sampling_rate=16000, return_tensors="pt")["input_features"][0]
    ids = [1, 5, 6, 7, IMG_ID, 8, 9, AUD_ID, 10]
    outs = llm.generate({"prompt_token_ids": ids, "multi_modal_data": {"image": [img], "audio": [feats]}},
                        sampling_params=SamplingParams(temperature=0.01, max_tokens=10, best_of=1,
                                                       skip_special_tokens=False))
    got = outs[0].outputs[0].token_ids
    assert isinstance(outs[0].outputs[0].text, str) and 1 <= len(got) <= 10

    # the same request through the HF-flavour boundary: raw fbank (CMVN inside the encoder), one sentinel per tile
    cfg = VitaConfig.tiny()
    m = VITAMixtralForCausalLM(cfg, sd, device="cuda:0", max_new_tokens=32, max_prefill=2048)
    m.get_vision_tower().load_model()
    tiles, _ = dynamic_preprocess(img, min_num=1, max_num=12, image_size=cfg.vision.image_size, use_thumbnail=True)
    assert len(tiles) > 1
    pix = m.process_images(tiles, m.config).to("cuda:0")
    raw = torch.from_numpy(kaldi_fbank(wav * (1 << 15), 16000))
    sent = [1, 5, 6, 7] + [-200] * len(tiles) + [8, 9, -500, 10]
    ref = m.generate(torch.tensor([sent], device="cuda:0"), images=pix,
                     audios={"audios": raw[None].to("cuda:0"), "lengths": torch.tensor([raw.shape[0]], device="cuda:0")},
                     do_sample=False, num_beams=1, return_dict_in_generate=True, max_new_tokens=10)
    exp = ref.sequences[0, len(sent):].tolist()
    assert got == exp, (got, exp)
    # ... and through the oracle: tiles + raw fbank -> fp64 encoders -> splice -> fp32 Mixtral greedy
    from tests.oracle_e2e import oracle_generate
    o_ids, _, o_emb = oracle_generate(sd, cfg, sent, pix=pix, fbank=raw, n_new=10)
    assert o_emb.shape[0] == len(sent) - len(tiles) - 1 + len(tiles) * (cfg.vision.image_size // cfg.vision.patch_size // 2) ** 2 + \
        audio_feature_size(raw.shape[0])
    assert got == o_ids[:len(got)], (got, o_ids)
    assert len(got) == 10 or got[-1] == 2
    assert audio_feature_size(raw.shape[0]) == m.get_audio_encoder()(raw[None], torch.tensor([raw.shape[0]]))["inputs_embeds"].shape[1]


def test_text_only_and_errors(served, dev):
    from oracle import mixtral as om
    from vita_amd.config import VitaConfig
    from vita_amd.serving import SamplingParams
    llm, sd, _ = served
    out = llm.generate({"prompt_token_ids": [1, 5, 6, 7, 8]}, sampling_params=SamplingParams(max_tokens=4))
    got = out[0].outputs[0].token_ids
    assert 1 <= len(got) <= 4
    exp, _ = om.MixtralOracle(sd, VitaConfig.tiny().text).greedy(sd["model.embed_tokens.weight"][[1, 5, 6, 7, 8]], 4)
    assert got == exp[:len(got)] and (len(got) == 4 or got[-1] == 2), (got, exp)
    with pytest.raises(ValueError):          # placeholder without data (mixtral.py:244-247)
        llm.generate({"prompt_token_ids": [1, IMG_ID, 5]}, sampling_params=SamplingParams(max_tokens=2))
    with pytest.raises(NotImplementedError):
        llm.generate({"prompt_token_ids": [1, 5]}, sampling_params=SamplingParams(temperature=0.8))


# ---- tensor_parallel_size = 2: two LLM processes (one GPU, gloo bootstrap, the library's IPC all-reduce) -------------
def _tp_llm_worker(rank, world, port, ckpt, prompts, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK="0", VITA_AMD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from vita_amd.serving import LLM, SamplingParams
    llm = LLM(model=ckpt, dtype="float16", tensor_parallel_size=world, max_new_tokens=16)
    try:
        outs = llm.generate([{"prompt_token_ids": p} for p in prompts],
                            sampling_params=SamplingParams(temperature=0.01, max_tokens=8))
        ret[rank] = (llm.collective, [o.outputs[0].token_ids for o in outs])
        dist.barrier()
    finally:
        llm.model.engine.close()
        if dist.is_initialized():
            dist.destroy_process_group()


def test_llm_tensor_parallel_two_processes(served, dev):
    """`LLM(..., tensor_parallel_size=2)` under a one-process-per-GPU launch: each process builds its rank's shard, brings up
    the collective itself and serves the same requests; both ranks return the single-process tokens."""
    import socket
    import torch.multiprocessing as mp
    from oracle import mixtral as om
    from vita_amd.config import VitaConfig
    from vita_amd.serving import SamplingParams
    llm, sd, d = served
    prompts = [[1, 5, 6, 7, 8], [1, 9, 10, 11, 12, 13, 14, 15, 16, 17]]
    exp = [llm.generate({"prompt_token_ids": p}, sampling_params=SamplingParams(temperature=0.01, max_tokens=8))[0]
           .outputs[0].token_ids for p in prompts]
    for p, e in zip(prompts, exp):                      # the single-process tokens are the oracle's
        o, _ = om.MixtralOracle(sd, VitaConfig.tiny().text).greedy(sd["model.embed_tokens.weight"][p], 8)
        assert e == o[:len(e)] and (len(e) == 8 or e[-1] == 2), (e, o)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_tp_llm_worker, args=(2, port, d, prompts, ret), nprocs=2, join=True)
    assert ret[0][0] == ret[1][0] and ret[0][0] in ("ipc", "torch"), dict(ret)
    assert ret[0][1] == exp and ret[1][1] == exp, (dict(ret), exp)
