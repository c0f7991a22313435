#!/usr/bin/env python
"""BASELINE configs[4] at the released geometry on ONE MI355X: two engine replicas (2 x 93 GB of bf16 weights
fit in 288 GB of HBM) under vita_amd.duplex.DuplexServer, fed with the interactive demo's request shape
(4 webcam frames + one spoken query, web_interactive_demo.py:240,694).  Reports, per request: the hand-off
latency (request taken -> first streamed chunk = encoders + prefill + first decode window + host pre-processing),
the streaming rate after the first chunk, and — for two overlapped requests — whether the monitor interrupted
the speaker.  Weights are synthetic (seeded N(0, 0.02)), so the text is meaningless; timing is not.

    python profiles/duplex_fullsize.py [out.json]        (needs ~200 GB of free HBM)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MAX_NEW = 48


class IdTokenizer:
    """stand-in tokenizer (no vocabulary offline): token id -> 't<id> '."""

    def decode(self, ids, skip_special_tokens=False):
        return "".join(f"t{int(i)} " for i in ids)


def make_llm(max_new=MAX_NEW):
    import torch
    from vita_amd.checkpoint import synth_mixtral_device, synth_state_dict
    from vita_amd.config import VitaConfig
    from vita_amd.model.vita_mixtral import VITAMixtralForCausalLM
    from vita_amd.serving import LLM
    cfg = VitaConfig()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    packed = synth_mixtral_device(cfg, dev, seed=0)
    sd = synth_state_dict(cfg, seed=1, rich=False, parts=("vision", "audio"))
    model = VITAMixtralForCausalLM(cfg, sd, device=dev, packed_llm=packed, max_new_tokens=max_new + 8,
                                   max_prefill=2048, keep_scores=False)
    model.get_vision_tower().load_model()
    llm = LLM.__new__(LLM)                      # the serving object around a prebuilt model (no checkpoint dir offline)
    llm.limit_mm = {"image": 256, "audio": 50}
    llm.tokenizer, llm.model = IdTokenizer(), model
    llm.image_processor = model.get_vision_tower().image_processor
    llm.image_token_index, llm.audio_token_index = 51000, 51001
    llm.min_dynamic_patch, llm.max_dynamic_patch, llm.use_thumbnail = 1, 12, True
    llm._n = 0
    # one warm-up request so the first measured one does not pay first-launch costs
    from vita_amd.serving import SamplingParams
    llm.generate(request(0), SamplingParams(temperature=0.01, max_tokens=4))
    torch.cuda.synchronize()
    return llm


def request(seed, n_frames=4, audio_frames=352):
    """4 frames of 448x448 (one tile each) + a 3.5 s query (352 fbank frames -> 44 tokens, the size of asset/q1.wav)
    + 140 stand-in system/text ids."""
    import numpy as np
    import torch
    from PIL import Image
    rng = np.random.default_rng(seed)
    imgs = [Image.fromarray(rng.integers(0, 255, size=(448, 448, 3), dtype=np.uint8)) for _ in range(n_frames)]
    feats = torch.from_numpy(rng.standard_normal((audio_frames, 80)).astype(np.float32))
    ids = [1] + rng.integers(3, 50000, size=139).tolist() + [51000] * n_frames + [51001]
    return {"prompt_token_ids": ids, "multi_modal_data": {"image": imgs, "audio": [feats]}, "request_id": seed,
            "prompt": f"q{seed}"}


def drain(q, n, timeout):
    out, t0 = [], time.time()
    while len(out) < n and time.time() - t0 < timeout:
        try:
            out.append(q.get(timeout=0.1))
        except Exception:
            pass
    return out


def main():
    from vita_amd.duplex import DuplexServer
    from vita_amd.serving import SamplingParams
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "duplex_fullsize.json")
    sp = SamplingParams(temperature=0.01, max_tokens=MAX_NEW)
    t0 = time.time()
    srv = DuplexServer(make_llm, (), sampling_params=sp)
    res = {"what": "two full-size VITA-Mixtral-8x7B replicas on one MI355X under DuplexServer; request = 4 frames "
                   "(448x448, 1 tile each -> 1024 tokens) + 352-frame audio (44 tokens) + 140 ids; "
                   f"max_tokens={MAX_NEW}; stream window = 2 tokens"}
    try:
        srv.wait_ready(timeout=400)
        res["both_engines_ready_s"] = round(time.time() - t0, 1)
        solo = []
        for i in (1, 2, 3, 4):                           # one at a time: engines alternate (baton), no interruption
            srv.submit(request(i))
            solo += drain(srv.stats, 1, timeout=60)
        res["solo"] = solo
        srv.submit(request(11))                          # overlapped: the second request goes to the OTHER engine,
        srv.submit(request(12))                          # whose first chunk interrupts the first speaker
        res["overlapped"] = sorted(drain(srv.stats, 2, timeout=60), key=lambda s: s["request"])
        ok = [s for s in solo if s.get("take_to_first_chunk_s") is not None]
        if ok:
            res["handoff_ms_median"] = round(1e3 * sorted(s["take_to_first_chunk_s"] for s in ok)[len(ok) // 2], 2)
            rates = [(s["n_tokens"] - 2) / s["first_chunk_to_end_s"] for s in ok if s["first_chunk_to_end_s"]]
            res["solo_stream_tokens_per_s_median"] = round(sorted(rates)[len(rates) // 2], 1) if rates else None
        res["engine_ids_solo"] = [s["id"] for s in solo]
    finally:
        srv.close()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
