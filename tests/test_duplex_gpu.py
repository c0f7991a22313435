"""Duplex serving with two REAL engine processes (replicas) sharing this GPU — BASELINE configs[4] in miniature:
requests alternate between the engines (baton), answers equal the single-engine answers, and the per-request
hand-off latency (request taken -> first streamed chunk) is reported."""
import json
import os
import time

import numpy as np
import pytest
from PIL import Image

from tests import tiny_ckpt

pytestmark = pytest.mark.gpu
IMG_ID, AUD_ID = 990, 991


def _drain(q, n, timeout):
    out, t0 = [], time.time()
    while len(out) < n and time.time() - t0 < timeout:
        try:
            out.append(q.get(timeout=0.1))
        except Exception:
            pass
    return out


@pytest.mark.timeout(300)
def test_two_engine_duplex(tmp_path, dev):
    from vita_amd.duplex import DuplexServer, make_serving_llm
    from vita_amd.serving import LLM, SamplingParams
    d = str(tmp_path)
    tiny_ckpt.write(d, seed=41)
    cj = os.path.join(d, "config.json")
    with open(cj) as f:
        j = json.load(f)
    j.update(image_token_index=IMG_ID, audio_token_index=AUD_ID)
    with open(cj, "w") as f:
        json.dump(j, f)
    rng = np.random.default_rng(1)
    imgs = [Image.fromarray(rng.integers(0, 255, size=(56, 56, 3), dtype=np.uint8)) for _ in range(2)]
    reqs = [{"prompt_token_ids": [1, 5, 6, IMG_ID, 7 + i], "multi_modal_data": {"image": [imgs[i]]}, "request_id": i,
             "prompt": f"q{i}"} for i in range(2)]
    sp = SamplingParams(temperature=0.01, max_tokens=12)
    ref = LLM(model=d, max_new_tokens=64)
    want = [ref.generate({k: r[k] for k in ("prompt_token_ids", "multi_modal_data")}, sp)[0].outputs[0] for r in reqs]

    srv = DuplexServer(make_serving_llm, (d, 64), sampling_params=sp)
    try:
        srv.wait_ready(timeout=240)
        stats = []
        for r in reqs:                                   # one at a time: no interruption, pure baton alternation
            srv.submit(r)
            stats += _drain(srv.stats, 1, timeout=60)
        assert [s["id"] for s in stats] == [0, 1], stats
        assert all(not s["negative"] and s["take_to_first_chunk_s"] is not None for s in stats)
        hist = list(srv.history)
        assert [h["prompt"] for h in hist] == ["q0", "q1"]
        for h, w in zip(hist, want):
            assert h["response"].replace("<1> ", "").replace("<1>", "") == w.text.replace("<1> ", "").replace("<1>", "")
        # overlapped requests: the second is taken by the OTHER engine while the first may still be speaking
        srv.submit(dict(reqs[0], request_id=10)); srv.submit(dict(reqs[1], request_id=11))
        st2 = sorted(_drain(srv.stats, 2, timeout=60), key=lambda s: s["request"])
        assert sorted(s["id"] for s in st2) == [0, 1]
        print("hand-off (request taken -> first chunk), s:", [round(s["take_to_first_chunk_s"], 4) for s in stats + st2],
              "interrupted:", [s["interrupted"] for s in st2])
    finally:
        srv.close()
