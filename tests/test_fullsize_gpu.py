"""Full BASELINE geometry (VITA-Mixtral-8x7B: 32 layers, 4096 hidden, 8 experts top-2, vocab 51760 —
93.7 GB of bf16 weights generated on the GPU).  The oracle comparison at this size is tests/test_realgeom_gpu.py
(layer-streamed fp32 oracle on the bench's own request); this file adds the size-independent properties of the path:

  * the two implementations of the SAME function must agree: a decode step (GEMV kernels, split-KV
    attention, on-device routing) == the last row of a prefill (MFMA GEMMs, flash attention, sorted
    grouped experts) over the same tokens;
  * chunked prefill (pos0 > 0, KV cache append) == one-shot prefill;
  * greedy decoding is deterministic: two runs give bit-identical ids and logits;
  * one C call for n steps == n calls of one step.
"""
import numpy as np
import pytest
import torch

from vita_amd.checkpoint import synth_mixtral_device
from vita_amd.config import VitaConfig

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north-star logits tolerance (fp32)


@pytest.fixture(scope="module")
def full(dev):
    from vita_amd.engine import MixtralEngine
    cfg = VitaConfig()
    packed = synth_mixtral_device(cfg, dev, seed=0)
    eng = MixtralEngine(cfg, packed, dev, max_ctx=512, max_prefill=256, max_new=40, logit_rows=40)
    yield cfg, packed, eng
    eng.close()


def _emb(packed, ids, dev):
    return packed["embed"][torch.as_tensor(ids, device=dev)].float()


def test_decode_step_equals_prefill_row(full, dev):
    cfg, packed, eng = full
    rng = np.random.default_rng(0)
    S, n_new = 130, 5            # context crosses two 64-key decode tiles
    ids = rng.integers(3, cfg.text.vocab_size, size=S).tolist()
    eng.prefill(_emb(packed, ids, dev))
    eng.decode(n_new - 1)
    torch.cuda.synchronize()
    toks = eng.generated()
    dec_logits = eng.logits_all[:n_new].clone()
    assert len(toks) == n_new
    # re-derive every decode step's logits with the PREFILL path over prompt + generated prefix
    for i in range(1, n_new):
        lg, _ = eng.prefill(_emb(packed, ids + toks[:i], dev))
        torch.cuda.synchronize()
        err = float((lg - dec_logits[i]).abs().max())
        print(f"step {i}: max|decode - prefill| = {err:.2e}, argmax {int(lg.argmax())} vs token {toks[i]}")
        assert err < TOL
        assert int(lg.argmax()) == toks[i]


def test_chunked_prefill_equals_one_shot(full, dev):
    cfg, packed, eng = full
    rng = np.random.default_rng(1)
    ids = rng.integers(3, cfg.text.vocab_size, size=150).tolist()
    one, _ = eng.prefill(_emb(packed, ids, dev))
    one = one.clone()
    eng.prefill(_emb(packed, ids[:83], dev))
    two, _ = eng.prefill(_emb(packed, ids[83:], dev), pos0=83)
    torch.cuda.synchronize()
    err = float((one - two).abs().max())
    print(f"max|one-shot - chunked| = {err:.2e}")
    assert err < TOL and int(one.argmax()) == int(two.argmax())


def test_deterministic_and_batched_steps(full, dev):
    cfg, packed, eng = full
    rng = np.random.default_rng(2)
    emb = _emb(packed, rng.integers(3, cfg.text.vocab_size, size=70).tolist(), dev)
    runs = []
    for mode in ("batched", "single", "batched"):
        eng.prefill(emb)
        if mode == "batched":
            eng.decode(15)
        else:
            for _ in range(15):
                eng.decode(1)
        torch.cuda.synchronize()
        runs.append((eng.generated(), eng.logits_all[:16].clone()))
    for toks, lg in runs[1:]:
        assert toks == runs[0][0]
        assert torch.equal(lg, runs[0][1])           # bit-identical: no atomics, fixed reduction order
    assert int(eng.counters[0].item()) == 70 + 15
