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
    """pos0 > 0 with KV-cache append against the one-shot prefill of the same 150 tokens.  The two schedules are not the same
    arithmetic: the K split of the projections / expert GEMMs is chosen on the device from the row count, so slabs are summed in
    a different order (1e-5 per layer), and a 32-layer top-2 router is discontinuous — ONE near-tied decision that flips moves
    the logits by 1e-2 (seen in r03: 9.2e-3 from this prompt once the attention rounding changed).  So the routing decisions of
    both runs are compared: every layer up to the first differing decision must agree tightly (that is the chunking logic:
    positions, KV append, causal offset), flips must be rare, and the logits bar applies whenever no decision flipped."""
    cfg, packed, eng = full
    L = cfg.text.num_hidden_layers
    rng = np.random.default_rng(1)
    ids = rng.integers(3, cfg.text.vocab_size, size=150).tolist()
    one, h1 = eng.prefill(_emb(packed, ids, dev), want_hidden=True, want_route=True)
    one, h1, r1 = one.clone(), h1[:, 83:].clone(), eng.route_ids[:, 83:].clone()
    eng.prefill(_emb(packed, ids[:83], dev))
    two, h2 = eng.prefill(_emb(packed, ids[83:], dev), pos0=83, want_hidden=True, want_route=True)
    r2 = eng.route_ids
    torch.cuda.synchronize()
    flipped = (torch.sort(r1, -1).values != torch.sort(r2, -1).values).any(-1)          # [layers, 67]
    layers_hit = torch.nonzero(flipped.any(-1)).flatten().tolist()
    first = layers_hit[0] if layers_hit else L
    n_flip = int(flipped.sum())
    err = float((one - two).abs().max())
    print(f"max|one-shot - chunked| = {err:.2e}; router decisions that differ: {n_flip} of {flipped.numel()}, first at layer {first if layers_hit else None}")
    for l in range(first):
        scale = max(1.0, float(h1[l].abs().max()))
        e = float((h1[l] - h2[l]).abs().max())
        assert e < 1e-3 * scale, f"hidden states after layer {l} differ by {e:.2e} before any routing decision does"
    n_first = int(flipped[first].sum()) if layers_hit else 0      # later layers see the consequences of the first flip
    assert first >= 2 and n_first <= 2, "chunked and one-shot prefill route differently early or often: not a near-tie effect"
    if not layers_hit:
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
