"""concurrent sequences over the paged KV cache (vh_mixtral_seq_*), the iteration-level scheduler and the
AsyncLLMEngine-shaped iterator — what vLLM gives the reference's serving plugin as `kv_caches` + block tables
(web_demo/vllm_tools/vllm_file/mixtral.py:491-501,1130-1186; web_interactive_demo.py:315-328,942-951).

Bar: every request's greedy ids equal the fp32 oracle's ids for THAT request alone (bit-exact), whatever else is
scheduled next to it and wherever its pages lie in the pool; prefill scores within 1e-3."""
import asyncio
import json
import os

import numpy as np
import pytest
import torch

from oracle import mixtral as om
from tests.util import assert_close, to_np
from vita_amd.checkpoint import pack_mixtral, synth_state_dict
from vita_amd.config import VitaConfig

pytestmark = pytest.mark.gpu


def _setup(dev, lens, n_new, seed=0, pool=1024, max_seqs=4, max_new=None, text=None):
    from vita_amd.engine import MixtralEngine
    cfg = VitaConfig.tiny()
    if text is not None:
        cfg.text = text
    sd = synth_state_dict(cfg, seed=seed, parts=("text",))
    rng = np.random.default_rng(seed + 7)
    orc = om.MixtralOracle(sd, cfg.text)
    reqs = []
    for S in lens:
        ids_in = rng.integers(3, cfg.text.vocab_size, size=S)
        emb = sd["model.embed_tokens.weight"][ids_in]
        orc.reset()
        ref_ids, ref_lg = orc.greedy(emb, n_new)
        reqs.append(dict(emb=torch.from_numpy(emb).to(dev), ref_ids=ref_ids, ref_lg=ref_lg))
    eng = MixtralEngine(cfg, pack_mixtral(sd, cfg, dev), dev, max_ctx=pool, max_prefill=max(lens) + n_new + 8,
                        max_new=max_new or (n_new + 4), max_seqs=max_seqs)
    return cfg, sd, eng, reqs


def _ids(eng, s, n):
    torch.cuda.synchronize()
    eng.check_device_flag(eng.seq_counters(s).tolist())
    return eng.seq_tokens(s)[:n].tolist()


@pytest.mark.parametrize("flash", [False, True])
def test_two_interleaved_sequences_match_oracle(dev, flash):
    """A (120 tokens) starts, decodes a little, B (70 tokens) arrives: B's pages land BETWEEN A's, then both advance in
    batched iterations across page boundaries (A crosses 128, B crosses 64 -> wait, 70 > 64: B crosses 128 never; A does).
    flash: the released 4 : 1 head grouping with the flash-form prefill attention forced (k_attn_fa through the page table)."""
    from vita_amd import _lib
    from vita_amd.config import TextConfig
    n_new = 24
    text = TextConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2, intermediate_size=1024,
                      num_local_experts=8, vocab_size=2000) if flash else None
    _lib.tune("attn_fa", 2 if flash else 1)
    try:
        _two_interleaved(dev, n_new, text)
    finally:
        _lib.tune("attn_fa", 1)


def _two_interleaved(dev, n_new, text):
    _, _, eng, (A, B) = _setup(dev, [120, 70], n_new, text=text)
    total = eng.pages_free()
    a = eng.seq_alloc()
    lg = eng.seq_prefill(a, A["emb"], want_logits=True)
    assert_close("A prefill scores", to_np(lg), A["ref_lg"][0], atol=1e-3)
    eng.seq_decode([a])
    eng.seq_decode([a])
    b = eng.seq_alloc()
    lg = eng.seq_prefill(b, B["emb"], want_logits=True)
    assert_close("B prefill scores", to_np(lg), B["ref_lg"][0], atol=1e-3)
    for _ in range(n_new - 3):
        eng.seq_decode([a, b])            # one continuous-batching iteration
    eng.seq_decode([b])
    eng.seq_decode([b])
    pa, pb = eng.seq_pages(a), eng.seq_pages(b)
    assert len(pa) == 3 and len(pb) == 2 and not set(pa) & set(pb)
    assert pa != sorted(pa) or pa[2] != pa[1] + 1, (pa, pb)      # A's third page was taken after B's: not contiguous
    assert _ids(eng, a, n_new) == A["ref_ids"], "sequence A differs from the oracle"
    assert _ids(eng, b, n_new) == B["ref_ids"], "sequence B differs from the oracle"
    assert eng.seq_pos(a) == 120 + n_new - 1 and eng.seq_pos(b) == 70 + n_new - 1
    eng.seq_free(a)
    eng.seq_free(b)
    assert eng.pages_free() == total
    eng.close()


def test_pages_are_reused_and_legacy_api_is_fenced(dev):
    from vita_amd._lib import VitaHipError
    n_new = 10
    _, _, eng, (A, B, Cq) = _setup(dev, [65, 33, 129], n_new, seed=3)
    a, b = eng.seq_alloc(), eng.seq_alloc()
    eng.seq_prefill(a, A["emb"])
    eng.seq_prefill(b, B["emb"])
    for _ in range(n_new - 1):
        eng.seq_decode([b, a])
    assert _ids(eng, a, n_new) == A["ref_ids"] and _ids(eng, b, n_new) == B["ref_ids"]
    with pytest.raises(VitaHipError):        # the single-sequence entry points would overwrite pages that are owned
        eng.prefill(A["emb"])
    freed = eng.seq_pages(a)
    eng.seq_free(a)
    c = eng.seq_alloc()
    eng.seq_prefill(c, Cq["emb"])            # takes A's pages (stale K/V inside) plus fresh ones
    assert set(freed) <= set(eng.seq_pages(c))
    for _ in range(n_new - 1):
        eng.seq_decode([c, b])               # b keeps decoding next to the newcomer
    assert _ids(eng, c, n_new) == Cq["ref_ids"]
    assert c == a                            # the freed SLOT is reused as well
    eng.seq_free(b)
    eng.seq_free(c)
    with pytest.raises(VitaHipError):
        eng.seq_decode([c])                  # freed slot
    # with no live sequence the single-sequence path works on the same engine (contiguous rows of the same pool)
    eng.prefill(B["emb"])
    eng.decode(n_new - 1)
    torch.cuda.synchronize()
    assert eng.generated() == B["ref_ids"]
    eng.close()


def test_pool_and_slot_exhaustion(dev):
    from vita_amd._lib import VitaHipError
    _, _, eng, (A, B) = _setup(dev, [100, 100], 4, pool=192, max_seqs=2)      # 3 pages: A takes 2
    a, b = eng.seq_alloc(), eng.seq_alloc()
    with pytest.raises(VitaHipError):
        eng.seq_alloc()
    eng.seq_prefill(a, A["emb"])
    with pytest.raises(VitaHipError, match="pool exhausted"):
        eng.seq_prefill(b, B["emb"])
    assert eng.seq_pos(b) == 0 and eng.pages_free() == 1                   # nothing leaked by the refused prefill
    eng.seq_free(a)
    eng.seq_prefill(b, B["emb"])
    for _ in range(3):
        eng.seq_decode([b])
    assert _ids(eng, b, 4) == B["ref_ids"]
    eng.close()


def _drain(batcher, n_expected, limit=10000):
    got, fin = {}, {}
    for _ in range(limit):
        if not batcher.has_work():
            break
        for rid, new, finished, reason in batcher.step():
            got.setdefault(rid, []).extend(new)
            if finished:
                fin[rid] = reason
    assert len(fin) == n_expected, (fin, got)
    return got, fin


def test_continuous_batcher_matches_oracle(dev):
    """five requests through two sequence slots: admission in arrival order, finished sequences make room, eos stops
    one request early; each request's ids are the oracle's for that request alone."""
    from vita_amd.serving import ContinuousBatcher
    n_new = 12
    lens = [40, 97, 64, 130, 20]
    _, _, eng, reqs = _setup(dev, lens, n_new, seed=5, pool=1024, max_seqs=2)
    bt = ContinuousBatcher(eng, window=1)
    eos3 = reqs[3]["ref_ids"][4]
    first_hit = reqs[3]["ref_ids"].index(eos3)
    for i, r in enumerate(reqs):
        bt.add(f"r{i}", r["emb"], max_tokens=n_new if i != 1 else 5, eos={eos3} if i == 3 else ())
    got, fin = _drain(bt, len(reqs))
    for i, r in enumerate(reqs):
        exp = r["ref_ids"][:5] if i == 1 else (r["ref_ids"][:first_hit + 1] if i == 3 else r["ref_ids"])
        assert got[f"r{i}"] == exp, (i, got[f"r{i}"], exp)
    assert fin["r3"] == "stop" and fin["r1"] == "length" and fin["r0"] == "length"
    assert bt.stats["prefills"] == 5 and bt.stats["preemptions"] == 0
    assert eng.pages_free() == 1024 // 64
    eng.close()


@pytest.mark.parametrize("batched", [1, 3, 0])
def test_six_sequences_batched_kernels_match_oracle(dev, batched):
    """six sequences advance together: groups of 4 + 2 through the batched decode kernels (shared attention weights and LM
    head, every distinct routed expert streamed once) — or one after the other (batch_decode = 0); ids equal the oracle's
    per-request ids either way, and a late joiner / an early leaver change nothing for the others."""
    from vita_amd import _lib
    n_new = 14
    lens = [40, 97, 64, 130, 20, 75]
    _, _, eng, reqs = _setup(dev, lens, n_new, seed=11, pool=2048, max_seqs=6)
    # 1: default (iterations of >= 3 sequences run the MoE once on the weight-streaming GEMM, smaller ones per sequence),
    # 3: expert GEMVs per sequence at every size, 0: one sequence after the other
    _lib.tune("batch_decode", min(batched, 1))
    _lib.tune("batch_moe_min", 3 if batched == 1 else 0)
    try:
        seqs = []
        for r in reqs[:5]:
            sq = eng.seq_alloc()
            eng.seq_prefill(sq, r["emb"])
            seqs.append(sq)
        for _ in range(3):
            eng.seq_decode(seqs)                       # 5 sequences: 4 + 1
        late = eng.seq_alloc()
        eng.seq_prefill(late, reqs[5]["emb"])          # joins after three iterations
        for _ in range(3):
            eng.seq_decode(seqs + [late])              # 4 + 2
        assert _ids(eng, seqs[1], 7) == reqs[1]["ref_ids"][:7]
        eng.seq_free(seqs[1])                          # leaves early
        rest = [q for q in seqs if q != seqs[1]] + [late]
        for _ in range(n_new - 7):
            eng.seq_decode(rest)                       # 4 + 1
        for _ in range(3):
            eng.seq_decode([late])
        for i, q in enumerate(seqs):
            if i != 1:
                assert _ids(eng, q, n_new) == reqs[i]["ref_ids"], i
        assert _ids(eng, late, n_new) == reqs[5]["ref_ids"]
    finally:
        _lib.tune("batch_decode", 1)
        _lib.tune("batch_moe_min", 3)
    eng.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("moe_min", [3, 0])
def test_six_sequences_batched_kernels_real_width_match_oracle(dev, moe_min):
    """VERDICT r05 weak #2c: batched decode iterations at the RELEASED widths (H 4096, 32 / 8 heads, 8 experts of 14336 columns; 4
    layers, paged pool) — six sequences advance together through vh_mixtral_seq_decode: groups of 4 + 2 on the batched GEMV /
    attention / LM-head kernels, and the layer's MoE ONCE per iteration on the weight-streaming GEMM (k_gemm_ps tiles of <= 64 rows,
    device-chosen K split of the down projection, the few-row rules of r05: moe_min = 3, the default) or per sequence on the
    batch-1 expert GEMVs (moe_min = 0).  Every sequence's greedy ids == the layer-streamed fp32 oracle's for THAT sequence alone
    (oracle/stream.py, teacher-forced over its own tokens)."""
    from oracle import stream
    from vita_amd import _lib
    from vita_amd.checkpoint import synth_mixtral_device
    from vita_amd.engine import MixtralEngine
    n_new, layers, seed = 8, 4, 0
    lens = [40, 97, 64, 130, 20, 75]
    cfg = VitaConfig()
    cfg.text.num_hidden_layers = layers
    t = cfg.text
    packed = synth_mixtral_device(cfg, dev, seed=seed)
    eng = MixtralEngine(cfg, packed, dev, max_ctx=1024, max_prefill=160, max_new=n_new + 4, max_seqs=6)
    rng = np.random.default_rng(23)
    prompts = [rng.integers(3, t.vocab_size, size=S).tolist() for S in lens]
    _lib.tune("batch_moe_min", moe_min)
    try:
        seqs = []
        for ids in prompts:
            sq = eng.seq_alloc()
            eng.seq_prefill(sq, packed["embed"][torch.as_tensor(ids, device=dev)].float())
            seqs.append(sq)
        for _ in range(n_new - 1):
            eng.seq_decode(seqs)                       # 6 sequences: 4 + 2
        toks = [_ids(eng, q, n_new) for q in seqs]
    finally:
        _lib.tune("batch_moe_min", 3)
    eng.close()
    del eng, packed
    torch.cuda.empty_cache()
    fulls = [stream.embed_rows(t, ids + tk[:-1], seed) for ids, tk in zip(prompts, toks)]
    refs = stream.forward_many(t, seed, fulls, n_layers=layers, logits_from=[len(ids) - 1 for ids in prompts])
    for i, (tk, ref) in enumerate(zip(toks, refs)):
        ref_ids = ref["logits"].argmax(-1).tolist()
        assert tk == ref_ids, f"sequence {i} (prompt {lens[i]}): device {tk} vs oracle {ref_ids}"
    print(f"six sequences at the released widths, {layers} layers, batch_moe_min = {moe_min}: ids == the streamed oracle's")


def test_batcher_preempts_by_recompute_when_the_pool_runs_dry(dev):
    """4 pages for two sequences of 60 and 62 tokens that each grow past a page boundary and then need a third and a
    fourth page... the younger one is preempted, re-queued with prompt + generated tokens, and still ends with the
    oracle's ids (vLLM's recompute preemption)."""
    from vita_amd.serving import ContinuousBatcher
    n_new = 80
    _, sd, eng, reqs = _setup(dev, [60, 62], n_new, seed=9, pool=256, max_seqs=2, max_new=n_new + 4)
    table = torch.from_numpy(sd["model.embed_tokens.weight"]).to(dev)
    bt = ContinuousBatcher(eng, embed_tokens=lambda ids: table[ids], window=4)
    for i, r in enumerate(reqs):
        bt.add(i, r["emb"], max_tokens=n_new)
    got, fin = _drain(bt, 2)
    assert bt.stats["preemptions"] >= 1
    for i, r in enumerate(reqs):
        assert got[i] == r["ref_ids"], (i, got[i], r["ref_ids"])
    assert eng.pages_free() == 4
    eng.close()


def test_async_engine_concurrent_requests(tmp_path, dev):
    """the interactive demo's usage: `async for out in llm.generate(inputs, sampling_params, request_id)` with several
    requests in flight; every stream is cumulative and ends with the ids the one-at-a-time LLM.generate gives."""
    from tests import tiny_ckpt
    from vita_amd.serving import AsyncEngineArgs, AsyncLLMEngine, SamplingParams
    d = str(tmp_path / "ckpt")
    os.makedirs(d)
    tiny_ckpt.write(d, seed=31)
    engine = AsyncLLMEngine.from_engine_args(AsyncEngineArgs(model=d, dtype="float16", max_num_seqs=3, max_new_tokens=24,
                                                             kv_pool_tokens=2048, device="cuda:0"))
    prompts = [[1, 5, 6, 7, 8], [1, 9, 10, 11, 12, 13, 14, 15], [1, 20, 21], [1, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39]]
    sp = SamplingParams(temperature=0.01, max_tokens=9, skip_special_tokens=False)

    async def one(i):
        seen = []
        async for out in engine.generate({"prompt_token_ids": prompts[i]}, sampling_params=sp, request_id=f"q{i}"):
            seen.append((list(out.outputs[0].token_ids), out.outputs[0].text, out.finished))
        return seen

    async def main():
        return await asyncio.gather(*[one(i) for i in range(len(prompts))])

    streams = asyncio.run(main())
    assert engine.batcher.stats["prefills"] == 4 and not engine.batcher.has_work()
    engine.shutdown()
    llm = engine.llm
    for i, seen in enumerate(streams):
        assert seen and seen[-1][2] and not any(f for _, _, f in seen[:-1])
        for (t0, x0, _), (t1, x1, _) in zip(seen, seen[1:]):
            assert t1[:len(t0)] == t0 and x1.startswith(x0)                      # cumulative, as the demo's diffing expects
        exp = llm.generate({"prompt_token_ids": prompts[i]}, sampling_params=sp)[0].outputs[0]
        assert seen[-1][0] == exp.token_ids and seen[-1][1] == exp.text, (i, seen[-1], exp)
