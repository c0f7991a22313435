"""AsyncLLMEngine (vita_amd/serving.py) plumbing on a scripted engine, no GPU: several asyncio consumers in flight,
cumulative outputs, a consumer that leaves early frees its pages (abort), an error while preparing ONE request reaches only
that request's iterator, duplicate request ids are refused, shutdown joins the scheduler thread."""
import asyncio
import os
import sys
import types

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_scheduler_cpu import FakeEngine, _emb, _expected   # noqa: E402


class _Tok:
    def decode(self, ids, skip_special_tokens=False):
        return " ".join(str(i) for i in ids)


class FakeLLM:
    """the surface AsyncLLMEngine uses: model.engine / model.device / model.model.embed_tokens, _prepare, _embed, tokenizer."""

    def __init__(self, pool_tokens=64 * 16, max_seqs=3):
        eng = FakeEngine(pool_tokens, max_seqs, max_new=64)
        table = torch.eye(FakeEngine.V)
        self.model = types.SimpleNamespace(engine=eng, device=torch.device("cpu"),
                                           model=types.SimpleNamespace(embed_tokens=lambda ids: table[ids]))
        self.tokenizer = _Tok()

    def _prepare(self, inputs, sp):
        ids = list(inputs["prompt_token_ids"])
        if any(i < 0 for i in ids):
            raise ValueError("placeholder without data")
        return dict(ids=ids, max_tokens=min(sp.max_tokens, 64), eos=set(sp.stop_token_ids or []))

    def _embed(self, req):
        return _emb(req["ids"])


@pytest.fixture(autouse=True)
def _no_cuda_sync(monkeypatch):
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: type("S", (), {"synchronize": lambda self: None})())


def test_concurrent_streams_abort_and_errors():
    from vita_amd.serving import AsyncLLMEngine, SamplingParams
    llm = FakeLLM()
    engine = AsyncLLMEngine(llm, max_num_seqs=3, window=1)
    eng = llm.model.engine
    prompts = {"a": [1, 2, 3], "b": [4, 5], "c": [6, 7, 8, 9], "d": [10]}

    async def full(rid, max_tokens):
        seen = []
        async for out in engine.generate({"prompt_token_ids": prompts[rid]}, SamplingParams(max_tokens=max_tokens), request_id=rid):
            seen.append((list(out.outputs[0].token_ids), out.outputs[0].text, out.finished, out.outputs[0].finish_reason))
        return seen

    async def leaves_early(rid):
        n = 0
        async for out in engine.generate({"prompt_token_ids": prompts[rid]}, SamplingParams(max_tokens=60), request_id=rid):
            n += 1
            if n == 3:
                break                                   # the demo does this on a noise verdict / interruption
        return n

    async def bad():
        with pytest.raises(ValueError):
            async for _ in engine.generate({"prompt_token_ids": [1, -5]}, SamplingParams(max_tokens=4), request_id="bad"):
                pass
        return True

    async def duplicate():
        g1 = engine.generate({"prompt_token_ids": prompts["d"]}, SamplingParams(max_tokens=30), request_id="dup")
        first = await g1.__anext__()
        with pytest.raises(ValueError):
            async for _ in engine.generate({"prompt_token_ids": prompts["d"]}, SamplingParams(max_tokens=2), request_id="dup"):
                pass
        await g1.aclose()                               # closing the first iterator aborts it
        return first

    async def main():
        return await asyncio.gather(full("a", 9), full("b", 5), leaves_early("c"), bad(), duplicate(), full("d", 7))

    a, b, n_c, ok, first, d = asyncio.run(main())
    for seen, rid, n in ((a, "a", 9), (b, "b", 5), (d, "d", 7)):
        assert seen[-1][2] and seen[-1][3] == "length" and not any(f for _, _, f, _ in seen[:-1])
        assert seen[-1][0] == _expected(prompts[rid], n)
        for (t0, x0, _, _), (t1, x1, _, _) in zip(seen, seen[1:]):
            assert t1[:len(t0)] == t0 and x1.startswith(x0)                 # cumulative
    got_first = first.outputs[0].token_ids
    assert n_c == 3 and ok and 1 <= len(got_first) <= 2 and got_first == _expected(prompts["d"], len(got_first))
    # everything was returned: the early leaver's and the duplicate's pages too (aborts are processed by the scheduler thread)
    for _ in range(200):
        if not engine.batcher.has_work() and eng.free == 16:
            break
        import time
        time.sleep(0.01)
    assert not engine.batcher.has_work() and eng.free == 16 and not eng.seqs
    engine.shutdown()
    assert not engine._thread.is_alive()
