"""ContinuousBatcher (vita_amd/serving.py) on a scripted engine: the scheduling decisions — FIFO admission under slot and
page limits, one decode step per running sequence per iteration, eos / max_tokens, pages returned, abort, youngest-first
recompute preemption — without a GPU.  The engine model: a sequence's next token is a pure function of the tokens it has
consumed (prompt rows are one-hot ids), so a preempted and re-prefilled request must continue exactly where it was."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Cfg:
    def __init__(self, max_seqs):
        self.max_seqs = max_seqs


class FakeEngine:
    """seq_* surface of MixtralEngine: 64-token pages, per-sequence positions, tokens = f(history)."""
    V = 97

    def __init__(self, pool_tokens, max_seqs, max_new=64, max_prefill=512):
        self.c = _Cfg(max_seqs)
        self.max_ctx, self.max_new, self.max_prefill = pool_tokens, max_new, max_prefill
        self.free = pool_tokens // 64
        self.seqs = {}
        self.calls = []

    @staticmethod
    def _next(hist):
        h = 7
        for t in hist:
            h = (h * 31 + int(t) + 3) % 1000003
        return h % FakeEngine.V

    def pages_free(self):
        return self.free

    def seq_alloc(self):
        for s in range(self.c.max_seqs):
            if s not in self.seqs:
                self.seqs[s] = dict(hist=[], toks=[], pages=0)
                return s
        raise RuntimeError("no slot")

    def seq_free(self, s):
        self.free += self.seqs.pop(s)["pages"]

    def _grow(self, q, n_tokens):
        need = -(-n_tokens // 64) - q["pages"]
        if need > self.free:
            raise RuntimeError("KV pool exhausted")
        self.free -= need
        q["pages"] += need

    def seq_prefill(self, s, emb):
        q = self.seqs[s]
        ids = emb.argmax(dim=1).tolist()                # rows are one-hot ids
        self._grow(q, len(q["hist"]) + len(ids))
        q["hist"] += ids
        q["toks"] = [self._next(q["hist"])]
        self.calls.append(("prefill", s, len(ids)))

    def seq_decode(self, seqs):
        for s in seqs:
            q = self.seqs[s]
            self._grow(q, len(q["hist"]) + 1)
            q["hist"].append(q["toks"][-1])
            q["toks"].append(self._next(q["hist"]))
        self.calls.append(("decode", tuple(seqs)))

    def seq_pos(self, s):
        return len(self.seqs[s]["hist"])

    def seq_pages(self, s):
        return list(range(self.seqs[s]["pages"]))

    def seq_tokens(self, s):
        return torch.tensor(self.seqs[s]["toks"] + [0] * (self.max_new - len(self.seqs[s]["toks"])))

    def seq_counters(self, s):
        return torch.tensor([len(self.seqs[s]["hist"]), len(self.seqs[s]["toks"]), 0, 0])

    def check_device_flag(self, c):
        return c


def _emb(ids):
    return torch.nn.functional.one_hot(torch.tensor(ids), FakeEngine.V).float()


def _expected(ids, n):
    hist, out = list(ids), []
    for _ in range(n):
        out.append(FakeEngine._next(hist))
        hist.append(out[-1])
    return out


def _drain(bt, limit=10000):
    got, fin = {}, {}
    for _ in range(limit):
        if not bt.has_work():
            return got, fin
        for rid, new, finished, reason in bt.step():
            got.setdefault(rid, []).extend(new)
            if finished:
                fin[rid] = reason
    raise AssertionError("scheduler did not terminate")


@pytest.fixture(autouse=True)
def _no_cuda_sync(monkeypatch):
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: type("S", (), {"synchronize": lambda self: None})())


def test_fifo_admission_slots_eos_and_length():
    from vita_amd.serving import ContinuousBatcher
    eng = FakeEngine(pool_tokens=64 * 32, max_seqs=2)
    bt = ContinuousBatcher(eng, window=1)
    prompts = [[1, 2, 3], [4, 5, 6, 7, 8], [9], [10, 11]]
    exp = [_expected(p, 12) for p in prompts]
    eos2 = exp[2][5]
    for i, p in enumerate(prompts):
        bt.add(i, _emb(p), max_tokens=12 if i != 1 else 4, eos={eos2} if i == 2 else ())
    got, fin = _drain(bt)
    assert got[0] == exp[0] and got[1] == exp[1][:4] and got[3] == exp[3]
    assert got[2] == exp[2][:exp[2].index(eos2) + 1] and fin[2] == "stop" and fin[1] == "length"
    prefills = [c[1] for c in eng.calls if c[0] == "prefill"]
    assert len(prefills) == 4 and eng.free == 32 and not eng.seqs          # everything returned
    # never more than two sequences in one iteration, and requests 2 / 3 start only after a slot was freed
    assert max(len(c[1]) for c in eng.calls if c[0] == "decode") == 2
    first_decode_with = lambda n: next(i for i, c in enumerate(eng.calls) if c[0] == "decode" and len(c[1]) == n)
    assert [c[0] for c in eng.calls[:2]] == ["prefill", "prefill"] and first_decode_with(2) == 2


def test_abort_frees_pages_and_waiting_requests():
    from vita_amd.serving import ContinuousBatcher
    eng = FakeEngine(pool_tokens=64 * 8, max_seqs=1)
    bt = ContinuousBatcher(eng, window=2)
    bt.add("a", _emb([1, 2]), max_tokens=30)
    bt.add("b", _emb([3]), max_tokens=3)
    bt.step()
    assert bt.abort("a") and eng.free == 8                               # running request: pages back at once
    assert bt.abort("nope") is False
    got, fin = _drain(bt)
    assert got["b"] == _expected([3], 3) and "a" not in fin
    bt.add("c", _emb([5]), max_tokens=2)
    assert bt.abort("c") and not bt.has_work()                           # waiting request: simply dropped


def test_preemption_recomputes_and_continues_identically():
    from vita_amd.serving import ContinuousBatcher
    # 4 pages: two 60-token prompts take one each, both need a second page at 64 and a third at 128
    eng = FakeEngine(pool_tokens=64 * 4, max_seqs=2, max_new=100, max_prefill=512)
    table = torch.eye(FakeEngine.V)
    bt = ContinuousBatcher(eng, embed_tokens=lambda ids: table[ids], window=4)
    pa, pb = list(range(1, 61)), list(range(20, 82))
    bt.add("a", _emb(pa), max_tokens=90)
    bt.add("b", _emb(pb), max_tokens=90)
    got, fin = _drain(bt)
    assert bt.stats["preemptions"] >= 1
    assert got["a"] == _expected(pa, 90) and got["b"] == _expected(pb, 90)
    assert eng.free == 4 and fin == {"a": "length", "b": "length"}
    # the preempted request came back as ONE prefill of prompt + everything it had generated
    re = [c for c in eng.calls if c[0] == "prefill"]
    assert len(re) == 3 and re[2][2] > 62


def test_oversized_prompt_is_refused_up_front():
    from vita_amd.serving import ContinuousBatcher
    eng = FakeEngine(pool_tokens=64 * 2, max_seqs=2, max_prefill=100)
    bt = ContinuousBatcher(eng)
    with pytest.raises(ValueError):
        bt.add("x", _emb(list(range(1, 90)) * 2), max_tokens=4)          # longer than max_prefill
    bt.add("y", _emb(list(range(1, 97))), max_tokens=50)                 # 96 tokens: 2 pages, the whole pool
    got, fin = _drain(bt)
    # alone in a dry pool: it ends ("length") where the pool ends instead of raising
    assert fin["y"] == "length" and got["y"] == _expected(list(range(1, 97)), len(got["y"])) and 1 <= len(got["y"]) < 50


def test_one_failing_request_does_not_touch_the_others():
    """a prefill that raises ends THAT request with an "error" event; running and later requests go on."""
    from vita_amd.serving import ContinuousBatcher

    class Flaky(FakeEngine):
        def seq_prefill(self, s, emb):
            if int(emb.argmax(dim=1)[0]) == 13:                          # the poisoned prompt
                raise RuntimeError("prefill rejected")
            return super().seq_prefill(s, emb)

    eng = Flaky(pool_tokens=64 * 8, max_seqs=2)
    bt = ContinuousBatcher(eng, window=1)
    bt.add("ok1", _emb([1, 2, 3]), max_tokens=5)
    bt.add("bad", _emb([13, 2]), max_tokens=5)
    bt.add("ok2", _emb([4, 5]), max_tokens=5)
    got, fin = _drain(bt)
    assert fin == {"ok1": "length", "bad": "error", "ok2": "length"}
    assert got["ok1"] == _expected([1, 2, 3], 5) and got["ok2"] == _expected([4, 5], 5) and got["bad"] == []
    assert isinstance(bt.errors.pop("bad"), RuntimeError) and bt.stats["failed"] == 1
    assert eng.free == 8 and not eng.seqs                                # the failed request's slot and pages came back


def test_prompt_larger_than_the_pool_fails_alone():
    from vita_amd.serving import ContinuousBatcher
    eng = FakeEngine(pool_tokens=64 * 2, max_seqs=2, max_prefill=512)
    bt = ContinuousBatcher(eng, window=1)
    bt.add("small", _emb([1, 2]), max_tokens=3)
    got, fin = _drain(bt)
    eng.max_ctx = 64 * 8                                                  # (lets add() accept the prompt; the POOL stays 2 pages)
    bt.add("huge", _emb(list(range(1, 90)) * 3), max_tokens=3)           # 267 tokens: 5 pages
    bt.add("after", _emb([7]), max_tokens=3)
    g2, f2 = _drain(bt)
    assert f2 == {"huge": "error", "after": "length"} and g2["after"] == _expected([7], 3)
    assert "KV pages" in str(bt.errors["huge"])


def test_recompute_longer_than_max_prefill_goes_in_chunks():
    """a preempted sequence comes back as prompt + generated tokens; past max_prefill that is several appending prefills."""
    from vita_amd.serving import ContinuousBatcher
    eng = FakeEngine(pool_tokens=64 * 4, max_seqs=2, max_new=100, max_prefill=64)
    table = torch.eye(FakeEngine.V)
    bt = ContinuousBatcher(eng, embed_tokens=lambda ids: table[ids], window=4)
    pa, pb = list(range(1, 61)), list(range(20, 82))[:60]
    bt.add("a", _emb(pa), max_tokens=90)
    bt.add("b", _emb(pb), max_tokens=90)
    got, fin = _drain(bt)
    assert bt.stats["preemptions"] >= 1 and bt.stats["failed"] == 0
    assert got["a"] == _expected(pa, 90) and got["b"] == _expected(pb, 90)
    assert all(c[2] <= 64 for c in eng.calls if c[0] == "prefill")       # no call above max_prefill
    assert eng.free == 4 and fin == {"a": "length", "b": "length"}
