"""Duplex baton / interrupt protocol (web_demo/web_interactive_demo.py:284-366) on scripted engines — CPU only.
Each scripted engine 'generates' one chunk every few ms; the test checks who takes which request, that a positive
answer from the monitor interrupts the speaker, and that `<2>` (noise) answers are dropped without interrupting."""
import threading
import time
from types import SimpleNamespace

import pytest

from vita_amd import duplex


class ScriptedLLM:
    """generate_stream yields cumulative text like vita_amd.serving.LLM.generate_stream."""

    def __init__(self, delay=0.01):
        self.delay = delay

    def generate_stream(self, inputs, sampling_params=None, request_id=None, should_stop=None):
        text = ""
        for chunk in inputs["script"]:
            time.sleep(self.delay)
            text += chunk
            yield SimpleNamespace(outputs=[SimpleNamespace(text=text, token_ids=[])])
            if should_stop is not None and should_stop():
                return


class _Val:
    def __init__(self):
        self.value = -1


def _setup(n=2):
    import queue
    ev = threading.Event
    sh = SimpleNamespace(inputs=queue.Queue(), outputs=queue.Queue(), stats=queue.Queue(), history=[],
                         shutdown=ev(), lock=threading.Lock(), interrupt=_Val(),
                         ready=[ev(), ev()], stop=[ev(), ev()], start=[ev(), ev()])
    sh.start[0].set()
    th = []
    for i in range(n):
        kw = dict(inputs_queue=sh.inputs, outputs_queue=sh.outputs, stop_event=sh.stop[i],
                  other_stop_event=sh.stop[1 - i], worker_ready=sh.ready[i], wait_workers_ready=sh.ready,
                  start_event=sh.start[i], other_start_event=sh.start[1 - i], start_event_lock=sh.lock,
                  interrupt_signal=sh.interrupt, global_history=sh.history, shutdown_event=sh.shutdown,
                  stats_queue=sh.stats, poll_s=0.001)
        t = threading.Thread(target=duplex.worker_loop, args=(i, ScriptedLLM, None), kwargs=kw, daemon=True)
        t.start()
        th.append(t)
    return sh, th


def _drain(q, n, timeout=5.0):
    out, t0 = [], time.time()
    while len(out) < n and time.time() - t0 < timeout:
        try:
            out.append(q.get(timeout=0.05))
        except Exception:
            pass
    return out


@pytest.mark.timeout(60)
def test_baton_alternates_and_monitor_interrupts():
    sh, th = _setup()
    try:
        long_answer = ["<1> ", "one", ",", " two", ",", " three", ",", " four", ",", " five", ".", " six", ".", " seven", "."]
        sh.inputs.put({"prompt": "q0", "request_id": 0, "script": long_answer})
        time.sleep(0.05)                                  # engine 0 is speaking; engine 1 holds the baton
        sh.inputs.put({"prompt": "q1", "request_id": 1, "script": ["<1> ", "stop", "."]})
        stats = sorted(_drain(sh.stats, 2), key=lambda s: s["request"])
        assert [s["id"] for s in stats] == [0, 1]          # the baton alternated
        assert stats[0]["interrupted"] and stats[0]["n_chunks"] < len(long_answer)   # speaker was cut off
        assert not stats[1]["interrupted"] and not stats[1]["negative"]
        assert sh.interrupt.value == 1
        texts = [o for o in _drain(sh.outputs, 1, timeout=1.0)]
        assert texts and texts[-1]["id"] == 1 and "stop" in texts[-1]["response"]   # queue was cleared for the new speaker
        assert [h["prompt"] for h in sh.history] == ["q0", "q1"]
    finally:
        sh.shutdown.set()
        [t.join(timeout=2) for t in th]


@pytest.mark.timeout(60)
def test_noise_query_is_dropped_without_interrupting():
    sh, th = _setup()
    try:
        sh.inputs.put({"prompt": "q0", "request_id": 0, "script": ["<1> ", "a", ",", " b", ",", " c", ",", " d", "."]})
        time.sleep(0.03)
        sh.inputs.put({"prompt": "noise", "request_id": 1, "script": ["<2>", " ignored", "."]})
        stats = sorted(_drain(sh.stats, 2), key=lambda s: s["request"])
        assert stats[1]["negative"] and stats[1]["n_chunks"] == 0 and stats[1]["id"] == 1
        assert not stats[0]["interrupted"] and stats[0]["n_chunks"] == 9          # speaker finished undisturbed
        assert [h["prompt"] for h in sh.history] == ["q0"]                         # noise leaves no history
        out = _drain(sh.outputs, 4, timeout=1.0)
        assert all(o["id"] == 0 for o in out) and "".join(o["response"] for o in out).replace(" ", "") == "a,b,c,d."
    finally:
        sh.shutdown.set()
        [t.join(timeout=2) for t in th]
