"""Duplex two-engine serving (BASELINE configs[4]) — the orchestration of the reference's
web_demo/web_interactive_demo.py:105-379,911-1029, without its Gradio / VAD / TTS front ends.

Two identical engines (replicas: no tensor traffic between them) alternate between two roles:

  * whoever holds the `start` baton takes the next request from the shared input queue and immediately hands
    the baton to the other engine (web_interactive_demo.py:284-293) — so a new query that arrives while this
    engine is still speaking is picked up by the other one ("monitor");
  * an engine streams its answer; if the first generated text starts with the state token `<2>` the query is
    classed as noise and dropped silently (:221-223,368-370); otherwise, on its FIRST positive chunk it clears its
    own stop flag, raises the OTHER engine's stop flag and clears the output queue (:340-352) — the monitor
    interrupts the generator;
  * an engine polls its own stop flag between decode windows and abandons its answer when it is set (:354-366);
  * finished sentences (split on punctuation) go to the output queue, the finished turn to the shared history.

`worker_loop` is the per-engine process body; it only needs an object with
`generate_stream(inputs, sampling_params, request_id, should_stop) -> iterator of RequestOutput` (vita_amd.serving.LLM
has it), so the protocol is unit-tested on the CPU with scripted engines and on the GPU with two real engines.
`DuplexServer` wires two worker processes with multiprocessing primitives, one process per engine."""
import multiprocessing as mp
import queue as _queue
import time
import uuid

NEGATIVE_PREFIX = "<2>"
SENTENCE_END = [",", "，", ".", "。", "?", "\n", "？", "!", "！", "、"]


def judge_negative(text):
    return text.startswith(NEGATIVE_PREFIX)          # web_interactive_demo.py:221-223


def clear_queue(q):
    while True:
        try:
            q.get_nowait()
        except _queue.Empty:
            return


def worker_loop(llm_id, make_llm, sampling_params, inputs_queue, outputs_queue, stop_event, other_stop_event,
                worker_ready, wait_workers_ready, start_event, other_start_event, start_event_lock, interrupt_signal,
                global_history, shutdown_event, stats_queue=None, history_limit=0, poll_s=0.005):
    """Body of one engine process (web_interactive_demo.py:105-379)."""
    llm = make_llm()
    worker_ready.set()
    while not shutdown_event.is_set():
        if not all(w.is_set() for w in wait_workers_ready):
            time.sleep(poll_s)
            continue
        if inputs_queue.empty():
            time.sleep(poll_s)
            continue
        with start_event_lock:
            if not start_event.is_set():
                continue
            try:
                inputs = inputs_queue.get_nowait()
            except _queue.Empty:
                continue
            other_start_event.set()                   # hand the baton over before starting to work
            start_event.clear()
        t_take = time.perf_counter()
        current = dict(inputs)
        results, pending, first_positive, t_first = [], "", True, None
        previous, n_tokens = "", 0
        st = {"speaking": False}     # the stop flag only counts once this engine has taken the floor (:354)
        error = None
        stream = llm.generate_stream(inputs, sampling_params, request_id=uuid.uuid4().hex,
                                     should_stop=lambda: st["speaking"] and stop_event.is_set())
        try:
            for out in stream:
                text = out.outputs[0].text
                n_tokens = len(out.outputs[0].token_ids)
                new = text[len(previous):]
                previous = text
                if new == "":
                    continue
                if t_first is None:
                    t_first = time.perf_counter()
                if judge_negative(new) or (first_positive and judge_negative(text)):
                    break                                   # noise: answer nothing
                pending += new
                if first_positive:                          # first real words: take the floor
                    stop_event.clear()
                    other_stop_event.set()
                    clear_queue(outputs_queue)
                    first_positive = False
                    st["speaking"] = True
                    interrupt_signal.value = llm_id
                if stop_event.is_set():
                    break                                   # the other engine took the floor
                results.append(new)
                pending = pending.replace("<1> ", "").replace("<1>", "")
                if new in SENTENCE_END or new[-1:] in SENTENCE_END:
                    outputs_queue.put({"id": llm_id, "response": pending})
                    pending = ""
        except Exception as e:      # one bad request must not kill the engine process: the baton would never return
            error = f"{type(e).__name__}: {e}"
            print(f"[duplex worker {llm_id}] request failed: {error}", flush=True)
        finally:
            stream.close()            # abandons the generation (noise verdict / interrupt / error) without decoding on
        if pending and not stop_event.is_set() and not first_positive:
            outputs_queue.put({"id": llm_id, "response": pending})
        current["response"] = "".join(results)
        if current["response"]:
            global_history.append({k: v for k, v in current.items() if k in ("prompt", "response")})
            if history_limit and len(global_history) > history_limit:
                del global_history[0]
        if stats_queue is not None:
            t_end = time.perf_counter()
            stats_queue.put({"id": llm_id, "request": current.get("request_id"),
                             "take_to_first_chunk_s": None if t_first is None else t_first - t_take,
                             "first_chunk_to_end_s": None if t_first is None else t_end - t_first, "n_tokens": n_tokens,
                             "interrupted": bool(stop_event.is_set()), "negative": first_positive,
                             "n_chunks": len(results), "error": error})


def _engine_process(llm_id, llm_factory, factory_args, sampling_params, shared, engines_share_device=True):
    if engines_share_device:
        # two replicas on ONE GPU: launches whose blocks wait for other blocks (the fused decode attention block) can be starved by the
        # other replica's waiting blocks — the library keeps the three-launch form for processes that share a device
        from . import _lib
        _lib.tune("comm_ranks_per_device", 2)
    worker_loop(llm_id, lambda: llm_factory(*factory_args), sampling_params, **shared)


class DuplexServer:
    """Two engine processes + the shared queues / events of web_interactive_demo.py:914-1029."""

    def __init__(self, llm_factory, factory_args=(), sampling_params=None, history_limit=0, ctx="spawn", engines_share_device=True):
        """engines_share_device: the two replicas drive the same GPU (the default assumption; pass False when the factory places
        them on different devices, as the reference's demo does: cuda_devices "0,1" / "2,3", web_interactive_demo.py:958,981)."""
        self.mpc = mp.get_context(ctx)
        self.mgr = self.mpc.Manager()
        m = self.mgr
        self.inputs, self.outputs, self.stats = m.Queue(), m.Queue(), m.Queue()
        self.history = m.list()
        self.shutdown = m.Event()
        self.lock = m.Lock()
        self.interrupt = m.Value("i", -1)
        ready = [m.Event(), m.Event()]
        stop = [m.Event(), m.Event()]
        start = [m.Event(), m.Event()]
        start[0].set()                                  # engine 0 holds the baton first (:1010)
        self.procs = []
        for i in range(2):
            shared = dict(inputs_queue=self.inputs, outputs_queue=self.outputs, stop_event=stop[i],
                          other_stop_event=stop[1 - i], worker_ready=ready[i], wait_workers_ready=ready,
                          start_event=start[i], other_start_event=start[1 - i], start_event_lock=self.lock,
                          interrupt_signal=self.interrupt, global_history=self.history,
                          shutdown_event=self.shutdown, stats_queue=self.stats, history_limit=history_limit)
            p = self.mpc.Process(target=_engine_process, args=(i, llm_factory, factory_args, sampling_params, shared, engines_share_device),
                                 daemon=True)
            p.start()
            self.procs.append(p)
        # keep the parent's proxies alive: a Manager drops an object whose last proxy is collected, and the
        # spawned children attach seconds later
        self._ready, self._stop, self._start = ready, stop, start

    def wait_ready(self, timeout=600):
        t0 = time.time()
        while not all(e.is_set() for e in self._ready):
            if time.time() - t0 > timeout or any(not p.is_alive() for p in self.procs):
                raise RuntimeError("duplex engines failed to start")
            time.sleep(0.05)

    def submit(self, inputs):
        self.inputs.put(dict(inputs))

    def close(self):
        self.shutdown.set()
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
        self.mgr.shutdown()


def make_serving_llm(model_path, max_new_tokens=128):
    """Default engine factory of a worker process: one vita_amd.serving.LLM replica on the visible GPU."""
    from .serving import LLM
    return LLM(model=model_path, tensor_parallel_size=1, max_new_tokens=max_new_tokens)
