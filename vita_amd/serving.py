"""vLLM-shaped serving surface (SURVEY §8(b) flavour 2 / §8(f)#1) over the HIP engine.

What the reference's web demo does with vLLM (web_demo/web_ability_demo.py:132-243,340-351):

    llm = LLM(model=path, dtype="float16", tensor_parallel_size=2, limit_mm_per_prompt={...})
    sampling_params = SamplingParams(temperature=0.01, max_tokens=512, best_of=1, skip_special_tokens=False)
    out = llm.generate({"prompt_token_ids": ids, "multi_modal_data": {"image": [PIL...], "audio": [Tensor[T,80]...]}},
                       sampling_params=sampling_params)
    text = out[0].outputs[0].text

keeps working with `from vita_amd.serving import LLM, SamplingParams`.  The prompt carries ONE
`image_token_index` (51000) per image and ONE `audio_token_index` (51001) per clip; like the plugin's
input processor (web_demo/vllm_tools/vllm_file/mixtral.py:194-311) we tile each image with
dynamic_preprocess (+thumbnail), expand its placeholder to 256 tokens per tile and each audio placeholder to
`((T-1)//2-1)//2 -> (.-1)//2+1` tokens.  Audio features arrive CMVN-normalised by WhaleFeatureExtractor (the
vLLM audio tower has no global_cmvn: mixtral.py:1245-1247), so the encoder is run with `normalized=True`.
temperature <= 0.01 is greedy (the demo's setting); other sampling is not implemented.

Concurrent requests (the interactive demo's AsyncLLMEngine, web_interactive_demo.py:126,315-328,942-951): the engine's
KV cache is a pool of 64-token pages addressed through per-sequence block tables (vh_mixtral_seq_* in
include/vita_hip.h); `ContinuousBatcher` is the iteration-level scheduler over it (admit -> prefill, one decode step for
every running sequence per iteration, free on finish, preempt-by-recompute when the pool runs dry) and
`AsyncLLMEngine.generate(inputs, sampling_params, request_id)` is the async iterator the demo consumes.  `LLM.generate`
(the offline demo) keeps serving one request at a time on the same engine."""
import asyncio
import collections
import json
import os
import threading
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .host.constants import AUDIO_TOKEN_INDEX, IMAGE_TOKEN_INDEX
from .host.image_processing import dynamic_preprocess


@dataclass
class SamplingParams:
    temperature: float = 0.01
    max_tokens: int = 512
    best_of: int = 1
    skip_special_tokens: bool = False
    stop_token_ids: Optional[List[int]] = None
    top_p: float = 1.0
    top_k: int = -1


@dataclass
class CompletionOutput:
    index: int
    text: str
    token_ids: List[int]
    finish_reason: str = "length"


@dataclass
class RequestOutput:
    request_id: str
    prompt_token_ids: List[int]
    outputs: List[CompletionOutput] = field(default_factory=list)
    finished: bool = True
    metrics: dict = field(default_factory=dict)


def audio_feature_size(n_frames: int) -> int:
    """mixtral.py:283-287."""
    down = ((int(n_frames) - 1) // 2 - 1) // 2
    return (down - 1) // 2 + 1


def expand_placeholders(ids, images, audios, *, image_token_index, audio_token_index, image_size, min_dynamic_patch=1,
                        max_dynamic_patch=12, use_thumbnail=True, limit_mm=None):
    """The plugin's input processor (web_demo/vllm_tools/vllm_file/mixtral.py:194-295) in the HF-flavour vocabulary:
    every image placeholder becomes one IMAGE sentinel per TILE of dynamic_preprocess (+ thumbnail), every audio
    placeholder one AUDIO sentinel; the splice later widens a sentinel to 256 tokens per tile / audio_feature_size
    tokens per clip, which is the reference's repeat_and_pad_image_tokens (:100-190) with repeat counts
    256 * tiles and ((T-1)//2-1)//2 -> (.-1)//2+1 (:283-287).  Returns (sentinel ids, tiles)."""
    limit_mm = limit_mm or {}
    n_img, n_aud = ids.count(image_token_index), ids.count(audio_token_index)
    if n_img != len(images) or n_aud != len(audios):
        raise ValueError(f"prompt has {n_img} image / {n_aud} audio placeholders but multi_modal_data holds "
                         f"{len(images)} / {len(audios)}")                    # mixtral.py:136-140,1110-1124
    if len(images) > limit_mm.get("image", 256) or len(audios) > limit_mm.get("audio", 50):
        raise ValueError("limit_mm_per_prompt exceeded")
    tiles, out, ii = [], [], 0
    for t in ids:
        if t == image_token_index:
            ts, _ = dynamic_preprocess(images[ii], min_num=min_dynamic_patch, max_num=max_dynamic_patch,
                                       image_size=image_size, use_thumbnail=use_thumbnail)
            tiles += ts
            out += [IMAGE_TOKEN_INDEX] * len(ts)
            ii += 1
        elif t == audio_token_index:
            out.append(AUDIO_TOKEN_INDEX)
        else:
            out.append(int(t))
    return out, tiles


def expanded_token_ids(sentinel_ids, audio_frames, *, image_token_index, audio_token_index, tokens_per_tile=256):
    """The id sequence the reference's input processor hands to vLLM (`new_token_ids`, mixtral.py:175-190,289-295),
    rebuilt from the sentinel form: what the spliced embedding sequence corresponds to, position by position."""
    out, ai = [], 0
    for t in sentinel_ids:
        if t == IMAGE_TOKEN_INDEX:
            out += [image_token_index] * tokens_per_tile
        elif t == AUDIO_TOKEN_INDEX:
            out += [audio_token_index] * audio_feature_size(audio_frames[ai])
            ai += 1
        else:
            out.append(int(t))
    return out


class LLM:
    def __init__(self, model, dtype=None, tensor_parallel_size=1, trust_remote_code=True, gpu_memory_utilization=None,
                 disable_custom_all_reduce=True, limit_mm_per_prompt=None, max_new_tokens=1024, device="cuda",
                 max_num_seqs=0, kv_pool_tokens=None, **_):
        from .model.builder import load_pretrained_model
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if tensor_parallel_size != 1 and tensor_parallel_size != world:
            raise RuntimeError(
                f"tensor_parallel_size={tensor_parallel_size}: vita_amd runs one process per GPU — launch this program "
                f"with `python -m torch.distributed.run --nproc-per-node {tensor_parallel_size} ...` (WORLD_SIZE={world})")
        self.limit_mm = dict(limit_mm_per_prompt or {"image": 256, "audio": 50})
        kw = {}
        if world > 1:
            kw.update(rank=int(os.environ.get("RANK", "0")), world=world)
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
        if world > 1:
            torch.cuda.set_device(torch.device(device))       # kernels launch on the CURRENT device's stream
        self.tokenizer, self.model, self.image_processor, _ = load_pretrained_model(
            model, None, os.path.basename(str(model).rstrip("/")), "mixtral-8x7b", device=device,
            max_new_tokens=max_new_tokens, max_seqs=int(max_num_seqs), kv_pool_tokens=kv_pool_tokens,
            gpu_memory_utilization=gpu_memory_utilization, **kw)
        self.collective = "none"
        if world > 1:
            from .parallel import setup_tensor_parallel
            self.collective = setup_tensor_parallel(self.model.engine, kw["rank"], world, device,
                                                    backend=os.environ.get("VITA_AMD_DIST_BACKEND", "nccl"))
        with open(os.path.join(model, "config.json")) as f:
            j = json.load(f)
        self.image_token_index = int(j.get("image_token_index", 51000))
        self.audio_token_index = int(j.get("audio_token_index", 51001))
        self.min_dynamic_patch = int(j.get("min_dynamic_patch", 1))
        self.max_dynamic_patch = int(j.get("max_dynamic_patch", 12))
        self.use_thumbnail = bool(j.get("use_thumbnail", True))
        self._n = 0

    def get_tokenizer(self):
        return self.tokenizer

    # ---- one request ----------------------------------------------------------------------------
    def _expand(self, ids, images, audios):
        return expand_placeholders(ids, images, audios, image_token_index=self.image_token_index,
                                   audio_token_index=self.audio_token_index,
                                   image_size=self.image_processor.crop_size["height"],
                                   min_dynamic_patch=self.min_dynamic_patch, max_dynamic_patch=self.max_dynamic_patch,
                                   use_thumbnail=self.use_thumbnail, limit_mm=self.limit_mm)

    @torch.no_grad()
    def _prepare(self, inp, sp: SamplingParams):
        """host side of one request: ids, placeholder expansion, pixel / fbank tensors (what vLLM's input processor and
        mapper do: mixtral.py:194-311).  Returns dict(ids, sent, pix, feats, lens, normalized, max_tokens, eos)."""
        if isinstance(inp, str):
            inp = {"prompt": inp}
        ids = inp.get("prompt_token_ids")
        if ids is None:
            ids = list(self.tokenizer(inp["prompt"]).input_ids)
        ids = [int(x) for x in (ids.tolist() if hasattr(ids, "tolist") else ids)]
        mm = inp.get("multi_modal_data") or {}
        images = mm.get("image", [])
        images = images if isinstance(images, list) else [images]
        audios = mm.get("audio", [])
        audios = audios if isinstance(audios, list) else [audios]
        if sp.temperature is not None and sp.temperature > 0.011:
            raise NotImplementedError("vita_amd serves greedy decoding (temperature <= 0.01, the reference demo's setting)")
        dev = self.model.device
        max_tokens = min(int(sp.max_tokens), int(self.model.max_new_tokens))   # the engine's output buffer is the hard cap
        sent, tiles = self._expand(ids, images, audios)
        size = self.image_processor.crop_size["height"]
        if tiles:
            pix = self.image_processor.preprocess(tiles, return_tensors="pt")["pixel_values"].to(dev)
        else:
            pix = torch.zeros((1, 3, size, size), device=dev)                  # the demo's dummy image
        enc = self.model.get_audio_encoder()
        normalized = False
        if audios:
            if len(audios) > 1:
                T = max(int(a.shape[0]) for a in audios)
                feats = torch.zeros((len(audios), T, audios[0].shape[1]))
                for i, a in enumerate(audios):
                    feats[i, :a.shape[0]] = a.float()
            else:
                feats = audios[0].float()[None]
            lens = torch.tensor([int(a.shape[0]) for a in audios])
            if len({int(a.shape[0]) for a in audios}) > 1:
                raise NotImplementedError("clips of different lengths in one request are not batched yet")
            normalized = True                # WhaleFeatureExtractor already applied CMVN
        else:
            feats, lens = torch.zeros((1, 400, enc.acfg.input_dim)), torch.tensor([400])
        eos = {self.model.generation_config.eos_token_id, *(sp.stop_token_ids or [])} - {None}
        return dict(ids=ids, sent=sent, pix=pix, feats=feats.to(dev), lens=lens.to(dev), normalized=normalized,
                    max_tokens=max_tokens, eos=eos)

    @torch.no_grad()
    def _embed(self, req):
        """encoders + splice of a prepared request -> inputs_embeds [S, H] fp32 on the device."""
        enc = self.model.get_audio_encoder()
        enc.normalized_input = req["normalized"]
        try:
            dev = self.model.device
            out = self.model.prepare_inputs_labels_for_multimodal(
                torch.tensor([req["sent"]], dtype=torch.long, device=dev), None, None, None, None, req["pix"],
                {"audios": req["feats"], "lengths": req["lens"]})
        finally:
            enc.normalized_input = False
        return out[4][0].to(torch.float32)

    @torch.no_grad()
    def _one(self, inp, sp: SamplingParams, streamer=None):
        req = self._prepare(inp, sp)
        ids, sent, pix, feats, lens, max_tokens = (req[k] for k in ("ids", "sent", "pix", "feats", "lens", "max_tokens"))
        dev = self.model.device
        enc = self.model.get_audio_encoder()
        enc.normalized_input = req["normalized"]
        try:
            out = self.model.generate(torch.tensor([sent], dtype=torch.long, device=dev), images=pix,
                                      audios={"audios": feats.to(dev), "lengths": lens.to(dev)}, do_sample=False,
                                      num_beams=1, return_dict_in_generate=True, max_new_tokens=max_tokens,
                                      eos_token_id=list(req["eos"]), streamer=streamer)
        finally:
            enc.normalized_input = False
        gen = out.sequences[0, len(sent):].tolist()
        eos = req["eos"]
        reason = "stop" if gen and gen[-1] in eos else "length"
        text = self.tokenizer.decode(gen, skip_special_tokens=sp.skip_special_tokens)
        self._n += 1
        return RequestOutput(request_id=str(self._n - 1), prompt_token_ids=ids,
                             outputs=[CompletionOutput(0, text, gen, reason)], metrics=dict(self.model.last_timing))

    def generate_stream(self, inputs, sampling_params: SamplingParams = None, request_id=None, should_stop=None):
        """Generator of RequestOutput with the CUMULATIVE text after every decode window — the shape of
        AsyncLLMEngine.generate's async iterator as web_interactive_demo.py:315-328 consumes it.  `should_stop()`
        is polled between windows; returning True interrupts the request (the duplex monitor hand-off)."""
        import queue
        import threading
        sp = sampling_params or SamplingParams()
        q = queue.Queue()
        state = {"toks": [], "abandoned": False}

        def streamer(new):
            state["toks"] += list(new)
            q.put(list(state["toks"]))
            # abandoned: the consumer left the iterator (noise verdict, interrupt): stop at the next window instead of
            # decoding to EOS / max_tokens while the worker waits in th.join() (the reference aborts the request)
            return not (state["abandoned"] or (should_stop is not None and should_stop()))

        result = {}

        def run():
            try:
                result["out"] = self._one(inputs, sp, streamer=streamer)
            except BaseException as e:  # surfaced to the consumer
                result["err"] = e
            q.put(None)

        old_la = self.model.lookahead
        self.model.lookahead = 2                       # short windows: the interrupt is seen within 2 tokens
        th = threading.Thread(target=run, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                yield RequestOutput(request_id=str(request_id), prompt_token_ids=[], finished=False,
                                    outputs=[CompletionOutput(0, self.tokenizer.decode(
                                        item, skip_special_tokens=sp.skip_special_tokens), item, "")])
        finally:
            state["abandoned"] = True
            th.join()
            self.model.lookahead = old_la
        if "err" in result:
            raise result["err"]
        yield result["out"]

    def generate(self, prompts, sampling_params: SamplingParams = None, use_tqdm=False, **_):
        sp = sampling_params or SamplingParams()
        batch = prompts if isinstance(prompts, list) else [prompts]
        return [self._one(p, sp) for p in batch]


# ---- concurrent requests: iteration-level scheduling over the paged KV cache -------------------------------------------
class ContinuousBatcher:
    """The scheduler half of vLLM's engine for this path (what drives the plugin's forward with `kv_caches` +
    `attn_metadata`, vllm_file/mixtral.py:1130-1186), over `MixtralEngine.seq_*`:

      add(request_id, embeds, max_tokens, eos)   queue a request (FIFO)
      step() -> [(request_id, new_token_ids, finished, finish_reason)]   one iteration:
          1. admit waiting requests while a sequence slot is free and the pool holds the prompt's pages plus one page of
             headroom per running sequence (prefill is one C call per admitted request);
          2. if the running sequences need more new pages than the pool has, preempt the youngest ones (free their pages,
             re-queue them at the front with prompt + generated tokens as the new prompt: vLLM's recompute preemption);
          3. `window` greedy decode steps for every running sequence (one C call per step), ONE host synchronisation,
             then eos / max_tokens per sequence; finished sequences return their pages.
      abort(request_id)

    A sequence's ids do not depend on what else is scheduled (same kernels, same arithmetic per sequence); tokens decoded
    past an eos inside a window are discarded, as MixtralEngine's single-request lookahead does."""

    def __init__(self, engine, embed_tokens=None, max_batch=None, window=1):
        self.eng, self.embed_tokens = engine, embed_tokens
        self.max_batch = int(max_batch or engine.c.max_seqs)
        if engine.c.max_seqs < 1:
            raise ValueError("the engine was created without sequence slots (max_seqs = 0)")
        self.window = int(window)
        self.waiting = collections.deque()
        self.running = []                 # admission order
        self.errors = {}                  # request id -> exception of a request that failed alone (event reason "error")
        self.stats = {"iterations": 0, "prefills": 0, "preemptions": 0, "decode_steps": 0, "failed": 0}

    def add(self, request_id, embeds, max_tokens, eos=()):
        S = int(embeds.shape[0])
        if S + 2 > self.eng.max_ctx or S > self.eng.max_prefill:
            raise ValueError(f"prompt of {S} tokens exceeds the engine (max_prefill {self.eng.max_prefill}, pool "
                             f"{self.eng.max_ctx} tokens)")
        self.waiting.append(dict(id=request_id, emb=embeds, max_tokens=min(int(max_tokens), self.eng.max_new),
                                 eos=set(eos), out=[], seq=None, base=0))

    def abort(self, request_id):
        for r in list(self.running):
            if r["id"] == request_id:
                self.eng.seq_free(r["seq"])
                self.running.remove(r)
                return True
        for r in list(self.waiting):
            if r["id"] == request_id:
                self.waiting.remove(r)
                return True
        return False

    def has_work(self):
        return bool(self.waiting or self.running)

    def _pages(self, n_tokens):
        return -(-int(n_tokens) // 64)

    def _prefill(self, r):
        """the prompt of a request — or, after a preemption, prompt + generated tokens, which may exceed what one prefill call
        takes — in chunks of max_prefill (seq_prefill appends at the sequence's position)."""
        emb, mp = r["emb"], int(self.eng.max_prefill)
        for s0 in range(0, int(emb.shape[0]), mp):
            self.eng.seq_prefill(r["seq"], emb[s0:s0 + mp])

    def _fail(self, r, exc, events):
        """ONE request failed (its prompt does not fit, its prefill was rejected): it ends with an "error" event, nobody else
        is touched.  Device-level failures (spin time-outs, launch errors seen at the synchronisation) still propagate."""
        self.errors[r["id"]] = exc
        self.stats["failed"] += 1
        events.append((r["id"], [], True, "error"))

    def _admit(self):
        events = []
        while self.waiting and len(self.running) < self.max_batch:
            r = self.waiting[0]
            need = self._pages(r["emb"].shape[0] + 1) + len(self.running)       # + one page of headroom per runner
            if need > self.eng.pages_free():
                if not self.running:
                    self.waiting.popleft()
                    self._fail(r, RuntimeError(f"request {r['id']}: prompt needs {need} KV pages, the pool has "
                                               f"{self.eng.pages_free()}"), events)
                    continue
                break
            self.waiting.popleft()
            try:
                r["seq"] = self.eng.seq_alloc()
            except Exception as e:
                self._fail(r, e, events)
                continue
            try:
                self._prefill(r)
            except Exception as e:
                self.eng.seq_free(r["seq"])
                r["seq"] = None
                self._fail(r, e, events)
                continue
            r["fresh"] = True                    # its first token (tokens[0]) is not reported yet
            self.running.append(r)
            self.stats["prefills"] += 1
        return events

    def _preempt_for(self, steps):
        """make sure `steps` decode steps of every running sequence find their pages; youngest sequences yield."""
        while True:
            new_pages = sum(self._pages(self.eng.seq_pos(r["seq"]) + steps) - len(self.eng.seq_pages(r["seq"]))
                            for r in self.running)
            if new_pages <= self.eng.pages_free() or len(self.running) <= 1:
                return
            r = self.running.pop()
            self.eng.seq_free(r["seq"])
            if r["out"] and self.embed_tokens is None:
                raise RuntimeError("KV pool exhausted and no embed_tokens hook to recompute a preempted sequence")
            if r["out"][r["base"]:]:
                tail = self.embed_tokens(torch.tensor(r["out"][r["base"]:], dtype=torch.long, device=r["emb"].device))
                r["emb"] = torch.cat([r["emb"], tail.to(torch.float32)], dim=0)
            r["base"], r["seq"] = len(r["out"]), None
            self.waiting.appendleft(r)
            self.stats["preemptions"] += 1

    @torch.no_grad()
    def step(self):
        failed = self._admit()
        if not self.running:
            return failed
        # decode window, bounded by every sequence's token buffer and by the pool's addressable context
        steps = self.window
        for r in self.running:
            n_dev = len(r["out"]) - r["base"] + (1 if r.get("fresh") else 0)
            steps = min(steps, self.eng.max_new - n_dev, self.eng.max_ctx - 2 - self.eng.seq_pos(r["seq"]))
        steps = max(steps, 0)
        starved = None
        if steps > 0:
            self._preempt_for(steps)
            if len(self.running) == 1:
                r = self.running[0]
                if self._pages(self.eng.seq_pos(r["seq"]) + steps) - len(self.eng.seq_pages(r["seq"])) > self.eng.pages_free():
                    starved, steps = r, 0                         # alone and the pool is dry: it ends here ("length")
        if steps > 0:
            ids = [r["seq"] for r in self.running]
            for _ in range(steps):
                self.eng.seq_decode(ids)
            self.stats["decode_steps"] += steps * len(ids)
        self.stats["iterations"] += 1
        torch.cuda.current_stream().synchronize()
        events = failed
        for r in list(self.running):
            cnt = self.eng.check_device_flag(self.eng.seq_counters(r["seq"]).tolist())
            n_dev = min(cnt[1], self.eng.max_new)                 # tokens of this (re)prefill + its decode steps
            have = len(r["out"]) - r["base"]
            new = self.eng.seq_tokens(r["seq"])[have:n_dev].tolist()
            r["fresh"] = False
            accepted, reason = [], None
            for tok in new:
                accepted.append(tok)
                if tok in r["eos"]:
                    reason = "stop"
                elif len(r["out"]) + len(accepted) >= r["max_tokens"]:
                    reason = "length"
                if reason:
                    break
            r["out"] += accepted
            if reason is None and (r is starved or self.eng.seq_pos(r["seq"]) + 2 >= self.eng.max_ctx):
                reason = "length"                                 # the pool cannot hold another token of this sequence
            if reason:
                self.eng.seq_free(r["seq"])
                self.running.remove(r)
            if accepted or reason:
                events.append((r["id"], accepted, reason is not None, reason or ""))
        return events


@dataclass
class AsyncEngineArgs:
    """the fields web_interactive_demo.py:942-951 sets, plus the scheduler's knobs (vLLM names)."""
    model: str = ""
    dtype: str = "float16"
    tensor_parallel_size: int = 1
    trust_remote_code: bool = True
    gpu_memory_utilization: float = 0.8
    disable_custom_all_reduce: bool = True
    limit_mm_per_prompt: Optional[dict] = None
    max_num_seqs: int = 8
    max_new_tokens: int = 1024
    kv_pool_tokens: Optional[int] = None     # size of the paged KV pool; default: max_num_seqs x (<= 1024-token prompt + max_new_tokens),
                                             # bounded by gpu_memory_utilization of the free memory (VITAMixtralForCausalLM.default_kv_pool_tokens)
    device: str = "cuda"


class AsyncLLMEngine:
    """`vllm.AsyncLLMEngine` as the interactive demo uses it (web_interactive_demo.py:126, 315-328):

        llm = AsyncLLMEngine.from_engine_args(AsyncEngineArgs(model=path, ...))
        async for request_output in llm.generate(inputs, sampling_params=sp, request_id=uuid):
            text = request_output.outputs[0].text          # cumulative

    Requests submitted while others are running are batched by `ContinuousBatcher`: one scheduler thread owns the GPU
    stream (encoders + prefill of admitted requests, then one decode step for every running sequence per iteration) and
    posts RequestOutputs to the callers' event loops."""

    def __init__(self, llm: "LLM", max_num_seqs=None, window=1):
        self.llm = llm
        eng = llm.model.engine
        self.batcher = ContinuousBatcher(eng, embed_tokens=lambda ids: llm.model.model.embed_tokens(ids),
                                         max_batch=max_num_seqs, window=window)
        self._lock = threading.Lock()
        self._cv = threading.Condition(self._lock)
        self._incoming, self._aborts, self._sinks = [], [], {}
        self._closed = False
        self._thread = threading.Thread(target=self._loop, daemon=True, name="vita-amd-scheduler")
        self._thread.start()

    @classmethod
    def from_engine_args(cls, args: AsyncEngineArgs, **kw):
        pool = args.kv_pool_tokens
        llm = LLM(args.model, dtype=args.dtype, tensor_parallel_size=args.tensor_parallel_size,
                  gpu_memory_utilization=args.gpu_memory_utilization,
                  limit_mm_per_prompt=args.limit_mm_per_prompt, max_new_tokens=args.max_new_tokens, device=args.device,
                  max_num_seqs=args.max_num_seqs, kv_pool_tokens=pool)
        return cls(llm, max_num_seqs=args.max_num_seqs, **kw)

    async def get_tokenizer(self):
        return self.llm.tokenizer

    # -- scheduler thread ----------------------------------------------------------------------------------------------
    def _post(self, rid, item):
        sink = self._sinks.get(rid)
        if sink is not None:
            loop, q = sink
            loop.call_soon_threadsafe(q.put_nowait, item)

    def _loop(self):
        if torch.cuda.is_available():
            torch.cuda.set_device(self.llm.model.device)      # kernels launch on the CURRENT device's stream
        state = {}                                   # request id -> (prompt ids, sampling params, token list)
        while True:
            with self._cv:
                while not (self._incoming or self._aborts or self.batcher.has_work() or self._closed):
                    self._cv.wait()
                if self._closed:
                    return
                incoming, self._incoming = self._incoming, []
                aborts, self._aborts = self._aborts, []
            for rid in aborts:
                self.batcher.abort(rid)
                state.pop(rid, None)
            for rid, inputs, sp in incoming:
                try:
                    req = self.llm._prepare(inputs, sp)
                    emb = self.llm._embed(req)
                    self.batcher.add(rid, emb, req["max_tokens"], req["eos"])
                    state[rid] = (req["ids"], sp, [])
                except BaseException as e:          # surfaced to the caller's iterator
                    self._post(rid, e)
            try:
                events = self.batcher.step()
            except BaseException as e:
                for rid in list(state):
                    self._post(rid, e)
                state.clear()
                self.batcher.waiting.clear()
                for r in list(self.batcher.running):
                    self.batcher.abort(r["id"])
                continue
            for rid, new, finished, reason in events:
                if rid not in state:
                    continue
                if reason == "error":                # this request alone failed (ContinuousBatcher._fail)
                    self._post(rid, self.batcher.errors.pop(rid, RuntimeError(f"request {rid} failed")))
                    state.pop(rid)
                    continue
                ids, sp, toks = state[rid]
                toks += new
                text = self.llm.tokenizer.decode(toks, skip_special_tokens=sp.skip_special_tokens)
                self._post(rid, RequestOutput(request_id=str(rid), prompt_token_ids=ids, finished=finished,
                                              outputs=[CompletionOutput(0, text, list(toks), reason)]))
                if finished:
                    state.pop(rid)

    # -- caller side -----------------------------------------------------------------------------------------------------
    async def generate(self, inputs, sampling_params: SamplingParams = None, request_id=None, **_):
        sp = sampling_params or SamplingParams()
        rid = request_id if request_id is not None else os.urandom(8).hex()
        q = asyncio.Queue()
        with self._cv:
            if rid in self._sinks:
                raise ValueError(f"request id {rid!r} is already running")
            self._sinks[rid] = (asyncio.get_running_loop(), q)
            self._incoming.append((rid, inputs, sp))
            self._cv.notify()
        try:
            while True:
                item = await q.get()
                if isinstance(item, BaseException):
                    raise item
                yield item
                if item.finished:
                    return
        finally:
            with self._cv:
                self._sinks.pop(rid, None)
                self._aborts.append(rid)          # a consumer that leaves early (interrupt, noise verdict) frees the pages
                self._cv.notify()

    async def abort(self, request_id):
        with self._cv:
            self._aborts.append(request_id)
            self._cv.notify()

    def shutdown(self):
        with self._cv:
            self._closed = True
            self._cv.notify()
        self._thread.join(timeout=30)
