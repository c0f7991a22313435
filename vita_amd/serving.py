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

Not here (out of scope for this path): vLLM's paged KV cache / continuous batching scheduler — requests
are served one at a time, as the offline demo does."""
import json
import os
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
                 disable_custom_all_reduce=True, limit_mm_per_prompt=None, max_new_tokens=1024, device="cuda", **_):
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
            max_new_tokens=max_new_tokens, **kw)
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
    def _one(self, inp, sp: SamplingParams, streamer=None):
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
            enc.normalized_input = True      # WhaleFeatureExtractor already applied CMVN
        else:
            feats, lens = torch.zeros((1, 400, enc.acfg.input_dim)), torch.tensor([400])
            enc.normalized_input = False
        try:
            out = self.model.generate(torch.tensor([sent], dtype=torch.long, device=dev), images=pix,
                                      audios={"audios": feats.to(dev), "lengths": lens.to(dev)}, do_sample=False,
                                      num_beams=1, return_dict_in_generate=True, max_new_tokens=max_tokens,
                                      eos_token_id=list({self.model.generation_config.eos_token_id,
                                                         *(sp.stop_token_ids or [])}), streamer=streamer)
        finally:
            enc.normalized_input = False
        gen = out.sequences[0, len(sent):].tolist()
        eos = {self.model.generation_config.eos_token_id, *(sp.stop_token_ids or [])}
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
