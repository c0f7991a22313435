"""VITAMixtralForCausalLM on HIP — the drop-in for the reference class of the same name
(vita/model/language_model/vita_mixtral.py:225-415) plus the multimodal assembly of
VITAMetaForCausalLM (vita/model/vita_arch.py:110-407) and the greedy loop of HF generate as
video_audio_demo.py:257-270 drives it.

What a caller of the reference finds unchanged: get_vision_tower(), get_audio_encoder(),
process_images(), encode_images(), prepare_inputs_labels_for_multimodal() (inference subset),
generate(input_ids, images=, audios=, ... ) -> object with .sequences / .scores, config, dtype,
eval(), resize_token_embeddings().  Only greedy decoding is implemented (the reference demos
use do_sample=False / temperature 0.01)."""
from types import SimpleNamespace

import numpy as np
import torch

from .. import ops
from ..config import VitaConfig
from ..engine import MixtralEngine
from ..host.constants import AUDIO_TOKEN_INDEX, IMAGE_TOKEN_INDEX
from ..host.image_processing import process_images as _process_images
from .encoders import InternViTVisionTower, VisionProjector, WhaleAudioEncoder, _HipModule


class GenerateOutput(SimpleNamespace):
    """Minimal stand-in for HF GenerateDecoderOnlyOutput: .sequences, .scores."""

    def __getitem__(self, k):
        return getattr(self, k)


class _Backbone(_HipModule):
    """`model.model` of the reference: holds the towers, projector, audio encoder, embed_tokens."""

    def __init__(self, vision_tower, mm_projector, audio_encoder, embed):
        super().__init__()
        self.vision_tower, self.mm_projector, self.audio_encoder = vision_tower, mm_projector, audio_encoder
        self._embed = embed

    def get_vision_tower(self):
        return self.vision_tower

    def get_audio_encoder(self):
        return self.audio_encoder

    def embed_tokens(self, ids):
        kind = torch.zeros_like(ids, dtype=torch.int32)
        H = self._embed.shape[1]
        return ops.embed_splice(kind, ids.to(torch.int32), self._embed, None, None, H)


class VITAMixtralForCausalLM(_HipModule):
    def __init__(self, cfg: VitaConfig, state_dict, device="cuda:0", packed_llm=None, max_new_tokens=1024,
                 max_prefill=None, rank=0, world=1, keep_scores=True, max_seqs=0, kv_pool_tokens=None,
                 gpu_memory_utilization=None):
        super().__init__()
        from ..checkpoint import pack_mixtral
        self.vcfg_all = cfg
        self._device = torch.device(device)
        t = cfg.text
        self.config = SimpleNamespace(
            model_type="vita-mixtral", hidden_size=t.hidden_size, vocab_size=t.vocab_size,
            num_hidden_layers=t.num_hidden_layers, image_aspect_ratio=cfg.image_aspect_ratio,
            tokenizer_model_max_length=cfg.tokenizer_model_max_length, tokenizer_padding_side="right",
            mm_vision_tower="InternViT-300M-448px", mm_projector_type="mlp2x_gelu", mm_hidden_size=cfg.vision.out_dim,
            mm_audio_encoder="audio-encoder", max_dynamic_patch=cfg.max_dynamic_patch,
            bos_token_id=t.bos_token_id, eos_token_id=t.eos_token_id)
        self.generation_config = SimpleNamespace(pad_token_id=None, eos_token_id=t.eos_token_id, max_new_tokens=None)
        tower = InternViTVisionTower("InternViT-300M-448px", vcfg=cfg.vision)
        audio = WhaleAudioEncoder(acfg=cfg.audio, llm_dim=t.hidden_size)
        proj = VisionProjector()
        if state_dict is not None:
            tower.set_state_dict(state_dict, device)
            if any(k.startswith("model.mm_projector.") for k in state_dict):
                proj.load(state_dict, device)
            if any(k.startswith("model.audio_encoder.") for k in state_dict):
                audio.load(state_dict, device)
        self.packed = packed_llm if packed_llm is not None else pack_mixtral(state_dict, cfg, device, rank, world)
        self.model = _Backbone(tower, proj, audio, self.packed["embed"])
        self.max_new_tokens = max_new_tokens
        self.max_prefill = max_prefill or cfg.tokenizer_model_max_length
        if max_seqs > 0 and kv_pool_tokens is None:
            kv_pool_tokens = self.default_kv_pool_tokens(max_seqs, gpu_memory_utilization, world)
        self.kv_pool_tokens = kv_pool_tokens
        self.engine = MixtralEngine(cfg, self.packed, self._device, max_prefill=self.max_prefill,
                                    max_new=max_new_tokens, rank=rank, world=world,
                                    logit_rows=max_new_tokens if keep_scores else 0,
                                    max_seqs=max_seqs, max_ctx=kv_pool_tokens)   # max_seqs > 0: paged KV pool (serving)
        self.lookahead = 8          # decode steps enqueued per host synchronisation in generate()
        self.last_timing = {}
        self.overlap_encoders = True      # audio tower on a side stream under the vision tower (prepare_inputs_labels_for_multimodal)
        self._enc_stream = None

    def default_kv_pool_tokens(self, max_seqs, gpu_memory_utilization=None, world=1):
        """Paged-KV pool for `max_seqs` concurrent requests when the caller names no size (vLLM sizes its block pool from
        gpu_memory_utilization; AsyncEngineArgs in web_interactive_demo.py:942-951 sets 0.8): room for every slot to hold a
        demo-shaped prompt (<= 1024 tokens: 1-2 tiles + audio + text) plus max_new_tokens, in whole 64-token pages, bounded by
        the stated fraction of the memory that is free once the weights are resident.  (r02 defaulted to ONE sequence's worth
        shared by all slots: eight demo requests kept preempting each other by recompute.)"""
        t = self.vcfg_all.text
        per_seq = -(-(min(self.max_prefill, 1024) + self.max_new_tokens + 1) // 64) * 64
        pool = max_seqs * per_seq
        floor = -(-(self.max_prefill + self.max_new_tokens + 1) // 64) * 64          # one full-length sequence always fits
        nkv_rank = max(1, t.num_key_value_heads // max(1, world))
        bytes_per_token = 2 * t.num_hidden_layers * nkv_rank * t.head_dim * 4        # fp32 K and V rows of every layer
        if torch.cuda.is_available():
            free, _ = torch.cuda.mem_get_info(self._device)
            cap = int(free * float(gpu_memory_utilization or 0.8)) // bytes_per_token // 64 * 64
            pool = min(pool, cap)
        pool = max(pool, floor)
        # SPMD tensor parallelism: every rank runs its own scheduler over its own pool; a rank whose free memory gave it a
        # smaller pool would admit / preempt differently and the per-layer collectives would fall out of step.  All ranks take
        # the smallest pool.
        pool = self._agree_min(pool, world)
        import logging
        logging.getLogger("vita_amd").info("paged KV pool: %d tokens (%d pages, %.2f GB) for %d sequence slots", pool,
                                           pool // 64, pool * bytes_per_token / 1e9, max_seqs)
        return pool

    def _agree_min(self, value, world):
        if world <= 1:
            return value
        import torch.distributed as dist
        if not dist.is_initialized():
            return value          # thread ranks / IpcComm-only set-ups share one process and one memory reading
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor([int(value)], dtype=torch.int64, device=self._device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    # ---- reference surface -------------------------------------------------------------------
    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self.model.get_vision_tower()

    def get_audio_encoder(self):
        return self.model.get_audio_encoder()

    def eval(self):
        return self

    def resize_token_embeddings(self, n):
        if n is not None and n > self.config.vocab_size:
            raise NotImplementedError(
                f"growing the embedding table ({self.config.vocab_size} -> {n}) is not supported by the HIP engine")
        return None

    def process_images(self, images, model_cfg=None):
        tower = self.get_vision_tower()
        if not tower.is_loaded:
            tower.load_model()
        return _process_images(images, tower.image_processor, getattr(model_cfg or self.config, "image_aspect_ratio", None))

    def encode_images(self, images):
        """tower -> projector (vita_arch.py:131-134)."""
        return self.model.mm_projector(self.get_vision_tower()(images))

    # ---- multimodal assembly -----------------------------------------------------------------
    @torch.no_grad()
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, audios):
        """Inference subset of vita_arch.py:151-407 for batch size 1: encode, check the placeholder
        counts, splice image / audio embeddings at the -200 / -500 sentinels, truncate to
        tokenizer_model_max_length.  Returns the reference's 6-tuple with inputs_embeds [1,S,H]."""
        if input_ids.shape[0] != 1:
            raise NotImplementedError("the HIP path serves one request per call (batch 1), like the demo")
        dev = self._device
        ids = input_ids[0]
        if attention_mask is not None:
            ids = ids[attention_mask[0].bool().to(ids.device)]
        if type(images) is list or images.ndim == 5:
            images = torch.cat([im for im in images], dim=0)
        if audios is None:
            raise ValueError("audios must be provided (the reference passes a dummy clip for text/image prompts)")
        if self.overlap_encoders and images.is_cuda:
            # (r06) the two towers are independent (vita_arch.py:189 encodes one after the other) and each is a chain of short,
            # latency-bound launches that leaves most of the chip idle: the audio tower runs on a side stream under the vision
            # tower + projector (scratch sets are per stream, vita_amd/ops.py)
            cur = torch.cuda.current_stream(dev)
            if self._enc_stream is None:
                self._enc_stream = torch.cuda.Stream(device=dev)
            side = self._enc_stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                audio_features = self.get_audio_encoder()(audios["audios"], audios["lengths"])
            image_features = self.encode_images(images)                   # [n_tiles, 256, H]
            cur.wait_stream(side)
            for v in audio_features.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
        else:
            image_features = self.encode_images(images)                   # [n_tiles, 256, H]
            audio_features = self.get_audio_encoder()(audios["audios"], audios["lengths"])
        ids_np = ids.detach().cpu().numpy()
        n_img, n_aud = int((ids_np == IMAGE_TOKEN_INDEX).sum()), int((ids_np == AUDIO_TOKEN_INDEX).sum())
        aud_emb = audio_features["inputs_embeds"]
        assert n_img + (0 if n_img else 1) == image_features.shape[0]       # vita_arch.py:227-231
        assert n_aud + (0 if n_aud else 1) == aud_emb.shape[0]             # vita_arch.py:232-236
        tiles_tok, aud_tok = image_features.shape[1], aud_emb.shape[1]
        kind, idx, ii, ai = [], [], 0, 0
        for t in ids_np.tolist():
            if t == IMAGE_TOKEN_INDEX:
                kind += [1] * tiles_tok
                idx += list(range(ii * tiles_tok, (ii + 1) * tiles_tok)); ii += 1
            elif t == AUDIO_TOKEN_INDEX:
                kind += [2] * aud_tok
                idx += list(range(ai * aud_tok, (ai + 1) * aud_tok)); ai += 1
            else:
                if t < 0 or t >= self.config.vocab_size:
                    raise ValueError(f"token id {t} outside the vocabulary")
                kind.append(0); idx.append(t)
        max_len = getattr(self.config, "tokenizer_model_max_length", None)
        if max_len is not None:
            kind, idx = kind[:max_len], idx[:max_len]
        H = self.config.hidden_size
        emb = ops.embed_splice(torch.tensor(kind, dtype=torch.int32, device=dev),
                               torch.tensor(idx, dtype=torch.int32, device=dev), self.packed["embed"],
                               image_features.reshape(-1, H).contiguous(), aud_emb.reshape(-1, H).contiguous(), H)
        S = emb.shape[0]
        pos = torch.arange(S, device=dev)[None] if position_ids is not None else None
        mask = torch.ones((1, S), dtype=attention_mask.dtype, device=dev) if attention_mask is not None else None
        return None, pos, mask, past_key_values, emb[None], labels

    # ---- forward / generate --------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, images=None, audios=None, **_):
        """Prefill forward.  Returns .logits [1,1,V] of the LAST position only (the reference computes
        all S rows, vita_mixtral.py:171-172, and greedy decoding reads the last)."""
        if inputs_embeds is None:
            _, _, _, _, inputs_embeds, _ = self.prepare_inputs_labels_for_multimodal(
                input_ids, position_ids, attention_mask, past_key_values, labels, images, audios)
        logits, _ = self.engine.prefill(inputs_embeds[0].to(torch.float32), gather_logits=True)   # full row (collective under a sharded head)
        return SimpleNamespace(logits=logits[None, None, :], past_key_values=self.engine)

    @torch.no_grad()
    def generate(self, input_ids=None, images=None, audios=None, do_sample=False, temperature=None, top_p=None,
                 num_beams=1, output_scores=False, return_dict_in_generate=False, max_new_tokens=None,
                 use_cache=True, stopping_criteria=None, inputs_embeds=None, attention_mask=None,
                 eos_token_id=None, streamer=None, **kwargs):
        if do_sample or num_beams != 1:
            raise NotImplementedError("vita_amd implements greedy decoding (do_sample=False, num_beams=1)")
        max_new = int(max_new_tokens or self.generation_config.max_new_tokens or 20)
        if max_new > self.max_new_tokens:
            raise ValueError(f"max_new_tokens={max_new} exceeds the engine capacity {self.max_new_tokens}")
        eos = self.generation_config.eos_token_id if eos_token_id is None else eos_token_id
        eos_set = set(eos if isinstance(eos, (list, tuple)) else [eos]) - {None}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        if inputs_embeds is None:
            _, _, _, _, inputs_embeds, _ = self.prepare_inputs_labels_for_multimodal(
                input_ids, None, attention_mask, None, None, images, audios)
        emb = inputs_embeds[0].to(torch.float32)
        if emb.shape[0] + max_new + 1 > self.engine.max_ctx:
            raise ValueError("prompt + max_new_tokens exceeds the KV-cache capacity")
        ev[1].record()
        eng = self.engine
        eng.prefill(emb)
        prompt = input_ids if input_ids is not None else torch.zeros((1, 0), dtype=torch.long, device=self._device)
        prompt = prompt.to(self._device)
        crit = list(stopping_criteria or [])
        keep_scores = output_scores and eng.logit_rows > 1
        generated, done, n_checked = [], False, 0
        # prompt + every token generated so far, written in place (the stopping criteria see a VIEW of it: no
        # concatenation per token; KeywordsStoppingCriteria decodes only the last max_keyword_len ids)
        seq_buf = None
        if crit:
            seq_buf = torch.empty((1, prompt.shape[1] + max_new), dtype=prompt.dtype, device=self._device)
            seq_buf[:, :prompt.shape[1]] = prompt
        while not done:
            # tokens [n_checked, eng.n_gen) are on the device; look at them one at a time, in order,
            # so stopping is decided exactly as a token-by-token loop would
            torch.cuda.current_stream().synchronize()
            eng.check_device_flag()        # a device-side spin time-out must not yield silently wrong tokens
            new = eng.tokens[n_checked:eng.n_gen].tolist()
            n_before = len(generated)
            if crit and new:
                p0 = prompt.shape[1] + n_checked
                seq_buf[0, p0:p0 + len(new)] = torch.tensor(new, dtype=prompt.dtype, device=self._device)
            for tok in new:
                generated.append(tok)
                n_checked += 1
                stop = tok in eos_set or len(generated) >= max_new
                if not stop and crit:
                    seq = seq_buf[:, :prompt.shape[1] + len(generated)]
                    stop = any(c(seq, None) for c in crit)
                if stop:
                    done = True
                    break
            # streaming hook (duplex serving): receives the tokens accepted in this window, returns False to
            # interrupt the generation (web_interactive_demo.py:340-352 breaks out of its stream the same way)
            if streamer is not None and len(generated) > n_before:
                if streamer(generated[n_before:]) is False:
                    done = True
            if not done:
                eng.decode(min(self.lookahead, max_new - eng.n_gen))
        ev[2].record()
        torch.cuda.synchronize()
        self.last_timing = {"encode_ms": ev[0].elapsed_time(ev[1]), "llm_ms": ev[1].elapsed_time(ev[2]),
                            "prompt_tokens": int(emb.shape[0]), "new_tokens": len(generated)}
        sequences = torch.cat([prompt, torch.tensor([generated], dtype=prompt.dtype, device=self._device)], dim=1)
        if not return_dict_in_generate:
            return sequences
        scores = None
        if keep_scores:
            # vocab-sharded head: a kept row holds this rank's slice and zeros elsewhere; the sum is the full row
            rows = eng.gather_vocab(eng.logits_all[:len(generated)]) if eng.vocab_sharded else eng.logits_all[:len(generated)].clone()
            scores = tuple(rows[i][None] for i in range(len(generated)))
        return GenerateOutput(sequences=sequences, scores=scores, past_key_values=None)
