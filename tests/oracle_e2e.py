"""The whole request on the ORACLE (test infrastructure only): encoders + projector + splice + greedy decode of
`oracle/encoders.py` / `oracle/mixtral.py` on the tensors a test handed to the HIP path.  Used by the drop-in and
serving tests so that the checkpoint-loader path, the LLM surface and the demo call sequences are compared with the
oracle directly (VERDICT r02 "self-comparing tests"), not with another run of the HIP kernels.

Restates: vita/model/vita_arch.py:131-134,151-329 (encode + splice) feeding HF Mixtral greedy decode
(video_audio_demo.py:257-270)."""
import numpy as np

from oracle import encoders as oe
from oracle import mixtral as om


def _np(t):
    return t.detach().float().cpu().numpy() if hasattr(t, "detach") else np.asarray(t, np.float32)


def oracle_embeds(sd, cfg, input_ids, pix=None, fbank=None, fbank_len=None):
    """inputs_embeds [S, H] of one request: `input_ids` with -200 / -500 sentinels, `pix` [n,3,h,w] (or None),
    `fbank` [T,80] raw features of the single clip (or None)."""
    ids = np.asarray(_np(input_ids), np.int64).reshape(-1)
    n_img, n_aud = int((ids == -200).sum()), int((ids == -500).sum())
    embed = np.asarray(sd["model.embed_tokens.weight"], np.float32)
    H = embed.shape[1]
    if n_img:
        vit = oe.internvit_tower(sd, cfg.vision, _np(pix))
        img = np.asarray(oe.projector(sd, vit), np.float32)
    else:
        img = np.zeros((1, 0, H), np.float32)          # the dummy image contributes a zero-length slice (vita_arch.py:240-251)
    if n_aud:
        f = _np(fbank)
        z, _mask = oe.whale_encoder(sd, cfg.audio, f, length=fbank_len)
        aud = np.asarray(z, np.float32)[None]          # every row is spliced, masked or not (vita_arch.py:294-297)
    else:
        aud = np.zeros((1, 0, H), np.float32)
    return np.asarray(oe.splice(ids, embed, img, aud, max_len=cfg.tokenizer_model_max_length), np.float32)


def oracle_generate(sd, cfg, input_ids, pix=None, fbank=None, n_new=8, eos=None, fbank_len=None):
    """-> (greedy ids, per-step fp32 logits [n, V], inputs_embeds)."""
    emb = oracle_embeds(sd, cfg, input_ids, pix, fbank, fbank_len)
    ids, logits = om.MixtralOracle(sd, cfg.text).greedy(emb, n_new, eos=eos)
    return ids, logits, emb
