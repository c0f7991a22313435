"""BASELINE configs[4]'s single-GPU shape on the current kernels: an 8-FRAME video prompt + 10 s audio (the request
`bench.py --frames 8` times: video_audio_demo.py:30-118 samples the frames, :199-212 puts one image sentinel per frame; the
duplex demo feeds the same shape, web_demo/web_interactive_demo.py:284-366) at the released geometry:

  8 frames of 448 x 448 -> ONE batch of n = 8 tiles through the 24-layer InternViT + projector: 8 x 256 image tokens
  10 s of audio (998 fbank frames) -> 24-layer Whale + adapter: 124 audio tokens
  S = 1 + 139 + 2048 + 32 + 124 = 2344 prompt rows (experts see ~586 rows each: four m-tiles per expert in the streaming GEMM,
  the longest prefill any test runs) through VITA_VIDEO_LAYERS backbone layers (default 16; the 32-layer run of r06 is
  profiles/r06_video_shape_parity_32.txt), 6 greedy steps.
  ONE of the 37 504 router decisions of the first 16 layers — row 2272 at layer 10 — sits on a tie of the oracle's own 2nd / 3rd expert logits
  (margin 6.4e-06, below fp32 re-association noise); the device takes the other expert there, that row's MoE output differs from layer 10 on, and
  from layer 11 on every LATER row (they attend to its K / V) moves with it — last row included (logits 1.3e-3 off the oracle's branch at 16
  layers).  The reference is ill-conditioned at such a decision, so the test follows BOTH of its branches: a decision may differ only where the
  oracle's margin is a tie, and the oracle is then re-run from that layer with the device's pair forced at that one decision (oracle/mixtral.py
  moe force=, oracle/stream.py route_override / resume); against that pass nothing is excused — every other router decision, hidden states,
  logits < 1e-3 and ids (the first form of this check excused the tied row alone).

against the fp64 encoder restatements and the layer-streamed fp32 oracle: encoder outputs, spliced embeddings, every router
decision, hidden states, logits < 1e-3, ids ==.  (n = 5 tiles: tests/test_assets_gpu.py; n = 1: tests/test_realgeom_gpu.py.)"""
import os
import time

import numpy as np
import pytest
import torch

from oracle import encoders as oe, hashw, stream
from tests.util import assert_close, report, to_np
from vita_amd.checkpoint import synth_mixtral_device, synth_state_dict
from vita_amd.config import VitaConfig
from vita_amd.host.synthetic import make_request

pytestmark = pytest.mark.gpu
T_NEW, SEED, FRAMES = 6, 0, 8
LAYERS = int(os.environ.get("VITA_VIDEO_LAYERS", "16"))


@pytest.mark.timeout(1800)
def test_eight_frame_video_prompt_matches_oracle(dev):
    from vita_amd.model.vita_mixtral import VITAMixtralForCausalLM
    cfg = VitaConfig()
    cfg.text.num_hidden_layers = LAYERS
    t, L = cfg.text, LAYERS
    t0 = time.time()
    packed = synth_mixtral_device(cfg, dev, seed=SEED)
    sd_enc = synth_state_dict(cfg, seed=1, rich=False, parts=("vision", "audio"))
    model = VITAMixtralForCausalLM(cfg, sd_enc, device=dev, packed_llm=packed, max_new_tokens=T_NEW + 8, max_prefill=2560,
                                   keep_scores=True)
    model.get_vision_tower().load_model()
    req = make_request(cfg, frames=FRAMES)
    pix = torch.from_numpy(req["pixel_values"]).to(dev)
    feats = torch.from_numpy(req["fbank"]).to(dev)
    ids = torch.tensor([req["input_ids"]], dtype=torch.long, device=dev)
    audios = {"audios": feats[None], "lengths": torch.tensor([feats.shape[0]], device=dev)}
    vit = model.get_vision_tower()(pix)                                  # n = 8 tiles in one batch
    img = model.model.mm_projector(vit)
    _, _, _, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix, audios)
    eng = model.engine
    _, hid = eng.prefill(emb[0], want_hidden=True, want_route=True)
    route = eng.route_ids.cpu().numpy()
    eng.decode(T_NEW - 1)
    torch.cuda.synchronize()
    toks, logits = eng.generated(), to_np(eng.logits_all[:T_NEW])
    d_vit, d_img, d_emb = to_np(vit), to_np(img), to_np(emb[0])
    d_hid = {l: to_np(hid[l]) for l in {0, L - 1}}
    S = d_emb.shape[0]
    print(f"[video] device side done in {time.time() - t0:.1f}s: S={S}, tokens {toks}")
    assert d_vit.shape == (FRAMES, 256, 4096) and S == 1 + 139 + FRAMES * 256 + 32 + 124 == 2344
    eng.close()
    del model, packed, hid
    torch.cuda.empty_cache()

    t0 = time.time()
    o_vit = oe.internvit_tower(sd_enc, cfg.vision, req["pixel_values"])
    o_img = oe.projector(sd_enc, o_vit)
    o_aud = oe.whale_encoder(sd_enc, cfg.audio, req["fbank"])[0]
    table = hashw.fill((t.vocab_size, t.hidden_size), hashw.tensor_seed("model.embed_tokens.weight", SEED))
    o_emb = np.asarray(oe.splice(np.asarray(req["input_ids"]), table, o_img, o_aud[None] if o_aud.ndim == 2 else o_aud), np.float32)
    del table
    print(f"[video] oracle encoders (8 tiles, 998 frames) + splice in {time.time() - t0:.1f}s")
    assert_close("InternViT (24 layers, 8 tiles in one batch) + pixel shuffle", d_vit, o_vit, atol=2e-3, rtol=1e-3)
    assert_close("projector", d_img, o_img, atol=2e-3, rtol=1e-3)
    assert_close("spliced inputs_embeds (S = 2344)", d_emb, o_emb, atol=2e-3, rtol=1e-3)

    full = np.concatenate([o_emb, stream.embed_rows(t, toks[:-1], SEED)], 0)
    t0 = time.time()
    ref = stream.forward(t, SEED, full, n_layers=L, capture=sorted(d_hid), logits_from=S - 1, margins=True, keep_inputs=True)
    print(f"[video] oracle backbone ({L} layers, {full.shape[0]} rows) in {time.time() - t0:.1f}s")
    # Router decisions: 2344 rows x L layers of a DISCONTINUOUS choice.  Where the ORACLE's own margin between its 2nd and 3rd expert is a
    # tie at fp32 re-association noise (< TIE) the device may legitimately take the other expert; that row's MoE output then differs,
    # and from the next layer on every LATER row (they attend to its K / V) moves with it — last row and logits included (r06 at 16
    # layers: row 2272 at layer 10, margin 6.4e-06, logits 1.3e-3 off the oracle's branch).  Both branches are "the reference": the
    # oracle is re-run FROM THE TIED LAYER with the device's pair forced at that one decision (om.moe force), and everything — every
    # other router decision, hidden states, logits < 1e-3, ids — must match THAT pass with no row excused.  A decision that differs
    # where the margin is not a tie fails at once.
    TIE, MAX_TIES = 3e-4, 4
    S_full = full.shape[0]
    dev_route = np.sort(route, -1)                                                     # [L, S]
    override, n_ties = {}, 0
    while True:
        diff = (dev_route != np.sort(ref["route"][:, :S], -1)).any(-1)                 # [L, S]
        if not diff.any():
            break
        l0 = int(np.argmax(diff.any(1)))                                               # first layer with a difference
        rows = np.flatnonzero(diff[l0])
        for r in rows:
            mg = float(ref["margin"][l0, r])
            print(f"  row {int(r)}: decision differs at layer {l0}, oracle margin (2nd - 3rd expert logit) {mg:.2e}")
            assert mg < TIE, f"row {int(r)} layer {l0}: the device took another expert where the oracle's margin is {mg:.2e} (not a tie)"
            override.setdefault(l0, {})[int(r)] = tuple(int(v) for v in route[l0, r])
        n_ties += len(rows)
        assert n_ties <= MAX_TIES, f"{n_ties} tied decisions"
        t0 = time.time()
        prev = ref
        ref = stream.forward(t, SEED, full, n_layers=L, capture=sorted(d_hid), logits_from=S - 1, margins=True, keep_inputs=True,
                             route_override=override, resume=(l0, prev["inputs"][l0]))
        for l in range(l0):                                                            # the layers in front of the tie are unchanged
            ref["route"][l], ref["margin"][l], ref["inputs"][l] = prev["route"][l], prev["margin"][l], prev["inputs"][l]
        for l in prev["hidden"]:
            if l < l0:
                ref["hidden"][l] = prev["hidden"][l]
        print(f"[video] oracle re-run from layer {l0} with the device's pair at {sorted(override[l0])} in {time.time() - t0:.1f}s")
        assert S_full == full.shape[0]
    print(f"router top-2 sets: {route.shape[0] * S} decisions identical" + (f" on the branch of {n_ties} tied decision(s) {override}" if n_ties else ""))
    for l in sorted(d_hid):
        h_ref = ref["hidden"][l][:S]
        assert_close(f"hidden after layer {l}", d_hid[l], h_ref, atol=3e-4 * float(np.abs(h_ref).max()), rtol=1e-3)
    ref_ids = ref["logits"].argmax(-1).tolist()
    print("device ids", toks, "oracle ids", ref_ids)
    print(report(f"logits of the {T_NEW} steps", logits, ref["logits"]))
    assert toks == ref_ids
    assert np.abs(logits - ref["logits"]).max() < 1e-3
