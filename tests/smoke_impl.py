"""__graft_entry__.smoke(): one tiny omni-modal request (2 image tiles + audio + text) through the
HIP path on cuda:0 — encoders, projector, splice, Mixtral prefill and a few greedy decode steps —
checked against the golden vectors recorded from the reference's own modules."""
import os

import numpy as np
import torch


def run():
    from vita_amd import _lib
    from vita_amd.config import VitaConfig
    from vita_amd.model import build_synthetic_model
    _lib.load()  # fails loudly if the HIP extension is missing
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_e2e.npz"))
    model, _ = build_synthetic_model(VitaConfig.tiny(), seed=int(g["seed"]), device=dev, max_new_tokens=16,
                                     max_prefill=256)
    ids = torch.from_numpy(g["input_ids"])[None].to(dev)
    audios = {"audios": torch.from_numpy(g["feats"])[None].to(dev), "lengths": torch.tensor([g["feats"].shape[0]]).to(dev)}
    out = model.generate(ids, images=torch.from_numpy(g["pix"]).to(dev), audios=audios, do_sample=False, num_beams=1,
                         output_scores=True, return_dict_in_generate=True, max_new_tokens=8, eos_token_id=-1)
    got = out.sequences[0, ids.shape[1]:].tolist()
    ref = g["gen_ids"][:8].tolist()
    err = max(float((out.scores[i][0].cpu() - torch.from_numpy(g["gen_logits"][i])).abs().max()) for i in range(8))
    print(f"smoke: ids {got} ref {ref} max|logit diff| {err:.2e}")
    assert got == ref, "greedy ids differ from the reference golden"
    assert err < 1e-3, "logits differ from the reference golden by more than 1e-3"
    print("smoke OK")
