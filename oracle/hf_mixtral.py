"""ORACLE (test infrastructure): the third-party implementation that carries the reference's
Mixtral arithmetic — HF transformers' MixtralForCausalLM, which VITAMixtralForCausalLM subclasses
(vita/model/language_model/vita_mixtral.py:7-13,225-233).  The reference pins 4.41.1; the image
ships 5.15.0, whose parameter names differ (experts fused into gate_up_proj / down_proj), so this
module maps a 4.41-named state dict onto the installed model and runs it in fp32 on the CPU with
a hand-rolled greedy loop.  Used to pin oracle/mixtral.py and to generate tests/golden/."""
import numpy as np
import torch


def build(tcfg, sd):
    from transformers import MixtralConfig, MixtralForCausalLM
    c = MixtralConfig(hidden_size=tcfg.hidden_size, num_hidden_layers=tcfg.num_hidden_layers,
                      num_attention_heads=tcfg.num_attention_heads, num_key_value_heads=tcfg.num_key_value_heads,
                      head_dim=tcfg.head_dim, intermediate_size=tcfg.intermediate_size,
                      num_local_experts=tcfg.num_local_experts, num_experts_per_tok=tcfg.num_experts_per_tok,
                      vocab_size=tcfg.vocab_size, rope_theta=tcfg.rope_theta, rms_norm_eps=tcfg.rms_norm_eps,
                      attn_implementation="eager", max_position_embeddings=32768, sliding_window=None)
    with torch.no_grad():
        m = MixtralForCausalLM(c).float().eval()
        T = lambda k: torch.from_numpy(np.asarray(sd[k], np.float32))
        new = {"model.embed_tokens.weight": T("model.embed_tokens.weight"), "model.norm.weight": T("model.norm.weight"),
               "lm_head.weight": T("lm_head.weight")}
        for l in range(tcfg.num_hidden_layers):
            p = f"model.layers.{l}."
            for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
                new[p + f"self_attn.{nm}.weight"] = T(p + f"self_attn.{nm}.weight")
            new[p + "input_layernorm.weight"] = T(p + "input_layernorm.weight")
            new[p + "post_attention_layernorm.weight"] = T(p + "post_attention_layernorm.weight")
            new[p + "mlp.gate.weight"] = T(p + "block_sparse_moe.gate.weight")
            E = tcfg.num_local_experts
            w1 = torch.stack([T(p + f"block_sparse_moe.experts.{e}.w1.weight") for e in range(E)])
            w3 = torch.stack([T(p + f"block_sparse_moe.experts.{e}.w3.weight") for e in range(E)])
            w2 = torch.stack([T(p + f"block_sparse_moe.experts.{e}.w2.weight") for e in range(E)])
            new[p + "mlp.experts.gate_up_proj"] = torch.cat([w1, w3], dim=1)  # gate rows then up rows (chunk(2))
            new[p + "mlp.experts.down_proj"] = w2
        missing, unexpected = m.load_state_dict(new, strict=False)
        assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    return m


@torch.no_grad()
def greedy(m, embeds, n_new):
    """embeds float32 [S, H] numpy.  Returns (ids, logits per step [n, V], hidden after each layer of the prefill)."""
    x = torch.from_numpy(np.asarray(embeds, np.float32))[None]
    out = m(inputs_embeds=x, use_cache=True, output_hidden_states=True)
    hidden = torch.stack(out.hidden_states[1:])[:, 0].numpy()  # last entry is pre-final-norm in 5.x? see note
    past = out.past_key_values
    ids, lg = [], []
    logits = out.logits[0, -1]
    for _ in range(n_new):
        tok = int(torch.argmax(logits))
        ids.append(tok)
        lg.append(logits.numpy().copy())
        out = m(input_ids=torch.tensor([[tok]]), past_key_values=past, use_cache=True)
        past = out.past_key_values
        logits = out.logits[0, -1]
    return ids, np.stack(lg), hidden
