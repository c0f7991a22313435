"""Not a test: prints what the GPU box looks like and times the decode/prefill path at the real
layer geometry with a few layers (cheap to generate).  Usage: python tests/gpu_probe.py [layers]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vita_amd.checkpoint import synth_mixtral_device  # noqa: E402
from vita_amd.config import TextConfig, VitaConfig  # noqa: E402
from vita_amd.engine import MixtralEngine  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    p = torch.cuda.get_device_properties(0)
    print("device:", p.name, "CUs", p.multi_processor_count, "mem GB", p.total_memory / 2**30)
    print("host: cores", os.cpu_count(), "mem GB",
          os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2**30)
    dev = torch.device("cuda:0")
    cfg = VitaConfig()
    cfg.text = TextConfig(num_hidden_layers=L)
    t0 = time.time()
    packed = synth_mixtral_device(cfg, dev, seed=0)
    torch.cuda.synchronize()
    print(f"weights for {L} layers generated in {time.time() - t0:.1f}s")
    S = 552
    eng = MixtralEngine(cfg, packed, dev, max_ctx=2048, max_prefill=S, max_new=512)
    emb = torch.randn(S, cfg.text.hidden_size, device=dev) * 0.02
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    eng.prefill(emb)  # warm-up
    torch.cuda.synchronize()
    ev[0].record(); eng.prefill(emb); ev[1].record()
    torch.cuda.synchronize()
    pre_ms = ev[0].elapsed_time(ev[1])
    t = cfg.text
    layer_bytes = 2 * ((t.num_attention_heads + 2 * t.num_key_value_heads) * 128 * t.hidden_size +
                       t.hidden_size * t.num_attention_heads * 128 + 2 * 3 * t.intermediate_size * t.hidden_size +
                       t.num_local_experts * t.hidden_size)
    head_bytes = 2 * t.vocab_size * t.hidden_size
    print(f"prefill S={S} {L} layers: {pre_ms:.2f} ms  ({pre_ms / L:.3f} ms/layer incl. lm_head)")
    eng.decode(8)
    torch.cuda.synchronize()
    n = 64
    ev[2].record(); eng.decode(n); ev[3].record()
    torch.cuda.synchronize()
    ms = ev[2].elapsed_time(ev[3]) / n
    tot = L * layer_bytes + head_bytes
    print(f"decode: {ms * 1000:.1f} us/token for {L} layers + lm_head; weights/token {tot / 1e9:.3f} GB "
          f"-> {tot / ms / 1e6:.1f} GB/s")
    # isolate lm_head cost by timing an L-layer vs 1-layer engine is overkill; report extrapolation instead
    per_layer_est = (ms - head_bytes / 5.0e9) / L  # assume ~5 TB/s for the head
    print(f"est per-layer {per_layer_est * 1000:.1f} us -> 32 layers ~ {32 * per_layer_est + head_bytes / 5.0e9:.2f} ms/token")
    print("tokens:", eng.generated()[:8])


if __name__ == "__main__":
    main()
