"""profiles/r02_pmc_FETCH_SIZE.txt -> profiles/r02_pmc_hbm_traffic.json (what bench.py reports as roofline.traffic).
FETCH_SIZE is in KiB and, on gfx950, tallies the 128-B requests of a wide coalesced streaming read at 64 B
(MI355X_MICROARCH.md, HBM section): HBM read bytes = counter * 1024 * 2."""
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "r02_pmc_FETCH_SIZE.txt")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(here, "r02_pmc_hbm_traffic.json")
H, I, E, V, NQKV = 4096, 14336, 8, 51760, 6144
alg = {   # algorithmic bytes per launch, released geometry, TP=1
    "k_dec_gateup": ("k_dec_gateup", 2 * 2 * I * H * 2 + E * H * 2),
    "k_dec_down": ("k_dec_down", 2 * H * I * 2),
    "k_dec_lmhead": ("k_dec_lmhead", V * H * 2),
    "k_dec_ablk": ("k_dec_ablk<2, 2, 8>", (NQKV + H) * H * 2 + 2 * 8 * 128 * 4 * 600),       # r06: fused QKV + attention + O launch (+ ~600 keys of fp32 K / V)
    "k_dec_gemv_qkv": (("k_dec_gemv<2, 8, true,", "k_dec_gemv<2, 8, true>"), NQKV * H * 2),       # r05: a fourth template argument (granule I/O)
    "k_dec_gemv_oproj": (("k_dec_gemv<2, 8, false,", "k_dec_gemv<2, 8, false>"), H * H * 2),
    "k_gemm_ps_moe_gateup (prefill S=552)": (("k_gemm_sp<true", "k_gemm_ps<true"), E * 2 * I * H * 2),   # r04: the specialised kernel is the default
}
rows = {}
for ln in open(src):
    if ln.startswith("#"):
        continue
    f = ln.rstrip("\n").split("\t")
    rows[f[0]] = (int(f[1]), float(f[2]), float(f[5]))
out = {"_how": open(src).readline().lstrip("# ").strip() + "  (profiles/r0N_measure.sh)",
       "_units": "FETCH_SIZE in KiB; bytes = counter * 1024 * 2 on gfx950 for wide coalesced streaming reads (MI355X_MICROARCH.md HBM section)"}
for key, (pat, ab) in alg.items():
    pats = pat if isinstance(pat, tuple) else (pat,)
    m = []
    for pt in pats:                       # first pattern that matches a kernel of the pass
        m = [(n, v) for n, v in rows.items() if pt in n]
        if m:
            break
    if not m:
        continue
    m.sort(key=lambda kv: -kv[1][0])      # several instantiations: the one launched most often
    n, (cnt, kib, us) = m[0]
    b = int(round(kib * 1024 * 2))
    out[key] = {"kernel": n[:80], "FETCH_SIZE_KiB_mean": kib, "launch_records": cnt, "avg_us": us, "hbm_read_bytes_per_launch": b,
                "algorithmic_bytes_per_launch": ab, "ratio": round(b / ab, 4)}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
