// vh_api.hip — the C ABI (include/vita_hip.h): argument checking, error reporting, and the
// Mixtral engine that owns the layer loop for prefill and greedy decode so that one host call
// enqueues a whole forward (HF MixtralModel.forward + GenerationMixin greedy loop, as driven
// by vita/model/language_model/vita_mixtral.py:101-215 and video_audio_demo.py:257-270).
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/vita_hip.h"
#include "vh_kernels.h"

namespace {
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int check_launch(const char* what, int rc) {
    if (rc != 0) return fail(VH_E_SHAPE, "%s: unsupported shape/arguments", what);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VH_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return VH_OK;
}
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
VhTuning g_tuning;
}  // namespace

VhTuning* vh_tuning() { return &g_tuning; }

extern "C" {

int vh_version(void) { return 100; }
const char* vh_last_error(void) { return g_err; }

int vh_tune(const char* key, int value) {
    if (!key) return fail(VH_E_ARG, "vh_tune: null key");
    if (!strcmp(key, "batch_moe_min")) { g_tuning.batch_moe_min = value; return VH_OK; }
    if (!strcmp(key, "batch_decode")) { g_tuning.batch_decode = value; return VH_OK; }
    if (!strcmp(key, "attn_impl")) { g_tuning.attn_impl = value; return VH_OK; }
    if (!strcmp(key, "attn_fa")) { g_tuning.attn_fa = value; return VH_OK; }
    if (!strcmp(key, "attn_rows")) { g_tuning.attn_rows = value; return VH_OK; }
    if (!strcmp(key, "attn_ksplit")) { g_tuning.attn_ksplit = value; return VH_OK; }
    if (!strcmp(key, "prefill_attn_gemm")) { g_tuning.prefill_attn_gemm = value; return VH_OK; }
    if (!strcmp(key, "prefill_fuse_rows")) { g_tuning.prefill_fuse_rows = value; return VH_OK; }
    if (!strcmp(key, "ps_cfg")) { g_tuning.ps_cfg = value; return VH_OK; }
    if (!strcmp(key, "ps_nt")) { g_tuning.ps_nt = value; return VH_OK; }
    if (!strcmp(key, "ps_xcd")) { g_tuning.ps_xcd = value; return VH_OK; }
    if (!strcmp(key, "tp_overlap")) { g_tuning.tp_overlap = value; return VH_OK; }
    if (!strcmp(key, "moe_ksplit")) { g_tuning.moe_ksplit = value; return VH_OK; }
    if (!strcmp(key, "force_allreduce")) { g_tuning.force_allreduce = value; return VH_OK; }
    if (!strcmp(key, "tp_fuse")) { g_tuning.tp_fuse = value; return VH_OK; }
    if (!strcmp(key, "comm_allow_coarse")) { g_tuning.comm_allow_coarse = value; return VH_OK; }
    if (!strcmp(key, "comm_ranks_per_device")) { g_tuning.comm_ranks_per_device = value; return VH_OK; }
    if (!strcmp(key, "dec_fused")) { g_tuning.dec_fused = value; return VH_OK; }
    if (!strcmp(key, "dec_gateup_grid")) { g_tuning.dec_gateup_grid = value; return VH_OK; }
    if (!strcmp(key, "attn_img")) { g_tuning.attn_img = value; return VH_OK; }
    if (!strcmp(key, "attn_xcd")) { g_tuning.attn_xcd = value; return VH_OK; }
    return fail(VH_E_ARG, "vh_tune: unknown key '%s'", key);
}

static int gemm_entry(const char* name, const vh_gemm_args* a, const float* ln_w, const float* ln_b, float ln_eps, float* ln_out,
                      long ld_ln, void* stream);
int vh_gemm(const vh_gemm_args* a, void* stream) { return gemm_entry("vh_gemm", a, nullptr, nullptr, 0.f, nullptr, 0, stream); }
int vh_gemm_ln(const vh_gemm_args* a, const float* ln_w, const float* ln_b, float ln_eps, float* ln_out, long ld_ln, void* stream) {
    if (!ln_w || !ln_out) return fail(VH_E_ARG, "vh_gemm_ln: null pointer");
    return gemm_entry("vh_gemm_ln", a, ln_w, ln_b, ln_eps, ln_out, ld_ln, stream);
}
static int gemm_entry(const char* name, const vh_gemm_args* a, const float* ln_w, const float* ln_b, float ln_eps, float* ln_out,
                      long ld_ln, void* stream) {
    if (!a || !a->A || !a->W || !a->C) return fail(VH_E_ARG, "%s: null pointer", name);
    VhGemmArgs g{};
    g.ln_w = ln_w; g.ln_b = ln_b; g.ln_eps = ln_eps; g.ln_out = ln_out; g.ld_ln = ld_ln;
    g.A = a->A; g.lda = a->lda; g.a_rows = a->a_rows; g.a_rowidx = a->a_rowidx;
    g.nseg = a->nseg; g.seglen = a->seglen;
    for (int i = 0; i < 16; ++i) g.segrow[i] = a->segrow[i];
    g.W = a->W; g.W_up = a->W_up; g.ldw = a->ldw; g.w_group_stride = a->w_group_stride;
    g.group_off = a->group_off; g.ngroups = a->ngroups;
    g.C = a->C; g.ldc = a->ldc; g.c_rowidx = a->c_rowidx;
    g.bias = a->bias; g.scale = a->scale; g.resid = a->resid; g.ldr = a->ldr;
    g.M = a->M; g.N = a->N; g.K = a->K; g.act = a->act;
    g.ws = a->ws; g.ws_bytes = a->ws_bytes; g.ksplit = a->ksplit;
    if ((a->lda % 4) != 0 || (a->ldw % 8) != 0) return fail(VH_E_SHAPE, "%s: lda%%4 / ldw%%8 alignment", name);
    return check_launch(name, vhk_gemm(S(stream), g));
}

int vh_gemm_ps(const vh_gemm_ps_args* a, void* stream) {
    if (!a || !a->A_hi || !a->A_lo || !a->W) return fail(VH_E_ARG, "vh_gemm_ps: null pointer");
    VhGemmPsArgs g{};
    g.A_hi = a->A_hi; g.A_lo = a->A_lo; g.lda = a->lda; g.a_rowidx = a->a_rowidx;
    g.W = a->W; g.W_up = a->W_up; g.ldw = a->ldw; g.w_group_stride = a->w_group_stride;
    g.group_off = a->group_off; g.ngroups = a->ngroups;
    g.C = a->C; g.ldc = a->ldc; g.C_hi = a->C_hi; g.C_lo = a->C_lo; g.ldc_split = a->ldc_split;
    g.c_rowidx = a->c_rowidx; g.bias = a->bias; g.scale = a->scale; g.resid = a->resid; g.ldr = a->ldr;
    g.M = a->M; g.N = a->N; g.K = a->K; g.act = a->act;
    g.ksplit = a->ksplit; g.c_split_stride = a->c_split_stride; g.nslab_out = a->nslab_out;
    if ((size_t)a->lda * 2 * (size_t)(a->M > 0 ? a->M : 1) >= (1ull << 32))
        return fail(VH_E_SHAPE, "vh_gemm_ps: activation plane above the 32-bit offset range");
    return check_launch("vh_gemm_ps", vhk_gemm_ps(S(stream), g));
}
int vh_split_planes(const float* x, long ldx, uint16_t* hi, uint16_t* lo, long ldo, int rows, int cols, void* stream) {
    if (!x || !hi || !lo) return fail(VH_E_ARG, "vh_split_planes: null pointer");
    return check_launch("vh_split_planes", vhk_split_planes(S(stream), x, ldx, hi, lo, ldo, rows, cols));
}

int vh_attention(const vh_attn_args* a, void* stream) {
    if (!a || !a->Q || !a->K || !a->V || !a->O) return fail(VH_E_ARG, "vh_attention: null pointer");
    VhAttnArgs g{};
    g.Q = a->Q; g.ldq = a->ldq; g.hsq = a->hsq; g.K = a->K; g.ldk = a->ldk; g.hsk = a->hsk;
    g.V = a->V; g.ldv = a->ldv; g.hsv = a->hsv; g.P = a->P; g.ldp = a->ldp; g.hsp = a->hsp;
    g.bias_u = a->bias_u; g.bias_v = a->bias_v; g.O = a->O; g.ldo = a->ldo;
    g.bsq = a->bsq; g.bsk = a->bsk; g.bso = a->bso;
    g.B = a->B; g.Hq = a->Hq; g.Hkv = a->Hkv; g.Sq = a->Sq; g.Sk = a->Sk; g.d = a->d;
    g.causal = a->causal; g.q_off = a->q_off; g.klen = a->klen; g.chunk = a->chunk; g.left = a->left;
    g.scale = a->scale;
    if (a->P && (!a->bias_u || !a->bias_v)) return fail(VH_E_ARG, "vh_attention: rel-pos needs bias_u/bias_v");
    if ((a->ldk % 4) || (a->ldv % 4) || (a->hsk % 4) || (a->hsv % 4) || (a->bsk % 4) ||
        (reinterpret_cast<uintptr_t>(a->K) & 15) || (reinterpret_cast<uintptr_t>(a->V) & 15) ||
        (a->P && ((a->ldp % 4) || (a->hsp % 4) || (reinterpret_cast<uintptr_t>(a->P) & 15))))
        return fail(VH_E_SHAPE, "vh_attention: K/V/P rows must be 16-byte aligned (strides multiples of 4 floats)");
    if ((a->ldq % 4) || (a->hsq % 4) || (a->bsq % 4) || (a->ldo % 4) || (a->bso % 4) || (reinterpret_cast<uintptr_t>(a->Q) & 15) ||
        (reinterpret_cast<uintptr_t>(a->O) & 15) || (a->d != 64 && a->d != 128) || (a->P && a->d != 64))
        return fail(VH_E_SHAPE, "vh_attention: Q/O rows must be 16-byte aligned, head_dim 64 or 128 (rel-pos: 64)");
    return check_launch("vh_attention", vhk_attn(S(stream), g));
}

int vh_layernorm(const float* x, long ldx, float* y, long ldy, const float* w, const float* b, int rows, int cols,
                 float eps, int act, float post_scale, void* stream) {
    return check_launch("vh_layernorm", vhk_layernorm(S(stream), x, ldx, y, ldy, w, b, rows, cols, eps, act, post_scale));
}
int vh_rmsnorm(const float* x, float* y, const float* w, int rows, int cols, float eps, void* stream) {
    return check_launch("vh_rmsnorm", vhk_rmsnorm(S(stream), x, y, w, rows, cols, eps));
}
int vh_add(float* x, const float* y, long n, void* stream) { return check_launch("vh_add", vhk_add(S(stream), x, y, n)); }
int vh_cast_bf16_f32(const uint16_t* in, float* out, long n, void* stream) {
    return check_launch("vh_cast_bf16_f32", vhk_cast_bf16_f32(S(stream), in, out, n));
}
int vh_fill_hash_bf16(uint16_t* dst, long rows, long cols, long ld_dst, long ld_src, long idx0, uint64_t seed, void* stream) {
    if (!dst) return fail(VH_E_ARG, "vh_fill_hash_bf16: null pointer");
    return check_launch("vh_fill_hash_bf16", vhk_fill_hash_bf16(S(stream), dst, rows, cols, ld_dst, ld_src, idx0, seed));
}
int vh_vit_embed(const vh_vit_embed_args* a, void* stream) {
    if (!a || !a->pix || !a->patch_w || !a->cls || !a->pos || !a->ln_w || !a->patches || !a->pe || !a->x || !a->h)
        return fail(VH_E_ARG, "vh_vit_embed: null pointer");
    if (a->n < 1 || a->patch < 1 || a->img % a->patch != 0 || a->kpad % 64 != 0 || a->kpad < 3 * a->patch * a->patch || a->C % 8 != 0)
        return fail(VH_E_SHAPE, "vh_vit_embed: n %d img %d patch %d kpad %d C %d", a->n, a->img, a->patch, a->kpad, a->C);
    const int g = a->img / a->patch, rows = a->n * g * g, M = a->n * a->ntok;
    if (a->ntok != g * g + 1) return fail(VH_E_SHAPE, "vh_vit_embed: ntok %d is not grid^2 + 1 (%d)", a->ntok, g * g + 1);
    hipStream_t st = S(stream);
    int rc = check_launch("vh_vit_embed (patchify)", vhk_vit_patchify(st, a->pix, a->patches, a->n, a->img, a->patch, a->kpad));
    if (rc != VH_OK) return rc;
    VhGemmArgs gm{};
    gm.A = a->patches; gm.lda = a->kpad; gm.a_rows = rows; gm.nseg = 1; gm.seglen = a->kpad;
    gm.W = a->patch_w; gm.ldw = a->kpad; gm.C = a->pe; gm.ldc = a->C; gm.M = rows; gm.N = a->C; gm.K = a->kpad;
    gm.bias = a->patch_b; gm.ws = a->ws; gm.ws_bytes = a->ws_bytes; gm.ksplit = a->ws ? 0 : 1;
    if ((rc = check_launch("vh_vit_embed (patch Linear)", vhk_gemm(st, gm))) != VH_OK) return rc;
    if ((rc = check_launch("vh_vit_embed (assemble)", vhk_vit_assemble(st, a->pe, a->cls, a->pos, a->x, a->n, a->ntok, a->C))) != VH_OK) return rc;
    if ((rc = check_launch("vh_vit_embed (norm1)", vhk_layernorm(st, a->x, a->C, a->h, a->C, a->ln_w, a->ln_b, M, a->C, a->eps, VH_ACT_NONE, 1.0f))) != VH_OK)
        return rc;
    if (a->h_planes) {
        uint16_t* hi = reinterpret_cast<uint16_t*>(a->h_planes);
        rc = check_launch("vh_vit_embed (planes)", vhk_split_planes(st, a->h, a->C, hi, hi + (size_t)M * a->C, a->C, M, a->C));
    }
    return rc;
}

int vh_vit_patchify(const float* pix, float* out, int n, int img, int patch, int kpad, void* stream) {
    return check_launch("vh_vit_patchify", vhk_vit_patchify(S(stream), pix, out, n, img, patch, kpad));
}
int vh_vit_assemble(const float* patches, const uint16_t* cls, const uint16_t* pos, float* x, int n, int ntok,
                    int hid, void* stream) {
    return check_launch("vh_vit_assemble", vhk_vit_assemble(S(stream), patches, cls, pos, x, n, ntok, hid));
}
int vh_vit_pixel_shuffle(const float* x, float* out, int n, int grid, int hid, float mul, void* stream) {
    return check_launch("vh_vit_pixel_shuffle", vhk_vit_pixel_shuffle(S(stream), x, out, n, grid, hid, mul));
}
int vh_audio_conv1(const float* feats, const float* mean, const float* istd, const uint16_t* w, const float* b,
                   float* out, int T, int F, int C, void* stream) {
    return check_launch("vh_audio_conv1", vhk_audio_conv1(S(stream), feats, mean, istd, w, b, out, T, F, C));
}
int vh_embed_splice(const int* src_kind, const int* src_idx, const uint16_t* embed, const float* img_feats,
                    const float* aud_feats, float* out, int Sn, int H, void* stream) {
    return check_launch("vh_embed_splice",
                        vhk_embed_splice(S(stream), src_kind, src_idx, embed, img_feats, aud_feats, out, Sn, H));
}

// ---- batch-1 decode operators, one by one ("one entry per kernel K1-K28 group") ----------------------------
// What vh_mixtral_decode chains per layer, for a maintainer who binds a single module of HF's MixtralDecoderLayer
// (reached from vita/model/language_model/vita_mixtral.py:158-169) at batch 1.  Scratch is caller-provided; the
// kernels and their arithmetic are the engine's own.
int vh_router_top2(const float* x, long ldx, const uint16_t* Wg, int E, int H, int rows, int* ids, float* wts, float* probs,
                   void* stream) {
    if (!x || !Wg || !ids || !wts) return fail(VH_E_ARG, "vh_router_top2: null pointer");
    return check_launch("vh_router_top2", vhk_router_top2(S(stream), x, ldx, Wg, E, H, rows, ids, wts, probs));
}

int vh_moe_decode(const float* x, const float* delta, const float* norm_w, float eps, const uint16_t* Wg, const uint16_t* W1,
                  const uint16_t* W3, const uint16_t* W2, int E, int I, int H, float* x_out, float* y, int* route, float* hbuf,
                  void* stream) {
    if (!x || !norm_w || !Wg || !W1 || !W3 || !W2 || !y || !route || !hbuf) return fail(VH_E_ARG, "vh_moe_decode: null pointer");
    if (H % 64 || I % 64 || H > 14336 || I > 14336) return fail(VH_E_SHAPE, "vh_moe_decode: H, I must be multiples of 64 and <= 14336");
    int rc = check_launch("vh_moe_decode (gate|up)", vhk_dec_gateup(S(stream), x, delta, x_out, norm_w, eps, Wg, E, W1, W3, I, H, route, hbuf, 0));
    if (rc != VH_OK) return rc;
    return check_launch("vh_moe_decode (down)", vhk_dec_down(S(stream), hbuf, route, W2, H, I, y));
}

int vh_rope_kv_append(const float* qkv, long ldqkv, float* q_out, float* kcache, float* vcache, const float* rope_cos,
                      const float* rope_sin, int Sn, int pos0, int nq, int nkv, int max_ctx, const int* table, void* stream) {
    if (!qkv || !q_out || !kcache || !vcache || !rope_cos || !rope_sin) return fail(VH_E_ARG, "vh_rope_kv_append: null pointer");
    if (Sn < 1 || pos0 < 0 || pos0 + Sn > max_ctx) return fail(VH_E_SHAPE, "vh_rope_kv_append: rows [%d, %d) outside the cache (%d)", pos0, pos0 + Sn, max_ctx);
    return check_launch("vh_rope_kv_append", vhk_rope_kv(S(stream), qkv, ldqkv, q_out, kcache, vcache, rope_cos, rope_sin, Sn, pos0,
                                                         nq, nkv, max_ctx, table, nullptr, 0));
}

int vh_attn_decode(const float* qkv, float* kcache, float* vcache, int pos, const float* rope_cos, const float* rope_sin,
                   int nq, int nkv, int max_ctx, float scale, const int* table, float* part_o, float* part_ml, int* tickets,
                   float* attn_out, void* stream) {
    if (!qkv || !kcache || !vcache || !rope_cos || !rope_sin || !part_o || !part_ml || !tickets || !attn_out)
        return fail(VH_E_ARG, "vh_attn_decode: null pointer");
    if (pos < 0 || pos >= max_ctx) return fail(VH_E_SHAPE, "vh_attn_decode: position %d outside the cache (%d)", pos, max_ctx);
    const int max_splits = (max_ctx + 63) / 64;
    return check_launch("vh_attn_decode", vhk_dec_attn(S(stream), qkv, kcache, vcache, nullptr, rope_cos, rope_sin, part_o, part_ml,
                                                      tickets, attn_out, nq, nkv, max_ctx, max_splits, pos + 1, scale, table));
}

int vh_lmhead_argmax(const float* x, const float* delta, const float* norm_w, float eps, const uint16_t* W, int V, int H,
                     float* logits, int* token_out, float* blk_val, int* blk_idx, int nblk, void* stream) {
    if (!x || !norm_w || !W || !logits || !token_out || !blk_val || !blk_idx) return fail(VH_E_ARG, "vh_lmhead_argmax: null pointer");
    if (nblk < 1 || nblk > 4096 || H % 8) return fail(VH_E_SHAPE, "vh_lmhead_argmax: 1 <= nblk <= 4096, H %% 8 == 0");
    const int grid = nblk < (V + 7) / 8 ? nblk : (V + 7) / 8;
    int rc = check_launch("vh_lmhead_argmax", vhk_dec_lmhead(S(stream), x, delta, norm_w, eps, W, V, H, logits, blk_val, blk_idx, grid,
                                                             nullptr, 1, 0, V));
    if (rc != VH_OK) return rc;
    return check_launch("vh_lmhead_argmax (pick)", vhk_dec_pick(S(stream), blk_val, blk_idx, grid, V, token_out, nullptr));
}

}  // extern "C"

// =========================================================================================
// Mixtral engine
// =========================================================================================
namespace {

// RCCL is resolved at run time (same librccl torch already mapped) so libvita_hip.so has no
// link-time dependency on it and loads on a single-GPU box without RCCL in the path.
struct Id128 { char b[128]; };  // ncclUniqueId is passed BY VALUE (128 bytes)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
bool load_rccl() {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }
    if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return false;
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce) return false;
    g_rccl.lib = h;
    return true;
}

inline size_t al(size_t n, size_t a) { return (n + a - 1) & ~(a - 1); }

// Workspace carver.  Buffers of 1 MiB and more start on 2 MiB boundaries (the allocation granule of the device page
// tables); small ones on 256 B.  The ORDER in vh_mixtral::carve is part of the performance contract: everything the
// prefill kernels stream sits in front of the KV cache, so its placement does not move with max_new / max_ctx.  (r03 measurement,
// profiles/r03_layout_*: prefill time does NOT depend on the placement — 52.5 ms under three step settings and five pads
// on one box, 40.0 ms under two on another; the r02 "39.7 vs 52.6 ms" was the box, not the layout.)
struct Carver {
    char* base; size_t off;
    template <typename T> T* take(size_t count) {
        const size_t bytes = count * sizeof(T);
        off = al(off, bytes >= (1u << 20) ? (size_t(2) << 20) : size_t(256));
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

}  // namespace

struct vh_mixtral {
    vh_mixtral_cfg c;
    std::vector<vh_mixtral_layer> L;
    const uint16_t* embed; const float* final_norm; const uint16_t* lm_head;
    const float* rope_cos; const float* rope_sin;
    int nq, nkv, hd, H, I, E, V, nqkv, max_splits, lm_grid;
    int v0 = 0, Vn = 0;         // this rank's rows of the LM head ([v0, v0 + Vn); the whole table when unsharded)
    float* cand = nullptr;      // [tp_world][2] (value, index) candidates of the vocab-sharded head
    int host_pos = 0;  // host mirror of counters[0] (sizes the split-KV grid without a device read)
    // state
    float *kcache, *vcache;  // [layer][nkv][max_ctx][hd]
    float *xa, *xb, *delta_attn, *delta_moe, *qkv, *part_o, *part_ml, *attn_out, *hbuf, *logits, *blk_val;
    int *blk_idx, *route, *counters, *out_tokens, *attn_cnt;
    // prefill scratch
    float *px, *pxn, *pqkv, *pq, *pattn, *py, *ptmp, *pwts;
    uint16_t *pxn_hi, *pxn_lo, *ph_hi, *ph_lo;   // bf16 hi/lo planes feeding the pre-split MoE GEMMs
    unsigned char* kv_img = nullptr; int img_tiles = 0;   // K / V tile images (VhAttnArgs::kv_img) of a one-shot prefill
    int *pids, *pgoff, *pstok, *psslot, *pnslab;
    // ---- concurrent sequences over a paged KV cache (vLLM's block tables).  The KV pool above is cut into
    // 64-token pages (= one decode-attention tile); a sequence owns a page table, a residual-stream state, its counters
    // and its generated ids.  The kernels see ONE sequence at a time: bind() points the engine's "current sequence"
    // members (xa .. out_tokens, table, host mirrors) at a sequence's slots, unbind() restores the default single-sequence
    // state, so prefill and decode are the same code for both.
    struct Seq {
        bool live = false;
        int host_pos = 0, poisoned = 0, npages = 0;
        std::vector<int> pages;        // host mirror of the device table (sized once: uploads read from it)
    };
    std::vector<Seq> seqs;
    std::vector<int> free_pages;
    std::vector<int> batch_ids_host;   // slot list last uploaded to seq_batch (storage reserved once)
    float* seq_x = nullptr;            // [max_seqs][4][H]: xa, xb, delta_attn, delta_moe
    int *seq_counters = nullptr, *seq_tokens = nullptr, *seq_table = nullptr, *seq_batch = nullptr;
    // scratch "lanes" 1..VH_BMAX-1 of the per-step buffers (lane 0 = the single-sequence ones) for batched decode iterations
    float *l_qkv[VH_BMAX] = {}, *l_part_o[VH_BMAX] = {}, *l_part_ml[VH_BMAX] = {}, *l_attn_out[VH_BMAX] = {}, *l_hbuf[VH_BMAX] = {},
          *l_blk_val[VH_BMAX] = {}, *l_cand[VH_BMAX] = {};
    int *l_attn_cnt[VH_BMAX] = {}, *l_route[VH_BMAX] = {}, *l_blk_idx[VH_BMAX] = {};
    const int* table = nullptr;        // device page table of the bound sequence; null = contiguous rows (default state)
    int bound = -1;
    struct { float *xa, *xb, *da, *dm; int *counters, *out_tokens; int host_pos, poisoned; } dflt{};
    int n_pages() const { return c.max_ctx / 64; }
    int live_seqs() const { int n = 0; for (const Seq& q : seqs) n += q.live; return n; }
    void bind(int s) {
        dflt = {xa, xb, delta_attn, delta_moe, counters, out_tokens, host_pos, poisoned};
        float* x = seq_x + (size_t)s * 4 * H;
        xa = x; xb = x + H; delta_attn = x + 2 * H; delta_moe = x + 3 * H;
        counters = seq_counters + 4 * s;
        out_tokens = seq_tokens + (size_t)s * (c.max_new > 0 ? c.max_new : 1);
        table = seq_table + (size_t)s * max_splits;
        host_pos = seqs[s].host_pos; poisoned = seqs[s].poisoned;
        bound = s;
    }
    void unbind() {
        Seq& q = seqs[bound];
        q.host_pos = host_pos; q.poisoned = poisoned;
        xa = dflt.xa; xb = dflt.xb; delta_attn = dflt.da; delta_moe = dflt.dm;
        counters = dflt.counters; out_tokens = dflt.out_tokens;
        host_pos = dflt.host_pos; poisoned = dflt.poisoned;
        table = nullptr; bound = -1;
    }
    // pages covering positions [0, n_tokens) of sequence s; new table entries are uploaded on st.  -1: pool exhausted.
    int ensure_pages(int s, int n_tokens, hipStream_t st) {
        Seq& q = seqs[s];
        const int need = (n_tokens + 63) / 64, had = q.npages;
        if (need <= had) return 0;
        if (need > max_splits || (int)free_pages.size() < need - had) return -1;
        for (int j = had; j < need; ++j) { q.pages[j] = free_pages.back(); free_pages.pop_back(); }
        q.npages = need;
        if (hipMemcpyAsync(seq_table + (size_t)s * max_splits + had, q.pages.data() + had, (size_t)(need - had) * sizeof(int),
                           hipMemcpyHostToDevice, st) != hipSuccess) return -2;
        return 0;
    }
    void release(int s) {
        Seq& q = seqs[s];
        for (int j = q.npages - 1; j >= 0; --j) free_pages.push_back(q.pages[j]);
        q.npages = 0; q.live = false; q.host_pos = 0; q.poisoned = 0;
    }
    void init_seqs() {
        seqs.assign(c.max_seqs > 0 ? c.max_seqs : 0, Seq{});
        for (Seq& q : seqs) q.pages.assign(max_splits, 0);
        free_pages.clear();
        for (int pg = n_pages() - 1; pg >= 0; --pg) free_pages.push_back(pg);   // page 0 is handed out first
        batch_ids_host.clear();
        batch_ids_host.reserve(seqs.size() + 1);
    }
    int rccl_gen = 0;           // bumped by vh_mixtral_cancel_rccl: a pending vh_mixtral_init_rccl then discards its communicator
    int poisoned = 0;           // a decode step failed half-way: only prefill / reset may follow
    int* route_dbg = nullptr;   // optional: per-layer top-2 expert ids of the next prefill, [layer][token][2]
    // tensor parallel
    vh_allreduce_fn ar_fn; void* ar_user; void* rccl_comm;
    // optional live timing of the dominant decode kernel (gate/up GEMV), sampled every prof_stride layers
    int prof_stride = 0;
    std::vector<hipEvent_t> prof_ev;  // start/stop pairs
    size_t prof_used = 0;

    size_t carve(void* ws) {
        Carver cv{reinterpret_cast<char*>(ws), 0};
        const size_t Sm = (size_t)c.max_prefill;
        // ---- prefill scratch first: fixed offsets for a given (max_prefill, geometry) --------------------------------
        pxn_hi = cv.take<uint16_t>(Sm * H); pxn_lo = cv.take<uint16_t>(Sm * H);
        ph_hi = cv.take<uint16_t>(2 * Sm * I); ph_lo = cv.take<uint16_t>(2 * Sm * I);
        px = cv.take<float>(Sm * H); pxn = cv.take<float>(Sm * H);
        pqkv = cv.take<float>(Sm * nqkv);
        pq = cv.take<float>(Sm * nq * hd); pattn = cv.take<float>(Sm * nq * hd);
        py = cv.take<float>(4 * 2 * Sm * H);   // py: up to 4 K-split slabs of the MoE down projection / 8 of the projections
        ptmp = cv.take<float>(Sm * H);
        pwts = cv.take<float>(2 * Sm);
        pids = cv.take<int>(2 * Sm); pgoff = cv.take<int>(E + 1);
        pstok = cv.take<int>(2 * Sm); psslot = cv.take<int>(2 * Sm);
        pnslab = cv.take<int>(4);
        // K / V tile images of a one-shot prefill for the flash attention kernel (64 KB per KV head and 64-row tile; only when the
        // head grouping is the 4 : 1 that kernel serves)
        img_tiles = (nq == 4 * nkv && hd == 128) ? (c.max_prefill + 63) / 64 : 0;
        kv_img = img_tiles ? cv.take<unsigned char>((size_t)nkv * img_tiles * 65536) : nullptr;
        // ---- decode state ------------------------------------------------------------------------------------------
        xa = cv.take<float>(H); xb = cv.take<float>(H);
        delta_attn = cv.take<float>(H); delta_moe = cv.take<float>(H);
        qkv = cv.take<float>(nqkv);
        attn_out = cv.take<float>((size_t)nq * hd);
        attn_cnt = cv.take<int>(nkv);  // arrival tickets; zero at creation, reset by each last arriver
        hbuf = cv.take<float>((size_t)2 * I);
        blk_val = cv.take<float>(lm_grid);
        blk_idx = cv.take<int>(lm_grid);
        cand = cv.take<float>(2 * (c.tp_world > 0 ? c.tp_world : 1));
        route = cv.take<int>(4);
        counters = cv.take<int>(4);  // {pos, n_generated, attn_done (monotonic), device error flag}
        g_qkv = cv.take<unsigned long long>(nqkv);
        g_attn = cv.take<unsigned long long>(vh_gran_gemv_len(nq * hd));
        // ---- everything whose size follows max_ctx / max_new / logit_rows: behind the fixed part ---------------------------
        out_tokens = cv.take<int>(c.max_new > 0 ? c.max_new : 1);
        part_o = cv.take<float>((size_t)nq * max_splits * hd);
        part_ml = cv.take<float>((size_t)nq * max_splits * 2);
        logits = cv.take<float>((size_t)hist_rows() * V);
        if (c.max_seqs > 0) {
            const size_t n = (size_t)c.max_seqs;
            seq_x = cv.take<float>(n * 4 * H);
            seq_counters = cv.take<int>(n * 4);
            seq_tokens = cv.take<int>(n * (c.max_new > 0 ? c.max_new : 1));
            seq_table = cv.take<int>(n * max_splits);
            seq_batch = cv.take<int>(n);
            l_qkv[0] = qkv; l_part_o[0] = part_o; l_part_ml[0] = part_ml; l_attn_out[0] = attn_out; l_attn_cnt[0] = attn_cnt;
            l_hbuf[0] = hbuf; l_route[0] = route; l_blk_val[0] = blk_val; l_blk_idx[0] = blk_idx; l_cand[0] = cand;
            for (int b = 1; b < VH_BMAX; ++b) {
                l_qkv[b] = cv.take<float>(nqkv);
                l_part_o[b] = cv.take<float>((size_t)nq * max_splits * hd);
                l_part_ml[b] = cv.take<float>((size_t)nq * max_splits * 2);
                l_attn_out[b] = cv.take<float>((size_t)nq * hd);
                l_attn_cnt[b] = cv.take<int>(nkv);
                l_hbuf[b] = cv.take<float>((size_t)2 * I);
                l_route[b] = cv.take<int>(4);
                l_blk_val[b] = cv.take<float>(lm_grid);
                l_blk_idx[b] = cv.take<int>(lm_grid);
                l_cand[b] = cv.take<float>(2 * (c.tp_world > 0 ? c.tp_world : 1));
            }
        }
        kcache = cv.take<float>((size_t)c.n_layers * nkv * c.max_ctx * hd);
        vcache = cv.take<float>((size_t)c.n_layers * nkv * c.max_ctx * hd);
        return al(cv.off, 256);
    }
    int hist_rows() const { return c.logit_rows > 1 ? c.logit_rows : 1; }
    int cand_world() const { return c.tp_world > 1 ? c.tp_world : 1; }   // slots of the vocab-sharded head's candidate vector (loop-back: this rank's only)
    void derive() {
        nq = c.n_q_heads; nkv = c.n_kv_heads; hd = c.head_dim; H = c.hidden; I = c.inter; E = c.n_experts;
        V = c.vocab; nqkv = (nq + 2 * nkv) * hd;
        max_splits = (c.max_ctx + 63) / 64;
        v0 = c.vocab_n > 0 ? c.vocab_lo : 0;
        Vn = c.vocab_n > 0 ? c.vocab_n : V;
        lm_grid = (Vn + 7) / 8;
        if (lm_grid > 1024) lm_grid = 1024;
        if (lm_grid < c.tp_world) lm_grid = c.tp_world;   // blk_val / blk_idx also hold the gathered candidates
    }
    // ---- fused attention-block launch (k_dec_ablk): q|k|v and the attention output travel between its work items as
    // tagged granules; a tag is used once per (step, layer, vector)
    unsigned long long *g_qkv = nullptr, *g_attn = nullptr;   // granule vectors (VhGranVec)
    unsigned gran_epoch = 0;                         // last granule tag handed out (0 = never written)
    int schedule_state = -1;                         // attention block of the last decode call: -1 none yet, 0 three launches, 1 one fused launch
    unsigned next_tag() { if (++gran_epoch == 0) ++gran_epoch; return gran_epoch; }
    vh_comm_t* comm = nullptr;   // the library's IPC all-reduce (not owned)
    hipStream_t cs = nullptr;    // communication stream of the overlapped tensor-parallel prefill
    hipEvent_t ev_c[2] = {nullptr, nullptr}, ev_r[2] = {nullptr, nullptr};   // half computed / half reduced
    // the collective can run on another stream than the compute stream (native RCCL, IPC) — the Python callback cannot
    bool stream_capable() const { return comm != nullptr || rccl_comm != nullptr; }
    int ensure_comm_stream() {
        if (cs) return 0;
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return -1;
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&ev_c[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ev_r[i], hipEventDisableTiming) != hipSuccess) return -1;
        return 0;
    }
    // exchanges run: a tensor-parallel rank, a test that forces the hook, or a loop-back communicator (one rank playing them all)
    bool exchanges() const { return c.tp_world > 1 || vh_tuning()->force_allreduce || (comm && vh_comm_is_loopback(comm)); }
    int allreduce(float* buf, long count, hipStream_t st) {
        if (!exchanges()) return 0;
        if (comm && (size_t)count <= vh_comm_capacity(comm)) return vh_comm_allreduce(comm, buf, count, st) == VH_OK ? 0 : -1;
        if (!ar_fn) return -1;
        return ar_fn(ar_user, buf, count, st);
    }
};

namespace {
int cfg_ok(const vh_mixtral_cfg* c) {
    if (!c) return fail(VH_E_ARG, "null cfg");
    if (c->head_dim != 128) return fail(VH_E_SHAPE, "head_dim must be 128 (got %d)", c->head_dim);
    if (c->top_k != 2) return fail(VH_E_SHAPE, "top_k must be 2");
    if (c->n_experts < 2 || c->n_experts > 8) return fail(VH_E_SHAPE, "n_experts must be in [2,8]");
    if (c->hidden % 64 || c->inter % 64) return fail(VH_E_SHAPE, "hidden and inter must be multiples of 64");
    if (c->hidden > 14336 || c->inter > 14336 || c->n_q_heads * c->head_dim > 14336)
        return fail(VH_E_SHAPE, "dimension above the GEMV register budget (14336)");
    if (c->n_q_heads % c->n_kv_heads || c->n_q_heads / c->n_kv_heads > 4)
        return fail(VH_E_SHAPE, "GQA group must divide and be <= 4");
    if (c->max_ctx < 1 || c->max_prefill < 1 || c->n_layers < 1) return fail(VH_E_SHAPE, "bad sizes");
    if (c->vocab_n < 0 || c->vocab_lo < 0 || (c->vocab_n > 0 && c->vocab_lo + c->vocab_n > c->vocab))
        return fail(VH_E_SHAPE, "vocab shard outside the table");
    if (c->vocab_n > 0 && c->vocab >= (1 << 24)) return fail(VH_E_SHAPE, "vocab-sharded head needs vocab < 2^24");
    if (c->max_seqs < 0 || c->max_seqs > 4096) return fail(VH_E_SHAPE, "max_seqs outside [0,4096]");
    if (c->max_seqs > 0 && (c->max_ctx % 64) != 0) return fail(VH_E_SHAPE, "paged KV cache: max_ctx must be a multiple of the 64-token page");
    return VH_OK;
}
int rccl_allreduce_cb(void* user, float* buf, long count, void* stream) {
    vh_mixtral* m = reinterpret_cast<vh_mixtral*>(user);
    const int rc = g_rccl.AllReduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, m->rccl_comm,
                                    reinterpret_cast<hipStream_t>(stream));
    return rc == 0 ? 0 : -1;
}
}  // namespace

extern "C" {

size_t vh_mixtral_workspace_bytes(const vh_mixtral_cfg* cfg) {
    if (cfg_ok(cfg) != VH_OK) return 0;
    vh_mixtral tmp{};
    tmp.c = *cfg;
    tmp.derive();
    return tmp.carve(nullptr);
}

vh_mixtral_t* vh_mixtral_create(const vh_mixtral_cfg* cfg, const vh_mixtral_layer* layers, const uint16_t* embed,
                                const float* final_norm, const uint16_t* lm_head, const float* rope_cos,
                                const float* rope_sin, void* workspace, size_t workspace_bytes) {
    if (cfg_ok(cfg) != VH_OK) return nullptr;
    if (!layers || !embed || !final_norm || !lm_head || !rope_cos || !rope_sin || !workspace) {
        fail(VH_E_ARG, "vh_mixtral_create: null pointer");
        return nullptr;
    }
    vh_mixtral* m = new vh_mixtral{};
    m->c = *cfg;
    m->derive();
    m->L.assign(layers, layers + cfg->n_layers);
    m->embed = embed; m->final_norm = final_norm; m->lm_head = lm_head;
    m->rope_cos = rope_cos; m->rope_sin = rope_sin;
    m->ar_fn = nullptr; m->ar_user = nullptr; m->rccl_comm = nullptr;
    const size_t need = m->carve(workspace);
    if (need > workspace_bytes) {
        fail(VH_E_ARG, "vh_mixtral_create: workspace too small (%zu < %zu)", workspace_bytes, need);
        delete m;
        return nullptr;
    }
    m->init_seqs();
    return m;
}

void vh_mixtral_destroy(vh_mixtral_t* m) {
    if (!m) return;
    for (hipEvent_t e : m->prof_ev) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) { if (m->ev_c[i]) (void)hipEventDestroy(m->ev_c[i]); if (m->ev_r[i]) (void)hipEventDestroy(m->ev_r[i]); }
    if (m->cs) (void)hipStreamDestroy(m->cs);
    if (m->rccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(m->rccl_comm);
    delete m;
}

int vh_mixtral_set_allreduce(vh_mixtral_t* m, vh_allreduce_fn fn, void* user) {
    if (!m) return fail(VH_E_ARG, "null engine");
    m->ar_fn = fn; m->ar_user = user;
    return VH_OK;
}

int vh_rccl_unique_id(void* out) {
    if (!load_rccl()) return fail(VH_E_COMM, "librccl.so not loadable: %s", dlerror());
    const int rc = g_rccl.GetUniqueId(out);
    return rc == 0 ? VH_OK : fail(VH_E_COMM, "ncclGetUniqueId failed (%d)", rc);
}

int vh_mixtral_init_rccl(vh_mixtral_t* m, const void* uid) {
    if (!m || !uid) return fail(VH_E_ARG, "null argument");
    if (!load_rccl()) return fail(VH_E_COMM, "librccl.so not loadable");
    Id128 id;
    memcpy(id.b, uid, 128);
    void* comm = nullptr;
    const int gen = __atomic_load_n(&m->rccl_gen, __ATOMIC_ACQUIRE);
    const int rc = g_rccl.CommInitRank(&comm, m->c.tp_world, id, m->c.tp_rank);
    if (rc != 0) return fail(VH_E_COMM, "ncclCommInitRank failed (%d)", rc);
    // a caller that gave up on this attempt (vh_mixtral_cancel_rccl: bring-up time-out, the ranks agreed on another
    // collective) must not find the communicator installed later, in the middle of a run
    if (__atomic_load_n(&m->rccl_gen, __ATOMIC_ACQUIRE) != gen) {
        if (g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
        return fail(VH_E_COMM, "RCCL bring-up was cancelled");
    }
    m->rccl_comm = comm;
    m->ar_fn = rccl_allreduce_cb; m->ar_user = m;
    return VH_OK;
}

int vh_mixtral_use_comm(vh_mixtral_t* m, vh_comm_t* c) {
    if (!m) return fail(VH_E_ARG, "null engine");
    m->comm = c;
    return VH_OK;
}

int vh_mixtral_cancel_rccl(vh_mixtral_t* m) {
    if (!m) return fail(VH_E_ARG, "null engine");
    __atomic_add_fetch(&m->rccl_gen, 1, __ATOMIC_ACQ_REL);
    return VH_OK;
}

int vh_mixtral_route_debug(vh_mixtral_t* m, int* ids_out) {
    if (!m) return fail(VH_E_ARG, "null engine");
    m->route_dbg = ids_out;   // device int[n_layers][S][2] filled by the following prefill calls; null = off
    return VH_OK;
}

int vh_mixtral_profile(vh_mixtral_t* m, int stride, int max_samples) {
    if (!m) return fail(VH_E_ARG, "null engine");
    m->prof_stride = stride;
    m->prof_used = 0;
    while ((int)m->prof_ev.size() < 2 * max_samples) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return fail(VH_E_HIP, "hipEventCreate failed");
        m->prof_ev.push_back(e);
    }
    return VH_OK;
}

int vh_mixtral_profile_read(vh_mixtral_t* m, double* total_ms, int* count) {
    if (!m || !total_ms || !count) return fail(VH_E_ARG, "null argument");
    double tot = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < m->prof_used; i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(m->prof_ev[i + 1]) != hipSuccess) return fail(VH_E_HIP, "event sync failed");
        if (hipEventElapsedTime(&ms, m->prof_ev[i], m->prof_ev[i + 1]) != hipSuccess)
            return fail(VH_E_HIP, "hipEventElapsedTime failed");
        tot += ms; ++n;
    }
    *total_ms = tot; *count = n;
    m->prof_used = 0;
    return VH_OK;
}

const int* vh_mixtral_tokens(const vh_mixtral_t* m) { return m ? m->out_tokens : nullptr; }
const int* vh_mixtral_counters(const vh_mixtral_t* m) { return m ? m->counters : nullptr; }
const float* vh_mixtral_logits(const vh_mixtral_t* m) { return m ? m->logits : nullptr; }

int vh_mixtral_reset(vh_mixtral_t* m, void* stream) {
    if (!m) return fail(VH_E_ARG, "null engine");
    if (hipMemsetAsync(m->counters, 0, 4 * sizeof(int), S(stream)) != hipSuccess)
        return fail(VH_E_HIP, "reset memset failed");
    m->host_pos = 0; m->poisoned = 0;
    for (int s = 0; s < (int)m->seqs.size(); ++s)
        if (m->seqs[s].live) m->release(s);      // the pool is one: a reset returns every page
    return VH_OK;
}

// a launcher returns non-zero when it REJECTS its arguments before launching (VH_E_SHAPE); a launch the runtime
// refused shows up in hipGetLastError (VH_E_HIP)
static int launch_failed(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VH_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return fail(VH_E_SHAPE, "%s: launch rejected (shape / arguments)", what);
}
static int head_and_select(vh_mixtral* m, hipStream_t st, const float* x_in, const float* delta, int mode, int set_pos,
                           const VhXchg* cx = nullptr);
#define VH_TRY(expr, what)                             \
    do {                                               \
        if ((expr) != 0) return launch_failed(what);   \
    } while (0)

static int prefill_impl(vh_mixtral* m, const float* embeds, int Sn, int pos0, float* logits_out, float* hidden_dbg,
                        hipStream_t st) {
    if (Sn < 1 || Sn > m->c.max_prefill) return fail(VH_E_SHAPE, "prefill length %d outside [1,%d]", Sn, m->c.max_prefill);
    if (pos0 < 0 || pos0 + Sn >= m->c.max_ctx) return fail(VH_E_SHAPE, "prefill exceeds KV capacity %d", m->c.max_ctx);
    const int H = m->H, I = m->I, E = m->E, nq = m->nq, nkv = m->nkv, hd = m->hd;
    const bool tp = m->c.tp_world > 1 || vh_tuning()->force_allreduce;
    const float scale = 1.0f / sqrtf((float)hd);
    if (hipMemcpyAsync(m->px, embeds, (size_t)Sn * H * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(VH_E_HIP, "prefill: embed copy failed");
    if (hipMemsetAsync(m->counters + 1, 0, 3 * sizeof(int), st) != hipSuccess) return fail(VH_E_HIP, "memset failed");
    m->poisoned = 0;
    if (m->c.vocab_n > 0 && m->c.tp_world > 1 &&
        hipMemsetAsync(m->logits, 0, (size_t)m->hist_rows() * m->V * sizeof(float), st) != hipSuccess)
        return fail(VH_E_HIP, "memset failed");   // vocab-sharded head: a kept score row holds THIS rank's slice, zeros elsewhere
    // overlapped tensor-parallel prefill: needs a collective that takes a stream, and halves that stay 16-byte rows
    const int H2 = H / 2;
    const bool overlap = tp && vh_tuning()->tp_overlap != 0 && m->stream_capable() && (H2 % 4) == 0 && Sn >= 16;
    if (overlap) {
        if (m->ensure_comm_stream() != 0) return fail(VH_E_HIP, "prefill: comm stream creation failed");
        // the comm stream starts behind everything already queued on the compute stream
        (void)hipEventRecord(m->ev_c[0], st);
        (void)hipStreamWaitEvent(m->cs, m->ev_c[0], 0);
    }

    // py holds 8 * max_prefill * H floats: K-split slabs of the projections (and of the MoE down projection later)
    const long Sm = m->c.max_prefill;
    int qkv_slabs = (int)((8L * H) / m->nqkv);
    if (qkv_slabs > 4) qkv_slabs = 4;
    const bool stream_attn = vh_tuning()->prefill_attn_gemm == 0 && qkv_slabs >= 1 && (H % 64) == 0 && ((nq * hd) % 64) == 0 &&
                             H <= 4096 && (m->nqkv % 4) == 0;
    // (r03) K-split slabs are summed by the norm kernel that consumes the rows (VhRowUpdate) on the single-rank path;
    // vh_tune("prefill_fuse_rows", 0) restores the separate slab-sum / combine launches
    const bool fuse_rows = !tp && vh_tuning()->prefill_fuse_rows != 0 && stream_attn;
    const bool attn_planes = stream_attn && vh_tuning()->prefill_fuse_rows != 0 && (hd == 64 || hd == 128);
    // (r06) one-shot prefills whose attention is the flash kernel get K / V as MFMA-ready tile images from the RoPE pass
    bool use_img = false;
    if (stream_attn && pos0 == 0 && m->kv_img && (Sn + 63) / 64 <= m->img_tiles && vh_tuning()->attn_img != 0) {
        VhAttnArgs q{};
        q.Sq = Sn; q.Sk = Sn; q.d = hd; q.Hq = nq; q.Hkv = nkv; q.B = 1; q.causal = 1;
        use_img = vhk_attn_fa_applies(q) != 0;
    }
    bool combine_pending = false;
    int pend_nslab = 1;
    const int* pend_nslab_dev = nullptr;
    const long pend_slab = (long)2 * m->c.max_prefill * H;
    for (int l = 0; l < m->c.n_layers; ++l) {
        const vh_mixtral_layer& w = m->L[l];
        float* kc = m->kcache + (size_t)l * nkv * m->c.max_ctx * hd;
        float* vc = m->vcache + (size_t)l * nkv * m->c.max_ctx * hd;
        bool o_pending = false;
        if (stream_attn) {
            // weight-streaming projections (vh_gemm_ps.hip): the norm emits the bf16 hi/lo planes, the 35 row tiles of
            // S = 552 run as 3 m-tiles per 256 weight rows and the kernel picks a K split that fills the CUs
            // (72 QKV tiles x 3 slabs); the partial slabs are summed by the consumer (RoPE / KV write)
            // (r03) the previous layer's expert outputs are still K-split slabs in py: this norm applies the routing-weighted
            // sum to the residual rows first (VhRowUpdate), so no combine kernel runs between the layers
            const VhRowUpdate cu{m->py, (long)H, pend_slab, pend_nslab_dev, pend_nslab, m->pwts};
            VH_TRY(vhk_rmsnorm_route(st, m->px, nullptr, m->pxn_hi, m->pxn_lo, w.attn_norm, Sn, H, m->c.rms_eps, nullptr, 0,
                                     nullptr, nullptr, combine_pending ? &cu : nullptr), "rmsnorm");
            if (combine_pending && hidden_dbg)
                (void)hipMemcpyAsync(hidden_dbg + (size_t)(l - 1) * Sn * H, m->px, (size_t)Sn * H * sizeof(float),
                               hipMemcpyDeviceToDevice, st);
            combine_pending = false;
            VhGemmPsArgs g{};
            g.A_hi = m->pxn_hi; g.A_lo = m->pxn_lo; g.lda = H;
            g.W = w.wqkv; g.ldw = H; g.C = m->py; g.ldc = m->nqkv; g.M = Sn; g.N = m->nqkv; g.K = H;
            g.ksplit = -qkv_slabs; g.c_split_stride = (long)Sm * m->nqkv; g.nslab_out = m->pnslab + 1;
            VH_TRY(vhk_gemm_ps(st, g), "qkv gemm");
            if (use_img)   // one-shot prefill under the flash kernel: RoPE + cache write + the K / V tile images in one pass
                VH_TRY(vhk_rope_kv_img(st, m->py, m->nqkv, m->pq, kc, vc, m->rope_cos, m->rope_sin, Sn, nq, nkv, m->c.max_ctx,
                                       m->table, m->pnslab + 1, g.c_split_stride, m->kv_img, m->img_tiles), "rope + images");
            else
                VH_TRY(vhk_rope_kv(st, m->py, m->nqkv, m->pq, kc, vc, m->rope_cos, m->rope_sin, Sn, pos0, nq, nkv,
                                   m->c.max_ctx, m->table, m->pnslab + 1, g.c_split_stride), "rope");
        } else {
            VH_TRY(vhk_rmsnorm(st, m->px, m->pxn, w.attn_norm, Sn, H, m->c.rms_eps), "rmsnorm");
            VhGemmArgs g{};
            g.A = m->pxn; g.lda = H; g.a_rows = Sn; g.nseg = 1; g.seglen = H;
            g.W = w.wqkv; g.ldw = H; g.C = m->pqkv; g.ldc = m->nqkv; g.M = Sn; g.N = m->nqkv; g.K = H;
            VH_TRY(vhk_gemm(st, g), "qkv gemm");
            VH_TRY(vhk_rope_kv(st, m->pqkv, m->nqkv, m->pq, kc, vc, m->rope_cos, m->rope_sin, Sn, pos0, nq, nkv,
                               m->c.max_ctx, m->table, nullptr, 0), "rope");
        }
        {
            VhAttnArgs a{};
            a.Q = m->pq; a.ldq = (long)nq * hd; a.hsq = hd;
            a.K = kc; a.ldk = hd; a.hsk = (long)m->c.max_ctx * hd;
            a.V = vc; a.ldv = hd; a.hsv = (long)m->c.max_ctx * hd;
            a.O = m->pattn; a.ldo = (long)nq * hd;
            a.B = 1; a.Hq = nq; a.Hkv = nkv; a.Sq = Sn; a.Sk = pos0 + Sn; a.d = hd;
            a.causal = 1; a.q_off = pos0; a.klen = pos0 + Sn; a.chunk = 0; a.left = -1; a.scale = scale;
            a.ktable = m->table; a.kv_rows = m->c.max_ctx;
            if (use_img) { a.kv_img = m->kv_img; a.img_tiles = m->img_tiles; }
            if (attn_planes) { a.O = nullptr; a.O_hi = m->ph_hi; a.O_lo = m->ph_lo; a.ldo_split = (long)nq * hd; }
            VH_TRY(vhk_attn(st, a), "attention");
        }
        if (stream_attn) {
            // O projection on the streaming kernel: planes of the attention output (written by the attention kernel itself
            // when it is the direct-operand one), K-split slabs, slab sum fused into the FFN norm (or feeding the all-reduce
            // under tensor parallelism)
            const int KO = nq * hd;
            if (!attn_planes) VH_TRY(vhk_split_planes(st, m->pattn, KO, m->ph_hi, m->ph_lo, KO, Sn, KO), "split planes");
            VhGemmPsArgs g{};
            g.A_hi = m->ph_hi; g.A_lo = m->ph_lo; g.lda = KO;
            g.W = w.wo; g.ldw = KO; g.M = Sn; g.K = KO; g.ksplit = -8;
            if (tp && overlap) {
                // column halves: the all-reduce of half 0 (comm stream) runs under the GEMM of half 1 (SURVEY 8(e);
                // o_proj is RowParallel in the reference: vllm_file/mixtral.py:470-476)
                for (int h = 0; h < 2; ++h) {
                    float* part = m->ptmp + (size_t)h * Sn * H2;
                    float* yh = m->py + (size_t)h * 8 * Sm * H2;
                    g.W = w.wo + (size_t)h * H2 * g.ldw; g.N = H2; g.C = yh; g.ldc = H2;
                    g.c_split_stride = Sm * H2; g.nslab_out = m->pnslab + 2 + h;
                    VH_TRY(vhk_gemm_ps(st, g), "o gemm");
                    VH_TRY(vhk_sum_slabs(st, part, H2, yh, H2, Sn, H2, m->pnslab + 2 + h, 1, g.c_split_stride, 0), "slab sum");
                    (void)hipEventRecord(m->ev_c[h], st);
                    (void)hipStreamWaitEvent(m->cs, m->ev_c[h], 0);
                    if (m->allreduce(part, (long)Sn * H2, m->cs) != 0) return fail(VH_E_COMM, "all-reduce failed");
                    (void)hipEventRecord(m->ev_r[h], m->cs);
                }
                (void)hipStreamWaitEvent(st, m->ev_r[0], 0);
                (void)hipStreamWaitEvent(st, m->ev_r[1], 0);
                VH_TRY(vhk_add_halves(st, m->px, m->ptmp, m->ptmp + (size_t)Sn * H2, Sn, H), "add");
            } else {
                g.N = H; g.C = m->py; g.ldc = H; g.c_split_stride = Sm * H; g.nslab_out = m->pnslab + 2;
                VH_TRY(vhk_gemm_ps(st, g), "o gemm");
                if (tp) {
                    VH_TRY(vhk_sum_slabs(st, m->ptmp, H, m->py, H, Sn, H, m->pnslab + 2, 1, g.c_split_stride, 0), "slab sum");
                    if (m->allreduce(m->ptmp, (long)Sn * H, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
                    VH_TRY(vhk_add(st, m->px, m->ptmp, (long)Sn * H), "add");
                } else if (fuse_rows) {
                    o_pending = true;                   // summed into px by the FFN norm below
                } else {
                    VH_TRY(vhk_sum_slabs(st, m->px, H, m->py, H, Sn, H, m->pnslab + 2, 1, g.c_split_stride, 1), "slab sum");
                }
            }
        } else
        {
            VhGemmArgs g{};
            g.A = m->pattn; g.lda = (long)nq * hd; g.a_rows = Sn; g.nseg = 1; g.seglen = nq * hd;
            g.W = w.wo; g.ldw = (long)nq * hd; g.M = Sn; g.N = H; g.K = nq * hd;
            if (tp && overlap) {
                // column halves: the all-reduce of half 0 (comm stream) runs under the GEMM of half 1 (SURVEY 8(e);
                // o_proj is RowParallel in the reference: vllm_file/mixtral.py:470-476)
                for (int h = 0; h < 2; ++h) {
                    float* part = m->ptmp + (size_t)h * Sn * H2;
                    g.W = w.wo + (size_t)h * H2 * g.ldw; g.N = H2; g.C = part; g.ldc = H2;
                    VH_TRY(vhk_gemm(st, g), "o gemm");
                    (void)hipEventRecord(m->ev_c[h], st);
                    (void)hipStreamWaitEvent(m->cs, m->ev_c[h], 0);
                    if (m->allreduce(part, (long)Sn * H2, m->cs) != 0) return fail(VH_E_COMM, "all-reduce failed");
                    (void)hipEventRecord(m->ev_r[h], m->cs);
                }
                (void)hipStreamWaitEvent(st, m->ev_r[0], 0);
                (void)hipStreamWaitEvent(st, m->ev_r[1], 0);
                VH_TRY(vhk_add_halves(st, m->px, m->ptmp, m->ptmp + (size_t)Sn * H2, Sn, H), "add");
            } else {
                if (tp) { g.C = m->ptmp; g.ldc = H; }
                else { g.C = m->px; g.ldc = H; g.resid = m->px; g.ldr = H; }
                VH_TRY(vhk_gemm(st, g), "o gemm");
                if (tp) {
                    if (m->allreduce(m->ptmp, (long)Sn * H, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
                    VH_TRY(vhk_add(st, m->px, m->ptmp, (long)Sn * H), "add");
                }
            }
        }
        bool moe_done = false;
        int nslab = 1;
        const int* nslab_dev = nullptr;
        const long slab = (long)2 * m->c.max_prefill * H;
        {
            // weight-streaming MoE: the norm kernel emits the bf16 hi/lo planes directly, one tall m-tile per
            // expert (weights cross the fabric once), gate|up emits the planes of h, the down projection is
            // K-split into `nslab` partial slabs that the combine kernel adds
            const VhRowUpdate ou{m->py, (long)H, (long)Sm * H, m->pnslab + 2, 1, nullptr};     // the O projection's slabs
            VH_TRY(vhk_rmsnorm_route(st, m->px, nullptr, m->pxn_hi, m->pxn_lo, w.ffn_norm, Sn, H, m->c.rms_eps, w.wrouter,
                                     E, m->pids, m->pwts, o_pending ? &ou : nullptr), "rmsnorm + route");
            VH_TRY(vhk_moe_sort(st, m->pids, Sn, E, m->pgoff, m->pstok, m->psslot), "sort");
            VhGemmPsArgs g{};
            g.A_hi = m->pxn_hi; g.A_lo = m->pxn_lo; g.lda = H; g.a_rowidx = m->pstok;
            g.W = w.w1; g.W_up = w.w3; g.ldw = H; g.w_group_stride = (long)I * H;
            g.group_off = m->pgoff; g.ngroups = E;
            g.C_hi = m->ph_hi; g.C_lo = m->ph_lo; g.ldc_split = I; g.M = 2 * Sn; g.N = I; g.K = H; g.ksplit = 1;
            const bool prof = m->prof_stride < 0 && (l % -m->prof_stride) == 0 && m->prof_used + 2 <= m->prof_ev.size();
            if (prof) (void)hipEventRecord(m->prof_ev[m->prof_used], st);
            VH_TRY(vhk_gemm_ps(st, g), "gate/up gemm");
            if (prof) { (void)hipEventRecord(m->prof_ev[m->prof_used + 1], st); m->prof_used += 2; }
            nslab = vh_tuning()->moe_ksplit;           // < 0: chosen by the kernel from the expert sizes (up to -n)
            if (nslab == 0) nslab = 1;
            if (nslab > 4) nslab = 4;
            if (nslab < -4) nslab = -4;
            if (nslab > (I >> 6)) nslab = 1;
            nslab_dev = nslab < 0 ? m->pnslab : nullptr;
            VhGemmPsArgs d{};
            d.A_hi = m->ph_hi; d.A_lo = m->ph_lo; d.lda = I;
            d.W = w.w2; d.ldw = I; d.w_group_stride = (long)H * I;
            d.group_off = m->pgoff; d.ngroups = E;
            d.C = m->py; d.ldc = H; d.c_rowidx = m->psslot; d.M = 2 * Sn; d.N = H; d.K = I;
            d.ksplit = nslab; d.c_split_stride = slab; d.nslab_out = m->pnslab;
            if (nslab < 0) nslab = 1;                   // (the reducer reads the device value)
            if (tp && overlap) {
                // "all-reduce over xGMI overlapped with the expert GEMMs" (FusedMoE reduce_results, vllm_file/
                // mixtral.py:405-414): the down projection runs as two COLUMN halves (disjoint weight rows); half
                // 0 is combined and all-reduced on the comm stream while half 1 streams its weights
                const long hslab = (long)2 * m->c.max_prefill * H2;            // one K-split slab of a half
                for (int h = 0; h < 2; ++h) {
                    float* yh = m->py + (size_t)h * 4 * hslab;
                    float* part = m->ptmp + (size_t)h * Sn * H2;
                    d.W = w.w2 + (size_t)h * H2 * I; d.N = H2; d.C = yh; d.ldc = H2; d.c_split_stride = hslab;
                    VH_TRY(vhk_gemm_ps(st, d), "down gemm");
                    if (hipMemsetAsync(part, 0, (size_t)Sn * H2 * sizeof(float), st) != hipSuccess)
                        return fail(VH_E_HIP, "memset failed");
                    VH_TRY(vhk_moe_combine(st, part, yh, m->pwts, Sn, H2, nslab, hslab, nslab_dev), "combine");
                    (void)hipEventRecord(m->ev_c[h], st);
                    (void)hipStreamWaitEvent(m->cs, m->ev_c[h], 0);
                    if (m->allreduce(part, (long)Sn * H2, m->cs) != 0) return fail(VH_E_COMM, "all-reduce failed");
                    (void)hipEventRecord(m->ev_r[h], m->cs);
                }
                (void)hipStreamWaitEvent(st, m->ev_r[0], 0);
                (void)hipStreamWaitEvent(st, m->ev_r[1], 0);
                VH_TRY(vhk_add_halves(st, m->px, m->ptmp, m->ptmp + (size_t)Sn * H2, Sn, H), "add");
                moe_done = true;
            } else {
                VH_TRY(vhk_gemm_ps(st, d), "down gemm");
            }
        }
        if (moe_done) {
        } else if (tp) {
            if (hipMemsetAsync(m->ptmp, 0, (size_t)Sn * H * sizeof(float), st) != hipSuccess)
                return fail(VH_E_HIP, "memset failed");
            VH_TRY(vhk_moe_combine(st, m->ptmp, m->py, m->pwts, Sn, H, nslab, slab, nslab_dev), "combine");
            if (m->allreduce(m->ptmp, (long)Sn * H, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
            VH_TRY(vhk_add(st, m->px, m->ptmp, (long)Sn * H), "add");
        } else if (fuse_rows && l + 1 < m->c.n_layers) {
            combine_pending = true;                     // applied by the next layer's attention norm
            pend_nslab = nslab; pend_nslab_dev = nslab_dev;
        } else {
            VH_TRY(vhk_moe_combine(st, m->px, m->py, m->pwts, Sn, H, nslab, slab, nslab_dev), "combine");
        }
        if (m->route_dbg)
            (void)hipMemcpyAsync(m->route_dbg + (size_t)l * 2 * Sn, m->pids, (size_t)2 * Sn * sizeof(int),
                           hipMemcpyDeviceToDevice, st);
        if (hidden_dbg && !combine_pending)
            (void)hipMemcpyAsync(hidden_dbg + (size_t)l * Sn * H, m->px, (size_t)Sn * H * sizeof(float),
                           hipMemcpyDeviceToDevice, st);
    }
    // logits of the last position only (the reference computes all S rows and uses the last:
    // vita_mixtral.py:171-172 + HF greedy argmax(logits[:, -1]))
    {
        const int rc = head_and_select(m, st, m->px + (size_t)(Sn - 1) * H, nullptr, /*mode=*/0, /*set_pos=*/pos0 + Sn);
        if (rc != VH_OK) return rc;
    }
    m->host_pos = pos0 + Sn;
    if (logits_out)
        (void)hipMemcpyAsync(logits_out, m->logits, (size_t)m->V * sizeof(float), hipMemcpyDeviceToDevice, st);
    hipError_t e = hipGetLastError();          // also what an event / copy enqueue above ((void) calls) left behind
    if (e != hipSuccess) return fail(VH_E_HIP, "prefill: %s", hipGetErrorString(e));
    return VH_OK;
}

static int api_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n;
}

int vh_encoder_layer(const vh_encoder_layer_args* a, void* stream) {
    if (!a || !a->x || !a->h_in || !a->qkv_w || !a->proj_w || !a->n2_w || !a->fc1_w || !a->fc2_w || !a->qkv || !a->attn ||
        !a->hmid || !a->mid)
        return fail(VH_E_ARG, "vh_encoder_layer: null pointer");
    const int M = a->M, Cw = a->C, F = a->F;
    if (M < 1 || a->B < 1 || M % a->B != 0 || a->heads < 1 || Cw % a->heads != 0 || Cw % 64 != 0 || F % 64 != 0)
        return fail(VH_E_SHAPE, "vh_encoder_layer: M %d B %d C %d F %d heads %d", M, a->B, Cw, F, a->heads);
    const int d = Cw / a->heads, Sq = M / a->B;
    if (d != 64 && d != 128) return fail(VH_E_SHAPE, "vh_encoder_layer: head_dim %d (64 or 128)", d);
    if (a->P && (!a->bias_u || !a->bias_v || a->B != 1)) return fail(VH_E_ARG, "vh_encoder_layer: rel-pos needs bias_u / bias_v and B = 1");
    if (a->h_out && !a->next_w) return fail(VH_E_ARG, "vh_encoder_layer: h_out without next_w");
    hipStream_t st = S(stream);
    auto lin = [&](const float* A, int K, const uint16_t* W, int N, const float* bias, int act, const float* scale, const float* resid,
                   float* Cout, const float* lw, const float* lb, float* lout) {
        VhGemmArgs g{};
        g.A = A; g.lda = K; g.a_rows = M; g.nseg = 1; g.seglen = K;
        g.W = W; g.ldw = K; g.C = Cout; g.ldc = N; g.M = M; g.N = N; g.K = K; g.act = act;
        g.bias = bias; g.scale = scale; g.resid = resid; g.ldr = N;
        g.ws = a->ws; g.ws_bytes = a->ws_bytes; g.ksplit = a->ws ? 0 : 1;
        g.ln_w = lw; g.ln_b = lb; g.ln_eps = a->eps; g.ln_out = lout; g.ld_ln = N;
        return vhk_gemm(st, g);
    };
    // ---- planes = 1: the four Linears on the streaming GEMM, operands as bf16 hi/lo planes (see include/vita_hip.h) -------------------
    const bool planes = a->planes != 0;
    if (planes && (a->P || !a->ws || a->ws_bytes < (size_t)M * Cw * sizeof(float)))
        return fail(VH_E_ARG, "vh_encoder_layer: planes = 1 needs ws >= 4 M C bytes and no rel-pos operand");
    auto pl_hi = [&](const void* base) { return reinterpret_cast<uint16_t*>(const_cast<void*>(base)); };
    auto pl_lo = [&](const void* base, long cols) { return reinterpret_cast<uint16_t*>(const_cast<void*>(base)) + (size_t)M * cols; };
    // K split of a Linear whose one-pass tiling cannot fill half the chip (N = C: 4 n-tiles): 2 .. 4 by depth, bounded by ws
    auto ks_for = [&](int N, int K) {
        const long nrt = (M + 15) >> 4, NT = (N + 255) / 256;
        if (((nrt + 2) / 3) * NT * 2 >= api_num_cus()) return 1;
        int ks = K / 1024 + 1;
        ks = ks < 2 ? 2 : (ks > 4 ? 4 : ks);
        while (ks > 1 && ((size_t)ks * M * N * sizeof(float) > a->ws_bytes || ks > (K >> 6))) --ks;
        return ks;
    };
    // out = A W^T on planes: raw partial sums into ws (ks slabs), or the finished rows (bias, act) as fp32 / as planes
    auto lin_ps = [&](const void* A, int K, const uint16_t* W, int N, const float* bias, int act, float* Cout, void* Cplanes, int ks) {
        VhGemmPsArgs g{};
        g.A_hi = pl_hi(A); g.A_lo = pl_lo(A, K); g.lda = K;
        g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K;
        if (ks >= 1 && !Cout && !Cplanes) { g.C = a->ws; g.ldc = N; g.ksplit = ks; g.c_split_stride = (long)M * N; }
        else { g.C = Cout; g.ldc = N; g.bias = bias; g.act = act; g.ksplit = 1;
               if (Cplanes) { g.C_hi = pl_hi(Cplanes); g.C_lo = pl_lo(Cplanes, N); g.ldc_split = N; } }
        return vhk_gemm_ps(st, g);
    };
    // x += scale * (sum of the ks slabs + bias); then LN(x; lw, lb) as planes into lout (nullable)
    auto reduce_into_x = [&](int N, int ks, const float* bias, const float* scale, const float* lw, const float* lb, void* lout) {
        VhGemmArgs g{};
        g.ws = a->ws; g.ws_bytes = a->ws_bytes; g.M = M; g.N = N; g.C = a->x; g.ldc = N;
        g.bias = bias; g.scale = scale; g.resid = a->x; g.ldr = N; g.act = VH_ACT_NONE;
        if (lout) { g.ln_w = lw; g.ln_b = lb; g.ln_eps = a->eps; g.ln_hi = pl_hi(lout); g.ln_lo = pl_lo(lout, N); g.ld_ln_split = N; }
        return vhk_gemm_reduce(st, g, ks);
    };
    if (planes) VH_TRY(lin_ps(a->h_in, Cw, a->qkv_w, 3 * Cw, a->qkv_b, VH_ACT_NONE, a->qkv, nullptr, 1), "encoder qkv (planes)");
    else
    VH_TRY(lin(a->h_in, Cw, a->qkv_w, 3 * Cw, a->qkv_b, VH_ACT_NONE, nullptr, nullptr, a->qkv, nullptr, nullptr, nullptr), "encoder qkv");
    {
        VhAttnArgs g{};
        g.Q = a->qkv; g.K = a->qkv + Cw; g.V = a->qkv + 2 * Cw;
        g.ldq = g.ldk = g.ldv = 3L * Cw; g.hsq = g.hsk = g.hsv = d;
        g.bsq = g.bsk = (long)Sq * 3 * Cw; g.bso = (long)Sq * Cw;
        if (planes) { g.O = nullptr; g.O_hi = pl_hi(a->attn); g.O_lo = pl_lo(a->attn, Cw); g.ldo_split = Cw; }
        else g.O = a->attn;
        g.ldo = Cw;
        g.B = a->B; g.Hq = g.Hkv = a->heads; g.Sq = g.Sk = Sq; g.d = d;
        g.klen = (a->klen >= 0 && a->klen < Sq) ? a->klen : Sq; g.chunk = a->chunk; g.left = a->left;
        g.scale = 1.0f / sqrtf((float)d);
        g.P = a->P; g.ldp = a->ldp; g.hsp = d; g.bias_u = a->bias_u; g.bias_v = a->bias_v;
        VH_TRY(vhk_attn(st, g), "encoder attention");
    }
    if (planes) {
        const int ks1 = ks_for(Cw, Cw), ks2 = ks_for(Cw, F);
        VH_TRY(lin_ps(a->attn, Cw, a->proj_w, Cw, nullptr, VH_ACT_NONE, nullptr, nullptr, ks1), "encoder proj (planes)");
        VH_TRY(reduce_into_x(Cw, ks1, a->proj_b, a->ls1, a->n2_w, a->n2_b, a->hmid), "encoder proj reducer");
        VH_TRY(lin_ps(a->hmid, Cw, a->fc1_w, F, a->fc1_b, a->act, nullptr, a->mid, 1), "encoder fc1 (planes)");
        VH_TRY(lin_ps(a->mid, F, a->fc2_w, Cw, nullptr, VH_ACT_NONE, nullptr, nullptr, ks2), "encoder fc2 (planes)");
        VH_TRY(reduce_into_x(Cw, ks2, a->fc2_b, a->ls2, a->next_w, a->next_b, a->h_out), "encoder fc2 reducer");
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) return fail(VH_E_HIP, "vh_encoder_layer: %s", hipGetErrorString(e2));
        return VH_OK;
    }
    VH_TRY(lin(a->attn, Cw, a->proj_w, Cw, a->proj_b, VH_ACT_NONE, a->ls1, a->x, a->x, a->n2_w, a->n2_b, a->hmid), "encoder proj");
    VH_TRY(lin(a->hmid, Cw, a->fc1_w, F, a->fc1_b, a->act, nullptr, nullptr, a->mid, nullptr, nullptr, nullptr), "encoder fc1");
    VH_TRY(lin(a->mid, F, a->fc2_w, Cw, a->fc2_b, VH_ACT_NONE, a->ls2, a->x, a->x, a->h_out ? a->next_w : nullptr, a->next_b,
               a->h_out), "encoder fc2");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VH_E_HIP, "vh_encoder_layer: %s", hipGetErrorString(e));
    return VH_OK;
}

int vh_mixtral_prefill(vh_mixtral_t* m, const float* embeds, int Sn, int pos0, float* logits_out, float* hidden_dbg,
                       void* stream) {
    if (!m || !embeds) return fail(VH_E_ARG, "vh_mixtral_prefill: null pointer");
    if (m->live_seqs() > 0) return fail(VH_E_ARG, "vh_mixtral_prefill: sequences own pages of the KV pool (free them or reset)");
    return prefill_impl(m, embeds, Sn, pos0, logits_out, hidden_dbg, S(stream));
}

// Final norm + LM head + greedy selection.  Vocab-sharded head: every rank scores its rows, the (max, index) candidates
// travel through the all-reduce hook, every rank takes the same global argmax; when scores are kept (logit_rows > 1)
// the full row is assembled by an all-reduce of the zero-filled slices.
static int head_and_select(vh_mixtral* m, hipStream_t st, const float* x_in, const float* delta, int mode, int set_pos,
                           const VhXchg* cx) {
    const bool sharded = m->c.vocab_n > 0 && (m->c.tp_world > 1 || (m->comm && vh_comm_is_loopback(m->comm)));
    VH_TRY(vhk_dec_lmhead(st, x_in, delta, m->final_norm, m->c.rms_eps, m->lm_head, m->Vn, m->H, m->logits, m->blk_val,
                          m->blk_idx, m->lm_grid, m->counters + 1, m->hist_rows(), m->v0, m->V, cx), "lm_head");
    int nblk = m->lm_grid;
    if (sharded) {
        const int cw = m->cand_world();
        VH_TRY(vhk_dec_cand(st, m->blk_val, m->blk_idx, m->lm_grid, m->cand, m->c.tp_rank, cw), "candidates");
        if (m->allreduce(m->cand, 2L * cw, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
        VH_TRY(vhk_dec_cand_unpack(st, m->cand, cw, m->blk_val, m->blk_idx), "candidates");
        nblk = cw;
    }
    VH_TRY(vhk_dec_select(st, m->blk_val, m->blk_idx, nblk, m->embed, m->H, m->V, m->xa, m->counters, m->counters + 1,
                          m->out_tokens, m->c.max_new, mode, set_pos), "select");
    return VH_OK;
}

int vh_mixtral_decode_schedule(const vh_mixtral_t* m) { return m ? m->schedule_state : -1; }

// Attention block of a decode layer as one fused launch (k_dec_ablk) or as three (QKV, attention, O projection): vh_tune("dec_fused").
static bool fused_wanted(const vh_mixtral* m) {
    const int want = vh_tuning()->dec_fused;        // -1 auto, 0 never, 1 wherever the kernel has an instantiation
    if (want == 0) return false;
    return vhk_dec_ablk_supported(m->H, m->nq, m->nkv) != 0;
}

// One decode step (all layers + LM head + token select) enqueued on st.  Returns VH_OK or an error code; the host
// mirror of the position (host_pos) is advanced by the CALLER only after the step was enqueued without error.
// Three launches per layer: the attention block (k_dec_ablk: fused QKV -> attention -> O projection; three launches with
// vh_tune("dec_fused", 0)), gate|up, down.
// Tensor parallel over the library's IPC transport: the two all-reduces of a layer are either FUSED into the kernels around
// them (VhXchg, vh_kernels.h: the O-projection items / the MoE down projection push their partial outputs straight into the
// peers' receive slots, the first blocks of the next launch — gate|up, the next layer's attention block, the LM head — sum the
// slots in rank order and every block of it waits for that sum behind its own weight loads: "all-reduce over xGMI overlapped
// with the expert GEMMs", web_demo/vllm_tools/vllm_file/mixtral.py:405-414,470-476; no launch per exchange) — the form the ranks
// vote for when each owns its device — or one small all-reduce kernel per exchange (ranks sharing a device: a waiting consumer
// grid would hold the CUs a peer's producer needs).
static int decode_one_step(vh_mixtral* m, hipStream_t st) {
    const int H = m->H, I = m->I, E = m->E, nq = m->nq, nkv = m->nkv, hd = m->hd;
    const float scale = 1.0f / sqrtf((float)hd);
    const float eps = m->c.rms_eps;
    const bool fuse = m->comm != nullptr && m->exchanges() && vh_tuning()->tp_fuse != 0 && (H % 2) == 0 &&
                      (size_t)H <= vh_comm_capacity(m->comm) && H <= 32768 && !vh_tuning()->force_allreduce;
    // Engine processes SHARING a device (ranks of the one-device tests, duplex replicas on one GPU) keep the three launches: a launch whose
    // blocks wait for other blocks is safe only while nobody else's waiting blocks can occupy the slots its producers need — blocks are
    // dispatched in index order PER XCD, so rank A's O blocks can fill one XCD while A's attention blocks queue on another behind rank B's
    // O blocks, which wait for B's attention blocks queued behind A's: r06 saw exactly this as an intermittent 0.5 s time-out with eight
    // ranks on one GPU.  A process that owns its device cannot starve itself (its own lower-index blocks are always dispatched first).
    const bool shared_dev = vh_tuning()->comm_ranks_per_device > 1 || (m->comm && vh_comm_ranks_per_device(m->comm) > 1);
    const bool fused_attn = fused_wanted(m) && !shared_dev;
    m->schedule_state = fused_attn ? 1 : 0;
    int* err = m->counters + 3;
    VhXchg xa{}, xm{};                           // attention / MoE exchange of the current layer
    bool have_xm = false;
    for (int l = 0; l < m->c.n_layers; ++l) {
        const vh_mixtral_layer& w = m->L[l];
        float* kc = m->kcache + (size_t)l * nkv * m->c.max_ctx * hd;
        float* vc = m->vcache + (size_t)l * nkv * m->c.max_ctx * hd;
        const bool prof = m->prof_stride > 0 && (l % m->prof_stride) == 0 && m->prof_used + 2 <= m->prof_ev.size();
        if (fuse && vh_comm_xchg_next(m->comm, H, 0, vhk_dec_consumer_blocks(1, 0, H, I), &xa, st) != VH_OK)
            return fail(VH_E_COMM, "fused exchange failed: %s", vh_comm_last_error());
        if (fused_attn) {
            VhDecAblk a{};
            a.x_in = m->xa; a.delta = l == 0 ? nullptr : m->delta_moe; a.x_out = m->xb; a.norm_w = w.attn_norm; a.eps = eps;
            a.Wqkv = w.wqkv; a.nqkv = m->nqkv; a.H = H;
            a.kcache = kc; a.vcache = vc; a.pos = m->host_pos; a.table = m->table; a.rope_cos = m->rope_cos; a.rope_sin = m->rope_sin;
            a.part_o = m->part_o; a.part_ml = m->part_ml; a.cnt = m->attn_cnt; a.nq = nq; a.nkv = nkv; a.max_ctx = m->c.max_ctx;
            a.max_splits = m->max_splits; a.nsplit = (m->host_pos + 1 + 63) / 64; a.scale = scale;
            a.Wo = w.wo; a.out = m->delta_attn;
            a.gq = VhGranVec{m->g_qkv, m->next_tag(), err}; a.ga = VhGranVec{m->g_attn, m->next_tag(), err};
            if (have_xm) a.cx = xm;
            if (fuse) a.px = xa;
            VH_TRY(vhk_dec_ablk(st, a), "dec attention block");
        } else {
            VH_TRY(vhk_dec_qkv(st, m->xa, l == 0 ? nullptr : m->delta_moe, m->xb, w.attn_norm, eps, w.wqkv, m->nqkv, H,
                               m->qkv, have_xm ? &xm : nullptr), "dec qkv");
            VH_TRY(vhk_dec_attn(st, m->qkv, kc, vc, m->counters, m->rope_cos, m->rope_sin, m->part_o, m->part_ml,
                                m->attn_cnt, m->attn_out, nq, nkv, m->c.max_ctx, m->max_splits, m->host_pos + 1,
                                scale, m->table), "dec attn");
            VH_TRY(vhk_dec_oproj(st, m->attn_out, w.wo, H, nq * hd, m->delta_attn, fuse ? &xa : nullptr), "dec oproj");
        }
        if (!fuse && m->allreduce(m->delta_attn, H, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
        // the consumer of the MoE exchange is the next layer's attention block (or fused-QKV GEMV), or the LM head after the last layer
        const int xm_blocks = l + 1 < m->c.n_layers ? (fused_attn ? vhk_dec_ablk_qkv_blocks(m->nqkv, H) : vhk_dec_consumer_blocks(0, m->nqkv, H, I))
                                                     : m->lm_grid;
        if (prof) (void)hipEventRecord(m->prof_ev[m->prof_used], st);
        {
            VH_TRY(vhk_dec_gateup(st, m->xb, m->delta_attn, m->xa, w.ffn_norm, eps, w.wrouter, E, w.w1, w.w3, I, H,
                                  m->route, m->hbuf, 0, fuse ? &xa : nullptr), "dec gateup");
            if (prof) { (void)hipEventRecord(m->prof_ev[m->prof_used + 1], st); m->prof_used += 2; }
            if (fuse) {
                if (vh_comm_xchg_next(m->comm, H, 1, xm_blocks, &xm, st) != VH_OK) return fail(VH_E_COMM, "fused exchange failed: %s", vh_comm_last_error());
                have_xm = true;
            }
            VH_TRY(vhk_dec_down(st, m->hbuf, m->route, w.w2, H, I, m->delta_moe, fuse ? &xm : nullptr), "dec down");
        }
        if (!fuse && m->allreduce(m->delta_moe, H, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
    }
    {
        const int rc = head_and_select(m, st, m->xa, m->delta_moe, /*mode=*/1, /*set_pos=*/0, have_xm ? &xm : nullptr);
        if (rc != VH_OK) return rc;
    }
    const hipError_t e = hipGetLastError();   // checked per step: the mirrors below must not run ahead of a failed launch
    if (e != hipSuccess) return fail(VH_E_HIP, "decode: %s", hipGetErrorString(e));
    return VH_OK;
}

int vh_mixtral_decode(vh_mixtral_t* m, int n_steps, void* stream) {
    if (!m) return fail(VH_E_ARG, "null engine");
    hipStream_t st = S(stream);
    if (m->poisoned) return fail(VH_E_ARG, "decode: a previous step failed; prefill or reset first");
    if (m->live_seqs() > 0) return fail(VH_E_ARG, "vh_mixtral_decode: sequences own pages of the KV pool (use vh_mixtral_seq_decode)");
    if (n_steps <= 0) return VH_OK;
    int rc = VH_OK;
    for (int step = 0; step < n_steps; ++step) {
        if (m->host_pos + 1 >= m->c.max_ctx) { rc = fail(VH_E_SHAPE, "decode: KV cache full (%d)", m->c.max_ctx); break; }
        rc = decode_one_step(m, st);
        if (rc != VH_OK) {
            // the step was not (fully) enqueued: host_pos keeps its value, i.e. it describes the
            // last COMPLETE step; the device state of the partial step is discarded by the next prefill / reset
            m->poisoned = 1;
            break;
        }
        m->host_pos += 1;
    }
    return rc;
}

// ---- concurrent sequences (paged KV cache) ---------------------------------------------------------------------------
static int seq_ok(vh_mixtral* m, int s, const char* what) {
    if (!m) return fail(VH_E_ARG, "%s: null engine", what);
    if (s < 0 || s >= (int)m->seqs.size() || !m->seqs[s].live) return fail(VH_E_ARG, "%s: sequence %d is not allocated", what, s);
    return VH_OK;
}

int vh_mixtral_seq_alloc(vh_mixtral_t* m) {
    if (!m) return fail(VH_E_ARG, "null engine");
    for (int s = 0; s < (int)m->seqs.size(); ++s)
        if (!m->seqs[s].live) { m->seqs[s].live = true; return s; }
    return fail(VH_E_FULL, "vh_mixtral_seq_alloc: all %d sequence slots are in use", (int)m->seqs.size());
}

int vh_mixtral_seq_free(vh_mixtral_t* m, int s) {
    const int rc = seq_ok(m, s, "vh_mixtral_seq_free");
    if (rc != VH_OK) return rc;
    m->release(s);
    return VH_OK;
}

int vh_mixtral_pages_free(const vh_mixtral_t* m) { return m ? (int)m->free_pages.size() : 0; }
int vh_mixtral_seq_pos(const vh_mixtral_t* m, int s) {
    return (m && s >= 0 && s < (int)m->seqs.size() && m->seqs[s].live) ? m->seqs[s].host_pos : -1;
}
const int* vh_mixtral_seq_tokens(const vh_mixtral_t* m, int s) {
    return (m && s >= 0 && s < (int)m->seqs.size()) ? m->seq_tokens + (size_t)s * (m->c.max_new > 0 ? m->c.max_new : 1) : nullptr;
}
const int* vh_mixtral_seq_counters(const vh_mixtral_t* m, int s) {
    return (m && s >= 0 && s < (int)m->seqs.size()) ? m->seq_counters + 4 * s : nullptr;
}
int vh_mixtral_seq_table(const vh_mixtral_t* m, int s, int* pages_out, int cap) {
    if (!m || s < 0 || s >= (int)m->seqs.size() || !m->seqs[s].live) return -1;
    const vh_mixtral::Seq& q = m->seqs[s];
    for (int j = 0; j < q.npages && j < cap; ++j) pages_out[j] = q.pages[j];
    return q.npages;
}

int vh_mixtral_seq_prefill(vh_mixtral_t* m, int s, const float* embeds, int Sn, float* logits_out, void* stream) {
    const int rc0 = seq_ok(m, s, "vh_mixtral_seq_prefill");
    if (rc0 != VH_OK) return rc0;
    if (!embeds) return fail(VH_E_ARG, "vh_mixtral_seq_prefill: null pointer");
    hipStream_t st = S(stream);
    const int pos0 = m->seqs[s].host_pos;       // appends to the sequence (0 for a new one): chunked prompts, later turns
    if (Sn < 1 || pos0 + Sn >= m->c.max_ctx) return fail(VH_E_SHAPE, "prefill of %d tokens at %d exceeds the pool", Sn, pos0);
    const int pr = m->ensure_pages(s, pos0 + Sn, st);
    if (pr == -1) return fail(VH_E_FULL, "KV pool exhausted: %d pages free, sequence %d needs %d more", (int)m->free_pages.size(),
                              s, (pos0 + Sn + 63) / 64 - m->seqs[s].npages);
    if (pr != 0) return fail(VH_E_HIP, "page table upload failed");
    m->bind(s);
    const int rc = prefill_impl(m, embeds, Sn, pos0, logits_out, nullptr, st);
    if (rc != VH_OK) m->poisoned = 1;
    m->unbind();
    return rc;
}

// One decode ITERATION of n concurrent sequences (n <= max_seqs, all distinct).  The attention side (fused QKV GEMV,
// split-KV attention, O projection) and the LM head run in groups of up to VH_BMAX sequences per launch (one pass over
// the shared weights per group).  The MoE of a layer runs
//   n >= batch_moe_min (default 3): ONCE for the whole iteration on the prefill's weight-streaming GEMM with S = n — rows
//       sorted by expert, every TOUCHED expert streamed exactly once at ~5 TB/s however many sequences picked it
//       (k_gemm_ps cfg 0, <= 64 rows per tile — with its few-row rules: vh_gemm_ps_inl.h K-split estimator, vh_gemm_ps.hip cache hint;
//       "continuous batching");
//   otherwise per sequence with the batch-1 GEMV kernels.
// Sequence state is addressed in place (slots); the MoE delta of a sequence is a row of the shared `ptmp` buffer.
static int decode_iteration(vh_mixtral* m, hipStream_t st, const int* ids, int n) {
    const int H = m->H, I = m->I, E = m->E, nq = m->nq, nkv = m->nkv, hd = m->hd;
    const float scale = 1.0f / sqrtf((float)hd);
    const float eps = m->c.rms_eps;
    const bool tp = m->c.tp_world > 1 || vh_tuning()->force_allreduce;
    std::vector<float*> xa(n), xb(n), da(n), dm(n);
    std::vector<int*> cnt(n);
    for (int b = 0; b < n; ++b) {
        float* x = m->seq_x + (size_t)ids[b] * 4 * H;
        xa[b] = x; xb[b] = x + H; da[b] = x + 2 * H; dm[b] = x + 3 * H;
        cnt[b] = m->seq_counters + 4 * ids[b];
    }
    const int moe_min = vh_tuning()->batch_moe_min;
    const bool stream_moe = moe_min > 0 && n >= moe_min && n <= m->c.max_prefill && H <= 4096 && (H % 64) == 0 && (I % 64) == 0;
    if (stream_moe) {
        // the slot list of the iteration lives on the device for k_gather_rows; it is re-uploaded only when the running
        // set changed (a pageable-memory upload waits for the stream: doing it every iteration would stop the host from
        // enqueueing ahead of the GPU).  The source is a member whose storage never moves (reserved at creation).
        if ((int)m->batch_ids_host.size() != n || memcmp(m->batch_ids_host.data(), ids, (size_t)n * sizeof(int)) != 0) {
            m->batch_ids_host.assign(ids, ids + n);
            if (hipMemcpyAsync(m->seq_batch, m->batch_ids_host.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess)
                return fail(VH_E_HIP, "batch id upload failed");
        }
        for (int b = 0; b < n; ++b) dm[b] = m->ptmp + (size_t)b * H;       // MoE delta rows of this iteration
    }
    for (int l = 0; l < m->c.n_layers; ++l) {
        const vh_mixtral_layer& w = m->L[l];
        float* kc = m->kcache + (size_t)l * nkv * m->c.max_ctx * hd;
        float* vc = m->vcache + (size_t)l * nkv * m->c.max_ctx * hd;
        for (int g0 = 0; g0 < n; g0 += VH_BMAX) {
            const int gn = n - g0 < VH_BMAX ? n - g0 : VH_BMAX;
            VhDecBatchVec q{};
            VhDecBatchAttn at{};
            VhDecBatchVec o{};
            q.n = gn; o.n = gn;
            for (int b = 0; b < gn; ++b) {
                const int sb = g0 + b;
                q.x_in[b] = xa[sb]; q.delta[b] = l == 0 ? nullptr : dm[sb]; q.x_out[b] = xb[sb]; q.out[b] = m->l_qkv[b];
                at.qkv[b] = m->l_qkv[b]; at.pos[b] = m->seqs[ids[sb]].host_pos;
                at.table[b] = m->seq_table + (size_t)ids[sb] * m->max_splits;
                at.part_o[b] = m->l_part_o[b]; at.part_ml[b] = m->l_part_ml[b]; at.cnt[b] = m->l_attn_cnt[b];
                at.attn_out[b] = m->l_attn_out[b];
                o.x_in[b] = m->l_attn_out[b]; o.out[b] = da[sb];
            }
            VH_TRY(vhk_decb_gemv(st, q, w.attn_norm, eps, w.wqkv, m->nqkv, H, 1), "batched qkv");
            VH_TRY(vhk_decb_attn(st, at, gn, kc, vc, m->rope_cos, m->rope_sin, nq, nkv, m->c.max_ctx, m->max_splits, scale), "batched attn");
            VH_TRY(vhk_decb_gemv(st, o, nullptr, 0.f, w.wo, H, nq * hd, 0), "batched oproj");
        }
        for (int b = 0; b < n; ++b)
            if (m->allreduce(da[b], H, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
        if (stream_moe) {
            // x = xb + delta_attn for every sequence -> rows of px (and each sequence's xa, as the GEMV path's x_out)
            VH_TRY(vhk_gather_rows(st, m->seq_x, m->seq_batch, n, H, m->px), "gather");
            VH_TRY(vhk_rmsnorm_route(st, m->px, nullptr, m->pxn_hi, m->pxn_lo, w.ffn_norm, n, H, eps, w.wrouter, E, m->pids,
                                     m->pwts), "rmsnorm + route");
            VH_TRY(vhk_moe_sort(st, m->pids, n, E, m->pgoff, m->pstok, m->psslot), "sort");
            VhGemmPsArgs g{};
            g.A_hi = m->pxn_hi; g.A_lo = m->pxn_lo; g.lda = H; g.a_rowidx = m->pstok;
            g.W = w.w1; g.W_up = w.w3; g.ldw = H; g.w_group_stride = (long)I * H;
            g.group_off = m->pgoff; g.ngroups = E;
            g.C_hi = m->ph_hi; g.C_lo = m->ph_lo; g.ldc_split = I; g.M = 2 * n; g.N = I; g.K = H; g.ksplit = 1;
            VH_TRY(vhk_gemm_ps(st, g), "gate/up gemm");
            int nslab = vh_tuning()->moe_ksplit;
            if (nslab == 0) nslab = 1;
            if (nslab > 4) nslab = 4;
            if (nslab < -4) nslab = -4;
            if (nslab > (I >> 6)) nslab = 1;
            const long slab = (long)2 * m->c.max_prefill * H;
            VhGemmPsArgs d{};
            d.A_hi = m->ph_hi; d.A_lo = m->ph_lo; d.lda = I;
            d.W = w.w2; d.ldw = I; d.w_group_stride = (long)H * I;
            d.group_off = m->pgoff; d.ngroups = E;
            d.C = m->py; d.ldc = H; d.c_rowidx = m->psslot; d.M = 2 * n; d.N = H; d.K = I;
            d.ksplit = nslab; d.c_split_stride = slab; d.nslab_out = m->pnslab;
            VH_TRY(vhk_gemm_ps(st, d), "down gemm");
            if (hipMemsetAsync(m->ptmp, 0, (size_t)n * H * sizeof(float), st) != hipSuccess) return fail(VH_E_HIP, "memset failed");
            VH_TRY(vhk_moe_combine(st, m->ptmp, m->py, m->pwts, n, H, nslab < 0 ? 1 : nslab, slab, nslab < 0 ? m->pnslab : nullptr), "combine");
            if (tp && m->allreduce(m->ptmp, (long)n * H, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
        } else {
            for (int g0 = 0; g0 < n; g0 += VH_BMAX) {
                const int gn = n - g0 < VH_BMAX ? n - g0 : VH_BMAX;
                    for (int b = 0; b < gn; ++b) {
                        VH_TRY(vhk_dec_gateup(st, xb[g0 + b], da[g0 + b], xa[g0 + b], w.ffn_norm, eps, w.wrouter, E, w.w1, w.w3, I, H,
                                              m->l_route[b], m->l_hbuf[b], 0), "dec gateup");
                        VH_TRY(vhk_dec_down(st, m->l_hbuf[b], m->l_route[b], w.w2, H, I, dm[g0 + b]), "dec down");
                    }
            }
            for (int b = 0; b < n; ++b)
                if (m->allreduce(dm[b], H, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
        }
    }
    const bool sharded = m->c.vocab_n > 0 && m->c.tp_world > 1;
    for (int g0 = 0; g0 < n; g0 += VH_BMAX) {
        const int gn = n - g0 < VH_BMAX ? n - g0 : VH_BMAX;
        VhDecBatchVec hv{};
        VhDecBatchHead hdp{};
        hv.n = gn;
        for (int b = 0; b < gn; ++b) {
            hv.x_in[b] = xa[g0 + b]; hv.delta[b] = dm[g0 + b];
            hdp.blk_val[b] = m->l_blk_val[b]; hdp.blk_idx[b] = m->l_blk_idx[b];
        }
        if (gn >= 2) {   // one pass over the table for the group: 97 us (four activation rows, always) against n x 70
            VH_TRY(vhk_decb_lmhead(st, hv, m->final_norm, eps, m->lm_head, m->Vn, H, hdp, m->lm_grid, m->v0), "batched lm_head");
        } else {
            VH_TRY(vhk_dec_lmhead(st, xa[g0], dm[g0], m->final_norm, eps, m->lm_head, m->Vn, H, m->logits, m->l_blk_val[0],
                                  m->l_blk_idx[0], m->lm_grid, cnt[g0] + 1, 1, m->v0, m->V), "lm_head");
        }
        for (int b = 0; b < gn; ++b) {
            const int sb = g0 + b;
            int nblk = m->lm_grid;
            if (sharded) {
                VH_TRY(vhk_dec_cand(st, m->l_blk_val[b], m->l_blk_idx[b], m->lm_grid, m->l_cand[b], m->c.tp_rank, m->c.tp_world), "candidates");
                if (m->allreduce(m->l_cand[b], 2L * m->c.tp_world, st) != 0) return fail(VH_E_COMM, "all-reduce failed");
                VH_TRY(vhk_dec_cand_unpack(st, m->l_cand[b], m->c.tp_world, m->l_blk_val[b], m->l_blk_idx[b]), "candidates");
                nblk = m->c.tp_world;
            }
            VH_TRY(vhk_dec_select(st, m->l_blk_val[b], m->l_blk_idx[b], nblk, m->embed, H, m->V, xa[sb], cnt[sb], cnt[sb] + 1,
                                  m->seq_tokens + (size_t)ids[sb] * (m->c.max_new > 0 ? m->c.max_new : 1), m->c.max_new, 1, 0), "select");
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VH_E_HIP, "batched decode: %s", hipGetErrorString(e));
    return VH_OK;
}

int vh_mixtral_seq_decode(vh_mixtral_t* m, const int* ids, int n, void* stream) {
    if (!m || (!ids && n > 0)) return fail(VH_E_ARG, "vh_mixtral_seq_decode: null pointer");
    hipStream_t st = S(stream);
    for (int i = 0; i < n; ++i) {               // validate the whole batch before anything is enqueued
        const int rc = seq_ok(m, ids[i], "vh_mixtral_seq_decode");
        if (rc != VH_OK) return rc;
        const vh_mixtral::Seq& q = m->seqs[ids[i]];
        if (q.poisoned) return fail(VH_E_ARG, "sequence %d: a previous step failed; free it", ids[i]);
        if (q.host_pos < 1) return fail(VH_E_ARG, "sequence %d has no prompt yet", ids[i]);
        if (q.host_pos + 1 >= m->c.max_ctx) return fail(VH_E_SHAPE, "sequence %d: context limit %d", ids[i], m->c.max_ctx);
    }
    // distinct sequences advance together through the batched iteration (a sequence listed twice in one call advances
    // twice, one step after the other, as before)
    if (n >= 2 && vh_tuning()->batch_decode != 0 && m->H <= 4096 && m->nq * m->hd <= 4096) {
        bool distinct = true;
        for (int i = 0; i < n && distinct; ++i)
            for (int j = 0; j < i; ++j) if (ids[i] == ids[j]) { distinct = false; break; }
        if (distinct) {
            for (int i = 0; i < n; ++i) {
                const int pr = m->ensure_pages(ids[i], m->seqs[ids[i]].host_pos + 1, st);
                if (pr == -1) return fail(VH_E_FULL, "KV pool exhausted at sequence %d (nothing of the batch advanced)", ids[i]);
                if (pr != 0) return fail(VH_E_HIP, "page table upload failed");
            }
            const int rc = decode_iteration(m, st, ids, n);
            if (rc != VH_OK) {
                for (int i = 0; i < n; ++i) m->seqs[ids[i]].poisoned = 1;
                return rc;
            }
            for (int i = 0; i < n; ++i) m->seqs[ids[i]].host_pos += 1;
            return VH_OK;
        }
    }
    if (n <= 0) return VH_OK;
    int rc = VH_OK;
    for (int i = 0; i < n; ++i) {
        const int s = ids[i];
        const int pr = m->ensure_pages(s, m->seqs[s].host_pos + 1, st);   // the step writes K/V of position host_pos
        if (pr == -1) { rc = fail(VH_E_FULL, "KV pool exhausted at sequence %d (%d of the batch advanced)", s, i); break; }
        if (pr != 0) { rc = fail(VH_E_HIP, "page table upload failed"); break; }
        m->bind(s);
        rc = decode_one_step(m, st);
        if (rc != VH_OK) m->poisoned = 1;
        else m->host_pos += 1;
        m->unbind();
        if (rc != VH_OK) break;
    }
    return rc;
}

}  // extern "C"
