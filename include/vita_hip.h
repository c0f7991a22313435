/* vita_hip.h — C ABI of libvita_hip.so, the MI355X (gfx950) hot path of VITA inference.
 *
 * The reference (VITA-MLLM/VITA @ 2024-10-22) is 100 % Python and has no FFI of its own:
 * its drop-in boundary is the `vita.model` Python API (vita/model/builder.py:14-24,306;
 * vita/model/language_model/vita_mixtral.py:249-415; vita/model/multimodal_encoder/builder.py:12,44;
 * vita/model/multimodal_projector/builder.py:154).  This header is the boundary we add
 * beneath it: the entry points a maintainer binds with ctypes from those Python classes
 * (see INTEGRATION.md).  Each function names the reference computation it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless marked host;
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it;
 *   - activations are fp32, large weights bf16 (uint16 storage), small vectors fp32;
 *   - return 0 on success, VH_E_* (<0) on error; vh_last_error() describes the last one;
 *   - no entry point allocates device memory; engines take a caller-provided workspace
 *     whose size comes from the matching *_workspace_bytes().
 */
#ifndef VITA_HIP_H
#define VITA_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VH_OK 0
#define VH_E_SHAPE (-1)
#define VH_E_ARG (-2)
#define VH_E_HIP (-3)
#define VH_E_COMM (-4)
#define VH_E_FULL (-5)   /* a pool is exhausted (sequence slots, KV pages): retry after freeing */

#define VH_ACT_NONE 0
#define VH_ACT_GELU 1 /* erf GELU (nn.GELU default) */
#define VH_ACT_RELU 2
#define VH_ACT_SILU 3

int vh_version(void);
const char* vh_last_error(void);
/* Kernel-variant knobs (21 keys: "batch_moe_min", "batch_decode", "attn_impl", "attn_fa", "attn_rows", "attn_ksplit",
 * "prefill_attn_gemm", "prefill_fuse_rows", "ps_cfg", "ps_nt", "tp_overlap", "moe_ksplit", "force_allreduce", "tp_fuse",
 * "comm_allow_coarse", "comm_ranks_per_device", "dec_fused", "dec_gateup_grid", "attn_img", "attn_xcd", "ps_xcd"; vita_amd/csrc/vh_kernels.h: VhTuning says what each selects); the defaults are the
 * measured-best variants, ids are identical across variants.  Unknown keys (e.g. of variants removed in r04) return VH_E_ARG. */
int vh_tune(const char* key, int value);

/* ---- generic operators --------------------------------------------------------------
 * vh_gemm: C = epilogue(A W^T).  Replaces every nn.Linear / conv-as-GEMM on the path:
 * InternViT qkv/proj/fc1/fc2 (internvit/modeling_intern_vit.py:144,175,213-217), the
 * mlp2x_gelu projector (multimodal_projector/builder.py:164-168), Whale linears and convs
 * (whale/module/component/subsampling.py:28-43, module/layer/attention.py:145-147,370-419,
 * adapter.py:93-136) and the Mixtral prefill projections + expert GEMMs (HF
 * modeling_mixtral.py MixtralAttention / MixtralExperts).
 *   A fp32 [*, lda]; logical row m reads source row a_rowidx[m] (or m) + segrow[k/seglen],
 *   column k%seglen; rows outside [0,a_rows) read as zero.  K = nseg*seglen, K%64==0.
 *   W bf16 [N][ldw]; W_up != NULL selects the gated form silu(A W^T) * (A W_up^T).
 *   group_off != NULL: grouped GEMM over `ngroups` row ranges, W advancing w_group_stride.
 *   epilogue: (+bias[n]) -> act -> (*scale[n]) -> (+resid[orow,n]) -> C[orow,n],
 *   orow = c_rowidx[m] or m.                                                            */
typedef struct {
    const float* A; long lda; int a_rows; const int* a_rowidx;
    int nseg; int seglen; int segrow[16];
    const uint16_t* W; const uint16_t* W_up; long ldw; long w_group_stride;
    const int* group_off; int ngroups;
    float* C; long ldc; const int* c_rowidx;
    const float* bias; const float* scale; const float* resid; long ldr;
    int M, N, K, act;
    float* ws; size_t ws_bytes;   /* optional scratch: lets small launches split K over more blocks (partial sums, then one
                                     reducing kernel that applies the epilogue); results differ from the unsplit launch
                                     only by fp32 summation order */
    int ksplit;                   /* 0 = decided by the library when ws != NULL, 1 = never split, n = n-way */
} vh_gemm_args;
int vh_gemm(const vh_gemm_args* args, void* stream);
/* vh_gemm_ln: vh_gemm followed by the LayerNorm that consumes its output, in one call:
 *   C = epilogue(A W^T)  and  ln_out[m, :] = LayerNorm(C[m, :]; ln_eps) * ln_w + ln_b      (ln_b nullable)
 * i.e. the encoders' `x = x + ls * proj(attn); h = norm2(x)` (InternVisionEncoderLayer.forward, modeling_intern_vit.py:245-253)
 * and Whale's `x = residual + feed_forward(x); x = norm(x)` steps.  When the library splits K (small launches) the
 * reducer holds whole rows and applies the norm itself; otherwise a norm launch follows.  Plain GEMMs only (no gate,
 * groups or output row map); N % 4 == 0. */
int vh_gemm_ln(const vh_gemm_args* args, const float* ln_w, const float* ln_b, float ln_eps, float* ln_out, long ld_ln,
               void* stream);

/* vh_gemm_ps: the same contraction as a WEIGHT STREAM for skinny M (Mixtral prefill, MoE grouped GEMMs: HF
 * MixtralExperts, modeling_mixtral.py:57-93) on activations already split into bf16 hi/lo planes (x = hi + lo
 * to 2^-17; vh_split_planes makes them, vh_gemm_ps can also emit them).  One m-tile covers all rows of a group
 * (up to 160 / 192; larger groups are cut into balanced m-tiles) so each weight byte is read once; K % 64 == 0,
 * lda % 8 == 0.  ksplit > 1 (plain fp32 output only): K is cut into ksplit ranges whose partial sums are
 * written to C + ks * c_split_stride (the caller adds the slabs).  `wide` is ignored (kept for ABI stability). */
typedef struct {
    const uint16_t* A_hi; const uint16_t* A_lo; long lda; const int* a_rowidx;
    const uint16_t* W; const uint16_t* W_up; long ldw; long w_group_stride;
    const int* group_off; int ngroups;
    float* C; long ldc; uint16_t* C_hi; uint16_t* C_lo; long ldc_split; const int* c_rowidx;
    const float* bias; const float* scale; const float* resid; long ldr;
    int M, N, K, act, wide;
    int ksplit; long c_split_stride;
    int* nslab_out;       /* ksplit < 0: the kernel picks the split (1 .. -ksplit) from the group sizes and stores it here (device int) */
} vh_gemm_ps_args;
int vh_gemm_ps(const vh_gemm_ps_args* args, void* stream);
int vh_split_planes(const float* x, long ldx, uint16_t* hi, uint16_t* lo, long ldo, int rows, int cols, void* stream);

/* vh_attention: softmax(scale * Q K^T [+ (Q+v) P^T]) V with causal / pad / chunk masks.
 * Replaces InternAttention._naive_attn (modeling_intern_vit.py:158-177), Whale
 * MultiHeadedAttention.forward score/softmax/PV (attention.py:380-416) and HF
 * eager_attention_forward for the Mixtral prefill.  head_dim 64 or 128.               */
typedef struct {
    const float* Q; long ldq; long hsq;
    const float* K; long ldk; long hsk;
    const float* V; long ldv; long hsv;
    const float* P; long ldp; long hsp;
    const float* bias_u; const float* bias_v;
    float* O; long ldo;
    long bsq, bsk, bso;
    int B, Hq, Hkv, Sq, Sk, d;
    int causal, q_off, klen, chunk, left;
    float scale;
} vh_attn_args;
int vh_attention(const vh_attn_args* args, void* stream);

/* vh_encoder_layer: ONE pre-norm transformer block of the two encoders, all launches from a single call (the host loop of
 * 24 layers x 8 operator calls was slower than the GPU at one tile / one clip):
 *   qkv = h_in W_qkv^T + b;  a = attention(qkv);  x += ls1 * (a W_proj^T + b);  h = LN(x; n2);
 *   x += ls2 * (act(h W_fc1^T + b) W_fc2^T + b);  h_out = LN(x; next)                     (ls1 / ls2 / next nullable)
 * = InternVisionEncoderLayer.forward (internvit/modeling_intern_vit.py:245-253: layer scale, GELU) and Whale's
 * TransformerEncoderLayer.forward (whale/module/encoder/transformer.py, rel-pos attention whale/module/layer/attention.py:358-419:
 * P = linear_pos(pos_emb), pos_bias_u / pos_bias_v, pad / chunk mask; ReLU).  h_in is LN(x; this layer's norm1): from
 * vh_layernorm for the first layer, from the previous call's h_out afterwards.  Attention runs inside each of the B
 * sequences of M / B rows; keys >= klen are masked (klen < 0: none).  Scratch (caller-owned): qkv [M, 3C], attn [M, C], hmid [M, C], mid [M, F], ws (split-K slabs,
 * >= 32 M max(C, F) bytes recommended; 0 disables the split).
 * planes = 1 (r04; no rel-pos, ws >= 4 M C bytes): the rows that only feed the next Linear travel as the bf16 hi/lo planes vh_gemm_ps consumes — h_in / h_out,
 * attn and hmid then hold two planes each, hi [M][C] followed by lo [M][C] (mid: [M][F] twice), in the same M C 4 (M F 4) bytes — and the four Linears run on the
 * weight-streaming GEMM with one-round tilings (qkv / fc1 one pass; proj / fc2 K-split into ws, summed by the reducer that also applies bias, layer scale,
 * residual and the next LayerNorm).  Same arithmetic as planes = 0 (the general GEMM splits its fp32 operand into the same planes per block). */
typedef struct {
    float* x; const float* h_in; float* h_out;
    int M, C, F, heads, B;
    const uint16_t* qkv_w; const float* qkv_b;
    const uint16_t* proj_w; const float* proj_b; const float* ls1;
    const float* n2_w; const float* n2_b;
    const uint16_t* fc1_w; const float* fc1_b;
    const uint16_t* fc2_w; const float* fc2_b; const float* ls2;
    const float* next_w; const float* next_b;
    int act; float eps;
    const float* P; long ldp; const float* bias_u; const float* bias_v; int klen, chunk, left;
    float* qkv; float* attn; float* hmid; float* mid; float* ws; size_t ws_bytes;
    int planes;
} vh_encoder_layer_args;
int vh_encoder_layer(const vh_encoder_layer_args* args, void* stream);

/* nn.LayerNorm over the last dim (+ optional act, then * post_scale). */
int vh_layernorm(const float* x, long ldx, float* y, long ldy, const float* w, const float* b, int rows, int cols,
                 float eps, int act, float post_scale, void* stream);
/* MixtralRMSNorm (HF modeling_mixtral.py:134-148). */
int vh_rmsnorm(const float* x, float* y, const float* w, int rows, int cols, float eps, void* stream);
int vh_add(float* x, const float* y, long n, void* stream);
int vh_cast_bf16_f32(const uint16_t* in, float* out, long n, void* stream);
/* Synthetic bf16 weights from a counter-based generator whose values are exact in bf16 (integers in [-126, 126] x 2^-11,
 * sigma 0.018): element (r, c) of the [rows, cols] view is sample idx0 + r * ld_src + c of stream `seed`, so shards of a
 * logical tensor hold the full tensor's values.  The host twin is oracle/hashfill.c — identical weights on both sides
 * at any size (no checkpoint exists offline: SURVEY 8(d)). */
int vh_fill_hash_bf16(uint16_t* dst, long rows, long cols, long ld_dst, long ld_src, long idx0, uint64_t seed, void* stream);

/* InternVisionEmbeddings + the first block's norm1 in ONE call (modeling_intern_vit.py:68-122, :243): patchify -> patch Linear
 * (+ bias) -> [cls | patches] + position table -> LayerNorm (-> bf16 hi / lo planes for vh_encoder_layer's planes mode).  Five
 * operator calls from Python left 177 us of host gaps in front of the first block of every tower pass (r04 trace); the pieces
 * stay available one by one below.  pos: the table ALREADY at this tile grid (the bicubic interpolation of the checkpoint's
 * table depends on the geometry only; the host does it once at load).  Scratch and outputs are caller-owned. */
typedef struct {
    const float* pix; int n, img, patch, kpad;            /* [n][3][img][img] fp32; kpad = padded 3 * patch^2 (multiple of 64) */
    const uint16_t* patch_w; const float* patch_b;         /* bf16 [C][kpad], fp32 [C] */
    const uint16_t* cls; const uint16_t* pos;              /* bf16 [C], bf16 [ntok][C] */
    const float* ln_w; const float* ln_b; float eps;
    int ntok, C;
    float* patches; float* pe;                             /* scratch: [n (img/patch)^2][kpad], [n (img/patch)^2][C] */
    float* x;                                              /* out: residual stream [n ntok][C] */
    float* h;                                              /* out: LayerNorm(x) fp32 [n ntok][C] */
    void* h_planes;                                        /* nullable out: the same rows as bf16 hi [n ntok][C] then lo [n ntok][C] */
    float* ws; size_t ws_bytes;                            /* nullable split-K scratch of the patch Linear */
} vh_vit_embed_args;
int vh_vit_embed(const vh_vit_embed_args* args, void* stream);

/* InternViT front/back (modeling_intern_vit.py:68-122; internvit_encoder.py:35-79). */
int vh_vit_patchify(const float* pix, float* out, int n, int img, int patch, int kpad, void* stream);
int vh_vit_assemble(const float* patches, const uint16_t* cls, const uint16_t* pos, float* x, int n, int ntok,
                    int hid, void* stream);
int vh_vit_pixel_shuffle(const float* x, float* out, int n, int grid, int hid, float mul, void* stream);

/* Whale GlobalCMVN + first Conv2d + ReLU, channels-last (cmvn.py:29-32; subsampling.py:28-31). */
int vh_audio_conv1(const float* feats, const float* mean, const float* istd, const uint16_t* w, const float* b,
                   float* out, int T, int F, int C, void* stream);

/* embed_tokens gather + image/audio splice (vita_arch.py:237-321); kind 0 text,1 image,2 audio. */
int vh_embed_splice(const int* src_kind, const int* src_idx, const uint16_t* embed, const float* img_feats,
                    const float* aud_feats, float* out, int S, int H, void* stream);

/* ---- Mixtral engine: prefill + greedy decode (HF MixtralModel/GenerationMixin as driven by
 * vita_mixtral.py:101-215,291-382 and video_audio_demo.py:257-270) ---------------------- */
typedef struct {
    int hidden, n_layers, n_q_heads, n_kv_heads, head_dim, inter, n_experts, top_k, vocab;
    float rms_eps;
    int max_ctx;          /* KV-cache capacity in tokens */
    int max_prefill;      /* largest S accepted by vh_mixtral_prefill */
    int max_new;          /* capacity of the generated-token buffer */
    int tp_rank, tp_world;/* n_q_heads / n_kv_heads / inter above are THIS RANK's slices */
    int nsplit;           /* decode split-KV factor, 0 = auto */
    int logit_rows;       /* >1: keep the fp32 logits of the first logit_rows generated tokens */
    int vocab_lo, vocab_n;/* vocab-sharded LM head (ParallelLMHead, vllm_file/mixtral.py:939-951): `lm_head` holds rows
                             [vocab_lo, vocab_lo + vocab_n) of the table; the ranks exchange (max, index) candidates
                             through the all-reduce hook.  vocab_n == 0: the whole table on every rank. */
    int max_seqs;         /* > 0: slots for concurrent sequences over the paged KV cache (vh_mixtral_seq_*); max_ctx is then
                             the POOL size in tokens (a multiple of 64) shared by all sequences */
} vh_mixtral_cfg;

typedef struct {
    const float* attn_norm;    /* [hidden] fp32 */
    const uint16_t* wqkv;      /* [(nq+2nkv)*head_dim][hidden]  q rows, then k rows, then v rows */
    const uint16_t* wo;        /* [hidden][nq*head_dim] */
    const float* ffn_norm;     /* [hidden] */
    const uint16_t* wrouter;   /* [n_experts][hidden]   block_sparse_moe.gate */
    const uint16_t* w1;        /* [n_experts][inter][hidden]  gate proj (experts.N.w1) */
    const uint16_t* w3;        /* [n_experts][inter][hidden]  up proj   (experts.N.w3) */
    const uint16_t* w2;        /* [n_experts][hidden][inter]  down proj (experts.N.w2) */
} vh_mixtral_layer;

typedef struct vh_mixtral vh_mixtral_t;

size_t vh_mixtral_workspace_bytes(const vh_mixtral_cfg* cfg);
/* `layers` is a host array of cfg->n_layers entries (copied). rope tables: fp32 [max_ctx][head_dim/2]. */
vh_mixtral_t* vh_mixtral_create(const vh_mixtral_cfg* cfg, const vh_mixtral_layer* layers, const uint16_t* embed,
                                const float* final_norm, const uint16_t* lm_head, const float* rope_cos,
                                const float* rope_sin, void* workspace, size_t workspace_bytes);
void vh_mixtral_destroy(vh_mixtral_t* m);

/* Tensor-parallel all-reduce hook (sum, fp32, in place).  Either the built-in RCCL path or a
 * caller-supplied callback (e.g. torch.distributed).  Unused when tp_world == 1. */
typedef int (*vh_allreduce_fn)(void* user, float* buf, long count, void* stream);
int vh_mixtral_set_allreduce(vh_mixtral_t* m, vh_allreduce_fn fn, void* user);
int vh_rccl_unique_id(void* out_128_bytes /* host */);
int vh_mixtral_init_rccl(vh_mixtral_t* m, const void* unique_id_128_bytes /* host */);

/* Prefill S embedded tokens at positions [pos0, pos0+S); leaves the first generated token in
 * out_tokens[0] (greedy) and the engine ready to decode.  If logits_out != NULL the fp32
 * logits of the last position are copied there ([vocab]).  hidden_dbg (nullable): fp32
 * [n_layers][S][hidden] receives the residual stream after every layer.                   */
/* ---- the library's own all-reduce over IPC-mapped peer buffers (vh_comm.hip): one process per GPU ------------------
 * vh_comm_create allocates this rank's receive buffer (the ONLY allocation the library makes: it must be fine-grained
 * device memory) and writes its 64-byte IPC handle to handle_out; the caller gathers the world's handles (any bootstrap:
 * torch.distributed, MPI, a file) and passes them, rank-major, to vh_comm_connect.  vh_comm_allreduce sums `count` fp32
 * values in place across the ranks (every rank must call it, in the same order, with the same count); results are
 * bit-identical on all ranks.  vh_comm_status: 0, or the phase whose bounded spin timed out (sticky: once set, every
 * later poll gives up at once).  vh_comm_create FAILS when fine-grained memory cannot be allocated, unless the caller
 * declared that all ranks share one device (vh_tune("comm_allow_coarse", 1): same-device tests); it reads
 * vh_tune("comm_ranks_per_device", n) — how many ranks drive this rank's device (1 on a node with a GPU per rank).  The 32-bit
 * granule tag wraps every 2^32 calls; the library then re-zeroes its regions between two barriers (collective, same call on all ranks).
 * vh_comm_create_loopback: a connected single-process communicator in which `rank` plays all `world` ranks into its own receive slots
 * (the stores, polls, tags and rank-ordered sums of a real exchange without a link; sums equal the inputs) — what
 * `bench.py --emulate-tp N --loopback` attaches to put the 65 exchanges of a decode step into one rank's measured time.
 * cap_elems <= 32768 (decode-sized messages only). */
typedef struct vh_comm vh_comm_t;
vh_comm_t* vh_comm_create(int rank, int world, size_t cap_elems, void* handle_out);
vh_comm_t* vh_comm_create_loopback(int rank, int world, size_t cap_elems);
int vh_comm_connect(vh_comm_t* c, const void* handles);
size_t vh_comm_capacity(const vh_comm_t* c);
int vh_comm_allreduce(vh_comm_t* c, float* buf, long count, void* stream);
int vh_comm_status(vh_comm_t* c);
void vh_comm_destroy(vh_comm_t* c);
const char* vh_comm_last_error(void);
int vh_comm_is_fine_grained(const vh_comm_t* c);              /* 1: the receive buffer is peer-coherent (required across devices) */
int vh_comm_debug_set_calls(vh_comm_t* c, uint64_t calls);    /* tests: continue from this call count on EVERY rank (tag wrap at 2^32) */
/* Route the engine's per-layer all-reduces through `c` (messages above its capacity keep using the RCCL / callback
 * collective installed before); null detaches it.  The engine does not own `c`. */
int vh_mixtral_use_comm(vh_mixtral_t* m, vh_comm_t* c);

/* Debug / parity hook: the top-2 expert ids of every layer of the following prefills are copied to ids_out
 * (device int[n_layers][S][2]; null switches it off) — SURVEY 8(c) golden list "router top-2 ids per layer". */
/* Give up on a vh_mixtral_init_rccl that is still running in another thread (bring-up time-out): when it returns it
 * destroys its communicator instead of installing it. */
int vh_mixtral_cancel_rccl(vh_mixtral_t* m);
int vh_mixtral_route_debug(vh_mixtral_t* m, int* ids_out);
int vh_mixtral_prefill(vh_mixtral_t* m, const float* embeds, int S, int pos0, float* logits_out, float* hidden_dbg,
                       void* stream);
/* Run n_steps greedy decode steps back to back with no host interaction.  Three launches per layer: the ATTENTION BLOCK as one
 * launch (k_dec_ablk: the fused-QKV rows, the split-KV attention tiles and the O-projection rows are work items of 2 persistent
 * blocks per CU; q|k|v and the attention output travel between items as tagged 8-byte granules, an item's weights / K-V tile are
 * in flight while it waits for its input; vh_tune("dec_fused", 0) runs it as three launches with the same arithmetic, bit for
 * bit), the router + gate|up GEMV of the two chosen experts, and the down projection.  Under tensor parallelism the two
 * all-reduces of a layer are fused into these launches (producer rows push to the peers, the first blocks of the consumer sum in
 * rank order: vh_tune("tp_fuse", 1), what ranks that own their device vote for) or run as one small kernel each.
 * Replaces the per-token forward of HF MixtralDecoderLayer x L as reached from vita/model/language_model/vita_mixtral.py:158-173. */
int vh_mixtral_decode(vh_mixtral_t* m, int n_steps, void* stream);
/* attention block of the last decode call: -1 none yet, 0 three launches (QKV, attention, O projection), 1 one fused launch. */
int vh_mixtral_decode_schedule(const vh_mixtral_t* m);
/* Device pointers into the engine state (for the host loop and the tests). */
const int* vh_mixtral_tokens(const vh_mixtral_t* m);     /* int[max_new]: generated ids   */
const int* vh_mixtral_counters(const vh_mixtral_t* m);   /* int[4]: {pos, n_generated, attn hand-off counter, device error flag (0 = ok)} */
const float* vh_mixtral_logits(const vh_mixtral_t* m);   /* fp32[max(1,logit_rows)][vocab]; row i = scores that produced token i */
int vh_mixtral_reset(vh_mixtral_t* m, void* stream);     /* n_generated = 0, pos = 0      */
/* ---- concurrent sequences over a paged KV cache: what the reference's serving plugin receives from vLLM as
 * `kv_caches` + `attn_metadata` block tables (web_demo/vllm_tools/vllm_file/mixtral.py:491-501,1130-1186; the engine
 * that schedules them: web_interactive_demo.py:942-996).  The KV pool is cut into 64-token pages; a sequence is a slot
 * with its own page table (grown on demand), residual state, counters and generated ids.  A sequence's arithmetic is
 * exactly the single-sequence path's (same kernels, physical rows looked up through the table), so its ids and scores
 * do not depend on what else is scheduled.  Not to be mixed with vh_mixtral_prefill/decode while sequences are live.
 *   seq_alloc   -> slot id >= 0, or VH_E_FULL
 *   seq_prefill    appends S embedded tokens at the sequence's current position and leaves its next greedy token in
 *                  seq_tokens[0] (n_generated restarts at 0, as vh_mixtral_prefill); VH_E_FULL when pages run out
 *   seq_decode     one greedy step for each listed sequence, in order, on `stream` (a continuous-batching iteration)
 *   seq_free       returns the pages;  vh_mixtral_reset frees every sequence */
int vh_mixtral_seq_alloc(vh_mixtral_t* m);
int vh_mixtral_seq_free(vh_mixtral_t* m, int seq);
int vh_mixtral_seq_prefill(vh_mixtral_t* m, int seq, const float* embeds, int S, float* logits_out, void* stream);
int vh_mixtral_seq_decode(vh_mixtral_t* m, const int* seqs /* host */, int n, void* stream);
int vh_mixtral_pages_free(const vh_mixtral_t* m);
int vh_mixtral_seq_pos(const vh_mixtral_t* m, int seq);                  /* tokens in the sequence's KV cache, -1 = not live */
const int* vh_mixtral_seq_tokens(const vh_mixtral_t* m, int seq);        /* device int[max_new]  */
const int* vh_mixtral_seq_counters(const vh_mixtral_t* m, int seq);      /* device int[4], as vh_mixtral_counters */
int vh_mixtral_seq_table(const vh_mixtral_t* m, int seq, int* pages_out /* host */, int cap);   /* -> number of pages */

/* ---- batch-1 decode operators, one entry per kernel group ---------------------------------------------------
 * The steps vh_mixtral_decode chains inside one HF MixtralDecoderLayer (third-party modeling_mixtral.py, reached from
 * vita/model/language_model/vita_mixtral.py:158-169), exposed one by one so that a single module can be bound.  All
 * scratch is the caller's.
 *
 * vh_router_top2 (K25): MixtralSparseMoeBlock's gate on ALREADY NORMED rows — logits = x Wg^T, fp32 softmax, top-2,
 *   renormalise (web_demo/vllm_tools/vllm_file/mixtral.py:398-411, renormalize=True).  x fp32 [rows][ldx], Wg bf16 [E][H],
 *   ids int[rows][2], wts fp32 [rows][2], probs (nullable) fp32 [rows][E].                                                */
int vh_router_top2(const float* x, long ldx, const uint16_t* Wg, int E, int H, int rows, int* ids, float* wts, float* probs,
                   void* stream);
/* vh_moe_decode (K20 + K25 + K26, one token): y = block_sparse_moe(post_attention_layernorm(x + delta)) — the second half
 *   of MixtralDecoderLayer.forward WITHOUT the residual add: RMSNorm (weight norm_w, eps), router as above, the two routed
 *   experts' w2(silu(w1 h) * w3 h) weighted and summed.  x, delta (nullable), y fp32 [H]; x_out (nullable) receives x + delta;
 *   W1, W3 bf16 [E][I][H], W2 bf16 [E][H][I]; route int[4] = {e0, e1, bits(w0), bits(w1)}; hbuf fp32 [2 I] scratch.          */
int vh_moe_decode(const float* x, const float* delta, const float* norm_w, float eps, const uint16_t* Wg, const uint16_t* W1,
                  const uint16_t* W3, const uint16_t* W2, int E, int I, int H, float* x_out, float* y, int* route, float* hbuf,
                  void* stream);
/* vh_rope_kv_append (K22): rotate-half RoPE on the q and k heads of S fused-QKV rows and append k, v to the cache at
 *   positions [pos0, pos0 + S) (HF apply_rotary_pos_emb + Cache.update; vLLM rotary_emb + KV write, mixtral.py:477-501).
 *   qkv fp32 [S][ldqkv] = q heads | k heads | v heads (head_dim 128); q_out fp32 [S][nq*128]; caches fp32
 *   [nkv][max_ctx][128]; rope_cos / rope_sin fp32 [max_pos][64]; table (nullable) = 64-row page table of a paged cache.      */
int vh_rope_kv_append(const float* qkv, long ldqkv, float* q_out, float* kcache, float* vcache, const float* rope_cos,
                      const float* rope_sin, int S, int pos0, int nq, int nkv, int max_ctx, const int* table, void* stream);
/* vh_attn_decode (K22 + K23, one token): RoPE(q, k_new), KV append at `pos`, causal GQA attention of the token against
 *   positions [0, pos] (MixtralAttention.forward between the projections).  qkv fp32 [(nq + 2 nkv) * 128]; attn_out fp32
 *   [nq * 128]; scratch: part_o fp32 [nq][ceil(max_ctx / 64)][128], part_ml fp32 [nq][ceil(max_ctx / 64)][2], tickets
 *   int[nkv] (zero before the first call; the kernel resets them).                                                           */
int vh_attn_decode(const float* qkv, float* kcache, float* vcache, int pos, const float* rope_cos, const float* rope_sin,
                   int nq, int nkv, int max_ctx, float scale, const int* table, float* part_o, float* part_ml, int* tickets,
                   float* attn_out, void* stream);
/* vh_lmhead_argmax (K27 + K28): logits = lm_head(norm(x + delta)) in fp32 (vita_mixtral.py:171-172) and their argmax
 *   (lowest index on ties, as torch.argmax in HF greedy search).  W bf16 [V][H]; logits fp32 [V]; token_out int[1];
 *   blk_val / blk_idx: nblk floats / ints of scratch (nblk = blocks of the launch, e.g. 1024).                               */
int vh_lmhead_argmax(const float* x, const float* delta, const float* norm_w, float eps, const uint16_t* W, int V, int H,
                     float* logits, int* token_out, float* blk_val, int* blk_idx, int nblk, void* stream);

/* Live HIP-event timing of the dominant kernel of a phase, on the stream the engine launches it on: stride > 0 samples the
 * DECODE gate|up expert GEMV of every stride-th layer, stride < 0 the PREFILL gate|up grouped GEMM of every |stride|-th layer
 * (0 = off), up to max_samples launches; read returns summed ms and sample count and clears the samples. */
int vh_mixtral_profile(vh_mixtral_t* m, int stride, int max_samples);
int vh_mixtral_profile_read(vh_mixtral_t* m, double* total_ms /* host */, int* count /* host */);

#ifdef __cplusplus
}
#endif
#endif
