// vh_kernels.h — internal launcher declarations (one per kernel family).  The public
// C ABI in include/vita_hip.h is a thin, argument-checking layer over these.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// ---- run-time tuning knobs (set through vh_tune(); defaults are the measured-best variants).  Round 4 pruned the variants that
// had lost their measurements in rounds 1-3 (double-buffered / 2-row gate|up GEMV, persistent down projection, O-projection
// prefetch blocks, fused attention + O-projection launch, de-duplicating batch GEMVs, the LDS-tiled fp32 attention kernel, the
// general-kernel MoE prefill path, the register-direct streaming GEMM): DESIGN.md 6.2 and the git history are their record. ---
struct VhTuning {
    int batch_moe_min = 3;     // concurrent sequences: from this many per iteration the layer's MoE runs ONCE on the weight-streaming GEMM
                               // (S = n rows sorted by expert, every touched expert streamed once); 0 = never
    int batch_decode = 1;      // concurrent sequences: 1 = groups of up to 4 sequences per batched decode step, 0 = one sequence after the other
    int attn_impl = 0;         // multi-row attention: 0 = bf16 x 3 MFMAs where the mask flavour allows (plain / causal), 2 = fp32-MFMA kernel everywhere
    int attn_fa = 1;           // bf16 x 3 attention: 1 = flash form (K / V tiles through LDS, k_attn_fa) when its blocks fill half the chip, 2 = always, 0 = never
    int attn_rows = 0;         // bf16 x 3 attention at d = 64: query rows per wave, 0 = auto (32 when the launch still fills the chip), 16, 32
    int attn_ksplit = 0;       // multi-row attention: key groups per block, 0 = auto (4 at d = 64, 2 at d = 128), 1 = single group
    int prefill_attn_gemm = 0; // prefill QKV / O projections: 0 = weight-streaming pre-split kernel with a K split, 1 = general kernel
    int prefill_fuse_rows = 1; // single-rank prefill: K-split slabs summed by the consuming norm kernel (VhRowUpdate); 0 = separate slab-sum / combine launches
    int ps_cfg = -1;           // vh_gemm_ps variant: -1 = by rows per group (0 up to 64 rows, else 2), 0 = 64 rows / 8-slot weight DMA ring, 1 = 192 rows / register-staged weights,
                               // 2 = 192 rows, 12 specialised waves (8 MFMA-only + 2 weight stagers + 2 activation DMA: vh_gemm_sp.hip)
    int ps_xcd = -1;           // specialised streaming GEMM: bit 0 = the down projection / plain GEMMs, bit 1 = gate|up take the XCD-contiguous placement
                               // of a round's tiles (VhGemmPsArgs::xcd_group); -1 = auto
    int ps_nt = -1;            // vh_gemm_ps non-temporal weight loads: -1 = default (8-wave kernels: unless the last round is M-split; specialised kernel: off), 0 = never, 1 = always
    int tp_overlap = 1;        // tensor-parallel prefill: all-reduce of one column half on a comm stream under the GEMM of the other half
    int moe_ksplit = -4;       // K split of the prefill MoE down projection (partial slabs, summed by the combine kernel); < 0: chosen on device up to -n
    int force_allreduce = 0;   // tests: run the collective hook even when tp_world == 1
    int tp_fuse = 0;           // batch-1 decode under the library's IPC all-reduce: 0 = one 16-block all-reduce kernel per exchange,
                               // 1 = exchange fused into the producer / consumer kernels (VhXchg).  Chosen at bring-up by
                               // vita_amd.parallel (timed on the ranks' own devices; "kernel" whenever ranks share a device)
    int dec_fused = -1;        // batch-1 decode: 1 = the attention block of a layer as ONE launch (k_dec_ablk: fused QKV GEMV -> split-KV attention
                               // -> O projection, q|k|v and the attention output handed over as tagged granules, a block's weights in flight
                               // while it waits), 0 = three launches (five per layer), -1 = auto (fused wherever the kernel has an instantiation:
                               // H and the heads' width <= 4096, i.e. the released geometry at any TP degree)
    int attn_xcd = 1;          // attention kernels: 1 = the q tiles of one (head, batch) pair all run on ONE XCD (its L2 holds that pair's K / V), 0 = plain x / y / z
    int attn_img = 1;          // one-shot prefills under the flash attention kernel: 1 = K / V as MFMA-ready tile images written by the RoPE pass
                               // (k_rope_kv_img -> LDS-DMA in k_attn_fa), 0 = fp32 K / V staged and converted by every attention block (r04-r05)
    int dec_gateup_grid = 0;   // debug: blocks of the batch-1 gate|up launch (0 = auto: 1.5 per CU, equal shares in few rounds for small shards)
    int comm_allow_coarse = 0; // vh_comm_create: 1 = a coarse-grained receive buffer is acceptable when the fine-grained allocation fails (only
                               // correct when every rank drives ONE device: same-device tests); 0 = fail loudly instead
    int comm_ranks_per_device = 1;   // engine processes that drive THIS process's device (vita_amd.parallel counts the ranks from the devices' PCI
                               // identities; vita_amd.duplex sets 2 for replicas on one GPU).  > 1: the bulk all-reduce divides its
                               // resident-block cap by it, the decode exchange is never fused into the kernels, and the attention block of a
                               // decode layer runs as three launches — any launch whose blocks WAIT for other blocks can be starved by another
                               // process's waiting blocks (block dispatch is in index order per XCD only)
};
VhTuning* vh_tuning();

// ---- tensor-parallel exchange fused into the batch-1 decode kernels (vh_comm.hip fills it, vh_decode.hip uses it) -------
// One all-reduce(sum) of a `count`-element fp32 vector = one VhXchg, used twice:
//   * the PRODUCER (O-projection blocks of the attention-block launch, MoE down projection) pushes every output element straight
//     into slot `rank` of every peer's receive region as an 8-byte {value, tag} granule instead of storing it locally;
//   * the CONSUMER launch (gate|up, the next layer's attention block, the LM head) needs the whole summed vector in every block
//     (RMSNorm): its first `nred` blocks poll the world slots of their slice in THIS rank's region — all slots of an element at
//     once —, sum them in rank order (bit-identical on all ranks) and publish the slice to `reduced_g` as tagged granules again
//     (GEMV layout, the exchange's tag); every block puts its weight loads in flight first and then reads the vector with the
//     granule sweep of the fused attention block (vh_decode.hip gran_read_gemv: one wave polls one granule, then every wave sweeps
//     its own) — the data is the flag: no counter, no fence, no drain between reducers and readers (r03-r05 published plain floats
//     behind an arrival counter: one write-through drain + barrier + atomic + counter poll more per exchange).
// world == 0: no exchange (plain local delta buffers, the single-GPU path).
struct VhXchg {
    uint64_t* peer[8];              // every rank's receive region of this call's parity (peer[rank] == local)
    uint64_t* local;
    unsigned long long* reduced_g;  // this rank's summed vector as granules in the GEMV layout (device memory of the communicator)
    int* err;                       // sticky time-out word (vh_comm_status)
    unsigned long long cap;         // slot stride in granules
    int rank, world;
    unsigned tag;
    int nred, count;
    int loopback;                   // 1: one rank plays all `world` ranks (every peer[] is the local region): the value goes to slot `rank`,
                                    // zeros to the other slots — the same stores, polls and rank-ordered sum as a real exchange, no link
};

// ---- decode activation vectors handed between work items of ONE launch (the fused attention block, k_dec_ablk) ----------------
// The fused-QKV rows, the split-KV attention tiles and the O-projection rows of a layer are items of one persistent launch; an item
// that consumes another item's output is resident — its weights or K / V tile in flight — while the producer still runs.  The data
// carries the dependency: every element travels as ONE naturally aligned 8-byte {tag, fp32 bits} granule written and read with
// agent-scope atomics (guide G16 R2: the data is the flag — no fence, no flag word; the form VhXchg uses across devices).
// tag = a per-engine counter that is different for every (step, layer, vector); 0 is never used (the buffers start zeroed).
//   layout 0 (VH_GRAN_LINEAR): element n in granule n (attention reads q / k / v of a head by lane)
//   layout 1 (VH_GRAN_GEMV):   the consumer is a GEMV block whose thread t owns the 16-byte weight chunks t + 256 j, i.e. elements
//                              [8 (t + 256 j), + 8): element n = 8 (t + 256 j) + e sits in granule (8 j + e) * 256 + t, so that the
//                              64 lanes of a wave read 512 contiguous bytes per load instruction
struct VhGranVec {
    unsigned long long* g;
    unsigned tag;
    int* err;                       // device error word (engine counters[3]): set when a bounded wait gives up
};
#define VH_GRAN_SPIN_LIMIT (1u << 21)      // polls of ~0.2-0.4 us: a producer that never publishes ends in the error word after ~0.5 s
__host__ __device__ inline size_t vhk_gran_pos_gemv(int n) { return (size_t)(((n >> 11) << 3) + (n & 7)) * 256 + ((n >> 3) & 255); }   // layout 1: granule of element n
__host__ __device__ inline size_t vh_gran_gemv_len(int K) { return (size_t)((K + 2047) / 2048) * 2048; }   // granules of a layout-1 vector

struct vh_comm;
int vh_comm_xchg_next(vh_comm* c, long count, int which, int consumer_blocks, VhXchg* out, void* stream);   // vh_comm.hip
int vh_comm_ranks_per_device(const vh_comm* c);   // declared at creation: ranks that drive this rank's device (1 on a node with a GPU per rank)
int vh_comm_is_loopback(const vh_comm* c);        // a single-rank communicator that plays `world` ranks into its own slots (bench: protocol cost without links)

// ---- decode (vh_decode.hip) ---------------------------------------------------------
// cx (nullable): the delta is the result of a fused exchange (then `delta` is ignored); px (nullable): push the outputs
int vhk_dec_qkv(hipStream_t st, const float* x_in, const float* delta, float* x_out, const float* norm_w, float eps,
                const uint16_t* W, int N, int K, float* out, const VhXchg* cx = nullptr);
int vhk_dec_consumer_blocks(int which, int N, int K, int I);   // grid of a consumer launch (0 qkv, 1 gate|up, 2 lm head, 3 fused attention block): bounds nred
int vhk_dec_attn(hipStream_t st, const float* qkv, float* kcache, float* vcache, const int* pos_ptr,
                 const float* rope_cos, const float* rope_sin, float* part_o, float* part_ml, int* cnt,
                 float* attn_out, int nq, int nkv, int max_ctx, int max_splits, int ctx_host, float scale,
                 const int* table);   // table: nullable page table of a paged KV cache (64-token pages)
int vhk_dec_oproj(hipStream_t st, const float* attn_out, const uint16_t* W, int N, int K, float* out, const VhXchg* px = nullptr);
// The attention block of one decode layer as ONE launch (k_dec_ablk): out[H] = Wo * attention(RoPE(Wqkv * rmsnorm(x_in + delta))), the
// new K / V row appended to the cache, x_out = x_in + delta.  cx (world > 0): delta is the result of a fused exchange; px (world > 0):
// the outputs are pushed to the peers instead of stored.  gq / ga: granule vectors of nqkv (linear) / vh_gran_gemv_len(nq * 128)
// (GEMV layout) elements with tags no earlier launch used.
struct VhDecAblk {
    const float* x_in; const float* delta; float* x_out; const float* norm_w; float eps;
    const uint16_t* Wqkv; int nqkv, H;
    float* kcache; float* vcache; int pos; const int* table; const float* rope_cos; const float* rope_sin;
    float* part_o; float* part_ml; int* cnt; int nq, nkv, max_ctx, max_splits, nsplit; float scale;
    const uint16_t* Wo; float* out;
    VhGranVec gq, ga;
    VhXchg cx, px;
};
int vhk_dec_ablk(hipStream_t st, const VhDecAblk& a);
int vhk_dec_ablk_supported(int H, int nq, int nkv);   // 1 when k_dec_ablk has an instantiation for these widths
int vhk_dec_ablk_qkv_blocks(int nqkv, int H);         // its fused-QKV blocks (the first of the launch: a fused exchange's reducers are among them)
// ---- batched decode (one iteration of up to VH_BMAX concurrent sequences; vh_decode.hip) -------------------------------
#define VH_BMAX 4
struct VhDecBatchVec {       // a GEMV-shaped step over the batch: out[b] = f(W, x_in[b] (+ delta[b]))
    int n;
    const float* x_in[VH_BMAX]; const float* delta[VH_BMAX];   // delta: all null or all set
    float* x_out[VH_BMAX];                                     // nullable: x_in + delta stored by block 0 (NORM kernels)
    float* out[VH_BMAX];
};
struct VhDecBatchAttn {
    const float* qkv[VH_BMAX]; int pos[VH_BMAX]; const int* table[VH_BMAX];
    float* part_o[VH_BMAX]; float* part_ml[VH_BMAX]; int* cnt[VH_BMAX]; float* attn_out[VH_BMAX];
};
struct VhDecBatchHead { float* logits[VH_BMAX]; float* blk_val[VH_BMAX]; int* blk_idx[VH_BMAX]; };   // logits[b]: nullable full-vocab row
int vhk_decb_gemv(hipStream_t st, const VhDecBatchVec& bt, const float* norm_w, float eps, const uint16_t* W, int N, int K, int norm);
int vhk_decb_attn(hipStream_t st, const VhDecBatchAttn& bt, int n, float* kcache, float* vcache, const float* rope_cos,
                  const float* rope_sin, int nq, int nkv, int max_ctx, int max_splits, float scale);
int vhk_decb_lmhead(hipStream_t st, const VhDecBatchVec& bt, const float* norm_w, float eps, const uint16_t* W, int V, int K,
                    const VhDecBatchHead& hd, int grid, int v0);
int vhk_dec_gateup(hipStream_t st, const float* x_in, const float* delta, float* x_out, const float* norm_w, float eps,
                   const uint16_t* Wg, int E, const uint16_t* W1, const uint16_t* W3, int I, int K, int* route_out,
                   float* hbuf, int grid, const VhXchg* cx = nullptr);
int vhk_dec_down(hipStream_t st, const float* hbuf, const int* route, const uint16_t* W2, int N, int I, float* out,
                 const VhXchg* px = nullptr);
int vhk_dec_lmhead(hipStream_t st, const float* x_in, const float* delta, const float* norm_w, float eps,
                   const uint16_t* W, int V, int K, float* logits, float* blk_val, int* blk_idx, int grid,
                   const int* ngen_ptr, int hist_rows, int v0, int Vfull, const VhXchg* cx = nullptr);
int vhk_dec_cand(hipStream_t st, const float* blk_val, const int* blk_idx, int nblk, float* cand, int rank, int world);
int vhk_dec_cand_unpack(hipStream_t st, const float* cand, int world, float* val, int* idx);
int vhk_dec_pick(hipStream_t st, const float* blk_val, const int* blk_idx, int nblk, int vocab, int* token_out, float* value_out);
int vhk_dec_select(hipStream_t st, const float* blk_val, const int* blk_idx, int nblk, const uint16_t* embed, int H,
                   int vocab, float* x_next, int* pos_ptr, int* ngen_ptr, int* out_tokens, int max_out, int mode, int set_pos);

// ---- GEMM (vh_gemm.hip) -------------------------------------------------------------
// C[orow(m), n] = epilogue( sum_k A[arow(m,k), k] * W[n, k] )
//   A fp32, W bf16 [N][K] (torch Linear layout), C fp32.  K % 64 == 0.
struct VhGemmArgs {
    const float* A; long lda; int a_rows;     // a_rows: source rows outside [0,a_rows) read as zero
    const int* a_rowidx;                      // nullable: source row of logical row m (else m)
    int nseg, seglen; int segrow[16];         // K = nseg*seglen; segment s reads source row + segrow[s]
    const uint16_t* W; const uint16_t* W_up;  // W_up != null => GLU: out = silu(A W^T) * (A W_up^T)
    long ldw; long w_group_stride;
    const int* group_off; int ngroups;        // nullable; device int[ngroups+1] row offsets (grouped GEMM)
    float* C; long ldc; const int* c_rowidx;  // nullable output row map
    const float* bias; const float* scale; const float* resid; long ldr;
    int M, N, K, act;
    int mt_slots;                             // set by the launcher: m-tile slots per n-tile
    // split-K for launches too small to fill the chip (encoder GEMMs: 136 blocks at M = 1025, N = 1024): `ws` is caller
    // scratch for ksplit * M * N partial sums; a second kernel adds them and applies the epilogue.  ksplit 0 = the
    // launcher decides (only when ws is given), 1 = off.  Plain (ungrouped, non-gated, no row maps) GEMMs only.
    float* ws; size_t ws_bytes; int ksplit;
    // (r03) the LayerNorm that consumes C, in the same call: ln_out[m, :] = LN(C[m, :]) * ln_w + ln_b (nullable ln_b).  With
    // split-K the reducer already holds whole rows, so the norm costs no launch of its own; otherwise a norm launch follows.
    const float* ln_w; const float* ln_b; float ln_eps; float* ln_out; long ld_ln;
    uint16_t* ln_hi; uint16_t* ln_lo; long ld_ln_split;   // (r04) the normed rows as bf16 hi/lo planes (what the streaming GEMM consumes); ln_out may then be null
};
int vhk_gemm(hipStream_t st, const VhGemmArgs& a);
// the split-K reducer of vh_gemm.hip on slabs some OTHER kernel wrote: ws = [ksp][M][N] fp32 partial sums; applies bias / act / scale /
// resid into C and, with ln_w, the LayerNorm of the row into ln_out and / or the ln_hi / ln_lo planes (N <= 4096 for the fused norm)
int vhk_gemm_reduce(hipStream_t st, const VhGemmArgs& a, int ksp);

// ---- weight-streaming GEMM on pre-split activations (vh_gemm_ps.hip) -------------------
// C[orow(m), n] = epilogue( sum_k (A_hi + A_lo)[arow(m), k] * W[n, k] ), A as bf16 hi/lo planes.  K % 64 == 0.
struct VhGemmPsArgs {
    const uint16_t* A_hi; const uint16_t* A_lo; long lda;   // planes [rows][lda] bf16
    const int* a_rowidx;                                     // nullable gather: source row of logical row m
    const uint16_t* W; const uint16_t* W_up; long ldw; long w_group_stride;   // W_up != null => SiLU(A W^T) * (A W_up^T)
    const int* group_off; int ngroups;                       // nullable: device int[ngroups+1] sorted-row offsets
    float* C; long ldc;                                      // nullable fp32 output
    uint16_t* C_hi; uint16_t* C_lo; long ldc_split;          // nullable split output planes
    const int* c_rowidx;
    const float* bias; const float* scale; const float* resid; long ldr;
    int M, N, K, act;
    int ksplit; long c_split_stride;                         // K split: partial sums go to C + ks * c_split_stride (plain fp32 output only);
                                                             // ksplit < 0: the kernel picks 1 .. -ksplit from the group sizes
    int* nslab_out;                                          // device int: the split the kernel used (required when ksplit < 0)
    int rt_cap;                                              // 0 = tile rows up to the kernel's maximum (192); n: m-tiles of at most 16 n rows (plain GEMMs at M ~ 1000: more, smaller tiles fill the chip)
    int xcd_group;                                           // specialised kernel, set by the launcher: 1 = XCD x takes the CONTIGUOUS positions [x nb, (x+1) nb) of every round of the
                                                             // tile list (the n-tiles of one (expert, K slice) on ONE XCD: they read the same activation rows in step, one L2 fill serves them)
};
int vhk_gemm_ps(hipStream_t st, const VhGemmPsArgs& a);
int vhk_gemm_sp(hipStream_t st, const VhGemmPsArgs& a, int grid, bool nt);   // vh_gemm_sp.hip: 12-wave specialised form (arguments already checked)
int vhk_split_planes(hipStream_t st, const float* x, long ldx, uint16_t* hi, uint16_t* lo, long ldo, int rows,
                     int cols);

// ---- attention (vh_attn.hip) --------------------------------------------------------
struct VhAttnArgs {
    const float* Q; long ldq; long hsq;   // Q[(b*Sq + q)*ldq + h*hsq + d]   (b strides below)
    const float* K; long ldk; long hsk;
    const float* V; long ldv; long hsv;
    const float* P; long ldp; long hsp;   // rel-pos keys (audio), nullable
    const float* bias_u; const float* bias_v;  // [H][d], nullable (rel-pos)
    float* O; long ldo;                   // O[(b*Sq + q)*ldo + h*d + dd]
    uint16_t* O_hi; uint16_t* O_lo; long ldo_split;   // optional bf16 hi/lo planes of O: row b * Sq + q, column h * d + dd; O may then be null
    long bsq, bsk, bso;                   // batch strides in elements for Q / K,V / O
    int B, Hq, Hkv, Sq, Sk, d;
    int causal; int q_off;                // causal: key <= q + q_off visible
    int klen;                             // keys >= klen are masked (pad mask); use Sk for none
    int chunk, left;                      // chunk>0: whale chunk mask (utils.py:88-103); left<0 = all left chunks
    float scale;
    const int* ktable;                    // nullable: keys / values live in 64-row pages, logical block j>>6 -> page ktable[j>>6]
    long kv_rows;                         // rows a page-table entry may address (the pool); 0 = Sk.  Bounds the 32-bit offsets of k_attn_x3
    // (r06) optional: K and V of this call as MFMA-READY bf16 hi/lo tile images written by the producer (vhk_rope_kv_img): image of
    // (kv head h, 64-key tile t) = 64 KB at kv_img + (h * img_tiles + t) * 65536 in exactly k_attn_fa's LDS layout (K hi | K lo |
    // V hi transposed | V lo transposed, 16-byte chunks XOR-swizzled; rows past Sk are zeros).  Used by the flash kernel for one-shot
    // causal prefills (q_off == 0, Sk == Sq, B == 1); ignored otherwise (K / V must always be valid too).
    const unsigned char* kv_img; int img_tiles;
    int xcd_map;                          // set by vhk_attn (vh_tune "attn_xcd"): block -> (q tile, head, batch) with all q tiles of a head on one XCD
};
int vhk_attn(hipStream_t st, const VhAttnArgs& a);
int vhk_attn_fa_applies(const VhAttnArgs& a);    // 1 when vhk_attn would run the flash kernel (k_attn_fa) for these arguments

// ---- element-wise / index kernels (vh_elem.hip) -------------------------------------
int vhk_layernorm(hipStream_t st, const float* x, long ldx, float* y, long ldy, const float* w, const float* b,
                  int rows, int cols, float eps, int act, float post_scale);
int vhk_rmsnorm(hipStream_t st, const float* x, float* y, const float* w, int rows, int cols, float eps);
int vhk_add(hipStream_t st, float* x, const float* y, long n);
int vhk_add_halves(hipStream_t st, float* x, const float* a, const float* b, int rows, int cols);
int vhk_vit_patchify(hipStream_t st, const float* pix, float* out, int n, int img, int patch, int kpad);
int vhk_vit_assemble(hipStream_t st, const float* patches, const uint16_t* cls, const uint16_t* pos, float* x, int n,
                     int ntok, int hid);
int vhk_vit_pixel_shuffle(hipStream_t st, const float* x, float* out, int n, int grid, int hid, float mul);
int vhk_audio_conv1(hipStream_t st, const float* feats, const float* mean, const float* istd, const uint16_t* w,
                    const float* b, float* out, int T, int F, int C);
int vhk_rope_kv(hipStream_t st, const float* qkv, long ldqkv, float* q_out, float* kcache, float* vcache,
                const float* rope_cos, const float* rope_sin, int S, int pos0, int nq, int nkv, int max_ctx,
                const int* table, const int* nslab_dev, long slab_stride);   // nslab_dev: qkv is *nslab_dev partial slabs
// the same for a one-shot prefill (positions 0 .. S-1) whose attention is the flash kernel: additionally writes the K / V tile images
// (VhAttnArgs::kv_img) — one block per (64-row tile, KV head) converts the tile once, where it is produced (nq == 4 * nkv)
int vhk_rope_kv_img(hipStream_t st, const float* qkv, long ldqkv, float* q_out, float* kcache, float* vcache,
                    const float* rope_cos, const float* rope_sin, int S, int nq, int nkv, int max_ctx,
                    const int* table, const int* nslab_dev, long slab_stride, unsigned char* img, int img_tiles);
int vhk_gather_rows(hipStream_t st, float* seq_x, const int* slots, int n, int H, float* rows);   // rows[b] = xa[b] = xb[b] + delta_attn[b]
int vhk_sum_slabs(hipStream_t st, float* dst, long ldd, const float* src, long lds, int rows, int cols,
                  const int* nslab_dev, int nslab, long stride, int accumulate);
int vhk_embed_splice(hipStream_t st, const int* src_kind, const int* src_idx, const uint16_t* embed,
                     const float* img_feats, const float* aud_feats, float* out, int S, int H);
// A pending update of the residual rows that the norm kernel applies (and writes back) before it norms them, so the
// producer's K-split slabs are summed where they are consumed instead of in a launch of their own:
//   wts == null:  x[r,:] += sum_k src[k*stride + r*ld + :]                                  (k_sum_slabs, accumulate)
//   wts != null:  x[r,:] += wts[2r] * sum_k src[k*stride + 2r*ld + :] + wts[2r+1] * sum_k src[.. (2r+1)*ld ..]  (k_moe_combine)
struct VhRowUpdate {
    const float* src; long ld; long stride;
    const int* nslab_dev; int nslab;        // the slab count is read on the device when nslab_dev != null
    const float* wts;
};
int vhk_rmsnorm_route(hipStream_t st, float* x, float* y, uint16_t* y_hi, uint16_t* y_lo, const float* w, int rows,
                      int cols, float eps, const uint16_t* Wg, int E, int* ids, float* wts, const VhRowUpdate* upd = nullptr);
int vhk_router_top2(hipStream_t st, const float* x, long ldx, const uint16_t* Wg, int E, int H, int rows, int* ids, float* wts,
                    float* probs);   // probs nullable: [rows][E] softmax
int vhk_moe_sort(hipStream_t st, const int* ids, int S, int E, int* group_off, int* sorted_tok, int* sorted_slot);
int vhk_moe_combine(hipStream_t st, float* x, const float* y, const float* wts, int S, int H, int nslab,
                    long slab_stride, const int* nslab_dev);
int vhk_cast_bf16_f32(hipStream_t st, const uint16_t* in, float* out, long n);
int vhk_fill_hash_bf16(hipStream_t st, uint16_t* dst, long rows, long cols, long ld_dst, long ld_src, long idx0,
                       uint64_t seed);
