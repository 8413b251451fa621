/* C ABI of libvpu_hip.so -- the MI355X (gfx950) kernels behind the VPUFormer hot path.
 *
 * The reference (XuZhang1211/PVPUFormer) is pure Python/PyTorch: its "FFI" for this path is the set of
 * ATen operator calls made by isegm/model/is_vpu_model.py and the modules it composes.  Each entry point
 * below names the reference lines whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions:
 *   - raw DEVICE pointers + explicit shapes / leading dimensions + a hipStream_t (passed as void*);
 *   - returns 0 on success, a negative VPU_ERR_* otherwise (vpu_last_error() gives the text);
 *   - never allocates, frees or synchronises; workspaces are passed in by the caller;
 *   - no global mutable state; safe to call from any host thread on distinct streams; capturable in a hipGraph;
 *   - dtype codes: VPU_BF16 = 0 (activations bf16, fp32 accumulate), VPU_F32 = 1 (exact-fp32 parity mode).
 *     Norm parameters, biases, statistics, losses and gradients of parameters are always fp32.
 */
#ifndef VPU_HIP_H
#define VPU_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPU_BF16 0
#define VPU_F32 1

/* GEMM epilogue flags (applied in this order) */
#define VPU_EPI_BIAS 1      /* v += bias[n]                                              */
#define VPU_EPI_PREACT 2    /* preact[m,n] = v                                           */
#define VPU_EPI_GELU 4      /* v = gelu_erf(v)              models_vit.py:21-27          */
#define VPU_EPI_RELU 8      /* v = max(v,0)                 common.py:28-42              */
#define VPU_EPI_DGELU 16    /* v *= gelu'(aux[m,n])         (backward of GELU)           */
#define VPU_EPI_DRELU 32    /* v *= aux[m,n] > 0            (backward of ReLU)           */
#define VPU_EPI_RESID 64    /* v += resid[m,n]              models_vit.py:72-75          */
#define VPU_EPI_AFFINE 128  /* v = v*post_mul + post_add    swin_transformer.py:752      */
#define VPU_EPI_ACCUM 256   /* v += C[m,n]   (fp32 output only; gradient accumulation)   */
#define VPU_EPI_OUT_F32 512 /* C is fp32 although dtype is bf16                          */
#define VPU_EPI_SAVE_DGELU 1024 /* with GELU: preact[m,n] = gelu'(v) (shares the erf) -- the backward of the MLP then
                                   needs no transcendental: its dgrad epilogue is VPU_EPI_MULAUX           */
#define VPU_EPI_MULAUX 2048 /* v *= aux[m,n]  (applied where DGELU/DRELU would be)                     */

typedef struct vpu_gemm_desc {
    const void* A;      /* transA=0: [M][lda] K-contiguous;  transA=1: [K][lda] (M contiguous) */
    const void* B;      /* transB=0: [N][ldb] K-contiguous (nn.Linear weight); transB=1: [K][ldb] */
    void* C;            /* [M][ldc] */
    const float* bias;  /* [N] or NULL */
    const void* resid;  /* element type = dtype */
    const void* aux;    /* element type = dtype */
    void* preact;       /* element type = dtype, leading dimension ldc */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldr, ldaux;
    int32_t batch, inner; /* blockIdx.z = zo*inner + zi */
    int64_t sAo, sAi, sBo, sBi, sCo, sCi, sRo, sRi; /* element strides per outer / inner batch index */
    int32_t transA, transB;
    int32_t dtype, flags;
    int32_t resid_period; /* > 0: resid row = m % period and no batch stride (broadcast pos_embed) */
    float alpha, post_mul, post_add;
    void* workspace;         /* optional fp32 scratch for split-K partial tiles (bf16 path); NULL disables split-K.
                                ZERO it once after allocation: its last 256 KiB are tile-arrival counters of the in-launch
                                split-K combine, which every launch leaves zeroed; one workspace per stream */
    int64_t workspace_bytes; /* split-K needs batch * slices * M * (N + 1) * 4 bytes + 256 KiB */
    float* colsum;           /* optional (bf16, transA=1, batch=1): colsum[m] += sum_k op(A)[m][k] -- the bias gradient
                                fused into the weight-gradient GEMM whose A operand is dY */
    int32_t cs_tn, cs_t0;    /* vpu_gemm_grouped, 256 x 256-tile weight-gradient kernels only.  cs_tn > 1: the column sums are
                                DISTRIBUTED over the cs_tn column tiles (256 columns each) of the whole problem -- column tile
                                t sums over K-steps [t nk / cs_tn, (t + 1) nk / cs_tn) and WRITES colsum[(z cs_tn + t) cs_ld + m]
                                (a slab the caller adds up); cs_t0 = the global index of this descriptor's first column
                                tile (a problem cut along its columns).  0 / 1: the classic form above */
    int64_t cs_ld;           /* row stride of that slab (the row count of the whole problem) */
} vpu_gemm_desc;

/* Up to VPU_GEMM_GROUP_MAX independent bf16 problems run by ONE launch of vpu_gemm_grouped: start[i] = first tile of
 * problem i in the concatenated list of 128x128 output tiles, start[n] = total. */
#define VPU_GEMM_GROUP_MAX 16   /* 16 descriptors = 3.5 KB of the 4-KB kernel-argument segment */
typedef struct vpu_gemm_group {
    int32_t n;
    int32_t start[VPU_GEMM_GROUP_MAX + 1];
    vpu_gemm_desc d[VPU_GEMM_GROUP_MAX];
} vpu_gemm_group;

const char* vpu_last_error(void);
int vpu_abi_version(void);

/* C = epilogue(alpha * op(A) op(B)).  Replaces every nn.Linear / 1x1 / patch / 2x2-stride-2 convolution and every
 * attention matmul on the path (models_vit.py:38-52,16-27,91; transformer.py:484-517; is_vpu_model.py:55-86;
 * swin_transformer.py:680-756) and their autograd backward (dgrad: transB=1, wgrad: transA=transB=1).
 * bf16: requires lda, ldb (and column offsets) to be multiples of 8 elements, 16-byte aligned bases; K padded with
 * zeros to a multiple of 8 by the producer of a K-contiguous operand.  f32: multiples of 4 / 16 bytes. */
int vpu_gemm(const vpu_gemm_desc* d, void* stream);
/* Kernel-selection knobs of vpu_gemm (tuning / tests; the defaults are the measured-fastest choices).
 * "ring": main loop of the 128x128 bf16 kernel = three LDS stages in a ring with a counted vmcnt instead of two stages:
 * -1 environment default (VPU_GEMM_RING, 0 if unset), 0 off, 1 one-wave problems (96..256 tiles, K <= 1536), 2 always.
 * "k2" (round 2): the 256-row-tile kernels (128 x 64 outputs per wave, one workgroup per CU): -1 environment default
 * (VPU_GEMM_K2, 2 if unset), 0 off, 1 the 256 x 128 form wherever legal, 2 the measured-fastest mix of 256 x 256 / 256 x 128 /
 * 128 x 128, 3 the 256 x 256 form wherever legal.
 * "reserve_cus": n (0..128, default 0): the persistent launches size their grids for (CUs - n), leaving room for the
 * channel kernels of a collective that runs beside them (pvpuformer_amd/parallel.py sets it while gradient buckets are
 * in flight).
 * "splitk_inlaunch": 0 (default; VPU_GEMM_INLAUNCH) a separate reduce launch sums the split-K slices; 1 they are summed,
 * in slice order, by the slice that arrives last at the tile's counter, inside the GEMM launch; n > 1: that, but only
 * when the slabs of the launch total at most n MiB.  Same results bit for bit; the default is the measured-faster one. */
int vpu_gemm_set_option(const char* name, int32_t value);
/* "k5" (round 6, csrc/gemm_k5.hip): the two-tile ping-pong family for the forward / dgrad forms of the blocks (the two 4-wave
 * groups of a workgroup own alternate 128-column tiles; one group's LDS-DMA + direct epilogue run beside the other group's main
 * loop): -1 environment default (VPU_GEMM_K5, 1 if unset), 0 off, 1 wherever legal except bias + GELU + GELU', 2 that too.
 * "k5_split" (-1 by flag set, 0 / 1: the LDS-DMA pieces issued by the producer waves alone / half by each wave group),
 * "k5_grid" (cap on its grid, 0 = none) and "k5_noepi" (main loops only) are test / diagnostic knobs.
 *
 * vpu_gemm_get_option: the EFFECTIVE value of "k2" / "k3" / "k5" as the dispatch of THIS library reads it -- environment default
 * resolved, laboratory-only bits masked in the product library -- so that a host that sizes its launches by the kernel family
 * (pvpuformer_amd/engine.py: the packed weight-gradient rounds) cannot disagree with the library about it. */
int vpu_gemm_get_option(const char* name, int32_t* value);
/* Name (as rocprofv3 prints it) of the kernel instantiation the calling host thread's last vpu_gemm / vpu_gemm_grouped call
 * launched -- lets a measurement harness label launches without mirroring the dispatch rules.  "" before the first call. */
const char* vpu_gemm_last_kernel(void);
/* n (1..VPU_GEMM_GROUP_MAX) independent bf16 GEMMs in one persistent launch, every problem un-split over its whole K:
 * meant for sets whose tiles together fill the chip (the four weight gradients of a ViT block: models_vit.py:38-40,16-18
 * backward) or whose K is short (the DMA neck's 576-row problems), where separate launches each pay split-K slabs + a
 * reduce launch.  All descriptors: dtype bf16, batch 1, the same transA / transB; epilogue flags, bias, colsum etc. are
 * per problem as in vpu_gemm; workspace is ignored. */
int vpu_gemm_grouped(const vpu_gemm_desc* d, int32_t n, void* stream);

/* ---- row-wise ops ---- */
/* nn.LayerNorm over the last dim (models_vit.py:126 eps 1e-6; transformer.py:417-426 eps 1e-5). */
int vpu_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                      int64_t rows, int32_t C, float eps, int32_t dtype, void* stream);
/* dx = [dres +] LN'(dy); dw/db partial sums go to part[nblk][2][C] (one vpu_colsum_f32 over [nblk][2C] yields both,
 * the two gradients being adjacent in the flat gradient buffer). nblk is returned
 * by vpu_layernorm_bwd_nblk(rows). */
int vpu_layernorm_bwd_nblk(int64_t rows);
int vpu_layernorm_bwd(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                      const void* dres, void* dx, float* part, int64_t rows, int32_t C, int32_t dtype, void* stream);
/* LayerNorm with the position-embedding add that follows it in the DMA neck (TwoWayAttentionBlock.forward,
 * transformer.py:438-460: queries = norm(queries); q = queries + query_pe -- likewise keys + key_pe) in the same launch:
 * y = LN(x) and y2 = y + pe[row % pe_rows] (pe: [pe_rows][C]; the sum is taken from the rounded y, as a separate add
 * would take it).  pe = y2 = NULL: plain vpu_layernorm_fwd.  Its backward: the gradient of the output is dy + dy2
 * (dy2 may be NULL), summed in fp32. */
int vpu_layernorm_fwd_pe(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                         int64_t rows, int32_t C, float eps, const void* pe, int64_t pe_rows, void* y2,
                         int32_t dtype, void* stream);
int vpu_layernorm_bwd2(const void* dy, const void* dy2, const void* x, const float* w, const float* mean,
                       const float* rstd, const void* dres, void* dx, float* part, int64_t rows, int32_t C,
                       int32_t dtype, void* stream);
/* n (<= VPU_COLSUM_BATCH_MAX) independent fp32 column sums in one launch: out_j[c] += sum_r in_j[r * ncols_j + c].
 * Used for the partial weight / bias gradient rows of every LayerNorm / GroupNorm backward of a step
 * (models_vit.py:72-75, transformer.py:417-426, is_vpu_model.py:55-86 backward). */
#define VPU_COLSUM_BATCH_MAX 64
/* row_len > 0: the output is a [ncols / row_len][row_len] block of a wider matrix with leading dimension out_ld -- column c
 * goes to out[(c / row_len) * out_ld + c % row_len] (the column blocks of the head's fusion weight: its slabs are summed
 * straight into the strided gradient); row_len = 0: out[c]. */
typedef struct vpu_colsum_job { const float* in; float* out; int32_t nrows, ncols, row_len, out_ld; } vpu_colsum_job;
typedef struct vpu_colsum_batch { vpu_colsum_job job[VPU_COLSUM_BATCH_MAX]; } vpu_colsum_batch;
int vpu_colsum_batched(const vpu_colsum_job* jobs, int32_t n, void* stream);
/* out[c] = beta*out[c] + sum_r in[r][c]  (fp32 in) */
int vpu_colsum_f32(const float* in, float* out, int64_t rows, int32_t C, float beta, void* stream);
/* out[c] = beta*out[c] + sum_r in[r*ld + c]  (activation dtype in; bias gradients) ; part = workspace [64][C] */
int vpu_colsum(const void* in, int32_t ld, float* out, float* part, int64_t rows, int32_t C, float beta,
               int32_t dtype, void* stream);
/* P = softmax(S) row-wise (models_vit.py:49; transformer.py:514).  S fp32 [rows][lds]; P dtype [rows][ldp];
 * columns [ncols, ldp) of P are written as zeros. */
int vpu_softmax_fwd(const float* S, int32_t lds, void* P, int32_t ldp, int64_t rows, int32_t ncols,
                    int32_t dtype, void* stream);
/* dS = P * (dP - sum(P*dP)) * scale ; dP fp32 [rows][lddp]; dS dtype [rows][ldp], pad columns zero */
int vpu_softmax_bwd(const void* P, int32_t ldp, const float* dP, int32_t lddp, void* dS, int64_t rows,
                    int32_t ncols, float scale, int32_t dtype, void* stream);
/* y = x / max(||x||_2, 1e-12) per row (F.normalize, swin_transformer.py:750-751); inv = 1/max(norm,eps) */
int vpu_l2norm_fwd(const void* x, void* y, float* inv, int64_t rows, int32_t C, int32_t dtype, void* stream);
int vpu_l2norm_bwd(const void* dy, const void* y, const float* inv, void* dx, int64_t rows, int32_t C,
                   int32_t dtype, void* stream);

/* Fused (flash-style) self-attention of the ViT blocks, bf16 (models_vit.py:43-52): q,k,v are
 * column slices of the fused qkv activation (row stride ld, head h at column h*hd); out [rows][ldo]; nb independent
 * runs of n consecutive rows (batch x windows); lse fp32 [nb*H][n] is saved for the backward. */
int vpu_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int32_t nb, int32_t H, int32_t n,
                 int32_t hd, int32_t ld, int32_t ldo, float scale, void* stream);
/* backward: dq/dk/dv (row stride ldg, same head layout) from o, d_o and lse; delta fp32 [nb*H][n] is workspace.
 * Deterministic (no atomics): one kernel owns key blocks (dk, dv), one owns query blocks (dq). */
int vpu_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                 float* delta, void* dq, void* dk, void* dv, int32_t nb, int32_t H, int32_t n, int32_t hd, int32_t ld,
                 int32_t ldo, int32_t ldg, float scale, void* stream);
/* Kernel-selection knob of the attention entry points: "lean" = 1 (default) runs the round-2 kernels (buffer-load staging
 * with hardware zero fill, thresholded running maximum, two tiles per wave), 0 the round-1 32-key-step kernels,
 * -1 = environment default (VPU_ATTN_LEAN).  Same results within bf16 rounding; both are covered by the tests.
 * "onepass": the backward of window-sized self-attention (head dim 64, n <= 256 keys = queries): 0 the dQ and dK / dV kernels,
 * 1 one workgroup per (window, head) problem, 2 key passes with three workgroups per CU, 3 (default) persistent workgroups that
 * fetch the next problem by LDS-DMA while they compute the current one (65 <= n <= 224; form 1 elsewhere; forms 1 and 3 give the
 * same bits), -1 = environment default (VPU_ATTN_ONEPASS). */
int vpu_attn_set_option(const char* name, int32_t value);
/* Name(s), as rocprofv3 prints them and separated by one space, of the kernel instantiation(s) the calling host thread's
 * last vpu_(x)attn_fwd / vpu_(x)attn_bwd call launched (the backward launches a dQ and a dK/dV kernel).  "" before the
 * first call.  Lets a test assert which head-dim path ran (ViT-H: head dim 80 in the 128-column image, 96 computed). */
const char* vpu_attn_last_kernel(void);
/* General form (the DMA neck's Attention, transformer.py:484-521): queries are rows [b*nq, (b+1)*nq) of a matrix with
 * row stride ldq, keys / values rows [b*nk, (b+1)*nk) of matrices with row stride ldk; hd % 16 == 0, hd <= 128;
 * lse / delta fp32 [nb*H][nq]; dq has row stride ldgq, dk / dv ldgk. */
int vpu_xattn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int32_t nb, int32_t H, int32_t nq,
                  int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, float scale, void* stream);
int vpu_xattn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                  float* delta, void* dq, void* dk, void* dv, int32_t nb, int32_t H, int32_t nq, int32_t nk, int32_t hd,
                  int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldgq, int32_t ldgk, float scale, void* stream);
/* Split launches of the attention kernels (round 5; the DMA neck's prompt<->image attentions, transformer.py:499-521: 48 prompt
 * tokens against 784 image tokens are 96 workgroups walking 25 key chunks each).  Batch entry b of the launch reads its queries
 * (q, o, d_o; in backward lse / delta) from entry b / qdiv and its keys / values from entry b / kdiv of the operands and writes
 * its outputs to entry b.  kdiv = S: S consecutive entries are S query ranges of one problem against the same keys -- forward
 * and dq are complete per entry, dk / dv are S partial sums (vpu_sum_groups).  qdiv = S: S key ranges of one problem for the
 * same queries -- the forward leaves S partial softmaxes (out, lse per entry: vpu_attn_combine), dq is S partial sums, dk / dv
 * are complete, lse / delta are the problem's (combined) ones.  One of qdiv / kdiv must be 1; nb counts the launch's entries. */
int vpu_xattn_fwd_split(const void* q, const void* k, const void* v, void* out, float* lse, int32_t nb, int32_t H, int32_t nq,
                        int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, float scale, int32_t qdiv, int32_t kdiv,
                        void* stream);
int vpu_xattn_bwd_split(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                        float* delta, void* dq, void* dk, void* dv, int32_t nb, int32_t H, int32_t nq, int32_t nk, int32_t hd,
                        int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldgq, int32_t ldgk, float scale, int32_t qdiv, int32_t kdiv,
                        void* stream);
/* out[b][q][h] = sum_s exp(lse_s - lse) o_s, lse = log sum_s exp(lse_s): the S partial softmaxes of vpu_xattn_fwd_split(qdiv = S)
 * (o_s bf16 [nb*S*nq][ld_s], entry b*S + s; lse_s fp32 [nb*S*H][nq]) -> out bf16 [nb*nq][ldo], lse fp32 [nb*H][nq]. */
int vpu_attn_combine(const void* o_s, const float* lse_s, void* out, float* lse, int32_t nb, int32_t H, int32_t nq, int32_t hd,
                     int32_t S, int32_t ld_s, int32_t ldo, void* stream);
/* out[g][i] = sum_s in[g][s][i], i < n (n % 8 == 0), summed in fp32 in the order of s. */
int vpu_sum_groups(const void* in, void* out, int64_t G, int32_t S, int64_t n, int32_t dtype, void* stream);

/* ---- element-wise ---- */
/* out[i] = a[i] + b[i % period_b]  (with_pos_embed, transformer.py:320, :430) */
int vpu_add_bcast(const void* a, const void* b, void* out, int64_t n, int64_t period_b, int32_t dtype, void* stream);
/* out = a + b + c + d (q_out, is_vpu_model.py:104); any of b,c,d may be NULL */
int vpu_add4(const void* a, const void* b, const void* c, const void* d, void* out, int64_t n, int32_t dtype,
             void* stream);
/* the adjoint of vpu_add4: dst[i] = src (accum[i] == 0) or dst[i] += src, i < ndst <= 4, ONE launch; dst / accum: host arrays */
int vpu_fanout_add(const void* src, void* const* dst, const int32_t* accum, int32_t ndst, int64_t n, int32_t dtype, void* stream);
/* dst[r][c] (dtype_dst, ld_dst) = src[r][c] (fp32/bf16, ld_src), zero-filling columns [cols, cols_pad) */
int vpu_cast2d(const void* src, int32_t src_dtype, int64_t ld_src, void* dst, int32_t dst_dtype, int64_t ld_dst,
               int64_t rows, int32_t cols, int32_t cols_pad, void* stream);
/* Up to 8 strided 2-D casts of fp32 sources in ONE launch: dst[map(r)][c] = src[r][c] (+ src2[r][c] when src2 != NULL, same
 * leading dimension), zeros in columns [cols, cols_pad); dst_dtype VPU_BF16 / VPU_F32; map = identity, or (perm_g > 0, rows ==
 * perm_g^2) raster -> window order of a perm_g x perm_g token grid with perm_wg-wide windows (models_vit.py:225-239).  The
 * per-step derived operands of the engine (fused patch-embed weight + bias, K-padded PuE weight, window-ordered pos_embed). */
typedef struct vpu_cast_job {
    const float* src;
    const float* src2;
    void* dst;
    int64_t ld_src, ld_dst, rows;
    int32_t cols, cols_pad, dst_dtype, perm_g, perm_wg;
} vpu_cast_job;
int vpu_cast2d_batched(const vpu_cast_job* jobs, int32_t n, void* stream);
/* Dropout2d channel mask of the segmentation head (reference: transformer_helper/decode_head.py:82-86,210-215,
 * nn.Dropout2d(0.1) in train mode): out[i] = Bernoulli(keep) / keep for n = B * channels entries, from a counter-based
 * generator keyed by (seed ^ state[1], call number, i); `state` is TWO uint64 in device memory: [0] the call number (zeroed by
 * the caller once, advanced by the launch itself, so the call is capturable in a hipGraph and every replay draws a new mask),
 * [1] a seed word the caller may change between replays (a captured launch follows it; 0 = the `seed` argument alone).
 * Same (seed ^ state[1], call number) -> same mask. */
int vpu_dropout_mask(float* out, int32_t n, float keep, uint64_t seed, uint64_t* state, void* stream);
int vpu_fill_f32(float* p, float v, int64_t n, void* stream);
/* base[off[r] .. off[r] + len[r]) <- v for n <= 160 ranges (host arrays; multiples of 4 floats from a 16-byte aligned base)
 * in one launch: the flat gradient buffer of engine.py minus the weights whose gradient the producing GEMM writes without
 * accumulating (Engine.zero_grad(lazy=True); reference semantics: optimizer.zero_grad() + backward, trainer.py:197-202). */
int vpu_fill_ranges_f32(float* base, const int64_t* off, const int64_t* len, int32_t n, float v, void* stream);
/* Diagnostic only (tools/reserve_cus_experiment.py): `wgs` workgroups of 512 threads, ~96 registers per thread, 16 KiB of
 * LDS, spinning for ~`cycles` shader cycles -- the footprint of a collective's channel workgroups. */
int vpu_debug_spin(float* sink, int32_t wgs, int64_t cycles, void* stream);
/* Diagnostic only (tools/k4_drift.py): a device buffer of 8 x uint64 per workgroup (or NULL to stop) into which the packed
 * weight-gradient kernel of vpu_gemm_grouped stamps the shader cycle counter at the start, the quarters and the end of each
 * tile's main loop -- how far the tiles of one XCD drift apart. */
int vpu_debug_gemm_times(void* dev_buf);
/* out[b][channel][:] = sigmoid(logits[b][:]) for an fp32 [B][channels][HW] tensor: the previous-mask channel of the next
 * click iteration's input (isegm/engine/trainer.py:428, :384) */
int vpu_sigmoid_to_channel(const float* logits, float* out, int32_t B, int64_t HW, int32_t channels, int32_t channel,
                           void* stream);
/* dz = dy * act'(aux) on strided 2-D views; kind 0 = ReLU (aux = its output), 1 = GELU (aux = pre-activation).
 * Backward of mmcv ConvModule's ReLU (swin_transformer.py:680-695) where no GEMM epilogue can absorb it. */
int vpu_act_bwd(const void* dy, int64_t ld_dy, const void* aux, int64_t ld_aux, void* dz, int64_t ld_dz, int64_t rows,
                int32_t cols, int32_t kind, int32_t dtype, void* stream);

/* ---- prompts (integer bookkeeping; bit-exact) ---- */
/* PuE Gaussian vectors: _guassinvector_click/_box (is_vpu_model.py:189-291) + GaussianVector(_box)
 * (ops.py:39-202).  points fp32 [B][2n][3] (row,col,order); boxes int32 [B][5] or NULL (click mode);
 * lut = the 19-tap clip (ops.py:51-61).  out dtype [B][2*num_max][ld] with ld >= 2*img+3, pad columns zero.
 * out64 (optional, may be NULL) receives the float64 rows exactly as the reference returns them. */
int vpu_pue_encode(const float* points, const int32_t* boxes, const float* lut, void* out, double* out64,
                   int32_t B, int32_t n, int32_t num_max, int32_t img, int32_t ld, int32_t dtype, void* stream);
/* Scribble prompts (prompt type 2).  vpu_pue_scribble_rows: after vpu_pue_encode (click mode), the LAST valid positive row
 * of every sample is replaced by that sample's scribble vectors vec[b] = (x profile [img] | y profile [img]) + label
 * one-hot 0 (is_vpu_model.py:294-352); a sample without a valid positive click keeps its rows.  The profiles come from the
 * host (GaussianVector_scribble, ops.py:244-296, draws from Python's `random` while deleting points: sequential by
 * construction).  out / out64 / ld as in vpu_pue_encode.
 * vpu_draw_polyline: ISModel.draw_scribble (is_model.py:123-146): cv2.polylines(image, [curve], False, 255, 3) -- the open
 * poly-line through curve[b][0..P) (int32 x, y), thickness 3, LINE_8 -- OR-ed into the positive channel of disks
 * [B][2][H][W].  OpenCV's algorithm restated (modules/imgproc/src/drawing.cpp: PolyLine -> ThickLine -> FillConvexPoly +
 * Line2 + Circle; oracle/vpu_oracle.py holds the same restatement in numpy, equal bit for bit); real cv2 is not available
 * to pin it against. */
int vpu_pue_scribble_rows(const float* points, const double* vec, void* out, double* out64, int32_t B, int32_t n,
                          int32_t num_max, int32_t img, int32_t ld, int32_t dtype, void* stream);
int vpu_draw_polyline(const int32_t* curve, float* disks, int32_t B, int32_t P, int32_t H, int32_t W, void* stream);
/* Exact Euclidean distance transform of B masks: dist[b][y][x] = distance of a non-zero pixel to the nearest zero pixel
 * (0 at zero pixels; +inf when there is none), float32(sqrt(float64)) of the exact integer squared distance -- the map
 * whose arg-max the click simulators take (clicker.py:29-56, trainer.py:628-629, 673-674, 736-737).  zero_border != 0:
 * the image is treated as surrounded by zero pixels (the reference pads the mask by one pixel first).
 * scratch: int32 [B][H][W]. */
int vpu_edt(const uint8_t* mask, int32_t* scratch, float* dist, int32_t B, int32_t H, int32_t W, int32_t zero_border,
            void* stream);
/* 5 x 5 chamfer transform = cv2.distanceTransform(mask, DIST_L2, 5) of the training simulators (trainer.py:628-629, 673-674,
 * 736-737), restated from OpenCV's two-pass fixed-point algorithm (weights 1, 1.4, 2.1969 in 16-bit fixed point); equals
 * oracle/vpu_oracle.py::chamfer_l2_5x5 bit for bit.  zero_border as in vpu_edt.  scratch: int32 [B][H + 2][W + 2].
 * W + 2 <= 1024. */
int vpu_chamfer5(const uint8_t* mask, int32_t* scratch, float* dist, int32_t B, int32_t H, int32_t W, int32_t zero_border,
                 void* stream);
/* NoBRS-loop reductions over maps that stay on the device (zoom_in.py:30-165, clicker.py:29-56).
 * vpu_mask_bbox: out[b] = {pixels with prob > thr, rmin, rmax, cmin, cmax} of image b of prob [B][H][W], the box joined
 * with the nclicks positive clicks pos_clicks (int32 (row, col) pairs; zoom_in.py:153-158 sets them in the mask first);
 * an empty mask without clicks gives {0, H, -1, W, -1}.
 * vpu_masked_argmax: per plane of dist [planes][H][W] the maximum of dist * keep (keep uint8 [H][W], shared) and the FIRST
 * raster index attaining it (numpy's tie rule), packed as (float bits of the maximum) << 32 | (0xFFFFFFFF - index). */
int vpu_mask_bbox(const float* prob, float thr, const int32_t* pos_clicks, int32_t nclicks, int32_t* out, int32_t B,
                  int32_t H, int32_t W, void* stream);
/* out uint8 [2][H][W] = (gt & !pred & valid, !gt & pred & valid): the false-negative / false-positive masks the Clicker
 * takes the distance transforms of (clicker.py:30-31); pred, gt, valid uint8 [H][W]. */
int vpu_error_masks(const uint8_t* pred, const uint8_t* gt, const uint8_t* valid, uint8_t* out, int32_t H, int32_t W,
                    void* stream);
int vpu_masked_argmax(const float* dist, const uint8_t* keep, uint64_t* out, int32_t planes, int32_t H, int32_t W,
                      void* stream);
/* 8-connected components of B masks [B][H][W]: roots[i] = smallest linear index (within the whole [B][H][W] array) of the
 * component of non-zero pixel i, -1 on zero pixels.  Sorted by root, the components are in the raster order of their first
 * pixel -- the label order of the reference's skimage.measure.label(connectivity=2) in max_connected_regions
 * (trainer.py:1175-1190). */
int vpu_cc_roots(const uint8_t* mask, int32_t* roots, int32_t B, int32_t H, int32_t W, void* stream);
/* Per-component table over the labels of vpu_cc_roots (what max_connected_regions / cal_box need of a labelling,
 * trainer.py:1061-1131,1175-1190: component sizes and bounding boxes -- no label image crosses to the host).
 * table: int32 [kmax][6] rows (root = smallest linear pixel index of the component, pixel count, ymin, ymax, xmin, xmax)
 * in no particular order, followed by ONE int32 = the number of components found (rows beyond kmax are dropped: the
 * caller compares the count with kmax).  slots: int32 scratch [B*H*W].  Integer atomics: exact, order-free. */
int vpu_cc_table(const int32_t* roots, int32_t* slots, int32_t* table, int32_t kmax, int32_t B, int32_t H, int32_t W,
                 void* stream);

/* DistMaps disks (ops.py:347-379, use_disks, spatial_scale 1) + optional draw_box outline (is_model.py:97-121:
 * cv2.rectangle((x0, y0), (x1, y1), 255, 3) = the closed 4-segment poly-line through the same restated ThickLine as
 * vpu_draw_polyline, into channel 0 (slot < n) or 1).  out fp32 [B][2][H][W] in {0,1}. */
int vpu_disk_maps(const float* points, const int32_t* boxes, float* out, int32_t B, int32_t n, int32_t H, int32_t W,
                  float radius, void* stream);

/* ---- patch embedding front end (is_model.py:59-95, ops.py:398-407, models_vit.py:94-104) ---- */
/* im2col of [normalised rgb | prev_mask | disks] into cols[B*T][6*P*P] (dtype), token rows in WINDOW order
 * (models_vit.py:225-239 folded into addressing).  image4 fp32 [B][4][H][W]; disks fp32 [B][2][H][W]. */
int vpu_patch_im2col(const float* image4, const float* disks, void* cols, int32_t B, int32_t H, int32_t W, int32_t P,
                     int32_t win_tokens, int32_t dtype, void* stream);
/* The same with the rgb planes of image4 normalised ALREADY: the public backbone_forward(image, coord_features, ...) of the
 * reference (is_vpu_model.py:383-419) receives what ISModel.prepare_input returned. */
int vpu_patch_im2col_prenorm(const float* image4, const float* disks, void* cols, int32_t B, int32_t H, int32_t W, int32_t P,
                             int32_t win_tokens, int32_t dtype, void* stream);
/* token re-ordering between window order and raster order; dir=0: raster->window, 1: window->raster.
 * x,y [B][g*g][C] */
int vpu_window_permute(const void* x, void* y, int32_t B, int32_t g, int32_t wg, int32_t C, int32_t dir, int32_t dtype,
                       void* stream);

/* ---- neck / head spatial ops (channels-last [B][H][W][C]) ---- */
/* depth-to-space for ConvTranspose2d(2, stride 2) written as a GEMM: in [B*h*w][C*4] with column (c,di,dj)
 * -> out [B][2h][2w][C], adding bias[c] (is_vpu_model.py:56-59,68).  dir=1: inverse (space-to-depth, no bias),
 * also the im2col of Conv2d(2, stride 2) (is_vpu_model.py:80). */
int vpu_pixel_shuffle2(const void* in, void* out, const float* bias, int32_t B, int32_t h, int32_t w, int32_t C,
                       int32_t dir, int32_t dtype, void* stream);
/* The backward of the depth-to-space of a ConvTranspose2d(2, stride 2) on channels-last maps (vpu_pixel_shuffle2 with
 * dir = 1) PLUS the bias gradient's partial sums in the same pass: part[block][c] = sum of in[..][c] over the fine pixels the
 * workgroup moved, vpu_pixel_unshuffle2_nblk(C) rows (a multiple of C / 8, <= 1024); the caller adds the rows up
 * (vpu_colsum_batched).  Reference: the ConvTranspose2d bias gradient of transformer_helper's FPN necks (sum over B, H, W). */
/* vpu_pixel_shuffle2 (dir 0: depth-to-space + bias) with the GroupNorm(1, C) statistics of its output in the same pass: stats =
 * the [B][vpu_groupnorm_nchunk()][2] double partials vpu_groupnorm_fwd computes in its first kernel; vpu_groupnorm_apply is
 * that function's second kernel alone (normalise [+ GELU] with given partials).  Reference: ConvTranspose2d -> GroupNorm /
 * LayerNorm2d pairs of the FPN necks (models_vit / transformer_helper SimpleFPN). */
int vpu_pixel_shuffle2_gn_stats(const void* in, void* out, const float* bias, double* stats, int32_t B, int32_t h, int32_t w,
                                int32_t C, int32_t dtype, void* stream);
int vpu_groupnorm_apply(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, const double* stats,
                        int32_t B, int64_t HW, int32_t C, float eps, int32_t gelu, int32_t dtype, void* stream);
int vpu_pixel_unshuffle2_nblk(int32_t C);
int vpu_pixel_unshuffle2_sums(const void* in, void* out, float* part, int32_t B, int32_t h, int32_t w, int32_t C,
                              int32_t dtype, void* stream);
/* GroupNorm(1, C) [+ GELU] on channels-last maps (is_vpu_model.py:57-85). stats = workspace fp64 [B][nchunk][2],
 * mean/rstd fp32 [B]. */
int vpu_groupnorm_nchunk(void);
int vpu_groupnorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, double* stats,
                      int32_t B, int64_t HW, int32_t C, float eps, int32_t gelu, int32_t dtype, void* stream);
/* dx; dw/db partials to part[B*nchunk][2][C] (reduce with vpu_colsum_f32); stats = workspace fp64 [B][nchunk][2] */
int vpu_groupnorm_bwd(const void* dy, const void* x, const float* w, const float* b, const float* mean,
                      const float* rstd, void* dx, float* part, double* stats, int32_t B, int64_t HW, int32_t C,
                      int32_t gelu, int32_t dtype, void* stream);
/* bilinear resize of channels-last maps (wrappers.py:8-28, align_corners False) in [B][h][w][C] (ld_in) ->
 * out [B][H][W][C] (ld_out: row stride in elements, lets the result land in a channel slice of the concat). */
int vpu_bilinear_cl_fwd(const void* in, int32_t ld_in, void* out, int32_t ld_out, int32_t B, int32_t h, int32_t w,
                        int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
int vpu_bilinear_cl_bwd(const void* dout, int32_t ld_out, void* din, int32_t ld_in, int32_t B, int32_t h, int32_t w,
                        int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
/* Head fusion without the channel concat (swin_transformer.py:744-756: 1x1 fusion_conv over cat(resize(convs_i(x_i)))):
 * by linearity io[B][H][W][C] = relu(io + sum_{i<n} resize(z_i)), z_i [B][h_i][w_i][C] = convs_i output times its column
 * block of the fusion weight at low resolution, io = the full-resolution level's product + bias; bilinear,
 * align_corners False (wrappers.py:8-28); in place; n <= 3. */
int vpu_upsum_relu(void* io, const void* const* z, const int32_t* h, const int32_t* w, int32_t n, int32_t B, int32_t H,
                   int32_t W, int32_t C, int32_t dtype, void* stream);
/* Head backward of the fused map in one pass (bf16, C in {64,128,256,512}): the gradient through the L2 normalisation of
 * the P2CL branch (dfn = gradient of fn = y, inv = 1/|fused| from vpu_l2norm_fwd) + conv_seg's input gradient (dout [rows]
 * fp32, w [C], mask [B][C] or NULL) with the ReLU' of the fused map x applied to the sum; conv_seg's weight / bias gradient
 * partials go to part[nblk][C] / part_b[nblk] with nblk = vpu_convseg_bwd_nblk(rows)
 * (swin_transformer.py:744-767, decode_head.py:210-215).  Equals vpu_l2norm_bwd followed by vpu_convseg_bwd(accum = 3). */
int vpu_head_grad_fused(const void* dfn, const void* y, const float* inv, const float* dout, const void* x, const float* w,
                        const float* mask, void* dx, float* part, float* part_b, int64_t rows, int64_t HW, int32_t C,
                        int32_t dtype, void* stream);
/* DMA gates (is_vpu_model.py:106-121): cg[b][c] = sigmoid(max_q Q[b][q][c]), sg[b][n] = sigmoid(max_c K[b][n][c]),
 * out = x*(1+cg+sg). arg* record the arg-max for backward. */
int vpu_gate_stats(const void* Q, const void* Kt, float* cg, int32_t* argq, float* sg, int32_t* argc, int32_t B,
                   int32_t nq, int32_t N, int32_t C, int32_t dtype, void* stream);
int vpu_gate_apply(const void* x, const float* cg, const float* sg, void* out, int32_t B, int32_t N, int32_t C,
                   int32_t dtype, void* stream);
/* backward: dx (+)= dout*(1+cg+sg) (accum!=0 adds into dx); dQ[b][argq][c] += dcg*cg*(1-cg);
 * dK[b][n][argc] += dsg*sg*(1-sg).  dQ / dK must already hold the upstream gradient (they are accumulated into).
 * part = workspace fp32 [B][64][C]. */
int vpu_gate_bwd(const void* dout, const void* x, const float* cg, const int32_t* argq, const float* sg,
                 const int32_t* argc, void* dx, int32_t accum, void* dQ, void* dK, float* part, int32_t B, int32_t nq,
                 int32_t N, int32_t C, int32_t dtype, void* stream);
/* The three gates of SimpleFPN (is_vpu_model.py:106-121: the same x against the three (queries, keys) pairs of the DMA neck)
 * in one launch per pass: vpu_gate_fwd_n = the statistics of all n <= 3 gates (ONE launch) + out[i] = x (1 + cg_i + sg_i) for
 * every gate (ONE launch, x read once); cg / argq [n][B][C], sg / argc [n][B][N].  vpu_gate_bwd_n: dout[i] = the gradient of
 * out[i]; dx (+)= sum_i dout_i (1 + cg_i + sg_i), summed in fp32 and rounded once; dQ[i] / dK[i] as in vpu_gate_bwd;
 * part = workspace fp32 [n][B][64][C].  Q, K, out, dout, dQ, dK: HOST arrays of n device pointers. */
int vpu_gate_fwd_n(const void* const* Q, const void* const* K, const void* x, void* const* out, float* cg, int32_t* argq,
                   float* sg, int32_t* argc, int32_t n, int32_t B, int32_t nq, int32_t N, int32_t C, int32_t dtype, void* stream);
int vpu_gate_bwd_n(const void* const* dout, const void* x, const float* cg, const int32_t* argq, const float* sg,
                   const int32_t* argc, void* dx, int32_t accum, void* const* dQ, void* const* dK, float* part, int32_t n,
                   int32_t B, int32_t nq, int32_t N, int32_t C, int32_t dtype, void* stream);
/* conv_seg: Dropout2d + 1x1 conv to one channel (decode_head.py:210-215).  x [rows][C] channels-last,
 * rows = B*HW; mask fp32 [B][C] (keep/(1-p)) or NULL; out fp32 [rows]. */
int vpu_convseg_fwd(const void* x, const float* w, const float* bias, const float* mask, float* out, int64_t rows,
                    int64_t HW, int32_t C, int32_t dtype, void* stream);
/* dx[r][c] = dout[r]*w[c]*mask (accum bit 0: added into dx; bit 1: the result is multiplied by [x > 0], i.e. x is a ReLU
 * output whose gradient is complete with this call and dx leaves as the pre-activation gradient);
 * dw partials -> part[nblk][C]; db partial -> part_b[nblk] */
int vpu_convseg_bwd_nblk(int64_t rows);
int vpu_convseg_bwd(const float* dout, const void* x, const float* w, const float* mask, void* dx, int32_t accum,
                    float* part, float* part_b, int64_t rows, int64_t HW, int32_t C, int32_t dtype, void* stream);

/* ---- output + losses ---- */
/* F.interpolate(bilinear, align_corners=True) of fp32 planes [P][h][w] -> [P][H][W] (is_vpu_model.py:431-436) */
int vpu_upsample_ac_fwd(const float* in, float* out, int64_t planes, int32_t h, int32_t w, int32_t H, int32_t W,
                        void* stream);
int vpu_upsample_ac_bwd(const float* dout, float* din, int64_t planes, int32_t h, int32_t w, int32_t H, int32_t W,
                        void* stream);
/* P2CL = SigmoidBinaryCrossEntropyLoss(from_sigmoid=True) (losses.py:155-176) against ed_mask_label
 * (trainer.py:329-331,756,764) built on the fly: label[b][s] = gt[b] for s < S/2, 1-gt[b] otherwise, unless
 * slot_mask_idx[b][s] >= 0, in which case it is override[slot_mask_idx[b][s]] (an error mask [H][W]).
 * prob fp32 [B][S][H][W]; loss_part fp32 [B][S] (per-plane sums); dprob (optional) = grad_scale * dloss/dprob. */
int vpu_p2cl_fwd_bwd(const float* prob, const float* gt, const int32_t* slot_mask_idx, const float* override_masks,
                     float* loss_part, float* dprob, float grad_scale, int32_t B, int32_t S, int32_t H, int32_t W,
                     void* stream);
/* The same loss taken on the LOW-resolution similarities: fuses the align_corners=True upsample (is_vpu_model.py:434-436),
 * the loss and both backward passes; sim_low fp32 [B][S][h][w], dsim_low (optional) its gradient.
 * loss_part fp32 [B][S][vpu_p2cl_up_nband(h, w)]: un-normalised sums per (plane, band of low-resolution rows). */
int vpu_p2cl_up_nband(int32_t h, int32_t w);
int vpu_p2cl_up_fwd_bwd(const float* sim_low, const float* gt, const int32_t* slot_mask_idx, const float* override_masks,
                        float* loss_part, float* dsim_low, float grad_scale, int32_t B, int32_t S, int32_t h, int32_t w,
                        int32_t H, int32_t W, void* stream);
/* NormalizedFocalLossSigmoid(alpha .5, gamma 2) + naive Dice on logits [B][HW] vs gt (losses.py:11-89,227-363).
 * sums fp64 [B][8] workspace; out fp32 [B][2] = (nfl_b, dice_b); dlogits = w_nfl*dNFL + w_dice*dDice (means over B
 * folded into w_*). */
/* sums: scratch of vpu_nfl_dice_scratch_doubles(B) doubles (per-chunk partial sums; need not be initialised) */
int vpu_nfl_dice_scratch_doubles(int32_t B);
int vpu_nfl_dice_fwd_bwd(const float* logits, const float* gt, double* sums, float* out, float* dlogits, float w_nfl,
                         float w_dice, int32_t B, int64_t HW, void* stream);

/* The logged scalars of one click iteration (trainer.py:399-419) from the loss kernels' partials, one launch:
 * res[4] = {total, nfl, dice, p2cl}; out [B][2] from vpu_nfl_dice_fwd_bwd, part [npart] from vpu_p2cl(_up)_fwd_bwd,
 * p2cl = sum(part) * inv_count, total = (w_nfl nfl + w_dice dice + w_pcl p2cl) * iter_weight. */
int vpu_loss_finalize(const float* out, const float* part, int32_t B, int32_t npart, double inv_count, float w_nfl,
                      float w_dice, float w_pcl, float iter_weight, float* res, void* stream);

/* ---- optimizer (torch.optim.Adam as configured at vpu_base448_cocolvis.py:149-154) ---- */
/* p,g,m,v fp32 [n]; shadow (bf16, optional) receives the rounded new parameters; lr_mult (optional) fp32 [n_seg] with
 * seg_of (int32 [n/seg_gran]) is not used in v1 (uniform lr). */
int vpu_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, float grad_scale, void* stream);
/* The same step with per-tensor learning rate and weight decay (layer-wise lr decay isegm/utils/lr_decay.py:15-66,
 * lr_mult isegm/engine/optimizer.py:15-17): segment s covers elements [seg_end[s-1], seg_end[s]) of the flat buffer
 * (device arrays; segments start on 8-element boundaries); decoupled_wd 0 = torch.optim.Adam (L2), 1 = AdamW. */
int vpu_adam_step_groups(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                         const int64_t* seg_end, const float* seg_lr, const float* seg_wd, int32_t nseg, float beta1,
                         float beta2, float eps, int32_t decoupled_wd, int32_t step, float grad_scale, void* stream);
/* The same update with the step-dependent scalars read from DEVICE memory, so that the launch can be captured in a
 * hipGraph and replayed: hyper fp32 [4] = {lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale}, uploaded by the host before
 * each replay.  nseg > 0: per-tensor lr = hyper[0] * seg_scale[s], weight decay seg_wd[s]; nseg == 0: weight_decay. */
int vpu_adam_step_hyper(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, const float* hyper,
                        const int64_t* seg_end, const float* seg_scale, const float* seg_wd, int32_t nseg, float beta1,
                        float beta2, float eps, float weight_decay, int32_t decoupled_wd, void* stream);


#ifdef __cplusplus
}
#endif
#endif
