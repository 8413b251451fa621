// GEMM for the VPUFormer hot path on gfx950 (MI355X).
//
//   C[M,N] = epilogue(alpha * op(A)[M,K] * op(B)[K,N])
//
// bf16 path: v_mfma_f32_16x16x32_bf16, 128x128x64 block tile, 4 waves (2x2) of 64x64, register-staged
// global->LDS with XOR-swizzled LDS images.  K-contiguous operands ([rows][K]) are read back with
// ds_read_b128; K-major operands ([K][cols], i.e. the "transposed" inputs of dgrad / wgrad / P.V) are read with
// the gfx950 transposing LDS read ds_read_b64_tr_b16, so no operand is ever transposed through HBM.
// f32 path (parity mode): v_mfma_f32_16x16x4_f32 -- bit-for-bit a k-ordered fp32 fma chain.
//
// Replaces cuBLAS/cuDNN calls behind nn.Linear / Conv2d / ConvTranspose2d / matmul of the reference
// (isegm/model/modeling/models_vit.py:38-52,16-27,91; transformer.py:484-517; is_vpu_model.py:55-86;
// swin_transformer.py:680-756).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "vpu_common.h"
#include "../../include/vpu_hip.h"
#include "gemm_tiles.h"

// K5 (gemm_k5.hip): the two-tile ping-pong form of the many-tile forward / dgrad launches
int vpu_k5_option();
void vpu_k5_set_option(int v);
void vpu_k5_set_grid(int v);
void vpu_k5_set_split(int v);
void vpu_k5_set_noepi(int v);
int vpu_k5_launch(const vpu_gemm_desc* d, int rb, int ncu, int vec, void* stream, char* name, size_t name_len);
#ifdef VPU_DIAG
void vpu_k5_set_dbg(unsigned long long* p);
#endif

namespace {


// ------------------------------------------------------------------------------------------------
// epilogue shared by both precisions
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void epilogue_store(const vpu_gemm_desc& p, int64_t coff, int64_t roff, int m, int n,
                                               float acc) {
    const int flags = p.flags;
    float v = acc * p.alpha;
    if (flags & VPU_EPI_BIAS) v += p.bias[n];
    const int64_t ci = coff + (int64_t)m * p.ldc + n;
    if (flags & VPU_EPI_PREACT) reinterpret_cast<T*>(p.preact)[ci] = from_f32<T>(v);
    if (flags & VPU_EPI_SAVE_DGELU) reinterpret_cast<T*>(p.preact)[ci] = from_f32<T>(dgelu_f(v));
    if (flags & VPU_EPI_GELU) v = gelu_f(v);
    if (flags & VPU_EPI_RELU) v = fmaxf(v, 0.f);
    if (flags & (VPU_EPI_DGELU | VPU_EPI_DRELU)) {
        const float a = to_f32(reinterpret_cast<const T*>(p.aux)[coff + (int64_t)m * p.ldaux + n]);
        v *= (flags & VPU_EPI_DGELU) ? dgelu_f(a) : (a > 0.f ? 1.f : 0.f);
    }
    if (flags & VPU_EPI_MULAUX) v *= to_f32(reinterpret_cast<const T*>(p.aux)[coff + (int64_t)m * p.ldaux + n]);
    if (flags & VPU_EPI_RESID) {
        const T* r = reinterpret_cast<const T*>(p.resid);
        const int64_t ri = p.resid_period > 0 ? (int64_t)(m % p.resid_period) * p.ldr + n
                                              : roff + (int64_t)m * p.ldr + n;
        v += to_f32(r[ri]);
    }
    if (flags & VPU_EPI_AFFINE) v = v * p.post_mul + p.post_add;
    if (flags & VPU_EPI_OUT_F32) {
        float* c = reinterpret_cast<float*>(p.C);
        if (flags & VPU_EPI_ACCUM) v += c[ci];
        c[ci] = v;
    } else {
        T* c = reinterpret_cast<T*>(p.C);
        if (flags & VPU_EPI_ACCUM) v += to_f32(c[ci]);
        c[ci] = from_f32<T>(v);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------

// Register staging of one tile: 4 x 16 B per thread, global -> VGPR -> ds_write_b128 into the swizzled image.
template <int TRANS>
struct TileLoader {
    uint4 r[4];
    __device__ __forceinline__ void load(const bf16_t* base, int ld, int x0, int X, int k0, int K, int tid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (TRANS == 0) {
                const int row = c >> 3, kc = c & 7;
                const int gx = x0 + row, gk = k0 + kc * 8;
                if (gx < X && gk < K) v = *reinterpret_cast<const uint4*>(base + (int64_t)gx * ld + gk);
            } else {
                const int k = c >> 4, cc = c & 15;
                const int gk = k0 + k, gx = x0 + cc * 8;
                if (gk < K && gx < X) v = *reinterpret_cast<const uint4*>(base + (int64_t)gk * ld + gx);
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256;
            const int off = TRANS == 0 ? kc_off(c >> 3, c & 7) : km_off(c >> 4, (c & 15) * 2);
            *reinterpret_cast<uint4*>(lds + off) = r[i];
        }
    }
};


// LDS-DMA staging of one 128x64 (K-contiguous) or 64x128 (K-major) bf16 tile: 16 pieces of 1 KiB, each one
// `buffer_load_dwordx4 ... lds` wave-instruction (64 lanes x 16 B, landing at piece_base + 16*lane, no VGPRs).  The LDS
// image is linear per piece, so the XOR swizzle is applied to the per-lane SOURCE chunk (cdna guide, rule 21); rows /
// k beyond the operand (ragged M, N, K, split-K slice ends) get the out-of-range offset and arrive as zeros.
template <int TRANS>
__device__ __forceinline__ void stage_piece(__amdgpu_buffer_rsrc_t rsrc, int ld, int x0, int X, int k0, int kend,
                                            char* lds_tile, int piece, int lane) {
    int voff;
    if (TRANS == 0) {
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int gx = x0 + row, gk = k0 + chunk * 8;
        voff = (gx < X && gk < kend) ? (gx * ld + gk) * 2 : OOB_OFFSET;
    } else {
        const int k = piece * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ ((k & 3) << 1) ^ (((k >> 3) & 1) << 3);
        const int gk = k0 + k, gx = x0 + chunk * 8;
        voff = (gk < kend && gx < X) ? (gk * ld + gx) * 2 : OOB_OFFSET;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_vptr)(lds_tile + piece * 1024), 16, voff, 0, 0, 0);
}

template <int TRANS>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, int ld, int x0, int X, int k0, int kend,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave + 4 * i;
        int voff;
        if (TRANS == 0) {
            const int row = piece * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const int gx = x0 + row, gk = k0 + chunk * 8;
            voff = (gx < X && gk < kend) ? (gx * ld + gk) * 2 : OOB_OFFSET;
        } else {
            const int k = piece * 4 + (lane >> 4);
            const int chunk = (lane & 15) ^ ((k & 3) << 1) ^ (((k >> 3) & 1) << 3);
            const int gk = k0 + k, gx = x0 + chunk * 8;
            voff = (gk < kend && gx < X) ? (gk * ld + gx) * 2 : OOB_OFFSET;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_vptr)(lds_tile + piece * 1024), 16, voff, 0, 0, 0);
    }
}


// 8 consecutive output columns of one row through the full epilogue with 16-byte accesses.
// Preconditions (checked on the host, flag `vec`): n % 8 == 0, n + 8 <= N, every leading dimension / batch offset /
// base pointer involved is a multiple of 8 elements (16 B for bf16, 32 B for fp32).
__device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8]) {
    const bf16x8_t e = __builtin_bit_cast(bf16x8_t, u);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)e[j];
}
// prefetch of the residual / aux values one lane will need in the epilogue: [pass][t] as laid out by the LDS staging
__device__ __forceinline__ void prefetch_epi(const vpu_gemm_desc& p, const int flags, int64_t coff, int64_t roff,
                                             int mbase, int nbase, int lane, uint4 (&pre)[2][4]) {
    const bool is_res = (flags & VPU_EPI_RESID) != 0;
    const bf16_t* src = reinterpret_cast<const bf16_t*>(is_res ? p.resid : p.aux);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int u = lane + 64 * t;
            const int m = mbase + pass * 32 + (u >> 3), n = nbase + (u & 7) * 8;
            pre[pass][t] = make_uint4(0, 0, 0, 0);
            if (m < p.M && n + 8 <= p.N) {
                const int64_t idx = is_res ? (p.resid_period > 0 ? (int64_t)(m % p.resid_period) * p.ldr + n
                                                                 : roff + (int64_t)m * p.ldr + n)
                                           : coff + (int64_t)m * p.ldaux + n;
                pre[pass][t] = *reinterpret_cast<const uint4*>(src + idx);
            }
        }
}
// `pre` (optional): the 8 bf16 of the residual (VPU_EPI_RESID) or of aux (DGELU/DRELU/MULAUX) for this position, fetched
// before the main loop so that their HBM latency is hidden behind the MFMA work.
// (everything by value / whole-array reference: a pointer to one element of a local array forces it into scratch)
struct EpiPre {
    bool has_pre, has_bias;
    uint4 pre;        // 8 bf16 of resid or aux for this position
    float bias[8];
};
__device__ __forceinline__ void epilogue_store8(const vpu_gemm_desc& p, const int flags, int64_t coff, int64_t roff,
                                                int m, int n, float (&v)[8], const EpiPre& e) {
    const bool has_pre = e.has_pre;
    const uint4* pre = &e.pre;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
    if (flags & VPU_EPI_BIAS) {
        if (e.has_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += e.bias[j];
        } else {
            float b[8];
            load8(p.bias + n, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += b[j];
        }
    }
    const int64_t ci = coff + (int64_t)m * p.ldc + n;
    if (flags & VPU_EPI_PREACT) store8(reinterpret_cast<bf16_t*>(p.preact) + ci, v);
    if (flags & (VPU_EPI_SAVE_DGELU | VPU_EPI_GELU)) {
        // gelu = x*Phi(x), gelu' = Phi(x) + x*pdf(x) from ONE exp and ONE rcp per element: with u = exp(-x^2/2),
        // erf(|x|/sqrt2) = 1 - poly(t)*u (Abramowitz-Stegun 7.1.26 on z = |x|/sqrt2, t = 1/(1+0.3275911 z),
        // |error| <= 1.5e-7, far below bf16 resolution) and pdf(x) = u/sqrt(2 pi).  (The fp32 parity path keeps erff.)
        float d[8];
        gelu_dgelu8(v, d);
        if (flags & VPU_EPI_SAVE_DGELU) store8(reinterpret_cast<bf16_t*>(p.preact) + ci, d);
    }
    if (flags & VPU_EPI_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    if (flags & (VPU_EPI_DGELU | VPU_EPI_DRELU | VPU_EPI_MULAUX)) {
        float a[8];
        if (has_pre) unpack8(*pre, a);
        else load8(reinterpret_cast<const bf16_t*>(p.aux) + coff + (int64_t)m * p.ldaux + n, a);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] *= (flags & VPU_EPI_MULAUX) ? a[j] : ((flags & VPU_EPI_DGELU) ? dgelu_f(a[j]) : (a[j] > 0.f ? 1.f : 0.f));
    }
    if (flags & VPU_EPI_RESID) {
        float r[8];
        if (has_pre) unpack8(*pre, r);
        else {
            const int64_t ri = p.resid_period > 0 ? (int64_t)(m % p.resid_period) * p.ldr + n
                                                  : roff + (int64_t)m * p.ldr + n;
            load8(reinterpret_cast<const bf16_t*>(p.resid) + ri, r);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
    }
    if (flags & VPU_EPI_AFFINE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * p.post_mul + p.post_add;
    }
    if (flags & VPU_EPI_OUT_F32) {
        float* c = reinterpret_cast<float*>(p.C) + ci;
        if (flags & VPU_EPI_ACCUM) {
            float o[8];
            load8(c, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += o[j];
        }
        store8(c, v);
    } else {
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + ci;
        if (flags & VPU_EPI_ACCUM) {
            float o[8];
            load8(c, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += o[j];
        }
        store8(c, v);
    }
}

// grid: x = tiles_m * tiles_n, y = split-K slices, z = batch.
// splitk > 1: every slice writes its raw fp32 partial tile to ws[(z*splitk + slice)][M][N]; splitk_reduce_kernel sums
// the slices in a fixed order (deterministic) and applies the epilogue.
// DMA = false: register staging, one 32-KiB LDS stage, ~3 blocks per CU (default: measured faster at this tile size).
// DMA = true : LDS-DMA staging, two 32-KiB stages, one barrier per K-tile (kept selectable: VPU_GEMM_DMA=1).
// CS: with the fused bias-gradient column sums (only instantiated for TA = 1); kept out of the plain kernels so that they
// do not carry its 16 accumulators + ones fragment.  __launch_bounds__(256, 2): two workgroups per CU.
// FL >= 0: the epilogue flags are the compile-time constant FL (host guarantees: vector epilogue, no split-K, N % 8 == 0)
// -- the generic epilogue tests a dozen flags per 8-element group at run time, ~15k lines of ISA for one kernel; the
// specialised ones are straight-line code.  FL = -1: generic (run-time flags, scalar tails, split-K slabs, diagnostics).
// One K-tile of the ring loop: DMA of a later K-tile into stage `wr`, fragments of the current one from stage `rd`.
// The two stages are __restrict__ parameters of an inlined function on purpose: the scoped no-alias information this
// leaves on the LDS-DMA and on the ds_reads is what stops SIInsertWaitcnts from putting s_waitcnt vmcnt(0) in front of
// the reads while a DMA is pending (it cannot tell the stages of one LDS array apart otherwise).
template <int TA, int TB, bool CS>
__device__ __forceinline__ void ring_step(__amdgpu_buffer_rsrc_t rA, __amdgpu_buffer_rsrc_t rB, int lda, int ldb, int m0,
                                          int M, int n0, int N, int k0, int kend, char* __restrict__ wr,
                                          const char* __restrict__ rd, int wave, int lane, int wm, int wn, bool do_cs,
                                          bf16x8_t ones, f32x4_t (&acc)[4][4], f32x4_t (&acc_cs)[4]) {
    stage_tile<TA>(rA, lda, m0, M, k0, kend, wr, wave, lane);
    stage_tile<TB>(rB, ldb, n0, N, k0, kend, wr + TILE_BYTES, wave, lane);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = read_frag<TA>(rd, wm * 64 + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = read_frag<TB>(rd + TILE_BYTES, wn * 64 + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if (CS && TA == 1 && do_cs) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, acc_cs[i], 0, 0, 0);
        }
    }
}

// RING = S > 0: S LDS stages of 32 KiB in a ring, the DMA of K-tile kt+S-1 is issued while tile kt is computed, a COUNTED
// s_waitcnt vmcnt(8*(S-2)) (a stage = 8 DMA instructions per wave; past the end of K the pieces are requested out of
// range, which costs no memory traffic and keeps the count uniform) + one raw s_barrier per K-tile.  For the small
// problems of the DMA neck (one tile per workgroup, <= 24 K-tiles): the two-stage loop pays one full L2/HBM round trip
// per K-tile there (~1.2 us), which split-K + a reduce launch used to paper over.
// 8 floats written through to memory (sc1): a slab that another workgroup of the same launch will read needs no release
// fence when every byte of it is stored this way and drained (s_waitcnt vmcnt(0)) before the arrival counter is bumped
// (cdna guide, Guideline 16, rule R1)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store8_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, const float (&v)[8]) {
    u32x4_t a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = __builtin_bit_cast(unsigned, v[j]); b[j] = __builtin_bit_cast(unsigned, v[4 + j]); }
    __builtin_amdgcn_raw_buffer_store_b128(a, r, byte_off, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(b, r, byte_off + 16, 0, 16);
}

constexpr int FL_SLAB = 0x10000;
// GRP: the work list is the concatenation of the tiles of up to VPU_GEMM_GROUP_MAX independent problems (same operand
// layouts, no split-K, batch 1), descriptors read from the kernel-argument segment -- one launch for e.g. the four weight
// gradients of a ViT block (432 full-K tiles: one round of the 512 persistent workgroups, no slabs, no reduce launches).
template <int TA, int TB, bool DMA, bool CS, int FL, int RING, bool GRP>
__device__ __forceinline__ void gemm_bf16_body(const vpu_gemm_desc& p_arg, const vpu_gemm_group* __restrict__ ga,
                                               const int tiles_n_arg, const int splitk, const int kchunk_arg,
                                               float* __restrict__ ws, const int vec_in, const int tiles_m_arg,
                                               const int nbatch, unsigned* __restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // 32 KiB (register staging) or 64 KiB (DMA)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // PERSISTENT tile loop: at most two workgroups per CU walk the (tile, split-K slice, batch) work list.  A workgroup
    // that exits holds its CU slot until its stores are acknowledged and its successor pays launch + first-tile latency;
    // at K = 768 that bubble was 30-45 % of a tile's life (round-1 measurement: main loop 880 vs 600 TFLOP/s end to end).
    const int ntiles0 = tiles_n_arg * tiles_m_arg;
    const int total_work = GRP ? ga->start[ga->n] : ntiles0 * splitk * nbatch;
    const int vec = vec_in & 255;
    constexpr bool GEN = FL < 0;
    constexpr bool SLAB = FL == FL_SLAB;   // split-K slice: the raw fp32 tile goes to its workspace slab, nothing else
    if (!GRP && (vec_in >> 8)) {
        // diagnostic, VPU_GEMM_STAGGER=c: the workgroups that own one work item FEWER than the others (total_work %
        // gridDim != 0) start about half a tile late (c cycles per K-tile), so that their epilogue (HBM stores) falls into
        // the main loop (L2 -> LDS traffic) of the workgroup they share the CU with.  Measured round 1 on the six ViT-B
        // forward / dgrad shapes with c = 300..1300: no effect beyond noise (fc1 82.9 -> 81.1..85.6 us) -- kept off.
        const int rem = total_work % (int)gridDim.x;
        if (rem != 0 && (int)blockIdx.x >= rem && total_work > (int)gridDim.x) {
            const int nk_all = (p_arg.K / splitk + BK - 1) / BK;
            const int steps = (nk_all * (vec_in >> 8)) >> 10;   // s_sleep(16) ~ 1024 cycles
            for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(16);
        }
    }
    for (int work = blockIdx.x; work < total_work; work += gridDim.x) {
    int grp = 0;
    if constexpr (GRP) {
        while (grp + 1 < ga->n && work >= ga->start[grp + 1]) ++grp;
    }
    const vpu_gemm_desc& p = GRP ? ga->d[grp] : p_arg;
    const int tiles_n = GRP ? (p.N + BN - 1) / BN : tiles_n_arg;
    const int ntiles = GRP ? ga->start[grp + 1] - ga->start[grp] : ntiles0;
    const int kchunk = GRP ? (p.K + BK - 1) / BK * BK : kchunk_arg;
    const int FLG = GEN ? p.flags : (SLAB ? 0 : FL);
    int tile_lin = GRP ? work - ga->start[grp] : work % ntiles, rest = GRP ? 0 : work / ntiles;
    int tile_m, tile_n;
    if (!GRP && splitk > 1) {
        // split-K: the few output tiles of ONE reduction slice read the same K-range of A and B -- put them on one XCD,
        // next to each other in its dispatch order, so that they share the panels through that XCD's L2 (in the plain
        // order the four 128 x 128 tiles of a 256 x 256 weight gradient sat on four XCDs and every panel came from HBM
        // twice: 308 MB for 154 MB of operands in the FPN's 150528-row reductions).  XCD x takes a contiguous range of
        // the (slice-major, tile-minor) order.
        const int xcd = work & 7, q = total_work >> 3, r = total_work & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (work >> 3);
        tile_lin = v % ntiles; rest = v / ntiles;
        const int tiles_m = ntiles / tiles_n;
        tile_n = tile_lin / tiles_m; tile_m = tile_lin - tile_n * tiles_m;
    } else {
        tile_coords(tile_lin, ntiles, tiles_n, tile_m, tile_n);
    }
    const int split = rest % splitk, z = rest / splitk;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int zo = z / p.inner, zi = z % p.inner;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + zo * p.sAo + zi * p.sAi;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + zo * p.sBo + zi * p.sBi;
    const int64_t coff = zo * p.sCo + zi * p.sCi;
    const int64_t roff = zo * p.sRo + zi * p.sRi;
    const int kbeg = split * kchunk;
    const int kend = (kbeg + kchunk < p.K) ? kbeg + kchunk : p.K;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, 0x7FFFFFFF, 0x00020000);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // fused bias gradient (weight-gradient form, A = dY K-major): colsum[m] += sum_k A[m][k] as one extra MFMA per
    // A fragment against a fragment that is 1 in output column 0 -- only the wn == 0 waves of the tile_n == 0 blocks.
    const bool do_cs = CS && TA == 1 && p.colsum != nullptr && tile_n == 0 && wn == 0;
    f32x4_t acc_cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_cs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)(((lane & 15) == 0) ? 1.0f : 0.0f);

    const int nk = (kend - kbeg + BK - 1) / BK;
    if constexpr (RING > 0) {
        static_assert(RING == 3 || RING == 4, "ring depth");
#pragma unroll
        for (int st = 0; st < RING - 1; ++st) {   // k0 >= kend: every piece is requested out of range (zeros, no traffic)
            stage_tile<TA>(rA, p.lda, m0, p.M, kbeg + st * BK, kend, lds + st * 2 * TILE_BYTES, wave, lane);
            stage_tile<TB>(rB, p.ldb, n0, p.N, kbeg + st * BK, kend, lds + st * 2 * TILE_BYTES + TILE_BYTES, wave, lane);
        }
        for (int kt = 0; kt < nk; ++kt) {
            if (RING == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // tile kt has landed for every wave; everyone is done reading tile kt-1
            ring_step<TA, TB, CS>(rA, rB, p.lda, p.ldb, m0, p.M, n0, p.N, kbeg + (kt + RING - 1) * BK, kend,
                                  lds + ((kt + RING - 1) % RING) * 2 * TILE_BYTES, lds + (kt % RING) * 2 * TILE_BYTES, wave,
                                  lane, wm, wn, do_cs, ones, acc, acc_cs);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range tail pieces still write (zeros) into LDS
        __syncthreads();
    } else {
    if (DMA) {
        stage_tile<TA>(rA, p.lda, m0, p.M, kbeg, kend, lds, wave, lane);
        stage_tile<TB>(rB, p.ldb, n0, p.N, kbeg, kend, lds + TILE_BYTES, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    TileLoader<TA> la;
    TileLoader<TB> lb;
    if (!DMA) {
        la.load(A, p.lda, m0, p.M, kbeg, kend, tid);
        lb.load(B, p.ldb, n0, p.N, kbeg, kend, tid);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const char* ldsA = DMA ? lds + (kt & 1) * 2 * TILE_BYTES : lds;
        const char* ldsB = ldsA + TILE_BYTES;
        if (!DMA) {
            la.store(lds, tid);
            lb.store(lds + TILE_BYTES, tid);
            __syncthreads();
            if (kt + 1 < nk) {  // global loads of the next tile fly during this tile's MFMAs
                la.load(A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, tid);
                lb.load(B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, tid);
            }
        }
        // All fragment reads of this K-tile are issued BEFORE the DMA of the next tile: hipcc puts s_waitcnt vmcnt(0) in
        // front of any ds_read that follows a pending LDS-DMA in program order (it cannot prove the stages disjoint),
        // which would drain the prefetch before the tile is computed (seen in the ISA, round 1).  In this order the DMA
        // flies under the 32 MFMAs of the tile.
        bf16x8_t af[2][4], bfr[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = read_frag<TA>(ldsA, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[ks][j] = read_frag<TB>(ldsB, wn * 64 + j * 16, ks, lane);
        }
        if (DMA) {
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) {
                char* nxt = lds + ((kt + 1) & 1) * 2 * TILE_BYTES;
                stage_tile<TA>(rA, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, nxt, wave, lane);
                stage_tile<TB>(rB, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, nxt + TILE_BYTES, wave, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the wave that holds its fragments owns the MFMA pipe; the co-resident workgroup's wave reads / waits in the gaps
        // (+3-5 % on the main loop, tools/gemm_bench.py with VPU_GEMM_NOEPI=1)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
            if (CS && TA == 1 && do_cs) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], ones, acc_cs[i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (DMA) {
            // keep the (register-only) MFMAs above the wait: hipcc otherwise sinks them below the barrier and the DMA
            // latency is exposed again (guide rule 18)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    }  // RING == 0

    if (GEN && vec == 9) {  // diagnostic (VPU_GEMM_NOEPI=1): main loop only; the impossible compare keeps the accumulators live
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 1.2345678e30f) reinterpret_cast<float*>(p.C)[0] = t;
        continue;
    }
    // Everything the epilogue READS from global memory is requested here, before its first store: vmcnt counts stores
    // too on CDNA4, so a load issued after a store makes the wave wait for that store's round trip (the ISA of the first
    // version had 8 such serialised waits per tile: 12-30 us per GEMM).  A lane always handles the same 8 columns.
    const bool use_pre = (!GEN || (vec == 1 && splitk == 1)) &&
                         (FLG & (VPU_EPI_RESID | VPU_EPI_MULAUX | VPU_EPI_DGELU | VPU_EPI_DRELU)) != 0 &&
                         !((FLG & VPU_EPI_RESID) && (FLG & (VPU_EPI_MULAUX | VPU_EPI_DGELU | VPU_EPI_DRELU)));
    uint4 pre[2][4];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_) pre[a_][b_] = make_uint4(0, 0, 0, 0);
    if (use_pre) prefetch_epi(p, FLG, coff, roff, m0 + wm * 64, n0 + wn * 64, lane, pre);
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int ncol = n0 + wn * 64 + (lane & 7) * 8;
    const bool use_bias8 = (!GEN || (vec == 1 && splitk == 1)) && (FLG & VPU_EPI_BIAS) && ncol + 8 <= p.N;
    if (use_bias8) load8(p.bias + ncol, bias8);
    // ---- epilogue: transpose the accumulators through LDS so that every lane owns 8 consecutive columns of one row
    // and all global accesses are 16-byte vectors.  Two passes of 32 rows per wave (8 KiB of fp32 per wave each).
    const int fr = lane & 15, fq = lane >> 4;
    float* wl = reinterpret_cast<float*>(lds) + wave * 2048;
    float* wsz = ws ? ws + ((int64_t)z * splitk + split) * (int64_t)p.M * p.N : nullptr;
    if (CS && TA == 1 && p.colsum != nullptr && tile_n == 0) {  // block-uniform condition
        float* red = reinterpret_cast<float*>(lds);
        if (do_cs && fr == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wm * 64 + i * 16 + fq * 4 + r] = acc_cs[i][r];
        }
        __syncthreads();
        if (tid < 128 && m0 + tid < p.M) {
            const float t = red[tid];
            if (splitk > 1) {
                float* cp = ws + (int64_t)nbatch * splitk * p.M * p.N + ((int64_t)z * splitk + split) * p.M + m0 + tid;
                if (SLAB && cnt) __hip_atomic_store(cp, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 store
                else *cp = t;
            } else p.colsum[m0 + tid] += t;
        }
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(wsz, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ii * 16 + fq * 4 + r;
                    wl[row * 64 + ((j * 16 + fr) ^ (fq << 4))] = acc[pass * 2 + ii][j][r];
                }
        // `wl` is private to this wave and a wave's LDS operations complete in program order: a wave-local wait is all
        // the transpose needs.  (A __syncthreads() here also waits vmcnt(0), i.e. for the HBM acknowledgement of every
        // global store of the previous pass -- with all 512 resident workgroups storing in lockstep that exposed the
        // full write burst twice per tile.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int u = lane + 64 * t;
            const int row = u >> 3, c8 = (u & 7) * 8;
            const int m = m0 + wm * 64 + pass * 32 + row;
            const int n = n0 + wn * 64 + c8;
            if (SLAB) {
                if (m < p.M && n < p.N) {
                    float v[8];
                    load8(wl + row * 64 + (c8 ^ (((row >> 2) & 3) << 4)), v);
                    if (cnt) store8_sc1(rws, (m * p.N + n) * 4, v);
                    else store8(wsz + (int64_t)m * p.N + n, v);
                }
            } else if (!GEN) {
                if (m < p.M && n < p.N) {
                    float v[8];
                    load8(wl + row * 64 + (c8 ^ (((row >> 2) & 3) << 4)), v);
                    EpiPre e;
                    e.has_pre = use_pre; e.has_bias = use_bias8; e.pre = pre[pass][t];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e.bias[j] = bias8[j];
                    epilogue_store8(p, FLG, coff, roff, m, n, v, e);
                }
            } else if (m < p.M && n < p.N) {
                float v[8];
                load8(wl + row * 64 + (c8 ^ (((row >> 2) & 3) << 4)), v);
                if (splitk > 1) {
                    float* o = wsz + (int64_t)m * p.N + n;
                    if ((p.N & 3) == 0 && n + 8 <= p.N) store8(o, v);
                    else
                        for (int j = 0; j < 8 && n + j < p.N; ++j) o[j] = v[j];
                } else if (vec == 8) {   // diagnostic: LDS transpose + math, no global store
                    float tsum = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) tsum += v[j];
                    if (tsum == 1.2345678e30f) reinterpret_cast<float*>(p.C)[0] = tsum;
                } else if (vec && n + 8 <= p.N) {
                    {
                        EpiPre e;
                        e.has_pre = use_pre; e.has_bias = use_bias8; e.pre = pre[pass][t];
#pragma unroll
                        for (int j = 0; j < 8; ++j) e.bias[j] = bias8[j];
                        epilogue_store8(p, FLG, coff, roff, m, n, v, e);
                    }
                } else {
                    for (int j = 0; j < 8 && n + j < p.N; ++j) epilogue_store<bf16_t>(p, coff, roff, m, n + j, v[j]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this pass's reads of `wl` are done before it is rewritten
    }
    if constexpr (SLAB) {
        if (cnt) {
            // ---- split-K combined inside the launch: the slice that arrives LAST at the tile's counter sums all slabs
            // in slice order (deterministic whatever the arrival order) and applies the epilogue -- no reduce launch.
            // Publish: every storing wave drains its write-through (sc1) slab stores, the workgroup meets, ONE lane
            // bumps the agent-scope counter.  Consume: the last arriver reads every slab byte with sc1 loads.
            // (cdna guide, Guideline 16 and "Projection GEMM", item 2.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* flag = reinterpret_cast<unsigned*>(lds);
            unsigned* tc = cnt + (int64_t)z * ntiles + tile_lin;
            if (tid == 0) *flag = __hip_atomic_fetch_add(tc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *flag == (unsigned)(splitk - 1);   // block-uniform
            if (last) {
                // (no agent-scope acquire: buffer_inv sc1 drops this XCD's whole L2, i.e. the A / B panels the remaining
                // tiles of the persistent loop still share -- measured +4.4 ms per training step.  Instead EVERY load of a
                // handed-off byte below is an sc1 load, which bypasses the caches that could hold a stale copy.)
                if (tid == 0) __hip_atomic_store(tc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the loads below the poll)
                const int64_t mn = (int64_t)p.M * p.N;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int u = lane + 64 * t;
                        const int row = u >> 3, c8 = (u & 7) * 8;
                        const int m = m0 + wm * 64 + pass * 32 + row;
                        const int n = n0 + wn * 64 + c8;
                        if (m < p.M && n < p.N) {
                            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            for (int sl = 0; sl < splitk; ++sl) {
                                // four 8-byte relaxed agent-scope atomic loads = global_load_dwordx2 ... sc1
                                // (__builtin_amdgcn_raw_buffer_load_b128 with aux = sc1 is mis-lowered to ONE dword
                                // load by ROCm 7.2's hipcc; checked in the ISA)
                                const unsigned long long* sp = reinterpret_cast<const unsigned long long*>(
                                    ws + ((int64_t)z * splitk + sl) * mn + (int64_t)m * p.N + n);
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const unsigned long long q = __hip_atomic_load(sp + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    v[2 * j] += __builtin_bit_cast(float, (unsigned)(q & 0xFFFFFFFFull));
                                    v[2 * j + 1] += __builtin_bit_cast(float, (unsigned)(q >> 32));
                                }
                            }
                            if (vec) {
                                EpiPre e;
                                e.has_pre = false; e.has_bias = false; e.pre = make_uint4(0, 0, 0, 0);
                                epilogue_store8(p, p.flags, coff, roff, m, n, v, e);
                            } else {
                                for (int j = 0; j < 8; ++j) epilogue_store<bf16_t>(p, coff, roff, m, n + j, v[j]);
                            }
                        }
                    }
                if (CS && TA == 1 && p.colsum != nullptr && tile_n == 0 && tid < 128 && m0 + tid < p.M) {
                    const float* wb = ws + (int64_t)nbatch * splitk * mn + (int64_t)z * splitk * p.M + m0 + tid;
                    float t = 0.f;
                    for (int sl = 0; sl < splitk; ++sl)
                        t += __hip_atomic_load(wb + (int64_t)sl * p.M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 load
                    p.colsum[m0 + tid] += t;
                }
            }
            __syncthreads();   // `flag` lives in stage 0
        }
    }
    // every wave is done with its epilogue LDS before the next tile's DMA overwrites stage 0; the global stores stay in
    // flight (raw barrier: no vmcnt wait) and drain under the next tile's first DMA
    __builtin_amdgcn_s_barrier();
    }  // persistent work loop
}

template <int TA, int TB, bool DMA, bool CS, int FL, int RING = 0>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const vpu_gemm_desc p, const int tiles_n, const int splitk,
                                                        const int kchunk, float* __restrict__ ws, const int vec_in,
                                                        const int tiles_m_arg, const int nbatch,
                                                        unsigned* __restrict__ cnt) {
    gemm_bf16_body<TA, TB, DMA, CS, FL, RING, false>(p, nullptr, tiles_n, splitk, kchunk, ws, vec_in, tiles_m_arg, nbatch, cnt);
}
template <int TA, int TB, bool CS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_grouped_kernel(const vpu_gemm_group ga, const int vec_in) {
    gemm_bf16_body<TA, TB, true, CS, -1, 0, true>(ga.d[0], &ga, 0, 1, 0, nullptr, vec_in, 0, 1, nullptr);
}

// ------------------------------------------------------------------------------------------------
// 256x128x64 tile, 8 waves (4x2, each 64x64 -- the same per-wave code as above), THREE 48-KiB LDS stages fed by LDS-DMA:
// tile kt+2 is issued before tile kt is computed, a counted s_waitcnt vmcnt(6) (the 6 pieces of the newest tile may stay
// in flight) + a raw s_barrier close the iteration, so two tiles of HBM/L2 latency are covered by MFMA work and the DMA
// is never drained inside the loop (cdna guide T3/T4).  One block per CU (144 KiB of LDS), two waves per SIMD.
// ------------------------------------------------------------------------------------------------
constexpr int BM2 = 256;
constexpr int STAGE2 = 3 * TILE_BYTES;  // A rows 0-127 | A rows 128-255 | B

template <int TA, int TB>
__global__ __launch_bounds__(512) void gemm_bf16_big_kernel(const vpu_gemm_desc p, const int tiles_n, const int splitk,
                                                            const int kchunk, float* __restrict__ ws, const int vec) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    tile_coords(blockIdx.x, gridDim.x, tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM2, n0 = tile_n * BN;
    const int z = blockIdx.z, zo = z / p.inner, zi = z % p.inner;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + zo * p.sAo + zi * p.sAi;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + zo * p.sBo + zi * p.sBi;
    const int64_t coff = zo * p.sCo + zi * p.sCi;
    const int64_t roff = zo * p.sRo + zi * p.sRi;
    const int kbeg = blockIdx.y * kchunk;
    const int kend = (kbeg + kchunk < p.K) ? kbeg + kchunk : p.K;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, 0x7FFFFFFF, 0x00020000);

    const int FLG = p.flags;
    // (measured: requesting the tile up front costs more than it hides at 2-3 blocks per CU -- disabled)
    const bool use_pre = false && vec == 1 && splitk == 1 &&
                         (p.flags & (VPU_EPI_RESID | VPU_EPI_MULAUX | VPU_EPI_DGELU | VPU_EPI_DRELU)) != 0 &&
                         !((p.flags & VPU_EPI_RESID) && (p.flags & (VPU_EPI_MULAUX | VPU_EPI_DGELU | VPU_EPI_DRELU)));
    uint4 pre[2][4];
    if (use_pre) prefetch_epi(p, FLG, coff, roff, m0 + wm * 64, n0 + wn * 64, lane, pre);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool do_cs = TA == 1 && p.colsum != nullptr && tile_n == 0 && wn == 0;
    f32x4_t acc_cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_cs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)(((lane & 15) == 0) ? 1.0f : 0.0f);

    // 48 pieces per stage; wave w issues pieces w, w+8, ..., w+40 (A: 0-31 in two 128-row sub-tiles, B: 32-47)
    auto issue = [&](int kt_, char* st) {
        const int k0 = kbeg + kt_ * BK;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pc = wave + 8 * i;
            if (i < 4) stage_piece<TA>(rA, p.lda, m0 + (pc >> 4) * 128, p.M, k0, kend, st + (pc >> 4) * TILE_BYTES, pc & 15, lane);
            else stage_piece<TB>(rB, p.ldb, n0, p.N, k0, kend, st + 2 * TILE_BYTES, pc - 32, lane);
        }
    };
    const int nk = (kend - kbeg + BK - 1) / BK;
    char* s0 = lds;
    char* s1 = lds + STAGE2;
    char* s2 = lds + 2 * STAGE2;
    issue(0, s0);
    if (nk > 1) {
        issue(1, s1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 2 < nk;
        if (more) issue(kt + 2, s2);
        const char* ldsA = s0 + (wm >> 1) * TILE_BYTES;
        const char* ldsB = s0 + 2 * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = read_frag<TA>(ldsA, (wm & 1) * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = read_frag<TB>(ldsB, wn * 64 + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (TA == 1 && do_cs) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, acc_cs[i], 0, 0, 0);
            }
        }
        if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        char* t = s0; s0 = s1; s1 = s2; s2 = t;   // rotate stages
    }

    if (vec == 9) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 1.2345678e30f) reinterpret_cast<float*>(p.C)[0] = t;
        return;
    }
    const int fr = lane & 15, fq = lane >> 4;
    float* wl = reinterpret_cast<float*>(lds) + wave * 2048;
    float* wsz = ws ? ws + ((int64_t)z * splitk + blockIdx.y) * (int64_t)p.M * p.N : nullptr;
    if (TA == 1 && p.colsum != nullptr && tile_n == 0) {
        float* red = reinterpret_cast<float*>(lds);
        if (do_cs && fr == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wm * 64 + i * 16 + fq * 4 + r] = acc_cs[i][r];
        }
        __syncthreads();
        if (tid < BM2 && m0 + tid < p.M) {
            const float t = red[tid];
            if (splitk > 1) ws[(int64_t)gridDim.z * splitk * p.M * p.N + ((int64_t)z * splitk + blockIdx.y) * p.M + m0 + tid] = t;
            else p.colsum[m0 + tid] += t;
        }
        __syncthreads();
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ii * 16 + fq * 4 + r;
                    wl[row * 64 + ((j * 16 + fr) ^ (fq << 4))] = acc[pass * 2 + ii][j][r];
                }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int u = lane + 64 * t;
            const int row = u >> 3, c8 = (u & 7) * 8;
            const int m = m0 + wm * 64 + pass * 32 + row;
            const int n = n0 + wn * 64 + c8;
            if (m < p.M && n < p.N) {
                float v[8];
                load8(wl + row * 64 + (c8 ^ (((row >> 2) & 3) << 4)), v);
                if (splitk > 1) {
                    float* o = wsz + (int64_t)m * p.N + n;
                    if ((p.N & 3) == 0 && n + 8 <= p.N) store8(o, v);
                    else
                        for (int j = 0; j < 8 && n + j < p.N; ++j) o[j] = v[j];
                } else if (vec && n + 8 <= p.N) {
                    {
                        EpiPre e;
                        e.has_pre = false; e.has_bias = false; e.pre = make_uint4(0, 0, 0, 0);
                        epilogue_store8(p, FLG, coff, roff, m, n, v, e);
                    }
                } else {
                    for (int j = 0; j < 8 && n + j < p.N; ++j) epilogue_store<bf16_t>(p, coff, roff, m, n + j, v[j]);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K2 (round 2): 256 x (64 WN) x 64 tile per 512-thread workgroup, 128 x 64 outputs per wave (8 x 4 accumulator tiles of
// v_mfma_f32_16x16x32_bf16), one workgroup per CU, persistent tile loop.
//   * twice the output per staged byte of the 128 x 128 tile and 0.375 instead of 0.5 LDS fragment bytes per MFMA -- the
//     two quantities round 1 measured the 128 x 128 kernel to be bound by (L2 -> LDS requests, LDS reads per MFMA).
//   * WN = 2 (256 x 128): the eight waves are 2 (K halves of every 64-deep stage) x 2 (M) x 2 (N); the two K-half groups
//     each accumulate the whole 128 x 64 wave tile over their half of K and exchange half of it through LDS at the end, so
//     every wave finishes a 64 x 64 quarter in the epilogue.  M = 9408 gives 222 / 666 / 888 tiles for N = 768 / 2304 /
//     3072 -- 87 % of the 256 CUs in every round -- where a 256 x 256 tile has 111 / 333 / 444.
//   * WN = 4 (256 x 256): 2 (M) x 4 (N) waves, no K split; for N >= 2048 shapes whose tile count fills the chip.
//   * LDS-DMA ring of S stages (3 x 48 KiB / 2 x 64 KiB), a COUNTED s_waitcnt vmcnt + one raw s_barrier per K-tile; the
//     per-lane DMA source offsets are computed once per tile, the K position rides in the scalar offset operand.
//   * grouped form: the tiles of all problems are cut into 8 contiguous ranges of one global order (short tile dimension
//     fastest inside a problem), one range per XCD, so that an XCD's concurrently running tiles share operand panels
//     through its L2 instead of every XCD streaming every panel (round 1: 700 MB read for 231 MB of operands).
// ------------------------------------------------------------------------------------------------
constexpr int K2_BM = 256;
template <int WN> struct K2Cfg {
    static constexpr int KG = WN == 2 ? 2 : 1;
    static constexpr int NSUB = 2 + WN / 2;   // 16-KiB sub-tiles of a stage: A rows 0-127 | A rows 128-255 | B cols 0-127 [| 128-255]
    static constexpr int STAGE = NSUB * TILE_BYTES;
    static constexpr int S = WN == 2 ? 3 : 2;
    static constexpr int PW = NSUB * 2;       // DMA pieces (1 KiB) per wave per stage
    static constexpr int LDS = S * STAGE;
    static constexpr int BN_ = 64 * WN;
};

template <int TA, int TB, int WN>
__device__ __forceinline__ void k2_issue(const __amdgpu_buffer_rsrc_t rA, const __amdgpu_buffer_rsrc_t rB,
                                         const int (&voff)[K2Cfg<WN>::PW], const int soffA, const int soffB,
                                         const bool live, char* __restrict__ wr, const int wave) {
#pragma unroll
    for (int i = 0; i < K2Cfg<WN>::PW; ++i) {
        const int sub = i >> 1, pis = wave + 8 * (i & 1);
        const int vo = live ? voff[i] : OOB_OFFSET;
        if (sub < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_vptr)(wr + sub * TILE_BYTES + pis * 1024), 16, vo, soffA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_vptr)(wr + sub * TILE_BYTES + pis * 1024), 16, vo, soffB, 0, 0);
    }
}

// One K-tile: DMA of a later K-tile into stage `wr`, fragments + MFMAs of the current one from stage `rd` (the stages are
// __restrict__ parameters of an inlined function for the reason given at ring_step).
template <int TA, int TB, int WN, bool CS, int WAITN, int RB, bool SW = false>
__device__ __forceinline__ void k2_step(const __amdgpu_buffer_rsrc_t rA, const __amdgpu_buffer_rsrc_t rB,
                                        const int (&voff)[K2Cfg<WN>::PW], const int soffA, const int soffB, const bool live,
                                        char* __restrict__ wr, const char* __restrict__ rd, const int wave, const int lane,
                                        const int wm, const int wn, const int g, const bool do_cs, const bf16x8_t ones,
                                        f32x4_t (&acc)[RB][4], f32x4_t (&acc_cs)[4]) {
    k2_issue<TA, TB, WN>(rA, rB, voff, soffA, soffB, live, wr, wave);
    const char* la = rd + wm * TILE_BYTES;
    const char* lb = rd + (2 + (wn >> 1)) * TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < (K2Cfg<WN>::KG == 2 ? 1 : 2); ++kk) {
        const int ks = K2Cfg<WN>::KG == 2 ? g : kk;
        bf16x8_t af[RB], bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = read_frag<TB>(lb, (wn & 1) * 64 + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < RB; ++i) af[i] = read_frag<TA>(la, i * 16, ks, lane);
        if (WAITN >= 0) {
            // ping-pong: the two K-half groups run one barrier apart, so that on every SIMD one wave is in this load
            // section (DMA issue + fragment reads + waits) while its partner is in the MFMA section below.  The load
            // section ends with this wave's pieces of the NEXT K-tile landed and its fragment reads complete (the stage
            // it read may be overwritten by the other group's DMA right after the barrier).
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN >= 0 ? WAITN : 0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // SW: operands swapped -- the accumulator tile is C^T, i.e. acc[i][j][r] = C[16 i + fr][16 j + 4 fq + r]
                if constexpr (SW) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        if constexpr (CS && RB == 8) if (do_cs) {   // the two N-halves of the wave grid share the A fragments: each sums four of the eight row blocks
            if (wn == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, acc_cs[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[4 + i], ones, acc_cs[i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (WAITN >= 0) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// epilogue of one 64 x 64 block held as 4 x 4 accumulator tiles by one wave: transposed through the wave's private LDS
// scratch (`wl`: RP rows x 64 fp32) in 64 / RP passes so that every lane owns 8 consecutive columns of a row (16-byte
// accesses).  Generic form: run-time flags, plain pointer accesses.
template <bool GEN, int RP>
__device__ __forceinline__ void k2_epi64(const vpu_gemm_desc& p, const int FLG, const int vec, f32x4_t (&a)[4][4],
                                         const int mrow0, const int ncol0, float* wl, const int lane,
                                         const int mend = 0x7FFFFFFF,     // rows >= mend belong to another wave / tile
                                         const int64_t coff = 0) {        // element offset of this batch entry's C (K3 grouped slices)
    const bool fast = !GEN || vec == 1;
    const bool use_pre = RP == 32 && fast && (FLG & (VPU_EPI_RESID | VPU_EPI_MULAUX | VPU_EPI_DGELU | VPU_EPI_DRELU)) != 0 &&
                         !((FLG & VPU_EPI_RESID) && (FLG & (VPU_EPI_MULAUX | VPU_EPI_DGELU | VPU_EPI_DRELU)));
    uint4 pre[2][4];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_) pre[a_][b_] = make_uint4(0, 0, 0, 0);
    if (use_pre) prefetch_epi(p, FLG, 0, 0, mrow0, ncol0, lane, pre);
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int ncol = ncol0 + (lane & 7) * 8;
    const bool use_bias8 = fast && (FLG & VPU_EPI_BIAS) && ncol + 8 <= p.N;
    if (use_bias8) load8(p.bias + ncol, bias8);
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int NI = RP / 16;
#pragma unroll
    for (int pass = 0; pass < 64 / RP; ++pass) {
#pragma unroll
        for (int ii = 0; ii < NI; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ii * 16 + fq * 4 + r;
                    wl[row * 64 + ((j * 16 + fr) ^ (fq << 4))] = a[pass * NI + ii][j][r];
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < RP / 8; ++t) {
            const int u = lane + 64 * t;
            const int row = u >> 3, c8 = (u & 7) * 8;
            const int m = mrow0 + pass * RP + row;
            const int n = ncol0 + c8;
            if (m < p.M && m < mend && n < p.N) {
                float v[8];
                load8(wl + row * 64 + (c8 ^ (((row >> 2) & 3) << 4)), v);
                if (fast && n + 8 <= p.N) {
                    EpiPre e;
                    e.has_pre = use_pre; e.has_bias = use_bias8; e.pre = RP == 32 ? pre[pass & 1][t & 3] : make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) e.bias[j] = bias8[j];
                    epilogue_store8(p, FLG, coff, 0, m, n, v, e);
                } else {
                    for (int j = 0; j < 8 && n + j < p.N; ++j) epilogue_store<bf16_t>(p, coff, 0, m, n + j, v[j]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// Exact-count form for the compile-time flag sets (bf16 output): EVERY global access is an unconditional raw-buffer
// instruction -- lanes outside the matrix carry an out-of-range offset (loads return zeros, stores are dropped) -- so the
// number of vector-memory operations a wave issues between the next tile's first LDS-DMA and that tile's main loop is a
// compile-time constant; the counted s_waitcnt vmcnt of that tile's first K-steps is built on it.
template <int FL> struct K2Pre {
    u32x4v x[4][2];   // residual or aux: [pass of 16 rows][group of 8 rows]
    float bias[8];
};
template <int FL>
__device__ __forceinline__ void k2_prefetch(const vpu_gemm_desc& p, const int mrow0, const int ncol0, const int lane,
                                            K2Pre<FL>& q, const int npass) {   // passes >= npass: rows of another wave
    constexpr bool IS_RES = (FL & VPU_EPI_RESID) != 0;
    if constexpr ((FL & (VPU_EPI_RESID | VPU_EPI_MULAUX | VPU_EPI_DRELU)) != 0) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(IS_RES ? p.resid : p.aux), 0, 0x7FFFFFFF, 0x00020000);
        const int ld = IS_RES ? p.ldr : p.ldaux;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int u = lane + 64 * t;
                const int m = mrow0 + pass * 16 + (u >> 3), n = ncol0 + (u & 7) * 8;
                const int row = (IS_RES && p.resid_period > 0) ? m % p.resid_period : m;
                const int off = (pass < npass && m < p.M && n < p.N) ? (row * ld + n) * 2 : OOB_OFFSET;
                q.x[pass][t] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
            }
    }
    if constexpr ((FL & VPU_EPI_BIAS) != 0) {
        // plain loads (a wave-uniform skip of out-of-range columns is fine: only the STORES of this epilogue are counted, the
        // loads are consumed before the first of them).  NOT raw_buffer_load_b128 here: on a float* resource ROCm 7.2's
        // hipcc lowered it to a ONE-dword load (all 8 bias values became the first; seen in the ISA and caught by the
        // exact-integer test at a K2-eligible shape)
        const int n = ncol0 + (lane & 7) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) q.bias[j] = 0.f;
        if (n + 8 <= p.N) load8(p.bias + n, q.bias);
    }
}
template <int FL>
__device__ __forceinline__ void k2_epi_fast(const vpu_gemm_desc& p, f32x4_t (&a)[4][4], const int mrow0, const int ncol0,
                                            float* wl, const int lane, const K2Pre<FL>& q, const int npass) {
    // (all four passes issue their stores -- the count is what the next tile's vmcnt is built on; a pass >= npass stores
    // out of range)
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(p.preact, 0, 0x7FFFFFFF, 0x00020000);
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) wl[(fq * 4 + r) * 64 + ((j * 16 + fr) ^ (fq << 4))] = a[pass][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int u = lane + 64 * t;
            const int row = u >> 3, c8 = (u & 7) * 8;
            const int m = mrow0 + pass * 16 + row, n = ncol0 + c8;
            float v[8];
            load8(wl + row * 64 + (c8 ^ (((row >> 2) & 3) << 4)), v);
            if constexpr ((FL & VPU_EPI_BIAS) != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += q.bias[j];
            }
            const int off = (pass < npass && m < p.M && n < p.N) ? (m * p.ldc + n) * 2 : OOB_OFFSET;
            if constexpr ((FL & VPU_EPI_GELU) != 0) {
                float d[8];
                gelu_dgelu8(v, d);
                if constexpr ((FL & VPU_EPI_SAVE_DGELU) != 0) __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(d), rP, off, 0, 0);
            }
            if constexpr ((FL & VPU_EPI_RELU) != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if constexpr ((FL & (VPU_EPI_RESID | VPU_EPI_MULAUX | VPU_EPI_DRELU)) != 0) {
                const bf16x8_t e = __builtin_bit_cast(bf16x8_t, q.x[pass][t]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = (float)e[j];
                    if (FL & VPU_EPI_MULAUX) v[j] *= x;
                    else if (FL & VPU_EPI_DRELU) v[j] *= (x > 0.f ? 1.f : 0.f);
                    else v[j] += x;
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v), rC, off, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// Direct form (SW kernels, round 4): the MFMAs ran with swapped operands, so a lane holds FOUR CONSECUTIVE COLUMNS of one
// row per accumulator tile: a[i][j][r] = C[16 i + fr][16 j + 4 fq + r].  One v_permlane16_swap per register pairs the
// column blocks 2t / 2t+1 (odd 16-lane rows of the first operand <-> even rows of the second), after which every lane owns
// 8 consecutive columns of its row -- 16-byte stores straight from the registers: no LDS transposition, no lgkmcnt waits,
// and the ring stages are free for the next tile's DMA while the stores go out.  Same store count as k2_epi_fast.
template <int FL> struct K2PreD {
    u32x4v x[4][2];    // residual or aux: [row block][column-block pair]
};
// The bias of a tile's 64 columns per wave travels as ONE 4-byte-per-lane LDS-DMA into a private 256-byte slot of the wave
// (two slots, alternating per tile), requested immediately BEFORE the tile's first stage: any wait that covers that stage
// covers it, so no counted s_waitcnt changes; the epilogue reads it back with two 16-byte LDS reads per 8 columns.  (As
// global loads the 16 values per lane were requested right before the K-half exchange and cost registers there.)
constexpr int K2_BIAS_LDS = 2 * 8 * 256;
__device__ __forceinline__ void k2_bias_issue(const vpu_gemm_desc& p, const int ncol0, const int lane, char* slot) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(p.bias)), 0, p.N * 4, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)slot, 4, (ncol0 + lane) * 4, 0, 0, 0);   // columns >= N: zeros
}
template <int FL>
__device__ __forceinline__ void k2_prefetch_direct(const vpu_gemm_desc& p, const int mrow0, const int ncol0, const int lane,
                                                   K2PreD<FL>& q, const int npass) {
    constexpr bool IS_RES = (FL & VPU_EPI_RESID) != 0;
    const int fr = lane & 15, cl = k2_direct_col(lane);
    if constexpr ((FL & (VPU_EPI_RESID | VPU_EPI_MULAUX | VPU_EPI_DRELU)) != 0) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(IS_RES ? p.resid : p.aux), 0, 0x7FFFFFFF, 0x00020000);
        const int ld = IS_RES ? p.ldr : p.ldaux;
        bool per = false;
        if constexpr (IS_RES) {
            int pv = __builtin_amdgcn_readfirstlane(p.resid_period > 0 ? 1 : 0);
            asm volatile("" : "+s"(pv));          // (opaque scalar: the modulo below stays behind a branch)
            per = pv != 0;
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int m = mrow0 + pass * 16 + fr, n = ncol0 + 32 * t + cl;
                // (the periodic residual -- a broadcast pos_embed -- behind a UNIFORM branch: as a select the compiler computed
                // the integer modulo for every load of every tile, ~300 instructions per wave: 1.2 us of the 4.5-us handover of
                // the ViT blocks' residual forms, which never use it; tools/k2_stamps.py)
                int row = m;
                if (IS_RES && per) row = m % p.resid_period;
                const int off = (pass < npass && m < p.M && n < p.N) ? (row * ld + n) * 2 : OOB_OFFSET;
                q.x[pass][t] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
            }
    }
}
template <int FL>
__device__ __forceinline__ void k2_epi_direct(const vpu_gemm_desc& p, f32x4_t (&a)[4][4], const int mrow0, const int ncol0,
                                              const int lane, const K2PreD<FL>& q, const int npass, const float* bl) {
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(p.preact, 0, 0x7FFFFFFF, 0x00020000);
    const int fr = lane & 15, cl = k2_direct_col(lane);
    // the wave's bias values out of its LDS slot, through inline asm: in front of a C++ LDS read of a slot that an LDS-DMA
    // wrote, hipcc puts s_waitcnt vmcnt(0) -- which drained the next tile's primed stages (in flight here by design) before
    // this tile's first store; the DMA that filled the slot is older than this tile's K-tile 0, which has been waited for
    f32x4_t bq[2][2];
    if constexpr ((FL & VPU_EPI_BIAS) != 0) {
        const unsigned ba = (unsigned)reinterpret_cast<uintptr_t>(bl + cl);
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:144\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(bq[0][0]), "=&v"(bq[0][1]), "=&v"(bq[1][0]), "=&v"(bq[1][1]) : "v"(ba) : "memory");
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (temporaries: __builtin_bit_cast applied to a vector-element lvalue read element 0 for every r)
                const float x0 = a[pass][2 * t][r], x1 = a[pass][2 * t + 1][r];
                const u32x2v sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
                v[r] = __uint_as_float(sw.x);
                v[4 + r] = __uint_as_float(sw.y);
            }
            const int m = mrow0 + pass * 16 + fr, n = ncol0 + 32 * t + cl;
            if constexpr ((FL & VPU_EPI_BIAS) != 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] += bq[t][0][j]; v[4 + j] += bq[t][1][j]; }
            }
            const int off = (pass < npass && m < p.M && n < p.N) ? (m * p.ldc + n) * 2 : OOB_OFFSET;
            if constexpr ((FL & VPU_EPI_GELU) != 0) {
                float d[8];
                gelu_dgelu8(v, d);
                if constexpr ((FL & VPU_EPI_SAVE_DGELU) != 0) __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(d), rP, off, 0, 0);
            }
            if constexpr ((FL & VPU_EPI_RELU) != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if constexpr ((FL & (VPU_EPI_RESID | VPU_EPI_MULAUX | VPU_EPI_DRELU)) != 0) {
                const bf16x8_t e = __builtin_bit_cast(bf16x8_t, q.x[pass][t]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = (float)e[j];
                    if (FL & VPU_EPI_MULAUX) v[j] *= x;
                    else if (FL & VPU_EPI_DRELU) v[j] *= (x > 0.f ? 1.f : 0.f);
                    else v[j] += x;
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v), rC, off, 0, 0);
        }
}

struct K2Tile {   // wave-uniform description of one output tile
    int m0, n0, tile_n, grp, M, N, K, lda, ldb;
    int z;            // batch entry (K3 grouped form: the reduction slices of one weight gradient), else 0
    int64_t coff;     // element offset of that entry's C
    const void* A;
    const void* B;
};
template <int WN, bool GRP, int RB, bool BATCH = false>
__device__ __forceinline__ void k2_tile_setup(const int work, const int total_work, const vpu_gemm_desc& p_arg,
                                              const vpu_gemm_group* __restrict__ ga, const int tiles_n_arg, K2Tile& t) {
    static_assert(!GRP || RB == 8, "grouped form: 256-row tiles");
    int grp = 0, tile_m, tile_n, z = 0;
    if constexpr (GRP) {
        // one contiguous range of the global tile order per XCD label (work & 7); inside a problem the shorter tile
        // dimension runs fastest, so a range is a compact block of the problem's tile grid
        const int xcd = work & 7, q = total_work >> 3, r = total_work & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (work >> 3);
        while (grp + 1 < ga->n && v >= ga->start[grp + 1]) ++grp;
        grp = rfl(grp);
        int local = v - ga->start[grp];
        const int tm = (ga->d[grp].M + K2_BM - 1) / K2_BM, tn = (ga->d[grp].N + K2Cfg<WN>::BN_ - 1) / K2Cfg<WN>::BN_;
        if constexpr (BATCH) {   // batch entries (reduction slices) of one problem: entry-major, so that an XCD's range stays within few slices
            z = local / (tm * tn);
            local -= z * (tm * tn);
        }
        if (tn <= tm) { tile_m = local / tn; tile_n = local - tile_m * tn; }
        else { tile_n = local / tm; tile_m = local - tile_n * tm; }
    } else {
        tile_coords(work, total_work, tiles_n_arg, tile_m, tile_n);
    }
    const vpu_gemm_desc& p = GRP ? ga->d[grp] : p_arg;
    t.grp = grp; t.tile_n = rfl(tile_n);
    t.m0 = rfl(tile_m * (32 * RB)); t.n0 = rfl(tile_n * K2Cfg<WN>::BN_);
    // (grouped form: the descriptor is picked by a run-time index; pin what the main loop uses to scalar registers)
    t.M = rfl(p.M); t.N = rfl(p.N); t.K = rfl(p.K); t.lda = rfl(p.lda); t.ldb = rfl(p.ldb);
    t.z = rfl(z);
    if constexpr (BATCH) {
        t.coff = (int64_t)t.z * p.sCo;
        t.A = rfl_ptr(reinterpret_cast<const bf16_t*>(p.A) + (int64_t)t.z * p.sAo);
        t.B = rfl_ptr(reinterpret_cast<const bf16_t*>(p.B) + (int64_t)t.z * p.sBo);
    } else {
        t.coff = 0;
        t.A = rfl_ptr(p.A); t.B = rfl_ptr(p.B);
    }
}
// per-lane byte offset of every DMA piece this wave issues per stage, at k = 0 (K % 64 == 0: no k bound inside a tile)
template <int TA, int TB, int WN, int RB>
__device__ __forceinline__ void k2_voff(const K2Tile& t, const int wave, const int lane, int (&voff)[K2Cfg<WN>::PW]) {
    static_assert(RB == 8 || TA == 0, "short tiles: row-major A only");
#pragma unroll
    for (int i = 0; i < K2Cfg<WN>::PW; ++i) {
        const int sub = i >> 1, pis = wave + 8 * (i & 1);
        const bool isA = sub < 2;
        const int tr = isA ? TA : TB;
        const int x0 = isA ? t.m0 + sub * (16 * RB) : t.n0 + (sub - 2) * 128;   // (A sub-tile = the rows of one wm)
        const int X = isA ? t.M : t.N;
        const int ld = isA ? t.lda : t.ldb;
        if (tr == 0) {
            const int row = pis * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const int gx = x0 + row;
            voff[i] = (gx < X && (!isA || row < 16 * RB)) ? (gx * ld + chunk * 8) * 2 : OOB_OFFSET;
        } else {
            const int k = pis * 4 + (lane >> 4);
            const int chunk = (lane & 15) ^ ((k & 3) << 1) ^ (((k >> 3) & 1) << 3);
            const int gx = x0 + chunk * 8;
            voff[i] = gx < X ? (k * ld + gx) * 2 : OOB_OFFSET;
        }
    }
}

// RB = 16-row blocks per wave: 8 (256-row tiles) or 7 (224-row tiles: M = 9408 = 42 x 224 fills 252 of 256 CUs per round
// where 36.75 x 256 fills 222; the LDS layout keeps its 128-row sub-tiles, rows 112-127 of each are dead).
template <int TA, int TB, int WN, bool CS, int FL, bool GRP, int RB, int SWP = 0>
__device__ __forceinline__ void k2_body(const vpu_gemm_desc& p_arg, const vpu_gemm_group* __restrict__ ga,
                                        const int tiles_m_arg, const int tiles_n_arg, const int vec
                                        VPU_DBG_PARAM_DEF) {   // (-DVPU_DIAG, vpu_debug_gemm_times: 16 stamps per workgroup)
    using Cf = K2Cfg<WN>;
    // SW: swapped MFMA operands + direct epilogue (k2_epi_direct); compile-time flag sets of the plain kernel only
    constexpr bool SW = SWP >= 1 && FL >= 0 && !CS;      // (also the grouped form with ONE compile-time flag set for all its problems)
    static_assert(!CS || (WN == 2 && TA == 1 && RB == 8), "fused column sums: weight-gradient form, 256 x 128 tile");
    constexpr bool PP = WN == 2;       // ping-pong schedule of the two K-half groups + next tile's first stages requested early
    constexpr bool GEN = FL < 0;
    // vector-memory stores one wave issues in the exact-count epilogue of its 64 x 64 quarter
    constexpr int NST = GEN ? 0 : 8 * ((FL & VPU_EPI_SAVE_DGELU) ? 2 : 1);
    constexpr int W1 = GEN ? Cf::PW : Cf::PW + NST;   // first waits of a tile whose stages 0 / 1 were requested before those stores
    constexpr int NSTW = 2 * NST;                     // WN = 4, SW: both 64-row halves' stores follow the next tile's first DMA
    int par = 0;                                      // WN = 4: ring stage that receives this tile's K-tile 0
    constexpr bool BL = SW && (FL & VPU_EPI_BIAS) != 0;  // bias through the wave's LDS slot (k2_bias_issue)
    // (measured and not kept, round 4: a cross-tile ring for the 256 x 128 form -- the last K-step's DMA slot requesting the
    // next tile's K-tile 0 and bias, the K-half exchange in two 8-KiB rounds through the two free stages, the ring position
    // carried from tile to tile: bit-exact, and no faster -- qkv 44.1 vs 44.6 us, fc2 49.8 vs 50.8, step 12.70 vs 12.69 ms --
    // the stage-0 latency is not what a tile boundary costs)
    int bs = 0;                                       // bias slot of the current tile
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = Cf::KG == 2 ? wave >> 2 : 0;
    const int wm = Cf::KG == 2 ? (wave >> 1) & 1 : wave >> 2;
    const int wn = Cf::KG == 2 ? wave & 1 : wave & 3;
    const int total_work = GRP ? ga->start[ga->n] : tiles_m_arg * tiles_n_arg;
    int work = blockIdx.x;
    K2Tile cur;
    int voff[Cf::PW];
    bool primed = false;
    if (work < total_work) {
        k2_tile_setup<WN, GRP, RB>(work, total_work, p_arg, ga, tiles_n_arg, cur);
        k2_voff<TA, TB, WN, RB>(cur, wave, lane, voff);
    }
    int dbg_t = 0;       // tile number of this workgroup (diagnostic stamps: tile start / main loop end / epilogue end)
    while (work < total_work) {
        VPU_STAMP(tid == 0 && dbg_t < 5, blockIdx.x * 16 + 3 * dbg_t);
        const vpu_gemm_desc& p = GRP ? ga->d[cur.grp] : p_arg;
        const int FLG = GEN ? p.flags : FL;
        const int m0 = cur.m0, n0 = cur.n0;
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(cur.A), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(cur.B), 0, 0x7FFFFFFF, 0x00020000);
        const int stepA = TA ? cur.lda * (BK * 2) : BK * 2, stepB = TB ? cur.ldb * (BK * 2) : BK * 2;
        const int nk = cur.K / BK;

        f32x4_t acc[RB][4];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const bool do_cs = CS && p.colsum != nullptr && cur.tile_n == 0;   // block-uniform
        f32x4_t acc_cs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc_cs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        bf16x8_t ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)(((lane & 15) == 0) ? 1.0f : 0.0f);

        char* s0 = PP ? lds : lds + par * Cf::STAGE;
        char* s1 = lds + Cf::STAGE;
        char* s2 = PP ? lds + (Cf::S - 1) * Cf::STAGE : lds + (1 - par) * Cf::STAGE;
        if constexpr (!PP) par ^= nk & 1;
        const int nxt = work + gridDim.x;
        const bool has_next = nxt < total_work;
        K2Tile nt = cur;
        int nvoff[Cf::PW];
#pragma unroll
        for (int i = 0; i < Cf::PW; ++i) nvoff[i] = voff[i];
        char* const bslot = lds + Cf::LDS + (bs * 8 + wave) * 256;
        char* const nbslot = lds + Cf::LDS + ((bs ^ 1) * 8 + wave) * 256;
        bs ^= 1;
        if (!primed) {
            if constexpr (BL) k2_bias_issue(p, n0 + wn * 64, lane, bslot);
            k2_issue<TA, TB, WN>(rA, rB, voff, 0, 0, true, s0, wave);
            if (Cf::S == 3) k2_issue<TA, TB, WN>(rA, rB, voff, stepA, stepB, 1 < nk, s1, wave);
        }
        if constexpr (PP) {
            // K-tile 0 has landed for every wave (a primed tile: its stages 0 / 1 were requested before the previous tile's
            // NST epilogue stores, which may still be in flight); then group 1 drops one barrier behind group 0
            if (primed) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cf::PW) : "memory");
            __builtin_amdgcn_s_barrier();
            if (g == 1) __builtin_amdgcn_s_barrier();
            if (primed) k2_step<TA, TB, WN, CS, W1, RB, SW>(rA, rB, voff, 2 * stepA, 2 * stepB, 2 < nk, s2, s0, wave, lane, wm, wn, g, do_cs, ones, acc, acc_cs);
            else k2_step<TA, TB, WN, CS, Cf::PW, RB, SW>(rA, rB, voff, 2 * stepA, 2 * stepB, 2 < nk, s2, s0, wave, lane, wm, wn, g, do_cs, ones, acc, acc_cs);
            { char* t = s0; s0 = s1; s1 = s2; s2 = t; }
            for (int kt = 1; kt < nk; ++kt) {
                const int kn = kt + 2;
                k2_step<TA, TB, WN, CS, Cf::PW, RB, SW>(rA, rB, voff, kn * stepA, kn * stepB, kn < nk, s2, s0, wave, lane, wm, wn, g, do_cs, ones, acc, acc_cs);
                char* t = s0; s0 = s1; s1 = s2; s2 = t;
            }
            if (g == 0) __builtin_amdgcn_s_barrier();   // the two groups meet again
        } else {
            for (int kt = 0; kt < nk; ++kt) {
                // (a primed tile: K-tile 0 was requested before the previous tile's NSTW direct stores, which may still be in flight)
                if (SW && primed && kt == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // K-tile kt has landed for every wave; every wave is done reading K-tile kt-1
                const int kn = kt + 1;
                k2_step<TA, TB, WN, CS, -1, RB, SW>(rA, rB, voff, kn * stepA, kn * stepB, kn < nk, s2, s0, wave, lane, wm, wn, g, do_cs, ones, acc, acc_cs);
                char* t = s0; s0 = s2; s2 = t;
            }
            // (measured and not kept: requesting the residual / aux operand of the direct 256-column form one K-step before the
            // loop ends -- 32 or 64 more live registers in that step spill, and a scratch reload drains the vmcnt queue)
        }
        // (the out-of-range tail pieces still write zeros into LDS; the direct 256-column form does not touch LDS in its
        // epilogue and the same wave overwrites the same words with the next tile's pieces, in order: nothing to wait for)
        if constexpr (!(SW && !PP)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VPU_STAMP(tid == 0 && dbg_t < 5, blockIdx.x * 16 + 3 * dbg_t + 1);

        if (vec == 9) {  // diagnostic (VPU_GEMM_NOEPI=1): main loop only; the impossible compare keeps the accumulators live
            __syncthreads();
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (t == 1.2345678e30f) reinterpret_cast<float*>(p.C)[0] = t;
        } else if constexpr (Cf::KG == 2) {
            // this wave's quarter: 64 x 64 (group 0: row blocks 0-3 of the wave tile; group 1: blocks 4 .. RB-1, 48 rows when RB = 7)
            const int mq = m0 + wm * (16 * RB) + g * 64, nq = n0 + wn * 64;
            const int npass = g == 0 ? 4 : RB - 4;
            K2Pre<(GEN || SW) ? 0 : FL> q;
            K2PreD<SW ? FL : 0> qd;
            // (an opaque copy of the lane index for everything between the main loops of the direct form: what is derived
            // from the lane is then recomputed per tile instead of being hoisted out of the tile loop, kept across the main
            // loop and spilled -- a scratch reload sits in the vector-memory queue and its wait drains the operand loads and
            // the primed stages)
            int el = lane;
            if constexpr (SW) asm volatile("" : "+v"(el));
            if constexpr (SW) k2_prefetch_direct<FL>(p, mq, nq, el, qd, npass);
            else if constexpr (!GEN) k2_prefetch<FL>(p, mq, nq, lane, q, npass);   // lands while the halves are exchanged
            // the next tile's coordinates and DMA offsets now (integer divisions, ~100 instructions): they overlap the waits of
            // the exchange instead of standing between its last barrier and the DMA that primes the next tile
            if (has_next) {
                k2_tile_setup<WN, GRP, RB>(nxt, total_work, p_arg, ga, tiles_n_arg, nt);
                k2_voff<TA, TB, WN, RB>(nt, wave, el, nvoff);
            }
            // (raw barriers + explicit LDS waits in this exchange: __syncthreads() carries a fence, i.e. s_waitcnt vmcnt(0),
            // which exposed the whole latency of the bias / residual / aux loads just requested -- once per tile)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();   // every wave is done with the ring (fragment reads consumed, own DMA pieces landed)
            __builtin_amdgcn_sched_barrier(0);
            VPU_STAMP(tid == 0 && dbg_t == 0, blockIdx.x * 16 + 10);
            // the two K-half groups exchange half of their 128 x 64 partial tile: group 0 finishes rows 0-63, group 1 rows
            // 64-127 of it.  Fragment layouts are identical in both waves, so the registers travel as they are (16-byte LDS
            // accesses, lane-linear).  Wave w sends through [w * 16 KiB, + 16 KiB).
            const int fr = lane & 15, fq = lane >> 4;
            f32x4_t* mine = reinterpret_cast<f32x4_t*>(lds + wave * 16384) + el;
            const f32x4_t* theirs = reinterpret_cast<const f32x4_t*>(lds + (wave ^ 4) * 16384) + el;
            float* red = reinterpret_cast<float*>(lds + 8 * 16384);   // [2][256] fused column sums
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mine[(i * 4 + j) * 64] = g == 0 ? acc[4 + i < RB ? 4 + i : 0][j] : acc[i][j];
            if (CS && do_cs && fr == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[g * 256 + wm * 128 + (wn * 4 + i) * 16 + fq * 4 + r] = acc_cs[i][r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (CS && do_cs && tid < K2_BM && m0 + tid < cur.M) p.colsum[m0 + tid] += red[tid] + red[256 + tid];
            f32x4_t fin[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) fin[i][j] = (g == 0 ? acc[i][j] : acc[4 + i < RB ? 4 + i : 0][j]) + theirs[(i * 4 + j) * 64];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();   // every wave has taken its partner's half: stages 0 and 1 can receive the next tile
            __builtin_amdgcn_sched_barrier(0);
            VPU_STAMP(tid == 0 && dbg_t == 0, blockIdx.x * 16 + 11);
            if (has_next) {
                const __amdgpu_buffer_rsrc_t nA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(nt.A), 0, 0x7FFFFFFF, 0x00020000);
                const __amdgpu_buffer_rsrc_t nB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(nt.B), 0, 0x7FFFFFFF, 0x00020000);
                const int nsA = TA ? nt.lda * (BK * 2) : BK * 2, nsB = TB ? nt.ldb * (BK * 2) : BK * 2;
                if constexpr (BL) k2_bias_issue(GRP ? ga->d[nt.grp] : p, nt.n0 + wn * 64, el, nbslot);
                k2_issue<TA, TB, WN>(nA, nB, nvoff, 0, 0, true, lds, wave);
                k2_issue<TA, TB, WN>(nA, nB, nvoff, nsA, nsB, BK < nt.K, lds + Cf::STAGE, wave);
            }
            float* wl = reinterpret_cast<float*>(lds + 2 * Cf::STAGE + wave * 4096);   // 16 rows x 64 fp32, in stage 2
            VPU_STAMP(tid == 0 && dbg_t == 0, blockIdx.x * 16 + 12);
            if constexpr (SW) k2_epi_direct<FL>(p, fin, mq, nq, el, qd, npass, reinterpret_cast<const float*>(bslot));
            VPU_STAMP(tid == 0 && dbg_t == 0, blockIdx.x * 16 + 13);
            if constexpr (SW) {}
            else if constexpr (GEN) k2_epi64<true, 16>(p, FLG, vec, fin, mq, nq, wl, lane, mq + 16 * npass);
            else k2_epi_fast<FL>(p, fin, mq, nq, wl, lane, q, npass);
        } else if constexpr (SW) {
            // WN = 4, direct: nothing of the epilogue touches LDS, so the next tile's K-tile 0 is requested first -- into the
            // stage the last K-step did not read (this wave's own dead tail pieces into it were requested earlier and land
            // first: the vector-memory queue is in order) -- and the stores of this tile go out behind it
            const int mw = m0 + wm * (16 * RB), nw = n0 + wn * 64;
            K2PreD<FL> q0, q1;
            k2_prefetch_direct<FL>(p, mw, nw, lane, q0, 4);
            k2_prefetch_direct<FL>(p, mw + 64, nw, lane, q1, RB - 4);
            VPU_STAMP(tid == 0 && dbg_t == 0, blockIdx.x * 16 + 10);
            if (has_next) {
                k2_tile_setup<WN, GRP, RB>(nxt, total_work, p_arg, ga, tiles_n_arg, nt);
                k2_voff<TA, TB, WN, RB>(nt, wave, lane, nvoff);
                const __amdgpu_buffer_rsrc_t nA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(nt.A), 0, 0x7FFFFFFF, 0x00020000);
                const __amdgpu_buffer_rsrc_t nB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(nt.B), 0, 0x7FFFFFFF, 0x00020000);
                if constexpr (BL) k2_bias_issue(GRP ? ga->d[nt.grp] : p, nt.n0 + wn * 64, lane, nbslot);
                k2_issue<TA, TB, WN>(nA, nB, nvoff, 0, 0, true, lds + par * Cf::STAGE, wave);
            }
            {
                f32x4_t fin[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) fin[i][j] = acc[i][j];
                VPU_STAMP(tid == 0 && dbg_t == 0, blockIdx.x * 16 + 11);
                k2_epi_direct<FL>(p, fin, mw, nw, lane, q0, 4, reinterpret_cast<const float*>(bslot));
                VPU_STAMP(tid == 0 && dbg_t == 0, blockIdx.x * 16 + 12);
            }
            {
                f32x4_t fin[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) fin[i][j] = acc[4 + i < RB ? 4 + i : 0][j];
                k2_epi_direct<FL>(p, fin, mw + 64, nw, lane, q1, RB - 4, reinterpret_cast<const float*>(bslot));
                VPU_STAMP(tid == 0 && dbg_t == 0, blockIdx.x * 16 + 13);
            }
        } else {
            __syncthreads();
            float* wl = reinterpret_cast<float*>(lds) + wave * 2048;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4_t fin[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) fin[i][j] = acc[h * 4 + i < RB ? h * 4 + i : 0][j];
                k2_epi64<GEN, 32>(p, FLG, vec, fin, m0 + wm * (16 * RB) + h * 64, n0 + wn * 64, wl, lane, m0 + (wm + 1) * (16 * RB));
            }
        }
        // every wave is done with its epilogue LDS before the next tile's DMA / fragment reads touch it (raw barrier: the
        // global stores stay in flight)
        __builtin_amdgcn_s_barrier();
        VPU_STAMP(tid == 0 && dbg_t < 5, blockIdx.x * 16 + 3 * dbg_t + 2);
        ++dbg_t;
        primed = (PP || SW) && has_next && vec != 9;
        if (has_next && !primed) {
            k2_tile_setup<WN, GRP, RB>(nxt, total_work, p_arg, ga, tiles_n_arg, nt);
            k2_voff<TA, TB, WN, RB>(nt, wave, lane, nvoff);
        }
        cur = nt;
#pragma unroll
        for (int i = 0; i < Cf::PW; ++i) voff[i] = nvoff[i];
        work = nxt;
    }
}

template <int TA, int TB, int WN, int FL, int RB, int SWP = 0>
__global__ __launch_bounds__(512) void gemm_bf16_k2_kernel(const vpu_gemm_desc p, const int tiles_m, const int tiles_n,
                                                           const int vec VPU_DBG_PARAM) {
    k2_body<TA, TB, WN, false, FL, false, RB, SWP>(p, nullptr, tiles_m, tiles_n, vec VPU_DBG_PASS);
}
// forward / dgrad groups whose problems share ONE compile-time flag set (round 4: the DMA neck's image-side K / V projections,
// bias only): the direct epilogue of the plain kernel instead of the run-time one
template <int TA, int TB, int FL>
__global__ __launch_bounds__(512) void gemm_bf16_k2_grouped_fl_kernel(const vpu_gemm_group ga_unused, const int vec) {
    const vpu_gemm_group* ga = (const vpu_gemm_group*)__builtin_amdgcn_kernarg_segment_ptr();
    k2_body<TA, TB, 2, false, FL, true, 8, 1>(ga->d[0], ga, 0, 0, vec);
}
template <int TA, int TB, bool CS>
__global__ __launch_bounds__(512) void gemm_bf16_k2_grouped_kernel(const vpu_gemm_group ga_unused, const int vec) {
    // the descriptors are read where they already are, in the kernel-argument segment (scalar loads with a run-time
    // index); taking the address of the by-value parameter makes hipcc copy all 3.5 KB of it to scratch first
    const vpu_gemm_group* ga = (const vpu_gemm_group*)__builtin_amdgcn_kernarg_segment_ptr();
    k2_body<TA, TB, 2, CS, -1, true, 8>(ga->d[0], ga, 0, 0, vec);
}

// ------------------------------------------------------------------------------------------------
// K3 (round 4): the K2 wave tile (128 x 64 outputs per wave, 8 x 4 accumulator tiles) in a 256-thread workgroup of
// 2 (M) x 2 (N) waves -- a 256 x 128 output tile with NO K split -- and TWO such workgroups per CU.
//   * K2's eight waves are coupled by the workgroup barrier of every K-step: its two K-half groups alternate load and MFMA
//     sections in lock step ("ping-pong"), so a barrier interval lasts as long as the LONGER of the two sections and the
//     MFMA pipe is busy Mf / max(L, Mf) of the time (measured 0.45: the load section -- six LDS-DMA issues, twelve
//     fragment reads, the counted wait -- takes ~1.7-2.4 x the 512 MFMA cycles it is paired with).  Two independent
//     workgroups are not coupled: each SIMD holds one wave of either, the hardware issues whichever is ready, load sections
//     overlap each other as well as the other wave's MFMAs (2 Mf / (L + Mf) at best), and one workgroup's prologue / epilogue
//     runs beside the other's main loop -- no priming of the next tile, no exact-count epilogue, no K-half exchange
//     (128 KiB through LDS per tile in K2).
//   * two workgroups must share the 160 KiB of LDS: a K-step is 32 deep (one MFMA k), a stage is 24 KiB (A rows 0-127 |
//     A rows 128-255 | B), three stages in a ring with a counted s_waitcnt vmcnt + ONE raw s_barrier per K-step
//     (K2: two).  K-major operands keep K2's image ([k][128 columns], 256-B rows, 4 k-rows per 1-KiB DMA piece);
//     K-contiguous ones are [128 rows][32 k] with 64-B rows, 16 rows per piece, 16-byte chunk index XOR ((row >> 3) & 1) << 1
//     (conflict-free for ds_read_b128's lane groups).
// ------------------------------------------------------------------------------------------------
constexpr int K3_BK = 32;
constexpr int K3_SUB = 8192;               // one 128 x 32 / 32 x 128 bf16 sub-tile
// NWN = 2: "K3", 2 x 2 waves, 256 x 128 tile, three 24-KiB stages (72 KiB: two workgroups per CU).
// NWN = 4: "K4", 2 x 4 waves, 256 x 256 tile, FIVE 32-KiB stages (all 160 KiB of one CU, one workgroup): the long-reduction
// weight gradients are bound by bytes in flight over the latency of an HBM miss (operands that no cache holds: 2 x 24 KiB
// per workgroup -> ~38 GB/s per CU whatever the schedule, K2 and K3 alike: ~900 TFLOP/s in the step); the wider tile needs
// 2/3 of the bytes per MFMA and the deeper ring keeps 4 x 32 KiB = 128 KiB in flight.
// TM = 128 ("K3S", NWN = 2): a 128 x 128 tile, 64 x 64 per wave, ONE A sub-tile; 16-KiB stages, three of them: 48 KiB, three
// workgroups per CU -- the round-1 128 x 128 kernel's shapes (one round of tiles over a short K, the neck / FPN / head maps)
// with a ring and a counted wait instead of one K-tile in flight and a drained vmcnt(0) per K-tile.
template <int NWN, int TM = 256> struct K3Cfg {
    static constexpr int NSUBA = TM / 128;
    static constexpr int NSUB = NSUBA + NWN / 2;
    static constexpr int STAGE = NSUB * K3_SUB;
    static constexpr int S = NWN == 2 ? 3 : 5;
    static constexpr int LDS = S * STAGE;
    static constexpr int NWAVE = 2 * NWN;
    static constexpr int PW = NSUB * 8 / NWAVE;     // DMA pieces per wave per stage: 6 / 4
    static constexpr int INFL = (S - 2) * PW;       // pieces that may stay in flight at the top of a K-step: 6 / 12
    static constexpr int NCS = 8 / NWN;             // row blocks whose column sums one wave takes
};
constexpr int K3_LDS = K3Cfg<2>::LDS;
constexpr int K3_PW = K3Cfg<2>::PW;

__device__ __forceinline__ int kc32_off(int row, int chunk) { return row * 64 + ((chunk ^ (((row >> 3) & 1) << 1)) << 4); }

template <int TRANS>
__device__ __forceinline__ bf16x8_t k3_frag(const char* lds, int x16, int lane) {
    if (TRANS == 0) return *reinterpret_cast<const bf16x8_t*>(lds + kc32_off(x16 + (lane & 15), lane >> 4));
    else return read_frag<1>(lds, x16, 0, lane);
}

template <int TA, int TB, int RB, int NWN, int TM = 256>
__device__ __forceinline__ void k3_voff(const K2Tile& t, const int wave, const int lane, int (&voff)[K3Cfg<NWN, TM>::PW]) {
    using Cf = K3Cfg<NWN, TM>;
    static_assert(RB == 8 || TA == 0 || TM == 128, "short tiles: row-major A only");
    constexpr int SUBROWS = TM == 256 ? 16 * RB : 128;   // live rows of an A sub-tile (TM = 256: the rows of one wave row)
#pragma unroll
    for (int i = 0; i < Cf::PW; ++i) {
        // (sub-tile and piece from compile-time arithmetic: `wave` is a run-time value, and a sub-tile index that depends
        // on it makes every DMA a run-time choice between the A and the B resource)
        constexpr int PPS = 8 / Cf::NWAVE;      // pieces per sub-tile and wave
        const int sub = i / PPS, pis = wave + Cf::NWAVE * (i % PPS);
        const bool isA = sub < Cf::NSUBA;
        const int tr = isA ? TA : TB;
        const int x0 = isA ? t.m0 + sub * SUBROWS : t.n0 + (sub - Cf::NSUBA) * 128;
        const int X = isA ? t.M : t.N;
        const int ld = isA ? t.lda : t.ldb;
        if (tr == 0) {
            const int row = pis * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ (((row >> 3) & 1) << 1);
            const int gx = x0 + row;
            voff[i] = (gx < X && (!isA || row < SUBROWS)) ? (gx * ld + chunk * 8) * 2 : OOB_OFFSET;
        } else {
            const int k = pis * 4 + (lane >> 4);
            const int chunk = (lane & 15) ^ ((k & 3) << 1) ^ (((k >> 3) & 1) << 3);
            const int gx = x0 + chunk * 8;
            voff[i] = gx < X ? (k * ld + gx) * 2 : OOB_OFFSET;
        }
    }
}

template <int NWN, int TM = 256>
__device__ __forceinline__ void k3_issue(const __amdgpu_buffer_rsrc_t rA, const __amdgpu_buffer_rsrc_t rB,
                                         const int (&voff)[K3Cfg<NWN, TM>::PW], const int soffA, const int soffB, const bool live,
                                         char* __restrict__ wr, const int wave) {
#pragma unroll
    for (int i = 0; i < K3Cfg<NWN, TM>::PW; ++i) {
        constexpr int PPS = 8 / K3Cfg<NWN, TM>::NWAVE;
        const int sub = i / PPS, pis = wave + K3Cfg<NWN, TM>::NWAVE * (i % PPS);
        const int vo = live ? voff[i] : OOB_OFFSET;
        if (sub < K3Cfg<NWN, TM>::NSUBA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_vptr)(wr + sub * K3_SUB + pis * 1024), 16, vo, soffA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_vptr)(wr + sub * K3_SUB + pis * 1024), 16, vo, soffB, 0, 0);
    }
}

// one K-step: the DMA of K-step kt+S-1 into stage `wr`, the fragments and the 32 MFMAs of K-step kt from stage `rd`
// (__restrict__ parameters of an inlined function: see ring_step)
template <int TA, int TB, bool CS, int RB, int NWN, int TM = 256>
__device__ __forceinline__ void k3_step(const __amdgpu_buffer_rsrc_t rA, const __amdgpu_buffer_rsrc_t rB,
                                        const int (&voff)[K3Cfg<NWN, TM>::PW], const int soffA, const int soffB, const bool live,
                                        char* __restrict__ wr, const char* __restrict__ rd, const int wave, const int lane,
                                        const int wm, const int wn, const bool do_cs, const bf16x8_t ones, f32x4_t (&acc)[RB][4],
                                        f32x4_t (&acc_cs)[K3Cfg<NWN>::NCS]) {
    k3_issue<NWN, TM>(rA, rB, voff, soffA, soffB, live, wr, wave);
    const char* la = rd + (TM == 256 ? wm : 0) * K3_SUB;           // TM = 128: one A sub-tile, this wave's rows at wm * 64
    const int arow = TM == 256 ? 0 : wm * 64;
    const char* lb = rd + (K3Cfg<NWN, TM>::NSUBA + (wn >> 1)) * K3_SUB;
    bf16x8_t af[RB], bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = k3_frag<TB>(lb, (wn & 1) * 64 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < RB; ++i) af[i] = k3_frag<TA>(la, arow + i * 16, lane);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if constexpr (CS && RB == 8) if (do_cs) {   // the N positions of the wave grid share the A fragments: each sums NCS of the eight row blocks
#pragma unroll
        for (int w = 0; w < NWN; ++w)
            if (wn == w) {
#pragma unroll
                for (int i = 0; i < K3Cfg<NWN>::NCS; ++i)
                    acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[w * K3Cfg<NWN>::NCS + i], ones, acc_cs[i], 0, 0, 0);
            }
    }
    __builtin_amdgcn_s_setprio(0);
}

// Software-pipelined K-step (round 4, second version), in two halves of 16 MFMAs.  In the plain step the DMA issues, the
// fragment reads and their wait sit in front of the 32 MFMAs with the matrix pipe idle -- both waves of a SIMD are in the
// same phase: ~780 of ~1800 cycles per K-step (measured through the per-CU rate: 56 % of the MFMA peak whatever the tile
// and however many CUs are busy).  Here every MFMA runs from fragments that are already in registers, and the DMA pieces
// and the NEXT fragments' LDS reads are issued BETWEEN the MFMAs (sched_group_barrier):
//   half 0: A rows 0-63 (aL) x B of K-step kt   |  reads A rows 64-127 (aH) of K-step kt, issues the DMA of K-step kt+S-1
//   half 1: aH x B                               |  reads aL and B of K-step kt+1 (landed: the barrier at the top of kt)
// Two register sets of half the A fragments (2 x 16 registers) and two of the B fragments (2 x 16) instead of one whole set
// (48): the whole-step double buffer (96) does not fit beside the 128 accumulators.
template <int TA, int TB, bool CS, int NWN, bool SWF = false>
__device__ __forceinline__ void k3_half0(const __amdgpu_buffer_rsrc_t rA, const __amdgpu_buffer_rsrc_t rB,
                                         const int (&voff)[K3Cfg<NWN>::PW], const int soffA, const int soffB, const bool live,
                                         char* __restrict__ wr, const char* __restrict__ rd, const int wave, const int lane,
                                         const int wm, const int wn, const bool do_cs, const bf16x8_t ones,
                                         const bf16x8_t (&aL)[4], const bf16x8_t (&bc)[4], bf16x8_t (&aH)[4],
                                         f32x4_t (&acc)[8][4], f32x4_t (&acc_cs)[K3Cfg<NWN>::NCS]) {
    k3_issue<NWN>(rA, rB, voff, soffA, soffB, live, wr, wave);
    const char* la = rd + wm * K3_SUB;
#pragma unroll
    for (int i = 0; i < 4; ++i) aH[i] = k3_frag<TA>(la, (4 + i) * 16, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = SWF ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[j], aL[i], acc[i][j], 0, 0, 0)    // (C^T tile: k3_epi_direct_f32)
                            : __builtin_amdgcn_mfma_f32_16x16x32_bf16(aL[i], bc[j], acc[i][j], 0, 0, 0);
    constexpr int NDS = (TA ? 2 : 1) * 4;      // LDS read instructions of this half
#pragma unroll
    for (int gI = 0; gI < 4; ++gI) {           // masks: MFMA 0x8, DS read 0x100, VMEM read 0x20
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, (K3Cfg<NWN>::PW + 3) / 4, 0);
    }
    if constexpr (CS) if (do_cs) {
#pragma unroll
        for (int w = 0; w < NWN / 2; ++w)      // the wave positions whose NCS row blocks lie in rows 0-63
            if (wn == w) {
#pragma unroll
                for (int i = 0; i < K3Cfg<NWN>::NCS; ++i)
                    acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aL[w * K3Cfg<NWN>::NCS + i], ones, acc_cs[i], 0, 0, 0);
            }
    }
}
template <int TA, int TB, bool CS, int NWN, bool SWF = false>
__device__ __forceinline__ void k3_half1(const char* __restrict__ rd, const int lane, const int wm, const int wn,
                                         const bool do_cs, const bf16x8_t ones, const bf16x8_t (&aH)[4], const bf16x8_t (&bc)[4],
                                         bf16x8_t (&aL)[4], bf16x8_t (&bn)[4], f32x4_t (&acc)[8][4],
                                         f32x4_t (&acc_cs)[K3Cfg<NWN>::NCS]) {
    const char* la = rd + wm * K3_SUB;
    const char* lb = rd + (2 + (wn >> 1)) * K3_SUB;
#pragma unroll
    for (int j = 0; j < 4; ++j) bn[j] = k3_frag<TB>(lb, (wn & 1) * 64 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) aL[i] = k3_frag<TA>(la, i * 16, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[4 + i][j] = SWF ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[j], aH[i], acc[4 + i][j], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(aH[i], bc[j], acc[4 + i][j], 0, 0, 0);
    constexpr int NDS = (TA ? 2 : 1) * 4 + (TB ? 2 : 1) * 4;
#pragma unroll
    for (int gI = 0; gI < 4; ++gI) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NDS / 4, 0);
    }
    if constexpr (CS) if (do_cs) {
#pragma unroll
        for (int w = NWN / 2; w < NWN; ++w)
            if (wn == w) {
#pragma unroll
                for (int i = 0; i < K3Cfg<NWN>::NCS; ++i)
                    acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aH[(w - NWN / 2) * K3Cfg<NWN>::NCS + i], ones, acc_cs[i], 0, 0, 0);
            }
    }
}

template <int FL, int RB, int H>
__device__ __forceinline__ void k3_epi_half(const vpu_gemm_desc& p, const int FLG, const int vec, f32x4_t (&acc)[RB][4],
                                            const int mw, const int nq, float* wl, const int lane, const int64_t coff) {
    f32x4_t fin[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) fin[i][j] = acc[H * 4 + i < RB ? H * 4 + i : 0][j];
    const int mq = mw + H * 64;
    constexpr int npass = H == 0 ? 4 : RB - 4;
    k2_epi64<true, 16>(p, FLG, vec, fin, mq, nq, wl, lane, mq + 16 * npass, coff);
}
template <int FL, int RB, int H>
__device__ __forceinline__ void k3_epi_fast_half(const vpu_gemm_desc& p, f32x4_t (&acc)[RB][4], const int mw, const int nq,
                                                 float* wl, const int lane, const K2Pre<FL>& q) {
    f32x4_t fin[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) fin[i][j] = acc[H * 4 + i < RB ? H * 4 + i : 0][j];
    k2_epi_fast<FL>(p, fin, mw + H * 64, nq, wl, lane, q, H == 0 ? 4 : RB - 4);
}

// Direct epilogue of the weight-gradient form (fp32 output, flags OUT_F32 [| ACCUM], alpha = 1: the host routes nothing else
// here).  The MFMAs ran with swapped operands, so a lane holds FOUR CONSECUTIVE fp32 COLUMNS of one row per accumulator
// tile -- acc[i][j][r] = C[mw + 16 i + fr][nq + 16 j + 4 fq + r] -- i.e. one 16-byte access, straight from the registers:
// no LDS transposition, no waits on it.  ACCUM: C is requested two row blocks ahead of the adds.
template <int RB>
__device__ __forceinline__ void k3_epi_direct_f32(const vpu_gemm_desc& p, const int FLG, f32x4_t (&acc)[RB][4], const int mw,
                                                  const int nq, const int lane, const int64_t coff, const int M) {
    const int fr = lane & 15, fq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(p.C) + coff, 0, 0x7FFFFFFF, 0x00020000);
    int off[RB][4];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mw + 16 * i + fr, n = nq + 16 * j + 4 * fq;
            off[i][j] = (m < M && n + 4 <= p.N) ? (m * p.ldc + n) * 4 : OOB_OFFSET;
        }
    if (FLG & VPU_EPI_ACCUM) {
        u32x4v c[3][4];
#pragma unroll
        for (int i = 0; i < 2 && i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rC, off[i][j], 0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (i + 2 < RB) {
#pragma unroll
                for (int j = 0; j < 4; ++j) c[(i + 2) % 3][j] = __builtin_amdgcn_raw_buffer_load_b128(rC, off[i + 2][j], 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t v = acc[i][j] + __builtin_bit_cast(f32x4_t, c[i % 3][j]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), rC, off[i][j], 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, acc[i][j]), rC, off[i][j], 0, 0);
    }
}

template <int TA, int TB, bool CS, int FL, bool GRP, int RB, int NWN = 2, bool PIPE = false, int TM = 256, bool SWF = false>
__device__ __forceinline__ void k3_body(const vpu_gemm_desc& p_arg, const vpu_gemm_group* __restrict__ ga,
                                        const int tiles_m_arg, const int tiles_n_arg, const int vec_in
                                        VPU_DBG_PARAM_DEF) {   // (-DVPU_DIAG, vpu_debug_gemm_times: 8 stamps per workgroup)
    using Cf = K3Cfg<NWN, TM>;
    static_assert(!CS || (TA == 1 && RB == 8), "fused column sums: weight-gradient form");
    static_assert(TM == 256 || (RB == 4 && NWN == 2 && !GRP && !PIPE && !CS), "128-row tiles: single problems, 64 x 64 per wave");
    constexpr bool GEN = FL < 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NWN == 2 ? wave >> 1 : wave >> 2, wn = wave & (NWN - 1);
    const int vec = vec_in & 255;
    const int total_work = GRP ? ga->start[ga->n] : tiles_m_arg * tiles_n_arg;
    if ((vec_in >> 8) && (int)blockIdx.x >= (int)(gridDim.x >> 1)) {
        // VPU_GEMM_K3_STAGGER=c (diagnostic): the second half of the grid starts c x 1024 cycles late.  Measured on the forward
        // forms (c = 4, 8, 12): every launch longer by about the delay -- the two workgroups of a CU do not make up for it
        for (int i = 0; i < (vec_in >> 8); ++i) __builtin_amdgcn_s_sleep(16);
    }
    for (int work = blockIdx.x; work < total_work; work += gridDim.x) {
        K2Tile cur;
        int voff[Cf::PW];
        k2_tile_setup<NWN, GRP, RB, GRP>(work, total_work, p_arg, ga, tiles_n_arg, cur);
        k3_voff<TA, TB, RB, NWN, TM>(cur, wave, lane, voff);
        const vpu_gemm_desc& p = GRP ? ga->d[cur.grp] : p_arg;
        const int FLG = GEN ? p.flags : FL;
        const int m0 = cur.m0, n0 = cur.n0;
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(cur.A), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(cur.B), 0, 0x7FFFFFFF, 0x00020000);
        const int stepA = TA ? cur.lda * (K3_BK * 2) : K3_BK * 2, stepB = TB ? cur.ldb * (K3_BK * 2) : K3_BK * 2;
        const int nk = cur.K / K3_BK;

        f32x4_t acc[RB][4];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // fused bias column sums.  Classic form: the tiles of the first column block sum over the whole reduction -- a third
        // of a ViT block's tiles then run 17 % longer than the rest, and a packed launch is ONE round: it lasts as long as
        // its slowest tile (tools/k4_drift.py: 58 of 256 workgroups at 262-277 us, the others at 225-230).  Distributed form
        // (cs_tn > 1, round 4): EVERY column tile of the row block sums over its own 1 / cs_tn of the K-steps and WRITES its
        // partial into row (z * cs_tn + global column tile) of a slab the host adds up afterwards (colsum_batched): equal
        // work per tile, one writer per word, no zeroing.
        const int cs_tn = (CS && p.colsum != nullptr && p.cs_tn > 1) ? p.cs_tn : 0;
        const int cs_gt = cs_tn ? p.cs_t0 + cur.tile_n : 0;
        const int cs_kb = cs_tn ? (int)((int64_t)cs_gt * nk / cs_tn) : 0, cs_ke = cs_tn ? (int)((int64_t)(cs_gt + 1) * nk / cs_tn) : nk;
        const bool do_cs = CS && p.colsum != nullptr && (cs_tn > 0 || cur.tile_n == 0);   // block-uniform
        f32x4_t acc_cs[Cf::NCS];
#pragma unroll
        for (int i = 0; i < Cf::NCS; ++i) acc_cs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        bf16x8_t ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)(((lane & 15) == 0) ? 1.0f : 0.0f);

        if constexpr (PIPE) {
            static_assert(!PIPE || RB == 8, "pipelined form: 256-row tiles");
            // ring: K-step kt lives in stage kt % S; S - 1 stages are requested up front.  At the top of K-step kt: its aL / B
            // fragments are in registers, its aH half is still in stage kt (read during half 0), K-step kt+1 has landed,
            // kt+2 .. kt+S-2 fly, kt+S-1 is issued into the stage K-step kt-1 has left.
#pragma unroll
            for (int st = 0; st < Cf::S - 1; ++st)
                k3_issue<NWN>(rA, rB, voff, st * stepA, st * stepB, st < nk, lds + st * Cf::STAGE, wave);
            bf16x8_t aL[4], aH[4], b0[4], b1[4];
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((Cf::S - 2) * Cf::PW) : "memory");
            __builtin_amdgcn_s_barrier();
            {
                const char* la = lds + wm * K3_SUB;
                const char* lb = lds + (2 + (wn >> 1)) * K3_SUB;
#pragma unroll
                for (int j = 0; j < 4; ++j) b0[j] = k3_frag<TB>(lb, (wn & 1) * 64 + j * 16, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) aL[i] = k3_frag<TA>(la, i * 16, lane);
            }
            int st_c = 0, st_w = Cf::S - 1;      // stage of K-step kt / stage the next DMA goes to
            VPU_STAMP(tid == 0, blockIdx.x * 8 + 0);
            const int q1 = (nk >> 2) & ~1, q2 = (nk >> 1) & ~1, q3 = (3 * nk >> 2) & ~1;
            for (int kt = 0; kt < nk; kt += 2) {
                VPU_STAMP(tid == 0 && (kt == q1 || kt == q2 || kt == q3), blockIdx.x * 8 + (kt == q1 ? 1 : kt == q2 ? 2 : 3));
                // (two K-steps per iteration: the B register sets swap roles, every index is a compile-time constant)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && kt + 1 >= nk) break;
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((Cf::S - 3) * Cf::PW) : "memory");   // K-step kt+1 has landed (mine)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             // aL / B of this K-step are in registers
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    const int st_n = st_c + 1 == Cf::S ? 0 : st_c + 1;
                    const int kn = kt + u + Cf::S - 1;
                    __builtin_amdgcn_s_setprio(1);
                    const bool cs_now = do_cs && kt + u >= cs_kb && kt + u < cs_ke;
                    if (u == 0) k3_half0<TA, TB, CS, NWN, SWF>(rA, rB, voff, kn * stepA, kn * stepB, kn < nk, lds + st_w * Cf::STAGE, lds + st_c * Cf::STAGE,
                                                               wave, lane, wm, wn, cs_now, ones, aL, b0, aH, acc, acc_cs);
                    else k3_half0<TA, TB, CS, NWN, SWF>(rA, rB, voff, kn * stepA, kn * stepB, kn < nk, lds + st_w * Cf::STAGE, lds + st_c * Cf::STAGE,
                                                        wave, lane, wm, wn, cs_now, ones, aL, b1, aH, acc, acc_cs);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             // aH is in registers
                    __builtin_amdgcn_sched_barrier(0);
                    if (u == 0) k3_half1<TA, TB, CS, NWN, SWF>(lds + st_n * Cf::STAGE, lane, wm, wn, cs_now, ones, aH, b0, aL, b1, acc, acc_cs);
                    else k3_half1<TA, TB, CS, NWN, SWF>(lds + st_n * Cf::STAGE, lane, wm, wn, cs_now, ones, aH, b1, aL, b0, acc, acc_cs);
                    __builtin_amdgcn_s_setprio(0);
                    st_w = st_c;
                    st_c = st_n;
                }
            }
        } else {
        // ring: K-step kt lives in stage kt % S; S - 1 K-steps are requested ahead
#pragma unroll
        for (int st = 0; st < Cf::S - 1; ++st)
            k3_issue<NWN, TM>(rA, rB, voff, st * stepA, st * stepB, st < nk, lds + st * Cf::STAGE, wave);
        int rd_i = 0, wr_i = Cf::S - 1;
        for (int kt = 0; kt < nk; ++kt) {
            // this wave's pieces of K-step kt have landed (those of the S - 2 later ones may still fly); after the barrier so
            // have everybody's, and every wave is done reading K-step kt-1, whose stage the next DMA overwrites
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cf::INFL) : "memory");
            __builtin_amdgcn_s_barrier();
            const int kn = kt + Cf::S - 1;
            k3_step<TA, TB, CS, RB, NWN, TM>(rA, rB, voff, kn * stepA, kn * stepB, kn < nk, lds + wr_i * Cf::STAGE, lds + rd_i * Cf::STAGE,
                                             wave, lane, wm, wn, do_cs && kt >= cs_kb && kt < cs_ke, ones, acc, acc_cs);
            rd_i = rd_i + 1 == Cf::S ? 0 : rd_i + 1;
            wr_i = wr_i + 1 == Cf::S ? 0 : wr_i + 1;
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the out-of-range tail pieces still write zeros into LDS)
        __syncthreads();
        VPU_STAMP(tid == 0, blockIdx.x * 8 + 4);
        float* wl = reinterpret_cast<float*>(lds + wave * 4096);   // 16 rows x 64 fp32 per wave
        const int mw = m0 + wm * (16 * RB), nq = n0 + wn * 64;
        if (vec == 9) {  // diagnostic (VPU_GEMM_NOEPI=1): main loop only
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (t == 1.2345678e30f) reinterpret_cast<float*>(p.C)[0] = t;
        } else if constexpr (GEN) {
            if constexpr (CS) if (do_cs && (lane & 15) == 0) {
                const int fq = lane >> 4;
#pragma unroll
                for (int i = 0; i < Cf::NCS; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + wm * 128 + (wn * Cf::NCS + i) * 16 + fq * 4 + r;
                        if (row < cur.M) {
                            if (cs_tn) p.colsum[((int64_t)cur.z * cs_tn + cs_gt) * p.cs_ld + row] = acc_cs[i][r];   // (this tile's slab row)
                            else p.colsum[(int64_t)cur.z * cur.M + row] += acc_cs[i][r];   // (a batch entry's sums: colsum + z M)
                        }
                    }
            }
            // (the two 64-row halves of the wave tile, written out twice: a loop over them that the compiler does not unroll
            // indexes the accumulators at run time and sends all 128 of them to scratch)
            if constexpr (SWF) {
                static_assert(!SWF || (PIPE && GRP), "direct fp32 epilogue: the pipelined grouped weight-gradient form");
                k3_epi_direct_f32<RB>(p, FLG, acc, mw, nq, lane, cur.coff, cur.M);
            } else {
                k3_epi_half<FL, RB, 0>(p, FLG, vec, acc, mw, nq, wl, lane, cur.coff);
                if constexpr (RB > 4) k3_epi_half<FL, RB, 1>(p, FLG, vec, acc, mw, nq, wl, lane, cur.coff);
            }
        } else {
            // everything the epilogue reads from global memory is requested before its first store
            K2Pre<GEN ? 0 : FL> q0, q1;
            k2_prefetch<GEN ? 0 : FL>(p, mw, nq, lane, q0, 4);
            if constexpr (RB > 4) k2_prefetch<GEN ? 0 : FL>(p, mw + 64, nq, lane, q1, RB - 4);
            k3_epi_fast_half<GEN ? 0 : FL, RB, 0>(p, acc, mw, nq, wl, lane, q0);
            if constexpr (RB > 4) k3_epi_fast_half<GEN ? 0 : FL, RB, 1>(p, acc, mw, nq, wl, lane, q1);
        }
        // every wave is done with its epilogue scratch before the next tile's DMA lands in stage 0 (raw barrier: the global
        // stores stay in flight)
        __builtin_amdgcn_s_barrier();
    }
}

template <int TA, int TB, int FL, int RB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_k3_kernel(const vpu_gemm_desc p, const int tiles_m, const int tiles_n,
                                                              const int vec) {
    k3_body<TA, TB, false, FL, false, RB>(p, nullptr, tiles_m, tiles_n, vec);
}
template <int TA, int TB, int FL>
__global__ __launch_bounds__(256, 3) void gemm_bf16_k3s_kernel(const vpu_gemm_desc p, const int tiles_m, const int tiles_n,
                                                               const int vec) {
    k3_body<TA, TB, false, FL, false, 4, 2, false, 128>(p, nullptr, tiles_m, tiles_n, vec);
}
template <int TA, int TB, bool CS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_k3_grouped_kernel(const vpu_gemm_group ga_unused, const int vec) {
    const vpu_gemm_group* ga = (const vpu_gemm_group*)__builtin_amdgcn_kernarg_segment_ptr();
    k3_body<TA, TB, CS, -1, true, 8>(ga->d[0], ga, 0, 0, vec);
}
template <int TA, int TB, bool CS>
__global__ __launch_bounds__(512) void gemm_bf16_k4_grouped_kernel(const vpu_gemm_group ga_unused, const int vec) {
    const vpu_gemm_group* ga = (const vpu_gemm_group*)__builtin_amdgcn_kernarg_segment_ptr();
    k3_body<TA, TB, CS, -1, true, 8, 4>(ga->d[0], ga, 0, 0, vec);
}
template <int TA, int TB, bool CS, bool SWF = false>
__global__ __launch_bounds__(512) void gemm_bf16_k4p_grouped_kernel(const vpu_gemm_group ga_unused, const int vec VPU_DBG_PARAM) {
    const vpu_gemm_group* ga = (const vpu_gemm_group*)__builtin_amdgcn_kernarg_segment_ptr();
    k3_body<TA, TB, CS, -1, true, 8, 4, true, 256, SWF>(ga->d[0], ga, 0, 0, vec VPU_DBG_PASS);
}

// ------------------------------------------------------------------------------------------------
// Skinny problems: the DMA neck's prompt-token GEMMs (M = B*48 = 576 rows, N, K <= 2048; at inference M = 96).  With the
// 128x128 tile they are 15-30 tiles of 12+ K-tiles each, run as split-K slabs + a reduce launch (~16 us per GEMM, 50 of
// them per training step).  Here a workgroup owns one 64x64 output tile and its FOUR WAVES SPLIT K: every wave walks a
// quarter of K for the whole tile through its own private LDS images (no block barrier in the loop, the next K-step's
// operands in registers while the current one is multiplied), the four partial tiles are summed through LDS in wave
// order (deterministic) and go through the full epilogue: one launch, no slabs.
// TB = 1 (dgrad: B is [K][N]): the B pieces go into a K-major image and are read with ds_read_b64_tr_b16.
// ------------------------------------------------------------------------------------------------
constexpr int SK_T = 64;

template <int TA, int TB>
__device__ __forceinline__ void skinny_body(const vpu_gemm_desc& p, const int tile_m, const int tile_n, const int kw,
                                            const int vec) {
    static_assert(TA == 0 || TB == 1, "K-major A (weight-gradient form) comes with a K-major B");
    extern __shared__ __attribute__((aligned(16))) char lds[];   // per wave: one K-contiguous image (+ one K-major image, TB = 1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = tile_m * SK_T, n0 = tile_n * SK_T;
    const int kbeg = wave * kw, kend = (kbeg + kw < p.K) ? kbeg + kw : p.K;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, 0x7FFFFFFF, 0x00020000);
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // Every operand goes global -> registers -> the wave's PRIVATE LDS image -> MFMA fragments: the global loads are whole
    // 128-byte row segments (8 rows x 8 chunks per wave-instruction); fragment-shaped loads straight from global memory
    // (16 rows x 64 B) measured 9.7 us against 7.3 us for the 576 x 768 x 768 problem.  Wave-local: a wave's LDS
    // operations complete in order, so no barrier is needed inside the loop.
    char* imgK = lds + wave * (TB ? 2 : 1) * TILE_BYTES;     // [128 rows][64 k]: rows 0..63 = A, rows 64..127 = B (TB = 0);
                                                             // TA = 1: [64 k][128 cols] K-major image of A, columns 0..63
    char* imgB = imgK + TILE_BYTES;                          // TB = 1: [64 k][128 cols] K-major image of B, columns 0..63
    // fused bias gradient of the weight-gradient form: column sums of A = one more MFMA per A fragment against a fragment
    // whose column 0 is all ones (the n = 0 tiles only)
    const bool do_cs = TA == 1 && p.colsum != nullptr && tile_n == 0;
    f32x4_t acc_cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_cs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)(((lane & 15) == 0) ? 1.0f : 0.0f);
    // piece pc of a K-contiguous operand: rows 8 pc .. + 8, lane -> (row 8 pc + lane / 8, 16-byte chunk lane % 8)
    auto ldKC = [&](__amdgpu_buffer_rsrc_t r, int ld, int x0, int X, int pc, int k0) -> u32x4v {
        const int gx = x0 + pc * 8 + (lane >> 3), gk = k0 + (lane & 7) * 8;
        const int off = (gx < X && gk < kend) ? (gx * ld + gk) * 2 : OOB_OFFSET;
        return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    };
    // piece pc of the K-major B: k rows 8 pc .. + 8, lane -> (k = 8 pc + lane / 8, 8 columns (lane % 8) * 8)
    auto ldKM = [&](int pc, int k0) -> u32x4v {
        const int gk = k0 + pc * 8 + (lane >> 3), gn = n0 + (lane & 7) * 8;
        const int off = (gk < kend && gn < p.N) ? (gk * p.ldb + gn) * 2 : OOB_OFFSET;
        return __builtin_amdgcn_raw_buffer_load_b128(rB, off, 0, 0);
    };
    // piece pc of a K-major A (TA = 1)
    auto ldKMa = [&](int pc, int k0) -> u32x4v {
        const int gk = k0 + pc * 8 + (lane >> 3), gm = m0 + (lane & 7) * 8;
        const int off = (gk < kend && gm < p.M) ? (gk * p.lda + gm) * 2 : OOB_OFFSET;
        return __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, 0);
    };
    const int nsteps = kend > kbeg ? (kend - kbeg + 63) / 64 : 0;
    u32x4v na[8], nb[8];
    if (nsteps > 0) {
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            na[pc] = TA ? ldKMa(pc, kbeg) : ldKC(rA, p.lda, m0, p.M, pc, kbeg);
            nb[pc] = TB ? ldKM(pc, kbeg) : ldKC(rB, p.ldb, n0, p.N, pc, kbeg);
        }
    }
    for (int st = 0; st < nsteps; ++st) {
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            const int r8 = pc * 8 + (lane >> 3);
            if (TA) *reinterpret_cast<u32x4v*>(imgK + km_off(r8, (lane & 7) * 2)) = na[pc];
            else *reinterpret_cast<u32x4v*>(imgK + kc_off(r8, lane & 7)) = na[pc];
            if (TB) *reinterpret_cast<u32x4v*>(imgB + km_off(r8, (lane & 7) * 2)) = nb[pc];
            else *reinterpret_cast<u32x4v*>(imgK + kc_off(64 + r8, lane & 7)) = nb[pc];
        }
        if (st + 1 < nsteps) {   // the next step's operands fly under this step's fragment reads and MFMAs
            const int k1 = kbeg + (st + 1) * 64;
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) {
                na[pc] = TA ? ldKMa(pc, k1) : ldKC(rA, p.lda, m0, p.M, pc, k1);
                nb[pc] = TB ? ldKM(pc, k1) : ldKC(rB, p.ldb, n0, p.N, pc, k1);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = TA ? read_frag<1>(imgK, i * 16, ks, lane) : read_frag<0>(imgK, i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = TB ? read_frag<1>(imgB, j * 16, ks, lane) : read_frag<0>(imgK, 64 + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if constexpr (TA == 1) {
                if (do_cs) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc_cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, acc_cs[i], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this step's fragment reads are done before the images are rewritten
    }
    __syncthreads();   // every wave is done with its images: the partial tiles reuse the same LDS
    // ---- the four waves' partial tiles -> LDS [wave][64 rows][64 cols] (column group XOR-swizzled by the row group)
    float* wl = reinterpret_cast<float*>(lds) + wave * (SK_T * SK_T);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + fq * 4 + r;
                wl[row * 64 + ((j * 16 + fr) ^ (fq << 4))] = acc[i][j][r];
            }
    float* csl = reinterpret_cast<float*>(lds) + 4 * (SK_T * SK_T);   // [wave][64]: the waves' partial column sums (TA = 1)
    if constexpr (TA == 1) {
        if (do_cs && fr == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) csl[wave * 64 + i * 16 + fq * 4 + r] = acc_cs[i][r];
        }
    }
    __syncthreads();
    if constexpr (TA == 1) {
        if (do_cs && tid < SK_T && m0 + tid < p.M)
            p.colsum[m0 + tid] += ((csl[tid] + csl[64 + tid]) + csl[128 + tid]) + csl[192 + tid];
    }
    // thread -> row tid / 4, 16 columns (tid % 4) * 16: two groups of 8, summed over the waves in wave order
    const float* all = reinterpret_cast<const float*>(lds);
    const int row = tid >> 2;
    const int m = m0 + row;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c8 = (tid & 3) * 16 + h * 8;
        const int n = n0 + c8;
        if (m < p.M && n < p.N) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w_ = 0; w_ < 4; ++w_) {
                float t[8];
                load8(all + w_ * (SK_T * SK_T) + row * 64 + (c8 ^ (((row >> 2) & 3) << 4)), t);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += t[j];
            }
            if (vec && n + 8 <= p.N) {
                EpiPre e;
                e.has_pre = false; e.has_bias = false; e.pre = make_uint4(0, 0, 0, 0);
                epilogue_store8(p, p.flags, 0, 0, m, n, v, e);
            } else {
                for (int j = 0; j < 8 && n + j < p.N; ++j) epilogue_store<bf16_t>(p, 0, 0, m, n + j, v[j]);
            }
        }
    }
}

template <int TB>
__global__ __launch_bounds__(256) void gemm_bf16_skinny_kernel(const vpu_gemm_desc p, const int tiles_n, const int kw,
                                                               const int vec) {
    const int tile_m = blockIdx.x / tiles_n;
    skinny_body<0, TB>(p, tile_m, blockIdx.x - tile_m * tiles_n, kw, vec);
}
// Grouped form: the 64 x 64 tiles of up to 16 independent skinny problems in one launch (ga.start[] = first tile of each
// problem; the q / k / v projections of the neck's prompt-token attentions and their dgrads: three 576-row problems took
// ~24 us as 128 x 128 tiles walking all of K in the general grouped kernel, or three launches of ~8 us each).
template <int TA, int TB>
__global__ __launch_bounds__(256) void gemm_bf16_skinny_grouped_kernel(const vpu_gemm_group ga_unused, const int vec) {
    const vpu_gemm_group* ga = (const vpu_gemm_group*)__builtin_amdgcn_kernarg_segment_ptr();
    int grp = 0;
    while (grp + 1 < ga->n && (int)blockIdx.x >= ga->start[grp + 1]) ++grp;
    grp = __builtin_amdgcn_readfirstlane(grp);
    const vpu_gemm_desc& p = ga->d[grp];
    const int local = blockIdx.x - ga->start[grp];
    const int tiles_n = (p.N + SK_T - 1) / SK_T;
    const int tile_m = local / tiles_n;
    const int kw = ((p.K + 3) / 4 + 63) / 64 * 64;
    skinny_body<TA, TB>(p, tile_m, local - tile_m * tiles_n, kw, vec);
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const vpu_gemm_desc p, const int splitk,
                                                            const float* __restrict__ ws, const int vec8) {
    const int z = blockIdx.z, zo = z / p.inner, zi = z % p.inner;
    const int64_t coff = zo * p.sCo + zi * p.sCi, roff = zo * p.sRo + zi * p.sRi;
    const int64_t mn = (int64_t)p.M * p.N;
    const float* w = ws + (int64_t)z * splitk * mn;
    if (vec8 == 2) {
        // many slices, few elements (weight gradients over 150528 rows: 128 slices of a 256 x 128 output): 8 lanes share one
        // group of 8 columns, lane j sums the slices j, j+8, ... and the eight partial sums are added in lane order through
        // LDS (fixed tree: deterministic).  One thread per group walked 128 slabs back to back: 63 us for 16 MB.
        // (32 slice lanes x 8 column groups per block: with 8 slice lanes x 32 groups a 256 x 128 output was 128 blocks of 16
        // dependent 32-byte loads per lane -- 35 us for 16.8 MB)
        __shared__ float red[32][8][8];
        const int eg = threadIdx.x & 7, sl = threadIdx.x >> 3;
        const int n8 = p.N >> 3;
        const int64_t i = (int64_t)blockIdx.x * 8 + eg;
        const bool live = i < (mn >> 3);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (live) {
#pragma unroll 4
            for (int s = sl; s < splitk; s += 32) {
                float t[8];
                load8(w + s * mn + i * 8, t);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += t[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[sl][eg][j] = v[j];
        __syncthreads();
        if (sl == 0 && live) {
#pragma unroll 4
            for (int q = 1; q < 32; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += red[q][eg][j];
            EpiPre e;
            e.has_pre = false; e.has_bias = false; e.pre = make_uint4(0, 0, 0, 0);
            epilogue_store8(p, p.flags, coff, roff, (int)(i / n8), (int)(i % n8) * 8, v, e);
        }
    } else if (vec8) {   // N % 8 == 0 and every epilogue operand 16/32-byte addressable: 8 columns of one row per thread
        const int n8 = p.N >> 3;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (mn >> 3); i += (int64_t)gridDim.x * 256) {
            const int m = (int)(i / n8), n = (int)(i % n8) * 8;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < splitk; ++s) {
                float t[8];
                load8(w + s * mn + i * 8, t);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += t[j];
            }
            EpiPre e;
            e.has_pre = false; e.has_bias = false; e.pre = make_uint4(0, 0, 0, 0);
            epilogue_store8(p, p.flags, coff, roff, m, n, v, e);
        }
    } else
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < mn; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < splitk; ++s) v += w[s * mn + i];
        epilogue_store<T>(p, coff, roff, (int)(i / p.N), (int)(i % p.N), v);
    }
    if (p.colsum != nullptr && z == 0) {
        // fused bias gradient: 32 slice lanes x 8 rows per block, partial sums added in lane order through LDS (one
        // thread per row walking up to 128 slices back to back was a ~30-us tail of the whole reduce launch)
        __shared__ float redc[32][8];
        const float* wb = ws + (int64_t)gridDim.z * splitk * mn;
        const int mi = threadIdx.x & 7, sl = threadIdx.x >> 3;
        for (int mb = blockIdx.x * 8; mb < p.M; mb += gridDim.x * 8) {   // block-uniform trip count
            const int m = mb + mi;
            float t = 0.f;
            if (m < p.M)
                for (int s = sl; s < splitk; s += 32) t += wb[(int64_t)s * p.M + m];
            __syncthreads();
            redc[sl][mi] = t;
            __syncthreads();
            if (sl == 0 && m < p.M) {
                for (int q = 1; q < 32; ++q) t += redc[q][mi];
                p.colsum[m] += t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// exact-fp32 MFMA kernel (parity mode).  64x64x16 block tile, 4 waves (2x2) of 32x32.
// LDS images are k-major [16][64+pad] for both operands whatever the source layout.
// ------------------------------------------------------------------------------------------------
constexpr int FM = 64, FN = 64, FK = 16, FLD = 68;

template <int TRANS>
__device__ __forceinline__ void f32_stage(const float* base, int ld, int x0, int X, int k0, int K, float* lds,
                                          int tid) {
    // 64 x 16 elements = 256 float4
    if (TRANS == 0) {
        const int row = tid >> 2, kc = (tid & 3) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int gx = x0 + row, gk = k0 + kc;
        if (gx < X && gk < K) v = *reinterpret_cast<const float4*>(base + (int64_t)gx * ld + gk);
        // K is a multiple of 4 for every caller (zero padded), so a float4 is all-valid or all-out
        lds[(kc + 0) * FLD + row] = v.x;
        lds[(kc + 1) * FLD + row] = v.y;
        lds[(kc + 2) * FLD + row] = v.z;
        lds[(kc + 3) * FLD + row] = v.w;
    } else {
        const int k = tid >> 4, cc = (tid & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int gk = k0 + k, gx = x0 + cc;
        if (gk < K && gx < X) v = *reinterpret_cast<const float4*>(base + (int64_t)gk * ld + gx);
        *reinterpret_cast<float4*>(&lds[k * FLD + cc]) = v;
    }
}

template <int TA, int TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const vpu_gemm_desc p, const int tiles_n) {
    __shared__ __attribute__((aligned(16))) float ldsA[FK * FLD];
    __shared__ __attribute__((aligned(16))) float ldsB[FK * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * FM, n0 = tile_n * FN;
    const int z = blockIdx.z, zo = z / p.inner, zi = z % p.inner;
    const float* A = reinterpret_cast<const float*>(p.A) + zo * p.sAo + zi * p.sAi;
    const float* B = reinterpret_cast<const float*>(p.B) + zo * p.sBo + zi * p.sBi;
    const int64_t coff = zo * p.sCo + zi * p.sCi;
    const int64_t roff = zo * p.sRo + zi * p.sRi;

    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int nk = (p.K + FK - 1) / FK;
    for (int kt = 0; kt < nk; ++kt) {
        f32_stage<TA>(A, p.lda, m0, p.M, kt * FK, p.K, ldsA, tid);
        f32_stage<TB>(B, p.ldb, n0, p.N, kt * FK, p.K, ldsB, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = ks * 4 + fq;
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = ldsA[k * FLD + wm * 32 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = ldsB[k * FLD + wn * 32 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 32 + i * 16 + fq * 4 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 32 + j * 16 + fr;
                if (n < p.N) epilogue_store<float>(p, coff, roff, m, n, acc[i][j][r]);
            }
        }
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ring-pipeline selection: 0 off (default), 1 one-wave problems, 2 everywhere; VPU_GEMM_RING at start-up,
// vpu_gemm_set_option("ring", v) at run time (tests)
// name of the kernel instantiation the last vpu_gemm / vpu_gemm_grouped call of this host thread launched, as rocprofv3
// prints it (vpu_gemm_last_kernel): lets a profiler harness label launches without mirroring the dispatch rules
thread_local char g_last_kernel[160] = "";
#define NOTE_KERNEL(...) snprintf(g_last_kernel, sizeof(g_last_kernel), __VA_ARGS__)
std::atomic<int> g_opt_ring{-1};
// split-K slices combined by a separate reduce launch (0, default) or by the last-arriving workgroup of the same launch
// (1: always, n > 1: only when all slabs of the launch total <= n MiB): VPU_GEMM_INLAUNCH at start-up,
// vpu_gemm_set_option("splitk_inlaunch", v) at run time.  Measured on the training step (round 1, bs 12): off 19.5 ms,
// slabs <= 2 MiB 19.8 ms, <= 8 MiB 21.5 ms, always 27.4 ms (24.1 ms with an agent-scope acquire + plain loads instead of
// sc1 loads: the acquire drops the XCD's L2) -- the write-through slab stores and sc1 loads cost more than the reduce
// launch they save, at every slab size of this model.  Kept (and tested) for shapes where a launch boundary is dearer.
std::atomic<int> g_opt_inlaunch{-1};
std::atomic<int> g_opt_skinny{-1};
// K2 kernels (256-row tiles, 128 x 64 per wave): -1 environment default (VPU_GEMM_K2, 2 if unset), 0 off, 1 the 256 x 128
// form only, 2 also the 256 x 256 form where its tile count fills the chip, 3 the 256 x 256 form wherever it is legal
std::atomic<int> g_opt_skinny_group{-1};
std::atomic<int> g_opt_k2{-1};
// K3 kernels (256 x 128 tiles in 256-thread workgroups, two per CU): bit 0 the grouped weight gradients, bit 1 the
// forward / dgrad forms of vpu_gemm, bit 2 "K3S": the 128 x 128 ring kernel for short-K launches of the round-1 kernel, bit 3
// "K4": grouped weight gradients as 256 x 256 tiles in one 512-thread workgroup per CU with a five-stage ring, bit 4 its
// software-pipelined K-step, bit 5 K3S for every K; -1 environment default (VPU_GEMM_K3, 28 if unset)
std::atomic<int> g_opt_k3{-1};
inline int k3_env0() {
    static const int v = [] { const char* e = getenv("VPU_GEMM_K3"); return e ? atoi(e) : 28; }();   // K3S + K4 + pipelined K4
    return v;
}
// The product library (no -DVPU_LAB) carries the kernel families the engine dispatches; the others -- K3 forward / dgrad forms
// and K3 / K2 grouped weight gradients (k3 bits 0-1), the non-pipelined K4 (bit 3 without bit 4), the three-stage ring forms of
// the 128 x 128 kernel, the 256 x 128 three-stage kernel (VPU_GEMM_BIG) and the LDS-transposed K2 epilogue
// (VPU_GEMM_K2_DIRECT=0) -- are compiled into the laboratory library only (`build.sh diag` -> libvpu_hip_diag.so, loaded with
// VPU_LIB_DIAG=1 by the tools and by the tests marked `lab`); their option bits read as off here.
inline int k3_opt() {
    int v = g_opt_k3.load(std::memory_order_relaxed);
    v = v >= 0 ? v : k3_env0();
#ifndef VPU_LAB
    v &= ~3;
    if (v & 8) v |= 16;
#endif
    return v;
}
inline int k2_env0() {
    static const int v = [] { const char* e = getenv("VPU_GEMM_K2"); return e ? atoi(e) : 2; }();
    return v;
}
inline int k2_opt() { const int v = g_opt_k2.load(std::memory_order_relaxed); return v >= 0 ? v : k2_env0(); }
// CUs the persistent launches leave unclaimed (vpu_gemm_set_option("reserve_cus")): room for the channel kernels of a
// collective that runs beside backward (pvpuformer_amd/parallel.py)
std::atomic<int> g_opt_reserve{0};
// diagnostic (vpu_debug_gemm_times): device buffer of 8 cycle-counter stamps per workgroup of the K4P kernel, or null
#ifdef VPU_DIAG
std::atomic<unsigned long long*> g_dbg_times{nullptr};
#endif
inline int cu_count() {
    static const int v = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int r = g_opt_reserve.load(std::memory_order_relaxed);
    const int left = (v - r) & ~7;       // a multiple of 8: the work index's low bits stay the XCD label
    return left >= 64 ? left : v;
}   // -1 environment default (VPU_GEMM_SKINNY, 1 if unset), 0 off, 1 on
inline int inlaunch_env0() {
    static const int v = [] { const char* e = vpu_lab_getenv("VPU_GEMM_INLAUNCH"); return e ? atoi(e) : 0; }();
    return v;
}
constexpr int64_t CNT_BYTES = 256 << 10;   // tile arrival counters at the end of the split-K workspace
inline int ring_env0() {
    static const int v = [] { const char* e = vpu_lab_getenv("VPU_GEMM_RING"); return e ? atoi(e) : 0; }();
    return v;
}

}  // namespace

extern "C" int vpu_gemm(const vpu_gemm_desc* d, void* stream) {
    vpu_clear_stale_error();
    if (!d || !d->A || !d->B || !d->C) { vpu_set_error("vpu_gemm: null operand"); return VPU_ERR_ARG; }
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0 || d->inner <= 0 || d->batch % d->inner) {
        vpu_set_error("vpu_gemm: bad sizes (M,N,K,batch > 0; batch % inner == 0)");
        return VPU_ERR_ARG;
    }
    if (d->colsum && d->cs_tn > 1) {
        vpu_set_error("vpu_gemm: cs_tn > 1 (distributed column sums) is a vpu_gemm_grouped form");
        return VPU_ERR_ARG;
    }
    const int f = d->flags;
    if (((f & VPU_EPI_BIAS) && !d->bias) || ((f & VPU_EPI_RESID) && !d->resid) ||
        ((f & (VPU_EPI_DGELU | VPU_EPI_DRELU | VPU_EPI_MULAUX)) && !d->aux) ||
        ((f & (VPU_EPI_PREACT | VPU_EPI_SAVE_DGELU)) && !d->preact) ||
        ((f & VPU_EPI_SAVE_DGELU) && (!(f & VPU_EPI_GELU) || (f & VPU_EPI_PREACT)))) {
        vpu_set_error("vpu_gemm: epilogue flag set but its pointer is null");
        return VPU_ERR_ARG;
    }
    const bool bf = d->dtype == VPU_BF16;
    if (!bf && d->dtype != VPU_F32) { vpu_set_error("vpu_gemm: dtype"); return VPU_ERR_ARG; }
    if (d->colsum && (!bf || !d->transA || d->batch != 1)) {
        vpu_set_error("vpu_gemm: colsum needs the bf16 path, transA=1 and batch=1");
        return VPU_ERR_ARG;
    }
    const int64_t q = bf ? 8 : 4;  // elements per 16 B
    const bool strides_ok = d->lda % q == 0 && d->ldb % q == 0 && d->sAo % q == 0 && d->sAi % q == 0 &&
                            d->sBo % q == 0 && d->sBi % q == 0;
    if (!strides_ok || !aligned_to(d->A, 16) || !aligned_to(d->B, 16)) {
        vpu_set_error("vpu_gemm: A/B base, leading dimensions and batch strides must be 16-byte multiples");
        return VPU_ERR_ALIGN;
    }
    if (bf) {  // LDS-DMA staging addresses each operand with a 31-bit byte offset from its (per-batch) base
        const int64_t ea = d->transA ? (int64_t)d->K * d->lda : (int64_t)d->M * d->lda;
        const int64_t eb = d->transB ? (int64_t)d->K * d->ldb : (int64_t)d->N * d->ldb;
        if (ea * 2 >= 0x7FFFFFF0LL || eb * 2 >= 0x7FFFFFF0LL) {
            vpu_set_error("vpu_gemm: an operand spans more than 2 GiB per batch entry");
            return VPU_ERR_ARG;
        }
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const int force_big = [] { const char* e = vpu_lab_getenv("VPU_GEMM_BIG"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    // the 256x128 three-stage kernel is kept selectable (VPU_GEMM_BIG=1) but is off by default: at these problem sizes
    // (<= 3.5 rounds of tiles) it measured 5-25 % slower than two co-resident 128x128 blocks per CU (round 1).
#ifdef VPU_LAB
    const bool big = bf && force_big == 1 && d->M >= 1024 && d->N >= 128 && d->K >= 256;
#else
    const bool big = false;
    (void)force_big;
#endif
    const int tm = bf ? (big ? BM2 : BM) : FM, tn = bf ? BN : FN;
    const int tiles_m = (d->M + tm - 1) / tm, tiles_n = (d->N + tn - 1) / tn;
    const int key = (d->transA ? 2 : 0) | (d->transB ? 1 : 0);
    if (bf) {
        // vector epilogue: everything the epilogue touches is addressable in 8-element (16/32 B) units
        const size_t cbytes = (f & VPU_EPI_OUT_F32) ? 32 : 16;
        // (a ragged N is fine: the kernel sends the last partial group of 8 columns through the scalar path)
        bool vec = d->ldc % 8 == 0 && d->sCo % 8 == 0 && d->sCi % 8 == 0 && aligned_to(d->C, cbytes);
        if (f & VPU_EPI_BIAS) vec = vec && aligned_to(d->bias, 32);
        if (f & (VPU_EPI_PREACT | VPU_EPI_SAVE_DGELU)) vec = vec && aligned_to(d->preact, 16);
        if (f & (VPU_EPI_DGELU | VPU_EPI_DRELU | VPU_EPI_MULAUX)) vec = vec && d->ldaux % 8 == 0 && aligned_to(d->aux, 16);
        if (f & VPU_EPI_RESID)
            vec = vec && d->ldr % 8 == 0 && d->sRo % 8 == 0 && d->sRi % 8 == 0 && aligned_to(d->resid, 16);
        // split-K: few output tiles and a long reduction (weight gradients, cosine-logit gradients)
        const int64_t tiles = (int64_t)tiles_m * tiles_n * d->batch;
        int splitk = 1, kchunk = (d->K + BK - 1) / BK * BK;
        // VPU_GEMM_RING=1: one-wave problems (96..256 tiles, e.g. the DMA neck's 9408 x 384 x 768 projections) run one tile
        // per workgroup, no split-K, on the three-stage ring.  Off by default -- measured from a hipGraph replay
        // (tools/gemm_bench.py, GEMM_BENCH_GRAPH=1): 9408x384x768 18.9 us (ring, generic epilogue) vs 16.1 us (two-stage,
        // specialised epilogue); the 576-row token GEMMs (15-30 tiles) 15.4 us vs 12.4 us for split-K + reduce: too few
        // bytes in flight per CU.  The ring pays at one workgroup per CU with long K; kept for the next tile shapes.
        const int ring_env = g_opt_ring.load(std::memory_order_relaxed) >= 0 ? g_opt_ring.load(std::memory_order_relaxed) : ring_env0();
        static const int ring_min = [] { const char* e = vpu_lab_getenv("VPU_GEMM_RING_MIN"); return e ? atoi(e) : 96; }();
#ifdef VPU_LAB
        const bool ring = !big && (ring_env == 2 || (ring_env == 1 && tiles >= ring_min && tiles <= 256 && d->K <= 24 * BK && d->K > 2 * BK));
#else
        const bool ring = false;
        (void)ring_env; (void)ring_min;
#endif
        // (192 ... 256 tiles walking a long K -- the FPN's 2352-row convolution over K = 3072: 228 tiles, 48 K-tiles each on
        // half of the 512 workgroup slots -- are cut in two as well: 55 -> ~35 us with the reduce)
        if (!ring && d->workspace && (tiles < 192 || (tiles <= 256 && d->K >= 32 * BK)) && d->K >= 8 * BK) {
            int64_t want = (384 + tiles - 1) / tiles;
            const int64_t max_by_k = d->K / (4 * BK);
            const int64_t max_by_ws = (d->workspace_bytes - CNT_BYTES) / ((int64_t)d->batch * d->M * (d->N + 1) * 4);
            if (want > max_by_k) want = max_by_k;
            if (want > max_by_ws) want = max_by_ws;
            if (want > 128) want = 128;
            if (want > 1) {
                kchunk = (int)(((d->K + want - 1) / want + BK - 1) / BK * BK);
                splitk = (d->K + kchunk - 1) / kchunk;
            }
        }
        // K3 (bit 1 of the k3 option): the same problems as K2 below, 256 x 128 tiles in 256-thread workgroups, two per CU
        // (VPU_GEMM_K3_FORMS: a bit mask of single forms -- 1 bias, 2 bias+residual, 4 bias+GELU, 8 plain dgrad, 16 x aux -- for A/B runs)
#ifdef VPU_LAB
        static const int forms3 = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K3_FORMS"); return e ? atoi(e) : 0; }();
        if (((k3_opt() & 2) || forms3) && !big && d->batch == 1 && !d->colsum && vec && d->N % 8 == 0 && d->K % K3_BK == 0 && d->K >= 256 &&
            !d->transA && d->alpha == 1.0f && (int64_t)d->M * d->ldc * 2 < 0x7FFFFFF0LL &&
            (int64_t)d->M * (d->ldr > d->ldaux ? d->ldr : d->ldaux) * 2 < 0x7FFFFFF0LL) {
            static const bool noepi3 = [] { const char* e = vpu_lab_getenv("VPU_GEMM_NOEPI"); return e && e[0] == '1'; }();
            static const int rb3_env = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2_RB"); return e ? atoi(e) : 0; }();
            constexpr int F_B = VPU_EPI_BIAS, F_BR = VPU_EPI_BIAS | VPU_EPI_RESID,
                          F_G = VPU_EPI_BIAS | VPU_EPI_GELU | VPU_EPI_SAVE_DGELU, F_M = VPU_EPI_MULAUX;
            const int tn3 = (d->N + 127) / 128;
            const int64_t t256 = (int64_t)((d->M + 255) / 256) * tn3, t224 = (int64_t)((d->M + 223) / 224) * tn3;
            const bool short3 = rb3_env != 8 && (rb3_env == 7 || t224 * 224 < t256 * 256);
            const int64_t tot3 = short3 ? t224 : t256;
            const int cap3 = 2 * cu_count();
            bool done3 = tot3 > cu_count();      // (one workgroup per CU: K2's ping-pong is the faster form)
            static const int stag3 = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K3_STAGGER"); return e ? atoi(e) : 0; }();
            const int vec3 = (noepi3 ? 9 : 1) | (stag3 << 8);
#define VPU_LAUNCH_K3_RB(TA_, TB_, FL_, RB_)                                                                         \
    do {                                                                                                             \
        static VpuDevOnce attr_;                                                                                   \
        auto kern_ = gemm_bf16_k3_kernel<TA_, TB_, FL_, RB_>;                                                         \
        if (auto todo_ = attr_.pending()) {                                                                                                \
            VPU_SET_LDS(K3_LDS, kern_); \
        }                                                                                                            \
        const int tm_ = (d->M + 32 * RB_ - 1) / (32 * RB_);                                                           \
        const int tot_ = tm_ * tn3;                                                                                  \
        NOTE_KERNEL("gemm_bf16_k3_kernel<%d, %d, %d, %d>", TA_, TB_, FL_, RB_);                                        \
        kern_<<<dim3((unsigned)(tot_ < cap3 ? tot_ : cap3)), dim3(256), K3_LDS, s>>>(*d, tm_, tn3, vec3);              \
    } while (0)
#define VPU_LAUNCH_K3(TA_, TB_, FL_) do { if (short3) VPU_LAUNCH_K3_RB(TA_, TB_, FL_, 7); else VPU_LAUNCH_K3_RB(TA_, TB_, FL_, 8); } while (0)
            const bool all3 = (k3_opt() & 2) != 0;
            if (!done3) {}
            else if (key == 0 && f == F_B && (all3 || (forms3 & 1))) VPU_LAUNCH_K3(0, 0, F_B);
            else if (key == 0 && f == F_BR && (all3 || (forms3 & 2))) VPU_LAUNCH_K3(0, 0, F_BR);
            else if (key == 0 && f == F_G && (all3 || (forms3 & 4))) VPU_LAUNCH_K3(0, 0, F_G);
            else if (key == 1 && f == 0 && (all3 || (forms3 & 8))) VPU_LAUNCH_K3(0, 1, 0);
            else if (key == 1 && f == F_M && (all3 || (forms3 & 16))) VPU_LAUNCH_K3(0, 1, F_M);
            else done3 = false;
#undef VPU_LAUNCH_K3
#undef VPU_LAUNCH_K3_RB
            if (done3) return vpu_check_launch("vpu_gemm");
        }
#endif
        // K2: large problems whose 256 x 128 tiles fill the chip and whose epilogue is one of the ViT-block flag sets
        {
            static const bool noepi2 = [] { const char* e = vpu_lab_getenv("VPU_GEMM_NOEPI"); return e && e[0] == '1'; }();
            const int k2 = k2_opt();
            // (VPU_GEMM_K2_MIN_TILES / VPU_GEMM_K2G_MIN_TILES: the smallest tile counts the plain / grouped-compile-time K2
            // forms take -- A/B knobs)
            static const int k2_min_tiles = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2_MIN_TILES"); return e ? atoi(e) : 120; }();   // (round 5: 160 -> 120 -- batch 8, 150 tiles: 745 -> 771 images/s; batch 4, 78-84 tiles, loses below 100)
            const int tm2 = (d->M + K2_BM - 1) / K2_BM, tn2 = (d->N + 127) / 128;
            // round 5: the x 128 form picks its tile height among 256 / 224 / 192 rows (RB = 8 / 7 / 6 row blocks per wave) by rounds
            // of tiles x rows per tile, and 160 rows (RB = 5) where nothing taller reaches the tile count the K2 forms start at
            // (batch 4: 3136 rows x N = 768 is 78 / 84 / 102 tiles of 256 / 224 / 192 rows, 120 of 160).  Same-box A/B
            // (VPU_GEMM_K2_RBMIN, laboratory build): 192 rows allowed: ViT-B bs 8 781.7 -> 788.6 images/s, ViT-L bs 8 369.6 -> 370.2,
            // ViT-H bs 8 161.5 -> 163.1, ViT-H bs 12 level; 160 rows allowed everywhere: ViT-B bs 8 773.8, ViT-L bs 8 364.0 (a
            // shorter tile reads the B panel once more per row of tiles) -- but batch 4 with it 500 -> 513.  ViT-B bs 12
            // (9408 = 42 x 224) keeps 224.
            const int ncu_k2 = cu_count();
            auto k2_cost = [&](int bm, int bn) {
                const int64_t t = (int64_t)((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn);
                return ((t + ncu_k2 - 1) / ncu_k2) * bm;
            };
            int rbn = 8;
            {
                static const int rb_min = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2_RBMIN"); const int v = e ? atoi(e) : 6; return v < 5 ? 5 : (v > 8 ? 8 : v); }();
                int64_t best = k2_cost(256, 128);
                for (int rb = 7; rb >= rb_min; --rb) {
                    const int64_t c = k2_cost(32 * rb, 128);
                    if (c < best) { best = c; rbn = rb; }
                }
                if ((int64_t)((d->M + 32 * rbn - 1) / (32 * rbn)) * tn2 < k2_min_tiles && (int64_t)((d->M + 159) / 160) * tn2 >= k2_min_tiles) rbn = 5;
            }
            const int64_t tiles_narrow = (int64_t)((d->M + 32 * rbn - 1) / (32 * rbn)) * tn2;
            if (k2 > 0 && !big && d->batch == 1 && !d->colsum && vec && d->N % 8 == 0 && d->K % BK == 0 && d->K >= 256 &&
                ((int64_t)tm2 * tn2 >= k2_min_tiles || tiles_narrow >= k2_min_tiles) && !d->transA && d->alpha == 1.0f &&
                (int64_t)d->M * d->ldc * 2 < 0x7FFFFFF0LL && (int64_t)d->M * (d->ldr > d->ldaux ? d->ldr : d->ldaux) * 2 < 0x7FFFFFF0LL) {
                constexpr int F_B = VPU_EPI_BIAS, F_BR = VPU_EPI_BIAS | VPU_EPI_RESID,
                              F_G = VPU_EPI_BIAS | VPU_EPI_GELU | VPU_EPI_SAVE_DGELU, F_M = VPU_EPI_MULAUX;
                const bool wide = d->N % 256 == 0 && (k2 == 3 || (k2 == 2 && (int64_t)tm2 * (d->N / 256) >= 384));
                // measured (tools/gemm_bench.py, GEMM_BENCH_K2=0,1,3, M = 9408): the 256 x 128 form wins for K >= 2304
                // (fc2 45.1 vs 47.6 us, fc1 dgrad 42.1 vs 46.7, qkv dgrad 33.9 vs 36.7) and for many-tile short-K problems
                // (qkv 42.3 vs 49.0); one round of 222 tiles over K = 768 stays with the 128 x 128 kernel (proj 21.9 vs 19.8)
                // (round 4, direct epilogue: the one-round K = 768 problems now win too -- proj 18.1 vs 19.6 us, its dgrad 15.4 vs 16.6)
                static const bool direct_env = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2_DIRECT"); return !e || e[0] != '0'; }();
                // (round 5: the short-K forms from 120 tiles of the chosen height on -- batch 4 / 6 / 8: +0.8-1.0 %; was 200)
                static const int narrow_min = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2_NARROW_MIN"); return e ? atoi(e) : 120; }();
                const bool narrow_ok = k2 == 1 || k2 == 3 || d->K >= 1024 || (int64_t)tm2 * tn2 >= 400 ||
                                       (direct_env && d->K >= 512 && ((int64_t)tm2 * tn2 >= 200 || tiles_narrow >= narrow_min));
                const int vec2 = noepi2 ? 9 : 1;
                const int ncu = cu_count();
                bool done = true;
                // tile height: 256 rows, or 224 when that needs less time per CU (rounds of tiles x rows per tile): M = 9408
                // is 36.75 x 256 -- 222 / 666 / 444 tiles, 87 % of the CUs busy in the last round -- but 42 x 224 exactly
                // (252 / 756 / 504 tiles).  VPU_GEMM_K2_RB=8 keeps 256 (A/B runs).
                static const int rb_env = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2_RB"); return e ? atoi(e) : 0; }();
                auto cost = [&](int bm, int bn) {
                    const int64_t tiles = (int64_t)((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn);
                    return ((tiles + ncu - 1) / ncu) * bm;
                };
                const int bn_sel = wide ? 256 : 128;
                const bool short_tile = rb_env != 8 && (rb_env == 7 || cost(224, bn_sel) < cost(256, bn_sel));
                // VPU_GEMM_K2_DIRECT: 0 = LDS-transposed epilogue, 1 (default) = direct epilogue (swapped MFMA operands, stores
                // from the registers)
                static const int direct2 = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2_DIRECT"); return e ? atoi(e) : 1; }();
#define VPU_LAUNCH_K2_SW(TA_, TB_, WN_, FL_, RB_, SW_)                                                               \
    do {                                                                                                             \
        static VpuDevOnce attr_;                                                                                   \
        auto kern_ = gemm_bf16_k2_kernel<TA_, TB_, WN_, FL_, RB_, SW_>;                                               \
        if (auto todo_ = attr_.pending()) {                                                                                                \
            VPU_SET_LDS(K2Cfg<WN_>::LDS + K2_BIAS_LDS, kern_); \
        }                                                                                                            \
        const int tn_ = (d->N + K2Cfg<WN_>::BN_ - 1) / K2Cfg<WN_>::BN_;                                               \
        const int tm_ = (d->M + 32 * RB_ - 1) / (32 * RB_);                                                           \
        const int tot_ = tm_ * tn_;                                                                                  \
        NOTE_KERNEL("gemm_bf16_k2_kernel<%d, %d, %d, %d, %d, %d>", TA_, TB_, WN_, FL_, RB_, SW_);                      \
        kern_<<<dim3((unsigned)(tot_ < ncu ? tot_ : ncu)), dim3(512), K2Cfg<WN_>::LDS + K2_BIAS_LDS, s>>>(*d, tm_, tn_, vec2 VPU_DBG_LOAD); \
    } while (0)
#ifdef VPU_LAB
#define VPU_LAUNCH_K2_RB(TA_, TB_, WN_, FL_, RB_)                                                                     \
    do {                                                                                                             \
        if (direct2 >= 1) VPU_LAUNCH_K2_SW(TA_, TB_, WN_, FL_, RB_, 1);                                               \
        else VPU_LAUNCH_K2_SW(TA_, TB_, WN_, FL_, RB_, 0);                                                            \
    } while (0)
#else
#define VPU_LAUNCH_K2_RB(TA_, TB_, WN_, FL_, RB_) do { (void)direct2; VPU_LAUNCH_K2_SW(TA_, TB_, WN_, FL_, RB_, 1); } while (0)
#endif
#define VPU_LAUNCH_K2(TA_, TB_, WN_, FL_) do { if (short_tile) VPU_LAUNCH_K2_RB(TA_, TB_, WN_, FL_, 7); else VPU_LAUNCH_K2_RB(TA_, TB_, WN_, FL_, 8); } while (0)
#define VPU_LAUNCH_K2N(TA_, TB_, FL_)                                                                                   \
    do {                                                                                                             \
        const int rb_ = rb_env == 7 || rb_env == 8 ? rb_env : rbn;                                                    \
        if (rb_ == 5) VPU_LAUNCH_K2_RB(TA_, TB_, 2, FL_, 5);                                                          \
        else if (rb_ == 6) VPU_LAUNCH_K2_RB(TA_, TB_, 2, FL_, 6);                                                     \
        else if (rb_ == 7) VPU_LAUNCH_K2_RB(TA_, TB_, 2, FL_, 7);                                                     \
        else VPU_LAUNCH_K2_RB(TA_, TB_, 2, FL_, 8);                                                                   \
    } while (0)
#define VPU_K2_BOTH(TA_, TB_, FL_) do { if (wide) VPU_LAUNCH_K2(TA_, TB_, 4, FL_); else if (narrow_ok) VPU_LAUNCH_K2N(TA_, TB_, FL_); else done = false; } while (0)
                // K5 (round 6, gemm_k5.hip): the two 4-wave groups of a workgroup own alternate 128-column tiles, one group's direct
                // epilogue + all LDS-DMA run beside the other group's main loop (no K-half exchange either).  Same-box A/B
                // (tools/gemm_bench.py GEMM_BENCH_K2=2,k5a, M = 9408): qkv 45.6 -> 39.6 us, fc2-dgrad x aux 54.9 -> 51.9, fc2 45.0 ->
                // 43.8, fc1-dgrad 42.8 -> 41.9, proj-dgrad 16.7 -> 16.1, the FPN's 37632 x 384 x 768 35.8 -> 32.2; level or better
                // on every ViT-B / -L / -H shape down to batch 4 -- one tile per workgroup included, so the same family (one
                // accumulation order) serves a batch and its micro-batches.  fc1's bias + GELU + GELU' stays on the 256-column K2
                // form (4.2 vector instructions per MFMA make K5's producer the slower role: 62.1 against 58.4-62.5 us).
                // Default rule of the k2 option only (an explicit k2 = 1 / 3 keeps the K2 forms: tests, A/B runs); k5 = 2: GELU too.
                {
                    const int k5 = vpu_k5_option();
                    if (k5 > 0 && (k2 == 2 || k5 == 2) && d->K >= 9 * BK && !((f & VPU_EPI_RESID) && d->resid_period > 0)) {
                        int rb5 = 8;
                        int64_t best5 = cost(256, 128);
                        for (int rb = 7; rb >= 6; --rb) {
                            const int64_t c5 = cost(32 * rb, 128);
                            if (c5 < best5) { best5 = c5; rb5 = rb; }
                        }
                        // (GELU: up to 6272 rows K5 wins -- 3136 rows 34.3 -> 30.0 us, ViT-L's 6272 x 4096 x 1024 74.9 -> 71.1 --, at 9408 rows
                        // it is level with the 256-column K2 form, at ViT-H's 12288 rows 3.6 % behind)
                        const bool gelu5 = (f & VPU_EPI_GELU) != 0 && d->M > 6272;
                        if (k5 == 2 || !gelu5) {
                            const int r5 = vpu_k5_launch(d, rb5, ncu, vec2, stream, g_last_kernel, sizeof(g_last_kernel));
                            if (r5 < 0) return r5;
                            if (r5 == 1) return vpu_check_launch("vpu_gemm");
                        }
                    }
                }
                if (key == 0 && f == F_B) VPU_K2_BOTH(0, 0, F_B);
                else if (key == 0 && f == F_BR) VPU_K2_BOTH(0, 0, F_BR);
                else if (key == 0 && f == F_G) VPU_K2_BOTH(0, 0, F_G);
                else if (key == 0 && f == 0) VPU_K2_BOTH(0, 0, 0);      // (round 4: the head's bias-free fusion / FPN convolutions)
                else if (key == 1 && f == 0) VPU_K2_BOTH(0, 1, 0);
                else if (key == 1 && f == F_M) VPU_K2_BOTH(0, 1, F_M);
                else done = false;
#undef VPU_K2_BOTH
#undef VPU_LAUNCH_K2N
#undef VPU_LAUNCH_K2
#undef VPU_LAUNCH_K2_RB
#undef VPU_LAUNCH_K2_SW
                if (done) return vpu_check_launch("vpu_gemm");
            }
        }
        // skinny problems (see gemm_bf16_skinny_kernel): few rows, moderate N and K, plain or K-major B
        static const int skinny_env = [] { const char* e = vpu_lab_getenv("VPU_GEMM_SKINNY"); return e ? atoi(e) : 1; }();
        const int skinny_opt = g_opt_skinny.load(std::memory_order_relaxed) >= 0 ? g_opt_skinny.load(std::memory_order_relaxed) : skinny_env;
        static const int skinny_m = [] { const char* e = vpu_lab_getenv("VPU_GEMM_SKINNY_M"); return e ? atoi(e) : 4096; }();   // (round 5: 2560 -> 4096, batch 4's 3136-row N = 768 GEMMs: 495 -> 502 images/s)
        static const int skinny_n = [] { const char* e = vpu_lab_getenv("VPU_GEMM_SKINNY_N"); return e ? atoi(e) : 4096; }();
        static const int skinny_k = [] { const char* e = vpu_lab_getenv("VPU_GEMM_SKINNY_K"); return e ? atoi(e) : 4096; }();
        // (under-filled launches only: fewer than 192 tiles of 128x128, the same bound as the split-K rule below)
        if (skinny_opt && !big && !d->transA && d->batch == 1 && !d->colsum && d->M <= skinny_m && d->N <= skinny_n &&
            d->K >= 64 && d->K <= skinny_k && (int64_t)tiles_m * tiles_n < 192) {
            const int tn64 = (d->N + SK_T - 1) / SK_T, tm64 = (d->M + SK_T - 1) / SK_T;
            const int kw = (int)(((d->K + 3) / 4 + 63) / 64 * 64);
            static VpuDevOnce attr_sk;
            if (auto todo_ = attr_sk.pending()) {
                VPU_SET_LDS(4 * TILE_BYTES, gemm_bf16_skinny_kernel<0>);
                VPU_SET_LDS(8 * TILE_BYTES, gemm_bf16_skinny_kernel<1>);
            }
            dim3 sgrid((unsigned)(tm64 * tn64)), sblock(256);
            NOTE_KERNEL("gemm_bf16_skinny_kernel<%d>", d->transB ? 1 : 0);
            if (d->transB) gemm_bf16_skinny_kernel<1><<<sgrid, sblock, 8 * TILE_BYTES, s>>>(*d, tn64, kw, vec ? 1 : 0);
            else gemm_bf16_skinny_kernel<0><<<sgrid, sblock, 4 * TILE_BYTES, s>>>(*d, tn64, kw, vec ? 1 : 0);
            return vpu_check_launch("vpu_gemm");
        }
        static const bool noepi = [] { const char* e = vpu_lab_getenv("VPU_GEMM_NOEPI"); return e && e[0] == '1'; }();
        static const bool nostore = [] { const char* e = vpu_lab_getenv("VPU_GEMM_NOSTORE"); return e && e[0] == '1'; }();
        static const int stagger = [] { const char* e = vpu_lab_getenv("VPU_GEMM_STAGGER"); return e ? atoi(e) : 0; }();
        const int vec_arg = (noepi ? 9 : (nostore ? 8 : (vec ? 1 : 0))) | (stagger << 8);
        float* ws = splitk > 1 ? reinterpret_cast<float*>(d->workspace) : nullptr;
        dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)splitk, (unsigned)d->batch), block(256);
        // persistent launch of the 128x128 kernel: at most VPU_GEMM_PERSIST (default 2 per CU = 512) workgroups walk the
        // (tile, split-K slice, batch) list
        const int64_t total_work = (int64_t)tiles_m * tiles_n * splitk * d->batch;
        static const int persist_env = [] { const char* e = vpu_lab_getenv("VPU_GEMM_PERSIST"); return e ? atoi(e) : 0; }();
        const int persist_cap = persist_env > 0 ? persist_env : 2 * cu_count();
        dim3 pgrid((unsigned)(total_work < persist_cap ? total_work : persist_cap), 1, 1);
        // staging: LDS-DMA + two stages (fragment reads -> DMA of the next tile -> MFMAs) is the default; it beats register
        // staging on every ViT-B shape once the DMA is no longer drained by the compiler's vmcnt(0) (tools/gemm_bench.py:
        // qkv fwd 599 vs 503, fc2 fwd 803 vs 692, wgrad 576 vs 347 TFLOP/s).  VPU_GEMM_DMA=0 selects register staging.
        static const int force_dma = [] { const char* e = vpu_lab_getenv("VPU_GEMM_DMA"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
        const bool use_dma = force_dma != 0;
#define VPU_LAUNCH(TA_, TB_)                                                                                         \
    do {                                                                                                             \
        NOTE_KERNEL("gemm_bf16_kernel<%d, %d, %s, false, -1, 0>", TA_, TB_, use_dma ? "true" : "false");              \
        if (use_dma) gemm_bf16_kernel<TA_, TB_, true, false, -1><<<pgrid, block, 4 * TILE_BYTES, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg, tiles_m, d->batch, cnt_arg); \
        else gemm_bf16_kernel<TA_, TB_, false, false, -1><<<pgrid, block, 2 * TILE_BYTES, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg, tiles_m, d->batch, cnt_arg);       \
    } while (0)
#define VPU_LAUNCH_FL(TA_, TB_, FL_)                                                                  \
    do {                                                                                              \
        NOTE_KERNEL("gemm_bf16_kernel<%d, %d, true, false, %d, 0>", TA_, TB_, FL_);                    \
        gemm_bf16_kernel<TA_, TB_, true, false, FL_><<<pgrid, block, 4 * TILE_BYTES, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg, tiles_m, d->batch, cnt_arg); \
    } while (0)
        // compile-time epilogues for the flag sets of the ViT blocks (engine.py: linear / mlp / _dgrad)
        static const bool no_spec = [] { const char* e = vpu_lab_getenv("VPU_GEMM_GENERIC"); return e && e[0] == '1'; }();
        const bool spec_ok = use_dma && !no_spec && !big && !d->colsum && vec && splitk == 1 && d->N % 8 == 0 && (vec_arg & 255) == 1;
        bool launched = false, inlaunch = false;
        unsigned* cnt_arg = nullptr;
#define VPU_LAUNCH_RING(TA_, TB_, CS_, FL_)                                                                            \
    do {                                                                                                             \
        static VpuDevOnce attr_;                                                                                   \
        auto kern_ = gemm_bf16_kernel<TA_, TB_, true, CS_, FL_, 3>;                                                   \
        if (auto todo_ = attr_.pending()) {                                                                                                \
            VPU_SET_LDS(6 * TILE_BYTES, kern_); \
        }                                                                                                            \
        NOTE_KERNEL("gemm_bf16_kernel<%d, %d, true, %s, %d, 3>", TA_, TB_, CS_ ? "true" : "false", FL_);              \
        kern_<<<pgrid, block, 6 * TILE_BYTES, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg, tiles_m, d->batch, cnt_arg);       \
    } while (0)
#ifdef VPU_LAB
        if (ring && use_dma && vec_arg <= 1) {
            launched = true;
            if (d->colsum && key == 3) VPU_LAUNCH_RING(1, 1, true, -1);
            else if (d->colsum) launched = false;
            else if (key == 0) VPU_LAUNCH_RING(0, 0, false, -1);
            else if (key == 1) VPU_LAUNCH_RING(0, 1, false, -1);
            else if (key == 2) VPU_LAUNCH_RING(1, 0, false, -1);
            else VPU_LAUNCH_RING(1, 1, false, -1);
        }
#endif
#undef VPU_LAUNCH_RING
        // K3S (bit 2 of the k3 option): the same flag sets on the 128 x 128 ring kernel, three workgroups per CU
        // (measured against the two-stage kernel, tools/gemm_bench.py GEMM_BENCH_K2=2,k3s: it wins where K is short -- the FPN /
        // head maps: 150528 x 128 x 192 23.3 -> 18.5 us, 150528 x 256 x 128 35.4 -> 27.0, 37632 x 256 x 256 16.1 -> 12.8, the
        // fusion dgrad 57.0 -> 46.8 -- and loses 3-10 % at K = 768 (proj 20.5 -> 21.8): K <= 384 only; "k3" option bit 5 lifts
        // the bound for tests)
        if (!launched && spec_ok && (k3_opt() & 4) && (d->K <= 384 || (k3_opt() & 32)) && d->batch == 1 && d->K % K3_BK == 0 && d->K >= 128 && d->alpha == 1.0f &&
            (int64_t)d->M * d->ldc * 2 < 0x7FFFFFF0LL && (int64_t)d->M * (d->ldr > d->ldaux ? d->ldr : d->ldaux) * 2 < 0x7FFFFFF0LL) {
            constexpr int F_B = VPU_EPI_BIAS, F_BR = VPU_EPI_BIAS | VPU_EPI_RESID, F_BL = VPU_EPI_BIAS | VPU_EPI_RELU,
                          F_G = VPU_EPI_BIAS | VPU_EPI_GELU | VPU_EPI_SAVE_DGELU, F_M = VPU_EPI_MULAUX, F_D = VPU_EPI_DRELU;
            const int cap3s = 3 * cu_count();
            const int tot3s = tiles_m * tiles_n;
            launched = true;
            // a bf16 output accumulated in place (a second gradient into the same activation gradient: the neck's 384 -> 768
            // projections, 9408 x 768 x 384, seven per step on the two-stage kernel's run-time epilogue, 20.6 us) IS the
            // residual form with the residual = C: same arithmetic (fp32 sum of the old bf16 value and the accumulator, one
            // rounding), every element read and written by the same lane, the reads requested before the first store
            vpu_gemm_desc dacc = *d;
            const bool acc_as_res = key == 1 && f == VPU_EPI_ACCUM && d->dtype == VPU_BF16 && !d->resid;
            if (acc_as_res) {
                dacc.flags = VPU_EPI_RESID; dacc.resid = d->C; dacc.ldr = d->ldc; dacc.resid_period = 0;
                dacc.sRo = d->sCo; dacc.sRi = d->sCi;
            }
            const vpu_gemm_desc* dk = acc_as_res ? &dacc : d;
#define VPU_LAUNCH_K3S(TA_, TB_, FL_)                                                                                \
    do {                                                                                                             \
        static VpuDevOnce attr_;                                                                                   \
        auto kern_ = gemm_bf16_k3s_kernel<TA_, TB_, FL_>;                                                             \
        if (auto todo_ = attr_.pending()) {                                                                                                \
            VPU_SET_LDS((K3Cfg<2, 128>::LDS), kern_); \
        }                                                                                                            \
        NOTE_KERNEL("gemm_bf16_k3s_kernel<%d, %d, %d>", TA_, TB_, FL_);                                                \
        kern_<<<dim3((unsigned)(tot3s < cap3s ? tot3s : cap3s)), dim3(256), K3Cfg<2, 128>::LDS, s>>>(*dk, tiles_m, tiles_n, 1); \
    } while (0)
            if (key == 0 && f == F_B) VPU_LAUNCH_K3S(0, 0, F_B);
            else if (key == 0 && f == F_BR) VPU_LAUNCH_K3S(0, 0, F_BR);
            else if (key == 0 && f == F_G) VPU_LAUNCH_K3S(0, 0, F_G);
            else if (key == 0 && f == F_BL) VPU_LAUNCH_K3S(0, 0, F_BL);
            else if (key == 0 && f == 0) VPU_LAUNCH_K3S(0, 0, 0);
            else if (key == 1 && f == F_D) VPU_LAUNCH_K3S(0, 1, F_D);
            else if (key == 1 && f == 0) VPU_LAUNCH_K3S(0, 1, 0);
            else if (key == 1 && f == F_M) VPU_LAUNCH_K3S(0, 1, F_M);
            else if (acc_as_res) VPU_LAUNCH_K3S(0, 1, VPU_EPI_RESID);
            else launched = false;
#undef VPU_LAUNCH_K3S
        }
        if (!launched && spec_ok) {
            launched = true;
            constexpr int F_B = VPU_EPI_BIAS, F_BR = VPU_EPI_BIAS | VPU_EPI_RESID,
                          F_G = VPU_EPI_BIAS | VPU_EPI_GELU | VPU_EPI_SAVE_DGELU, F_M = VPU_EPI_MULAUX;
            if (key == 0 && f == F_B) VPU_LAUNCH_FL(0, 0, F_B);
            else if (key == 0 && f == F_BR) VPU_LAUNCH_FL(0, 0, F_BR);
            else if (key == 0 && f == F_G) VPU_LAUNCH_FL(0, 0, F_G);
            else if (key == 0 && f == (VPU_EPI_BIAS | VPU_EPI_RELU)) VPU_LAUNCH_FL(0, 0, VPU_EPI_BIAS | VPU_EPI_RELU);
            else if (key == 0 && f == 0) VPU_LAUNCH_FL(0, 0, 0);
            else if (key == 1 && f == VPU_EPI_DRELU) VPU_LAUNCH_FL(0, 1, VPU_EPI_DRELU);
            else if (key == 1 && f == 0) VPU_LAUNCH_FL(0, 1, 0);
            else if (key == 1 && f == F_M) VPU_LAUNCH_FL(0, 1, F_M);
            else if (key == 1 && f == VPU_EPI_ACCUM) VPU_LAUNCH_FL(0, 1, VPU_EPI_ACCUM);
            else launched = false;
        }
        const bool slab_ok = use_dma && !no_spec && !big && splitk > 1 && d->N % 8 == 0 && vec_arg <= 1;
        if (!launched && slab_ok) {
            launched = true;
            // in-launch combine: tile counters in the last CNT_BYTES of the workspace (zero at first use, left zero)
            const int inl = g_opt_inlaunch.load(std::memory_order_relaxed) >= 0 ? g_opt_inlaunch.load(std::memory_order_relaxed) : inlaunch_env0();
            const int64_t slab_bytes = (int64_t)d->batch * splitk * d->M * d->N * 4;
            if (inl && (inl == 1 || slab_bytes <= ((int64_t)inl << 20)) && tiles <= CNT_BYTES / 4 &&
                (int64_t)d->M * d->N * 4 < 0x7FFFFFF0LL && d->workspace_bytes > 2 * CNT_BYTES) {
                cnt_arg = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(d->workspace) + d->workspace_bytes - CNT_BYTES);
                inlaunch = true;
            }
            if (d->colsum && key == 3) NOTE_KERNEL("gemm_bf16_kernel<1, 1, true, true, 65536, 0>");
            if (d->colsum && key == 3) gemm_bf16_kernel<1, 1, true, true, FL_SLAB><<<pgrid, block, 4 * TILE_BYTES, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg, tiles_m, d->batch, cnt_arg);
            else if (d->colsum) launched = false;
            else if (key == 3) VPU_LAUNCH_FL(1, 1, FL_SLAB);
            else if (key == 0) VPU_LAUNCH_FL(0, 0, FL_SLAB);
            else if (key == 1) VPU_LAUNCH_FL(0, 1, FL_SLAB);
            else launched = false;
        }
        if (launched) {
        } else
#ifdef VPU_LAB
        if (big) {
            static VpuDevOnce attr_done;
            if (auto todo_ = attr_done.pending()) {
                const int sz = 3 * STAGE2;
                VPU_SET_LDS(sz, gemm_bf16_big_kernel<0, 0>);
                VPU_SET_LDS(sz, gemm_bf16_big_kernel<0, 1>);
                VPU_SET_LDS(sz, gemm_bf16_big_kernel<1, 0>);
                VPU_SET_LDS(sz, gemm_bf16_big_kernel<1, 1>);
            }
            dim3 block2(512);
            NOTE_KERNEL("gemm_bf16_big_kernel<%d, %d>", key >> 1, key & 1);
            switch (key) {
                case 0: gemm_bf16_big_kernel<0, 0><<<grid, block2, 3 * STAGE2, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg); break;
                case 1: gemm_bf16_big_kernel<0, 1><<<grid, block2, 3 * STAGE2, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg); break;
                case 2: gemm_bf16_big_kernel<1, 0><<<grid, block2, 3 * STAGE2, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg); break;
                default: gemm_bf16_big_kernel<1, 1><<<grid, block2, 3 * STAGE2, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg); break;
            }
        } else
#endif
        {
            if (d->colsum) {  // weight-gradient GEMM with the fused bias gradient (always transA = transB = 1 in the engine)
                NOTE_KERNEL("gemm_bf16_kernel<1, %d, true, true, -1, 0>", key == 3 ? 1 : 0);
                if (key == 3) gemm_bf16_kernel<1, 1, true, true, -1><<<pgrid, block, 4 * TILE_BYTES, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg, tiles_m, d->batch, cnt_arg);
                else gemm_bf16_kernel<1, 0, true, true, -1><<<pgrid, block, 4 * TILE_BYTES, s>>>(*d, tiles_n, splitk, kchunk, ws, vec_arg, tiles_m, d->batch, cnt_arg);
            } else
            switch (key) {
                case 0: VPU_LAUNCH(0, 0); break;
                case 1: VPU_LAUNCH(0, 1); break;
                case 2: VPU_LAUNCH(1, 0); break;
                default: VPU_LAUNCH(1, 1); break;
            }
        }
#undef VPU_LAUNCH
#undef VPU_LAUNCH_FL
        if (splitk > 1 && !inlaunch) {
            const int64_t mn = (int64_t)d->M * d->N;
            int vec8 = vec && d->N % 8 == 0 ? 1 : 0;
            if (vec8 && splitk >= 16 && mn / 8 <= 8 * 65535) vec8 = 2;   // slice-parallel form
            dim3 rgrid((unsigned)(vec8 == 2 ? (mn / 8 + 7) / 8 : vpu_grid_for(vec8 ? mn / 8 : mn, 256, 4096)), 1,
                       (unsigned)d->batch);
            splitk_reduce_kernel<bf16_t><<<rgrid, block, 0, s>>>(*d, splitk, ws, vec8);
        }
    } else {
        dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)d->batch), block(256);
        NOTE_KERNEL("gemm_f32_kernel<%d, %d>", key >> 1, key & 1);
        switch (key) {
            case 0: gemm_f32_kernel<0, 0><<<grid, block, 0, s>>>(*d, tiles_n); break;
            case 1: gemm_f32_kernel<0, 1><<<grid, block, 0, s>>>(*d, tiles_n); break;
            case 2: gemm_f32_kernel<1, 0><<<grid, block, 0, s>>>(*d, tiles_n); break;
            default: gemm_f32_kernel<1, 1><<<grid, block, 0, s>>>(*d, tiles_n); break;
        }
    }
    return vpu_check_launch("vpu_gemm");
}

extern "C" const char* vpu_gemm_last_kernel(void) { return g_last_kernel; }

extern "C" int vpu_gemm_set_option(const char* name, int32_t value) {
    vpu_clear_stale_error();
#ifndef VPU_LAB
    if (name && ((!strcmp(name, "ring") && value > 0) || (!strcmp(name, "k3") && value > 0 && (value & 3)))) {
        vpu_set_error("vpu_gemm_set_option: the three-stage ring forms and the K3 forward / grouped forms (k3 bits 0-1) are compiled "
                      "into the laboratory library only (build.sh diag, VPU_LIB_DIAG=1)");
        return VPU_ERR_ARG;
    }
#endif
    if (name && !strcmp(name, "ring") && value >= -1 && value <= 2) {
        g_opt_ring.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    if (name && !strcmp(name, "skinny") && value >= -1 && value <= 1) {
        g_opt_skinny.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    if (name && !strcmp(name, "k2") && value >= -1 && value <= 3) {
        g_opt_k2.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    if (name && !strcmp(name, "k3") && value >= -1 && value <= 63) {
        g_opt_k3.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    if (name && !strcmp(name, "k5") && value >= -1 && value <= 2) {
        vpu_k5_set_option(value);
        return VPU_OK;
    }
    if (name && !strcmp(name, "k5_split") && value >= -1 && value <= 1) {   // LDS-DMA issue of K5 shared by both wave groups: -1 by flag set
        vpu_k5_set_split(value);
        return VPU_OK;
    }
    if (name && !strcmp(name, "k5_noepi") && value >= 0 && value <= 1) {   // diagnostic: K5 main loops only, nothing stored
        vpu_k5_set_noepi(value);
        return VPU_OK;
    }
    if (name && !strcmp(name, "k5_grid") && value >= 0 && value <= 1024) {   // cap on the K5 grid (tests: many tiles per workgroup)
        vpu_k5_set_grid(value);
        return VPU_OK;
    }
    if (name && !strcmp(name, "skinny_group") && value >= -1 && value <= 2) {
        g_opt_skinny_group.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    if (name && !strcmp(name, "reserve_cus") && value >= 0 && value <= 128) {
        g_opt_reserve.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    if (name && !strcmp(name, "splitk_inlaunch") && value >= -1 && value <= 4096) {
        g_opt_inlaunch.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    vpu_set_error("vpu_gemm_set_option: known options: ring (-1 environment default, 0 off, 1 one-wave problems, 2 everywhere), "
                  "splitk_inlaunch (-1 environment default, 0 separate reduce launch, 1 last-arriver combine), "
                  "skinny (-1 environment default, 0 off, 1 on), "
                  "k2 (-1 environment default, 0 off, 1 256x128 tiles, 2 + 256x256 where it fills the chip, 3 256x256 wherever legal), "
                  "k3 (-1 environment default, bit 0 grouped weight gradients, bit 1 forward / dgrad, bit 2 grouped forward / dgrad), "
                  "k5 (-1 environment default, 0 off, 1 wherever legal except bias + GELU, 2 that too), k5_grid (0 = none), "
                  "reserve_cus (0..128 CUs the persistent launches leave unclaimed)");
    return VPU_ERR_ARG;
}

// the CUs a persistent grid may claim right now (all of them, minus the "reserve_cus" a data-parallel step keeps for its collectives):
// the attention launcher sizes its persistent window backward with it (vpu_common.h)
int vpu_cu_budget() { return cu_count(); }

extern "C" int vpu_gemm_get_option(const char* name, int32_t* value) {
    vpu_clear_stale_error();
    if (!name || !value) { vpu_set_error("vpu_gemm_get_option: null argument"); return VPU_ERR_ARG; }
    if (!strcmp(name, "k2")) { *value = k2_opt(); return VPU_OK; }
    if (!strcmp(name, "k3")) { *value = k3_opt(); return VPU_OK; }
    if (!strcmp(name, "k5")) { *value = vpu_k5_option(); return VPU_OK; }
    vpu_set_error("vpu_gemm_get_option: known options: k2, k3, k5");
    return VPU_ERR_ARG;
}

extern "C" int vpu_gemm_grouped(const vpu_gemm_desc* descs, int32_t n, void* stream) {
    vpu_clear_stale_error();
    if (!descs || n < 1 || n > VPU_GEMM_GROUP_MAX) { vpu_set_error("vpu_gemm_grouped: 1 <= n <= VPU_GEMM_GROUP_MAX"); return VPU_ERR_ARG; }
    vpu_gemm_group ga;
    ga.n = n;
    int total = 0;
    bool vec = true, any_batch = false;
    const int key = (descs[0].transA ? 2 : 0) | (descs[0].transB ? 1 : 0);
    for (int i = 0; i < n; ++i) {
        const vpu_gemm_desc* d = descs + i;
        if (!d->A || !d->B || !d->C || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch < 1 || d->inner != 1 ||
            d->dtype != VPU_BF16 || ((d->transA ? 2 : 0) | (d->transB ? 1 : 0)) != key) {
            vpu_set_error("vpu_gemm_grouped: every problem bf16, inner 1, non-null operands, the same transA / transB");
            return VPU_ERR_ARG;
        }
        if (d->batch > 1) {
            // batch entries = the reduction slices of one weight gradient (entry z: A + z sAo, B + z sBo, C + z sCo, colsum +
            // z M): only the K3 weight-gradient form walks them
            any_batch = true;
            if (key != 3 || d->sAo % 8 || d->sBo % 8 || d->sCo % 8 || (int64_t)d->batch * d->M * d->N >= 0x7FFFFFFFLL) {
                vpu_set_error("vpu_gemm_grouped: batch > 1 needs transA = transB = 1 and batch strides that are multiples of 8 elements");
                return VPU_ERR_ARG;
            }
        }
        const int f = d->flags;
        if (((f & VPU_EPI_BIAS) && !d->bias) || ((f & VPU_EPI_RESID) && !d->resid) ||
            ((f & (VPU_EPI_DGELU | VPU_EPI_DRELU | VPU_EPI_MULAUX)) && !d->aux) ||
            ((f & (VPU_EPI_PREACT | VPU_EPI_SAVE_DGELU)) && !d->preact) ||
            ((f & VPU_EPI_SAVE_DGELU) && (!(f & VPU_EPI_GELU) || (f & VPU_EPI_PREACT))) || (d->colsum && key != 3)) {
            vpu_set_error("vpu_gemm_grouped: epilogue flag set but its pointer is null (colsum: transA = transB = 1 only)");
            return VPU_ERR_ARG;
        }
        if (d->lda % 8 || d->ldb % 8 || !aligned_to(d->A, 16) || !aligned_to(d->B, 16)) {
            vpu_set_error("vpu_gemm_grouped: A/B base and leading dimensions must be 16-byte multiples");
            return VPU_ERR_ALIGN;
        }
        const int64_t ea = d->transA ? (int64_t)d->K * d->lda : (int64_t)d->M * d->lda;
        const int64_t eb = d->transB ? (int64_t)d->K * d->ldb : (int64_t)d->N * d->ldb;
        if (ea * 2 >= 0x7FFFFFF0LL || eb * 2 >= 0x7FFFFFF0LL) {
            vpu_set_error("vpu_gemm_grouped: an operand spans more than 2 GiB");
            return VPU_ERR_ARG;
        }
        const size_t cbytes = (f & VPU_EPI_OUT_F32) ? 32 : 16;
        bool v = d->ldc % 8 == 0 && aligned_to(d->C, cbytes);
        if (f & VPU_EPI_BIAS) v = v && aligned_to(d->bias, 32);
        if (f & (VPU_EPI_PREACT | VPU_EPI_SAVE_DGELU)) v = v && aligned_to(d->preact, 16);
        if (f & (VPU_EPI_DGELU | VPU_EPI_DRELU | VPU_EPI_MULAUX)) v = v && d->ldaux % 8 == 0 && aligned_to(d->aux, 16);
        if (f & VPU_EPI_RESID) v = v && d->ldr % 8 == 0 && aligned_to(d->resid, 16);
        vec = vec && v;
        ga.start[i] = total;
        ga.d[i] = *d;
        const int64_t t = (int64_t)((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
        if (t + total > 0x3FFFFFFF) { vpu_set_error("vpu_gemm_grouped: too many tiles"); return VPU_ERR_ARG; }
        total += (int)t;
    }
    for (int i = n; i <= VPU_GEMM_GROUP_MAX; ++i) ga.start[i] = total;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // skinny form: every problem few rows, moderate N / K, plain or K-major B (the single-launch rule of vpu_gemm)
    // skinny_group option / VPU_GEMM_SKINNY_GROUP: 0 off, 1 (default) forward / dgrad orientations, 2 also the weight-gradient
    // orientation (measured: 852 / 855 / 848 images/s for 0 / 1 / 2 on one box -- for the 576-row weight gradients eight
    // problems of 128 x 128 tiles in the general grouped kernel beat 64 x 64 tiles with a four-way K split)
    static const int sk_grp_env0 = [] { const char* e = vpu_lab_getenv("VPU_GEMM_SKINNY_GROUP"); return e ? atoi(e) : 1; }();
    const int sk_grp_env = g_opt_skinny_group.load(std::memory_order_relaxed) >= 0 ? g_opt_skinny_group.load(std::memory_order_relaxed) : sk_grp_env0;
    if (!any_batch && sk_grp_env && (key <= 1 || (key == 3 && sk_grp_env >= 2)) && n >= 2) {
        bool ok = true;
        int total64 = 0;
        vpu_gemm_group g3 = ga;
        for (int i = 0; i < n; ++i) {
            const vpu_gemm_desc* d = descs + i;
            ok = ok && d->M <= 2560 && d->N <= 4096 && d->K >= 64 && d->K <= 4096 && d->alpha == 1.0f &&
                 (int64_t)((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN) < 192;
            g3.start[i] = total64;
            total64 += ((d->M + SK_T - 1) / SK_T) * ((d->N + SK_T - 1) / SK_T);
        }
        for (int i = n; i <= VPU_GEMM_GROUP_MAX; ++i) g3.start[i] = total64;
        if (ok && total64 <= 2048) {
            static VpuDevOnce attr_skg;
            if (auto todo_ = attr_skg.pending()) {
                VPU_SET_LDS(4 * TILE_BYTES, gemm_bf16_skinny_grouped_kernel<0, 0>);
                VPU_SET_LDS(8 * TILE_BYTES, gemm_bf16_skinny_grouped_kernel<0, 1>);
                VPU_SET_LDS(8 * TILE_BYTES, gemm_bf16_skinny_grouped_kernel<1, 1>);
            }
            NOTE_KERNEL("gemm_bf16_skinny_grouped_kernel<%d, %d>", key >> 1, key & 1);
            if (key == 3) gemm_bf16_skinny_grouped_kernel<1, 1><<<dim3((unsigned)total64), dim3(256), 8 * TILE_BYTES, s>>>(g3, vec ? 1 : 0);
            else if (key == 1) gemm_bf16_skinny_grouped_kernel<0, 1><<<dim3((unsigned)total64), dim3(256), 8 * TILE_BYTES, s>>>(g3, vec ? 1 : 0);
            else gemm_bf16_skinny_grouped_kernel<0, 0><<<dim3((unsigned)total64), dim3(256), 4 * TILE_BYTES, s>>>(g3, vec ? 1 : 0);
            return vpu_check_launch("vpu_gemm_grouped");
        }
    }
    // K2 form for forward / dgrad groups whose 256 x 128 tiles fill most of the chip (the DMA neck's image-side K / V
    // projections: two 9408 x 384 x 768 problems = 222 tiles; as 128 x 128 tiles in the general grouped kernel 28 us)
    static const int k2g_fwd = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2G_FWD"); return e ? atoi(e) : 1; }();
    if (k2g_fwd && key <= 1 && k2_opt() > 0 && vec) {
        bool ok = true;
        int total2 = 0;
        vpu_gemm_group g2 = ga;
        for (int i = 0; i < n; ++i) {
            const vpu_gemm_desc* d = descs + i;
            ok = ok && d->K % BK == 0 && d->K >= 512 && d->N % 8 == 0 && d->alpha == 1.0f && !d->colsum &&
                 (int64_t)d->M * d->ldc * 2 < 0x7FFFFFF0LL;
            g2.start[i] = total2;
            total2 += ((d->M + K2_BM - 1) / K2_BM) * ((d->N + 127) / 128);
        }
        for (int i = n; i <= VPU_GEMM_GROUP_MAX; ++i) g2.start[i] = total2;
        // all problems bias-only (forward) / plain (dgrad), bf16 output: the compile-time form with the direct epilogue
        bool same_fl = ok;
        for (int i = 0; i < n && same_fl; ++i)
            same_fl = descs[i].flags == (key == 0 ? VPU_EPI_BIAS : 0) && (key != 0 || descs[i].bias) && descs[i].dtype == VPU_BF16 &&
                      (int64_t)descs[i].M * descs[i].ldc * 2 < 0x7FFFFFF0LL;
        static const bool k2g_fl = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2G_FL"); return !e || e[0] != '0'; }();
        static const int k2g_min_tiles = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K2G_MIN_TILES"); return e ? atoi(e) : 100; }();   // (192 -> 100: 12.57 -> 12.53 ms, within noise; the run-time form keeps 192)
        if (ok && total2 >= k2g_min_tiles && same_fl && k2g_fl) {
            const int ncu = cu_count();
            static VpuDevOnce attrf0, attrf1;
            if (key == 0) {
                auto kern_ = gemm_bf16_k2_grouped_fl_kernel<0, 0, VPU_EPI_BIAS>;
                if (auto todo_ = attrf0.pending()) { VPU_SET_LDS(K2Cfg<2>::LDS + K2_BIAS_LDS, kern_); }
                NOTE_KERNEL("gemm_bf16_k2_grouped_fl_kernel<0, 0, 1>");
                kern_<<<dim3((unsigned)(total2 < ncu ? total2 : ncu)), dim3(512), K2Cfg<2>::LDS + K2_BIAS_LDS, s>>>(g2, 1);
            } else {
                auto kern_ = gemm_bf16_k2_grouped_fl_kernel<0, 1, 0>;
                if (auto todo_ = attrf1.pending()) { VPU_SET_LDS(K2Cfg<2>::LDS + K2_BIAS_LDS, kern_); }
                NOTE_KERNEL("gemm_bf16_k2_grouped_fl_kernel<0, 1, 0>");
                kern_<<<dim3((unsigned)(total2 < ncu ? total2 : ncu)), dim3(512), K2Cfg<2>::LDS + K2_BIAS_LDS, s>>>(g2, 1);
            }
            return vpu_check_launch("vpu_gemm_grouped");
        }
        if (ok && total2 >= 192) {
            const int ncu = cu_count();
            static VpuDevOnce attr0, attr1;
            if (key == 0) {
                auto kern_ = gemm_bf16_k2_grouped_kernel<0, 0, false>;
                if (auto todo_ = attr0.pending()) { VPU_SET_LDS(K2Cfg<2>::LDS, kern_); }
                NOTE_KERNEL("gemm_bf16_k2_grouped_kernel<0, 0, false>");
                kern_<<<dim3((unsigned)(total2 < ncu ? total2 : ncu)), dim3(512), K2Cfg<2>::LDS, s>>>(g2, 1);
            } else {
                auto kern_ = gemm_bf16_k2_grouped_kernel<0, 1, false>;
                if (auto todo_ = attr1.pending()) { VPU_SET_LDS(K2Cfg<2>::LDS, kern_); }
                NOTE_KERNEL("gemm_bf16_k2_grouped_kernel<0, 1, false>");
                kern_<<<dim3((unsigned)(total2 < ncu ? total2 : ncu)), dim3(512), K2Cfg<2>::LDS, s>>>(g2, 1);
            }
            return vpu_check_launch("vpu_gemm_grouped");
        }
    }
    // K2 form: weight-gradient groups over a long reduction whose 256 x 128 tiles fill most of the chip
    if (key == 3 && k2_opt() > 0 && vec) {
        // (VPU_GEMM_K4_MIN_K: the shortest reduction the 256 x 256-tile weight-gradient kernels take -- A/B knob)
        static const int k4_min_k = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K4_MIN_K"); return e ? atoi(e) : 2048; }();
        bool ok = true;
        int total2 = 0;
        vpu_gemm_group g2 = ga;
        for (int i = 0; i < n; ++i) {
            const vpu_gemm_desc* d = descs + i;
            ok = ok && d->K % BK == 0 && d->K >= k4_min_k && d->N % 8 == 0 && d->M % 8 == 0;
            g2.start[i] = total2;
            total2 += ((d->M + K2_BM - 1) / K2_BM) * ((d->N + 127) / 128) * d->batch;
        }
        for (int i = n; i <= VPU_GEMM_GROUP_MAX; ++i) g2.start[i] = total2;
        // (a tile of this kernel walks its whole K on a CU of its own: fewer than 192 tiles leave CUs idle, but over a very
        // long reduction -- the leftover of a step's packed weight-gradient launches, 147 K-tiles -- 96+ tiles still beat
        // the 128 x 128 grouped kernel's two tiles per CU: 165 vs 236 us)
        bool very_long = true;
        for (int i = 0; i < n; ++i) very_long = very_long && descs[i].K >= 8192;
        // (K3: two free-running workgroups per CU -- measured against K2 on grouped weight gradients over 9408 rows: 512 tiles
        // 246 vs 284 us, 432 tiles 280 vs 299, but 216 tiles -- one workgroup per CU, nobody to overlap with -- 201 vs 149)
        bool any_dcs = false;      // distributed column sums: only the 256 x 256-tile kernels below implement them
        for (int i = 0; i < n; ++i) any_dcs = any_dcs || (descs[i].colsum && descs[i].cs_tn > 1);
        if (any_dcs && !(ok && (k3_opt() & 8))) {
            vpu_set_error("vpu_gemm_grouped: cs_tn > 1 (distributed column sums) needs the K4 weight-gradient path (k3 option bit 8, "
                          "K % 64 == 0, K >= 2048, aligned operands)");
            return VPU_ERR_ARG;
        }
        if (ok && (k3_opt() & 8)) {
            // K4: 256 x 256 tiles, one 512-thread workgroup per CU, five-stage ring
            vpu_gemm_group g4 = ga;
            int total4 = 0;
            for (int i = 0; i < n; ++i) {
                g4.start[i] = total4;
                total4 += ((descs[i].M + 255) / 256) * ((descs[i].N + 255) / 256) * descs[i].batch;
            }
            for (int i = n; i <= VPU_GEMM_GROUP_MAX; ++i) g4.start[i] = total4;
            static VpuDevOnce attr4_;
            const bool pipe = (k3_opt() & 16) != 0;
            // direct fp32 epilogue (swapped MFMA operands): plain weight-gradient descriptors only -- fp32 output with or
            // without accumulation, alpha 1, 16-byte aligned rows; VPU_GEMM_K4_DIRECT=0 keeps the LDS-transposed epilogue
            static const bool direct4_env = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K4_DIRECT"); return !e || e[0] != '0'; }();
            bool direct4 = pipe && direct4_env;
            for (int i = 0; i < n && direct4; ++i) {
                const vpu_gemm_desc& q = descs[i];
                direct4 = (q.flags & ~(VPU_EPI_OUT_F32 | VPU_EPI_ACCUM)) == 0 && (q.flags & VPU_EPI_OUT_F32) && q.alpha == 1.0f &&
                          q.ldc % 4 == 0 && q.N % 4 == 0 && q.sCo % 4 == 0 && (reinterpret_cast<uintptr_t>(q.C) & 15) == 0 &&
                          (int64_t)q.M * q.ldc * 4 < 0x7FFFFFF0LL;
            }
            if (auto todo_ = attr4_.pending()) {
#ifdef VPU_LAB
                VPU_SET_LDS(K3Cfg<4>::LDS, gemm_bf16_k4_grouped_kernel<1, 1, true>);
#endif
                VPU_SET_LDS(K3Cfg<4>::LDS, gemm_bf16_k4p_grouped_kernel<1, 1, true>);
                VPU_SET_LDS(K3Cfg<4>::LDS, gemm_bf16_k4p_grouped_kernel<1, 1, true, true>);
            }
            const int ncu = cu_count();
            static const bool noepi4 = [] { const char* e = vpu_lab_getenv("VPU_GEMM_NOEPI"); return e && e[0] == '1'; }();
            NOTE_KERNEL("gemm_bf16_k4%s_grouped_kernel<1, 1, true%s>", pipe ? "p" : "", direct4 ? ", true" : "");
            if (direct4) gemm_bf16_k4p_grouped_kernel<1, 1, true, true><<<dim3((unsigned)(total4 < ncu ? total4 : ncu)), dim3(512), K3Cfg<4>::LDS, s>>>(g4, noepi4 ? 9 : 1 VPU_DBG_LOAD);
#ifdef VPU_LAB
            else if (!pipe) gemm_bf16_k4_grouped_kernel<1, 1, true><<<dim3((unsigned)(total4 < ncu ? total4 : ncu)), dim3(512), K3Cfg<4>::LDS, s>>>(g4, noepi4 ? 9 : 1);
#endif
            else gemm_bf16_k4p_grouped_kernel<1, 1, true><<<dim3((unsigned)(total4 < ncu ? total4 : ncu)), dim3(512), K3Cfg<4>::LDS, s>>>(g4, noepi4 ? 9 : 1 VPU_DBG_LOAD);
            return vpu_check_launch("vpu_gemm_grouped");
        }
        if (any_batch && !ok) {
            vpu_set_error("vpu_gemm_grouped: batch > 1 needs K % 64 == 0, K >= 2048, M % 8 == 0, N % 8 == 0 and aligned operands");
            return VPU_ERR_ARG;
        }
#ifdef VPU_LAB
        // (VPU_GEMM_K3G_MIN_K > 0: short-reduction weight-gradient groups -- the neck's token-side gradients, 576 rows -- with a
        // reduction of at least that many rows on the K3 form: A/B knob)
        static const int k3g_min_k = [] { const char* e = vpu_lab_getenv("VPU_GEMM_K3G_MIN_K"); return e ? atoi(e) : 0; }();
        bool ok_short = k3g_min_k > 0 && !any_batch && !any_dcs && total2 >= 96;
        for (int i = 0; i < n && ok_short; ++i)
            ok_short = descs[i].K % BK == 0 && descs[i].K >= k3g_min_k && descs[i].K < 2048 && descs[i].N % 8 == 0 && descs[i].M % 8 == 0;
        if ((ok && (any_batch || ((k3_opt() & 1) && total2 > cu_count()))) || ok_short) {
            static VpuDevOnce attr3_;
            if (auto todo_ = attr3_.pending()) {
                VPU_SET_LDS(K3_LDS, gemm_bf16_k3_grouped_kernel<1, 1, true>);
            }
            const int cap = 2 * cu_count();
            static const bool noepi3 = [] { const char* e = vpu_lab_getenv("VPU_GEMM_NOEPI"); return e && e[0] == '1'; }();
            NOTE_KERNEL("gemm_bf16_k3_grouped_kernel<1, 1, true>");
            gemm_bf16_k3_grouped_kernel<1, 1, true><<<dim3((unsigned)(total2 < cap ? total2 : cap)), dim3(256), K3_LDS, s>>>(g2, noepi3 ? 9 : 1);
            return vpu_check_launch("vpu_gemm_grouped");
        }
        if (ok && (total2 >= 192 || (very_long && total2 >= 96))) {
            static VpuDevOnce attr_;
            auto kern_ = gemm_bf16_k2_grouped_kernel<1, 1, true>;
            if (auto todo_ = attr_.pending()) {
                VPU_SET_LDS(K2Cfg<2>::LDS, kern_);
            }
            const int ncu = cu_count();
            static const bool noepi2 = [] { const char* e = vpu_lab_getenv("VPU_GEMM_NOEPI"); return e && e[0] == '1'; }();
            NOTE_KERNEL("gemm_bf16_k2_grouped_kernel<1, 1, true>");
            kern_<<<dim3((unsigned)(total2 < ncu ? total2 : ncu)), dim3(512), K2Cfg<2>::LDS, s>>>(g2, noepi2 ? 9 : 1);
            return vpu_check_launch("vpu_gemm_grouped");
        }
#else
        (void)very_long; (void)total2;
#endif
    }
    if (any_batch) { vpu_set_error("vpu_gemm_grouped: batch > 1 is only implemented for 16-byte-addressable weight-gradient groups"); return VPU_ERR_ARG; }
    static const int persist_env = [] { const char* e = vpu_lab_getenv("VPU_GEMM_PERSIST"); return e ? atoi(e) : 0; }();
    const int persist_cap = persist_env > 0 ? persist_env : 2 * cu_count();
    dim3 grid((unsigned)(total < persist_cap ? total : persist_cap)), block(256);
    const int vec_arg = vec ? 1 : 0;
    NOTE_KERNEL("gemm_bf16_grouped_kernel<%d, %d, %s>", key >> 1, key & 1, key == 3 ? "true" : "false");
    switch (key) {
        case 0: gemm_bf16_grouped_kernel<0, 0, false><<<grid, block, 4 * TILE_BYTES, s>>>(ga, vec_arg); break;
        case 1: gemm_bf16_grouped_kernel<0, 1, false><<<grid, block, 4 * TILE_BYTES, s>>>(ga, vec_arg); break;
        case 2: gemm_bf16_grouped_kernel<1, 0, false><<<grid, block, 4 * TILE_BYTES, s>>>(ga, vec_arg); break;
        default: gemm_bf16_grouped_kernel<1, 1, true><<<grid, block, 4 * TILE_BYTES, s>>>(ga, vec_arg); break;
    }
    return vpu_check_launch("vpu_gemm_grouped");
}

extern "C" int vpu_debug_gemm_times(void* dev_buf) {
#ifdef VPU_DIAG
    g_dbg_times.store(reinterpret_cast<unsigned long long*>(dev_buf), std::memory_order_relaxed);
    vpu_k5_set_dbg(reinterpret_cast<unsigned long long*>(dev_buf));
    return VPU_OK;
#else
    (void)dev_buf;
    vpu_set_error("vpu_debug_gemm_times: this library was built without -DVPU_DIAG (csrc/build.sh diag builds libvpu_hip_diag.so: the "
                  "product kernels carry no time-stamp code)");
    return VPU_ERR_ARG;
#endif
}
