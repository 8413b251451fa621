// K5 (round 6): two-tile ping-pong GEMM for the ViT blocks' many-tile forward / dgrad launches on gfx950.
//
//   C[M,N] = epilogue(A[M,K] * op(B)[K,N]),  bf16 operands, row-major A, one of the compile-time flag sets of K2.
//
// What it changes against K2 (gemm.hip): there the eight waves of a workgroup compute ONE tile in lock step and then all
// eight run its epilogue -- the MFMA pipes idle through every epilogue (3-4.7 us of ~13 per 256 x 128 tile, 12 of 30 for
// fc1 + GELU) and the vector ALUs / the store path idle through every main loop.  Here the two 4-wave groups of the
// 512-thread workgroup own ALTERNATE tiles (256 / 224 / 192 rows x 128 columns, 128 x 64 per wave, no K split, so no
// exchange of partial tiles either) and swap two roles tile by tile:
//   * consumer: LDS fragment reads + MFMAs of its tile over the whole K -- one MFMA wave per SIMD, nothing else in its
//     instruction stream but one barrier per 64-deep K-step;
//   * producer: every LDS-DMA piece of the consumer's K-steps (twelve 1-KiB pieces per wave and K-step, two steps
//     ahead, crossing into its OWN next tile at the end), and, in the gaps, the direct epilogue of the tile it has just
//     finished as consumer (swapped MFMA operands + v_permlane16_swap: 16-byte stores straight from the accumulators;
//     sixteen "units" of one store each, two per K-step interval).
// The K-steps of consecutive tiles form ONE continuous stream through a three-stage, 48-KiB-per-stage ring: it never
// drains at a tile boundary.  One workgroup barrier per K-step: before it the producer has waited (counted vmcnt) for the
// NEXT step's pieces, the consumer has finished reading the CURRENT one; after it the stage of the current step is free
// for the step three ahead.  Every global access of the epilogue is an unconditional raw-buffer instruction, so the number
// of vector-memory operations between two DMA issues is a compile-time constant and the waits are exact.
//
// Replaces the same reference calls as K2: nn.Linear of Attention / Mlp (isegm/model/modeling/models_vit.py:16-27,38-56)
// and their input gradients.
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include "gemm_tiles.h"

namespace {

// cache policy of the epilogue's streamed traffic (the residual / aux operand is read once, the output written once): 0 = default,
// 2 = non-temporal -- -DVPU_K5_NT=2 in an experiment build (csrc/build.sh x) for a same-box A/B
#ifndef VPU_K5_NT
#define VPU_K5_NT 0
#endif
constexpr int K5_STAGE = 3 * TILE_BYTES;          // A rows 0-127 | A rows 128-255 (of 16 RB each) | B 128 columns
constexpr int K5_PW = 12;                         // DMA pieces (1 KiB) per producer wave per stage: 48 / 4 waves
constexpr int K5_BIAS_LDS = 2 * 8 * 256;          // [tile parity of a group][wave]: 64 fp32 bias values per wave
constexpr int K5_LDS = 3 * K5_STAGE + K5_BIAS_LDS;
// epilogue intervals of a phase: the operand loads first, then eight intervals of two units each.  A form with a residual / aux
// operand takes one interval more: its operand (cold in the step: the saved GELU' of a block was written a forward pass ago)
// is requested TWO intervals before the first unit that consumes it -- the producer's vector-memory queue is in order, so a
// unit that waits for its operand also holds back the DMA pieces behind it, i.e. the consumer's next K-steps (measured in the
// step: fc2-dgrad x aux 60.6 us with one interval of lead against 50-54 us on warm operands)
template <int FL> constexpr int k5_ni() { return (FL & (VPU_EPI_RESID | VPU_EPI_MULAUX)) ? 10 : 9; }
template <int FL> constexpr int k5_u0() { return (FL & (VPU_EPI_RESID | VPU_EPI_MULAUX)) ? 2 : 1; }     // interval of units 0 and 1
template <int FL, int RB> constexpr int k5_xl1() { return (FL & (VPU_EPI_RESID | VPU_EPI_MULAUX)) ? (RB < 8 ? 1 : 5) : 4; }   // interval that requests the second half's operand
constexpr int K5_NI_MAX = 10;

template <int N> __device__ __forceinline__ void k5_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void k5_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// per-lane byte offset of the twelve pieces one producer wave issues per K-step, at k = 0
template <int TB, int RB>
__device__ __forceinline__ void k5_voff(const vpu_gemm_desc& p, const int m0, const int n0, const int w4, const int lane,
                                        int (&voff)[K5_PW]) {
#pragma unroll
    for (int i = 0; i < K5_PW; ++i) {
        const int sub = i >> 2, pis = w4 + 4 * (i & 3);
        if (sub < 2) {
            const int row = pis * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const int gx = m0 + sub * (16 * RB) + row;
            voff[i] = (gx < p.M && row < 16 * RB) ? (gx * p.lda + chunk * 8) * 2 : OOB_OFFSET;
        } else if (TB == 0) {
            const int row = pis * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const int gx = n0 + row;
            voff[i] = gx < p.N ? (gx * p.ldb + chunk * 8) * 2 : OOB_OFFSET;
        } else {
            const int k = pis * 4 + (lane >> 4);
            const int chunk = (lane & 15) ^ ((k & 3) << 1) ^ (((k >> 3) & 1) << 3);
            const int gx = n0 + chunk * 8;
            voff[i] = gx < p.N ? (k * p.ldb + gx) * 2 : OOB_OFFSET;
        }
    }
}
// pieces [P0, P1) of one K-step from this wave
template <int P0, int P1>
__device__ __forceinline__ void k5_issue(const __amdgpu_buffer_rsrc_t rA, const __amdgpu_buffer_rsrc_t rB, const int (&voff)[K5_PW],
                                         const int soffA, const int soffB, const bool live, char* __restrict__ wr, const int w4) {
#pragma unroll
    for (int i = P0; i < P1; ++i) {
        const int sub = i >> 2, pis = w4 + 4 * (i & 3);
        const int vo = live ? voff[i] : OOB_OFFSET;
        if (sub < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_vptr)(wr + sub * TILE_BYTES + pis * 1024), 16, vo, soffA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_vptr)(wr + 2 * TILE_BYTES + pis * 1024), 16, vo, soffB, 0, 0);
    }
}
__device__ __forceinline__ void k5_bias_issue(const vpu_gemm_desc& p, const int ncol0, const int lane, char* slot) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(p.bias)), 0, p.N * 4, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)slot, 4, (ncol0 + lane) * 4, 0, 0, 0);   // columns >= N: zeros
}

// fragments of one 32-deep half K-step of a consumer wave (its 16 RB x 64 block of the tile) and the MFMAs on them; operands
// swapped: the accumulator tile is C^T, acc[i][j][r] = C[16 i + fr][16 j + 4 fq + r]
template <int TB, int RB>
__device__ __forceinline__ void k5_read(const char* __restrict__ rd, const int lane, const int wm, const int wn, const int kk,
                                        bf16x8_t (&af)[RB], bf16x8_t (&bfr)[4]) {
    const char* la = rd + wm * TILE_BYTES;
    const char* lb = rd + 2 * TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = read_frag<TB>(lb, wn * 64 + j * 16, kk, lane);
#pragma unroll
    for (int i = 0; i < RB; ++i) af[i] = read_frag<0>(la, i * 16, kk, lane);
}
template <int RB, bool PRIO = true>
__device__ __forceinline__ void k5_mma(f32x4_t (&acc)[RB][4], const bf16x8_t (&af)[RB], const bf16x8_t (&bfr)[4]) {
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
}

// the residual / aux operand of one 64-row half of a wave's block: [row block of 16][32-column half], 16 bytes per lane each
struct K5X { u32x4v x[4][2]; };
template <int FL>
__device__ __forceinline__ void k5_xload(const vpu_gemm_desc& p, const int mrow0, const int ncol0, const int lane, K5X& q, const int npass) {
    constexpr bool IS_RES = (FL & VPU_EPI_RESID) != 0;
    if constexpr ((FL & (VPU_EPI_RESID | VPU_EPI_MULAUX)) != 0) {
        const int fr = lane & 15, cl = k2_direct_col(lane);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(IS_RES ? p.resid : p.aux), 0, 0x7FFFFFFF, 0x00020000);
        const int ld = IS_RES ? p.ldr : p.ldaux;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int m = mrow0 + pass * 16 + fr, n = ncol0 + 32 * t + cl;
                const int off = (pass < npass && m < p.M && n < p.N) ? (m * ld + n) * 2 : OOB_OFFSET;
                q.x[pass][t] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, VPU_K5_NT);
            }
    }
}
// one epilogue unit: 16 rows x 32 columns of the wave's block = two accumulator tiles = 8 consecutive columns of one row
// per lane = ONE 16-byte store (two with the saved GELU')
struct K5NoGap { template <int G> __device__ __forceinline__ void at() const {} };
template <int FL, typename GAP = K5NoGap>
__device__ __forceinline__ void k5_unit(const __amdgpu_buffer_rsrc_t rC, const __amdgpu_buffer_rsrc_t rP, const f32x4_t a0, const f32x4_t a1,
                                        const int off, const f32x4_t b0, const f32x4_t b1, const u32x4v xv, const GAP& gap = GAP()) {
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float x0 = a0[r], x1 = a1[r];
        const u32x2v sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
        v[r] = __uint_as_float(sw.x);
        v[4 + r] = __uint_as_float(sw.y);
    }
    if constexpr ((FL & VPU_EPI_BIAS) != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
    }
    if constexpr ((FL & VPU_EPI_GELU) != 0) {
        // the GELU / GELU' pairs two elements at a time, a gap behind each pair: the producer puts its LDS-DMA pieces there, one
        // or two per ~45 vector instructions -- the rate the memory pipe drains them at, so that the issue never finds the
        // queue full (twelve pieces in one burst wait ~130 cycles each at the issue, and the wave's arithmetic waits with them)
        float d[8];
        gelu_pair_fast(v[0], v[0], d[0]); gelu_pair_fast(v[1], v[1], d[1]);
        __builtin_amdgcn_sched_barrier(0); gap.template at<0>(); __builtin_amdgcn_sched_barrier(0);
        gelu_pair_fast(v[2], v[2], d[2]); gelu_pair_fast(v[3], v[3], d[3]);
        __builtin_amdgcn_sched_barrier(0); gap.template at<1>(); __builtin_amdgcn_sched_barrier(0);
        gelu_pair_fast(v[4], v[4], d[4]); gelu_pair_fast(v[5], v[5], d[5]);
        __builtin_amdgcn_sched_barrier(0); gap.template at<2>(); __builtin_amdgcn_sched_barrier(0);
        gelu_pair_fast(v[6], v[6], d[6]); gelu_pair_fast(v[7], v[7], d[7]);
        __builtin_amdgcn_sched_barrier(0); gap.template at<3>(); __builtin_amdgcn_sched_barrier(0);
        if constexpr ((FL & VPU_EPI_SAVE_DGELU) != 0) __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(d), rP, off, 0, VPU_K5_NT);
    }
    if constexpr ((FL & (VPU_EPI_RESID | VPU_EPI_MULAUX)) != 0) {
        const bf16x8_t e = __builtin_bit_cast(bf16x8_t, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (float)e[j];
            if (FL & VPU_EPI_MULAUX) v[j] *= x;
            else v[j] += x;
        }
    }
    __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v), rC, off, 0, VPU_K5_NT);
}

// compile-time bookkeeping of the epilogue schedule: unit u = (half, row block, column half) = (u / 8, (u % 8) / 2, u % 2);
// interval I >= 1 runs units 2 (I - 1) and 2 (I - 1) + 1; interval 0 requests the first half's operand, interval 4 the second's
template <int RB> constexpr bool k5_unit_valid(int u) { return u >= 0 && u < 16 && (u / 8) * 4 + (u % 8) / 2 < RB; }
template <int FL, int RB> constexpr int k5_eops(int I) {       // vector-memory operations of interval I beside its DMA pieces
    constexpr bool HAS_X = (FL & (VPU_EPI_RESID | VPU_EPI_MULAUX)) != 0;
    constexpr int ST_U = (FL & VPU_EPI_SAVE_DGELU) ? 2 : 1;
    int n = 0;
    if (I == 0 && HAS_X) n += 8;
    if (I == k5_xl1<FL, RB>() && HAS_X && RB > 4) n += 8;
    const int i = I - k5_u0<FL>();
    if (i >= 0)
        for (int u = 2 * i; u < 2 * i + 2; ++u) n += k5_unit_valid<RB>(u) ? ST_U : 0;
    return n;
}

template <int TB, int FL, int RB>
struct K5Ctx {
    const vpu_gemm_desc& p;
    __amdgpu_buffer_rsrc_t rA, rB, rC, rP;
    int lane, wave, w4, wm, wn;
    int nk, stepA, stepB, total, tiles_n, G, bid;
    char* lds;
    // DMA cursor (both roles walk it in step)
    int dvoff[K5_PW];
    int wst;            // ring stage that receives the next K-step requested
    int ph, ntl;
    // the tile this group finalizes
    int mq, nq;         // first row / column of the wave's block
    const float* bl;    // its bias slot
    f32x4_t bq[2][2];
    K5X x0, x1;
};

// Where the K-step requested in interval kt of phase ph goes and comes from: K-step kt + 2 of the consumer's tile, or K-step 0 / 1
// of the tile behind it (the PRODUCER group's own next tile; its bias travels in front of its first step)
struct K5Dma { char* wr; int soffA, soffB; bool live; };
template <bool PRODUCER, int TB, int FL, int RB>
__device__ __forceinline__ K5Dma k5_dma_prepare(K5Ctx<TB, FL, RB>& c, const int kt) {
    K5Dma d;
    const int s = kt + 2;
    d.wr = c.lds + c.wst * K5_STAGE;
    c.wst = c.wst == 2 ? 0 : c.wst + 1;
    d.live = true;
    int ks = s;
    if (s >= c.nk) {
        const bool has_next = c.ph + 1 < c.ntl;
        if (s == c.nk && has_next) {
            int tm, tn;
            tile_coords(c.bid + (c.ph + 1) * c.G, c.total, c.tiles_n, tm, tn);
            const int m0 = rfl(tm * (32 * RB)), n0 = rfl(tn * 128);
            k5_voff<TB, RB>(c.p, m0, n0, c.w4, c.lane, c.dvoff);
            if constexpr (PRODUCER && (FL & VPU_EPI_BIAS) != 0)
                k5_bias_issue(c.p, n0 + c.wn * 64, c.lane, c.lds + 3 * K5_STAGE + ((((c.ph + 1) >> 1) & 1) * 8 + c.wave) * 256);
        }
        d.live = has_next;
        ks = s - c.nk;
    }
    d.soffA = ks * c.stepA; d.soffB = ks * c.stepB;
    return d;
}
template <int P0, int P1, int TB, int FL, int RB>
__device__ __forceinline__ void k5_dma_issue(K5Ctx<TB, FL, RB>& c, const K5Dma& d) {
    if constexpr (P0 < P1) k5_issue<P0, P1>(c.rA, c.rB, c.dvoff, d.soffA, d.soffB, d.live, d.wr, c.w4);
}
template <int P0, int P1, bool PRODUCER, int TB, int FL, int RB>
__device__ __forceinline__ void k5_dma(K5Ctx<TB, FL, RB>& c, const int kt) {
    const K5Dma d = k5_dma_prepare<PRODUCER>(c, kt);
    k5_dma_issue<P0, P1>(c, d);
}

template <int TB, int FL, int RB>
__device__ __forceinline__ void k5_epi_begin(K5Ctx<TB, FL, RB>& c) {
    // bias values out of the wave's LDS slot through inline asm (in front of a C++ read of a DMA-written slot hipcc puts
    // s_waitcnt vmcnt(0), which would drain the ring); the DMA that filled the slot is older than the tile's K-step 0
    if constexpr ((FL & VPU_EPI_BIAS) != 0) {
        const unsigned ba = (unsigned)reinterpret_cast<uintptr_t>(c.bl + k2_direct_col(c.lane));
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:144\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(c.bq[0][0]), "=&v"(c.bq[0][1]), "=&v"(c.bq[1][0]), "=&v"(c.bq[1][1]) : "v"(ba) : "memory");
    }
    k5_xload<FL>(c.p, c.mq, c.nq, c.lane, c.x0, 4);
}
template <int TB, int FL, int RB, int U, typename GAP = K5NoGap>
__device__ __forceinline__ void k5_do_unit(K5Ctx<TB, FL, RB>& c, f32x4_t (&acc)[RB][4], const GAP& gap = GAP()) {
    if constexpr (k5_unit_valid<RB>(U)) {
        constexpr int h = U / 8, pass = (U % 8) / 2, t = U % 2;
        const int fr = c.lane & 15, cl = k2_direct_col(c.lane);
        const int m = c.mq + h * 64 + pass * 16 + fr, n = c.nq + 32 * t + cl;
        const int off = (m < c.p.M && n < c.p.N) ? (m * c.p.ldc + n) * 2 : OOB_OFFSET;
        k5_unit<FL, GAP>(c.rC, c.rP, acc[h * 4 + pass][2 * t], acc[h * 4 + pass][2 * t + 1], off, c.bq[t][0], c.bq[t][1],
                         h == 0 ? c.x0.x[pass][t] : c.x1.x[pass][t], gap);
    }
}
// the producer's pieces of one K-step spread over the four gaps of a GELU unit: unit slot S (0 / 1 of its interval) carries
// pieces [S * PWP / 2, (S + 1) * PWP / 2) as 2 + 1 + 2 + 1 (PWP = 12) or 1 + 1 + 1 + 0 (PWP = 6)
template <int TB, int FL, int RB, int PWP, int S>
struct K5PieceGaps {
    K5Ctx<TB, FL, RB>& c;
    const K5Dma& d;
    template <int G> __device__ __forceinline__ void at() const {
        constexpr int H = PWP / 2, B0 = S * H;
        constexpr int lo = PWP == 12 ? (G == 0 ? 0 : G == 1 ? 2 : G == 2 ? 3 : 5) : (G < 3 ? G : 3);
        constexpr int hi = PWP == 12 ? (G == 0 ? 2 : G == 1 ? 3 : G == 2 ? 5 : 6) : (G < 3 ? G + 1 : 3);
        k5_dma_issue<B0 + lo, B0 + hi>(c, d);
    }
};
// epilogue work of interval I in two parts, so that the producer's DMA pieces can go out between them: a wave that issues
// twelve pieces in one burst into a full vector-memory queue waits ~130 cycles per piece at the issue (in order: its vector ALU
// work waits with it); a third of them every ~half unit keeps the queue short of full (fc1 + GELU: the producer was the slower
// role by that blocked time -- see DESIGN section 5)
template <int TB, int FL, int RB, int I, int PART>
__device__ __forceinline__ void k5_epi_part(K5Ctx<TB, FL, RB>& c, f32x4_t (&acc)[RB][4]) {
    constexpr int i = I - k5_u0<FL>();
    if constexpr (PART == 0) {
        if constexpr (I == 0) k5_epi_begin(c);
        if constexpr (I == k5_xl1<FL, RB>() && RB > 4) k5_xload<FL>(c.p, c.mq + 64, c.nq, c.lane, c.x1, RB - 4);
        if constexpr (i >= 0) k5_do_unit<TB, FL, RB, 2 * i>(c, acc);
    } else {
        if constexpr (i >= 0) k5_do_unit<TB, FL, RB, 2 * i + 1>(c, acc);
    }
}
// producer intervals 0 .. k5_ni<FL>() - 1 of a phase with an epilogue: DMA in three bursts with the epilogue work between them, then
// the counted wait for everything this wave requested BEFORE this interval (the pieces of the next K-step among it: behind them
// the wave has issued this interval's PWP pieces and E(I) epilogue operations), barrier
template <int TB, int FL, int RB, int PWP, int I>
__device__ __forceinline__ void k5_producer_intervals(K5Ctx<TB, FL, RB>& c, f32x4_t (&acc)[RB][4]) {
    const K5Dma d = k5_dma_prepare<true>(c, I);
    constexpr int iu = I - k5_u0<FL>();
    if constexpr ((FL & VPU_EPI_GELU) != 0 && iu >= 0 && k5_unit_valid<RB>(2 * iu) && k5_unit_valid<RB>(2 * iu + 1)) {
        // vector-ALU-heavy units: the pieces inside them, at the rate the memory pipe takes them
        k5_do_unit<TB, FL, RB, 2 * iu>(c, acc, K5PieceGaps<TB, FL, RB, PWP, 0>{c, d});
        k5_do_unit<TB, FL, RB, 2 * iu + 1>(c, acc, K5PieceGaps<TB, FL, RB, PWP, 1>{c, d});
    } else {
        k5_dma_issue<0, PWP / 3>(c, d);
        __builtin_amdgcn_sched_barrier(0);
        k5_epi_part<TB, FL, RB, I, 0>(c, acc);
        __builtin_amdgcn_sched_barrier(0);
        k5_dma_issue<PWP / 3, 2 * PWP / 3>(c, d);
        __builtin_amdgcn_sched_barrier(0);
        k5_epi_part<TB, FL, RB, I, 1>(c, acc);
        __builtin_amdgcn_sched_barrier(0);
        k5_dma_issue<2 * PWP / 3, PWP>(c, d);
    }
    // (interval 0 with all pieces from the producer: nothing of this wave is older than this interval)
    if constexpr (I > 0 || PWP < K5_PW) k5_wait_vm<PWP + k5_eops<FL, RB>(I)>();
    k5_barrier();
    if constexpr (I + 1 < k5_ni<FL>()) k5_producer_intervals<TB, FL, RB, PWP, I + 1>(c, acc);
}
template <int TB, int FL, int RB, int I>
__device__ __forceinline__ void k5_epi_interval(K5Ctx<TB, FL, RB>& c, f32x4_t (&acc)[RB][4]) {
    k5_epi_part<TB, FL, RB, I, 0>(c, acc);
    k5_epi_part<TB, FL, RB, I, 1>(c, acc);
}
template <int TB, int FL, int RB, int I>
__device__ __forceinline__ void k5_epi_all(K5Ctx<TB, FL, RB>& c, f32x4_t (&acc)[RB][4]) {
    k5_epi_interval<TB, FL, RB, I>(c, acc);
    if constexpr (I + 1 < k5_ni<FL>()) k5_epi_all<TB, FL, RB, I + 1>(c, acc);
}

// PWP: LDS-DMA pieces per K-step a PRODUCER wave issues -- 12 (all of them) or 6 (the consumer waves issue the other half:
// for the epilogue-heavy flag sets, whose producer is the slower of the two roles)
template <int TB, int FL, int RB, int PWP>
__global__ __launch_bounds__(512) void gemm_bf16_k5_kernel(const vpu_gemm_desc p, const int tiles_m, const int tiles_n, const int vec
                                                           VPU_DBG_PARAM_DEF) {   // (-DVPU_DIAG, vpu_debug_gemm_times: per tile start / main loop end / epilogue end)
    constexpr int PWC = K5_PW - PWP;
    // the consumer's MFMA clusters at raised priority -- except beside a vector-ALU-heavy epilogue (GELU + GELU': 4.2 vector
    // instructions per MFMA of the partner wave), which the priority starves: the producer then is the slower role
    constexpr bool MPRIO = (FL & VPU_EPI_GELU) == 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    K5Ctx<TB, FL, RB> c{p};
    c.lane = tid & 63;
    c.wave = rfl(tid >> 6);
    const int g = c.wave >> 2;
    c.w4 = c.wave & 3; c.wm = c.w4 >> 1; c.wn = c.w4 & 1;
    c.total = tiles_m * tiles_n; c.tiles_n = tiles_n; c.G = gridDim.x; c.bid = blockIdx.x;
    c.ntl = (c.total - c.bid + c.G - 1) / c.G;
    c.nk = rfl(p.K / BK);
    c.stepA = BK * 2; c.stepB = TB ? p.ldb * (BK * 2) : BK * 2;
    c.lds = lds;
    c.rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0x7FFFFFFF, 0x00020000);
    c.rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, 0x7FFFFFFF, 0x00020000);
    c.rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7FFFFFFF, 0x00020000);
    c.rP = __builtin_amdgcn_make_buffer_rsrc(p.preact, 0, 0x7FFFFFFF, 0x00020000);
    c.mq = 0; c.nq = 0; c.bl = nullptr; c.wst = 0; c.ph = 0;
    const bool noepi = vec == 9;       // diagnostic (VPU_GEMM_NOEPI=1): main loops only
    VPU_STAMP(tid == 0, blockIdx.x * 16 + 15);

    f32x4_t acc[RB][4];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // K-steps 0 and 1 of tile 0 (group 0's): by group 0 alone, or half by each group -- as every later tile's first two steps
    // are requested in the last two intervals of the phase before
    {
        int tm, tn;
        tile_coords(c.bid, c.total, tiles_n, tm, tn);
        const int m0 = rfl(tm * (32 * RB)), n0 = rfl(tn * 128);
        k5_voff<TB, RB>(p, m0, n0, c.w4, c.lane, c.dvoff);
        if (g == 0) {
            if constexpr ((FL & VPU_EPI_BIAS) != 0) k5_bias_issue(p, n0 + c.wn * 64, c.lane, lds + 3 * K5_STAGE + c.wave * 256);
            k5_issue<PWC, K5_PW>(c.rA, c.rB, c.dvoff, 0, 0, true, lds, c.w4);
            k5_issue<PWC, K5_PW>(c.rA, c.rB, c.dvoff, c.stepA, c.stepB, true, lds + K5_STAGE, c.w4);
            k5_wait_vm<K5_PW - PWC>();
        } else if constexpr (PWC > 0) {
            k5_issue<0, PWC>(c.rA, c.rB, c.dvoff, 0, 0, true, lds, c.w4);
            k5_issue<0, PWC>(c.rA, c.rB, c.dvoff, c.stepA, c.stepB, true, lds + K5_STAGE, c.w4);
            k5_wait_vm<PWC>();
        }
    }
    k5_barrier();

    int q = 0;     // ring stage of K-step 0 of the current phase
    for (int ph = 0; ph <= c.ntl; ++ph) {
        const bool has_m = ph < c.ntl;
        c.ph = ph;
        if (has_m) {
            int tm, tn;
            tile_coords(c.bid + ph * c.G, c.total, tiles_n, tm, tn);
            const int m0 = rfl(tm * (32 * RB)), n0 = rfl(tn * 128);
            if (g == (ph & 1)) {
                // ---- consumer of tile ph: a pair of fragment sets, each read one half K-step ahead of its MFMAs; the barrier of
                // K-step kt sits between the two halves: behind it this wave has read all of stage kt (the producer may
                // overwrite it) and K-step kt + 1 has landed
                c.mq = m0 + c.wm * (16 * RB);
                c.nq = n0 + c.wn * 64;
                c.bl = reinterpret_cast<const float*>(lds + 3 * K5_STAGE + (((ph >> 1) & 1) * 8 + c.wave) * 256);
                if constexpr (PWC > 0) k5_voff<TB, RB>(p, m0, n0, c.w4, c.lane, c.dvoff);
                c.wst = q + 2 >= 3 ? q - 1 : q + 2;
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                VPU_STAMP((tid & 255) == 0 && ph < 5, blockIdx.x * 16 + 3 * ph);
                int rst = q;
                if constexpr (RB < 8) {
                    bf16x8_t a0[RB], b0[4], a1[RB], b1[4];
                    k5_read<TB, RB>(lds + rst * K5_STAGE, c.lane, c.wm, c.wn, 0, a0, b0);
                    for (int kt = 0; kt < c.nk; ++kt) {
                        k5_dma<PWP, K5_PW, false>(c, kt);
                        k5_read<TB, RB>(lds + rst * K5_STAGE, c.lane, c.wm, c.wn, 1, a1, b1);
                        __builtin_amdgcn_sched_barrier(0);
                        k5_mma<RB, MPRIO>(acc, a0, b0);
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        // this wave's pieces of K-step kt + 1 have landed (kt = 0: those it requested as the producer of the phase
                        // before, with everything older -- its epilogue stores)
                        k5_wait_vm<PWC>();
                        k5_barrier();
                        rst = rst == 2 ? 0 : rst + 1;
                        // (behind the last K-step this reads the next tile's first stage: landed, and not used)
                        k5_read<TB, RB>(lds + rst * K5_STAGE, c.lane, c.wm, c.wn, 0, a0, b0);
                        __builtin_amdgcn_sched_barrier(0);
                        k5_mma<RB, MPRIO>(acc, a1, b1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    // 256-row tiles: 128 accumulator registers leave room for ONE fragment set -- both halves of a K-step in front
                    // of its barrier
                    for (int kt = 0; kt < c.nk; ++kt) {
                        k5_dma<PWP, K5_PW, false>(c, kt);
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            bf16x8_t a0[RB], b0[4];
                            k5_read<TB, RB>(lds + rst * K5_STAGE, c.lane, c.wm, c.wn, kk, a0, b0);
                            k5_mma<RB, MPRIO>(acc, a0, b0);
                        }
                        k5_wait_vm<PWC>();
                        k5_barrier();
                        rst = rst == 2 ? 0 : rst + 1;
                    }
                }
                VPU_STAMP((tid & 255) == 0 && ph < 5, blockIdx.x * 16 + 3 * ph + 1);
            } else {
                // ---- producer for tile ph (+ the epilogue of tile ph - 1)
                const bool has_e = ph >= 1 && !noepi;
                k5_voff<TB, RB>(p, m0, n0, c.w4, c.lane, c.dvoff);
                c.wst = q + 2 >= 3 ? q - 1 : q + 2;
                int kt = 0;
                if (has_e) {
                    // (a vector-ALU-heavy epilogue is the slower role: it takes the issue priority its partner's MFMAs leave alone)
                    if constexpr (!MPRIO) __builtin_amdgcn_s_setprio(2);
                    k5_producer_intervals<TB, FL, RB, PWP, 0>(c, acc);
                    if constexpr (!MPRIO) __builtin_amdgcn_s_setprio(0);
                    kt = k5_ni<FL>();
                    VPU_STAMP((tid & 255) == 0 && ph - 1 < 5, blockIdx.x * 16 + 3 * (ph - 1) + 2);
                }
                for (; kt < c.nk; ++kt) {
                    k5_dma<0, PWP, true>(c, kt);
                    k5_wait_vm<PWP>();
                    k5_barrier();
                }
            }
        } else if (g != (ph & 1) && !noepi && ph >= 1) {
            k5_epi_all<TB, FL, RB, 0>(c, acc);
            VPU_STAMP((tid & 255) == 0 && ph - 1 < 5, blockIdx.x * 16 + 3 * (ph - 1) + 2);
        }
        q = (q + c.nk) % 3;
    }
    if (noepi) {   // keep the accumulators live
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 1.2345678e30f) reinterpret_cast<float*>(p.C)[0] = t;
    }
}

std::atomic<int> g_opt_k5{-1};
std::atomic<int> g_opt_k5_grid{0};
inline int k5_env0() {
    static const int v = [] { const char* e = getenv("VPU_GEMM_K5"); return e ? atoi(e) : 1; }();
    return v;
}

std::atomic<int> g_opt_k5_noepi{0};      // diagnostic: main loops only (the outputs are not written)
#ifdef VPU_DIAG
std::atomic<unsigned long long*> g_dbg_times{nullptr};     // (VPU_DBG_LOAD)
#endif
std::atomic<int> g_opt_k5_split{-1};     // -1 / 0: the producer waves issue every LDS-DMA piece; 1: half of them by the consumer waves (A/B runs, tests)
template <int TB, int FL, int RB, int PWP>
int k5_launch_pw(const vpu_gemm_desc* d, const int ncu, const int vec, hipStream_t s, char* name, size_t name_len) {
    static VpuDevOnce attr_;
    auto kern_ = gemm_bf16_k5_kernel<TB, FL, RB, PWP>;
    if (auto todo_ = attr_.pending()) { VPU_SET_LDS(K5_LDS, kern_); }
    const int tn_ = (d->N + 127) / 128, tm_ = (d->M + 32 * RB - 1) / (32 * RB);
    const int tot_ = tm_ * tn_;
    int grid = tot_ < ncu ? tot_ : ncu;
    const int cap = g_opt_k5_grid.load(std::memory_order_relaxed);
    if (cap > 0 && grid > cap) grid = cap;
    snprintf(name, name_len, "gemm_bf16_k5_kernel<%d, %d, %d, %d>", TB, FL, RB, PWP);
    kern_<<<dim3((unsigned)grid), dim3(512), K5_LDS, s>>>(*d, tm_, tn_, g_opt_k5_noepi.load(std::memory_order_relaxed) ? 9 : vec VPU_DBG_LOAD);
    return 1;
}
template <int TB, int FL, int RB>
int k5_launch_one(const vpu_gemm_desc* d, const int ncu, const int vec, hipStream_t s, char* name, size_t name_len) {
    const int sp = g_opt_k5_split.load(std::memory_order_relaxed);
    // (measured, tools/gemm_bench.py GEMM_BENCH_K2=k5n,k5s: ANY LDS-DMA piece in the consumer's instruction stream costs more
    // than it takes off the producer -- half of them: qkv 38.3 -> 44.6 us, fc2-dgrad x aux 49.8 -> 56.0; a quarter (nine by the
    // producer, three by the consumer): 43.1 / 55.1.  The split form stays selectable for the tests and for A/B runs only.)
    const int split = sp >= 0 ? sp : 0;
    if constexpr (RB < 8) {
        if (split == 1) return k5_launch_pw<TB, FL, RB, 6>(d, ncu, vec, s, name, name_len);
    }
    return k5_launch_pw<TB, FL, RB, 12>(d, ncu, vec, s, name, name_len);
}
template <int TB, int FL>
int k5_launch_rb(const vpu_gemm_desc* d, const int rb, const int ncu, const int vec, hipStream_t s, char* name, size_t name_len) {
    if (rb == 8) return k5_launch_one<TB, FL, 8>(d, ncu, vec, s, name, name_len);
    if (rb == 7) return k5_launch_one<TB, FL, 7>(d, ncu, vec, s, name, name_len);
    if (rb == 6) return k5_launch_one<TB, FL, 6>(d, ncu, vec, s, name, name_len);
    return 0;
}

}  // namespace

// option value of the K5 family: -1 environment default (VPU_GEMM_K5, 1 if unset), 0 off, 1 wherever the form is legal except
// the GELU flag set, 2 that one too (tests)
int vpu_k5_option() { const int v = g_opt_k5.load(std::memory_order_relaxed); return v >= 0 ? v : k5_env0(); }
void vpu_k5_set_option(int v) { g_opt_k5.store(v, std::memory_order_relaxed); }
void vpu_k5_set_grid(int v) { g_opt_k5_grid.store(v, std::memory_order_relaxed); }
int vpu_k5_grid() { return g_opt_k5_grid.load(std::memory_order_relaxed); }
void vpu_k5_set_split(int v) { g_opt_k5_split.store(v, std::memory_order_relaxed); }
void vpu_k5_set_noepi(int v) { g_opt_k5_noepi.store(v, std::memory_order_relaxed); }
#ifdef VPU_DIAG
void vpu_k5_set_dbg(unsigned long long* p) { g_dbg_times.store(p, std::memory_order_relaxed); }
#endif

// Launches the K5 instantiation for (transB, flag set, tile height); returns 1 when a kernel was enqueued, 0 when the form has no
// instantiation (the caller falls through to K2), a negative error code otherwise.  Preconditions (checked by vpu_gemm): bf16,
// row-major A, batch 1, no column sums, vector epilogue, N % 8 == 0, K % 64 == 0, K >= 9 * 64, alpha == 1, 31-bit offsets.
int vpu_k5_launch(const vpu_gemm_desc* d, int rb, int ncu, int vec, void* stream, char* name, size_t name_len) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int F_B = VPU_EPI_BIAS, F_BR = VPU_EPI_BIAS | VPU_EPI_RESID,
                  F_G = VPU_EPI_BIAS | VPU_EPI_GELU | VPU_EPI_SAVE_DGELU, F_M = VPU_EPI_MULAUX;
    const int f = d->flags;
    if (d->transA || d->K < 9 * BK || d->K % BK) return 0;
    if ((f & (VPU_EPI_RESID | VPU_EPI_MULAUX)) && d->K < K5_NI_MAX * BK) return 0;      // (these forms: ten epilogue intervals)
    if (!d->transB) {
        if (f == F_B) return k5_launch_rb<0, F_B>(d, rb, ncu, vec, s, name, name_len);
        if (f == F_BR) return k5_launch_rb<0, F_BR>(d, rb, ncu, vec, s, name, name_len);
        if (f == F_G) return k5_launch_rb<0, F_G>(d, rb, ncu, vec, s, name, name_len);
        if (f == 0) return k5_launch_rb<0, 0>(d, rb, ncu, vec, s, name, name_len);
    } else {
        if (f == 0) return k5_launch_rb<1, 0>(d, rb, ncu, vec, s, name, name_len);
        if (f == F_M) return k5_launch_rb<1, F_M>(d, rb, ncu, vec, s, name, name_len);
    }
    return 0;
}
