// Tile-level device helpers shared by the bf16 GEMM translation units (gemm.hip, gemm_k5.hip): LDS images of the operand
// tiles, the fragment reads that match them, the workgroup -> tile order, small register helpers.  gfx950 only.
#pragma once
#include "vpu_common.h"
#include "../../include/vpu_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
constexpr int OOB_OFFSET = (int)0x80000000;  // >= num_records of the buffer descriptors: the load returns zeros

// LDS image of a K-contiguous tile: [128 rows][64 k] bf16, 128-B rows, 16-B chunk index XOR ((row>>1)&7)
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// LDS image of a K-major tile: [64 k][128 cols] bf16, 256-B rows, 8-B unit index XOR f(k)
__device__ __forceinline__ int km_off(int k, int unit) {
    return k * 256 + ((unit ^ ((k & 3) << 2) ^ (((k >> 3) & 1) << 4)) << 3);
}

// fragment for 16 rows (or cols) starting at x16 within the tile, k-substep ks (0/1)
template <int TRANS>
__device__ __forceinline__ bf16x8_t read_frag(const char* lds, int x16, int ks, int lane) {
    if (TRANS == 0) {
        const int row = x16 + (lane & 15);
        const int chunk = ks * 4 + (lane >> 4);
        // (an ext-vector load: through HIP's uint4 struct the load carries TBAA info, and SIInsertWaitcnts then puts
        // s_waitcnt vmcnt(0) in front of every such ds_read while an LDS-DMA is pending -- the ring would never overlap)
        return *reinterpret_cast<const bf16x8_t*>(lds + kc_off(row, chunk));
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        const int k = ks * 32 + 8 * g + q;
        const int unit = (x16 >> 2) + pp;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + km_off(k, unit)));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + km_off(k + 4, unit)));
        s16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8_t, v);
    }
}

// Workgroup -> output tile.  Hardware deals consecutive workgroups round-robin over the 8 XCDs (private 4-MiB L2 each):
// (1) give XCD x a CONTIGUOUS range of a linear tile order (guide T1, bijective for any grid size);
// (2) make that order "M-chunks of tiles_m/8 row-panels, inside a chunk M-fastest": the A row-panels of a chunk
//     (~1.8 MB for fc1 at bs 12) stay resident in the XCD's L2 while each B (weight) tile is fetched once per chunk.
//     With the N-fastest order every row-panel re-streamed the whole weight (4.7 MB > L2): L2 hit rate 65 %,
//     fabric fetch 160 MB for 19 MB of unique inputs (rocprofv3 TCC_HIT/MISS, FETCH_SIZE, round 1).
// (chunking along the longer tile dimension instead -- column chunks for a weight gradient with few row-panels, e.g. fc2's
//  6 x 24 tiles -- was tried in round 1: no measurable change at ViT-B (706 vs 706 images/s), so the one rule is kept.)
__device__ __forceinline__ void tile_coords(int bid, int nwg, int tiles_n, int& tile_m, int& tile_n) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tiles_m = nwg / tiles_n;
    const int cm = (tiles_m + 7) >> 3;
    const int chunk = v / (cm * tiles_n), rem = v - chunk * (cm * tiles_n);
    const int mcount = (tiles_m - chunk * cm) < cm ? (tiles_m - chunk * cm) : cm;
    tile_n = rem / mcount;
    tile_m = chunk * cm + (rem - tile_n * mcount);
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const void* rfl_ptr(const void* q) {
    const uint64_t u = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = (uint32_t)rfl((int)(uint32_t)u), hi = (uint32_t)rfl((int)(uint32_t)(u >> 32));
    return reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
}

// v <- gelu(v), d <- gelu'(v) for 8 values (see the derivation at its use in epilogue_store8, gemm.hip)
__device__ __forceinline__ void gelu_dgelu8(float (&v)[8], float (&d)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) gelu_pair_fast(v[j], v[j], d[j]);
}
__device__ __forceinline__ u32x4v pack_bf16x8(const float (&v)[8]) {
    bf16x8_t a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    return __builtin_bit_cast(u32x4v, a);
}
// first of the 8 consecutive columns (of a wave's 32-column half) a lane owns after the v_permlane16_swap pairing of the
// swapped-operand accumulator tiles (the direct epilogues of K2 / K5)
__device__ __forceinline__ int k2_direct_col(const int lane) { const int fq = lane >> 4; return (fq & 1) * 16 + (fq >> 1) * 8; }

}  // namespace
