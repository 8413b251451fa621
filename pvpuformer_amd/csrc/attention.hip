// Fused (flash-style) multi-head attention, bf16, gfx950: the self-attention of the MAE-ViT blocks (head dim 32 / 64)
// and the prompt<->image attention of the DMA neck (transformer.py:484-521: 48 prompt tokens x 784 image tokens, head
// dim 48 or 96, query and key/value rows in different matrices).
//
// Replaces  attn = softmax(q k^T * scale); x = attn v   (isegm/model/modeling/models_vit.py:43-52) and its autograd
// backward without materialising the [B, heads, n, n] score / probability tensors (354 MB fp32 per global block at
// bs 12 in the unfused path).  Window attention (models_vit.py:225-255) is the same kernel with n = 196: tokens are
// kept in window order, so a (batch, window) pair is a contiguous run of rows.
//
// Layout: q is a column slice of a [*, ldq] matrix, k and v of [*, ldk] matrices (head h at column h*hd of each slice);
// o / d_o are [*, ldo].  Batch entry b = rows [b*nq, (b+1)*nq) of q/o and rows [b*nk, (b+1)*nk) of k/v.
// The kernels are instantiated for HD = 32, 64, 128 columns of LDS image; a head dim of 48 / 96 runs in the 64 / 128
// instantiation with the columns >= hd staged as zeros (hd % 16 == 0).
//
// MFMA plan (v_mfma_f32_16x16x32_bf16; D[row = 4*(lane>>4)+r][col = lane&15]):
//   forward, per wave 16 queries:  S^T[key][q] = K_tile . Q^T  puts the query on the lane, so the softmax statistics
//   are per-lane scalars (max/sum over keys = in-lane over r + two xor-shuffles over the 4 lane groups), and two
//   16-key accumulator tiles ARE the A fragment of the P.V product (k order permuted; V is read with the matching row
//   permutation through ds_read_b64_tr_b16) -- P never goes through LDS.
//   backward: the same trick with the roles swapped: dK/dV kernel keeps 16 keys per wave and sums over queries,
//   dQ kernel keeps 16 queries per wave and sums over keys (probabilities are recomputed from the saved
//   log-sum-exp; no atomics, bitwise reproducible).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "vpu_common.h"
#include "../../include/vpu_hip.h"

namespace {

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
constexpr int CH = 32;  // keys (or queries) staged per iteration

// row-contiguous image [CH][HD]: 16-B chunk index XOR-swizzled, read back with ds_read_b128 (MFMA A operand)
template <int HD> __device__ __forceinline__ int rc_off(int row, int chunk) {
    return row * (HD * 2) + ((chunk ^ ((row >> 1) & (HD / 8 - 1))) << 4);
}
// image for transposing reads [CH][HD]: 8-B unit index XOR-swizzled, read with ds_read_b64_tr_b16 (MFMA B operand, k = row)
template <int HD> __device__ __forceinline__ int tr_off(int row, int unit) {
    return row * (HD * 2) + ((unit ^ ((((row >> 1) & 3) << 2) & (HD / 4 - 1))) << 3);
}

// rows [r0, r0 + NS*CH) x HD columns of a [*, ld] matrix: fetch_rows pulls this thread's 16-B chunks into registers (rows
// >= n and columns >= hd as zeros), put_rows writes them into an LDS image.  A staged block is NS sub-chunks of CH rows:
// the kernels walk the sub-chunks of a block without any barrier, and fetch block i+1 right after block i has been put.
// (The swizzles depend on the row modulo 16 only, so a sub-chunk is just a byte offset into the image.  NS > 1 trades
// occupancy for fewer barrier pairs / round trips; the launchers use NS = 1, see the measurement there.)
template <int HD, int NS> struct RowChunk { uint4 v[(NS * CH * HD / 8 + 255) / 256]; };
template <int HD, int NS>
__device__ __forceinline__ void fetch_rows(const bf16_t* __restrict__ base, int ld, int r0, int n, int tid, int hd,
                                           RowChunk<HD, NS>& rc) {
    constexpr int CPR = HD / 8;  // 16-B chunks per row
#pragma unroll
    for (int i = 0; i < (NS * CH * CPR + 255) / 256; ++i) {
        const int c = tid + i * 256;
        const int row = c / CPR, ch = c % CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < NS * CH * CPR && r0 + row < n && ch * 8 < hd)
            v = *reinterpret_cast<const uint4*>(base + (int64_t)(r0 + row) * ld + ch * 8);
        rc.v[i] = v;
    }
}
template <int HD, int NS, bool TR>
__device__ __forceinline__ void put_rows(char* lds, int tid, const RowChunk<HD, NS>& rc) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < (NS * CH * CPR + 255) / 256; ++i) {
        const int c = tid + i * 256;
        if (c < NS * CH * CPR) {
            const int row = c / CPR, ch = c % CPR;
            const int off = TR ? tr_off<HD>(row, ch * 2) : rc_off<HD>(row, ch);
            *reinterpret_cast<uint4*>(lds + off) = rc.v[i];
        }
    }
}

template <int HD> __device__ __forceinline__ bf16x8_t frag_rc(const char* lds, int row16, int ks, int lane) {
    const uint4 v = *reinterpret_cast<const uint4*>(lds + rc_off<HD>(row16 + (lane & 15), ks * 4 + (lane >> 4)));
    return __builtin_bit_cast(bf16x8_t, v);
}
// B fragment of a product that sums over the CH staged rows in the PERMUTED k order of an accumulator pair:
// k-slot (g, j) <-> row 4g + j (j < 4) or 16 + 4g + (j - 4); columns [16*dt, 16*dt + 16)
template <int HD> __device__ __forceinline__ bf16x8_t frag_tr_perm(const char* lds, int dt, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int unit = dt * 4 + p;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + tr_off<HD>(4 * g + q, unit)));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + tr_off<HD>(16 + 4 * g + q, unit)));
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}
// 16 rows x HD of a global matrix as MFMA B fragments (col = row index on the lane), zero beyond n
template <int HD>
__device__ __forceinline__ void load_rows_as_b(const bf16_t* __restrict__ base, int ld, int row0, int n, int lane,
                                               bf16x8_t (&f)[HD / 32], int hd) {
    const int row = row0 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        const int col = ks * 32 + (lane >> 4) * 8;
        if (row < n && col < hd) v = *reinterpret_cast<const uint4*>(base + (int64_t)row * ld + col);
        f[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
}
// the same for the first NK 32-column groups only
template <int NK>
__device__ __forceinline__ void load_rows_as_bn(const bf16_t* __restrict__ base, int ld, int row0, int n, int lane,
                                                bf16x8_t (&f)[NK], int hd) {
    const int row = row0 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        const int col = ks * 32 + (lane >> 4) * 8;
        if (row < n && col < hd) v = *reinterpret_cast<const uint4*>(base + (int64_t)row * ld + col);
        f[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
}
__device__ __forceinline__ bf16x8_t pack_pair(const f32x4_t& a, const f32x4_t& b) {
    bf16x8_t f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = (bf16_t)a[j]; f[4 + j] = (bf16_t)b[j]; }
    return f;
}

struct AttnArgs {
    const bf16_t *q, *k, *v;   // slices of qkv (already offset to column 0 of the slice)
    const bf16_t *o, *d_o;
    bf16_t *out, *dq, *dk, *dv;
    float *lse, *delta;
    int nq, nk, H, hd;         // hd: real head dim (<= HD of the instantiation)
    int ldq, ldk, ldo;         // row strides of q, of k/v, of o/d_o/out
    int ldgq, ldgk;            // row strides of dq and of dk/dv
    float scale;
    int gx, nbh;               // see wg_problem
    // "split" launches (round 5, the lean kernels): batch entry b of the launch reads its queries (q, o, d_o, and in backward lse /
    // delta) from entry b / qdiv and its keys / values from entry b / kdiv of the operands, and writes its outputs (out + lse in
    // forward; dq; dk / dv) to entry b.  qdiv = kdiv = 1 (0 is taken as 1): the plain batch.  kdiv = S: S consecutive entries are
    // S query ranges of one problem against the same keys (dk / dv come out as S partial sums); qdiv = S: S key ranges of one
    // problem for the same queries (forward: S partial softmaxes, combined by attn_combine_kernel; dq: S partial sums).
    int qdiv, kdiv;
};

// ------------------------------------------------------------------------------------------------ forward
// QT query tiles of 16 per wave (QT = 2: the two tiles share every K / V fragment read and give the scheduler two
// independent softmax dependency chains to interleave; a workgroup covers 64 * QT queries).
template <int HD, int NS, int QT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char ldsK[NS * CH * HD * 2];
    __shared__ __attribute__((aligned(16))) char ldsV[NS * CH * HD * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.y, bw = bh / a.H, h = bh % a.H, nq = a.nq, nk = a.nk, hd = a.hd;
    const int64_t rbq = (int64_t)bw * nq, rbk = (int64_t)bw * nk;
    const bf16_t* q = a.q + rbq * a.ldq + h * hd;
    const bf16_t* k = a.k + rbk * a.ldk + h * hd;
    const bf16_t* v = a.v + rbk * a.ldk + h * hd;
    const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);
    const float sc2 = a.scale * 1.44269504089f;
    bf16x8_t qf[QT][HD / 32];
    float m[QT], l[QT];
    f32x4_t acc[QT][HD / 16];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        load_rows_as_b<HD>(q, a.ldq, q0 + 16 * u, nq, lane, qf[u], hd);
        m[u] = -INFINITY; l[u] = 0.f;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) acc[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    RowChunk<HD, NS> pk, pv;
    fetch_rows<HD, NS>(k, a.ldk, 0, nk, tid, hd, pk);
    fetch_rows<HD, NS>(v, a.ldk, 0, nk, tid, hd, pv);
    for (int kb = 0; kb < nk; kb += NS * CH) {
        __syncthreads();
        put_rows<HD, NS, false>(ldsK, tid, pk);
        put_rows<HD, NS, true>(ldsV, tid, pv);
        __syncthreads();
        if (kb + NS * CH < nk) {
            fetch_rows<HD, NS>(k, a.ldk, kb + NS * CH, nk, tid, hd, pk);
            fetch_rows<HD, NS>(v, a.ldk, kb + NS * CH, nk, tid, hd, pv);
        }
        for (int sub = 0; sub < NS && kb + sub * CH < nk; ++sub) {
        const int kc = kb + sub * CH;
        const char* sK = ldsK + sub * (CH * HD * 2);
        const char* sV = ldsV + sub * (CH * HD * 2);
        f32x4_t s[QT][2];
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) s[u][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                const bf16x8_t kfr = frag_rc<HD>(sK, 16 * t, ks, lane);
#pragma unroll
                for (int u = 0; u < QT; ++u)
                    s[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[u][ks], s[u][t], 0, 0, 0);
            }
        bf16x8_t pf[QT];
        float ar[QT][4];
        const bool tail = kc + CH > nk;   // only the last chunk has keys to mask (block-uniform)
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            // scores in the log2 domain: x = s * (scale * log2 e), p = exp2(x - max): one multiply, one subtract and one
            // v_exp_f32 per element (the running max / log-sum-exp are converted back to natural units at the end)
            float mx = m[u];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = s[u][t][r] * sc2;
                    if (tail) x = (kc + 16 * t + 4 * g + r < nk) ? x : -INFINITY;
                    s[u][t][r] = x;
                    mx = fmaxf(mx, x);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float alpha = __builtin_amdgcn_exp2f(m[u] - mx);
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[u][t][r] = __builtin_amdgcn_exp2f(s[u][t][r] - mx); ps += s[u][t][r]; }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            l[u] = l[u] * alpha + ps;
            m[u] = mx;
#pragma unroll
            for (int r = 0; r < 4; ++r) ar[u][r] = __shfl(alpha, 4 * g + r, 64);
            pf[u] = pack_pair(s[u][0], s[u][1]);
        }
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
            const bf16x8_t vfr = frag_tr_perm<HD>(sV, dt, lane);
#pragma unroll
            for (int u = 0; u < QT; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[u][dt][r] *= ar[u][r];
                acc[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[u], vfr, acc[u][dt], 0, 0, 0);
            }
        }
        }
    }
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int qu = q0 + 16 * u;
        const float il = 1.0f / l[u];
        if (g == 0 && qu + c < nq) a.lse[(int64_t)bh * nq + qu + c] = (m[u] + __builtin_amdgcn_logf(l[u])) * 0.69314718056f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s1 = __shfl(il, 4 * g + r, 64);
            const int qq = qu + 4 * g + r;
            if (qq < nq) {
                bf16_t* orow = a.out + (rbq + qq) * a.ldo + h * hd;
#pragma unroll
                for (int dt = 0; dt < HD / 16; ++dt)
                    if (dt * 16 < hd) orow[dt * 16 + c] = (bf16_t)(acc[u][dt][r] * s1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <int HD, int NS>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char ldsQr[NS * CH * HD * 2], ldsQt[NS * CH * HD * 2];
    __shared__ __attribute__((aligned(16))) char ldsOr[NS * CH * HD * 2], ldsOt[NS * CH * HD * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.y, bw = bh / a.H, h = bh % a.H, nq = a.nq, nk = a.nk, hd = a.hd;
    const int64_t rbq = (int64_t)bw * nq, rbk = (int64_t)bw * nk;
    const bf16_t* q = a.q + rbq * a.ldq + h * hd;
    const bf16_t* k = a.k + rbk * a.ldk + h * hd;
    const bf16_t* v = a.v + rbk * a.ldk + h * hd;
    const bf16_t* d_o = a.d_o + rbq * a.ldo + h * hd;
    const float* lse = a.lse + (int64_t)bh * nq;
    const float* dl = a.delta + (int64_t)bh * nq;
    const int key0 = blockIdx.x * 64 + wave * 16;
    const bool key_ok = key0 + c < nk;
    bf16x8_t kf[HD / 32], vf[HD / 32];
    load_rows_as_b<HD>(k, a.ldk, key0, nk, lane, kf, hd);
    load_rows_as_b<HD>(v, a.ldk, key0, nk, lane, vf, hd);
    f32x4_t adk[HD / 16], adv[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) { adk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; adv[dt] = adk[dt]; }
    RowChunk<HD, NS> pq, po;
    fetch_rows<HD, NS>(q, a.ldq, 0, nq, tid, hd, pq);
    fetch_rows<HD, NS>(d_o, a.ldo, 0, nq, tid, hd, po);
    for (int qb = 0; qb < nq; qb += NS * CH) {
        __syncthreads();
        put_rows<HD, NS, false>(ldsQr, tid, pq);
        put_rows<HD, NS, true>(ldsQt, tid, pq);
        put_rows<HD, NS, false>(ldsOr, tid, po);
        put_rows<HD, NS, true>(ldsOt, tid, po);
        __syncthreads();
        if (qb + NS * CH < nq) {
            fetch_rows<HD, NS>(q, a.ldq, qb + NS * CH, nq, tid, hd, pq);
            fetch_rows<HD, NS>(d_o, a.ldo, qb + NS * CH, nq, tid, hd, po);
        }
        for (int sub = 0; sub < NS && qb + sub * CH < nq; ++sub) {
        const int qc = qb + sub * CH;
        const int so = sub * (CH * HD * 2);
        f32x4_t P[2], dS[2];
        // the four log-sum-exp / delta values a lane needs per query tile are consecutive: one 16-byte load each, requested
        // before the MFMAs instead of eight scalar loads between the MFMAs and the exponentials (window backward 94 -> 81 us,
        // global 223 -> 184 us; a second register set that requests the staged Q / dO blocks two iterations ahead was
        // measured on top of this: slower, 88 / 203 us, not kept)
        f32x4_t lv[2], dv4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q4 = qc + 16 * t + 4 * g;
            if (q4 + 3 < nq && (nq & 3) == 0) {
                lv[t] = *reinterpret_cast<const f32x4_t*>(lse + q4);
                dv4[t] = *reinterpret_cast<const f32x4_t*>(dl + q4);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    lv[t][r] = q4 + r < nq ? lse[q4 + r] : 0.f;
                    dv4[t][r] = q4 + r < nq ? dl[q4 + r] : 0.f;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rc<HD>(ldsQr + so, 16 * t, ks, lane), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rc<HD>(ldsOr + so, 16 * t, ks, lane), vf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // element [query = qc+16t+4g+r][key = key0+c]
                const int qq = qc + 16 * t + 4 * g + r;
                float p = 0.f, ds = 0.f;
                if (qq < nq && key_ok) {
                    p = __expf(s[r] * a.scale - lv[t][r]);
                    ds = p * (dp[r] - dv4[t][r]) * a.scale;
                }
                P[t][r] = p; dS[t][r] = ds;
            }
        }
        const bf16x8_t pf = pack_pair(P[0], P[1]), dsf = pack_pair(dS[0], dS[1]);
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
            adv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, frag_tr_perm<HD>(ldsOt + so, dt, lane), adv[dt], 0, 0, 0);
            adk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, frag_tr_perm<HD>(ldsQt + so, dt, lane), adk[dt], 0, 0, 0);
        }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kk = key0 + 4 * g + r;
        if (kk < nk) {
            bf16_t* kr = a.dk + (rbk + kk) * a.ldgk + h * hd;
            bf16_t* vr = a.dv + (rbk + kk) * a.ldgk + h * hd;
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt)
                if (dt * 16 < hd) {
                    kr[dt * 16 + c] = (bf16_t)adk[dt][r];
                    vr[dt * 16 + c] = (bf16_t)adv[dt][r];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
template <int HD, int NS>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char ldsKr[NS * CH * HD * 2], ldsKt[NS * CH * HD * 2], ldsVr[NS * CH * HD * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.y, bw = bh / a.H, h = bh % a.H, nq = a.nq, nk = a.nk, hd = a.hd;
    const int64_t rbq = (int64_t)bw * nq, rbk = (int64_t)bw * nk;
    const bf16_t* q = a.q + rbq * a.ldq + h * hd;
    const bf16_t* k = a.k + rbk * a.ldk + h * hd;
    const bf16_t* v = a.v + rbk * a.ldk + h * hd;
    const bf16_t* d_o = a.d_o + rbq * a.ldo + h * hd;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const bool q_ok = q0 + c < nq;
    const float lse_q = q_ok ? a.lse[(int64_t)bh * nq + q0 + c] : 0.f;
    bf16x8_t qf[HD / 32], dof[HD / 32];
    load_rows_as_b<HD>(q, a.ldq, q0, nq, lane, qf, hd);
    load_rows_as_b<HD>(d_o, a.ldo, q0, nq, lane, dof, hd);
    // delta[q] = sum_d dO[q][d] * O[q][d], computed here from the dO fragments the wave holds anyway (lane c + 16 g has
    // columns 8 g .. 8 g + 7 of every 32-column group of row q0 + c) and published for the dK/dV kernel, which is
    // launched after this one -- the separate delta pass (19 launches per step, a second read of dO and O) is gone.
    float dl_q = 0.f;
    {
        bf16x8_t of[HD / 32];
        load_rows_as_b<HD>(a.o + rbq * a.ldo + h * hd, a.ldo, q0, nq, lane, of, hd);
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) dl_q += (float)dof[ks][j] * (float)of[ks][j];
        dl_q += __shfl_xor(dl_q, 16, 64);
        dl_q += __shfl_xor(dl_q, 32, 64);
        if (blockIdx.x * 64 + wave * 16 + c < nq && g == 0) a.delta[(int64_t)bh * nq + q0 + c] = dl_q;
    }
    f32x4_t adq[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) adq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    RowChunk<HD, NS> pk, pv;
    fetch_rows<HD, NS>(k, a.ldk, 0, nk, tid, hd, pk);
    fetch_rows<HD, NS>(v, a.ldk, 0, nk, tid, hd, pv);
    for (int kb = 0; kb < nk; kb += NS * CH) {
        __syncthreads();
        put_rows<HD, NS, false>(ldsKr, tid, pk);
        put_rows<HD, NS, true>(ldsKt, tid, pk);
        put_rows<HD, NS, false>(ldsVr, tid, pv);
        __syncthreads();
        if (kb + NS * CH < nk) {
            fetch_rows<HD, NS>(k, a.ldk, kb + NS * CH, nk, tid, hd, pk);
            fetch_rows<HD, NS>(v, a.ldk, kb + NS * CH, nk, tid, hd, pv);
        }
        for (int sub = 0; sub < NS && kb + sub * CH < nk; ++sub) {
        const int kc = kb + sub * CH;
        const int so = sub * (CH * HD * 2);
        f32x4_t dS[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rc<HD>(ldsKr + so, 16 * t, ks, lane), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rc<HD>(ldsVr + so, 16 * t, ks, lane), dof[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // element [key = kc+16t+4g+r][query = q0+c]
                float ds = 0.f;
                if (q_ok && kc + 16 * t + 4 * g + r < nk)
                    ds = __expf(s[r] * a.scale - lse_q) * (dp[r] - dl_q) * a.scale;
                dS[t][r] = ds;
            }
        }
        const bf16x8_t dsf = pack_pair(dS[0], dS[1]);
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt)
            adq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, frag_tr_perm<HD>(ldsKt + so, dt, lane), adq[dt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qq = q0 + 4 * g + r;
        if (qq < nq) {
            bf16_t* qr = a.dq + (rbq + qq) * a.ldgq + h * hd;
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt)
                if (dt * 16 < hd) qr[dt * 16 + c] = (bf16_t)adq[dt][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Lean kernels (round 2, the default).  The step kernels above spend ~240 VALU instructions per 32-key step of a wave for 16
// MFMAs (measured: SQ_INSTS_VALU 1677 per wave of the window forward, VALU pipe 45 % busy, MFMA pipe 13 %): address
// arithmetic and bounds checks of the staging loads re-done every step, the key mask applied on every step, an online-
// softmax rescale of the whole output tile plus eight ds_bpermute broadcasts per step, per-element branches in backward.
// Same data flow (32 rows per step through LDS, S^T trick, P never in LDS), but:
//   * staging through raw buffer loads whose resource ends with the last valid row: rows past the end read as zeros with
//     no compare, the per-thread global / LDS offsets are computed once, the LDS blocks are double-buffered (one barrier
//     per step instead of two);
//   * zero rows make every mask of the backward kernels unnecessary (a padded key row contributes dS * 0, a padded query
//     row has dO = 0); the forward masks keys only in the last step;
//   * forward: the running maximum is only raised when a step's scores exceed it by more than 2^8 (all probabilities of a
//     row stay relative to one reference, the row sum uses the same reference, so the result is exact up to rounding);
//     the common step has no rescale, no cross-lane traffic, and the row sums come out of one extra MFMA against a ones
//     fragment -- already in the layout of the output tile, so the final 1 / l needs no broadcast either;
//   * exp2 with the scale folded into one FMA per score; two 16-row tiles per wave share every LDS fragment read.
// Measured and NOT kept (tools/op_bench.py attn, ViT-B bs 12, window fwd / dQ / dKdV 22.7 / 28.8 / 40.3 us as shipped):
//   * a "resident" form for the 196-token windows -- every row requested in the prologue, one tr-layout image per matrix
//     (57 KiB per workgroup), no barrier or global load in the loop: 51.8 / 58.8 / 56.8 us (the earlier whole-chunk
//     kernels, same idea with a heavier loop, had measured 37.6 us forward); padding the streaming kernels' LDS to the
//     same 2 workgroups per CU only costs 22.7 -> 26.6 us, so it is not the occupancy;
//   * one workgroup of seven waves x two tiles per window problem (keys staged once): 23.6 vs 23.1 us;
//   * forcing the dQ kernel to 128 VGPRs (4 waves per SIMD, 20 bytes of scratch): 34.0 vs 28.6 us;
//   * a second register set in the dQ kernel (blocks i + 1 and i + 2 in flight while block i is multiplied): 29.2 vs 28.6 us
//     on the windows, 60.9 vs 57.9 global -- the steps do not wait for memory.  What is left per SIMD is issue time: ~100
//     VALU instructions + 16 quarter-rate v_exp_f32 per step and wave, plus a prologue / epilogue of ~600 instructions per
//     wave (delta, fragment loads, 2-byte stores), at 4.5 waves per SIMD in two rounds.
// ------------------------------------------------------------------------------------------------
constexpr float LOG2E = 1.44269504089f, LN2 = 0.69314718056f;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// buffer resource over rows [0, n) x cols [0, cols) of a [*, ld] matrix of elem-byte elements: everything past the last
// valid element of row n - 1 is out of range (reads as zero)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void* base, int n, int ld, int cols, int elem) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, ((n - 1) * ld + cols) * elem, 0x00020000);
}

// per-thread loop invariants of staging one [CH x HD] block with 256 threads (16-byte pieces)
template <int HD> struct Stg {
    static constexpr int CPR = HD / 8, NCH = CH * CPR, NI = (NCH + 255) / 256;
    int lrc[NI], ltr[NI];
    bool live[NI];
    __device__ __forceinline__ void init(int tid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * 256, row = c / CPR, ch = c % CPR;
            live[i] = c < NCH;
            lrc[i] = rc_off<HD>(row, ch);
            ltr[i] = tr_off<HD>(row, ch * 2);
        }
    }
    // byte offsets of this thread's pieces within rows [0, CH) of a matrix with row stride ld; columns >= hd: never valid
    __device__ __forceinline__ void offsets(int tid, int ld, int hd, int (&voff)[NI]) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * 256, row = c / CPR, ch = c % CPR;
            voff[i] = (live[i] && ch * 8 < hd) ? (row * ld + ch * 8) * 2 : 0x40000000;
        }
    }
    __device__ __forceinline__ static void fetch(const __amdgpu_buffer_rsrc_t rs, const int (&voff)[NI], int base, u32x4v (&v)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i] + base, 0, 0);
    }
    __device__ __forceinline__ void put_rc(char* lds, const u32x4v (&v)[NI]) const {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (live[i]) *reinterpret_cast<u32x4v*>(lds + lrc[i]) = v[i];
    }
    __device__ __forceinline__ void put_tr(char* lds, const u32x4v (&v)[NI]) const {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (live[i]) *reinterpret_cast<u32x4v*>(lds + ltr[i]) = v[i];
    }
};

// max of eight MFMA results: v_max3_f32 directly (fmaxf would first canonicalise every operand with a v_max_f32 x, x).
// ONE asm block that starts with its own wait states: the compiler's hazard recognizer does not look into inline asm, and an
// XDL result read by a VALU instruction needs up to 19 wait states after the MFMA was issued (no hardware interlock) --
// without them the v_max3 directly behind the score MFMAs read the PREVIOUS key chunk's scores some of the time: a running
// maximum that lags, exp2 of a positive difference, outputs that differ from launch to launch and NaN once the scores are
// tens apart (seen in the one-tile 128-column form at ViT-H's neck shapes; every form had the pattern in its ISA).
__device__ __forceinline__ float max8(const f32x4_t& a, const f32x4_t& b) {
    float d, t0, t1;
    asm volatile("s_nop 15\n\ts_nop 3\n\t"
                 "v_max3_f32 %0, %3, %4, %5\n\t"
                 "v_max3_f32 %1, %6, %7, %8\n\t"
                 "v_max3_f32 %2, %9, %10, %10\n\t"
                 "v_max3_f32 %0, %0, %1, %2"
                 : "=&v"(d), "=&v"(t0), "=&v"(t1)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    return d;
}

// Workgroup -> (problem bh, block x of it).  a.gx == 0: the 2-D grid (blockIdx.y = problem).  a.gx > 0: a 1-D grid whose
// workgroups of ONE problem are eight ids apart -- consecutive ids go to consecutive XCDs, so all blocks of a (window,
// head) problem share an XCD and its L2: the K / V (or Q / dO) rows every block of the problem streams are fetched from
// HBM once instead of once per block (window problems have two blocks: 81 -> 58 MB read in the forward).
__device__ __forceinline__ void wg_problem(int gx, int nbh, int& bh, int& xb) {
    if (gx == 0) { bh = blockIdx.y; xb = blockIdx.x; return; }
    const int id = blockIdx.x, per = 8 * gx, grp = id / per, r = id - grp * per;
    const int cnt = min(8, nbh - grp * 8);          // problems in this group (the last group may be short)
    bh = grp * 8 + r % cnt;
    xb = r / cnt;
}

constexpr float RESCALE_TH = 8.0f;   // log2 units: probabilities relative to the reference maximum stay below 2^8

template <int HD, int QT, int HC>
// (HD = 64, two tiles per wave: FOUR workgroups per CU -- 126 registers through round 5 without asking; the round-6 epilogue takes
// 134 unless the bound is stated, and three workgroups per CU are 1.5 rounds of the window forward's 1152 instead of 1.125.
// Round 4: occupancy 5 = 96 registers + 120 bytes of scratch in the loop: 51.5 us (window) / 106 us (global) against 22.9 / 37.)
#ifndef VPU_ATTN_FWD_OCC
#define VPU_ATTN_FWD_OCC 4
#endif
__global__ __launch_bounds__(256, (HD == 64 && QT == 2) ? VPU_ATTN_FWD_OCC : 1) void attn_fwd_lean_kernel(const AttnArgs a) {
    constexpr int KS = HC / 32, DT = HC / 16;   // computed width HC <= image width HD (head dim 80 / 96: 96 of the 128 columns)
    using S = Stg<HD>;
    __shared__ __attribute__((aligned(16))) char ldsK[2][CH * HD * 2];
    __shared__ __attribute__((aligned(16))) char ldsV[2][CH * HD * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    int bh, xb;
    wg_problem(a.gx, a.nbh, bh, xb);
    const int bw = bh / a.H, h = bh % a.H, nq = a.nq, nk = a.nk, hd = a.hd;
    const int bwq = bw / max(a.qdiv, 1), bwk = bw / max(a.kdiv, 1);      // (split launches: which entry the operands come from)
    const int64_t rbq = (int64_t)bwq * nq, rbk = (int64_t)bwk * nk, rbqo = (int64_t)bw * nq, rbko = (int64_t)bw * nk;
    const int bhq = bwq * a.H + h;
    const bf16_t* q = a.q + rbq * a.ldq + h * hd;
    const bf16_t* k = a.k + rbk * a.ldk + h * hd;
    const bf16_t* v = a.v + rbk * a.ldk + h * hd;
    const int q0 = xb * (64 * QT) + wave * (16 * QT);
    const float sc2 = a.scale * LOG2E;
    bf16x8_t qf[QT][KS];
    float mref[QT];
    f32x4_t acc[QT][DT], accl[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        load_rows_as_bn<KS>(q, a.ldq, q0 + 16 * u, nq, lane, qf[u], hd);
        mref[u] = -INFINITY;
        accl[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;
    S st;
    st.init(tid);
    int voff[S::NI];
    st.offsets(tid, a.ldk, hd, voff);
    const __amdgpu_buffer_rsrc_t rsK = rows_rsrc(k, nk, a.ldk, hd, 2), rsV = rows_rsrc(v, nk, a.ldk, hd, 2);
    const int nch = (nk + CH - 1) / CH, cstep = CH * a.ldk * 2;
    u32x4v pk[S::NI], pv[S::NI];
    S::fetch(rsK, voff, 0, pk);
    S::fetch(rsV, voff, 0, pv);
    st.put_rc(ldsK[0], pk);
    st.put_tr(ldsV[0], pv);
    if (nch > 1) { S::fetch(rsK, voff, cstep, pk); S::fetch(rsV, voff, cstep, pv); }
    f32x4_t s[QT][2];
    float lm[QT];
    // one step = stage (barrier, restage, next fetch) -> scores -> [raise] -> probs_pv.  The steps run in TWO loops: the
    // common one never touches the output tiles outside the MFMAs (a conditional rescale inside one loop makes the register
    // allocator copy every tile on every step); a wave whose scores outgrow the reference leaves it for the second loop,
    // which rescales on every step.  The waves of a workgroup may be in different loops: each still passes one barrier
    // per step.
    auto stage = [&](int i) {
        __syncthreads();    // block i is visible; every wave is done with block i - 1 (the buffer block i + 1 goes into)
        if (i + 1 < nch) { st.put_rc(ldsK[(i + 1) & 1], pk); st.put_tr(ldsV[(i + 1) & 1], pv); }
        if (i + 2 < nch) { S::fetch(rsK, voff, (i + 2) * cstep, pk); S::fetch(rsV, voff, (i + 2) * cstep, pv); }
    };
    auto scores = [&](int i) {
        const char* sK = ldsK[i & 1];
        const int kc = i * CH;
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) s[u][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kfr = frag_rc<HD>(sK, 16 * t, ks, lane);
#pragma unroll
                for (int u = 0; u < QT; ++u)
                    s[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[u][ks], s[u][t], 0, 0, 0);
            }
        if (kc + CH > nk) {   // the last step: keys past the end (block-uniform)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kc + 16 * t + 4 * g + r >= nk) {
#pragma unroll
                        for (int u = 0; u < QT; ++u) s[u][t][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int u = 0; u < QT; ++u) lm[u] = max8(s[u][0], s[u][1]) * sc2;
    };
    auto outgrown = [&]() {
        bool need = false;
#pragma unroll
        for (int u = 0; u < QT; ++u) need = need || (lm[u] > mref[u] + RESCALE_TH);
        return __builtin_amdgcn_ballot_w64(need) != 0;
    };
    auto raise = [&]() {   // reference <- this step's maximum where it is more than 2^8 above; everything rescaled
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            float xm = lm[u];
            xm = fmaxf(xm, __shfl_xor(xm, 16, 64));
            xm = fmaxf(xm, __shfl_xor(xm, 32, 64));
            const float mnew = (xm > mref[u] + RESCALE_TH) ? xm : mref[u];
            const float alpha = __builtin_amdgcn_exp2f(mref[u] - mnew);     // (of query c: the lane's own, see probs_pv)
            mref[u] = mnew;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accl[u][r] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc[u][dt][r] *= alpha;
            }
        }
    };
    auto probs_pv = [&](int i) {
        const char* sV = ldsV[i & 1];
        bf16x8_t pf[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const float nm = -mref[u];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[u][t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][t][r], sc2, nm));
            // round 6: the products are taken TRANSPOSED (O^T = V^T P^T, l^T = 1 P^T: the same operand registers, swapped) -- a lane
            // then holds four consecutive columns of ONE query (c), its running maximum and row sum are that query's own (no
            // broadcast in raise() or at the end) and the output leaves in 8-byte pieces instead of 2-byte ones
            pf[u] = pack_pair(s[u][0], s[u][1]);
            accl[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[u], accl[u], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x8_t vfr = frag_tr_perm<HD>(sV, dt, lane);
#pragma unroll
            for (int u = 0; u < QT; ++u) acc[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, pf[u], acc[u][dt], 0, 0, 0);
        }
    };
    stage(0); scores(0); raise();      // the first step sets the reference (alpha = 0 on all-zero tiles)
    int i = 0;
    bool go;
    do {    // single exit, the output tiles only change in the MFMAs: they accumulate in place
        probs_pv(i);
        ++i;
        go = i < nch;
        if (go) { stage(i); scores(i); go = !outgrown(); }
    } while (go);
    while (i < nch) {   // left the common loop with the scores of step i in hand
        raise(); probs_pv(i);
        ++i;
        if (i < nch) { stage(i); scores(i); }
    }
    // the wave's [16 QT queries][HD] output tile goes through the K / V buffers (nobody reads them any more) and leaves in whole
    // rows, 16 bytes per lane: the 2-byte stores of rounds 2-5 were 32 half-line write instructions per wave at the end of a
    // workgroup's life.  16-byte chunk ch of row r sits at chunk ch ^ (r & 7) of its row (bank spread; r & 3 in the 4-chunk rows of HD = 32).
    __syncthreads();
    {
        typedef __bf16 bf16x4o_t __attribute__((ext_vector_type(4)));
        constexpr int RB = HD * 2, NCHK = HD / 8, SW = NCHK >= 8 ? 7 : NCHK - 1;
        char* const stg = (wave < 2 ? &ldsK[0][0] : &ldsV[0][0]) + (wave & 1) * (CH * HD * 2);
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const float l = accl[u][0], il = __builtin_amdgcn_rcpf(l);   // (the quotient is rounded to bf16 anyway)
            const int qq = q0 + 16 * u + c;
            if (g == 0 && qq < nq) a.lse[(int64_t)bh * nq + qq] = (mref[u] + __builtin_amdgcn_logf(l)) * LN2;
            const int row = 16 * u + c;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4o_t w4;
#pragma unroll
                for (int r = 0; r < 4; ++r) w4[r] = (bf16_t)(acc[u][dt][r] * il);
                const int chk = (2 * dt + (g >> 1)) ^ (row & SW);
                *reinterpret_cast<bf16x4o_t*>(stg + row * RB + chk * 16 + (g & 1) * 8) = w4;
            }
        }
        // (same wave: the LDS writes above are ordered before these reads)
#pragma unroll
        for (int i = 0; i < 16 * QT * NCHK / 64; ++i) {
            const int idx = i * 64 + lane, row = idx / NCHK, ch = idx % NCHK;
            const u32x4v v16 = *reinterpret_cast<const u32x4v*>(stg + row * RB + ((ch ^ (row & SW)) * 16));
            const int qq = q0 + row;
            if (qq < nq && ch * 8 < hd) *reinterpret_cast<u32x4v*>(a.out + (rbqo + qq) * a.ldo + h * hd + ch * 8) = v16;
        }
    }
}

// dQ (and delta): QT query tiles per wave, keys streamed
template <int HD, int QT, int HC>
__global__ __launch_bounds__(256) void attn_bwd_dq_lean_kernel(const AttnArgs a) {
    constexpr int KS = HC / 32, DT = HC / 16;
    using S = Stg<HD>;
    __shared__ __attribute__((aligned(16))) char ldsKr[2][CH * HD * 2], ldsKt[2][CH * HD * 2], ldsVr[2][CH * HD * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    int bh, xb;
    wg_problem(a.gx, a.nbh, bh, xb);
    const int bw = bh / a.H, h = bh % a.H, nq = a.nq, nk = a.nk, hd = a.hd;
    const int bwq = bw / max(a.qdiv, 1), bwk = bw / max(a.kdiv, 1);      // (split launches: which entry the operands come from)
    const int64_t rbq = (int64_t)bwq * nq, rbk = (int64_t)bwk * nk, rbqo = (int64_t)bw * nq, rbko = (int64_t)bw * nk;
    const int bhq = bwq * a.H + h;
    const bf16_t* q = a.q + rbq * a.ldq + h * hd;
    const bf16_t* k = a.k + rbk * a.ldk + h * hd;
    const bf16_t* v = a.v + rbk * a.ldk + h * hd;
    const bf16_t* d_o = a.d_o + rbq * a.ldo + h * hd;
    const int q0 = xb * (64 * QT) + wave * (16 * QT);
    const float sc2 = a.scale * LOG2E;
    bf16x8_t qf[QT][KS], dof[QT][KS];
    float nl2[QT], nds[QT];    // -lse * log2 e, -delta * scale of query q0 + 16 u + c
    f32x4_t adq[QT][DT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int qu = q0 + 16 * u;
        const bool q_ok = qu + c < nq;
        nl2[u] = q_ok ? -a.lse[(int64_t)bhq * nq + qu + c] * LOG2E : 0.f;
        load_rows_as_bn<KS>(q, a.ldq, qu, nq, lane, qf[u], hd);
        load_rows_as_bn<KS>(d_o, a.ldo, qu, nq, lane, dof[u], hd);
        // delta[q] = sum_d dO[q][d] * O[q][d] from the dO fragments the wave holds anyway; published for the dK/dV kernel
        bf16x8_t of[KS];
        load_rows_as_bn<KS>(a.o + rbq * a.ldo + h * hd, a.ldo, qu, nq, lane, of, hd);
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)dof[u][ks][j] * (float)of[ks][j];
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (q_ok && g == 0) a.delta[(int64_t)bhq * nq + qu + c] = dl;      // (qdiv > 1: the same value from every key range)
        nds[u] = -dl * a.scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) adq[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    S st;
    st.init(tid);
    int voff[S::NI];
    st.offsets(tid, a.ldk, hd, voff);
    const __amdgpu_buffer_rsrc_t rsK = rows_rsrc(k, nk, a.ldk, hd, 2), rsV = rows_rsrc(v, nk, a.ldk, hd, 2);
    const int nch = (nk + CH - 1) / CH, cstep = CH * a.ldk * 2;
    u32x4v pk[S::NI], pv[S::NI];
    S::fetch(rsK, voff, 0, pk);
    S::fetch(rsV, voff, 0, pv);
    st.put_rc(ldsKr[0], pk); st.put_tr(ldsKt[0], pk); st.put_rc(ldsVr[0], pv);
    if (nch > 1) { S::fetch(rsK, voff, cstep, pk); S::fetch(rsV, voff, cstep, pv); }
    for (int i = 0; i < nch; ++i) {
        __syncthreads();
        if (i + 1 < nch) {
            const int nb = (i + 1) & 1;
            st.put_rc(ldsKr[nb], pk); st.put_tr(ldsKt[nb], pk); st.put_rc(ldsVr[nb], pv);
        }
        if (i + 2 < nch) { S::fetch(rsK, voff, (i + 2) * cstep, pk); S::fetch(rsV, voff, (i + 2) * cstep, pv); }
        const char* sKr = ldsKr[i & 1];
        const char* sKt = ldsKt[i & 1];
        const char* sVr = ldsVr[i & 1];
        const int kc = i * CH;
        f32x4_t dS[QT][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t s[QT], dp[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) { s[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[u] = s[u]; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t kfr = frag_rc<HD>(sKr, 16 * t, ks, lane), vfr = frag_rc<HD>(sVr, 16 * t, ks, lane);
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[u][ks], s[u], 0, 0, 0);
                    dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, dof[u][ks], dp[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)   // element [key = kc + 16 t + 4 g + r][query = q0 + 16 u + c]
                    dS[u][t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], sc2, nl2[u])) * __builtin_fmaf(dp[u][r], a.scale, nds[u]);
        }
        if (kc + CH > nk) {   // last step: a padded key has K = 0, so its dS only has to be finite
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kc + 16 * t + 4 * g + r >= nk) {
#pragma unroll
                        for (int u = 0; u < QT; ++u) dS[u][t][r] = 0.f;
                    }
        }
        bf16x8_t dsf[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) dsf[u] = pack_pair(dS[u][0], dS[u][1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x8_t ktf = frag_tr_perm<HD>(sKt, dt, lane);
#pragma unroll
            for (int u = 0; u < QT; ++u) adq[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf[u], ktf, adq[u][dt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = q0 + 16 * u + 4 * g + r;
            if (qq < nq) {
                bf16_t* qr = a.dq + (rbqo + qq) * a.ldgq + h * hd + c;
                if (hd == HC) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) qr[dt * 16] = (bf16_t)adq[u][dt][r];
                } else {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        if (dt * 16 < hd) qr[dt * 16] = (bf16_t)adq[u][dt][r];
                }
            }
        }
}

// dK, dV: KT key tiles per wave, queries streamed (needs nq % 4 == 0: the log-sum-exp / delta values of four consecutive
// queries are one 16-byte buffer load, zeros past the end)
template <int HD, int KT, int HC>
__global__ __launch_bounds__(256, 3) void attn_bwd_dkdv_lean_kernel(const AttnArgs a) {
    constexpr int KS = HC / 32, DT = HC / 16;
    using S = Stg<HD>;
    __shared__ __attribute__((aligned(16))) char ldsQr[2][CH * HD * 2], ldsQt[2][CH * HD * 2];
    __shared__ __attribute__((aligned(16))) char ldsOr[2][CH * HD * 2], ldsOt[2][CH * HD * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    int bh, xb;
    wg_problem(a.gx, a.nbh, bh, xb);
    const int bw = bh / a.H, h = bh % a.H, nq = a.nq, nk = a.nk, hd = a.hd;
    const int bwq = bw / max(a.qdiv, 1), bwk = bw / max(a.kdiv, 1);      // (split launches: which entry the operands come from)
    const int64_t rbq = (int64_t)bwq * nq, rbk = (int64_t)bwk * nk, rbqo = (int64_t)bw * nq, rbko = (int64_t)bw * nk;
    const int bhq = bwq * a.H + h;
    const bf16_t* q = a.q + rbq * a.ldq + h * hd;
    const bf16_t* k = a.k + rbk * a.ldk + h * hd;
    const bf16_t* v = a.v + rbk * a.ldk + h * hd;
    const bf16_t* d_o = a.d_o + rbq * a.ldo + h * hd;
    const int key0 = xb * (64 * KT) + wave * (16 * KT);
    const float sc2 = a.scale * LOG2E;
    bf16x8_t kf[KT][KS], vf[KT][KS];
    f32x4_t adk[KT][DT], adv[KT][DT];
#pragma unroll
    for (int u = 0; u < KT; ++u) {
        load_rows_as_bn<KS>(k, a.ldk, key0 + 16 * u, nk, lane, kf[u], hd);
        load_rows_as_bn<KS>(v, a.ldk, key0 + 16 * u, nk, lane, vf[u], hd);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { adk[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; adv[u][dt] = adk[u][dt]; }
    }
    S st;
    st.init(tid);
    int voq[S::NI], voo[S::NI];
    st.offsets(tid, a.ldq, hd, voq);
    st.offsets(tid, a.ldo, hd, voo);
    const __amdgpu_buffer_rsrc_t rsQ = rows_rsrc(q, nq, a.ldq, hd, 2), rsO = rows_rsrc(d_o, nq, a.ldo, hd, 2);
    const __amdgpu_buffer_rsrc_t rsL = rows_rsrc(reinterpret_cast<const char*>(a.lse + (int64_t)bhq * nq), 1, 0, nq, 4);
    const __amdgpu_buffer_rsrc_t rsD = rows_rsrc(reinterpret_cast<const char*>(a.delta + (int64_t)bhq * nq), 1, 0, nq, 4);
    const int nch = (nq + CH - 1) / CH, qstep = CH * a.ldq * 2, ostep = CH * a.ldo * 2;
    u32x4v pq[S::NI], po[S::NI];
    S::fetch(rsQ, voq, 0, pq);
    S::fetch(rsO, voo, 0, po);
    st.put_rc(ldsQr[0], pq); st.put_tr(ldsQt[0], pq); st.put_rc(ldsOr[0], po); st.put_tr(ldsOt[0], po);
    if (nch > 1) { S::fetch(rsQ, voq, qstep, pq); S::fetch(rsO, voo, ostep, po); }
    for (int i = 0; i < nch; ++i) {
        __syncthreads();
        if (i + 1 < nch) {
            const int nb = (i + 1) & 1;
            st.put_rc(ldsQr[nb], pq); st.put_tr(ldsQt[nb], pq); st.put_rc(ldsOr[nb], po); st.put_tr(ldsOt[nb], po);
        }
        if (i + 2 < nch) { S::fetch(rsQ, voq, (i + 2) * qstep, pq); S::fetch(rsO, voo, (i + 2) * ostep, po); }
        const int b = i & 1, qc = i * CH;
        f32x4_t nl2[2], nds[2];   // -lse * log2 e, -delta * scale of queries qc + 16 t + 4 g + r (zeros past the end)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int off = (qc + 16 * t + 4 * g) * 4;
            const f32x4_t lv = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsL, off, 0, 0));
            const f32x4_t dv = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsD, off, 0, 0));
            nl2[t] = lv * (-LOG2E);
            nds[t] = dv * (-a.scale);
        }
        f32x4_t P[KT][2], dS[KT][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t s[KT], dp[KT];
#pragma unroll
            for (int u = 0; u < KT; ++u) { s[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[u] = s[u]; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t qfr = frag_rc<HD>(ldsQr[b], 16 * t, ks, lane), ofr = frag_rc<HD>(ldsOr[b], 16 * t, ks, lane);
#pragma unroll
                for (int u = 0; u < KT; ++u) {
                    s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[u][ks], s[u], 0, 0, 0);
                    dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[u][ks], dp[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < KT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // element [query = qc + 16 t + 4 g + r][key = key0 + 16 u + c]
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], sc2, nl2[t][r]));
                    P[u][t][r] = p;
                    dS[u][t][r] = p * __builtin_fmaf(dp[u][r], a.scale, nds[t][r]);
                }
        }
        bf16x8_t pf[KT], dsf[KT];
#pragma unroll
        for (int u = 0; u < KT; ++u) { pf[u] = pack_pair(P[u][0], P[u][1]); dsf[u] = pack_pair(dS[u][0], dS[u][1]); }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x8_t otf = frag_tr_perm<HD>(ldsOt[b], dt, lane), qtf = frag_tr_perm<HD>(ldsQt[b], dt, lane);
#pragma unroll
            for (int u = 0; u < KT; ++u) {
                adv[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[u], otf, adv[u][dt], 0, 0, 0);
                adk[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf[u], qtf, adk[u][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = key0 + 16 * u + 4 * g + r;
            if (kk < nk) {
                bf16_t* kr = a.dk + (rbko + kk) * a.ldgk + h * hd + c;
                bf16_t* vr = a.dv + (rbko + kk) * a.ldgk + h * hd + c;
                if (hd == HC) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        kr[dt * 16] = (bf16_t)adk[u][dt][r];
                        vr[dt * 16] = (bf16_t)adv[u][dt][r];
                    }
                } else {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        if (dt * 16 < hd) {
                            kr[dt * 16] = (bf16_t)adk[u][dt][r];
                            vr[dt * 16] = (bf16_t)adv[u][dt][r];
                        }
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ one-pass backward
// Window-sized self-attention problems (n <= 256 tokens, head dim 64: the ViT-B / ViT-L blocks, 196 tokens): the two kernels
// above compute S and dP twice (once per orientation) and fetch Q, K, V, dO twice each -- and their two workgroups per
// (window, head) land on different XCDs, so the second fetch is not an L2 hit (measured 207 MB read for 72 MB of operands).
// Here ONE workgroup of eight waves owns a (window, head) problem and every score is computed once:
//   * the whole problem is resident: Q, dO and K (100 KB at 196 tokens) are requested in the prologue -- every global load
//     of the workgroup is in flight at once, the loop itself touches LDS only (the first version streamed 32-query blocks
//     with a two-block look-ahead like the kernels above: with one workgroup per CU nothing covers the latency a block
//     arrives with, 2 us per block against 0.7 us of work; 65 us per launch, no faster than the two kernels);
//   * wave w owns one or two 16-key tiles (13 tiles for 196 keys: 2,2,2,2,2,1,1,1), K / V fragments in registers, and
//     accumulates dK / dV for them over the 32-query blocks (as attn_bwd_dkdv_lean does);
//   * S = Q K^T comes out as [query 4g+r][key c]: P and dS in that layout are the A operands of dV += P^T dO and
//     dK += dS^T Q for free.  dQ += dS K sums over KEYS, i.e. over the lane index of that layout, so dS goes through
//     LDS once: each lane writes its four consecutive queries of one key as ONE 8-byte unit of a [key][32 queries] image
//     and the dQ product reads it back with ds_read_b64_tr_b16 as the A fragment [query][32 keys];
//   * after the block's barrier all keys' dS are there, so wave w computes the whole sum over keys for ONE of the eight
//     16 x 16 tiles of the block's dQ (2 query tiles x 4 column tiles; its K fragments stay in registers) and stores it:
//     no cross-wave reduction, no atomics.  The dQ of block i - 1 is computed in iteration i: one barrier per block.
//   * delta = rowsum(dO * O) is computed while dO is staged (the same threads fetch the matching O piece).
// One image per matrix, in the transposing-read layout; the row-fragment reads (ds_read_b128) of the score products use it
// too (two-way bank conflicts there instead of none: the LDS is not what bounds this kernel).
constexpr int WIN_MAX_KB = 8;                                   // 32-row blocks: n <= 256
constexpr int WIN_ROWS = WIN_MAX_KB * 32;
// LDS of a problem of NKB 32-row blocks: Q, dO, K images [32 NKB][64] bf16, two dS images [32 NKB keys + 16 spare rows][32 queries],
// -delta * scale and -lse * log2 e per query
constexpr int win_lds(int nkb) { return 3 * (32 * nkb * 64 * 2) + 2 * ((32 * nkb + 16) * 64) + 2 * (32 * nkb) * 4; }

// dS image [key][32 queries]: 64-byte rows of eight 8-byte units.  unit ^= {bit 2: row bit 2, bit 1: row bit 3, bit 0: row bit
// 1}: the 16 consecutive rows of a ds_write_b64 lane group fill 16 distinct 8-byte slots of the 128-byte bank row, the 8
// consecutive rows x 4 units of a ds_read_b64_tr_b16 half fill the 256-byte bank row exactly once.
__device__ __forceinline__ int ds_swz(int row) { return (((row >> 2) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 1) & 1); }
// [query][32 keys] fragment (k order permuted as frag_tr_perm's) out of the image: queries 16 t .. 16 t + 15
__device__ __forceinline__ bf16x8_t frag_ds(const char* lds, int t, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int r0 = 4 * g + q, u = ((t * 4 + p) ^ ds_swz(r0)) << 3;     // (rows r0 and 16 + r0 share the swizzle)
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + r0 * 64 + u));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + (16 + r0) * 64 + u));
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}
// A fragment (16 rows x 32 columns, ds_read_b128) out of an image in the transposing-read layout
template <int HD> __device__ __forceinline__ bf16x8_t frag_rc_tr(const char* lds, int row16, int ks, int lane) {
    const uint4 v = *reinterpret_cast<const uint4*>(lds + tr_off<HD>(row16 + (lane & 15), 2 * (ks * 4 + (lane >> 4))));
    return __builtin_bit_cast(bf16x8_t, v);
}
// frag_rc_tr as an ext-vector load (through HIP's uint4 struct the load carries TBAA info, and hipcc then puts s_waitcnt
// vmcnt(0) in front of it while an LDS-DMA is pending)
__device__ __forceinline__ bf16x8_t frag_rc_trv(const char* lds, int row16, int ks, int lane) {
    const u32x4v v = *reinterpret_cast<const u32x4v*>(lds + tr_off<64>(row16 + (lane & 15), 2 * (ks * 4 + (lane >> 4))));
    return __builtin_bit_cast(bf16x8_t, v);
}
typedef __attribute__((address_space(3))) void* lds_vptr3;
typedef __bf16 bf16x4w_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x4w_t pack4(const f32x4_t& a) {
    bf16x4w_t w;
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = (bf16_t)a[r];
    return w;
}

// Round 6.  The loop of the round-3 form was a chain: per 32-query block, fragment reads -> S / dP -> exp -> dS image -> barrier ->
// transposing reads -> dV / dK -> NKB x (two transposing reads, wait, ONE dQ MFMA) behind runtime branches, 1.63 us per block for
// 0.6 us of MFMA work (`tools/win_dbg_job.sh`: the loop 11.4 of a problem's 21.4 us).  Now NKB is a template parameter (no branch in
// the dQ sum: all its reads are issued with the dV / dK operands right after the barrier), the scores of block i + 1 share the basic
// block with the gradients of block i (independent MFMA chains and the exp / conversion arithmetic for the scheduler to interleave),
// and all three gradients leave through SWAPPED MFMA operands: a lane then holds four consecutive head-dim columns of one row, one
// 8-byte store instead of four 2-byte ones (64 -> 16 store instructions per wave for dK / dV, 4 -> 1 per block for dQ).
template <int NKB>
__global__ __launch_bounds__(512) void attn_bwd_win_kernel(const AttnArgs a) {
    constexpr int HD = 64, KS = 2, DT = 4, ROWS = 32 * NKB, NP = (NKB + 1) / 2;   // NP: 16-byte pieces per thread and matrix
    constexpr int LDS_M = ROWS * HD * 2, LDS_S1 = (ROWS + 16) * 64;
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    char* const ldsQ = wlds;
    char* const ldsO = wlds + LDS_M;          // dO
    char* const ldsK = wlds + 2 * LDS_M;
    char* const ldsS = wlds + 3 * LDS_M;      // [2][keys][32 queries] bf16, transposing-read layout
    float* const ldsD = reinterpret_cast<float*>(ldsS + 2 * LDS_S1);  // -delta * scale per query
    float* const ldsL = ldsD + ROWS;                                  // -lse * log2 e per query
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.x, bw = bh / a.H, h = bh % a.H, n = a.nq;
    const int64_t rb = (int64_t)bw * n;
    const bf16_t* q = a.q + rb * a.ldq + h * HD;
    const bf16_t* k = a.k + rb * a.ldk + h * HD;
    const bf16_t* v = a.v + rb * a.ldk + h * HD;
    const bf16_t* o = a.o + rb * a.ldo + h * HD;
    const bf16_t* d_o = a.d_o + rb * a.ldo + h * HD;
    const float sc2 = a.scale * LOG2E;
    const int NT = (n + 15) >> 4;             // (the launcher guarantees (NT + 1) >> 1 == NKB)
    // key tiles of this wave: NT = 8 base + rem, the first rem waves take one more
    const int base = NT >> 3, rem = NT & 7;
    const int nu = base + (wave < rem ? 1 : 0), tile0 = wave * base + (wave < rem ? wave : rem);
    const int my_t = wave >> 2, my_dt = wave & 3;     // this wave's tile of a block's dQ: queries 16 my_t.., columns 16 my_dt..
#ifdef VPU_WIN_STAMPS
    const unsigned t_start = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    // ---- prologue: every global load of the problem is requested before the first wait
    bf16x8_t kf[2][KS], vf[2][KS];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int key0 = (u < nu) ? (tile0 + u) * 16 : n;     // (an unowned slot reads nothing: rows >= n are zeros)
        load_rows_as_bn<KS>(k, a.ldk, key0, n, lane, kf[u], HD);
        load_rows_as_bn<KS>(v, a.ldk, key0, n, lane, vf[u], HD);
    }
    {
        const __amdgpu_buffer_rsrc_t rsQ = rows_rsrc(q, n, a.ldq, HD, 2), rsK = rows_rsrc(k, n, a.ldk, HD, 2);
        const __amdgpu_buffer_rsrc_t rsG = rows_rsrc(d_o, n, a.ldo, HD, 2), rsO = rows_rsrc(o, n, a.ldo, HD, 2);
        const __amdgpu_buffer_rsrc_t rsL = rows_rsrc(reinterpret_cast<const char*>(a.lse + (int64_t)bh * n), 1, 0, n, 4);
        u32x4v pq[NP], pk[NP], pg[NP], po[NP];
        const int ch = tid & 7, row0 = tid >> 3;      // piece i: row row0 + 64 i, 16-byte chunk ch
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = row0 + 64 * i;
            const bool live = row < ROWS;
            pq[i] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, live ? (row * a.ldq + ch * 8) * 2 : 0x40000000, 0, 0);
            pk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, live ? (row * a.ldk + ch * 8) * 2 : 0x40000000, 0, 0);
            pg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsG, live ? (row * a.ldo + ch * 8) * 2 : 0x40000000, 0, 0);
            po[i] = __builtin_amdgcn_raw_buffer_load_b128(rsO, live ? (row * a.ldo + ch * 8) * 2 : 0x40000000, 0, 0);
        }
        const float pl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, tid < ROWS ? tid * 4 : 0x40000000, 0, 0));
        // the dS rows nobody writes: the odd sixteenth tile of the last 32-key block, in both images (the dQ sum reads whole blocks)
        if ((NT & 1) && tid < 128)
            *reinterpret_cast<u32x4v*>(ldsS + (tid >> 6) * LDS_S1 + NT * 16 * 64 + (tid & 63) * 16) = (u32x4v){0u, 0u, 0u, 0u};
        if (tid < ROWS) ldsL[tid] = -pl * LOG2E;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = row0 + 64 * i;
            if (row < ROWS) {      // (wave-uniform for whole waves: a wave covers 8 consecutive rows)
                const int off = tr_off<HD>(row, ch * 2);
                *reinterpret_cast<u32x4v*>(ldsQ + off) = pq[i];
                *reinterpret_cast<u32x4v*>(ldsK + off) = pk[i];
                *reinterpret_cast<u32x4v*>(ldsO + off) = pg[i];
                // delta of the row: eight consecutive lanes hold one row of dO and of O
                const bf16x8_t x = __builtin_bit_cast(bf16x8_t, pg[i]), y = __builtin_bit_cast(bf16x8_t, po[i]);
                float dl = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)x[j] * (float)y[j];
                dl += __shfl_xor(dl, 1, 64);
                dl += __shfl_xor(dl, 2, 64);
                dl += __shfl_xor(dl, 4, 64);
                if (ch == 0) ldsD[row] = -dl * a.scale;
            }
        }
    }
    // where this lane's dS of slot u goes: key row 16 (tile0 + u) + c of the [key][32 queries] image -- an unowned slot (the
    // loop has no divergent branches: every wave computes two tiles, an unowned one on zero K / V) writes to the spare rows
    int srow[2], smask[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int krow = (u < nu ? (tile0 + u) * 16 : ROWS) + c;
        srow[u] = krow * 64;
        smask[u] = ds_swz(krow);
    }
    // dK^T, dV^T of this wave's key tiles: [16 head-dim columns of dt][16 keys], i.e. lane (g, c): key c, columns 16 dt + 4 g + r
    f32x4_t adk[2][DT], adv[2][DT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { adk[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; adv[u][dt] = adk[u][dt]; }
    __syncthreads();                 // the images are complete
    bf16x8_t ktf[NKB];               // K[32 keys of block kb][16 columns of my_dt]: the transposed operand of the dQ product
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) ktf[kb] = frag_tr_perm<HD>(ldsK + kb * 32 * HD * 2, my_dt, lane);
    // scores of query block i: P and dS as MFMA operands for this wave's key tiles, dS also into the block's image
    auto scores = [&](int i, bf16x8_t (&pf)[2], bf16x8_t (&dsf)[2]) {
        const int qc = i * CH;
        const char* sQ = ldsQ + qc * HD * 2;
        const char* sO = ldsO + qc * HD * 2;
        char* const sS = ldsS + (i & 1) * LDS_S1;
        f32x4_t P[2][2], dS[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4_t nl2 = *reinterpret_cast<const f32x4_t*>(ldsL + qc + 16 * t + 4 * g);
            const f32x4_t nds = *reinterpret_cast<const f32x4_t*>(ldsD + qc + 16 * t + 4 * g);
            f32x4_t s[2], dp[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { s[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[u] = s[u]; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t qfr = frag_rc_tr<HD>(sQ, 16 * t, ks, lane), ofr = frag_rc_tr<HD>(sO, 16 * t, ks, lane);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[u][ks], s[u], 0, 0, 0);
                    dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[u][ks], dp[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {   // element [query = qc + 16 t + 4 g + r][key = 16 (tile0 + u) + c]
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], sc2, nl2[r]));
                    P[u][t][r] = p;
                    dS[u][t][r] = p * __builtin_fmaf(dp[u][r], a.scale, nds[r]);
                }
                // this lane's four queries of key row 16 (tile0 + u) + c: one 8-byte unit of the [key][query] image
                *reinterpret_cast<bf16x4w_t*>(sS + srow[u] + ((4 * t + g) ^ smask[u]) * 8) = pack4(dS[u][t]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) { pf[u] = pack_pair(P[u][0], P[u][1]); dsf[u] = pack_pair(dS[u][0], dS[u][1]); }
    };
    const __amdgpu_buffer_rsrc_t rsDQ = rows_rsrc(a.dq + rb * a.ldgq + h * HD, n, a.ldgq, HD, 2);
    const int dq_ld2 = a.ldgq * 2, dq_col2 = (16 * my_dt + 4 * g) * 2;
    // gradients of query block i: dV^T += dO^T P, dK^T += Q^T dS for this wave's key tiles (operands in registers / the Q, dO images),
    // and this wave's 16 x 16 tile of the block's dQ^T = K^T dS^T, summed over ALL keys from the block's dS image
    auto grads = [&](int i, const bf16x8_t (&pf)[2], const bf16x8_t (&dsf)[2]) {
        const int qc = i * CH;
        const char* sQ = ldsQ + qc * HD * 2;
        const char* sO = ldsO + qc * HD * 2;
        const char* sS = ldsS + (i & 1) * LDS_S1;
        bf16x8_t otf[DT], qtf[DT], af[NKB];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { otf[dt] = frag_tr_perm<HD>(sO, dt, lane); qtf[dt] = frag_tr_perm<HD>(sQ, dt, lane); }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) af[kb] = frag_ds(sS + kb * 32 * 64, my_t, lane);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                adv[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otf[dt], pf[u], adv[u][dt], 0, 0, 0);
                adk[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf[dt], dsf[u], adk[u][dt], 0, 0, 0);
            }
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[kb], af[kb], acc, 0, 0, 0);
        // lane (g, c): query c of the tile, columns 16 my_dt + 4 g ..; a buffer store (rows >= n fall outside the resource and are
        // dropped): no branch, so that the scores of the next block share this basic block
        const int qq = qc + 16 * my_t + c;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2w, pack4(acc)), rsDQ, qq < n ? qq * dq_ld2 + dq_col2 : 0x40000000, 0, 0);
    };
#ifdef VPU_WIN_STAMPS
    const __amdgpu_buffer_rsrc_t rsDbg = rows_rsrc(a.delta + (int64_t)bh * n, 1, 0, n, 4);
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)__builtin_amdgcn_s_memtime(), rsDbg, lane == 0 ? (wave * 16 + 14) * 4 : 0x40000000, 0, 0);
#endif
    bf16x8_t pf[2], dsf[2];
    scores(0, pf, dsf);
#pragma nounroll
    for (int i = 0; i + 1 < NKB; ++i) {
        __syncthreads();     // the dS image of block i is complete (and everybody has finished reading the image of block i - 1)
#ifdef VPU_WIN_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)__builtin_amdgcn_s_memtime(), rsDbg, lane == 0 ? (wave * 16 + i * 2) * 4 : 0x40000000, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#endif
        bf16x8_t pn[2], dn[2];
        grads(i, pf, dsf);
        scores(i + 1, pn, dn);
#pragma unroll
        for (int u = 0; u < 2; ++u) { pf[u] = pn[u]; dsf[u] = dn[u]; }
#ifdef VPU_WIN_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the iteration's LDS traffic is complete
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)__builtin_amdgcn_s_memtime(), rsDbg, lane == 0 ? (wave * 16 + i * 2 + 1) * 4 : 0x40000000, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    __syncthreads();
    grads(NKB - 1, pf, dsf);
#ifdef VPU_WIN_STAMPS
    __builtin_amdgcn_raw_buffer_store_b32(t_start, rsDbg, lane == 0 ? (wave * 16 + 15) * 4 : 0x40000000, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)__builtin_amdgcn_s_memtime(), rsDbg, lane == 0 ? (wave * 16 + 13) * 4 : 0x40000000, 0, 0);
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (u < nu) {
            const int kk = (tile0 + u) * 16 + c;      // lane (g, c): key c of the tile, columns 16 dt + 4 g ..
            if (kk < n) {
                bf16_t* kr = a.dk + (rb + kk) * a.ldgk + h * HD + 4 * g;
                bf16_t* vr = a.dv + (rb + kk) * a.ldgk + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    *reinterpret_cast<bf16x4w_t*>(kr + dt * 16) = pack4(adk[u][dt]);
                    *reinterpret_cast<bf16x4w_t*>(vr + dt * 16) = pack4(adv[u][dt]);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ one-pass backward, persistent
// Round 6 ("onepass" = 3, the default where it applies: 65 <= n <= 224).  Time stamps in the kernel above (`tools/win_stamps.py`)
// split a problem's 19 us into 5.9 us until the first score (every CU requests its 128 KB in the same microsecond: the chip's
// memory system, not the CU, sets that wait), 1.2 us for the first block's scores, 6 x 1.4 us for the loop (the SIMD's vector issue
// port: MFMA 8, exp 8, fma 4, cvt 4-5 cycles each, both waves of a SIMD) and 2-3 us of gradient stores -- and a launch of 576
// problems is three such rounds in lock step.  Here a workgroup walks problems p, p + grid, ... and fetches the NEXT problem while it
// computes the current one:
//   * Q / dO images are rings of NKB + 1 32-row slots: the next problem's block j goes, by LDS-DMA, into the slot the current block
//     j - 1 left (block 0 into the spare one) during iteration j; K and V of the next problem go into their images during the first
//     iterations (the current problem's K / V fragments are in registers by then: they are read from the images at the switch);
//   * delta = rowsum(dO * O) of the next problem: waves 0-3 fetch the 4 KB of O that match their dO pieces into a scratch slot and
//     reduce them one iteration later (one 16-byte LDS read each of O and dO per lane, as the prologue above does from registers);
//   * -lse arrives raw (4-byte LDS-DMA) and is scaled in place in the final phase;
//   * counted waits only: before every barrier each wave leaves at most 6 vector-memory operations in flight (so whatever was issued
//     two phases ago is in LDS for everybody), waves 0-3 wait for exactly their two O / dO pieces before the delta step, and the
//     switch leaves the final phase's own operations (1-2 pieces, the dQ store, 8 dK / dV stores) in flight.
// Gradient values: the same instruction sequence per element as the kernel above (bit-identical results).
template <int NKB> struct WxCfg {
    static constexpr int ROWS = 32 * NKB, BLK = 32 * 64 * 2, NSLOT = NKB + 1, S1 = (ROWS + 16) * 64;
    static constexpr int OFF_Q = 0, OFF_O = OFF_Q + NSLOT * BLK, OFF_K = OFF_O + NSLOT * BLK, OFF_V = OFF_K + NKB * BLK;
    static constexpr int OFF_S = OFF_V + NKB * BLK, OFF_X = OFF_S + 2 * S1, OFF_D = OFF_X + BLK, OFF_L = OFF_D + 2 * 1024;
    static constexpr int LDS = OFF_L + 2 * 1024;      // NKB = 7: 161,792 bytes
};
template <int N> __device__ __forceinline__ void wx_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// The LDS-DMA instructions of this kernel are inline assembly: behind the builtin hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS
// read that MAY touch the bytes in flight -- here every read of the Q / dO rings, i.e. the whole prefetch would be waited for in the
// phase that issues it.  (An operation the compiler does not count can only make ITS counted waits longer, never shorter: the queue
// retires in order.)  rs: the descriptor as four scalars; lds: byte address of the instruction's 1 KB (or 256 bytes) in LDS.
// m0 is written without being declared (hipcc refuses a reserved register on the clobber list): nothing else in the kernel that
// uses these helpers reads m0 -- no builtin LDS-DMA, no s_movrel; gfx950's ds_* instructions do not use it.
__device__ __forceinline__ void wx_dma16(u32x4v rs, unsigned lds, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void wx_dma4(u32x4v rs, unsigned lds, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4v wx_rsrc(const void* base, int bytes) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    return (u32x4v){(unsigned)b, (unsigned)(b >> 32), (unsigned)bytes, 0x00020000u};
}
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)reinterpret_cast<uintptr_t>((lds_vptr3)const_cast<char*>(p)); }
__device__ __forceinline__ void wx_barrier() { asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct WxOff { int q, k, o, l, dq, dk; };     // byte offsets of a problem inside the operands (soffset of every access)

template <int NKB>
__global__ __launch_bounds__(512) void attn_bwd_winx_kernel(const AttnArgs a) {
    using C = WxCfg<NKB>;
    constexpr int HD = 64, KS = 2, DT = 4, ROWS = C::ROWS, BLK = C::BLK, NSLOT = C::NSLOT, NP = (NKB + 1) / 2;
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    char* const ldsQ = wlds + C::OFF_Q;
    char* const ldsO = wlds + C::OFF_O;       // dO
    char* const ldsK = wlds + C::OFF_K;
    char* const ldsV = wlds + C::OFF_V;
    char* const ldsS = wlds + C::OFF_S;       // [2][keys][32 queries] bf16
    char* const ldsX = wlds + C::OFF_X;       // the O pieces of the block whose delta is pending
    char* const ldsDb = wlds + C::OFF_D;      // [2][256] -delta * scale per query
    char* const ldsLb = wlds + C::OFF_L;      // [2][256] -lse * log2 e per query
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c = lane & 15;
    const int n = a.nq, H = a.H, nprob = a.nbh, stride = gridDim.x;
    const float sc2 = a.scale * LOG2E;
    const int NT = (n + 15) >> 4;
    const int base = NT >> 3, rem = NT & 7;
    const int nu = base + (wave < rem ? 1 : 0), tile0 = wave * base + (wave < rem ? wave : rem);
    const int my_t = wave >> 2, my_dt = wave & 3;
    // the operands whole (a problem is a soffset): rows of all windows, this launch's head columns
    const int trows = (nprob / H) * n;
    const u32x4v rsQ = wx_rsrc(a.q, ((trows - 1) * a.ldq + H * HD) * 2), rsK = wx_rsrc(a.k, ((trows - 1) * a.ldk + H * HD) * 2);
    const u32x4v rsV = wx_rsrc(a.v, ((trows - 1) * a.ldk + H * HD) * 2), rsG = wx_rsrc(a.d_o, ((trows - 1) * a.ldo + H * HD) * 2);
    const u32x4v rsO = wx_rsrc(a.o, ((trows - 1) * a.ldo + H * HD) * 2), rsL = wx_rsrc(a.lse, nprob * n * 4);
    const __amdgpu_buffer_rsrc_t rsDQ = rows_rsrc(a.dq, trows, a.ldgq, H * HD, 2), rsDK = rows_rsrc(a.dk, trows, a.ldgk, H * HD, 2);
    const __amdgpu_buffer_rsrc_t rsDV = rows_rsrc(a.dv, trows, a.ldgk, H * HD, 2);
    auto offsets = [&](int p) {
        const int bw = p / H, h = p - bw * H, rb = bw * n;
        WxOff o;
        o.q = (rb * a.ldq + h * HD) * 2; o.k = (rb * a.ldk + h * HD) * 2; o.o = (rb * a.ldo + h * HD) * 2; o.l = p * n * 4;
        o.dq = (rb * a.ldgq + h * HD) * 2; o.dk = (rb * a.ldgk + h * HD) * 2;
        return o;
    };
    // one LDS-DMA instruction = rows [row0, row0 + 8) x 64 columns -> 1 KB of a transposing-read image: lane -> row row0 + lane / 8,
    // LDS chunk lane % 8 <- the source chunk the image's swizzle puts there (row0 a multiple of 8: the swizzle reads row bits 1, 2)
    // (the lane's row and chunk are recomputed behind an opaque copy of the lane index at every use: hoisted out of the problem loop
    // they would be two more of the 256 registers the loop already fills)
    auto dma8 = [&](const u32x4v& rs, int soff, int ld, int row0, const char* dst) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int row = row0 + (l >> 3), lch16 = ((l & 7) ^ (((l >> 4) & 3) << 1)) * 16;
        wx_dma16(rs, lds_addr(dst), row < n ? row * ld * 2 + lch16 : OOB, soff);
    };
    int srow[2], smask[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int krow = (u < nu ? (tile0 + u) * 16 : ROWS) + c;
        srow[u] = krow * 64;
        smask[u] = ds_swz(krow);
    }
    // ---- the first problem: images by LDS-DMA, -lse and delta through registers
    int p = blockIdx.x;
    WxOff off = offsets(p);
    {
#pragma unroll
        for (int j = 0; j < (4 * NKB + 7) / 8; ++j) {
            const int q8 = wave + 8 * j;          // piece: rows 8 q8 .. 8 q8 + 7 (slot q8 / 4 of a ring with base 0)
            if (q8 < 4 * NKB) {
                dma8(rsQ, off.q, a.ldq, q8 * 8, ldsQ + q8 * 1024);
                dma8(rsG, off.o, a.ldo, q8 * 8, ldsO + q8 * 1024);
                dma8(rsK, off.k, a.ldk, q8 * 8, ldsK + q8 * 1024);
                dma8(rsV, off.k, a.ldk, q8 * 8, ldsV + q8 * 1024);
            }
        }
        u32x4v pg[NP], po[NP];
        const int ch = tid & 7, row0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = row0 + 64 * i;
            const int vo = row < n ? (row * a.ldo + ch * 8) * 2 : OOB;
            pg[i] = __builtin_amdgcn_raw_buffer_load_b128(rows_rsrc(a.d_o, trows, a.ldo, H * HD, 2), vo, off.o, 0);
            po[i] = __builtin_amdgcn_raw_buffer_load_b128(rows_rsrc(a.o, trows, a.ldo, H * HD, 2), vo, off.o, 0);
        }
        const float pl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rows_rsrc(a.lse, 1, 0, nprob * n, 4), tid < n ? tid * 4 : OOB, off.l, 0));
        if ((NT & 1) && tid < 128)
            *reinterpret_cast<u32x4v*>(ldsS + (tid >> 6) * C::S1 + NT * 16 * 64 + (tid & 63) * 16) = (u32x4v){0u, 0u, 0u, 0u};
        if (tid < 256) reinterpret_cast<float*>(ldsLb)[tid] = -pl * LOG2E;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = row0 + 64 * i;
            if (row < ROWS) {
                const bf16x8_t x = __builtin_bit_cast(bf16x8_t, pg[i]), y = __builtin_bit_cast(bf16x8_t, po[i]);
                float dl = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)x[j] * (float)y[j];
                dl += __shfl_xor(dl, 1, 64);
                dl += __shfl_xor(dl, 2, 64);
                dl += __shfl_xor(dl, 4, 64);
                if (ch == 0) reinterpret_cast<float*>(ldsDb)[row] = -dl * a.scale;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    f32x4_t adk[2][DT], adv[2][DT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { adk[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; adv[u][dt] = adk[u][dt]; }
#ifdef VPU_WIN_STAMPS
    int mloc = 0;
    const __amdgpu_buffer_rsrc_t rsDbg = rows_rsrc(a.delta + (int64_t)blockIdx.x * n, 1, 0, n, 4);
#define WX_STAMP(slot_) do { __builtin_amdgcn_sched_barrier(0); if (mloc == 1) __builtin_amdgcn_raw_buffer_store_b32((unsigned)__builtin_amdgcn_s_memtime(), rsDbg, lane == 0 ? (wave * 16 + (slot_)) * 4 : OOB, 0, 0); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WX_STAMP(slot_) do { } while (0)
#endif
    int cur = 0, par = 0;                      // ring slot of the current problem's block 0; buffer of its -lse / delta
    bool pend = false;                         // waves 0-3: a block of the next problem whose O / dO pieces are on their way
    int pend_row = 0, pend_slot = 0, pend_par = 0;
    const int dq_col2 = (16 * my_dt + 4 * g) * 2;
    for (;;) {
        const bool has_next = p + stride < nprob;
        const WxOff offn = offsets(has_next ? p + stride : p);
        const int nbase = cur == 0 ? NSLOT - 1 : cur - 1;             // the next problem's block j: slot (nbase + j) % NSLOT
        auto slot = [&](int b0, int j) { const int s = b0 + j; return s >= NSLOT ? s - NSLOT : s; };
        const float* const ldsL = reinterpret_cast<const float*>(ldsLb + par * 1024);
        const float* const ldsD = reinterpret_cast<const float*>(ldsDb + par * 1024);
        // ---- the problem's K / V operands out of the images
        bf16x8_t ktf[NKB], kf[2][KS], vf[2][KS];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) ktf[kb] = frag_tr_perm<HD>(ldsK + kb * BLK, my_dt, lane);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                kf[u][ks] = frag_rc_trv(ldsK, (tile0 + u) * 16, ks, lane);
                vf[u][ks] = frag_rc_trv(ldsV, (tile0 + u) * 16, ks, lane);
                if (u >= nu) { kf[u][ks] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0}; vf[u][ks] = kf[u][ks]; }     // (an unowned slot computes on zeros)
            }
        // what the phase after barrier i adds to the stream: the pending delta, the next problem's pieces
        auto fetch = [&](int i) {
            if (wave < 4) {
                if (pend) {
                    if (i == 0) wx_wait_vm<9>(); else wx_wait_vm<1>();       // (behind the two pieces: the dQ store; after a switch also 8 dK / dV stores)
                    const u32x4v xo = *reinterpret_cast<const u32x4v*>(ldsX + wave * 1024 + lane * 16);
                    const u32x4v xg = *reinterpret_cast<const u32x4v*>(ldsO + pend_slot * BLK + wave * 1024 + lane * 16);
                    const bf16x8_t x = __builtin_bit_cast(bf16x8_t, xg), y = __builtin_bit_cast(bf16x8_t, xo);
                    float dl = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) dl += (float)x[j] * (float)y[j];
                    dl += __shfl_xor(dl, 1, 64);
                    dl += __shfl_xor(dl, 2, 64);
                    dl += __shfl_xor(dl, 4, 64);
                    if ((lane & 7) == 0) reinterpret_cast<float*>(ldsDb + pend_par * 1024)[pend_row + wave * 8 + (lane >> 3)] = -dl * a.scale;
                    pend = false;
                }
            }
            if (has_next) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int q8 = i * 16 + 2 * wave + e;
                    if (q8 < 4 * NKB) dma8(rsK, offn.k, a.ldk, q8 * 8, ldsK + q8 * 1024);
                    else if (q8 < 8 * NKB) dma8(rsV, offn.k, a.ldk, (q8 - 4 * NKB) * 8, ldsV + (q8 - 4 * NKB) * 1024);
                }
                const int sl = slot(nbase, i);
                if (wave < 4) {
                    dma8(rsG, offn.o, a.ldo, i * 32 + wave * 8, ldsO + sl * BLK + wave * 1024);
                    dma8(rsO, offn.o, a.ldo, i * 32 + wave * 8, ldsX + wave * 1024);
                    pend = true; pend_row = i * 32; pend_slot = sl; pend_par = par ^ 1;
                } else {
                    dma8(rsQ, offn.q, a.ldq, i * 32 + (wave - 4) * 8, ldsQ + sl * BLK + (wave - 4) * 1024);
                    if (i == 0) {
                        const int f = (wave - 4) * 64 + lane;
                        wx_dma4(rsL, lds_addr(ldsLb + (par ^ 1) * 1024 + (wave - 4) * 256), f < n ? f * 4 : OOB, offn.l);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto scores = [&](int i, bf16x8_t (&pf)[2], bf16x8_t (&dsf)[2]) {
            const int qc = i * CH, sl = slot(cur, i);
            const char* sQ = ldsQ + sl * BLK;
            const char* sO = ldsO + sl * BLK;
            char* const sS = ldsS + (i & 1) * C::S1;
            bf16x4w_t p4[2][2], d4[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4_t nl2 = *reinterpret_cast<const f32x4_t*>(ldsL + qc + 16 * t + 4 * g);
                const f32x4_t nds = *reinterpret_cast<const f32x4_t*>(ldsD + qc + 16 * t + 4 * g);
                f32x4_t sv[2], dp[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { sv[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[u] = sv[u]; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8_t qfr = frag_rc_trv(sQ, 16 * t, ks, lane), ofr = frag_rc_trv(sO, 16 * t, ks, lane);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        sv[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[u][ks], sv[u], 0, 0, 0);
                        dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[u][ks], dp[u], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4_t pv, dv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[u][r], sc2, nl2[r]));
                        dv[r] = pv[r] * __builtin_fmaf(dp[u][r], a.scale, nds[r]);
                    }
                    p4[u][t] = pack4(pv);
                    d4[u][t] = pack4(dv);
                    *reinterpret_cast<bf16x4w_t*>(sS + srow[u] + ((4 * t + g) ^ smask[u]) * 8) = d4[u][t];
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pf[u] = __builtin_shufflevector(p4[u][0], p4[u][1], 0, 1, 2, 3, 4, 5, 6, 7);
                dsf[u] = __builtin_shufflevector(d4[u][0], d4[u][1], 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        // gradients of query block i: dV^T += dO^T P, dK^T += Q^T dS for this wave's key tiles, this wave's tile of the block's dQ^T
        auto grads = [&](int i, const bf16x8_t (&pf)[2], const bf16x8_t (&dsf)[2]) {
            const int qc = i * CH, sl = slot(cur, i);
            const char* sQ = ldsQ + sl * BLK;
            const char* sO = ldsO + sl * BLK;
            const char* sS = ldsS + (i & 1) * C::S1;
            // (the second half of the dS fragments is read behind the dV / dK products, into the registers their operands leave:
            // 256 registers per lane hold the accumulators, the K / V operands and ONE set of fragments)
            constexpr int NA = NKB > 4 ? NKB / 2 : NKB;
            bf16x8_t otf[DT], qtf[DT], af[NKB];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { otf[dt] = frag_tr_perm<HD>(sO, dt, lane); qtf[dt] = frag_tr_perm<HD>(sQ, dt, lane); }
#pragma unroll
            for (int kb = 0; kb < NA; ++kb) af[kb] = frag_ds(sS + kb * 32 * 64, my_t, lane);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    adv[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otf[dt], pf[u], adv[u][dt], 0, 0, 0);
                    adk[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf[dt], dsf[u], adk[u][dt], 0, 0, 0);
                }
            if (NA < NKB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = NA; kb < NKB; ++kb) af[kb] = frag_ds(sS + kb * 32 * 64, my_t, lane);
            f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[kb], af[kb], acc, 0, 0, 0);
            const int qq = qc + 16 * my_t + c;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2w, pack4(acc)), rsDQ, qq < n ? qq * a.ldgq * 2 + dq_col2 : OOB, off.dq, 0);
        };
        bf16x8_t pf[2], dsf[2];
        WX_STAMP(1);
        scores(0, pf, dsf);
        WX_STAMP(2);
#pragma nounroll
        for (int i = 0; i + 1 < NKB; ++i) {
            wx_barrier();        // the dS image of block i is complete (and everybody has finished with block i - 1 and its dS image)
            WX_STAMP(3 + i);
            fetch(i);
            bf16x8_t pn[2], dn[2];
            grads(i, pf, dsf);
            scores(i + 1, pn, dn);
#pragma unroll
            for (int u = 0; u < 2; ++u) { pf[u] = pn[u]; dsf[u] = dn[u]; }
        }
        // ---- the final phase: gradients of the last block, the problem's dK / dV
        wx_barrier();
        WX_STAMP(3 + NKB - 1);
        fetch(NKB - 1);
        grads(NKB - 1, pf, dsf);
        WX_STAMP(11);
        // dK / dV leave in whole 64-byte lines: a lane holds 8 bytes of a row (key c, columns 16 dt + 4 g ..), four lanes 32 contiguous
        // bytes -- stored like that every wave instruction was sixteen half-line writes and took 260-460 cycles to issue.  Two column
        // tiles at a time go through 1 KB of the dS image this phase does not read (16 keys x 32 columns), come back as 16 bytes per
        // lane (row lane / 4, 8 columns) and are stored as sixteen full 64-byte segments per instruction.
        // (the 1 KB: the rows of the wave's own key tile in that image -- nobody else's, never the zero rows; an unowned slot
        // stores nothing but issues its instructions: the counted waits below assume eight per wave)
        {
            const int srow16 = lane >> 2, sch = lane & 3;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                char* const stg = ldsS + (NKB & 1) * C::S1 + (u < nu ? (tile0 + u) * 16 : ROWS) * 64;
                const int kk = (tile0 + u) * 16 + srow16;
                const bool live = u < nu && kk < n;
                const int vo = (kk * a.ldgk + 8 * sch) * 2;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int kv = 0; kv < 2; ++kv) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int dt = 2 * hf + e;
                            *reinterpret_cast<bf16x4w_t*>(stg + c * 64 + e * 32 + g * 8) = pack4(kv ? adv[u][dt] : adk[u][dt]);
                        }
                        const u32x4v row = *reinterpret_cast<const u32x4v*>(stg + srow16 * 64 + sch * 16);
                        __builtin_amdgcn_raw_buffer_store_b128(row, kv ? rsDV : rsDK, live ? vo + hf * 64 : OOB, off.dk, 0);
                    }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) { adk[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; adv[u][dt] = adk[u][dt]; }
            }
        }
        WX_STAMP(12);
        if (!has_next) break;
        // ---- switch: everything the next problem starts from is in LDS once the operations older than this phase's own are done
        if (wave < 4) wx_wait_vm<11>();
        else {
            wx_wait_vm<10>();
            // the next problem's -lse arrived raw (phase 0, this wave's own 64 values): scale it in place
            float* const ln = reinterpret_cast<float*>(ldsLb + (par ^ 1) * 1024) + (wave - 4) * 64 + lane;
            *ln = -*ln * LOG2E;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        p += stride; off = offn; cur = nbase; par ^= 1;
#ifdef VPU_WIN_STAMPS
        ++mloc;
#endif
        WX_STAMP(0);
    }
}

#undef WX_STAMP
// ------------------------------------------------------------------------------------------------ one-pass backward, several per CU
// Round 5.  The kernel above holds a whole (window, head) problem in 132 KB of LDS and 210 registers: ONE workgroup per CU,
// two waves per SIMD, and 576 problems on 256 CUs are 2.25 rounds -- the third one a quarter full; inside a round nothing
// overlaps a workgroup's 6.4-us prologue or the latencies of its loop (LDS read -> MFMA -> exp -> LDS write -> barrier ->
// LDS read -> MFMA).  This form trades LDS traffic for occupancy: a 256-thread workgroup that needs 52 KB of LDS (WpCfg<1>::LDS = 53248: three per CU) and < 128
// registers, THREE per CU, so that all 576 problems of a ViT-B window block are resident at once (768 slots) and a SIMD
// always has another problem's wave to issue from:
//   * the keys are walked in PASSES of 4 NU tiles of 16 (NU per wave: K / V fragments and the dK / dV accumulators of one
//     tile are 48 registers); a pass streams every 32-query block of Q and dO through a three-stage ring of 8-KB stages
//     filled by LDS-DMA (no staging registers; rows >= n arrive as zeros), two steps ahead, and the ring runs on across the
//     pass boundaries -- the price: Q / dO are read from LDS once per pass instead of once;
//   * per step exactly as above: S and dP once, P / dS in the accumulator layout are the operands of dV / dK, dS goes
//     through a [key][32 queries] image for dQ (new swizzle: conflict-free for the 16-lane write groups and the 32-lane
//     transposing-read halves; the old image cost 2.45 M bank-conflict cycles per launch); ONE barrier per step, none
//     between passes (a wave's K / V fragments and its K^T operand of the dQ product come straight from global memory);
//   * dQ, dK and dV are computed TRANSPOSED (swapped MFMA operands: the same registers), so a lane owns four consecutive
//     columns of one row: 8-byte stores; the dQ of a later pass adds to what the earlier ones stored (same thread, same
//     address: read back as the MFMA's C operand -- one more bf16 rounding of the partial sum per pass);
//   * delta = rowsum(dO * O) and -lse come in through registers in the prologue (every load in flight at once).
constexpr int WP_NST = 5;                                   // Q / dO ring stages: a block is requested WP_NST - 2 steps before its first read
constexpr int WP_BLK = CH * 64 * 2;                         // one [32][64] bf16 block image: 4 KB
constexpr int WP_STAGE = 2 * WP_BLK;                        // Q block | dO block
template <int NU> struct WpCfg {
    static constexpr int SROWS = 64 * NU + 16;              // dS image rows: the pass's keys + 16 spare rows (unowned slots write there)
    static constexpr int SBUF = SROWS * 64;
    static constexpr int LDS = WP_NST * WP_STAGE + 2 * SBUF + 2 * WIN_ROWS * 4;      // NU = 1: 40960 + 10240 + 2048 = 53248: three per CU
};

typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

// rows [r0 + 8 piece, + 8) x 64 columns of a [*, ld] matrix -> 1 KB of a transposing-read image by ONE LDS-DMA instruction of
// this wave (lane -> row r0 + 8 piece + lane / 8, LDS chunk lane % 8 <- the source chunk the image's swizzle puts there)
__device__ __forceinline__ void wp_dma_rows(__amdgpu_buffer_rsrc_t rs, int ld, int r0, int n, char* img, int piece, int lane) {
    const int rl = piece * 8 + (lane >> 3), row = r0 + rl;
    const int chunk = (lane & 7) ^ (((rl >> 1) & 3) << 1);
    const int voff = row < n ? (row * ld + chunk * 8) * 2 : (int)0x80000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr3)(img + piece * 1024), 16, voff, 0, 0, 0);
}

template <int NU> struct WpAcc { f32x4_t dk[NU][4], dv[NU][4]; };

// One step = one 32-query block of one pass.  `wr` (the stage the DMA of step + WP_NST - 1 goes to), `rd` (this step's stage)
// and `sS` (this step's dS image) are __restrict__ parameters of an inlined function on purpose: the scoped no-alias
// information is what keeps hipcc from putting s_waitcnt vmcnt(0) in front of every LDS read that follows the DMA in program
// order.  `dqr`: this block's dQ^T tile of the wave (bf16, in registers across the passes).
template <int NU>
__device__ __forceinline__ void wp_step(const AttnArgs& a, __amdgpu_buffer_rsrc_t rsQ, __amdgpu_buffer_rsrc_t rsG, char* __restrict__ wr,
                                        const char* __restrict__ rd, char* __restrict__ sS, const float* __restrict__ ldsL,
                                        const float* __restrict__ ldsD, int qc, int qcn, int nn, bool first_pass, int nkb,
                                        const bf16x8_t (&kf)[NU][2], const bf16x8_t (&vf)[NU][2], const bf16x8_t (&ktf)[2 * NU],
                                        const int (&srow)[NU], const int (&sswz)[NU], WpAcc<NU>& acc, bf16x4v (&dqr)[2],
                                        int wave, int lane, float sc2) {
    constexpr int HD = 64, KS = 2, DT = 4;
    const int g = lane >> 4;
    const char* sQ = rd;
    const char* sO = rd + WP_BLK;
    bf16x8_t pf[NU], dsf[NU];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4_t nl2 = *reinterpret_cast<const f32x4_t*>(ldsL + qc + 16 * t + 4 * g);
        const f32x4_t nds = *reinterpret_cast<const f32x4_t*>(ldsD + qc + 16 * t + 4 * g);
        f32x4_t s[NU], dp[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) { s[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[u] = s[u]; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t qfr = frag_rc_trv(sQ, 16 * t, ks, lane), ofr = frag_rc_trv(sO, 16 * t, ks, lane);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[u][ks], s[u], 0, 0, 0);
                dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[u][ks], dp[u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            bf16x4v w4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // element [query = qc + 16 t + 4 g + r][key = the slot's tile, column lane & 15]
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], sc2, nl2[r]));
                const float ds = p * __builtin_fmaf(dp[u][r], a.scale, nds[r]);
                pf[u][4 * t + r] = (bf16_t)p;
                w4[r] = (bf16_t)ds;
                dsf[u][4 * t + r] = w4[r];
            }
            *reinterpret_cast<bf16x4v*>(sS + srow[u] + (((4 * t + g) ^ sswz[u]) << 3)) = w4;
        }
    }
    // my pieces of step + 1 have landed (requested WP_NST - 2 steps ago; the WP_NST - 3 younger requests -- two DMA instructions per
    // step, real or out of range -- stay in flight), my dS units are written; then everybody's
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (WP_NST - 3)) : "memory");
    // step + WP_NST - 1 -> the stage step - 1 used (everybody is past its last read: they all passed phase A of this step); past the
    // end of the problem the request is out of range: zeros, no memory traffic, the same count
    wp_dma_rows(rsQ, a.ldq, qcn, nn, wr, wave, lane);
    wp_dma_rows(rsG, a.ldo, qcn, nn, wr + WP_BLK, wave, lane);
    // dV^T += dO^T P, dK^T += Q^T dS (transposed products: swapped operands): [d = 16 dt + 4 g + r][key = lane & 15]
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const bf16x8_t otf = frag_tr_perm<HD>(sO, dt, lane), qtf = frag_tr_perm<HD>(sQ, dt, lane);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            acc.dv[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otf, pf[u], acc.dv[u][dt], 0, 0, 0);
            acc.dk[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf[u], acc.dk[u][dt], 0, 0, 0);
        }
    }
    // dQ^T tile [d = 16 wave + 4 g + r][query = qc + 16 t + (lane & 15)] over the pass's keys, on top of the earlier passes'
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x4_t dq;
#pragma unroll
        for (int r = 0; r < 4; ++r) dq[r] = first_pass ? 0.f : (float)dqr[t][r];
#pragma unroll
        for (int kb = 0; kb < 2 * NU; ++kb)
            if (kb < nkb) dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[kb], frag_ds(sS + kb * 32 * 64, t, lane), dq, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dqr[t][r] = (bf16_t)dq[r];
    }
}

// K / V fragments of a wave's tile(s) of a pass, and the K^T operand of its dQ columns over the pass's keys: straight from global
// memory (no LDS image, no barrier); rows >= n are zeros
template <int NU>
__device__ __forceinline__ void wp_pass_kv(const bf16_t* k, const bf16_t* v, int ldk, int tbase, int ntile, int n, int wave, int lane,
                                           bf16x8_t (&kf)[NU][2], bf16x8_t (&vf)[NU][2]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int slot = NU * wave + u;
        const int key0 = slot < ntile ? (tbase + slot) * 16 : n;          // (an unowned slot reads nothing)
        load_rows_as_bn<2>(k, ldk, key0, n, lane, kf[u], 64);
        load_rows_as_bn<2>(v, ldk, key0, n, lane, vf[u], 64);
    }
}
template <int NU>
__device__ __forceinline__ void wp_pass_kt(__amdgpu_buffer_rsrc_t rsK, int ldk, int tbase, int n, int wave, int lane, bf16x8_t (&ktf)[2 * NU]) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int kb = 0; kb < 2 * NU; ++kb) {
        // lane (g, c): K[key = 16 tbase + 32 kb + {4 g + j, 16 + 4 g + j}][column 16 wave + c], j = 0..3
        s16x8_t kk;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = tbase * 16 + 32 * kb + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
            kk[j] = __builtin_amdgcn_raw_buffer_load_b16(rsK, key < n ? (key * ldk + 16 * wave + c) * 2 : (int)0x80000000, 0, 0);
        }
        ktf[kb] = __builtin_bit_cast(bf16x8_t, kk);
    }
}

template <int NU>
__global__ __launch_bounds__(256, NU == 1 ? 3 : 2) void attn_bwd_winp_kernel(const AttnArgs a) {
    constexpr int HD = 64, KS = 2, DT = 4, TP = 4 * NU;     // tiles per pass
    typedef WpCfg<NU> Cf;
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    char* const ring = wlds;
    char* const ldsS = wlds + WP_NST * WP_STAGE;
    float* const ldsD = reinterpret_cast<float*>(ldsS + 2 * Cf::SBUF);   // -delta * scale per query
    float* const ldsL = ldsD + WIN_ROWS;                                 // -lse * log2 e per query
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.x, bw = bh / a.H, h = bh % a.H, n = a.nq;
    const int64_t rb = (int64_t)bw * n;
    const bf16_t* q = a.q + rb * a.ldq + h * HD;
    const bf16_t* k = a.k + rb * a.ldk + h * HD;
    const bf16_t* v = a.v + rb * a.ldk + h * HD;
    const bf16_t* o = a.o + rb * a.ldo + h * HD;
    const bf16_t* d_o = a.d_o + rb * a.ldo + h * HD;
    const float sc2 = a.scale * LOG2E;
    const int NT = (n + 15) >> 4, nch = (n + CH - 1) / CH, NPS = (NT + TP - 1) / TP, total = NPS * nch;
    const __amdgpu_buffer_rsrc_t rsQ = rows_rsrc(q, n, a.ldq, HD, 2), rsK = rows_rsrc(k, n, a.ldk, HD, 2);
    const __amdgpu_buffer_rsrc_t rsG = rows_rsrc(d_o, n, a.ldo, HD, 2), rsO = rows_rsrc(o, n, a.ldo, HD, 2);
    // ---- prologue: step 0 by DMA; -lse and delta = rowsum(dO * O) through registers; the first pass's operands; steps 1 .. WP_NST - 2
    // by DMA; the dS images zeroed (a pass with an odd tile count reads the 16 rows behind its last tile: they must hold finite
    // numbers -- their K rows are zeros)
    wp_dma_rows(rsQ, a.ldq, 0, n, ring, wave, lane);
    wp_dma_rows(rsG, a.ldo, 0, n, ring + WP_BLK, wave, lane);
    bf16x8_t kf[NU][KS], vf[NU][KS], ktf[2 * NU];
    {
        const __amdgpu_buffer_rsrc_t rsL = rows_rsrc(reinterpret_cast<const char*>(a.lse + (int64_t)bh * n), 1, 0, n, 4);
        const float pl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, tid * 4, 0, 0));   // (rows >= n: out of range = 0)
        u32x4v pg[WIN_MAX_KB], po[WIN_MAX_KB];
        const int ch = tid & 7, row0 = tid >> 3;      // piece i: row row0 + 32 i, 16-byte chunk ch
#pragma unroll
        for (int i = 0; i < WIN_MAX_KB; ++i) {
            const int row = row0 + 32 * i;
            const int off = (i < nch && row < n) ? (row * a.ldo + ch * 8) * 2 : (int)0x80000000;
            pg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsG, off, 0, 0);
            po[i] = __builtin_amdgcn_raw_buffer_load_b128(rsO, off, 0, 0);
        }
        wp_pass_kv<NU>(k, v, a.ldk, 0, min(TP, NT), n, wave, lane, kf, vf);
        wp_pass_kt<NU>(rsK, a.ldk, 0, n, wave, lane, ktf);
#pragma unroll
        for (int j = 1; j < WP_NST - 1; ++j) {       // step j = block j % nch of pass j / nch (out of range past the end)
            const int bj = j % nch;
            const int nn = j < total ? n : 0;
            wp_dma_rows(rsQ, a.ldq, bj * CH, nn, ring + j * WP_STAGE, wave, lane);
            wp_dma_rows(rsG, a.ldo, bj * CH, nn, ring + j * WP_STAGE + WP_BLK, wave, lane);
        }
#pragma unroll
        for (int i = 0; i < (2 * Cf::SBUF / 16 + 255) / 256; ++i)
            if (tid + i * 256 < 2 * Cf::SBUF / 16) *reinterpret_cast<u32x4v*>(ldsS + (tid + i * 256) * 16) = (u32x4v){0u, 0u, 0u, 0u};
        ldsL[tid] = -pl * LOG2E;
#pragma unroll
        for (int i = 0; i < WIN_MAX_KB; ++i) {
            const bf16x8_t x = __builtin_bit_cast(bf16x8_t, pg[i]), y = __builtin_bit_cast(bf16x8_t, po[i]);
            float dl = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)x[j] * (float)y[j];
            dl += __shfl_xor(dl, 1, 64);
            dl += __shfl_xor(dl, 2, 64);
            dl += __shfl_xor(dl, 4, 64);
            if (ch == 0) ldsD[row0 + 32 * i] = -dl * a.scale;
        }
    }
    // (step 0 and everything requested before the last 2 (WP_NST - 2) DMA instructions have landed)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (WP_NST - 2)) : "memory");
    bf16x4v dqr[WIN_MAX_KB][2];        // this wave's dQ^T tiles of every block: bf16 across the passes, stored once at the end
    int st = 0, step = 0;
    for (int ps = 0; ps < NPS; ++ps) {
        const int tbase = TP * ps, ntile = min(TP, NT - tbase), nkb = (ntile + 1) >> 1;
        int srow[NU], sswz[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int slot = NU * wave + u;
            const int krow = (slot < ntile ? slot * 16 : 64 * NU) + c;        // its dS rows in the pass's image; unowned: the spare rows
            srow[u] = krow * 64;
            sswz[u] = ds_swz(krow);
        }
        WpAcc<NU> acc;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { acc.dk[u][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc.dv[u][dt] = acc.dk[u][dt]; }
#pragma unroll
        for (int i = 0; i < WIN_MAX_KB; ++i) {
            if (i < nch) {
                const int wst = st == 0 ? WP_NST - 1 : st - 1;        // the stage of step - 1 = of step + WP_NST - 1
                const int sn = step + WP_NST - 1;                     // the step requested now: block sn % nch (the ring runs on into the next pass)
                wp_step<NU>(a, rsQ, rsG, ring + wst * WP_STAGE, ring + st * WP_STAGE, ldsS + (step & 1) * Cf::SBUF, ldsL, ldsD, i * CH,
                            (sn % nch) * CH, sn < total ? n : 0, ps == 0, nkb, kf, vf, ktf, srow, sswz, acc, dqr[i], wave, lane, sc2);
                st = st == WP_NST - 1 ? 0 : st + 1;
                ++step;
            }
        }
        // ---- dK / dV of my tile(s): lane (g, c) holds [d = 16 dt + 4 g + r][key = tile * 16 + c]; then the next pass's operands
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int slot = NU * wave + u, kk = (tbase + slot) * 16 + c;
            if (slot < ntile && kk < n) {
                bf16_t* kr = a.dk + (rb + kk) * a.ldgk + h * HD + 4 * g;
                bf16_t* vr = a.dv + (rb + kk) * a.ldgk + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16x4v k4, v4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { k4[r] = (bf16_t)acc.dk[u][dt][r]; v4[r] = (bf16_t)acc.dv[u][dt][r]; }
                    *reinterpret_cast<bf16x4v*>(kr + 16 * dt) = k4;
                    *reinterpret_cast<bf16x4v*>(vr + 16 * dt) = v4;
                }
            }
        }
        if (ps + 1 < NPS) {
            wp_pass_kv<NU>(k, v, a.ldk, tbase + TP, min(TP, NT - tbase - TP), n, wave, lane, kf, vf);
            wp_pass_kt<NU>(rsK, a.ldk, tbase + TP, n, wave, lane, ktf);
        }
    }
    // ---- dQ: lane (g, c) holds [d = 16 wave + 4 g + r][query = 32 i + 16 t + c]
    bf16_t* const dqp = a.dq + rb * a.ldgq + h * HD + 16 * wave + 4 * g;
#pragma unroll
    for (int i = 0; i < WIN_MAX_KB; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int qq = i * CH + 16 * t + c;
            if (i < nch && qq < n) *reinterpret_cast<bf16x4v*>(dqp + (int64_t)qq * a.ldgq) = dqr[i][t];
        }
}

// ------------------------------------------------------------------------------------------------ split forward: combine
// A forward launched with qdiv = S leaves S partial softmaxes per (batch, head, query): normalised outputs o_s over key range s
// and their log-sum-exp lse_s.  out = sum_s exp(lse_s - lse) o_s,  lse = log sum_s exp(lse_s).  One thread per 8 columns.
__global__ __launch_bounds__(256) void attn_combine_kernel(const bf16_t* __restrict__ o_s, const float* __restrict__ lse_s,
                                                           bf16_t* __restrict__ out, float* __restrict__ lse, int nb, int H, int nq,
                                                           int hd, int S, int ld_s, int ldo) {
    const int cpr = hd / 8;
    const int64_t total = (int64_t)nb * nq * H * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % cpr);
        const int h = (int)((i / cpr) % H);
        const int qq = (int)((i / ((int64_t)cpr * H)) % nq);
        const int b = (int)(i / ((int64_t)cpr * H * nq));
        float ls[8], m = -INFINITY;
        for (int sp = 0; sp < S; ++sp) {
            ls[sp] = lse_s[((int64_t)(b * S + sp) * H + h) * nq + qq];
            m = fmaxf(m, ls[sp]);
        }
        float tot = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < S; ++sp) {
            const float w = __expf(ls[sp] - m);
            tot += w;
            float v[8];
            load8(o_s + ((int64_t)(b * S + sp) * nq + qq) * ld_s + h * hd + ck * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w * v[j];
        }
        const float inv = 1.f / tot;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= inv;
        store8(out + ((int64_t)b * nq + qq) * ldo + h * hd + ck * 8, acc);
        if (ck == 0) lse[((int64_t)b * H + h) * nq + qq] = m + __logf(tot);
    }
}

// "lean" = 1 (default): the kernels above; 0: the round-1 step kernels (kept for A/B runs and as a second implementation in
// the tests).  VPU_ATTN_LEAN sets the process default.
std::atomic<int> g_opt_lean{-1}, g_opt_onepass{-1};
inline bool xcd_map_enabled() {   // VPU_ATTN_XCDMAP=0: the plain 2-D grid (A/B runs)
    static const int e0 = [] { const char* e = vpu_lab_getenv("VPU_ATTN_XCDMAP"); return e ? atoi(e) : 1; }();
    return e0 != 0;
}
inline int onepass_enabled() {   // "onepass": the one-pass (window, head) backward for n <= 256, head dim 64: 3 = persistent workgroups that fetch the next problem (round 6; 65 <= n <= 224, 1 elsewhere), 2 = key passes, three workgroups per CU (round 5), 1 = one workgroup per problem, 0 = two kernels
    static const int e0 = [] { const char* e = getenv("VPU_ATTN_ONEPASS"); return e ? atoi(e) : 3; }();
    const int v = g_opt_onepass.load(std::memory_order_relaxed);
    return v >= 0 ? v : e0;
}
inline bool wide_two() {   // two query tiles per wave also in the 128-column instantiation (head dims 80 / 96: ViT-H, the neck)
    static const int e0 = [] { const char* e = vpu_lab_getenv("VPU_ATTN_WIDE2"); return e ? atoi(e) : 1; }();
    return e0 != 0;
}
inline bool lean_enabled() {
    static const int e0 = [] { const char* e = getenv("VPU_ATTN_LEAN"); return e ? atoi(e) : 1; }();
    const int v = g_opt_lean.load(std::memory_order_relaxed);
    return (v >= 0 ? v : e0) != 0;
}

inline bool ok16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int hd_image(int hd) { return hd <= 32 ? 32 : (hd <= 64 ? 64 : 128); }
int hd_computed(int hd) { return hd_image(hd) == 128 && hd <= 96 ? 96 : hd_image(hd); }   // the HC template argument

// name(s) of the instantiation(s) the calling thread's last launch dispatched to, as rocprofv3 prints them
// (vpu_attn_last_kernel): lets a test assert that e.g. the head-dim-80 path of ViT-H really ran
thread_local char g_last_attn[192] = "";

}  // namespace

extern "C" const char* vpu_attn_last_kernel(void) { return g_last_attn; }

extern "C" int vpu_attn_set_option(const char* name, int32_t value) {
    vpu_clear_stale_error();
    if (name && !strcmp(name, "lean") && value >= -1 && value <= 1) {
        g_opt_lean.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    if (name && !strcmp(name, "onepass") && value >= -1 && value <= 3) {
        g_opt_onepass.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    vpu_set_error("vpu_attn_set_option: known options: lean (-1 environment default VPU_ATTN_LEAN, 0 round-1 step kernels, 1 lean kernels), "
                  "onepass (-1 environment default VPU_ATTN_ONEPASS, 0 two-kernel backward, 1 one-pass backward for window-sized problems with one workgroup per problem, 2 in key passes with three workgroups per CU, 3 persistent workgroups that fetch the next problem while they compute (65 <= n <= 224; 1 elsewhere))");
    return VPU_ERR_ARG;
}

static int xattn_fwd_impl(const void* q, const void* k, const void* v, void* out, float* lse, int32_t nb, int32_t H,
                          int32_t nq, int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, float scale, int32_t qdiv,
                          int32_t kdiv, void* stream) {
    vpu_clear_stale_error();
    if (qdiv < 1 || kdiv < 1 || (qdiv > 1 && kdiv > 1) || nb % (qdiv * kdiv) || ((qdiv > 1 || kdiv > 1) && !lean_enabled())) {
        vpu_set_error("xattn_fwd_split: qdiv, kdiv >= 1, one of them 1, nb a multiple of the other; the lean kernels only");
        return VPU_ERR_ARG;
    }
    if (hd <= 0 || hd > 128 || hd % 16 || ldq % 8 || ldk % 8 || ldo % 8 || !ok16(q) || !ok16(k) || !ok16(v) || !ok16(out) || nb <= 0 ||
        H <= 0 || nq <= 0 || nk <= 0 || (int64_t)nk * ldk >= (1 << 29)) {
        vpu_set_error("xattn_fwd: head dim a multiple of 16 up to 128, 16-byte aligned slices (q, k, v, out), row strides % 8 == 0, one batch entry of k/v below 1 GiB");
        return VPU_ERR_ARG;
    }
    AttnArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)out; a.lse = lse;
    a.nq = nq; a.nk = nk; a.H = H; a.hd = hd; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.scale = scale;
    a.qdiv = qdiv; a.kdiv = kdiv;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (lean_enabled()) {
        // two 16-query tiles per wave (128 queries per workgroup) once a problem has more than 64 queries
        // two 16-query tiles per wave (128 queries per workgroup) once a problem has more than 64 queries
        const bool two = nq > 64 && (hd_image(hd) <= 64 || wide_two());
        dim3 grid(two ? (nq + 127) / 128 : (nq + 63) / 64, nb * H);
        if (xcd_map_enabled() && grid.x > 1) { a.gx = grid.x; a.nbh = nb * H; grid = dim3(grid.x * grid.y); }
        snprintf(g_last_attn, sizeof(g_last_attn), "attn_fwd_lean_kernel<%d, %d, %d>", hd_image(hd), two ? 2 : 1, hd_computed(hd));
        switch (hd_image(hd) * 4 + (two ? 2 : 1)) {
            case 128 * 4 + 2:
                if (hd <= 96) attn_fwd_lean_kernel<128, 2, 96><<<grid, 256, 0, s>>>(a);   // head dim 80 / 96: 3 of 4 k-steps, 6 of 8 d-tiles
                else attn_fwd_lean_kernel<128, 2, 128><<<grid, 256, 0, s>>>(a);
                break;
            case 32 * 4 + 2: attn_fwd_lean_kernel<32, 2, 32><<<grid, 256, 0, s>>>(a); break;
            case 32 * 4 + 1: attn_fwd_lean_kernel<32, 1, 32><<<grid, 256, 0, s>>>(a); break;
            case 64 * 4 + 2: attn_fwd_lean_kernel<64, 2, 64><<<grid, 256, 0, s>>>(a); break;
            case 64 * 4 + 1: attn_fwd_lean_kernel<64, 1, 64><<<grid, 256, 0, s>>>(a); break;
            default:
                if (hd <= 96) attn_fwd_lean_kernel<128, 1, 96><<<grid, 256, 0, s>>>(a);
                else attn_fwd_lean_kernel<128, 1, 128><<<grid, 256, 0, s>>>(a);
                break;
        }
        return vpu_check_launch("vpu_xattn_fwd");
    }
    static const int qt2 = [] { const char* e = vpu_lab_getenv("VPU_ATTN_QT"); return e ? atoi(e) : 2; }();
    const bool two = qt2 == 2 && nq > 64 && nk <= 256 && hd_image(hd) <= 64;
    dim3 grid(two ? (nq + 127) / 128 : (nq + 63) / 64, nb * H);
    snprintf(g_last_attn, sizeof(g_last_attn), "attn_fwd_kernel<%d, 1, %d>", hd_image(hd), two ? 2 : 1);
    if (two) {
        if (hd_image(hd) == 32) attn_fwd_kernel<32, 1, 2><<<grid, 256, 0, s>>>(a);
        else attn_fwd_kernel<64, 1, 2><<<grid, 256, 0, s>>>(a);
        return vpu_check_launch("vpu_xattn_fwd");
    }
    switch (hd_image(hd)) {
        case 32: attn_fwd_kernel<32, 1, 1><<<grid, 256, 0, s>>>(a); break;
        case 64: attn_fwd_kernel<64, 1, 1><<<grid, 256, 0, s>>>(a); break;
        default: attn_fwd_kernel<128, 1, 1><<<grid, 256, 0, s>>>(a); break;
    }
    return vpu_check_launch("vpu_xattn_fwd");
}

extern "C" int vpu_xattn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int32_t nb, int32_t H,
                             int32_t nq, int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, float scale,
                             void* stream) {
    return xattn_fwd_impl(q, k, v, out, lse, nb, H, nq, nk, hd, ldq, ldk, ldo, scale, 1, 1, stream);
}
extern "C" int vpu_xattn_fwd_split(const void* q, const void* k, const void* v, void* out, float* lse, int32_t nb, int32_t H,
                                   int32_t nq, int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, float scale,
                                   int32_t qdiv, int32_t kdiv, void* stream) {
    return xattn_fwd_impl(q, k, v, out, lse, nb, H, nq, nk, hd, ldq, ldk, ldo, scale, qdiv, kdiv, stream);
}
extern "C" int vpu_attn_combine(const void* o_s, const float* lse_s, void* out, float* lse, int32_t nb, int32_t H, int32_t nq,
                                int32_t hd, int32_t S, int32_t ld_s, int32_t ldo, void* stream) {
    vpu_clear_stale_error();
    if (!o_s || !lse_s || !out || !lse || nb < 1 || H < 1 || nq < 1 || hd < 8 || hd % 8 || S < 1 || S > 8 || ld_s % 8 || ldo % 8) {
        vpu_set_error("attn_combine: non-null operands, head dim and row strides multiples of 8, 1 <= S <= 8");
        return VPU_ERR_ARG;
    }
    const int64_t total = (int64_t)nb * nq * H * (hd / 8);
    attn_combine_kernel<<<vpu_grid_for(total, 256, 4096), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
        (const bf16_t*)o_s, lse_s, (bf16_t*)out, lse, nb, H, nq, hd, S, ld_s, ldo);
    return vpu_check_launch("vpu_attn_combine");
}

static int xattn_bwd_impl(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                          const float* lse, float* delta, void* dq, void* dk, void* dv, int32_t nb, int32_t H,
                          int32_t nq, int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldgq,
                          int32_t ldgk, float scale, int32_t qdiv, int32_t kdiv, void* stream) {
    vpu_clear_stale_error();
    const bool split = qdiv > 1 || kdiv > 1;
    if (qdiv < 1 || kdiv < 1 || (qdiv > 1 && kdiv > 1) || nb % (qdiv * kdiv) || (split && (!lean_enabled() || nq % 4))) {
        vpu_set_error("xattn_bwd_split: qdiv, kdiv >= 1, one of them 1, nb a multiple of the other; the lean kernels only (nq % 4 == 0)");
        return VPU_ERR_ARG;
    }
    if (hd <= 0 || hd > 128 || hd % 16 || ldq % 8 || ldk % 8 || ldo % 8 || !ok16(q) || !ok16(k) || !ok16(v) || !ok16(o) ||
        !ok16(d_o) || nb <= 0 || H <= 0 || nq <= 0 || nk <= 0 || (int64_t)nk * ldk >= (1 << 29) ||
        (int64_t)nq * ldq >= (1 << 29) || (int64_t)nq * ldo >= (1 << 29)) {
        vpu_set_error("xattn_bwd: head dim a multiple of 16 up to 128, 16-byte aligned slices, row strides % 8 == 0, one batch entry per matrix below 1 GiB");
        return VPU_ERR_ARG;
    }
    AttnArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o;
    a.d_o = (const bf16_t*)d_o; a.lse = const_cast<float*>(lse); a.delta = delta;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
    a.nq = nq; a.nk = nk; a.H = H; a.hd = hd; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.ldgq = ldgq; a.ldgk = ldgk;
    a.scale = scale; a.qdiv = qdiv; a.kdiv = kdiv;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!split && lean_enabled() && onepass_enabled() == 2 && hd == 64 && nq == nk && nq <= 32 * WIN_MAX_KB && ldgq % 4 == 0 && ldgk % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(dq) & 7) == 0 && (reinterpret_cast<uintptr_t>(dk) & 7) == 0 && (reinterpret_cast<uintptr_t>(dv) & 7) == 0) {
        // three workgroups per CU (round 5); "onepass" = 1 selects the one-workgroup-per-CU form below
        static VpuDevOnce attrp;
        if (auto todo_ = attrp.pending()) {
            VPU_SET_LDS(WpCfg<1>::LDS, attn_bwd_winp_kernel<1>);
        }
        snprintf(g_last_attn, sizeof(g_last_attn), "attn_bwd_winp_kernel<1>");
        attn_bwd_winp_kernel<1><<<dim3(nb * H), 256, WpCfg<1>::LDS, s>>>(a);
        return vpu_check_launch("vpu_xattn_bwd");
    }
    const bool win_ok = !split && lean_enabled() && hd == 64 && nq == nk && nq <= 32 * WIN_MAX_KB && ldgq % 4 == 0 && ldgk % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(dq) & 7) == 0 && (reinterpret_cast<uintptr_t>(dk) & 7) == 0 && (reinterpret_cast<uintptr_t>(dv) & 7) == 0;
    {
        // the persistent form addresses the operands whole (a problem is an soffset): every operand below 2 GiB
        const int64_t trows = (int64_t)nb * nq, lim = (int64_t)1 << 30;
        const int nkb = (nq + 31) / 32;
        if (win_ok && onepass_enabled() == 3 && nkb >= 3 && nkb <= 7 && trows * ldq < lim && trows * ldk < lim && trows * ldo < lim &&
            trows * ldgq < lim && trows * ldgk < lim && (int64_t)nb * H * nq < lim / 2) {
            // one workgroup per CU the step may claim: a workgroup that finds its CU taken (a collective's kernel in a data-parallel
            // step) would start when another one has finished ALL its problems
            const int ncu = vpu_cu_budget();
            a.nbh = nb * H;
            const int grid = a.nbh < ncu ? a.nbh : ncu;
            snprintf(g_last_attn, sizeof(g_last_attn), "attn_bwd_winx_kernel");
#define WINX_CASE(NKB_)                                                                               \
            case NKB_: {                                                                              \
                static VpuDevOnce attr;                                                               \
                if (auto todo_ = attr.pending()) { VPU_SET_LDS(WxCfg<NKB_>::LDS, attn_bwd_winx_kernel<NKB_>); } \
                attn_bwd_winx_kernel<NKB_><<<dim3(grid), 512, WxCfg<NKB_>::LDS, s>>>(a);              \
            } break;
            switch (nkb) { WINX_CASE(3) WINX_CASE(4) WINX_CASE(5) WINX_CASE(6) WINX_CASE(7) }
#undef WINX_CASE
            return vpu_check_launch("vpu_xattn_bwd");
        }
    }
    if (win_ok && onepass_enabled()) {
        // (8-byte gradient stores: rows and slices 8-byte aligned; anything else takes the two kernels below)
        snprintf(g_last_attn, sizeof(g_last_attn), "attn_bwd_win_kernel");
#define WIN_CASE(NKB_)                                                                           \
        case NKB_: {                                                                             \
            static VpuDevOnce attr;                                                              \
            if (auto todo_ = attr.pending()) { VPU_SET_LDS(win_lds(NKB_), attn_bwd_win_kernel<NKB_>); } \
            attn_bwd_win_kernel<NKB_><<<dim3(nb * H), 512, win_lds(NKB_), s>>>(a);               \
        } break;
        switch ((nq + 31) / 32) {
            WIN_CASE(1) WIN_CASE(2) WIN_CASE(3) WIN_CASE(4) WIN_CASE(5) WIN_CASE(6) WIN_CASE(7) WIN_CASE(8)
        }
#undef WIN_CASE
        return vpu_check_launch("vpu_xattn_bwd");
    }
    if (lean_enabled() && nq % 4 == 0) {
        const bool q2 = nq > 64 && (hd_image(hd) <= 64 || wide_two()), k2 = nk > 64 && hd_image(hd) <= 64;
        dim3 gq(q2 ? (nq + 127) / 128 : (nq + 63) / 64, nb * H), gk(k2 ? (nk + 127) / 128 : (nk + 63) / 64, nb * H);
        AttnArgs ak = a;      // (the two kernels may have different block counts)
        if (xcd_map_enabled() && gq.x > 1) { a.gx = gq.x; a.nbh = nb * H; gq = dim3(gq.x * gq.y); }
        if (xcd_map_enabled() && gk.x > 1) { ak.gx = gk.x; ak.nbh = nb * H; gk = dim3(gk.x * gk.y); }
        snprintf(g_last_attn, sizeof(g_last_attn), "attn_bwd_dq_lean_kernel<%d, %d, %d> attn_bwd_dkdv_lean_kernel<%d, %d, %d>",
                 hd_image(hd), q2 ? 2 : 1, hd_computed(hd), hd_image(hd), k2 ? 2 : 1, hd_computed(hd));
        switch (hd_image(hd) * 4 + (q2 ? 2 : 1)) {     // also writes delta
            case 128 * 4 + 2:
                if (hd <= 96) attn_bwd_dq_lean_kernel<128, 2, 96><<<gq, 256, 0, s>>>(a);
                else attn_bwd_dq_lean_kernel<128, 2, 128><<<gq, 256, 0, s>>>(a);
                break;
            case 32 * 4 + 2: attn_bwd_dq_lean_kernel<32, 2, 32><<<gq, 256, 0, s>>>(a); break;
            case 32 * 4 + 1: attn_bwd_dq_lean_kernel<32, 1, 32><<<gq, 256, 0, s>>>(a); break;
            case 64 * 4 + 2: attn_bwd_dq_lean_kernel<64, 2, 64><<<gq, 256, 0, s>>>(a); break;
            case 64 * 4 + 1: attn_bwd_dq_lean_kernel<64, 1, 64><<<gq, 256, 0, s>>>(a); break;
            default:
                if (hd <= 96) attn_bwd_dq_lean_kernel<128, 1, 96><<<gq, 256, 0, s>>>(a);
                else attn_bwd_dq_lean_kernel<128, 1, 128><<<gq, 256, 0, s>>>(a);
                break;
        }
        switch (hd_image(hd) * 4 + (k2 ? 2 : 1)) {
            case 32 * 4 + 2: attn_bwd_dkdv_lean_kernel<32, 2, 32><<<gk, 256, 0, s>>>(ak); break;
            case 32 * 4 + 1: attn_bwd_dkdv_lean_kernel<32, 1, 32><<<gk, 256, 0, s>>>(ak); break;
            case 64 * 4 + 2: attn_bwd_dkdv_lean_kernel<64, 2, 64><<<gk, 256, 0, s>>>(ak); break;
            case 64 * 4 + 1: attn_bwd_dkdv_lean_kernel<64, 1, 64><<<gk, 256, 0, s>>>(ak); break;
            default:
                if (hd <= 96) attn_bwd_dkdv_lean_kernel<128, 1, 96><<<gk, 256, 0, s>>>(ak);
                else attn_bwd_dkdv_lean_kernel<128, 1, 128><<<gk, 256, 0, s>>>(ak);
                break;
        }
        return vpu_check_launch("vpu_xattn_bwd");
    }
    dim3 gk((nk + 63) / 64, nb * H), gq((nq + 63) / 64, nb * H);
    snprintf(g_last_attn, sizeof(g_last_attn), "attn_bwd_dq_kernel<%d, 1> attn_bwd_dkdv_kernel<%d, 1>", hd_image(hd), hd_image(hd));
    switch (hd_image(hd)) {
        case 32:
            attn_bwd_dq_kernel<32, 1><<<gq, 256, 0, s>>>(a);     // also writes delta
            attn_bwd_dkdv_kernel<32, 1><<<gk, 256, 0, s>>>(a);
            break;
        case 64:
            attn_bwd_dq_kernel<64, 1><<<gq, 256, 0, s>>>(a);     // also writes delta
            attn_bwd_dkdv_kernel<64, 1><<<gk, 256, 0, s>>>(a);
            break;
        default:
            attn_bwd_dq_kernel<128, 1><<<gq, 256, 0, s>>>(a);     // also writes delta
            attn_bwd_dkdv_kernel<128, 1><<<gk, 256, 0, s>>>(a);
            break;
    }
    return vpu_check_launch("vpu_xattn_bwd");
}

extern "C" int vpu_xattn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                             const float* lse, float* delta, void* dq, void* dk, void* dv, int32_t nb, int32_t H,
                             int32_t nq, int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldgq,
                             int32_t ldgk, float scale, void* stream) {
    return xattn_bwd_impl(q, k, v, o, d_o, lse, delta, dq, dk, dv, nb, H, nq, nk, hd, ldq, ldk, ldo, ldgq, ldgk, scale, 1, 1, stream);
}
extern "C" int vpu_xattn_bwd_split(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                   const float* lse, float* delta, void* dq, void* dk, void* dv, int32_t nb, int32_t H,
                                   int32_t nq, int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldgq,
                                   int32_t ldgk, float scale, int32_t qdiv, int32_t kdiv, void* stream) {
    return xattn_bwd_impl(q, k, v, o, d_o, lse, delta, dq, dk, dv, nb, H, nq, nk, hd, ldq, ldk, ldo, ldgq, ldgk, scale, qdiv, kdiv, stream);
}

// self-attention on a fused qkv activation: the same kernels with nq = nk and one row stride
extern "C" int vpu_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int32_t nb, int32_t H,
                            int32_t n, int32_t hd, int32_t ld, int32_t ldo, float scale, void* stream) {
    return vpu_xattn_fwd(q, k, v, out, lse, nb, H, n, n, hd, ld, ld, ldo, scale, stream);
}

extern "C" int vpu_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                            const float* lse, float* delta, void* dq, void* dk, void* dv, int32_t nb, int32_t H,
                            int32_t n, int32_t hd, int32_t ld, int32_t ldo, int32_t ldg, float scale, void* stream) {
    return vpu_xattn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, nb, H, n, n, hd, ld, ld, ldo, ldg, ldg, scale, stream);
}
