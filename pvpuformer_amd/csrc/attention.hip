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
// Whole-chunk kernels (round 2).  The kernels above walk the keys 32 at a time: two barriers, an LDS restage and -- in the
// forward -- one online-softmax step (running max, rescale of the output tile, two cross-lane reductions) per 8 MFMAs,
// ~170 instructions per 16-query x 32-key block.  Here a chunk of up to 208 keys (13 tiles of 16; 112 for head dims above 64)
// is staged ONCE per workgroup as one image per tensor and a wave takes all its score tiles for the chunk before it touches
// the softmax:
//   forward : 26 score MFMAs -> ONE max / exp2 / sum over the lane's 52 values -> 28 P.V MFMAs; a 196-token window is a
//             single chunk, i.e. a plain (not online) softmax with no rescale at all; longer rows (784 global tokens) carry
//             the running max / sum across 4 chunks;
//   backward: no reduction is needed (log-sum-exp and delta are known), so both kernels stream 32-row steps over the
//             resident chunk with no barrier between them.
// One image serves both operand forms: it is laid out for the transposing read (tr_off), and its 16-byte chunks stay
// contiguous under that swizzle, so the row-operand fragments are read from the same image with ds_read_b128
// (conflict-free for 128-byte rows: the eight rows of a lane group at one chunk index land on eight different 16-byte slots).
// ------------------------------------------------------------------------------------------------
template <int HD> struct WC {
    static constexpr int NKT = HD <= 64 ? 13 : 7;       // 16-row tiles per staged chunk
    static constexpr int ROWS = NKT * 16;                // 208 / 112
    static constexpr int NPS = (NKT + 1) / 2;            // 32-row steps of the products that sum over the chunk's rows
    static constexpr int IMG_ROWS = NPS * 32;            // 224 / 128 (rows >= the valid ones are zeros)
    static constexpr int IMG_BYTES = IMG_ROWS * HD * 2;  // 28672 / 32768
    static constexpr int NLD = IMG_ROWS * (HD / 8) / 256;   // 16-byte chunks per thread per image (256 threads)
};

// rows [r0, r0 + IMG_ROWS) of a [*, ld] matrix -> registers (zeros beyond row n / column hd / image row `rows`)
template <int HD>
__device__ __forceinline__ void wc_fetch(const bf16_t* __restrict__ base, int ld, int r0, int n, int hd, int rows, int tid,
                                         uint4 (&v)[WC<HD>::NLD]) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < WC<HD>::NLD; ++i) {
        const int c = tid + i * 256;
        const int row = c / CPR, ch = c % CPR;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (row < rows && r0 + row < n && ch * 8 < hd) x = *reinterpret_cast<const uint4*>(base + (int64_t)(r0 + row) * ld + ch * 8);
        v[i] = x;
    }
}
template <int HD>
__device__ __forceinline__ void wc_put(char* img, int rows, int tid, const uint4 (&v)[WC<HD>::NLD]) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < WC<HD>::NLD; ++i) {
        const int c = tid + i * 256;
        const int row = c / CPR, ch = c % CPR;
        if (row < rows) *reinterpret_cast<uint4*>(img + tr_off<HD>(row, ch * 2)) = v[i];
    }
}
// MFMA row-operand fragment (16 rows from row16, k-step ks) out of a tr-layout image
template <int HD> __device__ __forceinline__ bf16x8_t frag_img(const char* img, int row16, int ks, int lane) {
    const int row = row16 + (lane & 15);
    const int unit = (ks * 4 + (lane >> 4)) * 2;
    const uint4 v = *reinterpret_cast<const uint4*>(img + tr_off<HD>(row, unit));
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_fwd_wc_kernel(const AttnArgs a, const int qpw) {
    using W = WC<HD>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* imgK = lds;
    char* imgV = lds + W::IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.y, bw = bh / a.H, h = bh % a.H, nq = a.nq, nk = a.nk, hd = a.hd;
    const int64_t rbq = (int64_t)bw * nq, rbk = (int64_t)bw * nk;
    const bf16_t* q = a.q + rbq * a.ldq + h * hd;
    const bf16_t* k = a.k + rbk * a.ldk + h * hd;
    const bf16_t* v = a.v + rbk * a.ldk + h * hd;
    const int nchunks = (nk + W::ROWS - 1) / W::ROWS;
    const float sc2 = a.scale * 1.44269504089f;
    for (int qi = 0; qi < qpw; ++qi) {
        const int q0 = ((blockIdx.x * qpw + qi) * 4 + wave) * 16;
        const bool active = q0 < nq;   // wave-uniform; an idle wave still takes part in the staging barriers
        bf16x8_t qf[HD / 32];
        load_rows_as_b<HD>(q, a.ldq, q0, nq, lane, qf, hd);
        float m = -INFINITY, l = 0.f;
        f32x4_t acc[HD / 16];
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) acc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < nchunks; ++ch) {
            const int kc0 = ch * W::ROWS;
            const int kvalid = nk - kc0 < W::ROWS ? nk - kc0 : W::ROWS;   // keys of this chunk
            if (nchunks > 1 || qi == 0) {   // block-uniform
                uint4 rk[W::NLD], rv[W::NLD];
                wc_fetch<HD>(k, a.ldk, kc0, nk, hd, W::ROWS, tid, rk);
                wc_fetch<HD>(v, a.ldk, kc0, nk, hd, W::ROWS, tid, rv);
                __syncthreads();
                wc_put<HD>(imgK, W::IMG_ROWS, tid, rk);
                wc_put<HD>(imgV, W::IMG_ROWS, tid, rv);
                __syncthreads();
            }
            if (!active) continue;
            // every tile of the chunk is computed (rows beyond the chunk's keys are zeros in the image): straight-line code
            // with all 26 score MFMAs independent; only the masking differs -- a chunk that is full up to its last tile (a
            // 196-token window: 12 full tiles + 4 keys) masks that one tile, a shorter one masks by comparison everywhere
            f32x4_t s[W::NKT + 1];
#pragma unroll
            for (int kt = 0; kt < W::NKT; ++kt) {
                s[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < HD / 32; ++ks)
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_img<HD>(imgK, kt * 16, ks, lane), qf[ks], s[kt], 0, 0, 0);
            }
            s[W::NKT] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (kvalid > W::ROWS - 16) {   // uniform
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s[W::NKT - 1][r] = ((W::NKT - 1) * 16 + 4 * g + r < kvalid) ? s[W::NKT - 1][r] : -INFINITY;
            } else {
#pragma unroll
                for (int kt = 0; kt < W::NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][r] = (kt * 16 + 4 * g + r < kvalid) ? s[kt][r] : -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < W::NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // log2 domain: p = exp2(s * sc2 - m): one FMA and one v_exp_f32 per element
            const float mnew = fmaxf(m, mx * sc2);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < W::NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sc2, -mnew));
                    s[kt][r] = p;
                    ps += p;
                }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            l = l * alpha + ps;
            m = mnew;
            if (ch > 0) {   // only rows longer than one chunk ever rescale
                float ar[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
                for (int dt = 0; dt < HD / 16; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[dt][r] *= ar[r];
            }
#pragma unroll
            for (int pss = 0; pss < W::NPS; ++pss) {
                const bf16x8_t pf = pack_pair(s[2 * pss], s[2 * pss + 1]);
#pragma unroll
                for (int dt = 0; dt < HD / 16; ++dt)
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, frag_tr_perm<HD>(imgV + pss * 32 * HD * 2, dt, lane), acc[dt], 0, 0, 0);
            }
        }
        if (active) {
            const float il = 1.0f / l;
            if (g == 0 && q0 + c < nq) a.lse[(int64_t)bh * nq + q0 + c] = (m + __builtin_amdgcn_logf(l)) * 0.69314718056f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = __shfl(il, 4 * g + r, 64);
                const int qq = q0 + 4 * g + r;
                if (qq < nq) {
                    bf16_t* orow = a.out + (rbq + qq) * a.ldo + h * hd;
#pragma unroll
                    for (int dt = 0; dt < HD / 16; ++dt)
                        if (dt * 16 < hd) orow[dt * 16 + c] = (bf16_t)(acc[dt][r] * s1);
                }
            }
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_wc_kernel(const AttnArgs a, const int qpw) {
    using W = WC<HD>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* imgK = lds;
    char* imgV = lds + W::IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.y, bw = bh / a.H, h = bh % a.H, nq = a.nq, nk = a.nk, hd = a.hd;
    const int64_t rbq = (int64_t)bw * nq, rbk = (int64_t)bw * nk;
    const bf16_t* q = a.q + rbq * a.ldq + h * hd;
    const bf16_t* k = a.k + rbk * a.ldk + h * hd;
    const bf16_t* v = a.v + rbk * a.ldk + h * hd;
    const bf16_t* d_o = a.d_o + rbq * a.ldo + h * hd;
    const int nchunks = (nk + W::ROWS - 1) / W::ROWS;
    const float sc2 = a.scale * 1.44269504089f;
    for (int qi = 0; qi < qpw; ++qi) {
        const int q0 = ((blockIdx.x * qpw + qi) * 4 + wave) * 16;
        const bool active = q0 < nq;
        const bool q_ok = q0 + c < nq;
        const float lse2 = q_ok ? a.lse[(int64_t)bh * nq + q0 + c] * 1.44269504089f : 0.f;
        bf16x8_t qf[HD / 32], dof[HD / 32];
        load_rows_as_b<HD>(q, a.ldq, q0, nq, lane, qf, hd);
        load_rows_as_b<HD>(d_o, a.ldo, q0, nq, lane, dof, hd);
        // delta[q] = sum_d dO[q][d] * O[q][d] from the dO fragments the wave holds anyway; published for the dK/dV kernel
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
            if (q_ok && g == 0) a.delta[(int64_t)bh * nq + q0 + c] = dl_q;
        }
        f32x4_t adq[HD / 16];
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) adq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < nchunks; ++ch) {
            const int kc0 = ch * W::ROWS;
            if (nchunks > 1 || qi == 0) {
                uint4 rk[W::NLD], rv[W::NLD];
                wc_fetch<HD>(k, a.ldk, kc0, nk, hd, W::ROWS, tid, rk);
                wc_fetch<HD>(v, a.ldk, kc0, nk, hd, W::ROWS, tid, rv);
                __syncthreads();
                wc_put<HD>(imgK, W::IMG_ROWS, tid, rk);
                wc_put<HD>(imgV, W::IMG_ROWS, tid, rv);
                __syncthreads();
            }
            if (!active) continue;
            // straight-line over the whole image, no masks: a key beyond the chunk has an all-zero K row, so whatever its dS
            // is, it adds nothing to dQ (dQ += dS[q][k] K[k])
#pragma unroll
            for (int pss = 0; pss < W::NPS; ++pss) {
                f32x4_t ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int k16 = pss * 32 + t * 16;
                    f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
                    for (int ks = 0; ks < HD / 32; ++ks) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_img<HD>(imgK, k16, ks, lane), qf[ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_img<HD>(imgV, k16, ks, lane), dof[ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)   // element [key = kc0 + k16 + 4g + r][query = q0 + c]
                        ds[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sc2, -lse2)) * (dp[r] - dl_q);
                }
                const bf16x8_t dsf = pack_pair(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < HD / 16; ++dt)
                    adq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, frag_tr_perm<HD>(imgK + pss * 32 * HD * 2, dt, lane), adq[dt], 0, 0, 0);
            }
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + 4 * g + r;
                if (qq < nq) {
                    bf16_t* qr = a.dq + (rbq + qq) * a.ldgq + h * hd;
#pragma unroll
                    for (int dt = 0; dt < HD / 16; ++dt)
                        if (dt * 16 < hd) qr[dt * 16 + c] = (bf16_t)(adq[dt][r] * a.scale);
                }
            }
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_wc_kernel(const AttnArgs a, const int kpw) {
    using W = WC<HD>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* imgQ = lds;
    char* imgO = lds + W::IMG_BYTES;
    float* lseS = reinterpret_cast<float*>(lds + 2 * W::IMG_BYTES);   // log-sum-exp (log2 domain) and delta of the chunk's queries
    float* dlS = lseS + W::IMG_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.y, bw = bh / a.H, h = bh % a.H, nq = a.nq, nk = a.nk, hd = a.hd;
    const int64_t rbq = (int64_t)bw * nq, rbk = (int64_t)bw * nk;
    const bf16_t* q = a.q + rbq * a.ldq + h * hd;
    const bf16_t* k = a.k + rbk * a.ldk + h * hd;
    const bf16_t* v = a.v + rbk * a.ldk + h * hd;
    const bf16_t* d_o = a.d_o + rbq * a.ldo + h * hd;
    const float* lse = a.lse + (int64_t)bh * nq;
    const float* dl = a.delta + (int64_t)bh * nq;
    const int nchunks = (nq + W::ROWS - 1) / W::ROWS;
    const float sc2 = a.scale * 1.44269504089f;
    for (int ki = 0; ki < kpw; ++ki) {
        const int key0 = ((blockIdx.x * kpw + ki) * 4 + wave) * 16;
        const bool active = key0 < nk;
        bf16x8_t kf[HD / 32], vf[HD / 32];
        load_rows_as_b<HD>(k, a.ldk, key0, nk, lane, kf, hd);
        load_rows_as_b<HD>(v, a.ldk, key0, nk, lane, vf, hd);
        f32x4_t adk[HD / 16], adv[HD / 16];
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) { adk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; adv[dt] = adk[dt]; }
        for (int ch = 0; ch < nchunks; ++ch) {
            const int qc0 = ch * W::ROWS;
            const int qvalid = nq - qc0 < W::ROWS ? nq - qc0 : W::ROWS;
            if (nchunks > 1 || ki == 0) {
                uint4 rq[W::NLD], ro[W::NLD];
                wc_fetch<HD>(q, a.ldq, qc0, nq, hd, W::ROWS, tid, rq);
                wc_fetch<HD>(d_o, a.ldo, qc0, nq, hd, W::ROWS, tid, ro);
                const float l2 = tid < qvalid ? lse[qc0 + tid] * 1.44269504089f : 0.f;
                const float d2 = tid < qvalid ? dl[qc0 + tid] : 0.f;
                __syncthreads();
                wc_put<HD>(imgQ, W::IMG_ROWS, tid, rq);
                wc_put<HD>(imgO, W::IMG_ROWS, tid, ro);
                if (tid < W::IMG_ROWS) { lseS[tid] = l2; dlS[tid] = d2; }
                __syncthreads();
            }
            if (!active) continue;
            // straight-line, no masks: a query beyond the chunk has all-zero Q and dO rows, so its P / dS add nothing to dV
            // (+= P[q][k] dO[q]) and dK (+= dS[q][k] Q[q])
#pragma unroll
            for (int pss = 0; pss < W::NPS; ++pss) {
                f32x4_t P[2], dS[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int q16 = pss * 32 + t * 16;
                    const f32x4_t lv = *reinterpret_cast<const f32x4_t*>(lseS + q16 + 4 * g);
                    const f32x4_t dv4 = *reinterpret_cast<const f32x4_t*>(dlS + q16 + 4 * g);
                    f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
                    for (int ks = 0; ks < HD / 32; ++ks) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_img<HD>(imgQ, q16, ks, lane), kf[ks], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_img<HD>(imgO, q16, ks, lane), vf[ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {   // element [query = qc0 + q16 + 4g + r][key = key0 + c]
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sc2, -lv[r]));
                        P[t][r] = p;
                        dS[t][r] = p * (dp[r] - dv4[r]);
                    }
                }
                const bf16x8_t pf = pack_pair(P[0], P[1]), dsf = pack_pair(dS[0], dS[1]);
#pragma unroll
                for (int dt = 0; dt < HD / 16; ++dt) {
                    adv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, frag_tr_perm<HD>(imgO + pss * 32 * HD * 2, dt, lane), adv[dt], 0, 0, 0);
                    adk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, frag_tr_perm<HD>(imgQ + pss * 32 * HD * 2, dt, lane), adk[dt], 0, 0, 0);
                }
            }
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = key0 + 4 * g + r;
                if (kk < nk) {
                    bf16_t* kr = a.dk + (rbk + kk) * a.ldgk + h * hd;
                    bf16_t* vr = a.dv + (rbk + kk) * a.ldgk + h * hd;
#pragma unroll
                    for (int dt = 0; dt < HD / 16; ++dt)
                        if (dt * 16 < hd) {
                            kr[dt * 16 + c] = (bf16_t)(adk[dt][r] * a.scale);
                            vr[dt * 16 + c] = (bf16_t)adv[dt][r];
                        }
                }
            }
        }
    }
}

// tiles per wave of the whole-chunk kernels: a problem whose reduction side fits one chunk runs in ONE workgroup (every wave
// takes up to four 16-row tiles against the one staged image pair); longer reductions get one tile per wave
template <int HD> inline int wc_tiles_per_wave(int n_own, int n_red) {
    if (n_red > WC<HD>::ROWS) return 1;
    const int t = (n_own + 63) / 64;
    return t < 4 ? t : 4;
}
// Measured (tools/op_bench.py attn, ViT-B bs 12, both builds with MFMA results in VGPRs): window forward 37.6 us against
// 30.4 us for the 32-key-step kernels, window backward 79.8 / 79.1, global forward 79.4 / 77.2, global backward 173.1 /
// 176.9 -- the whole-chunk form issues ~2.5x fewer instructions per score element but runs two workgroups per CU (57 KiB
// of LDS, ~215 VGPRs) where the step kernels run eight, and a (window, head) problem is too short (2.25 workgroups per CU)
// for the leaner stream to make up for the exposed staging latency.  Kept selectable (VPU_ATTN_WC=1 /
// vpu_attn_set_option("whole_chunk", 1)) and tested; the step kernels stay the default.
std::atomic<int> g_opt_wc{-1};
inline bool wc_enabled() {
    static const int e0 = [] { const char* e = getenv("VPU_ATTN_WC"); return e ? atoi(e) : 0; }();
    const int v = g_opt_wc.load(std::memory_order_relaxed);
    return (v >= 0 ? v : e0) != 0;
}
template <typename K> inline void wc_attr(K kern, int bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

inline bool ok16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int hd_image(int hd) { return hd <= 32 ? 32 : (hd <= 64 ? 64 : 128); }

}  // namespace

extern "C" int vpu_attn_set_option(const char* name, int32_t value) {
    vpu_clear_stale_error();
    if (name && !strcmp(name, "whole_chunk") && value >= -1 && value <= 1) {
        g_opt_wc.store(value, std::memory_order_relaxed);
        return VPU_OK;
    }
    vpu_set_error("vpu_attn_set_option: known options: whole_chunk (-1 environment default VPU_ATTN_WC, 0 off, 1 on)");
    return VPU_ERR_ARG;
}

extern "C" int vpu_xattn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int32_t nb, int32_t H,
                             int32_t nq, int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, float scale,
                             void* stream) {
    vpu_clear_stale_error();
    if (hd <= 0 || hd > 128 || hd % 16 || ldq % 8 || ldk % 8 || ldo % 8 || !ok16(q) || !ok16(k) || !ok16(v) || nb <= 0 ||
        H <= 0 || nq <= 0 || nk <= 0) {
        vpu_set_error("xattn_fwd: head dim a multiple of 16 up to 128, 16-byte aligned slices, row strides % 8 == 0");
        return VPU_ERR_ARG;
    }
    AttnArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)out; a.lse = lse;
    a.nq = nq; a.nk = nk; a.H = H; a.hd = hd; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.scale = scale;
    hipStream_t s0 = reinterpret_cast<hipStream_t>(stream);
    if (wc_enabled() && hd_image(hd) >= 64) {
        if (hd_image(hd) == 64) {
            static const bool once = (wc_attr(attn_fwd_wc_kernel<64>, 2 * WC<64>::IMG_BYTES), true);
            (void)once;
            const int qpw = wc_tiles_per_wave<64>(nq, nk);
            attn_fwd_wc_kernel<64><<<dim3((nq + 64 * qpw - 1) / (64 * qpw), nb * H), 256, 2 * WC<64>::IMG_BYTES, s0>>>(a, qpw);
        } else {
            static const bool once = (wc_attr(attn_fwd_wc_kernel<128>, 2 * WC<128>::IMG_BYTES), true);
            (void)once;
            const int qpw = wc_tiles_per_wave<128>(nq, nk);
            attn_fwd_wc_kernel<128><<<dim3((nq + 64 * qpw - 1) / (64 * qpw), nb * H), 256, 2 * WC<128>::IMG_BYTES, s0>>>(a, qpw);
        }
        return vpu_check_launch("vpu_xattn_fwd");
    }
    static const int qt2 = [] { const char* e = getenv("VPU_ATTN_QT"); return e ? atoi(e) : 2; }();
    // two query tiles per wave (128 queries per workgroup) for the 196-token windows: 33.9 vs 35.4 us (bs 12, round 1);
    // for the global blocks one tile per wave stays ahead (84.1 vs 86.0 us).  Time grows linearly with the number of
    // (window, head) problems from ~300 workgroups on (tools/attn_scale.py): the kernel is issue-bound on the softmax's
    // dependent VALU / cross-lane chain (~160 instructions per 32-key step for 8 MFMAs), not on latency or LDS traffic.
    const bool two = qt2 == 2 && nq > 64 && nk <= 256 && hd_image(hd) <= 64;
    dim3 grid(two ? (nq + 127) / 128 : (nq + 63) / 64, nb * H);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (two) {
        if (hd_image(hd) == 32) attn_fwd_kernel<32, 1, 2><<<grid, 256, 0, s>>>(a);
        else attn_fwd_kernel<64, 1, 2><<<grid, 256, 0, s>>>(a);
        return vpu_check_launch("vpu_xattn_fwd");
    }
    switch (hd_image(hd)) {
        // staged block = NS x 32 keys.  NS = 1 everywhere: measured (tools/op_bench.py attn, round 1) window forward 35.8 us
        // at NS = 1, 36.2 at NS = 2, 41.8 at NS = 7 (a whole 196-token window resident, one barrier pair); backward 111.7 /
        // 119.2 / 144.3 us -- the loop is bound by its softmax dependency chain, and the 8-KiB blocks' occupancy (8 per CU)
        // hides more of it than fewer round trips save
        case 32: attn_fwd_kernel<32, 1, 1><<<grid, 256, 0, s>>>(a); break;
        case 64: attn_fwd_kernel<64, 1, 1><<<grid, 256, 0, s>>>(a); break;
        default: attn_fwd_kernel<128, 1, 1><<<grid, 256, 0, s>>>(a); break;
    }
    return vpu_check_launch("vpu_xattn_fwd");
}

extern "C" int vpu_xattn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                             const float* lse, float* delta, void* dq, void* dk, void* dv, int32_t nb, int32_t H,
                             int32_t nq, int32_t nk, int32_t hd, int32_t ldq, int32_t ldk, int32_t ldo, int32_t ldgq,
                             int32_t ldgk, float scale, void* stream) {
    vpu_clear_stale_error();
    if (hd <= 0 || hd > 128 || hd % 16 || ldq % 8 || ldk % 8 || ldo % 8 || !ok16(q) || !ok16(k) || !ok16(v) || !ok16(o) ||
        !ok16(d_o) || nb <= 0 || H <= 0 || nq <= 0 || nk <= 0) {
        vpu_set_error("xattn_bwd: head dim a multiple of 16 up to 128, 16-byte aligned slices, row strides % 8 == 0");
        return VPU_ERR_ARG;
    }
    AttnArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o;
    a.d_o = (const bf16_t*)d_o; a.lse = const_cast<float*>(lse); a.delta = delta;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
    a.nq = nq; a.nk = nk; a.H = H; a.hd = hd; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.ldgq = ldgq; a.ldgk = ldgk;
    a.scale = scale;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (wc_enabled() && hd_image(hd) >= 64) {
        if (hd_image(hd) == 64) {
            constexpr int LQ = 2 * WC<64>::IMG_BYTES, LK = LQ + 2 * WC<64>::IMG_ROWS * 4;
            static const bool once = (wc_attr(attn_bwd_dq_wc_kernel<64>, LQ), wc_attr(attn_bwd_dkdv_wc_kernel<64>, LK), true);
            (void)once;
            const int qpw = wc_tiles_per_wave<64>(nq, nk), kpw = wc_tiles_per_wave<64>(nk, nq);
            attn_bwd_dq_wc_kernel<64><<<dim3((nq + 64 * qpw - 1) / (64 * qpw), nb * H), 256, LQ, s>>>(a, qpw);     // also writes delta
            attn_bwd_dkdv_wc_kernel<64><<<dim3((nk + 64 * kpw - 1) / (64 * kpw), nb * H), 256, LK, s>>>(a, kpw);
        } else {
            constexpr int LQ = 2 * WC<128>::IMG_BYTES, LK = LQ + 2 * WC<128>::IMG_ROWS * 4;
            static const bool once = (wc_attr(attn_bwd_dq_wc_kernel<128>, LQ), wc_attr(attn_bwd_dkdv_wc_kernel<128>, LK), true);
            (void)once;
            const int qpw = wc_tiles_per_wave<128>(nq, nk), kpw = wc_tiles_per_wave<128>(nk, nq);
            attn_bwd_dq_wc_kernel<128><<<dim3((nq + 64 * qpw - 1) / (64 * qpw), nb * H), 256, LQ, s>>>(a, qpw);
            attn_bwd_dkdv_wc_kernel<128><<<dim3((nk + 64 * kpw - 1) / (64 * kpw), nb * H), 256, LK, s>>>(a, kpw);
        }
        return vpu_check_launch("vpu_xattn_bwd");
    }
    dim3 gk((nk + 63) / 64, nb * H), gq((nq + 63) / 64, nb * H);
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
