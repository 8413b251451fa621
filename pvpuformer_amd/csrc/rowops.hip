// Row-wise and element-wise kernels: LayerNorm, softmax, L2 normalise, column sums, broadcasts, casts.
// All HBM-bound; 16-byte vector accesses, one wave (64 lanes) per row, fp32 math.
#include <stdlib.h>
#include "vpu_common.h"
#include "../../include/vpu_hip.h"

#define DISPATCH_T(dtype, ...)                                   \
    if ((dtype) == VPU_BF16) { using T = bf16_t; __VA_ARGS__ }   \
    else if ((dtype) == VPU_F32) { using T = float; __VA_ARGS__ } \
    else { vpu_set_error("bad dtype"); return VPU_ERR_ARG; }

namespace {

constexpr int LN_MAXCH = 4;  // 8-element chunks per lane -> C <= 2048
constexpr int LN_BWD_WAVES = 8;   // part rows: rows / 16, at most 512

// ---------------------------------------------------------------------------------- LayerNorm fwd
// One row per wave and trip; a wave walks rows with the grid stride and requests the NEXT row before it reduces the
// current one (the reductions are two dependent cross-lane chains: without the prefetch every row pays a full HBM round
// trip with nothing in flight).  NCH: 512-column chunks per row.
// PE: a second output y2 = y + pe[row % pe_rows] (the position-embedding add every normalised token / image map of the DMA
// neck goes through before its q / k projections, transformer.py:439-457): the sum is taken from the ROUNDED y, as the
// separate add launch takes it.
template <typename T, int NCH, bool PE, int RW>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int64_t rows, int C, float eps, const T* __restrict__ pe,
                                                            int64_t pe_rows, T* __restrict__ y2) {
    // RW rows per wave and trip (round 5, RW = 2: the two rows' loads are in flight together and their reduction chains
    // interleave); a trip's rows are `half` apart (the number of waves in the grid)
    const int lane = threadIdx.x & 63;
    const int64_t half = (int64_t)gridDim.x * 4, stride = half * RW;
    int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row0 >= rows) return;
    float ww[NCH][8], bb[NCH][8];
    Raw8<T> cur[RW][NCH], nxt[RW][NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + i * 64) * 8;
        if (c < C) {
            load8(w + c, ww[i]); load8(b + c, bb[i]);
#pragma unroll
            for (int k = 0; k < RW; ++k)
                if (row0 + k * half < rows) cur[k][i].load(x + (row0 + k * half) * C + c);
        }
    }
    for (; row0 < rows; row0 += stride) {
        const int64_t nr = row0 + stride;
#pragma unroll
        for (int k = 0; k < RW; ++k)
            if (nr + k * half < rows) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = (lane + i * 64) * 8;
                    if (c < C) nxt[k][i].load(x + (nr + k * half) * C + c);
                }
            }
        float v[RW][NCH][8];
        float s[RW], q[RW], mu[RW], rs[RW];
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            s[k] = 0.f;
            if (row0 + k * half < rows) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = (lane + i * 64) * 8;
                    if (c < C) {
                        cur[k][i].get(v[k][i]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) s[k] += v[k][i][j];
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RW; ++k) mu[k] = wave_sum(s[k]) / C;
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            q[k] = 0.f;
            if (row0 + k * half < rows) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = (lane + i * 64) * 8;
                    if (c < C) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const float d = v[k][i][j] - mu[k]; q[k] += d * d; }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RW; ++k) rs[k] = rsqrtf(wave_sum(q[k]) / C + eps);
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int64_t row = row0 + k * half;
            if (row < rows) {
                if (lane == 0) { mean[row] = mu[k]; rstd[row] = rs[k]; }
                T* yr = y + row * C;
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = (lane + i * 64) * 8;
                    if (c < C) {
                        float o[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (v[k][i][j] - mu[k]) * rs[k] * ww[i][j] + bb[i][j];
                        store8(yr + c, o);
                        if (PE) {
                            float pv[8];
                            load8(pe + (row % pe_rows) * C + c, pv);
#pragma unroll
                            for (int j = 0; j < 8; ++j) o[j] = to_f32(from_f32<T>(o[j])) + pv[j];
                            store8(y2 + row * C + c, o);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RW; ++k)
#pragma unroll
            for (int i = 0; i < NCH; ++i) cur[k][i] = nxt[k][i];
    }
}

// ---------------------------------------------------------------------------------- LayerNorm bwd
// NCH: 512-column chunks per row (C <= 512*NCH); NW waves per workgroup.  The kernel is a latency-bound stream of 3-KB rows (x, dy,
// the residual gradient): a wave walks its rows with the grid stride and requests the next trip's rows before it reduces the
// current ones.  Round 5: RW rows per wave and TRIP (RW = 2 for C <= 1024) -- one row per trip left a wave with one row set in
// flight while it stalled on the previous one (4.6 dependent trips per wave at ViT-B bs 12, a memory round trip each: 15 us for
// 58 MB); two rows per trip halve the trips, double the bytes in flight and give the two cross-lane reduction chains of a trip
// a partner to overlap with.
// D2: the gradient of the output is dy + dy2 (the two outputs of the PE form of the forward), summed in fp32.
template <typename T, int NCH, int NW, bool PF, bool D2, int RW>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                            const T* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const T* __restrict__ dres,
                                                            T* __restrict__ dx, float* __restrict__ part, int64_t rows,
                                                            int C, int nblk) {
    __shared__ float red[NW][8 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dwa[NCH][8], dba[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { dwa[i][j] = 0.f; dba[i][j] = 0.f; }
    float ww[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + i * 64) * 8;
        if (c < C) load8(w + c, ww[i]);
    }
    // a trip = rows row0 + k * half, k < RW (half = the number of waves in the grid): the next trip's x / dy / residual gradient /
    // statistics are requested before the current trip's cross-lane reductions (raw registers, converted only when consumed)
    const int64_t half = (int64_t)nblk * NW, stride = half * RW;
    int64_t row0 = (int64_t)blockIdx.x * NW + wave;
    Raw8<T> cx[RW][NCH], cd[RW][NCH], cr[RW][NCH], nx[RW][NCH], nd[RW][NCH], nres[RW][NCH];
    Raw8<T> cd2[RW][D2 ? NCH : 1], nd2[RW][D2 ? NCH : 1];
    float mu[RW], rs[RW], nmu[RW], nrs[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) { mu[k] = 0.f; rs[k] = 0.f; nmu[k] = 0.f; nrs[k] = 0.f; }
    auto request = [&](int64_t r0, Raw8<T> (&ox)[RW][NCH], Raw8<T> (&od)[RW][NCH], Raw8<T> (&orr)[RW][NCH],
                       Raw8<T> (&od2)[RW][D2 ? NCH : 1], float (&om)[RW], float (&os)[RW]) {
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int64_t r = r0 + k * half;
            if (r < rows) {
                om[k] = mean[r]; os[k] = rstd[r];
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = (lane + i * 64) * 8;
                    if (c < C) {
                        ox[k][i].load(x + r * C + c);
                        od[k][i].load(dy + r * C + c);
                        if (D2) od2[k][i].load(dy2 + r * C + c);
                        if (dres) orr[k][i].load(dres + r * C + c);
                    }
                }
            }
        }
    };
    if (PF && row0 < rows) request(row0, cx, cd, cr, cd2, mu, rs);
    for (; row0 < rows; row0 += stride) {
        if (PF && row0 + stride < rows) request(row0 + stride, nx, nd, nres, nd2, nmu, nrs);
        if (!PF) request(row0, cx, cd, cr, cd2, mu, rs);   // 4 chunks per lane: no registers for a second trip in flight
        // (x-hat and g = dy * w are recomputed from the raw registers in the output loop instead of being kept: 64 registers)
        float s1[RW], s2[RW];
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            s1[k] = 0.f; s2[k] = 0.f;
            if (row0 + k * half < rows) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = (lane + i * 64) * 8;
                    if (c < C) {
                        float xv[8], dv[8];
                        cx[k][i].get(xv);
                        cd[k][i].get(dv);
                        if (D2) {
                            float d2[8];
                            cd2[k][i].get(d2);
#pragma unroll
                            for (int j = 0; j < 8; ++j) dv[j] += d2[j];
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float xh = (xv[j] - mu[k]) * rs[k], g = dv[j] * ww[i][j];
                            s1[k] += g;
                            s2[k] += g * xh;
                            dwa[i][j] += dv[j] * xh;
                            dba[i][j] += dv[j];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RW; ++k) { s1[k] = wave_sum(s1[k]) / C; s2[k] = wave_sum(s2[k]) / C; }
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int64_t row = row0 + k * half;
            if (row < rows) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = (lane + i * 64) * 8;
                    if (c < C) {
                        float o[8], xv[8], dv[8];
                        if (dres) cr[k][i].get(o);
                        else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) o[j] = 0.f;
                        }
                        cx[k][i].get(xv);
                        cd[k][i].get(dv);
                        if (D2) {
                            float d2[8];
                            cd2[k][i].get(d2);
#pragma unroll
                            for (int j = 0; j < 8; ++j) dv[j] += d2[j];
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float xh = (xv[j] - mu[k]) * rs[k], g = dv[j] * ww[i][j];      // (the same operations as above: same bits)
                            o[j] += rs[k] * (g - s1[k] - xh * s2[k]);
                        }
                        store8(dx + row * C + c, o);
                    }
                }
            }
        }
        if (PF) {
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                mu[k] = nmu[k]; rs[k] = nrs[k];
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    cx[k][i] = nx[k][i]; cd[k][i] = nd[k][i]; cr[k][i] = nres[k][i];
                    if (D2) cd2[k][i] = nd2[k][i];
                }
            }
        }
    }
    // reduce the waves' dw / db and write this block's partial row
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + i * 64) * 8;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave][j * 64 + lane] = which ? dba[i][j] : dwa[i][j];
            __syncthreads();
            if (wave == 0 && c < C) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = 0.f;
#pragma unroll
                    for (int w_ = 0; w_ < NW; ++w_) t += red[w_][j * 64 + lane];
                    o[j] = t;
                }
                store8(part + ((int64_t)blockIdx.x * 2 + which) * C + c, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------------- column sums
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         int64_t rows, int C, float beta) {
    __shared__ float red[8][33];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int64_t r = grp;
        for (; r + 8 < rows; r += 16) { s0 += in[r * C + c]; s1 += in[(r + 8) * C + c]; }
        if (r < rows) s0 += in[r * C + c];
    }
    red[grp][cl] = s0 + s1;
    __syncthreads();
    if (grp == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][cl];
        out[c] = (beta != 0.f ? beta * out[c] : 0.f) + t;
    }
}

// Batched form: blockIdx.y = job; out_j[c] += sum_r in_j[r*C_j + c].  The partial parameter-gradient rows that every
// LayerNorm / GroupNorm backward leaves behind (50 per training step) are reduced by ONE launch at the end of backward
// instead of one launch each.
__global__ __launch_bounds__(256) void colsum_batched_kernel(const vpu_colsum_batch jb) {
    __shared__ float red[8][33];
    const vpu_colsum_job& j = jb.job[blockIdx.y];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int C = j.ncols;
    const int64_t rows = j.nrows;
    const float* __restrict__ in = j.in;
    if (rows <= 16 && (C & 3) == 0 && !(j.row_len > 0 && ((j.row_len | j.out_ld) & 3)) &&
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(j.out)) & 15) == 0) {
        // few rows over many columns (the split-K style slabs of Engine._wgrad_sliced: 2-8 rows x ~300k columns): one
        // lane per 4 columns, 16-byte accesses, rows in order (deterministic); block-uniform branch, no barrier below
        const int C4 = C >> 2;
        const int rl = j.row_len;      // (> 0: a strided output block; row_len and out_ld are multiples of 4 here, host-checked)
        for (int c4 = blockIdx.x * 256 + threadIdx.x; c4 < C4; c4 += gridDim.x * 256) {
            const int64_t c = 4 * (int64_t)c4;
            float* o = j.out + (rl > 0 ? (c / rl) * j.out_ld + c % rl : c);
            f32x4_t a = *reinterpret_cast<const f32x4_t*>(o);
            for (int64_t r = 0; r < rows; ++r) a += *reinterpret_cast<const f32x4_t*>(in + r * C + c);
            *reinterpret_cast<f32x4_t*>(o) = a;
        }
        return;
    }
    for (int c0 = blockIdx.x * 32; c0 < C; c0 += gridDim.x * 32) {   // block-uniform trip count
        const int c = c0 + cl;
        float s0 = 0.f, s1 = 0.f;
        if (c < C) {
            int64_t r = grp;
            for (; r + 8 < rows; r += 16) { s0 += in[r * C + c]; s1 += in[(r + 8) * C + c]; }
            if (r < rows) s0 += in[r * C + c];
        }
        __syncthreads();
        red[grp][cl] = s0 + s1;
        __syncthreads();
        if (grp == 0 && c < C) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) t += red[g][cl];
            j.out[j.row_len > 0 ? (int64_t)(c / j.row_len) * j.out_ld + c % j.row_len : c] += t;
        }
    }
}

// few rows (<= 64): one pass, no partials.  out[c] = beta*out[c] + sum_r in[r*ld + c]
template <typename T>
__global__ __launch_bounds__(256) void colsum_small_kernel(const T* __restrict__ in, int64_t ld, float* __restrict__ out,
                                                           int rows, int64_t C, float beta) {
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (c >= C) return;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    for (int r = 0; r < rows; ++r) {
        float v[8];
        load8(in + r * ld + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
    if (beta != 0.f) {
        float o[8];
        load8(out + c, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += beta * o[j];
    }
    store8(out + c, a);
}

constexpr int CS_SLABS = 64;   // rows of the caller's partial buffer ([64][C], include/vpu_hip.h)
// block = CL column-lanes (8 columns each) x 256/CL row groups; grid = (C / (8 CL), 64 row slabs).  CL = 16 for wide maps;
// narrow ones (C <= 256 over 150528 rows: the bias gradient of the FPN's transposed convolutions) take CL = 8 / 4 so that
// the launch still has >= 256 workgroups (two column blocks x 64 slabs = 128 workgroups streamed 77 MB in 29 us).
template <typename T, int CL>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ in, int ld, float* __restrict__ part,
                                                          int64_t rows, int C) {
    constexpr int RG = 256 / CL;
    __shared__ float red[RG][CL * 8 + 1];
    const int cl = threadIdx.x % CL, grp = threadIdx.x / CL;
    const int c = (blockIdx.x * CL + cl) * 8;
    const int64_t per = (rows + CS_SLABS - 1) / CS_SLABS;
    const int64_t r0 = blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    if (c < C) {
        int64_t r = r0 + grp;
        for (; r + RG < r1; r += 2 * RG) {  // two independent 16-byte loads in flight per lane
            float v0[8], v1[8];
            load8(in + r * ld + c, v0);
            load8(in + (r + RG) * ld + c, v1);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += v0[j] + v1[j];
        }
        for (; r < r1; r += RG) {
            float v[8];
            load8(in + r * ld + c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[grp][cl * 8 + j] = a[j];
    __syncthreads();
    if (threadIdx.x < CL * 8) {
        const int cc = blockIdx.x * CL * 8 + threadIdx.x;
        if (cc < C) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < RG; ++g) t += red[g][threadIdx.x];
            part[(int64_t)blockIdx.y * C + cc] = t;
        }
    }
}

// ---------------------------------------------------------------------------------- softmax
// a wave holds a whole row in registers: SM_MAX * 64 columns.  16 covers the training shapes (784 / 1024 keys); 64 the global
// attention of an evaluation at a larger input (672^2: 1764 keys for patch 16, 2304 for patch 14) in the unfused parity path
template <typename T, int SM_MAX>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, int lds_, T* __restrict__ P,
                                                          int ldp, int64_t rows, int ncols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * lds_;
    float v[SM_MAX];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_MAX; ++i) {
        const int c = lane + i * 64;
        v[i] = c < ncols ? s[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAX; ++i) {
        const int c = lane + i * 64;
        v[i] = c < ncols ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
    const float inv = 1.0f / wave_sum(sum);
    T* p = P + row * ldp;
#pragma unroll
    for (int i = 0; i < SM_MAX; ++i) {
        const int c = lane + i * 64;
        if (c < ldp) p[c] = from_f32<T>(v[i] * inv);
    }
}

template <typename T, int SM_MAX>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, int ldp,
                                                          const float* __restrict__ dP, int lddp, T* __restrict__ dS,
                                                          int64_t rows, int ncols, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* p = P + row * ldp;
    const float* d = dP + row * lddp;
    float pv[SM_MAX], dv[SM_MAX];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAX; ++i) {
        const int c = lane + i * 64;
        pv[i] = c < ncols ? to_f32(p[c]) : 0.f;
        dv[i] = c < ncols ? d[c] : 0.f;
        dot += pv[i] * dv[i];
    }
    dot = wave_sum(dot);
    T* o = dS + row * ldp;
#pragma unroll
    for (int i = 0; i < SM_MAX; ++i) {
        const int c = lane + i * 64;
        if (c < ldp) o[c] = from_f32<T>(c < ncols ? pv[i] * (dv[i] - dot) * scale : 0.f);
    }
}

// ---------------------------------------------------------------------------------- L2 normalise
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         float* __restrict__ inv, int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = to_f32(x[row * C + c]); s += v * v; }
    const float nrm = sqrtf(wave_sum(s));
    const float iv = 1.0f / fmaxf(nrm, 1e-12f);
    if (lane == 0) inv[row] = iv;
    for (int c = lane; c < C; c += 64) y[row * C + c] = from_f32<T>(to_f32(x[row * C + c]) * iv);
}
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                         const float* __restrict__ inv, T* __restrict__ dx,
                                                         int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += to_f32(dy[row * C + c]) * to_f32(y[row * C + c]);
    s = wave_sum(s);
    const float iv = inv[row];
    for (int c = lane; c < C; c += 64)
        dx[row * C + c] = from_f32<T>(iv * (to_f32(dy[row * C + c]) - to_f32(y[row * C + c]) * s));
}

// C in {64, 128, 256, 512}: C / 8 lanes per row with 16-byte accesses, the row kept in registers between the reduction and
// the scaling, rows walked with the grid stride (the one-element-per-lane forms above: 52 / 71 us for the head's 150528 x
// 256 map).
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             float* __restrict__ inv, int64_t rows, int C) {
    const int lpr = C >> 3, rpw = 64 / lpr;
    const int lane = threadIdx.x & 63, sub = lane / lpr, cl = lane - sub * lpr;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r0 = wave_id * rpw; r0 < rows; r0 += nwaves * rpw) {
        const int64_t row = r0 + sub;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (row < rows) load8(x + row * C + cl * 8, v);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j] * v[j];
        for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float iv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
        if (row < rows) {
            if (cl == 0) inv[row] = iv;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= iv;
            store8(y + row * C + cl * 8, v);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                             const float* __restrict__ inv, T* __restrict__ dx,
                                                             int64_t rows, int C) {
    const int lpr = C >> 3, rpw = 64 / lpr;
    const int lane = threadIdx.x & 63, sub = lane / lpr, cl = lane - sub * lpr;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r0 = wave_id * rpw; r0 < rows; r0 += nwaves * rpw) {
        const int64_t row = r0 + sub;
        float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, yv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (row < rows) { load8(dy + row * C + cl * 8, d); load8(y + row * C + cl * 8, yv); }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += d[j] * yv[j];
        for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (row < rows) {
            const float iv = inv[row];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = iv * (d[j] - yv[j] * s);
            store8(dx + row * C + cl * 8, d);
        }
    }
}

// ---------------------------------------------------------------------------------- element-wise
template <typename T>
__global__ __launch_bounds__(256) void add_bcast_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                        T* __restrict__ out, int64_t nchunk, int64_t pchunk) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * 256) {
        float x[8], y[8];
        load8(a + i * 8, x);
        load8(b + (i % pchunk) * 8, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        store8(out + i * 8, x);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void add4_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                   const T* __restrict__ c, const T* __restrict__ d,
                                                   T* __restrict__ out, int64_t nchunk) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * 256) {
        float x[8], y[8];
        load8(a + i * 8, x);
        if (b) { load8(b + i * 8, y);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += y[j]; }
        if (c) { load8(c + i * 8, y);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += y[j]; }
        if (d) { load8(d + i * 8, y);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += y[j]; }
        store8(out + i * 8, x);
    }
}
// out[g][i] = sum_s in[g][s][i] (fp32 sum in order s = 0, 1, ...: deterministic), 8 elements per thread: the partial sums a split
// attention backward leaves per key / query range
template <typename T>
__global__ __launch_bounds__(256) void sum_groups_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t G, int S, int64_t nchunk) {
    const int64_t total = G * nchunk;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / nchunk, c = i - g * nchunk;
        float acc[8];
        load8(in + (g * S * nchunk + c) * 8, acc);
        for (int sp = 1; sp < S; ++sp) {
            float v[8];
            load8(in + ((g * S + sp) * nchunk + c) * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        store8(out + i * 8, acc);
    }
}
// the adjoint of add4: every dst_i (+)= src, ONE launch (q_out = q + q_1 + q_2 + q_3 hands its gradient to four tensors)
struct FanOut { void* dst[4]; int accum[4]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void fanout_add_kernel(const T* __restrict__ src, const FanOut f, int64_t nchunk) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * 256) {
        float x[8];
        load8(src + i * 8, x);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < f.n) {
                T* d = (T*)f.dst[k];
                float y[8];
                if (f.accum[k]) {
                    load8(d + i * 8, y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) y[j] += x[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) y[j] = x[j];
                }
                store8(d + i * 8, y);
            }
    }
}
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast2d_kernel(const TS* __restrict__ src, int64_t ld_src, TD* __restrict__ dst,
                                                     int64_t ld_dst, int64_t rows, int cols, int cols_pad) {
    const int64_t total = rows * cols_pad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols_pad;
        const int c = (int)(i - r * cols_pad);
        dst[r * ld_dst + c] = from_f32<TD>(c < cols ? to_f32(src[r * ld_src + c]) : 0.f);
    }
}
// up to CAST_MAXJ strided 2-D casts of fp32 sources in ONE launch (grid y = job): dst[map(r)][c] = src[r][c] (+ src2[r][c]), zeros
// in the pad columns; map = identity, or the window order of a g x g token grid (perm_g > 0: row r is a raster index, see
// window_permute_kernel) -- the per-step derived operands of Engine.refresh_weights (fused patch-embed weight and bias, the
// K-padded PuE weight, the window-ordered position embedding) were five launches
constexpr int CAST_MAXJ = 8;
struct CastJob {
    const float* src;
    const float* src2;
    void* dst;
    long long ld_src, ld_dst, rows;
    int cols, cols_pad, dst_dtype, perm_g, perm_wg;
};
struct CastBatch { CastJob j[CAST_MAXJ]; };
__global__ __launch_bounds__(256) void cast2d_batched_kernel(const CastBatch b) {
    const CastJob& q = b.j[blockIdx.y];
    const int64_t total = q.rows * q.cols_pad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / q.cols_pad;
        const int c = (int)(i - r * q.cols_pad);
        float v = 0.f;
        if (c < q.cols) {
            v = q.src[r * q.ld_src + c];
            if (q.src2) v += q.src2[r * q.ld_src + c];
        }
        int64_t ro = r;
        if (q.perm_g > 0) {
            const int g = q.perm_g, wg = q.perm_wg, nw = g / wg, t = (int)r, ty = t / g, tx = t % g;
            ro = ((ty / wg) * nw + tx / wg) * wg * wg + (ty % wg) * wg + tx % wg;
        }
        if (q.dst_dtype == VPU_BF16) reinterpret_cast<bf16_t*>(q.dst)[ro * q.ld_dst + c] = (bf16_t)v;
        else reinterpret_cast<float*>(q.dst)[ro * q.ld_dst + c] = v;
    }
}
// dz = dy * act'(aux) on a strided 2-D view (rows x cols, cols % 8 == 0); kind 0: relu (aux = output or input),
// 1: gelu (aux = pre-activation)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, int64_t ld_dy, const T* __restrict__ aux,
                                                      int64_t ld_aux, T* __restrict__ dz, int64_t ld_dz, int64_t rows,
                                                      int cols, int kind) {
    const int chunks = cols / 8;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks) * 8;
        float d[8], a[8];
        load8(dy + r * ld_dy + c, d);
        load8(aux + r * ld_aux + c, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] *= kind ? dgelu_f(a[j]) : (a[j] > 0.f ? 1.f : 0.f);
        store8(dz + r * ld_dz + c, d);
    }
}
// prev_mask = sigmoid(logits) written straight into channel `ch` of the next iteration's [B,C,H,W] network input
// (trainer.py:428 + :384)
__global__ __launch_bounds__(256) void sigmoid_to_channel_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                                 int64_t HW, int64_t batch_stride, int64_t ch_off,
                                                                 int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / HW, r = i - b * HW;
        out[b * batch_stride + ch_off + r] = 1.0f / (1.0f + expf(-logits[i]));
    }
}
// Dropout2d channel mask (decode_head.py:82-86,210-215: one Bernoulli(keep) draw per (sample, channel), scaled by 1 / keep):
// a counter-based generator -- splitmix64 of (seed, call number, element) -- with the call number kept in device memory
// and advanced by the kernel itself, so a hipGraph replay of the step draws a fresh mask every time without host work.
// One workgroup: every thread reads the call number before the barrier, thread 0 advances it after.
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, int n, float keep, float inv_keep,
                                                           unsigned long long seed, unsigned long long* __restrict__ state) {
    // state[0]: call number; state[1]: seed word kept in device memory (XORed into the `seed` argument: a launch captured
    // in a hipGraph follows a later re-seeding, and (seed, call number) can be saved and restored with a checkpoint)
    const unsigned long long call = state[0];
    seed ^= state[1];
    __syncthreads();
    if (threadIdx.x == 0) state[0] = call + 1;
    for (int i = threadIdx.x; i < n; i += 256) {
        unsigned long long x = seed + call * 0x9E3779B97F4A7C15ULL + (unsigned long long)(i + 1) * 0xD1B54A32D192ED03ULL;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
        x ^= x >> 27; x *= 0x94D049BB133111EBULL;
        x ^= x >> 31;
        const float u = (float)(x >> 40) * (1.0f / 16777216.0f);      // 24 uniform bits in [0, 1)
        out[i] = u < keep ? inv_keep : 0.f;
    }
}
__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, int64_t n) {
    // 16-byte stores over the aligned body (the 489-MB gradient buffer is cleared with this every step: 63 us), scalar
    // stores for the unaligned head / tail
    const int64_t head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2;
    const int64_t h = head < n ? head : n, n4 = (n - h) >> 2;
    float4* q = reinterpret_cast<float4*>(p + h);
    const float4 v4 = make_float4(v, v, v, v);
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    for (int64_t i = tid; i < n4; i += nth) q[i] = v4;
    if (tid < h) p[tid] = v;
    const int64_t tail = h + 4 * n4;
    if (tail + tid < n) p[tail + tid] = v;
}

// p[off[r] .. off[r] + len[r]) <- v for up to FILL_MAX_RANGES ranges in ONE launch (round 4: the gradient buffer minus the
// weights whose gradient is written, not accumulated, by the one GEMM that produces it).  A range is cut into chunks of
// FILL_CHUNK floats, a workgroup takes one chunk; offsets and lengths are multiples of 4 floats from a 16-byte aligned base.
constexpr int FILL_MAX_RANGES = 160;
constexpr int FILL_CHUNK = 16384;
struct FillRanges {
    int n;
    int first_chunk[FILL_MAX_RANGES + 1];
    long long off[FILL_MAX_RANGES], len[FILL_MAX_RANGES];
};
__global__ __launch_bounds__(256) void fill_ranges_kernel(float* __restrict__ base, const FillRanges r, const float v) {
    const int c = blockIdx.x;
    int lo = 0, hi = r.n;          // the range whose chunk interval holds c
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (r.first_chunk[mid] <= c) lo = mid; else hi = mid;
    }
    const long long start = (long long)(c - r.first_chunk[lo]) * FILL_CHUNK;
    const long long end = start + FILL_CHUNK < r.len[lo] ? start + FILL_CHUNK : r.len[lo];
    float4* q = reinterpret_cast<float4*>(base + r.off[lo]);
    const float4 v4 = make_float4(v, v, v, v);
    for (long long i = (start >> 2) + threadIdx.x; i < (end >> 2); i += 256) q[i] = v4;
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int vpu_layernorm_fwd_pe(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                    int64_t rows, int32_t C, float eps, const void* pe, int64_t pe_rows, void* y2,
                                    int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || C > LN_MAXCH * 512 || rows <= 0) { vpu_set_error("layernorm: C % 8 == 0, C <= 2048"); return VPU_ERR_ARG; }
    if ((pe != nullptr) != (y2 != nullptr) || (pe && pe_rows <= 0)) {
        vpu_set_error("layernorm_fwd_pe: pe, pe_rows > 0 and y2 go together");
        return VPU_ERR_ARG;
    }
    // balanced persistent grid: at most 2048 workgroups (8 per CU), every wave the same number of rows (+-1)
    static const int64_t capf = [] { const char* e = vpu_lab_getenv("VPU_LN_FWD_CAP"); const int v = e ? atoi(e) : 2048; return (int64_t)(v < 256 ? 256 : v); }();
    static const int rwf = [] { const char* e = vpu_lab_getenv("VPU_LN_FWD_RW"); return e ? atoi(e) : 1; }();     // rows per wave and trip (2: measured level with 1 in the step, 12.40 vs 12.40 ms -- unlike the backward)
    const int rw = (rwf == 2 && C <= 1024 && dtype == VPU_BF16 && !pe && rows >= 4096) ? 2 : 1;
    const int64_t nb = (rows + 4 * rw - 1) / (4 * rw), trips = (nb + capf - 1) / capf;
    const unsigned grid = (unsigned)((nb + trips - 1) / trips);
#define VPU_LN_FWD_(NCH_, RW_)                                                                                        \
    if (pe) { DISPATCH_T(dtype, (layernorm_fwd_kernel<T, NCH_, true, RW_><<<grid, 256, 0, ST>>>((const T*)x, w, b, (T*)y, mean, rstd, rows, C, eps, (const T*)pe, pe_rows, (T*)y2));) } \
    else { DISPATCH_T(dtype, (layernorm_fwd_kernel<T, NCH_, false, RW_><<<grid, 256, 0, ST>>>((const T*)x, w, b, (T*)y, mean, rstd, rows, C, eps, (const T*)nullptr, 1, (T*)nullptr));) }
#define VPU_LN_FWD(NCH_) if (rw == 2) { VPU_LN_FWD_(NCH_, 2) } else { VPU_LN_FWD_(NCH_, 1) }
    if (C <= 512) { VPU_LN_FWD(1) } else if (C <= 1024) { VPU_LN_FWD(2) } else { VPU_LN_FWD_(4, 1) }
#undef VPU_LN_FWD
#undef VPU_LN_FWD_
    return vpu_check_launch("vpu_layernorm_fwd");
}
extern "C" int vpu_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                 int64_t rows, int32_t C, float eps, int32_t dtype, void* stream) {
    return vpu_layernorm_fwd_pe(x, w, b, y, mean, rstd, rows, C, eps, nullptr, 0, nullptr, dtype, stream);
}
extern "C" int vpu_layernorm_bwd_nblk(int64_t rows) {
    vpu_clear_stale_error();
    int64_t n = rows / (2 * LN_BWD_WAVES);
    // workgroups per CU: 2 (round 1), VPU_LN_BWD_WGS = 3: the 4-wave form's ~170 registers allow three
    static const int cap = [] { const char* e = vpu_lab_getenv("VPU_LN_BWD_WGS"); const int v = e ? atoi(e) : 2; return 256 * (v < 1 ? 1 : (v > 4 ? 4 : v)); }();
    return (int)(n < 1 ? 1 : (n > cap ? cap : n));
}
extern "C" int vpu_layernorm_bwd2(const void* dy, const void* dy2, const void* x, const float* w, const float* mean,
                                  const float* rstd, const void* dres, void* dx, float* part, int64_t rows, int32_t C,
                                  int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || C > LN_MAXCH * 512 || rows <= 0) { vpu_set_error("layernorm_bwd: C"); return VPU_ERR_ARG; }
    const int nblk = vpu_layernorm_bwd_nblk(rows);
    static const int rw2 = [] { const char* e = vpu_lab_getenv("VPU_LN_BWD_RW"); return e ? atoi(e) : 2; }();     // rows per wave and trip (A/B: 1)
#define VPU_LN_BWD_(NCH_, NW_, RW_)                                                                                   \
    if (dy2) { DISPATCH_T(dtype, (layernorm_bwd_kernel<T, NCH_, NW_, (NCH_ <= 2), true, RW_><<<nblk, 64 * NW_, 0, ST>>>(   \
                          (const T*)dy, (const T*)dy2, (const T*)x, w, mean, rstd, (const T*)dres, (T*)dx, part, rows, C, nblk));) } \
    else { DISPATCH_T(dtype, (layernorm_bwd_kernel<T, NCH_, NW_, (NCH_ <= 2), false, RW_><<<nblk, 64 * NW_, 0, ST>>>(      \
                          (const T*)dy, (const T*)nullptr, (const T*)x, w, mean, rstd, (const T*)dres, (T*)dx, part, rows, C, nblk));) }
#define VPU_LN_BWD(NCH_, NW_)                                                                                         \
    if (NCH_ <= 2 && rw2 == 2 && dtype == VPU_BF16 && !dy2) { VPU_LN_BWD_(NCH_, NW_, 2) } else { VPU_LN_BWD_(NCH_, NW_, 1) }   /* (two gradients: 271 registers) */
    // (C > 1024 keeps 8 waves: its 4 chunks per lane need the 256-register budget)
    // (<= 2 chunks: 4-wave workgroups, ~170 VGPRs with the prefetched row -> three per CU; 4 chunks: no prefetch)
    if (C <= 512) { VPU_LN_BWD(1, 4) } else if (C <= 1024) { VPU_LN_BWD(2, 4) } else { VPU_LN_BWD(4, 8) }
#undef VPU_LN_BWD
#undef VPU_LN_BWD_
    return vpu_check_launch("vpu_layernorm_bwd");
}
extern "C" int vpu_layernorm_bwd(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                                 const void* dres, void* dx, float* part, int64_t rows, int32_t C, int32_t dtype,
                                 void* stream) {
    return vpu_layernorm_bwd2(dy, nullptr, x, w, mean, rstd, dres, dx, part, rows, C, dtype, stream);
}
extern "C" int vpu_colsum_f32(const float* in, float* out, int64_t rows, int32_t C, float beta, void* stream) {
    vpu_clear_stale_error();
    colsum_f32_kernel<<<(C + 31) / 32, 256, 0, ST>>>(in, out, rows, C, beta);
    return vpu_check_launch("vpu_colsum_f32");
}
extern "C" int vpu_colsum_batched(const vpu_colsum_job* jobs, int32_t n, void* stream) {
    vpu_clear_stale_error();
    if (!jobs || n < 1 || n > VPU_COLSUM_BATCH_MAX) { vpu_set_error("colsum_batched: 1 <= n <= VPU_COLSUM_BATCH_MAX"); return VPU_ERR_ARG; }
    vpu_colsum_batch jb;
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        if (!jobs[i].in || !jobs[i].out || jobs[i].nrows < 1 || jobs[i].ncols < 1) {
            vpu_set_error("colsum_batched: null pointer or empty job");
            return VPU_ERR_ARG;
        }
        if (jobs[i].row_len < 0 || (jobs[i].row_len > 0 && (jobs[i].ncols % jobs[i].row_len || jobs[i].out_ld < jobs[i].row_len))) {
            vpu_set_error("colsum_batched: row_len must divide ncols and out_ld >= row_len");
            return VPU_ERR_ARG;
        }
        jb.job[i] = jobs[i];
        // the 16-byte few-row form needs every output row start 16-byte aligned: otherwise the job runs as the generic form
        // blocks this job can use: 32 columns each, or (few-row form, same test as in the kernel) 1024 columns each
        const bool few = jobs[i].nrows <= 16 && (jobs[i].ncols & 3) == 0 && !(jobs[i].row_len > 0 && ((jobs[i].row_len | jobs[i].out_ld) & 3)) &&
                         ((reinterpret_cast<uintptr_t>(jobs[i].in) | reinterpret_cast<uintptr_t>(jobs[i].out)) & 15) == 0;
        const int need = few ? (jobs[i].ncols + 1023) / 1024 : (jobs[i].ncols + 31) / 32;
        cmax = need > cmax ? need : cmax;
    }
    dim3 grid(cmax, n);
    colsum_batched_kernel<<<grid, 256, 0, ST>>>(jb);
    return vpu_check_launch("vpu_colsum_batched");
}
extern "C" int vpu_colsum(const void* in, int32_t ld, float* out, float* part, int64_t rows, int32_t C, float beta,
                          int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || ld % 8) { vpu_set_error("colsum: C, ld % 8"); return VPU_ERR_ARG; }
    if (rows <= 64) {  // e.g. the batch sum of the pos_embed gradient: rows = B, C = tokens*dim
        const unsigned g = (unsigned)((C / 8 + 255) / 256);
        DISPATCH_T(dtype, colsum_small_kernel<T><<<g, 256, 0, ST>>>((const T*)in, ld, out, (int)rows, C, beta);)
        return vpu_check_launch("vpu_colsum");
    }
    if (C <= 128) {
        dim3 grid((C + 31) / 32, CS_SLABS);
        DISPATCH_T(dtype, colsum_part_kernel<T, 4><<<grid, 256, 0, ST>>>((const T*)in, ld, part, rows, C);)
    } else if (C <= 256) {
        dim3 grid((C + 63) / 64, CS_SLABS);
        DISPATCH_T(dtype, colsum_part_kernel<T, 8><<<grid, 256, 0, ST>>>((const T*)in, ld, part, rows, C);)
    } else {
        dim3 grid((C + 127) / 128, CS_SLABS);
        DISPATCH_T(dtype, colsum_part_kernel<T, 16><<<grid, 256, 0, ST>>>((const T*)in, ld, part, rows, C);)
    }
    colsum_f32_kernel<<<(C + 31) / 32, 256, 0, ST>>>(part, out, CS_SLABS, C, beta);
    return vpu_check_launch("vpu_colsum");
}
extern "C" int vpu_softmax_fwd(const float* S, int32_t lds_, void* P, int32_t ldp, int64_t rows, int32_t ncols,
                               int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (ncols > 4096 || ldp > 4096 || ldp < ncols) { vpu_set_error("softmax: ncols <= 4096"); return VPU_ERR_ARG; }
    if (ldp <= 1024) { DISPATCH_T(dtype, (softmax_fwd_kernel<T, 16><<<(unsigned)((rows + 3) / 4), 256, 0, ST>>>(S, lds_, (T*)P, ldp, rows, ncols));) }
    else { DISPATCH_T(dtype, (softmax_fwd_kernel<T, 64><<<(unsigned)((rows + 3) / 4), 256, 0, ST>>>(S, lds_, (T*)P, ldp, rows, ncols));) }
    return vpu_check_launch("vpu_softmax_fwd");
}
extern "C" int vpu_softmax_bwd(const void* P, int32_t ldp, const float* dP, int32_t lddp, void* dS, int64_t rows,
                               int32_t ncols, float scale, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (ncols > 4096 || ldp > 4096 || ldp < ncols) { vpu_set_error("softmax_bwd: ncols <= 4096"); return VPU_ERR_ARG; }
    if (ldp <= 1024) { DISPATCH_T(dtype, (softmax_bwd_kernel<T, 16><<<(unsigned)((rows + 3) / 4), 256, 0, ST>>>((const T*)P, ldp, dP, lddp, (T*)dS, rows, ncols, scale));) }
    else { DISPATCH_T(dtype, (softmax_bwd_kernel<T, 64><<<(unsigned)((rows + 3) / 4), 256, 0, ST>>>((const T*)P, ldp, dP, lddp, (T*)dS, rows, ncols, scale));) }
    return vpu_check_launch("vpu_softmax_bwd");
}
extern "C" int vpu_l2norm_fwd(const void* x, void* y, float* inv, int64_t rows, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if ((C == 64 || C == 128 || C == 256 || C == 512) && (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 32 == 0) {
        const int64_t trips = (rows * (C / 8) + 63) / 64;
        const unsigned grid = (unsigned)(trips / 4 < 1 ? 1 : (trips / 4 > 8192 ? 8192 : trips / 4));
        DISPATCH_T(dtype, l2norm_fwd_vec_kernel<T><<<grid, 256, 0, ST>>>((const T*)x, (T*)y, inv, rows, C);)
        return vpu_check_launch("vpu_l2norm_fwd");
    }
    DISPATCH_T(dtype, l2norm_fwd_kernel<T><<<(unsigned)((rows + 3) / 4), 256, 0, ST>>>((const T*)x, (T*)y, inv, rows, C);)
    return vpu_check_launch("vpu_l2norm_fwd");
}
extern "C" int vpu_l2norm_bwd(const void* dy, const void* y, const float* inv, void* dx, int64_t rows, int32_t C,
                              int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if ((C == 64 || C == 128 || C == 256 || C == 512) &&
        (reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dx)) % 32 == 0) {
        const int64_t trips = (rows * (C / 8) + 63) / 64;
        const unsigned grid = (unsigned)(trips / 4 < 1 ? 1 : (trips / 4 > 8192 ? 8192 : trips / 4));
        DISPATCH_T(dtype, l2norm_bwd_vec_kernel<T><<<grid, 256, 0, ST>>>((const T*)dy, (const T*)y, inv, (T*)dx, rows, C);)
        return vpu_check_launch("vpu_l2norm_bwd");
    }
    DISPATCH_T(dtype, l2norm_bwd_kernel<T><<<(unsigned)((rows + 3) / 4), 256, 0, ST>>>((const T*)dy, (const T*)y, inv,
                                                                                      (T*)dx, rows, C);)
    return vpu_check_launch("vpu_l2norm_bwd");
}
extern "C" int vpu_add_bcast(const void* a, const void* b, void* out, int64_t n, int64_t period_b, int32_t dtype,
                             void* stream) {
    vpu_clear_stale_error();
    if (n % 8 || period_b % 8 || period_b <= 0) { vpu_set_error("add_bcast: n, period % 8"); return VPU_ERR_ARG; }
    const int grid = vpu_grid_for(n / 8, 256, 4096);
    DISPATCH_T(dtype, add_bcast_kernel<T><<<grid, 256, 0, ST>>>((const T*)a, (const T*)b, (T*)out, n / 8, period_b / 8);)
    return vpu_check_launch("vpu_add_bcast");
}
extern "C" int vpu_add4(const void* a, const void* b, const void* c, const void* d, void* out, int64_t n,
                        int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (n % 8) { vpu_set_error("add4: n % 8"); return VPU_ERR_ARG; }
    const int grid = vpu_grid_for(n / 8, 256, 4096);
    DISPATCH_T(dtype, add4_kernel<T><<<grid, 256, 0, ST>>>((const T*)a, (const T*)b, (const T*)c, (const T*)d, (T*)out, n / 8);)
    return vpu_check_launch("vpu_add4");
}
extern "C" int vpu_sum_groups(const void* in, void* out, int64_t G, int32_t S, int64_t n, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (!in || !out || G < 1 || S < 1 || n < 8 || n % 8) { vpu_set_error("sum_groups: non-null operands, n % 8 == 0"); return VPU_ERR_ARG; }
    DISPATCH_T(dtype, sum_groups_kernel<T><<<vpu_grid_for(G * (n / 8), 256, 4096), 256, 0, ST>>>((const T*)in, (T*)out, G, S, n / 8);)
    return vpu_check_launch("vpu_sum_groups");
}
extern "C" int vpu_fanout_add(const void* src, void* const* dst, const int32_t* accum, int32_t ndst, int64_t n, int32_t dtype,
                              void* stream) {
    vpu_clear_stale_error();
    if (!src || !dst || !accum || ndst < 1 || ndst > 4 || n % 8) { vpu_set_error("fanout_add: 1 <= ndst <= 4, n % 8 == 0"); return VPU_ERR_ARG; }
    FanOut f{};
    f.n = ndst;
    for (int i = 0; i < ndst; ++i) {
        if (!dst[i]) { vpu_set_error("fanout_add: null destination"); return VPU_ERR_ARG; }
        f.dst[i] = dst[i]; f.accum[i] = accum[i];
    }
    const int grid = vpu_grid_for(n / 8, 256, 4096);
    DISPATCH_T(dtype, fanout_add_kernel<T><<<grid, 256, 0, ST>>>((const T*)src, f, n / 8);)
    return vpu_check_launch("vpu_fanout_add");
}
extern "C" int vpu_cast2d(const void* src, int32_t src_dtype, int64_t ld_src, void* dst, int32_t dst_dtype,
                          int64_t ld_dst, int64_t rows, int32_t cols, int32_t cols_pad, void* stream) {
    vpu_clear_stale_error();
    const int grid = vpu_grid_for(rows * cols_pad, 256, 8192);
    if (src_dtype == VPU_F32 && dst_dtype == VPU_BF16)
        cast2d_kernel<float, bf16_t><<<grid, 256, 0, ST>>>((const float*)src, ld_src, (bf16_t*)dst, ld_dst, rows, cols, cols_pad);
    else if (src_dtype == VPU_F32 && dst_dtype == VPU_F32)
        cast2d_kernel<float, float><<<grid, 256, 0, ST>>>((const float*)src, ld_src, (float*)dst, ld_dst, rows, cols, cols_pad);
    else if (src_dtype == VPU_BF16 && dst_dtype == VPU_F32)
        cast2d_kernel<bf16_t, float><<<grid, 256, 0, ST>>>((const bf16_t*)src, ld_src, (float*)dst, ld_dst, rows, cols, cols_pad);
    else if (src_dtype == VPU_BF16 && dst_dtype == VPU_BF16)
        cast2d_kernel<bf16_t, bf16_t><<<grid, 256, 0, ST>>>((const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, rows, cols, cols_pad);
    else { vpu_set_error("cast2d: dtype"); return VPU_ERR_ARG; }
    return vpu_check_launch("vpu_cast2d");
}
extern "C" int vpu_cast2d_batched(const vpu_cast_job* jobs, int32_t n, void* stream) {
    vpu_clear_stale_error();
    if (!jobs || n < 1 || n > CAST_MAXJ) { vpu_set_error("cast2d_batched: 1 <= n <= 8 jobs"); return VPU_ERR_ARG; }
    CastBatch b{};
    int64_t most = 1;
    for (int i = 0; i < n; ++i) {
        const vpu_cast_job& j = jobs[i];
        if (!j.src || !j.dst || j.rows < 1 || j.cols < 1 || j.cols_pad < j.cols || (j.dst_dtype != VPU_BF16 && j.dst_dtype != VPU_F32) ||
            (j.perm_g > 0 && (j.perm_wg < 1 || j.perm_g % j.perm_wg || j.rows != (int64_t)j.perm_g * j.perm_g))) {
            vpu_set_error("cast2d_batched: null operand, empty job, dtype, or a row map that does not fit the rows");
            return VPU_ERR_ARG;
        }
        b.j[i] = CastJob{j.src, j.src2, j.dst, j.ld_src, j.ld_dst, j.rows, j.cols, j.cols_pad, j.dst_dtype, j.perm_g, j.perm_wg};
        most = most > j.rows * j.cols_pad ? most : j.rows * j.cols_pad;
    }
    cast2d_batched_kernel<<<dim3(vpu_grid_for(most, 256, 2048), n), 256, 0, ST>>>(b);
    return vpu_check_launch("vpu_cast2d_batched");
}
extern "C" int vpu_act_bwd(const void* dy, int64_t ld_dy, const void* aux, int64_t ld_aux, void* dz, int64_t ld_dz,
                           int64_t rows, int32_t cols, int32_t kind, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (cols % 8 || ld_dy % 8 || ld_aux % 8 || ld_dz % 8) { vpu_set_error("act_bwd: cols, ld % 8"); return VPU_ERR_ARG; }
    const int grid = vpu_grid_for(rows * (cols / 8), 256, 8192);
    DISPATCH_T(dtype, act_bwd_kernel<T><<<grid, 256, 0, ST>>>((const T*)dy, ld_dy, (const T*)aux, ld_aux, (T*)dz, ld_dz,
                                                             rows, cols, kind);)
    return vpu_check_launch("vpu_act_bwd");
}
extern "C" int vpu_sigmoid_to_channel(const float* logits, float* out, int32_t B, int64_t HW, int32_t channels,
                                      int32_t channel, void* stream) {
    vpu_clear_stale_error();
    const int64_t total = (int64_t)B * HW;
    sigmoid_to_channel_kernel<<<vpu_grid_for(total, 256, 4096), 256, 0, ST>>>(logits, out, HW, (int64_t)channels * HW,
                                                                             (int64_t)channel * HW, total);
    return vpu_check_launch("vpu_sigmoid_to_channel");
}
// Diagnostic: `wgs` workgroups of 512 threads that hold ~96 registers per thread and 16 KiB of LDS and spin for `cycles`
// clock cycles -- the footprint of a collective's channel workgroup, for measuring what such workgroups do to the persistent
// GEMM launches on ONE GPU (tools/reserve_cus_experiment.py).  Every wave leaves after the same bounded wait.
namespace {
__global__ __launch_bounds__(512) void debug_spin_kernel(float* sink, long long cycles) {
    __shared__ float pad[4096];
    float r[80];
#pragma unroll
    for (int i = 0; i < 80; ++i) r[i] = (float)(threadIdx.x + i);
    pad[threadIdx.x] = r[3];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles / 24) {   // wall_clock64 ticks at 100 MHz: ~24 shader cycles per tick
#pragma unroll
        for (int i = 0; i < 80; ++i) r[i] = r[i] * 1.0001f + pad[(threadIdx.x + i) & 4095];
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 80; ++i) t += r[i];
    if (t == 1.2345e30f) sink[0] = t;      // keeps the registers live
}
}  // namespace
extern "C" int vpu_debug_spin(float* sink, int32_t wgs, int64_t cycles, void* stream) {
    vpu_clear_stale_error();
    if (wgs < 1 || wgs > 256 || cycles < 0 || cycles > 240000000LL) { vpu_set_error("debug_spin: 1 <= wgs <= 256, cycles <= 2.4e8"); return VPU_ERR_ARG; }
    debug_spin_kernel<<<wgs, 512, 0, ST>>>(sink, (long long)cycles);
    return vpu_check_launch("vpu_debug_spin");
}
extern "C" int vpu_dropout_mask(float* out, int32_t n, float keep, uint64_t seed, uint64_t* state, void* stream) {
    vpu_clear_stale_error();
    if (!out || !state || n < 1 || n > (1 << 22) || !(keep > 0.f && keep <= 1.f)) {
        vpu_set_error("dropout_mask: non-null out / state, 1 <= n <= 2^22, 0 < keep <= 1");
        return VPU_ERR_ARG;
    }
    dropout_mask_kernel<<<1, 256, 0, ST>>>(out, n, keep, 1.0f / keep, (unsigned long long)seed, reinterpret_cast<unsigned long long*>(state));
    return vpu_check_launch("vpu_dropout_mask");
}
extern "C" int vpu_fill_ranges_f32(float* base, const int64_t* off, const int64_t* len, int32_t n, float v, void* stream) {
    vpu_clear_stale_error();
    if (!base || !off || !len || n < 1 || n > FILL_MAX_RANGES || (reinterpret_cast<uintptr_t>(base) & 15)) {
        vpu_set_error("fill_ranges: 1 <= n <= 160 ranges, 16-byte aligned base");
        return VPU_ERR_ARG;
    }
    FillRanges r;
    r.n = n;
    int chunks = 0;
    for (int i = 0; i < n; ++i) {
        if (off[i] < 0 || len[i] <= 0 || (off[i] & 3) || (len[i] & 3)) {
            vpu_set_error("fill_ranges: offsets and lengths must be positive multiples of 4 floats");
            return VPU_ERR_ARG;
        }
        r.first_chunk[i] = chunks;
        r.off[i] = off[i];
        r.len[i] = len[i];
        chunks += (int)((len[i] + FILL_CHUNK - 1) / FILL_CHUNK);
    }
    r.first_chunk[n] = chunks;
    fill_ranges_kernel<<<(unsigned)chunks, 256, 0, ST>>>(base, r, v);
    return vpu_check_launch("vpu_fill_ranges_f32");
}
extern "C" int vpu_fill_f32(float* p, float v, int64_t n, void* stream) {
    vpu_clear_stale_error();
    // one 16-byte store per thread (a grid-stride loop over 4096 workgroups cleared the 489-MB gradient buffer in 74 us, this
    // form in ~63 us: 7.8 TB/s)
    const int64_t n4 = (n >> 2) + 8;
    fill_kernel<<<(unsigned)((n4 + 255) / 256 < 0x7FFFFFFF ? (n4 + 255) / 256 : 0x7FFFFFFF), 256, 0, ST>>>(p, v, n);
    return vpu_check_launch("vpu_fill_f32");
}
