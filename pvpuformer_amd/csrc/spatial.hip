// Spatial kernels of the VPUFormer path on channels-last maps: patch im2col (window token order), token
// permutation, pixel shuffle for 2x2/stride-2 (transposed) convolutions, GroupNorm(1,C)[+GELU], bilinear resize,
// DMA gates, conv_seg, final align_corners=True upsample.  All HBM-bound: vector accesses, fp32 math.
#include <stdlib.h>
#include "vpu_common.h"
#include "../../include/vpu_hip.h"

#define DISPATCH_T(dtype, ...)                                   \
    if ((dtype) == VPU_BF16) { using T = bf16_t; __VA_ARGS__ }   \
    else if ((dtype) == VPU_F32) { using T = float; __VA_ARGS__ } \
    else { vpu_set_error("bad dtype"); return VPU_ERR_ARG; }
#define ST reinterpret_cast<hipStream_t>(stream)

namespace {

__constant__ float c_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_std[3] = {0.229f, 0.224f, 0.225f};

// ------------------------------------------------------------------------------ patch im2col
// cols[(b*T + t_win)][ch*P*P + py*P + px], ch: 0-2 normalised rgb (ops.py:403-407), 3 prev mask, 4-5 disks
// NORM = false: the rgb planes of img4 are normalised already (ISModel.prepare_input ran on the caller's side: the public
// backbone_forward, is_vpu_model.py:383-419)
template <typename T, bool NORM = true>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* __restrict__ img4,
                                                           const float* __restrict__ disks, T* __restrict__ cols,
                                                           int B, int H, int W, int P, int wg) {
    const int gw = W / P, gh = H / P;
    // columns: [image half | coordinate half], each 3*P*P wide and zero-padded to a multiple of 8 (P = 14: 588 -> 592) so
    // that both halves start 16-byte aligned
    const int K3 = 3 * P * P, K3P = (K3 + 7) / 8 * 8;
    const int Tn = gw * gh, KK = 2 * K3P, chunks = KK / 8;
    const int64_t total = (int64_t)B * Tn * chunks;
    const int nwx = gw / wg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t row = i / chunks;
        const int tw = (int)(row % Tn), b = (int)(row / Tn);
        const int win = tw / (wg * wg), inner = tw % (wg * wg);
        const int ty = (win / nwx) * wg + inner / wg, tx = (win % nwx) * wg + inner % wg;
        float o[8];
        if ((P & 7) == 0) {
            // P % 8 == 0 (patch 16): the 8 columns of a chunk are 8 consecutive pixels of one patch row of one channel --
            // one set of index divisions and two 16-byte loads per thread (the element-wise form below: six divisions and
            // a 4-byte load per element, 53 us at ViT-B bs 12)
            const int k = ck * 8;
            const int half = k / K3P, kk = k - half * K3P;
            const int chl = kk / (P * P), rem = kk - chl * (P * P);
            const int ch = half * 3 + chl;
            const int y = ty * P + rem / P, x = tx * P + rem % P;
            const float* src = ch < 4 ? img4 + (((int64_t)b * 4 + ch) * H + y) * W + x
                                      : disks + (((int64_t)b * 2 + (ch - 4)) * H + y) * W + x;
            load8(src, o);
            if (NORM && ch < 3) {
                const float mu = c_mean[ch], isd = c_std[ch];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (o[j] - mu) / isd;
            }
            store8(cols + row * KK + ck * 8, o);
            continue;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ck * 8 + j;
            const int half = k / K3P, kk = k % K3P;
            float v = 0.f;
            if (kk < K3) {
                const int ch = half * 3 + kk / (P * P), rem = kk % (P * P);
                const int y = ty * P + rem / P, x = tx * P + rem % P;
                if (ch < 4) {
                    v = img4[(((int64_t)b * 4 + ch) * H + y) * W + x];
                    if (NORM && ch < 3) v = (v - c_mean[ch]) / c_std[ch];
                } else {
                    v = disks[(((int64_t)b * 2 + (ch - 4)) * H + y) * W + x];
                }
            }
            o[j] = v;
        }
        store8(cols + row * KK + ck * 8, o);
    }
}

// ------------------------------------------------------------------------------ window <-> raster token order
template <typename T>
__global__ __launch_bounds__(256) void window_permute_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int g,
                                                             int wg, int C, int dir) {
    const int Tn = g * g, chunks = C / 8, nw = g / wg;
    const int64_t total = (int64_t)B * Tn * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t row = i / chunks;
        const int t = (int)(row % Tn), b = (int)(row / Tn);
        // t is a raster index; tw its window-order index
        const int ty = t / g, tx = t % g;
        const int tw = ((ty / wg) * nw + tx / wg) * wg * wg + (ty % wg) * wg + tx % wg;
        const int64_t r_rast = (int64_t)b * Tn + t, r_win = (int64_t)b * Tn + tw;
        float v[8];
        if (dir == 0) { load8(x + r_rast * C + ck * 8, v); store8(y + r_win * C + ck * 8, v); }
        else { load8(x + r_win * C + ck * 8, v); store8(y + r_rast * C + ck * 8, v); }
    }
}

// ------------------------------------------------------------------------------ pixel shuffle (2x2)
// dir 0: in [B*h*w][C*4] (col = c*4+di*2+dj) -> out [B][2h][2w][C] (+bias[c])
// dir 1: x [B][2h][2w][C] -> in-layout [B*h*w][C*4]
// One thread moves 8 channels x the 4 sub-pixels of one coarse cell: 32 consecutive elements of the coarse row
// (64 B as 4 x 16-byte accesses; column = c*4 + dd) <-> one 16-byte vector in each of the 4 fine pixels -- the transpose
// happens in registers.  (The first version gathered 2-byte elements at stride 4: 100 us for the 2x 112^2 x 192 map.)
template <typename T>
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                             const float* __restrict__ bias, int B, int h, int w, int C,
                                                             int dir) {
    const int chunks = C / 8;
    const int W2 = 2 * w;
    const int64_t total = (int64_t)B * h * w * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t m = i / chunks;                     // coarse cell (b, y, x)
        const int x = (int)(m % w), y = (int)((m / w) % h);
        const int64_t b = m / ((int64_t)w * h);
        const int64_t coarse = m * (C * 4) + (int64_t)ck * 32;
        const int64_t fine0 = (((b * 2 * h) + 2 * y) * W2 + 2 * x) * C + ck * 8;   // sub-pixel (0, 0)
        float v[4][8];   // [column group of 8][k]: coarse element g*8+k = channel (g*8+k)/4, sub-pixel (g*8+k)%4
        if (dir == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) load8(src + coarse + g * 8, v[g]);
            float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (bias) load8(bias + ck * 8, bb);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = v[(j * 4 + dd) >> 3][(j * 4 + dd) & 7] + bb[j];
                store8(dst + fine0 + ((int64_t)(dd >> 1) * W2 + (dd & 1)) * C, o);
            }
        } else {
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                float o[8];
                load8(src + fine0 + ((int64_t)(dd >> 1) * W2 + (dd & 1)) * C, o);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[(j * 4 + dd) >> 3][(j * 4 + dd) & 7] = o[j];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) store8(dst + coarse + g * 8, v[g]);
        }
    }
}

// dir 1 with the bias gradient of the transposed convolution in the same pass (round 4): part[block][c] = this workgroup's sum
// over its fine pixels of x[..][c] -- the stand-alone column sum read the 2h x 2w map a second time (12 us per layer at
// ViT-B bs 12, three layers).  The grid is a multiple of C / 8, so a thread keeps its channel group over its whole loop;
// the workgroup's 256 threads are reduced through LDS in thread order, the rows of `part` by the caller's batched column
// sum: a fixed summation order.
template <typename T>
__global__ __launch_bounds__(256) void pixel_unshuffle2_sums_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                                    float* __restrict__ part, int B, int h, int w, int C) {
    __shared__ float red[256][8];
    const int chunks = C / 8;
    const int W2 = 2 * w;
    const int64_t total = (int64_t)B * h * w * chunks;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ck = (int)(i0 % chunks);
    for (int64_t i = i0; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / chunks;
        const int x = (int)(m % w), y = (int)((m / w) % h);
        const int64_t b = m / ((int64_t)w * h);
        const int64_t coarse = m * (C * 4) + (int64_t)ck * 32;
        const int64_t fine0 = (((b * 2 * h) + 2 * y) * W2 + 2 * x) * C + ck * 8;
        float v[4][8];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            float o[8];
            load8(src + fine0 + ((int64_t)(dd >> 1) * W2 + (dd & 1)) * C, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[(j * 4 + dd) >> 3][(j * 4 + dd) & 7] = o[j]; s[j] += o[j]; }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) store8(dst + coarse + g * 8, v[g]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = s[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        // the threads of this workgroup whose channel group is c / 8: t = first, first + chunks, ...
        const int g = c >> 3, base = (int)(((int64_t)blockIdx.x * 256) % chunks);
        int first = g - base;
        if (first < 0) first += chunks;
        float acc = 0.f;
        for (int t = first; t < 256; t += chunks) acc += red[t][c & 7];
        part[(int64_t)blockIdx.x * C + c] = acc;
    }
}

// ------------------------------------------------------------------------------ GroupNorm(1, C)
constexpr int GN_CHUNKS = 64;

// thread layout shared by the GroupNorm kernels: a block walks the pixels [p0,p1) of one sample;
// thread t owns channel chunks (t % tpp) + r * tpp (r < nrep) of pixel slot (t / tpp); nrep = 1 up to C = 2048, 2 up to
// C = 4096 (ViT-H's 2560-channel stride-32 branch).
constexpr int GN_MAXREP = 2;
struct GnMap {
    int tpp, slots, cc, slot, nrep, cstep;
    bool active;
    int64_t p0, p1;
    __device__ GnMap(int C, int64_t HW) {
        nrep = (C / 8 + 255) / 256;
        tpp = C / 8 / nrep;
        cstep = tpp * 8;
        slots = 256 / tpp;
        if (slots < 1) slots = 1;
        slot = threadIdx.x / tpp;
        cc = (threadIdx.x % tpp) * 8;
        active = slot < slots && tpp <= 256;
        const int64_t per = (HW + GN_CHUNKS - 1) / GN_CHUNKS;
        p0 = (int64_t)blockIdx.y * per;
        p1 = p0 + per < HW ? p0 + per : HW;
    }
};

// dir 0 of pixel_shuffle2 (depth-to-space + bias) with the GroupNorm(1, C) statistics of its OUTPUT in the same pass (round 4):
// block (b, chunk) moves the coarse cells [chunk * cells / GN_CHUNKS, ...) of image b and leaves the sum and the sum of
// squares of the ROUNDED values it wrote in stats[b][chunk] -- what gn_stats_kernel would read back from the 2h x 2w map.
// (gn_finalize adds the GN_CHUNKS partials of an image whatever part of the image each one covers.)
template <typename T>
__global__ __launch_bounds__(256) void pixel_shuffle2_gn_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                                const float* __restrict__ bias, double* __restrict__ stats,
                                                                int h, int w, int C) {
    __shared__ double red[8];
    const int chunks = C / 8, W2 = 2 * w;
    const int b = blockIdx.x;
    const int64_t cells = (int64_t)h * w;
    const int64_t per = (cells + GN_CHUNKS - 1) / GN_CHUNKS;
    const int64_t c0 = (int64_t)blockIdx.y * per, c1 = c0 + per < cells ? c0 + per : cells;
    double ds = 0.0, dq = 0.0;
    for (int64_t i = c0 * chunks + threadIdx.x; i < c1 * chunks; i += 256) {
        const int ck = (int)(i % chunks);
        const int64_t m = i / chunks;                     // coarse cell (y, x) of image b
        const int x = (int)(m % w), y = (int)(m / w);
        const int64_t coarse = ((int64_t)b * cells + m) * (C * 4) + (int64_t)ck * 32;
        const int64_t fine0 = ((((int64_t)b * 2 * h) + 2 * y) * W2 + 2 * x) * C + ck * 8;
        float v[4][8];
#pragma unroll
        for (int g = 0; g < 4; ++g) load8(src + coarse + g * 8, v[g]);
        float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bias) load8(bias + ck * 8, bb);
        float sq = 0.f, qq = 0.f;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = to_f32(from_f32<T>(v[(j * 4 + dd) >> 3][(j * 4 + dd) & 7] + bb[j]));
                sq += o[j]; qq += o[j] * o[j];
            }
            store8(dst + fine0 + ((int64_t)(dd >> 1) * W2 + (dd & 1)) * C, o);
        }
        ds += sq; dq += qq;
    }
    const double S = block_sum_d(ds, red);
    const double Q = block_sum_d(dq, red);
    if (threadIdx.x == 0) {
        stats[((int64_t)b * GN_CHUNKS + blockIdx.y) * 2 + 0] = S;
        stats[((int64_t)b * GN_CHUNKS + blockIdx.y) * 2 + 1] = Q;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, double* __restrict__ stats, int64_t HW,
                                                       int C) {
    __shared__ double red[8];
    GnMap mp(C, HW);
    const int b = blockIdx.x;
    float s = 0.f, q = 0.f;
    double ds = 0.0, dq = 0.0;
    if (mp.active)
        for (int64_t p = mp.p0 + mp.slot; p < mp.p1; p += mp.slots)
            for (int r = 0; r < mp.nrep; ++r) {
                float v[8];
                load8(x + ((int64_t)b * HW + p) * C + mp.cc + r * mp.cstep, v);
                s = 0.f; q = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { s += v[j]; q += v[j] * v[j]; }
                ds += s; dq += q;
            }
    const double S = block_sum_d(ds, red);
    const double Q = block_sum_d(dq, red);
    if (threadIdx.x == 0) {
        stats[((int64_t)b * GN_CHUNKS + blockIdx.y) * 2 + 0] = S;
        stats[((int64_t)b * GN_CHUNKS + blockIdx.y) * 2 + 1] = Q;
    }
}

__device__ __forceinline__ void gn_finalize(const double* stats, int b, int64_t n, float eps, float& mu, float& rs,
                                            double* red) {
    double s = 0.0, q = 0.0;
    if (threadIdx.x < GN_CHUNKS) {
        s = stats[((int64_t)b * GN_CHUNKS + threadIdx.x) * 2 + 0];
        q = stats[((int64_t)b * GN_CHUNKS + threadIdx.x) * 2 + 1];
    }
    const double S = block_sum_d(s, red);
    const double Q = block_sum_d(q, red);
    const double m = S / (double)n;
    double var = Q / (double)n - m * m;
    if (var < 0.0) var = 0.0;
    mu = (float)m;
    rs = (float)(1.0 / sqrt(var + (double)eps));
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bb, T* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                       const double* __restrict__ stats, int64_t HW, int C, float eps,
                                                       int gelu) {
    __shared__ double red[8];
    GnMap mp(C, HW);
    const int b = blockIdx.x;
    float mu, rs;
    gn_finalize(stats, b, HW * C, eps, mu, rs, red);
    if (blockIdx.y == 0 && threadIdx.x == 0) { mean[b] = mu; rstd[b] = rs; }
    if (!mp.active) return;
    float ww[GN_MAXREP][8], bv[GN_MAXREP][8];
#pragma unroll
    for (int r = 0; r < GN_MAXREP; ++r)
        if (r < mp.nrep) { load8(w + mp.cc + r * mp.cstep, ww[r]); load8(bb + mp.cc + r * mp.cstep, bv[r]); }
    for (int64_t p = mp.p0 + mp.slot; p < mp.p1; p += mp.slots)
#pragma unroll
        for (int r = 0; r < GN_MAXREP; ++r)
            if (r < mp.nrep) {
                float v[8];
                const int64_t off = ((int64_t)b * HW + p) * C + mp.cc + r * mp.cstep;
                load8(x + off, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = (v[j] - mu) * rs * ww[r][j] + bv[r][j];
                    v[j] = gelu ? gelu_t<T>(t) : t;
                }
                store8(y + off, v);
            }
}

// backward pass 1: per-sample sums of g and g*xhat (fp64 partials), per-channel dw/db partials
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ w, const float* __restrict__ bb,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ part,
                                                           double* __restrict__ stats, int B, int64_t HW, int C,
                                                           int gelu) {
    __shared__ double red[8];
    __shared__ float acc[4096];
    GnMap mp(C, HW);
    const int b = blockIdx.x;
    const float mu = mean[b], rs = rstd[b];
    float dwa[GN_MAXREP][8], dba[GN_MAXREP][8], ww[GN_MAXREP][8], bv[GN_MAXREP][8];
#pragma unroll
    for (int r = 0; r < GN_MAXREP; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) { dwa[r][j] = 0.f; dba[r][j] = 0.f; ww[r][j] = 0.f; bv[r][j] = 0.f; }
    double d1 = 0.0, d2 = 0.0;
    if (mp.active) {
#pragma unroll
        for (int r = 0; r < GN_MAXREP; ++r)
            if (r < mp.nrep) { load8(w + mp.cc + r * mp.cstep, ww[r]); load8(bb + mp.cc + r * mp.cstep, bv[r]); }
        for (int64_t p = mp.p0 + mp.slot; p < mp.p1; p += mp.slots)
#pragma unroll
            for (int r = 0; r < GN_MAXREP; ++r)
                if (r < mp.nrep) {
                    float xv[8], dv[8];
                    const int64_t off = ((int64_t)b * HW + p) * C + mp.cc + r * mp.cstep;
                    load8(x + off, xv);
                    load8(dy + off, dv);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xh = (xv[j] - mu) * rs;
                        float d = dv[j];
                        if (gelu) d *= dgelu_t<T>(xh * ww[r][j] + bv[r][j]);
                        const float g = d * ww[r][j];
                        s1 += g; s2 += g * xh;
                        dwa[r][j] += d * xh; dba[r][j] += d;
                    }
                    d1 += s1; d2 += s2;
                }
    }
    const double S1 = block_sum_d(d1, red);
    const double S2 = block_sum_d(d2, red);
    if (threadIdx.x == 0) {
        stats[((int64_t)b * GN_CHUNKS + blockIdx.y) * 2 + 0] = S1;
        stats[((int64_t)b * GN_CHUNKS + blockIdx.y) * 2 + 1] = S2;
    }
    const int64_t prow = (int64_t)b * GN_CHUNKS + blockIdx.y;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
        if (mp.active) {
#pragma unroll
            for (int r = 0; r < GN_MAXREP; ++r)
                if (r < mp.nrep) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc[mp.slot * C + mp.cc + r * mp.cstep + j] = which ? dba[r][j] : dwa[r][j];
                }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float t = 0.f;
            for (int s = 0; s < mp.slots; ++s) t += acc[s * C + c];
            part[(prow * 2 + which) * C + c] = t;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ bb,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        T* __restrict__ dx, const double* __restrict__ stats, int64_t HW,
                                                        int C, int gelu) {
    __shared__ double red[8];
    GnMap mp(C, HW);
    const int b = blockIdx.x;
    double s = 0.0, q = 0.0;
    if (threadIdx.x < GN_CHUNKS) {
        s = stats[((int64_t)b * GN_CHUNKS + threadIdx.x) * 2 + 0];
        q = stats[((int64_t)b * GN_CHUNKS + threadIdx.x) * 2 + 1];
    }
    const double n = (double)HW * (double)C;
    const float m1 = (float)(block_sum_d(s, red) / n);
    const float m2 = (float)(block_sum_d(q, red) / n);
    if (!mp.active) return;
    const float mu = mean[b], rs = rstd[b];
    float ww[GN_MAXREP][8], bv[GN_MAXREP][8];
#pragma unroll
    for (int r = 0; r < GN_MAXREP; ++r)
        if (r < mp.nrep) { load8(w + mp.cc + r * mp.cstep, ww[r]); load8(bb + mp.cc + r * mp.cstep, bv[r]); }
    for (int64_t p = mp.p0 + mp.slot; p < mp.p1; p += mp.slots)
#pragma unroll
        for (int r = 0; r < GN_MAXREP; ++r)
            if (r < mp.nrep) {
                float xv[8], dv[8];
                const int64_t off = ((int64_t)b * HW + p) * C + mp.cc + r * mp.cstep;
                load8(x + off, xv);
                load8(dy + off, dv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (xv[j] - mu) * rs;
                    float d = dv[j];
                    if (gelu) d *= dgelu_t<T>(xh * ww[r][j] + bv[r][j]);
                    dv[j] = rs * (d * ww[r][j] - m1 - xh * m2);
                }
                store8(dx + off, dv);
            }
}

// ------------------------------------------------------------------------------ bilinear, align_corners=False
__device__ __forceinline__ void src_index_half(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_cl_fwd_kernel(const T* __restrict__ in, int ld_in, T* __restrict__ out,
                                                              int ld_out, int B, int h, int w, int H, int W, int C) {
    const int chunks = C / 8;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int64_t total = (int64_t)B * H * W * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t pix = i / chunks;
        const int X = (int)(pix % W), Y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        int y0, y1, x0, x1;
        float ly, lx;
        src_index_half(Y, sh, h, y0, y1, ly);
        src_index_half(X, sw, w, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        float a[8], bq[8], c[8], d[8], o[8];
        const int64_t base = (int64_t)b * h * w;
        load8(in + (base + (int64_t)y0 * w + x0) * ld_in + ck * 8, a);
        load8(in + (base + (int64_t)y0 * w + x1) * ld_in + ck * 8, bq);
        load8(in + (base + (int64_t)y1 * w + x0) * ld_in + ck * 8, c);
        load8(in + (base + (int64_t)y1 * w + x1) * ld_in + ck * 8, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = hy * (hx * a[j] + lx * bq[j]) + ly * (hx * c[j] + lx * d[j]);
        store8(out + pix * ld_out + ck * 8, o);
    }
}

// Head fusion by linearity (swin_transformer.py:744-756): a 1x1 convolution over the channel concat of bilinearly resized
// maps equals the sum of the resized per-map 1x1 convolutions (both are linear, the resize acts per channel), so the
// [B*H*W, 4C] concat buffer (308 MB per step at ViT-B bs 12) and three quarters of the fusion GEMM never exist:
//     io[p] = relu(io[p] + sum_i resize(z_i)[p]),   io = pre-activation of the full-resolution level (+ bias), in place.
struct UpsumArgs {
    const void* z[3];
    int h[3], w[3];
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void upsum_relu_kernel(T* __restrict__ io, const UpsumArgs a, int B, int H, int W, int C) {
    const int chunks = C / 8;
    const int64_t total = (int64_t)B * H * W * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t pix = i / chunks;
        const int X = (int)(pix % W), Y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        float o[8];
        load8(io + pix * C + ck * 8, o);
#pragma unroll
        for (int m = 0; m < 3; ++m)
            if (m < a.n) {
                const int h = a.h[m], w = a.w[m];
                const T* __restrict__ in = reinterpret_cast<const T*>(a.z[m]);
                int y0, y1, x0, x1;
                float ly, lx;
                src_index_half(Y, (float)h / (float)H, h, y0, y1, ly);
                src_index_half(X, (float)w / (float)W, w, x0, x1, lx);
                const float hy = 1.f - ly, hx = 1.f - lx;
                float p[8], q[8], r[8], s[8];
                const int64_t base = (int64_t)b * h * w;
                load8(in + (base + (int64_t)y0 * w + x0) * C + ck * 8, p);
                load8(in + (base + (int64_t)y0 * w + x1) * C + ck * 8, q);
                load8(in + (base + (int64_t)y1 * w + x0) * C + ck * 8, r);
                load8(in + (base + (int64_t)y1 * w + x1) * C + ck * 8, s);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += hy * (hx * p[j] + lx * q[j]) + ly * (hx * r[j] + lx * s[j]);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.f);
        store8(io + pix * C + ck * 8, o);
    }
}

// 2 x 2 output pixels per lane (integer ratios >= 2, even H / W): the four pixels share a 3 x 3 (ratio 2) or 2 x 2 (ratio
// >= 4) neighbourhood of each low-resolution map, 9 / 4 tap loads instead of 16 -- the one-pixel form moves 12 x the bytes
// of the output through L2 (82 us at ViT-B bs 12).  Same association as above (horizontal, then vertical): identical bits.
template <typename T>
__global__ __launch_bounds__(256) void upsum_relu2_kernel(T* __restrict__ io, const UpsumArgs a, int B, int H, int W, int C) {
    const int chunks = C / 8, H2 = H >> 1, W2 = W >> 1;
    const int64_t total = (int64_t)B * H2 * W2 * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t blk = i / chunks;
        const int X = 2 * (int)(blk % W2), Y = 2 * (int)((blk / W2) % H2), b = (int)(blk / ((int64_t)W2 * H2));
        float o[2][2][8];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) load8(io + (((int64_t)b * H + Y + dy) * W + X + dx) * C + ck * 8, o[dy][dx]);
#pragma unroll
        for (int m = 0; m < 3; ++m)
            if (m < a.n) {
                const int h = a.h[m], w = a.w[m];
                const T* __restrict__ in = reinterpret_cast<const T*>(a.z[m]) + (int64_t)b * h * w * C + ck * 8;
                int y0a, y1a, y0b, y1b, x0a, x1a, x0b, x1b;
                float lya, lyb, lxa, lxb;
                src_index_half(Y, (float)h / (float)H, h, y0a, y1a, lya);
                src_index_half(Y + 1, (float)h / (float)H, h, y0b, y1b, lyb);
                src_index_half(X, (float)w / (float)W, w, x0a, x1a, lxa);
                src_index_half(X + 1, (float)w / (float)W, w, x0b, x1b, lxb);
                const bool sy = y0b != y0a, sx = x0b != x0a;   // second pixel moved on to the taps (1, 2) of the triple
                const int ry[3] = {y0a, y1a, y1b}, rx[3] = {x0a, x1a, x1b};
                float ha[3][8], hb[3][8];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (r == 2 && !sy) continue;
                    float t0[8], t1[8], t2[8];
                    load8(in + ((int64_t)ry[r] * w + rx[0]) * C, t0);
                    load8(in + ((int64_t)ry[r] * w + rx[1]) * C, t1);
                    if (sx) load8(in + ((int64_t)ry[r] * w + rx[2]) * C, t2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        ha[r][j] = (1.f - lxa) * t0[j] + lxa * t1[j];
                        const float u0 = sx ? t1[j] : t0[j], u1 = sx ? t2[j] : t1[j];
                        hb[r][j] = (1.f - lxb) * u0 + lxb * u1;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o[0][0][j] += (1.f - lya) * ha[0][j] + lya * ha[1][j];
                    o[0][1][j] += (1.f - lya) * hb[0][j] + lya * hb[1][j];
                    const float a0 = sy ? ha[1][j] : ha[0][j], a1 = sy ? ha[2][j] : ha[1][j];
                    const float b0 = sy ? hb[1][j] : hb[0][j], b1 = sy ? hb[2][j] : hb[1][j];
                    o[1][0][j] += (1.f - lyb) * a0 + lyb * a1;
                    o[1][1][j] += (1.f - lyb) * b0 + lyb * b1;
                }
            }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[dy][dx][j] = fmaxf(o[dy][dx][j], 0.f);
                store8(io + (((int64_t)b * H + Y + dy) * W + X + dx) * C + ck * 8, o[dy][dx]);
            }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_cl_bwd_kernel(const T* __restrict__ dout, int ld_out,
                                                              T* __restrict__ din, int ld_in, int B, int h, int w, int H,
                                                              int W, int C) {
    const int chunks = C / 8;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const float fh = (float)H / (float)h, fw = (float)W / (float)w;
    const int64_t total = (int64_t)B * h * w * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t pix = i / chunks;
        const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((int64_t)w * h));
        int Ylo = (int)floorf(fh * ((float)y - 0.5f) - 0.5f) - 1, Yhi = (int)ceilf(fh * ((float)y + 1.5f)) + 1;
        int Xlo = (int)floorf(fw * ((float)x - 0.5f) - 0.5f) - 1, Xhi = (int)ceilf(fw * ((float)x + 1.5f)) + 1;
        if (y == 0) Ylo = 0;
        if (x == 0) Xlo = 0;
        if (Ylo < 0) Ylo = 0;
        if (Xlo < 0) Xlo = 0;
        if (Yhi > H - 1 || y == h - 1) Yhi = H - 1;
        if (Xhi > W - 1 || x == w - 1) Xhi = W - 1;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int Y = Ylo; Y <= Yhi; ++Y) {
            int y0, y1; float ly;
            src_index_half(Y, sh, h, y0, y1, ly);
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int X = Xlo; X <= Xhi; ++X) {
                int x0, x1; float lx;
                src_index_half(X, sw, w, x0, x1, lx);
                const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
                if (wx == 0.f) continue;
                float g[8];
                load8(dout + (((int64_t)b * H + Y) * W + X) * ld_out + ck * 8, g);
                const float ww = wy * wx;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += ww * g[j];
            }
        }
        store8(din + pix * ld_in + ck * 8, acc);
    }
}

// Integer up-sampling ratio R (H = R*h, W = R*w): the outputs that touch input pixel x are exactly the 2R columns
// [R*x - R/2, R*x + 3R/2) (clipped), so the gather walks (2R)^2 taps with the column weights computed once per thread
// (the generic kernel above scans a (2R+4)^2 candidate window and re-derives both indices per candidate).
template <typename T, int R>
__global__ __launch_bounds__(256) void bilinear_cl_bwd_int_kernel(const T* __restrict__ dout, int ld_out,
                                                                  T* __restrict__ din, int ld_in, int B, int h, int w,
                                                                  int C) {
    const int H = R * h, W = R * w;
    const int chunks = C / 8;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int64_t total = (int64_t)B * h * w * chunks;
    // round 5: every output value is gathered by four input pixels (2 x 2 overlapping windows); with the workgroups dealt to the
    // eight XCDs round-robin the four readers sat behind four different L2s and the 77-MB map came from HBM 2.2-3 times (PMC).
    // Each XCD now walks ONE contiguous eighth of the input pixels (xcd_ranged_block), so the re-reads hit its own L2.
    const int64_t nblk = (total + 255) / 256, per = (nblk + 7) / 8;
    for (int64_t vb = blockIdx.x; vb < 8 * per; vb += gridDim.x) {
        const int64_t bb = (vb & 7) * per + (vb >> 3);
        const int64_t i = bb * 256 + threadIdx.x;
        if ((vb >> 3) >= per || bb >= nblk || i >= total) continue;
        const int ck = (int)(i % chunks);
        const int64_t pix = i / chunks;
        const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((int64_t)w * h));
        const int Xs = R * x - R / 2, Ys = R * y - R / 2;
        float wx[2 * R];
#pragma unroll
        for (int u = 0; u < 2 * R; ++u) {
            const int X = Xs + u;
            int x0, x1; float lx;
            src_index_half(X < 0 ? 0 : X, sw, w, x0, x1, lx);
            const float wgt = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
            wx[u] = (X >= 0 && X < W) ? wgt : 0.f;
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int v = 0; v < 2 * R; ++v) {
            const int Y = Ys + v;
            if (Y < 0 || Y >= H) continue;
            int y0, y1; float ly;
            src_index_half(Y, sh, h, y0, y1, ly);
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            const T* rowp = dout + (((int64_t)b * H + Y) * W) * ld_out + ck * 8;
#pragma unroll
            for (int u = 0; u < 2 * R; ++u) {
                const int X = Xs + u;
                if (X >= 0 && X < W) {
                    float g[8];
                    load8(rowp + (int64_t)X * ld_out, g);
                    const float ww = wy * wx[u];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += ww * g[j];
                }
            }
        }
        store8(din + pix * ld_in + ck * 8, acc);
    }
}

// The same gather with the (2R)^2 window of one input pixel cut into FOUR row slices (C = 256: 32 chunk lanes x 2 slices
// per wave, 2 waves per pixel, two pixels per block): at R = 8 one thread per (pixel, chunk) walked 256 taps back to back
// with only 75k threads in flight (93 us for the 14 x 14 level of the head); the slices are summed in a fixed order
// (lane pair by shuffle, wave pair through LDS).
template <typename T, int R>
__global__ __launch_bounds__(256) void bilinear_cl_bwd_int4_kernel(const T* __restrict__ dout, int ld_out,
                                                                   T* __restrict__ din, int ld_in, int B, int h, int w) {
    __shared__ float red[2][32][8];
    const int H = R * h, W = R * w;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ck = lane & 31, part = (lane >> 5) | ((wave & 1) << 1);
    // (XCD-ranged block order, as in bilinear_cl_bwd_int_kernel: the grid is 8 * per blocks)
    const int64_t nblk = ((int64_t)B * h * w + 1) / 2, per = (nblk + 7) / 8;
    const int64_t bb = ((int64_t)blockIdx.x & 7) * per + ((int64_t)blockIdx.x >> 3);
    const int64_t pix = bb * 2 + (wave >> 1);
    const bool live = bb < nblk && pix < (int64_t)B * h * w;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (live) {
        const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((int64_t)w * h));
        const int Xs = R * x - R / 2, Ys = R * y - R / 2;
        float wx[2 * R];
#pragma unroll
        for (int u = 0; u < 2 * R; ++u) {
            const int X = Xs + u;
            int x0, x1; float lx;
            src_index_half(X < 0 ? 0 : X, sw, w, x0, x1, lx);
            const float wgt = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
            wx[u] = (X >= 0 && X < W) ? wgt : 0.f;
        }
#pragma unroll
        for (int vv = 0; vv < R / 2; ++vv) {
            const int Y = Ys + part * (R / 2) + vv;
            if (Y < 0 || Y >= H) continue;
            int y0, y1; float ly;
            src_index_half(Y, sh, h, y0, y1, ly);
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            const T* rowp = dout + (((int64_t)b * H + Y) * W) * ld_out + ck * 8;
#pragma unroll
            for (int u = 0; u < 2 * R; ++u) {
                const int X = Xs + u;
                if (X >= 0 && X < W) {
                    float g[8];
                    load8(rowp + (int64_t)X * ld_out, g);
                    const float ww = wy * wx[u];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += ww * g[j];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], 32, 64);
    if ((wave & 1) && lane < 32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave >> 1][ck][j] = acc[j];
    }
    __syncthreads();
    if (!(wave & 1) && lane < 32 && live) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += red[wave >> 1][ck][j];
        store8(din + pix * ld_in + ck * 8, acc);
    }
}

// ------------------------------------------------------------------------------ DMA gates
template <typename T>
__global__ __launch_bounds__(256) void gate_colmax_kernel(const T* __restrict__ Q, float* __restrict__ cg,
                                                          int* __restrict__ argq, int B, int nq, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C) return;
    const int c = (int)(i % C), b = (int)(i / C);
    float mx = -INFINITY;
    int am = 0;
#pragma unroll 8
    for (int q = 0; q < nq; ++q) {   // (unrolled: eight independent loads in flight, the walk is latency-bound)
        const float v = to_f32(Q[((int64_t)b * nq + q) * C + c]);
        if (v > mx) { mx = v; am = q; }
    }
    cg[i] = sigmoid_f(mx);
    argq[i] = am;
}
template <typename T>
__global__ __launch_bounds__(256) void gate_rowmax_kernel(const T* __restrict__ K, float* __restrict__ sg,
                                                          int* __restrict__ argc, int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float mx = -INFINITY;
    int am = 0;
    if ((C & 7) == 0) {   // 16-byte loads; first maximum wins inside a lane (ascending columns), lowest index across lanes
        for (int ck = lane; ck < (C >> 3); ck += 64) {
            float v[8];
            load8(K + row * C + ck * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (v[j] > mx) { mx = v[j]; am = ck * 8 + j; }
        }
    } else
    for (int c = lane; c < C; c += 64) {
        const float v = to_f32(K[row * C + c]);
        if (v > mx) { mx = v; am = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    if (lane == 0) { sg[row] = sigmoid_f(mx); argc[row] = am; }
}
template <typename T>
__global__ __launch_bounds__(256) void gate_apply_kernel(const T* __restrict__ x, const float* __restrict__ cg,
                                                         const float* __restrict__ sg, T* __restrict__ out, int B, int N,
                                                         int C) {
    const int chunks = C / 8;
    const int64_t total = (int64_t)B * N * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t row = i / chunks;
        const int b = (int)(row / N);
        float v[8], g[8];
        load8(x + row * C + ck * 8, v);
        load8(cg + (int64_t)b * C + ck * 8, g);
        const float s = sg[row];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * (1.f + g[j] + s);
        store8(out + row * C + ck * 8, v);
    }
}

constexpr int GATE_NBLK = 64;   // row slices per sample (16: 192 workgroups for the whole launch, 31 us at ViT-B bs 12)
// one wave per row n: dx, dsg (+ scatter into dK), per-column partial of dcg
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_rows_kernel(const T* __restrict__ dout, const T* __restrict__ x,
                                                            const float* __restrict__ cg, const float* __restrict__ sg,
                                                            const int* __restrict__ argc, T* __restrict__ dx, int accum,
                                                            T* __restrict__ dK, float* __restrict__ part, int N, int C) {
    __shared__ float red[4][8 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int per = (N + GATE_NBLK - 1) / GATE_NBLK;
    const int n0 = blockIdx.y * per, n1 = n0 + per < N ? n0 + per : N;
    constexpr int MAXCH = 4;
    float dcg[MAXCH][8], g[MAXCH][8];
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = (lane + i * 64) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { dcg[i][j] = 0.f; g[i][j] = 0.f; }
        if (c < C) load8(cg + (int64_t)b * C + c, g[i]);
    }
    for (int n = n0 + wave; n < n1; n += 4) {
        const int64_t row = (int64_t)b * N + n;
        const float s = sg[row];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = (lane + i * 64) * 8;
            if (c < C) {
                float dv[8], xv[8], o[8];
                load8(dout + row * C + c, dv);
                load8(x + row * C + c, xv);
                if (accum) load8(dx + row * C + c, o);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = dv[j] * xv[j];
                    dot += t;
                    dcg[i][j] += t;
                    o[j] += dv[j] * (1.f + g[i][j] + s);
                }
                store8(dx + row * C + c, o);
            }
        }
        dot = wave_sum(dot);
        if (lane == 0) {
            const int64_t k = row * C + argc[row];
            dK[k] = from_f32<T>(to_f32(dK[k]) + dot * s * (1.f - s));
        }
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = (lane + i * 64) * 8;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][j * 64 + lane] = dcg[i][j];
        __syncthreads();
        if (wave == 0 && c < C) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = red[0][j * 64 + lane] + red[1][j * 64 + lane] + red[2][j * 64 + lane] + red[3][j * 64 + lane];
            store8(part + ((int64_t)b * GATE_NBLK + blockIdx.y) * C + c, o);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_cols_kernel(const float* __restrict__ part, const float* __restrict__ cg,
                                                            const int* __restrict__ argq, T* __restrict__ dQ, int B,
                                                            int nq, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C) return;
    const int c = (int)(i % C), b = (int)(i / C);
    float t = 0.f;
    for (int k = 0; k < GATE_NBLK; ++k) t += part[((int64_t)b * GATE_NBLK + k) * C + c];
    const float g = cg[i];
    const int64_t k = ((int64_t)b * nq + argq[i]) * C + c;
    dQ[k] = from_f32<T>(to_f32(dQ[k]) + t * g * (1.f - g));
}

// ---- the three gates of SimpleFPN in one launch each (round 5).  is_vpu_model.py:106-121 builds x_i = x + x cg_i + x sg_i for the
// three (queries, keys) pairs the DMA neck returns: same x, three gate pairs.  Per gate the round-1 kernels above took two
// statistics launches and one pass over x forward, two launches and a read-modify-write of dx backward: nine / six launches,
// x read three times, dx rewritten three times.  Here: ONE statistics launch (column blocks, then row blocks, gate = grid y),
// ONE apply launch (x read once, three maps written), one row pass backward (dx = sum_i dout_i (1 + cg_i + sg_i) in fp32,
// rounded once) + one column launch.  Values per gate are the single-gate kernels' (same operations in the same order);
// dx differs from three chained launches by the roundings it no longer makes.
constexpr int GATE_MAXN = 3;
struct GateSet {
    const void* Q[GATE_MAXN];      // queries [B][nq][C]
    const void* K[GATE_MAXN];      // keys [B][N][C]
    void* out[GATE_MAXN];          // forward: gated maps; backward: the maps' gradients (read)
    void* dQ[GATE_MAXN];
    void* dK[GATE_MAXN];
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void gate_stats_n_kernel(const GateSet gs, float* __restrict__ cg, int* __restrict__ argq,
                                                           float* __restrict__ sg, int* __restrict__ argc, int B, int nq, int N,
                                                           int C, int col_blocks) {
    const int gi = blockIdx.y;
    if ((int)blockIdx.x < col_blocks) {         // column maxima over the queries: thread -> (sample, channel)
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= (int64_t)B * C) return;
        const T* Q = (const T*)gs.Q[gi];
        const int c = (int)(i % C), b = (int)(i / C);
        float mx = -INFINITY;
        int am = 0;
#pragma unroll 8
        for (int q = 0; q < nq; ++q) {
            const float v = to_f32(Q[((int64_t)b * nq + q) * C + c]);
            if (v > mx) { mx = v; am = q; }
        }
        cg[(int64_t)gi * B * C + i] = sigmoid_f(mx);
        argq[(int64_t)gi * B * C + i] = am;
        return;
    }
    // row maxima over the channels: one wave per row
    const T* K = (const T*)gs.K[gi];
    const int lane = threadIdx.x & 63;
    const int64_t rows = (int64_t)B * N, row = (int64_t)(blockIdx.x - col_blocks) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float mx = -INFINITY;
    int am = 0;
    for (int ck = lane; ck < (C >> 3); ck += 64) {      // (C % 8 == 0: checked by the launcher)
        float v[8];
        load8(K + row * C + ck * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (v[j] > mx) { mx = v[j]; am = ck * 8 + j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    if (lane == 0) { sg[(int64_t)gi * rows + row] = sigmoid_f(mx); argc[(int64_t)gi * rows + row] = am; }
}
template <typename T>
__global__ __launch_bounds__(256) void gate_apply_n_kernel(const T* __restrict__ x, const float* __restrict__ cg,
                                                           const float* __restrict__ sg, const GateSet gs, int B, int N, int C) {
    const int chunks = C / 8;
    const int64_t rows = (int64_t)B * N, total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t row = i / chunks;
        const int b = (int)(row / N);
        float v[8];
        load8(x + row * C + ck * 8, v);
#pragma unroll
        for (int gi = 0; gi < GATE_MAXN; ++gi)
            if (gi < gs.n) {
                float g[8], o[8];
                load8(cg + ((int64_t)gi * B + b) * C + ck * 8, g);
                const float s = sg[(int64_t)gi * rows + row];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = v[j] * (1.f + g[j] + s);
                store8((T*)gs.out[gi] + row * C + ck * 8, o);
            }
    }
}
// eight consecutive elements as they sit in memory (bf16: four registers), expanded to fp32 where they are used
template <typename T> struct Raw8 {
    float v[8];
    __device__ __forceinline__ void load(const T* p) { load8(p, v); }
    __device__ __forceinline__ void get(float (&o)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = v[j];
    }
};
template <> struct Raw8<bf16_t> {
    uint4 r;
    __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void get(float (&o)[8]) const {
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(w[j] << 16); o[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
    }
};
// one wave per row n: dx, the dsg of every gate (+ scatter into its dK), per-column partials of every dcg
// (round 5: the column gates of the sample sit in LDS instead of 48 registers per lane and every load of a row -- x, the
// gates' gradients, dx -- is requested before the first multiply and stays packed until it is used: 170 -> 154 registers =
// three workgroups per CU, so the 768 workgroups of ViT-B at bs 12 are resident at once instead of a round and a half, and a
// row costs one memory latency instead of two)
template <typename T, int NCHK>
__global__ __launch_bounds__(256) void gate_bwd_rows_n_kernel(const GateSet gs, const T* __restrict__ x, const float* __restrict__ cg,
                                                              const float* __restrict__ sg, const int* __restrict__ argc,
                                                              T* __restrict__ dx, int accum, float* __restrict__ part, int B, int N,
                                                              int C) {
    __shared__ float red[4][8 * 64];
    __shared__ float gsh[GATE_MAXN][NCHK * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int per = (N + GATE_NBLK - 1) / GATE_NBLK;
    const int n0 = blockIdx.y * per, n1 = n0 + per < N ? n0 + per : N;
    const int64_t rows = (int64_t)B * N;
    for (int i = threadIdx.x; i < GATE_MAXN * NCHK * 512; i += 256) {
        const int gi = i / (NCHK * 512), c = i - gi * (NCHK * 512);
        gsh[gi][c] = (gi < gs.n && c < C) ? cg[((int64_t)gi * B + b) * C + c] : 0.f;
    }
    float dcg[GATE_MAXN][NCHK][8];
#pragma unroll
    for (int gi = 0; gi < GATE_MAXN; ++gi)
#pragma unroll
        for (int i = 0; i < NCHK; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) dcg[gi][i][j] = 0.f;
    __syncthreads();
    for (int n = n0 + wave; n < n1; n += 4) {
        const int64_t row = (int64_t)b * N + n;
        float s[GATE_MAXN], dot[GATE_MAXN];
#pragma unroll
        for (int gi = 0; gi < GATE_MAXN; ++gi) { s[gi] = gi < gs.n ? sg[(int64_t)gi * rows + row] : 0.f; dot[gi] = 0.f; }
        Raw8<T> xr[NCHK], orw[NCHK], dr[GATE_MAXN][NCHK];
#pragma unroll
        for (int i = 0; i < NCHK; ++i) {      // every load of the row first
            const int c = (lane + i * 64) * 8;
            if (c < C) {
                xr[i].load(x + row * C + c);
#pragma unroll
                for (int gi = 0; gi < GATE_MAXN; ++gi)
                    if (gi < gs.n) dr[gi][i].load((const T*)gs.out[gi] + row * C + c);
                if (accum) orw[i].load(dx + row * C + c);
            }
        }
#pragma unroll
        for (int i = 0; i < NCHK; ++i) {
            const int c = (lane + i * 64) * 8;
            if (c < C) {
                float xv[8], o[8];
                xr[i].get(xv);
                if (accum) orw[i].get(o);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = 0.f;
                }
#pragma unroll
                for (int gi = 0; gi < GATE_MAXN; ++gi)
                    if (gi < gs.n) {
                        const float4 ga = *reinterpret_cast<const float4*>(&gsh[gi][c]), gb = *reinterpret_cast<const float4*>(&gsh[gi][c + 4]);
                        const float g[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                        float dv[8];
                        dr[gi][i].get(dv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float t = dv[j] * xv[j];
                            dot[gi] += t;
                            dcg[gi][i][j] += t;
                            o[j] += dv[j] * (1.f + g[j] + s[gi]);
                        }
                    }
                store8(dx + row * C + c, o);
            }
        }
#pragma unroll
        for (int gi = 0; gi < GATE_MAXN; ++gi)
            if (gi < gs.n) {
                const float d = wave_sum(dot[gi]);
                if (lane == 0) {
                    T* dK = (T*)gs.dK[gi];
                    const int64_t k = row * C + argc[(int64_t)gi * rows + row];
                    dK[k] = from_f32<T>(to_f32(dK[k]) + d * s[gi] * (1.f - s[gi]));
                }
            }
    }
#pragma unroll
    for (int gi = 0; gi < GATE_MAXN; ++gi)
        if (gi < gs.n) {
#pragma unroll
            for (int i = 0; i < NCHK; ++i) {
                const int c = (lane + i * 64) * 8;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 8; ++j) red[wave][j * 64 + lane] = dcg[gi][i][j];
                __syncthreads();
                if (wave == 0 && c < C) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        o[j] = red[0][j * 64 + lane] + red[1][j * 64 + lane] + red[2][j * 64 + lane] + red[3][j * 64 + lane];
                    store8(part + (((int64_t)gi * B + b) * GATE_NBLK + blockIdx.y) * C + c, o);
                }
            }
        }
}
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_cols_n_kernel(const float* __restrict__ part, const float* __restrict__ cg,
                                                              const int* __restrict__ argq, const GateSet gs, int B, int nq, int C) {
    const int gi = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C) return;
    const int c = (int)(i % C), b = (int)(i / C);
    float t = 0.f;
    for (int k = 0; k < GATE_NBLK; ++k) t += part[(((int64_t)gi * B + b) * GATE_NBLK + k) * C + c];
    const float g = cg[(int64_t)gi * B * C + i];
    T* dQ = (T*)gs.dQ[gi];
    const int64_t k = ((int64_t)b * nq + argq[(int64_t)gi * B * C + i]) * C + c;
    dQ[k] = from_f32<T>(to_f32(dQ[k]) + t * g * (1.f - g));
}

// ------------------------------------------------------------------------------ conv_seg (C -> 1, Dropout2d mask)
template <typename T>
__global__ __launch_bounds__(256) void convseg_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ mask, float* __restrict__ out,
                                                          int64_t rows, int64_t HW, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t b = row / HW;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        float t = to_f32(x[row * C + c]) * w[c];
        if (mask) t *= mask[b * C + c];
        s += t;
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s + bias[0];
}
// C in {64, 128, 256, 512}: C / 8 lanes per pixel with 16-byte loads (a wave covers 64 / (C / 8) pixels per trip), weights
// times dropout mask held in registers per sample -- the one-element-per-lane form above read 128 B per wave-instruction
// (57 us for the 77-MB head map at ViT-B bs 12).
template <typename T>
__global__ __launch_bounds__(256) void convseg_fwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ mask, float* __restrict__ out,
                                                              int64_t rows, int64_t HW, int C) {
    const int lpr = C >> 3;                       // lanes per pixel
    const int ppw = 64 / lpr;                     // pixels per wave and trip
    const int lane = threadIdx.x & 63, sub = lane / lpr, cl = lane - sub * lpr;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    float ww[8], wm[8];
    load8(w + cl * 8, ww);
    int64_t cur_b = -1;
    const float b0 = bias[0];
    for (int64_t r0 = wave_id * ppw; r0 < rows; r0 += nwaves * ppw) {
        const int64_t row = r0 + sub;
        float s = 0.f;
        if (row < rows) {
            const int64_t b = row / HW;
            if (b != cur_b) {
                cur_b = b;
                if (mask) {
                    float mk[8];
                    load8(mask + b * C + cl * 8, mk);
#pragma unroll
                    for (int j = 0; j < 8; ++j) wm[j] = ww[j] * mk[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) wm[j] = ww[j];
                }
            }
            float v[8];
            load8(x + row * C + cl * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j] * wm[j];
        }
        for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (cl == 0 && row < rows) out[row] = s + b0;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void convseg_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ mask,
                                                          T* __restrict__ dx, int accum, float* __restrict__ part,
                                                          float* __restrict__ part_b, int64_t rows, int64_t HW, int C,
                                                          int nblk) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MAXC = 8;  // C <= 512
    float dwa[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) dwa[i] = 0.f;
    float dba = 0.f;
    const int64_t per = (rows + nblk - 1) / nblk;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    for (int64_t row = r0 + wave; row < r1; row += 4) {
        const float d = dout[row];
        const int64_t b = row / HW;
        dba += d;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < C) {
                const float mk = mask ? mask[b * C + c] : 1.f;
                const float xv = to_f32(x[row * C + c]);
                dwa[i] += d * xv * mk;
                float o = d * w[c] * mk;
                if (accum & 1) o += to_f32(dx[row * C + c]);
                if ((accum & 2) && !(to_f32(x[row * C + c]) > 0.f)) o = 0.f;   // ReLU' of the map x itself (see vec kernel)
                dx[row * C + c] = from_f32<T>(o);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        __syncthreads();
        red[wave][lane] = dwa[i];
        __syncthreads();
        if (wave == 0 && c < C) part[(int64_t)blockIdx.x * C + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    }
    __syncthreads();
    red[wave][lane] = dba;  // same value in every lane of a wave
    __syncthreads();
    if (threadIdx.x == 0) part_b[blockIdx.x] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
}

// 16-byte form of convseg_bwd for bf16 maps with C in {64, 128, 256, 512}: a lane owns 8 consecutive channels, a wave
// covers 512 / C rows per pass (the scalar form above moves 2 B per lane: 265 us for the 77 MB map, this one is HBM-bound).
__global__ __launch_bounds__(256) void convseg_bwd_vec_kernel(const float* __restrict__ dout, const bf16_t* __restrict__ x,
                                                              const float* __restrict__ w, const float* __restrict__ mask,
                                                              bf16_t* __restrict__ dx, int accum, float* __restrict__ part,
                                                              float* __restrict__ part_b, int64_t rows, int64_t HW, int C,
                                                              int nblk) {
    __shared__ float red[4][512];
    __shared__ float redb[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpr = C >> 3;            // lanes per row
    const int rpp = 64 / lpr;          // rows per wave pass
    const int sub = lane / lpr, c8 = (lane % lpr) * 8;
    float wv[8];
    load8(w + c8, wv);
    float dwa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dba = 0.f;
    const int64_t per = (rows + nblk - 1) / nblk;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    for (int64_t rb = r0 + (int64_t)wave * rpp; rb < r1; rb += 4 * rpp) {
        const int64_t row = rb + sub;
        if (row < r1) {
            const float d = dout[row];
            const int64_t b = row / HW;
            if (c8 == 0) dba += d;
            float mk[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            if (mask) load8(mask + b * C + c8, mk);
            float xv[8], o[8];
            load8(x + row * C + c8, xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { dwa[j] += d * xv[j] * mk[j]; o[j] = d * wv[j] * mk[j]; }
            if (accum & 1) {
                float old[8];
                load8(dx + row * C + c8, old);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += old[j];
            }
            if (accum & 2) {   // x is the output of a ReLU and dx its complete gradient: hand back the pre-activation gradient
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = xv[j] > 0.f ? o[j] : 0.f;
            }
            store8(dx + row * C + c8, o);
        }
    }
    // lanes with the same c8 (different sub-rows) -> one value per channel per wave, then over the 4 waves
#pragma unroll
    for (int j = 0; j < 8; ++j)
        for (int o = lpr; o < 64; o <<= 1) dwa[j] += __shfl_xor(dwa[j], o, 64);
    dba = wave_sum(dba);
    if (sub == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][c8 + j] = dwa[j];
    }
    if (lane == 0) redb[wave] = dba;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        part[(int64_t)blockIdx.x * C + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    if (threadIdx.x == 0) part_b[blockIdx.x] = redb[0] + redb[1] + redb[2] + redb[3];
}

// Head backward, both gradient paths of the fused map in ONE pass: the P2CL path (gradient of fn = fused / |fused| through
// the L2 normalisation, losses.py:155-176 via swin_transformer.py:759-767) and the mask path (conv_seg + Dropout2d,
// decode_head.py:210-215), with the ReLU' of the fused map applied to their sum:
//     dx[r][c] = [x > 0] * ( inv[r] * (dfn[r][c] - y[r][c] * sum_c' dfn[r][c'] y[r][c']) + dout[r] * w[c] * mask[b][c] )
// plus conv_seg's weight / bias gradient partials.  The separate kernels (l2norm_bwd, then convseg_bwd accumulating into
// its output) moved the 77-MB gradient map through HBM twice more.  bf16, C in {64, 128, 256, 512}.
__global__ __launch_bounds__(256) void head_grad_fused_kernel(const bf16_t* __restrict__ dfn, const bf16_t* __restrict__ y,
                                                              const float* __restrict__ inv, const float* __restrict__ dout,
                                                              const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ mask, bf16_t* __restrict__ dx,
                                                              float* __restrict__ part, float* __restrict__ part_b,
                                                              int64_t rows, int64_t HW, int C, int nblk) {
    __shared__ float red[4][512];
    __shared__ float redb[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpr = C >> 3, rpp = 64 / lpr;
    const int sub = lane / lpr, c8 = (lane % lpr) * 8;
    float wv[8];
    load8(w + c8, wv);
    float dwa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dba = 0.f;
    const int64_t per = (rows + nblk - 1) / nblk;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    float mk[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    int64_t mk_b = -1;
    // (the image of a row -- for its dropout-mask row -- by increments: a 64-bit division per row was half of this loop's instructions)
    int64_t bimg = (r0 + (int64_t)wave * rpp + sub) / HW, brem = (r0 + (int64_t)wave * rpp + sub) - bimg * HW;
    for (int64_t rb = r0 + (int64_t)wave * rpp; rb < r1; rb += 4 * rpp, brem += 4 * rpp) {
        const int64_t row = rb + sub;
        const bool live = row < r1;
        while (brem >= HW) { brem -= HW; ++bimg; }
        float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, yv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, xv[8];
        float d = 0.f, iv = 0.f;
        if (live) {
            load8(dfn + row * C + c8, g);
            load8(y + row * C + c8, yv);
            load8(x + row * C + c8, xv);
            d = dout[row];
            iv = inv[row];
        }
        float sdot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sdot += g[j] * yv[j];
        for (int o = lpr >> 1; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o, 64);   // (every lane takes part: rows beyond r1 add 0)
        if (live) {
            const int64_t b = bimg;
            if (c8 == 0) dba += d;
            // (the image's dropout-mask row: fetched again only when the image changes -- fetched per row it was 32 bytes per lane
            // beside 48 of operands, 92 against 71 us per launch)
            if (mask && b != mk_b) { load8(mask + b * C + c8, mk); mk_b = b; }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dwa[j] += d * xv[j] * mk[j];
                const float t = iv * (g[j] - yv[j] * sdot) + d * wv[j] * mk[j];
                o[j] = xv[j] > 0.f ? t : 0.f;
            }
            store8(dx + row * C + c8, o);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        for (int o = lpr; o < 64; o <<= 1) dwa[j] += __shfl_xor(dwa[j], o, 64);
    dba = wave_sum(dba);
    if (sub == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][c8 + j] = dwa[j];
    }
    if (lane == 0) redb[wave] = dba;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        part[(int64_t)blockIdx.x * C + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    if (threadIdx.x == 0) part_b[blockIdx.x] = redb[0] + redb[1] + redb[2] + redb[3];
}

// ------------------------------------------------------------------------------ align_corners=True upsample (planes)
__device__ __forceinline__ void src_index_ac(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    const float s = scale * (float)dst;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}
__global__ __launch_bounds__(256) void upsample_ac_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int64_t planes, int h, int w, int H, int W) {
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const int64_t total = planes * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int X = (int)(i % W), Y = (int)((i / W) % H);
        const int64_t pl = i / ((int64_t)W * H);
        int y0, y1, x0, x1; float ly, lx;
        src_index_ac(Y, sh, h, y0, y1, ly);
        src_index_ac(X, sw, w, x0, x1, lx);
        const float* p = in + pl * h * w;
        const float hy = 1.f - ly, hx = 1.f - lx;
        out[i] = hy * (hx * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * (hx * p[y1 * w + x0] + lx * p[y1 * w + x1]);
    }
}
__global__ __launch_bounds__(256) void upsample_ac_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                              int64_t planes, int h, int w, int H, int W) {
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float fh = sh > 0.f ? 1.f / sh : 0.f, fw = sw > 0.f ? 1.f / sw : 0.f;
    const int64_t total = planes * h * w;
    const bool small = total < ((int64_t)1 << 31);       // (three 64-bit divisions per cell were a quarter of its instructions)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int x, y;
        int64_t pl;
        if (small) {
            const unsigned iu = (unsigned)i, q = iu / (unsigned)w;
            x = (int)(iu - q * (unsigned)w); y = (int)(q % (unsigned)h); pl = q / (unsigned)h;
        } else {
            x = (int)(i % w); y = (int)((i / w) % h); pl = i / ((int64_t)w * h);
        }
        int Ylo = (int)floorf(fh * (float)(y - 1)) - 1, Yhi = (int)ceilf(fh * (float)(y + 1)) + 1;
        int Xlo = (int)floorf(fw * (float)(x - 1)) - 1, Xhi = (int)ceilf(fw * (float)(x + 1)) + 1;
        if (Ylo < 0) Ylo = 0;
        if (Xlo < 0) Xlo = 0;
        if (Yhi > H - 1) Yhi = H - 1;
        if (Xhi > W - 1) Xhi = W - 1;
        const float* g = dout + pl * H * W;
        float acc = 0.f;
        if (Xhi - Xlo < 16) {
            // column weights once per thread instead of once per candidate (same products, same order: identical bits);
            // the window of a 4x map is ~12 x 12 candidates, 34 us -> 12 us for the mask-logit gradient at bs 12
            float wxs[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int X = Xlo + u;
                int x0, x1; float lx;
                src_index_ac(X <= Xhi ? X : Xhi, sw, w, x0, x1, lx);
                const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
                wxs[u] = X <= Xhi ? wx : 0.f;
            }
            for (int Y = Ylo; Y <= Yhi; ++Y) {
                int y0, y1; float ly;
                src_index_ac(Y, sh, h, y0, y1, ly);
                const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
                if (wy == 0.f) continue;
                const float* row = g + (int64_t)Y * W + Xlo;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (wxs[u] != 0.f) acc += wy * wxs[u] * row[u];
            }
            din[i] = acc;
            continue;
        }
        for (int Y = Ylo; Y <= Yhi; ++Y) {
            int y0, y1; float ly;
            src_index_ac(Y, sh, h, y0, y1, ly);
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int X = Xlo; X <= Xhi; ++X) {
                int x0, x1; float lx;
                src_index_ac(X, sw, w, x0, x1, lx);
                const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
                if (wx != 0.f) acc += wy * wx * g[(int64_t)Y * W + X];
            }
        }
        din[i] = acc;
    }
}

}  // namespace

extern "C" int vpu_patch_im2col(const float* image4, const float* disks, void* cols, int32_t B, int32_t H, int32_t W,
                                int32_t P, int32_t win_tokens, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (H % P || W % P || (W / P) % win_tokens || (H / P) % win_tokens) {
        vpu_set_error("patch_im2col: H,W % P, grid % window");
        return VPU_ERR_ARG;
    }
    const int64_t total = (int64_t)B * (H / P) * (W / P) * (2 * ((3 * P * P + 7) / 8));
    DISPATCH_T(dtype, patch_im2col_kernel<T><<<vpu_grid_for(total, 256, 16384), 256, 0, ST>>>(
        image4, disks, (T*)cols, B, H, W, P, win_tokens);)
    return vpu_check_launch("vpu_patch_im2col");
}
extern "C" int vpu_patch_im2col_prenorm(const float* image4, const float* disks, void* cols, int32_t B, int32_t H, int32_t W,
                                        int32_t P, int32_t win_tokens, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (H % P || W % P || (W / P) % win_tokens || (H / P) % win_tokens) {
        vpu_set_error("patch_im2col_prenorm: H,W % P, grid % window");
        return VPU_ERR_ARG;
    }
    const int64_t total = (int64_t)B * (H / P) * (W / P) * (2 * ((3 * P * P + 7) / 8));
    DISPATCH_T(dtype, (patch_im2col_kernel<T, false><<<vpu_grid_for(total, 256, 16384), 256, 0, ST>>>(
        image4, disks, (T*)cols, B, H, W, P, win_tokens));)
    return vpu_check_launch("vpu_patch_im2col_prenorm");
}
extern "C" int vpu_window_permute(const void* x, void* y, int32_t B, int32_t g, int32_t wg, int32_t C, int32_t dir,
                                  int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || g % wg) { vpu_set_error("window_permute: C % 8, g % wg"); return VPU_ERR_ARG; }
    const int64_t total = (int64_t)B * g * g * (C / 8);
    DISPATCH_T(dtype, window_permute_kernel<T><<<vpu_grid_for(total, 256, 16384), 256, 0, ST>>>((const T*)x, (T*)y, B, g,
                                                                                               wg, C, dir);)
    return vpu_check_launch("vpu_window_permute");
}
extern "C" int vpu_pixel_shuffle2(const void* in, void* out, const float* bias, int32_t B, int32_t h, int32_t w,
                                  int32_t C, int32_t dir, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8) { vpu_set_error("pixel_shuffle2: C % 8"); return VPU_ERR_ARG; }
    const int64_t total = (int64_t)B * h * w * (C / 8);
    DISPATCH_T(dtype, pixel_shuffle2_kernel<T><<<vpu_grid_for(total, 256, 16384), 256, 0, ST>>>((const T*)in, (T*)out,
                                                                                               bias, B, h, w, C, dir);)
    return vpu_check_launch("vpu_pixel_shuffle2");
}
extern "C" int vpu_pixel_unshuffle2_nblk(int32_t C) {
    vpu_clear_stale_error();
    const int chunks = C / 8;
    return chunks < 1 || chunks > 1024 ? 0 : (1024 / chunks) * chunks;
}
extern "C" int vpu_pixel_unshuffle2_sums(const void* in, void* out, float* part, int32_t B, int32_t h, int32_t w, int32_t C,
                                         int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    const int nblk = C % 8 ? 0 : vpu_pixel_unshuffle2_nblk(C);
    if (!in || !out || !part || nblk < 1 || C > 2048 || B < 1 || h < 1 || w < 1) {
        vpu_set_error("pixel_unshuffle2_sums: C % 8 == 0, C <= 2048, non-null pointers");
        return VPU_ERR_ARG;
    }
    DISPATCH_T(dtype, pixel_unshuffle2_sums_kernel<T><<<nblk, 256, 0, ST>>>((const T*)in, (T*)out, part, B, h, w, C);)
    return vpu_check_launch("vpu_pixel_unshuffle2_sums");
}
extern "C" int vpu_groupnorm_nchunk(void) {
    vpu_clear_stale_error(); return GN_CHUNKS; }
extern "C" int vpu_groupnorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                 double* stats, int32_t B, int64_t HW, int32_t C, float eps, int32_t gelu,
                                 int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || C > 4096 || (C > 2048 && C % 16)) { vpu_set_error("groupnorm: C % 8 (16 above 2048), C <= 4096"); return VPU_ERR_ARG; }
    dim3 grid(B, GN_CHUNKS);
    DISPATCH_T(dtype, gn_stats_kernel<T><<<grid, 256, 0, ST>>>((const T*)x, stats, HW, C);
               gn_apply_kernel<T><<<grid, 256, 0, ST>>>((const T*)x, w, b, (T*)y, mean, rstd, stats, HW, C, eps, gelu);)
    return vpu_check_launch("vpu_groupnorm_fwd");
}
extern "C" int vpu_pixel_shuffle2_gn_stats(const void* in, void* out, const float* bias, double* stats, int32_t B, int32_t h,
                                           int32_t w, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || !in || !out || !stats || B < 1 || h < 1 || w < 1) { vpu_set_error("pixel_shuffle2_gn_stats: C % 8, non-null pointers"); return VPU_ERR_ARG; }
    dim3 grid(B, GN_CHUNKS);
    DISPATCH_T(dtype, pixel_shuffle2_gn_kernel<T><<<grid, 256, 0, ST>>>((const T*)in, (T*)out, bias, stats, h, w, C);)
    return vpu_check_launch("vpu_pixel_shuffle2_gn_stats");
}
extern "C" int vpu_groupnorm_apply(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                   const double* stats, int32_t B, int64_t HW, int32_t C, float eps, int32_t gelu,
                                   int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || C > 4096 || (C > 2048 && C % 16)) { vpu_set_error("groupnorm: C % 8 (16 above 2048), C <= 4096"); return VPU_ERR_ARG; }
    dim3 grid(B, GN_CHUNKS);
    DISPATCH_T(dtype, gn_apply_kernel<T><<<grid, 256, 0, ST>>>((const T*)x, w, b, (T*)y, mean, rstd, stats, HW, C, eps, gelu);)
    return vpu_check_launch("vpu_groupnorm_apply");
}
extern "C" int vpu_groupnorm_bwd(const void* dy, const void* x, const float* w, const float* b, const float* mean,
                                 const float* rstd, void* dx, float* part, double* stats, int32_t B, int64_t HW,
                                 int32_t C, int32_t gelu, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || C > 4096 || (C > 2048 && C % 16)) { vpu_set_error("groupnorm_bwd: C % 8 (16 above 2048), C <= 4096"); return VPU_ERR_ARG; }
    dim3 grid(B, GN_CHUNKS);
    DISPATCH_T(dtype,
               gn_bwd_stats_kernel<T><<<grid, 256, 0, ST>>>((const T*)dy, (const T*)x, w, b, mean, rstd, part, stats, B,
                                                            HW, C, gelu);
               gn_bwd_dx_kernel<T><<<grid, 256, 0, ST>>>((const T*)dy, (const T*)x, w, b, mean, rstd, (T*)dx, stats, HW,
                                                         C, gelu);)
    return vpu_check_launch("vpu_groupnorm_bwd");
}
extern "C" int vpu_bilinear_cl_fwd(const void* in, int32_t ld_in, void* out, int32_t ld_out, int32_t B, int32_t h,
                                   int32_t w, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || ld_in % 8 || ld_out % 8) { vpu_set_error("bilinear_cl: C, ld % 8"); return VPU_ERR_ARG; }
    const int64_t total = (int64_t)B * H * W * (C / 8);
    DISPATCH_T(dtype, bilinear_cl_fwd_kernel<T><<<vpu_grid_for(total, 256, 16384), 256, 0, ST>>>(
        (const T*)in, ld_in, (T*)out, ld_out, B, h, w, H, W, C);)
    return vpu_check_launch("vpu_bilinear_cl_fwd");
}
extern "C" int vpu_upsum_relu(void* io, const void* const* z, const int32_t* h, const int32_t* w, int32_t n, int32_t B,
                             int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (!io || !z || !h || !w || n < 0 || n > 3 || C % 8 || B < 1 || H < 1 || W < 1) {
        vpu_set_error("upsum_relu: 0 <= n <= 3 maps, C % 8 == 0, non-null pointers");
        return VPU_ERR_ARG;
    }
    UpsumArgs a;
    a.n = n;
    for (int i = 0; i < 3; ++i) {
        a.z[i] = i < n ? z[i] : nullptr; a.h[i] = i < n ? h[i] : 1; a.w[i] = i < n ? w[i] : 1;
        if (i < n && (!z[i] || h[i] < 1 || w[i] < 1)) { vpu_set_error("upsum_relu: null / empty map"); return VPU_ERR_ARG; }
    }
    const int64_t total = (int64_t)B * H * W * (C / 8);
    bool quad = n > 0 && H % 2 == 0 && W % 2 == 0;
    for (int i = 0; i < n; ++i) quad = quad && H % h[i] == 0 && W % w[i] == 0 && H / h[i] >= 2 && W / w[i] >= 2;
    if (quad) {
        DISPATCH_T(dtype, upsum_relu2_kernel<T><<<vpu_grid_for(total / 4, 256, 16384), 256, 0, ST>>>((T*)io, a, B, H, W, C);)
    } else {
        DISPATCH_T(dtype, upsum_relu_kernel<T><<<vpu_grid_for(total, 256, 16384), 256, 0, ST>>>((T*)io, a, B, H, W, C);)
    }
    return vpu_check_launch("vpu_upsum_relu");
}
extern "C" int vpu_bilinear_cl_bwd(const void* dout, int32_t ld_out, void* din, int32_t ld_in, int32_t B, int32_t h,
                                   int32_t w, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || ld_in % 8 || ld_out % 8) { vpu_set_error("bilinear_cl_bwd: C, ld % 8"); return VPU_ERR_ARG; }
    const int64_t total = (int64_t)B * h * w * (C / 8);
    const int R = (h > 0 && w > 0 && H % h == 0 && W % w == 0 && H / h == W / w) ? H / h : 0;
    static const int split_min = [] { const char* e = vpu_lab_getenv("VPU_BILINEAR_SPLIT_R"); return e ? atoi(e) : 8; }();
    if ((R == 4 || R == 8) && R >= split_min && C == 256) {   // few input pixels, long gathers: four slices per window
        const unsigned g2 = (unsigned)(8 * ((((int64_t)B * h * w + 1) / 2 + 7) / 8));     // (a multiple of 8: XCD-ranged block order)
        DISPATCH_T(dtype,
                   if (R == 4) bilinear_cl_bwd_int4_kernel<T, 4><<<g2, 256, 0, ST>>>((const T*)dout, ld_out, (T*)din, ld_in, B, h, w);
                   else bilinear_cl_bwd_int4_kernel<T, 8><<<g2, 256, 0, ST>>>((const T*)dout, ld_out, (T*)din, ld_in, B, h, w);)
        return vpu_check_launch("vpu_bilinear_cl_bwd");
    }
    if (R == 2 || R == 4 || R == 8) {
        const int64_t nb8 = 8 * (((total + 255) / 256 + 7) / 8);
        const unsigned gi = (unsigned)(nb8 < 16384 ? nb8 : 16384);      // (a multiple of 8: XCD-ranged block order)
        DISPATCH_T(dtype,
                   if (R == 2) bilinear_cl_bwd_int_kernel<T, 2><<<gi, 256, 0, ST>>>(
                       (const T*)dout, ld_out, (T*)din, ld_in, B, h, w, C);
                   else if (R == 4) bilinear_cl_bwd_int_kernel<T, 4><<<gi, 256, 0, ST>>>(
                       (const T*)dout, ld_out, (T*)din, ld_in, B, h, w, C);
                   else bilinear_cl_bwd_int_kernel<T, 8><<<gi, 256, 0, ST>>>(
                       (const T*)dout, ld_out, (T*)din, ld_in, B, h, w, C);)
        return vpu_check_launch("vpu_bilinear_cl_bwd");
    }
    DISPATCH_T(dtype, bilinear_cl_bwd_kernel<T><<<vpu_grid_for(total, 256, 16384), 256, 0, ST>>>(
        (const T*)dout, ld_out, (T*)din, ld_in, B, h, w, H, W, C);)
    return vpu_check_launch("vpu_bilinear_cl_bwd");
}
extern "C" int vpu_gate_stats(const void* Q, const void* Kt, float* cg, int32_t* argq, float* sg, int32_t* argc,
                              int32_t B, int32_t nq, int32_t N, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    DISPATCH_T(dtype,
               gate_colmax_kernel<T><<<vpu_grid_for((int64_t)B * C, 256), 256, 0, ST>>>((const T*)Q, cg, argq, B, nq, C);
               gate_rowmax_kernel<T><<<(unsigned)(((int64_t)B * N + 3) / 4), 256, 0, ST>>>((const T*)Kt, sg, argc,
                                                                                          (int64_t)B * N, C);)
    return vpu_check_launch("vpu_gate_stats");
}
extern "C" int vpu_gate_apply(const void* x, const float* cg, const float* sg, void* out, int32_t B, int32_t N,
                              int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8) { vpu_set_error("gate_apply: C % 8"); return VPU_ERR_ARG; }
    const int64_t total = (int64_t)B * N * (C / 8);
    DISPATCH_T(dtype, gate_apply_kernel<T><<<vpu_grid_for(total, 256, 16384), 256, 0, ST>>>((const T*)x, cg, sg, (T*)out,
                                                                                           B, N, C);)
    return vpu_check_launch("vpu_gate_apply");
}
extern "C" int vpu_gate_bwd(const void* dout, const void* x, const float* cg, const int32_t* argq, const float* sg,
                            const int32_t* argc, void* dx, int32_t accum, void* dQ, void* dK, float* part, int32_t B,
                            int32_t nq, int32_t N, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C % 8 || C > 2048) { vpu_set_error("gate_bwd: C % 8, C <= 2048"); return VPU_ERR_ARG; }
    dim3 grid(B, GATE_NBLK);
    DISPATCH_T(dtype,
               gate_bwd_rows_kernel<T><<<grid, 256, 0, ST>>>((const T*)dout, (const T*)x, cg, sg, argc, (T*)dx, accum,
                                                             (T*)dK, part, N, C);
               gate_bwd_cols_kernel<T><<<vpu_grid_for((int64_t)B * C, 256), 256, 0, ST>>>(part, cg, argq, (T*)dQ, B, nq,
                                                                                         C);)
    return vpu_check_launch("vpu_gate_bwd");
}
extern "C" int vpu_gate_fwd_n(const void* const* Q, const void* const* K, const void* x, void* const* out, float* cg, int32_t* argq,
                              float* sg, int32_t* argc, int32_t n, int32_t B, int32_t nq, int32_t N, int32_t C, int32_t dtype,
                              void* stream) {
    vpu_clear_stale_error();
    if (n < 1 || n > GATE_MAXN || C % 8 || !Q || !K || !x || !out || !cg || !argq || !sg || !argc || B < 1 || nq < 1 || N < 1) {
        vpu_set_error("gate_fwd_n: 1 <= n <= 3 gates, C % 8 == 0, non-null pointers");
        return VPU_ERR_ARG;
    }
    GateSet gs{};
    gs.n = n;
    for (int i = 0; i < n; ++i) {
        gs.Q[i] = Q[i]; gs.K[i] = K[i]; gs.out[i] = out[i];
        if (!Q[i] || !K[i] || !out[i]) { vpu_set_error("gate_fwd_n: null operand"); return VPU_ERR_ARG; }
    }
    const int col_blocks = vpu_grid_for((int64_t)B * C, 256);
    const unsigned row_blocks = (unsigned)(((int64_t)B * N + 3) / 4);
    const int64_t total = (int64_t)B * N * (C / 8);
    DISPATCH_T(dtype,
               gate_stats_n_kernel<T><<<dim3(col_blocks + row_blocks, n), 256, 0, ST>>>(gs, cg, argq, sg, argc, B, nq, N, C, col_blocks);
               gate_apply_n_kernel<T><<<vpu_grid_for(total, 256, 16384), 256, 0, ST>>>((const T*)x, cg, sg, gs, B, N, C);)
    return vpu_check_launch("vpu_gate_fwd_n");
}
extern "C" int vpu_gate_bwd_n(const void* const* dout, const void* x, const float* cg, const int32_t* argq, const float* sg,
                              const int32_t* argc, void* dx, int32_t accum, void* const* dQ, void* const* dK, float* part, int32_t n,
                              int32_t B, int32_t nq, int32_t N, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (n < 1 || n > GATE_MAXN || C % 8 || C > 2048 || !dout || !x || !dx || !dQ || !dK || !part) {
        vpu_set_error("gate_bwd_n: 1 <= n <= 3 gates, C % 8 == 0, C <= 2048, non-null pointers");
        return VPU_ERR_ARG;
    }
    GateSet gs{};
    gs.n = n;
    for (int i = 0; i < n; ++i) {
        gs.out[i] = const_cast<void*>(dout[i]); gs.dQ[i] = dQ[i]; gs.dK[i] = dK[i];
        if (!dout[i] || !dQ[i] || !dK[i]) { vpu_set_error("gate_bwd_n: null operand"); return VPU_ERR_ARG; }
    }
    dim3 grid(B, GATE_NBLK);
    const int nchk = (C + 511) / 512;
#define VPU_GATE_ROWS(NCHK_)                                                                                                   \
    DISPATCH_T(dtype, (gate_bwd_rows_n_kernel<T, NCHK_><<<grid, 256, 0, ST>>>(gs, (const T*)x, cg, sg, argc, (T*)dx, accum, part, B, N, C));)
    if (nchk <= 1) { VPU_GATE_ROWS(1) } else if (nchk == 2) { VPU_GATE_ROWS(2) } else if (nchk == 3) { VPU_GATE_ROWS(3) } else { VPU_GATE_ROWS(4) }
#undef VPU_GATE_ROWS
    DISPATCH_T(dtype, gate_bwd_cols_n_kernel<T><<<dim3(vpu_grid_for((int64_t)B * C, 256), n), 256, 0, ST>>>(part, cg, argq, gs, B, nq, C);)
    return vpu_check_launch("vpu_gate_bwd_n");
}
extern "C" int vpu_convseg_fwd(const void* x, const float* w, const float* bias, const float* mask, float* out,
                               int64_t rows, int64_t HW, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if ((C == 64 || C == 128 || C == 256 || C == 512) && reinterpret_cast<uintptr_t>(x) % 32 == 0 &&
        (reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(mask)) % 32 == 0) {
        const int64_t trips = (rows + (64 / (C / 8)) - 1) / (64 / (C / 8));   // wave trips
        const unsigned grid = (unsigned)(trips / 4 < 1 ? 1 : (trips / 4 > 4096 ? 4096 : trips / 4));
        DISPATCH_T(dtype, convseg_fwd_vec_kernel<T><<<grid, 256, 0, ST>>>((const T*)x, w, bias, mask, out, rows, HW, C);)
        return vpu_check_launch("vpu_convseg_fwd");
    }
    DISPATCH_T(dtype, convseg_fwd_kernel<T><<<(unsigned)((rows + 3) / 4), 256, 0, ST>>>((const T*)x, w, bias, mask, out,
                                                                                       rows, HW, C);)
    return vpu_check_launch("vpu_convseg_fwd");
}
extern "C" int vpu_convseg_bwd_nblk(int64_t rows) {
    vpu_clear_stale_error();
    // (workgroups of the conv_seg / fused head backward = rows of their partial-sum output; VPU_CONVSEG_NBLK sets the cap for
    // A/B runs -- measured in the step, round 4: 512 / 1024 / 2048 / 4096 workgroups 12.66-12.68 / 12.61 / 12.68-12.71 / 12.69 ms)
    static const int cap = [] { const char* e = vpu_lab_getenv("VPU_CONVSEG_NBLK"); const int v = e ? atoi(e) : 512; return v < 1 ? 1 : v; }();
    int64_t n = rows / 64;
    return (int)(n < 1 ? 1 : (n > cap ? cap : n));
}
extern "C" int vpu_convseg_bwd(const float* dout, const void* x, const float* w, const float* mask, void* dx,
                               int32_t accum, float* part, float* part_b, int64_t rows, int64_t HW, int32_t C,
                               int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (C > 512) { vpu_set_error("convseg_bwd: C <= 512"); return VPU_ERR_ARG; }
    const int nblk = vpu_convseg_bwd_nblk(rows);
    if (dtype == VPU_BF16 && (C == 64 || C == 128 || C == 256 || C == 512) &&
        (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) % 16 == 0 &&
        (reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(mask)) % 32 == 0) {
        convseg_bwd_vec_kernel<<<nblk, 256, 0, ST>>>(dout, (const bf16_t*)x, w, mask, (bf16_t*)dx, accum, part, part_b, rows,
                                                    HW, C, nblk);
        return vpu_check_launch("vpu_convseg_bwd");
    }
    DISPATCH_T(dtype, convseg_bwd_kernel<T><<<nblk, 256, 0, ST>>>(dout, (const T*)x, w, mask, (T*)dx, accum, part,
                                                                 part_b, rows, HW, C, nblk);)
    return vpu_check_launch("vpu_convseg_bwd");
}
extern "C" int vpu_head_grad_fused(const void* dfn, const void* y, const float* inv, const float* dout, const void* x,
                                   const float* w, const float* mask, void* dx, float* part, float* part_b, int64_t rows,
                                   int64_t HW, int32_t C, int32_t dtype, void* stream) {
    vpu_clear_stale_error();
    if (dtype != VPU_BF16 || !(C == 64 || C == 128 || C == 256 || C == 512) || !dfn || !y || !inv || !dout || !x || !w ||
        !dx || !part || !part_b || rows < 1 ||
        (reinterpret_cast<uintptr_t>(dfn) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x) |
         reinterpret_cast<uintptr_t>(dx)) % 16 != 0 ||
        (reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(mask)) % 32 != 0) {
        vpu_set_error("head_grad_fused: bf16, C in {64,128,256,512}, 16-byte aligned maps, 32-byte aligned w / mask");
        return VPU_ERR_ARG;
    }
    const int nblk = vpu_convseg_bwd_nblk(rows);
    head_grad_fused_kernel<<<nblk, 256, 0, ST>>>((const bf16_t*)dfn, (const bf16_t*)y, inv, dout, (const bf16_t*)x, w, mask,
                                                 (bf16_t*)dx, part, part_b, rows, HW, C, nblk);
    return vpu_check_launch("vpu_head_grad_fused");
}
extern "C" int vpu_upsample_ac_fwd(const float* in, float* out, int64_t planes, int32_t h, int32_t w, int32_t H,
                                   int32_t W, void* stream) {
    vpu_clear_stale_error();
    upsample_ac_fwd_kernel<<<vpu_grid_for(planes * H * W, 256, 65536), 256, 0, ST>>>(in, out, planes, h, w, H, W);
    return vpu_check_launch("vpu_upsample_ac_fwd");
}
extern "C" int vpu_upsample_ac_bwd(const float* dout, float* din, int64_t planes, int32_t h, int32_t w, int32_t H,
                                   int32_t W, void* stream) {
    vpu_clear_stale_error();
    upsample_ac_bwd_kernel<<<vpu_grid_for(planes * h * w, 256, 65536), 256, 0, ST>>>(dout, din, planes, h, w, H, W);
    return vpu_check_launch("vpu_upsample_ac_bwd");
}
