// Losses of the VPUFormer training step with their gradients, fp32:
//   P2CL  = SigmoidBinaryCrossEntropyLoss(from_sigmoid=True)   isegm/model/losses.py:155-176
//           against ed_mask_label built on the fly             isegm/engine/trainer.py:329-331,756,764
//   NFL   = NormalizedFocalLossSigmoid(alpha .5, gamma 2)      isegm/model/losses.py:11-89
//   Dice  = DiceLoss(sigmoid, naive_dice)                      isegm/model/losses.py:227-363
// Bandwidth-bound reductions: one block per plane / per sample, deterministic (no atomics).
#include "vpu_common.h"
#include "../../include/vpu_hip.h"

#define ST reinterpret_cast<hipStream_t>(stream)

namespace {

__global__ __launch_bounds__(1024) void p2cl_kernel(const float* __restrict__ prob, const float* __restrict__ gt,
                                                    const int* __restrict__ slot_idx,
                                                    const float* __restrict__ override_masks,
                                                    float* __restrict__ loss_part, float* __restrict__ dprob,
                                                    float grad_scale, int S, int64_t HW) {
    __shared__ double red[16];
    const int plane = blockIdx.x;
    const int b = plane / S, s = plane % S;
    const float* p = prob + (int64_t)plane * HW;
    const int ov = slot_idx ? slot_idx[plane] : -1;
    const float* lab = ov >= 0 ? override_masks + (int64_t)ov * HW : gt + (int64_t)b * HW;
    const bool invert = ov < 0 && s >= S / 2;
    float* dp = dprob ? dprob + (int64_t)plane * HW : nullptr;
    double acc = 0.0;
    for (int64_t i4 = threadIdx.x; i4 < HW / 4; i4 += 1024) {
        const float4 pv = *reinterpret_cast<const float4*>(p + i4 * 4);
        const float4 lv = *reinterpret_cast<const float4*>(lab + i4 * 4);
        const float pp[4] = {pv.x, pv.y, pv.z, pv.w};
        const float ll[4] = {lv.x, lv.y, lv.z, lv.w};
        float g[4];
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = ll[j];
            const bool valid = y != -1.0f;  // ignore_label (never set by ed_mask_label, kept for fidelity)
            if (invert) y = (y != 0.f) ? 0.f : 1.f;  // logical_not (trainer.py:330)
            if (!valid) y = 0.f;
            const float a = pp[j] + 1e-12f, c = 1.f - pp[j] + 1e-12f;
            const float l = -(logf(a) * y + logf(c) * (1.f - y));
            part += valid ? l : 0.f;
            g[j] = valid ? grad_scale * (-(y / a) + (1.f - y) / c) : 0.f;
        }
        acc += part;
        if (dp) *reinterpret_cast<float4*>(dp + i4 * 4) = make_float4(g[0], g[1], g[2], g[3]);
    }
    const double t = block_sum_d(acc, red);
    if (threadIdx.x == 0) loss_part[plane] = (float)t;
}

// Fused  F.interpolate(sim, align_corners=True)  ->  P2CL  ->  gradient w.r.t. the LOW-resolution similarities
// (is_vpu_model.py:431-436 + losses.py:155-176 + their backward).  The unfused chain writes / reads the [B,S,H,W]
// upsampled tensor and its gradient four times (1.85 GB per step at bs 12); this kernel reads one 50-KB plane into LDS,
// evaluates every full-resolution pixel exactly once and writes one 50-KB gradient plane.
// One block per (b, slot) plane.  Each thread owns low-resolution cells (y0, x0); the pixels whose bilinear anchor is
// that cell contribute to the 4 cell corners, accumulated in registers and added to the LDS gradient plane in four
// barrier-separated phases (in each phase every LDS word has exactly one writer -> bitwise reproducible, no atomics).
__device__ __forceinline__ int ac_i0(int dst, float scale, int in_size) {
    int i0 = (int)(scale * (float)dst);
    return i0 > in_size - 1 ? in_size - 1 : i0;
}
// One block per (plane, band of P2_BAND low-resolution anchor rows) plus one halo row above.  Two passes:
//  1. pixel pass -- every full-resolution pixel anchored in the band (and in the halo row) is evaluated ONCE by one lane
//     (4 consecutive pixels per thread, 16-byte label loads): interpolated probability, loss term, d loss / d prob -> LDS;
//  2. cell pass -- one thread per low-resolution cell gathers the gradients of its <= P2_SPAN x P2_SPAN pixels from LDS,
//     weighted per corner; the corner sums are added to the LDS gradient band in four barrier-separated phases (in each
//     phase every LDS word has exactly one writer -> bitwise reproducible, no atomics).
// The block owns the gradient rows [r0, r1) of its band: the halo row r0-1 contributes its lower corners (row r0), the
// lower corners of the last anchor row are dropped (the next band's halo recomputes them); losses are counted for the
// band's own anchor rows only.
constexpr int P2_BAND_MAX = 8;   // anchor rows per block: min(8, 1024 / w - 1) so that (band + 1) * w threads fit one block
constexpr int P2_SPAN = 5;   // most full-resolution pixels per low-resolution cell and axis
constexpr int P2_W_MAX = 512;   // widest low-resolution map (host-side check)

// workgroup barrier that waits for this wave's LDS operations only: `__syncthreads()` also waits for every global load in flight --
// the next item's labels, requested on purpose long before they are needed
__device__ __forceinline__ void p2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Round 6: PERSISTENT.  The kernel is bound by instruction issue (16 waves on 4 SIMDs: whatever a wave executes costs the SIMD four
// times its cycles), and a block per (plane, band) paid its setup 8064 times: time stamps (`tools/p2_stamps.py`, experiment build)
// split such a block's 23,600 cycles into 7,250 until its operands were in LDS (anchor-row searches, 64-bit address arithmetic, the
// loads), 3,070 for the horizontal interpolation (an integer division and an anchor computation per element), 8,290 for the pixel pass
// with the loss reduction (every thread summed the sixteen wave partials in double precision), 3,740 for the cell pass and 1,270 for
// the folds and the store.  Now a workgroup walks a contiguous range of (plane, band) items: the row-anchor table of the WHOLE map and
// every per-thread constant are computed once per workgroup; the labels of the next item go into the label registers as soon as the
// pixel pass has consumed the current ones, its low-resolution rows into two registers (in flight under the cell pass, the folds and
// the store: the barriers behind the request wait for LDS only); the interpolation pass gives a thread one column and every second row;
// thread 0 alone adds the wave partials (same order, same bits) behind ONE barrier; the test for a label that is neither 0 nor 1 is
// one fma and one OR per label; no SLP packing (build.sh).  16,600 -> ~14,500 cycles per item, of which the pixel pass is 7,200 (the
// last wave's; the first one's 3,500); 344 -> 216 us per launch, bit-identical (tools/p2_compare.py: soft and ignore labels included).
// (experiment builds only -- VPU_X_loss="-DVPU_P2_STAMPS -DP2_STAMP_THREAD=0" bash build.sh x, tools/p2_stamps.py: one thread's
// s_memtime at the phases of every item)
#ifdef VPU_P2_STAMPS
__device__ unsigned g_p2_dbg[8192 * 8];
#define P2_STAMP(k) do { if (threadIdx.x == P2_STAMP_THREAD && it < 8192) g_p2_dbg[it * 8 + (k)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define P2_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(1024) void p2cl_up_kernel(const float* __restrict__ low, const float* __restrict__ gt,
                                                       const int* __restrict__ slot_idx,
                                                       const float* __restrict__ override_masks,
                                                       float* __restrict__ loss_part, float* __restrict__ dlow,
                                                       float grad_scale, int S, int h, int w, int H, int W, int nband,
                                                       int max_rows, int BAND, int nitems) {
    extern __shared__ float sm[];   // [(BAND+2)*w] values (rows r0-1..r1) | [BAND*w] gradients | [(BAND+2)*W] hr | [max_rows*W] d loss / d prob
    __shared__ double red[16];
    // first full-resolution column / row whose bilinear anchor is >= a given low-resolution column / anchor row of the block
    // (the anchor index is monotone, so cell x0 owns columns [tXa[x0], tXa[x0 + 1])).  Built once per block: the search loops
    // used to run in every thread of both passes (PMC: 1969 VALU instructions per wave, the kernel is VALU-bound).
    __shared__ int tXa[P2_W_MAX + 2], tYall[P2_W_MAX + 4];      // tYall[1 + y]: first full-resolution row anchored at or below row y, y = -1 .. h + 1
    float* const sv = sm;                        // sv[(y - (r0 - 1)) * w + x]
    float* const sg = sm + (BAND + 2) * w;    // sg[(y - r0) * w + x]
    float* const hr = sg + BAND * w;          // hr[(y - (r0 - 1)) * W + X]: rows of sv interpolated along x
    // (gp in raster order: the cell pass's stride-4 gathers are 4-way bank conflicts -- 27 % of the kernel's LDS cycles, PMC --
    // but a column-permuted layout that makes them conflict-free costs four 4-byte stores per pixel quad in the pixel pass
    // and measured slower, 570 vs 547 us: the kernel is issue-bound, not LDS-bound)
    float* const gp = hr + (BAND + 2) * W;    // gp[(Y - Y0) * W + X]
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float fh = sh > 0.f ? 1.f / sh : 0.f;
    // first full-resolution row anchored at or below low-resolution row `arow`
    auto first_row = [&](int arow) {
        int Yf = 0;
        if (arow > 0) {
            Yf = (int)(fh * (float)arow) - 1; Yf = Yf < 0 ? 0 : (Yf > H ? H : Yf);
            while (Yf < H && ac_i0(Yf, sh, h) < arow) ++Yf;
        }
        return Yf;
    };
    // the column table and the row table are the same for every item
    for (int i = threadIdx.x; i <= w; i += blockDim.x) tXa[i] = W;
    for (int i = threadIdx.x; i < h + 3; i += blockDim.x) tYall[i] = first_row(i - 1);
    __syncthreads();
    for (int X = threadIdx.x; X < W; X += blockDim.x) {
        const int cur = ac_i0(X, sw, w), prev = X > 0 ? ac_i0(X - 1, sw, w) : -1;
        for (int x = prev + 1; x <= cur; ++x) tXa[x] = X;
    }
    const int W4 = W >> 2;
    // pixel-pass mapping: thread -> (anchor row yq = r0-1+g, 4-pixel column group X4); it walks the <= P2_SPAN pixel
    // rows anchored at yq
    const int pg = threadIdx.x / W4, X4 = (threadIdx.x - pg * W4) * 4;
    const int nsv = (BAND + 2) * w;
    // this workgroup's items: a contiguous range (consecutive bands of a plane, then the next plane)
    const int it0 = (int)(((int64_t)blockIdx.x * nitems) / gridDim.x), it1 = (int)(((int64_t)(blockIdx.x + 1) * nitems) / gridDim.x);
    // per-thread operands of an item's pixel pass and its share of the low-resolution rows: requested one item ahead
    float4 lv[P2_SPAN];
    float svn[2];
    int Ya = 0, Yb = 0;
    int svy[2], svx[2];          // this thread's two elements of the (BAND + 2) x w rows: row (or -1: none) and column
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = threadIdx.x + j * (int)blockDim.x;
        svy[j] = i < nsv ? i / w : -1;
        svx[j] = i < nsv ? i % w : 0;
    }
    auto request = [&](int itn) {
        const int plane = itn / nband, band = itn - plane * nband;
        const int r0 = band * BAND, r1 = (r0 + BAND < h) ? r0 + BAND : h;
        const int ov = slot_idx ? slot_idx[plane] : -1;
        const float* lab = ov >= 0 ? override_masks + (int64_t)ov * H * W : gt + (int64_t)(plane / S) * H * W;
        const int yq = r0 - 1 + pg;
        Ya = 0; Yb = 0;
        if (pg <= (r1 - r0) && yq >= 0 && yq < h) { Ya = tYall[1 + yq]; Yb = tYall[2 + yq]; }
#pragma unroll
        for (int k = 0; k < P2_SPAN; ++k) {
            lv[k] = make_float4(0.f, 0.f, 0.f, 0.f);      // (a row this thread does not own: never used, and "plain" below)
            if (Ya + k < Yb) lv[k] = *reinterpret_cast<const float4*>(lab + ((Ya + k) * W + X4));
        }
        const float* lowp = low + (int64_t)plane * h * w;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int y = r0 - 1 + svy[j];
            svn[j] = (svy[j] >= 0 && y >= 0 && y < h) ? lowp[y * w + svx[j]] : 0.f;
        }
    };
    if (it0 < it1) request(it0);
#pragma nounroll
    for (int it = it0; it < it1; ++it) {
        const int plane = it / nband, band = it - plane * nband;
        const int s = plane % S;
        P2_STAMP(0);
        const int r0 = band * BAND, r1 = (r0 + BAND < h) ? r0 + BAND : h;
        const int ov = slot_idx ? slot_idx[plane] : -1;
        const bool invert = ov < 0 && s >= S / 2;
        // pixel rows of the block: the first row anchored at max(r0-1, 0) .. the last row anchored at r1-1
        const int ya = r0 > 0 ? r0 - 1 : 0;
        const int Y0 = tYall[1 + ya];
        int Y1 = tYall[1 + r1];                                // one past the last row anchored below r1
        if (Y1 - Y0 > max_rows) Y1 = Y0 + max_rows;            // (cannot happen: host-side bound)
        const int* const tYa = tYall + r0;                     // tYa[k]: anchor row r0 - 1 + k
        const int yq = r0 - 1 + pg;
        const bool plive = pg <= (r1 - r0) && yq >= 0 && yq < h;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = threadIdx.x + j * (int)blockDim.x;
            if (i < nsv) sv[i] = svn[j];
        }
        for (int i = threadIdx.x; i < BAND * w; i += blockDim.x) sg[i] = 0.f;
        __syncthreads();
        P2_STAMP(1);
        // ---- pass 0: horizontal interpolation of the band's low-resolution rows, hr[row][X] = hx*v[x0] + lx*v[x1]
        // (the inner sums of the align_corners=True formula  hy*(hx*v00 + lx*v01) + ly*(hx*v10 + lx*v11), same rounding):
        // a thread keeps one column (anchor and weights once) and walks every (blockDim / W)-th row
        {
            const int nrp = (int)blockDim.x / W;
            if (nrp > 0) {
                if ((int)threadIdx.x < nrp * W) {
                    const int rr0 = threadIdx.x / W, X = threadIdx.x - rr0 * W;
                    const int x0 = ac_i0(X, sw, w);
                    const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
                    const float lx = sw * (float)X - (float)x0, hx = 1.f - lx;
                    for (int rr = rr0; rr < BAND + 2; rr += nrp) hr[rr * W + X] = hx * sv[rr * w + x0] + lx * sv[rr * w + x1];
                }
            } else {
                for (int i = threadIdx.x; i < (BAND + 2) * W; i += blockDim.x) {
                    const int rr = i / W, X = i - rr * W;
                    const int x0 = ac_i0(X, sw, w);
                    const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
                    const float lx = sw * (float)X - (float)x0, hx = 1.f - lx;
                    hr[i] = hx * sv[rr * w + x0] + lx * sv[rr * w + x1];
                }
            }
        }
        __syncthreads();
        P2_STAMP(2);
        // ---- pass 1: pixels
        float part = 0.f;
        if (plive) {
            const bool own = yq >= r0;
            const int yq1 = yq + (yq < h - 1 ? 1 : 0);
            const float4 t0 = *reinterpret_cast<const float4*>(hr + (yq - r0 + 1) * W + X4);
            const float4 t1 = *reinterpret_cast<const float4*>(hr + (yq1 - r0 + 1) * W + X4);
            const float top[4] = {t0.x, t0.y, t0.z, t0.w}, bot[4] = {t1.x, t1.y, t1.z, t1.w};
            // one test per THREAD for what almost never occurs -- a soft label (neither 0 nor 1) or the ignore label among its
            // <= 20 values -- instead of three compares per pixel: the common loop below knows every label is 0 or 1
            // (l * l - l is +0 exactly for l = 0, -0, 1 and for nothing else -- a NaN stays a NaN: the bit patterns OR-ed, one fma and one
            // OR per label instead of two compares, two mask operations and the row test)
            unsigned nz = 0u;
#pragma unroll
            for (int k = 0; k < P2_SPAN; ++k) {
                const float ls[4] = {lv[k].x, lv[k].y, lv[k].z, lv[k].w};
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) nz |= __float_as_uint(__builtin_fmaf(ls[q4], ls[q4], -ls[q4]));
            }
            const bool plain = nz == 0u;
            if (plain) {
#pragma unroll
                for (int k = 0; k < P2_SPAN; ++k) {
                    const int Y = Ya + k;
                    if (Y < Yb) {
                        const float ly = sh * (float)Y - (float)yq, hy = 1.f - ly;
                        const float ls[4] = {lv[k].x, lv[k].y, lv[k].z, lv[k].w};
                        float gout[4];
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const float pr = hy * top[q4] + ly * bot[q4];
                            const bool yb = (ls[q4] != 0.f) != invert;     // the label after logical_not (trainer.py:330)
                            const float a = pr + 1e-12f, c = 1.f - pr + 1e-12f;
                            // one raw v_log_f32 (log2; the arguments are >= 1e-12, far from the denormals that logf's
                            // ~14-instruction wrapper guards) and one reciprocal, no branch
                            const float q = yb ? a : c;
                            part += -0.69314718056f * __builtin_amdgcn_logf(q);
                            const float rq = __builtin_amdgcn_rcpf(q) * grad_scale;
                            gout[q4] = yb ? -rq : rq;
                        }
                        *reinterpret_cast<float4*>(gp + (Y - Y0) * W + X4) = make_float4(gout[0], gout[1], gout[2], gout[3]);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < P2_SPAN; ++k) {
                    const int Y = Ya + k;
                    if (Y < Yb) {
                        const float ly = sh * (float)Y - (float)yq, hy = 1.f - ly;
                        const float ls[4] = {lv[k].x, lv[k].y, lv[k].z, lv[k].w};
                        float gout[4];
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const float pr = hy * top[q4] + ly * bot[q4];
                            float y = ls[q4];
                            const bool valid = y != -1.0f;   // ignore_label (never set by ed_mask_label, kept for fidelity)
                            if (invert) y = (y != 0.f) ? 0.f : 1.f;   // logical_not (trainer.py:330)
                            const float a = pr + 1e-12f, c = 1.f - pr + 1e-12f;
                            const bool yb = y != 0.f;
                            const float q = yb ? a : c;
                            float l = -0.69314718056f * __builtin_amdgcn_logf(q);
                            const float rq = __builtin_amdgcn_rcpf(q);
                            float g = yb ? -rq : rq;
                            if (valid && yb && y != 1.f) {   // soft label: the general form
                                l = -0.69314718056f * (__builtin_amdgcn_logf(a) * y + __builtin_amdgcn_logf(c) * (1.f - y));
                                g = -(y * __builtin_amdgcn_rcpf(a)) + (1.f - y) * __builtin_amdgcn_rcpf(c);
                            }
                            part += valid ? l : 0.f;
                            gout[q4] = valid ? g * grad_scale : 0.f;
                        }
                        *reinterpret_cast<float4*>(gp + (Y - Y0) * W + X4) = make_float4(gout[0], gout[1], gout[2], gout[3]);
                    }
                }
            }
            if (!own) part = 0.f;      // (the halo row's pixels belong to the band above)
        }
        // the next item's labels take the registers the pixel pass has just consumed; with its low-resolution rows they are in flight
        // under the reduction, the cell pass, the folds and the store
        P2_STAMP(7);
        if (it + 1 < it1) request(it + 1);
        P2_STAMP(3);
        // the block's loss: wave partials in double precision, added by ONE thread in wave order (the barriers also publish gp)
        {
            const double wv = wave_sum_d((double)part);
            const int wi = threadIdx.x >> 6, nw = blockDim.x >> 6;
            if ((threadIdx.x & 63) == 0) red[wi] = wv;      // (thread 0 read the previous item's partials six barriers ago)
            p2_barrier();
            if (threadIdx.x == 0) {
                double t = 0.0;
                for (int i = 0; i < nw; ++i) t += red[i];
                loss_part[(int64_t)plane * nband + band] = (float)t;
            }
        }
        P2_STAMP(4);
        if (dlow) {
            // ---- pass 2: thread -> cell (y0, x0) with y0 in [r0-1, r1)
            const int ly_ = threadIdx.x / w, x0 = threadIdx.x % w;
            const int y0 = r0 - 1 + ly_;
            const bool live = ly_ <= (r1 - r0) && y0 >= 0 && y0 < h;
            const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
            float g00 = 0.f, g01 = 0.f, g10 = 0.f, g11 = 0.f;
            if (live) {
                // the pixels anchored at this cell form a contiguous range in each axis (the anchor index is monotone): <= P2_SPAN
                // columns x <= P2_SPAN rows.  Column weights once per cell; a tap outside the range reads the cell's first pixel
                // with weight zero (every gradient in gp is finite: |g| <= 1e12 * grad_scale), so the 25 taps are one LDS read with
                // an immediate offset and two FMAs each, no compare
                const int Ya = tYa[ly_], Yb = tYa[ly_ + 1], Xa = tXa[x0], Xb = tXa[x0 + 1];
                float wl0[P2_SPAN], wl1[P2_SPAN];
                int xo[P2_SPAN];
#pragma unroll
                for (int ix = 0; ix < P2_SPAN; ++ix) {
                    const int X = Xa + ix;
                    const bool in = X < Xb;
                    const float lx = sw * (float)X - (float)x0;
                    wl0[ix] = in ? 1.f - lx : 0.f;
                    wl1[ix] = in ? lx : 0.f;
                    xo[ix] = in ? ix : 0;
                }
#pragma unroll
                for (int iy = 0; iy < P2_SPAN; ++iy) {
                    const int Y = Ya + iy;
                    if (Y < Yb && Xa < Xb) {     // (a cell that owns no column reads nothing: grow[0] would be the NEXT row's first pixel)
                        const float ly = sh * (float)Y - (float)y0, hy = 1.f - ly;
                        const float* grow = gp + (Y - Y0) * W + Xa;
                        float rx0 = 0.f, rx1 = 0.f;   // this row's gradient split over the left / right corner columns
#pragma unroll
                        for (int ix = 0; ix < P2_SPAN; ++ix) {
                            const float g = grow[xo[ix]];
                            rx0 += g * wl0[ix]; rx1 += g * wl1[ix];
                        }
                        g00 += hy * rx0; g01 += hy * rx1; g10 += ly * rx0; g11 += ly * rx1;
                    }
                }
            }
            // clamped edge cells (x1 == x0 or y1 == y0) would alias a neighbour's target word: fold their (zero-weight)
            // corner terms into the cell's own word and skip the write
            const bool bx = x1 != x0, by = y1 != y0;
            if (!bx) { g00 += g01; g10 += g11; }
            if (!by) { g00 += g10; if (bx) g01 += g11; }
            const bool up = live && y0 >= r0;              // upper corners land on row y0 (inside the band unless halo)
            const bool dn = live && by && y1 < r1;         // lower corners land on row y1 (dropped for the last anchor row)
            if (up) sg[(y0 - r0) * w + x0] += g00;
            p2_barrier();
            if (up && bx) sg[(y0 - r0) * w + x1] += g01;
            p2_barrier();
            if (dn) sg[(y1 - r0) * w + x0] += g10;
            p2_barrier();
            if (dn && bx) sg[(y1 - r0) * w + x1] += g11;
            p2_barrier();
            P2_STAMP(5);
            for (int i = threadIdx.x; i < (r1 - r0) * w; i += blockDim.x)
                dlow[(int64_t)plane * h * w + (int64_t)r0 * w + i] = sg[i];
        }
        p2_barrier();      // the item's LDS is free
        P2_STAMP(6);
    }
}

// NFL + Dice in three launches over (sample, chunk) blocks: per-chunk partial sums -> gradients + per-chunk loss partials
// (every block re-reduces the NFL_NBLK partial rows of its sample in a fixed order) -> per-sample losses.
// scratch layout (double): sums[b][1 + chunk][0..4] = sum w, sum beta, sum p*t, sum p, sum t ; [5] = NFL loss partial.
constexpr int NFL_NBLK = 32;

__global__ __launch_bounds__(256) void nfl_dice_sums_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ gt, double* __restrict__ sums,
                                                            int64_t HW) {
    __shared__ double red[16];
    const int b = blockIdx.y, blk = blockIdx.x;
    const int64_t n4 = HW / 4, per = (n4 + NFL_NBLK - 1) / NFL_NBLK;
    const int64_t lo = blk * per, hi = lo + per < n4 ? lo + per : n4;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t i4 = lo + threadIdx.x; i4 < hi; i4 += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(logits + (int64_t)b * HW + i4 * 4);
        const float4 tv = *reinterpret_cast<const float4*>(gt + (int64_t)b * HW + i4 * 4);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ts[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = xs[j], t = ts[j];
            const float p = 1.f / (1.f + expf(-x));
            const float w = t != -1.0f ? 1.f : 0.f;
            const float pt = w > 0.f ? 1.f - fabsf(t - p) : 1.f;
            const float beta = (1.f - pt) * (1.f - pt);
            a[0] += w; a[1] += beta; a[2] += p * t; a[3] += p; a[4] += t;
        }
    }
    for (int k = 0; k < 5; ++k) {
        const double v = block_sum_d((double)a[k], red);
        if (threadIdx.x == 0) sums[((int64_t)b * (NFL_NBLK + 1) + 1 + blk) * 8 + k] = v;
    }
}

__device__ __forceinline__ void nfl_sample_sums(const double* __restrict__ sums, int b, float (&o)[5]) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double t = 0.0;
        for (int c = 0; c < NFL_NBLK; ++c) t += sums[((int64_t)b * (NFL_NBLK + 1) + 1 + c) * 8 + k];
        o[k] = (float)t;
    }
}

__global__ __launch_bounds__(256) void nfl_dice_grad_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ gt, double* __restrict__ sums,
                                                            float* __restrict__ dlogits, float w_nfl, float w_dice,
                                                            int64_t HW) {
    __shared__ double red[16];
    const int b = blockIdx.y, blk = blockIdx.x;
    const float eps = 1e-12f;
    float sm[5];
    nfl_sample_sums(sums, b, sm);
    const float sw = sm[0], bs = sm[1], A = sm[2], Bp = sm[3], Ct = sm[4];
    const float mult = sw / (bs + eps);          // detached (losses.py:57-59)
    const float inv_bsum = 1.f / (sw + eps);     // size_average (losses.py:80-82)
    const float deps = 1e-3f;
    const float den = Bp + Ct + deps;
    const int64_t n4 = HW / 4, per = (n4 + NFL_NBLK - 1) / NFL_NBLK;
    const int64_t lo = blk * per, hi = lo + per < n4 ? lo + per : n4;
    float acc = 0.f;
    for (int64_t i4 = lo + threadIdx.x; i4 < hi; i4 += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(logits + (int64_t)b * HW + i4 * 4);
        const float4 tv = *reinterpret_cast<const float4*>(gt + (int64_t)b * HW + i4 * 4);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ts[4] = {tv.x, tv.y, tv.z, tv.w};
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = xs[j], t = ts[j];
            const float p = 1.f / (1.f + expf(-x));
            const float w = t != -1.0f ? 1.f : 0.f;
            const float alpha = 0.5f * w;  // alpha = 1-alpha = 0.5
            const float pt = w > 0.f ? 1.f - fabsf(t - p) : 1.f;
            const float om = 1.f - pt;
            const float beta = om * om * mult;
            const float arg = fminf(pt + eps, 1.f);
            const float lg = logf(arg);
            acc += -alpha * beta * lg * w;
            // d/dpt of (1-pt)^2 * log(min(pt+eps,1))
            const float dfd = -2.f * om * lg + om * om * ((pt + eps < 1.f) ? 1.f / (pt + eps) : 0.f);
            const float dptdp = (t - p) > 0.f ? 1.f : ((t - p) < 0.f ? -1.f : 0.f);
            const float dp_dx = p * (1.f - p);
            const float g_nfl = -alpha * mult * w * dfd * dptdp * dp_dx * inv_bsum;
            const float dd_dp = (2.f * t * den - (2.f * A + deps)) / (den * den);
            const float g_dice = -dd_dp * dp_dx;
            g[j] = w_nfl * g_nfl + w_dice * g_dice;
        }
        if (dlogits) *reinterpret_cast<float4*>(dlogits + (int64_t)b * HW + i4 * 4) = make_float4(g[0], g[1], g[2], g[3]);
    }
    const double L = block_sum_d((double)acc, red);
    if (threadIdx.x == 0) sums[((int64_t)b * (NFL_NBLK + 1) + 1 + blk) * 8 + 5] = L;
}

__global__ void nfl_dice_final_kernel(const double* __restrict__ sums, float* __restrict__ out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float sm[5];
    nfl_sample_sums(sums, b, sm);
    double L = 0.0;
    for (int c = 0; c < NFL_NBLK; ++c) L += sums[((int64_t)b * (NFL_NBLK + 1) + 1 + c) * 8 + 5];
    const float deps = 1e-3f;
    out[b * 2 + 0] = (float)L * (1.f / (sm[0] + 1e-12f));
    out[b * 2 + 1] = 1.f - (2.f * sm[2] + deps) / (sm[3] + sm[4] + deps);
}

// The logged scalars of one click iteration from the kernels' per-sample / per-plane partials in ONE launch (the torch
// expression -- two means, a sum, a weighted total -- was eleven 5-us launches per step):
// res = {total, nfl, dice, p2cl},  nfl = mean_b out[b][0], dice = mean_b out[b][1], p2cl = sum(part) * inv_count,
// total = (w_nfl nfl + w_dice dice + w_pcl p2cl) * iter_weight  (trainer.py:399-419).  Sums in double, fixed order.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ out, const float* __restrict__ part,
                                                            int B, int npart, double inv_count, float w_nfl, float w_dice,
                                                            float w_pcl, float iter_weight, float* __restrict__ res) {
    __shared__ double red[16];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) { a += out[2 * i]; b += out[2 * i + 1]; }
#pragma unroll 8
    for (int i = threadIdx.x; i < npart; i += 256) c += part[i];      // (eight loads in flight: 8064 band partials at bs 12)
    a = block_sum_d(a, red);
    b = block_sum_d(b, red);
    c = block_sum_d(c, red);
    if (threadIdx.x == 0) {
        const float nfl = (float)(a / B), dice = (float)(b / B), pcl = (float)(c * inv_count);
        res[0] = (w_nfl * nfl + w_dice * dice + w_pcl * pcl) * iter_weight;
        res[1] = nfl; res[2] = dice; res[3] = pcl;
    }
}

}  // namespace

extern "C" int vpu_p2cl_fwd_bwd(const float* prob, const float* gt, const int32_t* slot_mask_idx,
                                const float* override_masks, float* loss_part, float* dprob, float grad_scale, int32_t B,
                                int32_t S, int32_t H, int32_t W, void* stream) {
    vpu_clear_stale_error();
    const int64_t HW = (int64_t)H * W;
    if (HW % 4 || S % 2) { vpu_set_error("p2cl: H*W % 4, S % 2"); return VPU_ERR_ARG; }
    p2cl_kernel<<<(unsigned)(B * S), 1024, 0, ST>>>(prob, gt, slot_mask_idx, override_masks, loss_part, dprob, grad_scale,
                                                   S, HW);
    return vpu_check_launch("vpu_p2cl_fwd_bwd");
}

static inline int p2cl_band(int w) {
    // VPU_P2CL_BAND: anchor rows per workgroup (A/B runs; default below)
    static const int env = [] { const char* e = vpu_lab_getenv("VPU_P2CL_BAND"); return e ? atoi(e) : 0; }();
    int b = 1024 / (w > 0 ? w : 1) - 1;
    const int cap = env > 0 && env <= P2_BAND_MAX ? env : P2_BAND_MAX;
    return b > cap ? cap : b;
}
#ifdef VPU_P2_STAMPS
extern "C" int vpu_dbg_p2(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_p2_dbg), sizeof(unsigned) * 8192 * 8, 0, hipMemcpyDeviceToHost); }
#endif
extern "C" int vpu_p2cl_up_nband(int32_t h, int32_t w) {
    const int band = p2cl_band(w);
    return band < 1 ? 0 : (h + band - 1) / band;
}

extern "C" int vpu_p2cl_up_fwd_bwd(const float* sim_low, const float* gt, const int32_t* slot_mask_idx,
                                   const float* override_masks, float* loss_part, float* dsim_low, float grad_scale,
                                   int32_t B, int32_t S, int32_t h, int32_t w, int32_t H, int32_t W, void* stream) {
    vpu_clear_stale_error();
    const int band = p2cl_band(w);
    if (S % 2 || band < 1 || h < 2 || h > P2_W_MAX || w < 2 || W % 4 || (int64_t)(H - 1) >= (int64_t)P2_SPAN * (h - 1) ||
        (int64_t)(W - 1) >= (int64_t)P2_SPAN * (w - 1)) {
        vpu_set_error("p2cl_up: S % 2, W % 4, 2 <= h, w <= 512, upsampling factor (H-1)/(h-1) < 5");
        return VPU_ERR_ARG;
    }
    const int nband = (h + band - 1) / band;
    // pixel rows one block can own: (band + 1) anchor rows x (H-1)/(h-1) rows per anchor row, + 2 for rounding
    const int max_rows = (int)(((int64_t)(band + 1) * (H - 1)) / (h - 1)) + 2;
    const size_t shmem = ((size_t)(2 * band + 2) * w + (size_t)(band + 2 + max_rows) * W) * sizeof(float);
    if (shmem > 160 * 1024 - 6144 || (int64_t)(band + 1) * (W / 4) > 1024) {
        vpu_set_error("p2cl_up: band does not fit LDS / one block ((band + 1) * W/4 <= 1024)");
        return VPU_ERR_ARG;
    }
    static VpuDevOnce attr_set;
    if (auto todo_ = attr_set.pending()) {
        VPU_SET_LDS(160 * 1024 - 6144, p2cl_up_kernel);   // (static: reduction scratch + the anchor tables)
    }
    // (as many threads as the pixel / cell passes use: (band + 1) * W / 4, a multiple of 64)
    const int nthr = (int)((((int64_t)(band + 1) * (W / 4 > w ? W / 4 : w)) + 63) / 64 * 64);
    // persistent: one workgroup per CU walks nitems / CUs consecutive (plane, band) items
    const int nitems = B * S * nband;
    const int ncu = vpu_cu_budget();
    const int grid = nitems < ncu ? nitems : ncu;
    p2cl_up_kernel<<<(unsigned)grid, nthr < 1024 ? nthr : 1024, shmem, ST>>>(sim_low, gt, slot_mask_idx, override_masks, loss_part,
                                                                  dsim_low, grad_scale, S, h, w, H, W, nband, max_rows, band, nitems);
    return vpu_check_launch("vpu_p2cl_up_fwd_bwd");
}

extern "C" int vpu_loss_finalize(const float* out, const float* part, int32_t B, int32_t npart, double inv_count,
                                 float w_nfl, float w_dice, float w_pcl, float iter_weight, float* res, void* stream) {
    vpu_clear_stale_error();
    if (!out || !part || !res || B < 1 || npart < 1) { vpu_set_error("loss_finalize: null pointer or empty input"); return VPU_ERR_ARG; }
    loss_finalize_kernel<<<1, 256, 0, ST>>>(out, part, B, npart, inv_count, w_nfl, w_dice, w_pcl, iter_weight, res);
    return vpu_check_launch("vpu_loss_finalize");
}

extern "C" int vpu_nfl_dice_scratch_doubles(int32_t B) { return B * (NFL_NBLK + 1) * 8; }

extern "C" int vpu_nfl_dice_fwd_bwd(const float* logits, const float* gt, double* sums, float* out, float* dlogits,
                                    float w_nfl, float w_dice, int32_t B, int64_t HW, void* stream) {
    vpu_clear_stale_error();
    if (HW % 4) { vpu_set_error("nfl_dice: H*W % 4"); return VPU_ERR_ARG; }
    dim3 grid(NFL_NBLK, B);
    nfl_dice_sums_kernel<<<grid, 256, 0, ST>>>(logits, gt, sums, HW);
    nfl_dice_grad_kernel<<<grid, 256, 0, ST>>>(logits, gt, sums, dlogits, w_nfl, w_dice, HW);
    nfl_dice_final_kernel<<<(B + 63) / 64, 64, 0, ST>>>(sums, out, B);
    return vpu_check_launch("vpu_nfl_dice_fwd_bwd");
}
