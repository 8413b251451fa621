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
__global__ __launch_bounds__(1024) void p2cl_up_kernel(const float* __restrict__ low, const float* __restrict__ gt,
                                                       const int* __restrict__ slot_idx,
                                                       const float* __restrict__ override_masks,
                                                       float* __restrict__ loss_part, float* __restrict__ dlow,
                                                       float grad_scale, int S, int h, int w, int H, int W) {
    extern __shared__ float sm[];          // [h*w] values, [h*w] gradients
    __shared__ double red[16];
    float* sv = sm;
    float* sg = sm + h * w;
    const int plane = blockIdx.x, b = plane / S, s = plane % S;
    const int ov = slot_idx ? slot_idx[plane] : -1;
    const float* lab = ov >= 0 ? override_masks + (int64_t)ov * H * W : gt + (int64_t)b * H * W;
    const bool invert = ov < 0 && s >= S / 2;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float fh = sh > 0.f ? 1.f / sh : 0.f, fw = sw > 0.f ? 1.f / sw : 0.f;
    for (int i = threadIdx.x; i < h * w; i += blockDim.x) { sv[i] = low[(int64_t)plane * h * w + i]; sg[i] = 0.f; }
    __syncthreads();
    double acc = 0.0;
    const int ncell = h * w;
    const int iters = (ncell + blockDim.x - 1) / blockDim.x;
    for (int it = 0; it < iters; ++it) {
        const int cell = it * blockDim.x + threadIdx.x;
        const bool live = cell < ncell;
        const int y0 = live ? cell / w : 0, x0 = live ? cell % w : 0;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        float g00 = 0.f, g01 = 0.f, g10 = 0.f, g11 = 0.f, part = 0.f;
        if (live) {
            const float v00 = sv[y0 * w + x0], v01 = sv[y0 * w + x1], v10 = sv[y1 * w + x0], v11 = sv[y1 * w + x1];
            int Ya = (int)floorf(fh * (float)y0) - 1, Yb = (int)ceilf(fh * (float)(y0 + 1)) + 1;
            int Xa = (int)floorf(fw * (float)x0) - 1, Xb = (int)ceilf(fw * (float)(x0 + 1)) + 1;
            Ya = Ya < 0 ? 0 : Ya; Xa = Xa < 0 ? 0 : Xa;
            Yb = Yb > H - 1 ? H - 1 : Yb; Xb = Xb > W - 1 ? W - 1 : Xb;
            for (int Y = Ya; Y <= Yb; ++Y) {
                if (ac_i0(Y, sh, h) != y0) continue;
                const float ly = sh * (float)Y - (float)y0, hy = 1.f - ly;
                for (int X = Xa; X <= Xb; ++X) {
                    if (ac_i0(X, sw, w) != x0) continue;
                    const float lx = sw * (float)X - (float)x0, hx = 1.f - lx;
                    const float p = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
                    float y = lab[(int64_t)Y * W + X];
                    const bool valid = y != -1.0f;
                    if (invert) y = (y != 0.f) ? 0.f : 1.f;
                    if (!valid) y = 0.f;
                    const float a = p + 1e-12f, c = 1.f - p + 1e-12f;
                    if (valid) {
                        part += -(logf(a) * y + logf(c) * (1.f - y));
                        const float g = grad_scale * (-(y / a) + (1.f - y) / c);
                        g00 += g * hy * hx; g01 += g * hy * lx; g10 += g * ly * hx; g11 += g * ly * lx;
                    }
                }
            }
        }
        acc += part;
        if (dlow) {   // four phases: one writer per LDS word in each
            // clamped edge cells (x1 == x0 or y1 == y0) would alias a neighbour's target word: fold their (zero-weight)
            // corner terms into the cell's own word and skip the write
            const bool bx = x1 != x0, by = y1 != y0;
            if (!bx) { g00 += g01; g10 += g11; }
            if (!by) { g00 += g10; if (bx) g01 += g11; }
            if (live) sg[y0 * w + x0] += g00;
            __syncthreads();
            if (live && bx) sg[y0 * w + x1] += g01;
            __syncthreads();
            if (live && by) sg[y1 * w + x0] += g10;
            __syncthreads();
            if (live && bx && by) sg[y1 * w + x1] += g11;
            __syncthreads();
        }
    }
    const double t = block_sum_d(acc, red);
    if (threadIdx.x == 0) loss_part[plane] = (float)t;
    if (dlow) {
        __syncthreads();
        for (int i = threadIdx.x; i < h * w; i += blockDim.x) dlow[(int64_t)plane * h * w + i] = sg[i];
    }
}

// sums[b][0..4] = sum w, sum beta, sum p*t, sum p, sum t
__global__ __launch_bounds__(1024) void nfl_dice_sums_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ gt, double* __restrict__ sums,
                                                             int64_t HW) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    double a[5] = {0, 0, 0, 0, 0};
    for (int64_t i = threadIdx.x; i < HW; i += 1024) {
        const float x = logits[(int64_t)b * HW + i], t = gt[(int64_t)b * HW + i];
        const float p = 1.f / (1.f + expf(-x));
        const float w = t != -1.0f ? 1.f : 0.f;
        const float pt = w > 0.f ? 1.f - fabsf(t - p) : 1.f;
        const float beta = (1.f - pt) * (1.f - pt);
        a[0] += w; a[1] += beta; a[2] += p * t; a[3] += p; a[4] += t;
    }
    for (int k = 0; k < 5; ++k) {
        const double v = block_sum_d(a[k], red);
        if (threadIdx.x == 0) sums[b * 8 + k] = v;
    }
}

__global__ __launch_bounds__(1024) void nfl_dice_grad_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ gt,
                                                             const double* __restrict__ sums, float* __restrict__ out,
                                                             float* __restrict__ dlogits, float w_nfl, float w_dice,
                                                             int64_t HW) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    const float eps = 1e-12f;
    const float sw = (float)sums[b * 8 + 0], bs = (float)sums[b * 8 + 1];
    const float A = (float)sums[b * 8 + 2], Bp = (float)sums[b * 8 + 3], Ct = (float)sums[b * 8 + 4];
    const float mult = sw / (bs + eps);          // detached (losses.py:57-59)
    const float inv_bsum = 1.f / (sw + eps);     // size_average (losses.py:80-82)
    const float deps = 1e-3f;
    const float den = Bp + Ct + deps;
    const float dice = (2.f * A + deps) / den;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < HW; i += 1024) {
        const float x = logits[(int64_t)b * HW + i], t = gt[(int64_t)b * HW + i];
        const float p = 1.f / (1.f + expf(-x));
        const float w = t != -1.0f ? 1.f : 0.f;
        const bool pos = t > 0.5f;
        const float alpha = 0.5f * w;  // alpha = 1-alpha = 0.5
        const float pt = w > 0.f ? 1.f - fabsf(t - p) : 1.f;
        const float om = 1.f - pt;
        const float beta = om * om * mult;
        const float arg = fminf(pt + eps, 1.f);
        const float lg = logf(arg);
        acc += (double)(-alpha * beta * lg * w);
        if (dlogits) {
            // d/dpt of (1-pt)^2 * log(min(pt+eps,1))
            const float dfd = -2.f * om * lg + om * om * ((pt + eps < 1.f) ? 1.f / (pt + eps) : 0.f);
            const float dptdp = (t - p) > 0.f ? 1.f : ((t - p) < 0.f ? -1.f : 0.f);
            const float dp_dx = p * (1.f - p);
            const float g_nfl = -alpha * mult * w * dfd * dptdp * dp_dx * inv_bsum;
            const float dd_dp = (2.f * t * den - (2.f * A + deps)) / (den * den);
            const float g_dice = -dd_dp * dp_dx;
            dlogits[(int64_t)b * HW + i] = w_nfl * g_nfl + w_dice * g_dice;
        }
        (void)pos;
    }
    const double L = block_sum_d(acc, red);
    if (threadIdx.x == 0) {
        out[b * 2 + 0] = (float)L * inv_bsum;
        out[b * 2 + 1] = 1.f - dice;
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

extern "C" int vpu_p2cl_up_fwd_bwd(const float* sim_low, const float* gt, const int32_t* slot_mask_idx,
                                   const float* override_masks, float* loss_part, float* dsim_low, float grad_scale,
                                   int32_t B, int32_t S, int32_t h, int32_t w, int32_t H, int32_t W, void* stream) {
    vpu_clear_stale_error();
    const size_t shmem = (size_t)2 * h * w * sizeof(float);
    if (S % 2 || shmem > 150 * 1024) { vpu_set_error("p2cl_up: S % 2, low-res plane must fit LDS twice"); return VPU_ERR_ARG; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(p2cl_up_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024);
        attr_set = true;
    }
    p2cl_up_kernel<<<(unsigned)(B * S), 1024, shmem, ST>>>(sim_low, gt, slot_mask_idx, override_masks, loss_part, dsim_low,
                                                         grad_scale, S, h, w, H, W);
    return vpu_check_launch("vpu_p2cl_up_fwd_bwd");
}

extern "C" int vpu_nfl_dice_fwd_bwd(const float* logits, const float* gt, double* sums, float* out, float* dlogits,
                                    float w_nfl, float w_dice, int32_t B, int64_t HW, void* stream) {
    vpu_clear_stale_error();
    nfl_dice_sums_kernel<<<B, 1024, 0, ST>>>(logits, gt, sums, HW);
    nfl_dice_grad_kernel<<<B, 1024, 0, ST>>>(logits, gt, sums, out, dlogits, w_nfl, w_dice, HW);
    return vpu_check_launch("vpu_nfl_dice_fwd_bwd");
}
