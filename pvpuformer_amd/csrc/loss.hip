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

extern "C" int vpu_nfl_dice_fwd_bwd(const float* logits, const float* gt, double* sums, float* out, float* dlogits,
                                    float w_nfl, float w_dice, int32_t B, int64_t HW, void* stream) {
    vpu_clear_stale_error();
    nfl_dice_sums_kernel<<<B, 1024, 0, ST>>>(logits, gt, sums, HW);
    nfl_dice_grad_kernel<<<B, 1024, 0, ST>>>(logits, gt, sums, out, dlogits, w_nfl, w_dice, HW);
    return vpu_check_launch("vpu_nfl_dice_fwd_bwd");
}
