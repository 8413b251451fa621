// Fused Adam over the flat fp32 parameter buffer (torch.optim.Adam semantics, no amsgrad; configured at
// models/iSegNet/vpu_base448_cocolvis.py:149-154), writing the bf16 shadow used by the MFMA GEMMs in the same pass.
// HBM-bound: 16 B reads of p,g,m,v and writes of p,m,v (+2 B shadow) per element.
#include "vpu_common.h"
#include <stdlib.h>
#include "../../include/vpu_hip.h"

namespace {
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16_t* __restrict__ shadow, int64_t n4, int64_t n, float lr_in, float b1,
                                                   float b2, float eps, float wd_in, float bc1_in, float bc2s_in, float gs_in,
                                                   const int64_t* __restrict__ seg_end, const float* __restrict__ seg_lr,
                                                   const float* __restrict__ seg_wd, int nseg, int decoupled,
                                                   const float* __restrict__ hyper) {
    float bc1 = bc1_in, bc2s = bc2s_in, gs = gs_in, lr_scalar = lr_in;
    if (hyper) {   // step-dependent scalars from device memory: the launch can be replayed from a hipGraph
        lr_scalar = hyper[0]; bc1 = hyper[1]; bc2s = hyper[2]; gs = hyper[3];
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float pv[4], gv[4], mv[4], vv[4];
        const int64_t base = i * 4;
        // per-tensor learning rate / weight decay (layer-wise decay, lr_mult): the segment that holds element `base`
        // (tensors start on 8-element boundaries, so four consecutive elements never straddle two of them)
        float lr = lr_scalar, wd = wd_in;
        if (nseg > 0) {
            int lo = 0, hi = nseg - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (seg_end[mid] > base) hi = mid; else lo = mid + 1;
            }
            lr = hyper ? lr_scalar * seg_lr[lo] : seg_lr[lo]; wd = seg_wd[lo];
        }
        const bool full = base + 4 <= n;
        if (full) {
            const float4 a = *reinterpret_cast<const float4*>(p + base), b = *reinterpret_cast<const float4*>(g + base);
            const float4 c = *reinterpret_cast<const float4*>(m + base), d = *reinterpret_cast<const float4*>(v + base);
            pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
            mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
        } else {
            for (int j = 0; j < 4; ++j) {
                const bool ok = base + j < n;
                pv[j] = ok ? p[base + j] : 0.f; gv[j] = ok ? g[base + j] : 0.f;
                mv[j] = ok ? m[base + j] : 0.f; vv[j] = ok ? v[base + j] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gg = gv[j] * gs;
            if (wd != 0.f) {
                if (decoupled) pv[j] *= 1.f - lr * wd;     // AdamW
                else gg += wd * pv[j];                     // Adam with L2 regularisation
            }
            mv[j] = b1 * mv[j] + (1.f - b1) * gg;
            vv[j] = b2 * vv[j] + (1.f - b2) * gg * gg;
            const float denom = sqrtf(vv[j]) / bc2s + eps;
            pv[j] -= (lr / bc1) * (mv[j] / denom);
        }
        if (full) {
            *reinterpret_cast<float4*>(p + base) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            *reinterpret_cast<float4*>(m + base) = make_float4(mv[0], mv[1], mv[2], mv[3]);
            *reinterpret_cast<float4*>(v + base) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            if (shadow) {   // one 8-byte store (tensors start on 8-element boundaries: base * 2 bytes is 8-byte aligned)
                typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                bf16x4_t sv;
#pragma unroll
                for (int j = 0; j < 4; ++j) sv[j] = (bf16_t)pv[j];
                *reinterpret_cast<bf16x4_t*>(shadow + base) = sv;
            }
        } else {
            for (int j = 0; j < 4; ++j)
                if (base + j < n) {
                    p[base + j] = pv[j]; m[base + j] = mv[j]; v[base + j] = vv[j];
                    if (shadow) shadow[base + j] = (bf16_t)pv[j];
                }
        }
    }
}
// one 16-byte group per thread and no grid-stride loop: 8192 workgroups walking the 123 M parameters took 851 us per launch,
// 32768 took 731, one group per thread (120 k workgroups) 671 (tools/run_stats.sh, VPU_ADAM_GRID)
inline int adam_grid_cap() {
    static const int v = [] { const char* e = vpu_lab_getenv("VPU_ADAM_GRID"); return e ? atoi(e) : (1 << 22); }();
    return v;
}
}  // namespace

extern "C" int vpu_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                             void* stream) {
    vpu_clear_stale_error();
    if (n <= 0 || step < 1) { vpu_set_error("adam: n > 0, step >= 1"); return VPU_ERR_ARG; }
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    const int64_t n4 = (n + 3) / 4;
    adam_kernel<<<vpu_grid_for(n4, 256, adam_grid_cap()), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
        p, g, m, v, (bf16_t*)shadow_bf16, n4, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, nullptr,
        nullptr, nullptr, 0, 0, nullptr);
    return vpu_check_launch("vpu_adam_step");
}

extern "C" int vpu_adam_step_groups(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                                    const int64_t* seg_end, const float* seg_lr, const float* seg_wd, int32_t nseg,
                                    float beta1, float beta2, float eps, int32_t decoupled_wd, int32_t step,
                                    float grad_scale, void* stream) {
    vpu_clear_stale_error();
    if (n <= 0 || step < 1 || nseg < 1 || !seg_end || !seg_lr || !seg_wd) {
        vpu_set_error("adam_groups: n > 0, step >= 1, nseg >= 1, segment tables");
        return VPU_ERR_ARG;
    }
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    const int64_t n4 = (n + 3) / 4;
    adam_kernel<<<vpu_grid_for(n4, 256, adam_grid_cap()), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
        p, g, m, v, (bf16_t*)shadow_bf16, n4, n, 0.f, beta1, beta2, eps, 0.f, bc1, bc2s, grad_scale, seg_end, seg_lr,
        seg_wd, nseg, decoupled_wd, nullptr);
    return vpu_check_launch("vpu_adam_step_groups");
}

extern "C" int vpu_adam_step_hyper(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                                   const float* hyper, const int64_t* seg_end, const float* seg_scale,
                                   const float* seg_wd, int32_t nseg, float beta1, float beta2, float eps,
                                   float weight_decay, int32_t decoupled_wd, void* stream) {
    vpu_clear_stale_error();
    if (n <= 0 || !hyper || (nseg > 0 && (!seg_end || !seg_scale || !seg_wd))) {
        vpu_set_error("adam_hyper: n > 0, hyper, segment tables when nseg > 0");
        return VPU_ERR_ARG;
    }
    const int64_t n4 = (n + 3) / 4;
    adam_kernel<<<vpu_grid_for(n4, 256, adam_grid_cap()), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
        p, g, m, v, (bf16_t*)shadow_bf16, n4, n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, 1.f, seg_end, seg_scale,
        seg_wd, nseg > 0 ? nseg : 0, decoupled_wd, hyper);
    return vpu_check_launch("vpu_adam_step_hyper");
}
