// Shared device helpers for the VPUFormer gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>


// "done once per DEVICE" flag of a call site (hipFuncSetAttribute is a per-device setting; a process that moves from one
// device to another must repeat it there): one bit per device id.  Use:
//     static VpuDevOnce once;  if (auto todo = once.pending()) { VPU_SET_LDS(bytes, kernel); ... }
// The bit is committed when `todo` goes out of scope -- i.e. AFTER the attribute calls -- and only if none of them failed (a
// failing VPU_SET_LDS returns from the C-ABI call with VPU_ERR_LAUNCH: the next call tries again instead of launching with a
// dynamic-LDS request the kernel was never granted).  A thread that races the first one repeats the (idempotent)
// calls rather than launching before they have completed.
inline thread_local bool g_vpu_attr_failed = false;
struct VpuDevOnce {
    std::atomic<unsigned long long> done{0};
    struct Todo {
        VpuDevOnce* o;
        unsigned long long bit;
        explicit operator bool() const { return o != nullptr; }
        ~Todo() { if (o && !g_vpu_attr_failed) o->done.fetch_or(bit, std::memory_order_release); }
    };
    Todo pending() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return Todo{nullptr, 0};
        g_vpu_attr_failed = false;
        return Todo{this, bit};
    }
};


typedef __bf16 bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define VPU_WAVE 64

// error codes of the C ABI (include/vpu_hip.h)
#define VPU_OK 0
#define VPU_ERR_ARG (-1)
#define VPU_ERR_ALIGN (-2)
#define VPU_ERR_LAUNCH (-3)

void vpu_set_error(const char* msg);
int vpu_check_launch(const char* what);
// hipGetLastError() is per-thread and NOT cleared by successful calls: drop whatever an earlier runtime call of the
// host program left behind so that vpu_check_launch() reports this launch only.
static inline void vpu_clear_stale_error() { (void)hipGetLastError(); }
// Raises a kernel's dynamic-LDS limit (done once per device by the callers, see VpuDevOnce); a refusal ends the C-ABI call that
// asked for it with VPU_ERR_LAUNCH instead of surfacing later as a launch failure with no cause attached.
#define VPU_SET_LDS(bytes, ...)                                                                                          \
    do {                                                                                                                 \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (bytes)) != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                                                     \
            g_vpu_attr_failed = true;                                                                                    \
            vpu_set_error("hipFuncSetAttribute(" #__VA_ARGS__ ", hipFuncAttributeMaxDynamicSharedMemorySize) failed");   \
            return VPU_ERR_LAUNCH;                                                                                       \
        }                                                                                                                \
    } while (0)

// -DVPU_DIAG (csrc/build.sh diag -> libvpu_hip_diag.so, for tools/k2_stamps.py and tools/k4_drift.py): the K2 / K4P GEMM kernels
// take a device buffer and stamp the 100-MHz real-time counter at the phases of their tiles (vpu_debug_gemm_times).  The
// product library is built without it: no stamp pointer among the kernel arguments, no stamp code in the loops.
#ifdef VPU_DIAG
#define VPU_DBG_PARAM_DEF , unsigned long long* __restrict__ dbg = nullptr
#define VPU_DBG_PARAM , unsigned long long* dbg
#define VPU_DBG_PASS , dbg
#define VPU_DBG_LOAD , g_dbg_times.load(std::memory_order_relaxed)
#define VPU_STAMP(cond, idx)                                                  \
    do {                                                                      \
        if (dbg && (cond)) dbg[idx] = __builtin_amdgcn_s_memrealtime();       \
    } while (0)
#else
#define VPU_DBG_PARAM_DEF
#define VPU_DBG_PARAM
#define VPU_DBG_PASS
#define VPU_DBG_LOAD
#define VPU_STAMP(cond, idx) do { } while (0)
#endif

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x a multiple of 64 (<= 1024); every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 16 floats */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}
__device__ __forceinline__ double block_sum_d(double v, double* smem) {
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* smem) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    float r = smem[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, smem[i]);
    return r;
}

// Environment knobs.  The product library reads four (VPU_GEMM_K2, VPU_GEMM_K3, VPU_ATTN_LEAN, VPU_ATTN_ONEPASS: which of its tested
// kernel families a call takes; the same choices as vpu_gemm_set_option / vpu_attn_set_option).  Every other one is an A/B knob of
// the laboratory build (-DVPU_LAB): in the product library it reads as unset and the code takes its default.
int vpu_cu_budget();     // gemm.hip: CUs a persistent grid may claim (the "reserve_cus" option of vpu_gemm_set_option taken off)

inline const char* vpu_lab_getenv(const char* name) {
#ifdef VPU_LAB
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// GELU(x) = x Phi(x) and GELU'(x) = Phi(x) + x phi(x) through erf(|x|/sqrt 2) = 1 - (a1 t + .. + a5 t^5) exp(-x^2/2),
// t = 1 / (1 + p |x| / sqrt 2)   (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7): one v_exp, one v_rcp and ~10 FMAs for the
// pair, against ~35 instructions per erff() call.  Used where the result is rounded to bf16 (the fp32 parity path keeps erff).
__device__ __forceinline__ void gelu_pair_fast(float x, float& y, float& dy) {
    const float u = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);       // exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(x), 0.23164188892868984f, 1.0f));
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(t, poly, 1.421413741f);
    poly = __builtin_fmaf(t, poly, -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.254829592f);
    const float erfa = __builtin_fmaf(-(poly * t), u, 1.0f);       // erf(|x|/sqrt2)
    const float phi = __builtin_fmaf(0.5f, __builtin_copysignf(erfa, x), 0.5f);
    dy = __builtin_fmaf(x * 0.3989422804014327f, u, phi);
    y = x * phi;
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) {
    if constexpr (sizeof(T) == 2) { float y, d; gelu_pair_fast(x, y, d); return y; }
    else return gelu_f(x);
}
template <typename T> __device__ __forceinline__ float dgelu_t(float x) {
    if constexpr (sizeof(T) == 2) { float y, d; gelu_pair_fast(x, y, d); return d; }
    else return dgelu_f(x);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// 8 contiguous elements <-> 8 floats
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8_t a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8_t*>(p) = a;
}

// 8 contiguous elements kept RAW (unconverted) in registers: a software prefetch must not touch the loaded value before
// the iteration that consumes it, or the wave waits for the load right where it was issued.
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
    bf16x8_t r;
    __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const bf16x8_t*>(p); }
    __device__ __forceinline__ void get(float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
    }
};
template <> struct Raw8<float> {
    f32x4_t a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4_t*>(p);
        b = *reinterpret_cast<const f32x4_t*>(p + 4);
    }
    __device__ __forceinline__ void get(float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
};

static inline int vpu_grid_for(long long n_items, int per_block, int cap = 1 << 20) {
    long long g = (n_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
