// Shared device helpers for the SBA-GAN gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sbagan_hip.h"

typedef uint16_t bf16_t;   // storage type of bf16 activations / packed weights

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;    // MFMA A/B fragment (8 bf16)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

// IEEE binary16 storage of the pre-BatchNorm conv outputs (SBA_BF16_YH): same bytes as bf16, 3 more mantissa bits
typedef _Float16 f16_t;
__device__ __forceinline__ bf16_t f2h_bits(float f) {      // saturating, RNE; returned as the 16 raw bits
    f = fminf(fmaxf(f, -65504.f), 65504.f);
    const f16_t h = (f16_t)f;
    return __builtin_bit_cast(bf16_t, h);
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// VEC<T>: 16-byte vector of T (4 floats / 8 bf16) for coalesced elementwise kernels
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float4 v;
    __device__ __forceinline__ float get(int i) const { return ((const float*)&v)[i]; }
    __device__ __forceinline__ void set(int i, float f) { ((float*)&v)[i] = f; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    uint4 v;
    __device__ __forceinline__ float get(int i) const { return bf2f(((const bf16_t*)&v)[i]); }
    __device__ __forceinline__ void set(int i, float f) { ((bf16_t*)&v)[i] = f2bf(f); }
};
template <> struct Vec16<f16_t> {
    static constexpr int N = 8;
    uint4 v;
    __device__ __forceinline__ float get(int i) const { return (float)((const f16_t*)&v)[i]; }
    __device__ __forceinline__ void set(int i, float f) { ((f16_t*)&v)[i] = (f16_t)f; }
};
template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

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
// block-wide sum for blockDim.x <= 1024 threads; `sh` holds >= 16 floats; result in all threads
__device__ __forceinline__ float block_sum(float v, float* sh) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}

// (IEEE division on purpose.  `__builtin_amdgcn_rcpf(1.f + __expf(-x))` -- v_rcp_f32, 1 ulp -- makes the VALU-bound GLU
// BatchNorm backward reduce 10 % faster (63 -> 56 us at 256 px, profiles/r03_bn_passes.txt), ~20 us per step; but the
// bf16 step-0 losses sit at the 1e-3 bar as the end of a random walk of roundings, and that one-ulp
// change moved errD2 of the B = 20 golden step from 9.3e-4 to 1.6e-3.  Not worth the margin.)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
// the BACKWARD passes' recomputation of the gate (bn_bwd_reduce / apply / fused): v_rcp_f32.  Step-0 losses depend on
// forward kernels only; gradients move by one ulp of the gate (their bounds are 10x wider than the losses')
__device__ __forceinline__ float sigmoid_bwd_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// hipGetLastError() is per-thread and STICKY across unrelated runtime calls of the host framework (an
// event query that returned hipErrorNotReady is enough): clear it before every launch so that
// SBA_CHECK_LAUNCH reports this launch only.
#define SBA_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define SBA_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? SBA_OK : SBA_E_LAUNCH)
#define SBA_DISPATCH(dtype, CALL)                       \
    do {                                                \
        if ((dtype) == SBA_F32) { using T = float; CALL; }        \
        else if ((dtype) == SBA_BF16) { using T = bf16_t; CALL; } \
        else return SBA_E_ARG;                          \
    } while (0)
// T = activation storage type, YT = storage type of the raw conv output y (SBA_BF16_YH: binary16)
#define SBA_DISPATCH_Y(dtype, CALL)                     \
    do {                                                \
        if ((dtype) == SBA_F32) { using T [[maybe_unused]] = float; using YT = float; CALL; }            \
        else if ((dtype) == SBA_BF16) { using T [[maybe_unused]] = bf16_t; using YT = bf16_t; CALL; }    \
        else if ((dtype) == SBA_BF16_YH) { using T [[maybe_unused]] = bf16_t; using YT = f16_t; CALL; }  \
        else return SBA_E_ARG;                          \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Accumulators are cleared by a KERNEL, never by hipMemsetAsync: a memset node captured into a hipGraph is not
// reliably ordered before the kernel node that follows it on ROCm 7.2 (the accumulation can run first and the
// memset then wipes it; seen as instance-norm statistics of 0 / eps on replays, tools/debug_gA.py).
static __global__ void sba_zero_f32_kernel(float* __restrict__ a, float* __restrict__ b, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        a[i] = 0.f;
        if (b) b[i] = 0.f;
    }
}
static inline void sba_zero_f32(float* a, float* b, int64_t n, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(sba_zero_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, n);
}

// ---------------------------------------------------------------------------
// Deterministic-reduction mode (det.hip; sba_set_deterministic in sbagan_hip.h).  When on, no result depends on
// the order in which workgroups or waves reach an f32 atomic: every kernel that adds partial sums into a shared
// destination writes them to a private slot of the scratch ring instead (plain stores), and sba_det_fold adds the
// slots up in slot order; in-workgroup LDS accumulation is done wave by wave; split-K is off.
// ---------------------------------------------------------------------------
bool sba_det_on();
// `nfloats` f32 of scratch for ONE launch (+ its fold), 256-byte aligned; nullptr when the ring is smaller than that
float* sba_det_alloc(int64_t nfloats);
// dst[j * dst_stride + i] (+)= sum_{p < P, in order} part[(j * P + p) * n + i]   for j < J, i < n
//   mode 0: dst += sum;  mode 1: dst = sum
void sba_det_fold(const float* part, int J, int P, int64_t n, float* dst, int64_t dst_stride, int mode, hipStream_t st);
// DEFAULT mode, two-stage reduction (sba_set_reduce_scratch in sbagan_hip.h): a launch whose workgroups would all add into
// the same few hundred addresses AT ITS END (the stem / image-head weight gradients: ~800 workgroups x 3072 same-address
// f32 atomics = a 30 us tail) stores per-workgroup partial sums into this ring instead and sba_fold_add adds them to the
// destination (32 slots per thread, then one atomic).  nullptr = no ring set or request too large: fall back to atomics.
float* sba_reduce_alloc(int64_t nfloats);
// dst[i] += sum_{p < P} part[p * n + i]   (unordered: f32 atomics between groups of 32 slots)
void sba_fold_add(const float* part, int P, int64_t n, float* dst, hipStream_t st);
