// Scalar losses of the adversarial step and the fused Adam + EMA update (gfx950).
//   BCE       miscc/losses.py:144-158,175-178 (nn.BCELoss on sigmoid outputs)
//   KL        miscc/losses.py:210-214
//   Adam/EMA  trainer.py:136-143 (lr 2e-4, betas (0.5, 0.999), eps 1e-8), :298-299
// Loss scalars stay on the device (the reference's .item() syncs are not reproduced).
#include "common.h"

namespace {

struct AdamState {
    int32_t step;
    float step_size;       // lr / (1 - beta1^step)
    float inv_sqrt_bc2;    // 1 / sqrt(1 - beta2^step)
    float pad;
};

__global__ void bce_multi_kernel(const float* __restrict__ prob, const int32_t* __restrict__ offsets,
                                 const float* __restrict__ target, const float* __restrict__ weight, int nseg,
                                 float* __restrict__ loss, float* __restrict__ dprob) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int s = 0; s < nseg; ++s) {
        const int lo = offsets[s], hi = offsets[s + 1];
        const float t = target[s], wn = weight[s] / (float)(hi - lo);
        for (int k = lo + threadIdx.x; k < hi; k += blockDim.x) {
            const float p = prob[k];
            // torch.nn.BCELoss: logs clamped at -100; grad (p - t) / max(p (1 - p), 1e-12)
            const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
            acc += -wn * (t * lp + (1.f - t) * lq);
            if (dprob) dprob[k] = wn * (p - t) / fmaxf(p * (1.f - p), 1e-12f);
        }
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = acc;
}

__global__ void kl_loss_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                               float* __restrict__ loss, float* __restrict__ dmu, float* __restrict__ dlogvar,
                               int n) {
    __shared__ float sh[16];
    float acc = 0.f;
    const float inv = 1.f / (float)n;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const float m = mu[k], lv = logvar[k], e = expf(lv);
        acc += 1.f + lv - m * m - e;
        if (dmu) dmu[k] = m * inv;                       // d(-0.5 mean(...))/dmu = mu / n
        if (dlogvar) dlogvar[k] = -0.5f * (1.f - e) * inv;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = -0.5f * acc * inv;
}

__global__ void adam_prepare_kernel(AdamState* st, float lr, float beta1, float beta2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int step = st->step + 1;
        st->step = step;
        const double bc1 = 1.0 - pow((double)beta1, (double)step);
        const double bc2 = 1.0 - pow((double)beta2, (double)step);
        st->step_size = (float)((double)lr / bc1);
        st->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    }
}

// NT: non-temporal loads and stores.  The update of D_NET256 streams 2 GB (28 B per parameter) through a 32 MB L2 while two
// or three other chains of the step run beside it: with ordinary accesses it evicts their working sets and the small
// latency-bound launches next to it stretch 5-10x (profiles/r04_step_inception_v1_timeline.txt: a 7 us global-average-pool
// backward at 292 us, a 36 us stem data gradient at 387 us, a 10 us 8x8 conv at 280 us while adam_step runs).  Nothing in
// the update is read twice, so every access is marked streaming.  MEASURED (same box, two runs each,
// gpurun_out/r4_ab_adam_nt*.json): 10.54 / 10.69 ms with, 10.59 / 10.67 ms without -- no difference: the neighbours are
// slowed by the HBM queues, not by L2 evictions.  Kept (harmless, SBA_ADAM_NT=0 turns it off).
template <bool NT>
__device__ __forceinline__ f32x4_t ld4(const float* a, int64_t i) {
    const f32x4_t* q = reinterpret_cast<const f32x4_t*>(a) + i;
    return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT>
__device__ __forceinline__ void st4(float* a, int64_t i, f32x4_t x) {
    f32x4_t* q = reinterpret_cast<f32x4_t*>(a) + i;
    if (NT) __builtin_nontemporal_store(x, q); else *q = x;
}

template <bool NT>
__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, float* __restrict__ avg, bf16_t* __restrict__ shadow,
                                 const AdamState* __restrict__ st, int64_t n, float beta1, float beta2, float eps,
                                 float gscale) {
    const float step_size = st->step_size, isb2 = st->inv_sqrt_bc2;
    const int64_t n4 = n / 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        f32x4_t pv = ld4<NT>(p, i);
        const f32x4_t gv = ld4<NT>(g, i);
        f32x4_t mv = ld4<NT>(m, i), vv = ld4<NT>(v, i);
        f32x4_t av;
        if (avg) av = ld4<NT>(avg, i);
        float* pp = (float*)&pv; const float* gp = (const float*)&gv;
        float* mp = (float*)&mv; float* vp = (float*)&vv; float* ap = (float*)&av;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gp[k] * gscale;
            mp[k] = beta1 * mp[k] + (1.f - beta1) * gg;
            vp[k] = beta2 * vp[k] + (1.f - beta2) * gg * gg;
            const float denom = sqrtf(vp[k]) * isb2 + eps;
            pp[k] -= step_size * mp[k] / denom;
            if (avg) ap[k] = 0.999f * ap[k] + 0.001f * pp[k];
        }
        st4<NT>(p, i, pv);
        st4<NT>(m, i, mv);
        st4<NT>(v, i, vv);
        if (avg) st4<NT>(avg, i, av);
        if (shadow) {
            ushort4 s;
            s.x = f2bf(pp[0]); s.y = f2bf(pp[1]); s.z = f2bf(pp[2]); s.w = f2bf(pp[3]);
            reinterpret_cast<ushort4*>(shadow)[i] = s;
        }
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            const float gg = g[i] * gscale;
            m[i] = beta1 * m[i] + (1.f - beta1) * gg;
            v[i] = beta2 * v[i] + (1.f - beta2) * gg * gg;
            p[i] -= step_size * m[i] / (sqrtf(v[i]) * isb2 + eps);
            if (avg) avg[i] = 0.999f * avg[i] + 0.001f * p[i];
            if (shadow) shadow[i] = f2bf(p[i]);
        }
    }
}

template <typename D, typename S>
__global__ void cast_kernel(D* __restrict__ dst, const S* __restrict__ src, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = from_f<D>(to_f<S>(src[i]));
}

}  // namespace

extern "C" const char* sba_version(void) { return "sbagan_hip 0.1 (gfx950)"; }

extern "C" int sba_bce_multi(const float* prob, const int32_t* offsets, const float* target, const float* weight,
                             int nseg, float* loss, float* dprob, void* stream) {
    if (!prob || !offsets || !target || !weight || !loss || nseg <= 0 || nseg > 64) return SBA_E_ARG;
    SBA_LAUNCH(bce_multi_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, prob, offsets, target, weight,
                       nseg, loss, dprob);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_kl_loss(const float* mu, const float* logvar, float* loss, float* dmu, float* dlogvar, int n,
                           void* stream) {
    if (!mu || !logvar || !loss || n <= 0) return SBA_E_ARG;
    SBA_LAUNCH(kl_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mu, logvar, loss, dmu, dlogvar, n);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_adam_prepare(void* state, float lr, float beta1, float beta2, void* stream) {
    if (!state) return SBA_E_ARG;
    SBA_LAUNCH(adam_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (AdamState*)state, lr, beta1,
                       beta2);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_adam_step(float* p, const float* g, float* m, float* v, float* avg, void* shadow,
                             const void* state, int64_t n, float beta1, float beta2, float eps, float grad_scale,
                             void* stream) {
    if (!p || !g || !m || !v || !state || n <= 0) return SBA_E_ARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)avg) & 15) return SBA_E_ARG;
    if ((uintptr_t)shadow & 7) return SBA_E_ARG;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    static int nt = -1;         // SBA_ADAM_NT=0: ordinary (cached) accesses (A/B aid)
    if (nt < 0) { const char* e = getenv("SBA_ADAM_NT"); nt = (e && e[0] == '0') ? 0 : 1; }
    if (nt)
        SBA_LAUNCH(adam_step_kernel<true>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, avg,
                   (bf16_t*)shadow, (const AdamState*)state, n, beta1, beta2, eps, grad_scale);
    else
        SBA_LAUNCH(adam_step_kernel<false>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, avg,
                   (bf16_t*)shadow, (const AdamState*)state, n, beta1, beta2, eps, grad_scale);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_cast(int dtype_dst, void* dst, int dtype_src, const void* src, int64_t n, void* stream) {
    if (!dst || !src || n <= 0) return SBA_E_ARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (dtype_dst == SBA_BF16 && dtype_src == SBA_F32)
        SBA_LAUNCH((cast_kernel<bf16_t, float>), dim3((int)blocks), dim3(256), 0, st, (bf16_t*)dst,
                           (const float*)src, n);
    else if (dtype_dst == SBA_F32 && dtype_src == SBA_BF16)
        SBA_LAUNCH((cast_kernel<float, bf16_t>), dim3((int)blocks), dim3(256), 0, st, (float*)dst,
                           (const bf16_t*)src, n);
    else if (dtype_dst == SBA_F32 && dtype_src == SBA_F32)
        SBA_LAUNCH((cast_kernel<float, float>), dim3((int)blocks), dim3(256), 0, st, (float*)dst,
                           (const float*)src, n);
    else
        return SBA_E_ARG;
    return SBA_CHECK_LAUNCH();
}
