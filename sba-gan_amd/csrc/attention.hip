// Word-level spatial attention of the generator (GlobalAttentionGeneral.forward,
// GlobalAttention.py:82-121) as one fused kernel per direction.
//
// One thread owns one query pixel: its idf-channel vector is read once (16-byte
// loads, NHWC), the projected word keys src[idf][L] (<= 12.8 KB) are staged in
// LDS, the L <= 32 scores live in registers, softmax is in-register, and the
// context vector is written once.  No B x Q x L tensor is materialised unless the
// caller asks for the attention map.  HBM-bound: 2 * idf * sizeof(T) bytes / query.
//
// Mask quirk (GlobalAttention.py:105-108): the reference masks row r = b*Q + q of
// the (B*Q) x L score matrix with mask.repeat(Q, 1)[r] = mask[r % B], not mask[b].
// mask_mode 0 reproduces that bit-for-bit; mask_mode 1 is the per-sample mask.
#include "common.h"

namespace {

constexpr int LMAX = 32;

template <typename T, int IDF>
__global__ __launch_bounds__(256) void word_attn_fwd_kernel(
    const T* __restrict__ h, const float* __restrict__ src, const uint8_t* __restrict__ mask,
    T* __restrict__ ctx, float* __restrict__ att, int B, int Q, int L, int mask_mode, int ocs, int oco) {
    constexpr int V = Vec16<T>::N;
    __shared__ float s_src[IDF * LMAX];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < IDF * LMAX; i += blockDim.x) {
        const int c = i / LMAX, l = i - c * LMAX;      // columns >= L are zero (never NaN * 0)
        s_src[i] = l < L ? src[((int64_t)b * IDF + c) * L + l] : 0.f;
    }
    __syncthreads();
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const int64_t r = (int64_t)b * Q + q;
    const int mrow = mask_mode == 0 ? (int)(r % B) : b;

    float s[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) s[l] = 0.f;
    const T* hp = h + r * IDF;
#pragma unroll
    for (int cv = 0; cv < IDF / V; ++cv) {
        Vec16<T> hv = ld16(hp + cv * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float hh = hv.get(k);
            const float* sr = &s_src[(cv * V + k) * LMAX];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) s[l] += hh * sr[l];
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        const bool dead = l >= L || (mask && mask[mrow * L + l]);
        s[l] = dead ? -INFINITY : s[l];
        mx = fmaxf(mx, s[l]);
    }
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        s[l] = __expf(s[l] - mx);     // exp(-inf) == 0 for masked / padded words
        sum += s[l];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) s[l] *= inv;
    if (att) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
            if (l < L) att[((int64_t)b * L + l) * Q + q] = s[l];
    }
    T* op = ctx + r * ocs + oco;
#pragma unroll
    for (int cv = 0; cv < IDF / V; ++cv) {
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float* sr = &s_src[(cv * V + k) * LMAX];
            float acc = 0.f;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) acc += s[l] * sr[l];
            o.set(k, acc);
        }
        st16(op + cv * V, o);
    }
}

// backward: recompute the attention row, then
//   dA[l]  = sum_c dctx[c] src[c][l];  dS[l] = a[l] (dA[l] - sum_l' a dA)
//   dh[c]  = sum_l dS[l] src[c][l]
//   dsrc[c][l] += sum_q (h[q][c] dS[q][l] + dctx[q][c] a[q][l])     (block reduce through LDS)
template <typename T, int IDF>
__global__ __launch_bounds__(128) void word_attn_bwd_kernel(
    const T* __restrict__ h, const float* __restrict__ src, const uint8_t* __restrict__ mask,
    const T* __restrict__ dctx, T* __restrict__ dh, float* __restrict__ dsrc, int B, int Q, int L,
    int mask_mode, int dcs, int dco, int accumulate) {
    constexpr int V = Vec16<T>::N;
    constexpr int NT = 128;
    constexpr int HS = IDF + 1;
    extern __shared__ float sm[];
    float* s_src = sm;                         // [IDF][LMAX]
    float* s_h = s_src + IDF * LMAX;           // [NT][HS]
    float* s_dc = s_h + NT * HS;               // [NT][HS]
    float* s_ds = s_dc + NT * HS;              // [NT][LMAX+1]
    float* s_a = s_ds + NT * (LMAX + 1);       // [NT][LMAX+1]
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < IDF * LMAX; i += NT) {
        const int c = i / LMAX, l = i - c * LMAX;
        s_src[i] = l < L ? src[((int64_t)b * IDF + c) * L + l] : 0.f;
    }
    __syncthreads();
    const int q = blockIdx.x * NT + tid;
    const bool live = q < Q;
    const int64_t r = (int64_t)b * Q + (live ? q : 0);
    const int mrow = mask_mode == 0 ? (int)(r % B) : b;

    float s[LMAX], dA[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) { s[l] = 0.f; dA[l] = 0.f; }
    const T* hp = h + r * IDF;
    const T* dp = dctx + r * dcs + dco;
#pragma unroll
    for (int cv = 0; cv < IDF / V; ++cv) {
        Vec16<T> hv = ld16(hp + cv * V), dv = ld16(dp + cv * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float hh = live ? hv.get(k) : 0.f, dd = live ? dv.get(k) : 0.f;
            s_h[tid * HS + cv * V + k] = hh;
            s_dc[tid * HS + cv * V + k] = dd;
            const float* sr = &s_src[(cv * V + k) * LMAX];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { s[l] += hh * sr[l]; dA[l] += dd * sr[l]; }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        const bool dead = l >= L || (mask && mask[mrow * L + l]);
        s[l] = dead ? -INFINITY : s[l];
        mx = fmaxf(mx, s[l]);
    }
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) { s[l] = __expf(s[l] - mx); sum += s[l]; }
    const float inv = 1.f / sum;
    float dot = 0.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) { s[l] *= inv; dot += s[l] * dA[l]; }
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        const float ds = s[l] * (dA[l] - dot);
        dA[l] = ds;                                   // dA now holds dS
        s_ds[tid * (LMAX + 1) + l] = live ? ds : 0.f;
        s_a[tid * (LMAX + 1) + l] = live ? s[l] : 0.f;
    }
    if (live) {
        T* op = dh + r * IDF;
#pragma unroll
        for (int cv = 0; cv < IDF / V; ++cv) {
            Vec16<T> o;
            if (accumulate) o = ld16(op + cv * V);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float* sr = &s_src[(cv * V + k) * LMAX];
                float acc = 0.f;
#pragma unroll
                for (int l = 0; l < LMAX; ++l) acc += dA[l] * sr[l];
                if (accumulate) acc += o.get(k);
                o.set(k, acc);
            }
            st16(op + cv * V, o);
        }
    }
    __syncthreads();
    for (int o = tid; o < IDF * L; o += NT) {
        const int c = o / L, l = o - c * L;
        float acc = 0.f;
        for (int t = 0; t < NT; ++t)
            acc += s_h[t * HS + c] * s_ds[t * (LMAX + 1) + l] + s_dc[t * HS + c] * s_a[t * (LMAX + 1) + l];
        atomicAdd(&dsrc[((int64_t)b * IDF + c) * L + l], acc);
    }
}

template <typename T, int IDF>
int launch_fwd(const void* h, const float* src, const uint8_t* mask, void* ctx, float* att, int B, int Q, int L,
               int mode, int ocs, int oco, hipStream_t st) {
    dim3 grid(cdiv(Q, 256), B);
    hipLaunchKernelGGL((word_attn_fwd_kernel<T, IDF>), grid, dim3(256), 0, st, (const T*)h, src, mask, (T*)ctx, att,
                       B, Q, L, mode, ocs, oco);
    return SBA_CHECK_LAUNCH();
}

template <typename T, int IDF>
int launch_bwd(const void* h, const float* src, const uint8_t* mask, const void* dctx, void* dh, float* dsrc, int B,
               int Q, int L, int mode, int dcs, int dco, int acc, hipStream_t st) {
    dim3 grid(cdiv(Q, 128), B);
    const size_t sh = sizeof(float) * (IDF * LMAX + 2 * 128 * (IDF + 1) + 2 * 128 * (LMAX + 1));
    if (sh > 160 * 1024) return SBA_E_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)word_attn_bwd_kernel<T, IDF>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        attr_set = true;
    }
    hipLaunchKernelGGL((word_attn_bwd_kernel<T, IDF>), grid, dim3(128), sh, st, (const T*)h, src, mask,
                       (const T*)dctx, (T*)dh, dsrc, B, Q, L, mode, dcs, dco, acc);
    return SBA_CHECK_LAUNCH();
}

}  // namespace

#define IDF_SWITCH(idf, CALL)                               \
    switch (idf) {                                          \
        case 32: { constexpr int IDF = 32; CALL; } break;   \
        case 64: { constexpr int IDF = 64; CALL; } break;   \
        case 128: { constexpr int IDF = 128; CALL; } break; \
        default: return SBA_E_ARG;                          \
    }

extern "C" int sba_word_attn_fwd(int dtype, const void* h, const float* src, const uint8_t* mask, void* ctx,
                                 float* att, int B, int Q, int idf, int L, int mask_mode, int out_cstride,
                                 int out_coff, void* stream) {
    if (!h || !src || !ctx || B <= 0 || Q <= 0 || L <= 0 || L > LMAX || B > 65535) return SBA_E_ARG;
    if (mask_mode != 0 && mask_mode != 1) return SBA_E_ARG;
    const int V = dtype == SBA_BF16 ? 8 : 4;
    if (out_cstride < idf + out_coff || out_cstride % V || out_coff % V) return SBA_E_ARG;
    SBA_DISPATCH(dtype, IDF_SWITCH(idf, return (launch_fwd<T, IDF>(h, src, mask, ctx, att, B, Q, L, mask_mode,
                                                                     out_cstride, out_coff, (hipStream_t)stream))));
    return SBA_E_ARG;
}

extern "C" int sba_word_attn_bwd(int dtype, const void* h, const float* src, const uint8_t* mask, const void* dctx,
                                 void* dh, float* dsrc, int B, int Q, int idf, int L, int mask_mode,
                                 int dctx_cstride, int dctx_coff, int accumulate, void* stream) {
    if (!h || !src || !dctx || !dh || !dsrc || B <= 0 || Q <= 0 || L <= 0 || L > LMAX || B > 65535)
        return SBA_E_ARG;
    if (mask_mode != 0 && mask_mode != 1) return SBA_E_ARG;
    const int V = dtype == SBA_BF16 ? 8 : 4;
    if (dctx_cstride < idf + dctx_coff || dctx_cstride % V || dctx_coff % V) return SBA_E_ARG;
    SBA_DISPATCH(dtype, IDF_SWITCH(idf, return (launch_bwd<T, IDF>(h, src, mask, dctx, dh, dsrc, B, Q, L, mask_mode,
                                                                     dctx_cstride, dctx_coff, accumulate,
                                                                     (hipStream_t)stream))));
    return SBA_E_ARG;
}
