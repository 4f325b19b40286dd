// BertEncoder forward on the device (model_bert.py:161-189; SURVEY.md 8f-2, second half): the frozen BERT-base
// trunk of the `bert` / `mix` variants' text side.  The dense layers (QKV, attention output, the two feed-forward
// layers, the 1x1 word projection) are M = B*L = 400-row GEMMs and run on the implicit-GEMM kernels as 1x1
// convolutions with a bias epilogue (sba_conv_igemm_bias); this file holds what sits between them:
//   bert_embed_ln     word + position + token-type embedding gather, LayerNorm (eps 1e-12)
//   bert_attention    per (caption, head): softmax(q k^T / sqrt(64)) v for L <= 32 tokens, NO attention mask
//                     (the reference passes none, model_bert.py:181), q / k / v staged in LDS as f32
//   bert_add_ln       LayerNorm(x + residual)
//   bert_gelu         exact (erf) GELU, in place
//   bert_tanh_t       tanh + transpose [B*L][nef] -> [B][nef][L] f32 (words_embs layout of the generator)
// Activations are the compute dtype T (bf16 / f32), statistics and softmax in f32.
#include <math.h>

#include "common.h"

namespace {

// one wave per row of C = 768 values (12 per lane)
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_ln_kernel(const int64_t* __restrict__ tok, const float* __restrict__ we,
                                                            const float* __restrict__ pe, const float* __restrict__ te,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ out, int rows, int L, int C, int ntoken,
                                                            float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    int64_t t = tok[row];
    if (t < 0 || t >= ntoken) t = 0;
    const int pos = row % L;
    float v[16];
    float s = 0.f;
    const int per = C / 64;
    for (int i = 0; i < per; ++i) {
        const int c = lane + 64 * i;
        v[i] = we[t * C + c] + pe[(int64_t)pos * C + c] + te[c];
        s += v[i];
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
    for (int i = 0; i < per; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
    for (int i = 0; i < per; ++i) {
        const int c = lane + 64 * i;
        out[(int64_t)row * C + c] = from_f<T>((v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bert_add_ln_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          T* __restrict__ out, int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[16];
    float s = 0.f;
    const int per = C / 64;
    for (int i = 0; i < per; ++i) {
        const int64_t o = (int64_t)row * C + lane + 64 * i;
        v[i] = to_f<T>(x[o]) + to_f<T>(res[o]);
        s += v[i];
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
    for (int i = 0; i < per; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
    for (int i = 0; i < per; ++i) {
        const int c = lane + 64 * i;
        out[(int64_t)row * C + c] = from_f<T>((v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
}

// grid (B, heads), block 256: qkv [B*L][3*C] (q | k | v), ctx [B*L][C]; head dim 64, L <= 32
template <typename T>
__global__ __launch_bounds__(256) void bert_attention_kernel(const T* __restrict__ qkv, T* __restrict__ ctx, int L, int C) {
    constexpr int D = 64, LM = 32;
    __shared__ float sq[LM][D + 1], sk[LM][D + 1], sv[LM][D + 1], sp[LM][LM + 1];
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < L * D; i += 256) {
        const int t = i / D, d = i - t * D;
        const int64_t o = ((int64_t)b * L + t) * 3 * C + h * D + d;
        sq[t][d] = to_f<T>(qkv[o]);
        sk[t][d] = to_f<T>(qkv[o + C]);
        sv[t][d] = to_f<T>(qkv[o + 2 * C]);
    }
    __syncthreads();
    for (int i = tid; i < L * L; i += 256) {
        const int a = i / L, c = i - a * L;
        float s = 0.f;
#pragma unroll 16
        for (int d = 0; d < D; ++d) s += sq[a][d] * sk[c][d];
        sp[a][c] = s * 0.125f;                     // 1 / sqrt(64)
    }
    __syncthreads();
    if (tid < L) {
        float m = -INFINITY;
        for (int c = 0; c < L; ++c) m = fmaxf(m, sp[tid][c]);
        float z = 0.f;
        for (int c = 0; c < L; ++c) { const float e = expf(sp[tid][c] - m); sp[tid][c] = e; z += e; }
        const float iz = 1.f / z;
        for (int c = 0; c < L; ++c) sp[tid][c] *= iz;
    }
    __syncthreads();
    for (int i = tid; i < L * D; i += 256) {
        const int t = i / D, d = i - t * D;
        float s = 0.f;
        for (int c = 0; c < L; ++c) s += sp[t][c] * sv[c][d];
        ctx[((int64_t)b * L + t) * C + h * D + d] = from_f<T>(s);
    }
}

template <typename T>
__global__ void bert_gelu_kernel(T* __restrict__ x, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = to_f<T>(x[i]);
        x[i] = from_f<T>(0.5f * v * (1.f + erff(v * 0.70710678118654752f)));
    }
}

// y[b][c][l] = tanh(x[b*L + l][c])
template <typename T>
__global__ void bert_tanh_t_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int L, int C) {
    const int64_t n = (int64_t)B * L * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        const int l = (int)(r % L), b = (int)(r / L);
        y[((int64_t)b * C + c) * L + l] = tanhf(to_f<T>(x[i]));
    }
}

}  // namespace

extern "C" int sba_bert_embed_ln(int dtype, const int64_t* tokens, const float* word_emb, const float* pos_emb,
                                 const float* type_emb, const float* gamma, const float* beta, void* out, int B, int L,
                                 int C, int ntoken, float eps, void* stream) {
    if (!tokens || !word_emb || !pos_emb || !type_emb || !gamma || !beta || !out) return SBA_E_ARG;
    if (B <= 0 || L <= 0 || C % 64 != 0 || C > 1024 || ntoken <= 0) return SBA_E_ARG;
    const int rows = B * L;
    SBA_DISPATCH(dtype, SBA_LAUNCH((bert_embed_ln_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                                   tokens, word_emb, pos_emb, type_emb, gamma, beta, (T*)out, rows, L, C, ntoken, eps));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bert_add_ln(int dtype, const void* x, const void* residual, const float* gamma, const float* beta,
                               void* out, int rows, int C, float eps, void* stream) {
    if (!x || !residual || !gamma || !beta || !out || rows <= 0 || C % 64 != 0 || C > 1024) return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((bert_add_ln_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                                   (const T*)x, (const T*)residual, gamma, beta, (T*)out, rows, C, eps));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bert_attention(int dtype, const void* qkv, void* ctx, int B, int L, int C, int heads, void* stream) {
    if (!qkv || !ctx || B <= 0 || L <= 0 || L > 32 || heads <= 0 || C != heads * 64) return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((bert_attention_kernel<T>), dim3(B, heads), dim3(256), 0, (hipStream_t)stream,
                                   (const T*)qkv, (T*)ctx, L, C));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bert_gelu(int dtype, void* x, int64_t n, void* stream) {
    if (!x || n <= 0) return SBA_E_ARG;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    SBA_DISPATCH(dtype, SBA_LAUNCH((bert_gelu_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (T*)x, n));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bert_tanh_transpose(int dtype, const void* x, float* y, int B, int L, int C, void* stream) {
    if (!x || !y || B <= 0 || L <= 0 || C <= 0) return SBA_E_ARG;
    const int64_t n = (int64_t)B * L * C;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    SBA_DISPATCH(dtype, SBA_LAUNCH((bert_tanh_t_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                   (const T*)x, y, B, L, C));
    return SBA_CHECK_LAUNCH();
}
