// Small dense f32 layers of the generator's conditioning path (gfx950): CA_NET
// (model.py:271-299), MAPPING_NET (:301-321), INIT_STAGE_G.fc (:354), AdaIN style
// (:330) and the attention key projection conv_context (GlobalAttention.py:75,97).
// Batch is 20-64 rows, so these are weight-bandwidth bound GEMVs, not MFMA work:
// one wave streams one weight row with coalesced loads, activations sit in LDS.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int B, int K, int N) {
    extern __shared__ float s_x[];       // [B][K]
    for (int i = threadIdx.x; i < B * K; i += blockDim.x) s_x[i] = x[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int n = blockIdx.x * 4 + wid; n < N; n += gridDim.x * 4) {
        const float* wr = w + (int64_t)n * K;
        for (int b = 0; b < B; ++b) {
            float acc = 0.f;
            for (int k = lane; k < K; k += 64) acc += wr[k] * s_x[b * K + k];
            acc = wave_sum(acc);
            if (lane == 0) y[(int64_t)b * N + n] = acc + (bias ? bias[n] : 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dw, float* __restrict__ dbias, int B,
                                                           int K, int N) {
    extern __shared__ float s_x[];       // [B][K]
    for (int i = threadIdx.x; i < B * K; i += blockDim.x) s_x[i] = x[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int n = blockIdx.x * 4 + wid; n < N; n += gridDim.x * 4) {
        float sb = 0.f;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            float acc = 0.f;
            for (int b = 0; b < B; ++b) {
                const float d = dy[(int64_t)b * N + n];
                if (k < K) acc += d * s_x[b * K + k];
                if (k0 == 0) sb += d;
            }
            if (k < K) dw[(int64_t)n * K + k] += acc;
        }
        if (dbias && lane == 0) dbias[n] += sb;
    }
}

// dx[b][k] += sum over this block's n-range of dy[b][n] w[n][k]   (dx zeroed by the caller)
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int B, int K, int N, int nper) {
    const int n0 = blockIdx.x * nper, n1 = min(n0 + nper, N);
    for (int i = threadIdx.x; i < B * K; i += blockDim.x) {
        const int b = i / K, k = i - b * K;
        float acc = 0.f;
        for (int n = n0; n < n1; ++n) acc += dy[(int64_t)b * N + n] * w[(int64_t)n * K + k];
        atomicAdd(&dx[i], acc);
    }
}

// ---------------------------------------------------------------------------
// The small dense layers with a batch of <= 32 rows (MAPPING_NET, CA_NET.fc, AdaIN style) on the f32
// matrix cores: one wave per 32 x 32 output tile, both operands read straight from L2 with 16-byte loads
// (lane (r, h) of v_mfma_f32_32x32x2_f32 needs element k = 2*step + h of row r: one float4 feeds two
// steps).  As wave-per-row GEMVs with a cross-lane reduction per (row, sample) these layers took
// 19 / 31 / 25 us (forward / dx / dW) each, pure latency.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void linear_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int B, int K, int N) {
    const int lane = threadIdx.x, rl = lane & 31, hf = lane >> 5;
    const int n = blockIdx.x * 32 + rl;
    const bool nv = n < N, bv = rl < B;
    const float* xr = x + (int64_t)(bv ? rl : 0) * K;
    const float* wr = w + (int64_t)(nv ? n : 0) * K;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int K4 = K / 4;
    for (int j0 = 0; j0 < K4; j0 += 8) {
        float4 xv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = j0 + u < K4;
            xv[u] = (ok && bv) ? *reinterpret_cast<const float4*>(xr + 4 * (j0 + u)) : make_float4(0.f, 0.f, 0.f, 0.f);
            wv[u] = (ok && nv) ? *reinterpret_cast<const float4*>(wr + 4 * (j0 + u)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hf ? xv[u].y : xv[u].x, hf ? wv[u].y : wv[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hf ? xv[u].w : xv[u].z, hf ? wv[u].w : wv[u].z, acc, 0, 0, 0);
        }
    }
    if (nv) {
        const float bn = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (b < B) y[(int64_t)b * N + n] = acc[r] + bn;
        }
    }
}

// dx[b][k] = sum_n dy[b][n] w[n][k]: tile = 32 input features, reduction over N
// blockIdx.y = slice of `nper` output features of the reduction (atomic accumulation when sliced)
__device__ __forceinline__ void linear_bwd_x_mfma_body(const float* __restrict__ w, const float* __restrict__ dy,
                                                       float* __restrict__ dx, int B, int K, int N, int nper,
                                                       const int bx, const int by, const int ny) {
    const int lane = threadIdx.x, rl = lane & 31, hf = lane >> 5;
    const int k = bx * 32 + rl;
    const bool kv = k < K, bv = rl < B;
    const float* dr = dy + (int64_t)(bv ? rl : 0) * N;
    const float* wc = w + (kv ? k : 0);
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int n_lo = by * nper, n_hi = min(n_lo + nper, N);
    for (int n0 = n_lo; n0 < n_hi; n0 += 32) {
        float4 dv[8];
        float wv[16];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            dv[u] = (bv && n0 + 4 * u < n_hi) ? *reinterpret_cast<const float4*>(dr + n0 + 4 * u)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int n = n0 + 2 * kk + hf;
            wv[kk] = (kv && n < n_hi) ? wc[(int64_t)n * K] : 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float4 d = dv[kk >> 1];
            const float a = (kk & 1) ? (hf ? d.w : d.z) : (hf ? d.y : d.x);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wv[kk], acc, 0, 0, 0);
        }
    }
    if (kv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (b < B) {
                if (ny > 1) atomicAdd(&dx[(int64_t)b * K + k], acc[r]);
                else dx[(int64_t)b * K + k] = acc[r];
            }
        }
    }
}

__global__ __launch_bounds__(64) void linear_bwd_x_mfma_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                               float* __restrict__ dx, int B, int K, int N, int nper) {
    linear_bwd_x_mfma_body(w, dy, dx, B, K, N, nper, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// dw[n][k] += sum_b dy[b][n] x[b][k]; dbias[n] += sum_b dy[b][n].  grid = (K tiles, N tiles)
__device__ __forceinline__ void linear_bwd_w_mfma_body(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dw, float* __restrict__ dbias,
                                                       int B, int K, int N, const int bx, const int by) {
    const int lane = threadIdx.x, rl = lane & 31, hf = lane >> 5;
    const int k = bx * 32 + rl, n = by * 32 + rl;
    const bool kv = k < K, nv = n < N;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float av[16], bvv[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int b = 2 * kk + hf;
        av[kk] = (nv && b < B) ? dy[(int64_t)b * N + n] : 0.f;
        bvv[kk] = (kv && b < B) ? x[(int64_t)b * K + k] : 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bvv[kk], acc, 0, 0, 0);
    if (kv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = by * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (nn < N) dw[(int64_t)nn * K + k] += acc[r];
        }
    }
    if (dbias && bx == 0) {
        float sb = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) sb += av[kk];
        sb += __shfl_xor(sb, 32, 64);
        if (hf == 0 && nv) dbias[n] += sb;
    }
}

__global__ __launch_bounds__(64) void linear_bwd_w_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dw, float* __restrict__ dbias,
                                                               int B, int K, int N) {
    linear_bwd_w_mfma_body(x, dy, dw, dbias, B, K, N, (int)blockIdx.x, (int)blockIdx.y);
}

// dx AND dW of one small layer in ONE launch: workgroups [0, kt) are the dx tiles (one slice: no atomics), the rest the
// (K tile, N tile) grid of dW -- they read the same three tensors and write disjoint outputs.  The mapping network's backward
// pass is a chain of 6 / 8 such layers at the very end of the step (two ~15 us launches each, pure latency).
__global__ __launch_bounds__(64) void linear_bwd_xw_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ dy, float* __restrict__ dx,
                                                                float* __restrict__ dw, float* __restrict__ dbias,
                                                                int B, int K, int N, int kt) {
    const int b = (int)blockIdx.x;
    if (b < kt) linear_bwd_x_mfma_body(w, dy, dx, B, K, N, N, b, 0, 1);
    else linear_bwd_w_mfma_body(x, dy, dw, dbias, B, K, N, (b - kt) % kt, (b - kt) / kt);
}

// attention key projection (conv_context, GlobalAttention.py:75,97) on the f32 matrix cores.
// forward: src[b][i][l] = sum_c W[i][c] words[b][c][l]: rows = (b, l) pairs, 32 per wave; columns = i.
__global__ __launch_bounds__(64) void ctx_proj_fwd_mfma_kernel(const float* __restrict__ words, const float* __restrict__ W,
                                                               float* __restrict__ src, int B, int idf, int cdf, int L) {
    const int lane = threadIdx.x, rl = lane & 31, hf = lane >> 5;
    const int row = blockIdx.x * 32 + rl;                 // (b, l)
    const bool rv = row < B * L;
    const int b = rv ? row / L : 0, l = rv ? row - b * L : 0;
    const int i = blockIdx.y * 32 + rl;
    const bool iv = i < idf;
    const float* wp = words + ((int64_t)b * cdf + hf) * L + l;      // + 2*kk*L per step
    const float* Wr = W + (int64_t)(iv ? i : 0) * cdf;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int c0 = 0; c0 < cdf; c0 += 32) {                 // 16 steps of 2 channels
        float av[16];
        float4 wv[8];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) av[kk] = (rv && c0 + 2 * kk + hf < cdf) ? wp[(int64_t)(c0 + 2 * kk) * L] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            wv[u] = (iv && c0 + 4 * u < cdf) ? *reinterpret_cast<const float4*>(Wr + c0 + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float4 q = wv[kk >> 1];
            const float bq = (kk & 1) ? (hf ? q.w : q.z) : (hf ? q.y : q.x);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bq, acc, 0, 0, 0);
        }
    }
    if (iv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (rr < B * L) {
                const int bb = rr / L, ll = rr - bb * L;
                src[((int64_t)bb * idf + i) * L + ll] = acc[r];
            }
        }
    }
}

// The same projection with FP8 operands (BASELINE config 5: "fp8 MFMA for the attention / context GEMM"):
// v_mfma_f32_32x32x16_fp8_fp8 (OCP e4m3 on gfx950), f32 accumulate.  Block-scaled: every 32-row tile of
// (b, l) positions and every 32-row tile of W is scaled by the largest power of two <= 448 / amax(tile) before the conversion
// (v_cvt_pk_fp8_f32) and the product is scaled back in the epilogue, so the 3-bit mantissa is spent on the tile's
// own range.  Forward only: the backward keeps the f32 operands (straight-through w.r.t. the quantisation).
// Stated tolerance: relative L2 error of the projected keys <= 8e-2 on N(0,1) data (e4m3: 2^-4 relative per
// element, ~5e-2 on a 256-term dot product of two quantised operands); exact on data that e4m3 represents.
__device__ __forceinline__ long pack8_fp8(const float (&v)[8], const float s) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * s, v[1] * s, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * s, v[3] * s, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * s, v[5] * s, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * s, v[7] * s, hi, true);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}

__global__ __launch_bounds__(64) void ctx_proj_fwd_fp8_kernel(const float* __restrict__ words, const float* __restrict__ W,
                                                              float* __restrict__ src, int B, int idf, int cdf, int L) {
    const int lane = threadIdx.x, rl = lane & 31, hf = lane >> 5;
    const int row = blockIdx.x * 32 + rl;                 // (b, l)
    const bool rv = row < B * L;
    const int b = rv ? row / L : 0, l = rv ? row - b * L : 0;
    const int i = blockIdx.y * 32 + rl;
    const bool iv = i < idf;
    const float* wp = words + (int64_t)b * cdf * L + l;             // channel k of this position: wp[k * L]
    const float* Wr = W + (int64_t)(iv ? i : 0) * cdf;
    // lane (rl, hf) owns k = c0 + 8 * hf + j (j < 8) of row rl in every 16-deep step: A[row][k], B[k][col]
    float ax = 0.f, aw = 0.f;
    for (int c0 = 0; c0 < cdf; c0 += 16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = c0 + 8 * hf + j;
            if (rv) ax = fmaxf(ax, fabsf(wp[(int64_t)k * L]));
            if (iv) aw = fmaxf(aw, fabsf(Wr[k]));
        }
    }
    ax = wave_max(ax);
    aw = wave_max(aw);
    // power-of-two scales: scaling and un-scaling are exact, at the cost of < 1 bit of the e4m3 range
    const float sx = ax > 0.f ? exp2f(floorf(log2f(448.f / ax))) : 1.f;
    const float sw = aw > 0.f ? exp2f(floorf(log2f(448.f / aw))) : 1.f;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int c0 = 0; c0 < cdf; c0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = c0 + 8 * hf + j;
            av[j] = rv ? wp[(int64_t)k * L] : 0.f;
            bv[j] = iv ? Wr[k] : 0.f;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pack8_fp8(av, sx), pack8_fp8(bv, sw), acc, 0, 0, 0);
    }
    const float inv = 1.f / (sx * sw);
    if (iv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (rr < B * L) {
                const int bb = rr / L, ll = rr - bb * L;
                src[((int64_t)bb * idf + i) * L + ll] = acc[r] * inv;
            }
        }
    }
}

// dW[i][c] += sum_{b,l} dsrc[b][i][l] words[b][c][l]: tile = 32 i x 32 c, reduction over the B*L positions
__global__ __launch_bounds__(64) void ctx_proj_bwd_w_mfma_kernel(const float* __restrict__ words,
                                                                 const float* __restrict__ dsrc, float* __restrict__ dW,
                                                                 int B, int idf, int cdf, int L) {
    const int lane = threadIdx.x, rl = lane & 31, hf = lane >> 5;
    const int c = blockIdx.x * 32 + rl, i = blockIdx.y * 32 + rl;
    const bool cv = c < cdf, iv = i < idf;
    const int P = B * L;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int p0 = 0; p0 < P; p0 += 16) {                   // 8 steps of 2 positions
        float av[8], bv[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int p = p0 + 2 * kk + hf;
            const bool ok = p < P;
            const int b = ok ? p / L : 0, l = ok ? p - b * L : 0;
            av[kk] = (ok && iv) ? dsrc[((int64_t)b * idf + i) * L + l] : 0.f;
            bv[kk] = (ok && cv) ? words[((int64_t)b * cdf + c) * L + l] : 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
    }
    if (cv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = blockIdx.y * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (ii < idf) dW[(int64_t)ii * cdf + c] += acc[r];
        }
    }
}

// CA_NET tail: h[B][4C]: GLU -> x[0..2C) = h[:, :2C] * sigmoid(h[:, 2C:]); mu = x[:C], logvar = x[C:]
__global__ void ca_fwd_kernel(const float* __restrict__ h, const float* __restrict__ eps, float* __restrict__ c,
                              float* __restrict__ mu, float* __restrict__ logvar, int B, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, j = i - b * C;
    const float* hr = h + (int64_t)b * 4 * C;
    const float m = hr[j] * sigmoidf_(hr[2 * C + j]);
    const float lv = hr[C + j] * sigmoidf_(hr[3 * C + j]);
    mu[i] = m;
    logvar[i] = lv;
    c[i] = eps[i] * expf(0.5f * lv) + m;
}

__global__ void ca_bwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                              const float* __restrict__ dc, const float* __restrict__ dmu,
                              const float* __restrict__ dlogvar, float* __restrict__ dh, int B, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, j = i - b * C;
    const float* hr = h + (int64_t)b * 4 * C;
    float* dr = dh + (int64_t)b * 4 * C;
    const float sm = sigmoidf_(hr[2 * C + j]), sl = sigmoidf_(hr[3 * C + j]);
    const float lv = hr[C + j] * sl;
    const float gc = dc ? dc[i] : 0.f;
    const float gm = gc + (dmu ? dmu[i] : 0.f);
    const float gl = gc * eps[i] * 0.5f * expf(0.5f * lv) + (dlogvar ? dlogvar[i] : 0.f);
    dr[j] = gm * sm;
    dr[2 * C + j] = gm * hr[j] * sm * (1.f - sm);
    dr[C + j] = gl * sl;
    dr[3 * C + j] = gl * hr[C + j] * sl * (1.f - sl);
}

// attention key projection: src[b][i][l] = sum_c W[i][c] words[b][c][l]
__global__ void ctx_proj_fwd_kernel(const float* __restrict__ words, const float* __restrict__ W,
                                    float* __restrict__ src, int B, int idf, int cdf, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * idf * L) return;
    const int l = i % L, o = (i / L) % idf, b = i / (L * idf);
    float acc = 0.f;
    for (int c = 0; c < cdf; ++c) acc += W[o * cdf + c] * words[((int64_t)b * cdf + c) * L + l];
    src[i] = acc;
}

// dW[i][c] += sum_{b,l} dsrc[b][i][l] words[b][c][l];  dwords[b][c][l] = sum_i W[i][c] dsrc[b][i][l]
__global__ void ctx_proj_bwd_kernel(const float* __restrict__ words, const float* __restrict__ W,
                                    const float* __restrict__ dsrc, float* __restrict__ dW,
                                    float* __restrict__ dwords, int B, int idf, int cdf, int L) {
    const int nW = idf * cdf;
    const int total = nW + (dwords ? B * cdf * L : 0);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < nW) {
            const int o = i / cdf, c = i - o * cdf;
            float acc = 0.f;
            for (int b = 0; b < B; ++b)
                for (int l = 0; l < L; ++l)
                    acc += dsrc[((int64_t)b * idf + o) * L + l] * words[((int64_t)b * cdf + c) * L + l];
            dW[i] += acc;
        } else {
            const int j = i - nW;
            const int l = j % L, c = (j / L) % cdf, b = j / (L * cdf);
            float acc = 0.f;
            for (int o = 0; o < idf; ++o) acc += W[o * cdf + c] * dsrc[((int64_t)b * idf + o) * L + l];
            dwords[j] = acc;
        }
    }
}

}  // namespace

extern "C" int sba_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                              void* stream) {
    if (!x || !w || !y || B <= 0 || K <= 0 || N <= 0) return SBA_E_ARG;
    if (B <= 32 && K % 4 == 0) {            // batch <= 32: matrix cores, one wave per 32 columns (INIT_STAGE_G.fc, 16384 columns:
                                            // 49 us as wave-per-column GEMVs)
        SBA_LAUNCH(linear_fwd_mfma_kernel, dim3(cdiv(N, 32)), dim3(64), 0, (hipStream_t)stream, x, w, bias, y, B, K, N);
        return SBA_CHECK_LAUNCH();
    }
    const size_t sh = sizeof(float) * B * K;
    if (sh > 64 * 1024) return SBA_E_ARG;
    int blocks = cdiv(N, 4);
    if (blocks > 2048) blocks = 2048;
    SBA_LAUNCH(linear_fwd_kernel, dim3(blocks), dim3(256), sh, (hipStream_t)stream, x, w, bias, y, B, K, N);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* dbias,
                              int B, int K, int N, void* stream) {
    if (!x || !w || !dy || B <= 0 || K <= 0 || N <= 0) return SBA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (B <= 32 && N % 4 == 0 && (N <= 2048 || (N % 32 == 0 && cdiv(N, 32) <= 65535))) {
        if (dw && dx && N <= 2048) {        // one launch for both (linear_bwd_xw_mfma_kernel)
            const int kt = cdiv(K, 32);
            SBA_LAUNCH(linear_bwd_xw_mfma_kernel, dim3(kt + kt * cdiv(N, 32)), dim3(64), 0, st, x, w, dy, dx, dw, dbias, B, K, N, kt);
            return SBA_CHECK_LAUNCH();
        }
        if (dw) SBA_LAUNCH(linear_bwd_w_mfma_kernel, dim3(cdiv(K, 32), cdiv(N, 32)), dim3(64), 0, st, x, dy, dw, dbias, B, K, N);
        if (dx && (N <= 2048 || sba_det_on())) {     // (deterministic mode: one slice, no atomics)
            SBA_LAUNCH(linear_bwd_x_mfma_kernel, dim3(cdiv(K, 32)), dim3(64), 0, st, w, dy, dx, B, K, N, N);
        } else if (dx) {       // wide layer (INIT_STAGE_G.fc): the reduction over N sliced over workgroups, f32 atomics into dx
            sba_zero_f32(dx, nullptr, (int64_t)B * K, st);
            const int nper = 128;
            SBA_LAUNCH(linear_bwd_x_mfma_kernel, dim3(cdiv(K, 32), cdiv(N, nper)), dim3(64), 0, st, w, dy, dx, B, K, N, nper);
        }
        return SBA_CHECK_LAUNCH();
    }
    const size_t sh = sizeof(float) * B * K;
    if (sh > 64 * 1024) return SBA_E_ARG;
    if (dw) {
        int blocks = cdiv(N, 4);
        if (blocks > 2048) blocks = 2048;
        SBA_LAUNCH(linear_bwd_w_kernel, dim3(blocks), dim3(256), sh, st, x, dy, dw, dbias, B, K, N);
    }
    if (dx && B <= 32 && N % 32 == 0) {    // wide layer (INIT_STAGE_G.fc): the reduction sliced over workgroups
        sba_zero_f32(dx, nullptr, (int64_t)B * K, st);
        const int nper = sba_det_on() ? N : 512;       // deterministic mode: one slice, no atomics
        SBA_LAUNCH(linear_bwd_x_mfma_kernel, dim3(cdiv(K, 32), cdiv(N, nper)), dim3(64), 0, st, w, dy, dx, B, K, N, nper);
    } else if (dx) {
        sba_zero_f32(dx, nullptr, (int64_t)B * K, st);
        const int nper = sba_det_on() ? N : (N >= 4096 ? 64 : (N >= 512 ? 16 : 4));    // deterministic: one workgroup
        SBA_LAUNCH(linear_bwd_x_kernel, dim3(cdiv(N, nper)), dim3(256), 0, st, w, dy, dx, B, K, N, nper);
    }
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_ca_fwd(const float* h, const float* eps, float* c, float* mu, float* logvar, int B, int C,
                          void* stream) {
    if (!h || !eps || !c || !mu || !logvar || B <= 0 || C <= 0) return SBA_E_ARG;
    SBA_LAUNCH(ca_fwd_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, h, eps, c, mu,
                       logvar, B, C);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_ca_bwd(const float* h, const float* eps, const float* dc, const float* dmu, const float* dlogvar,
                          float* dh, int B, int C, void* stream) {
    if (!h || !eps || !dh || B <= 0 || C <= 0) return SBA_E_ARG;
    SBA_LAUNCH(ca_bwd_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, h, eps, dc, dmu,
                       dlogvar, dh, B, C);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_ctx_proj_fwd(const float* words, const float* W, float* src, int B, int idf, int cdf, int L,
                                void* stream) {
    if (!words || !W || !src || B <= 0 || idf <= 0 || cdf <= 0 || L <= 0) return SBA_E_ARG;
    if (cdf % 4 == 0) {
        SBA_LAUNCH(ctx_proj_fwd_mfma_kernel, dim3(cdiv((int64_t)B * L, 32), cdiv(idf, 32)), dim3(64), 0,
                   (hipStream_t)stream, words, W, src, B, idf, cdf, L);
        return SBA_CHECK_LAUNCH();
    }
    SBA_LAUNCH(ctx_proj_fwd_kernel, dim3(cdiv((int64_t)B * idf * L, 256)), dim3(256), 0,
                       (hipStream_t)stream, words, W, src, B, idf, cdf, L);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_ctx_proj_fwd_fp8(const float* words, const float* W, float* src, int B, int idf, int cdf, int L,
                                    void* stream) {
    if (!words || !W || !src || B <= 0 || idf <= 0 || cdf <= 0 || L <= 0 || cdf % 16 != 0) return SBA_E_ARG;
    SBA_LAUNCH(ctx_proj_fwd_fp8_kernel, dim3(cdiv((int64_t)B * L, 32), cdiv(idf, 32)), dim3(64), 0, (hipStream_t)stream,
               words, W, src, B, idf, cdf, L);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_ctx_proj_bwd(const float* words, const float* W, const float* dsrc, float* dW, float* dwords,
                                int B, int idf, int cdf, int L, void* stream) {
    if (!words || !W || !dsrc || !dW || B <= 0 || idf <= 0 || cdf <= 0 || L <= 0) return SBA_E_ARG;
    if (!dwords) {      // GAN training: the word embeddings carry no gradient -- only dW, on the matrix cores
        SBA_LAUNCH(ctx_proj_bwd_w_mfma_kernel, dim3(cdiv(cdf, 32), cdiv(idf, 32)), dim3(64), 0, (hipStream_t)stream,
                   words, dsrc, dW, B, idf, cdf, L);
        return SBA_CHECK_LAUNCH();
    }
    const int total = idf * cdf + (dwords ? B * cdf * L : 0);
    SBA_LAUNCH(ctx_proj_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, words, W,
                       dsrc, dW, dwords, B, idf, cdf, L);
    return SBA_CHECK_LAUNCH();
}
