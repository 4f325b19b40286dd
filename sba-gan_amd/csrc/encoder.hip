// Non-GEMM pieces of the frozen image encoder (CNN_ENCODER, model.py:162-267: Inception-v3 trunk on a
// 299 x 299 bilinear resize) for gfx950.  The convolutions themselves run on the implicit-GEMM
// kernels of igemm.hip (BatchNorm(eval) folded into weights + bias, ReLU in the epilogue, concat by
// writing channel slices); this file holds the resize, the 3-channel stem conv, the pools and the
// ReLU-mask / layout helpers, forward and backward-data (the encoder's weights are frozen:
// trainer.py:57-63).  All are HBM-bound streaming kernels.
#include "common.h"

namespace {

// ---- bilinear resize, align_corners=True (model.py:210), NCHW f32 -> NCHW f32 ----
__global__ void resize_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int NC, int S, int D) {
    const int64_t total = (int64_t)NC * D * D;
    const float sc = (float)(S - 1) / (float)(D - 1);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % D), y = (int)((i / D) % D);
        const int64_t nc = i / ((int64_t)D * D);
        const float fy = y * sc, fx = x * sc;
        int y0 = (int)fy, x0 = (int)fx;
        y0 = min(y0, S - 1); x0 = min(x0, S - 1);
        const int y1 = min(y0 + 1, S - 1), x1 = min(x0 + 1, S - 1);
        const float wy = fy - y0, wx = fx - x0;
        const float* p = in + nc * S * S;
        const float v = (1.f - wy) * ((1.f - wx) * p[y0 * S + x0] + wx * p[y0 * S + x1]) +
                        wy * ((1.f - wx) * p[y1 * S + x0] + wx * p[y1 * S + x1]);
        out[i] = v;
    }
}

// adjoint of resize_fwd as a GATHER (one thread per input pixel, no atomics): input row r receives
// (1-wy) from the output rows whose y0 == r and wy from those whose y1 == r; the candidate rows are the
// few y with y*sc in (r-1, r+1).  The weights are recomputed with exactly the forward's arithmetic.
__device__ __forceinline__ float resize_adj_w(int o, int r, float sc, int S) {
    const float f = o * sc;
    int i0 = (int)f;
    i0 = min(i0, S - 1);
    const int i1 = min(i0 + 1, S - 1);
    const float wgt = f - i0;
    return (i0 == r ? 1.f - wgt : 0.f) + (i1 == r ? wgt : 0.f);
}

__global__ void resize_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int NC, int S, int D) {
    const int64_t total = (int64_t)NC * S * S;
    const float sc = (float)(S - 1) / (float)(D - 1);
    const float inv = 1.f / sc;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % S), r = (int)((i / S) % S);
        const int64_t nc = i / ((int64_t)S * S);
        const int ylo = max(0, (int)floorf((r - 1) * inv) - 1), yhi = min(D - 1, (int)ceilf((r + 1) * inv) + 1);
        const int xlo = max(0, (int)floorf((c - 1) * inv) - 1), xhi = min(D - 1, (int)ceilf((c + 1) * inv) + 1);
        const float* g = dout + nc * D * D;
        float acc = 0.f;
        for (int y = ylo; y <= yhi; ++y) {
            const float wy = resize_adj_w(y, r, sc, S);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int x = xlo; x <= xhi; ++x) {
                const float wx = resize_adj_w(x, c, sc, S);
                if (wx != 0.f) row += g[y * D + x] * wx;
            }
            acc += wy * row;
        }
        din[i] = acc;
    }
}

// The same with the candidates of every input index tabulated once per workgroup (the image is square: one table serves rows and
// columns): entry k of index r = (output index, weight) of the k-th output row / column with a non-zero weight, in ascending
// order -- the loop above spends its time re-deriving them (floorf / ceilf ranges, ~16 weight evaluations per pixel: 55 us at
// B = 20).  Same weights, same summation order.  Upscaling only (D >= S: at most RESIZE_MAXC candidates per index).
constexpr int RESIZE_MAXC = 4;
__global__ __launch_bounds__(256) void resize_bwd_tab_kernel(const float* __restrict__ dout, float* __restrict__ din, int NC,
                                                             int S, int D) {
    extern __shared__ unsigned char rs_lds[];
    int* t_o = reinterpret_cast<int*>(rs_lds);                          // [S][RESIZE_MAXC]
    float* t_w = reinterpret_cast<float*>(rs_lds + (size_t)S * RESIZE_MAXC * 4);
    const float sc = (float)(S - 1) / (float)(D - 1);
    const float inv = 1.f / sc;
    for (int r = threadIdx.x; r < S; r += blockDim.x) {
        const int lo = max(0, (int)floorf((r - 1) * inv) - 1), hi = min(D - 1, (int)ceilf((r + 1) * inv) + 1);
        int k = 0;
        for (int o = lo; o <= hi && k < RESIZE_MAXC; ++o) {
            const float wgt = resize_adj_w(o, r, sc, S);
            if (wgt != 0.f) { t_o[r * RESIZE_MAXC + k] = o; t_w[r * RESIZE_MAXC + k] = wgt; ++k; }
        }
        for (; k < RESIZE_MAXC; ++k) { t_o[r * RESIZE_MAXC + k] = 0; t_w[r * RESIZE_MAXC + k] = 0.f; }
    }
    __syncthreads();
    const int64_t total = (int64_t)NC * S * S;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % S), r = (int)((i / S) % S);
        const int64_t nc = i / ((int64_t)S * S);
        const float* g = dout + nc * D * D;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < RESIZE_MAXC; ++ky) {
            const float wy = t_w[r * RESIZE_MAXC + ky];
            if (wy == 0.f) continue;
            const float* grow = g + t_o[r * RESIZE_MAXC + ky] * D;
            float row = 0.f;
#pragma unroll
            for (int kx = 0; kx < RESIZE_MAXC; ++kx) {
                const float wx = t_w[c * RESIZE_MAXC + kx];
                if (wx != 0.f) row += grow[t_o[c * RESIZE_MAXC + kx]] * wx;
            }
            acc += wy * row;
        }
        din[i] = acc;
    }
}

// ---- stem: conv3x3 s2 p0 (3 -> C) + bias + ReLU, NCHW f32 in, NHWC T out ----
// thread = (output pixel, V channels); weights [C][3][3][3] (channels_last OIHW) transposed to [27][C] in LDS
template <typename T>
__global__ __launch_bounds__(256) void enc_stem_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ out,
                                                           int N, int S, int O, int C) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ float s_w[];       // [27][C]
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) {
        const int co = i / 27, k = i - co * 27;
        s_w[k * C + co] = w[i];
    }
    __syncthreads();
    const int cv = C / V;
    const int64_t total = (int64_t)N * O * O * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t p = i / cv;
        const int ox = (int)(p % O), oy = (int)((p / O) % O), n = (int)(p / ((int64_t)O * O));
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = bias ? bias[c + k] : 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float v = img[(((int64_t)n * 3 + ci) * S + 2 * oy + kh) * S + 2 * ox + kw];
                    const float* wr = &s_w[((kh * 3 + kw) * 3 + ci) * C + c];
#pragma unroll
                    for (int k = 0; k < V; ++k) acc[k] += v * wr[k];
                }
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.set(k, bias ? fmaxf(acc[k], 0.f) : acc[k]);     // bias == NULL: the raw conv output
        st16(out + p * C + c, o);
    }
}

// The same conv reading the bilinear resize (resize_fwd_kernel: align_corners, S -> D) of the image ON THE FLY: the D x D
// tensor is never written.  Same thread mapping as above (output pixel, V channels): the C / V threads of a pixel each
// interpolate its 27 inputs (L1 hits; the expression is resize_fwd_kernel's term by term); one kernel row at a time keeps
// the live loads at 36 and the occupancy high -- the kernel is load-latency bound.
template <typename T>
__global__ __launch_bounds__(256) void enc_stem_resize_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, T* __restrict__ out,
                                                                  int N, int S, int D, int O, int C) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ float s_w[];       // [27][C]
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) {
        const int co = i / 27, k = i - co * 27;
        s_w[k * C + co] = w[i];
    }
    __syncthreads();
    const float sc = (float)(S - 1) / (float)(D - 1);
    const int cv = C / V;
    const int64_t total = (int64_t)N * O * O * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t p = i / cv;
        const int ox = (int)(p % O), oy = (int)((p / O) % O), n = (int)(p / ((int64_t)O * O));
        int x0[3], x1[3];
        float wx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float fx = (2 * ox + k) * sc;
            x0[k] = min((int)fx, S - 1);
            x1[k] = min(x0[k] + 1, S - 1);
            wx[k] = fx - x0[k];
        }
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = bias ? bias[c + k] : 0.f;
        const float* q0 = img + (int64_t)n * 3 * S * S;
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
            const float fy = (2 * oy + kh) * sc;
            const int y0 = min((int)fy, S - 1), y1 = min(y0 + 1, S - 1);
            const float wy = fy - y0;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float* q = q0 + (int64_t)ci * S * S;
                    const float v = (1.f - wy) * ((1.f - wx[kw]) * q[y0 * S + x0[kw]] + wx[kw] * q[y0 * S + x1[kw]]) +
                                    wy * ((1.f - wx[kw]) * q[y1 * S + x0[kw]] + wx[kw] * q[y1 * S + x1[kw]]);
                    const float* wr = &s_w[((kh * 3 + kw) * 3 + ci) * C + c];
#pragma unroll
                    for (int k = 0; k < V; ++k) acc[k] += v * wr[k];
                }
        }
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.set(k, bias ? fmaxf(acc[k], 0.f) : acc[k]);
        st16(out + p * C + c, o);
    }
}

// dimg[n][ci][iy][ix] = sum over output pixels (<= 2 x 2) reading it and all co of dout*(out>0)*w
template <typename T>
__global__ __launch_bounds__(256) void enc_stem_bwd_kernel(const float* __restrict__ w, const T* __restrict__ out,
                                                           const T* __restrict__ dout, float* __restrict__ dimg,
                                                           int N, int S, int O, int C) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ float s_w[];       // [27][C]
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) {
        const int co = i / 27, k = i - co * 27;
        s_w[k * C + co] = w[i];
    }
    __syncthreads();
    const int64_t total = (int64_t)N * S * S;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(p % S), iy = (int)((p / S) % S), n = (int)(p / ((int64_t)S * S));
        float acc[3] = {0.f, 0.f, 0.f};
        for (int kh = iy & 1; kh < 3; kh += 2) {
            const int oy = (iy - kh) >> 1;
            if (iy - kh < 0 || oy >= O) continue;
            for (int kw = ix & 1; kw < 3; kw += 2) {
                const int ox = (ix - kw) >> 1;
                if (ix - kw < 0 || ox >= O) continue;
                const int64_t q = (((int64_t)n * O + oy) * O + ox) * C;
                for (int cv = 0; cv < C / V; ++cv) {
                    Vec16<T> ov = ld16(out + q + cv * V), dv = ld16(dout + q + cv * V);
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        const float dp = ov.get(k) > 0.f ? dv.get(k) : 0.f;
                        const int co = cv * V + k;
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) acc[ci] += dp * s_w[((kh * 3 + kw) * 3 + ci) * C + co];
                    }
                }
            }
        }
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) dimg[(((int64_t)n * 3 + ci) * S + iy) * S + ix] = acc[ci];
    }
}

// The same gradient on the matrix cores (bf16 features, C = 32).  The 2 x 2 block of image pixels (2 by + a, 2 bx + b) is read
// by the 2 x 2 neighbourhood of output positions (by - dy, bx - dx), dy, dx in {0, 1}, through the taps kh = a + 2 dy,
// kw = b + 2 dx (no term where a tap index exceeds 2):
//   dimg[ci][2 by + a][2 bx + b] = sum_{dy, dx} sum_co dpre[by - dy][bx - dx][co] w[co][kh][kw][ci]
// -- a 2 x 2 convolution of dpre = dout (out > 0) with 12 "channels" (ci, a, b) and K = 4 x 32: per tile of 16 consecutive bx
// of one row by, four steps of v_mfma_f32_16x16x32_bf16 (x 2: the f32 weights as hi + lo bf16 parts, so that the products are
// the VALU kernel's to within f32 rounding).  The weights (A operand: 16 rows, 12 used) live in registers for the whole
// launch; a lane's B fragment -- 8 channels of one position -- is ONE 16-byte load of dout and one of out straight from
// global memory (64 lanes = 16 positions x 64 B: 1 KB contiguous), no LDS; the accumulator of lane (position, ci) is exactly
// the 2 x 2 pixel block of channel ci.  (enc_stem_bwd_kernel: 4 x 32 x 3 FMAs per pixel on the vector pipe, 80 us at B = 20.)
__global__ __launch_bounds__(256) void enc_stem_bwd_mfma_kernel(const float* __restrict__ w, const bf16_t* __restrict__ out,
                                                                const bf16_t* __restrict__ dout, float* __restrict__ dimg,
                                                                const int N, const int S, const int O, const int ntiles) {
    constexpr int C = 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    bf16x8_t a_hi[4], a_lo[4];
    {
        const int ci = m >> 2, a = (m >> 1) & 1, b = m & 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kh = a + 2 * (t >> 1), kw = b + 2 * (t & 1);
            const bool ok = m < 12 && kh < 3 && kw < 3;
            bf16x8_t hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int co = 8 * g + j;
                const float v = ok ? w[co * 27 + (kh * 3 + kw) * 3 + ci] : 0.f;
                const bf16_t h = f2bf(v);
                hi[j] = (short)h;
                lo[j] = (short)f2bf(v - bf2f(h));
            }
            a_hi[t] = hi;
            a_lo[t] = lo;
        }
    }
    const int BH = (S + 1) / 2, tiles_x = (BH + 15) / 16;          // 2 x 2 pixel blocks per image side, 16-block tiles per row
    for (int tile = blockIdx.x * 4 + wid; tile < ntiles; tile += gridDim.x * 4) {
        const int txi = tile % tiles_x, by = (tile / tiles_x) % BH, n = tile / (tiles_x * BH);
        const int bx = txi * 16 + m;
        f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int oy = by - (t >> 1), ox = bx - (t & 1);
            uint4 dv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0);
            if (oy >= 0 && oy < O && ox >= 0 && ox < O) {
                const int64_t q = (((int64_t)n * O + oy) * O + ox) * C + 8 * g;
                dv = *reinterpret_cast<const uint4*>(dout + q);
                ov = *reinterpret_cast<const uint4*>(out + q);
            }
            // dpre = dout where out > 0 (bf16 bits: positive and not zero)
            const uint32_t d[4] = {dv.x, dv.y, dv.z, dv.w}, o[4] = {ov.x, ov.y, ov.z, ov.w};
            bf16x8_t bfr;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t lo_ok = ((o[k] & 0x7fffu) != 0u && (o[k] & 0x8000u) == 0u) ? 0x0000ffffu : 0u;
                const uint32_t hi_ok = ((o[k] & 0x7fff0000u) != 0u && (o[k] & 0x80000000u) == 0u) ? 0xffff0000u : 0u;
                const uint32_t v = d[k] & (lo_ok | hi_ok);
                bfr[2 * k] = (short)(v & 0xffffu);
                bfr[2 * k + 1] = (short)(v >> 16);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[t], bfr, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[t], bfr, acc, 0, 0, 0);
        }
        // D layout: column = lane & 15 (block bx), rows 4 g .. 4 g + 3 = (ci = g; a, b)
        if (g < 3 && bx < BH) {
            float* o0 = dimg + (((int64_t)n * 3 + g) * S + 2 * by) * S + 2 * bx;
            const bool x1 = 2 * bx + 1 < S, y1 = 2 * by + 1 < S;
            o0[0] = acc[0];
            if (x1) o0[1] = acc[1];
            if (y1) {
                o0[S] = acc[2];
                if (x1) o0[S + 1] = acc[3];
            }
        }
    }
}

// ---- max pool 3x3 stride 2 (no padding), NHWC with channel strides ----
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int xcs,
                                   int xco, int ycs, int yco) {
    constexpr int V = Vec16<T>::N;
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1, cv = C / V;
    const int64_t total = (int64_t)N * OH * OW * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t p = i / cv;
        const int ox = (int)(p % OW), oy = (int)((p / OW) % OH), n = (int)(p / ((int64_t)OW * OH));
        float m[V];
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                Vec16<T> v = ld16(x + (((int64_t)n * H + 2 * oy + kh) * W + 2 * ox + kw) * xcs + xco + c);
#pragma unroll
                for (int k = 0; k < V; ++k) m[k] = fmaxf(m[k], v.get(k));
            }
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.set(k, m[k]);
        st16(y + p * ycs + yco + c, o);
    }
}

// forward that also stores the window position (kh * 3 + kw, FIRST maximum in scan order: torch's tie rule) of every
// output element: the backward then reads <= 4 argmax bytes + 4 dy vectors per input vector instead of re-reading the
// 9 inputs of each of its <= 4 windows (36 loads) to recompute them
template <typename T>
__global__ void maxpool_fwd_arg_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ arg, int N,
                                       int H, int W, int C, int xcs, int xco, int ycs, int yco) {
    constexpr int V = Vec16<T>::N;
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1, cv = C / V;
    const int64_t total = (int64_t)N * OH * OW * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t p = i / cv;
        const int ox = (int)(p % OW), oy = (int)((p / OW) % OH), n = (int)(p / ((int64_t)OW * OH));
        float m[V];
        uint8_t a[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { m[k] = -INFINITY; a[k] = 255; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                Vec16<T> v = ld16(x + (((int64_t)n * H + 2 * oy + kh) * W + 2 * ox + kw) * xcs + xco + c);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float f = v.get(k);
                    if (f > m[k]) { m[k] = f; a[k] = (uint8_t)(kh * 3 + kw); }
                }
            }
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.set(k, m[k]);
        st16(y + p * ycs + yco + c, o);
        uint8_t* ap = arg + p * C + c;          // V bytes, V-aligned (C % V == 0)
        if (V == 8) {
            uint2 w;
            w.x = a[0] | (a[1] << 8) | (a[2] << 16) | ((uint32_t)a[3] << 24);
            w.y = a[4] | (a[5] << 8) | (a[6] << 16) | ((uint32_t)a[7] << 24);
            *reinterpret_cast<uint2*>(ap) = w;
        } else {
            *reinterpret_cast<uint32_t*>(ap) = a[0] | (a[1] << 8) | (a[2] << 16) | ((uint32_t)a[3] << 24);
        }
    }
}

template <typename T>
__global__ void maxpool_bwd_arg_kernel(const uint8_t* __restrict__ arg, const T* __restrict__ dy, T* __restrict__ dx,
                                       int N, int H, int W, int C, int dycs, int dyco, int dxcs, int dxco,
                                       int accumulate, const T* __restrict__ relu_mask) {
    constexpr int V = Vec16<T>::N;
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1, cv = C / V;
    const int64_t total = (int64_t)N * H * W * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t p = i / cv;
        const int ix = (int)(p % W), iy = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
        float g[V];
#pragma unroll
        for (int k = 0; k < V; ++k) g[k] = 0.f;
        for (int oy = max(0, (iy - 1) >> 1); oy <= min(OH - 1, iy >> 1); ++oy) {
            if (iy - 2 * oy > 2) continue;
            for (int ox = max(0, (ix - 1) >> 1); ox <= min(OW - 1, ix >> 1); ++ox) {
                if (ix - 2 * ox > 2) continue;
                const uint32_t mypos = (uint32_t)((iy - 2 * oy) * 3 + (ix - 2 * ox));
                const int64_t q = ((int64_t)n * OH + oy) * OW + ox;
                uint32_t aw[2] = {0u, 0u};
                if (V == 8) {
                    const uint2 w = *reinterpret_cast<const uint2*>(arg + q * C + c);
                    aw[0] = w.x; aw[1] = w.y;
                } else {
                    aw[0] = *reinterpret_cast<const uint32_t*>(arg + q * C + c);
                }
                Vec16<T> d = ld16(dy + q * dycs + dyco + c);
#pragma unroll
                for (int k = 0; k < V; ++k)
                    if (((aw[k >> 2] >> (8 * (k & 3))) & 255u) == mypos) g[k] += d.get(k);
            }
        }
        T* op = dx + p * dxcs + dxco + c;
        Vec16<T> o;
        if (accumulate) o = ld16(op);
#pragma unroll
        for (int k = 0; k < V; ++k) o.set(k, g[k] + (accumulate ? o.get(k) : 0.f));
        if (relu_mask) {        // laid out like dx: the backward of the ReLU that produced the pooled tensor
            const Vec16<T> m = ld16(relu_mask + p * dxcs + dxco + c);
#pragma unroll
            for (int k = 0; k < V; ++k) if (!(m.get(k) > 0.f)) o.set(k, 0.f);
        }
        st16(op, o);
    }
}

// dx[p] (+)= sum over the windows containing p whose FIRST maximum (scan order kh, kw: torch's tie rule)
// is p, of dy[window]
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int N, int H,
                                   int W, int C, int xcs, int xco, int dycs, int dyco, int dxcs, int dxco,
                                   int accumulate) {
    constexpr int V = Vec16<T>::N;
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1, cv = C / V;
    const int64_t total = (int64_t)N * H * W * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t p = i / cv;
        const int ix = (int)(p % W), iy = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
        float g[V];
#pragma unroll
        for (int k = 0; k < V; ++k) g[k] = 0.f;
        for (int oy = max(0, (iy - 1) >> 1); oy <= min(OH - 1, iy >> 1); ++oy) {
            if (iy - 2 * oy > 2) continue;
            for (int ox = max(0, (ix - 1) >> 1); ox <= min(OW - 1, ix >> 1); ++ox) {
                if (ix - 2 * ox > 2) continue;
                const int mypos = (iy - 2 * oy) * 3 + (ix - 2 * ox);
                float m[V];
                int arg[V];
#pragma unroll
                for (int k = 0; k < V; ++k) { m[k] = -INFINITY; arg[k] = -1; }
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        Vec16<T> v = ld16(x + (((int64_t)n * H + 2 * oy + kh) * W + 2 * ox + kw) * xcs + xco + c);
#pragma unroll
                        for (int k = 0; k < V; ++k) {
                            const float f = v.get(k);
                            if (f > m[k]) { m[k] = f; arg[k] = kh * 3 + kw; }
                        }
                    }
                Vec16<T> d = ld16(dy + (((int64_t)n * OH + oy) * OW + ox) * dycs + dyco + c);
#pragma unroll
                for (int k = 0; k < V; ++k)
                    if (arg[k] == mypos) g[k] += d.get(k);
            }
        }
        T* op = dx + p * dxcs + dxco + c;
        Vec16<T> o;
        if (accumulate) o = ld16(op);
#pragma unroll
        for (int k = 0; k < V; ++k) o.set(k, g[k] + (accumulate ? o.get(k) : 0.f));
        st16(op, o);
    }
}

// ---- avg pool 3x3 stride 1 pad 1, count_include_pad (divide by 9); its backward is the same operator ----
template <typename T>
__global__ void avgpool3_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int xcs,
                                int xco, int ycs, int yco, int accumulate) {
    constexpr int V = Vec16<T>::N;
    const int cv = C / V;
    const int64_t total = (int64_t)N * H * W * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t p = i / cv;
        const int ix = (int)(p % W), iy = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
        float s[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = iy + dy, xx = ix + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                Vec16<T> v = ld16(x + (((int64_t)n * H + yy) * W + xx) * xcs + xco + c);
#pragma unroll
                for (int k = 0; k < V; ++k) s[k] += v.get(k);
            }
        T* op = y + p * ycs + yco + c;
        Vec16<T> o;
        if (accumulate) o = ld16(op);
#pragma unroll
        for (int k = 0; k < V; ++k) o.set(k, s[k] * (1.f / 9.f) + (accumulate ? o.get(k) : 0.f));
        st16(op, o);
    }
}

// ---- dpre = (out > 0) ? dout : 0 on a channel slice ----
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ out, const T* __restrict__ dout, T* __restrict__ dpre,
                                int64_t rows, int C, int ocs, int oco, int dcs, int dco) {
    constexpr int V = Vec16<T>::N;
    const int cv = C / V;
    const int64_t total = rows * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t r = i / cv;
        Vec16<T> o = ld16(out + r * ocs + oco + c), d = ld16(dout + r * dcs + dco + c), q;
#pragma unroll
        for (int k = 0; k < V; ++k) q.set(k, o.get(k) > 0.f ? d.get(k) : 0.f);
        st16(dpre + r * C + c, q);
    }
}

// ---- global average pool over HW (NHWC T -> [N][C] f32) and its backward ----
template <typename T>
__global__ void gap_fwd_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += to_f<T>(x[((int64_t)n * HW + p) * C + c]);
    y[i] = s / HW;
}
// one thread = one 16-byte vector of channels of one pixel (the scalar version -- a 64-bit divide and a 2-byte store per
// element -- took 190-290 us for 5 MB on the image encoder's critical chain)
template <typename T>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dy, T* __restrict__ dx, int N, int HW, int C) {
    constexpr int V = Vec16<T>::N;
    const int cv = C / V;
    const int total = N * HW * cv;                       // (checked by the entry point: fits 31 bits)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / cv, c = (i - r * cv) * V;      // r = n * HW + pixel
        const int n = r / HW;
        const float* src = dy + (int64_t)n * C + c;
        Vec16<T> q;
#pragma unroll
        for (int k = 0; k < V; ++k) q.set(k, src[k] / HW);      // (the same expression as before: bit-identical)
        st16(dx + (int64_t)r * C + c, q);
    }
}

// ---- NHWC T <-> NCHW f32 (region features handed to the DAMSM loss) ----
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    const int64_t total = (int64_t)N * HW * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW), c = (int)((i / HW) % C), n = (int)(i / ((int64_t)HW * C));
        y[i] = to_f<T>(x[((int64_t)n * HW + p) * C + c]);
    }
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int HW, int C) {
    const int64_t total = (int64_t)N * HW * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), p = (int)((i / C) % HW), n = (int)(i / ((int64_t)HW * C));
        y[i] = from_f<T>(x[((int64_t)n * C + c) * HW + p]);
    }
}

inline int grid_for(int64_t items, int cap = 8192) {
    int64_t b = (items + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}
inline bool slice_ok(int dtype, int C, int cs, int co) {
    const int V = dtype == SBA_BF16 ? 8 : 4;
    return C > 0 && C % V == 0 && cs >= C + co && co >= 0 && cs % V == 0 && co % V == 0;
}

}  // namespace

extern "C" int sba_resize_bilinear(const float* in, float* out, int NC, int S, int D, int backward, void* stream) {
    if (!in || !out || NC <= 0 || S < 2 || D < 2) return SBA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (!backward) {
        SBA_LAUNCH(resize_fwd_kernel, dim3(grid_for((int64_t)NC * D * D)), dim3(256), 0, st, in, out, NC, S, D);
    } else {     // in = d(out) [NC][D][D], out = d(in) [NC][S][S]
        static int tab = -1;        // SBA_RESIZE_BWD_TAB=0: the kernel that re-derives the candidates per pixel (A/B aid)
        if (tab < 0) { const char* e = getenv("SBA_RESIZE_BWD_TAB"); tab = (e && e[0] == '0') ? 0 : 1; }
        // (an input index collects the outputs o with o (S-1)/(D-1) within (r-1, r+1): at most 2 (D-1)/(S-1) + 1 <= RESIZE_MAXC)
        if (tab && D >= S && S >= 2 && S <= 2048 && 2 * (int64_t)(D - 1) <= 3 * (int64_t)(S - 1))
            SBA_LAUNCH(resize_bwd_tab_kernel, dim3(grid_for((int64_t)NC * S * S, 2048)), dim3(256), (size_t)S * RESIZE_MAXC * 8,
                       st, in, out, NC, S, D);
        else
        SBA_LAUNCH(resize_bwd_kernel, dim3(grid_for((int64_t)NC * S * S)), dim3(256), 0, st, in, out, NC, S, D);
    }
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_enc_stem_fwd(int dtype, const float* img, const float* w, const float* bias, void* out, int N,
                                int S, int C, void* stream) {
    if (!img || !w || !out || N <= 0 || S < 3 || C <= 0 || C % 8 || C > 256) return SBA_E_ARG;
    const int O = (S - 3) / 2 + 1, V = dtype == SBA_BF16 ? 8 : 4;
    SBA_DISPATCH(dtype, SBA_LAUNCH((enc_stem_fwd_kernel<T>), dim3(grid_for((int64_t)N * O * O * (C / V), 4096)),
                                           dim3(256), sizeof(float) * 27 * C, (hipStream_t)stream, img, w, bias,
                                           (T*)out, N, S, O, C));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_enc_stem_bwd(int dtype, const float* w, const void* out, const void* dout, float* dimg, int N,
                                int S, int C, void* stream) {
    if (!w || !out || !dout || !dimg || N <= 0 || S < 3 || C <= 0 || C % 8 || C > 256) return SBA_E_ARG;
    const int O = (S - 3) / 2 + 1;
    static int mf = -1;         // SBA_ENC_STEM_BWD_MFMA=0: the VALU kernel also for bf16 (A/B aid)
    if (mf < 0) { const char* e = getenv("SBA_ENC_STEM_BWD_MFMA"); mf = (e && e[0] == '0') ? 0 : 1; }
    if (mf && dtype == SBA_BF16 && C == 32 && (((uintptr_t)out | (uintptr_t)dout) & 15) == 0) {
        const int BH = (S + 1) / 2, ntiles = N * BH * ((BH + 15) / 16);
        SBA_LAUNCH(enc_stem_bwd_mfma_kernel, dim3((ntiles + 3) / 4 < 8192 ? (ntiles + 3) / 4 : 8192), dim3(256), 0,
                   (hipStream_t)stream, w, (const bf16_t*)out, (const bf16_t*)dout, dimg, N, S, O, ntiles);
        return SBA_CHECK_LAUNCH();
    }
    SBA_DISPATCH(dtype, SBA_LAUNCH((enc_stem_bwd_kernel<T>), dim3(grid_for((int64_t)N * S * S, 4096)), dim3(256),
                                           sizeof(float) * 27 * C, (hipStream_t)stream, w, (const T*)out,
                                           (const T*)dout, dimg, N, S, O, C));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_maxpool3x3s2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, int xcs, int xco,
                                    int ycs, int yco, void* stream) {
    if (!x || !y || N <= 0 || H < 3 || W < 3 || !slice_ok(dtype, C, xcs, xco) || !slice_ok(dtype, C, ycs, yco))
        return SBA_E_ARG;
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1, V = dtype == SBA_BF16 ? 8 : 4;
    SBA_DISPATCH(dtype, SBA_LAUNCH((maxpool_fwd_kernel<T>), dim3(grid_for((int64_t)N * OH * OW * (C / V))),
                                           dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, N, H, W, C, xcs, xco,
                                           ycs, yco));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_maxpool3x3s2_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int C,
                                    int xcs, int xco, int dycs, int dyco, int dxcs, int dxco, int accumulate,
                                    void* stream) {
    if (!x || !dy || !dx || N <= 0 || H < 3 || W < 3 || !slice_ok(dtype, C, xcs, xco) ||
        !slice_ok(dtype, C, dycs, dyco) || !slice_ok(dtype, C, dxcs, dxco))
        return SBA_E_ARG;
    const int V = dtype == SBA_BF16 ? 8 : 4;
    SBA_DISPATCH(dtype, SBA_LAUNCH((maxpool_bwd_kernel<T>), dim3(grid_for((int64_t)N * H * W * (C / V))),
                                           dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, N, H,
                                           W, C, xcs, xco, dycs, dyco, dxcs, dxco, accumulate));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_maxpool3x3s2_fwd_arg(int dtype, const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C,
                                        int xcs, int xco, int ycs, int yco, void* stream) {
    if (!x || !y || !argmax || N <= 0 || H < 3 || W < 3 || !slice_ok(dtype, C, xcs, xco) || !slice_ok(dtype, C, ycs, yco))
        return SBA_E_ARG;
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1, V = dtype == SBA_BF16 ? 8 : 4;
    if (((uintptr_t)argmax & 7) != 0) return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((maxpool_fwd_arg_kernel<T>), dim3(grid_for((int64_t)N * OH * OW * (C / V))),
                                           dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, argmax, N, H, W, C,
                                           xcs, xco, ycs, yco));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_maxpool3x3s2_bwd_arg(int dtype, const uint8_t* argmax, const void* dy, void* dx, int N, int H, int W,
                                        int C, int dycs, int dyco, int dxcs, int dxco, int accumulate,
                                        const void* relu_mask, void* stream) {
    if (!argmax || !dy || !dx || N <= 0 || H < 3 || W < 3 || !slice_ok(dtype, C, dycs, dyco) ||
        !slice_ok(dtype, C, dxcs, dxco) || ((uintptr_t)argmax & 7) != 0)
        return SBA_E_ARG;
    const int V = dtype == SBA_BF16 ? 8 : 4;
    SBA_DISPATCH(dtype, SBA_LAUNCH((maxpool_bwd_arg_kernel<T>), dim3(grid_for((int64_t)N * H * W * (C / V))),
                                           dim3(256), 0, (hipStream_t)stream, argmax, (const T*)dy, (T*)dx, N, H, W, C,
                                           dycs, dyco, dxcs, dxco, accumulate, (const T*)relu_mask));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_enc_stem_resize_fwd(int dtype, const float* img, const float* w, const float* bias, void* out, int N,
                                       int S, int D, int C, void* stream) {
    if (!img || !w || !out || N <= 0 || S < 2 || D < 3 || C <= 0 || C % 8 || C > 256) return SBA_E_ARG;
    const int O = (D - 3) / 2 + 1, V = dtype == SBA_BF16 ? 8 : 4;
    SBA_DISPATCH(dtype, SBA_LAUNCH((enc_stem_resize_fwd_kernel<T>), dim3(grid_for((int64_t)N * O * O * (C / V), 8192)),
                                           dim3(256), sizeof(float) * 27 * C, (hipStream_t)stream, img, w, bias, (T*)out, N,
                                           S, D, O, C));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_avgpool3x3(int dtype, const void* x, void* y, int N, int H, int W, int C, int xcs, int xco,
                              int ycs, int yco, int accumulate, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || !slice_ok(dtype, C, xcs, xco) || !slice_ok(dtype, C, ycs, yco))
        return SBA_E_ARG;
    const int V = dtype == SBA_BF16 ? 8 : 4;
    SBA_DISPATCH(dtype, SBA_LAUNCH((avgpool3_kernel<T>), dim3(grid_for((int64_t)N * H * W * (C / V))), dim3(256),
                                           0, (hipStream_t)stream, (const T*)x, (T*)y, N, H, W, C, xcs, xco, ycs, yco,
                                           accumulate));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_relu_bwd(int dtype, const void* out, const void* dout, void* dpre, int64_t rows, int C, int ocs,
                            int oco, int dcs, int dco, void* stream) {
    if (!out || !dout || !dpre || rows <= 0 || !slice_ok(dtype, C, ocs, oco) || !slice_ok(dtype, C, dcs, dco))
        return SBA_E_ARG;
    const int V = dtype == SBA_BF16 ? 8 : 4;
    SBA_DISPATCH(dtype, SBA_LAUNCH((relu_bwd_kernel<T>), dim3(grid_for(rows * (C / V))), dim3(256), 0,
                                           (hipStream_t)stream, (const T*)out, (const T*)dout, (T*)dpre, rows, C, ocs,
                                           oco, dcs, dco));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_global_avgpool(int dtype, void* x, float* y, int N, int HW, int C, int backward, void* stream) {
    if (!x || !y || N <= 0 || HW <= 0 || C <= 0) return SBA_E_ARG;
    if (!backward) {
        SBA_DISPATCH(dtype, SBA_LAUNCH((gap_fwd_kernel<T>), dim3(cdiv((int64_t)N * C, 256)), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)x, y, N, HW, C));
    } else {     // x = dx (output, T), y = dy (input, f32)
        if (C % 8 != 0 || (int64_t)N * HW * C >= (1ll << 31)) return SBA_E_ARG;
        SBA_DISPATCH(dtype, SBA_LAUNCH((gap_bwd_kernel<T>), dim3(grid_for((int64_t)N * HW * C / (dtype == SBA_BF16 ? 8 : 4))), dim3(256), 0,
                                               (hipStream_t)stream, (const float*)y, (T*)x, N, HW, C));
    }
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_layout_nhwc_nchw(int dtype, void* nhwc, float* nchw, int N, int HW, int C, int to_nhwc, void* stream) {
    if (!nhwc || !nchw || N <= 0 || HW <= 0 || C <= 0) return SBA_E_ARG;
    if (!to_nhwc) {
        SBA_DISPATCH(dtype, SBA_LAUNCH((nhwc_to_nchw_kernel<T>), dim3(grid_for((int64_t)N * HW * C)), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)nhwc, nchw, N, HW, C));
    } else {
        SBA_DISPATCH(dtype, SBA_LAUNCH((nchw_to_nhwc_kernel<T>), dim3(grid_for((int64_t)N * HW * C)), dim3(256), 0,
                                               (hipStream_t)stream, (const float*)nchw, (T*)nhwc, N, HW, C));
    }
    return SBA_CHECK_LAUNCH();
}
