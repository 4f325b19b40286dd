// Narrow-channel layers at the two ends of the networks (gfx950).  These are
// HBM / issue-bound, not MFMA-shaped (3 image channels on one side):
//   - generator image head  conv3x3(ngf->3)+tanh      (GET_IMAGE_G, model.py:426-437)
//   - discriminator stem    conv4x4 s2 (3->ndf)+LReLU (encode_image_by_16times, model.py:563-564)
//   - logits head           conv4x4 s4 (8ndf->1)+sigmoid on the 4x4 map (D_GET_LOGITS, model.py:590-607)
//   - conditioning concat   (model.py:597-600)
// Images stay in the reference's NCHW f32 layout at the boundary; features are NHWC.
#include <stdlib.h>

#include "common.h"

namespace {

// ------------------------------------------------------------------ image head
template <typename T, int C>
__global__ __launch_bounds__(256) void img_head_fwd_kernel(const T* __restrict__ h, const float* __restrict__ w,
                                                           float* __restrict__ img, int N, int H, int W) {
    constexpr int V = Vec16<T>::N;
    __shared__ float s_w[27 * C];            // [co][kh][kw][ci]
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) s_w[i] = w[i];
    __syncthreads();
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= (int64_t)N * H * W) return;
    const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = y + kh - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = x + kw - 1;
            if (ix < 0 || ix >= W) continue;
            const T* hp = h + (((int64_t)n * H + iy) * W + ix) * C;
#pragma unroll
            for (int cv = 0; cv < C / V; ++cv) {
                Vec16<T> v = ld16(hp + cv * V);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float hv = v.get(k);
#pragma unroll
                    for (int co = 0; co < 3; ++co) acc[co] += hv * s_w[((co * 3 + kh) * 3 + kw) * C + cv * V + k];
                }
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 3; ++co) img[(((int64_t)n * 3 + co) * H + y) * W + x] = tanhf(acc[co]);
}

// one thread per INPUT pixel p: d[27] = dpre of the 9 output pixels that read p (x 3 channels);
// dh[p][ci] = sum d[co,kh,kw] w[co][kh][kw][ci]   (27 x C FMAs per pixel, weights via ds_read_b128)
// dw[co][kh][kw][ci] += sum_p d[p][co,kh,kw] h[p][ci]: a (27 -> 32) x C x 256-pixel contraction per
// workgroup on the f32 matrix cores (each wave takes 64 of the 256 pixels), reduced through LDS
template <typename T, int C>
__global__ __launch_bounds__(256) void img_head_bwd_kernel(const T* __restrict__ h, const float* __restrict__ w,
                                                           const float* __restrict__ img,
                                                           const float* __restrict__ dimg, T* __restrict__ dh,
                                                           float* __restrict__ dw, int N, int H, int W,
                                                           int accumulate, float* __restrict__ det_part) {
    constexpr int V = Vec16<T>::N;
    // row strides ODD: the per-pixel scalar writes of a wave (lane = pixel, stride = row) then spread over all 64 banks
    // (strides of 32 floats put the 64 lanes on two banks: 32 + C 32-way conflicting ds_write_b32 per lane), and the
    // MFMA operand reads (32 consecutive floats of one row per half-wave, rows one apart) overlap on a single bank
    constexpr int HS = C + 1, DS = 33;
    constexpr int NTL = C / 32;                  // 32-channel tiles of dw
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_w = sm;                 // [27][C]
    float* s_h = s_w + 27 * C;       // [256][HS]
    float* s_d = s_h + 256 * HS;     // [256][DS]   columns 27..31 zero
    float* s_r = s_d + 256 * DS;     // [32][C]     dw partial sums of the workgroup
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 27 * C; i += 256) s_w[i] = w[i];
    const int64_t total = (int64_t)N * H * W;
    const int64_t p = blockIdx.x * (int64_t)256 + tid;
    const bool live = p < total;
    float d[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) d[i] = 0.f;
    if (live) {
        const int x = (int)(p % W), y = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int oy = y - kh + 1;
            const bool oky = oy >= 0 && oy < H;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ox = x - kw + 1;
                const bool ok = oky && ox >= 0 && ox < W;
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    const int64_t o = (((int64_t)n * 3 + co) * H + (oky ? oy : 0)) * W + ((ox >= 0 && ox < W) ? ox : 0);
                    const float t = img[o];
                    const float g = dimg[o];
                    d[(co * 3 + kh) * 3 + kw] = ok ? g * (1.f - t * t) : 0.f;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) s_d[tid * DS + i] = i < 27 ? d[i < 27 ? i : 0] : 0.f;
    __syncthreads();     // s_w ready
    const T* hp = h + p * C;
    T* op = dh + p * C;
#pragma unroll
    for (int cv = 0; cv < C / V; ++cv) {
        Vec16<T> hv, o, prev;
        if (live) hv = ld16(hp + cv * V);
        if (live && accumulate) prev = ld16(op + cv * V);
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            s_h[tid * HS + cv * V + k] = live ? hv.get(k) : 0.f;
            acc[k] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 27; ++i) {
            const float4* wr = reinterpret_cast<const float4*>(&s_w[i * C + cv * V]);
#pragma unroll
            for (int k4 = 0; k4 < V / 4; ++k4) {
                const float4 wv = wr[k4];
                acc[4 * k4] += d[i] * wv.x; acc[4 * k4 + 1] += d[i] * wv.y;
                acc[4 * k4 + 2] += d[i] * wv.z; acc[4 * k4 + 3] += d[i] * wv.w;
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) o.set(k, acc[k] + ((live && accumulate) ? prev.get(k) : 0.f));
        if (live) st16(op + cv * V, o);
    }
    __syncthreads();
    // dw tile(s): rows = i (27 of 32), columns = ci; this wave contracts its own 64 pixels; the four waves' tiles
    // are then added wave by wave (a fixed order: no LDS atomics)
    {
        const int rl = lane & 31, hf = lane >> 5;
        const float* ap = s_d + (wid * 64 + hf) * DS + rl;
        f32x16_t acc[NTL];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
            const float* bp = s_h + (wid * 64 + hf) * HS + nt * 32 + rl;
#pragma unroll 8
            for (int kk = 0; kk < 32; ++kk)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk * DS], bp[2 * kk * HS], acc[nt], 0, 0, 0);
        }
#pragma unroll 1
        for (int wv = 0; wv < 4; ++wv) {
            if (wid == wv) {
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = (r & 3) + 8 * (r >> 2) + 4 * hf;
                        if (i < 27) {
                            float* q = &s_r[i * C + nt * 32 + rl];
                            *q = wv == 0 ? acc[nt][r] : *q + acc[nt][r];
                        }
                    }
            }
            __syncthreads();
        }
    }
    if (det_part) {         // deterministic mode: this workgroup's own slot; sba_det_fold adds the slots in order
        for (int o = tid; o < 27 * C; o += 256) det_part[(int64_t)blockIdx.x * 27 * C + o] = s_r[o];
        return;
    }
    for (int o = tid; o < 27 * C; o += 256) atomicAdd(&dw[o], s_r[o]);
}

// ------------------------------------------------------------------ discriminator stem
// conv4x4 s2 (3 -> C) + LeakyReLU(0.2) on the NCHW f32 image.  K = 48 is too thin for the implicit-GEMM
// tiles, so forward and data-gradient are VALU kernels organised so that every LDS access is a
// ds_read_b128 feeding 4 FMAs; the weight gradient (a 64-pixel-deep K loop per tile) runs on the f32
// matrix cores.
constexpr int STEM_CB = 32;          // output channels per thread (forward)

// forward: thread = (output pixel, 32-channel block); the 48 patch values live in registers, the
// block's weights in LDS as [48][32]
template <typename T>
__global__ __launch_bounds__(256) void d_stem_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                         T* __restrict__ out, int N, int S, int C) {
    constexpr int V = Vec16<T>::N;
    __shared__ __attribute__((aligned(16))) float s_w[48 * STEM_CB];
    const int c0 = blockIdx.y * STEM_CB;
    for (int i = threadIdx.x; i < 48 * STEM_CB; i += blockDim.x) {
        const int k = i / STEM_CB, cl = i - k * STEM_CB;       // w is [co][kh][kw][ci] = [co][48]
        s_w[i] = w[(c0 + cl) * 48 + k];
    }
    __syncthreads();
    const int O = S / 2;
    const int64_t total = (int64_t)N * O * O;
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= total) return;
    const int ox = (int)(p % O), oy = (int)((p / O) % O), n = (int)(p / ((int64_t)O * O));
    float pv[48];
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
        const int iy = 2 * oy + kh - 1;
        const bool oky = iy >= 0 && iy < S;
        const int iyc = oky ? iy : 0;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const int ix = 2 * ox + kw - 1;
            const bool ok = oky && ix >= 0 && ix < S;
            const int ixc = (ix >= 0 && ix < S) ? ix : 0;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = img[(((int64_t)n * 3 + ci) * S + iyc) * S + ixc];
                pv[(kh * 4 + kw) * 3 + ci] = ok ? v : 0.f;
            }
        }
    }
    float acc[STEM_CB];
#pragma unroll
    for (int c = 0; c < STEM_CB; ++c) acc[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 48; ++k) {
        const float4* wr = reinterpret_cast<const float4*>(&s_w[k * STEM_CB]);
#pragma unroll
        for (int c4 = 0; c4 < STEM_CB / 4; ++c4) {
            const float4 wv = wr[c4];
            acc[4 * c4] += pv[k] * wv.x; acc[4 * c4 + 1] += pv[k] * wv.y;
            acc[4 * c4 + 2] += pv[k] * wv.z; acc[4 * c4 + 3] += pv[k] * wv.w;
        }
    }
    T* op = out + p * C + c0;
#pragma unroll
    for (int cv = 0; cv < STEM_CB / V; ++cv) {
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float a = acc[cv * V + k];
            o.set(k, a > 0.f ? a : 0.2f * a);
        }
        st16(op + cv * V, o);
    }
}

// The same forward on the bf16 matrix cores (bf16 output): out^T[co][pix] = sum_k w[co][k] patch[pix][k], K = 48 = three
// 16-deep steps of v_mfma_f32_32x32x16_bf16 with the PIXEL axis as the MFMA's N (lane) axis -- lane (pixel, group g) loads
// the 8 patch values k = 16 s + 8 g .. + 8 of its pixel straight from the f32 image (no LDS), the weights are the A
// operands (registers, loaded once per wave).  Image values and weights are f32: both are split into hi + lo bf16 parts
// (3 MFMAs per step: hi*hi + hi*lo + lo*hi), so the products carry ~16 mantissa bits and the result, rounded to bf16 like
// the VALU kernel's, agrees with it to that rounding.  The VALU kernel above spends 48 x C FMAs per pixel on the vector
// pipe (4 GFLOP at 256 px, 2B images: 90 us); here the 115 MB of HBM traffic are what is left.
template <int MT>
__global__ __launch_bounds__(256) void d_stem_fwd_mfma_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                              bf16_t* __restrict__ out, int N, int S, int tiles_per_wave) {
    constexpr int C = 32 * MT;
    constexpr int ROWB = C * 2 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char s_out[4 * 32 * ROWB];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pl = lane & 31, kg = lane >> 5;
    bf16x8_t a_hi[MT][3], a_lo[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            bf16x8_t hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // contraction index of step s: k = (ci = s, kh = 2 kg + (j >> 2), kw = j & 3) -- any order will do as long as
                // both operands use it; this one makes a lane's 8 patch values two runs of 4 consecutive image floats
                const float v = w[(32 * mt + pl) * 48 + ((2 * kg + (j >> 2)) * 4 + (j & 3)) * 3 + s];
                const bf16_t h = f2bf(v);
                hi[j] = (short)h;
                lo[j] = (short)f2bf(v - bf2f(h));
            }
            a_hi[mt][s] = hi;
            a_lo[mt][s] = lo;
        }
    const int O = S / 2;
    const int64_t total = (int64_t)N * O * O;
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wid) * tiles_per_wave;
    // the 24 image floats of lane (pixel, kg) for one tile; the NEXT tile's are fetched before the current tile's MFMAs
    // (a load -> wait -> compute -> store loop pays the HBM latency once per tile)
    float vn[3][8];
    auto fetch = [&](const int64_t p0) __attribute__((always_inline)) {
        const int64_t p = p0 + pl;
        const bool live = p < total;
        const int64_t pc = live ? p : 0;
        // (32-bit: the host checks N * O * O < 2^31 -- a 64-bit division is ~200 instructions)
        const uint32_t pq = (uint32_t)pc / (uint32_t)O;
        const int ox = (int)((uint32_t)pc - pq * (uint32_t)O), n = (int)(pq / (uint32_t)O), oy = (int)(pq - (uint32_t)n * (uint32_t)O);
        const float* ib = img + (int64_t)n * 3 * S * S;
        const int ix0 = 2 * ox - 1;
        const bool inner = ix0 >= 0 && ix0 + 3 < S;           // all four columns inside the image: one 16-byte load per row
#pragma unroll
        for (int s = 0; s < 3; ++s) {                         // s = input channel
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int iy = 2 * oy + 2 * kg + hh - 1;
                const bool rok = live && iy >= 0 && iy < S;
                const float* rp = ib + ((int64_t)s * S + (rok ? iy : 0)) * S;
                if (rok && inner) {
                    struct __attribute__((packed, aligned(4))) F4 { float f[4]; };
                    const F4 q = *reinterpret_cast<const F4*>(rp + ix0);
                    vn[s][4 * hh] = q.f[0]; vn[s][4 * hh + 1] = q.f[1]; vn[s][4 * hh + 2] = q.f[2]; vn[s][4 * hh + 3] = q.f[3];
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ix = ix0 + u;
                        vn[s][4 * hh + u] = (rok && ix >= 0 && ix < S) ? rp[ix] : 0.f;
                    }
                }
            }
        }
    };
    if (tile0 * 32 < total) fetch(tile0 * 32);
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int64_t p0 = (tile0 + t) * 32;
        if (p0 >= total) break;                               // (wave-uniform)
        bf16x8_t b_hi[3], b_lo[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            bf16x8_t hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16_t h = f2bf(vn[s][j]);
                hi[j] = (short)h;
                lo[j] = (short)f2bf(vn[s][j] - bf2f(h));
            }
            b_hi[s] = hi;
            b_lo[s] = lo;
        }
        if (t + 1 < tiles_per_wave && p0 + 32 < total) fetch(p0 + 32);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[mt][s], b_hi[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[mt][s], b_lo[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[mt][s], b_hi[s], acc, 0, 0, 0);
            }
            // this lane holds channels 32 mt + 8 g4 + 4 kg .. + 4 of its pixel: through the wave's LDS tile [pixel][C]
            // (rows padded by 16 B) so that the global stores are whole 16-byte row pieces, 1 KB per instruction
            unsigned char* row = s_out + wid * (32 * ROWB) + pl * ROWB + (32 * mt + 4 * kg) * 2;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const float a = acc[4 * g4 + u]; v[u] = a > 0.f ? a : 0.2f * a; }
                uint2 o;
                o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(row + 16 * g4) = o;
            }
        }
        // (wave-private tile: a wave's LDS accesses complete in order, no barrier needed)
        constexpr int CPR = C / 8;                                // 16-byte pieces per pixel row
#pragma unroll
        for (int i = 0; i < 32 * CPR / 64; ++i) {
            const int piece = lane + 64 * i, r = piece / CPR, cc = piece - r * CPR;
            if (p0 + r < total) {
                const uint4 v = *reinterpret_cast<const uint4*>(s_out + wid * (32 * ROWB) + r * ROWB + cc * 16);
                *reinterpret_cast<uint4*>(out + (p0 + r) * C + cc * 8) = v;
            }
        }
    }
}

// dw[co][k] += sum_pix dpre[pix][co] * patch[pix][k] on the f32 matrix cores: per 64-pixel tile the
// workgroup stages dpre [64][C] and the patches [64][48 (+16 zero)] in LDS (row strides = 32 mod 64
// floats, so the two half-waves of an MFMA operand read hit disjoint banks) and each wave accumulates
// its 32(co) x 32(k) tiles over the 64 pixels with 32 v_mfma_f32_32x32x2_f32.
template <typename T>
__global__ __launch_bounds__(256) void d_stem_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ out,
                                                           const T* __restrict__ dout, float* __restrict__ dw, int N,
                                                           int S, int C, int tiles_per_block, int det) {
    constexpr int V = Vec16<T>::N;
    constexpr int PS = 96;                       // patch row stride (64 columns used, 48 real)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int DS = C + (C % 64 == 0 ? 32 : 0);   // C is a multiple of 32 -> DS = 32 mod 64
    float* s_d = sm;                             // [64][DS]
    float* s_p = s_d + 64 * DS;                  // [64][PS]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int O = S / 2;
    const int64_t total = (int64_t)N * O * O;
    const int ntile = (C / 32) * 2;              // (co tile, k tile) pairs, <= 16
    f32x16_t acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int i = tid; i < 64 * PS; i += 256) s_p[i] = 0.f;       // padding columns stay zero
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t p0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * 64;
        if (p0 >= total) break;
        __syncthreads();
        const int cvn = C / V;
        for (int i = tid; i < 64 * cvn; i += 256) {
            const int t = i / cvn, c = (i - t * cvn) * V;
            const int64_t pp = p0 + t;
            Vec16<T> ov, dv;
            const bool live = pp < total;
            if (live) { ov = ld16(out + pp * C + c); dv = ld16(dout + pp * C + c); }
#pragma unroll
            for (int k = 0; k < V; ++k)
                s_d[t * DS + c + k] = live ? dv.get(k) * (ov.get(k) > 0.f ? 1.f : 0.2f) : 0.f;
        }
        for (int i = tid; i < 64 * 48; i += 256) {
            const int t = i / 48, k = i - t * 48;
            const int64_t pp = p0 + t;
            float v = 0.f;
            if (pp < total) {
                const int ox = (int)(pp % O), oy = (int)((pp / O) % O), n = (int)(pp / ((int64_t)O * O));
                const int ci = k % 3, kw = (k / 3) % 4, kh = k / 12;
                const int iy = 2 * oy + kh - 1, ix = 2 * ox + kw - 1;
                if (iy >= 0 && iy < S && ix >= 0 && ix < S) v = img[(((int64_t)n * 3 + ci) * S + iy) * S + ix];
            }
            s_p[t * PS + k] = v;
        }
        __syncthreads();
        const int rl = lane & 31, hf = lane >> 5;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int tl = wid + 4 * a;
            if (tl < ntile) {
                const int mt = tl >> 1, nt = tl & 1;
                const float* ap = s_d + hf * DS + mt * 32 + rl;
                const float* bp = s_p + hf * PS + nt * 32 + rl;
#pragma unroll 8
                for (int kk = 0; kk < 32; ++kk)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk * DS], bp[2 * kk * PS], acc[a], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int tl = wid + 4 * a;
        if (tl < ntile) {
            const int mt = tl >> 1, nt = tl & 1;
            const int k = nt * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                // deterministic mode: dw = the scratch ring, one [C][48] slot per workgroup, folded in order afterwards
                if (k < 48) {
                    if (det) dw[(int64_t)blockIdx.x * C * 48 + co * 48 + k] = acc[a][r];
                    else atomicAdd(&dw[co * 48 + k], acc[a][r]);
                }
            }
        }
    }
}

// dimg[n][ci][iy][ix] = sum over the (<=4) output pixels reading it, all co.  Workgroup = 16x16 input
// pixels: the 10x10 output pixels they touch are staged once in LDS as dpre = dout * lrelu'(out) (f32,
// zero outside the map, so the tap loop needs no bounds checks), instead of every thread pulling its
// four 2*C-element rows through L1 (4x redundant).  Weights in LDS as [48][C] (co contiguous): per 4
// channels of dpre one ds_read_b128 + three weight ds_read_b128 feed 12 FMAs.
template <typename T>
__global__ __launch_bounds__(256) void d_stem_dgrad_kernel(const float* __restrict__ w, const T* __restrict__ out,
                                                           const T* __restrict__ dout, float* __restrict__ dimg,
                                                           int N, int S, int C) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int DSd = C + 4;
    float* s_w = sm;                     // [48][C]
    float* s_d = sm + 48 * C;            // [100][C + 4]
    const int tid = threadIdx.x;
    for (int i = tid; i < 48 * C; i += 256) {
        const int co = i / 48, k = i - co * 48;
        s_w[k * C + co] = w[i];
    }
    const int O = S / 2;
    const int tiles_x = (S + 15) / 16;
    const int n = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * 16, tx0 = (blockIdx.x % tiles_x) * 16;
    const int oy0 = ty0 / 2 - 1, ox0 = tx0 / 2 - 1;
    const int cvn = C / V;
    for (int i = tid; i < 100 * cvn; i += 256) {
        const int t = i / cvn, c = (i - t * cvn) * V;
        const int oy = oy0 + t / 10, ox = ox0 + t % 10;
        const bool ok = oy >= 0 && oy < O && ox >= 0 && ox < O;
        Vec16<T> ov, dv;
        if (ok) {
            const int64_t q = (((int64_t)n * O + oy) * O + ox) * C + c;
            ov = ld16(out + q);
            dv = ld16(dout + q);
        }
#pragma unroll
        for (int k = 0; k < V; ++k)
            s_d[t * DSd + c + k] = ok ? dv.get(k) * (ov.get(k) > 0.f ? 1.f : 0.2f) : 0.f;
    }
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15;
    const int iy = ty0 + ly, ix = tx0 + lx;
    if (iy >= S || ix >= S) return;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kh = ((iy + 1) & 1) + 2 * j;
        const int ry = ((iy + 1 - kh) >> 1) - oy0;          // row inside the staged 10x10 patch (0..9)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int kw = ((ix + 1) & 1) + 2 * i2;
            const int rx = ((ix + 1 - kw) >> 1) - ox0;
            const float4* dp = reinterpret_cast<const float4*>(s_d + (ry * 10 + rx) * DSd);
            const float* wt = s_w + (kh * 4 + kw) * 3 * C;
            for (int c4 = 0; c4 < C / 4; ++c4) {
                const float4 dv = dp[c4];
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float4 wv = reinterpret_cast<const float4*>(wt + ci * C)[c4];
                    acc[ci] += dv.x * wv.x + dv.y * wv.y + dv.z * wv.z + dv.w * wv.w;
                }
            }
        }
    }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) dimg[(((int64_t)n * 3 + ci) * S + iy) * S + ix] = acc[ci];
}

// fragment of columns [c32, c32 + 32) over the 16 pixel rows of a slice whose rows are 192 B apart: 8 consecutive
// pixels per lane via ds_read_b64_tr_b16 (same addressing as WgFrag<bf16_t>::load of igemm.hip)
__device__ __forceinline__ bf16x8_t stem_tr_frag(const unsigned char* slice, const int c32, const int lane) {
    constexpr int ROWS = 192;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int cbase = c32 + 16 * (g16 & 1), kbase = 8 * (g16 >> 1);
    const int q = i16 >> 2, p = i16 & 3;
    const unsigned char* a0 = slice + (kbase + q) * ROWS + (cbase + 4 * p) * 2;
    typedef __attribute__((address_space(3))) s16x4_t* lptr;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * ROWS));
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// ------------------------------------------------------------------ stem backward on the bf16 matrix cores
// (bf16 activations, C = 64: every discriminator of the 3-stage model).  f32 quantities -- dpre = dout * lrelu'(out),
// the weights, the image -- enter the MFMAs as hi + lo bf16 parts (hi*hi + hi*lo + lo*hi, as in d_stem_fwd_mfma_kernel):
// the products carry ~16 mantissa bits, the sums are f32.
__device__ __forceinline__ void split_hi_lo(const float v, short& hi, short& lo) {
    const bf16_t h = f2bf(v);
    hi = (short)h;
    lo = (short)f2bf(v - bf2f(h));
}
// dpre of 8 channels (one 16-byte chunk of dout / out) as hi / lo fragments
__device__ __forceinline__ void dpre_hi_lo(const uint4 dv, const uint4 ov, uint4& hi, uint4& lo) {
    const uint32_t d[4] = {dv.x, dv.y, dv.z, dv.w}, o[4] = {ov.x, ov.y, ov.z, ov.w};
    uint32_t h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float d0 = __uint_as_float(d[q] << 16), d1 = __uint_as_float(d[q] & 0xffff0000u);
        const float o0 = __uint_as_float(o[q] << 16), o1 = __uint_as_float(o[q] & 0xffff0000u);
        short h0, l0, h1, l1;
        split_hi_lo(o0 > 0.f ? d0 : 0.2f * d0, h0, l0);
        split_hi_lo(o1 > 0.f ? d1 : 0.2f * d1, h1, l1);
        h[q] = (uint32_t)(uint16_t)h0 | ((uint32_t)(uint16_t)h1 << 16);
        l[q] = (uint32_t)(uint16_t)l0 | ((uint32_t)(uint16_t)l1 << 16);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// DATA GRADIENT.  The 2 x 2 block of image pixels (2 by + a, 2 bx + b) is read by the 3 x 3 neighbourhood of output
// positions around (by, bx):
//   dimg[ci][2 by + a][2 bx + b] = sum_{dy, dx = -1..1} sum_co dpre[by + dy][bx + dx][co] * w[co][kh][kw][ci],
//   kh = a + 1 - 2 dy, kw = b + 1 - 2 dx  (no term unless both lie in 0..3)
// -- a 3 x 3 stride-1 convolution of dpre with 12 "channels" (ci, a, b), K = 9 * 64 = 576 = 18 steps of
// v_mfma_f32_16x16x32_bf16: the (zero-filled) weights are the A operand, 16 rows (12 used), kept in registers for the
// whole launch (persistent workgroups); 16 consecutive positions of one map row are the B columns.  A workgroup stages
// the (8 + 2) x (16 + 2) positions of its tile ONCE, as hi / lo bf16 rows of 144 B (conflict-free ds_read_b128); the
// accumulator of lane (position, ci) is exactly the 2 x 2 pixel block of channel ci: two 8-byte stores, 128 B per
// 16 lanes.  (The VALU kernel above: 27 x 64 FMAs per pixel on the vector pipe, 154 us at 256 px.)
__global__ __launch_bounds__(256, 2) void d_stem_dgrad_mfma_kernel(const float* __restrict__ w,
                                                                   const bf16_t* __restrict__ out,
                                                                   const bf16_t* __restrict__ dout,
                                                                   float* __restrict__ dimg, const int N, const int S,
                                                                   const int ntiles) {
    constexpr int C = 64, TH = 8, TW = 16, HR = TH + 2, HC = TW + 2, ROWB = 2 * C + 16, NCH = HR * HC * 8;
    constexpr int NI = (NCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char s_hi[HR * HC * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char s_lo[HR * HC * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    bf16x8_t a_hi[18], a_lo[18];
    {
        const int ci = m >> 2, a = (m >> 1) & 1, b = m & 1;
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            const int tap = st >> 1, s2 = st & 1;
            const int kh = a + 1 - 2 * (tap / 3 - 1), kw = b + 1 - 2 * (tap % 3 - 1);
            const bool ok = m < 12 && kh >= 0 && kh < 4 && kw >= 0 && kw < 4;
            bf16x8_t hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int co = 32 * s2 + 8 * g + j;
                const float v = ok ? w[co * 48 + (kh * 4 + kw) * 3 + ci] : 0.f;
                short h, l;
                split_hi_lo(v, h, l);
                hi[j] = h;
                lo[j] = l;
            }
            a_hi[st] = hi;
            a_lo[st] = lo;
        }
    }
    const int O = S / 2, tiles_x = O / TW, tiles_y = O / TH;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int by0 = tyi * TH, bx0 = txi * TW;
        __syncthreads();                // the previous tile's fragment reads are done
        uint4 dv[NI], ov[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u, p = i >> 3, ch = i & 7;
            const int hy = p / HC, hx = p - hy * HC;
            const int oy = by0 + hy - 1, ox = bx0 + hx - 1;
            const bool ok = i < NCH && oy >= 0 && oy < O && ox >= 0 && ox < O;
            dv[u] = make_uint4(0, 0, 0, 0);
            ov[u] = make_uint4(0, 0, 0, 0);
            if (ok) {
                const int64_t q = (((int64_t)n * O + oy) * O + ox) * C + ch * 8;
                dv[u] = *reinterpret_cast<const uint4*>(dout + q);
                ov[u] = *reinterpret_cast<const uint4*>(out + q);
            }
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u, p = i >> 3, ch = i & 7;
            if (i < NCH) {
                uint4 hi, lo;
                dpre_hi_lo(dv[u], ov[u], hi, lo);
                *reinterpret_cast<uint4*>(s_hi + p * ROWB + ch * 16) = hi;
                *reinterpret_cast<uint4*>(s_lo + p * ROWB + ch * 16) = lo;
            }
        }
        __syncthreads();
        f32x4_t acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            const int tap = st >> 1, s2 = st & 1;
            const int ty = tap / 3, tx = tap % 3;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int p = (2 * wid + r + ty) * HC + m + tx;
                const bf16x8_t bh = *reinterpret_cast<const bf16x8_t*>(s_hi + p * ROWB + (4 * s2 + g) * 16);
                const bf16x8_t bl = *reinterpret_cast<const bf16x8_t*>(s_lo + p * ROWB + (4 * s2 + g) * 16);
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[st], bh, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[st], bl, acc[r], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[st], bh, acc[r], 0, 0, 0);
            }
        }
        // D layout: column = lane & 15 (position), rows 4 g .. 4 g + 3 = (ci = g; a, b)
        if (g < 3) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int by = by0 + 2 * wid + r, bx = bx0 + m;
                float* o0 = dimg + (((int64_t)n * 3 + g) * S + 2 * by) * S + 2 * bx;
                *reinterpret_cast<float2*>(o0) = make_float2(acc[r][0], acc[r][1]);
                *reinterpret_cast<float2*>(o0 + S) = make_float2(acc[r][2], acc[r][3]);
            }
        }
    }
}

// WEIGHT GRADIENT.  dw[co][k] += sum_pix dpre[pix][co] * patch[pix][k]: the contraction runs over PIXELS, both operands
// are pixel-major, so the fragments are read with ds_read_b64_tr_b16 (WgFrag of igemm.hip: 4 pixels x 16 columns
// transposed per 16-lane group).  Per 64-pixel tile the workgroup stages dpre [64][64] and the patches [64][48 (+16 zero)]
// as hi / lo bf16 rows of 192 B; wave (i, j) accumulates its 32 (co) x 32 (k') tile with 4 x 3 v_mfma_f32_32x32x16_bf16.
// Patch columns are kept as k' = ci * 16 + kh * 4 + kw (a thread's four consecutive image floats = one 8-byte LDS
// write); the epilogue maps them back to the weight layout [kh][kw][ci].  (Before: f32 operands and
// v_mfma_f32_32x32x2_f32 -- 1/8 of the bf16 rate -- with a scalar patch gather: 194 us at 256 px, 2B images.)
__global__ __launch_bounds__(256, 2) void d_stem_wgrad_mfma_kernel(const float* __restrict__ img,
                                                                   const bf16_t* __restrict__ out,
                                                                   const bf16_t* __restrict__ dout,
                                                                   float* __restrict__ dw, const int N, const int S,
                                                                   const int tiles_per_block, const int det) {
    constexpr int C = 64, ROWS = 192;
    __shared__ __attribute__((aligned(16))) unsigned char s_dh[64 * ROWS];
    __shared__ __attribute__((aligned(16))) unsigned char s_dl[64 * ROWS];
    __shared__ __attribute__((aligned(16))) unsigned char s_ph[64 * ROWS];
    __shared__ __attribute__((aligned(16))) unsigned char s_pl[64 * ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int it = wid >> 1, jt = wid & 1;
    const int O = S / 2;
    const int64_t total = (int64_t)N * O * O;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (tid < 128) {        // patch columns 48 .. 63 stay zero
        unsigned char* base = ((tid & 64) ? s_pl : s_ph) + (tid & 63) * ROWS + 96;
        *reinterpret_cast<uint4*>(base) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(base + 16) = make_uint4(0, 0, 0, 0);
    }
    // the next tile's global loads are issued before the current tile's MFMAs: their latency hides behind them
    uint4 dv[2], ov[2];
    float pv[3][4];
    auto fetch = [&](const int64_t p0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, t = i >> 3, ch = i & 7;
            const int64_t pp = p0 + t;
            dv[u] = make_uint4(0, 0, 0, 0);
            ov[u] = make_uint4(0, 0, 0, 0);
            if (pp < total) {
                dv[u] = *reinterpret_cast<const uint4*>(dout + pp * C + ch * 8);
                ov[u] = *reinterpret_cast<const uint4*>(out + pp * C + ch * 8);
            }
        }
        const int t = tid & 63, kh = wid;
        const int64_t pp = p0 + t;
        const bool live = pp < total;
        const int64_t pc = live ? pp : 0;
        const uint32_t pq = (uint32_t)pc / (uint32_t)O;          // (32-bit divisions: N * O * O < 2^31, checked by the host)
        const int ox = (int)((uint32_t)pc - pq * (uint32_t)O), n = (int)(pq / (uint32_t)O), oy = (int)(pq - (uint32_t)n * (uint32_t)O);
        const int iy = 2 * oy + kh - 1, ix0 = 2 * ox - 1;
        const bool rok = live && iy >= 0 && iy < S;
        const bool inner = ix0 >= 0 && ix0 + 3 < S;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const float* rp = img + (((int64_t)n * 3 + ci) * S + (rok ? iy : 0)) * S;
            if (rok && inner) {
                struct __attribute__((packed, aligned(4))) F4 { float f[4]; };
                const F4 q = *reinterpret_cast<const F4*>(rp + ix0);
                pv[ci][0] = q.f[0]; pv[ci][1] = q.f[1]; pv[ci][2] = q.f[2]; pv[ci][3] = q.f[3];
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ix = ix0 + u;
                    pv[ci][u] = (rok && ix >= 0 && ix < S) ? rp[ix] : 0.f;
                }
            }
        }
    };
    const int64_t pbase = (int64_t)blockIdx.x * tiles_per_block * 64;
    if (pbase < total) fetch(pbase);
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t p0 = pbase + (int64_t)tile * 64;
        if (p0 >= total) break;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, t = i >> 3, ch = i & 7;
            uint4 hi, lo;
            dpre_hi_lo(dv[u], ov[u], hi, lo);
            *reinterpret_cast<uint4*>(s_dh + t * ROWS + ch * 16) = hi;
            *reinterpret_cast<uint4*>(s_dl + t * ROWS + ch * 16) = lo;
        }
        {
            const int t = tid & 63, kh = wid;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                short h[4], l[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) split_hi_lo(pv[ci][u], h[u], l[u]);
                uint2 hv, lv;
                hv.x = (uint32_t)(uint16_t)h[0] | ((uint32_t)(uint16_t)h[1] << 16);
                hv.y = (uint32_t)(uint16_t)h[2] | ((uint32_t)(uint16_t)h[3] << 16);
                lv.x = (uint32_t)(uint16_t)l[0] | ((uint32_t)(uint16_t)l[1] << 16);
                lv.y = (uint32_t)(uint16_t)l[2] | ((uint32_t)(uint16_t)l[3] << 16);
                *reinterpret_cast<uint2*>(s_ph + t * ROWS + (ci * 16 + kh * 4) * 2) = hv;
                *reinterpret_cast<uint2*>(s_pl + t * ROWS + (ci * 16 + kh * 4) * 2) = lv;
            }
        }
        __syncthreads();
        if (tile + 1 < tiles_per_block && p0 + 64 < total) fetch(p0 + 64);
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            const bf16x8_t ah = stem_tr_frag(s_dh + 16 * k16 * ROWS, 32 * it, lane);
            const bf16x8_t al = stem_tr_frag(s_dl + 16 * k16 * ROWS, 32 * it, lane);
            const bf16x8_t bh = stem_tr_frag(s_ph + 16 * k16 * ROWS, 32 * jt, lane);
            const bf16x8_t bl = stem_tr_frag(s_pl + 16 * k16 * ROWS, 32 * jt, lane);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    const int kp = 32 * jt + (lane & 31);
    if (kp < 48) {
        const int ci = kp >> 4, kh = (kp >> 2) & 3, kw = kp & 3;
        const int k = (kh * 4 + kw) * 3 + ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            // deterministic mode: dw = the scratch ring, one [C][48] slot per workgroup, folded in order afterwards
            // det = 1: dw = a scratch ring (deterministic mode, or the default mode's two-stage reduction): one [C][48] slot
            // per workgroup, folded afterwards
            if (det) dw[(int64_t)blockIdx.x * C * 48 + co * 48 + k] = acc[r];
            else atomicAdd(&dw[co * 48 + k], acc[r]);
        }
    }
}

// ------------------------------------------------------------------ image head forward on the bf16 matrix cores
// (bf16 features, C = 32).  img^T[co][pix] = tanh(sum_tap sum_ci w[co][tap][ci] h[pix + tap][ci]): one
// v_mfma_f32_16x16x32_bf16 per tap -- K = 32 IS one tap's channels -- with 16 consecutive pixels of a map row as the B
// columns; the weights (f32: hi + lo parts, 2 MFMAs per tap) are the A rows (3 of 16 used) and stay in registers.
// Persistent workgroups own 8 x 32-pixel tiles and stage the (8 + 2) x (32 + 2) feature rows ONCE in LDS (80-byte rows:
// conflict-free ds_read_b128 for the nine shifted views); the next tile's global loads are issued before the current
// tile's MFMAs.  Lanes 0..15 hold the three channels of their pixel: 64-byte row stores into the NCHW image.
// (Reading the nine shifted rows straight from global memory instead was latency-bound at 1.2 TB/s: 9x the bytes in
// flight per pixel.  The VALU kernel above: 27 x 32 FMAs per pixel, 70 us at 256 px.)
// tanh with the hardware exponential and reciprocal (the library tanhf is ~90 instructions, executed for a whole wave
// while only the 16 lanes that hold an image row have data: more issue cycles than the tile's MFMAs): absolute error
// <= 2e-7, and the odd series below 0.1 keeps that RELATIVE accuracy for small arguments
__device__ __forceinline__ float tanh_fast(const float x) {
    const float e = __expf(2.f * x);
    const float big = 1.f - 2.f * __frcp_rn(e + 1.f);
    const float x2 = x * x;
    const float small = x * (1.f + x2 * (-0.33333334f + x2 * 0.13333334f));
    return fabsf(x) < 0.1f ? small : big;
}
__global__ __launch_bounds__(256, 2) void img_head_fwd_mfma_kernel(const bf16_t* __restrict__ h,
                                                                   const float* __restrict__ w,
                                                                   float* __restrict__ img, const int N, const int H,
                                                                   const int W, const int ntiles) {
    constexpr int C = 32, TH = 8, TW = 32, HC = TW + 2, ROWB = 80, NCH = (TH + 2) * HC * 4, NI = (NCH + 255) / 256;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    __shared__ __attribute__((aligned(16))) unsigned char s_h[(TH + 2) * HC * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    bf16x8_t a_hi[9], a_lo[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        bf16x8_t hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = m < 3 ? w[(m * 9 + tap) * C + 8 * g + j] : 0.f;
            short hh, ll;
            split_hi_lo(v, hh, ll);
            hi[j] = hh;
            lo[j] = ll;
        }
        a_hi[tap] = hi;
        a_lo[tap] = lo;
    }
    const int tiles_x = W / TW, tiles_y = H / TH;
    u32x4_t ph[NI];
    auto fetch = [&](const int tile) __attribute__((always_inline)) {
        const int txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int y0 = tyi * TH, x0 = txi * TW;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u, p = i >> 2, ch = i & 3;
            const int hy = p / HC, hx = p - hy * HC;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            ph[u] = u32x4_t{0u, 0u, 0u, 0u};
            if (i < NCH && y >= 0 && y < H && x >= 0 && x < W)
                ph[u] = *reinterpret_cast<const u32x4_t*>(h + (((int64_t)n * H + y) * W + x) * C + ch * 8);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int y0 = tyi * TH, x0 = txi * TW;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u, p = i >> 2, ch = i & 3;
            if (i < NCH) *reinterpret_cast<u32x4_t*>(s_h + p * ROWB + ch * 16) = ph[u];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yl = 2 * wid + (q >> 1), xl = 16 * (q & 1) + m;
            f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(s_h + ((yl + tap / 3) * HC + xl + tap % 3) * ROWB + g * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[tap], bf, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[tap], bf, acc, 0, 0, 0);
            }
            if (g == 0) {           // D rows 0..2 = co, column = pixel
                float* o = img + (((int64_t)n * 3) * H + y0 + yl) * W + x0 + xl;
                const int64_t cs = (int64_t)H * W;
                o[0] = tanh_fast(acc[0]);
                o[cs] = tanh_fast(acc[1]);
                o[2 * cs] = tanh_fast(acc[2]);
            }
        }
    }
}

// ------------------------------------------------------------------ image head backward on the bf16 matrix cores
// (bf16 activations, C = 32: GET_IMAGE_G of every stage, model.py:426-437).  With D[pix][k] = dpre[co][y - kh + 1][x - kw + 1],
// k = (co, kh, kw) (27 of 32 columns), dpre = dimg * (1 - img^2):
//   dh[pix][ci] = sum_k D[pix][k] w[k][ci]          -- K = 32: two v_mfma_f32_32x32x16_bf16 steps per 32 pixels
//   dw[k][ci]  += sum_pix D[pix][k] h[pix][ci]      -- K = pixels: D^T fragments = 8 consecutive floats of a dpre row,
//                                                      h fragments via ds_read_b64_tr_b16 from the staged bf16 tile
// D and w are f32 quantities: hi + lo bf16 parts (hi*hi + hi*lo + lo*hi); h is bf16 as stored.  A workgroup is
// persistent, owns 8 x 32-pixel tiles, stages dpre with a one-pixel halo (f32, 4 KB) and the h tile (16 KB), keeps its
// dw tile in registers over all its tiles and leaves ONE [27][32] partial sum per workgroup (scratch ring + fold, or
// f32 atomics).  The next tile's global loads are issued before the current tile's MFMAs.  (The VALU / f32-MFMA kernel
// above: 27 x 32 FMAs per pixel on the vector pipe + 106 KB of LDS, one workgroup per CU: 201 us at 256 px.)
__host__ __device__ constexpr int img_head_doff(const int k) {     // offset of D[.][k] inside the halo tile (rows of 36, 10 per co)
    return k < 27 ? (k / 9) * 360 + (2 - (k % 9) / 3) * 36 + (2 - k % 3) : -1;
}
__global__ __launch_bounds__(256, 2) void img_head_bwd_mfma_kernel(const bf16_t* __restrict__ h,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ img,
                                                                   const float* __restrict__ dimg,
                                                                   bf16_t* __restrict__ dh, float* __restrict__ dw,
                                                                   const int N, const int H, const int W,
                                                                   const int accumulate, const int ntiles,
                                                                   float* __restrict__ part) {
    constexpr int C = 32, TH = 8, TW = 32, HC = TW + 2, RS = 36, CS = (TH + 2) * RS, NDP = 3 * (TH + 2) * HC;
    constexpr int NU = (NDP + 255) / 256;
    __shared__ __attribute__((aligned(16))) float s_dp[3 * CS + 64];
    __shared__ __attribute__((aligned(16))) unsigned char s_h[TH * TW * 64];        // [pixel][32 ci] bf16; later: [4][27*32] f32
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l32 = lane & 31, g = lane >> 5;
    // dh: A = w^T (rows ci), B = D (columns = the 32 pixels of a tile row)
    bf16x8_t wa_hi[2], wa_lo[2];
    static_assert(RS == 36 && CS == 360, "img_head_doff");
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        bf16x8_t hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * st + 8 * g + j;
            const float v = k < 27 ? w[k * C + l32] : 0.f;
            short hh, ll;
            split_hi_lo(v, hh, ll);
            hi[j] = hh;
            lo[j] = ll;
        }
        wa_hi[st] = hi;
        wa_lo[st] = lo;
    }
    // dw: A = D^T (row k = l32: 8 consecutive pixels of one dpre row), B = h (column ci = l32, transposing read)
    const bool krow = l32 < 27;
    const int offk = krow ? (l32 / 9) * CS + (2 - (l32 % 9) / 3) * RS + (2 - l32 % 3) : 0;
    f32x16_t wacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
    for (int i = tid; i < 3 * CS + 64; i += 256) s_dp[i] = 0.f;

    const int tiles_x = W / TW, tiles_y = H / TH;
    float pt[NU], pg[NU];
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    u32x4_t ph[4];
    auto fetch = [&](const int tile) __attribute__((always_inline)) {
        const int txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int y0 = tyi * TH, x0 = txi * TW;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = tid + 256 * u;
            const int co = i / ((TH + 2) * HC), rem = i - co * ((TH + 2) * HC);
            const int hy = rem / HC, hx = rem - hy * HC;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            const bool ok = i < NDP && y >= 0 && y < H && x >= 0 && x < W;
            pt[u] = 0.f;
            pg[u] = 0.f;
            if (ok) {
                const int64_t o = (((int64_t)n * 3 + co) * H + y) * W + x;
                pt[u] = img[o];
                pg[u] = dimg[o];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + 256 * u, p = i >> 2, ch = i & 3;
            const int64_t pix = ((int64_t)n * H + y0 + (p >> 5)) * W + x0 + (p & 31);
            ph[u] = *reinterpret_cast<const u32x4_t*>(h + pix * C + ch * 8);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int y0 = tyi * TH, x0 = txi * TW;
        __syncthreads();            // the previous tile's reads are done (first tile: s_dp cleared)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = tid + 256 * u;
            if (i < NDP) {
                const int co = i / ((TH + 2) * HC), rem = i - co * ((TH + 2) * HC);
                const int hy = rem / HC, hx = rem - hy * HC;
                s_dp[co * CS + hy * RS + hx] = pg[u] * (1.f - pt[u] * pt[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4_t*>(s_h + (tid + 256 * u) * 16) = ph[u];
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int yl = 2 * wid + rr;
            // ---- dh of the 32 pixels of tile row yl
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* db = s_dp + yl * RS + l32;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8_t bh, bl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int oa = img_head_doff(16 * st + j), ob = img_head_doff(16 * st + 8 + j);     // (compile-time)
                    const int o = g ? ob : oa;
                    const float v = o >= 0 ? db[o >= 0 ? o : 0] : 0.f;
                    short hh, ll;
                    split_hi_lo(v, hh, ll);
                    bh[j] = hh;
                    bl[j] = ll;
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_lo[st], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_hi[st], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_hi[st], bh, acc, 0, 0, 0);
            }
            // D layout: column = l32 (pixel), rows (r & 3) + 8 (r >> 2) + 4 g (ci): 4 consecutive channels per r >> 2
            {
                const int64_t pix = ((int64_t)n * H + y0 + yl) * W + x0 + l32;
                bf16_t* op = dh + pix * C + 4 * g;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4] = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    if (accumulate) {
                        const uint2 pr = *reinterpret_cast<const uint2*>(op + 8 * q);
                        v[0] += __uint_as_float(pr.x << 16); v[1] += __uint_as_float(pr.x & 0xffff0000u);
                        v[2] += __uint_as_float(pr.y << 16); v[3] += __uint_as_float(pr.y & 0xffff0000u);
                    }
                    uint2 o;
                    o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                    o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                    *reinterpret_cast<uint2*>(op + 8 * q) = o;
                }
            }
            // ---- dw += D^T h over the same 32 pixels: two 16-pixel steps
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const float* ab = s_dp + offk + yl * RS + 16 * half + 8 * g;
                bf16x8_t ah, al;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = krow ? ab[j] : 0.f;
                    short hh, ll;
                    split_hi_lo(v, hh, ll);
                    ah[j] = hh;
                    al[j] = ll;
                }
                const unsigned char* slice = s_h + (yl * TW + 16 * half) * 64;
                const int g16 = lane >> 4, i16 = lane & 15;
                const unsigned char* a0 = slice + (8 * (g16 >> 1) + (i16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;
                typedef __attribute__((address_space(3))) s16x4_t* lptr;
                const s16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
                const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * 64));
                bf16x8_t hb;
                hb[0] = lo4[0]; hb[1] = lo4[1]; hb[2] = lo4[2]; hb[3] = lo4[3];
                hb[4] = hi4[0]; hb[5] = hi4[1]; hb[6] = hi4[2]; hb[7] = hi4[3];
                wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, hb, wacc, 0, 0, 0);
                wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, hb, wacc, 0, 0, 0);
            }
        }
    }
    // the four waves' dw tiles, added in wave order, then one [27][32] partial sum per workgroup
    __syncthreads();
    float* s_r = reinterpret_cast<float*>(s_h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * g;
        if (k < 27) s_r[wid * (27 * C) + k * C + l32] = wacc[r];
    }
    __syncthreads();
    for (int o = tid; o < 27 * C; o += 256) {
        const float v = ((s_r[o] + s_r[27 * C + o]) + s_r[2 * 27 * C + o]) + s_r[3 * 27 * C + o];
        if (part) part[(int64_t)blockIdx.x * 27 * C + o] = v;
        else atomicAdd(&dw[o], v);
    }
}

// ------------------------------------------------------------------ logits head
template <typename T>
__global__ __launch_bounds__(256) void logits_fwd_kernel(const T* __restrict__ h, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ prob,
                                                         int K) {
    constexpr int V = Vec16<T>::N;
    __shared__ float sh[16];
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int k = threadIdx.x * V; k < K; k += blockDim.x * V) {
        Vec16<T> v = ld16(h + (int64_t)b * K + k);
#pragma unroll
        for (int i = 0; i < V; ++i) acc += v.get(i) * w[k + i];
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) prob[b] = 1.f / (1.f + expf(-(acc + bias[0])));
}

// blockIdx.y = chunk of LB samples: the per-sample loop was a chain of dependent 16-byte loads
// (15 us for a 160 KB problem); LB independent samples per thread, weight gradient by f32 atomics
constexpr int LOGITS_LB = 4;
template <typename T>
__global__ __launch_bounds__(256) void logits_bwd_kernel(const T* __restrict__ h, const float* __restrict__ w,
                                                         const float* __restrict__ prob,
                                                         const float* __restrict__ dprob, T* __restrict__ dh,
                                                         float* __restrict__ dw, float* __restrict__ dbias, int B,
                                                         int K, int accumulate, int chunk0) {
    constexpr int V = Vec16<T>::N;
    const int b0 = (chunk0 + blockIdx.y) * LOGITS_LB;
    float dl[LOGITS_LB];
#pragma unroll
    for (int j = 0; j < LOGITS_LB; ++j) {
        const int b = b0 + j;
        const float p = b < B ? prob[b] : 0.f;
        dl[j] = b < B ? dprob[b] * p * (1.f - p) : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && dbias) {
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < LOGITS_LB; ++j) sacc += dl[j];
        atomicAdd(dbias, sacc);
    }
    const int k = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (k >= K) return;
    float wv[V], gw[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { wv[i] = w[k + i]; gw[i] = 0.f; }
    Vec16<T> hv[LOGITS_LB], ov[LOGITS_LB];
#pragma unroll
    for (int j = 0; j < LOGITS_LB; ++j) {
        const int b = b0 + j < B ? b0 + j : b0;
        hv[j] = ld16(h + (int64_t)b * K + k);
        if (accumulate) ov[j] = ld16(dh + (int64_t)b * K + k);
    }
#pragma unroll
    for (int j = 0; j < LOGITS_LB; ++j) {
        if (b0 + j >= B) break;
        Vec16<T> o;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            gw[i] += dl[j] * hv[j].get(i);
            o.set(i, dl[j] * wv[i] + (accumulate ? ov[j].get(i) : 0.f));
        }
        st16(dh + (int64_t)(b0 + j) * K + k, o);
    }
    if (dw) {
#pragma unroll
        for (int i = 0; i < V; ++i) atomicAdd(&dw[k + i], gw[i]);
    }
}

// ------------------------------------------------------------------ conditioning concat
template <typename T>
__global__ void cond_cat_fwd_kernel(const T* __restrict__ h, const float* __restrict__ sent, T* __restrict__ out,
                                    int B, int C, int E) {
    const int CE = C + E;
    const int64_t total = (int64_t)B * 16 * CE;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CE);
        const int64_t row = i / CE;          // b*16 + s
        out[i] = c < C ? h[row * C + c] : from_f<T>(sent[(row / 16) * E + (c - C)]);
    }
}

template <typename T>
__global__ void cond_cat_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dh, float* __restrict__ dsent,
                                    int B, int C, int E, int accumulate) {
    const int CE = C + E;
    const int64_t nh = (int64_t)B * 16 * C;
    const int64_t total = nh + (dsent ? (int64_t)B * E : 0);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nh) {
            const int c = (int)(i % C);
            const int64_t row = i / C;
            float v = to_f<T>(dout[row * CE + c]);
            if (accumulate) v += to_f<T>(dh[i]);
            dh[i] = from_f<T>(v);
        } else {
            const int64_t j = i - nh;
            const int e = (int)(j % E), b = (int)(j / E);
            float s = 0.f;
            for (int sp = 0; sp < 16; ++sp) s += to_f<T>(dout[((int64_t)b * 16 + sp) * CE + C + e]);
            dsent[j] += s;
        }
    }
}

inline int grid_for(int64_t items, int cap = 8192) {
    int64_t b = (items + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

template <typename K> void set_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

#define CH_SWITCH(C_, CALL)                              \
    switch (C_) {                                        \
        case 32: { constexpr int C = 32; CALL; } break;  \
        case 64: { constexpr int C = 64; CALL; } break;  \
        case 128: { constexpr int C = 128; CALL; } break;\
        default: return SBA_E_ARG;                       \
    }

extern "C" int sba_img_head_fwd(int dtype, const void* h, const float* w, float* img, int N, int H, int W, int C_,
                                void* stream) {
    if (!h || !w || !img || N <= 0 || H <= 0 || W <= 0) return SBA_E_ARG;
    const int64_t total = (int64_t)N * H * W;
    if (total > 0x7fffffffLL * 64) return SBA_E_ARG;
    static int mfma = -1;       // SBA_HEAD_MFMA=0: the VALU kernel for bf16 as well (A/B aid)
    if (mfma < 0) { const char* e = getenv("SBA_HEAD_MFMA"); mfma = (e && e[0] == '0') ? 0 : 1; }
    if (dtype == SBA_BF16 && mfma && C_ == 32 && W % 32 == 0 && H % 8 == 0 && total / 256 <= 0x7fffffff) {
        const int ntiles = (int)(total / 256);
        const int blocks = ntiles < 1024 ? ntiles : 1024;       // persistent: four workgroups per CU
        SBA_LAUNCH(img_head_fwd_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, w, img,
                   N, H, W, ntiles);
        return SBA_CHECK_LAUNCH();
    }
    SBA_DISPATCH(dtype, CH_SWITCH(C_, SBA_LAUNCH((img_head_fwd_kernel<T, C>), dim3(cdiv(total, 256)),
                                                        dim3(256), 0, (hipStream_t)stream, (const T*)h, w, img, N,
                                                        H, W)));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_img_head_bwd(int dtype, const void* h, const float* w, const float* img, const float* dimg,
                                void* dh, float* dw, int N, int H, int W, int C_, int accumulate, void* stream) {
    if (!h || !w || !img || !dimg || !dh || !dw || N <= 0 || H <= 0 || W <= 0) return SBA_E_ARG;
    if (C_ > 64) return SBA_E_ARG;       // LDS budget of the fused wgrad contraction
    const int64_t total = (int64_t)N * H * W;
    static int mfma = -1;       // SBA_HEAD_MFMA=0: the VALU / f32-MFMA kernel for bf16 as well (A/B aid)
    if (mfma < 0) { const char* e = getenv("SBA_HEAD_MFMA"); mfma = (e && e[0] == '0') ? 0 : 1; }
    if (dtype == SBA_BF16 && mfma && C_ == 32 && W % 32 == 0 && H % 8 == 0 && total / 256 <= 0x7fffffff) {
        const int ntiles = (int)(total / 256);
        const int blocks = ntiles < 1024 ? ntiles : 1024;       // persistent: four workgroups per CU
        hipStream_t st = (hipStream_t)stream;
        float* part = nullptr;
        const bool det = sba_det_on();
        if (det) { part = sba_det_alloc((int64_t)blocks * 27 * 32); if (!part) return SBA_E_ARG; }
        else if (blocks > 64) part = sba_reduce_alloc((int64_t)blocks * 27 * 32);      // (none: atomics)
        SBA_LAUNCH(img_head_bwd_mfma_kernel, dim3(blocks), dim3(256), 0, st, (const bf16_t*)h, w, img, dimg,
                   (bf16_t*)dh, dw, N, H, W, accumulate, ntiles, part);
        if (part && det) sba_det_fold(part, 1, blocks, 27 * 32, dw, 0, 0, st);
        else if (part) sba_fold_add(part, blocks, 27 * 32, dw, st);
        return SBA_CHECK_LAUNCH();
    }
    SBA_DISPATCH(dtype, CH_SWITCH(C_, {
        const size_t sh = sizeof(float) * (27 * C + 256 * (C + 1) + 256 * 33 + 32 * C);
        set_lds(img_head_bwd_kernel<T, C>, sh);
        const int blocks = cdiv(total, 256);
        float* part = nullptr;
        if (sba_det_on()) { part = sba_det_alloc((int64_t)blocks * 27 * C); if (!part) return SBA_E_ARG; }
        SBA_LAUNCH((img_head_bwd_kernel<T, C>), dim3(blocks), dim3(256), sh, (hipStream_t)stream,
                           (const T*)h, w, img, dimg, (T*)dh, dw, N, H, W, accumulate, part);
        if (part) sba_det_fold(part, 1, blocks, 27 * C, dw, 0, 0, (hipStream_t)stream);
    }));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_d_stem_fwd(int dtype, const float* img, const float* w, void* out, int N, int S, int C,
                              void* stream) {
    if (!img || !w || !out || N <= 0 || S <= 0 || S % 2 || C <= 0 || C % STEM_CB || C > 256) return SBA_E_ARG;
    const int64_t pix = (int64_t)N * (S / 2) * (S / 2);
    if ((pix + 255) / 256 > 0x7fffffff) return SBA_E_ARG;
    static int mfma = -1;       // SBA_STEM_MFMA=0: the VALU kernel for bf16 as well (A/B aid)
    if (mfma < 0) { const char* e = getenv("SBA_STEM_MFMA"); mfma = (e && e[0] == '0') ? 0 : 1; }
    if (dtype == SBA_BF16 && mfma && (C == 32 || C == 64 || C == 128) && pix < 0x7fffffff) {
        const int64_t tiles = (pix + 31) / 32;
        const int tpw = tiles >= 16384 ? 4 : (tiles >= 4096 ? 2 : 1);
        const dim3 grid((unsigned)((tiles + 4 * tpw - 1) / (4 * tpw)));
        hipStream_t st = (hipStream_t)stream;
        if (C == 32) SBA_LAUNCH((d_stem_fwd_mfma_kernel<1>), grid, dim3(256), 0, st, img, w, (bf16_t*)out, N, S, tpw);
        else if (C == 64) SBA_LAUNCH((d_stem_fwd_mfma_kernel<2>), grid, dim3(256), 0, st, img, w, (bf16_t*)out, N, S, tpw);
        else SBA_LAUNCH((d_stem_fwd_mfma_kernel<4>), grid, dim3(256), 0, st, img, w, (bf16_t*)out, N, S, tpw);
        return SBA_CHECK_LAUNCH();
    }
    SBA_DISPATCH(dtype, {
        SBA_LAUNCH((d_stem_fwd_kernel<T>), dim3((unsigned)((pix + 255) / 256), C / STEM_CB), dim3(256), 0,
                           (hipStream_t)stream, img, w, (T*)out, N, S, C);
    });
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_d_stem_bwd(int dtype, const float* img, const float* w, const void* out, const void* dout,
                              float* dimg, float* dw, int N, int S, int C, void* stream) {
    if (!img || !w || !out || !dout || N <= 0 || S <= 0 || S % 2 || C <= 0 || C % 32 || C > 256) return SBA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    static int mfma = -1;       // SBA_STEM_MFMA=0: the VALU / f32-MFMA kernels for bf16 as well (A/B aid)
    if (mfma < 0) { const char* e = getenv("SBA_STEM_MFMA"); mfma = (e && e[0] == '0') ? 0 : 1; }
    const bool mm = dtype == SBA_BF16 && mfma && C == 64 && (int64_t)N * (S / 2) * (S / 2) < 0x7fffffff;
    if (dw && mm) {
        const int64_t pix = (int64_t)N * (S / 2) * (S / 2);
        const int64_t tiles = (pix + 63) / 64;
        int tpb = (int)((tiles + 767) / 768);       // three workgroups per CU (48 KB of LDS each)
        if (tpb < 1) tpb = 1;
        const int blocks = (int)((tiles + tpb - 1) / tpb);
        float* part = nullptr;
        const bool det = sba_det_on();
        if (det) { part = sba_det_alloc((int64_t)blocks * C * 48); if (!part) return SBA_E_ARG; }
        else if (blocks > 64) part = sba_reduce_alloc((int64_t)blocks * C * 48);    // (none: atomics)
        SBA_LAUNCH(d_stem_wgrad_mfma_kernel, dim3(blocks), dim3(256), 0, st, img, (const bf16_t*)out,
                   (const bf16_t*)dout, part ? part : dw, N, S, tpb, part ? 1 : 0);
        if (part && det) sba_det_fold(part, 1, blocks, (int64_t)C * 48, dw, 0, 0, st);
        else if (part) sba_fold_add(part, blocks, (int64_t)C * 48, dw, st);
        dw = nullptr;
    }
    if (dimg && mm && (S / 2) % 16 == 0) {
        const int64_t ntiles = (int64_t)N * (S / 2 / 16) * (S / 2 / 8);
        if (ntiles > 0x7fffffff) return SBA_E_ARG;
        const int blocks = ntiles < 512 ? (int)ntiles : 512;       // persistent: two workgroups per CU
        SBA_LAUNCH(d_stem_dgrad_mfma_kernel, dim3(blocks), dim3(256), 0, st, w, (const bf16_t*)out,
                   (const bf16_t*)dout, dimg, N, S, (int)ntiles);
        dimg = nullptr;
    }
    if (dw) {
        const int64_t pix = (int64_t)N * (S / 2) * (S / 2);
        const int64_t tiles = (pix + 63) / 64;
        int tpb = (int)((tiles + 767) / 768);      // three workgroups per CU (LDS 49 KB, 138 registers): staging is not
                                                   // software-pipelined, the co-resident workgroups hide it
        if (tpb < 1) tpb = 1;
        const int blocks = (int)((tiles + tpb - 1) / tpb);
        const size_t sh = sizeof(float) * (64 * (C + (C % 64 == 0 ? 32 : 0)) + 64 * 96);
        float* part = nullptr;
        if (sba_det_on()) { part = sba_det_alloc((int64_t)blocks * C * 48); if (!part) return SBA_E_ARG; }
        SBA_DISPATCH(dtype, {
            set_lds(d_stem_wgrad_kernel<T>, sh);
            SBA_LAUNCH((d_stem_wgrad_kernel<T>), dim3(blocks), dim3(256), sh, st, img, (const T*)out,
                               (const T*)dout, part ? part : dw, N, S, C, tpb, part ? 1 : 0);
        });
        if (part) sba_det_fold(part, 1, blocks, (int64_t)C * 48, dw, 0, 0, st);
    }
    if (dimg) {
        const size_t sh = sizeof(float) * (48 * C + 100 * (C + 4));
        const int tiles = ((S + 15) / 16) * ((S + 15) / 16);
        if (N > 65535) return SBA_E_ARG;
        SBA_DISPATCH(dtype, {
            set_lds(d_stem_dgrad_kernel<T>, sh);
            SBA_LAUNCH((d_stem_dgrad_kernel<T>), dim3(tiles, N), dim3(256), sh, st, w, (const T*)out,
                               (const T*)dout, dimg, N, S, C);
        });
    }
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_logits_fwd(int dtype, const void* h, const float* w, const float* bias, float* prob, int B,
                              int K, void* stream) {
    if (!h || !w || !bias || !prob || B <= 0 || K <= 0 || K % 8) return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((logits_fwd_kernel<T>), dim3(B), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)h, w, bias, prob, K));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_logits_bwd(int dtype, const void* h, const float* w, const float* prob, const float* dprob,
                              void* dh, float* dw, float* dbias, int B, int K, int accumulate, void* stream) {
    if (!h || !w || !prob || !dprob || !dh || B <= 0 || B > 4096 || K <= 0 || K % 8) return SBA_E_ARG;
    const int V = dtype == SBA_BF16 ? 8 : 4;
    if (sba_det_on() && (dw || dbias)) {
        // the sample chunks add into the same dw / dbias: one launch per chunk, in order
        for (int ck = 0; ck < cdiv(B, LOGITS_LB); ++ck)
            SBA_DISPATCH(dtype, SBA_LAUNCH((logits_bwd_kernel<T>), dim3(cdiv(K / V, 256), 1), dim3(256), 0,
                                           (hipStream_t)stream, (const T*)h, w, prob, dprob, (T*)dh, dw, dbias, B, K,
                                           accumulate, ck));
        return SBA_CHECK_LAUNCH();
    }
    SBA_DISPATCH(dtype, SBA_LAUNCH((logits_bwd_kernel<T>), dim3(cdiv(K / V, 256), cdiv(B, LOGITS_LB)), dim3(256),
                                           0, (hipStream_t)stream, (const T*)h, w, prob, dprob,
                                           (T*)dh, dw, dbias, B, K, accumulate, 0));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_cond_cat_fwd(int dtype, const void* h, const float* sent, void* out, int B, int C, int E,
                                void* stream) {
    if (!h || !sent || !out || B <= 0 || C <= 0 || E <= 0) return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((cond_cat_fwd_kernel<T>), dim3(grid_for((int64_t)B * 16 * (C + E))),
                                           dim3(256), 0, (hipStream_t)stream, (const T*)h, sent, (T*)out, B, C, E));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_cond_cat_bwd(int dtype, const void* dout, void* dh, float* dsent, int B, int C, int E,
                                int accumulate, void* stream) {
    if (!dout || !dh || B <= 0 || C <= 0 || E <= 0) return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((cond_cat_bwd_kernel<T>), dim3(grid_for((int64_t)B * 16 * C + B * E)),
                                           dim3(256), 0, (hipStream_t)stream, (const T*)dout, (T*)dh, dsent, B, C, E,
                                           accumulate));
    return SBA_CHECK_LAUNCH();
}
