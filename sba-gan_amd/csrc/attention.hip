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
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int LMAX = 32;

// bit l set <=> word l takes no part in the softmax (padding column l >= L, or masked in row `mrow`)
__device__ __forceinline__ uint32_t dead_words(const uint8_t* __restrict__ mask, int mrow, int L) {
    uint32_t bits = L < 32 ? ~((1u << L) - 1u) : 0u;
    if (mask) {
#pragma unroll 1
        for (int l = 0; l < L; ++l) bits |= (mask[mrow * L + l] ? 1u : 0u) << l;
    }
    return bits;
}

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
    // The channel loops stay ROLLED: fully unrolled, the compiler hoists all IDF*LMAX LDS reads to the
    // top of the kernel and spills them to scratch (2-15 KB per lane).
#pragma unroll 1
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
    const uint32_t deadbits = dead_words(mask, mrow, L);
    float mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        s[l] = ((deadbits >> l) & 1u) ? -INFINITY : s[l];
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
#pragma unroll 1
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

// ---------------------------------------------------------------------------
// The forward on the matrix cores (bf16 activations): both contractions of GlobalAttention.py:103,117 as
// v_mfma_f32_32x32x16 tiles with the QUERY axis as the MFMA's N (lane) axis, so that no operand ever passes
// through LDS and the softmax is in-lane:
//   S^T[l][q]  = sum_c src[c][l] h[q][c]      A = src^T (32 words x 16 channels per k-step, constant per image:
//                                              kept in registers), B = h^T: lane (q, k-group g) needs channels
//                                              16 s + 8 g .. +8 of its query = ONE 16-byte load straight from HBM
//   the accumulator of lane (q, g) then holds words l = (r&3) + 8 (r>>2) + 4 g, r < 16, of query q: softmax over
//   the words = 16 registers + ONE exchange with lane q of the other half-wave (__shfl_xor 32)
//   ctx^T[c][q] = sum_l src[c][l] a[q][l]     A = src (32 channels x 16 words per k-step, registers), B = a^T: the
//                                              contraction index is walked in the order the accumulator already holds
//                                              the words (k-step s, group g, j -> word (j&3) + 8 (2s + (j>>2)) + 4 g),
//                                              applied to A and B alike: no cross-lane movement at all
// MODE 0: bf16 operands, the f32 src and the probabilities split into hi + lo bf16 parts (5 MFMAs instead of 2 per
//         k-step pair: the matrix cores idle anyway, the kernel is HBM-bound at 128 B per query) -- results within
//         f32 rounding of the VALU kernel above;
// MODE 1: BASELINE config 5 -- FP8 (OCP e4m3) operands on v_mfma_f32_32x32x16_fp8_fp8 for BOTH contractions:
//         h scaled per 32-query tile, src per image, by powers of two (exact un-scaling), probabilities x 256.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t pack8_bf16(const float (&v)[8]) {
    bf16x8_t r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(v[j]);
    return r;
}
__device__ __forceinline__ long pack8_e4m3(const float (&v)[8], const float s) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * s, v[1] * s, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * s, v[3] * s, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * s, v[5] * s, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * s, v[7] * s, hi, true);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}
__device__ __forceinline__ float pow2_scale(const float amax, const float top) {
    return amax > 0.f ? exp2f(floorf(log2f(top / amax))) : 1.f;
}

template <int IDF, int MODE>
__global__ __launch_bounds__(256) void word_attn_fwd_mfma_kernel(
    const bf16_t* __restrict__ h, const float* __restrict__ src, const uint8_t* __restrict__ mask,
    bf16_t* __restrict__ ctx, float* __restrict__ att, int B, int Q, int L, int mask_mode, int ocs, int oco,
    int tiles_per_wave) {
    constexpr int KS = IDF / 16;            // k-steps of the score contraction (channels)
    constexpr int MT = IDF / 32;            // 32-channel tiles of the context
    constexpr int NDEAD = 1024;
    __shared__ uint32_t s_dead[NDEAD];      // mask_mode 0: dead-word bits of mask row (r % B), r = b*Q + q
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ql = lane & 31, kg = lane >> 5;
    const int nrows = mask_mode == 0 ? B : 1;
    for (int i = tid; i < nrows && i < NDEAD; i += 256) s_dead[i] = dead_words(mask, mask_mode == 0 ? i : b, L);
    // ---- the image's keys as MFMA A operands (registers)
    const float* sb = src + (int64_t)b * IDF * L;
    float smax = 0.f;
    if (MODE == 1) {
        for (int i = lane; i < IDF * L; i += 64) smax = fmaxf(smax, fabsf(sb[i]));
        smax = wave_max(smax);
    }
    const float ss = MODE == 1 ? pow2_scale(smax, 448.f) : 1.f;
    bf16x8_t st_hi[KS], st_lo[KS];          // score contraction: row = word ql, k = channel 16 s + 8 kg + j
    long st_f8[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float v[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = ql < L ? sb[(16 * s + 8 * kg + j) * L + ql] : 0.f;
            lo[j] = v[j] - bf2f(f2bf(v[j]));
        }
        st_hi[s] = pack8_bf16(v);
        st_lo[s] = pack8_bf16(lo);
        st_f8[s] = MODE == 1 ? pack8_e4m3(v, ss) : 0;
    }
    bf16x8_t sc_hi[MT][2], sc_lo[MT][2];    // context contraction: row = channel 32 mt + ql, k = word wmap(s, kg, j)
    long sc_f8[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8], lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int l = (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * kg;
                v[j] = l < L ? sb[(32 * mt + ql) * L + l] : 0.f;
                lo[j] = v[j] - bf2f(f2bf(v[j]));
            }
            sc_hi[mt][s] = pack8_bf16(v);
            sc_lo[mt][s] = pack8_bf16(lo);
            sc_f8[mt][s] = MODE == 1 ? pack8_e4m3(v, ss) : 0;
        }
    __syncthreads();

    const int tile0 = (blockIdx.x * 4 + wid) * tiles_per_wave;
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int q0 = (tile0 + t) * 32;
        if (q0 >= Q) break;                                   // (wave-uniform)
        const int q = q0 + ql;
        const bool live = q < Q;
        const int64_t r = (int64_t)b * Q + (live ? q : 0);
        // ---- scores
        bf16x8_t hb[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) hb[s] = *reinterpret_cast<const bf16x8_t*>(h + r * IDF + 16 * s + 8 * kg);
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float sh = 1.f;
        if (MODE == 1) {
            float hm = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) hm = fmaxf(hm, fabsf(bf2f((bf16_t)hb[s][j])));
            sh = pow2_scale(wave_max(hm), 448.f);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = bf2f((bf16_t)hb[s][j]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(st_f8[s], pack8_e4m3(v, sh), acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(st_hi[s], hb[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(st_lo[s], hb[s], acc, 0, 0, 0);
            }
        }
        const float unscale = MODE == 1 ? 1.f / (ss * sh) : 1.f;
        // ---- softmax over the words: 16 in this lane, 16 in lane ^ 32
        const int mrow = mask_mode == 0 ? (int)(r % B) : 0;
        const uint32_t deadbits = (mask_mode == 0 && mrow >= NDEAD) ? dead_words(mask, mrow, L) : s_dead[mrow];
        float a[16];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int l = (i & 3) + 8 * (i >> 2) + 4 * kg;
            a[i] = ((deadbits >> l) & 1u) ? -INFINITY : acc[i] * unscale;
            mx = fmaxf(mx, a[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a[i] = __expf(a[i] - mx); sum += a[i]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] *= inv;
        if (att && live) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int l = (i & 3) + 8 * (i >> 2) + 4 * kg;
                if (l < L) att[((int64_t)b * L + l) * Q + q] = a[i];
            }
        }
        // ---- context
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x16_t c;
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8], lo[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] = a[8 * s + j]; lo[j] = v[j] - bf2f(f2bf(v[j])); }
                if (MODE == 1) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(sc_f8[mt][s], pack8_e4m3(v, 256.f), c, 0, 0, 0);
                } else {
                    const bf16x8_t ah = pack8_bf16(v), al = pack8_bf16(lo);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sc_hi[mt][s], ah, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sc_hi[mt][s], al, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sc_lo[mt][s], ah, c, 0, 0, 0);
                }
            }
            const float cs = MODE == 1 ? 1.f / (ss * 256.f) : 1.f;
            if (live) {
                bf16_t* op = ctx + r * ocs + oco + 32 * mt + 4 * kg;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {        // channels 32 mt + 8 g4 + 4 kg .. + 4
                    uint2 o;
                    o.x = (uint32_t)f2bf(c[4 * g4] * cs) | ((uint32_t)f2bf(c[4 * g4 + 1] * cs) << 16);
                    o.y = (uint32_t)f2bf(c[4 * g4 + 2] * cs) | ((uint32_t)f2bf(c[4 * g4 + 3] * cs) << 16);
                    *reinterpret_cast<uint2*>(op + 8 * g4) = o;
                }
            }
        }
    }
}

// backward: recompute the attention row, then
//   dA[l]  = sum_c dctx[c] src[c][l];  dS[l] = a[l] (dA[l] - sum_l' a dA)
//   dh[c]  = sum_l dS[l] src[c][l]
//   dsrc[c][l] += sum_q (h[q][c] dS[q][l] + dctx[q][c] a[q][l])
// The last line is a (idf x Q) . (Q x L) contraction over the query axis: each wave stages its
// 64 queries' rows ([q][32] tiles of T) in LDS and accumulates 32 x 32 tiles with MFMA
// (bf16: 32x32x16 through ds_read_b64_tr_b16; f32: 32x32x2), over all the query chunks the
// workgroup walks; one LDS reduction + idf*L global atomics per workgroup at the end.
template <typename T> struct AttnMma;
template <> struct AttnMma<bf16_t> {
    static constexpr int NW = 4;
    static constexpr int RS = 64;                      // bytes per [q][32] row
    // fragment of columns [0,32) over rows [16*ks, 16*ks+16) of a [64][32] bf16 tile
    static __device__ __forceinline__ bf16x8_t frag(const unsigned char* tile, int ks, int lane) {
        const int g16 = lane >> 4, i16 = lane & 15;
        const int cbase = 16 * (g16 & 1), kbase = 16 * ks + 8 * (g16 >> 1);
        const int q = i16 >> 2, p = i16 & 3;
        const unsigned char* a0 = tile + (kbase + q) * RS + (cbase + 4 * p) * 2;
        typedef __attribute__((address_space(3))) s16x4_t* lptr;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * RS));
        bf16x8_t r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
    // acc[c][l] += sum_q X[q][c] * Y[q][l]
    static __device__ __forceinline__ void mma(const unsigned char* X, const unsigned char* Y, int lane, f32x16_t& acc) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(X, ks, lane), frag(Y, ks, lane), acc, 0, 0, 0);
    }
};
template <> struct AttnMma<float> {
    static constexpr int NW = 2;
    static constexpr int RS = 128;
    static __device__ __forceinline__ void mma(const unsigned char* X, const unsigned char* Y, int lane, f32x16_t& acc) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const float a = *reinterpret_cast<const float*>(X + (2 * kk + h) * RS + r * 4);
            const float b = *reinterpret_cast<const float*>(Y + (2 * kk + h) * RS + r * 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
};

template <typename T, int IDF>
__global__ __launch_bounds__(AttnMma<T>::NW * 64) void word_attn_bwd_kernel(
    const T* __restrict__ h, const float* __restrict__ src, const uint8_t* __restrict__ mask,
    const T* __restrict__ dctx, T* __restrict__ dh, float* __restrict__ dsrc, int B, int Q, int L,
    int mask_mode, int dcs, int dco, int accumulate, int chunks, float* __restrict__ det_part) {
    constexpr int V = Vec16<T>::N;
    constexpr int NW = AttnMma<T>::NW;                   // waves per workgroup (LDS budget: 4 bf16, 2 f32)
    constexpr int NT = NW * 64;
    constexpr int CT = IDF / 32;                         // 32-channel tiles
    constexpr int RS = AttnMma<T>::RS;
    constexpr int TILE = 64 * RS;                        // one [64 q][32] tile
    constexpr int WAVE_BYTES = (2 * CT + 2) * TILE;      // per wave: h tiles, dctx tiles, dS tile, a tile
    __shared__ float s_src[IDF * LMAX];
    __shared__ __attribute__((aligned(16))) unsigned char s_t[NW * WAVE_BYTES];
    __shared__ float s_red[IDF * 32];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < IDF * LMAX; i += NT) {
        const int c = i / LMAX, l = i - c * LMAX;
        s_src[i] = l < L ? src[((int64_t)b * IDF + c) * L + l] : 0.f;
    }
    __syncthreads();
    unsigned char* XH = s_t + wid * WAVE_BYTES;          // [CT][64 q][32 ch] of h
    unsigned char* XD = XH + CT * TILE;                  // [CT][64 q][32 ch] of dctx
    unsigned char* YS = XD + CT * TILE;                  // [64 q][32 l] of dS
    unsigned char* YA = YS + TILE;                       // [64 q][32 l] of a

    f32x16_t acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int ck = 0; ck < chunks; ++ck) {
        const int q = (blockIdx.x * chunks + ck) * NT + tid;
        const bool live = q < Q;
        const int64_t r = (int64_t)b * Q + (live ? q : 0);
        const int mrow = mask_mode == 0 ? (int)(r % B) : b;

        float s[LMAX], dA[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { s[l] = 0.f; dA[l] = 0.f; }
        const T* hp = h + r * IDF;
        const T* dp = dctx + r * dcs + dco;
        // rolled channel loops (see the forward kernel); the rows go to the wave's LDS tiles as they
        // are read, so nothing but the 2 x LMAX scores stays in registers
#pragma unroll 1
        for (int cv = 0; cv < IDF / V; ++cv) {
            Vec16<T> hv = ld16(hp + cv * V), dv = ld16(dp + cv * V);
            if (!live) {
#pragma unroll
                for (int k = 0; k < V; ++k) { hv.set(k, 0.f); dv.set(k, 0.f); }
            }
            const int t = (cv * V) / 32, cin = cv * V - t * 32;
            st16(reinterpret_cast<T*>(XH + t * TILE + lane * RS) + cin, hv);
            st16(reinterpret_cast<T*>(XD + t * TILE + lane * RS) + cin, dv);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float hh = hv.get(k), dd = dv.get(k);
                const float* sr = &s_src[(cv * V + k) * LMAX];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) { s[l] += hh * sr[l]; dA[l] += dd * sr[l]; }
            }
        }
        const uint32_t deadbits = dead_words(mask, mrow, L);
        float mx = -INFINITY;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            s[l] = ((deadbits >> l) & 1u) ? -INFINITY : s[l];
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
        for (int l = 0; l < LMAX; ++l) dA[l] = live ? s[l] * (dA[l] - dot) : 0.f;     // dA now holds dS
        {
            T* ys = reinterpret_cast<T*>(YS + lane * RS);
            T* ya = reinterpret_cast<T*>(YA + lane * RS);
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                ys[l] = from_f<T>(dA[l]);
                ya[l] = from_f<T>(live ? s[l] : 0.f);
            }
        }
        if (live) {
            T* op = dh + r * IDF;
#pragma unroll 1
            for (int cv = 0; cv < IDF / V; ++cv) {
                Vec16<T> o;
                if (accumulate) o = ld16(op + cv * V);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float* sr = &s_src[(cv * V + k) * LMAX];
                    float a2 = 0.f;
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) a2 += dA[l] * sr[l];
                    if (accumulate) a2 += o.get(k);
                    o.set(k, a2);
                }
                st16(op + cv * V, o);
            }
        }
        // ---- dsrc contraction on the matrix cores, 32 channels at a time (tiles are wave-private)
        __syncthreads();
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            AttnMma<T>::mma(XH + t * TILE, YS, lane, acc[t]);
            AttnMma<T>::mma(XD + t * TILE, YA, lane, acc[t]);
        }
        __syncthreads();                            // tiles consumed before the next chunk overwrites them
    }
    // acc[t]: rows = channel (r&3)+8(r>>2)+4(lane>>5) of tile t, column = l = lane&31; the waves' tiles are added
    // wave by wave (a fixed order: no LDS atomics)
#pragma unroll 1
    for (int wv = 0; wv < NW; ++wv) {
        if (wid == wv) {
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float* q = &s_red[c * 32 + (lane & 31)];
                    *q = wv == 0 ? acc[t][r] : *q + acc[t][r];
                }
        }
        __syncthreads();
    }
    if (det_part) {         // deterministic mode: this workgroup's own slot; sba_det_fold adds the slots in order
        float* part = det_part + ((int64_t)b * gridDim.x + blockIdx.x) * IDF * L;
        for (int o = tid; o < IDF * L; o += NT) part[o] = s_red[(o / L) * 32 + (o - (o / L) * L)];
        return;
    }
    for (int o = tid; o < IDF * L; o += NT) {
        const int c = o / L, l = o - c * L;
        atomicAdd(&dsrc[((int64_t)b * IDF + c) * L + l], s_red[c * 32 + l]);
    }
}

// ---------------------------------------------------------------------------
// The backward on the matrix cores (bf16 activations, idf = 32: NEXT_STAGE_G of every stage), in the forward kernel's
// formulation -- the QUERY axis is the MFMA's lane axis, nothing but the dsrc contraction touches LDS:
//   S^T[l][q]  = sum_c src[c][l] h[q][c]        recomputed as in the forward (A = src^T hi + lo, B = h from HBM)
//   dA^T[l][q] = sum_c src[c][l] dctx[q][c]     the SAME A operands, B = dctx (one 16-byte load per k-step)
//   softmax and dS = a (dA - <a, dA>)           in-lane over the lane's 16 words + ONE exchange with lane ^ 32
//   dh^T[c][q] = sum_l src[c][l] dS[q][l]       the forward's context contraction with dS in place of a (hi + lo)
//   dsrc[c][l] += sum_q (h[q][c] dS[q][l] + dctx[q][c] a[q][l])   over the query axis: the wave's four [32 q][32] bf16
//                                               tiles go to LDS and come back through ds_read_b64_tr_b16 (as before)
// The VALU kernel above spends 3 x 32 x 32 FMAs per query on the vector pipe, each fed by an LDS read of src: 89 us at
// 128 x 128 queries (0.7 TB/s); here 18 MFMAs per 32 queries.
// ---------------------------------------------------------------------------
template <int IDF>
__global__ __launch_bounds__(256) void word_attn_bwd_mfma_kernel(
    const bf16_t* __restrict__ h, const float* __restrict__ src, const uint8_t* __restrict__ mask,
    const bf16_t* __restrict__ dctx, bf16_t* __restrict__ dh, float* __restrict__ dsrc, int B, int Q, int L,
    int mask_mode, int dcs, int dco, int accumulate, int tiles_per_wave, float* __restrict__ det_part) {
    static_assert(IDF == 32, "one 32-channel tile");
    constexpr int KS = IDF / 16, NDEAD = 1024, TILE = 32 * 64;      // a [32 q][32] bf16 tile: 2 KB
    __shared__ uint32_t s_dead[NDEAD];
    __shared__ __attribute__((aligned(16))) unsigned char s_t[4 * 4 * TILE];     // per wave: h, dctx, dS, a
    __shared__ float s_red[IDF * 32];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ql = lane & 31, kg = lane >> 5;
    const int nrows = mask_mode == 0 ? B : 1;
    for (int i = tid; i < nrows && i < NDEAD; i += 256) s_dead[i] = dead_words(mask, mask_mode == 0 ? i : b, L);
    const float* sb = src + (int64_t)b * IDF * L;
    bf16x8_t st_hi[KS], st_lo[KS];          // score / dA contraction: row = word ql, k = channel 16 s + 8 kg + j
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float v[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = ql < L ? sb[(16 * s + 8 * kg + j) * L + ql] : 0.f;
            lo[j] = v[j] - bf2f(f2bf(v[j]));
        }
        st_hi[s] = pack8_bf16(v);
        st_lo[s] = pack8_bf16(lo);
    }
    bf16x8_t sc_hi[2], sc_lo[2];            // dh contraction: row = channel ql, k = word wmap(s, kg, j)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int l = (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * kg;
            v[j] = l < L ? sb[ql * L + l] : 0.f;
            lo[j] = v[j] - bf2f(f2bf(v[j]));
        }
        sc_hi[s] = pack8_bf16(v);
        sc_lo[s] = pack8_bf16(lo);
    }
    __syncthreads();
    unsigned char* XH = s_t + wid * 4 * TILE;
    unsigned char* XD = XH + TILE;
    unsigned char* YS = XD + TILE;
    unsigned char* YA = YS + TILE;
    f32x16_t wacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) wacc[i] = 0.f;

    const int tile0 = (blockIdx.x * 4 + wid) * tiles_per_wave;
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int q0 = (tile0 + t) * 32;
        if (q0 >= Q) break;                                   // (wave-uniform)
        const int q = q0 + ql;
        const bool live = q < Q;
        const int64_t r = (int64_t)b * Q + (live ? q : 0);
        bf16x8_t hb[KS], db[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            hb[s] = *reinterpret_cast<const bf16x8_t*>(h + r * IDF + 16 * s + 8 * kg);
            db[s] = *reinterpret_cast<const bf16x8_t*>(dctx + r * dcs + dco + 16 * s + 8 * kg);
            if (!live) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { hb[s][j] = 0; db[s][j] = 0; }
            }
        }
        f32x16_t sa, da;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.f; da[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(st_hi[s], hb[s], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(st_lo[s], hb[s], sa, 0, 0, 0);
            da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(st_hi[s], db[s], da, 0, 0, 0);
            da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(st_lo[s], db[s], da, 0, 0, 0);
        }
        // ---- softmax over the words (16 in this lane, 16 in lane ^ 32), then dS = a (dA - <a, dA>)
        const int mrow = mask_mode == 0 ? (int)(r % B) : 0;
        const uint32_t deadbits = (mask_mode == 0 && mrow >= NDEAD) ? dead_words(mask, mrow, L) : s_dead[mrow];
        float a[16], ds[16];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int l = (i & 3) + 8 * (i >> 2) + 4 * kg;
            a[i] = ((deadbits >> l) & 1u) ? -INFINITY : sa[i];
            mx = fmaxf(mx, a[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a[i] = __expf(a[i] - mx); sum += a[i]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a[i] *= inv; dot += a[i] * da[i]; }
        dot += __shfl_xor(dot, 32, 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ds[i] = live ? a[i] * (da[i] - dot) : 0.f;
            if (!live) a[i] = 0.f;
        }
        // ---- dh
        {
            f32x16_t c;
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8], lo[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] = ds[8 * s + j]; lo[j] = v[j] - bf2f(f2bf(v[j])); }
                const bf16x8_t ah = pack8_bf16(v), al = pack8_bf16(lo);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sc_hi[s], ah, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sc_hi[s], al, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sc_lo[s], ah, c, 0, 0, 0);
            }
            if (live) {
                bf16_t* op = dh + r * IDF + 4 * kg;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {        // channels 8 g4 + 4 kg .. + 4
                    float v[4] = {c[4 * g4], c[4 * g4 + 1], c[4 * g4 + 2], c[4 * g4 + 3]};
                    if (accumulate) {
                        const uint2 pr = *reinterpret_cast<const uint2*>(op + 8 * g4);
                        v[0] += __uint_as_float(pr.x << 16); v[1] += __uint_as_float(pr.x & 0xffff0000u);
                        v[2] += __uint_as_float(pr.y << 16); v[3] += __uint_as_float(pr.y & 0xffff0000u);
                    }
                    uint2 o;
                    o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                    o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                    *reinterpret_cast<uint2*>(op + 8 * g4) = o;
                }
            }
        }
        // ---- dsrc: rows of the four [32 q][32] tiles (a wave's LDS accesses complete in order: no barrier)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<bf16x8_t*>(XH + ql * 64 + (16 * s + 8 * kg) * 2) = hb[s];
            *reinterpret_cast<bf16x8_t*>(XD + ql * 64 + (16 * s + 8 * kg) * 2) = db[s];
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {                // words 8 g4 + 4 kg .. + 4
            uint2 o, p;
            o.x = (uint32_t)f2bf(ds[4 * g4]) | ((uint32_t)f2bf(ds[4 * g4 + 1]) << 16);
            o.y = (uint32_t)f2bf(ds[4 * g4 + 2]) | ((uint32_t)f2bf(ds[4 * g4 + 3]) << 16);
            p.x = (uint32_t)f2bf(a[4 * g4]) | ((uint32_t)f2bf(a[4 * g4 + 1]) << 16);
            p.y = (uint32_t)f2bf(a[4 * g4 + 2]) | ((uint32_t)f2bf(a[4 * g4 + 3]) << 16);
            *reinterpret_cast<uint2*>(YS + ql * 64 + (8 * g4 + 4 * kg) * 2) = o;
            *reinterpret_cast<uint2*>(YA + ql * 64 + (8 * g4 + 4 * kg) * 2) = p;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AttnMma<bf16_t>::frag(XH, ks, lane),
                                                           AttnMma<bf16_t>::frag(YS, ks, lane), wacc, 0, 0, 0);
            wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AttnMma<bf16_t>::frag(XD, ks, lane),
                                                           AttnMma<bf16_t>::frag(YA, ks, lane), wacc, 0, 0, 0);
        }
    }
    // wacc: rows = channel (r&3)+8(r>>2)+4 kg, column = word ql; the waves' tiles are added in wave order
#pragma unroll 1
    for (int wv = 0; wv < 4; ++wv) {
        if (wid == wv) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = (i & 3) + 8 * (i >> 2) + 4 * kg;
                float* qd = &s_red[c * 32 + ql];
                *qd = wv == 0 ? wacc[i] : *qd + wacc[i];
            }
        }
        __syncthreads();
    }
    if (det_part) {         // deterministic mode: this workgroup's own slot; sba_det_fold adds the slots in order
        float* part = det_part + ((int64_t)b * gridDim.x + blockIdx.x) * IDF * L;
        for (int o = tid; o < IDF * L; o += 256) part[o] = s_red[(o / L) * 32 + (o - (o / L) * L)];
        return;
    }
    for (int o = tid; o < IDF * L; o += 256) {
        const int c = o / L, l = o - c * L;
        atomicAdd(&dsrc[((int64_t)b * IDF + c) * L + l], s_red[c * 32 + l]);
    }
}

// SBA_ATTN_MFMA=0: the VALU kernel for bf16 as well (A/B aid)
static int attn_mfma_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SBA_ATTN_MFMA"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}

template <int IDF, int MODE>
int launch_fwd_mfma(const void* h, const float* src, const uint8_t* mask, void* ctx, float* att, int B, int Q, int L,
                    int mode, int ocs, int oco, hipStream_t st) {
    const int tiles = cdiv(Q, 32);
    int tpw = tiles * B >= 16384 ? 4 : (tiles * B >= 4096 ? 2 : 1);        // >= ~1000 workgroups where the map allows
    dim3 grid(cdiv(tiles, 4 * tpw), B);
    SBA_LAUNCH((word_attn_fwd_mfma_kernel<IDF, MODE>), grid, dim3(256), 0, st, (const bf16_t*)h, src, mask,
               (bf16_t*)ctx, att, B, Q, L, mode, ocs, oco, tpw);
    return SBA_CHECK_LAUNCH();
}

template <typename T, int IDF>
int launch_fwd(const void* h, const float* src, const uint8_t* mask, void* ctx, float* att, int B, int Q, int L,
               int mode, int ocs, int oco, hipStream_t st) {
    if (sizeof(T) == 2 && IDF <= 64 && ocs % 4 == 0 && oco % 4 == 0 && attn_mfma_enabled())
        return launch_fwd_mfma<(IDF <= 64 ? IDF : 32), 0>(h, src, mask, ctx, att, B, Q, L, mode, ocs, oco, st);
    dim3 grid(cdiv(Q, 256), B);
    SBA_LAUNCH((word_attn_fwd_kernel<T, IDF>), grid, dim3(256), 0, st, (const T*)h, src, mask, (T*)ctx, att,
                       B, Q, L, mode, ocs, oco);
    return SBA_CHECK_LAUNCH();
}

template <typename T, int IDF>
int launch_bwd(const void* h, const float* src, const uint8_t* mask, const void* dctx, void* dh, float* dsrc, int B,
               int Q, int L, int mode, int dcs, int dco, int acc, hipStream_t st) {
    if (sizeof(T) == 2 && IDF == 32 && dcs % 8 == 0 && dco % 8 == 0 && attn_mfma_enabled()) {
        const int tiles = cdiv(Q, 32);
        const int tpw = tiles * B >= 16384 ? 4 : (tiles * B >= 4096 ? 2 : 1);
        dim3 grid(cdiv(tiles, 4 * tpw), B);
        float* part = nullptr;
        if (sba_det_on()) {
            part = sba_det_alloc((int64_t)B * grid.x * IDF * L);
            if (!part) return SBA_E_ARG;
        }
        SBA_LAUNCH((word_attn_bwd_mfma_kernel<32>), grid, dim3(256), 0, st, (const bf16_t*)h, src, mask,
                   (const bf16_t*)dctx, (bf16_t*)dh, dsrc, B, Q, L, mode, dcs, dco, acc, tpw, part);
        if (part) sba_det_fold(part, B, (int)grid.x, (int64_t)IDF * L, dsrc, (int64_t)IDF * L, 0, st);
        return SBA_CHECK_LAUNCH();
    }
    constexpr int NT = AttnMma<T>::NW * 64;
    int chunks = Q >= 16384 ? 4 : (Q >= 4096 ? 2 : 1);           // NT-query chunks per workgroup
    dim3 grid(cdiv(Q, NT * chunks), B);
    float* part = nullptr;
    if (sba_det_on()) {
        part = sba_det_alloc((int64_t)B * grid.x * IDF * L);
        if (!part) return SBA_E_ARG;
    }
    SBA_LAUNCH((word_attn_bwd_kernel<T, IDF>), grid, dim3(NT), 0, st, (const T*)h, src, mask,
                       (const T*)dctx, (T*)dh, dsrc, B, Q, L, mode, dcs, dco, acc, chunks, part);
    if (part) sba_det_fold(part, B, (int)grid.x, (int64_t)IDF * L, dsrc, (int64_t)IDF * L, 0, st);
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

extern "C" int sba_word_attn_fwd_fp8(const void* h, const float* src, const uint8_t* mask, void* ctx, float* att, int B,
                                     int Q, int idf, int L, int mask_mode, int out_cstride, int out_coff, void* stream) {
    if (!h || !src || !ctx || B <= 0 || Q <= 0 || L <= 0 || L > LMAX || B > 65535) return SBA_E_ARG;
    if (mask_mode != 0 && mask_mode != 1) return SBA_E_ARG;
    if (out_cstride < idf + out_coff || out_cstride % 8 || out_coff % 8) return SBA_E_ARG;
    if (idf == 32) return launch_fwd_mfma<32, 1>(h, src, mask, ctx, att, B, Q, L, mask_mode, out_cstride, out_coff,
                                                 (hipStream_t)stream);
    if (idf == 64) return launch_fwd_mfma<64, 1>(h, src, mask, ctx, att, B, Q, L, mask_mode, out_cstride, out_coff,
                                                 (hipStream_t)stream);
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
    if (idf != 32 && idf != 64) return SBA_E_ARG;
    SBA_DISPATCH(dtype, {
        if (idf == 32) return (launch_bwd<T, 32>(h, src, mask, dctx, dh, dsrc, B, Q, L, mask_mode, dctx_cstride,
                                                 dctx_coff, accumulate, (hipStream_t)stream));
        return (launch_bwd<T, 64>(h, src, mask, dctx, dh, dsrc, B, Q, L, mask_mode, dctx_cstride, dctx_coff,
                                  accumulate, (hipStream_t)stream));
    });
    return SBA_E_ARG;
}
