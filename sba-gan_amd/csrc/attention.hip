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

template <typename T, int IDF>
int launch_fwd(const void* h, const float* src, const uint8_t* mask, void* ctx, float* att, int B, int Q, int L,
               int mode, int ocs, int oco, hipStream_t st) {
    dim3 grid(cdiv(Q, 256), B);
    SBA_LAUNCH((word_attn_fwd_kernel<T, IDF>), grid, dim3(256), 0, st, (const T*)h, src, mask, (T*)ctx, att,
                       B, Q, L, mode, ocs, oco);
    return SBA_CHECK_LAUNCH();
}

template <typename T, int IDF>
int launch_bwd(const void* h, const float* src, const uint8_t* mask, const void* dctx, void* dh, float* dsrc, int B,
               int Q, int L, int mode, int dcs, int dco, int acc, hipStream_t st) {
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
