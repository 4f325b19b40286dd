// DAMSM words loss (miscc/losses.py:62-132 with func_attention, GlobalAttention.py:31-69) on the bf16 matrix cores.
//
// The f32 kernels of damsm.hip run the 289 x T x 256 contractions of every (caption i, image j) pair on the VALU and add
// every pair's d(features) with f32 atomics (29.6 M of them at B = 20): 184 + 318 us, undiluted on the critical chain of
// the step (generator forward -> image encoder -> THIS -> image encoder backward -> generator backward).  Here:
//
//   prep   f32 features [B][nef][R] and words [B][nef][L] -> bf16 hi + lo parts in both operand layouts
//          (F [B][RP][nef] / FT [B][nef][RP], Q [B][32][nef] / QT [B][nef][32]; zero beyond R / the caption length).
//          Every product below is xh*yh + xl*yh + xh*yl: ~16 mantissa bits per product, f32 sums.
//   fwd    one workgroup per pair (i, j): scores S[t][r] (MFMA 32x32x16, words = rows, K = nef, regions = the lane axis:
//          softmax over words in-lane + ONE __shfl_xor 32, as csrc/attention.hip), x gamma1, softmax over regions
//          (a1 <= 1, so exp(gamma1 a1) needs no max; region sums: 5 shuffles + LDS), attended context
//          wctx[c][t] = sum_r FT[c][r] A[t][r] (K = regions, attention through LDS as hi + lo), cosine, log-sum-exp.
//   bwd1   one workgroup per pair: dcos -> dwctx (LDS, and f32 to scratch), dA = dwctx . F (the scores' loop), the two
//          softmax backward passes in registers -> dS (f32 to scratch); optional d(words) (DAMSM pre-training).
//   bwd2   d(features)[j] = sum_{i,t} A dwctx + dS q as ONE contraction per image over K = (caption, word): every
//          output element has exactly one owner -- no atomics, no per-pair partial tensors, deterministic by construction.
#include "common.h"

namespace {

constexpr int MF_NT = 256;          // 4 waves
constexpr int TP = 32;              // word axis padded to one MFMA tile

__device__ __forceinline__ bf16x8_t pack8(const float (&v)[8]) {
    bf16x8_t r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(v[j]);
    return r;
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8_t& hi, bf16x8_t& lo) {
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = v[j] - bf2f(f2bf(v[j]));
    hi = pack8(v);
    lo = pack8(l);
}
__device__ __forceinline__ f32x16_t mma3(const bf16x8_t ah, const bf16x8_t al, const bf16x8_t bh, const bf16x8_t bl,
                                         f32x16_t acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return acc;
}
// accumulator element e of lane (col, g) is row arow(e, g)
__device__ __forceinline__ int arow(int e, int g) { return (e & 3) + 8 * (e >> 2) + 4 * g; }

// ---- prep: src [b][C][X] f32 (x contiguous, valid x < Xv) -> T [b][XP][C] and D [b][C][XP], bf16 hi + lo -----------
__global__ __launch_bounds__(256) void damsm_prep_kernel(const float* __restrict__ src, const int64_t* __restrict__ lens,
                                                         bf16_t* __restrict__ Th, bf16_t* __restrict__ Tl,
                                                         bf16_t* __restrict__ Dh, bf16_t* __restrict__ Dl, int C, int X,
                                                         int XP) {
    __shared__ uint32_t tile[32][33];           // [c][x] : hi | lo << 16
    const int b = blockIdx.z, c0 = blockIdx.y * 32, x0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    int Xv = X;
    if (lens) { Xv = (int)lens[b]; Xv = Xv < 1 ? 1 : (Xv > X ? X : Xv); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, x = x0 + tx;
        float v = 0.f;
        if (x < Xv) v = src[((int64_t)b * C + c) * X + x];
        const bf16_t h = f2bf(v), l = f2bf(v - bf2f(h));
        tile[ty + 8 * k][tx] = (uint32_t)h | ((uint32_t)l << 16);
        Dh[((int64_t)b * C + c) * XP + x] = h;
        Dl[((int64_t)b * C + c) * XP + x] = l;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xl = ty + 8 * k;
        const uint32_t w = tile[tx][xl];
        const int64_t o = ((int64_t)b * XP + x0 + xl) * C + c0 + tx;
        Th[o] = (bf16_t)(w & 0xffffu);
        Tl[o] = (bf16_t)(w >> 16);
    }
}

struct Prep {       // views into the prep scratch
    const bf16_t *Fh, *Fl, *FTh, *FTl, *Qh, *Ql, *QTh, *QTl;
};
inline int64_t prep_elems_feat(int B, int nef, int RP) { return (int64_t)B * RP * nef; }
inline int64_t prep_elems_word(int B, int nef) { return (int64_t)B * TP * nef; }
inline Prep prep_views(const void* p, int B, int nef, int RP) {
    const bf16_t* b = (const bf16_t*)p;
    const int64_t nf = prep_elems_feat(B, nef, RP), nq = prep_elems_word(B, nef);
    Prep v;
    v.Fh = b; v.Fl = b + nf; v.FTh = b + 2 * nf; v.FTl = b + 3 * nf;
    v.Qh = b + 4 * nf; v.Ql = v.Qh + nq; v.QTh = v.Qh + 2 * nq; v.QTl = v.Qh + 3 * nq;
    return v;
}

// LDS of the per-pair kernels: xh/xl = the 32-row A operand of the score-shaped contraction ([32][nef], rows 16 bytes
// apart from a multiple of 256: conflict-free 16-byte fragment reads), ah/al = a [32][RP] word x region matrix, red = floats
struct PairLds {
    unsigned char *xh, *xl, *ah, *al;
    float* red;
    int XS, AS;
};
__device__ __forceinline__ PairLds pair_lds(unsigned char* sm, int nef, int RP) {
    PairLds l;
    l.XS = nef * 2 + 16;
    l.AS = RP * 2 + 16;
    l.xh = sm;
    l.xl = l.xh + TP * l.XS;
    l.ah = l.xl + TP * l.XS;
    l.al = l.ah + TP * l.AS;
    l.red = reinterpret_cast<float*>(l.al + TP * l.AS);
    return l;
}
inline size_t pair_lds_bytes(int nef, int RP) { return (size_t)2 * TP * (nef * 2 + 16) + (size_t)2 * TP * (RP * 2 + 16) + 4 * TP * 4 * 4; }

// acc[t][r] += sum_c X[t][c] F[r][c]: X from LDS (row = lane & 31), F rows of this lane from global; nef % 64 == 0
__device__ __forceinline__ f32x16_t score_tile(const bf16_t* __restrict__ fh, const bf16_t* __restrict__ fl,
                                               const unsigned char* xh, const unsigned char* xl, const int nef) {
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int c0 = 0; c0 < nef; c0 += 64) {
        bf16x8_t bh[4], bl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bh[s] = *reinterpret_cast<const bf16x8_t*>(fh + c0 + 16 * s);
            bl[s] = *reinterpret_cast<const bf16x8_t*>(fl + c0 + 16 * s);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(xh + (c0 + 16 * s) * 2);
            const bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(xl + (c0 + 16 * s) * 2);
            acc = mma3(ah, al, bh[s], bl[s], acc);
        }
    }
    return acc;
}

// acc[c][t] += sum_r FT[c][r] A[t][r]: FT rows of this lane from global (A operand), A from LDS (B operand); RP % 32 == 0
__device__ __forceinline__ f32x16_t ctx_tile(const bf16_t* __restrict__ fth, const bf16_t* __restrict__ ftl,
                                             const unsigned char* ah, const unsigned char* al, const int RP) {
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int r0 = 0; r0 < RP; r0 += 32) {
        bf16x8_t xh[2], xl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            xh[s] = *reinterpret_cast<const bf16x8_t*>(fth + r0 + 16 * s);
            xl[s] = *reinterpret_cast<const bf16x8_t*>(ftl + r0 + 16 * s);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8_t bh = *reinterpret_cast<const bf16x8_t*>(ah + (r0 + 16 * s) * 2);
            const bf16x8_t bl = *reinterpret_cast<const bf16x8_t*>(al + (r0 + 16 * s) * 2);
            acc = mma3(xh[s], xl[s], bh, bl, acc);
        }
    }
    return acc;
}

// sum of v[e] over the 32 lanes that share g = lane >> 5 (every lane gets the sum)
__device__ __forceinline__ void half_wave_sum16(float (&v)[16]) {
#pragma unroll
    for (int o = 1; o < 32; o <<= 1)
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += __shfl_xor(v[e], o, 64);
}

constexpr int MAX_RT = 3;           // region tiles per wave (RP <= 384)

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MF_NT, 2) void damsm_words_fwd_mfma_kernel(
    const Prep P, const float* __restrict__ words, const int64_t* __restrict__ cap_lens, float* __restrict__ sim,
    float* __restrict__ attn, float* __restrict__ attn1, float* __restrict__ wctx_o, const int B, const int nef,
    const int R, const int RP, const int Lw, const float gamma1, const float gamma2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const PairLds L = pair_lds(sm, nef, RP);
    const int i = blockIdx.x, j = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, g = lane >> 5;
    int T = (int)cap_lens[i];
    T = T < 1 ? 1 : (T > Lw ? Lw : T);
    const int64_t pair = (int64_t)j * B + i;
    // ---- the caption's words as the A operand
    {
        const int cpr = nef / 8;                    // 16-byte chunks per row
        const uint4* sh = reinterpret_cast<const uint4*>(P.Qh + (int64_t)i * TP * nef);
        const uint4* sl = reinterpret_cast<const uint4*>(P.Ql + (int64_t)i * TP * nef);
        for (int k = tid; k < TP * cpr; k += MF_NT) {
            const int t = k / cpr, ch = k - t * cpr;
            *reinterpret_cast<uint4*>(L.xh + t * L.XS + ch * 16) = sh[k];
            *reinterpret_cast<uint4*>(L.xl + t * L.XS + ch * 16) = sl[k];
        }
    }
    __syncthreads();
    // ---- scores, softmax over the words, x gamma1, exp
    const int ntile = RP / 32;
    float e2[MAX_RT][16], psum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) psum[e] = 0.f;
    const unsigned char* xh = L.xh + col * L.XS + 16 * g;
    const unsigned char* xl = L.xl + col * L.XS + 16 * g;
#pragma unroll
    for (int u = 0; u < MAX_RT; ++u) {
        const int rt = wid + 4 * u;
#pragma unroll
        for (int e = 0; e < 16; ++e) e2[u][e] = 0.f;
        if (rt < ntile) {                               // (wave-uniform)
            const int r = rt * 32 + col;
            const int64_t frow = ((int64_t)j * RP + r) * nef + 8 * g;
            const f32x16_t acc = score_tile(P.Fh + frow, P.Fl + frow, xh, xl, nef);
            float a[16], mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                a[e] = arow(e, g) < T ? acc[e] : -INFINITY;
                mx = fmaxf(mx, a[e]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { a[e] = __expf(a[e] - mx); sum += a[e]; }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            const bool rv = r < R;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = arow(e, g);
                const float a1 = a[e] * inv;
                if (rv && t < T) {
                    attn1[(pair * Lw + t) * R + r] = a1;
                    e2[u][e] = __expf(gamma1 * a1);
                    psum[e] += e2[u][e];
                }
            }
        }
    }
    // ---- softmax over the regions: sums per word over lanes, tiles and waves
    half_wave_sum16(psum);
    if (col == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) L.red[wid * TP + arow(e, g)] = psum[e];
    }
    __syncthreads();
    float inv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int t = arow(e, g);
        const float tot = L.red[t] + L.red[TP + t] + L.red[2 * TP + t] + L.red[3 * TP + t];
        inv[e] = t < T ? 1.f / tot : 0.f;
    }
#pragma unroll
    for (int u = 0; u < MAX_RT; ++u) {
        const int rt = wid + 4 * u;
        if (rt < ntile) {
            const int r = rt * 32 + col;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = arow(e, g);
                const float A = e2[u][e] * inv[e];
                if (r < R && t < T) attn[(pair * Lw + t) * R + r] = A;
                const bf16_t h = f2bf(A);
                *reinterpret_cast<bf16_t*>(L.ah + t * L.AS + r * 2) = h;
                *reinterpret_cast<bf16_t*>(L.al + t * L.AS + r * 2) = f2bf(A - bf2f(h));
            }
        }
    }
    __syncthreads();
    // ---- attended context wctx[c][t], cosine partial sums per word (lane = word)
    float w12 = 0.f, n1 = 0.f, n2 = 0.f;
    {
        const unsigned char* ah = L.ah + col * L.AS + 16 * g;
        const unsigned char* al = L.al + col * L.AS + 16 * g;
        const int t = col;
        for (int ct = wid; ct < nef / 32; ct += 4) {
            const int64_t trow = ((int64_t)j * nef + ct * 32 + col) * RP + 8 * g;
            const f32x16_t acc = ctx_tile(P.FTh + trow, P.FTl + trow, ah, al, RP);
            if (t < T) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int c = ct * 32 + 8 * m + 4 * g;
                    *reinterpret_cast<float4*>(wctx_o + (pair * Lw + t) * nef + c) =
                        make_float4(acc[4 * m], acc[4 * m + 1], acc[4 * m + 2], acc[4 * m + 3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float q = words[((int64_t)i * nef + c + k) * Lw + t], wv = acc[4 * m + k];
                        w12 += q * wv; n1 += q * q; n2 += wv * wv;
                    }
                }
            }
        }
    }
    w12 += __shfl_xor(w12, 32, 64); n1 += __shfl_xor(n1, 32, 64); n2 += __shfl_xor(n2, 32, 64);
    __syncthreads();                        // (red is reused)
    if (g == 0) { L.red[(wid * TP + col) * 3] = w12; L.red[(wid * TP + col) * 3 + 1] = n1; L.red[(wid * TP + col) * 3 + 2] = n2; }
    __syncthreads();
    if (wid == 0) {
        float ez = 0.f;
        if (lane < T) {
            float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += L.red[(w * TP + lane) * 3]; b += L.red[(w * TP + lane) * 3 + 1]; c += L.red[(w * TP + lane) * 3 + 2]; }
            ez = expf(gamma2 * (a / fmaxf(sqrtf(b) * sqrtf(c), 1e-8f)));
        }
        ez = wave_sum(ez);
        if (lane == 0) sim[pair] = logf(ez);
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MF_NT, 2) void damsm_words_bwd1_mfma_kernel(
    const Prep P, const float* __restrict__ words, const int64_t* __restrict__ cap_lens, const float* __restrict__ sim,
    const float* __restrict__ attn, const float* __restrict__ attn1, const float* __restrict__ wctx_i,
    const float* __restrict__ dsim, float* __restrict__ dwbuf, float* __restrict__ dsbuf, float* __restrict__ dwords,
    const int B, const int nef, const int R, const int RP, const int Lw, const float gamma1, const float gamma2,
    float* __restrict__ det_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const PairLds L = pair_lds(sm, nef, RP);
    const int i = blockIdx.x, j = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, g = lane >> 5;
    int T = (int)cap_lens[i];
    T = T < 1 ? 1 : (T > Lw ? Lw : T);
    const int64_t pair = (int64_t)j * B + i;
    float* const pwords = det_words ? det_words + ((int64_t)i * B + j) * nef * Lw : nullptr;
    const float gup = dsim[pair];
    const float zsum = expf(sim[pair]);
    float* scal = L.red + 4 * TP;           // [3][TP]: dcos/den, dcos cos/n2, dcos cos/n1   (red itself: [4][TP])
    // ---- per word: cosine -> d cos (one wave per word, lanes over the channels)
    for (int t = wid; t < T; t += 4) {
        float w12 = 0.f, n1 = 0.f, n2 = 0.f;
        for (int c = lane; c < nef; c += 64) {
            const float q = words[((int64_t)i * nef + c) * Lw + t], wv = wctx_i[(pair * Lw + t) * nef + c];
            w12 += q * wv; n1 += q * q; n2 += wv * wv;
        }
        w12 = wave_sum(w12); n1 = wave_sum(n1); n2 = wave_sum(n2);
        if (lane == 0) {
            const float den = sqrtf(n1) * sqrtf(n2);
            const bool clamped = den < 1e-8f;
            const float cosv = w12 / fmaxf(den, 1e-8f);
            const float dcos = gup * gamma2 * expf(gamma2 * cosv) / zsum;
            scal[t] = dcos / fmaxf(den, 1e-8f);
            scal[TP + t] = clamped ? 0.f : dcos * cosv / n2;
            scal[2 * TP + t] = clamped ? 0.f : dcos * cosv / n1;
        }
    }
    __syncthreads();
    // ---- dwctx[t][c] = ka q - kb wctx -> LDS (hi + lo) and scratch (f32); direct word gradient ka wctx - kc q
    for (int k = tid; k < TP * nef; k += MF_NT) {
        const int t = k / nef, c = k - t * nef;
        float dw = 0.f;
        if (t < T) {
            const float q = words[((int64_t)i * nef + c) * Lw + t], wv = wctx_i[(pair * Lw + t) * nef + c];
            dw = scal[t] * q - scal[TP + t] * wv;
            dwbuf[(pair * Lw + t) * nef + c] = dw;
            const float dq = scal[t] * wv - scal[2 * TP + t] * q;
            if (pwords) pwords[c * Lw + t] = dq;
            else if (dwords) atomicAdd(&dwords[((int64_t)i * nef + c) * Lw + t], dq);
        } else if (pwords && t < Lw) {
            pwords[c * Lw + t] = 0.f;
        }
        const bf16_t h = f2bf(dw);
        *reinterpret_cast<bf16_t*>(L.xh + t * L.XS + c * 2) = h;
        *reinterpret_cast<bf16_t*>(L.xl + t * L.XS + c * 2) = f2bf(dw - bf2f(h));
    }
    __syncthreads();
    // ---- dA[t][r] = sum_c dwctx[t][c] F[r][c];  dot[t] = sum_r A dA
    const int ntile = RP / 32;
    float dA[MAX_RT][16], Av[MAX_RT][16], pd[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) pd[e] = 0.f;
    const unsigned char* xh = L.xh + col * L.XS + 16 * g;
    const unsigned char* xl = L.xl + col * L.XS + 16 * g;
#pragma unroll
    for (int u = 0; u < MAX_RT; ++u) {
        const int rt = wid + 4 * u;
#pragma unroll
        for (int e = 0; e < 16; ++e) { dA[u][e] = 0.f; Av[u][e] = 0.f; }
        if (rt < ntile) {
            const int r = rt * 32 + col;
            const int64_t frow = ((int64_t)j * RP + r) * nef + 8 * g;
            const f32x16_t acc = score_tile(P.Fh + frow, P.Fl + frow, xh, xl, nef);
            if (r < R) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int t = arow(e, g);
                    if (t < T) {
                        Av[u][e] = attn[(pair * Lw + t) * R + r];
                        dA[u][e] = acc[e];
                        pd[e] += Av[u][e] * acc[e];
                    }
                }
            }
        }
    }
    half_wave_sum16(pd);
    if (col == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) L.red[wid * TP + arow(e, g)] = pd[e];
    }
    __syncthreads();
    float dot[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int t = arow(e, g);
        dot[e] = L.red[t] + L.red[TP + t] + L.red[2 * TP + t] + L.red[3 * TP + t];
    }
    // ---- dz = A (dA - dot); da1 = gamma1 dz; dS = a1 (da1 - sum_t a1 da1)
#pragma unroll
    for (int u = 0; u < MAX_RT; ++u) {
        const int rt = wid + 4 * u;
        if (rt < ntile) {
            const int r = rt * 32 + col;
            const bool rv = r < R;
            float a1[16], da1[16], d1 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = arow(e, g);
                a1[e] = (rv && t < T) ? attn1[(pair * Lw + t) * R + r] : 0.f;
                da1[e] = gamma1 * Av[u][e] * (dA[u][e] - dot[e]);
                d1 += a1[e] * da1[e];
            }
            d1 += __shfl_xor(d1, 32, 64);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = arow(e, g);
                const float ds = a1[e] * (da1[e] - d1);
                if (rv && t < T) dsbuf[(pair * Lw + t) * R + r] = ds;
                if (dwords || pwords) {
                    const bf16_t h = f2bf(ds);
                    *reinterpret_cast<bf16_t*>(L.ah + t * L.AS + r * 2) = h;
                    *reinterpret_cast<bf16_t*>(L.al + t * L.AS + r * 2) = f2bf(ds - bf2f(h));
                }
            }
        }
    }
    if (!(dwords || pwords)) return;
    // ---- d(words)[c][t] += sum_r dS[t][r] F[r][c]   (DAMSM pre-training: the text side has a gradient)
    __syncthreads();
    {
        const unsigned char* ah = L.ah + col * L.AS + 16 * g;
        const unsigned char* al = L.al + col * L.AS + 16 * g;
        const int t = col;
        for (int ct = wid; ct < nef / 32; ct += 4) {
            const int64_t trow = ((int64_t)j * nef + ct * 32 + col) * RP + 8 * g;
            const f32x16_t acc = ctx_tile(P.FTh + trow, P.FTl + trow, ah, al, RP);
            if (t < T) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = ct * 32 + arow(e, g);
                    // (deterministic mode: the direct term was stored before the barriers above, by another thread)
                    if (pwords) pwords[c * Lw + t] += acc[e];
                    else atomicAdd(&dwords[((int64_t)i * nef + c) * Lw + t], acc[e]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dfeat[j][c][r] += sum_i sum_t ( dwctx[(j,i)][t][c] A[(j,i)][t][r] + q[i][t][c] dS[(j,i)][t][r] ): wave = one
// 32 x 32 (channel x region) tile, K = (caption, word) walked in 16-word steps
__global__ __launch_bounds__(MF_NT) void damsm_words_bwd2_mfma_kernel(
    const Prep P, const int64_t* __restrict__ cap_lens, const float* __restrict__ attn, const float* __restrict__ dwbuf,
    const float* __restrict__ dsbuf, float* __restrict__ dfeat, const int B, const int nef, const int R, const int Lw) {
    const int j = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, g = lane >> 5;
    const int ct = blockIdx.y * 4 + wid;
    if (ct >= nef / 32) return;
    const int c = ct * 32 + col;                // A-operand row of this lane
    const int r = blockIdx.x * 32 + col;        // B-operand column of this lane
    const bool rv = r < R;
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int i = 0; i < B; ++i) {
        int T = (int)cap_lens[i];
        T = T < 1 ? 1 : (T > Lw ? Lw : T);
        const int64_t pair = (int64_t)j * B + i;
        for (int s = 0; 16 * s < T; ++s) {
            const int t0 = 16 * s + 8 * g;
            float xa[8], ya[8], yd[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int t = t0 + k;
                const bool tv = t < T;
                xa[k] = tv ? dwbuf[(pair * Lw + t) * nef + c] : 0.f;
                ya[k] = (tv && rv) ? attn[(pair * Lw + t) * R + r] : 0.f;
                yd[k] = (tv && rv) ? dsbuf[(pair * Lw + t) * R + r] : 0.f;
            }
            const bf16x8_t qh = *reinterpret_cast<const bf16x8_t*>(P.QTh + ((int64_t)i * nef + c) * TP + t0);
            const bf16x8_t ql = *reinterpret_cast<const bf16x8_t*>(P.QTl + ((int64_t)i * nef + c) * TP + t0);
            bf16x8_t xh, xl, yh, yl, zh, zl;
            split8(xa, xh, xl);
            split8(ya, yh, yl);
            split8(yd, zh, zl);
            acc = mma3(xh, xl, yh, yl, acc);
            acc = mma3(qh, ql, zh, zl, acc);
        }
    }
    if (rv) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float* p = dfeat + ((int64_t)j * nef + ct * 32 + arow(e, g)) * R + r;
            *p += acc[e];
        }
    }
}

inline bool mf_ok(int B, int nef, int R, int L) {
    return B > 0 && B <= 1024 && nef > 0 && nef % 64 == 0 && nef <= 1024 && R > 0 && R <= 32 * 4 * MAX_RT && L > 0 && L <= TP;
}
inline int round32(int v) { return (v + 31) / 32 * 32; }

}  // namespace

extern "C" int64_t sba_damsm_prep_bytes(int B, int nef, int R, int L) {
    if (!mf_ok(B, nef, R, L)) return -1;
    return 2 * (4 * prep_elems_feat(B, nef, round32(R)) + 4 * prep_elems_word(B, nef));
}

extern "C" int sba_damsm_prep(const float* feat, const float* words, const int64_t* cap_lens, void* prep,
                              int64_t prep_bytes, int B, int nef, int R, int L, void* stream) {
    if (!feat || !words || !cap_lens || !prep || !mf_ok(B, nef, R, L)) return SBA_E_ARG;
    if (prep_bytes < sba_damsm_prep_bytes(B, nef, R, L) || ((uintptr_t)prep & 15)) return SBA_E_ARG;
    const int RP = round32(R);
    const Prep v = prep_views(prep, B, nef, RP);
    hipStream_t st = (hipStream_t)stream;
    SBA_LAUNCH(damsm_prep_kernel, dim3(RP / 32, nef / 32, B), dim3(256), 0, st, feat, (const int64_t*)nullptr,
               (bf16_t*)v.Fh, (bf16_t*)v.Fl, (bf16_t*)v.FTh, (bf16_t*)v.FTl, nef, R, RP);
    SBA_LAUNCH(damsm_prep_kernel, dim3(1, nef / 32, B), dim3(256), 0, st, words, cap_lens, (bf16_t*)v.Qh, (bf16_t*)v.Ql,
               (bf16_t*)v.QTh, (bf16_t*)v.QTl, nef, L, TP);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_damsm_words_fwd_mfma(const void* prep, const float* words, const int64_t* cap_lens, float* sim,
                                        float* attn, float* attn1, float* wctx, int B, int nef, int R, int L,
                                        float gamma1, float gamma2, void* stream) {
    if (!prep || !words || !cap_lens || !sim || !attn || !attn1 || !wctx || !mf_ok(B, nef, R, L)) return SBA_E_ARG;
    const int RP = round32(R);
    const size_t sh = pair_lds_bytes(nef, RP);
    if (sh > 160 * 1024) return SBA_E_ARG;
    (void)hipFuncSetAttribute((const void*)damsm_words_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    SBA_LAUNCH(damsm_words_fwd_mfma_kernel, dim3(B, B), dim3(MF_NT), sh, (hipStream_t)stream, prep_views(prep, B, nef, RP),
               words, cap_lens, sim, attn, attn1, wctx, B, nef, R, RP, L, gamma1, gamma2);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_damsm_words_bwd_mfma(const void* prep, const float* words, const int64_t* cap_lens, const float* sim,
                                        const float* attn, const float* attn1, const float* wctx, const float* dsim,
                                        float* dwbuf, float* dsbuf, float* dfeat, float* dwords, int B, int nef, int R,
                                        int L, float gamma1, float gamma2, void* stream) {
    if (!prep || !words || !cap_lens || !sim || !attn || !attn1 || !wctx || !dsim || !dwbuf || !dsbuf || !dfeat ||
        !mf_ok(B, nef, R, L))
        return SBA_E_ARG;
    const int RP = round32(R);
    const size_t sh = pair_lds_bytes(nef, RP);
    if (sh > 160 * 1024) return SBA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* pw = nullptr;
    if (sba_det_on() && dwords) {
        pw = sba_det_alloc((int64_t)B * B * nef * L);
        if (!pw) return SBA_E_ARG;
    }
    const Prep v = prep_views(prep, B, nef, RP);
    (void)hipFuncSetAttribute((const void*)damsm_words_bwd1_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    SBA_LAUNCH(damsm_words_bwd1_mfma_kernel, dim3(B, B), dim3(MF_NT), sh, st, v, words, cap_lens, sim, attn, attn1, wctx,
               dsim, dwbuf, dsbuf, dwords, B, nef, R, RP, L, gamma1, gamma2, pw);
    SBA_LAUNCH(damsm_words_bwd2_mfma_kernel, dim3(RP / 32, cdiv(nef / 32, 4), B), dim3(MF_NT), 0, st, v, cap_lens, attn,
               dwbuf, dsbuf, dfeat, B, nef, R, L);
    if (pw) sba_det_fold(pw, B, B, (int64_t)nef * L, dwords, (int64_t)nef * L, 0, st);
    return SBA_CHECK_LAUNCH();
}
