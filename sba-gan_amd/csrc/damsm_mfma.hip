// DAMSM words loss (miscc/losses.py:62-132 with func_attention, GlobalAttention.py:31-69) and sentence loss
// (losses.py:20-59) on the bf16 matrix cores.
//
// The f32 kernels of damsm.hip run the 289 x T x 256 contractions of every (caption i, image j) pair on the VALU and add
// every pair's d(features) with f32 atomics (29.6 M of them at B = 20): 184 + 318 us, undiluted on the critical chain of
// the step (generator forward -> image encoder -> THIS -> image encoder backward -> generator backward).  Here:
//
//   prep   f32 features [B][nef][R] and words [B][nef][L] -> bf16 hi + lo parts, laid out FRAGMENT-MAJOR: the 64 lanes'
//          16-byte MFMA fragments of one (32-row tile, 16-deep k-step) are 1 KB contiguous, so every operand load of the
//          kernels below is one fully coalesced wave instruction (row-major operands with 16 bytes per lane per row cost
//          32 cache lines per instruction: the first version of these kernels sat at 50 us each on that).
//          Every product below is xh*yh + xl*yh + xh*yl: ~16 mantissa bits per product, f32 sums; the xh*yl term of
//          the features is skipped when the prep pass found all their lo parts zero (bf16 compute path).
//   fwd    one workgroup per pair (i, j): scores S[t][r] (MFMA 32x32x16, words = rows, K = nef, regions = the lane axis:
//          softmax over words in-lane + ONE __shfl_xor 32, as csrc/attention.hip), x gamma1, softmax over regions
//          (a1 <= 1, so exp(gamma1 a1) needs no max; region sums: 5 shuffles + LDS), attended context
//          wctx[c][t] = sum_r f[c][r] A[t][r] (K = regions, attention through LDS as hi + lo), cosine, log-sum-exp.
//   bwd1   one workgroup per pair: dcos -> dwctx, dA = dwctx . F (the scores' loop), the two softmax backward passes in
//          registers -> dS; A, dS and dwctx leave as bf16 hi + lo fragments for pass 2; optional d(words) (pre-training).
//   bwd2   d(features)[j] = sum_{i,t} A dwctx + dS q as ONE contraction per image over K = (caption, word): 8 coalesced
//          loads + 6 MFMAs per 16 words, every output element has exactly one owner -- no atomics, no per-pair partial
//          tensors, deterministic by construction.
//   ce / sentence: the two cross entropies over the B x B score matrix with their LAMBDA-weighted gradient in one launch
//          (sba_ce_pair_direct), the whole sentence loss forward + backward in one launch (sba_damsm_sent_direct).
#include "common.h"

namespace {

constexpr int MF_NT = 256;          // 4 waves
constexpr int TP = 32;              // word axis padded to one MFMA tile

__device__ __forceinline__ f32x16_t mma3(const bf16x8_t ah, const bf16x8_t al, const bf16x8_t bh, const bf16x8_t bl,
                                         f32x16_t acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return acc;
}
// accumulator element e of lane (col, g) is row arow(e, g)
__device__ __forceinline__ int arow(int e, int g) { return (e & 3) + 8 * (e >> 2) + 4 * g; }

// fragment-major offset (elements) of element (row, k) of a [rows][K] operand: tiles of 32 rows, k-steps of 16,
// lane = ((k >> 3) & 1) * 32 + (row & 31), 8 elements per lane
__device__ __forceinline__ int64_t frag_off(const int row, const int k, const int ksteps) {
    return ((((int64_t)(row >> 5) * ksteps + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (row & 31)) << 3) + (k & 7);
}

// ---- prep: src [b][C][X] f32 (x contiguous, valid x < Xv, zero beyond) -> bf16 hi + lo:
//   D = operand with rows c, K = x (fragment-major, XP / 16 k-steps)
//   T = operand with rows x, K = c: fragment-major (tmode 1) or row-major [XP][C] (tmode 0: staged through LDS later)
__global__ __launch_bounds__(256) void damsm_prep_kernel(const float* __restrict__ src, const int64_t* __restrict__ lens,
                                                         bf16_t* __restrict__ Th, bf16_t* __restrict__ Tl,
                                                         bf16_t* __restrict__ Dh, bf16_t* __restrict__ Dl, int C, int X,
                                                         int XP, int tmode, int* __restrict__ flag, const int flag_mode) {
    // flag_mode 1: clear *flag (the words pass, launched first); 2: set it when any lo part is non-zero (features)
    __shared__ uint32_t tile[32][33];           // [c][x] : hi | lo << 16
    if (flag_mode == 1 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *flag = 0;
    bool anylo = false;
    const int b = blockIdx.z, c0 = blockIdx.y * 32, x0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    int Xv = X;
    if (lens) { Xv = (int)lens[b]; Xv = Xv < 1 ? 1 : (Xv > X ? X : Xv); }
    const int64_t dbase = (int64_t)b * C * XP, tbase = (int64_t)b * XP * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, x = x0 + tx;
        float v = 0.f;
        if (x < Xv) v = src[((int64_t)b * C + c) * X + x];
        const bf16_t h = f2bf(v), l = f2bf(v - bf2f(h));
        anylo = anylo || (l & 0x7fffu) != 0;
        tile[ty + 8 * k][tx] = (uint32_t)h | ((uint32_t)l << 16);
        const int64_t o = dbase + frag_off(c, x, XP / 16);
        Dh[o] = h;
        Dl[o] = l;
    }
    if (flag_mode == 2 && __syncthreads_or(anylo ? 1 : 0) && threadIdx.x == 0) *flag = 1;   // (identical racing stores)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xl = ty + 8 * k;
        const uint32_t w = tile[tx][xl];
        const int64_t o = tbase + (tmode ? frag_off(x0 + xl, c0 + tx, C / 16) : (int64_t)(x0 + xl) * C + c0 + tx);
        Th[o] = (bf16_t)(w & 0xffffu);
        Tl[o] = (bf16_t)(w >> 16);
    }
}

struct Prep {       // views into the prep scratch (all bf16, h = hi, l = lo part)
    const bf16_t *Fh, *Fl;      // features, rows = regions, K = channels  (fragment-major)
    const bf16_t *FTh, *FTl;    // features, rows = channels, K = regions  (fragment-major)
    const bf16_t *Qh, *Ql;      // words, [32][nef] row-major
    const bf16_t *QTh, *QTl;    // words, rows = channels, K = words       (fragment-major, 2 k-steps)
    const int* flo;             // != 0: some feature has a non-zero lo part (f32 features); 0: exact bf16 values
};
inline int64_t prep_elems_feat(int B, int nef, int RP) { return (int64_t)B * RP * nef; }
inline int64_t prep_elems_word(int B, int nef) { return (int64_t)B * TP * nef; }
inline Prep prep_views(const void* p, int B, int nef, int RP) {
    const bf16_t* b = (const bf16_t*)p;
    const int64_t nf = prep_elems_feat(B, nef, RP), nq = prep_elems_word(B, nef);
    Prep v;
    v.Fh = b; v.Fl = b + nf; v.FTh = b + 2 * nf; v.FTl = b + 3 * nf;
    v.Qh = b + 4 * nf; v.Ql = v.Qh + nq; v.QTh = v.Qh + 2 * nq; v.QTl = v.Qh + 3 * nq;
    v.flo = reinterpret_cast<const int*>(v.Qh + 4 * nq);
    return v;
}
struct Bwd {        // views into the backward scratch: pass 1 -> pass 2 operands (bf16 fragments, hi + lo)
    bf16_t *XAh, *XAl;          // dwctx, rows = channels, K = words   [pair][nef/32][2][64][8]
    bf16_t *YAh, *YAl;          // attention, K = words, cols = regions [pair][RP/32][2][64][8]
    bf16_t *YDh, *YDl;          // d(scores), same layout
};
inline int64_t bwd_elems(int B, int nef, int RP) { return (int64_t)B * B * TP * (2 * (int64_t)nef + 4 * (int64_t)RP); }
inline Bwd bwd_views(void* p, int B, int nef, int RP) {
    bf16_t* b = (bf16_t*)p;
    const int64_t nx = (int64_t)B * B * TP * nef, ny = (int64_t)B * B * TP * RP;
    Bwd v;
    v.XAh = b; v.XAl = b + nx; v.YAh = b + 2 * nx; v.YAl = v.YAh + ny; v.YDh = v.YAh + 2 * ny; v.YDl = v.YAh + 3 * ny;
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

// One 32 x 32 output tile of  acc[m][n] += sum_k X[m][k] Y[k][n]  with ONE operand streamed from global memory in
// fragment-major order (gh / gl: this lane's 8 elements of k-step 0; consecutive k-steps 512 elements apart) and the other
// read from an LDS matrix whose rows are this lane's row (lh / ll: row base + 16 g bytes; k-step s at + 32 s bytes).
// GLOBAL_IS_A: the global operand is the A (row) operand.  GLO: the global operand has a lo part.
// Chunks of 4 k-steps, the next chunk's loads in flight while the current one is multiplied.
struct StreamChunk { bf16x8_t h[4], l[4]; };
template <bool GLO>
__device__ __forceinline__ void stream_load(StreamChunk& c, const bf16_t* __restrict__ gh, const bf16_t* __restrict__ gl,
                                            const int k0, const int ksteps) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bool ok = k0 + s < ksteps;            // (wave-uniform)
        c.h[s] = ok ? *reinterpret_cast<const bf16x8_t*>(gh + (int64_t)(k0 + s) * 512) : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (GLO) c.l[s] = ok ? *reinterpret_cast<const bf16x8_t*>(gl + (int64_t)(k0 + s) * 512) : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    }
}
template <bool GLOBAL_IS_A, bool GLO>
__device__ __forceinline__ f32x16_t stream_mul(const StreamChunk& c, const unsigned char* lh, const unsigned char* ll,
                                               const int k0, const int ksteps, f32x16_t acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (k0 + s < ksteps) {
            const bf16x8_t xh = *reinterpret_cast<const bf16x8_t*>(lh + (k0 + s) * 32);
            const bf16x8_t xl = *reinterpret_cast<const bf16x8_t*>(ll + (k0 + s) * 32);
            if (GLOBAL_IS_A) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.h[s], xh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.h[s], xl, acc, 0, 0, 0);
                if (GLO) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.l[s], xh, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, c.h[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, c.h[s], acc, 0, 0, 0);
                if (GLO) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, c.l[s], acc, 0, 0, 0);
            }
        }
    }
    return acc;
}
template <bool GLOBAL_IS_A, bool GLO>
__device__ __forceinline__ f32x16_t stream_tile(const bf16_t* __restrict__ gh, const bf16_t* __restrict__ gl,
                                                const unsigned char* lh, const unsigned char* ll, const int ksteps) {
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    StreamChunk A, B;
    stream_load<GLO>(A, gh, gl, 0, ksteps);
    for (int k0 = 0; k0 < ksteps; k0 += 8) {
        if (k0 + 4 < ksteps) stream_load<GLO>(B, gh, gl, k0 + 4, ksteps);
        acc = stream_mul<GLOBAL_IS_A, GLO>(A, lh, ll, k0, ksteps, acc);
        if (k0 + 4 < ksteps) {
            if (k0 + 8 < ksteps) stream_load<GLO>(A, gh, gl, k0 + 8, ksteps);
            acc = stream_mul<GLOBAL_IS_A, GLO>(B, lh, ll, k0 + 4, ksteps, acc);
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
template <bool FLO>
__device__ __forceinline__ void damsm_words_fwd_body(unsigned char* sm,
    const Prep& P, const float* __restrict__ words, const int64_t* __restrict__ cap_lens, float* __restrict__ sim,
    float* __restrict__ attn, float* __restrict__ attn1, float* __restrict__ wctx_o, const int B, const int nef,
    const int R, const int RP, const int Lw, const float gamma1, const float gamma2) {
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
    const int ntile = RP / 32, ks = nef / 16, ksr = RP / 16;
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
            const int64_t fo = (((int64_t)j * ntile + rt) * ks * 64 + lane) * 8;
            const f32x16_t acc = stream_tile<false, FLO>(P.Fh + fo, P.Fl + fo, xh, xl, ks);
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
            const int64_t fo = (((int64_t)j * (nef / 32) + ct) * ksr * 64 + lane) * 8;
            const f32x16_t acc = stream_tile<true, FLO>(P.FTh + fo, P.FTl + fo, ah, al, ksr);
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
template <bool FLO>
__device__ __forceinline__ void damsm_words_bwd1_body(unsigned char* sm,
    const Prep& P, const Bwd& W, const float* __restrict__ words, const int64_t* __restrict__ cap_lens,
    const float* __restrict__ sim, const float* __restrict__ attn, const float* __restrict__ attn1,
    const float* __restrict__ wctx_i, const float* __restrict__ dsim, float* __restrict__ dwords,
    const int B, const int nef, const int R, const int RP, const int Lw, const float gamma1, const float gamma2,
    float* __restrict__ det_words) {
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
    for (int t = wid; t < TP; t += 4) {
        float w12 = 0.f, n1 = 0.f, n2 = 0.f;
        if (t < T) {
            for (int c = lane; c < nef; c += 64) {
                const float q = words[((int64_t)i * nef + c) * Lw + t], wv = wctx_i[(pair * Lw + t) * nef + c];
                w12 += q * wv; n1 += q * q; n2 += wv * wv;
            }
        }
        w12 = wave_sum(w12); n1 = wave_sum(n1); n2 = wave_sum(n2);
        if (lane == 0) {
            float ka = 0.f, kb = 0.f, kc = 0.f;
            if (t < T) {
                const float den = sqrtf(n1) * sqrtf(n2);
                const bool clamped = den < 1e-8f;
                const float cosv = w12 / fmaxf(den, 1e-8f);
                const float dcos = gup * gamma2 * expf(gamma2 * cosv) / zsum;
                ka = dcos / fmaxf(den, 1e-8f);
                kb = clamped ? 0.f : dcos * cosv / n2;
                kc = clamped ? 0.f : dcos * cosv / n1;
            }
            scal[t] = ka; scal[TP + t] = kb; scal[2 * TP + t] = kc;
        }
    }
    __syncthreads();
    // ---- dwctx[t][c] = ka q - kb wctx -> LDS (the A operand of dA) and pass 2's fragments; the direct word gradient
    //      ka wctx - kc q.  One item = 8 consecutive words of one channel (16 bytes of pass 2's A operand).
    {
        bf16_t* xah = W.XAh + pair * (int64_t)TP * nef;
        bf16_t* xal = W.XAl + pair * (int64_t)TP * nef;
        for (int k = tid; k < nef * 4; k += MF_NT) {
            const int oct = k / nef, c = k - oct * nef;         // (consecutive lanes: consecutive channels)
            const int t0 = 8 * oct;
            uint32_t ph[4] = {0u, 0u, 0u, 0u}, pl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u;
                float dw = 0.f;
                if (t < T) {
                    const float q = words[((int64_t)i * nef + c) * Lw + t], wv = wctx_i[(pair * Lw + t) * nef + c];
                    dw = scal[t] * q - scal[TP + t] * wv;
                    const float dq = scal[t] * wv - scal[2 * TP + t] * q;
                    if (pwords) pwords[c * Lw + t] = dq;
                    else if (dwords) atomicAdd(&dwords[((int64_t)i * nef + c) * Lw + t], dq);
                } else if (pwords && t < Lw) {
                    pwords[c * Lw + t] = 0.f;
                }
                const bf16_t h = f2bf(dw), l = f2bf(dw - bf2f(h));
                *reinterpret_cast<bf16_t*>(L.xh + t * L.XS + c * 2) = h;
                *reinterpret_cast<bf16_t*>(L.xl + t * L.XS + c * 2) = l;
                ph[u >> 1] |= (uint32_t)h << (16 * (u & 1));
                pl[u >> 1] |= (uint32_t)l << (16 * (u & 1));
            }
            // rows = channels, K = words: tile c >> 5, k-step oct >> 1, lane (oct & 1) * 32 + (c & 31)
            const int64_t o = ((((int64_t)(c >> 5) * 2 + (oct >> 1)) * 64 + (oct & 1) * 32 + (c & 31)) << 3);
            *reinterpret_cast<uint4*>(xah + o) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            *reinterpret_cast<uint4*>(xal + o) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        }
    }
    __syncthreads();
    // ---- dA[t][r] = sum_c dwctx[t][c] F[r][c];  dot[t] = sum_r A dA
    const int ntile = RP / 32, ks = nef / 16, ksr = RP / 16;
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
            const int64_t fo = (((int64_t)j * ntile + rt) * ks * 64 + lane) * 8;
            const f32x16_t acc = stream_tile<false, FLO>(P.Fh + fo, P.Fl + fo, xh, xl, ks);
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
    // ---- dz = A (dA - dot); da1 = gamma1 dz; dS = a1 (da1 - sum_t a1 da1); A and dS -> pass 2's fragments
    const bool need_words = dwords || pwords;
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
            // K = words, cols = regions: tile rt, k-step m >> 1 (words 8 m + 4 g + 0..3), lane (m & 1) * 32 + col,
            // element offset 4 g
            const int64_t yb = (pair * ntile + rt) * (int64_t)(2 * 64 * 8);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                uint32_t ah[2] = {0u, 0u}, al[2] = {0u, 0u}, dh[2] = {0u, 0u}, dl[2] = {0u, 0u};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * m + k;
                    const float ds = a1[e] * (da1[e] - d1), av = Av[u][e];
                    const bf16_t h = f2bf(ds), l = f2bf(ds - bf2f(h));
                    const bf16_t vh = f2bf(av), vl = f2bf(av - bf2f(vh));
                    if (need_words) {
                        const int t = arow(e, g);
                        *reinterpret_cast<bf16_t*>(L.ah + t * L.AS + r * 2) = h;
                        *reinterpret_cast<bf16_t*>(L.al + t * L.AS + r * 2) = l;
                    }
                    const int sft = 16 * (k & 1);
                    ah[k >> 1] |= (uint32_t)vh << sft; al[k >> 1] |= (uint32_t)vl << sft;
                    dh[k >> 1] |= (uint32_t)h << sft; dl[k >> 1] |= (uint32_t)l << sft;
                }
                const int64_t o = yb + ((((m >> 1) * 64) + (m & 1) * 32 + col) << 3) + 4 * g;
                *reinterpret_cast<uint2*>(W.YAh + o) = make_uint2(ah[0], ah[1]);
                *reinterpret_cast<uint2*>(W.YAl + o) = make_uint2(al[0], al[1]);
                *reinterpret_cast<uint2*>(W.YDh + o) = make_uint2(dh[0], dh[1]);
                *reinterpret_cast<uint2*>(W.YDl + o) = make_uint2(dl[0], dl[1]);
            }
        }
    }
    if (!need_words) return;
    // ---- d(words)[c][t] += sum_r dS[t][r] f[c][r]   (DAMSM pre-training: the text side has a gradient)
    __syncthreads();
    {
        const unsigned char* ah = L.ah + col * L.AS + 16 * g;
        const unsigned char* al = L.al + col * L.AS + 16 * g;
        const int t = col;
        for (int ct = wid; ct < nef / 32; ct += 4) {
            const int64_t fo = (((int64_t)j * (nef / 32) + ct) * ksr * 64 + lane) * 8;
            const f32x16_t acc = stream_tile<true, FLO>(P.FTh + fo, P.FTl + fo, ah, al, ksr);
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

__global__ __launch_bounds__(MF_NT, 2) void damsm_words_fwd_mfma_kernel(
    const Prep P, const float* __restrict__ words, const int64_t* __restrict__ cap_lens, float* __restrict__ sim,
    float* __restrict__ attn, float* __restrict__ attn1, float* __restrict__ wctx_o, const int B, const int nef,
    const int R, const int RP, const int Lw, const float gamma1, const float gamma2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    if (__builtin_amdgcn_readfirstlane(*P.flo))
        damsm_words_fwd_body<true>(sm, P, words, cap_lens, sim, attn, attn1, wctx_o, B, nef, R, RP, Lw, gamma1, gamma2);
    else
        damsm_words_fwd_body<false>(sm, P, words, cap_lens, sim, attn, attn1, wctx_o, B, nef, R, RP, Lw, gamma1, gamma2);
}

__global__ __launch_bounds__(MF_NT, 2) void damsm_words_bwd1_mfma_kernel(
    const Prep P, const Bwd W, const float* __restrict__ words, const int64_t* __restrict__ cap_lens,
    const float* __restrict__ sim, const float* __restrict__ attn, const float* __restrict__ attn1,
    const float* __restrict__ wctx_i, const float* __restrict__ dsim, float* __restrict__ dwords,
    const int B, const int nef, const int R, const int RP, const int Lw, const float gamma1, const float gamma2,
    float* __restrict__ det_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    if (__builtin_amdgcn_readfirstlane(*P.flo))
        damsm_words_bwd1_body<true>(sm, P, W, words, cap_lens, sim, attn, attn1, wctx_i, dsim, dwords, B, nef, R, RP, Lw,
                                    gamma1, gamma2, det_words);
    else
        damsm_words_bwd1_body<false>(sm, P, W, words, cap_lens, sim, attn, attn1, wctx_i, dsim, dwords, B, nef, R, RP, Lw,
                                     gamma1, gamma2, det_words);
}

// ---------------------------------------------------------------------------------------------------------------
// dfeat[j][c][r] = sum_i sum_t ( dwctx[(j,i)][t][c] A[(j,i)][t][r] + q[i][t][c] dS[(j,i)][t][r] ): wave = one
// 32 x 32 (channel x region) tile, K = (caption, 16 words): 8 coalesced 1 KB loads + 6 MFMAs per step, the next step's
// loads in flight while the current one is multiplied
struct Bwd2Step { bf16x8_t xh, xl, qh, ql, yh, yl, zh, zl; bool live; };

__global__ __launch_bounds__(MF_NT) void damsm_words_bwd2_mfma_kernel(
    const Prep P, const Bwd W, const int64_t* __restrict__ cap_lens, float* __restrict__ dfeat, const int B,
    const int nef, const int R, const int RP, const int Lw, const int accumulate) {
    const int j = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, g = lane >> 5;
    const int nct = nef / 32, ntile = RP / 32;
    const int ct = blockIdx.y * 4 + wid, rt = blockIdx.x;
    if (ct >= nct) return;
    const int r = rt * 32 + col;
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto load = [&](Bwd2Step& S, const int n) {
        const int i = n >> 1, s = n & 1;
        S.live = false;
        if (i >= B) return;
        int T = (int)cap_lens[i];
        T = T < 1 ? 1 : (T > Lw ? Lw : T);
        if (16 * s >= T) return;                            // (wave-uniform)
        S.live = true;
        const int64_t pair = (int64_t)j * B + i;
        const int64_t xo = (((pair * nct + ct) * 2 + s) * 64 + lane) * 8;
        const int64_t qo = ((((int64_t)i * nct + ct) * 2 + s) * 64 + lane) * 8;
        const int64_t yo = (((pair * ntile + rt) * 2 + s) * 64 + lane) * 8;
        S.xh = *reinterpret_cast<const bf16x8_t*>(W.XAh + xo);
        S.xl = *reinterpret_cast<const bf16x8_t*>(W.XAl + xo);
        S.qh = *reinterpret_cast<const bf16x8_t*>(P.QTh + qo);
        S.ql = *reinterpret_cast<const bf16x8_t*>(P.QTl + qo);
        S.yh = *reinterpret_cast<const bf16x8_t*>(W.YAh + yo);
        S.yl = *reinterpret_cast<const bf16x8_t*>(W.YAl + yo);
        S.zh = *reinterpret_cast<const bf16x8_t*>(W.YDh + yo);
        S.zl = *reinterpret_cast<const bf16x8_t*>(W.YDl + yo);
    };
    auto mul = [&](const Bwd2Step& S) {
        if (!S.live) return;
        acc = mma3(S.xh, S.xl, S.yh, S.yl, acc);
        acc = mma3(S.qh, S.ql, S.zh, S.zl, acc);
    };
    Bwd2Step S0, S1;
    load(S0, 0);
    for (int n = 0; n < 2 * B; n += 2) {
        load(S1, n + 1);
        mul(S0);
        load(S0, n + 2);
        mul(S1);
    }
    if (r < R) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float* p = dfeat + ((int64_t)j * nef + ct * 32 + arow(e, g)) * R + r;
            *p = accumulate ? *p + acc[e] : acc[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Two cross entropies (rows / columns) over the B x B score matrix, labels arange(B) (losses.py:113-130 / 47-57), AND the
// gradient of lam * (loss0 + loss1) w.r.t. the unscaled scores, in one launch:  loss_out[0] = lam (loss0 + loss1),
// dscore = lam (d0 + d1)  (sba_ce_pair + sba_combine2 + three scalar launches otherwise, all on the critical chain)
__global__ __launch_bounds__(256) void ce_pair_direct_kernel(const float* __restrict__ score, const uint8_t* __restrict__ mask,
                                                             const float scale, const float lam, float* __restrict__ loss_out,
                                                             float* __restrict__ dscore, const int B) {
    extern __shared__ float s[];      // [B*B] scaled masked scores, then [2*B] row/col lse
    float* lse = s + B * B;
    const int tid = threadIdx.x;
    for (int k = tid; k < B * B; k += blockDim.x) s[k] = (mask && mask[k]) ? -INFINITY : score[k] * scale;
    __syncthreads();
    for (int k = tid; k < 2 * B; k += blockDim.x) {
        const bool colw = k >= B;
        const int a = colw ? k - B : k;
        float mx = -INFINITY;
        for (int b = 0; b < B; ++b) mx = fmaxf(mx, colw ? s[b * B + a] : s[a * B + b]);
        float sum = 0.f;
        for (int b = 0; b < B; ++b) sum += expf((colw ? s[b * B + a] : s[a * B + b]) - mx);
        lse[k] = mx + logf(sum);
    }
    __syncthreads();
    if (tid == 0) {
        float l0 = 0.f, l1 = 0.f;
        for (int a = 0; a < B; ++a) { l0 += lse[a] - s[a * B + a]; l1 += lse[B + a] - s[a * B + a]; }
        loss_out[0] = (l0 / B + l1 / B) * lam;
    }
    const float invB = 1.f / B;
    for (int k = tid; k < B * B; k += blockDim.x) {
        const int j = k / B, i = k - j * B;
        const float dl = j == i ? 1.f : 0.f;
        const float d0 = (expf(s[k] - lse[j]) - dl) * invB * scale, d1 = (expf(s[k] - lse[B + i]) - dl) * invB * scale;
        dscore[k] = lam * d0 + lam * d1;
    }
}

// The whole sentence loss (losses.py:20-59) forward + backward to the image side in ONE launch (one workgroup):
// s[j][i] = gamma3 cos(cnn_j, rnn_i), the two cross entropies, loss_out[0] = lam (loss0 + loss1), and
// dcnn[j] = sum_i ds[j][i] d s[j][i] / d cnn_j (walked in caption order: deterministic).
__global__ __launch_bounds__(1024) void damsm_sent_direct_kernel(const float* __restrict__ cnn, const float* __restrict__ rnn,
                                                                 const uint8_t* __restrict__ mask, const float gamma3,
                                                                 const float eps, const float lam,
                                                                 float* __restrict__ loss_out, float* __restrict__ dcnn,
                                                                 const int B, const int nef) {
    extern __shared__ float sh[];
    float* s = sh;                  // [B*B] scores (masked: -inf)
    float* w12 = s + B * B;         // [B*B] raw dot products
    float* ds = w12 + B * B;        // [B*B] d loss / d s (w.r.t. the gamma3-scaled score)
    float* lse = ds + B * B;        // [2B]
    float* nc = lse + 2 * B;        // [B] |cnn_j|^2
    float* nr = nc + B;             // [B] |rnn_i|^2
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    for (int p = wid; p < 2 * B; p += nw) {             // squared norms, one wave per vector
        const float* v = p < B ? cnn + (int64_t)p * nef : rnn + (int64_t)(p - B) * nef;
        float a = 0.f;
        for (int c = lane; c < nef; c += 64) a += v[c] * v[c];
        a = wave_sum(a);
        if (lane == 0) { if (p < B) nc[p] = a; else nr[p - B] = a; }
    }
    for (int p = wid; p < B * B; p += nw) {             // dot products, one wave per pair
        const int j = p / B, i = p - j * B;
        float a = 0.f;
        for (int c = lane; c < nef; c += 64) a += cnn[(int64_t)j * nef + c] * rnn[(int64_t)i * nef + c];
        a = wave_sum(a);
        if (lane == 0) w12[p] = a;
    }
    __syncthreads();
    for (int p = tid; p < B * B; p += blockDim.x) {
        const int j = p / B, i = p - j * B;
        const float v = w12[p] / fmaxf(sqrtf(nc[j]) * sqrtf(nr[i]), eps) * gamma3;
        s[p] = (mask && mask[p]) ? -INFINITY : v;
    }
    __syncthreads();
    for (int k = tid; k < 2 * B; k += blockDim.x) {
        const bool colw = k >= B;
        const int a = colw ? k - B : k;
        float mx = -INFINITY;
        for (int b = 0; b < B; ++b) mx = fmaxf(mx, colw ? s[b * B + a] : s[a * B + b]);
        float sum = 0.f;
        for (int b = 0; b < B; ++b) sum += expf((colw ? s[b * B + a] : s[a * B + b]) - mx);
        lse[k] = mx + logf(sum);
    }
    __syncthreads();
    if (tid == 0) {
        float l0 = 0.f, l1 = 0.f;
        for (int a = 0; a < B; ++a) { l0 += lse[a] - s[a * B + a]; l1 += lse[B + a] - s[a * B + a]; }
        loss_out[0] = (l0 / B + l1 / B) * lam;
    }
    const float invB = 1.f / B;
    for (int p = tid; p < B * B; p += blockDim.x) {
        const int j = p / B, i = p - j * B;
        const float dl = j == i ? 1.f : 0.f;
        // (the same expression as sba_ce_pair at scale 1 + sba_combine2: the autograd path's bits)
        const float d0 = (expf(s[p] - lse[j]) - dl) * invB * 1.f, d1 = (expf(s[p] - lse[B + i]) - dl) * invB * 1.f;
        ds[p] = lam * d0 + lam * d1;
    }
    __syncthreads();
    if (!dcnn) return;
    // the two coefficients of every pair once (they were recomputed -- two square roots, three divisions -- for each of the
    // nef channels): k0 -> s[p], ka -> w12[p]
    for (int p = tid; p < B * B; p += blockDim.x) {
        const int j = p / B, i = p - j * B;
        const float den = sqrtf(nc[j]) * sqrtf(nr[i]);
        const bool clamped = den < eps;
        const float gq = ds[p] * gamma3;
        const float k0 = gq / fmaxf(den, eps);
        const float ka = clamped ? 0.f : gq * (w12[p] / den) / nc[j];
        s[p] = k0;
        w12[p] = ka;
    }
    __syncthreads();
    for (int o = tid; o < B * nef; o += blockDim.x) {
        const int j = o / nef, c = o - j * nef;
        const float a = cnn[o];
        float acc = 0.f;
        for (int i = 0; i < B; ++i) acc += s[j * B + i] * rnn[(int64_t)i * nef + c] - w12[j * B + i] * a;
        dcnn[o] = acc;
    }
}

inline bool mf_ok(int B, int nef, int R, int L) {
    return B > 0 && B <= 1024 && nef > 0 && nef % 64 == 0 && nef <= 1024 && R > 0 && R <= 32 * 4 * MAX_RT && L > 0 && L <= TP;
}
inline int round32(int v) { return (v + 31) / 32 * 32; }

}  // namespace

extern "C" int64_t sba_damsm_prep_bytes(int B, int nef, int R, int L) {
    if (!mf_ok(B, nef, R, L)) return -1;
    return 2 * (4 * prep_elems_feat(B, nef, round32(R)) + 4 * prep_elems_word(B, nef)) + 256;     // (+ the lo flag)
}

extern "C" int64_t sba_damsm_bwd_bytes(int B, int nef, int R, int L) {
    if (!mf_ok(B, nef, R, L)) return -1;
    return 2 * bwd_elems(B, nef, round32(R));
}

extern "C" int sba_damsm_prep(const float* feat, const float* words, const int64_t* cap_lens, void* prep,
                              int64_t prep_bytes, int B, int nef, int R, int L, void* stream) {
    if (!feat || !words || !cap_lens || !prep || !mf_ok(B, nef, R, L)) return SBA_E_ARG;
    if (prep_bytes < sba_damsm_prep_bytes(B, nef, R, L) || ((uintptr_t)prep & 15)) return SBA_E_ARG;
    const int RP = round32(R);
    const Prep v = prep_views(prep, B, nef, RP);
    hipStream_t st = (hipStream_t)stream;
    SBA_LAUNCH(damsm_prep_kernel, dim3(1, nef / 32, B), dim3(256), 0, st, words, cap_lens, (bf16_t*)v.Qh, (bf16_t*)v.Ql,
               (bf16_t*)v.QTh, (bf16_t*)v.QTl, nef, L, TP, 0, (int*)v.flo, 1);
    SBA_LAUNCH(damsm_prep_kernel, dim3(RP / 32, nef / 32, B), dim3(256), 0, st, feat, (const int64_t*)nullptr,
               (bf16_t*)v.Fh, (bf16_t*)v.Fl, (bf16_t*)v.FTh, (bf16_t*)v.FTl, nef, R, RP, 1, (int*)v.flo, 2);
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
                                        void* scratch, int64_t scratch_bytes, float* dfeat, int accumulate,
                                        float* dwords, int B, int nef, int R, int L, float gamma1, float gamma2,
                                        void* stream) {
    if (!prep || !words || !cap_lens || !sim || !attn || !attn1 || !wctx || !dsim || !scratch || !dfeat ||
        !mf_ok(B, nef, R, L))
        return SBA_E_ARG;
    if (scratch_bytes < sba_damsm_bwd_bytes(B, nef, R, L) || ((uintptr_t)scratch & 15)) return SBA_E_ARG;
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
    const Bwd w = bwd_views(scratch, B, nef, RP);
    (void)hipFuncSetAttribute((const void*)damsm_words_bwd1_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    SBA_LAUNCH(damsm_words_bwd1_mfma_kernel, dim3(B, B), dim3(MF_NT), sh, st, v, w, words, cap_lens, sim, attn, attn1, wctx,
               dsim, dwords, B, nef, R, RP, L, gamma1, gamma2, pw);
    SBA_LAUNCH(damsm_words_bwd2_mfma_kernel, dim3(RP / 32, cdiv(nef / 32, 4), B), dim3(MF_NT), 0, st, v, w, cap_lens, dfeat,
               B, nef, R, RP, L, accumulate);
    if (pw) sba_det_fold(pw, B, B, (int64_t)nef * L, dwords, (int64_t)nef * L, 0, st);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_ce_pair_direct(const float* score, const uint8_t* mask, float scale, float lam, float* loss_out,
                                  float* dscore, int B, void* stream) {
    if (!score || !loss_out || !dscore || B <= 0 || B > 96) return SBA_E_ARG;
    const size_t sh = sizeof(float) * ((size_t)B * B + 2 * B);
    SBA_LAUNCH(ce_pair_direct_kernel, dim3(1), dim3(256), sh, (hipStream_t)stream, score, mask, scale, lam, loss_out, dscore, B);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_damsm_sent_direct(const float* cnn, const float* rnn, const uint8_t* mask, float gamma3, float eps,
                                     float lam, float* loss_out, float* dcnn, int B, int nef, void* stream) {
    if (!cnn || !rnn || !loss_out || B <= 0 || B > 96 || nef <= 0) return SBA_E_ARG;
    const size_t sh = sizeof(float) * ((size_t)3 * B * B + 4 * B);
    if (sh > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)damsm_sent_direct_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    }
    SBA_LAUNCH(damsm_sent_direct_kernel, dim3(1), dim3(1024), sh, (hipStream_t)stream, cnn, rnn, mask, gamma3, eps, lam,
               loss_out, dcnn, B, nef);
    return SBA_CHECK_LAUNCH();
}
