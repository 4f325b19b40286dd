// DAMSM word / sentence matching loss (miscc/losses.py:11-132 with func_attention,
// GlobalAttention.py:31-69) for gfx950, f32 throughout.
//
// The reference loops over captions in Python and runs ~25 tiny kernels per caption;
// here one workgroup handles one (caption i, image j) pair end to end: region-word
// scores (289 x T), softmax over words, x gamma1, softmax over regions, attended
// context (nef x T), per-word cosine, log-sum-exp.  cap_lens is read on the device
// (no .tolist() sync).  Work is ~5 MFLOP per pair: latency/launch bound, so the
// win is the fusion; the 289 x T x 256 contractions run on the VALU from LDS.
#include "common.h"

namespace {

constexpr int NT = 320;       // threads per pair: >= R (289), 5 waves
constexpr int CCH = 16;       // feature channels staged per LDS chunk (sized so that two pairs fit one CU)
constexpr int TMAX = 32;

// The word axis of the LDS-resident [nef][words] matrices is padded to a COMPILE-TIME stride LP (20 for the
// 18-word captions of the reference configs, else 32) and zero-filled beyond the caption length, so that
// the nef x words inner products are unconditional, fully unrolled FMA runs fed by ds_read_b128.  (With a
// run-time word count every single FMA became branch + ds_read_b32 + full LDS wait: ~10x slower.)
struct Lds {
    float* q;     // [nef][LP]   words of caption i (zero beyond T)
    float* z;     // [Lw][R]     attention (scratch / A)
    float* w;     // [nef][LP]   weighted context (fwd) / dwctx (bwd)
    float* f;     // [CCH][R]    staged feature chunk
    float* t;     // [4*TMAX]    per-word scalars
};
__device__ __forceinline__ Lds carve(float* sm, int nef, int LP, int Lw, int R) {
    Lds l;
    l.q = sm;
    l.z = l.q + nef * LP;
    l.w = l.z + ((Lw * R + 3) & ~3);
    l.f = l.w + nef * LP;
    l.t = l.f + CCH * R;
    return l;
}
inline size_t lds_bytes(int nef, int LP, int Lw, int R) {
    const size_t zf = ((size_t)Lw * R + 3) & ~(size_t)3;       // keep the later arrays 16-byte aligned
    return sizeof(float) * ((size_t)2 * nef * LP + zf + (size_t)CCH * R + 4 * TMAX);
}

// ---------------------------------------------------------------------------
template <int LP>
__global__ __launch_bounds__(NT) void damsm_words_fwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ words, const int64_t* __restrict__ cap_lens,
    float* __restrict__ sim, float* __restrict__ attn, float* __restrict__ attn1, float* __restrict__ wctx_o,
    int B, int nef, int R, int Lw, float gamma1, float gamma2) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    Lds L = carve(sm, nef, LP, Lw, R);
    const int i = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6, nw = NT / 64;
    int T = (int)cap_lens[i];
    T = T < 1 ? 1 : (T > Lw ? Lw : T);
    const int64_t pair = (int64_t)j * B + i;
    const float* fj = feat + (int64_t)j * nef * R;

    for (int k = tid; k < nef * LP; k += NT) {
        const int c = k / LP, t = k - c * LP;
        L.q[k] = t < T ? words[((int64_t)i * nef + c) * Lw + t] : 0.f;
    }
    __syncthreads();

    // phase 1: S[r][t] = sum_c f[c][r] q[c][t]; softmax over t; x gamma1
    float s[LP];
#pragma unroll
    for (int t = 0; t < LP; ++t) s[t] = 0.f;
    if (tid < R) {
        // 8 independent loads in flight per thread (a dependent load -> FMA chain made this
        // phase pure latency)
        for (int c0 = 0; c0 < nef; c0 += 8) {
            float fv8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) fv8[k] = fj[(int64_t)(c0 + k) * R + tid];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4* qr = reinterpret_cast<const float4*>(&L.q[(c0 + k) * LP]);
#pragma unroll
                for (int t4 = 0; t4 < LP / 4; ++t4) {
                    const float4 qv = qr[t4];
                    s[4 * t4] += fv8[k] * qv.x; s[4 * t4 + 1] += fv8[k] * qv.y;
                    s[4 * t4 + 2] += fv8[k] * qv.z; s[4 * t4 + 3] += fv8[k] * qv.w;
                }
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < LP; ++t)
            if (t < T) mx = fmaxf(mx, s[t]);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < LP; ++t) {
            s[t] = t < T ? expf(s[t] - mx) : 0.f;
            sum += s[t];
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int t = 0; t < LP; ++t)
            if (t < T) {
                const float a1 = s[t] * inv;
                attn1[(pair * Lw + t) * R + tid] = a1;
                L.z[t * R + tid] = a1 * gamma1;
            }
    }
    __syncthreads();
    // phase 2: softmax over regions for each word (one wave per word)
    for (int t = wid; t < T; t += nw) {
        float mx = -INFINITY;
        for (int r = lane; r < R; r += 64) mx = fmaxf(mx, L.z[t * R + r]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int r = lane; r < R; r += 64) {
            const float e = expf(L.z[t * R + r] - mx);
            L.z[t * R + r] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int r = lane; r < R; r += 64) {
            const float a = L.z[t * R + r] * inv;
            L.z[t * R + r] = a;
            attn[(pair * Lw + t) * R + r] = a;
        }
    }
    __syncthreads();
    // phase 3: wctx[c][t] = sum_r f[c][r] A[t][r] on the f32 matrix cores: M = 32 channels per tile, N = words
    // (32 columns, T valid), K = regions.  The feature rows are read straight from L2 (296 KB per image,
    // shared by the B workgroups of that image), the attention from LDS; 8 loads per lane in flight.
    // (As a VALU loop with two LDS reads per FMA this phase was 78 % of the kernel.)
    {
        const int rl = lane & 31, hf = lane >> 5;
        const bool tv = rl < T;
        const int steps = (R + 1) / 2;
        for (int ct = wid; ct < nef / 32; ct += nw) {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* fr = fj + (int64_t)(ct * 32 + rl) * R + hf;
            const float* zr = L.z + rl * R + hf;
            for (int kk0 = 0; kk0 < steps; kk0 += 8) {
                float av[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = kk0 + u;
                    const bool ok = kk < steps && 2 * kk + hf < R;
                    av[u] = ok ? fr[2 * kk] : 0.f;
                    bv[u] = (ok && tv) ? zr[2 * kk] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
            }
            if (tv) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    L.w[c * LP + rl] = acc[r];
                    wctx_o[(pair * Lw + rl) * nef + c] = acc[r];
                }
            }
        }
        __syncthreads();
    }
    // phase 4: cosine per word, then log-sum-exp
    for (int t = wid; t < T; t += nw) {
        float w12 = 0.f, n1 = 0.f, n2 = 0.f;
        for (int c = lane; c < nef; c += 64) {
            const float qv = L.q[c * LP + t], wv = L.w[c * LP + t];
            w12 += qv * wv; n1 += qv * qv; n2 += wv * wv;
        }
        w12 = wave_sum(w12); n1 = wave_sum(n1); n2 = wave_sum(n2);
        if (lane == 0) L.t[t] = w12 / fmaxf(sqrtf(n1) * sqrtf(n2), 1e-8f);
    }
    __syncthreads();
    if (tid == 0) {
        float z = 0.f;
        for (int t = 0; t < T; ++t) z += expf(gamma2 * L.t[t]);
        sim[pair] = logf(z);
    }
}

// ---------------------------------------------------------------------------
template <int LP>
__global__ __launch_bounds__(NT) void damsm_words_bwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ words, const int64_t* __restrict__ cap_lens,
    const float* __restrict__ sim, const float* __restrict__ attn, const float* __restrict__ attn1,
    const float* __restrict__ wctx_i, const float* __restrict__ dsim, float* __restrict__ dfeat,
    float* __restrict__ dwords, int B, int nef, int R, int Lw, float gamma1, float gamma2,
    float* __restrict__ det_feat, float* __restrict__ det_words) {
    // deterministic mode: pair (i, j) owns slot [j][i] of det_feat ([nef][R]) and slot [i][j] of det_words
    // ([nef][Lw]) and fills them with plain stores; sba_det_fold adds the slots of an image / a caption in order
    extern __shared__ __attribute__((aligned(16))) float sm[];
    Lds L = carve(sm, nef, LP, Lw, R);
    const int i = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    float* const pfeat = det_feat ? det_feat + ((int64_t)j * B + i) * nef * R : nullptr;
    float* const pwords = det_words ? det_words + ((int64_t)i * B + j) * nef * Lw : nullptr;
    const int lane = tid & 63, wid = tid >> 6, nw = NT / 64;
    int T = (int)cap_lens[i];
    T = T < 1 ? 1 : (T > Lw ? Lw : T);
    const int64_t pair = (int64_t)j * B + i;
    const float* fj = feat + (int64_t)j * nef * R;
    const float g = dsim[pair];
    const float zsum = expf(sim[pair]);

    for (int k = tid; k < nef * LP; k += NT) {
        const int c = k / LP, t = k - c * LP;
        L.q[k] = t < T ? words[((int64_t)i * nef + c) * Lw + t] : 0.f;
        L.w[k] = t < T ? wctx_i[(pair * Lw + t) * nef + c] : 0.f;
    }
    __syncthreads();
    // per word: cos, norms -> dcos; scalars kept in L.t: [t]=dcos/(n1 n2), [TMAX+t]=dcos*cos/n2^2,
    // [2TMAX+t]=dcos*cos/n1^2 (for the direct word gradient), [3TMAX+t]=clamped flag
    for (int t = wid; t < T; t += nw) {
        float w12 = 0.f, n1 = 0.f, n2 = 0.f;
        for (int c = lane; c < nef; c += 64) {
            const float qv = L.q[c * LP + t], wv = L.w[c * LP + t];
            w12 += qv * wv; n1 += qv * qv; n2 += wv * wv;
        }
        w12 = wave_sum(w12); n1 = wave_sum(n1); n2 = wave_sum(n2);
        if (lane == 0) {
            const float den = sqrtf(n1) * sqrtf(n2);
            const bool clamped = den < 1e-8f;
            const float cosv = w12 / fmaxf(den, 1e-8f);
            const float dcos = g * gamma2 * expf(gamma2 * cosv) / zsum;
            L.t[t] = dcos / fmaxf(den, 1e-8f);
            L.t[TMAX + t] = clamped ? 0.f : dcos * cosv / n2;
            L.t[2 * TMAX + t] = clamped ? 0.f : dcos * cosv / n1;
        }
    }
    __syncthreads();
    // dq_direct[c][t] (kept in registers of the owning thread is not possible: write to dwords now),
    // then overwrite L.w with dwctx[c][t] = a*q - b*wctx
    for (int k = tid; k < nef * LP; k += NT) {
        const int c = k / LP, t = k - c * LP;
        if (t < T) {
            const float qv = L.q[k], wv = L.w[k];
            if (pwords) pwords[c * Lw + t] = L.t[t] * wv - L.t[2 * TMAX + t] * qv;
            else if (dwords)
                atomicAdd(&dwords[((int64_t)i * nef + c) * Lw + t], L.t[t] * wv - L.t[2 * TMAX + t] * qv);
            L.w[k] = L.t[t] * qv - L.t[TMAX + t] * wv;
        } else {
            if (pwords && t < Lw) pwords[c * Lw + t] = 0.f;
            L.w[k] = 0.f;
        }
    }
    __syncthreads();
    // pass A over the features: dA[t][r] = sum_c dwctx[c][t] f[c][r]
    float dA[LP];
#pragma unroll
    for (int t = 0; t < LP; ++t) dA[t] = 0.f;
    if (tid < R) {
        for (int c0 = 0; c0 < nef; c0 += 8) {
            float fv8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) fv8[k] = fj[(int64_t)(c0 + k) * R + tid];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4* wr = reinterpret_cast<const float4*>(&L.w[(c0 + k) * LP]);
#pragma unroll
                for (int t4 = 0; t4 < LP / 4; ++t4) {
                    const float4 wv = wr[t4];
                    dA[4 * t4] += fv8[k] * wv.x; dA[4 * t4 + 1] += fv8[k] * wv.y;
                    dA[4 * t4 + 2] += fv8[k] * wv.z; dA[4 * t4 + 3] += fv8[k] * wv.w;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < LP; ++t)
            if (t < T) L.z[t * R + tid] = dA[t] * attn[(pair * Lw + t) * R + tid];    // A*dA
    }
    __syncthreads();
    // per word: dot_t = sum_r A dA
    for (int t = wid; t < T; t += nw) {
        float d = 0.f;
        for (int r = lane; r < R; r += 64) d += L.z[t * R + r];
        d = wave_sum(d);
        if (lane == 0) L.t[3 * TMAX + t] = d;
    }
    __syncthreads();
    // per region: dz = A (dA - dot), da1 = gamma1 dz, dS = a1 (da1 - sum_t a1 da1); keep A and dS in registers
    float av[LP], ds[LP];
#pragma unroll
    for (int t = 0; t < LP; ++t) { av[t] = 0.f; ds[t] = 0.f; }
    if (tid < R) {
        float dot1 = 0.f;
#pragma unroll
        for (int t = 0; t < LP; ++t)
            if (t < T) {
                const float a = attn[(pair * Lw + t) * R + tid];
                const float a1 = attn1[(pair * Lw + t) * R + tid];
                const float da1 = gamma1 * a * (dA[t] - L.t[3 * TMAX + t]);
                av[t] = a;
                ds[t] = da1;          // temporarily da1
                dA[t] = a1;           // reuse dA[] for a1
                dot1 += a1 * da1;
            }
#pragma unroll
        for (int t = 0; t < LP; ++t) {
            ds[t] = t < T ? dA[t] * (ds[t] - dot1) : 0.f;
            if (t < T) L.z[t * R + tid] = ds[t];
        }
    }
    __syncthreads();
    // pass B: dfeat[c][r] += sum_t dwctx[c][t] A[t][r] + q[c][t] dS[t][r] on the f32 matrix cores: 32 x 32
    // (channel x region) tiles, K = the (<= LP) words, two MFMA chains per step.  A[t][r] comes from the
    // attention the forward stored (L2-resident, coalesced over regions), dS from LDS.
    {
        const int rl = lane & 31, hf = lane >> 5;
        const int rtiles = (R + 31) / 32;
        for (int tl = wid; tl < (nef / 32) * rtiles; tl += nw) {
            const int ct = tl / rtiles, rt = tl - ct * rtiles;
            const int r = rt * 32 + rl;
            const bool rv = r < R;
            f32x16_t acc;
#pragma unroll
            for (int q2 = 0; q2 < 16; ++q2) acc[q2] = 0.f;
            const float* wr = L.w + (ct * 32 + rl) * LP + hf;
            const float* qr = L.q + (ct * 32 + rl) * LP + hf;
            float a1[LP / 2], b1[LP / 2], b2[LP / 2];
#pragma unroll
            for (int kk = 0; kk < LP / 2; ++kk) {
                const int t = 2 * kk + hf;
                const bool ok = rv && t < T;
                b1[kk] = ok ? attn[(pair * Lw + t) * R + r] : 0.f;
                b2[kk] = ok ? L.z[t * R + r] : 0.f;
                a1[kk] = wr[2 * kk];
            }
#pragma unroll
            for (int kk = 0; kk < LP / 2; ++kk) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b1[kk], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[2 * kk], b2[kk], acc, 0, 0, 0);
            }
            if (rv) {
#pragma unroll
                for (int q2 = 0; q2 < 16; ++q2) {
                    const int c = ct * 32 + (q2 & 3) + 8 * (q2 >> 2) + 4 * hf;
                    if (pfeat) pfeat[c * R + r] = acc[q2];
                    else atomicAdd(&dfeat[((int64_t)j * nef + c) * R + r], acc[q2]);
                }
            }
        }
    }
    if (dwords) {
        __syncthreads();
        // dq[c][t] += sum_r f[c][r] dS[r][t], features staged through LDS
        for (int c0 = 0; c0 < nef; c0 += CCH) {
            for (int k = tid; k < CCH * R; k += NT) L.f[k] = fj[(int64_t)c0 * R + k];
            __syncthreads();
            for (int o = tid; o < CCH * T; o += NT) {
                const int cl = o / T, t = o - cl * T;
                float acc = 0.f;
                for (int r = 0; r < R; ++r) acc += L.f[cl * R + r] * L.z[t * R + r];
                // (deterministic mode: the direct term was stored before the barriers above, by another thread)
                if (pwords) pwords[(c0 + cl) * Lw + t] += acc;
                else atomicAdd(&dwords[((int64_t)i * nef + c0 + cl) * Lw + t], acc);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// sentence scores: s[j][i] = gamma3 * <cnn_j, rnn_i> / max(|cnn_j| |rnn_i|, eps)
__global__ void damsm_sent_fwd_kernel(const float* __restrict__ cnn, const float* __restrict__ rnn,
                                      float* __restrict__ s, int B, int nef, float gamma3, float eps) {
    const int i = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    float w12 = 0.f, n0 = 0.f, n1 = 0.f;
    for (int c = lane; c < nef; c += 64) {
        const float a = cnn[j * nef + c], b = rnn[i * nef + c];
        w12 += a * b; n0 += a * a; n1 += b * b;
    }
    w12 = wave_sum(w12); n0 = wave_sum(n0); n1 = wave_sum(n1);
    if (lane == 0) s[j * B + i] = w12 / fmaxf(sqrtf(n0) * sqrtf(n1), eps) * gamma3;
}

__global__ void damsm_sent_bwd_kernel(const float* __restrict__ cnn, const float* __restrict__ rnn,
                                      const float* __restrict__ ds, float* __restrict__ dcnn,
                                      float* __restrict__ drnn, int B, int nef, float gamma3, float eps) {
    const int i = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    float w12 = 0.f, n0 = 0.f, n1 = 0.f;
    for (int c = lane; c < nef; c += 64) {
        const float a = cnn[j * nef + c], b = rnn[i * nef + c];
        w12 += a * b; n0 += a * a; n1 += b * b;
    }
    w12 = wave_sum(w12); n0 = wave_sum(n0); n1 = wave_sum(n1);
    const float den = sqrtf(n0) * sqrtf(n1);
    const bool clamped = den < eps;
    const float g = ds[j * B + i] * gamma3;
    const float k0 = g / fmaxf(den, eps);
    const float ka = clamped ? 0.f : g * (w12 / den) / n0;
    const float kb = clamped ? 0.f : g * (w12 / den) / n1;
    for (int c = lane; c < nef; c += 64) {
        const float a = cnn[j * nef + c], b = rnn[i * nef + c];
        if (dcnn) atomicAdd(&dcnn[j * nef + c], k0 * b - ka * a);
        if (drnn) atomicAdd(&drnn[i * nef + c], k0 * a - kb * b);
    }
}

// deterministic mode: blockIdx.y = 0: dcnn[j = blockIdx.x] += sum_i ..., blockIdx.y = 1: drnn[i = blockIdx.x] += sum_j ...
// (one workgroup per output row, the partners walked in order, no atomics)
__global__ __launch_bounds__(64) void damsm_sent_bwd_det_kernel(const float* __restrict__ cnn, const float* __restrict__ rnn,
                                                                const float* __restrict__ ds, float* __restrict__ dcnn,
                                                                float* __restrict__ drnn, int B, int nef, float gamma3,
                                                                float eps) {
    const int lane = threadIdx.x, side = blockIdx.y, o = blockIdx.x;
    float* out = side == 0 ? dcnn : drnn;
    if (!out) return;
    for (int p = 0; p < B; ++p) {
        const int j = side == 0 ? o : p, i = side == 0 ? p : o;
        float w12 = 0.f, n0 = 0.f, n1 = 0.f;
        for (int c = lane; c < nef; c += 64) {
            const float a = cnn[j * nef + c], b = rnn[i * nef + c];
            w12 += a * b; n0 += a * a; n1 += b * b;
        }
        w12 = wave_sum(w12); n0 = wave_sum(n0); n1 = wave_sum(n1);
        const float den = sqrtf(n0) * sqrtf(n1);
        const bool clamped = den < eps;
        const float g = ds[j * B + i] * gamma3;
        const float k0 = g / fmaxf(den, eps);
        const float ka = clamped ? 0.f : g * (w12 / den) / n0;
        const float kb = clamped ? 0.f : g * (w12 / den) / n1;
        for (int c = lane; c < nef; c += 64) {
            const float a = cnn[j * nef + c], b = rnn[i * nef + c];
            if (side == 0) out[j * nef + c] += k0 * b - ka * a;
            else out[i * nef + c] += k0 * a - kb * b;
        }
    }
}

// two cross entropies (rows / columns) over a B x B score matrix with labels arange(B)
__global__ void ce_pair_kernel(const float* __restrict__ score, const uint8_t* __restrict__ mask, float scale,
                               float* __restrict__ loss, float* __restrict__ d0, float* __restrict__ d1, int B) {
    extern __shared__ float s[];      // [B*B] scaled masked scores, then [2*B] row/col lse
    float* lse = s + B * B;
    const int tid = threadIdx.x;
    for (int k = tid; k < B * B; k += blockDim.x)
        s[k] = (mask && mask[k]) ? -INFINITY : score[k] * scale;
    __syncthreads();
    for (int k = tid; k < 2 * B; k += blockDim.x) {
        const bool col = k >= B;
        const int a = col ? k - B : k;
        float mx = -INFINITY;
        for (int b = 0; b < B; ++b) mx = fmaxf(mx, col ? s[b * B + a] : s[a * B + b]);
        float sum = 0.f;
        for (int b = 0; b < B; ++b) sum += expf((col ? s[b * B + a] : s[a * B + b]) - mx);
        lse[k] = mx + logf(sum);
    }
    __syncthreads();
    if (tid == 0) {
        float l0 = 0.f, l1 = 0.f;
        for (int a = 0; a < B; ++a) { l0 += lse[a] - s[a * B + a]; l1 += lse[B + a] - s[a * B + a]; }
        loss[0] = l0 / B;
        loss[1] = l1 / B;
    }
    const float invB = 1.f / B;
    for (int k = tid; k < B * B; k += blockDim.x) {
        const int j = k / B, i = k - j * B;
        const float dl = j == i ? 1.f : 0.f;
        // d loss / d (unscaled score): softmax - onehot, times scale / B
        d0[k] = (expf(s[k] - lse[j]) - dl) * invB * scale;
        d1[k] = (expf(s[k] - lse[B + i]) - dl) * invB * scale;
    }
}

__global__ void combine2_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ ga,
                                const float* __restrict__ b, const float* __restrict__ gb, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ga[0] * a[i] + gb[0] * b[i];
}

}  // namespace

extern "C" int sba_damsm_words_fwd(const float* feat, const float* words, const int64_t* cap_lens, float* sim,
                                   float* attn, float* attn1, float* wctx, int B, int nef, int R, int L,
                                   float gamma1, float gamma2, void* stream) {
    if (!feat || !words || !cap_lens || !sim || !attn || !attn1 || !wctx) return SBA_E_ARG;
    if (B <= 0 || B > 1024 || nef <= 0 || nef % 32 || R <= 0 || R > NT || L <= 0 || L > TMAX) return SBA_E_ARG;
    const int LP = L <= 20 ? 20 : 32;
    const size_t sh = lds_bytes(nef, LP, L, R);
    if (sh > 160 * 1024) return SBA_E_ARG;
    if (LP == 20) {
        (void)hipFuncSetAttribute((const void*)damsm_words_fwd_kernel<20>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        SBA_LAUNCH(damsm_words_fwd_kernel<20>, dim3(B, B), dim3(NT), sh, (hipStream_t)stream, feat, words,
                           cap_lens, sim, attn, attn1, wctx, B, nef, R, L, gamma1, gamma2);
    } else {
        (void)hipFuncSetAttribute((const void*)damsm_words_fwd_kernel<32>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        SBA_LAUNCH(damsm_words_fwd_kernel<32>, dim3(B, B), dim3(NT), sh, (hipStream_t)stream, feat, words,
                           cap_lens, sim, attn, attn1, wctx, B, nef, R, L, gamma1, gamma2);
    }
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_damsm_words_bwd(const float* feat, const float* words, const int64_t* cap_lens, const float* sim,
                                   const float* attn, const float* attn1, const float* wctx, const float* dsim,
                                   float* dfeat, float* dwords, int B, int nef, int R, int L, float gamma1,
                                   float gamma2, void* stream) {
    if (!feat || !words || !cap_lens || !sim || !attn || !attn1 || !wctx || !dsim || !dfeat) return SBA_E_ARG;
    if (B <= 0 || B > 1024 || nef <= 0 || nef % 32 || R <= 0 || R > NT || L <= 0 || L > TMAX) return SBA_E_ARG;
    const int LP = L <= 20 ? 20 : 32;
    const size_t sh = lds_bytes(nef, LP, L, R);
    if (sh > 160 * 1024) return SBA_E_ARG;
    float *pf = nullptr, *pw = nullptr;
    if (sba_det_on()) {
        pf = sba_det_alloc((int64_t)B * B * nef * R);
        if (!pf) return SBA_E_ARG;
        if (dwords) { pw = sba_det_alloc((int64_t)B * B * nef * L); if (!pw) return SBA_E_ARG; }
    }
    if (LP == 20) {
        (void)hipFuncSetAttribute((const void*)damsm_words_bwd_kernel<20>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        SBA_LAUNCH(damsm_words_bwd_kernel<20>, dim3(B, B), dim3(NT), sh, (hipStream_t)stream, feat, words,
                           cap_lens, sim, attn, attn1, wctx, dsim, dfeat, dwords, B, nef, R, L, gamma1, gamma2, pf, pw);
    } else {
        (void)hipFuncSetAttribute((const void*)damsm_words_bwd_kernel<32>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        SBA_LAUNCH(damsm_words_bwd_kernel<32>, dim3(B, B), dim3(NT), sh, (hipStream_t)stream, feat, words,
                           cap_lens, sim, attn, attn1, wctx, dsim, dfeat, dwords, B, nef, R, L, gamma1, gamma2, pf, pw);
    }
    if (pf) sba_det_fold(pf, B, B, (int64_t)nef * R, dfeat, (int64_t)nef * R, 0, (hipStream_t)stream);
    if (pw) sba_det_fold(pw, B, B, (int64_t)nef * L, dwords, (int64_t)nef * L, 0, (hipStream_t)stream);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_damsm_sent_fwd(const float* cnn, const float* rnn, float* s, int B, int nef, float gamma3,
                                  float eps, void* stream) {
    if (!cnn || !rnn || !s || B <= 0 || B > 1024 || nef <= 0) return SBA_E_ARG;
    SBA_LAUNCH(damsm_sent_fwd_kernel, dim3(B, B), dim3(64), 0, (hipStream_t)stream, cnn, rnn, s, B, nef,
                       gamma3, eps);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_damsm_sent_bwd(const float* cnn, const float* rnn, const float* ds, float* dcnn, float* drnn,
                                  int B, int nef, float gamma3, float eps, void* stream) {
    if (!cnn || !rnn || !ds || B <= 0 || B > 1024 || nef <= 0) return SBA_E_ARG;
    if (sba_det_on()) {
        SBA_LAUNCH(damsm_sent_bwd_det_kernel, dim3(B, 2), dim3(64), 0, (hipStream_t)stream, cnn, rnn, ds, dcnn, drnn, B,
                   nef, gamma3, eps);
        return SBA_CHECK_LAUNCH();
    }
    SBA_LAUNCH(damsm_sent_bwd_kernel, dim3(B, B), dim3(64), 0, (hipStream_t)stream, cnn, rnn, ds, dcnn, drnn,
                       B, nef, gamma3, eps);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_ce_pair(const float* score, const uint8_t* mask, float scale, float* loss, float* dscore0,
                           float* dscore1, int B, void* stream) {
    if (!score || !loss || !dscore0 || !dscore1 || B <= 0 || B > 96) return SBA_E_ARG;
    const size_t sh = sizeof(float) * ((size_t)B * B + 2 * B);
    SBA_LAUNCH(ce_pair_kernel, dim3(1), dim3(256), sh, (hipStream_t)stream, score, mask, scale, loss,
                       dscore0, dscore1, B);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_combine2(float* out, const float* a, const float* ga, const float* b, const float* gb, int n,
                            void* stream) {
    if (!out || !a || !ga || !b || !gb || n <= 0) return SBA_E_ARG;
    SBA_LAUNCH(combine2_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, a, ga, b, gb, n);
    return SBA_CHECK_LAUNCH();
}
