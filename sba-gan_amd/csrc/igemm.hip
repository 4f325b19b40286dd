// Implicit-GEMM convolution kernels for gfx950 (MI355X): forward / data-gradient
// (one kernel, driven by a tap table) and weight-gradient.  NHWC activations,
// [Cout][tap][Cin] packed weights, MFMA 32x32 tiles with f32 accumulation
// (v_mfma_f32_32x32x16_bf16 for bf16 storage, v_mfma_f32_32x32x2_f32 for f32).
//
// Replaces cuDNN under nn.Conv2d forward/backward in the reference
// (model.py:32-35 conv3x3, :552 downBlock conv4x4 s2, :41 fused nearest x2).
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include <type_traits>

namespace {

// Division of a pixel index (< 2^21) by a launch-constant: one 64-bit multiply instead of the
// ~30-instruction integer division sequence (the weight-gradient kernels decode (n, oy, ox) for
// every staged pixel, which made address generation their bottleneck).
struct FastDiv {
    uint64_t magic;     // ceil(2^42 / d), 0 = use the plain division
    uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d, int64_t max_n) {
    FastDiv f;
    f.d = d;
    f.magic = (max_n < (1 << 21) && d > 0) ? (((uint64_t)1 << 42) + d - 1) / d : 0;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    return f.magic ? (uint32_t)(((uint64_t)n * f.magic) >> 42) : n / f.d;
}

// ---------------------------------------------------------------------------
// MFMA over one 64-byte K slab held in LDS as rows of ROWB bytes
// ---------------------------------------------------------------------------
// optional epilogue operands: per-channel bias, and a ReLU mask source (same layout as y): outputs are
// zeroed where mask <= 0 -- the backward of the ReLU that produced the tensor whose gradient this is
// yh: store the (bf16-typed) output as IEEE binary16 bits (SBA_BF16_YH: the pre-BatchNorm tensor)
struct EpiX { const float* bias; const void* mask; int yh; };

// keep the bf16 halves of v whose counterpart in m is > 0
__device__ __forceinline__ uint32_t relu_mask_bf16x2(uint32_t v, uint32_t m) {
    const uint32_t lo = ((m & 0x7fffu) != 0u && (m & 0x8000u) == 0u) ? 0x0000ffffu : 0u;
    const uint32_t hi = ((m & 0x7fff0000u) != 0u && (m & 0x80000000u) == 0u) ? 0xffff0000u : 0u;
    return v & (lo | hi);
}

// two packed bf16 values + two packed bf16 values, rounded back to bf16
__device__ __forceinline__ uint32_t add_bf16x2(uint32_t p, uint32_t q) {
    const float lo = __uint_as_float(p << 16) + __uint_as_float(q << 16);
    const float hi = __uint_as_float(p & 0xffff0000u) + __uint_as_float(q & 0xffff0000u);
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

template <typename T> struct Mma;

template <> struct Mma<bf16_t> {
    // slab = 32 bf16 = two 16-deep MFMA steps; lane l holds row (l&31), k = 8*(l>>5)+j
    template <int TM, int TN, int ROWB>
    static __device__ __forceinline__ void slab(const unsigned char* a_rows, const unsigned char* b_rows,
                                                int lane, f32x16_t (&acc)[TM][TN]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8_t*>(a_rows + (i * 32 + r) * ROWB + kk * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8_t*>(b_rows + (j * 32 + r) * ROWB + kk * 32 + h * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <> struct Mma<float> {
    // slab = 16 floats = eight 2-deep MFMA steps; lane l holds row (l&31), k = (l>>5)
    template <int TM, int TN, int ROWB>
    static __device__ __forceinline__ void slab(const unsigned char* a_rows, const unsigned char* b_rows,
                                                int lane, f32x16_t (&acc)[TM][TN]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const float*>(a_rows + (i * 32 + r) * ROWB + (2 * kk + h) * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const float*>(b_rows + (j * 32 + r) * ROWB + (2 * kk + h) * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

// ---------------------------------------------------------------------------
// shared tile epilogue: per-channel BatchNorm statistics, bias / ReLU, residual addend, NHWC store
// (bf16 tiles are transposed through LDS into 16-byte row stores).  `lead` marks the threads that
// own accumulators; rowoff[row] = output pixel index of tile row `row` (or -1).
// ---------------------------------------------------------------------------
template <typename T, int BM, int BN, int TM, int TN, int NTT, int STAGE_BYTES>
__device__ __forceinline__ void tile_epilogue(f32x16_t (&acc)[TM][TN], const bool lead, unsigned char* lds_all,
                                              const int* rowoff, float* s_stat, const int wm0, const int wn0,
                                              const int lane, const int n_base, const int ycs,
                                              const sba_conv_geom& g, T* __restrict__ y,
                                              const T* __restrict__ addend, float* __restrict__ stats,
                                              const EpiX ex, const int slot_id = -1) {
    const float* __restrict__ bias = ex.bias;
    const T* __restrict__ rmask = reinterpret_cast<const T*>(ex.mask);
    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col_l = lane & 31, rsel = 4 * (lane >> 5);
    constexpr bool kStageOut = sizeof(T) == 2;       // bf16: transpose through LDS -> 16-byte row stores
    constexpr int OROW = BN * 2 + 16;                // staged output row: BN bf16 + 16 B pad
    static_assert(!kStageOut || BM * OROW <= STAGE_BYTES, "output tile fits in the staging buffers");
    if (lead) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n_base + wn0 + j * 32 + col_l;
        float csum = 0.f, csq = 0.f;
        const float bco = bias ? bias[co < g.Cout ? co : 0] : 0.f;      // one load per column, not per element
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                float v = acc[i][j][r];
                csum += v;
                csq += v * v;
                if (bias) v += bco;
                if (g.relu) v = fmaxf(v, 0.f);
                if (kStageOut) {
                    // (the main loop's last barrier has passed: the staging buffers are free)
                    *reinterpret_cast<bf16_t*>(lds_all + row * OROW + (wn0 + j * 32 + col_l) * 2) =
                        ex.yh ? f2h_bits(v) : f2bf(v);
                } else {
                    const int pix = rowoff[row];
                    if (pix >= 0 && co < g.Cout) {
                        const int64_t o = (int64_t)pix * ycs + g.y_coff + co;
                        if (addend) v += to_f<T>(addend[o]);
                        if (rmask && !(to_f<T>(rmask[o]) > 0.f)) v = 0.f;
                        y[o] = from_f<T>(v);
                    }
                }
            }
        }
        if (stats) {
            csum += __shfl_xor(csum, 32, 64);
            csq += __shfl_xor(csq, 32, 64);
            if (lane < 32) {
                atomicAdd(&s_stat[wn0 + j * 32 + col_l], csum);
                atomicAdd(&s_stat[BN + wn0 + j * 32 + col_l], csq);
            }
        }
    }
    }
    if (kStageOut || stats) __syncthreads();
    if (kStageOut) {
        constexpr int CPRO = BN / 8;                 // 16-byte chunks per output row
        for (int idx = threadIdx.x; idx < BM * CPRO; idx += NTT) {
            const int row = idx / CPRO, cc = idx - row * CPRO;
            const int pix = rowoff[row];
            const int co = n_base + cc * 8;
            if (pix < 0 || co >= g.Cout) continue;
            uint4 v = *reinterpret_cast<const uint4*>(lds_all + row * OROW + cc * 16);
            const int64_t o = (int64_t)pix * ycs + g.y_coff + co;
            if (co + 8 <= g.Cout) {
                if (addend) {
                    const uint4 a = *reinterpret_cast<const uint4*>(addend + o);
                    v.x = add_bf16x2(v.x, a.x);
                    v.y = add_bf16x2(v.y, a.y);
                    v.z = add_bf16x2(v.z, a.z);
                    v.w = add_bf16x2(v.w, a.w);
                }
                if (rmask) {
                    const uint4 m = *reinterpret_cast<const uint4*>(rmask + o);
                    v.x = relu_mask_bf16x2(v.x, m.x);
                    v.y = relu_mask_bf16x2(v.y, m.y);
                    v.z = relu_mask_bf16x2(v.z, m.z);
                    v.w = relu_mask_bf16x2(v.w, m.w);
                }
                *reinterpret_cast<uint4*>(y + o) = v;
            } else {                                  // ragged Cout tail: scalar
                // (fully unrolled with static indices: a dynamically indexed private array would be
                // promoted to LDS and make every wave read the AQL dispatch packet for its flat id)
                const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (co + k < g.Cout) {
                        float f = bf2f((bf16_t)((k & 1) ? (vw[k >> 1] >> 16) : (vw[k >> 1] & 0xffffu)));
                        if (addend) f += to_f<T>(addend[o + k]);
                        if (rmask && !(to_f<T>(rmask[o + k]) > 0.f)) f = 0.f;
                        y[o + k] = from_f<T>(f);
                    }
                }
            }
        }
    }
    if (stats) {
        // one of SBA_BN_STAT_SLOTS replicas per workgroup: 1/SLOTS of the same-address atomic traffic
        const int sid = slot_id >= 0 ? slot_id : (int)(blockIdx.x + blockIdx.z);
        float* slot = stats + (int64_t)(sid & (SBA_BN_STAT_SLOTS - 1)) * 2 * g.Cout;
        for (int c = threadIdx.x; c < BN; c += NTT) {
            const int co = n_base + c;
            if (co < g.Cout) {
                atomicAdd(&slot[co], s_stat[c]);
                atomicAdd(&slot[g.Cout + co], s_stat[BN + c]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// forward / data-gradient implicit GEMM
//   rows  = output pixels of the (OHs x OWs) sub-grid, M = N*OHs*OWs
//   cols  = output channels
//   K     = ntaps * Cin, walked in 64-byte slabs (one tap, 32 bf16 / 16 f32 channels)
// 256 threads = 4 waves laid out (BM/WM) x (BN/WN); each wave owns WM x WN.
// LDS: double-buffered A[BM] and B[BN] rows of 80 B (64 data + 16 pad: the
// pad makes the 16-lane ds_read_b128 groups hit 16 distinct 4-bank slots).
// ---------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int KS, int KG, int PF>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64 * KG, (sizeof(T) == 2 && BM * BN == 128 * 128 && KS == 1) ? 3 : 1) void igemm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                    T* __restrict__ y, const T* __restrict__ addend,
                                                    float* __restrict__ stats, const sba_conv_geom g,
                                                    const int M, float* __restrict__ ws,
                                                    const int slabs_per_split, const EpiX ex) {
    constexpr int ROWB = 80;
    constexpr int KS_CH = 64 / (int)sizeof(T);   // channels per slab
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;   // threads of one K-group: one wave per WM x WN sub-tile
    constexpr int NTT = NT * KG;                     // KG groups walk disjoint K ranges of the same tile
    constexpr int RP = NT / 4;                       // tile rows staged per pass (4 threads x 16 B per row)
    constexpr int AI = (BM + RP - 1) / RP, BI = (BN + RP - 1) / RP;    // 16-byte loads per thread per slab
    static_assert(BM % WM == 0 && BN % WN == 0 && WM % 32 == 0 && WN % 32 == 0, "tile");
    constexpr int TILE_BYTES = (BM + BN) * ROWB;
    // PF = stages of global loads kept in flight (register sets); PF == 1: double-buffered LDS, one
    // stage ahead.  PF == 3 (small tiles, ~1 workgroup per CU, nothing else to hide the load latency
    // behind): three register sets + three LDS buffers, loads issued three stages ahead.
    static_assert(PF == 1 || PF == 3, "prefetch depth");
    constexpr int NBUF = PF == 1 ? 2 : 3;
    constexpr int GROUP_BYTES = NBUF * KS * TILE_BYTES;
    static_assert(KG == 1 || (KG - 1) * BM * BN * 4 <= KG * GROUP_BYTES, "K-group partials fit in the staging buffers");

    // KS slabs are staged per barrier (KS > 1 for the small tiles, whose MFMA work per slab is short);
    // each K-group has its own double buffer
    // The epilogue's row table and statistics accumulators live INSIDE the (by then free) staging buffers,
    // behind the staged bf16 output tile: the 64x64 tile then needs 36 KB instead of 40.75 KB of LDS and a CU
    // holds four workgroups instead of three.
    constexpr int EPI_OFF = sizeof(T) == 2 ? BM * (BN * 2 + 16) : 0;
    constexpr int EPI_END = EPI_OFF + BM * 4 + BN * 8;
    constexpr int LDS_BYTES = KG * GROUP_BYTES > EPI_END ? KG * GROUP_BYTES : EPI_END;
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[LDS_BYTES];
    int* rowoff = reinterpret_cast<int*>(lds_all + EPI_OFF);
    float* s_stat = reinterpret_cast<float*>(lds_all + EPI_OFF + BM * 4);

    const int kg = KG > 1 ? (int)threadIdx.x / NT : 0;
    const int tid = KG > 1 ? (int)threadIdx.x - kg * NT : (int)threadIdx.x;
    unsigned char* const lds = lds_all + kg * GROUP_BYTES;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int m_base = blockIdx.x * BM, n_base = blockIdx.y * BN;
    const int IHL = g.ups ? 2 * g.IH : g.IH, IWL = g.ups ? 2 * g.IW : g.IW;
    const int sub = g.OHs * g.OWs;

    // per-thread description of the A rows it stages (fixed over the K loop)
    const int chunk = tid & 3;
    int a_iy0[AI], a_ix0[AI], a_nb[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m_base + (tid >> 2) + RP * i;
        if (m < M && (tid >> 2) + RP * i < BM) {
            const int n = m / sub, rem = m - n * sub;
            const int oy = rem / g.OWs, ox = rem - oy * g.OWs;
            a_iy0[i] = oy * g.sy;
            a_ix0[i] = ox * g.sx;
            a_nb[i] = n * g.IH * g.IW;
        } else {
            a_iy0[i] = -100000;   // always out of bounds -> zero rows
            a_ix0[i] = 0;
            a_nb[i] = 0;
        }
    }
    const int cpt = g.Cin / KS_CH;            // slabs per tap
    const int nsteps = g.ntaps * cpt;
    // tap offsets packed 4 bits each (offset + 8) so that the per-step lookup is
    // scalar shifts instead of a dynamically indexed kernarg array
    uint64_t tyb[2] = {0, 0}, txb[2] = {0, 0};
#pragma unroll
    for (int t = 0; t < SBA_MAX_TAPS; ++t) {
        tyb[t >> 4] |= (uint64_t)((g.ty[t] + 8) & 15) << (4 * (t & 15));
        txb[t >> 4] |= (uint64_t)((g.tx[t] + 8) & 15) << (4 * (t & 15));
    }
    const int xcs = g.x_cstride ? g.x_cstride : g.Cin;       // input pixel stride (channels)
    const int ycs = g.y_cstride ? g.y_cstride : g.Cout;      // output pixel stride (channels)

    // split-K: this block walks slabs [b_begin, b_end), its K-group kg the sub-range [s_begin, s_end)
    const int b_begin = blockIdx.z * slabs_per_split;
    const int b_end = min(b_begin + slabs_per_split, nsteps);
    const int per_group = KG > 1 ? ((b_end - b_begin + KG * KS - 1) / (KG * KS)) * KS : b_end - b_begin;
    const int s_begin = b_begin + kg * per_group;
    const int s_end = min(s_begin + per_group, b_end);

    // Address generation is hoisted out of the per-slab path: slabs are consumed in order, so the
    // (tap, channel-slab) position is tracked incrementally (no division), the per-row gather
    // offsets are recomputed only when the tap changes, and everything is 32-bit byte offsets
    // (tensors are < 4 GiB, checked on the host).
    // Loads are BRANCH-FREE buffer loads: an out-of-image tap, a row beyond M or a slab beyond this
    // split's range gets the offset 0xFFFFFFFF, which the buffer bounds check turns into zeros.
    int g_tap = s_begin / cpt, g_c = s_begin - g_tap * cpt, g_step = s_begin;
    int cur_tap = -1;
    constexpr uint32_t OOB = 0xFFFFFFFFu;
    uint32_t a_off[AI];           // byte offset of (n, iy, ix, channel 0) for the current tap, or OOB
#pragma unroll
    for (int i = 0; i < AI; ++i) a_off[i] = OOB;
    uint32_t w_off[BI];           // byte offset of (co, slab, chunk); K is contiguous per co
    const uint32_t krow_bytes = (uint32_t)g.ntaps * (uint32_t)g.Cin * (uint32_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int r = (tid >> 2) + RP * i;
        const int co = n_base + r;
        const bool ok = (co < g.Cout) && (r < BN);
        w_off[i] = ok ? (uint32_t)co * krow_bytes + (uint32_t)s_begin * 64u + (uint32_t)chunk * 16u : OOB;
    }
    const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.IH * g.IW * xcs * (int64_t)sizeof(T));
    const uint32_t w_bytes = (uint32_t)g.Cout * krow_bytes;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

    uint4 rra[PF][KS][AI], rrb[PF][KS][BI];
    auto gload = [&](uint4 (&ra)[KS][AI], uint4 (&rb)[KS][BI]) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const bool live = g_step < s_end;
            if (g_tap != cur_tap) {                     // (uniform) new tap: refresh the gather offsets
                cur_tap = g_tap;
                const int tsel = g_tap < SBA_MAX_TAPS ? g_tap : 0;
                const uint64_t tyw = tsel < 16 ? tyb[0] : tyb[1], txw = tsel < 16 ? txb[0] : txb[1];
                const int ty = (int)((tyw >> (4 * (tsel & 15))) & 15) - 8;
                const int tx = (int)((txw >> (4 * (tsel & 15))) & 15) - 8;
#pragma unroll
                for (int i = 0; i < AI; ++i) {
                    int iy = a_iy0[i] + ty, ix = a_ix0[i] + tx;
                    const bool ok = (iy >= 0) & (iy < IHL) & (ix >= 0) & (ix < IWL);
                    if (g.ups) { iy >>= 1; ix >>= 1; }
                    const uint32_t o = (uint32_t)(a_nb[i] + iy * g.IW + ix) * (uint32_t)(xcs * (int)sizeof(T)) +
                                       (uint32_t)(g.x_coff * (int)sizeof(T)) + (uint32_t)chunk * 16u;
                    a_off[i] = ok ? o : OOB;
                }
            }
            const uint32_t cbytes = (uint32_t)g_c * 64u;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const uint32_t o = (live && a_off[i] != OOB) ? a_off[i] + cbytes : OOB;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0, 0);
                ra[k][i] = make_uint4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const uint32_t o = (live && w_off[i] != OOB) ? w_off[i] : OOB;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wr, o, 0, 0);
                rb[k][i] = make_uint4(v[0], v[1], v[2], v[3]);
                if (w_off[i] != OOB) w_off[i] += 64u;
            }
            ++g_step;
            if (++g_c == cpt) { g_c = 0; ++g_tap; }
        }
    };
    auto lstore = [&](int buf, const uint4 (&ra)[KS][AI], const uint4 (&rb)[KS][BI]) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            unsigned char* base = lds + (buf * KS + k) * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < AI; ++i)
                if ((tid >> 2) + RP * i < BM)
                    *reinterpret_cast<uint4*>(base + ((tid >> 2) + RP * i) * ROWB + chunk * 16) = ra[k][i];
#pragma unroll
            for (int i = 0; i < BI; ++i)
                if ((tid >> 2) + RP * i < BN)
                    *reinterpret_cast<uint4*>(base + (BM + (tid >> 2) + RP * i) * ROWB + chunk * 16) = rb[k][i];
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nstages = (per_group + KS - 1) / KS;      // uniform over the groups (dead slabs load zeros)
    if (PF == 1) {
        gload(rra[0], rrb[0]);
        lstore(0, rra[0], rrb[0]);
        __syncthreads();
        for (int s = 0; s < nstages; ++s) {
            const int buf = s & 1;
            if (s + 1 < nstages) gload(rra[0], rrb[0]);
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const unsigned char* base = lds + (buf * KS + k) * TILE_BYTES;
                Mma<T>::template slab<TM, TN, ROWB>(base + wm0 * ROWB, base + (BM + wn0) * ROWB, lane, acc);
            }
            if (s + 1 < nstages) lstore(buf ^ 1, rra[0], rrb[0]);
            __syncthreads();
        }
    } else {
        // stage s: LDS buffer s % 3 holds it; register set (s+1) % 3 and (s+2) % 3 hold the loads of
        // stages s+1, s+2 (in flight); set s % 3 is free -> issue stage s+3 into it.  Stages past the
        // end load zeros (OOB offsets), so the trip count is simply rounded up to a multiple of 3.
        constexpr int P1 = PF > 1 ? 1 : 0, P2 = PF > 2 ? 2 : 0;
        gload(rra[0], rrb[0]);
        gload(rra[P1], rrb[P1]);
        gload(rra[P2], rrb[P2]);
        lstore(0, rra[0], rrb[0]);
        __syncthreads();
        auto stage = [&](int buf, uint4 (&fa)[KS][AI], uint4 (&fb)[KS][BI], const uint4 (&na)[KS][AI],
                         const uint4 (&nb)[KS][BI]) {
            gload(fa, fb);
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const unsigned char* base = lds + (buf * KS + k) * TILE_BYTES;
                Mma<T>::template slab<TM, TN, ROWB>(base + wm0 * ROWB, base + (BM + wn0) * ROWB, lane, acc);
            }
            lstore(buf == 2 ? 0 : buf + 1, na, nb);
            __syncthreads();
        };
        for (int s = 0; s < nstages; s += 3) {
            stage(0, rra[0], rrb[0], rra[P1], rrb[P1]);
            stage(1, rra[P1], rrb[P1], rra[P2], rrb[P2]);
            stage(2, rra[P2], rrb[P2], rra[0], rrb[0]);
        }
    }

    if (KG > 1) {
        // sum the K-groups' partial tiles into group 0 through the (now free) staging buffers
        float* red = reinterpret_cast<float*>(lds_all);
        if (kg > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[(((kg - 1) * TM * TN + i * TN + j) * 16 + r) * NT + tid] = acc[i][j][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int q = 0; q < KG - 1; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[i][j][r] += red[((q * TM * TN + i * TN + j) * 16 + r) * NT + tid];
        }
        __syncthreads();
    }
    const bool lead = KG == 1 || kg == 0;       // the group that owns the summed tile

    if (ws) {
        if (!lead) return;
        // split-K partial: f32 atomics into ws[m][co]; y / addend / stats are done by splitk_finish_kernel
        const int col_s = lane & 31, rsel_s = 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n_base + wn0 + j * 32 + col_s;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_base + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + rsel_s;
                    if (m < M && co < g.Cout) atomicAdd(&ws[(int64_t)m * g.Cout + co], acc[i][j][r]);
                }
        }
        return;
    }

    // (the main loop / K-group reduction ended with a barrier: the staging buffers are free)
    for (int r = threadIdx.x; r < BM; r += NTT) {
        const int m = m_base + r;
        int off = -1;
        if (m < M) {
            const int n = m / sub, rem = m - n * sub;
            const int oy = rem / g.OWs, ox = rem - oy * g.OWs;
            off = (n * g.OH + oy * g.osy + g.ooy) * g.OW + ox * g.osx + g.oox;
        }
        rowoff[r] = off;
    }
    for (int c = threadIdx.x; c < 2 * BN; c += NTT) s_stat[c] = 0.f;
    __syncthreads();
    tile_epilogue<T, BM, BN, TM, TN, NTT, LDS_BYTES>(acc, lead, lds_all, rowoff, s_stat, wm0, wn0, lane, n_base, ycs, g,
                                                     y, addend, stats, ex);
}

// ---------------------------------------------------------------------------
// The same implicit GEMM with the operands staged by LDS-DMA (bf16): `buffer_load_dwordx4 ... lds` moves
// 16 bytes per lane from a per-lane global offset straight into LDS (wave-uniform base + lane * 16), so a
// K stage costs no staging VGPRs and no ds_write pass, and a ring of D stages keeps D-1 of them in flight:
// the generic kernel above exposes one full global-load latency per stage on the launch-latency-bound
// layers (273-workgroup grids of the Inception trunk: ~1440 cycles per 64-deep stage for 128 cycles of MFMA).
//   * LDS image of one 32-channel slab: rows of 64 B, unpadded (the DMA destination is lane-linear: 4 lanes
//     per row, 16 rows per wave-instruction); the 16-byte chunk c of row r lives at chunk c ^ ((r >> 2) & 3),
//     applied on the SOURCE side (each lane loads the chunk its slot holds) and on the fragment reads: every
//     16-lane phase of a ds_read_b128 then covers all 64 banks.
//   * out-of-image taps, rows beyond M and slabs beyond the split's range are out-of-range buffer offsets:
//     the DMA writes zeros (tools/ldsdma_probe.hip).
//   * the DMA is issued from inline asm (hipcc would otherwise drain ALL of it -- vmcnt(0) -- in front of
//     every LDS read); completion is counted by hand: before stage s is read every wave waits until only the
//     (D-2) younger stages it issued are outstanding, then the workgroup barrier makes all waves' parts visible
//     and proves that nobody still reads the buffer the next issue overwrites.
//   * workgroups are numbered so that the N tiles of one M tile run on one XCD, back to back (shared A rows
//     are served by that XCD's L2).
// ---------------------------------------------------------------------------
// M0 = LDS destination of the DMA.  M0 is compiler-reserved; nothing else in these kernels uses it (LDS
// instructions need no M0 on gfx9+, no dynamic register indexing, no LDS-DMA builtins), so it is simply
// overwritten: saving and restoring it around every load cost two of the ~10 scalar instructions per load, and
// with one wave per SIMD the main loop is instruction-ISSUE bound (rocprofv3: SQ_ACTIVE_INST_ANY 45 % of the wave
// cycles against 10 % SQ_VALU_MFMA_BUSY_CYCLES on the 17x17 layers, profiles/r02_pmc_igemm_dma_v1.txt).
// voff: per-lane byte offset (bounds-checked: 0xFFFFFFFF -> zeros); soff: wave-uniform byte offset added after
// the bounds check -- the walk along K costs no per-lane arithmetic.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16(const __amdgpu_buffer_rsrc_t rsrc, const uint32_t voff, const uint32_t soff,
                                          const uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

// Split-K without a finishing launch: every split adds its partial tile into the zero-filled f32 workspace with
// device-scope atomics, then takes a ticket for its output tile; the LAST arrival reads the complete sums back
// (device-scope loads: the partial sums were performed at the memory side, not in this XCD's L2), leaves the
// workspace and the ticket zero for the next user, and runs the ordinary epilogue (bias / ReLU / addend / mask / BN
// statistics / y) as if it had computed the whole K range itself.  Nobody waits for anybody: a workgroup that is
// not last simply exits.  Returns true for the workgroup that has to run the epilogue.
constexpr int SPLITK_TICKET_BYTES = 64 * 1024;      // tail of the workspace: one int32 ticket per output tile
template <int TM, int TN>
__device__ __forceinline__ bool splitk_arrive(float* __restrict__ ws, int* __restrict__ tickets, f32x16_t (&acc)[TM][TN],
                                              const int m0, const int n0, const int lane, const int M, const int Cout,
                                              const int tile_id, int* s_ticket) {
    const int col_s = lane & 31, rsel_s = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + j * 32 + col_s;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + rsel_s;
                if (m < M && co < Cout) atomicAdd(&ws[(int64_t)m * Cout + co], acc[i][j][r]);
            }
    }
    if (!tickets) return false;                 // the caller launches splitk_finish_kernel
    // Order "my partial sums, then my ticket" WITHOUT a release fence: at agent scope a fence writes back and
    // invalidates the whole L2 of this XCD (measured: the step went from 14.4 to 18.6 ms).  The partial sums are
    // device-scope atomics, performed at the memory side and acknowledged once performed; vmcnt counts them, so after
    // s_waitcnt vmcnt(0) they are visible to every later device-scope access -- the ticket, and the last arrival's
    // device-scope loads, which bypass the L2 themselves.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *s_ticket = atomicAdd(&tickets[tile_id], 1);
    __syncthreads();
    const int ticket = *s_ticket;
    __syncthreads();                            // s_ticket is a word of the staging ring: the epilogue reuses it
    if (ticket != (int)gridDim.z - 1) return false;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + j * 32 + col_s;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + rsel_s;
                float v = 0.f;
                if (m < M && co < Cout) {
                    float* p = &ws[(int64_t)m * Cout + co];
                    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                acc[i][j][r] = v;
            }
    }
    if (threadIdx.x == 0) __hip_atomic_store(&tickets[tile_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

#ifdef SBA_DMA_TRACE     // tools/trace_dma.py: per-stage s_memtime stamps of wave 0 of the first workgroups
static unsigned long long* g_dma_trace = nullptr;
extern "C" void sba_set_dma_trace(unsigned long long* p) { g_dma_trace = p; }
#define DMA_TRACE_PARAM , unsigned long long* __restrict__ trace
#define DMA_TRACE_ARG , g_dma_trace
#define DMA_TRACE_ARG_FWD , trace
#define DMA_STAMP(slot) do { if (trace && L < 32 && tid == 0 && s < 30) { asm volatile("" ::: "memory"); \
    trace[(L * 32 + s) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } } while (0)
#else
#define DMA_TRACE_PARAM
#define DMA_TRACE_ARG
#define DMA_TRACE_ARG_FWD
#define DMA_STAMP(slot) do { } while (0)
#endif

template <int BM, int BN, int WM, int WN, int KS, int D>
__device__ __forceinline__ void igemm_dma_body(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
    const bf16_t* __restrict__ addend, float* __restrict__ stats, const sba_conv_geom& g, const int M,
    float* __restrict__ ws, int* __restrict__ tickets, const int slabs_per_split, const EpiX ex, const int gx,
    const int gy, const int L, const int bz, const int nmajor DMA_TRACE_PARAM) {
    typedef bf16_t T;
    constexpr int TM = WM / 32, TN = WN / 32, WAVES_N = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN), NT = NW * 64;
    // every wave issues the same number of DMA loads per slab (the vmcnt bookkeeping is per wave): the weight
    // rows are padded to BNL, a multiple of 16 * NW; the padding rows load zeros (out of range) and are never read
    static_assert(BM % (16 * NW) == 0, "A rows divide evenly over the waves");
    constexpr int BNL = ((BN + 16 * NW - 1) / (16 * NW)) * (16 * NW);
    constexpr int AI = BM / (16 * NW), BI = BNL / (16 * NW);
    constexpr int SLAB_BYTES = (BM + BNL) * 64, STAGE_BYTES = KS * SLAB_BYTES, RING_BYTES = D * STAGE_BYTES;
    constexpr int LPS = KS * (AI + BI);                       // DMA loads per thread per stage
    static_assert(D >= 3 && (D - 2) * LPS <= 63, "vmcnt field");
    static_assert(RING_BYTES <= 160 * 1024, "LDS");           // (DMA destinations beyond 64 KiB work: tools/ldsdma_probe.hip)
    constexpr int EPI_OFF = BM * (BN * 2 + 16);
    constexpr int EPI_END = EPI_OFF + BM * 4 + BN * 8;
    constexpr int LDS_BYTES = RING_BYTES > EPI_END ? RING_BYTES : EPI_END;
    __shared__ __attribute__((aligned(1024))) unsigned char lds_all[LDS_BYTES];
    int* rowoff = reinterpret_cast<int*>(lds_all + EPI_OFF);
    float* s_stat = reinterpret_cast<float*>(lds_all + EPI_OFF + BM * 4);

    // workgroup -> tile: ids L and L + 8 share an XCD; the N tiles of an M tile take consecutive slots of one XCD
    // M-major: the N tiles of an M tile (they share its input rows) take consecutive slots of one XCD, each XCD's L2 fetches
    // the weights once (x read ~once, w up to 8 times); N-major (weight-heavy GEMM-like layers: a few M tiles against
    // megabytes of weights): the M tiles of an N tile on one XCD -- its weight slice is fetched by that XCD only
    const int xcd = L & 7, q = L >> 3;
    int mt, nt;
    if (nmajor) {
        nt = xcd + 8 * (q / gx);
        mt = q - (q / gx) * gx;
        if (nt >= gy) return;
    } else {
        mt = xcd + 8 * (q / gy);
        nt = q - (q / gy) * gy;
        if (mt >= gx) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int m_base = mt * BM, n_base = nt * BN;
    const int IHL = g.ups ? 2 * g.IH : g.IH, IWL = g.ups ? 2 * g.IW : g.IW;
    const int sub = g.OHs * g.OWs;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_all;

    // rows this thread's lane slot belongs to: instruction i of wave `wid` covers tile rows 16 * (wid + NW * i) ..+16
    const int rsub = lane >> 2;
    const uint32_t chunk = (uint32_t)((lane & 3) ^ ((lane >> 4) & 3));       // logical 16-byte chunk of the slot
    int a_iy0[AI], a_ix0[AI], a_nb[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m_base + 16 * (wid + NW * i) + rsub;
        if (m < M) {
            const int n = m / sub, rem = m - n * sub;
            const int oy = rem / g.OWs, ox = rem - oy * g.OWs;
            a_iy0[i] = oy * g.sy;
            a_ix0[i] = ox * g.sx;
            a_nb[i] = n * g.IH * g.IW;
        } else {
            a_iy0[i] = -100000;
            a_ix0[i] = 0;
            a_nb[i] = 0;
        }
    }
    const int cpt = g.Cin / 32;
    const int nsteps = g.ntaps * cpt;
    uint64_t tyb[2] = {0, 0}, txb[2] = {0, 0};
#pragma unroll
    for (int t = 0; t < SBA_MAX_TAPS; ++t) {
        tyb[t >> 4] |= (uint64_t)((g.ty[t] + 8) & 15) << (4 * (t & 15));
        txb[t >> 4] |= (uint64_t)((g.tx[t] + 8) & 15) << (4 * (t & 15));
    }
    const int xcs = g.x_cstride ? g.x_cstride : g.Cin;
    const int ycs = g.y_cstride ? g.y_cstride : g.Cout;
    const int s_begin = bz * slabs_per_split;
    const int s_end = min(s_begin + slabs_per_split, nsteps);

    int g_tap = s_begin / cpt, g_c = s_begin - g_tap * cpt, g_step = s_begin;
    int cur_tap = -1;
    constexpr uint32_t OOB = 0xFFFFFFFFu;
    uint32_t a_off[AI];           // per-lane byte offset of (pixel of the current tap, channel 0, this lane's chunk)
#pragma unroll
    for (int i = 0; i < AI; ++i) a_off[i] = OOB;
    uint32_t w_off[BI];           // per-lane byte offset of (co, K = 0, this lane's chunk): constant over the K loop
    const uint32_t krow_bytes = (uint32_t)g.ntaps * (uint32_t)g.Cin * 2u;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int r = 16 * (wid + NW * i) + rsub, co = n_base + r;
        w_off[i] = (r < BN && co < g.Cout) ? (uint32_t)co * krow_bytes + chunk * 16u : OOB;
    }
    const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.IH * g.IW * xcs * 2);
    const uint32_t w_bytes = (uint32_t)g.Cout * krow_bytes;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);
    const uint32_t lds_wave = lds_base + (uint32_t)(wid * 1024);
    uint32_t v_oob = OOB;
    asm volatile("" : "+v"(v_oob));         // a VGPR that holds the out-of-range offset (dead stages)

    // issue the DMA of one stage (KS consecutive slabs) into the ring buffer at LDS offset `dst`
    auto issue = [&](const uint32_t dst0) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const uint32_t dst = dst0 + (uint32_t)(k * SLAB_BYTES);
            if (g_step < s_end) {
                if (g_tap != cur_tap) {             // (uniform) new tap: per-lane pixel offsets
                    cur_tap = g_tap;
                    const int tsel = g_tap < SBA_MAX_TAPS ? g_tap : 0;
                    const uint64_t tyw = tsel < 16 ? tyb[0] : tyb[1], txw = tsel < 16 ? txb[0] : txb[1];
                    const int ty = (int)((tyw >> (4 * (tsel & 15))) & 15) - 8;
                    const int tx = (int)((txw >> (4 * (tsel & 15))) & 15) - 8;
#pragma unroll
                    for (int i = 0; i < AI; ++i) {
                        int iy = a_iy0[i] + ty, ix = a_ix0[i] + tx;
                        const bool ok = (iy >= 0) & (iy < IHL) & (ix >= 0) & (ix < IWL);
                        if (g.ups) { iy >>= 1; ix >>= 1; }
                        const uint32_t o = (uint32_t)(a_nb[i] + iy * g.IW + ix) * (uint32_t)(xcs * 2) +
                                           (uint32_t)(g.x_coff * 2) + chunk * 16u;
                        a_off[i] = ok ? o : OOB;
                    }
                }
                const uint32_t sa = (uint32_t)g_c * 64u, sw = (uint32_t)g_step * 64u;
#pragma unroll
                for (int i = 0; i < AI; ++i) lds_dma16(xr, a_off[i], sa, dst + (uint32_t)(NW * i * 1024));
#pragma unroll
                for (int i = 0; i < BI; ++i) lds_dma16(wr, w_off[i], sw, dst + (uint32_t)(BM * 64 + NW * i * 1024));
                ++g_step;
                if (++g_c == cpt) { g_c = 0; ++g_tap; }
            } else {                                 // past this split's K range: zeros (keeps the vmcnt count uniform)
#pragma unroll
                for (int i = 0; i < AI; ++i) lds_dma16(xr, v_oob, 0u, dst + (uint32_t)(NW * i * 1024));
#pragma unroll
                for (int i = 0; i < BI; ++i) lds_dma16(wr, v_oob, 0u, dst + (uint32_t)(BM * 64 + NW * i * 1024));
            }
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: lane l reads row (l & 31), logical chunk 2 * kk + (l >> 5), swizzled by its row
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 2) & 3;
    const int foff0 = fr * 64 + ((fh ^ fsw) << 4), foff1 = fr * 64 + (((2 + fh) ^ fsw) << 4);

    const int nstages = (s_end - s_begin + KS - 1) / KS;
    uint32_t idst = lds_wave;               // LDS destination (this wave's 1 KB slot) of the next stage to issue
    const uint32_t idst_end = lds_wave + (uint32_t)RING_BYTES;
#pragma unroll
    for (int p = 0; p < D - 1; ++p) { issue(idst); idst += STAGE_BYTES; }
    if (idst == idst_end) idst = lds_wave;
    const unsigned char* cptr = lds_all;
    for (int s = 0; s < nstages; ++s) {
        DMA_STAMP(0);
        wait_vmcnt<(D - 2) * LPS>();        // this wave's part of stage s has landed ...
        DMA_STAMP(1);
        wg_barrier();                       // ... and everybody else's; nobody reads buffer (s - 1) % D any more
        DMA_STAMP(2);
        issue(idst);                        // stage s + D - 1 (zeros past the end) into that buffer
        idst += STAGE_BYTES;
        if (idst == idst_end) idst = lds_wave;
        DMA_STAMP(3);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const unsigned char* base = cptr + k * SLAB_BYTES;
            const unsigned char* ar = base + wm0 * 64;
            const unsigned char* br = base + (BM + wn0) * 64;
            bf16x8_t a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a0[i] = *reinterpret_cast<const bf16x8_t*>(ar + i * 2048 + foff0);
                a1[i] = *reinterpret_cast<const bf16x8_t*>(ar + i * 2048 + foff1);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                b0[j] = *reinterpret_cast<const bf16x8_t*>(br + j * 2048 + foff0);
                b1[j] = *reinterpret_cast<const bf16x8_t*>(br + j * 2048 + foff1);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
        }
        cptr += STAGE_BYTES;
        if (cptr == lds_all + RING_BYTES) cptr = lds_all;
#ifdef SBA_DMA_TRACE
        { float keep = 0.f;             // make the stamp wait for the MFMA results of this stage
#pragma unroll
          for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) keep += acc[i][j][0];
          asm volatile("" :: "v"(keep)); }
        DMA_STAMP(4);
#endif
    }
    wait_vmcnt<0>();        // the dead stages issued past the end still write (zeros) into the ring
    wg_barrier();

    if (ws) {
        if (!splitk_arrive<TM, TN>(ws, tickets, acc, m_base + wm0, n_base + wn0, lane, M, g.Cout, L,
                                   reinterpret_cast<int*>(lds_all)))
            return;
    }
    for (int r = threadIdx.x; r < BM; r += NT) {
        const int m = m_base + r;
        int off = -1;
        if (m < M) {
            const int n = m / sub, rem = m - n * sub;
            const int oy = rem / g.OWs, ox = rem - oy * g.OWs;
            off = (n * g.OH + oy * g.osy + g.ooy) * g.OW + ox * g.osx + g.oox;
        }
        rowoff[r] = off;
    }
    for (int c = threadIdx.x; c < 2 * BN; c += NT) s_stat[c] = 0.f;
    __syncthreads();
    tile_epilogue<T, BM, BN, TM, TN, NT, LDS_BYTES>(acc, true, lds_all, rowoff, s_stat, wm0, wn0, lane, n_base, ycs, g, y,
                                                    addend, stats, ex, mt + bz);
}

template <int BM, int BN, int WM, int WN, int KS, int D>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void igemm_dma_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
    const bf16_t* __restrict__ addend, float* __restrict__ stats, const sba_conv_geom g, const int M,
    float* __restrict__ ws, int* __restrict__ tickets, const int slabs_per_split, const EpiX ex, const int gx,
    const int gy, const int nmajor DMA_TRACE_PARAM) {
    igemm_dma_body<BM, BN, WM, WN, KS, D>(x, w, y, addend, stats, g, M, ws, tickets, slabs_per_split, ex, gx, gy,
                                          (int)blockIdx.x, (int)blockIdx.z, nmajor DMA_TRACE_ARG_FWD);
}

// ---------------------------------------------------------------------------
// Second-generation LDS-DMA implicit GEMM (bf16, Cin % 64 == 0).  Per-stage s_memtime stamps of the kernel above
// (tools/trace_dma.py, profiles/r02_dma_stage_trace.txt) showed where a 64-deep stage of a 64x64 tile spends its
// ~1200 cycles: ~0 waiting for data, 80 in the barrier, ~430 ISSUING four DMA loads per wave (~100 cycles per
// buffer_load ... lds: each instruction touches 16 half cache lines) and ~450 in ds_read -> MFMA with nothing
// overlapping either.  Hence:
//   * slabs of 64 channels: LDS rows of 128 B = whole cache lines, 8 rows per DMA instruction (half the lines
//     per instruction); chunk c of row r lives at chunk c ^ ((r >> 1) & 7) -- every 16-lane phase of a
//     ds_read_b128 covers all 64 banks;
//   * the fragments of stage t+1 are read into a second register set right after the barrier, and the MFMAs of
//     stage t run interleaved with the DMA issue of stage t+D: LDS latency, matrix pipe and the load-issue
//     stall of one wave overlap each other;
//   * the buffer of stage t is free as soon as its fragments sit in registers (all waves' reads are drained
//     before the barrier), so the ring keeps D full stages in flight with D buffers.
// Everything else (zero-filled out-of-range taps / rows / dead stages, XCD-aware tile numbering, split-K into
// the f32 workspace, shared epilogue) is as in the kernel above.
// ---------------------------------------------------------------------------
// (the body is shared by the one-conv kernel and the grouped kernel below: L = the workgroup's index among this
// conv's tiles, bz = its K split)
template <int BM, int BN, int WM, int WN, int D>
__device__ __forceinline__ void igemm_dma2_body(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
    const bf16_t* __restrict__ addend, float* __restrict__ stats, const sba_conv_geom& g, const int M,
    float* __restrict__ ws, int* __restrict__ tickets, const int slabs_per_split, const EpiX ex, const int gx,
    const int gy, const int L, const int bz, const int nmajor DMA_TRACE_PARAM) {
    typedef bf16_t T;
    constexpr int TM = WM / 32, TN = WN / 32, WAVES_N = BN / WN;
    constexpr int NW = (BM / WM) * (BN / WN), NT = NW * 64;
    static_assert(BM % (8 * NW) == 0, "A rows divide evenly over the waves");
    constexpr int BNL = ((BN + 8 * NW - 1) / (8 * NW)) * (8 * NW);     // weight rows padded: same DMA count per wave
    constexpr int AI = BM / (8 * NW), BI = BNL / (8 * NW);
    constexpr int STAGE_BYTES = (BM + BNL) * 128, RING_BYTES = D * STAGE_BYTES;
    constexpr int LPS = AI + BI;                               // DMA loads per thread per stage
    constexpr int NM = 4 * TM * TN;                            // MFMAs per wave per stage
    static_assert(D >= 2 && (D - 1) * LPS <= 63, "vmcnt field");
    static_assert(RING_BYTES <= 160 * 1024, "LDS");
    constexpr int EPI_OFF = BM * (BN * 2 + 16);
    constexpr int EPI_END = EPI_OFF + BM * 4 + BN * 8;
    constexpr int LDS_BYTES = RING_BYTES > EPI_END ? RING_BYTES : EPI_END;
    __shared__ __attribute__((aligned(1024))) unsigned char lds_all[LDS_BYTES];
    int* rowoff = reinterpret_cast<int*>(lds_all + EPI_OFF);
    float* s_stat = reinterpret_cast<float*>(lds_all + EPI_OFF + BM * 4);

    // M-major: the N tiles of an M tile (they share its input rows) take consecutive slots of one XCD, each XCD's L2 fetches
    // the weights once (x read ~once, w up to 8 times); N-major (weight-heavy GEMM-like layers: a few M tiles against
    // megabytes of weights): the M tiles of an N tile on one XCD -- its weight slice is fetched by that XCD only
    const int xcd = L & 7, q = L >> 3;
    int mt, nt;
    if (nmajor) {
        nt = xcd + 8 * (q / gx);
        mt = q - (q / gx) * gx;
        if (nt >= gy) return;
    } else {
        mt = xcd + 8 * (q / gy);
        nt = q - (q / gy) * gy;
        if (mt >= gx) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int m_base = mt * BM, n_base = nt * BN;
    const int IHL = g.ups ? 2 * g.IH : g.IH, IWL = g.ups ? 2 * g.IW : g.IW;
    const int sub = g.OHs * g.OWs;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_all;

    // DMA instruction i of wave `wid` covers tile rows 8 * (wid + NW * i) .. +8, 8 lanes (128 B) per row
    const int rsub = lane >> 3;
    int a_iy0[AI], a_ix0[AI], a_nb[AI];
    uint32_t a_chunk[AI], w_off[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int r = 8 * (wid + NW * i) + rsub, m = m_base + r;
        a_chunk[i] = (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
        if (m < M) {
            const int n = m / sub, rem = m - n * sub;
            const int oy = rem / g.OWs, ox = rem - oy * g.OWs;
            a_iy0[i] = oy * g.sy;
            a_ix0[i] = ox * g.sx;
            a_nb[i] = n * g.IH * g.IW;
        } else {
            a_iy0[i] = -100000;
            a_ix0[i] = 0;
            a_nb[i] = 0;
        }
    }
    const int cpt = g.Cin / 64;               // slabs per tap
    const int nsteps = g.ntaps * cpt;
    uint64_t tyb[2] = {0, 0}, txb[2] = {0, 0};
#pragma unroll
    for (int t = 0; t < SBA_MAX_TAPS; ++t) {
        tyb[t >> 4] |= (uint64_t)((g.ty[t] + 8) & 15) << (4 * (t & 15));
        txb[t >> 4] |= (uint64_t)((g.tx[t] + 8) & 15) << (4 * (t & 15));
    }
    const int xcs = g.x_cstride ? g.x_cstride : g.Cin;
    const int ycs = g.y_cstride ? g.y_cstride : g.Cout;
    const int s_begin = bz * slabs_per_split;
    const int s_end = min(s_begin + slabs_per_split, nsteps);
    int g_tap = s_begin / cpt, g_c = s_begin - g_tap * cpt, g_step = s_begin;
    int cur_tap = -1;
    constexpr uint32_t OOB = 0xFFFFFFFFu;
    uint32_t a_off[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) a_off[i] = OOB;
    const uint32_t krow_bytes = (uint32_t)g.ntaps * (uint32_t)g.Cin * 2u;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int r = 8 * (wid + NW * i) + rsub, co = n_base + r;
        w_off[i] = (r < BN && co < g.Cout) ? (uint32_t)co * krow_bytes + (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 16)
                                           : OOB;
    }
    const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.IH * g.IW * xcs * 2);
    const uint32_t w_bytes = (uint32_t)g.Cout * krow_bytes;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);
    const uint32_t lds_wave = lds_base + (uint32_t)(wid * 1024);

    // per-stage bookkeeping, split from the loads so that the loads can be spread between the MFMAs
    bool st_live = false;
    uint32_t st_sa = 0, st_sw = 0;
    auto stage_begin = [&]() {
        st_live = g_step < s_end;
        if (st_live) {
            if (g_tap != cur_tap) {             // (uniform) new tap: per-lane pixel offsets
                cur_tap = g_tap;
                const int tsel = g_tap < SBA_MAX_TAPS ? g_tap : 0;
                const uint64_t tyw = tsel < 16 ? tyb[0] : tyb[1], txw = tsel < 16 ? txb[0] : txb[1];
                const int ty = (int)((tyw >> (4 * (tsel & 15))) & 15) - 8;
                const int tx = (int)((txw >> (4 * (tsel & 15))) & 15) - 8;
#pragma unroll
                for (int i = 0; i < AI; ++i) {
                    int iy = a_iy0[i] + ty, ix = a_ix0[i] + tx;
                    const bool ok = (iy >= 0) & (iy < IHL) & (ix >= 0) & (ix < IWL);
                    if (g.ups) { iy >>= 1; ix >>= 1; }
                    const uint32_t o = (uint32_t)(a_nb[i] + iy * g.IW + ix) * (uint32_t)(xcs * 2) +
                                       (uint32_t)(g.x_coff * 2) + a_chunk[i];
                    a_off[i] = ok ? o : OOB;
                }
            }
            st_sa = (uint32_t)g_c * 128u;
            st_sw = (uint32_t)g_step * 128u;
            ++g_step;
            if (++g_c == cpt) { g_c = 0; ++g_tap; }
        } else {
            st_sa = st_sw = 0u;
        }
    };
    auto stage_load = [&](const int qd, const uint32_t dst0) {       // qd = 0 .. LPS-1 (compile-time after unrolling)
        if (qd < AI) {
            const uint32_t o = st_live ? a_off[qd < AI ? qd : 0] : OOB;
            lds_dma16(xr, o, st_sa, dst0 + (uint32_t)(NW * qd * 1024));
        } else {
            const int i = qd - AI;
            const uint32_t o = st_live ? w_off[(i >= 0 && i < BI) ? i : 0] : OOB;
            lds_dma16(wr, o, st_sw, dst0 + (uint32_t)(BM * 128 + NW * i * 1024));
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: lane l reads row (l & 31), logical chunk 2 * kk + (l >> 5), swizzled by its row
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) foff[kk] = fr * 128 + (((2 * kk + fh) ^ fsw) << 4);

    struct Frags { bf16x8_t a[4][TM], b[4][TN]; };
    auto read_frags = [&](Frags& F, const unsigned char* base) {
        const unsigned char* ar = base + wm0 * 128;
        const unsigned char* br = base + (BM + wn0) * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) F.a[kk][i] = *reinterpret_cast<const bf16x8_t*>(ar + i * 4096 + foff[kk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) F.b[kk][j] = *reinterpret_cast<const bf16x8_t*>(br + j * 4096 + foff[kk]);
        }
    };

    const int nstages = s_end - s_begin;
    uint32_t idst = lds_wave;
    const uint32_t idst_end = lds_wave + (uint32_t)RING_BYTES;
#pragma unroll
    for (int p = 0; p < D; ++p) {
        stage_begin();
#pragma unroll
        for (int qd = 0; qd < LPS; ++qd) stage_load(qd, idst);
        idst += STAGE_BYTES;
    }
    idst = lds_wave;
    const unsigned char* rptr = lds_all;            // buffer of the stage whose fragments are read next
    Frags F0, F1;
    wait_vmcnt<(D - 1) * LPS>();
    wg_barrier();
    read_frags(F0, rptr);
    rptr += STAGE_BYTES;
    if (rptr == lds_all + RING_BYTES) rptr = lds_all;

    // one stage: fragments of stage t in Fc (their reads are in flight), stage t+1 -> Fn, DMA of stage t+D
    int s = 0;              // (stage counter of the trace build)
    auto stage = [&](Frags& Fc, Frags& Fn) {
        DMA_STAMP(0);
        wait_vmcnt<(D - 2) * LPS>();                // this wave's part of stage t+1 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ... and its reads of stage t are in registers
        DMA_STAMP(1);
        wg_barrier();                               // everybody's: buffer t is free, buffer t+1 is complete
        DMA_STAMP(2);
        read_frags(Fn, rptr);
        rptr += STAGE_BYTES;
        if (rptr == lds_all + RING_BYTES) rptr = lds_all;
        stage_begin();
        DMA_STAMP(3);
        int qd = 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fc.a[kk][i], Fc.b[kk][j], acc[i][j], 0, 0, 0);
                    const int m = (kk * TM + i) * TN + j;
                    // spread the LPS loads evenly between the NM MFMAs
                    if (((m + 1) * LPS) / NM > (m * LPS) / NM) {
#pragma unroll
                        for (int u = (m * LPS) / NM; u < ((m + 1) * LPS) / NM; ++u) stage_load(u, idst);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        (void)qd;
        idst += STAGE_BYTES;
        if (idst == idst_end) idst = lds_wave;
#ifdef SBA_DMA_TRACE
        { float keep = 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) keep += acc[i][j][0];
          asm volatile("" :: "v"(keep)); }
        DMA_STAMP(4);
#endif
        ++s;
    };
    (void)s;
    for (int t = 0; t < nstages; t += 2) {
        stage(F0, F1);
        if (t + 1 < nstages) stage(F1, F0);
    }
    wait_vmcnt<0>();        // the dead stages issued past the end still write (zeros) into the ring
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();

    if (ws) {
        if (!splitk_arrive<TM, TN>(ws, tickets, acc, m_base + wm0, n_base + wn0, lane, M, g.Cout, L,
                                   reinterpret_cast<int*>(lds_all)))
            return;
    }
    for (int r = threadIdx.x; r < BM; r += NT) {
        const int m = m_base + r;
        int off = -1;
        if (m < M) {
            const int n = m / sub, rem = m - n * sub;
            const int oy = rem / g.OWs, ox = rem - oy * g.OWs;
            off = (n * g.OH + oy * g.osy + g.ooy) * g.OW + ox * g.osx + g.oox;
        }
        rowoff[r] = off;
    }
    for (int c = threadIdx.x; c < 2 * BN; c += NT) s_stat[c] = 0.f;
    __syncthreads();
    tile_epilogue<T, BM, BN, TM, TN, NT, LDS_BYTES>(acc, true, lds_all, rowoff, s_stat, wm0, wn0, lane, n_base, ycs, g, y,
                                                    addend, stats, ex, mt + bz);
}

template <int BM, int BN, int WM, int WN, int D>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void igemm_dma2_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
    const bf16_t* __restrict__ addend, float* __restrict__ stats, const sba_conv_geom g, const int M,
    float* __restrict__ ws, int* __restrict__ tickets, const int slabs_per_split, const EpiX ex, const int gx,
    const int gy, const int nmajor DMA_TRACE_PARAM) {
    igemm_dma2_body<BM, BN, WM, WN, D>(x, w, y, addend, stats, g, M, ws, tickets, slabs_per_split, ex, gx, gy,
                                       (int)blockIdx.x, (int)blockIdx.z, nmajor DMA_TRACE_ARG_FWD);
}

// ---------------------------------------------------------------------------
// GROUPED launch: up to SBA_GROUP_MAX independent convolutions (the branches of one Inception block at one depth
// level: model.py:226-262 runs them one after the other) as ONE grid.  Each of them alone is 120..273 workgroups of
// a 64 x 64 tile on 256 CUs -- one wave per SIMD, nothing to overlap the LDS / DMA-issue latency of a stage with, a
// nearly empty second round, ~4.4 us of launch floor -- and hipGraph replay runs the branches' streams back to back.
// Grouped, their tiles share the chip: two workgroups per CU co-resident, one launch floor, one tail.  The item
// descriptors travel BY VALUE in the kernarg segment (pointers change every eager step; no device-side table to
// refresh, nothing for a captured graph to copy).
// ---------------------------------------------------------------------------
struct GroupItem {
    const bf16_t* x; const bf16_t* w; bf16_t* y; const bf16_t* addend; const float* bias; const void* mask;
    sba_conv_geom g;
    int M, gx, gy, tile_begin, nmajor, pad;
    float* ws;              // split-K partial sums of this item ([M][Cout] f32, zero-filled), or NULL
};
struct GroupArgs { int n; int sps; GroupItem it[SBA_GROUP_MAX]; };      // sps: 64-channel slabs per K split (0 = no split)

template <int BM, int BN, int WM, int WN, int D>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void igemm_dma2_group_kernel(const GroupArgs A DMA_TRACE_PARAM) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < SBA_GROUP_MAX; ++k)
        if (k < A.n && (int)blockIdx.x >= A.it[k].tile_begin) i = k;
    const GroupItem& it = A.it[i];
    const sba_conv_geom g = it.g;
    // A.sps != 0: grid.z K splits, partial sums added into the item's f32 workspace, finished by splitk_finish_group_kernel
    igemm_dma2_body<BM, BN, WM, WN, D>(it.x, it.w, it.y, it.addend, nullptr, g, it.M, A.sps ? it.ws : nullptr, nullptr,
                                       A.sps ? A.sps : g.ntaps * (g.Cin / 64), EpiX{it.bias, it.mask, 0}, it.gx, it.gy,
                                       (int)blockIdx.x - it.tile_begin, (int)blockIdx.z, it.nmajor DMA_TRACE_ARG_FWD);
}

// the same for members whose Cin is a multiple of 32 only (32-channel slabs, first-generation body)
template <int BM, int BN, int WM, int WN, int KS, int D>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void igemm_dma_group_kernel(const GroupArgs A DMA_TRACE_PARAM) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < SBA_GROUP_MAX; ++k)
        if (k < A.n && (int)blockIdx.x >= A.it[k].tile_begin) i = k;
    const GroupItem& it = A.it[i];
    const sba_conv_geom g = it.g;
    igemm_dma_body<BM, BN, WM, WN, KS, D>(it.x, it.w, it.y, it.addend, nullptr, g, it.M, nullptr, nullptr,
                                          g.ntaps * (g.Cin / 32), EpiX{it.bias, it.mask, 0}, it.gx, it.gy,
                                          (int)blockIdx.x - it.tile_begin, 0, it.nmajor DMA_TRACE_ARG_FWD);
}

// ---------------------------------------------------------------------------
// 3x3 stride-1 convolution (optionally behind a nearest x2 upsample) from a HALO TILE:
// a workgroup owns an 8 x 32 block of output pixels of one image and stages the (8+2) x (32+2)
// input pixels it touches (6 x 18 source pixels when upsampling) in LDS ONCE; the nine taps are then
// nine shifted views of that tile.  Compared with the generic implicit GEMM above this removes 8/9 of
// the A-operand global->LDS traffic, all per-slab address generation and half of the barriers (one per
// tap, for the double-buffered weight rows), so the main loop is just ds_read_b128 + MFMA.
// Pixel / weight rows are (2*CIN + 16) bytes apart: the 16-lane phases of a ds_read_b128 then hit 16
// distinct 4-bank groups (x2 upsampling: lane pairs share a pixel -> broadcast).
// Waves: wave w owns tile rows 2w, 2w+1 (two 32-pixel M tiles) x all BN output channels.
// ---------------------------------------------------------------------------
template <int CIN, int BN, int UPS>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ y, const bf16_t* __restrict__ addend,
                                                           float* __restrict__ stats, const sba_conv_geom g,
                                                           const EpiX ex) {
    typedef bf16_t T;
    constexpr int TH = 8, TW = 32, BM = TH * TW;
    constexpr int PIXB = CIN * 2 + 16;
    constexpr int HR = UPS ? TH / 2 + 2 : TH + 2, HC = UPS ? TW / 2 + 2 : TW + 2;
    constexpr int A_BYTES = HR * HC * PIXB, B_BYTES = BN * PIXB;
    constexpr int OUT_BYTES = BM * (BN * 2 + 16);         // the epilogue's bf16 staging tile reuses the buffers
    constexpr int STAGE = A_BYTES + 2 * B_BYTES > OUT_BYTES ? A_BYTES + 2 * B_BYTES : OUT_BYTES;
    constexpr int TM = 2, TN = BN / 32;
    constexpr int CPP = CIN / 8;                          // 16-byte chunks per pixel / weight row
    constexpr int BI = (BN * CPP + 255) / 256;            // weight chunks per thread per tap
    __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE + BM * 4 + BN * 8];
    unsigned char* const lA = lds;
    unsigned char* const lB = lds + A_BYTES;
    int* rowoff = reinterpret_cast<int*>(lds + STAGE);
    float* s_stat = reinterpret_cast<float*>(lds + STAGE + BM * 4);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_x = g.OW / TW, tiles_y = g.OH / TH;
    const int tx_ = blockIdx.x % tiles_x, ty_ = (blockIdx.x / tiles_x) % tiles_y, n = blockIdx.x / (tiles_x * tiles_y);
    const int oy0 = ty_ * TH, ox0 = tx_ * TW;
    const int n_base = blockIdx.y * BN;
    const int xcs = g.x_cstride ? g.x_cstride : g.Cin;
    const int ycs = g.y_cstride ? g.y_cstride : g.Cout;
    {
        const int r = tid;                                // BM == 256 threads
        rowoff[r] = (n * g.OH + oy0 + (r >> 5)) * g.OW + ox0 + (r & 31);
    }
    for (int c = tid; c < 2 * BN; c += 256) s_stat[c] = 0.f;

    constexpr uint32_t OOB = 0xFFFFFFFFu;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.IH * g.IW * xcs * 2);
    const uint32_t w_bytes = (uint32_t)((int64_t)g.Cout * 9 * g.Cin * 2);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);
    // Cin = nchunks x CIN: the halo tile is staged and walked once per CIN-channel chunk (same LDS footprint and
    // occupancy as Cin = CIN); the weight taps stream through the two B buffers across the chunk boundary
    const int nchunks = g.Cin / CIN, ntaps_all = 9 * nchunks;

    // ---- halo tile: source rows sy0 .. sy0+HR-1, columns sx0 .. sx0+HC-1 (zeros outside the image)
    const int sy0 = UPS ? (oy0 >> 1) - 1 : oy0 - 1, sx0 = UPS ? (ox0 >> 1) - 1 : ox0 - 1;
    // (loads are issued in batches of HB before the first LDS write: a load -> wait -> ds_write loop pays the
    // global latency once per iteration, 4 (upsampling) to 11 times per workgroup)
    constexpr int NCH = HR * HC * CPP, NI = (NCH + 255) / 256, HB = 6;
    auto stage_halo = [&](const int chunk) {
        const uint32_t cbytes = (uint32_t)(g.x_coff * 2 + chunk * CIN * 2);
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += HB) {
            u32x4_t hv[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const int idx = tid + 256 * (i0 + u);
                const int p = idx / CPP, ch = idx - p * CPP;
                const int hr = p / HC, hc = p - hr * HC;
                const int iy = sy0 + hr, ix = sx0 + hc;
                const bool ok = i0 + u < NI && idx < NCH && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
                const uint32_t o = ok ? (uint32_t)((n * g.IH + iy) * g.IW + ix) * (uint32_t)(xcs * 2) + cbytes +
                                            (uint32_t)ch * 16u
                                      : OOB;
                hv[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const int idx = tid + 256 * (i0 + u);
                const int p = idx / CPP, ch = idx - p * CPP;
                if (i0 + u < NI && idx < NCH)
                    *reinterpret_cast<uint4*>(lA + p * PIXB + ch * 16) = make_uint4(hv[u][0], hv[u][1], hv[u][2], hv[u][3]);
            }
        }
    };
    // ---- weights of one (chunk, tap): BN rows of CIN channels; t = chunk * 9 + tap
    uint4 rb[BI];
    auto bload = [&](const int t) {
        const int chunk = t / 9, tap = t - chunk * 9;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CPP, ch = idx - row * CPP;
            const int co = n_base + row;
            const uint32_t o = (row < BN && co < g.Cout)
                                   ? (((uint32_t)co * 9u + (uint32_t)tap) * (uint32_t)g.Cin + (uint32_t)(chunk * CIN)) * 2u +
                                         (uint32_t)ch * 16u
                                   : OOB;
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wr, o, 0, 0);
            rb[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto bstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CPP, ch = idx - row * CPP;
            if (row < BN) *reinterpret_cast<uint4*>(lB + buf * B_BYTES + row * PIXB + ch * 16) = rb[i];
        }
    };
    bload(0);
    bstore(0);

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int rl = lane & 31, hf = lane >> 5;
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        stage_halo(chunk);          // (chunk > 0: the barrier that ended the previous chunk's last tap freed lA)
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int t = chunk * 9 + tap;
            if (t + 1 < ntaps_all) bload(t + 1);
            const int ky = tap / 3, kx = tap - ky * 3;
            const unsigned char* ap[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int hr, hc;
                if (UPS) {
                    hr = ((2 * wid + i + ky - 1) >> 1) + 1;
                    hc = ((rl + kx - 1) >> 1) + 1;
                } else {
                    hr = 2 * wid + i + ky;
                    hc = rl + kx;
                }
                ap[i] = lA + (hr * HC + hc) * PIXB + hf * 16;
            }
            const unsigned char* bp = lB + (t & 1) * B_BYTES + rl * PIXB + hf * 16;
#pragma unroll
            for (int k16 = 0; k16 < CIN / 16; ++k16) {
                bf16x8_t a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(ap[i] + k16 * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(bp + j * 32 * PIXB + k16 * 32);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (t + 1 < ntaps_all) bstore((t + 1) & 1);
            __syncthreads();
        }
    }
    tile_epilogue<T, BM, BN, TM, TN, 256, STAGE>(acc, true, lds, rowoff, s_stat, wid * 64, 0, lane, n_base, ycs, g, y,
                                                 addend, stats, ex);
}

// ---------------------------------------------------------------------------
// Halo-tile 3x3 convolution, third form: WEIGHT FRAGMENTS IN REGISTERS.  PMC of the kernel above on the ResBlock conv
// (64 -> 64 at 128 x 128, B = 20; profiles/r04_pmc_halo_res128.txt): waves parked at a barrier / s_waitcnt 53 % of their
// cycles, matrix cores busy 12 %, LDS array active 11 % -- it is bound by the nine per-tap barriers of the
// double-buffered weight rows and by two workgroups per CU (69 KB of LDS each), not by LDS bandwidth or the MFMAs.
// Here the weights never enter LDS: they are packed FRAGMENT-MAJOR (sba_pack_frag_multi: the 64 lanes' 16-byte B
// fragments of one (tap, 32-channel column tile, 16-deep k-step) are 1 KB contiguous), every wave loads the eight
// fragments of the NEXT tap with eight coalesced 1 KB instructions while it multiplies the current one (they are L1 / L2
// hits: 72 KB per layer shared by the whole grid), and the main loop has no barrier at all: the halo tile is read-only
// after the staging barrier.  LDS = the halo tile alone (49 KB; 37 KB behind the nearest x2 upsample): three workgroups
// per CU, and half the LDS reads per MFMA (A fragments only).
// ---------------------------------------------------------------------------
template <int CIN, int UPS>
__global__ __launch_bounds__(256, 3) void conv3x3_halo3_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wf,
                                                               bf16_t* __restrict__ y, const bf16_t* __restrict__ addend,
                                                               float* __restrict__ stats, const sba_conv_geom g,
                                                               const EpiX ex) {
    typedef bf16_t T;
    constexpr int TH = 8, TW = 32, BM = TH * TW, BN = 64;
    constexpr int PIXB = CIN * 2 + 16;
    constexpr int HR = UPS ? TH / 2 + 2 : TH + 2, HC = UPS ? TW / 2 + 2 : TW + 2;
    constexpr int A_BYTES = HR * HC * PIXB;
    constexpr int OUT_BYTES = BM * (BN * 2 + 16);         // the epilogue's bf16 staging tile reuses the halo buffer
    constexpr int STAGE = A_BYTES > OUT_BYTES ? A_BYTES : OUT_BYTES;
    constexpr int TM = 2, TN = 2;
    constexpr int CPP = CIN / 8;                          // 16-byte chunks per pixel
    __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE + BM * 4 + BN * 8];
    unsigned char* const lA = lds;
    int* rowoff = reinterpret_cast<int*>(lds + STAGE);
    float* s_stat = reinterpret_cast<float*>(lds + STAGE + BM * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = g.OW / TW, tiles_y = g.OH / TH;
    const int tx_ = blockIdx.x % tiles_x, ty_ = (blockIdx.x / tiles_x) % tiles_y, n = blockIdx.x / (tiles_x * tiles_y);
    const int oy0 = ty_ * TH, ox0 = tx_ * TW;
    const int n_base = blockIdx.y * BN;
    const int xcs = g.x_cstride ? g.x_cstride : g.Cin;
    const int ycs = g.y_cstride ? g.y_cstride : g.Cout;
    rowoff[tid] = (n * g.OH + oy0 + (tid >> 5)) * g.OW + ox0 + (tid & 31);      // BM == 256 threads
    for (int c = tid; c < 2 * BN; c += 256) s_stat[c] = 0.f;

    constexpr uint32_t OOB = 0xFFFFFFFFu;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.IH * g.IW * xcs * 2);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const int nchunks = g.Cin / CIN, ntaps_all = 9 * nchunks;
    const int sy0 = UPS ? (oy0 >> 1) - 1 : oy0 - 1, sx0 = UPS ? (ox0 >> 1) - 1 : ox0 - 1;
    constexpr int NCH = HR * HC * CPP, NI = (NCH + 255) / 256, HB = 4;
    auto stage_halo = [&](const int chunk) {
        const uint32_t cbytes = (uint32_t)(g.x_coff * 2 + chunk * CIN * 2);
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += HB) {
            u32x4_t hv[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const int idx = tid + 256 * (i0 + u);
                const int p = idx / CPP, ch = idx - p * CPP;
                const int hr = p / HC, hc = p - hr * HC;
                const int iy = sy0 + hr, ix = sx0 + hc;
                const bool ok = i0 + u < NI && idx < NCH && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
                const uint32_t o = ok ? (uint32_t)((n * g.IH + iy) * g.IW + ix) * (uint32_t)(xcs * 2) + cbytes +
                                            (uint32_t)ch * 16u
                                      : OOB;
                hv[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const int idx = tid + 256 * (i0 + u);
                const int p = idx / CPP, ch = idx - p * CPP;
                if (i0 + u < NI && idx < NCH)
                    *reinterpret_cast<uint4*>(lA + p * PIXB + ch * 16) = make_uint4(hv[u][0], hv[u][1], hv[u][2], hv[u][3]);
            }
        }
    };
    // weight fragments of (chunk, tap): [n-block][chunk][tap][j][k16][lane][8]
    // weight fragments: [64-row block][tap][j][K / 16 k-steps][lane][8]; (chunk, tap) t -> k-steps chunk * CIN/16 ..
    struct BFrag { bf16x8_t v[TN][CIN / 16]; };
    const int KSW = g.Cin / 16;                 // k-steps of one (tap, j) row
    const bf16_t* const wbase = wf + ((int64_t)blockIdx.y * 9 * 2) * KSW * 512 + lane * 8;
    auto wptr = [&](const int t) {              // t = chunk * 9 + tap
        const int chunk = t / 9, tap = t - chunk * 9;
        return wbase + ((int64_t)(tap * 2) * KSW + chunk * (CIN / 16)) * 512;
    };
    auto bload = [&](BFrag& b, const int t) {
        const bf16_t* p = wptr(t);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int k = 0; k < CIN / 16; ++k)
                b.v[j][k] = *reinterpret_cast<const bf16x8_t*>(p + ((int64_t)j * KSW + k) * 512);
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int rl = lane & 31, hf = lane >> 5;
    // ONE set of weight fragments (32 registers): fragment (j, k16) of the NEXT tap is loaded into its registers right
    // behind the last MFMA that reads the current tap's -- a 16-MFMA (~0.2 us) head start on an L1 / L2 hit, the other
    // waves of the SIMD cover the rest (two sets, 64 registers, spilled at three waves per SIMD)
    BFrag b;
    bload(b, 0);
#pragma unroll 1
    for (int t = 0; t < ntaps_all; ++t) {
        const int chunk = t / 9, tap = t - chunk * 9;
        if (tap == 0) {                         // (wave-uniform) a new 64-channel chunk of Cin: (re)stage the halo tile
            if (chunk) __syncthreads();         // everybody has finished reading the previous chunk's tile
            stage_halo(chunk);
            __syncthreads();
        }
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned char* ap[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int hr, hc;
            if (UPS) {
                hr = ((2 * wid + i + ky - 1) >> 1) + 1;
                hc = ((rl + kx - 1) >> 1) + 1;
            } else {
                hr = 2 * wid + i + ky;
                hc = rl + kx;
            }
            ap[i] = lA + (hr * HC + hc) * PIXB + hf * 16;
        }
        const bool more = t + 1 < ntaps_all;
        const bf16_t* pn = wptr(more ? t + 1 : t);
#pragma unroll
        for (int k16 = 0; k16 < CIN / 16; ++k16) {
            bf16x8_t a[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(ap[i] + k16 * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b.v[j][k16], acc[i][j], 0, 0, 0);
            if (more) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b.v[j][k16] = *reinterpret_cast<const bf16x8_t*>(pn + ((int64_t)j * KSW + k16) * 512);
            }
        }
    }
    __syncthreads();                            // the epilogue stages through the halo buffer
    tile_epilogue<T, BM, BN, TM, TN, 256, STAGE>(acc, true, lds, rowoff, s_stat, wid * 64, 0, lane, n_base, ycs, g, y,
                                                 addend, stats, ex);
}

// ---------------------------------------------------------------------------
// The register-weight halo-tile kernel for ANY stride-1 3 x 3 window (round 4): ragged maps (tiles of 8 x 32 output
// pixels cut at the border), taps anywhere in a 3 x 3 window of offsets ('same', 'valid', their data gradients, flipped
// orders), Cin in chunks of CIN = 32 or 64 channels, 32 * TN output channels per workgroup.  Written for the first 3 x 3
// layers of the Inception trunk (model.py:170-199: 32 -> 32 and 32 -> 64 at 147 x 147, 80 -> 192 at 71 x 71) and their data
// gradients: the implicit-GEMM kernels gather every tap separately through L2 -> LDS (9x the input bytes) and run them
// at ~85 TFLOP/s -- 94 us for a layer whose tensors take 11 us to stream -- six launches on the image encoder's chain,
// the critical one of the step.
// ---------------------------------------------------------------------------
template <int CIN, int TN>
__global__ __launch_bounds__(256, 3) void conv3x3_halo3g_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wf,
                                                                bf16_t* __restrict__ y, const bf16_t* __restrict__ addend,
                                                                const sba_conv_geom g, const EpiX ex) {
    typedef bf16_t T;
    constexpr int TH = 8, TW = 32, BM = TH * TW, BN = 32 * TN;
    constexpr int PIXB = CIN * 2 + 16;
    constexpr int HR = TH + 2, HC = TW + 2;
    constexpr int A_BYTES = HR * HC * PIXB;
    constexpr int OUT_BYTES = BM * (BN * 2 + 16);
    constexpr int STAGE = A_BYTES > OUT_BYTES ? A_BYTES : OUT_BYTES;
    constexpr int TM = 2;
    constexpr int CPP = CIN / 8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE + BM * 4 + BN * 8];
    unsigned char* const lA = lds;
    int* rowoff = reinterpret_cast<int*>(lds + STAGE);
    float* s_stat = reinterpret_cast<float*>(lds + STAGE + BM * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (g.OW + TW - 1) / TW, tiles_y = (g.OH + TH - 1) / TH;
    const int tx_ = blockIdx.x % tiles_x, ty_ = (blockIdx.x / tiles_x) % tiles_y, n = blockIdx.x / (tiles_x * tiles_y);
    const int oy0 = ty_ * TH, ox0 = tx_ * TW;
    const int n_base = blockIdx.y * BN;
    const int xcs = g.x_cstride ? g.x_cstride : g.Cin;
    const int ycs = g.y_cstride ? g.y_cstride : g.Cout;
    {
        const int oy = oy0 + (tid >> 5), ox = ox0 + (tid & 31);
        rowoff[tid] = (oy < g.OH && ox < g.OW) ? (n * g.OH + oy) * g.OW + ox : -1;
    }
    for (int c = tid; c < 2 * BN; c += 256) s_stat[c] = 0.f;
    // the 3 x 3 window of tap offsets
    int ymin = g.ty[0], xmin = g.tx[0];
#pragma unroll
    for (int t = 1; t < 9; ++t) { ymin = min(ymin, (int)g.ty[t]); xmin = min(xmin, (int)g.tx[t]); }
    uint32_t kyx = 0;                               // 4 bits per tap: ky * 4 + kx (taps 0..7; tap 8 apart)
#pragma unroll
    for (int t = 0; t < 8; ++t) kyx |= (uint32_t)((g.ty[t] - ymin) * 4 + (g.tx[t] - xmin)) << (4 * t);
    const uint32_t kyx8 = (uint32_t)((g.ty[8] - ymin) * 4 + (g.tx[8] - xmin));

    constexpr uint32_t OOB = 0xFFFFFFFFu;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.IH * g.IW * xcs * 2);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const int nchunks = g.Cin / CIN, ntaps_all = 9 * nchunks;
    const int sy0 = oy0 + ymin, sx0 = ox0 + xmin;
    constexpr int NCH = HR * HC * CPP, NI = (NCH + 255) / 256, HB = 4;
    auto stage_halo = [&](const int chunk) {
        const uint32_t cbytes = (uint32_t)(g.x_coff * 2 + chunk * CIN * 2);
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += HB) {
            u32x4_t hv[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const int idx = tid + 256 * (i0 + u);
                const int p = idx / CPP, ch = idx - p * CPP;
                const int hr = p / HC, hc = p - hr * HC;
                const int iy = sy0 + hr, ix = sx0 + hc;
                const bool ok = i0 + u < NI && idx < NCH && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
                const uint32_t o = ok ? (uint32_t)((n * g.IH + iy) * g.IW + ix) * (uint32_t)(xcs * 2) + cbytes +
                                            (uint32_t)ch * 16u
                                      : OOB;
                hv[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const int idx = tid + 256 * (i0 + u);
                const int p = idx / CPP, ch = idx - p * CPP;
                if (i0 + u < NI && idx < NCH)
                    *reinterpret_cast<uint4*>(lA + p * PIXB + ch * 16) = make_uint4(hv[u][0], hv[u][1], hv[u][2], hv[u][3]);
            }
        }
    };
    // weight fragments: [64-row block][tap][j][K / 16][lane][8]; this workgroup's rows n_base .. n_base + BN
    struct BFrag { bf16x8_t v[TN][CIN / 16]; };
    const int KSW = g.Cin / 16;
    const int j0 = (n_base >> 5) & 1;               // BN = 32: the odd 32-row tiles are j = 1 of their 64-row block
    const bf16_t* const wbase = wf + ((int64_t)(n_base >> 6) * 9 * 2) * KSW * 512 + lane * 8;
    auto wptr = [&](const int t) {                  // t = chunk * 9 + tap
        const int chunk = t / 9, tap = t - chunk * 9;
        return wbase + ((int64_t)(tap * 2 + j0) * KSW + chunk * (CIN / 16)) * 512;
    };
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int rl = lane & 31, hf = lane >> 5;
    BFrag b;
    {
        const bf16_t* p = wptr(0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int k = 0; k < CIN / 16; ++k) b.v[j][k] = *reinterpret_cast<const bf16x8_t*>(p + ((int64_t)j * KSW + k) * 512);
    }
#pragma unroll 1
    for (int t = 0; t < ntaps_all; ++t) {
        const int chunk = t / 9, tap = t - chunk * 9;
        if (tap == 0) {
            if (chunk) __syncthreads();
            stage_halo(chunk);
            __syncthreads();
        }
        const uint32_t code = tap < 8 ? (kyx >> (4 * tap)) & 15u : kyx8;
        const int ky = (int)(code >> 2), kx = (int)(code & 3u);
        const unsigned char* ap[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) ap[i] = lA + ((2 * wid + i + ky) * HC + rl + kx) * PIXB + hf * 16;
        const bool more = t + 1 < ntaps_all;
        const bf16_t* pn = wptr(more ? t + 1 : t);
#pragma unroll
        for (int k16 = 0; k16 < CIN / 16; ++k16) {
            bf16x8_t a[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(ap[i] + k16 * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b.v[j][k16], acc[i][j], 0, 0, 0);
            if (more) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b.v[j][k16] = *reinterpret_cast<const bf16x8_t*>(pn + ((int64_t)j * KSW + k16) * 512);
            }
        }
    }
    __syncthreads();
    tile_epilogue<T, BM, BN, TM, TN, 256, STAGE>(acc, true, lds, rowoff, s_stat, wid * 64, 0, lane, n_base, ycs, g, y,
                                                 addend, nullptr, ex);
}

// row-major packed conv operand [R][taps][K] (bf16; R a multiple of 32, K of 16) -> fragment-major
// [ceil(R/64)][taps][2][K/16][64 lanes][8]: lane = ((k >> 3) & 1) * 32 + (r & 31).  One thread = 16 bytes.
__global__ __launch_bounds__(256) void pack_frag_kernel(const sba_frag_desc* __restrict__ descs, const int ndesc,
                                                        const int total_units) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= total_units) return;
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].unit_begin <= u) lo = mid; else hi = mid - 1;
    }
    const sba_frag_desc d = descs[lo];
    const int local = u - d.unit_begin;
    const int k8n = d.K / 8;
    const int k8 = local % k8n, rest = local / k8n;
    const int t = rest % d.taps, r = rest / d.taps;
    const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(d.src) + ((int64_t)r * d.taps + t) * d.K + 8 * k8);
    const int nb = r >> 6, j = (r >> 5) & 1, nn = r & 31, k16 = k8 >> 1, gg = k8 & 1, ks = d.K / 16;
    const int64_t o = (((((int64_t)nb * d.taps + t) * 2 + j) * ks + k16) * 512) + (gg * 32 + nn) * 8;
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(d.dst) + o) = v;
}

// ---------------------------------------------------------------------------
// Persistent halo-tile 3x3 convolution (Cin = 64): the kernel above pays, per 8 x 32 output tile, a serialised
// prologue (halo staging through registers, first weight tap), nine barriers for the double-buffered weight
// taps and an epilogue -- 17.8 us per workgroup-round for 5.8 us of MFMA issue.  Here
//   * a workgroup is PERSISTENT (one per CU, 160 KB of LDS) and walks tiles t = id, id + G, ...;
//   * the weights of all nine taps stay in LDS for the whole launch (9 x 64 x 128 B = 72 KB, loaded once);
//   * the halo tile of tile t+1 is fetched by LDS-DMA into the second halo buffer while tile t's 144 MFMAs per
//     wave run -- no barrier inside a tile's main loop, one counted vmcnt wait + barrier per tile;
//   * rows are 128 B (64 channels) with the chunk swizzle c ^ ((row >> 1) & 7): conflict-free ds_read_b128 for
//     the shifted tap views and for the weight rows alike; out-of-image halo pixels are out-of-range DMA lanes
//     (zeros).
// Epilogue (BatchNorm statistics, bias / ReLU / mask, addend, LDS-transposed NHWC store) is the shared one and
// stages through the halo buffer the tile has just finished reading.
// ---------------------------------------------------------------------------
template <int UPS>
__global__ __launch_bounds__(256) void conv3x3_halo2_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            bf16_t* __restrict__ y, const bf16_t* __restrict__ addend,
                                                            float* __restrict__ stats, const sba_conv_geom g,
                                                            const EpiX ex, const int tiles, const int wgs_per_nblock) {
    typedef bf16_t T;
    constexpr int TH = 8, TW = 32, BM = TH * TW, BN = 64;
    constexpr int HR = UPS ? TH / 2 + 2 : TH + 2, HC = UPS ? TW / 2 + 2 : TW + 2, HP = HR * HC;
    constexpr int W_BYTES = 9 * BN * 128;                       // 73,728
    constexpr int OUT_BYTES = BM * (BN * 2 + 16);               // 36,864: the epilogue's staging tile
    constexpr int HI = (HP * 8 + 255) / 256;                    // DMA instructions per wave per halo tile
    constexpr int WI = 9 * BN * 8 / 256;                        // ... for the weights (18)
    // a halo buffer holds the DMA image (HI * 4 slots of 1 KB: the last slots are partly out-of-range lanes, which
    // write zeros) and, during the epilogue, the staging tile + the row table + the statistics accumulators
    constexpr int EPI_OFF = OUT_BYTES;
    constexpr int HALO_NEED = HI * 4 * 1024 > EPI_OFF + BM * 4 + BN * 8 ? HI * 4 * 1024 : EPI_OFF + BM * 4 + BN * 8;
    constexpr int HALO_BYTES = (HALO_NEED + 1023) / 1024 * 1024;
    constexpr int TM = 2, TN = 2;
    constexpr int LDS_BYTES = W_BYTES + 2 * HALO_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(HI <= 63, "vmcnt field");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    unsigned char* const lW = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = blockIdx.x / wgs_per_nblock, wg = blockIdx.x - nb * wgs_per_nblock;
    const int n_base = nb * BN;
    const int tiles_x = g.OW / TW, tiles_y = g.OH / TH;
    const int xcs = g.x_cstride ? g.x_cstride : g.Cin;
    const int ycs = g.y_cstride ? g.y_cstride : g.Cout;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    constexpr uint32_t OOB = 0xFFFFFFFFu;
    const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.IH * g.IW * xcs * 2);
    const uint32_t w_bytes = (uint32_t)((int64_t)g.Cout * 9 * 64 * 2);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);

    // ---- weights: LDS row R = tap * 64 + co (128 B), DMA instruction q of wave `wid` covers rows 8 * (wid + 4 q) ..
#pragma unroll
    for (int q = 0; q < WI; ++q) {
        const int R = 8 * (wid + 4 * q) + (lane >> 3);
        const int tap = R >> 6, co = n_base + (R & 63);
        const uint32_t c = (uint32_t)((lane & 7) ^ ((R >> 1) & 7));
        const uint32_t o = co < g.Cout ? ((uint32_t)co * 9u + (uint32_t)tap) * 128u + c * 16u : OOB;
        lds_dma16(wr, o, 0u, lds_base + (uint32_t)((wid + 4 * q) * 1024));
    }
    // ---- halo tile of output tile `t` into buffer `buf`: lane slot idx = 64 * (wid + 4 q) + lane -> pixel idx >> 3
    auto issue_halo = [&](const int t, const int buf) {
        const int tx_ = t % tiles_x, ty_ = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int oy0 = ty_ * TH, ox0 = tx_ * TW;
        const int sy0 = UPS ? (oy0 >> 1) - 1 : oy0 - 1, sx0 = UPS ? (ox0 >> 1) - 1 : ox0 - 1;
        const uint32_t dst = lds_base + (uint32_t)(W_BYTES + buf * HALO_BYTES + wid * 1024);
        const bool live = t < tiles;
#pragma unroll
        for (int q = 0; q < HI; ++q) {
            const int idx = 64 * (wid + 4 * q) + lane;
            const int p = idx >> 3;
            const int hr = p / HC, hc = p - hr * HC;
            const int iy = sy0 + hr, ix = sx0 + hc;
            const bool ok = live && p < HP && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
            const uint32_t c = (uint32_t)((idx & 7) ^ ((p >> 1) & 7));
            const uint32_t o = ok ? (uint32_t)((n * g.IH + iy) * g.IW + ix) * (uint32_t)(xcs * 2) +
                                        (uint32_t)(g.x_coff * 2) + c * 16u
                                  : OOB;
            lds_dma16(xr, o, 0u, dst + (uint32_t)(4 * q * 1024));
        }
    };
    int t = wg;
    issue_halo(t, 0);
    const int rl = lane & 31, hf = lane >> 5;
    int buf = 0;
    for (; t < tiles; t += wgs_per_nblock, buf ^= 1) {
        // next tile's halo into the other buffer (its last reader was tile t-1's epilogue, which ended with a barrier)
        issue_halo(t + wgs_per_nblock, buf ^ 1);
        wait_vmcnt<HI>();                   // everything but the HI loads just issued: this tile's halo (and the weights)
        wg_barrier();
        const unsigned char* lA = lds + W_BYTES + buf * HALO_BYTES;
        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int bsw = (rl >> 1) & 7;
        // one wave per SIMD: nothing else hides the LDS latency, so the 16 fragments of tap t+1 are read into a
        // second register set while the 16 MFMAs of tap t (512 cycles) run
        struct TapFrags { bf16x8_t a[4][TM], b[4][TN]; };
        auto load_tap = [&](TapFrags& F, const int tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            int hp[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int hr, hc;
                if (UPS) {
                    hr = ((2 * wid + i + ky - 1) >> 1) + 1;
                    hc = ((rl + kx - 1) >> 1) + 1;
                } else {
                    hr = 2 * wid + i + ky;
                    hc = rl + kx;
                }
                hp[i] = hr * HC + hc;
            }
            const unsigned char* bp = lW + (tap * 64 + rl) * 128;
#pragma unroll
            for (int k16 = 0; k16 < 4; ++k16) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    F.a[k16][i] = *reinterpret_cast<const bf16x8_t*>(lA + hp[i] * 128 + (((2 * k16 + hf) ^ ((hp[i] >> 1) & 7)) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    F.b[k16][j] = *reinterpret_cast<const bf16x8_t*>(bp + j * 32 * 128 + (((2 * k16 + hf) ^ bsw) << 4));
            }
        };
        auto mma_tap = [&](const TapFrags& F) {
#pragma unroll
            for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[k16][i], F.b[k16][j], acc[i][j], 0, 0, 0);
        };
        TapFrags F0, F1;
        load_tap(F0, 0);
#pragma unroll
        for (int tap = 0; tap < 8; tap += 2) {
            load_tap(F1, tap + 1);
            mma_tap(F0);
            __builtin_amdgcn_sched_barrier(0);
            load_tap(F0, tap + 2);
            mma_tap(F1);
            __builtin_amdgcn_sched_barrier(0);
        }
        mma_tap(F0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();           // every wave has finished reading this tile's halo: the epilogue may stage through it
        unsigned char* const stage = lds + W_BYTES + buf * HALO_BYTES;
        int* rowoff = reinterpret_cast<int*>(stage + EPI_OFF);
        float* s_stat = reinterpret_cast<float*>(stage + EPI_OFF + BM * 4);
        {
            const int tx_ = t % tiles_x, ty_ = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
            rowoff[tid] = (n * g.OH + ty_ * TH + (tid >> 5)) * g.OW + tx_ * TW + (tid & 31);      // BM == 256 threads
            if (tid < 2 * BN) s_stat[tid] = 0.f;
        }
        __syncthreads();
        tile_epilogue<T, BM, BN, TM, TN, 256, HALO_BYTES>(acc, true, stage, rowoff, s_stat, wid * 64, 0, lane, n_base, ycs,
                                                          g, y, addend, stats, ex, t);
        __syncthreads();        // staging tile / row table / accumulators free again (next tile's halo DMA lands here)
    }
    wait_vmcnt<0>();            // the dead halo issued for the tile past the end
}

// ---------------------------------------------------------------------------
// weight gradient: dw[co][tap][ci] += sum_pixels dy[pixel][co] * x[gather(pixel,tap)][ci]
// Workgroup = one 64(co) x 64(ci) tile of one tap; its 4 waves each walk their
// own 16-pixel slices of the workgroup's pixel range, then reduce through LDS.
// Both operands are pixel-major in memory; the MFMA wants 8 consecutive
// pixels per lane, so bf16 fragments are read with ds_read_b64_tr_b16 (4
// pixels x 16 channels transposed per 16-lane group); f32 fragments are single
// elements and need no transpose.
// ---------------------------------------------------------------------------
template <typename T> struct WgFrag;

template <> struct WgFrag<bf16_t> {
    static constexpr int ROWS = 64 * 2 + 64;   // bytes per pixel row: 128 data + 64 pad (bank spread)
    // fragment of channels [c32, c32+32) over pixels [0,16) of a slice
    static __device__ __forceinline__ bf16x8_t load(const unsigned char* slice, int c32, int lane) {
        const int g16 = lane >> 4, i16 = lane & 15;
        const int cbase = c32 + 16 * (g16 & 1), kbase = 8 * (g16 >> 1);
        const int q = i16 >> 2, p = i16 & 3;
        const unsigned char* a0 = slice + (kbase + q) * ROWS + (cbase + 4 * p) * 2;
        typedef __attribute__((address_space(3))) s16x4_t* lptr;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * ROWS));
        bf16x8_t r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
    static __device__ __forceinline__ void mma(const unsigned char* sa, const unsigned char* sb, int lane,
                                               f32x16_t (&acc)[2][2]) {
        bf16x8_t a[2], b[2];
        a[0] = load(sa, 0, lane); a[1] = load(sa, 32, lane);
        b[0] = load(sb, 0, lane); b[1] = load(sb, 32, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
};

template <> struct WgFrag<float> {
    static constexpr int ROWS = 64 * 4 + 64;
    static __device__ __forceinline__ void mma(const unsigned char* sa, const unsigned char* sb, int lane,
                                               f32x16_t (&acc)[2][2]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float*>(sa + (2 * kk + h) * ROWS + (i * 32 + r) * 4);
                b[i] = *reinterpret_cast<const float*>(sb + (2 * kk + h) * ROWS + (i * 32 + r) * 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                    float* __restrict__ dw, const sba_conv_geom g,
                                                    const int M, const int chunks_per_split,
                                                    const int use_atomic, const FastDiv dsub, const FastDiv dow,
                                                    const int64_t zstride) {
    constexpr int ROWS = WgFrag<T>::ROWS;
    constexpr int CH = 16 / (int)sizeof(T);          // elements per 16-byte chunk
    constexpr int CPR = 64 / CH;                     // chunks per 64-channel pixel row
    constexpr int LPT = 16 * CPR / 64;               // 16-byte loads per lane per slice
    constexpr int SLICE = 16 * ROWS;
    static_assert(4 * 2 * SLICE >= 64 * 64 * 4, "reduction buffer fits in the staging area");
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 2 * SLICE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int co0 = blockIdx.x * 64;
    const int ci_tiles = (g.Cin + 63) / 64;
    const int tap = blockIdx.y / ci_tiles, ci0 = (blockIdx.y - tap * ci_tiles) * 64;
    int ty = 0, tx = 0;
#pragma unroll
    for (int t = 0; t < SBA_MAX_TAPS; ++t)
        if (t == tap) { ty = g.ty[t]; tx = g.tx[t]; }
    const int IHL = g.ups ? 2 * g.IH : g.IH, IWL = g.ups ? 2 * g.IW : g.IW;
    const int sub = g.OHs * g.OWs;

    unsigned char* sa = lds + wid * 2 * SLICE;
    unsigned char* sb = sa + SLICE;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int chunk_lo = blockIdx.z * chunks_per_split;
    const int total_chunks = (M + 63) / 64;
    const int chunk_hi = min(chunk_lo + chunks_per_split, total_chunks);

    uint4 va[LPT], vb[LPT];
    auto gload = [&](int ck) {
        const int m0 = ck * 64 + wid * 16;          // this wave's 16 pixels
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int idx = lane + 64 * u;
            const int pix = idx / CPR, cc = idx - pix * CPR;
            const int m = m0 + pix;
            va[u] = make_uint4(0, 0, 0, 0);
            vb[u] = make_uint4(0, 0, 0, 0);
            if (m < M) {
                const int n = (int)fdiv(m, dsub), rem = m - n * sub;
                const int oy = (int)fdiv(rem, dow), ox = rem - oy * g.OWs;
                const int co = co0 + cc * CH;
                if (co < g.Cout) {
                    const int64_t po = (int64_t)(n * g.OH + oy * g.osy + g.ooy) * g.OW + ox * g.osx + g.oox;
                    va[u] = *reinterpret_cast<const uint4*>(dy + po * g.Cout + co);
                }
                int iy = oy * g.sy + ty, ix = ox * g.sx + tx;
                const bool ok = (iy >= 0) & (iy < IHL) & (ix >= 0) & (ix < IWL);
                if (g.ups) { iy >>= 1; ix >>= 1; }
                const int ci = ci0 + cc * CH;
                if (ok && ci < g.Cin) {
                    const int64_t pi = (int64_t)(n * g.IH + iy) * g.IW + ix;
                    vb[u] = *reinterpret_cast<const uint4*>(x + pi * g.Cin + ci);
                }
            }
        }
    };
    if (chunk_lo < chunk_hi) gload(chunk_lo);
    for (int ck = chunk_lo; ck < chunk_hi; ++ck) {
        __syncthreads();    // previous slice fully consumed by this wave's MFMA reads
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int idx = lane + 64 * u;
            const int pix = idx / CPR, cc = idx - pix * CPR;
            *reinterpret_cast<uint4*>(sa + pix * ROWS + cc * 16) = va[u];
            *reinterpret_cast<uint4*>(sb + pix * ROWS + cc * 16) = vb[u];
        }
        __syncthreads();
        if (ck + 1 < chunk_hi) gload(ck + 1);     // next chunk's loads fly under this chunk's MFMAs
        WgFrag<T>::mma(sa, sb, lane, acc);
    }

    // cross-wave reduction through LDS, wave by wave (a fixed order: no LDS atomics), then one add per element
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
    const int col_l = lane & 31, rsel = 4 * (lane >> 5);
#pragma unroll 1
    for (int wv = 0; wv < 4; ++wv) {
        if (wid == wv) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + rsel;   // co
                        float* q = &red[row * 64 + j * 32 + col_l];
                        *q = wv == 0 ? acc[i][j][r] : *q + acc[i][j][r];
                    }
        }
        __syncthreads();
    }
    float* dwz = dw + (int64_t)blockIdx.z * zstride;    // deterministic mode: this pixel split's own partial tensor
    for (int i = tid; i < 64 * 64; i += 256) {
        const int co = co0 + (i >> 6), ci = ci0 + (i & 63);
        if (co < g.Cout && ci < g.Cin) {
            float* p = dwz + ((int64_t)co * g.ntaps + tap) * g.Cin + ci;
            if (use_atomic == 1) atomicAdd(p, red[i]);
            else if (use_atomic == 2) *p = red[i];
            else *p += red[i];
        }
    }
}

// ---------------------------------------------------------------------------
// weight gradient, small-pixel-count regime (GEMM-like layers at 4x4 / 8x8 maps with
// thousands of channels): every wave owns its own 64(co) x 64(ci) tile of one tap and walks
// ALL pixels of the block's range, so there is no cross-wave reduction; the four waves of a
// workgroup share the dy slice (same co tile) and differ in (tap, ci tile).  CT = 2: every wave owns
// TWO co tiles (128 x 64 outputs) against the same x slice -- these launches are bound by the L2 traffic
// of the operand slices (each wave streams its own), 0.375 instead of 0.625 KB per MFMA.
// ---------------------------------------------------------------------------
template <typename T, int CT>
__global__ __launch_bounds__(256, 2) void wgrad_small_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          float* __restrict__ dw, const sba_conv_geom g,
                                                          const int M, const int chunks_per_split,
                                                          const int use_atomic, const FastDiv dsub,
                                                          const FastDiv dow, const int64_t zstride) {
    constexpr int ROWS = WgFrag<T>::ROWS;
    constexpr int CH = 16 / (int)sizeof(T);
    constexpr int CPR = 64 / CH;
    constexpr int LPT = 16 * CPR / 64;               // 16-byte loads per lane for a wave-private slice
    constexpr int APT = (16 * CPR * CT + 255) / 256; // 16-byte loads per thread for the shared dy slice(s)
    constexpr int SLICE = 16 * ROWS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[(CT + 4) * SLICE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int co0 = blockIdx.x * (64 * CT);
    const int ci_tiles = (g.Cin + 63) / 64;
    const int item = blockIdx.y * 4 + wid;
    const bool active = item < g.ntaps * ci_tiles;
    const int tap = active ? item / ci_tiles : 0;
    const int ci0 = active ? (item - tap * ci_tiles) * 64 : 0;
    int ty = 0, tx = 0;
#pragma unroll
    for (int t = 0; t < SBA_MAX_TAPS; ++t)
        if (t == tap) { ty = g.ty[t]; tx = g.tx[t]; }
    const int IHL = g.ups ? 2 * g.IH : g.IH, IWL = g.ups ? 2 * g.IW : g.IW;
    const int sub = g.OHs * g.OWs;

    unsigned char* sa = lds;
    unsigned char* sb = lds + (CT + wid) * SLICE;

    f32x16_t acc[CT][2][2];
#pragma unroll
    for (int s = 0; s < CT; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][i][j][r] = 0.f;

    const int total_chunks = (M + 15) / 16;
    const int chunk_lo = blockIdx.z * chunks_per_split;
    const int chunk_hi = min(chunk_lo + chunks_per_split, total_chunks);

    uint4 va[APT], vb[LPT];
    auto gload = [&](int ck) {
        const int m0 = ck * 16;
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const int idx = tid + 256 * u;
            va[u] = make_uint4(0, 0, 0, 0);
            if (idx < 16 * CPR * CT) {
                const int pix = idx / (CPR * CT), cc = idx - pix * (CPR * CT);
                const int m = m0 + pix, co = co0 + cc * CH;
                if (m < M && co < g.Cout) {
                    const int n = (int)fdiv(m, dsub), rem = m - n * sub;
                    const int oy = (int)fdiv(rem, dow), ox = rem - oy * g.OWs;
                    const int64_t po = (int64_t)(n * g.OH + oy * g.osy + g.ooy) * g.OW + ox * g.osx + g.oox;
                    va[u] = *reinterpret_cast<const uint4*>(dy + po * g.Cout + co);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int idx = lane + 64 * u;
            const int pix = idx / CPR, cc = idx - pix * CPR;
            const int m = m0 + pix;
            vb[u] = make_uint4(0, 0, 0, 0);
            if (active && m < M) {
                const int n = (int)fdiv(m, dsub), rem = m - n * sub;
                const int oy = (int)fdiv(rem, dow), ox = rem - oy * g.OWs;
                int iy = oy * g.sy + ty, ix = ox * g.sx + tx;
                const bool ok = (iy >= 0) & (iy < IHL) & (ix >= 0) & (ix < IWL);
                if (g.ups) { iy >>= 1; ix >>= 1; }
                const int ci = ci0 + cc * CH;
                if (ok && ci < g.Cin) {
                    const int64_t pi = (int64_t)(n * g.IH + iy) * g.IW + ix;
                    vb[u] = *reinterpret_cast<const uint4*>(x + pi * g.Cin + ci);
                }
            }
        }
    };
    if (chunk_lo < chunk_hi) gload(chunk_lo);
    for (int ck = chunk_lo; ck < chunk_hi; ++ck) {
        __syncthreads();          // everyone is done reading the previous slices
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const int idx = tid + 256 * u;
            if (idx < 16 * CPR * CT) {
                const int pix = idx / (CPR * CT), cc = idx - pix * (CPR * CT);
                *reinterpret_cast<uint4*>(sa + (cc / CPR) * SLICE + pix * ROWS + (cc % CPR) * 16) = va[u];
            }
        }
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int idx = lane + 64 * u;
            const int pix = idx / CPR, cc = idx - pix * CPR;
            *reinterpret_cast<uint4*>(sb + pix * ROWS + cc * 16) = vb[u];
        }
        __syncthreads();
        if (ck + 1 < chunk_hi) gload(ck + 1);
#pragma unroll
        for (int s = 0; s < CT; ++s) WgFrag<T>::mma(sa + s * SLICE, sb, lane, acc[s]);
    }

    if (!active) return;
    const int col_l = lane & 31, rsel = 4 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < CT; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + s * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                    const int ci = ci0 + j * 32 + col_l;
                    if (co < g.Cout && ci < g.Cin) {
                        float* p = dw + (int64_t)blockIdx.z * zstride + ((int64_t)co * g.ntaps + tap) * g.Cin + ci;
                        if (use_atomic == 1) atomicAdd(p, acc[s][i][j][r]);
                        else if (use_atomic == 2) *p = acc[s][i][j][r];       // first write of a cleared gradient
                        else *p += acc[s][i][j][r];
                    }
                }
}

// ---------------------------------------------------------------------------
// The same decomposition (one 64 x 64 tile of one tap per wave, the workgroup's waves share the dy slices) fed by
// LDS-DMA through a D-deep ring instead of register staging.  The register-staged kernel above prefetches ONE
// 16-pixel chunk (four MFMAs, ~0.1 us) ahead of an L2 round trip of ~1 us, so every chunk costs a full memory
// latency: 320 pixels = 20 chunks ~ 30 us per workgroup whatever the MFMA work is.  Here a stage is 32 pixels
// (two MFMA k-steps): per wave four 1 KB DMA instructions for its own x slice (the im2col gather is the per-lane
// address) and CT for its share of the dy slices; D - 1 stages are in flight, completion is counted with
// s_waitcnt vmcnt + one workgroup barrier per stage, exactly like igemm_dma_kernel.  Rows are 128 bytes (64
// channels, no padding: the DMA writes lane l at M0 + 16 l); the 16-byte chunk position is XORed with bit 1 of the
// row so that the four rows x 32 bytes a 16-lane group reads with ds_read_b64_tr_b16 fall in different banks.
// ---------------------------------------------------------------------------
struct WgFragDma {
    // fragment of channels [c32, c32+32) over pixels [16 k16, 16 k16 + 16) of a 32-row slice
    static __device__ __forceinline__ bf16x8_t load(const unsigned char* slice, int k16, int c32, int lane) {
        const int g16 = lane >> 4, i16 = lane & 15;
        const int cbase = c32 + 16 * (g16 & 1), row = 16 * k16 + 8 * (g16 >> 1) + (i16 >> 2), p = i16 & 3;
        const int cpos = (cbase >> 3) ^ (((row >> 1) & 1) << 2);
        const unsigned char* a0 = slice + row * 128 + cpos * 16 + p * 8;
        typedef __attribute__((address_space(3))) s16x4_t* lptr;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * 128));
        bf16x8_t r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
};

template <int CT, int D>
__global__ __launch_bounds__(256, 2) void wgrad_small_dma_kernel(const bf16_t* __restrict__ x,
                                                              const bf16_t* __restrict__ dy,
                                                              float* __restrict__ dw, const sba_conv_geom g,
                                                              const int M, const int chunks_per_split,
                                                              const int use_atomic, const FastDiv dsub,
                                                              const FastDiv dow, const int64_t zstride) {
    constexpr int SL = 32 * 128;                 // one slice: 32 pixels x 64 channels
    constexpr int STAGE = (CT + 4) * SL;         // [dy slices (shared)] [x slice of wave 0..3]
    constexpr int LPS = 4 + CT;                  // DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(1024))) unsigned char wg_lds[];

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.x * (64 * CT);
    const int ci_tiles = (g.Cin + 63) / 64;
    const int item = blockIdx.y * 4 + wid;
    const bool active = item < g.ntaps * ci_tiles;
    const int tap = active ? item / ci_tiles : 0;
    const int ci0 = active ? (item - tap * ci_tiles) * 64 : 0;
    int ty = 0, tx = 0;
#pragma unroll
    for (int t = 0; t < SBA_MAX_TAPS; ++t)
        if (t == tap) { ty = g.ty[t]; tx = g.tx[t]; }
    const int IHL = g.ups ? 2 * g.IH : g.IH, IWL = g.ups ? 2 * g.IW : g.IW;
    const int sub = g.OHs * g.OWs;

    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wg_lds;
    const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.IH * g.IW * g.Cin * 2);
    const uint32_t dy_bytes = (uint32_t)((int64_t)g.N * g.OH * g.OW * g.Cout * 2);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);
    constexpr uint32_t OOB = 0xFFFFFFFFu;

    // lane l of a DMA instruction fills 16-byte position (l & 7) of row (l >> 3) of its 8-row block; that position
    // holds channel chunk (l & 7) ^ 4 * bit1(row)
    const int rsub = lane >> 3, cg = (lane & 7) ^ (((lane >> 4) & 1) << 2);
    const bool x_ok = active && ci0 + cg * 8 < g.Cin;
    const uint32_t x_coff = (uint32_t)(ci0 + cg * 8) * 2u;
    const int dslice = CT == 1 ? 0 : (wid >> 1);
    const bool d_ok = co0 + dslice * 64 + cg * 8 < g.Cout;
    const uint32_t d_coff = (uint32_t)(co0 + dslice * 64 + cg * 8) * 2u;
    const uint32_t x_pix = (uint32_t)g.Cin * 2u, d_pix = (uint32_t)g.Cout * 2u;

    const int total_chunks = (M + 31) / 32;
    const int chunk_lo = blockIdx.z * chunks_per_split;
    const int chunk_hi = min(chunk_lo + chunks_per_split, total_chunks);
    int g_ck = chunk_lo;

    auto issue = [&](const uint32_t dst) {
        const bool live = g_ck < chunk_hi;
        const int m0 = g_ck * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 8 * i + rsub;
            uint32_t off = OOB;
            if (live && x_ok && m < M) {
                const int n = (int)fdiv(m, dsub), rem = m - n * sub;
                const int oy = (int)fdiv(rem, dow), ox = rem - oy * g.OWs;
                int iy = oy * g.sy + ty, ix = ox * g.sx + tx;
                const bool ok = (iy >= 0) & (iy < IHL) & (ix >= 0) & (ix < IWL);
                if (g.ups) { iy >>= 1; ix >>= 1; }
                if (ok) off = (uint32_t)((n * g.IH + iy) * g.IW + ix) * x_pix + x_coff;
            }
            lds_dma16(xr, off, 0u, dst + (uint32_t)((CT + wid) * SL + i * 1024));
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int i = CT == 1 ? wid : 2 * (wid & 1) + j;
            const int m = m0 + 8 * i + rsub;
            uint32_t off = OOB;
            if (live && d_ok && m < M) {
                const int n = (int)fdiv(m, dsub), rem = m - n * sub;
                const int oy = (int)fdiv(rem, dow), ox = rem - oy * g.OWs;
                off = (uint32_t)((n * g.OH + oy * g.osy + g.ooy) * g.OW + ox * g.osx + g.oox) * d_pix + d_coff;
            }
            lds_dma16(dr, off, 0u, dst + (uint32_t)(dslice * SL + i * 1024));
        }
        ++g_ck;
    };

    f32x16_t acc[CT][2][2];
#pragma unroll
    for (int s = 0; s < CT; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][i][j][r] = 0.f;

    int islot = 0;
#pragma unroll
    for (int p = 0; p < D - 1; ++p) { issue(lds_base + (uint32_t)(islot * STAGE)); ++islot; }
    if (islot == D) islot = 0;
    int cslot = 0;
    for (int ck = chunk_lo; ck < chunk_hi; ++ck) {
        wait_vmcnt<(D - 2) * LPS>();         // this wave's part of stage ck has landed ...
        wg_barrier();                        // ... and everybody else's; nobody reads slot (ck - 1) % D any more
        issue(lds_base + (uint32_t)(islot * STAGE));
        if (++islot == D) islot = 0;
        const unsigned char* st = wg_lds + cslot * STAGE;
        const unsigned char* sb = st + (CT + wid) * SL;
#pragma unroll
        for (int k16 = 0; k16 < 2; ++k16) {
            bf16x8_t b[2];
            b[0] = WgFragDma::load(sb, k16, 0, lane);
            b[1] = WgFragDma::load(sb, k16, 32, lane);
#pragma unroll
            for (int s = 0; s < CT; ++s) {
                bf16x8_t a[2];
                a[0] = WgFragDma::load(st + s * SL, k16, 0, lane);
                a[1] = WgFragDma::load(st + s * SL, k16, 32, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[s][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[s][i][j], 0, 0, 0);
            }
        }
        if (++cslot == D) cslot = 0;
    }
    wait_vmcnt<0>();            // the dead stages issued past the end still write (zeros) into the ring
    wg_barrier();

    if (!active) return;
    const int col_l = lane & 31, rsel = 4 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < CT; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + s * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                    const int ci = ci0 + j * 32 + col_l;
                    if (co < g.Cout && ci < g.Cin) {
                        float* p = dw + (int64_t)blockIdx.z * zstride + ((int64_t)co * g.ntaps + tap) * g.Cin + ci;
                        if (use_atomic == 1) atomicAdd(p, acc[s][i][j][r]);
                        else if (use_atomic == 2) *p = acc[s][i][j][r];       // first write of a cleared gradient
                        else *p += acc[s][i][j][r];
                    }
                }
}

// ---------------------------------------------------------------------------
// Kernel-row decomposition: the workgroup's waves = the KW taps of ONE kernel row kh and one 64-channel input tile.
// wgrad_small_dma_kernel gathers a 32-pixel x slice per wave (tap): 32 KW pixel rows per stage.  The kw taps of a kernel
// row read the SAME input row(s) at column offsets 0..KW-1 of a stride-SX walk, so here the stage holds that input row
// segment once -- R output rows x Wc output columns = 32 pixels (Wc = min(OW, 32)) need R x (SX (Wc - 1) + KW) input
// pixels -- and wave kw reads pixel (r, j) of it at LDS row r XW + SX j + kw.
//   <4, 2>: the 4x4 / stride-2 down blocks (encode_image_by_16times and the extra down blocks, model.py:540-575): <= 72
//           input pixels instead of 128: 4 KB (dy) + 9 KB per stage instead of 20 KB, four DMA instructions per wave
//           instead of five, 1.6x fewer bytes L2 -> LDS per MFMA (the roofline of this family);
//   <3, 1>: 3x3 stride-1 convs (the generator's 64 x 64 maps): <= 40 input pixels instead of 96, three waves.
// DMA blocks (8 rows x 128 B) of a stage: 0..3 = the dy slice, 4.. = the x rows, dealt round-robin to the waves.
// Stride 2: channel chunk c of LDS row L sits at 16-byte position c ^ 2 ((L >> 1) & 3) -- the four stride-2 rows a 16-lane
// group reads with ds_read_b64_tr_b16 fall into four different 32-byte bank groups; stride 1: c ^ 4 bit1(L) as above.
// ---------------------------------------------------------------------------
template <int SX>
struct WgFragRow {
    // fragment of channels [c32, c32+32) over output pixels [16 k16, 16 k16 + 16) of the chunk, tap column kw
    static __device__ __forceinline__ int swz(int L) { return SX == 2 ? (((L >> 1) & 3) << 1) : (((L >> 1) & 1) << 2); }
    // Stride 2: a wave's rows all have the parity of kw, and a 128-byte row covers half the banks -- every read would use 32 of
    // the 64 banks (PMC: SQ_LDS_BANK_CONFLICT = a third of the LDS cycles).  Segment row L therefore lives in LDS row
    // L ^ bit1(L) (an involution; swz() does not see bit 0): rows L and L + 2 fall into different halves.
    static __device__ __forceinline__ int slot(int L) { return SX == 2 ? (L ^ ((L >> 1) & 1)) : L; }
    static __device__ __forceinline__ bf16x8_t load(const unsigned char* xs, int k16, int c32, int lane, int kw, int wclog,
                                                    int xw, int ups, int tx0) {
        const int g16 = lane >> 4, i16 = lane & 15;
        const int cbase = c32 + 16 * (g16 & 1), p = 16 * k16 + 8 * (g16 >> 1) + (i16 >> 2), q = i16 & 3;
        const int r = p >> wclog, j = p & ((1 << wclog) - 1);
        // (behind a nearest x2 upsample the segment holds LOW-resolution pixels: column (ox0 + j + kw + tx0) >> 1, ox0 even)
        const int col = ups ? ((j + kw + tx0) >> 1) - (tx0 >> 1) : SX * j + kw;
        // pixel p + 4: four columns on in the same output row (Wc >= 8), or the next output row (Wc = 4)
        const int L0 = r * xw + col, L1 = L0 + (wclog == 2 ? xw : (ups ? 2 : 4 * SX));
        const int c0 = (cbase >> 3) ^ swz(L0), c1 = (cbase >> 3) ^ swz(L1);
        typedef __attribute__((address_space(3))) s16x4_t* lptr;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xs + slot(L0) * 128 + c0 * 16 + q * 8));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xs + slot(L1) * 128 + c1 * 16 + q * 8));
        bf16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    }
};

// KR = 1: one kernel row per workgroup (its KW waves); KR = KH: ALL kernel rows (KH x KW waves, the segment of every row
// staged side by side): the dy slice is then shared by all taps -- for layers whose dW is small enough that the extra
// pixel splits (fewer workgroups per split) cost nothing: the generator's 3x3 convs (dW <= 0.3 MB).
template <int KW, int XB, int KR = 1> struct WgRowCfg {         // XB: 8-row DMA blocks of one x segment
    static constexpr int NW = KW * KR;
    static constexpr int LPS = (4 + KR * XB + NW - 1) / NW;      // DMA instructions per wave per stage
    static constexpr int STAGE = LPS * NW * 1024;
};

template <int KW, int SX, int XB, int D, int KR = 1>
__global__ __launch_bounds__(64 * KW * KR, KR == 1 ? 2 : 1) void wgrad_row_dma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                float* __restrict__ dw, const sba_conv_geom g, const int M,
                                                                const int chunks_per_split, const int use_atomic,
                                                                const FastDiv dsub, const FastDiv dow, const int64_t zstride,
                                                                const int wclog) {
    typedef WgRowCfg<KW, XB, KR> Cfg;
    constexpr int LPS = Cfg::LPS, STAGE = Cfg::STAGE, NW = Cfg::NW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char wg_lds[];

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khl = wv / KW, kw = wv - khl * KW;            // this wave's kernel row inside the workgroup, its tap column
    const int co0 = blockIdx.x * 64;
    const int ci_tiles = g.Cin / 64;
    const int kh0 = KR == 1 ? blockIdx.y / ci_tiles : 0;    // first kernel row of the workgroup
    const int ci0 = (KR == 1 ? blockIdx.y - kh0 * ci_tiles : blockIdx.y) * 64;
    const int kh = kh0 + khl;
    const int tap = kh * KW + kw;
    // taps are row-structured (checked by the host): row k starts at (ty, tx0) = (g.ty[k KW], g.tx[k KW])
    auto row_ty = [&](const int k) { int v = 0;
#pragma unroll
        for (int t = 0; t < SBA_MAX_TAPS; ++t) if (t == k * KW) v = g.ty[t];
        return v; };
    int tx0 = 0;
#pragma unroll
    for (int t = 0; t < SBA_MAX_TAPS; ++t) {
        if (t == kh * KW) tx0 = g.tx[t];
    }
    const int ups = g.ups;            // (KW = 3, SX = 1 only) x is the LOW-resolution input of a nearest x2 upsample
    const int Wc = 1 << wclog, R = 32 >> wclog, XW = ups ? (Wc >> 1) + 2 : SX * (Wc - 1) + KW, XR = R * XW;
    const int sub = g.OH * g.OW;

    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wg_lds;
    const uint32_t x_bytes = (uint32_t)((int64_t)g.N * g.IH * g.IW * g.Cin * 2);
    const uint32_t dy_bytes = (uint32_t)((int64_t)g.N * g.OH * g.OW * g.Cout * 2);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);
    constexpr uint32_t OOB = 0xFFFFFFFFu;

    // lane l of a DMA instruction fills 16-byte position (l & 7) of row (l >> 3) of its 8-row block
    const int rsub = lane >> 3;
    const int cgx = (lane & 7) ^ WgFragRow<SX>::swz(rsub);              // x rows (8-row blocks: swz(L) = swz(L & 7))
    const int cgd = (lane & 7) ^ (((lane >> 4) & 1) << 2);              // dy rows: as in wgrad_small_dma_kernel
    const uint32_t x_coff = (uint32_t)(ci0 + cgx * 8) * 2u, d_coff = (uint32_t)(co0 + cgd * 8) * 2u;
    const uint32_t x_pix = (uint32_t)g.Cin * 2u, d_pix = (uint32_t)g.Cout * 2u;
    // this wave's blocks b = kw + KW i.  b < 4: rows 8 b .. of the dy slice (role 0: rr = pixel of the chunk);
    // 4 <= b < 4 + XB: rows of the x segment (role 1: rr / rc = input row / column relative to the chunk's first pixel);
    // else a dummy that zero-fills its block (role 2)
    // (a 4 x 4 map has 16 pixels: a chunk then spans rows_img = OH rows of each of 32 / 16 = 2 images)
    const int rows_img = g.OH < R ? g.OH : R;
    int role[LPS], rr[LPS], rc[LPS], rn[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int b = wv + NW * i;
        role[i] = 2; rr[i] = 0; rc[i] = 0; rn[i] = 0;
        if (b < 4) { role[i] = 0; rr[i] = 8 * b + rsub; }
        else {
            const int sg = (b - 4) / XB;                                 // which kernel row's segment
            const int L = WgFragRow<SX>::slot(8 * (b - 4 - sg * XB) + rsub);  // the segment row this LDS row holds
            if (sg < KR && L < XR) {
                const int r = L / XW;
                role[i] = 1;
                rn[i] = r / rows_img;
                rr[i] = (r - rn[i] * rows_img) * g.sy + row_ty(kh0 + sg); // (ups: an offset in UPSAMPLED rows)
                rc[i] = ups ? L - r * XW : L - r * XW + tx0;            // (ups: the segment's low-resolution column index)
            }
        }
    }

    const int chunk_lo = blockIdx.z * chunks_per_split;
    const int chunk_hi = min(chunk_lo + chunks_per_split, M >> 5);
    int g_ck = chunk_lo;

    auto issue = [&](const uint32_t dst) {
        const bool live = g_ck < chunk_hi;
        const int m0 = g_ck * 32;
        const int n = (int)fdiv(m0, dsub), rem = m0 - n * sub;
        const int oy0 = (int)fdiv(rem, dow), ox0 = rem - oy0 * g.OW;
        const int iy0 = oy0 * g.sy, ix0 = ox0 * g.sx;
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            uint32_t off = OOB;
            if (role[i] == 0) {
                if (live) off = (uint32_t)(m0 + rr[i]) * d_pix + d_coff;
                lds_dma16(dr, off, 0u, dst + (uint32_t)((wv + NW * i) * 1024));
            } else {
                int iy = iy0 + rr[i], ix = ix0 + rc[i];
                if (ups) {              // upsampled row v -> low-resolution row v >> 1 (v = -1 and v = 2 IH are the padding)
                    iy >>= 1;
                    ix = ((ox0 + tx0) >> 1) + rc[i];
                }
                const bool ok = live & (role[i] == 1) & (iy >= 0) & (iy < g.IH) & (ix >= 0) & (ix < g.IW);
                if (ok) off = (uint32_t)(((n + rn[i]) * g.IH + iy) * g.IW + ix) * x_pix + x_coff;
                lds_dma16(xr, off, 0u, dst + (uint32_t)((wv + NW * i) * 1024));
            }
        }
        ++g_ck;
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int islot = 0;
#pragma unroll
    for (int p = 0; p < D - 1; ++p) { issue(lds_base + (uint32_t)(islot * STAGE)); ++islot; }
    if (islot == D) islot = 0;
    int cslot = 0;
    for (int ck = chunk_lo; ck < chunk_hi; ++ck) {
        wait_vmcnt<(D - 2) * LPS>();         // this wave's part of stage ck has landed ...
        wg_barrier();                        // ... and everybody else's; nobody reads slot (ck - 1) % D any more
        issue(lds_base + (uint32_t)(islot * STAGE));
        if (++islot == D) islot = 0;
        const unsigned char* st = wg_lds + cslot * STAGE;
        const unsigned char* xs = st + (4 + khl * XB) * 1024;
#pragma unroll
        for (int k16 = 0; k16 < 2; ++k16) {
            bf16x8_t a[2], b[2];
            b[0] = WgFragRow<SX>::load(xs, k16, 0, lane, kw, wclog, XW, ups, tx0);
            b[1] = WgFragRow<SX>::load(xs, k16, 32, lane, kw, wclog, XW, ups, tx0);
            a[0] = WgFragDma::load(st, k16, 0, lane);
            a[1] = WgFragDma::load(st, k16, 32, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (++cslot == D) cslot = 0;
    }
    wait_vmcnt<0>();            // the dead stages issued past the end still write (zeros) into the ring
    wg_barrier();

    const int col_l = lane & 31, rsel = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                const int ci = ci0 + j * 32 + col_l;
                float* p = dw + (int64_t)blockIdx.z * zstride + ((int64_t)co * g.ntaps + tap) * g.Cin + ci;
                if (use_atomic == 1) atomicAdd(p, acc[i][j][r]);
                else if (use_atomic == 2) *p = acc[i][j][r];       // first write of a cleared gradient
                else *p += acc[i][j][r];
            }
}

// ---------------------------------------------------------------------------
// weight gradient, large-map 3x3 stride-1 convs (optionally over a nearest-x2 upsampled input)
// with OW % 64 == 0: the generator's 64..256 px layers, where ~all wgrad FLOPs are.
// A workgroup owns one 64(co) x 64(ci) tile for ALL nine taps: per 64-pixel segment of an output
// row it stages the dy tile once and, per kernel row kh, ONE input row with a 1-pixel halo
// (66 x 64 channels); the three kw taps are the same LDS rows read at a +kw row offset, so each
// input pixel is fetched once per kh instead of once per tap.  Wave kh (3 waves) accumulates its
// three taps (192 accumulator registers) over the workgroup's whole pixel range, then adds them
// to dw with f32 atomics.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(192) void wgrad_rows_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         float* __restrict__ dw, const sba_conv_geom g,
                                                         const int total_segs, const int segs_per_wg,
                                                         const int store, const int64_t zstride) {
    constexpr int ROWS = WgFrag<T>::ROWS;
    constexpr int CH = 16 / (int)sizeof(T);
    constexpr int CPR = 64 / CH;                       // 16-byte chunks per 64-channel pixel row
    constexpr int XR = 66;                             // 64 pixels + halo
    constexpr int XROWS_ALLOC = 80;                    // rows reserved per x tile (>= 64 + 2 + 15 read slack)
    __shared__ __attribute__((aligned(16))) unsigned char lds[(64 + 3 * XROWS_ALLOC) * ROWS];

    const int tid = threadIdx.x, lane = tid & 63, kh = tid >> 6;
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64;
    const int IHL = g.ups ? 2 * g.IH : g.IH, IWL = g.ups ? 2 * g.IW : g.IW;
    const int segs_per_row = g.OW / 64;
    unsigned char* sa = lds;
    unsigned char* sx = lds + (64 + kh * XROWS_ALLOC) * ROWS;

    // rows 66..79 of the x tile are only ever read by discarded k positions? no: every read row
    // index is < 16*3 + 2 + 16 = 66, so the slack rows are never touched; zero them once anyway
    for (int i = lane; i < (XROWS_ALLOC - XR) * ROWS / 16; i += 64)
        *reinterpret_cast<uint4*>(sx + XR * ROWS + i * 16) = make_uint4(0, 0, 0, 0);

    f32x16_t acc[3][2][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    const int seg_lo = blockIdx.z * segs_per_wg;
    const int seg_hi = min(seg_lo + segs_per_wg, total_segs);
    constexpr int A_PT = (64 * CPR + 191) / 192;
    constexpr int X_PT = (XR * CPR + 63) / 64;
    uint4 va[A_PT], vx[X_PT];
    auto gload = [&](int seg) {
        const int n = seg / (g.OH * segs_per_row);
        const int rem = seg - n * g.OH * segs_per_row;
        const int oy = rem / segs_per_row, ox0 = (rem - oy * segs_per_row) * 64;
#pragma unroll
        for (int u = 0; u < A_PT; ++u) {
            const int idx = tid + 192 * u;
            va[u] = make_uint4(0, 0, 0, 0);
            if (idx < 64 * CPR) {
                const int pix = idx / CPR, cc = idx - pix * CPR;
                const int co = co0 + cc * CH;
                if (co < g.Cout) {
                    const int64_t po = ((int64_t)(n * g.OH + oy) * g.OW + ox0 + pix);
                    va[u] = *reinterpret_cast<const uint4*>(dy + po * g.Cout + co);
                }
            }
        }
        int iy = oy + kh - 1;
        const bool row_ok = (iy >= 0) & (iy < IHL);
        if (g.ups) iy >>= 1;
#pragma unroll
        for (int u = 0; u < X_PT; ++u) {
            const int idx = lane + 64 * u;
            vx[u] = make_uint4(0, 0, 0, 0);
            if (idx < XR * CPR) {
                const int j = idx / CPR, cc = idx - j * CPR;
                int ix = ox0 - 1 + j;
                const bool ok = row_ok & (ix >= 0) & (ix < IWL);
                if (g.ups) ix >>= 1;
                const int ci = ci0 + cc * CH;
                if (ok && ci < g.Cin) {
                    const int64_t pi = (int64_t)(n * g.IH + iy) * g.IW + ix;
                    vx[u] = *reinterpret_cast<const uint4*>(x + pi * g.Cin + ci);
                }
            }
        }
    };
    if (seg_lo < seg_hi) gload(seg_lo);
    for (int seg = seg_lo; seg < seg_hi; ++seg) {
        __syncthreads();        // previous segment's tiles fully consumed
#pragma unroll
        for (int u = 0; u < A_PT; ++u) {
            const int idx = tid + 192 * u;
            if (idx < 64 * CPR) {
                const int pix = idx / CPR, cc = idx - pix * CPR;
                *reinterpret_cast<uint4*>(sa + pix * ROWS + cc * 16) = va[u];
            }
        }
#pragma unroll
        for (int u = 0; u < X_PT; ++u) {
            const int idx = lane + 64 * u;
            if (idx < XR * CPR) {
                const int j = idx / CPR, cc = idx - j * CPR;
                *reinterpret_cast<uint4*>(sx + j * ROWS + cc * 16) = vx[u];
            }
        }
        __syncthreads();
        if (seg + 1 < seg_hi) gload(seg + 1);     // prefetch the next segment under the 48 MFMAs below
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                WgFrag<T>::mma(sa + 16 * ks * ROWS, sx + (16 * ks + kw) * ROWS, lane, acc[kw]);
        }
    }

    const int col_l = lane & 31, rsel = 4 * (lane >> 5);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int tap = kh * 3 + kw;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                    const int ci = ci0 + j * 32 + col_l;
                    if (co < g.Cout && ci < g.Cin) {
                        float* p = dw + (int64_t)blockIdx.z * zstride + ((int64_t)co * 9 + tap) * g.Cin + ci;
                        if (store) *p = acc[kw][i][j][r];       // deterministic mode: this split's own partial tensor
                        else atomicAdd(p, acc[kw][i][j][r]);
                    }
                }
    }
}

// ---------------------------------------------------------------------------
// weight packing and 2x2 sum pooling
// ---------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int KH,
                                   int KW, int Cin, int mode) {
    const int64_t n = (int64_t)Cout * (mode == 3 ? 16 : KH * KW) * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT (so that writes are coalesced)
        float v;
        if (mode == 0) {
            v = w[i];
        } else if (mode == 1) {
            // out[ci][kh'][kw'][co] = w[co][KH-1-kh'][KW-1-kw'][ci]
            const int co = (int)(i % Cout);
            int64_t t = i / Cout;
            const int kw = (int)(t % KW); t /= KW;
            const int kh = (int)(t % KH);
            const int ci = (int)(t / KH);
            v = w[(((int64_t)co * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * Cin + ci];
        } else if (mode == 3) {
            // data-gradient of (nearest x2 -> conv3x3) as ONE 4x4 stride-2 pad-1 conv over dy:
            // out[ci][dd*4+ee][co] = sum_{kh in S(dd)} sum_{kw in S(ee)} w[co][kh][kw][ci],
            // S(0) = {2}, S(1) = {1,2}, S(2) = {0,1}, S(3) = {0}
            const int co = (int)(i % Cout);
            int64_t t = i / Cout;
            const int sl = (int)(t % 16);
            const int ci = (int)(t / 16);
            const int dd = sl >> 2, ee = sl & 3;
            const int kh0 = dd == 0 ? 2 : (dd == 1 ? 1 : 0), khn = (dd == 1 || dd == 2) ? 2 : 1;
            const int kw0 = ee == 0 ? 2 : (ee == 1 ? 1 : 0), kwn = (ee == 1 || ee == 2) ? 2 : 1;
            v = 0.f;
            for (int a = 0; a < khn; ++a)
                for (int b2 = 0; b2 < kwn; ++b2)
                    v += w[(((int64_t)co * 3 + kh0 + a) * 3 + kw0 + b2) * Cin + ci];
        } else {
            // out[cls][ci][j*2+i2][co], cls = py*2+px, kh = (1-py)+2j, kw = (1-px)+2*i2  (KH=KW=4)
            const int co = (int)(i % Cout);
            int64_t t = i / Cout;
            const int tp = (int)(t % 4); t /= 4;
            const int ci = (int)(t % Cin);
            const int cls = (int)(t / Cin);
            const int py = cls >> 1, px = cls & 1, j = tp >> 1, i2 = tp & 1;
            const int kh = (1 - py) + 2 * j, kw = (1 - px) + 2 * i2;
            v = w[(((int64_t)co * 4 + kh) * 4 + kw) * Cin + ci];
        }
        out[i] = from_f<T>(v);
    }
}

// modes 1 / 2 as tiled transposes: for one (source tap -> destination tap) pair the job is
// out[ci][co] = w[co][ci] with row strides of taps*Cin resp. dtaps*Cout; 32x32 tiles through LDS
// keep both the global reads (along ci) and writes (along co) coalesced.
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_tr_kernel(const float* __restrict__ w, T* __restrict__ out,
                                                             int Cout, int KH, int KW, int Cin, int mode) {
    __shared__ float tile[32][33];
    const int z = blockIdx.z;
    int src_tap, dst_tap, dtaps;
    int64_t dst_base = 0;
    if (mode == 1) {
        const int kh = z / KW, kw = z - kh * KW;
        dst_tap = z;
        src_tap = (KH - 1 - kh) * KW + (KW - 1 - kw);
        dtaps = KH * KW;
    } else {
        const int cls = z >> 2, tp = z & 3;
        const int py = cls >> 1, px = cls & 1, j = tp >> 1, i2 = tp & 1;
        src_tap = ((1 - py) + 2 * j) * 4 + (1 - px) + 2 * i2;
        dst_tap = tp;
        dtaps = 4;
        dst_base = (int64_t)cls * Cin * 4 * Cout;
    }
    const int taps = KH * KW;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((int64_t)co * taps + src_tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) out[dst_base + ((int64_t)ci * dtaps + dst_tap) * Cout + co] = from_f<T>(tile[tx][r]);
    }
}

// every packed copy of one network in one launch: a workgroup owns one 64(co) x 64(ci) tile of one
// source tap of one tensor, reads it once (float4 rows), writes the forward copy (same order) and
// the transposed data-gradient copy (64 co contiguous = full 128-byte lines) through LDS
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const sba_pack_desc* __restrict__ descs, int ndesc) {
    __shared__ float tile[64][65];
    int lo = 0, hi = ndesc - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                                   // last desc with tile_begin <= b (uniform)
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile_begin <= b) lo = mid; else hi = mid - 1;
    }
    const sba_pack_desc d = descs[lo];
    const int local = b - d.tile_begin;
    const int per_tap = d.co_tiles * d.ci_tiles;
    const int src_tap = local / per_tap;                 // mode 3: the destination tap slot (0..15)
    const int rem = local - src_tap * per_tap;
    const int co0 = (rem / d.ci_tiles) * 64, ci0 = (rem % d.ci_tiles) * 64;
    const int taps = d.KH * d.KW;
    const int slots = d.mode == 3 ? 16 : taps;
    if (src_tap >= slots) return;
    const int tid = threadIdx.x;
    const int c4 = (tid & 15) * 4;
    T* fwd = reinterpret_cast<T*>(d.fwd);
    // mode 3 (data-gradient of nearest x2 -> conv3x3 as one 4x4/s2 conv): slot (dd, ee) sums the source
    // taps kh in S(dd), kw in S(ee);  S(0) = {2}, S(1) = {1,2}, S(2) = {0,1}, S(3) = {0}
    int kh0 = 0, khn = 1, kw0 = 0, kwn = 1;
    if (d.mode == 3) {
        const int dd = src_tap >> 2, ee = src_tap & 3;
        kh0 = dd == 0 ? 2 : (dd == 1 ? 1 : 0); khn = (dd == 1 || dd == 2) ? 2 : 1;
        kw0 = ee == 0 ? 2 : (ee == 1 ? 1 : 0); kwn = (ee == 1 || ee == 2) ? 2 : 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 4) + 16 * i;
        const int co = co0 + r, ci = ci0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (co < d.Cout && ci < d.Cin) {
            if (d.mode != 3 || src_tap < taps) {           // forward copy of source tap `src_tap`
                const int64_t o = ((int64_t)co * taps + src_tap) * d.Cin + ci;
                const float4 f = *reinterpret_cast<const float4*>(d.w + o);
                if (fwd) {
                    fwd[o] = from_f<T>(f.x); fwd[o + 1] = from_f<T>(f.y);
                    fwd[o + 2] = from_f<T>(f.z); fwd[o + 3] = from_f<T>(f.w);
                }
                if (d.mode != 3) v = f;
            }
            if (d.mode == 3) {
                for (int a = 0; a < khn; ++a)
                    for (int b2 = 0; b2 < kwn; ++b2) {
                        const float4 f = *reinterpret_cast<const float4*>(
                            d.w + ((int64_t)co * 9 + (kh0 + a) * 3 + kw0 + b2) * d.Cin + ci);
                        v.x += f.x; v.y += f.y; v.z += f.z; v.w += f.w;
                    }
            }
        }
        tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    }
    if (!d.tr) return;
    __syncthreads();
    int dst_tap, dtaps;
    int64_t dst_base = 0;
    if (d.mode == 1) {
        const int kh = src_tap / d.KW, kw = src_tap - kh * d.KW;
        dst_tap = (d.KH - 1 - kh) * d.KW + (d.KW - 1 - kw);
        dtaps = taps;
    } else if (d.mode == 3) {
        dst_tap = src_tap;
        dtaps = 16;
    } else {
        const int kh = src_tap >> 2, kw = src_tap & 3;      // kh = (1-py) + 2j, kw = (1-px) + 2i
        const int py = 1 - (kh & 1), px = 1 - (kw & 1);
        dst_tap = (kh >> 1) * 2 + (kw >> 1);
        dtaps = 4;
        dst_base = (int64_t)(py * 2 + px) * d.Cin * 4 * d.Cout;
    }
    T* tr = reinterpret_cast<T*>(d.tr);
    const int c8 = (tid & 7) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 3) + 32 * i;                  // ci within the tile
        const int ci = ci0 + r, co = co0 + c8;
        if (ci >= d.Cin || co >= d.Cout) continue;
        T* op = tr + dst_base + ((int64_t)ci * dtaps + dst_tap) * d.Cout + co;
        if (co + 8 <= d.Cout && sizeof(T) == 2) {
            uint4 o;
            o.x = (uint32_t)f2bf(tile[c8 + 0][r]) | ((uint32_t)f2bf(tile[c8 + 1][r]) << 16);
            o.y = (uint32_t)f2bf(tile[c8 + 2][r]) | ((uint32_t)f2bf(tile[c8 + 3][r]) << 16);
            o.z = (uint32_t)f2bf(tile[c8 + 4][r]) | ((uint32_t)f2bf(tile[c8 + 5][r]) << 16);
            o.w = (uint32_t)f2bf(tile[c8 + 6][r]) | ((uint32_t)f2bf(tile[c8 + 7][r]) << 16);
            *reinterpret_cast<uint4*>(op) = o;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (co + k < d.Cout) op[k] = from_f<T>(tile[c8 + k][r]);
        }
    }
}

template <typename T>
__global__ void pool2x2_kernel(const T* __restrict__ up, T* __restrict__ dx, int N, int H, int W, int C) {
    constexpr int V = Vec16<T>::N;
    const int cv = C / V;
    const int64_t total = (int64_t)N * H * W * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        int64_t p = i / cv;
        const int xx = (int)(p % W); p /= W;
        const int yy = (int)(p % H);
        const int n = (int)(p / H);
        const T* b = up + (((int64_t)n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + c;
        Vec16<T> v00 = ld16(b), v01 = ld16(b + C), v10 = ld16(b + (int64_t)2 * W * C),
                 v11 = ld16(b + (int64_t)2 * W * C + C), o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.set(k, v00.get(k) + v01.get(k) + v10.get(k) + v11.get(k));
        st16(dx + (((int64_t)n * H + yy) * W + xx) * C + c, o);
    }
}

// split-K finish: y[pix(m)][co] = ws[m][co] (+ addend), per-channel stats; thread = 4 channels x 8 rows
template <typename T>
__device__ __forceinline__ void splitk_finish_body(float* __restrict__ ws, T* __restrict__ y,
                                                   const T* __restrict__ addend,
                                                   float* __restrict__ stats, const sba_conv_geom& g,
                                                   const int M, const EpiX ex) {
    const int ycs = g.y_cstride ? g.y_cstride : g.Cout;
    const int cq = g.Cout / 4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cq) return;
    const int c = t * 4;
    const int m0 = blockIdx.y * 8, m1 = min(m0 + 8, M);
    if (m0 >= M) return;
    const int sub = g.OHs * g.OWs;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    float4 rows[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {                       // all loads in flight before any use
        const int m = m0 + r;
        float4* wp = reinterpret_cast<float4*>(ws + (int64_t)(m < m1 ? m : m0) * g.Cout + c);
        rows[r] = *wp;
        if (m < m1) *wp = make_float4(0.f, 0.f, 0.f, 0.f);      // leave the workspace zero-filled for its next user
    }
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (ex.bias) {
#pragma unroll
        for (int k = 0; k < 4; ++k) bv[k] = ex.bias[c + k];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int m = m0 + r;
        if (m >= m1) break;
        float v[4] = {rows[r].x, rows[r].y, rows[r].z, rows[r].w};
        const int n = m / sub, rem = m - n * sub;
        const int oy = rem / g.OWs, ox = rem - oy * g.OWs;
        const int64_t o = ((int64_t)(n * g.OH + oy * g.osy + g.ooy) * g.OW + ox * g.osx + g.oox) * ycs + g.y_coff + c;
        // 4 consecutive channels: one 8-byte (bf16) / 16-byte (f32) access for y, addend and the ReLU mask
        typedef typename std::conditional<sizeof(T) == 2, uint2, uint4>::type V4;
        T av[4], mv[4], ov[4];
        if (addend) *reinterpret_cast<V4*>(av) = *reinterpret_cast<const V4*>(addend + o);
        if (ex.mask) *reinterpret_cast<V4*>(mv) = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(ex.mask) + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s0[k] += v[k];
            s1[k] += v[k] * v[k];
            v[k] += bv[k];
            if (g.relu) v[k] = fmaxf(v[k], 0.f);
            if (addend) v[k] += to_f<T>(av[k]);
            if (ex.mask && !(to_f<T>(mv[k]) > 0.f)) v[k] = 0.f;
            if (sizeof(T) == 2 && ex.yh) *reinterpret_cast<bf16_t*>(&ov[k]) = f2h_bits(v[k]);
            else ov[k] = from_f<T>(v[k]);
        }
        *reinterpret_cast<V4*>(y + o) = *reinterpret_cast<const V4*>(ov);
    }
    if (stats) {
        float* slot = stats + (int64_t)(blockIdx.y & (SBA_BN_STAT_SLOTS - 1)) * 2 * g.Cout;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&slot[c + k], s0[k]);
            atomicAdd(&slot[g.Cout + c + k], s1[k]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(float* __restrict__ ws, T* __restrict__ y,
                                                            const T* __restrict__ addend,
                                                            float* __restrict__ stats, const sba_conv_geom g,
                                                            const int M, const EpiX ex) {
    splitk_finish_body<T>(ws, y, addend, stats, g, M, ex);
}

// the finishing pass of a grouped split-K launch: blockIdx.z = item
__global__ __launch_bounds__(256) void splitk_finish_group_kernel(const GroupArgs A) {
    const GroupItem& it = A.it[blockIdx.z];
    const sba_conv_geom g = it.g;
    splitk_finish_body<bf16_t>(it.ws, it.y, it.addend, nullptr, g, it.M, EpiX{it.bias, it.mask, 0});
}

// ---- tile configurations and their selection --------------------------------------------
struct IgemmCfg { int bm, bn, ks, occ; float eff; bool split; };
// A: big square tile, B: wide-M tile for Cout = 64, C: mid tile, D: small tile (+split-K),
// E: skinny GEMM tile for the 4x4 / 8x8 maps with thousands of channels (+split-K)
// (The kernel also instantiates with KG > 1 in-workgroup K-groups and PF = 3 stages of loads in flight;
// on the latency-bound small layers neither beat D -- 1 KB of LDS fragments per MFMA and one wave per
// SIMD bound them, not the K loop -- so no shipped configuration uses them.  A 256x128 tile with 128x64
// per-wave tiles (less LDS traffic per MFMA, but 4 waves and one workgroup per CU) lost to E everywhere.)
static const IgemmCfg kCfg[5] = {
    {128, 128, 1, 3, 1.00f, false}, {256, 64, 1, 3, 1.00f, false}, {128, 64, 2, 2, 0.80f, false},
    {64, 64, 2, 4, 0.50f, true},    {320, 128, 2, 1, 0.90f, true}};

template <typename T, int BM, int BN, int WM, int WN, int KS, int KG = 1, int PF = 1>
static void launch_cfg(const T* xp, const T* wp, T* yp, const T* ap, float* stats, const sba_conv_geom& g, int M,
                       int nslabs, int split, float* ws, hipStream_t st, const EpiX ex) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64 * KG;
    int sps = nslabs;
    if (split > 1) {
        sps = cdiv(cdiv(nslabs, split), KS) * KS;
        split = cdiv(nslabs, sps);
    }
    dim3 grid(cdiv(M, BM), cdiv(g.Cout, BN), split);
    SBA_LAUNCH((igemm_kernel<T, BM, BN, WM, WN, KS, KG, PF>), grid, dim3(NT), 0, st, xp, wp, yp, ap, stats, g, M,
                       split > 1 ? ws : (float*)nullptr, sps, ex);
    if (split > 1) {
        dim3 fgrid(cdiv(g.Cout / 4, 256), cdiv(M, 8));
        SBA_LAUNCH((splitk_finish_kernel<T>), fgrid, dim3(256), 0, st, ws, yp, ap, stats, g, M, ex);
    }
}

// SBA_SPLITK_FUSED=1: the last split to arrive finishes its tile inside the GEMM kernel (splitk_arrive); the tickets
// live in the last SPLITK_TICKET_BYTES of the (zero-filled, left zero-filled) workspace.  Default 0 = separate
// splitk_finish_kernel launch: the fused form passes the whole GPU suite but measured no gain on the step (14.96 vs
// 14.78 ms, 75 finishing launches fewer): the last arrival's 16 KB device-scope reload + epilogue sits at the tail of
// every tile, where the finishing launch spreads the same work over the whole chip.
static int* splitk_tickets(float* ws, int64_t ws_bytes, int M, int Cout, int tiles) {
    static int fused = -1;
    if (fused < 0) { const char* e = getenv("SBA_SPLITK_FUSED"); fused = (e && e[0] == '1') ? 1 : 0; }
    if (!fused || !ws || ws_bytes < 2 * SPLITK_TICKET_BYTES) return nullptr;
    if ((int64_t)M * Cout * 4 > ws_bytes - SPLITK_TICKET_BYTES || tiles > SPLITK_TICKET_BYTES / 4) return nullptr;
    return reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + ws_bytes - SPLITK_TICKET_BYTES);
}

// N-major tile numbering (see igemm_dma2_body) when the weights outweigh the input tensor.  SBA_IGEMM_NMAJOR=0 disables it.
static int nmajor_for(const sba_conv_geom& g) {
    static int en = -1;
    if (en < 0) { const char* e = getenv("SBA_IGEMM_NMAJOR"); en = (e && e[0] == '0') ? 0 : 1; }
    const int64_t wb = (int64_t)g.Cout * g.ntaps * g.Cin, xb = (int64_t)g.N * g.IH * g.IW * g.Cin;
    return (en && wb > xb) ? 1 : 0;
}

template <int BM, int BN, int WM, int WN, int KS, int D>
static void launch_dma(const bf16_t* xp, const bf16_t* wp, bf16_t* yp, const bf16_t* ap, float* stats,
                       const sba_conv_geom& g, int M, int nslabs, int split, float* ws, int64_t ws_bytes, hipStream_t st,
                       const EpiX ex) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    int sps = nslabs;
    if (split > 1) {
        sps = cdiv(cdiv(nslabs, split), KS) * KS;
        split = cdiv(nslabs, sps);
    }
    const int gx = cdiv(M, BM), gy = cdiv(g.Cout, BN);
    const int nmajor = nmajor_for(g);
    dim3 grid(nmajor ? 8 * cdiv(gy, 8) * gx : 8 * cdiv(gx, 8) * gy, 1, split);
    int* tickets = splitk_tickets(ws, ws_bytes, M, g.Cout, (int)grid.x);
    SBA_LAUNCH((igemm_dma_kernel<BM, BN, WM, WN, KS, D>), grid, dim3(NT), 0, st, xp, wp, yp, ap, stats, g, M,
               split > 1 ? ws : (float*)nullptr, split > 1 ? tickets : (int*)nullptr, sps, ex, gx, gy, nmajor DMA_TRACE_ARG);
    if (split > 1 && !tickets) {
        dim3 fgrid(cdiv(g.Cout / 4, 256), cdiv(M, 8));
        SBA_LAUNCH((splitk_finish_kernel<bf16_t>), fgrid, dim3(256), 0, st, ws, yp, ap, stats, g, M, ex);
    }
}

template <int BM, int BN, int WM, int WN, int D>
static void launch_dma2(const bf16_t* xp, const bf16_t* wp, bf16_t* yp, const bf16_t* ap, float* stats,
                        const sba_conv_geom& g, int M, int nslabs64, int split, float* ws, int64_t ws_bytes, hipStream_t st,
                        const EpiX ex) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    int sps = nslabs64;
    if (split > 1) {
        sps = cdiv(nslabs64, split);
        split = cdiv(nslabs64, sps);
    }
    const int gx = cdiv(M, BM), gy = cdiv(g.Cout, BN);
    const int nmajor = nmajor_for(g);
    dim3 grid(nmajor ? 8 * cdiv(gy, 8) * gx : 8 * cdiv(gx, 8) * gy, 1, split);
    int* tickets = splitk_tickets(ws, ws_bytes, M, g.Cout, (int)grid.x);
    SBA_LAUNCH((igemm_dma2_kernel<BM, BN, WM, WN, D>), grid, dim3(NT), 0, st, xp, wp, yp, ap, stats, g, M,
               split > 1 ? ws : (float*)nullptr, split > 1 ? tickets : (int*)nullptr, sps, ex, gx, gy, nmajor DMA_TRACE_ARG);
    if (split > 1 && !tickets) {
        dim3 fgrid(cdiv(g.Cout / 4, 256), cdiv(M, 8));
        SBA_LAUNCH((splitk_finish_kernel<bf16_t>), fgrid, dim3(256), 0, st, ws, yp, ap, stats, g, M, ex);
    }
}

// SBA_IGEMM_DMA: 1 (default) = LDS-DMA staged kernels for the bf16 tiles A-D, 0 = register-staged kernels
static int dma_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SBA_IGEMM_DMA"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}

static int forced_cfg() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("SBA_IGEMM_CFG");       // tuning aid only: A..E
        v = (e && e[0] >= 'A' && e[0] <= 'E') ? e[0] - 'A' : -1;
    }
    return v;
}

// ---- halo-tile 3x3 path: which geometries qualify, and its launch
static bool halo_ok(const sba_conv_geom& g) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("SBA_CONV_HALO"); enabled = (e && e[0] == '0') ? 0 : 1; }
    if (!enabled) return false;
    if (g.ntaps != 9 || g.sy != 1 || g.sx != 1 || g.osy != 1 || g.osx != 1 || g.ooy || g.oox) return false;
    // Cin = 128 (data gradient of the ResBlocks' 64 -> 128 conv): the kernel walks the tile once per 64-channel chunk
    static int halo128 = -1;
    if (halo128 < 0) { const char* e = getenv("SBA_CONV_HALO128"); halo128 = (e && e[0] == '0') ? 0 : 1; }
    const bool cin_ok = g.Cin == 64 || (g.Cin == 128 && halo128);
    if (g.OHs != g.OH || g.OWs != g.OW || !cin_ok || g.Cout % 64) return false;
    if (g.OH % 8 || g.OW % 32) return false;
    if (g.ups ? (g.IH * 2 != g.OH || g.IW * 2 != g.OW) : (g.IH != g.OH || g.IW != g.OW)) return false;
    for (int t = 0; t < 9; ++t)
        if (g.ty[t] != t / 3 - 1 || g.tx[t] != t % 3 - 1) return false;
    const int64_t tiles = (int64_t)g.N * (g.OH / 8) * (g.OW / 32);
    return tiles >= 128 && tiles <= 0x7fffffff;
}

// the general register-weight halo kernel (fragment-major weights only): any stride-1 3 x 3 window, ragged maps
static bool halo3g_ok(const sba_conv_geom& g) {
    if (g.w_layout != 1 || g.ups) return false;
    if (g.ntaps != 9 || g.sy != 1 || g.sx != 1 || g.osy != 1 || g.osx != 1 || g.ooy || g.oox) return false;
    if (g.OHs != g.OH || g.OWs != g.OW || g.Cin % 32 || g.Cout % 32) return false;
    int ymin = g.ty[0], ymax = g.ty[0], xmin = g.tx[0], xmax = g.tx[0];
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        ymin = g.ty[t] < ymin ? g.ty[t] : ymin; ymax = g.ty[t] > ymax ? g.ty[t] : ymax;
        xmin = g.tx[t] < xmin ? g.tx[t] : xmin; xmax = g.tx[t] > xmax ? g.tx[t] : xmax;
    }
    if (ymax - ymin != 2 || xmax - xmin != 2) return false;
    for (int t = 0; t < 9; ++t) seen |= 1u << ((g.ty[t] - ymin) * 3 + (g.tx[t] - xmin));
    if (seen != 0x1ffu) return false;               // the nine taps are the nine cells of the window
    const int64_t tiles = (int64_t)g.N * cdiv(g.OH, 8) * cdiv(g.OW, 32);
    return tiles >= 128 && tiles <= 0x7fffffff;
}

static void launch_halo3g(const sba_conv_geom& g, const bf16_t* x, const bf16_t* w, bf16_t* y, const bf16_t* addend,
                          const EpiX ex, hipStream_t st) {
    const int tiles = g.N * cdiv(g.OH, 8) * cdiv(g.OW, 32);
    const bool tn2 = g.Cout % 64 == 0;
    dim3 grid(tiles, g.Cout / (tn2 ? 64 : 32));
    if (g.Cin % 64 == 0) {
        if (tn2) SBA_LAUNCH((conv3x3_halo3g_kernel<64, 2>), grid, dim3(256), 0, st, x, w, y, addend, g, ex);
        else SBA_LAUNCH((conv3x3_halo3g_kernel<64, 1>), grid, dim3(256), 0, st, x, w, y, addend, g, ex);
    } else {
        if (tn2) SBA_LAUNCH((conv3x3_halo3g_kernel<32, 2>), grid, dim3(256), 0, st, x, w, y, addend, g, ex);
        else SBA_LAUNCH((conv3x3_halo3g_kernel<32, 1>), grid, dim3(256), 0, st, x, w, y, addend, g, ex);
    }
}

static void launch_halo(const sba_conv_geom& g, const bf16_t* x, const bf16_t* w, bf16_t* y, const bf16_t* addend,
                        float* stats, const EpiX ex, hipStream_t st) {
    const int tiles = g.N * (g.OH / 8) * (g.OW / 32);
    // SBA_CONV_HALO2=1: the persistent variant (one workgroup per CU, weights resident, halo by LDS-DMA).  Measured
    // SLOWER than the per-tile kernel at three workgroups per CU (G3 upBlock 151 vs 113 us, ResBlock 86 vs 80 us):
    // with one wave per SIMD the epilogue of a tile overlaps nothing.  Kept as an experiment, off by default.
    static int v2 = -1;
    if (v2 < 0) { const char* e = getenv("SBA_CONV_HALO2"); v2 = (e && e[0] == '1') ? 1 : 0; }
    const int nblocks = g.Cout / 64;
    if (v2 && tiles >= 512 && g.Cin == 64) {
        // persistent workgroups, one per CU, split evenly over the 64-channel blocks of Cout
        int per = 256 / nblocks;
        if (per > tiles) per = tiles;
        dim3 grid(per * nblocks);
        if (g.ups) SBA_LAUNCH((conv3x3_halo2_kernel<1>), grid, dim3(256), 0, st, x, w, y, addend, stats, g, ex, tiles, per);
        else SBA_LAUNCH((conv3x3_halo2_kernel<0>), grid, dim3(256), 0, st, x, w, y, addend, stats, g, ex, tiles, per);
        return;
    }
    // BN = 64 for every Cout: the 128-wide variant needs 86 KB of LDS (one workgroup per CU) and
    // measured slower; re-staging the halo tile for the second channel block is cheap
    dim3 grid(tiles, nblocks);
    if (g.w_layout == 1) {      // fragment-major weights: the register-resident form (geom_ok has checked the layout's needs)
        if (g.ups) SBA_LAUNCH((conv3x3_halo3_kernel<64, 1>), grid, dim3(256), 0, st, x, w, y, addend, stats, g, ex);
        else SBA_LAUNCH((conv3x3_halo3_kernel<64, 0>), grid, dim3(256), 0, st, x, w, y, addend, stats, g, ex);
        return;
    }
    if (g.ups) SBA_LAUNCH((conv3x3_halo_kernel<64, 64, 1>), grid, dim3(256), 0, st, x, w, y, addend, stats, g, ex);
    else SBA_LAUNCH((conv3x3_halo_kernel<64, 64, 0>), grid, dim3(256), 0, st, x, w, y, addend, stats, g, ex);
}

template <typename T>
int launch_igemm(const void* x, const void* w, void* y, const void* addend, float* stats,
                 const sba_conv_geom& g, void* workspace, int64_t ws_bytes, hipStream_t st,
                 const EpiX ex = EpiX{nullptr, nullptr, 0}, int* plan = nullptr) {
    // plan != NULL: do not launch -- report the kernel this geometry goes to: plan[0] = family (0 halo-tile 3x3, 1 LDS-DMA
    // gen 2 (64-channel slabs), 2 LDS-DMA gen 1, 3 register-staged), plan[1] = tile id / configuration, plan[2] = K splits
    const int M = g.N * g.OHs * g.OWs;
    const T* xp = (const T*)x; const T* wp = (const T*)w; T* yp = (T*)y; const T* ap = (const T*)addend;
    // fragment-major weights: the halo-tile kernels only (family 0: the strict form; family 4: the general form)
    if (g.w_layout != 0 && !(g.w_layout == 1 && sizeof(T) == 2 && (halo_ok(g) || halo3g_ok(g)))) return SBA_E_ARG;
    if (sizeof(T) == 2 && g.w_layout == 1 && !halo_ok(g)) {
        if (plan) { plan[0] = 4; plan[1] = g.Cin % 64 == 0 ? 64 : 32; plan[2] = 1; return SBA_OK; }
        if (stats) return SBA_E_ARG;
        launch_halo3g(g, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, (const bf16_t*)addend, ex, st);
        return SBA_CHECK_LAUNCH();
    }
    if (sizeof(T) == 2 && halo_ok(g)) {
        if (plan) { plan[0] = 0; plan[1] = g.ups ? 1 : 0; plan[2] = 1; return SBA_OK; }
        launch_halo(g, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, (const bf16_t*)addend, stats, ex, st);
        return SBA_CHECK_LAUNCH();
    }
    const int nslabs = g.ntaps * (g.Cin / (64 / (int)sizeof(T)));
    const bool can_split = workspace && g.Cout % 4 == 0 && (int64_t)M * g.Cout * 4 <= ws_bytes && nslabs >= 16;
    // Rule table calibrated with tools/bench_conv.py on the B=20 layer shapes (profiles/r01_conv_tiles.txt):
    //  - GEMM-like maps (M <= 2048): small tile D with split-K;
    //  - Cout multiple of 128: A when it yields >= 320 workgroups, E (320x128, 10 waves) when it yields
    //    thousands, otherwise the mid tile C;
    //  - narrow Cout (64): B when >= 256 workgroups, otherwise C.
    int best, best_split = 1;
    const int wgA = (g.Cout % 128 == 0) ? cdiv(M, 128) * (g.Cout / 128) : 0;
    if (M <= 2048) {
        best = (g.Cout >= 1024 && (nslabs >= 512 || (M >= 1280 && nslabs >= 256))) ? 4 : 3;
    } else if ((g.Cout == 128 && M >= 40000) || (g.Cout == 256 && M >= 40000 && nslabs >= 48)) {
        best = 4;
    } else if (wgA >= 320) {
        best = 0;
    } else if (g.Cout <= 64 && cdiv(M, 256) >= 256) {
        best = 1;
    } else if (cdiv(M, 128) * cdiv(g.Cout, 64) >= 256) {
        best = 2;
    } else {
        best = 3;       // too few mid tiles to fill the chip: small tiles (+ split-K when K is long)
    }
    if (forced_cfg() >= 0) best = forced_cfg();
    {
        const IgemmCfg& k = kCfg[best];
        const int tiles = cdiv(M, k.bm) * cdiv(g.Cout, k.bn);
        const int slots = 256 * k.occ;
        static int split_m = -1;
        if (split_m < 0) { const char* e = getenv("SBA_IGEMM_SPLIT_M"); split_m = e ? atoi(e) : 2048; }
        if (k.split && can_split && tiles < slots && M <= split_m) {     // split-K only pays on the GEMM-like maps
            int split = slots / tiles;          // floor: one more split than fits leaves a nearly empty second round
            if (split > nslabs / 8) split = nslabs / 8;
            if (split > 32) split = 32;
            if (split > 1) best_split = split;
        }
    }
    const bool det = sba_det_on();      // deterministic mode: no split-K (its partial sums meet in f32 atomics)
    if (det) best_split = 1;
    float* ws = (float*)workspace;
    if (sizeof(T) == 2 && dma_enabled()) {
        const bf16_t* xb = (const bf16_t*)x; const bf16_t* wb = (const bf16_t*)w; bf16_t* yb = (bf16_t*)y;
        const bf16_t* ab = (const bf16_t*)addend;
        // tile ids (include/sbagan_hip.h: sba_conv_geom.tile): BM x BN, slabs per stage, ring depth
        //   1/2: 64x64 (64 / 128 KB of LDS)   3/4: 96x64 (72 / 144 KB)   5/6: 128x64 (48 / 144 KB)
        //   7/8: 128x128 (64 / 128 KB)        9/10: 256x64 (60 / 120 KB) 11: 320x128 register-staged   12: 96x128 (120 KB)
        //   13: 320x64 (5 waves, 150 KB) / 14: 160x64 (5 waves, 120 KB) / 15: 160x64 with a 2-stage ring (60 KB: two
        //   workgroups per CU): the WHOLE batch of a 4x4 map (M = 16 B = 320) in one
        //   M tile -- every weight byte is staged once per M tile, so these weight-streaming layers (4..38 MB of weights
        //   against 320..640 rows) move 1/3..1/5 of the L2->LDS bytes of the 64- / 96-row tiles; Cin % 64 == 0 only
        // the deep rings keep ~100 KB of loads in flight per CU: an L2-hit load takes ~1 us under load, so a
        // workgroup alone on its CU moves bytes_in_flight / 1 us (measured 36-42 GB/s with 48 KB in flight)
        int tile = g.tile;
        int split = best_split;
        if (tile <= 0 || tile > SBA_IGEMM_TILES) {
            static const int rule_tile[4] = {7, 9, 5, 1};
            tile = best <= 3 ? rule_tile[best] : 11;
        } else if (g.ksplit >= 1) {
            split = g.ksplit;
            if (split > 1 && !(workspace && g.Cout % 4 == 0 && (int64_t)M * g.Cout * 4 <= ws_bytes)) split = 1;
            if (split > nslabs / 2) split = nslabs / 2 > 0 ? nslabs / 2 : 1;
        }
        if (det) split = 1;
        static int gen2 = -1;       // SBA_IGEMM_DMA2=0: first-generation kernels only (A/B aid)
        if (gen2 < 0) { const char* e = getenv("SBA_IGEMM_DMA2"); gen2 = (e && e[0] == '0') ? 0 : 1; }
        if (tile >= 13 && tile != 12 && !(gen2 && g.Cin % 64 == 0)) {       // gen-2 only: back to the rules
            static const int rule_tile[4] = {7, 9, 5, 1};
            tile = best <= 3 ? rule_tile[best] : 11;
            split = det ? 1 : best_split;
        }
        if (gen2 && g.Cin % 64 == 0 && tile != 11) {
            // 128-byte rows, fragment double buffering, DMA issue between the MFMAs (igemm_dma2_kernel)
            const int ns64 = nslabs / 2;
            int sp = split;
            if (sp > ns64 / 2) sp = ns64 / 2 > 0 ? ns64 / 2 : 1;
            if (plan) { plan[0] = 1; plan[1] = tile; plan[2] = sp; return SBA_OK; }
            switch (tile) {
                case 1: launch_dma2<64, 64, 32, 32, 4>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 2: launch_dma2<64, 64, 32, 32, 8>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 3: launch_dma2<96, 64, 32, 64, 3>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 4: launch_dma2<96, 64, 32, 64, 6>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 5: launch_dma2<128, 64, 32, 64, 3>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 6: launch_dma2<128, 64, 32, 64, 6>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 7: launch_dma2<128, 128, 64, 64, 3>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 8: launch_dma2<128, 128, 64, 64, 4>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 9: launch_dma2<256, 64, 64, 64, 3>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 10: launch_dma2<256, 64, 64, 64, 4>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 13: launch_dma2<320, 64, 64, 64, 3>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 14: launch_dma2<160, 64, 32, 64, 4>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 15: launch_dma2<160, 64, 32, 64, 2>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 16: launch_dma2<256, 128, 128, 64, 3>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 17: launch_dma2<256, 128, 64, 64, 3>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                case 18: launch_dma2<256, 128, 128, 64, 2>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
                default: launch_dma2<96, 128, 32, 128, 5>(xb, wb, yb, ab, stats, g, M, ns64, sp, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            }
        }
        if (plan && tile != 11) { plan[0] = 2; plan[1] = tile; plan[2] = split; return SBA_OK; }
        switch (tile) {
            case 1: launch_dma<64, 64, 32, 32, 2, 4>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 2: launch_dma<64, 64, 32, 32, 2, 8>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 3: launch_dma<96, 64, 32, 64, 2, 3>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 4: launch_dma<96, 64, 32, 64, 2, 6>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 5: launch_dma<128, 64, 32, 64, 1, 4>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 6: launch_dma<128, 64, 32, 64, 2, 6>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 7: launch_dma<128, 128, 64, 64, 1, 4>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 8: launch_dma<128, 128, 64, 64, 1, 8>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 9: launch_dma<256, 64, 64, 64, 1, 3>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 10: launch_dma<256, 64, 64, 64, 1, 6>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            case 12: launch_dma<96, 128, 32, 128, 1, 8>(xb, wb, yb, ab, stats, g, M, nslabs, split, ws, ws_bytes, st, ex); return SBA_CHECK_LAUNCH();
            default: best = 4; best_split = split; break;      // 11: the register-staged 320x128 tile below
        }
    }
    if (plan) { plan[0] = 3; plan[1] = best; plan[2] = best_split; return SBA_OK; }
    switch (best) {
        case 0: launch_cfg<T, 128, 128, 64, 64, 1>(xp, wp, yp, ap, stats, g, M, nslabs, best_split, ws, st, ex); break;
        case 1: launch_cfg<T, 256, 64, 64, 64, 1>(xp, wp, yp, ap, stats, g, M, nslabs, best_split, ws, st, ex); break;
        case 2: launch_cfg<T, 128, 64, 32, 64, 2>(xp, wp, yp, ap, stats, g, M, nslabs, best_split, ws, st, ex); break;
        case 3: launch_cfg<T, 64, 64, 32, 32, 2>(xp, wp, yp, ap, stats, g, M, nslabs, best_split, ws, st, ex); break;
        default: launch_cfg<T, 320, 128, 64, 64, 2>(xp, wp, yp, ap, stats, g, M, nslabs, best_split, ws, st, ex); break;
    }
    return SBA_CHECK_LAUNCH();
}

bool geom_ok(const sba_conv_geom* g, int dtype) {
    if (!g) return false;
    const int ks = dtype == SBA_BF16 ? 32 : 16;
    if (g->ntaps < 1 || g->ntaps > SBA_MAX_TAPS) return false;
    for (int t = 0; t < g->ntaps; ++t)
        if (g->ty[t] < -8 || g->ty[t] > 7 || g->tx[t] < -8 || g->tx[t] > 7) return false;
    if (g->Cin <= 0 || g->Cin % ks != 0) return false;
    if (g->N <= 0 || g->IH <= 0 || g->IW <= 0 || g->OH <= 0 || g->OW <= 0 || g->Cout <= 0) return false;
    if (g->OHs <= 0 || g->OWs <= 0 || g->osy <= 0 || g->osx <= 0) return false;
    // every written output pixel must lie inside OH x OW
    if ((g->OHs - 1) * g->osy + g->ooy >= g->OH || (g->OWs - 1) * g->osx + g->oox >= g->OW) return false;
    if (g->ooy < 0 || g->oox < 0) return false;
    if ((int64_t)g->N * g->OHs * g->OWs > 0x7fffffff) return false;
    const int64_t esz = dtype == SBA_BF16 ? 2 : 4;
    const int vec = dtype == SBA_BF16 ? 8 : 4;
    const int xcs = g->x_cstride ? g->x_cstride : g->Cin, ycs = g->y_cstride ? g->y_cstride : g->Cout;
    if (xcs < g->Cin + g->x_coff || ycs < g->Cout + g->y_coff || g->x_coff < 0 || g->y_coff < 0) return false;
    if (xcs % vec || g->x_coff % vec || ycs % vec || g->y_coff % vec) return false;
    if ((int64_t)g->N * g->IH * g->IW * xcs * esz >= ((int64_t)1 << 32)) return false;    // 32-bit byte offsets
    if ((int64_t)g->Cout * g->ntaps * g->Cin * esz >= ((int64_t)1 << 32)) return false;
    if ((int64_t)g->N * g->IH * g->IW > 0x7fffffff / 2 || (int64_t)g->N * g->OH * g->OW > 0x7fffffff / 2)
        return false;
    return true;
}

// the launch writes every pixel and channel of a dense [N*OH*OW][Cout] tensor
bool dense_output(const sba_conv_geom& g) {
    return g.OHs == g.OH && g.OWs == g.OW && g.osy == 1 && g.osx == 1 && g.ooy == 0 && g.oox == 0 &&
           (g.y_cstride == 0 || g.y_cstride == g.Cout) && g.y_coff == 0;
}

}  // namespace

extern "C" int sba_conv_igemm(int dtype, const void* x, const void* w, void* y, const void* addend,
                              float* stats, const sba_conv_geom* g, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    if (dtype == SBA_BF16_YH) {
        // bf16 operands, y stored as binary16 (the pre-BatchNorm tensor: include/sbagan_hip.h)
        if (!x || !w || !y || !geom_ok(g, SBA_BF16) || addend || g->relu || g->Cout % 8) return SBA_E_ARG;
        if (((uintptr_t)workspace & 15) != 0) return SBA_E_ARG;
        const EpiX ex{nullptr, nullptr, 1};
        if (sba_det_on() && stats) {
            if (!dense_output(*g)) return SBA_E_ARG;
            const int rc = launch_igemm<bf16_t>(x, w, y, nullptr, nullptr, *g, workspace, workspace_bytes, (hipStream_t)stream, ex);
            if (rc != SBA_OK) return rc;
            return sba_bn_stats(dtype, y, stats, (int64_t)g->N * g->OH * g->OW, 1, g->Cout, stream);
        }
        return launch_igemm<bf16_t>(x, w, y, nullptr, stats, *g, workspace, workspace_bytes, (hipStream_t)stream, ex);
    }
    if (!x || !w || !y || !geom_ok(g, dtype)) return SBA_E_ARG;
    if (((uintptr_t)workspace & 15) != 0) return SBA_E_ARG;
    if (sba_det_on() && stats) {
        // deterministic mode: the epilogue's statistics meet in f32 atomics (LDS and global); take them from the
        // stored tensor with the ordered bn_stats pass instead (needs a dense output: it is one BatchNorm batch)
        if (!dense_output(*g) || addend) return SBA_E_ARG;
        int rc = SBA_E_ARG;
        SBA_DISPATCH(dtype, rc = launch_igemm<T>(x, w, y, addend, nullptr, *g, workspace, workspace_bytes,
                                                 (hipStream_t)stream));
        if (rc != SBA_OK) return rc;
        return sba_bn_stats(dtype, y, stats, (int64_t)g->N * g->OH * g->OW, 1, g->Cout, stream);
    }
    SBA_DISPATCH(dtype, return launch_igemm<T>(x, w, y, addend, stats, *g, workspace, workspace_bytes,
                                               (hipStream_t)stream));
    return SBA_E_ARG;
}

extern "C" int sba_conv_igemm_bias(int dtype, const void* x, const void* w, void* y, const void* addend,
                                   float* stats, const float* bias, const void* relu_mask,
                                   const sba_conv_geom* g, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!x || !w || !y || !geom_ok(g, dtype)) return SBA_E_ARG;
    if (((uintptr_t)workspace & 15) != 0) return SBA_E_ARG;
    if (sba_det_on() && stats) return SBA_E_ARG;        // (no caller asks for statistics behind a bias / ReLU epilogue)
    SBA_DISPATCH(dtype, return launch_igemm<T>(x, w, y, addend, stats, *g, workspace, workspace_bytes,
                                               (hipStream_t)stream, EpiX{bias, relu_mask, 0}));
    return SBA_E_ARG;
}

extern "C" int sba_conv_igemm_plan(int dtype, const sba_conv_geom* g, int64_t workspace_bytes, int* plan) {
    if (!plan || (dtype != SBA_F32 && dtype != SBA_BF16) || !geom_ok(g, dtype)) return SBA_E_ARG;
    static char dummy_ws[16];
    void* ws = workspace_bytes > 0 ? (void*)dummy_ws : nullptr;       // (only its presence and size enter the decision)
    if (dtype == SBA_F32) return launch_igemm<float>(nullptr, nullptr, nullptr, nullptr, nullptr, *g, ws, workspace_bytes,
                                                     nullptr, EpiX{nullptr, nullptr, 0}, plan);
    return launch_igemm<bf16_t>(nullptr, nullptr, nullptr, nullptr, nullptr, *g, ws, workspace_bytes, nullptr,
                                EpiX{nullptr, nullptr, 0}, plan);
}

template <int BM, int BN, int WM, int WN, int D, int KS = 0>
static int launch_group(const sba_conv_group_item* items, int n, hipStream_t st, int split = 1, float* ws = nullptr) {
    GroupArgs A;
    A.n = n;
    A.sps = 0;
    int tiles = 0, max_m = 0, max_co = 0, ns64 = 0;
    for (int i = 0; i < n; ++i) {
        const sba_conv_geom& g = *items[i].g;
        GroupItem& it = A.it[i];
        it.x = (const bf16_t*)items[i].x; it.w = (const bf16_t*)items[i].w; it.y = (bf16_t*)items[i].y;
        it.addend = (const bf16_t*)items[i].addend; it.bias = items[i].bias; it.mask = items[i].relu_mask;
        it.g = g;
        it.M = g.N * g.OHs * g.OWs;
        it.gx = cdiv(it.M, BM);
        it.gy = cdiv(g.Cout, BN);
        it.nmajor = nmajor_for(g);
        it.pad = 0;
        it.ws = nullptr;
        it.tile_begin = tiles;
        tiles += it.nmajor ? 8 * cdiv(it.gy, 8) * it.gx : 8 * cdiv(it.gx, 8) * it.gy;
        if (it.M > max_m) max_m = it.M;
        if (g.Cout > max_co) max_co = g.Cout;
        const int k = g.ntaps * (g.Cin / 64);
        if (k > ns64) ns64 = k;
    }
    if (split > 1 && KS == 0 && ws) {
        // every item is cut into the same number of K splits (slabs per split from the longest K); item i adds its
        // partial sums into its own [M][Cout] f32 slice of the zero-filled workspace
        A.sps = cdiv(ns64, split);
        split = cdiv(ns64, A.sps);
        float* p = ws;
        for (int i = 0; i < n; ++i) { A.it[i].ws = p; p += (int64_t)A.it[i].M * A.it[i].g.Cout; }
        if (split <= 1) { A.sps = 0; split = 1; }
    } else {
        split = 1;
    }
    for (int i = n; i < SBA_GROUP_MAX; ++i) A.it[i] = A.it[0];
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    if (KS == 0) SBA_LAUNCH((igemm_dma2_group_kernel<BM, BN, WM, WN, D>), dim3(tiles, 1, split), dim3(NT), 0, st, A DMA_TRACE_ARG);
    else SBA_LAUNCH((igemm_dma_group_kernel<BM, BN, WM, WN, (KS ? KS : 1), D>), dim3(tiles), dim3(NT), 0, st, A DMA_TRACE_ARG);
    if (A.sps) {
        dim3 fgrid(cdiv(max_co / 4, 256), cdiv(max_m, 8), n);
        SBA_LAUNCH(splitk_finish_group_kernel, fgrid, dim3(256), 0, st, A);
    }
    return SBA_CHECK_LAUNCH();
}

static int group_dispatch(int dtype, int n, const sba_conv_group_item* items, int tile, int ksplit, void* workspace,
                          int64_t ws_bytes, hipStream_t st) {
    if (dtype != SBA_BF16 || !items || n < 1 || n > SBA_GROUP_MAX) return SBA_E_ARG;
    bool all64 = true;
    int64_t need = 0;
    for (int i = 0; i < n; ++i) {
        const sba_conv_group_item& it = items[i];
        if (!it.x || !it.w || !it.y || !geom_ok(it.g, dtype)) return SBA_E_ARG;     // (geom_ok: Cin % 32 == 0)
        all64 = all64 && it.g->Cin % 64 == 0;
        need += (int64_t)it.g->N * it.g->OHs * it.g->OWs * it.g->Cout * 4;
        if (ksplit > 1 && it.g->Cout % 4) ksplit = 1;
    }
    if (ksplit > 1 && (!workspace || need > ws_bytes || ((uintptr_t)workspace & 15) || !all64 || sba_det_on())) ksplit = 1;
    if (ksplit < 1) ksplit = 1;
    float* ws = (float*)workspace;
    if (all64) {        // 64-channel slabs, second-generation body
        switch (tile) {
            case 0: case 1: return launch_group<64, 64, 32, 32, 4>(items, n, st, ksplit, ws);
            case 3: return launch_group<96, 64, 32, 64, 3>(items, n, st, ksplit, ws);
            case 5: return launch_group<128, 64, 32, 64, 3>(items, n, st, ksplit, ws);
            case 7: return launch_group<128, 128, 64, 64, 3>(items, n, st, ksplit, ws);
            default: return SBA_E_ARG;
        }
    }
    switch (tile) {     // some member has Cin % 64 == 32: 32-channel slabs for the whole group
        case 0: case 1: return launch_group<64, 64, 32, 32, 4, 2>(items, n, st);
        case 3: return launch_group<96, 64, 32, 64, 3, 2>(items, n, st);
        case 5: case 7: return launch_group<128, 64, 32, 64, 4, 1>(items, n, st);
        default: return SBA_E_ARG;
    }
}

extern "C" int sba_conv_igemm_group(int dtype, int n, const sba_conv_group_item* items, int tile, void* stream) {
    return group_dispatch(dtype, n, items, tile, 1, nullptr, 0, (hipStream_t)stream);
}

extern "C" int sba_conv_igemm_group_splitk(int dtype, int n, const sba_conv_group_item* items, int tile, int ksplit,
                                           void* workspace, int64_t workspace_bytes, void* stream) {
    return group_dispatch(dtype, n, items, tile, ksplit, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int sba_conv_wgrad(int dtype, const void* x, const void* dy, float* dw, const sba_conv_geom* g,
                              int ksplit, void* stream) {
    if (!x || !dy || !dw || !geom_ok(g, dtype)) return SBA_E_ARG;
    if (g->Cin % 8 != 0 || g->Cout % 8 != 0) return SBA_E_ARG;
    if (g->x_cstride || g->x_coff || g->y_cstride || g->y_coff || g->ntaps > 16) return SBA_E_ARG;
    const int M = g->N * g->OHs * g->OWs;
    if (ksplit < 1) ksplit = 1;
    const int fw = g->first_write ? 2 : 0;       // epilogue mode of the exclusive-owner kernels: 0 = +=, 1 = atomic, 2 = store
    const int co_tiles = cdiv(g->Cout, 64), items = cdiv(g->Cin, 64) * g->ntaps;
    const FastDiv dsub = make_fastdiv((uint32_t)(g->OHs * g->OWs), (int64_t)M + 64);
    const FastDiv dow = make_fastdiv((uint32_t)g->OWs, (int64_t)M + 64);
    // Deterministic mode: pixel splits do not meet in f32 atomics -- split z STORES its partial gradient into its own
    // tensor of the scratch ring (mode 2, offset z * zstride) and sba_det_fold adds the splits up in order.
    hipStream_t st = (hipStream_t)stream;
    const int64_t dwn = (int64_t)g->Cout * g->ntaps * g->Cin;
    float* part = nullptr;
    auto det_begin = [&](int nsplit) -> bool {
        part = nullptr;
        if (sba_det_on() && nsplit > 1) { part = sba_det_alloc((int64_t)nsplit * dwn); return part != nullptr; }
        return true;
    };
    auto det_end = [&](int nsplit) { if (part) sba_det_fold(part, 1, nsplit, dwn, dw, 0, fw == 2 ? 1 : 0, st); };
    // Kernel-row decomposition (wgrad_row_dma_kernel): the kw taps of a kernel row share one staged input row segment.
    // 4x4 / stride-2 down blocks and 3x3 / stride-1 convs on maps of 8 x 8 .. (below the all-taps halo-row kernel's range).
    // SBA_WGRAD_S2=0 / SBA_WGRAD_S1=0: off (A/B aids); SBA_WGRAD_S1=1: only below the halo-row kernel's range; 2: also instead
    // of it; 3 (default): also behind the nearest x2 upsample (G upsample1..4 73 / 92 / 102 / 80 -> 41 / 56 / 55 / 53 us,
    // upBlock -> 128 px 123 -> 100, -> 256 px 192 -> 187) -- tools/bench_wgrad.py, B = 20: ResBlock 64 x 64 44.0 -> 28.4 us, 64->128 @64 61.6 -> 38.3; at 128 x 128
    // against wgrad_rows_kernel: 64->64 95.8 -> 54.9 us, 64->128 122.8 -> 92.8 (profiles/r04_wgrad_s2_rows.txt).
    {
        static int s2 = -1, s1 = -1, row_wgs = -1, rows_m1 = -1;
        if (s2 < 0) { const char* e = getenv("SBA_WGRAD_S2"); s2 = e ? atoi(e) : 1; }
        if (s1 < 0) { const char* e = getenv("SBA_WGRAD_S1"); s1 = e ? atoi(e) : 3; }
        if (row_wgs < 0) { const char* e = getenv("SBA_WGRAD_S2_WGS"); row_wgs = e ? atoi(e) : 512; }
        if (rows_m1 < 0) { const char* e = getenv("SBA_WGRAD_ROWS_M"); rows_m1 = e ? atoi(e) : 131072; }
        const int kwn = g->ntaps == 16 ? 4 : (g->ntaps == 9 ? 3 : 0), sxy = g->sx;
        bool ok = dtype == SBA_BF16 && kwn && g->sy == sxy && (!g->ups || (kwn == 3 && s1 >= 3)) && g->osy == 1 && g->osx == 1 && g->ooy == 0 &&
                  g->oox == 0 && g->OHs == g->OH && g->OWs == g->OW && g->Cin % 64 == 0 && g->Cout % 64 == 0 &&
                  g->OW >= 4 && (g->OW & (g->OW - 1)) == 0 && !(g->OW == 4 && (g->ups || g->OH != 4)) && M % 32 == 0 &&
                  ((g->OH * g->OW) % 32 == 0 || g->OW == 4);
        ok = ok && ((kwn == 4 && sxy == 2 && s2) ||
                    (kwn == 3 && sxy == 1 && s1 && (s1 >= 2 || M < rows_m1 || g->OW % 64 != 0)));
        // SBA_WGRAD_ALLROWS=1 (experiment, off): all three kernel rows of a small-dW 3x3 conv per workgroup (KR = 3: nine waves,
        // dy shared by nine taps, 1.2 instead of 3 KB staged per tap).  Measured SLOWER almost everywhere (one workgroup of
        // nine waves per CU behind one barrier per stage): ResBlock 128 x 128 58.9 -> 71.4 us, 64 x 64 29.5 -> 42.3, upsample4
        // 54.5 -> 68.9 at 256 workgroups (worse at 320 / 512); only the 256 px upBlock gains (183.6 -> 156.8).
        static int allrows = -1, all9_wgs = -1;
        if (allrows < 0) { const char* e = getenv("SBA_WGRAD_ALLROWS"); allrows = e ? atoi(e) : 0; }
        if (all9_wgs < 0) { const char* e = getenv("SBA_WGRAD_ALL9_WGS"); all9_wgs = e ? atoi(e) : 256; }
        static int ow4 = -1;        // SBA_WGRAD_ROW_OW4=0: not on the 4 x 4 maps (two images per chunk; A/B aid).  B = 40: D256's 3x3
                                    // 2048->1024 78.6 -> 55.8 us, D128's 1024->512 38.1 -> 27.3, 512->1024 4x4/s2 35.8 -> 27.9
        if (ow4 < 0) { const char* e = getenv("SBA_WGRAD_ROW_OW4"); ow4 = e ? atoi(e) : 1; }
        ok = ok && (g->OW > 4 || ow4);
        for (int t = 0; t < g->ntaps && ok; ++t)
            ok = g->ty[t] == g->ty[(t / kwn) * kwn] && g->tx[t] == g->tx[(t / kwn) * kwn] + (t % kwn);
        const int64_t xb = (int64_t)g->N * g->IH * g->IW * g->Cin * 2, db = (int64_t)g->N * g->OH * g->OW * g->Cout * 2;
        if (ok && xb < (1ll << 32) && db < (1ll << 32)) {
            const int wc = g->OW < 32 ? g->OW : 32;
            int wclog = 0;
            while ((1 << wclog) < wc) ++wclog;
            // all three kernel rows of a 3x3 conv in one workgroup (nine waves, the dy slice shared by the nine taps) where dW
            // is small: with 1..4 workgroups per pixel split the extra splits' f32 atomics cost nothing
            const bool all9 = allrows && kwn == 3 && g->OW >= 8 && co_tiles * (g->Cin / 64) <= 4;
            const int wgs = all9 ? co_tiles * (g->Cin / 64) : co_tiles * kwn * (g->Cin / 64);
            const int tc32 = M / 32;
            // pixel splits: each one adds a full f32-atomic copy of dW (~1.3 TB/s chip-wide): fill the chip about twice,
            // keep >= 12 chunks behind a copy
            // (tools/bench_wgrad.py, B = 40: 128->256 @64 71 us at 512 workgroups, 78 at 384, 94 at 256; the 64->128 layers,
            // 8 workgroups per split: @128 89 / 86 / 94, @64 42 / 37 / 36; all9: one workgroup per CU)
            int sp = cdiv(all9 ? all9_wgs : (wgs <= 8 ? (row_wgs * 3) / 4 : row_wgs), wgs);
            if (sp > tc32 / 12) sp = tc32 / 12 > 0 ? tc32 / 12 : 1;
            const int cps32 = cdiv(tc32, sp);
            sp = cdiv(tc32, cps32);
            dim3 gd(co_tiles, (all9 ? 1 : kwn) * (g->Cin / 64), sp);
            if (gd.y <= 65535 && gd.z <= 65535) {
                if (!det_begin(sp)) return SBA_E_ARG;
                float* dwa = part ? part : dw;
                const int md = part ? 2 : (sp > 1 ? 1 : fw);
                const int64_t zs = part ? dwn : 0;
                // (ring depth: 3 and 4 measure the same, 6 is 30-60 % slower -- occupancy: profiles/r04_wgrad_s2_rows.txt)
                if (kwn == 4) {         // x segment: <= 72 rows (OW >= 8), 80 rows (two 4 x 4 maps)
                    constexpr int LDS = 4 * WgRowCfg<4, 10>::STAGE;
                    static bool once = false;
                    if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_row_dma_kernel<4, 2, 10, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
                    SBA_LAUNCH((wgrad_row_dma_kernel<4, 2, 10, 4>), gd, dim3(256), LDS, (hipStream_t)stream, (const bf16_t*)x,
                               (const bf16_t*)dy, dwa, *g, M, cps32, md, dsub, dow, zs, wclog);
                } else if (g->OW == 4) {       // 8 rows x 6 = 48 rows
                    constexpr int LDS = 4 * WgRowCfg<3, 6>::STAGE;
                    static bool once = false;
                    if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_row_dma_kernel<3, 1, 6, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
                    SBA_LAUNCH((wgrad_row_dma_kernel<3, 1, 6, 4>), gd, dim3(192), LDS, (hipStream_t)stream, (const bf16_t*)x,
                               (const bf16_t*)dy, dwa, *g, M, cps32, md, dsub, dow, zs, wclog);
                } else if (all9) {
                    constexpr int LDS = 3 * WgRowCfg<3, 5, 3>::STAGE;
                    static bool once = false;
                    if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_row_dma_kernel<3, 1, 5, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
                    SBA_LAUNCH((wgrad_row_dma_kernel<3, 1, 5, 3, 3>), gd, dim3(576), LDS, (hipStream_t)stream, (const bf16_t*)x,
                               (const bf16_t*)dy, dwa, *g, M, cps32, md, dsub, dow, zs, wclog);
                } else {                       // <= 40 rows
                    constexpr int LDS = 4 * WgRowCfg<3, 5>::STAGE;
                    static bool once = false;
                    if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_row_dma_kernel<3, 1, 5, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
                    SBA_LAUNCH((wgrad_row_dma_kernel<3, 1, 5, 4>), gd, dim3(192), LDS, (hipStream_t)stream, (const bf16_t*)x,
                               (const bf16_t*)dy, dwa, *g, M, cps32, md, dsub, dow, zs, wclog);
                }
                det_end(sp);
                return SBA_CHECK_LAUNCH();
            }
        }
    }
    static int small_m = -1;
    if (small_m < 0) { const char* e = getenv("SBA_WGRAD_SMALL_M"); small_m = e ? atoi(e) : 12000; }
    if (M <= small_m && co_tiles * items >= 256) {
        // GEMM-like layer: one tile per wave, all pixels (ksplit re-derived for this decomposition)
        const int total_chunks = cdiv(M, 16);
        const int wgs = co_tiles * cdiv(items, 4);
        // pixel splits: each one adds a full f32-atomic copy of every 64x64 tile (the atomics run at ~1.3 TB/s),
        // so split only up to ~3 workgroups per CU and keep >= 24 chunks (384 pixels) of MFMA work behind a copy
        // (measured on the discriminator shapes: joint conv 68 -> 44 us, s64_2 68 -> 46, c4 166 -> 133)
        static int tgt = -1, minc = -1;
        if (tgt < 0) { const char* e = getenv("SBA_WGRAD_SMALL_WGS"); tgt = e ? atoi(e) : 768; }
        if (minc < 0) { const char* e = getenv("SBA_WGRAD_SMALL_MINC"); minc = e ? atoi(e) : 24; }
        int split = wgs >= tgt / 2 ? 1 : cdiv(tgt, wgs);
        if (split > total_chunks / minc) split = total_chunks / minc > 0 ? total_chunks / minc : 1;
        const int cps = cdiv(total_chunks, split);
        split = cdiv(total_chunks, cps);
        dim3 grid(co_tiles, cdiv(items, 4), split);
        if (grid.y > 65535 || grid.z > 65535) return SBA_E_ARG;
        static int ct2 = -1;        // SBA_WGRAD_SMALL_CT2: minimum workgroups for the 128-wide variant (0 = never,
                                    // the default: measured 20-35 % SLOWER than CT = 1 at 4 waves per SIMD)
        if (ct2 < 0) { const char* e = getenv("SBA_WGRAD_SMALL_CT2"); ct2 = e ? atoi(e) : 0; }
        if (ct2 > 0 && split == 1 && g->Cout % 128 == 0 && (co_tiles / 2) * (int)grid.y >= ct2) {
            grid.x = co_tiles / 2;
            SBA_DISPATCH(dtype, SBA_LAUNCH((wgrad_small_kernel<T, 2>), grid, dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, (const T*)dy, dw, *g, M, cps, fw, dsub, dow, (int64_t)0));
            return SBA_CHECK_LAUNCH();
        }
        static int dma = -1;        // SBA_WGRAD_DMA: 0 = register-staged kernel; D = ring depth of the LDS-DMA kernel
        if (dma < 0) { const char* e = getenv("SBA_WGRAD_DMA"); dma = e ? atoi(e) : 4; }
        const int64_t xb = (int64_t)g->N * g->IH * g->IW * g->Cin * 2, db = (int64_t)g->N * g->OH * g->OW * g->Cout * 2;
        // measured (tools/bench_wgrad.py, B = 20): 15-25 % faster than the register-staged kernel up to ~512
        // workgroups (joint conv 29 -> 25 us, D s32 61 -> 50, s32_1 49 -> 37); beyond that the launches are bound by
        // the L2 traffic of the operand slices either way and the 80 KB ring costs occupancy (s64 145 -> 175 us)
        static int dma_wgs = -1;
        if (dma_wgs < 0) { const char* e = getenv("SBA_WGRAD_DMA_WGS"); dma_wgs = e ? atoi(e) : 512; }
        // beyond dma_wgs: the DMA kernel with TWO co tiles per wave (0.375 KB of operands per MFMA) -- pays once the
        // epilogue is a plain store (first write: D256 s64 119 -> 95 us, s64_1 89 -> 79, G upsample1 84 -> 72); with the
        // read-modify-write epilogue it is no faster than the register-staged kernel (148 vs 146 us).
        // SBA_WGRAD_DMA_CT2: 1 = always, 0 = never, unset = on first writes.
        static int dma_ct2 = -2;
        if (dma_ct2 == -2) { const char* e = getenv("SBA_WGRAD_DMA_CT2"); dma_ct2 = e ? atoi(e) : -1; }
        if (dma > 0 && (dma_ct2 > 0 || (dma_ct2 < 0 && fw == 2)) && wgs > dma_wgs && g->Cout % 128 == 0 &&
            dtype == SBA_BF16 && xb < (1ll << 32) && db < (1ll << 32)) {
            const int tc32 = cdiv(M, 32);
            dim3 gd(co_tiles / 2, cdiv(items, 4), 1);
            constexpr int LDS = 3 * 6 * 32 * 128;
            static bool once = false;
            if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_small_dma_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
            SBA_LAUNCH((wgrad_small_dma_kernel<2, 3>), gd, dim3(256), LDS, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)dy, dw, *g, M, tc32, fw, dsub, dow, (int64_t)0);
            return SBA_CHECK_LAUNCH();
        }
        if (dma > 0 && wgs <= dma_wgs && dtype == SBA_BF16 && xb < (1ll << 32) && db < (1ll << 32)) {
            // stages of 32 pixels; the same split rule restated in 32-pixel chunks
            const int tc32 = cdiv(M, 32);
            int sp = wgs >= tgt / 2 ? 1 : cdiv(tgt, wgs);
            const int mc = minc / 2 > 0 ? minc / 2 : 1;
            if (sp > tc32 / mc) sp = tc32 / mc > 0 ? tc32 / mc : 1;
            const int cps32 = cdiv(tc32, sp);
            sp = cdiv(tc32, cps32);
            dim3 gd(co_tiles, cdiv(items, 4), sp);
            if (gd.z > 65535) return SBA_E_ARG;
            if (!det_begin(sp)) return SBA_E_ARG;
            float* dwa = part ? part : dw;
            const int md = part ? 2 : (sp > 1 ? 1 : fw);
            const int64_t zs = part ? dwn : 0;
            if (dma == 3) {
                constexpr int LDS = 3 * 5 * 32 * 128;
                static bool once = false;
                if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_small_dma_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
                SBA_LAUNCH((wgrad_small_dma_kernel<1, 3>), gd, dim3(256), LDS, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)dy, dwa, *g, M, cps32, md, dsub, dow, zs);
            } else {
                constexpr int LDS = 4 * 5 * 32 * 128;
                static bool once = false;
                if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_small_dma_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
                SBA_LAUNCH((wgrad_small_dma_kernel<1, 4>), gd, dim3(256), LDS, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)dy, dwa, *g, M, cps32, md, dsub, dow, zs);
            }
            det_end(sp);
            return SBA_CHECK_LAUNCH();
        }
        if (!det_begin(split)) return SBA_E_ARG;
        SBA_DISPATCH(dtype, SBA_LAUNCH((wgrad_small_kernel<T, 1>), grid, dim3(256), 0, (hipStream_t)stream,
                                               (const T*)x, (const T*)dy, part ? part : dw, *g, M, cps,
                                               part ? 2 : (split > 1 ? 1 : fw), dsub, dow, part ? dwn : (int64_t)0));
        det_end(split);
        return SBA_CHECK_LAUNCH();
    }
    // generator-style 3x3 stride-1 conv on a wide map: all nine taps per workgroup from halo tiles.  From 128x128 maps
    // up (B = 20: M >= 327 k) it beats the LDS-DMA decomposition below (upBlock -> 256 px 189 vs 349 us); at 64x64
    // (M = 82 k) the DMA kernel wins (ResBlock 51 -> 40 us, 64->128: 69 -> 60), tools/bench_wgrad.py.
    static int rows_m = -1;
    if (rows_m < 0) { const char* e = getenv("SBA_WGRAD_ROWS_M"); rows_m = e ? atoi(e) : 131072; }
    bool rows_ok = g->ntaps == 9 && g->sy == 1 && g->sx == 1 && g->osy == 1 && g->osx == 1 && g->ooy == 0 &&
                   g->oox == 0 && g->OHs == g->OH && g->OWs == g->OW && g->OW % 64 == 0 && M >= rows_m;
    for (int t = 0; t < 9 && rows_ok; ++t) rows_ok = g->ty[t] == t / 3 - 1 && g->tx[t] == t % 3 - 1;
    static int rows_en = -1;        // SBA_WGRAD_ROWS=0: skip the all-taps halo-row kernel (A/B aid)
    if (rows_en < 0) { const char* e = getenv("SBA_WGRAD_ROWS"); rows_en = (e && e[0] == '0') ? 0 : 1; }
    if (rows_ok && rows_en) {
        const int total_segs = g->N * g->OH * (g->OW / 64);
        const int ci_t = cdiv(g->Cin, 64);
        // every pixel split adds a full copy of the tile's 9 x 64 x 64 outputs to the f32 atomics
        // (~1.3 TB/s chip-wide), so use few, fat workgroups: ~1 per CU and >= 16 segments each
        int nz = cdiv(256, co_tiles * ci_t);
        if (nz > total_segs / 16) nz = total_segs / 16;
        if (nz < 1) nz = 1;
        const int spw = cdiv(total_segs, nz);
        nz = cdiv(total_segs, spw);
        dim3 grid(co_tiles, ci_t, nz);
        if (grid.z > 65535) return SBA_E_ARG;
        // (deterministic mode: also for nz == 1 -- every workgroup's three waves store, nothing adds)
        part = nullptr;
        if (sba_det_on()) { part = sba_det_alloc((int64_t)nz * dwn); if (!part) return SBA_E_ARG; }
        SBA_DISPATCH(dtype, SBA_LAUNCH((wgrad_rows_kernel<T>), grid, dim3(192), 0, (hipStream_t)stream,
                                               (const T*)x, (const T*)dy, part ? part : dw, *g, total_segs, spw,
                                               part ? 1 : 0, part ? dwn : (int64_t)0));
        det_end(nz);
        return SBA_CHECK_LAUNCH();
    }
    {
        // Big-M layers that are not 3x3 / OW % 64 == 0 (the discriminators' 4x4/s2 down blocks at 32..128 px): the
        // register-staged kernel below shares nothing between its waves (1 KB of operands per MFMA from L2 = the
        // 300 TFLOP/s on-chip-bandwidth roofline of a 64x64 tile); the small-pixel-count decomposition shares the dy
        // slices between the four (tap, ci tile) items of a workgroup (0.625 KB, CT = 2: 0.375 KB per MFMA) and walks
        // its pixel split through the LDS-DMA ring.  Measured (tools/bench_wgrad.py, B = 20): D256 down 64->128 @128 px
        // 200 -> 110 us, 128->256 @64 193 -> 107, D128 down @64 99 -> 48, D64 down @32 43 -> 24; CT = 2 is no better.
        // SBA_WGRAD_GEN_DMA: 0 = off, 1 = CT 1 (default), 2 = CT 2 where Cout % 128 == 0.
        static int gen = -1, gen_wgs = -1;
        if (gen < 0) { const char* e = getenv("SBA_WGRAD_GEN_DMA"); gen = e ? atoi(e) : 1; }
        if (gen_wgs < 0) { const char* e = getenv("SBA_WGRAD_GEN_WGS"); gen_wgs = e ? atoi(e) : 512; }
        const int64_t xb = (int64_t)g->N * g->IH * g->IW * g->Cin * 2, db = (int64_t)g->N * g->OH * g->OW * g->Cout * 2;
        if (gen > 0 && dtype == SBA_BF16 && xb < (1ll << 32) && db < (1ll << 32)) {
            const bool ct2 = gen >= 2 && g->Cout % 128 == 0;
            const int cot = ct2 ? co_tiles / 2 : co_tiles;
            const int wgs = cot * cdiv(items, 4);
            const int tc32 = cdiv(M, 32);
            int sp = cdiv(gen_wgs, wgs);
            if (sp > tc32 / 12) sp = tc32 / 12 > 0 ? tc32 / 12 : 1;
            const int cps32 = cdiv(tc32, sp);
            sp = cdiv(tc32, cps32);
            dim3 gd(cot, cdiv(items, 4), sp);
            if (gd.y <= 65535 && gd.z <= 65535) {
                if (!det_begin(sp)) return SBA_E_ARG;
                float* dwa = part ? part : dw;
                const int md = part ? 2 : (sp > 1 ? 1 : fw);
                const int64_t zs = part ? dwn : 0;
                if (ct2) {
                    constexpr int LDS = 3 * 6 * 32 * 128;
                    static bool once = false;
                    if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_small_dma_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
                    SBA_LAUNCH((wgrad_small_dma_kernel<2, 3>), gd, dim3(256), LDS, (hipStream_t)stream, (const bf16_t*)x,
                               (const bf16_t*)dy, dwa, *g, M, cps32, md, dsub, dow, zs);
                } else {
                    constexpr int LDS = 4 * 5 * 32 * 128;
                    static bool once = false;
                    if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_small_dma_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); once = true; }
                    SBA_LAUNCH((wgrad_small_dma_kernel<1, 4>), gd, dim3(256), LDS, (hipStream_t)stream, (const bf16_t*)x,
                               (const bf16_t*)dy, dwa, *g, M, cps32, md, dsub, dow, zs);
                }
                det_end(sp);
                return SBA_CHECK_LAUNCH();
            }
        }
    }
    const int total_chunks = cdiv(M, 64);
    if (ksplit > total_chunks) ksplit = total_chunks;
    const int cps = cdiv(total_chunks, ksplit);
    ksplit = cdiv(total_chunks, cps);
    dim3 grid(co_tiles, items, ksplit);
    if (grid.y > 65535 || grid.z > 65535) return SBA_E_ARG;
    if (!det_begin(ksplit)) return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((wgrad_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream,
                                           (const T*)x, (const T*)dy, part ? part : dw, *g, M, cps,
                                           part ? 2 : (ksplit > 1 ? 1 : 0), dsub, dow, part ? dwn : (int64_t)0));
    det_end(ksplit);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_pack_weight(int dtype, const float* w, void* out, int Cout, int KH, int KW, int Cin,
                               int mode, void* stream) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || mode < 0 || mode > 3) return SBA_E_ARG;
    if (mode == 3 && (KH != 3 || KW != 3)) return SBA_E_ARG;
    if (mode == 2 && (KH != 4 || KW != 4)) return SBA_E_ARG;
    if (mode == 1 || mode == 2) {
        dim3 grid(cdiv(Cin, 32), cdiv(Cout, 32), mode == 1 ? KH * KW : 16);
        if (grid.y > 65535) return SBA_E_ARG;
        SBA_DISPATCH(dtype, SBA_LAUNCH((pack_weight_tr_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream,
                                               w, (T*)out, Cout, KH, KW, Cin, mode));
        return SBA_CHECK_LAUNCH();
    }
    const int64_t n = (int64_t)Cout * (mode == 3 ? 16 : KH * KW) * Cin;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    SBA_DISPATCH(dtype, SBA_LAUNCH((pack_weight_kernel<T>), dim3(blocks), dim3(256), 0,
                                           (hipStream_t)stream, w, (T*)out, Cout, KH, KW, Cin, mode));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_pack_weights_multi(int dtype, const sba_pack_desc* descs, int ndesc, int total_tiles,
                                      void* stream) {
    if (!descs || ndesc <= 0 || total_tiles <= 0) return SBA_E_ARG;
    if (((uintptr_t)descs & 7) != 0) return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((pack_multi_kernel<T>), dim3(total_tiles), dim3(256), 0,
                                           (hipStream_t)stream, descs, ndesc));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_pack_frag_multi(const sba_frag_desc* descs, int ndesc, int total_units, void* stream) {
    if (!descs || ndesc <= 0 || total_units <= 0 || ((uintptr_t)descs & 7) != 0) return SBA_E_ARG;
    SBA_LAUNCH(pack_frag_kernel, dim3(cdiv(total_units, 256)), dim3(256), 0, (hipStream_t)stream, descs, ndesc, total_units);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_pool2x2_sum(int dtype, const void* dup, void* dx, int N, int H, int W, int C, void* stream) {
    if (!dup || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0) return SBA_E_ARG;
    const int64_t total = (int64_t)N * H * W * (C / (dtype == SBA_BF16 ? 8 : 4));
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    SBA_DISPATCH(dtype, SBA_LAUNCH((pool2x2_kernel<T>), dim3(blocks), dim3(256), 0,
                                           (hipStream_t)stream, (const T*)dup, (T*)dx, N, H, W, C));
    return SBA_CHECK_LAUNCH();
}
