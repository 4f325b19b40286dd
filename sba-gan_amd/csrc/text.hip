// RNN_ENCODER forward (model.py:75-159) for the frozen text encoder of the GAN step (trainer.py:248-252):
// Embedding -> (eval-mode dropout = identity) -> one-layer bidirectional LSTM over PACKED sequences ->
// words_emb [B][2H][L] (zeros past each caption's length, as pad_packed_sequence leaves them) and
// sent_emb [B][2H] (the last valid hidden state of each direction).  Caption lengths are read ON THE
// DEVICE: the reference's cap_lens.tolist() host sync (model.py:139) is gone, so the encode of the next
// batch can be queued behind the generator's gradient exchange.
//
// Two launches:
//   lstm_xproj_kernel   gx[dir][b][t][4H] = W_ih[dir] . emb[captions[b][t]] + b_ih[dir] + b_hh[dir]
//                       (all time steps at once: the only part with parallelism over t)
//   lstm_recur_kernel   one workgroup per (caption, direction), one thread per gate row; the thread keeps
//                       its W_hh row (H floats) in REGISTERS for the whole sequence, h lives in LDS and is
//                       read as broadcast float4; two barriers per time step.
// f32 throughout (the reference runs this encoder in f32; 1e-5 parity with torch.nn.LSTM).
#include "common.h"

namespace {

constexpr int XR = 8;          // (caption, step) rows per workgroup of the input projection

// grid (ceil(B*T / XR), 2 * ceil(4H / 256)), block 256: thread = one gate row of one direction
__global__ __launch_bounds__(256) void lstm_xproj_kernel(const int64_t* __restrict__ captions,
                                                         const float* __restrict__ emb,
                                                         const float* __restrict__ w_ih,
                                                         const float* __restrict__ b_ih,
                                                         const float* __restrict__ b_hh, float* __restrict__ gx,
                                                         int BT, int ntoken, int ninput, int G4) {
    extern __shared__ float s_x[];                       // [XR][ninput]
    const int nb = (G4 + 255) / 256;
    const int dir = blockIdx.y / nb, j = (blockIdx.y - dir * nb) * 256 + threadIdx.x;
    const int r0 = blockIdx.x * XR;
    for (int i = threadIdx.x; i < XR * ninput; i += 256) {
        const int r = i / ninput, k = i - r * ninput;
        float v = 0.f;
        if (r0 + r < BT) {
            int64_t tok = captions[r0 + r];
            if (tok < 0 || tok >= ntoken) tok = 0;       // (nn.Embedding would raise; stay in bounds)
            v = emb[tok * ninput + k];
        }
        s_x[i] = v;
    }
    __syncthreads();
    if (j >= G4) return;
    const float* wr = w_ih + ((int64_t)dir * G4 + j) * ninput;
    float acc[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) acc[r] = 0.f;
    for (int k = 0; k < ninput; k += 4) {                // ninput % 4 == 0 (checked by the wrapper)
        const float4 w = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const float4 x = *reinterpret_cast<const float4*>(s_x + r * ninput + k);
            acc[r] += w.x * x.x + w.y * x.y + w.z * x.z + w.w * x.w;
        }
    }
    const float bias = b_ih[dir * G4 + j] + b_hh[dir * G4 + j];
#pragma unroll
    for (int r = 0; r < XR; ++r)
        if (r0 + r < BT) gx[((int64_t)dir * BT + r0 + r) * G4 + j] = acc[r] + bias;
}

// grid (B, 2), block 4H (<= 1024): thread j = gate row j (PyTorch order i | f | g | o)
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_recur_kernel(const float* __restrict__ gx,
                                                           const float* __restrict__ w_hh,
                                                           const int64_t* __restrict__ cap_lens,
                                                           const float* __restrict__ h0, const float* __restrict__ c0,
                                                           float* __restrict__ words, float* __restrict__ sent,
                                                           int B, int T, int Lout) {
    constexpr int G4 = 4 * H;
    __shared__ __attribute__((aligned(16))) float s_h[H];
    __shared__ float s_g[G4];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    int len = (int)cap_lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    float w[H];
    {
        const float* wr = w_hh + ((int64_t)dir * G4 + j) * H;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(wr + k);
            w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
        }
    }
    float c = 0.f, h = 0.f;
    if (j < H) {
        h = h0 ? h0[((int64_t)dir * B + b) * H + j] : 0.f;
        c = c0 ? c0[((int64_t)dir * B + b) * H + j] : 0.f;
        s_h[j] = h;
    }
    // zero padding past the caption (pad_packed_sequence) -- this direction's half of the channels
    for (int i = j; i < H * (Lout - (len < Lout ? len : Lout)); i += G4) {
        const int span = Lout - len;
        const int ch = i / span, t = len + (i - ch * span);
        words[((int64_t)b * 2 * H + dir * H + ch) * Lout + t] = 0.f;
    }
    __syncthreads();
    const float* gxb = gx + ((int64_t)dir * B + b) * T * G4;
    for (int s = 0; s < len; ++s) {
        const int t = dir == 0 ? s : len - 1 - s;
        float a = gxb[(int64_t)t * G4 + j];
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(s_h + k);
            a += w[k] * hv.x + w[k + 1] * hv.y + w[k + 2] * hv.z + w[k + 3] * hv.w;
        }
        s_g[j] = a;
        __syncthreads();
        if (j < H) {
            const float ig = 1.f / (1.f + expf(-s_g[j]));
            const float fg = 1.f / (1.f + expf(-s_g[H + j]));
            const float gg = tanhf(s_g[2 * H + j]);
            const float og = 1.f / (1.f + expf(-s_g[3 * H + j]));
            c = fg * c + ig * gg;
            h = og * tanhf(c);
            s_h[j] = h;
            if (t < Lout) words[((int64_t)b * 2 * H + dir * H + j) * Lout + t] = h;
        }
        __syncthreads();
    }
    if (j < H) sent[(int64_t)b * 2 * H + dir * H + j] = h;
}

// ---------------------------------------------------------------------------
// Training path of the same encoder (DAMSM pre-training, pretrain_DAMSM.py:49-130: the text encoder is trained
// with the words / sentence losses, loss.backward() -> clip_grad_norm -> Adam).
//   lstm_recur_train_kernel: the recurrence above that also keeps what back-propagation through time needs:
//       gates[dir][b][t][4H] (i, f, g, o AFTER their nonlinearities), cs[dir][b][t][H], hs[dir][b][t][H]
//   lstm_recur_bwd_kernel:   BPTT of one (caption, direction) per workgroup, walking the time steps in
//       reverse processing order: dh = d words[:, :, t] + W_hh^T . dgates(next) (+ d sent at the last step),
//       dc likewise; writes dG[dir][b][t][4H] (gradient w.r.t. the gate PRE-activations, zeros past the
//       caption) and hprev[dir][b][t][H] (the hidden state each step started from).  Thread (k, q) keeps the
//       W_hh^T slice {W_hh[q*H + jj][k]}_jj in registers, so W_hh^T . dgates is four partial sums per k.
// The dense parts around the recurrence are plain GEMMs (input projection, dW_ih = dG^T X, dW_hh = dG^T hprev,
// dX = dG W_ih) and go through the BLAS library on the host side (sbagan/ops.py LstmBidirTrainFn).
// ---------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_recur_train_kernel(const float* __restrict__ gx,
                                                                 const float* __restrict__ w_hh,
                                                                 const int64_t* __restrict__ cap_lens,
                                                                 const float* __restrict__ h0, const float* __restrict__ c0,
                                                                 float* __restrict__ words, float* __restrict__ sent,
                                                                 float* __restrict__ gates, float* __restrict__ cs,
                                                                 float* __restrict__ hs, int B, int T, int Lout) {
    constexpr int G4 = 4 * H;
    __shared__ __attribute__((aligned(16))) float s_h[H];
    __shared__ float s_g[G4];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    int len = (int)cap_lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    float w[H];
    {
        const float* wr = w_hh + ((int64_t)dir * G4 + j) * H;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(wr + k);
            w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
        }
    }
    float c = 0.f, h = 0.f;
    if (j < H) {
        h = h0 ? h0[((int64_t)dir * B + b) * H + j] : 0.f;
        c = c0 ? c0[((int64_t)dir * B + b) * H + j] : 0.f;
        s_h[j] = h;
    }
    for (int i = j; i < H * (Lout - (len < Lout ? len : Lout)); i += G4) {
        const int span = Lout - len;
        const int ch = i / span, t = len + (i - ch * span);
        words[((int64_t)b * 2 * H + dir * H + ch) * Lout + t] = 0.f;
    }
    __syncthreads();
    const int64_t row0 = ((int64_t)dir * B + b) * T;
    const float* gxb = gx + row0 * G4;
    for (int s = 0; s < len; ++s) {
        const int t = dir == 0 ? s : len - 1 - s;
        float a = gxb[(int64_t)t * G4 + j];
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(s_h + k);
            a += w[k] * hv.x + w[k + 1] * hv.y + w[k + 2] * hv.z + w[k + 3] * hv.w;
        }
        // thread j applies its own gate's nonlinearity (i | f | g | o blocks of H rows)
        const float act = (j >= 2 * H && j < 3 * H) ? tanhf(a) : 1.f / (1.f + expf(-a));
        s_g[j] = act;
        gates[(row0 + t) * G4 + j] = act;
        __syncthreads();
        if (j < H) {
            c = s_g[H + j] * c + s_g[j] * s_g[2 * H + j];
            h = s_g[3 * H + j] * tanhf(c);
            s_h[j] = h;
            cs[(row0 + t) * H + j] = c;
            hs[(row0 + t) * H + j] = h;
            if (t < Lout) words[((int64_t)b * 2 * H + dir * H + j) * Lout + t] = h;
        }
        __syncthreads();
    }
    if (j < H) sent[(int64_t)b * 2 * H + dir * H + j] = h;
}

template <int H>
__global__ __launch_bounds__(4 * H) void lstm_recur_bwd_kernel(const float* __restrict__ w_hh,
                                                               const int64_t* __restrict__ cap_lens,
                                                               const float* __restrict__ h0, const float* __restrict__ c0,
                                                               const float* __restrict__ gates, const float* __restrict__ cs,
                                                               const float* __restrict__ hs,
                                                               const float* __restrict__ dwords,
                                                               const float* __restrict__ dsent, float* __restrict__ dG,
                                                               float* __restrict__ hprev, int B, int T, int Lout) {
    constexpr int G4 = 4 * H;
    __shared__ float s_dg[G4];
    __shared__ float s_part[4][H];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int k = tid % H, q = tid / H;
    int len = (int)cap_lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    float wt[H];                    // W_hh[q*H + jj][k], jj = 0..H-1
    {
        const float* wc = w_hh + ((int64_t)dir * G4 + (int64_t)q * H) * H + k;
#pragma unroll
        for (int jj = 0; jj < H; ++jj) wt[jj] = wc[(int64_t)jj * H];
    }
    const int64_t row0 = ((int64_t)dir * B + b) * T;
    // rows past the caption carry no gradient (the host zero-fills dG / hprev once; nothing to do here)
    float dh_rec = 0.f, dc_next = 0.f;          // (threads q == 0 own channel k)
    for (int s = len - 1; s >= 0; --s) {
        const int t = dir == 0 ? s : len - 1 - s;
        const int tp = dir == 0 ? t - 1 : t + 1;        // the step processed before this one
        const bool first = s == 0;
        if (q == 0) {
            float dh = dh_rec;
            if (t < Lout) dh += dwords[((int64_t)b * 2 * H + dir * H + k) * Lout + t];
            if (s == len - 1) dh += dsent[(int64_t)b * 2 * H + dir * H + k];
            const float* gt = gates + (row0 + t) * G4;
            const float ig = gt[k], fg = gt[H + k], gg = gt[2 * H + k], og = gt[3 * H + k];
            const float c = cs[(row0 + t) * H + k];
            const float cp = first ? (c0 ? c0[((int64_t)dir * B + b) * H + k] : 0.f) : cs[(row0 + tp) * H + k];
            const float hp = first ? (h0 ? h0[((int64_t)dir * B + b) * H + k] : 0.f) : hs[(row0 + tp) * H + k];
            const float tc = tanhf(c);
            const float dc = dh * og * (1.f - tc * tc) + dc_next;
            s_dg[k] = dc * gg * ig * (1.f - ig);
            s_dg[H + k] = dc * cp * fg * (1.f - fg);
            s_dg[2 * H + k] = dc * ig * (1.f - gg * gg);
            s_dg[3 * H + k] = dh * tc * og * (1.f - og);
            dc_next = dc * fg;
            hprev[(row0 + t) * H + k] = hp;
        }
        __syncthreads();
        dG[(row0 + t) * G4 + tid] = s_dg[tid];
        float p = 0.f;
#pragma unroll
        for (int jj = 0; jj < H; ++jj) p += wt[jj] * s_dg[q * H + jj];
        s_part[q][k] = p;
        __syncthreads();
        if (q == 0) dh_rec = s_part[0][k] + s_part[1][k] + s_part[2][k] + s_part[3][k];
        // (s_dg / s_part are rewritten only after the next iteration's first barrier ... by q == 0 threads, which
        // have passed this point; the other threads read s_dg before the second barrier above)
    }
}

}  // namespace

extern "C" int sba_lstm_bidir_fwd(const int64_t* captions, const int64_t* cap_lens, const float* emb_weight,
                                  const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                  const float* h0, const float* c0, float* gx_scratch, float* words, float* sent,
                                  int B, int T, int Lout, int ntoken, int ninput, int H, void* stream) {
    if (!captions || !cap_lens || !emb_weight || !w_ih || !w_hh || !b_ih || !b_hh || !gx_scratch || !words || !sent)
        return SBA_E_ARG;
    if (B <= 0 || T <= 0 || Lout <= 0 || Lout > T || ntoken <= 0 || ninput <= 0 || ninput % 4) return SBA_E_ARG;
    if ((h0 == nullptr) != (c0 == nullptr)) return SBA_E_ARG;
    if (H != 64 && H != 128) return SBA_E_ARG;                        // nhidden 128 / 256 (cfg.TEXT.EMBEDDING_DIM)
    if ((size_t)XR * ninput * sizeof(float) > 64 * 1024) return SBA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int G4 = 4 * H, BT = B * T;
    SBA_LAUNCH(lstm_xproj_kernel, dim3(cdiv(BT, XR), 2 * cdiv(G4, 256)), dim3(256), XR * ninput * sizeof(float), st,
               captions, emb_weight, w_ih, b_ih, b_hh, gx_scratch, BT, ntoken, ninput, G4);
    if (H == 128)
        SBA_LAUNCH((lstm_recur_kernel<128>), dim3(B, 2), dim3(512), 0, st, gx_scratch, w_hh, cap_lens, h0, c0, words,
                   sent, B, T, Lout);
    else
        SBA_LAUNCH((lstm_recur_kernel<64>), dim3(B, 2), dim3(256), 0, st, gx_scratch, w_hh, cap_lens, h0, c0, words,
                   sent, B, T, Lout);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_lstm_recur_train(const float* gx, const int64_t* cap_lens, const float* w_hh, const float* h0,
                                    const float* c0, float* words, float* sent, float* gates, float* cs, float* hs,
                                    int B, int T, int Lout, int H, void* stream) {
    if (!gx || !cap_lens || !w_hh || !words || !sent || !gates || !cs || !hs) return SBA_E_ARG;
    if (B <= 0 || T <= 0 || Lout <= 0 || Lout > T || (H != 64 && H != 128)) return SBA_E_ARG;
    if ((h0 == nullptr) != (c0 == nullptr)) return SBA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (H == 128)
        SBA_LAUNCH((lstm_recur_train_kernel<128>), dim3(B, 2), dim3(512), 0, st, gx, w_hh, cap_lens, h0, c0, words, sent,
                   gates, cs, hs, B, T, Lout);
    else
        SBA_LAUNCH((lstm_recur_train_kernel<64>), dim3(B, 2), dim3(256), 0, st, gx, w_hh, cap_lens, h0, c0, words, sent,
                   gates, cs, hs, B, T, Lout);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_lstm_recur_bwd(const int64_t* cap_lens, const float* w_hh, const float* h0, const float* c0,
                                  const float* gates, const float* cs, const float* hs, const float* dwords,
                                  const float* dsent, float* dG, float* hprev, int B, int T, int Lout, int H,
                                  void* stream) {
    if (!cap_lens || !w_hh || !gates || !cs || !hs || !dwords || !dsent || !dG || !hprev) return SBA_E_ARG;
    if (B <= 0 || T <= 0 || Lout <= 0 || Lout > T || (H != 64 && H != 128)) return SBA_E_ARG;
    if ((h0 == nullptr) != (c0 == nullptr)) return SBA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (H == 128)
        SBA_LAUNCH((lstm_recur_bwd_kernel<128>), dim3(B, 2), dim3(512), 0, st, w_hh, cap_lens, h0, c0, gates, cs, hs,
                   dwords, dsent, dG, hprev, B, T, Lout);
    else
        SBA_LAUNCH((lstm_recur_bwd_kernel<64>), dim3(B, 2), dim3(256), 0, st, w_hh, cap_lens, h0, c0, gates, cs, hs,
                   dwords, dsent, dG, hprev, B, T, Lout);
    return SBA_CHECK_LAUNCH();
}
