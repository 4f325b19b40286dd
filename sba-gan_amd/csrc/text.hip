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
