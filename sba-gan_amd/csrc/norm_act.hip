// BatchNorm(train)+activation, BatchNorm1d+GLU, InstanceNorm/AdaIN kernels (gfx950).
// All are HBM-bound streaming kernels: 16-byte vector accesses along the NHWC
// channel axis, f32 math, per-channel reductions through LDS then global atomics.
//
// Reference: model.py:43-44 (BN+GLU), :62-65,70 (ResBlock BN, residual add),
// :543-544,553-554 (BN+LeakyReLU 0.2), :353-356 (BatchNorm1d+GLU), :324-339 (AdaIN).
#include "common.h"

namespace {

constexpr float LRELU_SLOPE = 0.2f;

// per-channel (scale, shift, mean, rstd) of one BatchNorm batch from its (sum, sumsq)
struct BnCoef { float scale, shift, mean, rstd, var; };
__device__ __forceinline__ BnCoef bn_coef(const float* __restrict__ stats, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, int C, int c, float count, float eps) {
    BnCoef k;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int sl = 0; sl < SBA_BN_STAT_SLOTS; ++sl) {            // add the replicas up
        s0 += stats[sl * 2 * C + c];
        s1 += stats[sl * 2 * C + C + c];
    }
    k.mean = s0 / count;
    k.var = fmaxf(s1 / count - k.mean * k.mean, 0.f);
    k.rstd = rsqrtf(k.var + eps);
    k.scale = gamma[c] * k.rstd;
    k.shift = beta[c] - k.mean * k.scale;
    return k;
}

// ---- batch statistics of an NHWC tensor (used when the conv epilogue cannot provide them:
//      grouped passes, where one conv launch covers several BatchNorm batches); blockIdx.y = group ----
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ y_all, float* __restrict__ stats_all, int64_t rows, int C,
                                                       float* __restrict__ det_part) {
    constexpr int V = Vec16<T>::N;
    const T* y = y_all + (int64_t)blockIdx.y * rows * C;
    float* stats = stats_all + ((int64_t)blockIdx.y * SBA_BN_STAT_SLOTS + (blockIdx.x & (SBA_BN_STAT_SLOTS - 1))) * 2 * C;
    const int cv = C / V;
    extern __shared__ float s_acc[];                        // [2*C]; deterministic mode: [rows per iteration][2*C]
    if (!det_part) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_acc[i] = 0.f;
        __syncthreads();
    }
    const int tpr = cv < (int)blockDim.x ? cv : (int)blockDim.x;
    const int rpi = blockDim.x / tpr;           // (cv not a power of two: the last blockDim.x % tpr threads idle)
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    for (int cvi = tc; cvi < cv && tr < rpi; cvi += tpr) {
        const int c = cvi * V;
        float s0[V], s1[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s0[k] = s1[k] = 0.f;
        for (int64_t row = (int64_t)blockIdx.x * rpi + tr; row < rows; row += (int64_t)gridDim.x * rpi) {
            Vec16<T> a = ld16(y + row * C + c);
#pragma unroll
            for (int k = 0; k < V; ++k) { const float v = a.get(k); s0[k] += v; s1[k] += v * v; }
        }
        if (det_part) {                                     // private LDS row per row-group: no atomics
#pragma unroll
            for (int k = 0; k < V; ++k) {
                s_acc[tr * 2 * C + c + k] = s0[k];
                s_acc[tr * 2 * C + C + c + k] = s1[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                atomicAdd(&s_acc[c + k], s0[k]);
                atomicAdd(&s_acc[C + c + k], s1[k]);
            }
        }
    }
    __syncthreads();
    if (det_part) {         // this workgroup's partial sums, row-groups added in order; sba_det_fold adds the workgroups
        float* part = det_part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float v = 0.f;
            for (int r = 0; r < rpi; ++r) v += s_acc[r * 2 * C + i];
            part[i] = v;
        }
        return;
    }
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(&stats[i], s_acc[i]);
}

// ---- forward: finalize + out = act(y*scale+shift) (+residual) in one launch; blockIdx.y = group.
// Every workgroup derives scale/shift of its group's C channels from the (sum, sumsq) statistics into
// LDS (C rsqrt: noise next to the streaming pass); workgroup (0, g) also stores aux[g] = scale,
// shift, mean, rstd for the backward, and workgroup (0, 0) applies the running-statistics updates
// of all groups in order (momentum, unbiased variance), as consecutive module calls would.
template <typename T, typename YT, int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const YT* __restrict__ y_all, const float* __restrict__ stats_all,
                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                  float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                  float* __restrict__ aux_all, const T* __restrict__ residual_all,
                                  T* __restrict__ out_all, int64_t rows, int C, int out_cstride, int out_coff,
                                  float eps, float momentum, int training) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ float s_co[];                         // scale[C], shift[C]
    const int g = blockIdx.y;
    const int Co = ACT == SBA_ACT_GLU ? C / 2 : C;
    const float count = (float)rows;
    float* aux = aux_all + (int64_t)g * 4 * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        BnCoef k;
        if (training) {
            k = bn_coef(stats_all + (int64_t)g * SBA_BN_STAT_SLOTS * 2 * C, gamma, beta, C, c, count, eps);
        } else {                                            // inference: running statistics
            k.mean = rmean[c];
            k.rstd = rsqrtf(rvar[c] + eps);
            k.scale = gamma[c] * k.rstd;
            k.shift = beta[c] - k.mean * k.scale;
        }
        s_co[c] = k.scale;
        s_co[C + c] = k.shift;
        if (blockIdx.x == 0) {
            aux[c] = k.scale; aux[C + c] = k.shift; aux[2 * C + c] = k.mean; aux[3 * C + c] = k.rstd;
        }
        if (blockIdx.x == 0 && g == 0 && training && rmean) {
            float rm = rmean[c], rv = rvar[c];
            for (int gg = 0; gg < (int)gridDim.y; ++gg) {
                const BnCoef q = bn_coef(stats_all + (int64_t)gg * SBA_BN_STAT_SLOTS * 2 * C, gamma, beta, C, c, count, eps);
                const float unb = count > 1.f ? q.var * count / (count - 1.f) : q.var;
                rm = (1.f - momentum) * rm + momentum * q.mean;
                rv = (1.f - momentum) * rv + momentum * unb;
            }
            rmean[c] = rm;
            rvar[c] = rv;
        }
    }
    if (blockIdx.x == 0 && g == 0 && threadIdx.x == 0 && training && nbt) *nbt += gridDim.y;
    __syncthreads();
    const float* scale = s_co;
    const float* shift = s_co + C;
    const YT* y = y_all + (int64_t)g * rows * C;
    const T* residual = residual_all ? residual_all + (int64_t)g * rows * Co : nullptr;
    T* out = out_all + (int64_t)g * rows * out_cstride;
    const int cv = Co / V;
    const int64_t total = rows * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cv;
        const int c = (int)(i - row * cv) * V;
        Vec16<YT> a = ld16(y + row * C + c);
        Vec16<T> o;
        if (ACT == SBA_ACT_GLU) {
            Vec16<YT> gt = ld16(y + row * C + Co + c);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float n = a.get(k) * scale[c + k] + shift[c + k];
                const float gp = gt.get(k) * scale[Co + c + k] + shift[Co + c + k];
                o.set(k, n * sigmoidf_(gp));
            }
        } else {
            Vec16<T> r;
            if (residual) r = ld16(residual + row * Co + c);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float n = a.get(k) * scale[c + k] + shift[c + k];
                if (ACT == SBA_ACT_LRELU) n = n > 0.f ? n : LRELU_SLOPE * n;
                if (ACT == SBA_ACT_RELU) n = fmaxf(n, 0.f);
                if (residual) n += r.get(k);
                o.set(k, n);
            }
        }
        st16(out + row * out_cstride + out_coff + c, o);
    }
}

// ---- backward pass 1: per-channel sum(dz), sum(dz*xhat); blockIdx.y = group ----
// thread mapping: each thread keeps a fixed set of channel vectors and strides over rows
template <typename T, typename YT, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const YT* __restrict__ y_all, const T* __restrict__ dout_all,
                                     const float* __restrict__ aux_all, float* __restrict__ red_all,
                                     int64_t rows, int C, int dcs, int dco, float* __restrict__ det_part) {
    constexpr int V = Vec16<T>::N;
    const int g = blockIdx.y;
    const YT* y = y_all + (int64_t)g * rows * C;
    const T* dout = dout_all + (int64_t)g * rows * dcs;
    const float* scale = aux_all + (int64_t)g * 4 * C;
    const float* shift = scale + C;
    const float* mean = scale + 2 * C;
    const float* rstd = scale + 3 * C;
    // one of SBA_BN_STAT_SLOTS replicas per workgroup (same-address f32 atomics serialise at the memory side)
    float* red = red_all + ((int64_t)g * SBA_BN_STAT_SLOTS + (blockIdx.x & (SBA_BN_STAT_SLOTS - 1))) * 2 * C;
    const int Co = ACT == SBA_ACT_GLU ? C / 2 : C;
    const int cv = Co / V;                                  // power of two
    extern __shared__ float s_acc[];                        // [2*C]; deterministic mode: [rows per iteration][2*C]
    if (!det_part) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_acc[i] = 0.f;
        __syncthreads();
    }
    const int tpr = cv < (int)blockDim.x ? cv : (int)blockDim.x;   // threads per row
    const int rpi = blockDim.x / tpr;                               // rows per iteration
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    for (int cvi = tc; cvi < cv; cvi += tpr) {
        const int c = cvi * V;
        float s0[V], s1[V], g0[V], g1[V];
        float sc[V], sh[V], mn[V], rs[V], scg[V], shg[V], mng[V], rsg[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            s0[k] = s1[k] = g0[k] = g1[k] = 0.f;
            sc[k] = scale[c + k]; sh[k] = shift[c + k]; mn[k] = mean[c + k]; rs[k] = rstd[c + k];
            if (ACT == SBA_ACT_GLU) {
                scg[k] = scale[Co + c + k]; shg[k] = shift[Co + c + k];
                mng[k] = mean[Co + c + k]; rsg[k] = rstd[Co + c + k];
            }
        }
#pragma unroll 2
        for (int64_t row = (int64_t)blockIdx.x * rpi + tr; row < rows; row += (int64_t)gridDim.x * rpi) {
            Vec16<YT> a = ld16(y + row * C + c);
            Vec16<T> d = ld16(dout + row * dcs + dco + c);
            if (ACT == SBA_ACT_GLU) {
                Vec16<YT> gt = ld16(y + row * C + Co + c);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float n = a.get(k) * sc[k] + sh[k];
                    const float gp = gt.get(k) * scg[k] + shg[k];
                    const float s = sigmoid_bwd_(gp), dd = d.get(k);
                    const float dza = dd * s, dzg = dd * n * s * (1.f - s);
                    s0[k] += dza;
                    s1[k] += dza * (a.get(k) - mn[k]) * rs[k];
                    g0[k] += dzg;
                    g1[k] += dzg * (gt.get(k) - mng[k]) * rsg[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float dz = d.get(k);
                    if (ACT == SBA_ACT_LRELU) {
                        const float n = a.get(k) * sc[k] + sh[k];
                        dz = n > 0.f ? dz : LRELU_SLOPE * dz;
                    }
                    s0[k] += dz;
                    s1[k] += dz * (a.get(k) - mn[k]) * rs[k];
                }
            }
        }
        if (det_part) {
            float* row = s_acc + tr * 2 * C;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                row[c + k] = s0[k];
                row[C + c + k] = s1[k];
                if (ACT == SBA_ACT_GLU) {
                    row[Co + c + k] = g0[k];
                    row[C + Co + c + k] = g1[k];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                atomicAdd(&s_acc[c + k], s0[k]);
                atomicAdd(&s_acc[C + c + k], s1[k]);
                if (ACT == SBA_ACT_GLU) {
                    atomicAdd(&s_acc[Co + c + k], g0[k]);
                    atomicAdd(&s_acc[C + Co + c + k], g1[k]);
                }
            }
        }
    }
    __syncthreads();
    if (det_part) {
        float* part = det_part + ((int64_t)g * gridDim.x + blockIdx.x) * 2 * C;
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float v = 0.f;
            for (int r = 0; r < rpi; ++r) v += s_acc[r * 2 * C + i];
            part[i] = v;
        }
        return;
    }
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(&red[i], s_acc[i]);
}

// ---- backward pass 2: dy = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)); blockIdx.y = group ----
// per-channel coefficients live in LDS: A = scale, B = shift, M = mean, R = rstd, P = red0/rows, Q = red1/rows
template <typename T, typename YT, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const YT* __restrict__ y_all, const T* __restrict__ dout_all,
                                    const float* __restrict__ aux_all, const float* __restrict__ red_all,
                                    T* __restrict__ dy_all, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    int64_t rows, int C, int dcs, int dco) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ float s_k[];                          // [6][C]
    const int g = blockIdx.y;
    const float* aux = aux_all + (int64_t)g * 4 * C;
    const float* red = red_all + (int64_t)g * SBA_BN_STAT_SLOTS * 2 * C;
    const float inv = 1.f / (float)rows;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        s_k[c] = aux[c]; s_k[C + c] = aux[C + c]; s_k[2 * C + c] = aux[2 * C + c]; s_k[3 * C + c] = aux[3 * C + c];
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int sl = 0; sl < SBA_BN_STAT_SLOTS; ++sl) {    // add the replicas up
            r0 += red[(int64_t)sl * 2 * C + c];
            r1 += red[(int64_t)sl * 2 * C + C + c];
        }
        s_k[4 * C + c] = r0 * inv;
        s_k[5 * C + c] = r1 * inv;
        if (blockIdx.x == 0 && dgamma) {                    // groups accumulate into the same parameter
            atomicAdd(&dgamma[c], r1);
            atomicAdd(&dbeta[c], r0);
        }
    }
    __syncthreads();
    const float* scale = s_k;
    const float* shift = s_k + C;
    const float* mean = s_k + 2 * C;
    const float* rstd = s_k + 3 * C;
    const float* P = s_k + 4 * C;
    const float* Q = s_k + 5 * C;
    const YT* y = y_all + (int64_t)g * rows * C;
    const T* dout = dout_all + (int64_t)g * rows * dcs;
    T* dy = dy_all + (int64_t)g * rows * C;
    const int Co = ACT == SBA_ACT_GLU ? C / 2 : C;
    const int cv = Co / V;
    const int64_t total = rows * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cv;
        const int c = (int)(i - row * cv) * V;
        Vec16<YT> a = ld16(y + row * C + c);
        Vec16<T> d = ld16(dout + row * dcs + dco + c);
        Vec16<T> o;
        if (ACT == SBA_ACT_GLU) {
            Vec16<YT> gt = ld16(y + row * C + Co + c);
            Vec16<T> og;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int ca = c + k, cg = Co + c + k;
                const float n = a.get(k) * scale[ca] + shift[ca];
                const float gp = gt.get(k) * scale[cg] + shift[cg];
                const float s = sigmoid_bwd_(gp), dd = d.get(k);
                const float dza = dd * s, dzg = dd * n * s * (1.f - s);
                const float xa = (a.get(k) - mean[ca]) * rstd[ca];
                const float xg = (gt.get(k) - mean[cg]) * rstd[cg];
                o.set(k, scale[ca] * (dza - P[ca] - xa * Q[ca]));
                og.set(k, scale[cg] * (dzg - P[cg] - xg * Q[cg]));
            }
            st16(dy + row * C + Co + c, og);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int ca = c + k;
                float dz = d.get(k);
                if (ACT == SBA_ACT_LRELU) {
                    const float n = a.get(k) * scale[ca] + shift[ca];
                    dz = n > 0.f ? dz : LRELU_SLOPE * dz;
                }
                const float xa = (a.get(k) - mean[ca]) * rstd[ca];
                o.set(k, scale[ca] * (dz - P[ca] - xa * Q[ca]));
            }
        }
        st16(dy + row * C + c, o);
    }
}

// ---- forward of BatchNorm(train) + activation for SMALL tensors whose statistics are not available from
// a conv epilogue (grouped real|fake passes): statistics, finalize, running-stat update and normalise in
// ONE launch.  A workgroup owns V channels (+ their V gate channels for GLU) over all rows and walks
// the groups in order, so the running statistics see the same sequence as consecutive module calls.
template <typename T, typename YT, int ACT>
__global__ __launch_bounds__(256) void bn_fwd_fused_kernel(const YT* __restrict__ y_all, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ rmean,
                                                           float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                                           float* __restrict__ aux_all, T* __restrict__ out_all,
                                                           int64_t rows, int groups, int C, int out_cstride,
                                                           int out_coff, float eps, float momentum) {
    constexpr int V = Vec16<T>::N;
    constexpr int NV = ACT == SBA_ACT_GLU ? 2 : 1;
    __shared__ float s_red[4][2 * NV * V];
    __shared__ float s_co[2 * NV * V];                      // scale, shift per owned channel
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Co = ACT == SBA_ACT_GLU ? C / 2 : C;
    const int c = blockIdx.x * V;
    const float count = (float)rows;
    if (blockIdx.x == 0 && tid == 0 && nbt) *nbt += groups;
    for (int g = 0; g < groups; ++g) {
        const YT* y = y_all + (int64_t)g * rows * C;
        T* out = out_all + (int64_t)g * rows * out_cstride;
        float* aux = aux_all + (int64_t)g * 4 * C;
        float s0[NV][V], s1[NV][V];
#pragma unroll
        for (int h = 0; h < NV; ++h)
#pragma unroll
            for (int k = 0; k < V; ++k) { s0[h][k] = 0.f; s1[h][k] = 0.f; }
        for (int64_t row = tid; row < rows; row += 256) {
#pragma unroll
            for (int h = 0; h < NV; ++h) {
                Vec16<YT> a = ld16(y + row * C + c + h * Co);
#pragma unroll
                for (int k = 0; k < V; ++k) { const float v = a.get(k); s0[h][k] += v; s1[h][k] += v * v; }
            }
        }
#pragma unroll
        for (int h = 0; h < NV; ++h)
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float a0 = wave_sum(s0[h][k]), a1 = wave_sum(s1[h][k]);
                if (lane == 0) { s_red[wid][(h * V + k) * 2] = a0; s_red[wid][(h * V + k) * 2 + 1] = a1; }
            }
        __syncthreads();
        if (tid < NV * V) {
            const int h = tid / V, k = tid - h * V, ch = c + h * Co + k;
            const float t0 = s_red[0][tid * 2] + s_red[1][tid * 2] + s_red[2][tid * 2] + s_red[3][tid * 2];
            const float t1 = s_red[0][tid * 2 + 1] + s_red[1][tid * 2 + 1] + s_red[2][tid * 2 + 1] + s_red[3][tid * 2 + 1];
            const float mean = t0 / count;
            const float var = fmaxf(t1 / count - mean * mean, 0.f);
            const float rstd = rsqrtf(var + eps);
            const float scale = gamma[ch] * rstd, shift = beta[ch] - mean * scale;
            s_co[tid * 2] = scale;
            s_co[tid * 2 + 1] = shift;
            aux[ch] = scale; aux[C + ch] = shift; aux[2 * C + ch] = mean; aux[3 * C + ch] = rstd;
            if (rmean) {
                const float unb = count > 1.f ? var * count / (count - 1.f) : var;
                rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * mean;
                rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * unb;
            }
        }
        __syncthreads();
        for (int64_t row = tid; row < rows; row += 256) {
            Vec16<YT> a = ld16(y + row * C + c);
            Vec16<T> o;
            if (ACT == SBA_ACT_GLU) {
                Vec16<YT> gt = ld16(y + row * C + Co + c);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float n = a.get(k) * s_co[k * 2] + s_co[k * 2 + 1];
                    const float gp = gt.get(k) * s_co[(V + k) * 2] + s_co[(V + k) * 2 + 1];
                    o.set(k, n * sigmoidf_(gp));
                }
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float n = a.get(k) * s_co[k * 2] + s_co[k * 2 + 1];
                    if (ACT == SBA_ACT_LRELU) n = n > 0.f ? n : LRELU_SLOPE * n;
                    o.set(k, n);
                }
            }
            st16(out + row * out_cstride + out_coff + c, o);
        }
        __syncthreads();                                    // s_red / s_co reused by the next group
    }
}

// ---- backward of BatchNorm + activation for SMALL tensors in ONE launch (the 4x4 .. 16x16 maps of
// the discriminator tails and the generator's first stage): a workgroup owns V channels (and, for GLU,
// their V gate channels) over ALL rows of one group, so both the reduction and the apply pass are
// workgroup-local -- no global accumulator, no second launch.  blockIdx.x = channel vector, .y = group.
template <typename T, typename YT, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_fused_kernel(const YT* __restrict__ y_all, const T* __restrict__ dout_all,
                                                           const float* __restrict__ aux_all, T* __restrict__ dy_all,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int64_t rows, int C, int dcs, int dco) {
    constexpr int V = Vec16<T>::N;
    constexpr int NV = ACT == SBA_ACT_GLU ? 2 : 1;          // channel vectors owned (value [+ gate])
    __shared__ float s_red[4][2 * NV * V];
    __shared__ float s_tot[2 * NV * V];
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Co = ACT == SBA_ACT_GLU ? C / 2 : C;
    const int c = blockIdx.x * V;                           // first owned (value) channel
    const YT* y = y_all + (int64_t)g * rows * C;
    const T* dout = dout_all + (int64_t)g * rows * dcs;
    T* dy = dy_all + (int64_t)g * rows * C;
    const float* aux = aux_all + (int64_t)g * 4 * C;
    float sc[NV][V], sh[NV][V], mn[NV][V], rs[NV][V];
#pragma unroll
    for (int h = 0; h < NV; ++h)
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int ch = c + h * Co + k;
            sc[h][k] = aux[ch]; sh[h][k] = aux[C + ch]; mn[h][k] = aux[2 * C + ch]; rs[h][k] = aux[3 * C + ch];
        }
    float s0[NV][V], s1[NV][V];
#pragma unroll
    for (int h = 0; h < NV; ++h)
#pragma unroll
        for (int k = 0; k < V; ++k) { s0[h][k] = 0.f; s1[h][k] = 0.f; }
    // dz of one row for the owned channels
    auto dz_row = [&](int64_t row, float (&dz)[NV][V], float (&xh)[NV][V]) {
        Vec16<YT> a = ld16(y + row * C + c);
        Vec16<T> d = ld16(dout + row * dcs + dco + c);
        if (ACT == SBA_ACT_GLU) {
            Vec16<YT> gt = ld16(y + row * C + Co + c);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float n = a.get(k) * sc[0][k] + sh[0][k];
                const float gp = gt.get(k) * sc[NV - 1][k] + sh[NV - 1][k];
                const float sg = sigmoid_bwd_(gp), dd = d.get(k);
                dz[0][k] = dd * sg;
                dz[NV - 1][k] = dd * n * sg * (1.f - sg);
                xh[0][k] = (a.get(k) - mn[0][k]) * rs[0][k];
                xh[NV - 1][k] = (gt.get(k) - mn[NV - 1][k]) * rs[NV - 1][k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float z = d.get(k);
                if (ACT == SBA_ACT_LRELU) {
                    const float n = a.get(k) * sc[0][k] + sh[0][k];
                    z = n > 0.f ? z : LRELU_SLOPE * z;
                }
                dz[0][k] = z;
                xh[0][k] = (a.get(k) - mn[0][k]) * rs[0][k];
            }
        }
    };
    for (int64_t row = tid; row < rows; row += 256) {
        float dz[NV][V], xh[NV][V];
        dz_row(row, dz, xh);
#pragma unroll
        for (int h = 0; h < NV; ++h)
#pragma unroll
            for (int k = 0; k < V; ++k) { s0[h][k] += dz[h][k]; s1[h][k] += dz[h][k] * xh[h][k]; }
    }
#pragma unroll
    for (int h = 0; h < NV; ++h)
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float a0 = wave_sum(s0[h][k]), a1 = wave_sum(s1[h][k]);
            if (lane == 0) { s_red[wid][(h * V + k) * 2] = a0; s_red[wid][(h * V + k) * 2 + 1] = a1; }
        }
    __syncthreads();
    if (tid < 2 * NV * V) s_tot[tid] = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    __syncthreads();
    if (tid < NV * V && dgamma) {                           // groups accumulate into the same parameter
        const int h = tid / V, k = tid - h * V;
        atomicAdd(&dgamma[c + h * Co + k], s_tot[tid * 2 + 1]);
        atomicAdd(&dbeta[c + h * Co + k], s_tot[tid * 2]);
    }
    const float inv = 1.f / (float)rows;
    for (int64_t row = tid; row < rows; row += 256) {
        float dz[NV][V], xh[NV][V];
        dz_row(row, dz, xh);
#pragma unroll
        for (int h = 0; h < NV; ++h) {
            Vec16<T> o;
#pragma unroll
            for (int k = 0; k < V; ++k)
                o.set(k, sc[h][k] * (dz[h][k] - s_tot[(h * V + k) * 2] * inv - xh[h][k] * s_tot[(h * V + k) * 2 + 1] * inv));
            st16(dy + row * C + c + h * Co, o);
        }
    }
}

// ---- BatchNorm1d + GLU on [B][F] f32 with the NCHW->NHWC view permutation ----
// feature f' in [0,F/2) pairs with gate f'+F/2; view(B, F/2/16, 4, 4): f' = c*16 + s
template <typename T>
__global__ __launch_bounds__(256) void bn1d_glu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float* __restrict__ rmean,
                                    float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                    float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                    T* __restrict__ out, int B, int F, float eps, float momentum) {
    const int fh = F / 2, Cg = fh / 16;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f == 0 && nbt) *nbt += 1;
    if (f >= fh) return;
    float m[2], r[2], sc[2], sh[2];
    // B <= 32 (the training batch): the 2 B values of this feature pair are loaded ONCE, all loads in flight together (three
    // passes of dependent loads took 27 us for 20 x 16384 values at the head of the generator's forward pass); same sums in
    // the same order
    constexpr int BM = 32;
    float yv[2][BM];
    const bool cached = B <= BM;
    if (cached) {
#pragma unroll
        for (int b = 0; b < BM; ++b) {
            yv[0][b] = b < B ? y[(int64_t)b * F + f] : 0.f;
            yv[1][b] = b < B ? y[(int64_t)b * F + f + fh] : 0.f;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ff = f + h * fh;
        float s = 0.f, q = 0.f;
        if (cached) {
#pragma unroll
            for (int b = 0; b < BM; ++b) if (b < B) s += yv[h][b];
        } else {
            for (int b = 0; b < B; ++b) { const float v = y[(int64_t)b * F + ff]; s += v; }
        }
        const float mean = s / B;
        if (cached) {
#pragma unroll
            for (int b = 0; b < BM; ++b) if (b < B) { const float v = yv[h][b] - mean; q += v * v; }
        } else {
            for (int b = 0; b < B; ++b) { const float v = y[(int64_t)b * F + ff] - mean; q += v * v; }
        }
        const float var = q / B;
        m[h] = mean; r[h] = rsqrtf(var + eps);
        sc[h] = gamma[ff] * r[h]; sh[h] = beta[ff] - mean * sc[h];
        mean_o[ff] = mean; rstd_o[ff] = r[h];
        if (rmean) {
            const float unb = B > 1 ? var * B / (B - 1.f) : var;
            rmean[ff] = (1.f - momentum) * rmean[ff] + momentum * mean;
            rvar[ff] = (1.f - momentum) * rvar[ff] + momentum * unb;
        }
    }
    const int c = f / 16, s16 = f % 16;
    if (cached) {
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (b < B) {
                const float n = yv[0][b] * sc[0] + sh[0], gp = yv[1][b] * sc[1] + sh[1];
                out[((int64_t)b * 16 + s16) * Cg + c] = from_f<T>(n * sigmoidf_(gp));
            }
        return;
    }
    for (int b = 0; b < B; ++b) {
        const float n = y[(int64_t)b * F + f] * sc[0] + sh[0];
        const float gp = y[(int64_t)b * F + f + fh] * sc[1] + sh[1];
        out[((int64_t)b * 16 + s16) * Cg + c] = from_f<T>(n * sigmoidf_(gp));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn1d_glu_bwd_kernel(const float* __restrict__ y, const T* __restrict__ dout,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    float* __restrict__ dy, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int B, int F) {
    const int fh = F / 2, Cg = fh / 16;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= fh) return;
    const int c = f / 16, s16 = f % 16;
    const int fa = f, fg = f + fh;
    const float sca = gamma[fa] * rstd[fa], sha = beta[fa] - mean[fa] * sca;
    const float scg = gamma[fg] * rstd[fg], shg = beta[fg] - mean[fg] * scg;
    float a0 = 0.f, a1 = 0.f, g0 = 0.f, g1 = 0.f;
    constexpr int BM = 32;
    if (B <= BM) {          // one pass of loads, all in flight together (see bn1d_glu_fwd_kernel); same sums, same order
        float ya_[BM], yg_[BM], dd_[BM];
        const float ma = mean[fa], ra = rstd[fa], mg = mean[fg], rg = rstd[fg];
#pragma unroll
        for (int b = 0; b < BM; ++b) {
            ya_[b] = b < B ? y[(int64_t)b * F + fa] : 0.f;
            yg_[b] = b < B ? y[(int64_t)b * F + fg] : 0.f;
            dd_[b] = b < B ? to_f<T>(dout[((int64_t)b * 16 + s16) * Cg + c]) : 0.f;
        }
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (b < B) {
                const float n = ya_[b] * sca + sha, gp = yg_[b] * scg + shg, s = sigmoidf_(gp);
                const float dza = dd_[b] * s, dzg = dd_[b] * n * s * (1.f - s);
                a0 += dza; a1 += dza * (ya_[b] - ma) * ra;
                g0 += dzg; g1 += dzg * (yg_[b] - mg) * rg;
            }
        dgamma[fa] += a1; dbeta[fa] += a0;
        dgamma[fg] += g1; dbeta[fg] += g0;
        const float inv = 1.f / B;
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (b < B) {
                const float n = ya_[b] * sca + sha, gp = yg_[b] * scg + shg, s = sigmoidf_(gp);
                const float dza = dd_[b] * s, dzg = dd_[b] * n * s * (1.f - s);
                dy[(int64_t)b * F + fa] = sca * (dza - a0 * inv - (ya_[b] - ma) * ra * a1 * inv);
                dy[(int64_t)b * F + fg] = scg * (dzg - g0 * inv - (yg_[b] - mg) * rg * g1 * inv);
            }
        return;
    }
    for (int b = 0; b < B; ++b) {
        const float ya = y[(int64_t)b * F + fa], yg = y[(int64_t)b * F + fg];
        const float n = ya * sca + sha, gp = yg * scg + shg, s = sigmoidf_(gp);
        const float dd = to_f<T>(dout[((int64_t)b * 16 + s16) * Cg + c]);
        const float dza = dd * s, dzg = dd * n * s * (1.f - s);
        a0 += dza; a1 += dza * (ya - mean[fa]) * rstd[fa];
        g0 += dzg; g1 += dzg * (yg - mean[fg]) * rstd[fg];
    }
    dgamma[fa] += a1; dbeta[fa] += a0;
    dgamma[fg] += g1; dbeta[fg] += g0;
    const float inv = 1.f / B;
    for (int b = 0; b < B; ++b) {
        const float ya = y[(int64_t)b * F + fa], yg = y[(int64_t)b * F + fg];
        const float n = ya * sca + sha, gp = yg * scg + shg, s = sigmoidf_(gp);
        const float dd = to_f<T>(dout[((int64_t)b * 16 + s16) * Cg + c]);
        const float dza = dd * s, dzg = dd * n * s * (1.f - s);
        dy[(int64_t)b * F + fa] = sca * (dza - a0 * inv - (ya - mean[fa]) * rstd[fa] * a1 * inv);
        dy[(int64_t)b * F + fg] = scg * (dzg - g0 * inv - (yg - mean[fg]) * rstd[fg] * g1 * inv);
    }
}

// ---- InstanceNorm statistics / AdaIN ----
// grid (N, splits); threads [rows][C/V]; accumulates (sum, sumsq) into mean/rstd buffers
template <typename T>
__global__ __launch_bounds__(256) void instnorm_accum_kernel(const T* __restrict__ h, float* __restrict__ sum,
                                      float* __restrict__ sumsq, int HW, int C, int det) {
    constexpr int V = Vec16<T>::N;
    const int cv = C / V, n = blockIdx.x;
    const int rpi = blockDim.x / cv;
    const int tc = threadIdx.x % cv, tr = threadIdx.x / cv;
    extern __shared__ float s_acc[];   // [2*C]; deterministic mode (one workgroup per image): [rpi][2*C]
    if (!det) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_acc[i] = 0.f;
        __syncthreads();
    }
    float s0[V], s1[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s0[k] = s1[k] = 0.f;
    if (tr < rpi) {
        for (int p = blockIdx.y * rpi + tr; p < HW; p += gridDim.y * rpi) {
            Vec16<T> a = ld16(h + ((int64_t)n * HW + p) * C + tc * V);
#pragma unroll
            for (int k = 0; k < V; ++k) { const float v = a.get(k); s0[k] += v; s1[k] += v * v; }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (det) {
                s_acc[tr * 2 * C + tc * V + k] = s0[k];
                s_acc[tr * 2 * C + C + tc * V + k] = s1[k];
            } else {
                atomicAdd(&s_acc[tc * V + k], s0[k]);
                atomicAdd(&s_acc[C + tc * V + k], s1[k]);
            }
        }
    }
    __syncthreads();
    if (det) {              // gridDim.y == 1: this workgroup owns image n; row-groups added in order, plain stores
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float v = 0.f;
            for (int r = 0; r < rpi; ++r) v += s_acc[r * 2 * C + i];
            if (i < C) sum[n * C + i] = v; else sumsq[n * C + i - C] = v;
        }
        return;
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        atomicAdd(&sum[n * C + i], s_acc[i]);
        atomicAdd(&sumsq[n * C + i], s_acc[C + i]);
    }
}

// The same statistics in ONE launch, no atomics, no clearing, no finalize pass: workgroup (n, tc) owns 16-byte channel vector tc
// of image n over all HW pixels -- each thread sums its pixels (four loads in flight), the 256 partial sums meet in a fixed
// order through LDS, thread 0..V-1 write mean and rstd.  Three launches (clear, accumulate, finalize: 28-35 us alone at the
// entry of a generator stage, serial time of the step) become one; deterministic by construction.
template <typename T>
__global__ __launch_bounds__(256) void instnorm_stats_fused_kernel(const T* __restrict__ h, float* __restrict__ mean,
                                                                   float* __restrict__ rstd, int HW, int C, float eps) {
    constexpr int V = Vec16<T>::N;
    const int n = blockIdx.x, tc = blockIdx.y, tid = threadIdx.x;
    const T* base = h + (int64_t)n * HW * C + tc * V;
    float s0[V], s1[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s0[k] = s1[k] = 0.f;
    int p = tid;
    for (; p + 3 * 256 < HW; p += 4 * 256) {
        Vec16<T> a0 = ld16(base + (int64_t)p * C), a1 = ld16(base + (int64_t)(p + 256) * C);
        Vec16<T> a2 = ld16(base + (int64_t)(p + 512) * C), a3 = ld16(base + (int64_t)(p + 768) * C);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float v0 = a0.get(k), v1 = a1.get(k), v2 = a2.get(k), v3 = a3.get(k);
            s0[k] += (v0 + v1) + (v2 + v3);
            s1[k] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
        }
    }
    for (; p < HW; p += 256) {
        Vec16<T> a = ld16(base + (int64_t)p * C);
#pragma unroll
        for (int k = 0; k < V; ++k) { const float v = a.get(k); s0[k] += v; s1[k] += v * v; }
    }
    __shared__ float s_part[4][2 * V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        float a = s0[k], b = s1[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if ((tid & 63) == 0) { s_part[tid >> 6][k] = a; s_part[tid >> 6][V + k] = b; }
    }
    __syncthreads();
    if (tid < V) {
        const float sm = (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
        const float sq = (s_part[0][V + tid] + s_part[1][V + tid]) + (s_part[2][V + tid] + s_part[3][V + tid]);
        const float m = sm / (float)HW;
        const float var = fmaxf(sq / (float)HW - m * m, 0.f);
        mean[n * C + tc * V + tid] = m;
        rstd[n * C + tc * V + tid] = rsqrtf(var + eps);
    }
}

__global__ __launch_bounds__(256) void instnorm_finalize_kernel(float* __restrict__ mean, float* __restrict__ rstd, int NC,
                                         float HW, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    const float m = mean[i] / HW;
    const float var = fmaxf(rstd[i] / HW - m * m, 0.f);
    mean[i] = m;
    rstd[i] = rsqrtf(var + eps);
}

template <typename T>
__global__ __launch_bounds__(256) void adain_fwd_kernel(const T* __restrict__ h, const float* __restrict__ mean,
                                 const float* __restrict__ rstd, const float* __restrict__ style,
                                 T* __restrict__ out, int N, int HW, int C, int ocs, int oco) {
    constexpr int V = Vec16<T>::N;
    const int cv = C / V;
    const int64_t total = (int64_t)N * HW * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cv;
        const int c = (int)(i - row * cv) * V;
        const int n = (int)(row / HW);
        Vec16<T> a = ld16(h + row * C + c), o;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = (a.get(k) - mean[n * C + c + k]) * rstd[n * C + c + k];
            o.set(k, (style[n * 2 * C + c + k] + 1.f) * xh + style[n * 2 * C + C + c + k]);
        }
        st16(out + row * ocs + oco + c, o);
    }
}

// red[n][c][0] += sum dout*xhat, red[n][c][1] += sum dout
template <typename T>
__global__ __launch_bounds__(256) void adain_bwd_reduce_kernel(const T* __restrict__ h, const T* __restrict__ dout,
                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                        float* __restrict__ red, int HW, int C, int dcs, int dco, int det) {
    constexpr int V = Vec16<T>::N;
    const int cv = C / V, n = blockIdx.x;
    const int rpi = blockDim.x / cv;
    const int tc = threadIdx.x % cv, tr = threadIdx.x / cv;
    extern __shared__ float s_acc[];   // [2*C]; deterministic mode (one workgroup per image): [rpi][2*C]
    if (!det) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_acc[i] = 0.f;
        __syncthreads();
    }
    float s0[V], s1[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s0[k] = s1[k] = 0.f;
    if (tr < rpi) {
        for (int p = blockIdx.y * rpi + tr; p < HW; p += gridDim.y * rpi) {
            const int64_t row = (int64_t)n * HW + p;
            Vec16<T> a = ld16(h + row * C + tc * V);
            Vec16<T> d = ld16(dout + row * dcs + dco + tc * V);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int c = tc * V + k;
                const float xh = (a.get(k) - mean[n * C + c]) * rstd[n * C + c];
                s0[k] += d.get(k) * xh;
                s1[k] += d.get(k);
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (det) {
                s_acc[tr * 2 * C + tc * V + k] = s0[k];
                s_acc[tr * 2 * C + C + tc * V + k] = s1[k];
            } else {
                atomicAdd(&s_acc[tc * V + k], s0[k]);
                atomicAdd(&s_acc[C + tc * V + k], s1[k]);
            }
        }
    }
    __syncthreads();
    if (det) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float v = 0.f;
            for (int r = 0; r < rpi; ++r) v += s_acc[r * 2 * C + i];
            if (i < C) red[(n * C + i) * 2 + 0] += v; else red[(n * C + i - C) * 2 + 1] += v;
        }
        return;
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        atomicAdd(&red[(n * C + i) * 2 + 0], s_acc[i]);
        atomicAdd(&red[(n * C + i) * 2 + 1], s_acc[C + i]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void adain_bwd_apply_kernel(const T* __restrict__ h, const T* __restrict__ dout,
                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                       const float* __restrict__ style, const float* __restrict__ red,
                                       T* __restrict__ dh, float* __restrict__ dstyle, int N, int HW,
                                       int C, int dcs, int dco, int accumulate) {
    constexpr int V = Vec16<T>::N;
    const int cv = C / V;
    const float inv = 1.f / HW;
    if (blockIdx.x == 0 && dstyle) {
        for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
            const int n = i / C, c = i - n * C;
            dstyle[n * 2 * C + c] = red[i * 2 + 0];
            dstyle[n * 2 * C + C + c] = red[i * 2 + 1];
        }
    }
    const int64_t total = (int64_t)N * HW * cv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cv;
        const int c = (int)(i - row * cv) * V;
        const int n = (int)(row / HW);
        Vec16<T> a = ld16(h + row * C + c);
        Vec16<T> d = ld16(dout + row * dcs + dco + c);
        Vec16<T> o;
        if (accumulate) o = ld16(dh + row * C + c);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int nc = n * C + c + k;
            const float xh = (a.get(k) - mean[nc]) * rstd[nc];
            const float g = (style[n * 2 * C + c + k] + 1.f) * rstd[nc];
            float v = g * (d.get(k) - red[nc * 2 + 1] * inv - xh * red[nc * 2 + 0] * inv);
            if (accumulate) v += o.get(k);
            o.set(k, v);
        }
        st16(dh + row * C + c, o);
    }
}

inline int grid_for(int64_t items, int cap = 4096) {
    int64_t b = (items + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int sba_bn_stats(int dtype, const void* y, float* stats, int64_t rows, int groups, int C,
                            void* stream) {
    const int V = dtype != SBA_F32 ? 8 : 4;
    if (!y || !stats || rows <= 0 || groups <= 0 || groups > 65535 || C <= 0 || C % V || C > 4096)
        return SBA_E_ARG;
    const int cv = C / V;
    const int rpi = cv < 256 ? 256 / cv : 1;
    int blocks = cdiv(rows, (int64_t)rpi * 8);
    // every workgroup ends with 2C same-address f32 atomics per replica: measured (tools/bench_bn.py) 85 -> 63 us at
    // 256x256 and 64 -> 43 us at 128x128 going from 1024 to 512 workgroups in total (two per CU)
    static int cap = -1;
    if (cap < 0) { const char* e = getenv("SBA_BN_RED_BLOCKS"); cap = e ? atoi(e) : 512; }
    const int cap_g = cap / groups > 64 ? cap / groups : 64;
    if (blocks > cap_g) blocks = cap_g;
    if (blocks < 1) blocks = 1;
    float* part = nullptr;
    size_t sh = 2 * (size_t)C * sizeof(float);
    if (sba_det_on()) {
        part = sba_det_alloc((int64_t)groups * blocks * 2 * C);
        if (!part) return SBA_E_ARG;
        sh *= rpi;
    }
    SBA_DISPATCH_Y(dtype, SBA_LAUNCH((bn_stats_kernel<YT>), dim3(blocks, groups), dim3(256), sh, (hipStream_t)stream,
                                     (const YT*)y, stats, rows, C, part));
    // deterministic mode: the workgroups' partial sums are added in workgroup order into replica 0 (the others stay zero)
    if (part) sba_det_fold(part, groups, blocks, 2 * C, stats, (int64_t)SBA_BN_STAT_SLOTS * 2 * C, 0, (hipStream_t)stream);
    return SBA_CHECK_LAUNCH();
}

#define ACT_SWITCH(act, CALL)                                                   \
    switch (act) {                                                              \
        case SBA_ACT_NONE: { constexpr int ACT = SBA_ACT_NONE; CALL; } break;   \
        case SBA_ACT_GLU: { constexpr int ACT = SBA_ACT_GLU; CALL; } break;     \
        case SBA_ACT_LRELU: { constexpr int ACT = SBA_ACT_LRELU; CALL; } break; \
        default: return SBA_E_ARG;                                              \
    }

static bool bn_shape_ok(int dtype, int64_t rows, int groups, int C, int act) {
    const int V = dtype != SBA_F32 ? 8 : 4;
    const int Co = act == SBA_ACT_GLU ? C / 2 : C;
    return rows > 0 && groups > 0 && groups <= 65535 && C > 0 && pow2(C) && Co % V == 0 && C <= 4096;
}

extern "C" int sba_bn_act_fwd(int dtype, const void* y, const float* stats, const float* gamma,
                              const float* beta, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float* aux, const void* residual, void* out,
                              int64_t rows, int groups, int C, int act, int out_cstride, int out_coff, float eps,
                              float momentum, int training, void* stream) {
    // (the forward kernel indexes channels generically: any C that is a multiple of the vector width -- the Inception
    // trunk's 80 / 96 / 160 / 192 / 320 / 448 ... channel BatchNorms in the DAMSM pre-training loop)
    const int Vw = dtype != SBA_F32 ? 8 : 4;
    const bool shape_ok = rows > 0 && groups > 0 && groups <= 65535 && C > 0 && C <= 4096 &&
                          (act == SBA_ACT_GLU ? (C / 2) % Vw == 0 && C % 2 == 0 : C % Vw == 0);
    if (!y || !gamma || !beta || !aux || !out || !shape_ok) return SBA_E_ARG;
    if (training ? !stats : (!running_mean || !running_var)) return SBA_E_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SBA_E_ARG;
    if (act == SBA_ACT_GLU && residual) return SBA_E_ARG;
    const int Co = act == SBA_ACT_GLU ? C / 2 : C;
    const int V = dtype != SBA_F32 ? 8 : 4;
    if (out_cstride < Co + out_coff || out_cstride % V || out_coff % V) return SBA_E_ARG;
    const int blocks = grid_for(rows * (Co / V));
    const size_t sh = 2 * (size_t)C * sizeof(float);
    if (act == SBA_ACT_RELU) {
        SBA_DISPATCH_Y(dtype, SBA_LAUNCH((bn_act_fwd_kernel<T, YT, SBA_ACT_RELU>), dim3(blocks, groups), dim3(256), sh,
                                         (hipStream_t)stream, (const YT*)y, stats, gamma, beta, running_mean, running_var,
                                         num_batches_tracked, aux, (const T*)residual, (T*)out, rows, C, out_cstride,
                                         out_coff, eps, momentum, training));
        return SBA_CHECK_LAUNCH();
    }
    SBA_DISPATCH_Y(dtype, ACT_SWITCH(act, SBA_LAUNCH((bn_act_fwd_kernel<T, YT, ACT>), dim3(blocks, groups),
                                                           dim3(256), sh, (hipStream_t)stream, (const YT*)y, stats,
                                                           gamma, beta, running_mean, running_var,
                                                           num_batches_tracked, aux, (const T*)residual, (T*)out,
                                                           rows, C, out_cstride, out_coff, eps, momentum,
                                                           training)));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bn_act_bwd_reduce(int dtype, const void* y, const void* dout, const float* aux, float* red,
                                     int64_t rows, int groups, int C, int act, int dcs, int dco, void* stream) {
    if (!y || !dout || !aux || !red || !bn_shape_ok(dtype, rows, groups, C, act)) return SBA_E_ARG;
    const int Co = act == SBA_ACT_GLU ? C / 2 : C;
    const int V = dtype != SBA_F32 ? 8 : 4;
    if (dcs < Co + dco || dcs % V || dco % V) return SBA_E_ARG;
    const int cv = Co / V;
    const int rpi = cv < 256 ? 256 / cv : 1;
    int blocks = cdiv(rows, (int64_t)rpi * 8);
    // every workgroup ends with 2C same-address f32 atomics per replica: measured (tools/bench_bn.py) 85 -> 63 us at
    // 256x256 and 64 -> 43 us at 128x128 going from 1024 to 512 workgroups in total (two per CU)
    static int cap = -1;
    if (cap < 0) { const char* e = getenv("SBA_BN_RED_BLOCKS"); cap = e ? atoi(e) : 512; }
    const int cap_g = cap / groups > 64 ? cap / groups : 64;
    if (blocks > cap_g) blocks = cap_g;
    if (blocks < 1) blocks = 1;
    size_t sh = 2 * (size_t)C * sizeof(float);
    float* part = nullptr;
    if (sba_det_on()) {
        part = sba_det_alloc((int64_t)groups * blocks * 2 * C);
        if (!part) return SBA_E_ARG;
        sh *= rpi;
    }
    SBA_DISPATCH_Y(dtype, ACT_SWITCH(act, SBA_LAUNCH((bn_bwd_reduce_kernel<T, YT, ACT>), dim3(blocks, groups),
                                                           dim3(256), sh, (hipStream_t)stream, (const YT*)y,
                                                           (const T*)dout, aux, red, rows, C, dcs, dco, part)));
    if (part) sba_det_fold(part, groups, blocks, 2 * C, red, (int64_t)SBA_BN_STAT_SLOTS * 2 * C, 0, (hipStream_t)stream);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bn_act_bwd_apply(int dtype, const void* y, const void* dout, const float* aux,
                                    const float* red, void* dy, float* dgamma, float* dbeta, int64_t rows,
                                    int groups, int C, int act, int dcs, int dco, void* stream) {
    if (!y || !dout || !aux || !red || !dy || !bn_shape_ok(dtype, rows, groups, C, act)) return SBA_E_ARG;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return SBA_E_ARG;
    const int Co = act == SBA_ACT_GLU ? C / 2 : C;
    const int V = dtype != SBA_F32 ? 8 : 4;
    if (dcs < Co + dco || dcs % V || dco % V) return SBA_E_ARG;
    const int blocks = grid_for(rows * (Co / V));
    const size_t sh = 6 * (size_t)C * sizeof(float);
    if (sba_det_on() && groups > 1 && dgamma) {
        // the groups add into the same dgamma / dbeta: one launch per group, in group order
        const int64_t esz = dtype != SBA_F32 ? 2 : 4;
        for (int g = 0; g < groups; ++g) {
            const char* yg = (const char*)y + (int64_t)g * rows * C * esz;
            const char* dg = (const char*)dout + (int64_t)g * rows * dcs * esz;
            char* dyg = (char*)dy + (int64_t)g * rows * C * esz;
            const float* auxg = aux + (int64_t)g * 4 * C;
            const float* redg = red + (int64_t)g * SBA_BN_STAT_SLOTS * 2 * C;
            SBA_DISPATCH_Y(dtype, ACT_SWITCH(act, SBA_LAUNCH((bn_bwd_apply_kernel<T, YT, ACT>), dim3(blocks, 1), dim3(256), sh,
                                                           (hipStream_t)stream, (const YT*)yg, (const T*)dg, auxg, redg,
                                                           (T*)dyg, dgamma, dbeta, rows, C, dcs, dco)));
        }
        return SBA_CHECK_LAUNCH();
    }
    SBA_DISPATCH_Y(dtype, ACT_SWITCH(act, SBA_LAUNCH((bn_bwd_apply_kernel<T, YT, ACT>), dim3(blocks, groups),
                                                           dim3(256), sh, (hipStream_t)stream, (const YT*)y,
                                                           (const T*)dout, aux, red, (T*)dy, dgamma, dbeta, rows, C,
                                                           dcs, dco)));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bn_act_fwd_fused(int dtype, const void* y, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                    float* aux, void* out, int64_t rows, int groups, int C, int act,
                                    int out_cstride, int out_coff, float eps, float momentum, void* stream) {
    if (!y || !gamma || !beta || !aux || !out || !bn_shape_ok(dtype, rows, groups, C, act)) return SBA_E_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SBA_E_ARG;
    const int Co = act == SBA_ACT_GLU ? C / 2 : C;
    const int V = dtype != SBA_F32 ? 8 : 4;
    if (out_cstride < Co + out_coff || out_cstride % V || out_coff % V) return SBA_E_ARG;
    SBA_DISPATCH_Y(dtype, ACT_SWITCH(act, SBA_LAUNCH((bn_fwd_fused_kernel<T, YT, ACT>), dim3(Co / V), dim3(256), 0,
                                                           (hipStream_t)stream, (const YT*)y, gamma, beta,
                                                           running_mean, running_var, num_batches_tracked, aux,
                                                           (T*)out, rows, groups, C, out_cstride, out_coff, eps,
                                                           momentum)));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bn_act_bwd_fused(int dtype, const void* y, const void* dout, const float* aux, void* dy,
                                    float* dgamma, float* dbeta, int64_t rows, int groups, int C, int act, int dcs,
                                    int dco, void* stream) {
    if (!y || !dout || !aux || !dy || !bn_shape_ok(dtype, rows, groups, C, act)) return SBA_E_ARG;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return SBA_E_ARG;
    const int Co = act == SBA_ACT_GLU ? C / 2 : C;
    const int V = dtype != SBA_F32 ? 8 : 4;
    if (dcs < Co + dco || dcs % V || dco % V) return SBA_E_ARG;
    if (sba_det_on() && groups > 1 && dgamma) {
        const int64_t esz = dtype != SBA_F32 ? 2 : 4;
        for (int g = 0; g < groups; ++g) {
            const char* yg = (const char*)y + (int64_t)g * rows * C * esz;
            const char* dg = (const char*)dout + (int64_t)g * rows * dcs * esz;
            char* dyg = (char*)dy + (int64_t)g * rows * C * esz;
            const float* auxg = aux + (int64_t)g * 4 * C;
            SBA_DISPATCH_Y(dtype, ACT_SWITCH(act, SBA_LAUNCH((bn_bwd_fused_kernel<T, YT, ACT>), dim3(Co / V, 1), dim3(256), 0,
                                                           (hipStream_t)stream, (const YT*)yg, (const T*)dg, auxg,
                                                           (T*)dyg, dgamma, dbeta, rows, C, dcs, dco)));
        }
        return SBA_CHECK_LAUNCH();
    }
    SBA_DISPATCH_Y(dtype, ACT_SWITCH(act, SBA_LAUNCH((bn_bwd_fused_kernel<T, YT, ACT>), dim3(Co / V, groups),
                                                           dim3(256), 0, (hipStream_t)stream, (const YT*)y,
                                                           (const T*)dout, aux, (T*)dy, dgamma, dbeta, rows, C, dcs,
                                                           dco)));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bn1d_glu_fwd(int dtype, const float* y, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* nbt, float* mean, float* rstd,
                                void* out, int B, int F, float eps, float momentum, void* stream) {
    if (!y || !gamma || !beta || !mean || !rstd || !out || B <= 0 || F <= 0 || F % 32 != 0) return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((bn1d_glu_fwd_kernel<T>), dim3(cdiv(F / 2, 64)), dim3(64), 0,
                                           (hipStream_t)stream, y, gamma, beta, running_mean, running_var, nbt,
                                           mean, rstd, (T*)out, B, F, eps, momentum));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_bn1d_glu_bwd(int dtype, const float* y, const void* dout, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, float* dy, float* dgamma, float* dbeta,
                                int B, int F, void* stream) {
    if (!y || !dout || !gamma || !beta || !mean || !rstd || !dy || !dgamma || !dbeta || B <= 0 || F <= 0 ||
        F % 32 != 0)
        return SBA_E_ARG;
    SBA_DISPATCH(dtype, SBA_LAUNCH((bn1d_glu_bwd_kernel<T>), dim3(cdiv(F / 2, 64)), dim3(64), 0,
                                           (hipStream_t)stream, y, (const T*)dout, gamma, beta, mean, rstd, dy,
                                           dgamma, dbeta, B, F));
    return SBA_CHECK_LAUNCH();
}

static bool in_shape_ok(int dtype, int N, int HW, int C) {
    const int V = dtype != SBA_F32 ? 8 : 4;
    return N > 0 && HW > 0 && C > 0 && C % V == 0 && C / V <= 256 && pow2(C / V);
}

extern "C" int sba_instnorm_stats(int dtype, const void* h, float* mean, float* rstd, int N, int HW, int C,
                                  float eps, void* stream) {
    if (!h || !mean || !rstd || !in_shape_ok(dtype, N, HW, C)) return SBA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int V = dtype != SBA_F32 ? 8 : 4;
    static int fused = -1;      // SBA_INSTNORM_FUSED=0: clear + accumulate (atomics) + finalize (A/B aid)
    if (fused < 0) { const char* e = getenv("SBA_INSTNORM_FUSED"); fused = (e && e[0] == '0') ? 0 : 1; }
    if (fused && N * (C / V) >= 64 && HW >= 1024) {        // enough workgroups of enough pixels: one launch
        SBA_DISPATCH(dtype, SBA_LAUNCH((instnorm_stats_fused_kernel<T>), dim3(N, C / V), dim3(256), 0, st, (const T*)h, mean,
                                               rstd, HW, C, eps));
        return SBA_CHECK_LAUNCH();
    }
    sba_zero_f32(mean, rstd, (int64_t)N * C, st);
    const int rpi = 256 / (C / V);
    int splits = cdiv(HW, rpi * 16);
    if (splits > 256) splits = 256;
    const int det = sba_det_on() ? 1 : 0;
    if (det) splits = 1;            // one workgroup per image, row-groups added in order
    SBA_DISPATCH(dtype, SBA_LAUNCH((instnorm_accum_kernel<T>), dim3(N, splits), dim3(256),
                                           (det ? rpi : 1) * 2 * C * sizeof(float), st, (const T*)h, mean, rstd, HW, C, det));
    SBA_LAUNCH(instnorm_finalize_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, st, mean, rstd, N * C,
                       (float)HW, eps);
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_adain_fwd(int dtype, const void* h, const float* mean, const float* rstd, const float* style,
                             void* out, int N, int HW, int C, int ocs, int oco, void* stream) {
    if (!h || !mean || !rstd || !style || !out || !in_shape_ok(dtype, N, HW, C)) return SBA_E_ARG;
    const int V = dtype != SBA_F32 ? 8 : 4;
    if (ocs < C + oco || ocs % V || oco % V) return SBA_E_ARG;
    const int blocks = grid_for((int64_t)N * HW * (C / V));
    SBA_DISPATCH(dtype, SBA_LAUNCH((adain_fwd_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)h, mean, rstd, style, (T*)out, N, HW, C, ocs, oco));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_adain_bwd_reduce(int dtype, const void* h, const void* dout, const float* mean,
                                    const float* rstd, float* red, int N, int HW, int C, int dcs, int dco,
                                    void* stream) {
    if (!h || !dout || !mean || !rstd || !red || !in_shape_ok(dtype, N, HW, C)) return SBA_E_ARG;
    const int V = dtype != SBA_F32 ? 8 : 4;
    if (dcs < C + dco || dcs % V || dco % V) return SBA_E_ARG;
    const int rpi = 256 / (C / V);
    int splits = cdiv(HW, rpi * 16);
    if (splits > 256) splits = 256;
    const int det = sba_det_on() ? 1 : 0;
    if (det) splits = 1;
    SBA_DISPATCH(dtype, SBA_LAUNCH((adain_bwd_reduce_kernel<T>), dim3(N, splits), dim3(256),
                                           (det ? rpi : 1) * 2 * C * sizeof(float), (hipStream_t)stream, (const T*)h,
                                           (const T*)dout, mean, rstd, red, HW, C, dcs, dco, det));
    return SBA_CHECK_LAUNCH();
}

extern "C" int sba_adain_bwd_apply(int dtype, const void* h, const void* dout, const float* mean,
                                   const float* rstd, const float* style, const float* red, void* dh,
                                   float* dstyle, int N, int HW, int C, int dcs, int dco, int accumulate,
                                   void* stream) {
    if (!h || !dout || !mean || !rstd || !style || !red || !dh || !in_shape_ok(dtype, N, HW, C)) return SBA_E_ARG;
    const int V = dtype != SBA_F32 ? 8 : 4;
    if (dcs < C + dco || dcs % V || dco % V) return SBA_E_ARG;
    const int blocks = grid_for((int64_t)N * HW * (C / V));
    SBA_DISPATCH(dtype, SBA_LAUNCH((adain_bwd_apply_kernel<T>), dim3(blocks), dim3(256), 0,
                                           (hipStream_t)stream, (const T*)h, (const T*)dout, mean, rstd, style,
                                           red, (T*)dh, dstyle, N, HW, C, dcs, dco, accumulate));
    return SBA_CHECK_LAUNCH();
}
