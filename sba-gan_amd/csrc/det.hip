// Deterministic-reduction mode: the scratch ring and the ordered fold (see common.h).
//
// The reference is PyTorch on cuDNN, whose default algorithms are not run-to-run reproducible either; this mode
// exists so that the parity tests can tell a race from floating-point reassociation: with it on, two runs of the
// same step from the same state are bit-identical in every launch mode (tests/test_step_gpu.py).
#include <cstdio>
#include <mutex>

#include "common.h"

namespace {
std::mutex g_mu;
bool g_on = false;
char* g_ring = nullptr;
int64_t g_bytes = 0, g_off = 0, g_used_since_reset = 0, g_high_water = 0;

__global__ __launch_bounds__(256) void det_fold_kernel(const float* __restrict__ part, const int P, const int64_t n,
                                                       float* __restrict__ dst, const int64_t dst_stride,
                                                       const int mode) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = blockIdx.y;
    const float* p = part + (int64_t)j * P * n + i;
    float s = 0.f;
    for (int q = 0; q < P; ++q) s += p[(int64_t)q * n];
    float* d = dst + (int64_t)j * dst_stride + i;
    *d = mode == 1 ? s : *d + s;
}
// default mode: the same fold spread over the chip -- a thread adds up 32 partial sums, then ONE atomic per (32 slots, output)
__global__ __launch_bounds__(256) void fold_add_kernel(const float* __restrict__ part, const int P, const int64_t n,
                                                       float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int q0 = blockIdx.y * 32, q1 = min(q0 + 32, P);
    float s = 0.f;
    for (int q = q0; q < q1; ++q) s += part[(int64_t)q * n + i];
    atomicAdd(&dst[i], s);
}

char* r_ring = nullptr;
int64_t r_bytes = 0, r_off = 0;
}  // namespace

bool sba_det_on() { return g_on; }

float* sba_reduce_alloc(int64_t nfloats) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int64_t bytes = ((nfloats * 4 + 255) / 256) * 256;
    if (!r_ring || bytes > r_bytes / 4) return nullptr;     // (a launch may take a quarter of the ring at most)
    if (r_off + bytes > r_bytes) r_off = 0;
    float* p = reinterpret_cast<float*>(r_ring + r_off);
    r_off += bytes;
    return p;
}

void sba_fold_add(const float* part, int P, int64_t n, float* dst, hipStream_t st) {
    if (P <= 0 || n <= 0) return;
    SBA_LAUNCH(fold_add_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)((P + 31) / 32)), dim3(256), 0, st, part, P, n, dst);
}

extern "C" int sba_set_reduce_scratch(void* scratch, int64_t scratch_bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (scratch && (scratch_bytes < (1 << 20) || ((uintptr_t)scratch & 255) != 0)) return SBA_E_ARG;
    r_ring = reinterpret_cast<char*>(scratch);
    r_bytes = scratch ? scratch_bytes : 0;
    r_off = 0;
    return SBA_OK;
}

float* sba_det_alloc(int64_t nfloats) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int64_t bytes = ((nfloats * 4 + 255) / 256) * 256;
    if (!g_ring || bytes > g_bytes) return nullptr;
    if (g_off + bytes > g_bytes) {
        // NO wrap: slots handed out since the last sba_det_reset() may still be waiting for their fold, and this is the
        // mode that exists to prove bit equality -- fail loudly (SBA_E_ARG from the launching entry point) instead
        static bool said = false;
        if (!said) {
            said = true;
            fprintf(stderr, "sbagan_hip: deterministic scratch ring exhausted (%lld of %lld bytes used since the last "
                            "sba_det_reset): reset it at the start of every step or enlarge it (SBA_DET_SCRATCH_MB)\n",
                    (long long)g_used_since_reset, (long long)g_bytes);
        }
        return nullptr;
    }
    float* p = reinterpret_cast<float*>(g_ring + g_off);
    g_off += bytes;
    g_used_since_reset += bytes;
    if (g_used_since_reset > g_high_water) g_high_water = g_used_since_reset;
    return p;
}

void sba_det_fold(const float* part, int J, int P, int64_t n, float* dst, int64_t dst_stride, int mode, hipStream_t st) {
    if (J <= 0 || P <= 0 || n <= 0) return;
    SBA_LAUNCH(det_fold_kernel, dim3((unsigned)((n + 255) / 256), J), dim3(256), 0, st, part, P, n, dst, dst_stride, mode);
}

extern "C" int sba_set_deterministic(int on, void* scratch, int64_t scratch_bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (on && (!scratch || scratch_bytes < (1 << 20) || ((uintptr_t)scratch & 255) != 0)) return SBA_E_ARG;
    g_on = on != 0;
    g_ring = on ? reinterpret_cast<char*>(scratch) : nullptr;
    g_bytes = on ? scratch_bytes : 0;
    g_off = g_used_since_reset = 0;
    return SBA_OK;
}

extern "C" int sba_get_deterministic(void) { return g_on ? 1 : 0; }

extern "C" int sba_det_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_off = g_used_since_reset = 0;
    return SBA_OK;
}

extern "C" int64_t sba_det_high_water(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_high_water;
}

extern "C" int sba_bn_stat_slots(void) { return SBA_BN_STAT_SLOTS; }
