#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(r), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__global__ void k(const uint4* __restrict__ x, uint4* __restrict__ y, int n, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[98304];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)(n * 16), 0x00020000);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) ((uint32_t*)lds)[i] = 0xdeadbeefu;
    __syncthreads();
    uint32_t off = threadIdx.x * 16u;
    if (mode == 1 && (threadIdx.x & 1)) off = 0xFFFFFFFFu;           // OOB lanes
    if (mode == 2) off = (threadIdx.x ^ 3) * 16u;                      // per-lane permuted source
    const uint32_t hi = mode == 3 ? 81920u : 0u;          // mode 3: destination beyond 64 KiB
    if (mode == 3) { for (int i = threadIdx.x; i < 1024; i += blockDim.x) ((uint32_t*)(lds + 81920))[i] = 0xdeadbeefu; __syncthreads(); }
    dma16(r, off, base + hi + wid * 1024);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    y[threadIdx.x] = *reinterpret_cast<uint4*>(lds + hi + threadIdx.x * 16);
}
int main() {
    const int n = 256;
    uint4 *x, *y; hipMalloc(&x, n * 16); hipMalloc(&y, n * 16);
    uint32_t h[n * 4]; for (int i = 0; i < n * 4; ++i) h[i] = 0x1000 + i;
    hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, y, n, mode);
        uint32_t o[n * 4]; hipMemcpy(o, y, sizeof(o), hipMemcpyDeviceToHost);
        int ok = 1, zeros = 1, untouched = 1;
        for (int t = 0; t < n; ++t) for (int j = 0; j < 4; ++j) {
            uint32_t v = o[t * 4 + j];
            if ((mode == 0 || mode == 3) && v != h[t * 4 + j]) ok = 0;
            if (mode == 2 && v != h[(t ^ 3) * 4 + j]) ok = 0;
            if (mode == 1) { if (t & 1) { if (v != 0) zeros = 0; if (v != 0xdeadbeefu) untouched = 0; } else if (v != h[t * 4 + j]) ok = 0; }
        }
        printf("mode %d: ok=%d oob_lanes_zero=%d oob_lanes_untouched=%d sample %08x %08x\n", mode, ok, zeros, untouched, o[4], o[8]);
    }
    return 0;
}
