// Do HIP stream priorities change how two concurrent kernel chains share an MI355X?   hipcc --offload-arch=gfx950 -O2 -o tools/_probe/prio_probe tools/prio_probe.hip
// Two streams each issue `n` launches of a kernel of `wgs` workgroups; the time from the common start to each stream's last
// kernel is printed for (normal, normal) and (high, low) priorities.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void spin_kernel(float* p, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
    if (a == 12345.f) p[0] = a;
}
static int run(int pa, int pb, int wgs_a, int wgs_b, int n, int iters, float* buf) {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, pa));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, pb));
    hipEvent_t e0, ea, eb;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sa));
        CK(hipStreamWaitEvent(sb, e0, 0));
        for (int i = 0; i < n; ++i) {
            spin_kernel<<<wgs_a, 256, 0, sa>>>(buf, iters);
            spin_kernel<<<wgs_b, 256, 0, sb>>>(buf, iters);
        }
        CK(hipEventRecord(ea, sa)); CK(hipEventRecord(eb, sb));
        CK(hipDeviceSynchronize());
    }
    float ta, tb;
    CK(hipEventElapsedTime(&ta, e0, ea)); CK(hipEventElapsedTime(&tb, e0, eb));
    printf("  priorities (%2d, %2d), workgroups (%5d, %5d): stream A done at %7.3f ms, stream B at %7.3f ms\n", pa, pb, wgs_a, wgs_b, ta, tb);
    CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
    return 0;
}
int main() {
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    float* buf; CK(hipMalloc(&buf, 1024));
    const int n = 100, iters = 20000;
    for (int wgs : {256, 1024, 4096}) {
        for (int wb : {wgs, 4 * wgs}) {
            if (run(0, 0, wgs, wb, n, iters, buf)) return 1;
            if (run(hi, lo, wgs, wb, n, iters, buf)) return 1;
            if (run(lo, hi, wgs, wb, n, iters, buf)) return 1;
        }
    }
    return 0;
}
