// Fixed cost of a kernel in a recorded chain: N dependent kernel nodes in ONE hipGraph, (a) one tiny workgroup each, (b) 512 workgroups of
// 512 threads with 74 KB of dynamic LDS that return at once, (c) the same with a 3 us spin per workgroup (does the drain of one overlap
// the ramp of the next?).  Prints us per node.   hipcc --offload-arch=gfx950 -O2 tools/launch_gap.hip -o tools/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ __launch_bounds__(512) void k_lds(int* p, int spin) {
    extern __shared__ char lds[];
    if (spin) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) {}
    }
    if (p && threadIdx.x == 9999) p[0] = lds[0];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static float run(hipStream_t s, int n, int mode) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) {
        if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, nullptr);
        else hipLaunchKernelGGL(k_lds, dim3(512), dim3(512), 74 * 1024, s, nullptr, mode == 2 ? 300 : 0);   // wall clock: 100 MHz -> 3 us
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.f / 10 / n;
}
int main() {
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 74 * 1024));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int n = 200;
    printf("{\"nodes\": %d, \"us_per_node_tiny\": %.2f, \"us_per_node_512x512_74KB\": %.2f, \"us_per_node_512x512_74KB_3us_spin\": %.2f}\n", n, run(s, n, 0), run(s, n, 1),
           run(s, n, 2));
    return 0;
}
