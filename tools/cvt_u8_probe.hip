// What v_cvt_pk_u8_f32 and v_max_f32(x, 0) do with halves, out-of-range values, signed zeros and NaN (quant.hip relies on it):
//   hipcc -O2 --offload-arch=gfx950 tools/cvt_u8_probe.hip -o tools/cvt_u8_probe && ./tools/cvt_u8_probe   (output: profiles/r05_cvt_pk_u8_probe.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) {
        out[3 * i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
        out[3 * i + 1] = __float_as_uint(__builtin_fmaxf(in[i], 0.0f));
        out[3 * i + 2] = __float_as_uint(in[i] > 0.0f ? in[i] : 0.0f);
    }
}
int main() {
    float h[] = {0.5f, 1.5f, 2.5f, 3.5f, 0.49999997f, 254.5f, 255.5f, 256.0f, 300.0f, -0.5f, -0.0f, -1.0f, 1.4999999f, 2.5000002f, 126.5f, 127.5f, __builtin_nanf(""), 1e-45f, -1e-45f, __builtin_inff(), -__builtin_inff()};
    int n = sizeof(h) / 4;
    float* d; unsigned* o; unsigned r[64 * 3];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(r, o, n * 12, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%g -> cvt_pk_u8 %u  fmax bits %08x  select bits %08x\n", h[i], r[3 * i] & 255, r[3 * i + 1], r[3 * i + 2]);
    return 0;
}
