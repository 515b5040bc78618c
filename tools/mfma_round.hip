// How does v_mfma_f32_32x32x16_bf16 round its f32 accumulation on gfx950?  (hipcc --offload-arch=gfx950 tools/mfma_round.hip -o tools/mfma_round)
// One wave.  A = one non-zero column (k = 0) of value a, B = one non-zero row of value b, C = c everywhere: D = c + a * b.
// With a * b = 1.5 * 2^-24 and c = 1: round-to-nearest-even gives 1 + 2^-23, truncation gives 1.  Also: many small addends (the
// bias a chain of accumulations collects), and a sum of 16 products inside ONE instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float* av, const float* bv, float c, int reps, int fill_k, float* out) {
    const int lane = threadIdx.x;
    bf16x8 a, b;
    // A fragment: lane holds row (lane % 32), k = 8 * (lane / 32) .. +7; B likewise with column (lane % 32)
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * (lane / 32) + i;
        a[i] = (__bf16)(k < fill_k ? av[k] : 0.0f);
        b[i] = (__bf16)(k < fill_k ? bv[k] : 0.0f);
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = c;
    for (int r = 0; r < reps; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

static float run(const float* a, const float* b, float c, int reps, int fill_k) {
    float *da, *db, *dout, h;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    hipMemcpy(da, a, 64, hipMemcpyHostToDevice); hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, c, reps, fill_k, dout);
    hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return h;
}

int main() {
    float a[16] = {0}, b[16] = {0};
    a[0] = ldexpf(1.0f, -12); b[0] = ldexpf(1.5f, -12);
    printf("{\"one_plus_0.75ulp\": %.10g, \"rne\": %.10g, \"rz\": 1, ", run(a, b, 1.0f, 1, 1), 1.0 + ldexp(1.0, -23));
    b[0] = -ldexpf(1.5f, -12);
    printf("\"minus_one_minus_0.75ulp\": %.10g, ", run(a, b, -1.0f, 1, 1));
    b[0] = ldexpf(1.5f, -12);
    printf("\"one_minus... c=-1 plus 0.75ulp(toward zero)\": %.10g, ", run(a, b, -1.0f, 1, 1));
    // 1000 accumulations of 0.25 ulp(1) each onto 1.0: exact 1 + 250 ulp; RNE keeps 1.0 (each addend below half an ulp); RZ keeps 1.0 too.
    // 1000 accumulations of 0.75 ulp: RNE -> 1 + 1000 ulp (each rounds up to a full ulp), RZ -> 1.0, exact 1 + 750 ulp.
    printf("\"1000_x_0.75ulp_minus_1_in_ulps\": %.10g, ", (run(a, b, 1.0f, 1000, 1) - 1.0) / ldexp(1.0, -23));
    // 16 products in one instruction, each 0.25 ulp(1): their sum is 4 ulp exactly.  If the products are summed first (wide) and the
    // total rounded once: 1 + 4 ulp.  If each is added to c separately with truncation: 1.0.
    for (int k = 0; k < 16; ++k) { a[k] = ldexpf(1.0f, -12); b[k] = ldexpf(1.0f, -13); }
    printf("\"16_products_of_0.25ulp_in_one_mfma_minus_1_in_ulps\": %.10g, ", (run(a, b, 1.0f, 1, 16) - 1.0) / ldexp(1.0, -23));
    // 16 products of 0.4375 ulp = 7 ulp: exact sum is representable; 3 products of ... a non-representable total: 16 x 0.46875 ulp = 7.5 ulp
    for (int k = 0; k < 16; ++k) { a[k] = ldexpf(1.0f, -12); b[k] = ldexpf(1.875f, -14); }
    printf("\"16_products_summing_to_7.5ulp_minus_1_in_ulps\": %.10g}\n", (run(a, b, 1.0f, 1, 16) - 1.0) / ldexp(1.0, -23));
    return 0;
}
