// Practical MFMA ceiling of the box: register-only v_mfma_f32_16x16x32_bf16 loop (no LDS, no HBM) at 1 and 2 waves/SIMD,
// plus the shader clock seen under that load (clock64 vs the 100 MHz wall clock). Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, long long* clk, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)(float)(i & 1); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, NACC = 16;
    for (int wpb : {256, 512}) {            // 4 or 8 waves per CU-resident block -> 1 or 2 waves / SIMD
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            mfma_loop<NACC><<<256, wpb>>>(out, clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            double flops = 256.0 * (wpb / 64) * iters * NACC * 2.0 * 16 * 16 * 32;
            printf("waves/CU=%d  %.1f TFLOP/s  (%.2f ms)  shader clock %.0f MHz (clock64 %lld / wall %lld @100MHz)\n", wpb / 64,
                   flops / ms * 1e-9, ms, (double)h[0] / ((double)h[1] / 100.0), h[0], h[1]);
        }
    }
    return 0;
}
