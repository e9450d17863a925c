// Probe of v_permlane16_swap_b32 / v_permlane32_swap_b32 lane semantics on gfx950 (the attention_w64 epilogue and row reductions rely on them).
//   hipcc --offload-arch=gfx950 tools/probe/permlane_probe.hip -o tools/probe/permlane_probe && tools/probe/permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap16 vdst", "swap16 src ", "swap32 vdst", "swap32 src "};
    for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int r = 0; r < 4; ++r) printf(" row%d=%u..%u", r, h[t * 64 + r * 16], h[t * 64 + r * 16 + 15]); printf("\n"); }
    // expected (a = 100 + lane, b = 200 + lane): swap16 vdst rows = a0 b0 a2 b2 -> 100.. 200.. 132.. 232..; src rows = a1 b1 a3 b3 -> 116.. 216.. 148.. 248..
    //           swap32 vdst = a.lo b.lo -> 100..131 200..231; src = a.hi b.hi -> 132..163 232..263
    const bool ok = h[0] == 100 && h[16] == 200 && h[32] == 132 && h[48] == 232 && h[64] == 116 && h[80] == 216 && h[96] == 148 && h[112] == 248 &&
                    h[128] == 100 && h[160] == 200 && h[192] == 132 && h[224] == 232;
    printf("permlane semantics %s\n", ok ? "AS ASSUMED" : "DIFFERENT");
    return ok ? 0 : 1;
}
