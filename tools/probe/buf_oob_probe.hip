// Probe: does a raw buffer load's range check include the scalar offset (soffset)?  (tools only; not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* p, unsigned bytes, int soff, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
    int voff = threadIdx.x * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = ((float*)smem)[threadIdx.x * 4];
}
int main() {
    float *d, *o; const int n = 4096;
    hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 4);
    float h[n]; for (int i = 0; i < n; ++i) h[i] = 1000 + i;
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    // resource covers the first 1024 bytes only; lanes read 64 x 16 B = 1024 B starting at soff
    for (int soff : {0, 512, 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, 1024u, soff, o);
        float r[64]; hipMemcpy(r, o, 64 * 4, hipMemcpyDeviceToHost);
        printf("soff %4d: lane0 %.0f lane31 %.0f lane32 %.0f lane63 %.0f\n", soff, r[0], r[31], r[32], r[63]);
    }
    return 0;
}
