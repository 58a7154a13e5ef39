// Probe of ds_read_b64_tr_b16 (gfx950): which 16-bit elements does lane l receive when every lane supplies its own 8-byte address?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o tools/probe/tr_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[64 * 4];
    const int l = threadIdx.x;
    // lane i owns elements 4 i .. 4 i + 3, value = lane * 4 + e
    for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (short)(l * 4 + e);
    __syncthreads();
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)&lds[l * 4]);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = r[e];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
