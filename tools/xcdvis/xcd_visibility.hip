// Does a kernel see every store of the previous kernel ON THE SAME STREAM when a second stream keeps the device busy?
// Each stream loops { produce(buf, it); consume(buf, it) }: workgroup b of `consume` checks the chunk written by workgroup
// (b + shift) of `produce` (another XCD: consecutive workgroups go round the 8 XCDs).  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void produce(float* buf, int chunk, float val) {
    float4* p = reinterpret_cast<float4*>(buf + (size_t)blockIdx.x * chunk);
    for (int i = threadIdx.x; i < chunk / 4; i += blockDim.x) p[i] = make_float4(val, val, val, val);
}
__global__ void consume(const float* buf, int chunk, int nchunks, float val, unsigned* bad, int shift) {
    const int src = (blockIdx.x + shift) % nchunks;
    const float* p = buf + (size_t)src * chunk;
    unsigned n = 0;
    for (int i = threadIdx.x; i < chunk; i += blockDim.x) n += p[i] != val;
    if (n) atomicAdd(bad, n);
}
// a long-running kernel of another kind for the second stream (keeps CUs busy, touches its own buffer only)
__global__ void busy(float* buf, int n, int rounds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = buf[i % n];
    for (int r = 0; r < rounds; ++r) v = v * 1.0001f + 0.5f;
    buf[i % n] = v;
}

int main(int argc, char** argv) {
    const int nstreams = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 300;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;       // 0: every stream runs the pair; 1: stream 1.. run `busy`
    const int nchunks = 4096, chunk = 2048;
    std::vector<hipStream_t> st(nstreams);
    std::vector<float*> buf(nstreams);
    std::vector<unsigned*> bad(nstreams);
    for (int s = 0; s < nstreams; ++s) {
        hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
        hipMalloc(&buf[s], (size_t)nchunks * chunk * 4);
        hipMalloc(&bad[s], 4);
        hipMemset(bad[s], 0, 4);
    }
    hipDeviceSynchronize();
    for (int it = 1; it <= iters; ++it)
        for (int s = 0; s < nstreams; ++s) {
            if (mode == 1 && s > 0) {
                hipLaunchKernelGGL(busy, dim3(3000), dim3(256), 0, st[s], buf[s], nchunks * chunk, 2000 + 37 * (it % 7));
                continue;
            }
            hipLaunchKernelGGL(produce, dim3(nchunks), dim3(256), 0, st[s], buf[s], chunk, (float)it);
            hipLaunchKernelGGL(consume, dim3(nchunks), dim3(256), 0, st[s], buf[s], chunk, nchunks, (float)it, bad[s], 1 + (it % 5));
        }
    hipDeviceSynchronize();
    for (int s = 0; s < nstreams; ++s) {
        unsigned h = 0;
        hipMemcpy(&h, bad[s], 4, hipMemcpyDeviceToHost);
        printf("streams %d mode %d: stream %d stale values seen by the consumer: %u of %.3g\n", nstreams, mode, s, h,
               (double)iters * nchunks * chunk);
    }
    return 0;
}
