// lstm.hip -- the recurrence of an LSTM layer as ONE launch (the AutoVC content converter in front of
// Module1, Module1/src/autovc/retrain_version/model_vc_37_1.py:48-112 -- a 2-layer bidirectional LSTM with 16 hidden units and a
// 3-layer LSTM with 512, run over the clip's ~830 spectrogram frames at batch 1).
//
//   gates_t = xproj[t] + W_hh h_{t-1}          (xproj = W_ih x_t + b_ih + b_hh for ALL t: one library GEMM on the host side)
//   i, f, g, o = sigmoid, sigmoid, tanh, sigmoid of the four H-row blocks (PyTorch's gate order)
//   c_t = f c_{t-1} + i g,   h_t = o tanh(c_t)
//
// Through torch / MIOpen the recurrence is a few tiny kernels per time step -- ~11 k launches and 0.20 s of host launch time per
// 10 s clip for ~30 ms of device work (bench.py stream leg, `autovc_converter`), and it cannot be captured in a hipGraph
// (MIOpen's RNN calls hipBLASLt, which synchronises inside the call and aborts a capturing process).  Here the time loop runs
// inside the kernel:
//   * lstm_small_kernel (H <= 64): one workgroup per (direction, batch row); thread = gate row, its W_hh row in registers,
//     h in LDS; no inter-workgroup traffic at all.
//   * lstm_dist_kernel<H> (H = 256, 512): 8 hidden units per workgroup (H / 8 workgroups), a thread keeps 1/8 of a gate row of
//     W_hh in registers for the whole sequence (W_hh is read from memory ONCE), h_{t-1} is exchanged through a 2 x H global
//     buffer and a monotonic arrival counter per step.  All accesses to the exchange data are agent-scope relaxed atomics
//     (performed at the coherence point: the per-CU L1 is never refreshed by another CU's stores), ordered by waiting for the
//     stores' acknowledgement before the workgroup barrier that precedes the counter bump (the protocol of conv_bf16x3.h's
//     in-kernel InstanceNorm).  The workers are the blocks b with b % 8 == 0 of an 8 x larger grid -- one XCD.
#include "common.h"

#include <cstdint>

namespace apamd {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // (exp-based, accurate to ~1e-7 relative over the range that matters; saturates cleanly)
    const float e = __expf(-2.f * fabsf(x));
    const float t = (1.f - e) / (1.f + e);
    return x < 0.f ? -t : t;
}

struct LstmParams {
    const float* xproj;       // [B][T][4H]
    const float* whh;         // [4H][H]
    const float* h0;          // [B][H] or null (zeros)
    const float* c0;
    float* out;               // [B][T][out_stride] at column out_off
    float* hn;                // [B][H] or null
    float* cn;
    int B, T, H, reverse, out_stride, out_off;
    float* hx;                // dist kernel: exchange buffer [2][H]
    unsigned* ctr;            // dist kernel: [0] arrival counter (zero on entry), [1] error flag
};

// grid: (B), 4H threads (<= 256)
__global__ __launch_bounds__(256) void lstm_small_kernel(const LstmParams p) {
    __shared__ float hs[64], gs[256];
    const int b = blockIdx.x, tid = threadIdx.x, H = p.H, G4 = 4 * H;
    float w[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) w[k] = (tid < G4 && k < H) ? p.whh[(long long)tid * H + k] : 0.f;
    float c = 0.f;
    if (tid < H) {
        hs[tid] = p.h0 ? p.h0[b * H + tid] : 0.f;
        c = p.c0 ? p.c0[b * H + tid] : 0.f;
    }
    __syncthreads();
    const float* xp = p.xproj + (long long)b * p.T * G4;
    float xnext = tid < G4 ? xp[(long long)(p.reverse ? p.T - 1 : 0) * G4 + tid] : 0.f;
    for (int s = 0; s < p.T; ++s) {
        const int t = p.reverse ? p.T - 1 - s : s;
        float a = xnext;
        if (s + 1 < p.T && tid < G4) xnext = xp[(long long)(p.reverse ? t - 1 : t + 1) * G4 + tid];     // (in flight under the dot product)
#pragma unroll
        for (int k = 0; k < 64; ++k)
            if (k < H) a += w[k] * hs[k];
        gs[tid] = a;
        __syncthreads();
        if (tid < H) {
            const float gi = sigmoidf_(gs[tid]), gf = sigmoidf_(gs[H + tid]), gg = tanhf_(gs[2 * H + tid]), go = sigmoidf_(gs[3 * H + tid]);
            c = gf * c + gi * gg;
            const float h = go * tanhf_(c);
            hs[tid] = h;
            p.out[((long long)b * p.T + t) * p.out_stride + p.out_off + tid] = h;
        }
        __syncthreads();
    }
    if (tid < H) {
        if (p.hn) p.hn[b * H + tid] = hs[tid];
        if (p.cn) p.cn[b * H + tid] = c;
    }
}

// grid: 8 * (H / 8) blocks of 256 threads; block b works iff b % 8 == 0 (worker b / 8: hidden units 8 w .. 8 w + 7).  B == 1.
template <int H>
__global__ __launch_bounds__(256) void lstm_dist_kernel(const LstmParams p) {
    constexpr int UPW = 8, NJ = H / 8;
    if (blockIdx.x & 7) return;
    const int wk = blockIdx.x >> 3, nworkers = H / UPW;
    __shared__ float hs[H], gs[32];
    const int tid = threadIdx.x, r = tid >> 3, part = tid & 7;          // gate row r = gate * 8 + unit, 1/8 of its columns
    const int gate = r >> 3, unit = r & 7, u0 = wk * UPW;
    float w[NJ];
    {
        const float* wr = p.whh + (long long)(gate * H + u0 + unit) * H;
#pragma unroll
        for (int j = 0; j < NJ; ++j) w[j] = wr[part + 8 * j];
    }
    float c = 0.f;
    if (tid < UPW) c = p.c0 ? p.c0[u0 + tid] : 0.f;
    const int G4 = 4 * H;
    float xn[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch_x = [&](int t) {
        if (tid < UPW) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xn[g] = p.xproj[(long long)t * G4 + g * H + u0 + tid];
        }
    };
    fetch_x(p.reverse ? p.T - 1 : 0);
    for (int s = 0; s < p.T; ++s) {
        const int t = p.reverse ? p.T - 1 - s : s;
        // h_{t-1}: the initial state, or what the workers published in step s - 1 (buffer (s - 1) & 1)
        for (int k = tid; k < H; k += 256) {
            float v;
            if (s == 0) v = p.h0 ? p.h0[k] : 0.f;
            else v = __hip_atomic_load(p.hx + ((s - 1) & 1) * H + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hs[k] = v;
        }
        __syncthreads();
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) a += w[j] * hs[part + 8 * j];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 4, 64);
        if (part == 0) gs[r] = a;
        __syncthreads();
        if (tid < UPW) {
            const float gi = sigmoidf_(gs[tid] + xn[0]), gf = sigmoidf_(gs[8 + tid] + xn[1]);
            const float gg = tanhf_(gs[16 + tid] + xn[2]), go = sigmoidf_(gs[24 + tid] + xn[3]);
            c = gf * c + gi * gg;
            const float h = go * tanhf_(c);
            __hip_atomic_store(p.hx + (s & 1) * H + u0 + tid, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.out[(long long)t * p.out_stride + p.out_off + u0 + tid] = h;
            if (s + 1 == p.T) {
                if (p.hn) p.hn[u0 + tid] = h;
                if (p.cn) p.cn[u0 + tid] = c;
            }
        }
        if (s + 1 < p.T) fetch_x(p.reverse ? t - 1 : t + 1);           // (lands while the workers meet)
        if (s + 1 == p.T) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the published values are acknowledged ...
        __syncthreads();
        if (tid == 0) {                                               // ... before this worker says so
            __hip_atomic_fetch_add(p.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)nworkers * (unsigned)(s + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(p.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) {                           // the other workers never arrived (a shared device): fail loudly
                    __hip_atomic_store(p.ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace apamd

using namespace apamd;

extern "C" int64_t ap_lstm_workspace_bytes(int32_t H) { return (int64_t)(2 * H) * 4 + 64; }

// One direction of one LSTM layer over a whole sequence.  xproj: [B][T][4H] = W_ih x + b_ih + b_hh (gate order i, f, g, o);
// whh: [4H][H]; h0 / c0: [B][H] or null; out: [B][T][out_stride], this direction's H values at column out_off; hn / cn: the
// final state or null; workspace: ap_lstm_workspace_bytes(H) bytes (only touched for H > 64).  H <= 64 with any B, or
// H in {256, 512} with B == 1; everything else is AP_ERR_UNSUPPORTED (the caller keeps the library path).
extern "C" int ap_lstm_recurrence(const float* xproj, const float* whh, const float* h0, const float* c0, float* out, float* hn, float* cn,
                                  int32_t B, int32_t T, int32_t H, int32_t reverse, int32_t out_stride, int32_t out_off,
                                  void* workspace, ap_stream_t stream_) {
    if (!xproj || !whh || !out) return fail(AP_ERR_INVALID, "lstm_recurrence: null pointer");
    if (B < 1 || T < 1 || H < 1 || out_stride < H || out_off < 0 || out_off + H > out_stride)
        return fail(AP_ERR_INVALID, "lstm_recurrence: bad sizes (B=%d T=%d H=%d stride=%d off=%d)", B, T, H, out_stride, out_off);
    hipStream_t stream = (hipStream_t)stream_;
    LstmParams p;
    p.xproj = xproj; p.whh = whh; p.h0 = h0; p.c0 = c0; p.out = out; p.hn = hn; p.cn = cn;
    p.B = B; p.T = T; p.H = H; p.reverse = reverse ? 1 : 0; p.out_stride = out_stride; p.out_off = out_off;
    p.hx = nullptr; p.ctr = nullptr;
    if (H <= 64) {
        hipLaunchKernelGGL(lstm_small_kernel, dim3(B), dim3(256), 0, stream, p);
        return check_launch("lstm_small_kernel");
    }
    if (B != 1 || (H != 256 && H != 512))
        return fail(AP_ERR_UNSUPPORTED, "lstm_recurrence: H=%d B=%d (built for H <= 64, or H in {256, 512} at batch 1)", H, B);
    if (!workspace) return fail(AP_ERR_INVALID, "lstm_recurrence: H=%d needs its workspace", H);
    p.hx = reinterpret_cast<float*>(workspace);
    p.ctr = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + (size_t)2 * H * 4);
    if (hipMemsetAsync(p.ctr, 0, 64, stream) != hipSuccess) return fail(AP_ERR_LAUNCH, "lstm_recurrence: memset");
    if (H == 512) hipLaunchKernelGGL(lstm_dist_kernel<512>, dim3(8 * (512 / 8)), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(lstm_dist_kernel<256>, dim3(8 * (256 / 8)), dim3(256), 0, stream, p);
    return check_launch("lstm_dist_kernel");
}

// 1 when the last distributed launch on this workspace gave up waiting for its peers (read after a synchronisation)
extern "C" int ap_lstm_timed_out(const void* workspace, int32_t H) {
    unsigned v = 0;
    if (!workspace || H <= 64) return 0;
    if (hipMemcpy(&v, reinterpret_cast<const char*>(workspace) + (size_t)2 * H * 4 + 4, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v != 0 ? 1 : 0;
}
