// wgrad_final.h -- weight gradient of the generator's last layer: ReflectionPad2d(3) + Conv2d(ngf, 1, 7)
// (Module2/models/networks.py:1277-1279).
//
//   dW[0][ci][ky][kx] = sum_{n, y, x} g[n][y][x] * pad3(act(IN(src)))[n][ci][y + ky][x + kx]
//
// One gradient channel against 64 input channels: 3136 sums over 2 M pixels each.  As a GEMM (roles swapped, M = Cin,
// N = taps) the fp32 matrix kernel needs 2.0 ms including the padded operand copy; the arithmetic is the same 6.6 GFLOP
// as the layer's forward (conv_direct.h) and runs the same way on the vector ALUs: a workgroup owns one input channel
// and a range of 32 x 64 pixel tiles; per tile the activation window (with halo; InstanceNorm + ReLU + reflection
// applied by the loader) is staged in LDS, a lane holds 1 x 4 strips of the gradient (two per tile) and the 7 x 12 window over each
// (conflict-free ds_read_b128), and adds the 4 x 49 products into 49 per-tap accumulators kept for the whole range.
// Block sums go to partial[P][Cin][49], summed in fixed order by wgrad_reduce_kernel.
#pragma once
#include "conv_igemm.h"

namespace apamd {

struct WgradFinalParams {
    SrcSeg src;               // [N][C][H][W], possibly virtual
    const float* g;           // [N][1][H][W]
    int N, C, H, W, pad_mode;
    int tiles_x, tiles_y, tiles_per_block;
    float* partial;           // [gridDim.x][C][49]
};

constexpr int kFinalStrips = 2;          // 1 x 4 strips per lane and staged tile: pixel tile = 16 kFinalStrips x 64

// grid: (P, C), 256 threads
template <bool WVEC>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_final_kernel(const WgradFinalParams p) {
    constexpr int K = 7, R = 3, NS = kFinalStrips, TH = 16 * NS, TW = 64, LPAD = 1, IH = TH + K - 1, IWS = 72, PLANE = IH * IWS, T = K * K;
    constexpr int NE = (PLANE + 255) / 256;
    __shared__ __attribute__((aligned(16))) float xbuf[2][PLANE];
    __shared__ float red[4][T];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int ci = blockIdx.y;
    const int H = p.H, W = p.W, HW = H * W;
    const int per_img = p.tiles_y * p.tiles_x, total = p.N * per_img;
    const int t0 = blockIdx.x * p.tiles_per_block;
    int t1 = t0 + p.tiles_per_block;
    if (t1 > total) t1 = total;
    const float slope = p.src.act == 1 ? 0.f : (p.src.act == 2 ? 0.2f : 1.f);

    float acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.f;
    if (t0 < t1) {
        float xr[NE];
        bool okr[NE];
        int lyx[NE];                                             // (row << 8) | column of this thread's window elements
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = tid + k * 256, ly = e / IWS;
            lyx[k] = (ly << 8) | (e - ly * IWS);
        }
        float m = 0.f, rs = 1.f;
        // The gradient strips of a tile are fetched WITH its window, one tile ahead (round 6).  They used to be loaded inside the
        // strip loop and consumed at once: two exposed global round trips per tile, ~2 us of a tile's ~2.2 us -- the kernel ran
        // at 1.03 TB/s (553 us for 0.57 GB, profiles/r05_train_hbm_per_kernel.md) with its vector ALUs idle.
        // (raw words only in the prefetch: a select on the loaded value is a use, and the compiler waits for the load right there)
        float4 gq[NS];
        float g1[NS][4];
        auto strip_ok = [&](int t, int s, int j) {
            const int r = t % per_img;
            return (r / p.tiles_x) * TH + s * 16 + ty < H && (r % p.tiles_x) * TW + tx * 4 + j < W;
        };
        // fetch the window of tile t into registers (normalisation applied at commit: m / rs belong to the tile's image)
        auto issue = [&](int t) {
            const int n = t / per_img, r = t - n * per_img;
            const int oy0 = (r / p.tiles_x) * TH, ox0 = (r % p.tiles_x) * TW;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int oy = oy0 + s * 16 + ty, ox = ox0 + tx * 4;
                if constexpr (WVEC) {                               // whole 16-byte groups (W % 4 == 0); a dead lane reads pixel 0
                    gq[s] = *reinterpret_cast<const float4*>(p.g + (long long)n * HW + ((oy < H && ox + 3 < W) ? oy * W + ox : 0));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) g1[s][j] = p.g[(long long)n * HW + ((oy < H && ox + j < W) ? oy * W + ox + j : 0)];
                }
            }
            const float* base = p.src.data + ((long long)n * p.C + ci) * HW;
            if (p.src.mean != nullptr) { m = p.src.mean[n * p.C + ci]; rs = p.src.rstd[n * p.C + ci]; }
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int e = tid + k * 256;
                int gy = oy0 - R + (lyx[k] >> 8), gx = ox0 - R - LPAD + (lyx[k] & 255);
                bool ok = e < PLANE;
                if (p.pad_mode == 1) {
                    gy = reflect_clamp(gy, H);
                    gx = reflect_clamp(gx, W);
                } else {
                    ok = ok && gy >= 0 && gy < H && gx >= 0 && gx < W;
                }
                okr[k] = ok;
                xr[k] = base[ok ? gy * W + gx : 0];
            }
        };
        auto commit = [&](float* dst) {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int e = tid + k * 256;
                float v = (xr[k] - m) * rs;
                v = v > 0.f ? v : slope * v;
                if (e < PLANE) dst[e] = okr[k] ? v : 0.f;
            }
        };
        issue(t0);
        commit(xbuf[0]);
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            const int cur = (t - t0) & 1;
            const bool more = t + 1 < t1;
            float gc[NS][4];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float raw[4] = {WVEC ? gq[s].x : g1[s][0], WVEC ? gq[s].y : g1[s][1], WVEC ? gq[s].z : g1[s][2], WVEC ? gq[s].w : g1[s][3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) gc[s][j] = strip_ok(t, s, WVEC ? 3 : j) ? raw[j] : 0.f;
            }
            if (more) issue(t + 1);
            static_assert(NS == 2, "the strip's gradient is picked with a select");
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {                       // this lane's NS gradient strips of the tile (rolled: one 84-register window)
            const float gv[4] = {s ? gc[1][0] : gc[0][0], s ? gc[1][1] : gc[0][1], s ? gc[1][2] : gc[0][2], s ? gc[1][3] : gc[0][3]};
            // window rows: whole 16-byte lanes, all in flight before one wait (see conv_direct.h)
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v q[K][3];
            {
                const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)(&xbuf[cur][(s * 16 + ty) * IWS + tx * 4]);
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const unsigned a = a0 + ky * IWS * 4;
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32"
                                 : "=&v"(q[ky][0]), "=&v"(q[ky][1]), "=&v"(q[ky][2]) : "v"(a));
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]),
                               "+v"(q[2][0]), "+v"(q[2][1]), "+v"(q[2][2]), "+v"(q[3][0]), "+v"(q[3][1]), "+v"(q[3][2]),
                               "+v"(q[4][0]), "+v"(q[4][1]), "+v"(q[4][2]), "+v"(q[5][0]), "+v"(q[5][1]), "+v"(q[5][2]),
                               "+v"(q[6][0]), "+v"(q[6][1]), "+v"(q[6][2]));
            }
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float win[12];
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    win[v * 4 + 0] = q[ky][v].x; win[v * 4 + 1] = q[ky][v].y; win[v * 4 + 2] = q[ky][v].z; win[v * 4 + 3] = q[ky][v].w;
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
                    acc[ky * K + kx] += (gv[0] * win[LPAD + kx] + gv[1] * win[LPAD + 1 + kx]) +
                                        (gv[2] * win[LPAD + 2 + kx] + gv[3] * win[LPAD + 3 + kx]);
            }
            }
            if (more) commit(xbuf[cur ^ 1]);
            __syncthreads();
        }
    }
    // block sums: lanes by halving exchanges (common.h), then the four waves in fixed order
    wave_sums_to_lds<T>(acc, tid, red[tid >> 6]);
    __syncthreads();
    if (tid < T) p.partial[((long long)blockIdx.x * p.C + ci) * T + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

}  // namespace apamd
