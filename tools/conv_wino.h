// conv_wino.h -- Winograd F(2x2, 3x3) for the wide stride-1 3x3 layers, on the split-bf16 matrix pipe.
//
// The 21 ResNet-block convolutions and the merge convolution of the generator (Module2/models/networks.py:1251,
// 2329-2421; 75 + 10 % of its FLOPs) multiply 9 taps x 3 split products per output on conv_bf16x3.  Scheduling of that
// kernel is exhausted; this path issues 2.25x fewer MFMAs instead:
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A           (Lavin & Gray; 16 products per 2x2 outputs instead of 36)
//   * wino_input_kernel  (the pass between two convolutions, as norm_split_kernel): InstanceNorm + activation
//     [+ residual], optional fp32 output, and V = B^T d B of every 4x4 input patch (stride 2, padding applied here),
//     split into bf16 head + tail:  VS[n][pos 16][part 2][C/8][T tiles][8 x bf16]  (16-byte slots);
//   * wino_pack_kernel: U = G g G^T in fp32, split, as the conv's LDS image per (cout tile, position, K stage);
//   * conv_wino: per position a [Cout x C] . [C x tiles] GEMM with the three split products (u_l v_h + u_h v_l + u_h v_h,
//     fp32 accumulation).  A workgroup owns a 128-cout x 128-tile block and walks POSITION-OUTER: one working
//     accumulator set collects M_pos over all channels, then is folded (+-1 coefficients of A^T . A) into the four
//     output accumulator sets (the 2x2 pixels of each tile), so M never leaves the registers.  Both operands are
//     staged by LDS-DMA only (four 32-channel stages of 32 KB in flight), fragments are conflict-free ds_read_b128.
//     Epilogue: bias / activation, 8-byte stores (the two x-neighbours of a tile), InstanceNorm partial sums.
// Precision (tools/wino_precision.py, whole generator, ngf = 64): L-inf 1.4e-4..1.6e-4 against the exact-fp32 result
// (direct split-bf16: 0.9e-4; budget 1e-3).
#pragma once
#include "conv_bf16x3.h"

namespace apamd {

struct WinoSeg {
    const unsigned char* vs;   // transformed split tensor of this segment (wino_input_kernel)
    int CG;                    // its channel groups (C / 8)
    int cg_begin;              // first channel group of the segment in the concatenation
};

struct WinoParams {
    WinoSeg seg[kMaxSeg];
    int nseg;
    int N, H, W;               // map size (input == output: stride 1, pad 1)
    int TW, T;                 // tile columns (W / 2), tiles per image (H / 2 * W / 2)
    int Cout;
    int kstages;               // 32-channel K stages (channels padded with zero weights)
    int cg_real;               // channel groups that exist in the sources
    const unsigned char* up;   // packed transformed weights: [cout tile][pos][kstage] blocks of W_BYTES
    const float* bias;         // Cout or null
    int act;                   // epilogue activation
    float* y;                  // N x Cout x H x W
    float* stats;              // [N * Cout][stat_tiles][2] or null
    int stat_tiles;            // px_tiles * 2
    int co_tiles, px_tiles;    // Cout / 128 (rounded up), T / 128
};

struct WinoCfg {
    static constexpr int CO_TILE = 128, PX_TILE = 128, KS = 32, KG = 4, NSTG = 4, MT = 2, NT = 2;
    static constexpr int W_SLOTS = 2 * KG * CO_TILE;           // [part][kgroup][cout]
    static constexpr int X_SLOTS = 2 * KG * PX_TILE;           // [part][kgroup][tile]
    static constexpr int STAGE = W_SLOTS + X_SLOTS;            // 16-byte slots: 32 KB
    static constexpr int W_BYTES = W_SLOTS * 16;
    static constexpr size_t lds_bytes() { return (size_t)NSTG * STAGE * 16; }
};

// a wave-uniform pointer the compiler cannot prove uniform (selected through a run-time segment index): pin it in SGPRs
__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* q) {
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const unsigned char*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}

// A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]
__device__ __forceinline__ float wino_at(int i, int a) {
    return i == 0 ? (a < 3 ? 1.f : 0.f) : (a == 0 ? 0.f : (a == 1 ? 1.f : -1.f));
}

// ABL: experiment builds only (tools/wino_bench.hip): 1 no LDS-DMA, 2 no MFMAs, 4 no fold, 8 no output stores,
// 16 weight stream only, 32 activation stream only
template <int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv_wino(const WinoParams p) {
    using C = WinoCfg;
    constexpr int MT = C::MT, NT = C::NT, KG = C::KG, STAGE = C::STAGE, W_SLOTS = C::W_SLOTS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint4* const smem = reinterpret_cast<const uint4*>(smem_raw);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wco = wave & 1, wpx = wave >> 1;
    const int T = p.T, kst = p.kstages, G = 16 * kst;

    // ---- jobs: (image, tile block, cout tile), cout tile fastest; workgroup b runs on XCD b % 8 and every XCD owns a
    // contiguous range, so the cout tiles of one tile block share their activation stream through that XCD's L2
    int job, job_end, job_step;
    {
        const int Gd = gridDim.x, b = blockIdx.x;
        const int nx = Gd < 8 ? Gd : 8;
        const int xcd = b % nx, idx = b / nx;
        const int njobs = p.N * p.px_tiles * p.co_tiles;
        const int q = njobs / nx, r = njobs % nx;
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        job_step = (Gd - xcd + nx - 1) / nx;
        job = base + idx;
        job_end = base + q + (xcd < r ? 1 : 0);
    }

    // this wave's pieces of a stage: weight pieces 4 wave .. 4 wave + 3 of the 16 (the packed block IS the LDS image), and
    // channel group `wave` of the activation image: (part, half of the 128 tiles) -- 8 wave-wide 1 KiB pieces per wave.
    // Everything that changes per stage is scalar; the per-lane offset is the constant lane * 16.
    const unsigned lane16 = (unsigned)lane * 16u;
    struct DmaCtx {
        const unsigned char* w;    // this wave's quarter of the (cout tile, position, K stage) weight block
        const unsigned char* x0;   // head plane of channel group (stage, wave) at the job's first tile
        const unsigned char* x1;   // tail plane
        unsigned wdst, xdst;
    };
    auto dma_setup = [&](int n, int cot, int tp0, int pos, int ks, int buf) __attribute__((always_inline)) {
        DmaCtx d;
        d.w = p.up + ((long long)(cot * 16 + pos) * kst + ks) * C::W_BYTES + wave * 4096;
        d.wdst = lds0 + (unsigned)(buf * STAGE) * 16u + (unsigned)wave * 4096u;
        int cg = ks * KG + wave;
        if (cg >= p.cg_real) cg = p.cg_real - 1;               // padding channels: zero weights x any finite data
        int s = 0;
        if (p.nseg > 1 && cg >= p.seg[1].cg_begin) s = 1;
        if (p.nseg > 2 && cg >= p.seg[2].cg_begin) s = 2;
        const int CG = p.seg[s].CG;
        d.x0 = uniform_ptr(p.seg[s].vs + ((((long long)(n * 16 + pos) * 2 + 0) * CG + (cg - p.seg[s].cg_begin)) * T + tp0) * 16);
        d.x1 = uniform_ptr(d.x0 + (long long)CG * T * 16);
        d.w = uniform_ptr(d.w);
        d.xdst = lds0 + (unsigned)(buf * STAGE + W_SLOTS + wave * C::PX_TILE) * 16u;
        return d;
    };
    auto dma_piece = [&](const DmaCtx& d, int j) __attribute__((always_inline)) {
        if (ABL & 1) return;
        if ((ABL & 16) && j >= 4) return;
        if ((ABL & 32) && j < 4) return;
        if (j < 4) glds16_sv(d.w + j * 1024, lane16, d.wdst + j * 1024);
        else if (j < 6) glds16_sv(d.x0 + (j - 4) * 1024, lane16, d.xdst + (j - 4) * 1024);
        else glds16_sv(d.x1 + (j - 6) * 1024, lane16, d.xdst + KG * C::PX_TILE * 16 + (j - 6) * 1024);
    };
    constexpr int NPIECE = 8;

    // fragment slots: chunk kk of a stage = channel groups (2 kk, 2 kk + 1), the half-wave picks one
    const int a_slot = half * C::CO_TILE + wco * 64 + l32;     // + (part * KG + 2 kk) * CO_TILE + m * 32
    const int b_slot = W_SLOTS + half * C::PX_TILE + wpx * 64 + l32;
    bf16x8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
    auto fetch_one = [&](int buf, int kk, int fb, int r) __attribute__((always_inline)) {
        const uint4* S = smem + buf * STAGE;
        if (r < 2 * MT) {
            const int m = r >> 1;
            const uint4* q = S + a_slot + ((r & 1) * KG + 2 * kk) * C::CO_TILE + m * 32;
            if (r & 1) al[fb][m] = *reinterpret_cast<const bf16x8*>(q);
            else ah[fb][m] = *reinterpret_cast<const bf16x8*>(q);
        } else {
            const int q_ = (r - 2 * MT) >> 1;
            const uint4* q = S + b_slot + ((r & 1) * KG + 2 * kk) * C::PX_TILE + q_ * 32;
            if (r & 1) bl[fb][q_] = *reinterpret_cast<const bf16x8*>(q);
            else bh[fb][q_] = *reinterpret_cast<const bf16x8*>(q);
        }
    };
    constexpr int NRD = 2 * (MT + NT);                           // fragment reads per chunk: 8

    f32x16 Y[2][2][MT][NT];                                      // the four pixels of each tile
    f32x16 M[MT][NT];                                            // working set: the current position

    for (; job < job_end; job += job_step) {
        const int cot = job % p.co_tiles;
        int t_ = job / p.co_tiles;
        const int pt = t_ % p.px_tiles;
        const int n = t_ / p.px_tiles;
        const int tp0 = pt * C::PX_TILE;

#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    M[m][q][r] = 0.f;
                    Y[0][0][m][q][r] = Y[0][1][m][q][r] = Y[1][0][m][q][r] = Y[1][1][m][q][r] = 0.f;
                }

        __syncthreads();                                         // the previous job's readers are done with the ring
        int ipos = 0, iks = 0;                                   // (position, K stage) of the next stage to issue
        auto issue_next = [&](int buf) __attribute__((always_inline)) {
            const DmaCtx d = dma_setup(n, cot, tp0, ipos, iks, buf);
            if (++iks == kst) { iks = 0; ++ipos; }
            return d;
        };
#pragma unroll
        for (int b = 0; b < C::NSTG; ++b) {
            const DmaCtx d = issue_next(b);
#pragma unroll
            for (int j = 0; j < NPIECE; ++j) dma_piece(d, j);
        }
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");        // stage 0 has landed (this wave's pieces)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NRD; ++r) fetch_one(0, 0, 0, r);

        int pos = 0, ks = 0;
        auto stage = [&](auto ptag, int g) __attribute__((always_inline)) {
            constexpr int P = decltype(ptag)::value;
            // product j of tile (m, q): small terms first
            auto mfma1 = [&](int fb, int m, int q, int j) __attribute__((always_inline)) {
                if (ABL & 2) { M[m][q][j] += (float)ah[fb][m][0] * (float)bh[fb][q][0] + (float)al[fb][m][1] + (float)bl[fb][q][2]; return; }
                if (j == 0) M[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[fb][m], bh[fb][q], M[m][q], 0, 0, 0);
                else if (j == 1) M[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[fb][m], bl[fb][q], M[m][q], 0, 0, 0);
                else M[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[fb][m], bh[fb][q], M[m][q], 0, 0, 0);
            };
            constexpr int NM = 3 * MT * NT;                      // 12 MFMAs per chunk
            // ---- chunk 0 (fragments in set 0); the fragments of chunk 1 are fetched under its MFMAs
#pragma unroll
            for (int r = 0; r < NRD; ++r) fetch_one(P, 1, 1, r);
#pragma unroll
            for (int i = 0; i < NM; ++i) mfma1(0, (i / 3) / NT, (i / 3) % NT, i % 3);
#pragma unroll
            for (int i = 0; i < NRD / 2; ++i) {                  // 8 reads spread over 12 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            // ---- chunk 1: every fragment of this stage is in registers once lgkmcnt drains; stage g + 1 has landed once
            // everybody is past the barrier; then ring slot P takes stage g + 4
            if (g + 3 < G) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else if (g + 2 < G) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            const bool more = g + 1 < G;
            const bool do_dma = g + C::NSTG < G;
            const bool last_of_pos = ks == kst - 1;
            DmaCtx d;
            if (do_dma) d = issue_next(P);
            // coefficients of A^T . A for this position (0 / +-1)
            const int a = pos >> 2, b = pos & 3;
            const float c00 = wino_at(0, a) * wino_at(0, b), c01 = wino_at(0, a) * wino_at(1, b);
            const float c10 = wino_at(1, a) * wino_at(0, b), c11 = wino_at(1, a) * wino_at(1, b);
            auto fold = [&](int t) __attribute__((always_inline)) {
                // the position is complete for tile t: fold M into the four output sets and clear it for the next one
                const int m = t / NT, q = t % NT;
                if (ABL & 4) { Y[0][0][m][q] = M[m][q]; return; }
                if (c00 != 0.f) Y[0][0][m][q] += c00 * M[m][q];
                if (c01 != 0.f) Y[0][1][m][q] += c01 * M[m][q];
                if (c10 != 0.f) Y[1][0][m][q] += c10 * M[m][q];
                if (c11 != 0.f) Y[1][1][m][q] += c11 * M[m][q];
#pragma unroll
                for (int r = 0; r < 16; ++r) M[m][q][r] = 0.f;
            };
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                mfma1(1, (i / 3) / NT, (i / 3) % NT, i % 3);
                if (do_dma && i < NPIECE) dma_piece(d, i);
                if (more && i >= NM - NRD) fetch_one((P + 1) & 3, 0, 0, i - (NM - NRD));
                // tile t is folded once the MFMAs of tile t + 1 are in the pipe (the matrix pipe keeps running)
                if (last_of_pos && i % 3 == 2 && i >= 5) fold(i / 3 - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (last_of_pos) fold(MT * NT - 1);
            if (++ks == kst) { ks = 0; ++pos; }
        };
        for (int g = 0; g < G; g += 4) {
            stage(std::integral_constant<int, 0>{}, g);
            stage(std::integral_constant<int, 1>{}, g + 1);
            stage(std::integral_constant<int, 2>{}, g + 2);
            stage(std::integral_constant<int, 3>{}, g + 3);
        }

        // ---- epilogue.  MFMA C/D layout: column = lane & 31 = tile (tx within one tile row when TW % 32 == 0), row
        // (cout) = (r & 3) + 8 (r >> 2) + 4 half.  A lane holds the 2 x 2 pixels of its tile: two 8-byte stores per cout.
        const int co_base = cot * C::CO_TILE + wco * 64;
        const bool want_stats = p.stats != nullptr;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float s[16], sq[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = sq[r] = 0.f;
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int tp = tp0 + wpx * 64 + q * 32 + l32;
                const int ty = tp / p.TW, tx = tp - ty * p.TW;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co_base + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co >= p.Cout) continue;
                    const float bv = p.bias != nullptr ? p.bias[co] : 0.f;
                    float* dst = p.y + (((long long)n * p.Cout + co) * p.H + 2 * ty) * p.W + 2 * tx;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float v0 = Y[i][0][m][q][r] + bv, v1 = Y[i][1][m][q][r] + bv;
                        s[r] += v0 + v1;
                        sq[r] += v0 * v0 + v1 * v1;
                        if (!(ABL & 8) || v0 == 123.456f)
                            *reinterpret_cast<float2*>(dst + i * p.W) = make_float2(apply_act(v0, p.act), apply_act(v1, p.act));
                    }
                }
            }
            if (want_stats) {
                // transpose-reduce over the 32 lanes of a half-wave: after step k a lane keeps 16 >> k of the 16 rows
                // (lane bit 4-k picks which half), so 16 + 8 + 4 + 2 + 1 exchanges instead of 16 x 5; the last step
                // leaves row R(l32) complete in lanes l32 and l32 ^ 1
                float vs_[16], vq_[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { vs_[r] = s[r]; vq_[r] = sq[r]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cnt = 8 >> k, bit = 16 >> k;
                    const bool up = (l32 & bit) != 0;
#pragma unroll
                    for (int j = 0; j < cnt; ++j) {
                        // the lane keeps rows [up ? cnt : 0, ...) of its current set and sends the other half
                        const float ks_ = up ? vs_[cnt + j] : vs_[j], gs_ = up ? vs_[j] : vs_[cnt + j];
                        const float kq_ = up ? vq_[cnt + j] : vq_[j], gq_ = up ? vq_[j] : vq_[cnt + j];
                        vs_[j] = ks_ + __shfl_xor(gs_, bit, 64);
                        vq_[j] = kq_ + __shfl_xor(gq_, bit, 64);
                    }
                }
                vs_[0] += __shfl_xor(vs_[0], 1, 64);
                vq_[0] += __shfl_xor(vq_[0], 1, 64);
                if ((l32 & 1) == 0) {
                    // row index accumulated from the kept halves: bit 4 of l32 -> row bit 3, ... bit 1 -> row bit 0
                    const int r = ((l32 >> 4) & 1) * 8 + ((l32 >> 3) & 1) * 4 + ((l32 >> 2) & 1) * 2 + ((l32 >> 1) & 1);
                    const int co = co_base + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < p.Cout) {
                        float* d = p.stats + (((long long)n * p.Cout + co) * p.stat_tiles + pt * 2 + wpx) * 2;
                        d[0] = vs_[0];
                        d[1] = vq_[0];
                    }
                }
            }
        }
    }
}

// ---- weights: U = G g G^T (fp32), split, as the LDS image per (cout tile, position, K stage):
//   out[cot][pos][ks][part][kg 0..3][cout 0..127][8]   (bf16)
// G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]]
struct WinoPackParams {
    const float* w;            // OIHW (layout 0) or IOHW (layout 1), 3 x 3
    unsigned short* out;
    int Cin, Cout, layout, flip;
    int kstages, co_tiles;
};

static __global__ void wino_pack_kernel(const WinoPackParams p) {
    // one thread per (cot, ks, kg, cout, c): computes the 16 positions of one (cout, cin) filter
    const long long total = (long long)p.co_tiles * p.kstages * 4 * 128 * 8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        const int c = (int)(r % 8); r /= 8;
        const int col = (int)(r % 128); r /= 128;
        const int kg = (int)(r % 4); r /= 4;
        const int ks = (int)(r % p.kstages);
        const int cot = (int)(r / p.kstages);
        const int co = cot * 128 + col, cin = ks * 32 + kg * 8 + c;
        float g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float v = 0.f;
                if (co < p.Cout && cin < p.Cin) {
                    const int yy = p.flip ? 2 - ky : ky, xx = p.flip ? 2 - kx : kx;
                    const long long off = p.layout == 0 ? (((long long)co * p.Cin + cin) * 3 + yy) * 3 + xx
                                                        : (((long long)cin * p.Cout + co) * 3 + yy) * 3 + xx;
                    v = p.w[off];
                }
                g[ky][kx] = v;
            }
        float t[4][3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            t[0][kx] = g[0][kx];
            t[1][kx] = 0.5f * ((g[0][kx] + g[2][kx]) + g[1][kx]);
            t[2][kx] = 0.5f * ((g[0][kx] + g[2][kx]) - g[1][kx]);
            t[3][kx] = g[2][kx];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float u[4];
            u[0] = t[a][0];
            u[1] = 0.5f * ((t[a][0] + t[a][2]) + t[a][1]);
            u[2] = 0.5f * ((t[a][0] + t[a][2]) - t[a][1]);
            u[3] = t[a][2];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                __bf16 h, l;
                split_bf16(u[b], h, l);
                const long long blk = ((long long)(cot * 16 + a * 4 + b) * p.kstages + ks) * (WinoCfg::W_BYTES / 2);
                const long long o = blk + ((0 * 4 + kg) * 128 + col) * 8 + c;
                p.out[o] = __builtin_bit_cast(unsigned short, h);
                p.out[o + 4 * 128 * 8] = __builtin_bit_cast(unsigned short, l);
            }
        }
    }
}

// ---- activations: the pass between two convolutions, Winograd form.
//   v = act((x - mean) * rstd) [+ (res - res_mean) * res_rstd]      (as norm_split_kernel; statistics finalised from the
//                                                                     producer's partial tiles when given)
//   y  = v (fp32, optional)
//   VS[n][pos][part][cg][tile] = split(B^T d B),  d = the 4 x 4 patch of pad1(v) at rows 2 ty - 1.., columns 2 tx - 1..
// B^T = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
// grid: (T / 256, C / 8, N); a workgroup owns 256 consecutive tiles (R = 256 / TW tile rows) of 8 channels and stages
// their 2 R + 2 input rows through LDS.
struct WinoInParams {
    const float* x;
    const float* mean;
    const float* rstd;
    const float* partials;
    int tiles;
    double inv_count;
    float eps;
    float* mean_out;
    float* rstd_out;
    int act;
    const float* res;
    const float* res_mean;
    const float* res_rstd;
    float* y;
    uint4* vs;
    int pad_mode;              // 0 zero, 1 reflect
    int N, C, H, W, TW, T;
};

__global__ __launch_bounds__(256) void wino_input_kernel(const WinoInParams p) {
    extern __shared__ __attribute__((aligned(16))) float wi_smem[];
    __shared__ float s_m[8], s_r[8];
    const int tid = threadIdx.x, cg = blockIdx.y, n = blockIdx.z;
    const int C = p.C, H = p.H, W = p.W, HW = H * W, TW = p.TW;
    const int R = 256 / TW;                                  // tile rows of this workgroup
    const int ty0 = blockIdx.x * R;
    const int rows = 2 * R + 2;                              // staged input rows: 2 ty0 - 1 .. 2 ty0 + 2 R
    const int LW = W + 4;                                    // LDS row: columns -1 .. W (+2 spare), 8-byte aligned pairs at even c
    const bool normed = p.partials != nullptr || p.mean != nullptr;
    if (p.partials != nullptr) {
        if (tid < 64) {
            const int c = tid >> 3, sub = tid & 7;
            const float2* pp = reinterpret_cast<const float2*>(p.partials) + ((long long)n * C + cg * 8 + c) * p.tiles;
            double s = 0.0, q = 0.0;
            for (int t = sub; t < p.tiles; t += 8) {
                const float2 v = pp[t];
                s += (double)v.x;
                q += (double)v.y;
            }
#pragma unroll
            for (int sh = 1; sh < 8; sh <<= 1) {
                s += __shfl_xor(s, sh, 64);
                q += __shfl_xor(q, sh, 64);
            }
            if (sub == 0) {
                const double m = s * p.inv_count;
                double var = q * p.inv_count - m * m;
                var = var > 0.0 ? var : 0.0;
                s_m[c] = (float)m;
                s_r[c] = (float)(1.0 / sqrt(var + (double)p.eps));
            }
        }
        __syncthreads();
        if (blockIdx.x == 0 && tid < 8) {
            p.mean_out[n * C + cg * 8 + tid] = s_m[tid];
            p.rstd_out[n * C + cg * 8 + tid] = s_r[tid];
        }
    } else if (p.mean != nullptr) {
        if (tid < 8) {
            s_m[tid] = p.mean[n * C + cg * 8 + tid];
            s_r[tid] = p.rstd[n * C + cg * 8 + tid];
        }
        __syncthreads();
    }
    // ---- stage rows: LDS [c][row][LW], column index + 1 (so column -1 sits at 0); 16-byte loads of 4 pixels
    const int W4 = W >> 2;
    const int items = 8 * rows * W4;
    for (int it = tid; it < items; it += 256) {
        const int x4 = it % W4;
        int t2 = it / W4;
        const int lr = t2 % rows, c = t2 / rows;
        int gy = 2 * ty0 - 1 + lr;
        bool ok = true;
        if (p.pad_mode == 1) gy = gy < 0 ? -gy : (gy >= H ? 2 * (H - 1) - gy : gy);
        else ok = gy >= 0 && gy < H;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            const long long off = ((long long)n * C + cg * 8 + c) * HW + (long long)gy * W + x4 * 4;
            v = *reinterpret_cast<const float4*>(p.x + off);
            const float m = normed ? s_m[c] : 0.f, r = normed ? s_r[c] : 1.f;
            float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = normed ? (vv[j] - m) * r : vv[j];
                t = p.act == 1 ? fmaxf(t, 0.f) : (p.act == 2 ? (t > 0.f ? t : 0.2f * t) : t);
                vv[j] = t;
            }
            if (p.res != nullptr) {
                const float4 rv = *reinterpret_cast<const float4*>(p.res + off);
                float rm = 0.f, rr = 1.f;
                if (p.res_mean != nullptr) { rm = p.res_mean[n * C + cg * 8 + c]; rr = p.res_rstd[n * C + cg * 8 + c]; }
                vv[0] += (rv.x - rm) * rr; vv[1] += (rv.y - rm) * rr; vv[2] += (rv.z - rm) * rr; vv[3] += (rv.w - rm) * rr;
            }
            v = make_float4(vv[0], vv[1], vv[2], vv[3]);
            // rows this workgroup owns (not the halo) go out as the fp32 tensor
            if (p.y != nullptr && lr >= 1 && lr <= 2 * R) *reinterpret_cast<float4*>(p.y + off) = v;
        }
        float* d = wi_smem + (c * rows + lr) * LW + 1 + x4 * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // border columns -1 and W
    for (int it = tid; it < 8 * rows; it += 256) {
        float* d = wi_smem + it * LW;
        d[0] = p.pad_mode == 1 ? d[2] : 0.f;               // column -1 <- column 1
        d[W + 1] = p.pad_mode == 1 ? d[W - 1] : 0.f;       // column W <- column W - 2
    }
    __syncthreads();
    // ---- one tile per thread
    const int tyl = tid / TW, tx = tid - tyl * TW;
    const int tile = (ty0 + tyl) * TW + tx;
    bf16x8 vh[16], vl[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* s = wi_smem + (c * rows + 2 * tyl + r) * LW + 2 * tx;      // columns 2 tx - 1 .. 2 tx + 2 (index + 1)
            const float2 a = *reinterpret_cast<const float2*>(s), b = *reinterpret_cast<const float2*>(s + 2);
            d[r][0] = a.x; d[r][1] = a.y; d[r][2] = b.x; d[r][3] = b.y;
        }
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j];
            t[1][j] = d[1][j] + d[2][j];
            t[2][j] = d[2][j] - d[1][j];
            t[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float v0 = t[a][0] - t[a][2], v1 = t[a][1] + t[a][2], v2 = t[a][2] - t[a][1], v3 = t[a][1] - t[a][3];
            __bf16 h, l;
            split_bf16(v0, h, l); vh[a * 4 + 0][c] = h; vl[a * 4 + 0][c] = l;
            split_bf16(v1, h, l); vh[a * 4 + 1][c] = h; vl[a * 4 + 1][c] = l;
            split_bf16(v2, h, l); vh[a * 4 + 2][c] = h; vl[a * 4 + 2][c] = l;
            split_bf16(v3, h, l); vh[a * 4 + 3][c] = h; vl[a * 4 + 3][c] = l;
        }
    }
    const int CG = C >> 3;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        uint4* o = p.vs + ((((long long)(n * 16 + pos) * 2 + 0) * CG + cg) * p.T + tile);
        *reinterpret_cast<bf16x8*>(o) = vh[pos];
        *reinterpret_cast<bf16x8*>(o + (long long)CG * p.T) = vl[pos];
    }
}

}  // namespace apamd
