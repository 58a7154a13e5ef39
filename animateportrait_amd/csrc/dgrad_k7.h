// dgrad_k7.h -- data gradient of the generator's last layer, ReflectionPad2d(3) + Conv2d(ngf, 1, 7) (Module2/models/networks.py:
// 1277-1279), on the bf16 matrix pipe (plain-bf16 arithmetic): the gradient w.r.t. the PADDED input,
//
//     gp[n][c][py][px] = sum_{ky, kx} w[0][c][ky][kx] * g[n][0][py - ky][px - kx]        (py < H + 6, px < W + 6; g zero outside),
//
// which its consumer folds back over the reflection (ap_instnorm_bwd's g1_pad = 3).  One input channel against 64 outputs: on the
// fp32 matrix pipe (conv_igemm_f32<ConvCfg<2,1,7,...>>, one real channel of a two-channel chunk) the layer took 377 us for 0.56 GB
// of output; as a GEMM with M = channels, N = 32 pixels of a row, K = (ky, j) = 8 x 8 (ky = 7 and j = 7 carry zero weights) it is
// 8 MFMAs per 64 x 32 outputs and the kernel is a store stream.
//   * A (weights, [c][(ky, j)] = w[c][ky][6 - j]): 8 fragments per lane, built once per workgroup, kept in registers;
//   * B ([(ky, j)][px] = g[py - ky][px - 6 + j] = Nr[py - ky + 6][px + j], Nr = g zero-padded by 6): the rows that
//     wgrad_k7_narrow_kernel prepares for the weight gradient of the same layer (bf16, two copies one element apart), staged in
//     LDS per 8 output rows; a lane's 8 elements start at column px, on a dword in the copy of px's parity;
//   * output through an LDS row buffer, stored as whole channel rows (see the kernel).
#pragma once
#include "wgrad_k7.h"

namespace apamd {

struct DgradK7Params {
    const unsigned short* narrow;   // [N][1][A][2][NW] bf16 (A = H + 12, NW = W + 16)
    const float* w;                 // [C][7][7] = weight[0][c][ky][kx]
    float* gp;                      // [N][C][HP][WP]
    int N, C, H, W, HP, WP, A, NW, RB, blocks_per_img;
};

constexpr int kDgradK7Rows = 8;     // output rows per staging of the gradient rows
typedef float dk7_float4u __attribute__((ext_vector_type(4), aligned(4)));

// A workgroup (eight waves) owns RB rows of one image (one workgroup per CU).  Per output row the waves share the
// ceil(WP / 32) pixel blocks; the 64 x 32 accumulator tiles go to an LDS row buffer [c][WP] (double-buffered: one barrier per
// row) and leave it as WHOLE channel rows -- a wave instruction stores 1 KiB of one channel.  (Stored straight from the
// accumulators -- 128-byte runs, 24 bytes off the cache lines with WP = 262 -- the kernel wrote at 2.2 TB/s: 258 us.)
template <int MT>
static __global__ __launch_bounds__(512) void dgrad_k7_final_kernel(const DgradK7Params p) {
    constexpr int SB = kDgradK7Rows, LR = SB + 7;                   // staged rows: Nr rows py0 - 1 .. py0 + SB + 5
    constexpr int NWV = 8, NTH = NWV * 64;                          // eight waves: two per SIMD, to cover the LDS / store latencies
    constexpr int C = MT * 32, CPW = C / NWV;                       // channels; channel rows a wave stores
    extern __shared__ __attribute__((aligned(16))) unsigned char dk7_smem[];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = p.NW, WP = p.WP, HP = p.HP;
    const int NWLB = (NW + 8) * 2, ROWB = 2 * NWLB;
    const int OP = (WP + 7) & ~7;                                   // row buffer pitch, floats
    float* const obuf = reinterpret_cast<float*>(dk7_smem);         // [2][C][OP]
    unsigned char* const nar_l = dk7_smem + (size_t)2 * C * OP * 4; // [LR][2][NWLB]
    const int n = blockIdx.x / p.blocks_per_img, rb = blockIdx.x - n * p.blocks_per_img;
    const int row0 = rb * p.RB;
    const int row1 = row0 + p.RB < HP ? row0 + p.RB : HP;

    // ---- weights as A fragments: step s, lane (c, half): ky = 2 s + half, k-slot j <-> kx = 6 - j
    k7_bf16x8 af[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = mt * 32 + l32, ky = 2 * s + half;
            const bool live = ky < 7;
            const float* wr = p.w + (live ? c * 49 + ky * 7 : 0);
            float v[8];
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = wr[6 - j];
            v[7] = 0.f;
            uint4 pk = make_uint4(k7_pack(v[0], v[1]), k7_pack(v[2], v[3]), k7_pack(v[4], v[5]), k7_pack(v[6], v[7]));
            if (!live) pk = make_uint4(0u, 0u, 0u, 0u);
            af[mt][s] = __builtin_bit_cast(k7_bf16x8, pk);
        }

    const int NV = NW >> 3, total = LR * 2 * NV;
    constexpr int NLD = 2;                                          // ceil(LR * 2 * (256 + 16) / 8 / 512): W <= 256
    const int NPB = (WP + 31) >> 5, nchunks = (WP + 3) >> 2;
    int buf = 0;
    for (int py0 = row0; py0 < row1; py0 += SB) {
        // ---- stage the gradient rows of these SB output rows (raw 16-byte loads first, then the LDS writes)
        auto stage_load = [&](int k) -> uint4 {
            const int i = tid + k * NTH, ic = i < total ? i : 0;
            const int rowc = ic / NV, v = ic - rowc * NV;           // rowc = staged row * 2 + copy
            int a = py0 - 1 + (rowc >> 1);
            a = a < 0 ? 0 : (a > p.A - 1 ? p.A - 1 : a);
            return *(reinterpret_cast<const uint4*>(p.narrow + (((long long)n * p.A + a) * 2 + (rowc & 1)) * NW) + v);
        };
        auto stage_store = [&](int k, const uint4& val) {
            const int i = tid + k * NTH;
            if (i < total) {
                const int rowc = i / NV, v = i - rowc * NV;
                *(reinterpret_cast<uint4*>(nar_l + rowc * NWLB) + v) = val;
            }
        };
        static_assert(NLD == 2, "two named registers");
        const uint4 ld0 = stage_load(0), ld1 = stage_load(1);
        __syncthreads();                                            // the previous rows' fragment reads are done
        stage_store(0, ld0);
        stage_store(1, ld1);
        __syncthreads();
        const int pyE = py0 + SB < row1 ? py0 + SB : row1;
        for (int py = py0; py < pyE; ++py) {
            const int lrow = py - py0 + 7;                          // staged row of ky = 0
            float* const ob = obuf + (size_t)buf * C * OP;
            for (int pb = wave; pb < NPB; pb += NWV) {
                const int px = pb * 32 + l32, pxc = px < WP ? px : WP - 1, copy = pxc & 1;
                k7_bf16x8 bf[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned* b = reinterpret_cast<const unsigned*>(nar_l + (lrow - 2 * s - half) * ROWB + copy * NWLB + (pxc + copy) * 2);
                    const uint4 w4 = make_uint4(b[0], b[1], b[2], b[3]);
                    bf[s] = __builtin_bit_cast(k7_bf16x8, w4);
                }
                k7_f32x16 acc[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][s], bf[s], acc[mt], 0, 0, 0);
                }
                if (px < OP) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ob[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * OP + px] = acc[mt][r];
                }
            }
            __syncthreads();
            // ---- whole channel rows out: lane = a 16-byte chunk of the row; all LDS reads of the wave first, then its stores
            // (W <= 256: a row has at most 66 chunks -- the one or two beyond a wave's 64 lanes go out in one more instruction,
            // lane = (channel, chunk))
            float* const grow = p.gp + (((long long)n * C + wave * CPW) * HP + py) * WP;
            const float* const src0 = ob + wave * CPW * OP;
            const int R = nchunks > 64 ? nchunks - 64 : 0;
            float4 v[CPW];
            const int qa = lane < nchunks ? lane : 0;
#pragma unroll
            for (int cc = 0; cc < CPW; ++cc) v[cc] = *reinterpret_cast<const float4*>(src0 + cc * OP + 4 * qa);
            const int rcc = R ? lane / R : 0, rq = 64 + (R ? lane - rcc * R : 0);
            const bool ract = R > 0 && lane < CPW * R;
            const float4 vr = *reinterpret_cast<const float4*>(src0 + (ract ? rcc * OP + 4 * rq : 0));
            auto put = [&](float* dst, int q, const float4& t) {
                if (4 * q + 3 < WP) {
                    const dk7_float4u u = {t.x, t.y, t.z, t.w};
                    *reinterpret_cast<dk7_float4u*>(dst + 4 * q) = u;
                } else {
                    dst[4 * q] = t.x;
                    if (4 * q + 1 < WP) dst[4 * q + 1] = t.y;
                    if (4 * q + 2 < WP) dst[4 * q + 2] = t.z;
                }
            };
            if (lane < nchunks) {
#pragma unroll
                for (int cc = 0; cc < CPW; ++cc) put(grow + (long long)cc * HP * WP, lane, v[cc]);
            }
            if (ract) put(grow + (long long)rcc * HP * WP, rq, vr);
            buf ^= 1;
        }
    }
}


// ---- data gradient of the PatchGAN's output layer, Conv2d(C, 1, 4, stride 1, pad 1) on a small map (networks.py:2643), plain bf16:
//     gx[n][c][iy][ix] = sum_{ky, kx} w[0][c][ky][kx] * g[n][0][iy - ky + 1][ix - kx + 1]        (g: (H-1) x (W-1), zero outside)
// One gradient channel against 512 outputs of 31 x 31: the fp32 matrix kernel (conv_igemm_f32<ConvCfg<2,1,4,..>>) took 32 us for
// 31-94 MB of output.  Same form as dgrad_k7_final_kernel with K = (ky, kx) = 16 = ONE K-step: a workgroup owns 32 channels of
// one image -- their planes are one contiguous 32 H W block of the output, which is assembled in LDS (lane = flat pixel) and
// leaves as a linear copy: 20.5 us.  (16 channels per workgroup, two workgroups per CU: 23.9 us.)
struct DgradHeadParams {
    const float* g;           // [N][1][H-1][W-1]
    const float* w;           // [C][4][4] = weight[0][c][ky][kx]
    float* gx;                // [N][C][H][W]
    int N, C, H, W;
};

static __global__ __launch_bounds__(256) void dgrad_head_kernel(const DgradHeadParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dh_smem[];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, HW = H * W, GH = H - 1, GW = W - 1;
    const int cg = p.C >> 5, n = blockIdx.x / cg, c0 = (blockIdx.x - n * cg) * 32;
    const int RP = W + 8, RPB = RP * 2, LR = H + 3;                 // staged row: element e <-> gradient column e - 2
    float* const ob = reinterpret_cast<float*>(dh_smem);            // [32][HW]
    unsigned short* const gt = reinterpret_cast<unsigned short*>(dh_smem + (size_t)32 * HW * 4);   // [LR][2 copies][RP]: row t <-> gradient row t - 2
    // ---- the gradient map as zero-framed bf16 rows in two copies one element apart
    // (all of a thread's loads first, from clamped addresses, then the stores)
    const float* gsrc = p.g + (long long)n * GH * GW;
    constexpr int NI = 7;                                           // ceil((34 + 3) * (34 + 8) / 256): H W <= 1156
    float g0[NI], g1[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = tid + k * 256, ic = i < LR * RP ? i : 0;
        const int t = ic / RP, e = ic - t * RP;
        const int gy = t - 2, gx0 = e - 2;
        const int gyc = gy < 0 ? 0 : (gy > GH - 1 ? GH - 1 : gy);
        const int xa = gx0 < 0 ? 0 : (gx0 > GW - 1 ? GW - 1 : gx0), xb = gx0 - 1 < 0 ? 0 : (gx0 - 1 > GW - 1 ? GW - 1 : gx0 - 1);
        g0[k] = gsrc[gyc * GW + xa];
        g1[k] = gsrc[gyc * GW + xb];
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = tid + k * 256;
        if (i < LR * RP) {
            const int t = i / RP, e = i - t * RP;
            const int gy = t - 2, gx0 = e - 2;
            const bool rowok = gy >= 0 && gy < GH;
            const __bf16 v0 = (__bf16)((rowok && gx0 >= 0 && gx0 < GW) ? g0[k] : 0.f);
            const __bf16 v1 = (__bf16)((rowok && gx0 - 1 >= 0 && gx0 - 1 < GW) ? g1[k] : 0.f);
            gt[(t * 2) * RP + e] = __builtin_bit_cast(unsigned short, v0);        // copy 0: element e at position e
            gt[(t * 2 + 1) * RP + e] = __builtin_bit_cast(unsigned short, v1);    // copy 1: element e - 1 at position e
        }
    }
    // ---- weights: lane (c, half): k-slot j <-> (ky = 2 half + (j >> 2), kx = 3 - (j & 3))
    k7_bf16x8 af;
    {
        const float* wr = p.w + (c0 + l32) * 16 + 8 * half;
        const float4 a = reinterpret_cast<const float4*>(wr)[0], b = reinterpret_cast<const float4*>(wr)[1];
        const uint4 pk = make_uint4(k7_pack(a.w, a.z), k7_pack(a.y, a.x), k7_pack(b.w, b.z), k7_pack(b.y, b.x));
        af = __builtin_bit_cast(k7_bf16x8, pk);
    }
    __syncthreads();
    const int NB = (HW + 31) >> 5;
    for (int b = wave; b < NB; b += 4) {
        const int f = b * 32 + l32, fc = f < HW ? f : HW - 1;
        const int iy = fc / W, ix = fc - iy * W;
        // tap row ky reads gradient row iy - ky + 1 = staged row iy + 3 - ky; its four columns ix - 2 .. ix + 1 = elements ix .. ix + 3
        const int copy = ix & 1;
        const unsigned* r0 = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(gt) + ((iy + 3 - 2 * half) * 2 + copy) * RPB + (ix + copy) * 2);
        const unsigned* r1 = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(gt) + ((iy + 2 - 2 * half) * 2 + copy) * RPB + (ix + copy) * 2);
        const uint4 w4 = make_uint4(r0[0], r0[1], r1[0], r1[1]);
        const k7_bf16x8 bf = __builtin_bit_cast(k7_bf16x8, w4);
        k7_f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
        if (f < HW) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ob[((r & 3) + 8 * (r >> 2) + 4 * half) * HW + f] = acc[r];
        }
    }
    __syncthreads();
    // ---- the 32 planes are one contiguous block of the output (32 H W floats, 16-byte aligned for C % 32 == 0)
    float4* dst = reinterpret_cast<float4*>(p.gx + ((long long)n * p.C + c0) * HW);
    const float4* src = reinterpret_cast<const float4*>(ob);
    for (int i = tid; i < 8 * HW; i += 256) dst[i] = src[i];
}

}  // namespace apamd
