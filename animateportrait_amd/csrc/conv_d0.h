// conv_d0.h -- the PatchGAN's first layer, Conv2d(1 | 2, 64, 4, stride 2, pad 1) + LeakyReLU on a full-resolution image
// (Module2/models/networks.py:2620-2623), in plain-bf16 arithmetic on the bf16 matrix pipe.
//
// One or two input channels against 64 outputs: K = 16 taps per channel.  The implicit-GEMM kernels (conv_igemm_f32<ConvCfg<2,2,4,..>>,
// fifteen launches per train step) spend their time in per-tile prologues and epilogues -- one channel chunk per tile -- and wrote
// their 64 x 128 x 128 outputs at 2 TB/s (58 us per launch).  Here the layer is an output stream:
//   * M = 64 channels (A = the weights, [m][(ky, kx)] per input channel: two fragments per lane, built once per workgroup),
//     N = 32 output pixels of a row, K-step = one input channel: a lane's 8 k-slots are taps (ky, kx) = (2h, 0..3), (2h+1, 0..3) =
//     four consecutive pixels from input column 2 ox - 1 of rows 2 oy - 1 + 2h and 2 oy + 2h.  The input rows are staged as bf16
//     with one zero column on the left, so that the four pixels are the two dwords at dword ox of the row (ds_read2_b32);
//   * a workgroup (eight waves) owns 8 output rows of one image: 18 input rows per channel in LDS;
//   * bias + LeakyReLU on the accumulators, which are stored directly: register r of a lane = one channel, 32 lanes = 32
//     consecutive pixels = one 128-byte line (OW a multiple of 32).
#pragma once
#include "wgrad_k7.h"

namespace apamd {

struct ConvD0Params {
    const float* x;           // [N][CIN][H][W]
    const float* w;           // [64][CIN][4][4]
    const float* bias;        // [64] or null
    float* y;                 // [N][64][OH][OW]
    int N, H, W, OH, OW, act, blocks_per_img;
};

constexpr int kConvD0Rows = 8;

template <int CIN>
static __global__ __launch_bounds__(512) void conv_d0_kernel(const ConvD0Params p) {
    constexpr int SB = kConvD0Rows, IR = 2 * SB + 2, MT = 2, NTH = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char d0_smem[];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, OH = p.OH, OW = p.OW;
    const int RPB = (W + 8) * 2;                                    // bytes per staged row: element e <-> input column e - 1
    const int n = blockIdx.x / p.blocks_per_img, rb = blockIdx.x - n * p.blocks_per_img;
    const int oy0 = rb * SB, iy0 = 2 * oy0 - 1;

    // ---- stage the input rows: raw 16-byte loads first (clamped rows), then bf16 into LDS (rows outside the image: zeros)
    const int W4 = W >> 2, total = CIN * IR * W4;
    constexpr int NLD = (CIN * IR * 64 + NTH - 1) / NTH;            // W <= 256
    float4 ld[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * NTH, ic = i < total ? i : 0;
        const int rowc = ic / W4, q = ic - rowc * W4;               // rowc = ci * IR + staged row
        const int ci = rowc / IR, t = rowc - ci * IR;
        int iy = iy0 + t;
        iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
        ld[k] = *reinterpret_cast<const float4*>(p.x + (((long long)n * CIN + ci) * H + iy) * W + 4 * q);
    }
    // ---- weights as A fragments (channel ci = K-step): lane (m, half): taps (2 half, 0..3), (2 half + 1, 0..3)
    k7_bf16x8 af[MT][CIN];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float4* wr = reinterpret_cast<const float4*>(p.w + ((mt * 32 + l32) * CIN + ci) * 16 + 8 * half);
            const float4 a = wr[0], b = wr[1];
            const uint4 pk = make_uint4(k7_pack(a.x, a.y), k7_pack(a.z, a.w), k7_pack(b.x, b.y), k7_pack(b.z, b.w));
            af[mt][ci] = __builtin_bit_cast(k7_bf16x8, pk);
        }
    float bv[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[mt][r] = p.bias ? p.bias[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
    const float slope = p.act == 1 ? 0.f : (p.act == 2 ? 0.2f : 1.f);
    // the zero columns: element 0 and W + 1 .. W + 7 of every staged row
    for (int i = tid; i < CIN * IR * 8; i += NTH) {
        const int rowc = i >> 3, e = i & 7;
        reinterpret_cast<unsigned short*>(d0_smem + rowc * RPB)[e == 0 ? 0 : W + e] = 0;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * NTH;
        if (i < total) {
            const int rowc = i / W4, q = i - rowc * W4;
            const int t = rowc % IR, iy = iy0 + t;
            const bool in = iy >= 0 && iy < H;
            k7_bf16x2 lo = {(__bf16)(in ? ld[k].x : 0.f), (__bf16)(in ? ld[k].y : 0.f)};
            k7_bf16x2 hi = {(__bf16)(in ? ld[k].z : 0.f), (__bf16)(in ? ld[k].w : 0.f)};
            unsigned short* dst = reinterpret_cast<unsigned short*>(d0_smem + rowc * RPB) + 1 + 4 * q;
            const unsigned a = __builtin_bit_cast(unsigned, lo), b = __builtin_bit_cast(unsigned, hi);
            dst[0] = (unsigned short)a;                             // (odd element offset: 2-byte stores)
            dst[1] = (unsigned short)(a >> 16);
            dst[2] = (unsigned short)b;
            dst[3] = (unsigned short)(b >> 16);
        }
    }
    __syncthreads();

    const int NPB = (OW + 31) >> 5;
    for (int oy = oy0 + (wave >> 2); oy < oy0 + SB && oy < OH; oy += 2) {
        const int t = 2 * (oy - oy0) + 2 * half;                    // staged row of this half's first tap row
        for (int pb = wave & 3; pb < NPB; pb += 4) {
            const int ox = pb * 32 + l32, oxc = ox < OW ? ox : OW - 1;
            k7_bf16x8 bf[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const unsigned* r0 = reinterpret_cast<const unsigned*>(d0_smem + (ci * IR + t) * RPB) + oxc;
                const unsigned* r1 = reinterpret_cast<const unsigned*>(d0_smem + (ci * IR + t + 1) * RPB) + oxc;
                const uint4 w4 = make_uint4(r0[0], r0[1], r1[0], r1[1]);
                bf[ci] = __builtin_bit_cast(k7_bf16x8, w4);
            }
            k7_f32x16 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = bv[mt][i];
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][ci], bf[ci], acc[mt], 0, 0, 0);
            }
            if (ox < OW) {
                float* out = p.y + (((long long)n * 64 + 4 * half) * OH + oy) * OW + ox;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[mt][r];
                        out[(long long)(mt * 32 + (r & 3) + 8 * (r >> 2)) * OH * OW] = v > 0.f ? v : slope * v;
                    }
            }
        }
    }
}

}  // namespace apamd
