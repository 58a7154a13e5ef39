// wgrad_k7.h -- weight gradients of the generator's 7x7 edge layers at full resolution on the bf16 matrix pipe (plain-bf16
// arithmetic, AP_PRECISION_BF16): the three stems ReflectionPad2d(3) + Conv2d(3, ngf | ngf/2, 7) and the last layer
// ReflectionPad2d(3) + Conv2d(ngf, 1, 7) (Module2/models/networks.py:1251-1279).
//
// Both are the product of a WIDE tensor (32 / 64 channels, 2 M pixels per channel: streamed from HBM exactly once) with the 49
// shifted views of a NARROW one (3 channels / 1 channel):
//
//     dWn[m][c][ky][kx] = sum_{n, (r, q) in D}  Wd[n][m][r][q] * Nr[n][c][r + ky][q + kx]
//
//   stem form:   D = H x W,           Wd = the layer's output gradient,         Nr = the reflection-padded input (pad 3);
//                dW[m][c][ky][kx] = dWn[m][c][ky][kx]
//   final form:  D = (H+6) x (W+6),   Wd = the reflection-padded input (with its InstanceNorm + activation applied on the fly),
//                Nr = the one-channel output gradient zero-padded by 6;         dW[0][m][ky][kx] = dWn[m][0][6-ky][6-kx]
//
// i.e. a GEMM with M = wide channels, N = (c, ky, kx) taps (147 -> 160 / 49 -> 64 columns), K = pixels.  On the fp32 pipe the stem
// form took 417 us per layer (wgrad_igemm_f32<WgradCfg<1,7,1,3,2>>) and the final form 552 us on the vector ALUs
// (wgrad_final.h) -- both compute-bound -- while the matrix work is 20 us of bf16 MFMAs: this kernel is bound by the one pass
// over the wide tensor.
//
// A workgroup owns RB consecutive rows r of one image.  Per row ("tile"):
//   * wide: each wave fetches MT*8 channel rows with one coalesced 16-byte load per lane and row (raw loads, TWO tiles ahead, in
//     registers -- nothing between the loads: DESIGN 3.5), converts to bf16 and stores them as the LDS tile [m][x] (row stride
//     2 W + 16 bytes: the A fragment -- lane = channel, 8 consecutive pixels -- is one conflict-free ds_read_b128);
//   * narrow: prepared once per launch by wgrad_k7_narrow_kernel as bf16 rows in TWO copies, the second shifted by one
//     element, so that the B fragment -- lane = tap, 8 consecutive pixels from column 16 s + 8 h + kx (+3) -- starts on a dword
//     in one of them whatever the parity of kx: four dword reads (ds_read2_b32 x 2), no funnel shifts.  A ring of 8 rows in
//     LDS; a tile adds one row;
//   * the W/16 K-steps of a row are dealt to the four waves (each wave holds the whole MT x NT accumulator tile; the waves are
//     summed through LDS at the end, in fixed order);
//   * final form: the six reflected columns q = 0..2, W+3..W+5 of D are one extra K-step whose A fragment comes from a
//     per-channel border block written with the tile and whose B fragment is gathered element by element.
// Partial sums leave in accumulator order, partial[workgroup][tile][reg][lane]; wgrad_k7_reduce_kernel adds the workgroups in
// fixed order and scatters into OIHW.
#pragma once
#include <type_traits>
#include <utility>

#include "conv_igemm.h"

namespace apamd {

typedef __bf16 k7_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 k7_bf16x2 __attribute__((ext_vector_type(2)));
typedef float k7_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned k7_pack(float a, float b) {
    k7_bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

struct K7NarrowParams {
    const float* src;         // [N][CN][H][W]
    unsigned* dst;            // [N][CN][A][2][NW] bf16, as dwords
    int N, CN, H, W, A, NW, final_form;
};

// Nr rows as the main kernel reads them: copy 0 holds element e at position e, copy 1 at position e + 1.
static __global__ __launch_bounds__(256) void wgrad_k7_narrow_kernel(const K7NarrowParams p) {
    const int D = p.NW / 2;
    const long long total = (long long)p.N * p.CN * p.A * D;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < total; j += (long long)gridDim.x * 256) {
        const int d = (int)(j % D);
        const long long row = j / D;                                // (n * CN + c) * A + a
        const int a = (int)(row % p.A);
        const long long nc = row / p.A;
        const float* plane = p.src + nc * p.H * p.W;
        auto val = [&](int b) -> float {
            if (b < 0) return 0.f;
            if (p.final_form) {
                const int y = a - 6, x = b - 6;
                return (y >= 0 && y < p.H && x >= 0 && x < p.W) ? plane[y * p.W + x] : 0.f;
            }
            if (b >= p.W + 6) return 0.f;
            return plane[reflect_clamp(a - 3, p.H) * p.W + reflect_clamp(b - 3, p.W)];
        };
        const float em = val(2 * d - 1), e0 = val(2 * d), e1 = val(2 * d + 1);
        p.dst[row * 2 * D + d] = k7_pack(e0, e1);
        p.dst[(row * 2 + 1) * D + d] = k7_pack(em, e0);
    }
}

// Form 2 (PatchGAN first layer): rows a <-> input row a - 1 (zero outside), planes (phase, copy): phase 0 holds the even columns,
// phase 1 the odd ones, element j <-> column 2 (j - 1) + phase (zero outside); copy 1 is copy 0 shifted by one element.
static __global__ __launch_bounds__(256) void wgrad_d0_narrow_kernel(const K7NarrowParams p) {
    const int D = p.NW / 2;
    const long long total = (long long)p.N * p.CN * p.A * 2 * D;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < total; j += (long long)gridDim.x * 256) {
        const int d = (int)(j % D);
        const long long t = j / D;
        const int phase = (int)(t & 1);
        const long long row = t >> 1;                               // (n * CN + c) * A + a
        const int iy = (int)(row % p.A) - 1;
        const float* plane = p.src + (row / p.A) * p.H * p.W;
        auto val = [&](int e) -> float {
            const int x = 2 * (e - 1) + phase;
            return (e >= 1 && iy >= 0 && iy < p.H && x < p.W) ? plane[iy * p.W + x] : 0.f;
        };
        const float em = val(2 * d - 1), e0 = val(2 * d), e1 = val(2 * d + 1);
        p.dst[((row * 2 + phase) * 2) * D + d] = k7_pack(e0, e1);
        p.dst[((row * 2 + phase) * 2 + 1) * D + d] = k7_pack(em, e0);
    }
}

struct WgradK7Params {
    const float* wide;        // [N][MW][H][W] (fp32, or bf16 values: wgrad_k7_kernel<.., WB16>)
    const float* wmean;       // [N*MW] or null: InstanceNorm of the wide operand ...
    const float* wrstd;
    int wact;                 // ... and its activation (AP_ACT_*)
    const unsigned short* narrow;   // wgrad_k7_narrow_kernel's rows
    int N, MW, CN, H, W, R, A, NW, RB, blocks_per_img;
    float* partial;           // [gridDim.x][MT*NT][16][64]
};

// WB16: the wide tensor holds bf16 values (the stems' gradient as ap_instnorm_bwd stores it on request): half the bytes, no conversion
// FORM 0: stem, 1: final (see the head of the file); 2: the PatchGAN's first layer Conv2d(1 | 2, 64, 4, stride 2, pad 1)
// (networks.py:2620-2623): D = OH x OW, Wd = the layer's output gradient, and the narrow operand is the input image read at
// (2 r + ky - 1, 2 q + kx - 1) -- its rows are prepared by wgrad_d0_narrow_kernel as the EVEN and the ODD columns apart (two copies
// each), so that a lane's 8 pixels of tap kx are contiguous again; a tile advances the ring by two input rows.
template <int MT, int NT, int FORM, bool WB16 = false>
static __global__ __launch_bounds__(256, 1) void wgrad_k7_kernel(const WgradK7Params p) {
    constexpr bool FINAL = FORM == 1, D0 = FORM == 2;
    static_assert(!WB16 || FORM == 0, "bf16 wide operand: the stem form");
    constexpr int TAPS = D0 ? 16 : 49, KW = D0 ? 4 : 7, KH = KW;   // taps per narrow channel
    constexpr int NPL = D0 ? 4 : 2;                                 // planes of a narrow row: copies (x phases)
    constexpr int RS = D0 ? 2 : 1;                                  // narrow rows a tile advances by
    constexpr int NQ = D0 ? 2 : 1;                                  // 16-byte narrow loads per thread and tile
    constexpr int CPW = MT * 8;                                     // channel rows per wave and tile
    using QT = std::conditional_t<WB16, uint2, float4>;             // four pixels of a channel row, as loaded
    extern __shared__ __attribute__((aligned(16))) unsigned char k7_smem[];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, NW = p.NW, CN = p.CN, A = p.A;
    const int WSB = (W + 8) * 2;                                    // wide tile: bytes per channel row
    const int NWLB = (NW + 8) * 2;                                  // narrow ring: bytes per (channel, copy) row
    const int SLOT = CN * NPL * NWLB;                               // ... per ring slot
    unsigned char* const wide_l = k7_smem;                          // [MT*32][WSB]
    unsigned char* const bord_l = wide_l + MT * 32 * WSB;           // [MT*32][16]: the reflected columns (final form)
    unsigned char* const nar_l = bord_l + MT * 32 * 16;             // [8][CN][NPL][NWLB]
    const int n = blockIdx.x / p.blocks_per_img, rb = blockIdx.x - n * p.blocks_per_img;
    const int r0 = rb * p.RB;
    const int r1 = r0 + p.RB < p.R ? r0 + p.RB : p.R;
    const int S = W >> 4;                                           // K-steps of a row

    // ---- per-lane constants of the B fragments: tap (c, ky, kx) of column nt*32 + l32
    int nb_off[NT], nky[NT], nkx[NT], ncc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int tap = nt * 32 + l32;
        if (tap >= CN * TAPS) tap = CN * TAPS - 1;                  // padding columns: any legal address, the sums are dropped
        const int c = tap / TAPS, t = tap - c * TAPS, ky = t / KW, kx = t - ky * KW;
        // first of the lane's 8 elements in its plane row.  D0: tap kx reads the odd columns from q - 1 (kx 0) / q (kx 2) and the
        // even ones from q (kx 1) / q + 1 (kx 3); the phase rows carry one zero element on the left
        const int c0 = 8 * half + (D0 ? (kx + 1) >> 1 : (FINAL ? 3 : 0) + kx), copy = c0 & 1;
        const int plane = D0 ? (1 - (kx & 1)) * 2 + copy : copy;
        nb_off[nt] = (c * NPL + plane) * NWLB + (c0 + copy) * 2;
        nky[nt] = ky;
        nkx[nt] = kx;
        ncc[nt] = c;
    }
    // InstanceNorm constants of this wave's channels (one image per workgroup)
    // (final form only: the stems' wide operand is a gradient, plain by contract)
    float wm[CPW], wr[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        wm[i] = 0.f;
        wr[i] = 1.f;
        if (FINAL && p.wmean != nullptr) {
            wm[i] = p.wmean[n * p.MW + wave * CPW + i];
            wr[i] = p.wrstd[n * p.MW + wave * CPW + i];
        }
    }
    const float slope = p.wact == 1 ? 0.f : (p.wact == 2 ? 0.2f : 1.f);

    k7_f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    // ---- narrow ring: RS rows per tile, each CN x NPL plane rows of NW elements; 16 bytes per thread and load
    const int NV = NW >> 3, per_row = CN * NPL * NV;
    // (per-load values as named scalars -- indexed arrays of them were kept in scratch memory; load 1 exists in form 2 only)
    auto nq_decode = [&](int j, bool& act, int& row, int& plane, int& k) {
        const int e = tid + j * 256;
        act = e < RS * per_row && j < NQ;
        const int ec = act ? e : 0;
        row = ec / per_row;
        const int rem = ec - row * per_row;
        plane = rem / NV;
        k = rem - plane * NV;
    };
    bool nact0, nact1;
    int nrow0, nrow1, nplane0, nplane1, nk0, nk1;
    nq_decode(0, nact0, nrow0, nplane0, nk0);
    nq_decode(1, nact1, nrow1, nplane1, nk1);
    auto narrow_src = [&](int a, int j) {
        const int pl = j ? nplane1 : nplane0, k = j ? nk1 : nk0;
        return reinterpret_cast<const uint4*>(p.narrow + ((((long long)n * CN + pl / NPL) * A + a) * NPL + pl % NPL) * NW) + k;
    };
    auto narrow_dst = [&](int a, int j) { return reinterpret_cast<uint4*>(nar_l + (a & 7) * SLOT + (j ? nplane1 : nplane0) * NWLB) + (j ? nk1 : nk0); };
    // the rows tile r adds to the ring: RS r + KH - RS .. RS r + KH - 1
    auto new_row = [&](int r, int j) { return RS * r + KH - RS + (j ? nrow1 : nrow0); };

    QT q[2][CPW];
    uint4 nq00, nq01, nq10, nq11;                                   // [set][load]
    const bool wact_lane = 4 * lane < W;
    auto issue = [&](auto setc, int r) __attribute__((always_inline)) {
        constexpr int set = decltype(setc)::value;
        const int srow = FINAL ? reflect_clamp(r - 3, H) : r;
        const long long e0 = (((long long)n * p.MW + wave * CPW) * H + srow) * W + (wact_lane ? 4 * lane : 0);
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            if constexpr (WB16) q[set][i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.wide) + e0 + (long long)i * H * W);
            else q[set][i] = *reinterpret_cast<const float4*>(p.wide + e0 + (long long)i * H * W);
        }
        if constexpr (set == 0) {
            nq00 = *narrow_src(new_row(r, 0), 0);
            if constexpr (NQ > 1) nq01 = *narrow_src(new_row(r, 1), 1);
        } else {
            nq10 = *narrow_src(new_row(r, 0), 0);
            if constexpr (NQ > 1) nq11 = *narrow_src(new_row(r, 1), 1);
        }
    };
    auto commit = [&](auto setc, int r) __attribute__((always_inline)) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int ch = wave * CPW + i;
            if constexpr (WB16) {
                if (wact_lane) *reinterpret_cast<uint2*>(wide_l + ch * WSB + 8 * lane) = q[set][i];
            } else {
                float v[4] = {q[set][i].x, q[set][i].y, q[set][i].z, q[set][i].w};
                if constexpr (FINAL) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = (v[k] - wm[i]) * wr[i];
                        v[k] = t > 0.f ? t : slope * t;
                    }
                }
                if (wact_lane) *reinterpret_cast<uint2*>(wide_l + ch * WSB + 8 * lane) = make_uint2(k7_pack(v[0], v[1]), k7_pack(v[2], v[3]));
                if constexpr (FINAL) {
                    // border block: k-slots 0..2 = columns 3, 2, 1 (q = 0, 1, 2); 4..6 = columns W-2, W-3, W-4 (q = W+3, W+4, W+5)
                    if (lane == 0) *reinterpret_cast<uint2*>(bord_l + ch * 16) = make_uint2(k7_pack(v[3], v[2]), k7_pack(v[1], 0.f));
                    if (lane == (W >> 2) - 1) *reinterpret_cast<uint2*>(bord_l + ch * 16 + 8) = make_uint2(k7_pack(v[2], v[1]), k7_pack(v[0], 0.f));
                }
            }
        }
        if (nact0) *narrow_dst(new_row(r, 0), 0) = set == 0 ? nq00 : nq10;
        if constexpr (NQ > 1) {
            if (nact1) *narrow_dst(new_row(r, 1), 1) = set == 0 ? nq01 : nq11;
        }
    };
    auto compute = [&](int r) __attribute__((always_inline)) {
        const unsigned char* bp[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bp[nt] = nar_l + ((RS * r + nky[nt]) & 7) * SLOT + nb_off[nt];
        const unsigned char* ap = wide_l + l32 * WSB + 16 * half;
        for (int s = wave; s < S; s += 4) {
            k7_bf16x8 af[MT], bf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const k7_bf16x8*>(ap + mt * 32 * WSB + 32 * s);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned* b = reinterpret_cast<const unsigned*>(bp[nt] + 32 * s);
                const uint4 w = make_uint4(b[0], b[1], b[2], b[3]);
                bf[nt] = __builtin_bit_cast(k7_bf16x8, w);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
        if constexpr (FINAL) {
            if (wave == (S & 3)) {                                  // the reflected columns: one more K-step, k-slots 8..15 empty
                k7_bf16x8 af[MT], bf[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    uint4 w = *reinterpret_cast<const uint4*>(bord_l + (mt * 32 + l32) * 16);
                    if (half) w = make_uint4(0u, 0u, 0u, 0u);
                    af[mt] = __builtin_bit_cast(k7_bf16x8, w);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const unsigned short* row = reinterpret_cast<const unsigned short*>(nar_l + ((r + nky[nt]) & 7) * SLOT + ncc[nt] * 2 * NWLB);
                    const int kx = nkx[nt];
                    const unsigned e0 = row[kx], e1 = row[1 + kx], e2 = row[2 + kx];
                    const unsigned e4 = row[W + 3 + kx], e5 = row[W + 4 + kx], e6 = row[W + 5 + kx];
                    uint4 w = make_uint4(e0 | (e1 << 16), e2, e4 | (e5 << 16), e6);
                    if (half) w = make_uint4(0u, 0u, 0u, 0u);
                    bf[nt] = __builtin_bit_cast(k7_bf16x8, w);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;

    // ---- prologue: the six ring rows under the first tile's, then two tiles of loads in flight
    {
        // (rows RS r0 .. RS r0 + KH - RS - 1, in groups of RS as the tiles load them: the group of "tile" r0 - g)
        // (named registers: an array here was kept in scratch memory)
        if constexpr (D0) {
            const uint4 pre0 = *narrow_src(RS * r0 + nrow0, 0), pre1 = *narrow_src(RS * r0 + nrow1, 1);
            issue(Set0{}, r0);
            issue(Set1{}, r0 + 1 < r1 ? r0 + 1 : r1 - 1);
            if (nact0) *narrow_dst(RS * r0 + nrow0, 0) = pre0;
            if (nact1) *narrow_dst(RS * r0 + nrow1, 1) = pre1;
        } else {
            const uint4 pre0 = *narrow_src(r0, 0), pre1 = *narrow_src(r0 + 1, 0), pre2 = *narrow_src(r0 + 2, 0);
            const uint4 pre3 = *narrow_src(r0 + 3, 0), pre4 = *narrow_src(r0 + 4, 0), pre5 = *narrow_src(r0 + 5, 0);
            issue(Set0{}, r0);
            issue(Set1{}, r0 + 1 < r1 ? r0 + 1 : r1 - 1);
            if (nact0) {
                *narrow_dst(r0, 0) = pre0;
                *narrow_dst(r0 + 1, 0) = pre1;
                *narrow_dst(r0 + 2, 0) = pre2;
                *narrow_dst(r0 + 3, 0) = pre3;
                *narrow_dst(r0 + 4, 0) = pre4;
                *narrow_dst(r0 + 5, 0) = pre5;
            }
        }
    }
    // a tile: registers -> LDS, loads of the tile after next, K-steps.  Row indices beyond the block are clamped (a repeated
    // load, a ring slot nobody reads) so that the load sequence has no branch in it.
    auto tile = [&](auto setc, int r) __attribute__((always_inline)) {
        const int rc = r < r1 ? r : r1 - 1;
        commit(setc, rc);
        __syncthreads();
        issue(setc, r + 2 < r1 ? r + 2 : r1 - 1);
        if (r < r1) compute(r);
        __syncthreads();
    };
    for (int r = r0; r < r1; r += 2) {
        tile(Set0{}, r);
        tile(Set1{}, r + 1);
    }

    // ---- the four waves' accumulators, summed in fixed order (wave 0 + 1 + 2 + 3) through LDS
    float* const red = reinterpret_cast<float*>(k7_smem);
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) red[((mt * NT + nt) * 16 + i) * 64 + lane] = acc[mt][nt][i];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][nt][i] += red[((mt * NT + nt) * 16 + i) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* out = p.partial + (long long)blockIdx.x * (MT * NT * 1024) + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) out[((mt * NT + nt) * 16 + i) * 64] = acc[mt][nt][i];
    }
}

// dW = sum over the P workgroups' partial tiles, in fixed order: a block owns 64 consecutive elements, wave w adds the
// workgroups [w P/4, (w+1) P/4) one after the other (eight loads in flight), the four sums are added as (0 + 1) + (2 + 3).
static __global__ __launch_bounds__(256) void wgrad_k7_reduce_kernel(const float* __restrict__ partial, int P, int total, int NT,
                                                                     int MW, int CN, int final_form, float* __restrict__ dw, int TAPS = 49) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int k0 = (int)((long long)P * wave / 4), k1 = (int)((long long)P * (wave + 1) / 4);
    float s = 0.f;
    if (j < total) {
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(k + u) * total + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < k1; ++k) s += partial[(long long)k * total + j];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || j >= total) return;
    s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const int l = j & 63, r = (j >> 6) & 15, tl = j >> 10;
    const int mt = tl / NT, nt = tl - mt * NT;
    const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    const int col = nt * 32 + (l & 31);
    if (m >= MW || col >= CN * TAPS) return;
    if (final_form) dw[m * 49 + (48 - col)] = s;                     // dW[0][m][6 - ky][6 - kx]
    else dw[(long long)m * CN * TAPS + col] = s;                     // dW[m][c][ky][kx]
}

}  // namespace apamd
