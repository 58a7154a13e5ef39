// wgrad_bf16x3.h -- the weight gradient of wgrad_igemm.h on the bf16 matrix pipe, at fp32-class accuracy
// (operands split into bf16 head + tail, three MFMAs per product, fp32 accumulation: see conv_bf16x3.h).
//
//   dW[m][ci][ky][kx] = sum_{n, oy, ox} G[n][m][oy][ox] * A[n][ci][oy + ky - pad][ox + kx - pad]        (stride 1)
//
// GEMM view: M = m (32 per wave), N = ci (32 per wave) with one accumulator tile PER TAP, K = pixels: an MFMA's
// 16-deep K step is 16 x-adjacent output pixels, 8 per half-wave.  A lane's operand fragment is therefore 8
// consecutive pixels of one channel -- for the shifted operand, at a column offset of kx pixels.  Both tensors are
// prepared once per layer (split_transpose_kernel) as pixel-octet slots with the CHANNEL innermost,
//     T[n][head|tail][row][x / 8][channel][8 x bf16],
// so that (a) every LDS-DMA piece is 64 consecutive channels of one octet = 1 KiB contiguous in HBM at a scalar
// address (no per-lane address work at all), (b) fragment reads are conflict-free ds_read_b128 (lane = channel), and
// (c) a tap's fragment is a funnel shift of two neighbouring octets by kx halfwords (v_alignbit; free for even kx).
// The kx / ky shifts are register / row-index arithmetic on ONE staged tile: no im2col, no per-tap copies.
//
// Workgroup = 4 waves = 2 (m) x 2 (ci) -> 64 x 64 channels x K*K taps; one pipeline stage = 2 output rows x 32
// columns of one image (4 K steps); stages of the workgroup's pixel share are double-buffered in LDS with the same
// flat schedule as conv_bf16x3 (stage g+2 issued from inside the last MFMA group of stage g).  The pixel range is
// split over P workgroups per (m, ci) tile; partials are summed by wgrad_reduce_kernel in a fixed order.
#pragma once
#include <type_traits>

#include "conv_bf16x3.h"
#include "wgrad_igemm.h"

namespace apamd {

struct WgradBf3Params {
    const uint4* gt;          // [N][2][GHp][GX8][Mp]  slots of 8 pixels
    const uint4* at;          // [N][2][Hp][AX8][Cp]
    int N, M, Cin, Q;         // Q = Cin * K * K
    int GHp, GX8, Mp, Hp, AX8, Cp;
    int tiles_x, tiles_y;     // pixel tiles (2 rows x 32 columns) per image
    int nstages, P;
    int m_tiles, c_tiles;
    float* partial;           // [P][M][Q]
    int ablate;               // debugging only (APAMD_ABLATE): 1 = no refill of the stage buffers, 2 = no barrier
};

// PARTS_ = 2: head + tail staged, three products per tap.  PARTS_ = 1 (AP_PRECISION_BF16): head planes only -- half the
// LDS stage and LDS-DMA traffic, one product per tap, two workgroups per CU.
// KY_ = 1: the ROW form of a K x K layer on few channels (the 7x7 stems): the shifted operand is prepared with one channel per
// (input channel, kernel row) -- channel ci * K + ky of row y is the padded input row y + ky (split_transpose_kernel, rows_k) --
// so the kernel sees a 1 x K layer over K * Cin channels: K taps, no row shift, and dW[m][ci * K + ky][kx] IS the OIHW tensor.
// WM_ = 4: EIGHT waves, 4 (m) x 2 (ci) -> a 128 x 64 channel tile.  With head + tail staged a 4-wave workgroup fills a CU's LDS, so
// every SIMD holds ONE wave and nothing covers its LDS / MFMA latencies; the 8-wave workgroup stages 1.3x the bytes (the shifted
// operand is shared by twice the products) for 2x the products and puts two waves on every SIMD.
template <int K_, int PARTS_ = 2, int KY_ = K_, int WM_ = 2>
struct WgradBf3Cfg {
    static constexpr int K = K_, KY = KY_, T = K * KY, PR = 2, PARTS = PARTS_, WM = WM_;
    static_assert(WM == 2 || WM == 4, "2 x 2 or 4 x 2 waves");
    static constexpr int NWAVES = 2 * WM, M_TILE = 32 * WM, MH = WM / 2;   // MH: 64-channel DMA pieces per G octet
    static_assert(PARTS == 1 || PARTS == 2, "head only, or head + tail");
    static_assert(KY == K || KY == 1, "square taps, or the row form");
    static constexpr int ROWS = PR + KY - 1;                // staged rows of the shifted operand
    static constexpr int NXG = 5;                           // staged octets per row: 4 + 1 for the column shift
    static constexpr int G_SLOTS = PARTS * PR * 4 * M_TILE; // [part][row][octet][m]
    static constexpr int A_SLOTS = PARTS * ROWS * NXG * 64; // [part][row][octet][ci]
    static constexpr int NPIECE = (G_SLOTS + A_SLOTS) / 64; // 1 KiB DMA pieces per stage
    static constexpr size_t lds_bytes() { return (size_t)2 * (G_SLOTS + A_SLOTS) * 16; }
    // one accumulator tile per tap: 16 K^2 registers -- with head-only staging two workgroups share a CU when that
    // leaves room for the operand registers (K <= 3; a 4x4 layer's 256 accumulators need a SIMD's whole file)
    static constexpr int WG_PER_CU = (PARTS == 1 && T * 16 <= 160 && WM == 2) ? 2 : 1;
    static_assert(K - 1 < 8, "the column shift must stay inside one extra octet");
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 8 consecutive bf16 starting SH elements into the 16-element sequence lo ++ hi
template <int SH>
__device__ __forceinline__ bf16x8 funnel8(const u32x4 lo, const u32x4 hi) {
    unsigned c[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    u32x4 r;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if constexpr ((SH & 1) == 0) r[d] = c[d + SH / 2];
        else r[d] = __builtin_amdgcn_alignbit(c[d + SH / 2 + 1], c[d + SH / 2], 16);
    }
    return __builtin_bit_cast(bf16x8, r);
}

// PARTS = 2: split-bf16 arithmetic (tail x head, head x tail, head x head per tap).  PARTS = 1: plain bf16 arithmetic
// (AP_PRECISION_BF16): the head x head product only, head planes only.
template <class C>
__global__ __launch_bounds__(C::NWAVES * 64, C::WG_PER_CU) void wgrad_bf16x3(const WgradBf3Params p) {
    constexpr int WM = C::WM, NWAVES = C::NWAVES, M_TILE = C::M_TILE, MH = C::MH;
    constexpr int PARTS = C::PARTS, PROD = PARTS == 1 ? 1 : 3;
    constexpr int K = C::K, KY = C::KY, T = C::T, ROWS = C::ROWS, NXG = C::NXG;
    constexpr int G_SLOTS = C::G_SLOTS, A_SLOTS = C::A_SLOTS, STAGE = G_SLOTS + A_SLOTS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint4* const smem = reinterpret_cast<const uint4*>(smem_raw);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wm = wave % WM, wq = wave / WM;
    int b = blockIdx.x;
    const int split = b % p.P; b /= p.P;
    const int ct = b % p.c_tiles;
    const int mt = b / p.c_tiles;
    const int st0 = (int)((long long)p.nstages * split / p.P), st1 = (int)((long long)p.nstages * (split + 1) / p.P);

    // ---- LDS-DMA: piece j of a stage = 64 consecutive channels of one (part, row, octet); this wave issues the
    // pieces j = wave, wave + 4, ...  Source = scalar slot index, + lane.
    // Wave w moves part (w & 1); of that part, G row (w >> 1) (4 octets) and every second (row, octet) pair of the
    // shifted operand -- so a piece's kind is static and only scalar offsets depend on the wave.  The position of
    // the NEXT stage to issue advances incrementally (no divisions in the loop).
    const unsigned lane16 = lane * 16;
    // (head-only staging: every wave moves head pieces -- G row (w >> 1), octets 2 (w & 1) and + 1; every fourth
    // (row, octet) pair of the shifted operand)
    const int dpart = PARTS == 2 ? (wave & 1) : 0, dhw = wave >> 1;
    constexpr int NGP = PARTS == 2 ? 4 : 2;                            // G pieces per wave and stage
    constexpr int NAE = ROWS * NXG, NAP = PARTS == 2 ? (NAE + 1) / 2 : (NAE + 3) / 4;   // (row, octet) pairs; per wave
    constexpr int NPW = WM == 2 ? NGP + NAP                            // pieces per wave and stage
                                : (PARTS * C::PR * 4 * MH + PARTS * ROWS * NXG + NWAVES - 1) / NWAVES;
    const long long g_row = (long long)p.GX8 * p.Mp, g_part = g_row * p.GHp;
    const long long a_row = (long long)p.AX8 * p.Cp, a_part = a_row * p.Hp;
    int in_ = 0, ity = 0, itx = 0;                                      // (image, tile row, tile column) to issue next
    {
        itx = st0 % p.tiles_x;
        const int t2 = st0 / p.tiles_x;
        ity = t2 % p.tiles_y;
        in_ = t2 / p.tiles_y;
    }
    long long gbase = 0, abase = 0;                                     // slot of (image, this wave's part, tile origin)
    auto locate = [&]() __attribute__((always_inline)) {
        if constexpr (WM == 2) {
            gbase = ((long long)(in_ * 2 + dpart) * p.GHp + ity * C::PR + dhw) * g_row + (long long)(itx * 4) * p.Mp + mt * 64;
            abase = ((long long)(in_ * 2 + dpart) * p.Hp + ity * C::PR) * a_row + (long long)(itx * 4) * p.Cp + ct * 64;
        } else {                                                      // (part 0, row 0 of the tile: the rest per piece)
            gbase = ((long long)(in_ * 2) * p.GHp + ity * C::PR) * g_row + (long long)(itx * 4) * p.Mp + mt * M_TILE;
            abase = ((long long)(in_ * 2) * p.Hp + ity * C::PR) * a_row + (long long)(itx * 4) * p.Cp + ct * 64;
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (++itx == p.tiles_x) {
            itx = 0;
            if (++ity == p.tiles_y) { ity = 0; ++in_; }
        }
    };
    // 8 waves: piece j of the wave = piece (wave + 8 j) of the stage's list [G: part][row][octet][64-channel half], then
    // [A: part][row][octet]; the decode is scalar arithmetic on the wave index
    constexpr int NGPIECE = PARTS * C::PR * 4 * MH, NAPIECE = PARTS * ROWS * NXG;
    auto issue_piece8 = [&](int buf, int i) __attribute__((always_inline)) {
        const int j = wave + NWAVES * i;
        if (j < NGPIECE) {
            const int mh = j % MH, t1 = j / MH, o = t1 % 4, t2 = t1 / 4, row = t2 % C::PR, part = t2 / C::PR;
            glds16_sv(p.gt + gbase + part * g_part + row * g_row + (long long)o * p.Mp + mh * 64, lane16,
                      lds0 + (buf * STAGE + (((part * C::PR + row) * 4 + o) * MH + mh) * 64) * 16);
        } else if (j < NGPIECE + NAPIECE) {
            const int e = j - NGPIECE, o = e % NXG, t1 = e / NXG, r = t1 % ROWS, part = t1 / ROWS;
            glds16_sv(p.at + abase + part * a_part + r * a_row + (long long)o * p.Cp, lane16,
                      lds0 + (buf * STAGE + G_SLOTS + ((part * ROWS + r) * NXG + o) * 64) * 16);
        }
    };
    auto issue_piece = [&](int buf, int j) __attribute__((always_inline)) {   // uses gbase / abase of the located stage
        if constexpr (WM == 4) {
            issue_piece8(buf, j);
        } else if (j < NGP) {
            const int o = PARTS == 2 ? j : 2 * (wave & 1) + j;          // octet of the G row
            glds16_sv(p.gt + gbase + (long long)o * p.Mp, lane16,
                      lds0 + (buf * STAGE + ((dpart * C::PR + dhw) * 4 + o) * 64) * 16);
        } else {
            const int e = PARTS == 2 ? (j - NGP) * 2 + dhw : (j - NGP) * 4 + wave;   // (row, octet) pair of this wave
            if ((PARTS == 2 && (j - NGP) * 2 + 1 < NAE) || e < NAE) {
                const int r = e / NXG, o = e - r * NXG;
                glds16_sv(p.at + abase + r * a_row + (long long)o * p.Cp, lane16,
                          lds0 + (buf * STAGE + G_SLOTS + ((dpart * ROWS + r) * NXG + o) * 64) * 16);
            }
        }
    };
    auto issue_stage = [&](int buf) __attribute__((always_inline)) {    // issues the located stage and moves on
        locate();
#pragma unroll
        for (int j = 0; j < NPW; ++j) issue_piece(buf, j);
        advance();
    };

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // fragment slots (16 bytes each) relative to a stage buffer
    const int ga = (half) * M_TILE + wm * 32 + l32;                    // + (part*PR*4 + py*4 + xh*2) * M_TILE
    const int ab = G_SLOTS + half * 64 + wq * 32 + l32;                // + ((part*ROWS + row)*NXG + xh*2 + j) * 64

    if (st0 < st1) {
        issue_stage(0);
        if (st0 + 1 < st1) issue_stage(1);
        dma_wait_all();
    }
    __syncthreads();

    // one group = (K step ks, kernel row ky): 4 raw octets of the shifted operand (+ 2 fragments of G at ky == 0),
    // then K taps x 3 MFMAs.  Groups run as a 3-deep software pipeline, flat across stages:
    //     while group g multiplies:  the raw octets of group g+2 are read from LDS,
    //                                the odd-shift operands of group g+1 are funnelled out of ITS raw octets,
    // so every LDS read and every vector-ALU result has a whole group (>= 9 MFMAs) before its first use, and the
    // products go round the K accumulators of the kernel row (consecutive MFMAs never wait on each other).
    constexpr int NG = 4 * KY;
    constexpr int RB = KY == 3 ? 3 : 4;                                // raw-octet register sets (RB divides NG)
    static_assert(NG % RB == 0 && NG % 2 == 0, "register set indices must be stage-invariant");
    // G fragments by ks parity -- in the row form (one group per K step) by group index mod RB, as the raw octets: the fetch of
    // group g + 2 precedes the products of group g in program order
    constexpr int NGF = KY == 1 ? RB : 2;
    bf16x8 ah[NGF], al[NGF];
    u32x4 rh[RB][2], rl[RB][2];                                        // by group index mod RB: [octet j]
    bf16x8 sh[2][K], sl[2][K];                                         // funnelled operands by group parity (odd shifts)
    // live = false: the stage does not exist (the padding stage of an odd count, see the stage loop): its G fragments are
    // forced to zero, so its products are exact zeros whatever finite data the buffer holds
    auto fetch_group = [&](int buf, int gidx, bool live) __attribute__((always_inline)) {
        const int ks = gidx / KY, ky = gidx % KY, rb = gidx % RB;
        const int py = ks >> 1, xh = ks & 1;
        const uint4* S0 = smem + buf * STAGE;
        if (ky == 0) {
            u32x4 th = *reinterpret_cast<const u32x4*>(S0 + ga + ((0 * C::PR + py) * 4 + xh * 2) * M_TILE);
#pragma unroll
            for (int d = 0; d < 4; ++d) th[d] = live ? th[d] : 0u;
            ah[KY == 1 ? rb : (ks & 1)] = __builtin_bit_cast(bf16x8, th);
            if constexpr (PARTS == 2) {
                u32x4 tl = *reinterpret_cast<const u32x4*>(S0 + ga + ((1 * C::PR + py) * 4 + xh * 2) * M_TILE);
#pragma unroll
                for (int d = 0; d < 4; ++d) tl[d] = live ? tl[d] : 0u;
                al[KY == 1 ? rb : (ks & 1)] = __builtin_bit_cast(bf16x8, tl);
            }
        }
        const int row = py + ky;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rh[rb][j] = *reinterpret_cast<const u32x4*>(S0 + ab + ((0 * ROWS + row) * NXG + xh * 2 + j) * 64);
            if constexpr (PARTS == 2) rl[rb][j] = *reinterpret_cast<const u32x4*>(S0 + ab + ((1 * ROWS + row) * NXG + xh * 2 + j) * 64);
        }
    };
    auto shift_group = [&](int gidx) __attribute__((always_inline)) {  // odd shifts only: even ones are register picks
        const int rb = gidx % RB, pb = gidx & 1;
        sh[pb][1] = funnel8<1>(rh[rb][0], rh[rb][1]);
        if constexpr (PARTS == 2) sl[pb][1] = funnel8<1>(rl[rb][0], rl[rb][1]);
        if constexpr (K > 3) {
            sh[pb][3] = funnel8<3>(rh[rb][0], rh[rb][1]);
            if constexpr (PARTS == 2) sl[pb][3] = funnel8<3>(rl[rb][0], rl[rb][1]);
        }
        if constexpr (K > 5) {
            sh[pb][5] = funnel8<5>(rh[rb][0], rh[rb][1]);
            if constexpr (PARTS == 2) sl[pb][5] = funnel8<5>(rl[rb][0], rl[rb][1]);
        }
    };
    auto mfma_one = [&](int gidx, int i) __attribute__((always_inline)) {
        const int ks = gidx / KY, ky = gidx % KY, rb = gidx % RB, pb = gidx & 1;
        const int pr = PROD == 1 ? 2 : i / K, kx = i % K;               // product-major: round the K accumulators
        bf16x8 bh, bl;
        if (kx == 0) { bh = funnel8<0>(rh[rb][0], rh[rb][1]); if constexpr (PARTS == 2) bl = funnel8<0>(rl[rb][0], rl[rb][1]); }
        else if (kx == 2) { bh = funnel8<2>(rh[rb][0], rh[rb][1]); if constexpr (PARTS == 2) bl = funnel8<2>(rl[rb][0], rl[rb][1]); }
        else if (kx == 4) { bh = funnel8<4>(rh[rb][0], rh[rb][1]); if constexpr (PARTS == 2) bl = funnel8<4>(rl[rb][0], rl[rb][1]); }
        else if (kx == 6) { bh = funnel8<6>(rh[rb][0], rh[rb][1]); if constexpr (PARTS == 2) bl = funnel8<6>(rl[rb][0], rl[rb][1]); }
        else { bh = sh[pb][kx]; if constexpr (PARTS == 2) bl = sl[pb][kx]; }
        f32x16& a = acc[ky * K + kx];
        const int gs = KY == 1 ? rb : (ks & 1);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pr == 0 ? al[gs] : ah[gs], pr == 1 ? bl : bh, a, 0, 0, 0);
    };

    auto stage = [&](auto ptag, int st) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;                       // stage buffer
        constexpr int NM = PROD * K, PPS = (NPW + 2 * NM - 1) / (2 * NM);  // DMA pieces per MFMA slot (last two groups)
        const bool live = st < st1, more = st + 1 < st1, dma = st + 2 < st1 && !AP_ABLATE(p, 1);
        if (st == st0) {
            fetch_group(P, 0, true);
            fetch_group(P, 1, true);
            shift_group(0);
        }
        if (!live) {                                                   // (the fragments of its first groups were not fetched)
#pragma unroll
            for (int i = 0; i < NGF; ++i) {
                ah[i] = bf16x8{};
                if constexpr (PARTS == 2) al[i] = bf16x8{};
            }
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi == NG - 2) {
                // the reads of this stage's last group were issued one group ago: once everybody is past the
                // barrier the buffer can be refilled (stage st+2), and stage st+1 has landed for the reads below
                if (dma) locate();
                if (!AP_ABLATE(p, 2)) {
                    dma_wait_all();
                    __syncthreads();
                }
            }
            // (no run-time conditions in the steady-state groups: a branch would split the basic block and the
            // compiler's wait-count bookkeeping turns conservative across blocks)
            if (gi + 2 < NG) fetch_group(P, gi + 2, live);
            else if (more) fetch_group(P ^ 1, gi + 2 - NG, true);
            if (gi + 1 < NG) shift_group(gi + 1);
            else if (more) shift_group(0);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                mfma_one(gi, i);
                if (gi >= NG - 2 && dma) {
#pragma unroll
                    for (int pp = 0; pp < PPS; ++pp) {
                        const int j = ((gi - (NG - 2)) * NM + i) * PPS + pp;
                        if (j < NPW) issue_piece(P, j);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (gi < NG - 2) {
                // pin: the LDS reads go one by one after the first MFMAs (left alone, the scheduler sinks every
                // read to just before its first use and waits on it)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dma) advance();
    };
    // Stages run in PAIRS, both halves unconditionally: with the second stage behind `if (st + 1 < st1)` its MFMAs sat in a
    // conditional block, every accumulator became a phi node and the allocator moved them between AGPRs and VGPRs around it
    // (K = 4: 998 v_accvgpr moves and 128-252 bytes of scratch for 128 MFMAs; round 5).  The padding stage of an odd count
    // multiplies zeroed G fragments with whatever the buffer holds -- which must be FINITE: a workgroup with a single stage never
    // fills buffer 1, so it is cleared here.
    if (st0 + 1 == st1) {
        uint4* z = reinterpret_cast<uint4*>(smem_raw) + STAGE;
        for (int i = tid; i < STAGE; i += NWAVES * 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    for (int st = st0; st < st1; st += 2) {
        stage(std::integral_constant<int, 0>{}, st);
        stage(std::integral_constant<int, 1>{}, st + 1);
    }

    // Partial sums leave in the accumulators' own order, partial[split][tile = mt * c_tiles + ct][wave][tap][r][lane]:
    // every store is 256 contiguous bytes per wave.  (The OIHW order -- element (m, ci, tap) at m * Q + ci * T + tap --
    // put a lane's 4 bytes 36..64 bytes from its neighbour's: 144 fully scattered store instructions per wave, 115 of
    // the kernel's 250 us.)  wgrad_bf3_reduce_kernel sums the splits in this order and scatters once.
    if (AP_ABLATE(p, 16)) {
        if (acc[0][0] == 123.456f) p.partial[0] = 1.f;
        return;
    }
    const long long tile_floats = (long long)NWAVES * T * 1024;
    float* out = p.partial + ((long long)split * p.m_tiles * p.c_tiles + (long long)mt * p.c_tiles + ct) * tile_floats +
                 (long long)wave * T * 1024 + lane;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64] = acc[t][r];
}

// dW = sum over the P splits of the partial tiles written by wgrad_bf16x3 (fixed order), scattered to the caller's
// layout: s2d_c == 0: dW[m][ci][ky][kx] (OIHW / IOHW rows as the plan defines M and Cin); s2d_c = C > 0: the kernel saw
// the space-to-depth form (4C channels, 2 x 2 taps) of a stride-2 K x K layer: ci' = (ry*2+rx)*C + c, tap = ty*2+tx ->
// dW[m][c][2 ty + ry][2 tx + rx], taps >= K (a 3x3 layer's seven all-zero ones) dropped.
__global__ __launch_bounds__(256) void wgrad_bf3_reduce_kernel(const float* __restrict__ partial, int P, int M, int Cin,
                                                               int T, int c_tiles, long long total, int s2d_c, int K,
                                                               float* __restrict__ dw, int WM = 2) {
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < total; j += (long long)gridDim.x * 256) {
        float s = 0.f;
        int k = 0;
        for (; k + 4 <= P; k += 4) {                 // four loads in flight (same summation order as one by one)
            const float a = partial[(long long)k * total + j], b = partial[(long long)(k + 1) * total + j];
            const float c = partial[(long long)(k + 2) * total + j], d = partial[(long long)(k + 3) * total + j];
            s += a; s += b; s += c; s += d;
        }
        for (; k < P; ++k) s += partial[(long long)k * total + j];
        const int lane = (int)(j & 63), r = (int)((j >> 6) & 15);
        const long long jt = j >> 10;
        const int t = (int)(jt % T);
        const int nw = 2 * WM;                       // waves per tile: WM (m) x 2 (ci)
        const int wave = (int)((jt / T) % nw);
        const int tile = (int)(jt / T / nw);
        const int mt = tile / c_tiles, ct = tile - mt * c_tiles;
        const int mm = mt * (32 * WM) + (wave % WM) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int ci = ct * 64 + (wave / WM) * 32 + (lane & 31);
        if (mm >= M || ci >= Cin) continue;
        if (s2d_c == 0) {
            dw[((long long)mm * Cin + ci) * T + t] = s;
        } else {
            const int r2 = ci / s2d_c, c = ci - r2 * s2d_c;
            const int ky = 2 * (t >> 1) + (r2 >> 1), kx = 2 * (t & 1) + (r2 & 1);
            if (ky < K && kx < K) dw[(((long long)mm * s2d_c + c) * K + ky) * K + kx] = s;
        }
    }
}

// ---- operand preparation: T[n][part][y][x/8][c][8 px] (bf16 head / tail) = padded view of act(IN(concat(src))),
// (y, x) <-> source (y - pad, x - pad); reflection or zero padding inside [0, H+2pad) x [0, W+2pad); zeros beyond,
// and for channels >= C.  The transposition goes through LDS: fp32 reads run along x, slot writes along c.
struct SplitTParams {
    SrcSeg seg[kMaxSeg];      // chunk_begin = first concat channel
    int nseg;
    int N, C, H, W, pad, pad_mode, Hp, X8, Cp;
    int s2d_c;                // > 0: space-to-depth view of a pad-1 source with s2d_c channels (see below)
    int heads_only;           // 1: the consumer multiplies head parts only (AP_PRECISION_BF16): the tail planes are not written
    int rows_k;               // > 0: row view of a K x K layer's source, K = rows_k (see below; split_transpose_kernel only)
    uint4* out;
};

// Space-to-depth view (s2d_c = C0 > 0): channel c' = (ry * 2 + rx) * C0 + c of the (H, W) = (H0/2 + 1, W0/2 + 1) map is
// pad1(act(IN(concat(src))))[c][2 y + ry][2 x + rx] -- the operand of a stride-2 K x K (K = 3, 4) layer rewritten as
// the 2 x 2 stride-1 layer over 4 C0 channels (as ap_split_prepass_s2d does for the forward pass), so that strided
// weight gradients run on the same bf16 GEMM kernel.  p.H / p.W are then the source's own size and pad must be 0.
// Row view (rows_k = K > 0): channel c' = c * K + ky of the H x (W + 2 pad) map is pad(act(IN(src)))[c][y + ky][x] -- the operand of
// a K x K stride-1 layer on few channels rewritten as the 1 x K layer over K C channels (as ap_split_prepass_rows does for the
// forward pass: WgradBf3Cfg KY = 1).  p.C is then the number of VIEW channels (K times the source's).
// grid: (ceil(Hp * X8 / 8), Cp / 64, N): a workgroup transposes 8 consecutive octets (64 padded pixels, possibly
// across a row boundary) of 64 channels.
__global__ __launch_bounds__(256) void split_transpose_kernel(const SplitTParams p) {
    __shared__ float f[64][65];
    const int cg = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
    const int He = p.H + 2 * p.pad, We = p.W + 2 * p.pad, HW = p.H * p.W;
    const int oct0 = blockIdx.x * 8, noct = p.Hp * p.X8;
    // phase 1: thread = (pixel of the span, channel quarter): 16 independent loads, 8-pixel runs per octet
    {
        const int px = tid & 63, oct = oct0 + (px >> 3);
        const int y = oct / p.X8, x = (oct - y * p.X8) * 8 + (px & 7);
        int sy = y - p.pad, sx = x - p.pad;
        bool ok = oct < noct && y < He && x < We;
        if (p.s2d_c > 0) {
            ok = oct < noct && y <= p.H / 2 && x <= p.W / 2;       // the view is (H/2 + 1) x (W/2 + 1)
        } else if (p.rows_k > 0) {
            ok = oct < noct && y < p.H && x < We;                  // the view is H x (W + 2 pad); rows resolved per channel
        } else if (p.pad_mode == 1) {
            sy = reflect_clamp(sy, p.H);
            sx = reflect_clamp(sx, p.W);
        } else {
            ok = ok && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
        }
        const int soff = ok ? sy * p.W + sx : 0;
        const int cq = __builtin_amdgcn_readfirstlane(tid >> 6) * 16;   // this wave's 16 channels of the group (scalar: see split_transpose_pad_kernel)
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int c = cg * 64 + cq + i;
            v[i] = 0.f;
            if (c < p.C) {
                bool okc = ok;
                int so = soff;
                if (p.s2d_c > 0) {                                 // (wave-uniform: 16 consecutive channels share r
                    const int r = c / p.s2d_c;                     //  unless they straddle a multiple of C0)
                    c -= r * p.s2d_c;
                    const int yy = 2 * y + (r >> 1) - 1, xx = 2 * x + (r & 1) - 1;
                    okc = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                    so = okc ? yy * p.W + xx : 0;
                } else if (p.rows_k > 0) {
                    const int cc = c / p.rows_k;
                    int yy = y + (c - cc * p.rows_k) - p.pad, xx = sx;
                    c = cc;
                    if (p.pad_mode == 1) {
                        yy = reflect_clamp(yy, p.H);
                        xx = reflect_clamp(xx, p.W);
                    } else {
                        okc = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                    }
                    so = okc ? yy * p.W + xx : 0;
                }
                int s = 0;
                if (p.nseg > 1 && c >= p.seg[1].chunk_begin) s = 1;
                if (p.nseg > 2 && c >= p.seg[2].chunk_begin) s = 2;
                const SrcSeg sg = s == 0 ? p.seg[0] : (s == 1 ? p.seg[1] : p.seg[2]);   // wave-uniform
                const int cs = c - sg.chunk_begin;
                float t = sg.data[((long long)n * sg.C + cs) * HW + so];
                if (sg.mean != nullptr) t = (t - sg.mean[n * sg.C + cs]) * sg.rstd[n * sg.C + cs];
                t = sg.act == 1 ? fmaxf(t, 0.f) : (sg.act == 2 ? (t > 0.f ? t : 0.2f * t) : t);
                v[i] = okc ? t : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) f[cq + i][px] = v[i];
    }
    __syncthreads();
    // phase 2: thread = (channel, octet): 64 consecutive channels of one octet per wave-store (1 KiB contiguous)
    const int c = tid & 63;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int o = (tid >> 6) + 4 * k, oct = oct0 + o;
        if (oct < noct) {
            bf16x8 hv, lv;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 h, l;
                split_bf16(f[c][o * 8 + j], h, l);
                hv[j] = h;
                lv[j] = l;
            }
            *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 0) * noct + oct) * p.Cp + cg * 64 + c) = hv;
            if (!p.heads_only) *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 1) * noct + oct) * p.Cp + cg * 64 + c) = lv;
        }
    }
}

__device__ __forceinline__ float4 norm_act4(float4 v, const SrcSeg& sg, int nc) {
    if (sg.mean != nullptr) {
        const float m = sg.mean[nc], r = sg.rstd[nc];
        v.x = (v.x - m) * r; v.y = (v.y - m) * r; v.z = (v.z - m) * r; v.w = (v.w - m) * r;
    }
    if (sg.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (sg.act == 2) {
        v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y;
        v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w;
    }
    return v;
}

// Fast path of split_transpose_kernel for an unpadded operand whose rows are whole octet rows (pad == 0, X8 * 8 == W:
// the gradient operand of every 64 / 128 / 256-pixel-wide layer): a workgroup converts 32 consecutive octets = 256
// consecutive pixels of 64 channels.  Loads are 16 bytes per lane and 1 KiB contiguous per wave (the general kernel
// reads 256-byte runs from 16 planes per thread: 2.6 TB/s); stores are the same 1 KiB slot rows.
// grid: (ceil(Hp * X8 / 32), Cp / 64, N)
__global__ __launch_bounds__(256) void split_transpose_vec_kernel(const SplitTParams p) {
    extern __shared__ float fv[];                    // [64][257]
    constexpr int LP = 257;
    const int cg = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int HW = p.H * p.W, noct = p.Hp * p.X8;
    const int oct0 = blockIdx.x * 32;
    const SrcSeg sg = p.seg[0];
    {
        const int oct = oct0 + (lane >> 1);
        const int y = oct / p.X8, x = (oct - y * p.X8) * 8 + (lane & 1) * 4;
        const bool ok = oct < noct && y < p.H;       // (x < W by construction: X8 * 8 == W)
        const long long soff = (long long)y * p.W + x;
        float4 vv[16];                               // all loads first (see split_transpose_pad_kernel)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = cg * 64 + i * 4 + wave;
            vv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && c < p.C) vv[i] = *reinterpret_cast<const float4*>(sg.data + ((long long)n * sg.C + c) * HW + soff);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cl = i * 4 + wave, c = cg * 64 + cl;
            float4 v = vv[i];
            if (ok && c < p.C) v = norm_act4(v, sg, n * sg.C + c);
            float* d = fv + cl * LP + lane * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();
    const int c = lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int o = wave + 4 * k, oct = oct0 + o;
        if (oct < noct) {
            bf16x8 hv, lv;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 h, l;
                split_bf16(fv[c * LP + o * 8 + j], h, l);
                hv[j] = h;
                lv[j] = l;
            }
            *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 0) * noct + oct) * p.Cp + cg * 64 + c) = hv;
            if (!p.heads_only) *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 1) * noct + oct) * p.Cp + cg * 64 + c) = lv;
        }
    }
}

// Row forms of split_transpose_kernel: a workgroup converts whole padded rows of 64 channels, so that every source row
// is read with 16-byte loads (1 KiB contiguous per wave in the interior) instead of 256-byte runs of 4-byte loads.
//
// (1) pad == 1 (reflection or zero), W in {32, 64, 128, 256}: R = 256 / W padded rows per workgroup; one wave-load
//     covers the R rows of one channel; the two border columns and the zero slots up to the octet boundary are written
//     by the lanes that hold the neighbouring values.            grid: (ceil(Hp / R), Cp / 64, N)
__device__ __forceinline__ void split_store_octets(const SplitTParams& p, const float* fv, int LP, int n, int cg, int oct0,
                                                   int nocts_tile, int wave, int lane) {
    const int noct = p.Hp * p.X8;
    for (int o = wave; o < nocts_tile; o += 4) {
        const int oct = oct0 + o;
        if (oct >= noct) break;
        bf16x8 hv, lv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 h, l;
            split_bf16(fv[lane * LP + o * 8 + j], h, l);
            hv[j] = h;
            lv[j] = l;
        }
        *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 0) * noct + oct) * p.Cp + cg * 64 + lane) = hv;
        if (!p.heads_only) *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 1) * noct + oct) * p.Cp + cg * 64 + lane) = lv;
    }
}

__global__ __launch_bounds__(256) void split_transpose_pad_kernel(const SplitTParams p) {
    extern __shared__ float fv[];                    // [64][R * X8 * 8 + 1]
    // (the wave index as a SCALAR: the channel a wave works on, its segment, base pointer and statistics then live in
    // SGPRs; left in a VGPR the compiler fetches the selected SrcSeg with vector loads and every iteration becomes a
    // chain of three dependent memory round trips)
    const int cg = blockIdx.y, n = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int LPR = p.W >> 2, R = 64 / LPR;          // lanes per row, rows per workgroup
    const int RS = p.X8 * 8, LP = R * RS + 1;        // slots per padded row, LDS row pitch
    const int He = p.H + 2, HW = p.H * p.W;
    const int y0 = blockIdx.x * R;
    {
        const int r = lane / LPR, xl = (lane - r * LPR) * 4;
        const int y = y0 + r;
        int sy = y - 1;
        bool ok = y < He;
        if (p.pad_mode == 1) sy = reflect_clamp(sy, p.H);
        else ok = ok && sy >= 0 && sy < p.H;
        const long long soff = ok ? (long long)sy * p.W + xl : 0;
        const bool refl = p.pad_mode == 1 && ok;
        // all 16 loads of the wave first (the compiler does not hoist them over the LDS stores on its own: one memory
        // round trip per channel otherwise), then the arithmetic and the LDS writes
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = cg * 64 + i * 4 + wave;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < p.C) {
                int sgi = 0;
                if (p.nseg > 1 && c >= p.seg[1].chunk_begin) sgi = 1;
                if (p.nseg > 2 && c >= p.seg[2].chunk_begin) sgi = 2;
                const float* base = (sgi == 0 ? p.seg[0].data : (sgi == 1 ? p.seg[1].data : p.seg[2].data));
                const int sc = sgi == 0 ? p.seg[0].C : (sgi == 1 ? p.seg[1].C : p.seg[2].C);
                const int cs = c - (sgi == 0 ? 0 : (sgi == 1 ? p.seg[1].chunk_begin : p.seg[2].chunk_begin));
                const int b16 = sgi == 0 ? p.seg[0].pad_ : (sgi == 1 ? p.seg[1].pad_ : p.seg[2].pad_);   // the segment holds bf16 values
                if (ok) {
                    if (b16) {
                        const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + ((long long)n * sc + cs) * HW + soff);
                        v[i] = make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                                           __uint_as_float(t.y & 0xffff0000u));
                    } else {
                        v[i] = *reinterpret_cast<const float4*>(base + ((long long)n * sc + cs) * HW + soff);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cl = i * 4 + wave, c = cg * 64 + cl;
            float4 w = v[i];
            if (c < p.C && ok) {
                int sgi = 0;
                if (p.nseg > 1 && c >= p.seg[1].chunk_begin) sgi = 1;
                if (p.nseg > 2 && c >= p.seg[2].chunk_begin) sgi = 2;
                const SrcSeg sg = sgi == 0 ? p.seg[0] : (sgi == 1 ? p.seg[1] : p.seg[2]);   // wave-uniform, scalar
                w = norm_act4(w, sg, n * sg.C + c - sg.chunk_begin);
            }
            float* d = fv + cl * LP + r * RS;
            d[1 + xl] = w.x; d[2 + xl] = w.y; d[3 + xl] = w.z; d[4 + xl] = w.w;
            if (xl == 0) d[0] = refl ? w.y : 0.f;                        // padded column 0 <- source column 1
            if (xl == p.W - 4) d[p.W + 1] = refl ? w.z : 0.f;            // padded column W + 1 <- source column W - 2
        }
        // the slots between the padded row's end and the octet boundary
        const int rsh = __ffs(R) - 1;
        for (int e = tid; e < 64 * R; e += 256) {
            float* d = fv + (e >> rsh) * LP + (e & (R - 1)) * RS;
            for (int z = p.W + 2; z < RS; ++z) d[z] = 0.f;
        }
    }
    __syncthreads();
    split_store_octets(p, fv, LP, n, cg, y0 * p.X8, R * p.X8, wave, lane);
}

// (2) the space-to-depth view (s2d_c = C0, C0 % 32 == 0, source W0 in {64, 128, 256}): a workgroup reads R = 256 / W0
//     source rows of equal parity of 32 source channels -- whole rows, every element used -- and writes both column
//     parities rx = 0, 1: view channels (ry * 2 + rx) * C0 + c of view rows y = (sy + 1 - ry) / 2.  The 64 LDS rows are
//     [rx][32 channels]; a wave-store is two 512-byte runs.
//     grid: (ceil(Hv / R) * 2 [ry], C0 / 32, N), Hv = H0 / 2 + 1 view rows (the operand has Hp >= Hv rows: the rest is
//     zero-filled by the workgroups of the last row group)
__global__ __launch_bounds__(256) void split_transpose_s2d_kernel(const SplitTParams p) {
    extern __shared__ float fv[];                    // [2 * 32][R * X8 * 8 + 1]
    const int n = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C0 = p.s2d_c;
    const int LPR = p.W >> 2, R = 64 / LPR;
    const int RS = p.X8 * 8, LP = R * RS + 1;
    const int Hv = p.H / 2 + 1, Wv = p.W / 2 + 1, HW = p.H * p.W;
    const int ry = blockIdx.x & 1, y0 = (blockIdx.x >> 1) * R;
    const int cb = blockIdx.y * 32;
    const SrcSeg sg = p.seg[0];
    {
        const int r = lane / LPR, l = lane - r * LPR, xl = l * 4;
        const int y = y0 + r, sy = 2 * y + ry - 1;
        const bool ok = y < Hv && sy >= 0 && sy < p.H;
        const long long soff = ok ? (long long)sy * p.W + xl : 0;
        float4 vv[8];                                // all loads first (see split_transpose_pad_kernel)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            vv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) vv[i] = *reinterpret_cast<const float4*>(sg.data + ((long long)n * sg.C + cb + i * 4 + wave) * HW + soff);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cl = i * 4 + wave, c = cb + cl;
            float4 v = vv[i];
            if (ok) v = norm_act4(v, sg, n * sg.C + c);
            // rx = 0: view column x <- source column 2x - 1 (odd columns; x = 0 is the zero border)
            float* d0 = fv + cl * LP + r * RS;
            d0[2 * l + 1] = v.y; d0[2 * l + 2] = v.w;
            // rx = 1: view column x <- source column 2x (even columns; x = W0 / 2 is the zero border)
            float* d1 = fv + (32 + cl) * LP + r * RS;
            d1[2 * l] = v.x; d1[2 * l + 1] = v.z;
            if (l == 0) d0[0] = 0.f;
            if (l == LPR - 1) {
                d1[Wv - 1] = 0.f;
                for (int z = Wv; z < RS; ++z) { d0[z] = 0.f; d1[z] = 0.f; }
            }
        }
    }
    __syncthreads();
    // thread = (rx, channel) of the 64 LDS rows; view channel (ry * 2 + rx) * C0 + cb + c
    const int noct = p.Hp * p.X8;
    const int rx = lane >> 5, c = lane & 31;
    const long long cofs = (long long)(ry * 2 + rx) * C0 + cb + c;
    for (int o = wave; o < R * p.X8; o += 4) {
        const int oct = y0 * p.X8 + o;
        if (oct >= noct) break;
        bf16x8 hv, lv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 h, l2;
            split_bf16(fv[lane * LP + o * 8 + j], h, l2);
            hv[j] = h;
            lv[j] = l2;
        }
        *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 0) * noct + oct) * p.Cp + cofs) = hv;
        if (!p.heads_only) *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 1) * noct + oct) * p.Cp + cofs) = lv;
    }
}

// ---- the same operand from the FORWARD pass's split copy.  The convolution that produced the layer's output staged
// XS[n][part][C/8][HW + 1][8 channels] (conv_prepass.h: act(IN(.)) applied, split, slot HW all-zero) of every source; it is
// still alive when the backward pass runs, and it holds exactly the values the operand needs -- so the operand is a
// re-tiling of 16-byte slots (8 channels of a pixel -> 8 pixels of a channel), not a second normalisation pass over the
// fp32 tensor: a thread loads the 8 slots of one pixel octet (padding = index arithmetic; out-of-image taps of a
// zero-padded layer read slot HW), transposes the 8 x 8 halfwords in registers (32 v_perm_b32) and stores 128 contiguous
// bytes.  No LDS, no statistics, bf16 in and out: 33 MB read + 39 MB written for a 256-channel 64 x 64 layer at B = 16
// in plain-bf16 mode, where split_transpose_pad_kernel reads 67 (bf16 source) .. 134 MB.
// The space-to-depth operand of a stride-2 layer comes the same way from ITS forward copy (ap_split_prepass_s2d: the
// view (4 C0, H0/2 + 1, W0/2 + 1), zero ring included): pad = 0 on the view.
// Lanes: 8 consecutive lanes own one pixel octet, lane e its pixel e -- a load instruction fetches 64 consecutive pixels'
// slots (1 KiB contiguous) and, after the transposition, lane c holds the octet's slot of channel c, so a store
// instruction writes whole 128-byte lines (8 channels x 16 bytes per octet).  The 8 x 8 halfword transposition runs
// ACROSS the 8 lanes as three butterfly exchanges (lane ^ 4 on dword pairs: ds_swizzle; lane ^ 2 on dwords and lane ^ 1
// on halfwords: DPP quad permutes + v_perm_b32).  (First form, one thread per (octet, channel octet) with a register
// transposition: every load and store instruction touched 64 different lines, 16 bytes of each -- 38 us per launch in
// the train step where this form takes ~20.)
// grid: (ceil(Hp * X8 / 32), ceil(Cp / 8 / 4), N): a workgroup = 32 pixel octets x 4 channel octets.
struct XsTParams {
    const uint4* xs[kMaxSeg];     // split copy of each source segment
    int cg_begin[kMaxSeg + 1];    // first channel octet of each segment; [nseg] = total octets (C / 8)
    int nseg;
    int N, H, W, pad, pad_mode, Hp, X8, Cp;
    int parts;                    // 1: head planes only
    int s2d_c, H0, W0;            // s2d_c = C0 > 0: the space-to-depth view (4 C0, H, W) = (.., H0/2 + 1, W0/2 + 1) gathered from the
                                  // PLAIN copy of the (C0, H0, W0) source (a stride-2 layer whose forward pass staged that one):
                                  // view channel r * C0 + c, pixel (y, x) = pad1(source)[c][2 y + (r >> 1)][2 x + (r & 1)]
    uint4* out;
};

constexpr int kXsCgPerThread = 4;

template <int XOR>
__device__ __forceinline__ unsigned lane_xor(unsigned v) {
    if constexpr (XOR == 4) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);               // bit mode: and 0x1f, xor 4
    else if constexpr (XOR == 2) return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);                       // quad_perm [1,0,3,2]
}

__global__ __launch_bounds__(256) void xs_transpose_kernel(const XsTParams p) {
    const int tid = threadIdx.x, e = tid & 7;
    const int pos = blockIdx.x * 32 + (tid >> 3);
    if (pos >= p.Hp * p.X8) return;                                 // (whole 8-lane groups)
    const int n = blockIdx.z;
    const int y = pos / p.X8, k = pos - y * p.X8;
    const int HW = p.H * p.W, He = p.H + 2 * p.pad, We = p.W + 2 * p.pad;
    int idx;
    {
        const int x = k * 8 + e;
        int sy = y - p.pad, sx = x - p.pad;
        bool ok = y < He && x < We;
        if (p.pad_mode == 1) {
            sy = reflect_clamp(sy, p.H);
            sx = reflect_clamp(sx, p.W);
        } else {
            ok = ok && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
        }
        idx = ok ? sy * p.W + sx : HW;                              // slot HW: zeros
    }
    const int total_cg = p.cg_begin[p.nseg];
    for (int part = 0; part < p.parts; ++part) {
        uint4 v[kXsCgPerThread];
#pragma unroll
        for (int i = 0; i < kXsCgPerThread; ++i) {
            const int cg = blockIdx.y * kXsCgPerThread + i;         // (block-uniform)
            v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (cg < total_cg) {
                if (p.s2d_c > 0) {
                    const int CG0 = p.s2d_c >> 3, r = cg / CG0, cgl = cg - r * CG0, HW0 = p.H0 * p.W0;
                    const int yy = 2 * y + (r >> 1) - 1, xx = 2 * (k * 8 + e) + (r & 1) - 1;
                    const bool ok = y < p.H && k * 8 + e < p.W && yy >= 0 && yy < p.H0 && xx >= 0 && xx < p.W0;
                    v[i] = p.xs[0][((long long)(n * 2 + part) * CG0 + cgl) * (HW0 + 1) + (ok ? yy * p.W0 + xx : HW0)];
                    continue;
                }
                int s = 0;
                if (p.nseg > 1 && cg >= p.cg_begin[1]) s = 1;
                if (p.nseg > 2 && cg >= p.cg_begin[2]) s = 2;
                const int CGs = p.cg_begin[s + 1] - p.cg_begin[s], cgl = cg - p.cg_begin[s];
                const uint4* src = s == 0 ? p.xs[0] : (s == 1 ? p.xs[1] : p.xs[2]);
                v[i] = src[((long long)(n * 2 + part) * CGs + cgl) * (HW + 1) + idx];
            }
        }
#pragma unroll
        for (int i = 0; i < kXsCgPerThread; ++i) {
            const int cg = blockIdx.y * kXsCgPerThread + i;
            if (cg * 8 >= p.Cp) continue;
            unsigned d0 = v[i].x, d1 = v[i].y, d2 = v[i].z, d3 = v[i].w;
            {   // lane ^ 4: dword pairs
                const bool hi = e & 4;
                const unsigned s0 = hi ? d0 : d2, s1 = hi ? d1 : d3;
                const unsigned r0 = lane_xor<4>(s0), r1 = lane_xor<4>(s1);
                if (hi) { d0 = r0; d1 = r1; } else { d2 = r0; d3 = r1; }
            }
            {   // lane ^ 2: dwords inside each pair
                const bool hi = e & 2;
                const unsigned s0 = hi ? d0 : d1, s1 = hi ? d2 : d3;
                const unsigned r0 = lane_xor<2>(s0), r1 = lane_xor<2>(s1);
                if (hi) { d0 = r0; d2 = r1; } else { d1 = r0; d3 = r1; }
            }
            {   // lane ^ 1: halfwords inside each dword.  even lane: (own.lo, other.lo); odd lane: (other.hi, own.hi)
                const bool hi = e & 1;
                const unsigned r0 = lane_xor<1>(d0), r1 = lane_xor<1>(d1), r2 = lane_xor<1>(d2), r3 = lane_xor<1>(d3);
                d0 = hi ? __builtin_amdgcn_perm(d0, r0, 0x07060302u) : __builtin_amdgcn_perm(r0, d0, 0x05040100u);
                d1 = hi ? __builtin_amdgcn_perm(d1, r1, 0x07060302u) : __builtin_amdgcn_perm(r1, d1, 0x05040100u);
                d2 = hi ? __builtin_amdgcn_perm(d2, r2, 0x07060302u) : __builtin_amdgcn_perm(r2, d2, 0x05040100u);
                d3 = hi ? __builtin_amdgcn_perm(d3, r3, 0x07060302u) : __builtin_amdgcn_perm(r3, d3, 0x05040100u);
            }
            // lane e now holds channel e of the octet: pixels 0..7
            p.out[(((long long)(n * 2 + part) * p.Hp + y) * p.X8 + k) * p.Cp + cg * 8 + e] = make_uint4(d0, d1, d2, d3);
        }
    }
}

}  // namespace apamd
