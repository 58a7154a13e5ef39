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
};

template <int K_>
struct WgradBf3Cfg {
    static constexpr int K = K_, T = K * K, PR = 2;
    static constexpr int ROWS = PR + K - 1;                 // staged rows of the shifted operand
    static constexpr int NXG = 5;                           // staged octets per row: 4 + 1 for the column shift
    static constexpr int G_SLOTS = 2 * PR * 4 * 64;         // [part][row][octet][m]
    static constexpr int A_SLOTS = 2 * ROWS * NXG * 64;     // [part][row][octet][ci]
    static constexpr int NPIECE = (G_SLOTS + A_SLOTS) / 64; // 1 KiB DMA pieces per stage
    static constexpr size_t lds_bytes() { return (size_t)2 * (G_SLOTS + A_SLOTS) * 16; }
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

template <class C>
__global__ __launch_bounds__(256, 1) void wgrad_bf16x3(const WgradBf3Params p) {
    constexpr int K = C::K, T = C::T, ROWS = C::ROWS, NXG = C::NXG;
    constexpr int G_SLOTS = C::G_SLOTS, A_SLOTS = C::A_SLOTS, STAGE = G_SLOTS + A_SLOTS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint4* const smem = reinterpret_cast<const uint4*>(smem_raw);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wm = wave & 1, wq = wave >> 1;
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
    const int dpart = wave & 1, dhw = wave >> 1;
    constexpr int NAE = ROWS * NXG, NAP = (NAE + 1) / 2;               // (row, octet) pairs; per wave
    constexpr int NPW = 4 + NAP;                                       // pieces per wave and stage
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
        gbase = ((long long)(in_ * 2 + dpart) * p.GHp + ity * C::PR + dhw) * g_row + (long long)(itx * 4) * p.Mp + mt * 64;
        abase = ((long long)(in_ * 2 + dpart) * p.Hp + ity * C::PR) * a_row + (long long)(itx * 4) * p.Cp + ct * 64;
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (++itx == p.tiles_x) {
            itx = 0;
            if (++ity == p.tiles_y) { ity = 0; ++in_; }
        }
    };
    auto issue_piece = [&](int buf, int j) __attribute__((always_inline)) {   // uses gbase / abase of the located stage
        if (j < 4) {
            glds16_sv(p.gt + gbase + (long long)j * p.Mp, lane16,
                      lds0 + (buf * STAGE + ((dpart * C::PR + dhw) * 4 + j) * 64) * 16);
        } else {
            const int e = (j - 4) * 2 + dhw;                            // (row, octet) pair of this wave
            if ((j - 4) * 2 + 1 < NAE || e < NAE) {
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
    const int ga = (half) * 64 + wm * 32 + l32;                        // + (part*PR*4 + py*4 + xh*2) * 64
    const int ab = G_SLOTS + half * 64 + wq * 32 + l32;                // + ((part*ROWS + row)*NXG + xh*2 + j) * 64

    if (st0 < st1) {
        issue_stage(0);
        if (st0 + 1 < st1) issue_stage(1);
        dma_wait_all();
    }
    __syncthreads();

    // one group = (K step ks, kernel row ky): 4 raw octets of the shifted operand (+ 2 fragments of G at ky == 0),
    // then K taps x 3 MFMAs
    constexpr int NG = 4 * K;
    bf16x8 ah[2], al[2];                                               // by ks parity
    u32x4 rh[2][2], rl[2][2];                                          // by group parity: [octet j]
    auto fetch_group = [&](int buf, int gidx, int pb) __attribute__((always_inline)) {
        const int ks = gidx / K, ky = gidx % K;
        const int py = ks >> 1, xh = ks & 1;
        const uint4* S0 = smem + buf * STAGE;
        if (ky == 0) {
            ah[ks & 1] = *reinterpret_cast<const bf16x8*>(S0 + ga + ((0 * C::PR + py) * 4 + xh * 2) * 64);
            al[ks & 1] = *reinterpret_cast<const bf16x8*>(S0 + ga + ((1 * C::PR + py) * 4 + xh * 2) * 64);
        }
        const int row = py + ky;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rh[pb][j] = *reinterpret_cast<const u32x4*>(S0 + ab + ((0 * ROWS + row) * NXG + xh * 2 + j) * 64);
            rl[pb][j] = *reinterpret_cast<const u32x4*>(S0 + ab + ((1 * ROWS + row) * NXG + xh * 2 + j) * 64);
        }
    };
    // shifted fragments of one group, then K taps x 3 products.  The products go round the K accumulators of the
    // kernel row in turn (small terms first), so consecutive MFMAs never wait on each other's result.
    auto shift_group = [&](int pb, bf16x8 (&bh)[K], bf16x8 (&bl)[K]) __attribute__((always_inline)) {
        auto one = [&](auto kxtag) __attribute__((always_inline)) {
            constexpr int KX = decltype(kxtag)::value;
            bh[KX] = funnel8<KX>(rh[pb][0], rh[pb][1]);
            bl[KX] = funnel8<KX>(rl[pb][0], rl[pb][1]);
        };
        one(std::integral_constant<int, 0>{});
        one(std::integral_constant<int, 1>{});
        one(std::integral_constant<int, 2>{});
        if constexpr (K > 3) one(std::integral_constant<int, 3>{});
    };
    auto mfma_slot = [&](int gidx, int i, const bf16x8 (&bh)[K], const bf16x8 (&bl)[K]) __attribute__((always_inline)) {
        const int ks = gidx / K, ky = gidx % K;
        const int pr = i / K, kx = i % K;
        f32x16& a = acc[ky * K + kx];
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pr == 0 ? al[ks & 1] : ah[ks & 1], pr == 1 ? bl[kx] : bh[kx], a, 0, 0, 0);
    };
    auto mfma_group = [&](int gidx, int pb) __attribute__((always_inline)) {
        bf16x8 bh[K], bl[K];
        shift_group(pb, bh, bl);
#pragma unroll
        for (int i = 0; i < 3 * K; ++i) mfma_slot(gidx, i, bh, bl);
    };

    auto stage = [&](auto ptag, int st) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;                       // stage buffer
        // group parity continues across stages: NG is even, so group 0 always uses register set 0
        if (st == st0) fetch_group(P, 0, 0);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int pb = gi & 1;
            if (gi + 1 < NG) {
                fetch_group(P, gi + 1, pb ^ 1);
                mfma_group(gi, pb);
                // pin the schedule: the next group's LDS reads go one by one between this group's MFMAs (left
                // alone, the scheduler sinks each read to just before its first use and waits on it)
#pragma unroll
                for (int i = 0; i < 3 * K; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // every fragment of this stage is in registers: after the barrier the buffer is refilled with
                // stage st+2, one DMA piece per MFMA triple, and the first group of stage st+1 is fetched
                const bool more = st + 1 < st1, dma = st + 2 < st1;
                if (dma) locate();
                dma_wait_all();
                __syncthreads();
                constexpr int NM = 3 * K, PPS = (NPW + NM - 1) / NM;
                bf16x8 bh[K], bl[K];
                shift_group(pb, bh, bl);
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    mfma_slot(gi, i, bh, bl);
                    if (i == 0 && more) fetch_group(P ^ 1, 0, pb ^ 1);
                    if (dma) {
#pragma unroll
                        for (int pp = 0; pp < PPS; ++pp)
                            if (i * PPS + pp < NPW) issue_piece(P, i * PPS + pp);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (dma) advance();
            }
        }
    };
    for (int st = st0; st < st1; st += 2) {
        stage(std::integral_constant<int, 0>{}, st);
        if (st + 1 < st1) stage(std::integral_constant<int, 1>{}, st + 1);
    }

    // partial[split][m][q], q = ci*T + tap.  C/D layout: column j = lane & 31 (ci), row i = (r&3) + 8*(r>>2) + 4*half (m)
    float* out = p.partial + (long long)split * p.M * p.Q;
    const int ci = ct * 64 + wq * 32 + l32;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = mt * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (mm < p.M && ci < p.Cin) out[(long long)mm * p.Q + ci * T + t] = acc[t][r];
        }
}

// ---- operand preparation: T[n][part][y][x/8][c][8 px] (bf16 head / tail) = padded view of act(IN(concat(src))),
// (y, x) <-> source (y - pad, x - pad); reflection or zero padding inside [0, H+2pad) x [0, W+2pad); zeros beyond,
// and for channels >= C.  grid: (Hp, Cp/64, N); a workgroup transposes one padded row of 64 channels through LDS:
// coalesced fp32 reads along x, 1 KiB contiguous slot writes along c.
struct SplitTParams {
    SrcSeg seg[kMaxSeg];      // chunk_begin = first concat channel
    int nseg;
    int N, C, H, W, pad, pad_mode, Hp, X8, Cp;
    uint4* out;
};

__global__ __launch_bounds__(256) void split_transpose_kernel(const SplitTParams p) {
    __shared__ float f[64][65];
    const int y = blockIdx.x, cg = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
    const int He = p.H + 2 * p.pad, We = p.W + 2 * p.pad;
    int sy = y - p.pad;
    bool yok = y < He;
    if (p.pad_mode == 1) sy = reflect_clamp(sy, p.H);
    else yok = yok && sy >= 0 && sy < p.H;
    uint4* const oh = p.out + (((long long)(n * 2 + 0) * p.Hp + y) * p.X8) * p.Cp + cg * 64;
    uint4* const ol = p.out + (((long long)(n * 2 + 1) * p.Hp + y) * p.X8) * p.Cp + cg * 64;
    for (int x0 = 0; x0 < p.X8 * 8; x0 += 64) {
        // phase 1: thread = (x, channel quarter): 64 consecutive pixels of one channel per wave-load
        const int x = x0 + (tid & 63);
        int sx = x - p.pad;
        bool ok = yok && x < We;
        if (p.pad_mode == 1) sx = reflect_clamp(sx, p.W);
        else ok = ok && sx >= 0 && sx < p.W;
        for (int cl = tid >> 6; cl < 64; cl += 4) {
            const int c = cg * 64 + cl;
            float v = 0.f;
            if (ok && c < p.C) {
                int s = 0;
                if (p.nseg > 1 && c >= p.seg[1].chunk_begin) s = 1;
                if (p.nseg > 2 && c >= p.seg[2].chunk_begin) s = 2;
                const SrcSeg sg = s == 0 ? p.seg[0] : (s == 1 ? p.seg[1] : p.seg[2]);
                const int cs = c - sg.chunk_begin;
                v = sg.data[((long long)n * sg.C + cs) * p.H * p.W + sy * p.W + sx];
                if (sg.mean != nullptr) v = (v - sg.mean[n * sg.C + cs]) * sg.rstd[n * sg.C + cs];
                v = sg.act == 1 ? fmaxf(v, 0.f) : (sg.act == 2 ? (v > 0.f ? v : 0.2f * v) : v);
            }
            f[cl][tid & 63] = v;
        }
        __syncthreads();
        // phase 2: thread = (channel, octet pair): 64 consecutive channels of one octet per wave-store
        const int c = tid & 63;
        for (int o = tid >> 6; o < 8; o += 4) {
            const int xg = x0 / 8 + o;
            if (xg < p.X8) {
                bf16x8 hv, lv;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 h, l;
                    split_bf16(f[c][o * 8 + j], h, l);
                    hv[j] = h;
                    lv[j] = l;
                }
                *reinterpret_cast<bf16x8*>(oh + (long long)xg * p.Cp + c) = hv;
                *reinterpret_cast<bf16x8*>(ol + (long long)xg * p.Cp + c) = lv;
            }
        }
        __syncthreads();
    }
}

}  // namespace apamd
