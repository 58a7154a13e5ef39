// conv_ph4.h -- the four sub-pixel phases of a stride-2 transposed convolution (ConvTranspose2d(3, 2, 1, 1) of the
// generator's decoder; the data gradients of the stride-2 3x3 / 4x4 layers) in ONE tile.
//
// conv_bf16x3's fused-phase launch walks (phase, cout tile) pairs as separate tiles: every pair stages the same
// activation tile again (8 times for 128 output channels) and a 4-tap weight block of which a 3x3 layer uses 9 of 16
// taps, so its stages are LDS-DMA-bound (12..48 MFMAs against 36 KB of DMA).  Here a tile owns 32 output channels of
// ALL four phases: one activation tile per stage, four phase accumulators, only the taps that exist are multiplied
// (compile-time tables for the two geometries that occur: K = 3 and K = 4 with pad 1), and the two x-phases of an
// output row leave as 8-byte stores (the phases interleave along x).
//
// Operands are the ones conv_bf16x3 uses: XS split tensors, and the packed phase blocks [phase][cout tile][chunk] of
// [part][tap][k-group][32 couts] 16-byte slots written by pack_bf16x3_kernel for a 32-cout tile.
#pragma once
#include <utility>

#include "conv_bf16x3.h"

namespace apamd {

template <int KK_, int PARTS_>
struct Ph4Cfg {
    static_assert(KK_ == 3 || KK_ == 4, "pad-1 3x3 and 4x4 stride-2 layers");
    static constexpr int KK = KK_, PARTS = PARTS_;
    static constexpr int CO_TILE = 32, NT = 2, WPX = 4, TH = WPX * NT;
    static constexpr int EXT = KK == 3 ? 1 : 2;                    // reach of the union window of the four phases
    static constexpr int IH = TH + EXT, IW = 32 + EXT, PLANE = IH * IW;
    static constexpr int XP = (2 * PLANE + 63) / 64 * 64;          // slots per part: [k-group][pixel]
    static constexpr int NXP = XP / 64;                            // wave-wide DMA pieces per part
    static constexpr int WB = 256;                                 // slots of one part of a packed phase block: 4 taps x 2 x 32
    static constexpr int NTAPS = KK == 3 ? 9 : 16;                 // taps that exist: only those are staged, one 64-slot
    static constexpr int W_SLOTS = NTAPS * PARTS * 64;             // piece (2 k-groups x 32 couts) per tap and part
    static constexpr int X_SLOTS = PARTS * XP, STAGE = W_SLOTS + X_SLOTS;
    static int wfloats(int) { return 2 * 4 * 2 * CO_TILE * 4; }    // a packed phase block always holds both parts
    static size_t lds_bytes(int) { return (size_t)2 * STAGE * 16; }   // (the statistics slots live in the free stage buffer)
    // phase ph = phy * 2 + phx; its window origin inside the union window and the taps (ly * 2 + lx) it has
    static constexpr int org(int phz) { return KK == 3 ? 0 : phz; }           // K = 4: phase 0 starts one pixel earlier
    static constexpr unsigned mask1(int phz) { return KK == 3 ? (phz ? 3u : 1u) : 3u; }   // taps along one axis
    static constexpr bool has(int ph, int tp) {
        return ((mask1(ph >> 1) >> (tp >> 1)) & 1u) && ((mask1(ph & 1) >> (tp & 1)) & 1u);
    }
    static constexpr int base_d = KK == 3 ? 0 : -1;                // input coordinate of the union window's origin
    static constexpr int tap_code(int i) {                         // phase * 4 + tap of the i-th existing tap
        int n = 0;
        for (int c = 0; c < 16; ++c)
            if (has(c >> 2, c & 3)) {
                if (n == i) return c;
                ++n;
            }
        return 0;
    }
    static constexpr int tap_ph(int i) { return tap_code(i) >> 2; }
    static constexpr int tap_tp(int i) { return tap_code(i) & 3; }
    // Position of existing tap i inside the union window (row * 4 + column).  Several (phase, tap) pairs multiply the SAME
    // activation fragment -- K = 3: 9 pairs over 4 positions, K = 4: 16 pairs over 9 -- so the pairs are walked position by
    // position and a position's fragments are read once: 34 instead of 54 ds_read_b128 per wave and chunk for K = 3 (68 / 96
    // for K = 4).  With one weight fragment against NT = 2 pixel rows the kernel issued one LDS read per MFMA and two
    // workgroups per CU kept the LDS port, not the matrix pipe, busy (30 % MFMA busy, profiles/r03zz_pmc_mfma.md).
    static constexpr int pos_of(int i) { return (org(tap_ph(i) >> 1) + (tap_tp(i) >> 1)) * 4 + org(tap_ph(i) & 1) + (tap_tp(i) & 1); }
    static constexpr int order(int j) {                           // the j-th pair in (position, staging index) order
        int n = 0;
        for (int pos = 0; pos < 16; ++pos)
            for (int i = 0; i < NTAPS; ++i)
                if (pos_of(i) == pos) {
                    if (n == j) return i;
                    ++n;
                }
        return 0;
    }
    static constexpr bool new_pos(int j) { return j == 0 || pos_of(order(j)) != pos_of(order(j - 1)); }
    static constexpr int xbuf_of(int j) {                         // register buffer of the j-th pair's activation fragments
        int b = 0;
        for (int k = 1; k <= j; ++k) b += new_pos(k) ? 1 : 0;
        return b & 1;
    }
};

// 16 per-lane values summed over the 32 lanes of a half-wave as a transpose-reduce: at every step a lane hands half of its
// values to the partner and keeps the sums of the other half (8 + 4 + 2 + 1 exchanges, then one plain step): 16 shuffles per
// quantity instead of 16 x 5.  Lanes with even l32 end up with the total of value index
// ((l32 >> 4) & 1) * 8 + ((l32 >> 3) & 1) * 4 + ((l32 >> 2) & 1) * 2 + ((l32 >> 1) & 1)   (conv_bf16x3's octet epilogue).
__device__ __forceinline__ float fold16_half(const float (&v)[16], int l32) {
    const bool b16 = l32 & 16, b8 = l32 & 8, b4 = l32 & 4, b2 = l32 & 2;
    float w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) w8[i] = (b16 ? v[i + 8] : v[i]) + __shfl_xor(b16 ? v[i] : v[i + 8], 16, 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = (b8 ? w8[i + 4] : w8[i]) + __shfl_xor(b8 ? w8[i] : w8[i + 4], 8, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[i] = (b4 ? w4[i + 2] : w4[i]) + __shfl_xor(b4 ? w4[i] : w4[i + 2], 4, 64);
    const float w1 = (b2 ? w2[1] : w2[0]) + __shfl_xor(b2 ? w2[0] : w2[1], 2, 64);
    return w1 + __shfl_xor(w1, 1, 64);
}

template <class C>
__global__ __launch_bounds__(256, 2) void conv_ph4(const ConvKParams p) {
    constexpr int PARTS = C::PARTS, NT = C::NT, IW = C::IW, PLANE = C::PLANE, XP = C::XP, NXP = C::NXP, WB = C::WB;
    constexpr int W_SLOTS = C::W_SLOTS, STAGE = C::STAGE, CO_TILE = C::CO_TILE;
    constexpr int NXW = (NXP + 3) / 4;                             // activation pieces per wave and part
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* const smem = reinterpret_cast<uint4*>(smem_raw);
    float* const sred = reinterpret_cast<float*>(smem + STAGE);       // [4 phases][WPX][CO_TILE][2]: stage buffer 1, free in the epilogue
    static_assert(STAGE * 16 >= 4 * C::WPX * CO_TILE * 2 * 4, "statistics slots fit a stage buffer");
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wpx = wave;
    const int H = p.H, W = p.W, HW = H * W;
    const int nchunks = p.nchunks, nreal = p.cin_pad >> 4;
    const int co_tiles = p.co_tiles_phase;

    // persistent workgroups over (image, pixel tile, cout tile), cout tile fastest (conv_bf16x3's order)
    int tile, tile_end, tile_step;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int nx = G < 8 ? G : 8;
        const int xcd = b % nx, idx = b / nx;
        const int ntl = p.N * p.tiles_y * p.tiles_x * co_tiles;
        const int q = ntl / nx, r = ntl % nx;
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile_step = (G - xcd + nx - 1) / nx;
        tile = base + idx;
        tile_end = base + q + (xcd < r ? 1 : 0);
    }
    if (tile >= tile_end) return;

    // ---- DMA geometry of this lane: activation piece k of the wave covers slot (wave + 4k) * 64 + lane of a part
    int pgeo[NXW];
#pragma unroll
    for (int k = 0; k < NXW; ++k) {
        const int s = (wave + 4 * k) * 64 + lane;
        const int kg = s >= PLANE ? 1 : 0;
        const int pix = s - kg * PLANE;
        const int ly = pix / IW, lx = pix - ly * IW;
        pgeo[k] = (wave + 4 * k < NXP && s < 2 * PLANE) ? ((kg << 15) | (ly << 8) | lx) : -1;
    }
    struct TileId { int n, cot, ty, tx; };
    auto locate = [&](int logical, TileId& t, int (&goff)[NXW]) __attribute__((always_inline)) {
        t.cot = logical % co_tiles;
        int t_ = logical / co_tiles;
        t.tx = t_ % p.tiles_x;
        t_ /= p.tiles_x;
        t.ty = t_ % p.tiles_y;
        t.n = t_ / p.tiles_y;
        const int iy0 = t.ty * C::TH + C::base_d, ix0 = t.tx * 32 + C::base_d;
#pragma unroll
        for (int k = 0; k < NXW; ++k) {
            int gy = iy0 + ((pgeo[k] >> 8) & 127), gx = ix0 + (pgeo[k] & 255);
            bool ok = pgeo[k] >= 0;
            if (p.pad_mode == 1) {
                gy = reflect_clamp(gy, H);
                gx = reflect_clamp(gx, W);
            } else {
                ok = ok && gy >= 0 && gy < H && gx >= 0 && gx < W;
            }
            goff[k] = (ok ? ((pgeo[k] >> 15) & 1) * (HW + 1) + gy * W + gx : HW) * 16;
        }
    };
    auto seg_of = [&](int chunk) {
        int s = 0;
        if (p.nseg > 1 && chunk >= p.seg[1].chunk_begin) s = 1;
        if (p.nseg > 2 && chunk >= p.seg[2].chunk_begin) s = 2;
        return s;
    };
    // weight piece j = (tap i = j / PARTS of the list, part j % PARTS) goes to wave j % 4.  Its source is the packed block of the
    // tap's PHASE: block (phase, cout tile, chunk) lies phase * co_tiles * nchunks blocks behind block (0, cout tile, chunk), so
    // the phase is folded into the piece's per-lane byte offset (a constant of the kernel) and a stage has ONE scalar weight base.
    constexpr int NWJ = (C::NTAPS * PARTS + 3) / 4;
    unsigned woff[NWJ];
    {
        const long long phase_bytes = (long long)co_tiles * nchunks * p.wfloats * 4;    // < 2^30 (host: conv2d_fwd_impl checks)
#pragma unroll
        for (int jj = 0; jj < NWJ; ++jj) {
            const int j = jj * 4 + wave, i = j / PARTS, part = j - i * PARTS;
            int code = 0, n = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (C::has(c >> 2, c & 3)) {
                    if (n == i) code = c;
                    ++n;
                }
            woff[jj] = (unsigned)((code >> 2) * phase_bytes) + (unsigned)((part * WB + (code & 3) * 64) * 16) + lane * 16;
        }
    }
    // One stage = the existing taps of the four phase blocks of (cout tile, chunk) + the activation tile of the chunk.
    // As in conv_bf16x3 (profiles/r05_dominant_cycle_account.md): the scalar bases are computed once per stage and pinned in
    // SGPRs, a piece is {s_add m0, base, literal; global_load_lds v_off, s[base]} and nothing else.  (Before: every piece
    // re-derived its 64-bit source address -- ten to fourteen scalar multiplies / adds --, read spilled scalars back with
    // v_readlane and tested its own validity: 260-350 scalar multiplies in a kernel of 18-96 MFMAs.)  alt / use_alt: the tile's
    // last chunk stages the NEXT tile's chunk 0 with that tile's offsets.
    auto issue = [&](int n_, int cot_, const int (&goff)[NXW], const int (&alt)[NXW], bool use_alt, int chunk_, int buf)
                     __attribute__((always_inline)) {
        const int n = __builtin_amdgcn_readfirstlane(n_), cot = __builtin_amdgcn_readfirstlane(cot_);
        chunk_ = __builtin_amdgcn_readfirstlane(chunk_);
        const int chunk = chunk_ < nreal ? chunk_ : nreal - 1;      // a padding chunk (zero weights) re-stages real, finite data
        const int s = seg_of(chunk);
        const int cg0 = (chunk - p.seg[s].chunk_begin) * 2;
        const int CG = p.seg[s].C >> 3;
        const unsigned char* xs = reinterpret_cast<const unsigned char*>(p.seg[s].data);
        const unsigned char* xh = xs + ((long long)(n * 2 + 0) * CG + cg0) * (HW + 1) * 16;
        const unsigned char* xl = xs + ((long long)(n * 2 + 1) * CG + cg0) * (HW + 1) * 16;
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp + ((long long)cot * nchunks + chunk_) * p.wfloats);
        unsigned wdst = lds0 + (buf * STAGE + wave * 64) * 16;
        unsigned xdst = lds0 + (buf * STAGE + W_SLOTS + wave * 64) * 16;
        asm volatile("" : "+s"(xh), "+s"(xl), "+s"(wsrc), "+s"(wdst), "+s"(xdst));
        static_for<NWJ>([&](auto jt) __attribute__((always_inline)) {
            constexpr int jj = decltype(jt)::value;
            constexpr bool whole = jj * 4 + 3 < C::NTAPS * PARTS;   // every wave has a piece jj
            if (whole || jj * 4 + wave < C::NTAPS * PARTS) glds16_si<jj * 4096>(wsrc, woff[jj], wdst);
        });
        static_for<PARTS * NXW>([&](auto jt) __attribute__((always_inline)) {
            constexpr int part = decltype(jt)::value / NXW, k = decltype(jt)::value % NXW;
            constexpr bool whole = 4 * k + 3 < NXP;
            if (whole || wave + 4 * k < NXP)
                glds16_si<(part * XP + k * 256) * 16>(part ? xl : xh, (unsigned)(use_alt ? alt[k] : goff[k]), xdst);
        });
    };

    f32x16 acc[4][NT];
    TileId cur, nxt;
    int cgoff[NXW], ngoff[NXW];
    locate(tile, cur, cgoff);
    issue(cur.n, cur.cot, cgoff, cgoff, false, 0, 0);
    for (;;) {
        const bool has_next = tile + tile_step < tile_end;
        locate(has_next ? tile + tile_step : tile, nxt, ngoff);
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][q][r] = 0.f;
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;                                  // (nchunks is even: every tile starts in buffer 0)
            dma_wait_all();
            __syncthreads();        // stage c has landed for every wave; nobody reads the other buffer any more
            {
                // the tile's last chunk stages chunk 0 of the next tile -- without a next tile this tile's own chunk 0 again,
                // which nobody reads: one unconditional issue sequence (the two call sites were two copies of it)
                const bool tailc = c + 1 >= nchunks;
                issue(tailc ? nxt.n : cur.n, tailc ? nxt.cot : cur.cot, cgoff, ngoff, tailc, tailc ? 0 : c + 1, buf ^ 1);
            }
            const uint4* Wc = smem + buf * STAGE + half * CO_TILE + l32;
            const uint4* Xc = smem + buf * STAGE + W_SLOTS + half * PLANE + (wpx * NT) * IW + l32;
            // the pairs that exist, position by position; the fragments of pair j + 1 are read while pair j multiplies
            bf16x8 ah[2], al[2], xh[2][NT], xl[2][NT];
            auto fetch_w = [&](int i, int fb) __attribute__((always_inline)) {
                ah[fb] = *reinterpret_cast<const bf16x8*>(Wc + (i * PARTS) * 64);
                if constexpr (PARTS == 2) al[fb] = *reinterpret_cast<const bf16x8*>(Wc + (i * PARTS + 1) * 64);
            };
            auto fetch_x = [&](int pos, int xb) __attribute__((always_inline)) {
                const int toff = (pos >> 2) * IW + (pos & 3);
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    xh[xb][q] = *reinterpret_cast<const bf16x8*>(Xc + toff + q * IW);
                    if constexpr (PARTS == 2) xl[xb][q] = *reinterpret_cast<const bf16x8*>(Xc + XP + toff + q * IW);
                }
            };
            fetch_w(C::order(0), 0);
            fetch_x(C::pos_of(C::order(0)), 0);
            // (the pair index is a TEMPLATE constant: with a run-time loop variable the constexpr tables above are evaluated at run
            // time for K = 4 -- 16 pairs -- and the fragment buffers, indexed by their results, move to scratch memory)
            static_for<C::NTAPS>([&](auto jt) __attribute__((always_inline)) {
                constexpr int j = decltype(jt)::value;
                constexpr int fb = j & 1, xb = C::xbuf_of(j), ph = C::tap_ph(C::order(j));
                if constexpr (j + 1 < C::NTAPS) {
                    fetch_w(C::order(j + 1), fb ^ 1);
                    if constexpr (C::new_pos(j + 1)) fetch_x(C::pos_of(C::order(j + 1)), C::xbuf_of(j + 1));
                }
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    if constexpr (PARTS == 2) {
                        acc[ph][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[fb], xh[xb][q], acc[ph][q], 0, 0, 0);
                        acc[ph][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[fb], xl[xb][q], acc[ph][q], 0, 0, 0);
                    }
                    acc[ph][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[fb], xh[xb][q], acc[ph][q], 0, 0, 0);
                }
            });
        }
        // ---- epilogue.  MFMA C/D layout: pixel column = lane & 31, cout row = (r & 3) + 8 (r >> 2) + 4 half.  The phases
        // (phy, 0) and (phy, 1) are the even / odd output columns of one row.
        // FAST: no activation, no bias (compile time), whole pixel pairs and 16-byte aligned rows -- what every InstanceNorm-ed layer and
        // every data gradient is.  The two lanes of a pixel pair (ox even, ox + 1) hold output columns 2 ox .. 2 ox + 3 of BOTH
        // pixel rows of the tile; the even lane hands its row-1 pair to the odd lane and takes that lane's row-0 pair (DPP
        // quad_perm [1,0,3,2]), so each lane stores one row as 4 consecutive floats: half the store instructions of the 8-byte
        // form.  The destination is a per-lane pointer advanced by additions (the generic form below re-derives the 64-bit
        // address, tests the activation code and loads its bias in front of every store: 26 k instructions for 64 stores).
        auto epilogue = [&](auto fasttag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fasttag)::value;
            const int n = cur.n, cot = cur.cot;
            const int ox = cur.tx * 32 + l32;
            const bool want = p.stats != nullptr;
            if (want) __syncthreads();          // every wave is done with the fragments of stage buffer 1: it holds sred now
            const bool odd = l32 & 1;
            const int oyq = cur.ty * C::TH + wpx * NT + (odd ? 1 : 0);            // FAST: the pixel row this lane stores
            float* lane_dst = p.y + (long long)n * p.o_nstride + (long long)(cot * CO_TILE + 4 * half) * p.o_cstride +
                              (long long)(oyq * 2) * p.o_rstride + (ox & ~1) * 2;
            const long long cs = p.o_cstride;
#pragma unroll
            for (int phy = 0; phy < 2; ++phy) {
                float* dst_r = lane_dst + (long long)phy * p.o_rstride;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int co = cot * CO_TILE + row;
                    const bool cok = co < p.Cout;
                    const float bv = (!FAST && p.bias != nullptr && cok) ? p.bias[co] : 0.f;     // (FAST: no bias)
                    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
                    if constexpr (FAST) {
                        static_assert(NT == 2, "one pixel row per lane of a pair");
                        float own[2][2];
#pragma unroll
                        for (int q = 0; q < NT; ++q) {
                            own[q][0] = acc[phy * 2][q][r] + bv;
                            own[q][1] = acc[phy * 2 + 1][q][r] + bv;
                        }
                        const float g0 = odd ? own[0][0] : own[1][0], g1 = odd ? own[0][1] : own[1][1];     // the pair this lane gives away
                        const float r0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(g0), 0xB1, 0xF, 0xF, true));
                        const float r1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(g1), 0xB1, 0xF, 0xF, true));
                        if (cok && oyq < p.OH && ox < p.OW)
                            *reinterpret_cast<float4*>(dst_r) = odd ? make_float4(r0, r1, own[1][0], own[1][1])
                                                                    : make_float4(own[0][0], own[0][1], r0, r1);
                        dst_r += ((r & 3) == 3 ? 5 : 1) * cs;                     // rows 0..3, 8..11, 16..19, 24..27 (+ 4 half)
                    } else {
#pragma unroll
                        for (int q = 0; q < NT; ++q) {
                            const int oy = cur.ty * C::TH + wpx * NT + q;
                            const bool inside = oy < p.OH && ox < p.OW;
                            const float v0 = acc[phy * 2][q][r] + bv, v1 = acc[phy * 2 + 1][q][r] + bv;
                            if (inside) { s0 += v0; q0 += v0 * v0; s1 += v1; q1 += v1 * v1; }
                            if (inside && cok) {
                                float* dst = p.y + (long long)n * p.o_nstride + (long long)co * p.o_cstride +
                                             (long long)(oy * 2 + phy) * p.o_rstride + ox * 2;
                                *reinterpret_cast<float2*>(dst) = make_float2(apply_act(v0, p.act), apply_act(v1, p.act));
                            }
                        }
                    }
                    if (!FAST && want) {
                        float v[4] = {s0, q0, s1, q1};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {               // sum over the 32 lanes of the half-wave
                            v[k] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[k]), (1 << 10) | 0x1f));
                            v[k] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[k]), (2 << 10) | 0x1f));
                            v[k] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[k]), (4 << 10) | 0x1f));
                            v[k] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[k]), (8 << 10) | 0x1f));
                            v[k] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[k]), (16 << 10) | 0x1f));
                        }
                        if (l32 == 0) {
                            float* d0 = sred + (((phy * 2) * C::WPX + wpx) * CO_TILE + row) * 2;
                            float* d1 = sred + (((phy * 2 + 1) * C::WPX + wpx) * CO_TILE + row) * 2;
                            d0[0] = v[0]; d0[1] = v[1];
                            d1[0] = v[2]; d1[1] = v[3];
                        }
                    }
                }
            }
            if constexpr (FAST) {
                if (want) {
                    // row sums of the four phases: one transpose-reduce per (phase, quantity) over the 16 cout rows a lane holds
                    const bool in0 = cur.ty * C::TH + wpx * NT < p.OH && ox < p.OW, in1 = cur.ty * C::TH + wpx * NT + 1 < p.OH && ox < p.OW;
                    const int rr = ((l32 >> 4) & 1) * 8 + ((l32 >> 3) & 1) * 4 + ((l32 >> 2) & 1) * 2 + ((l32 >> 1) & 1);
                    const int my_row = (rr & 3) + 8 * (rr >> 2) + 4 * half;
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph) {
                        float st, qt;
                        {
                            float v[16];
#pragma unroll
                            for (int r = 0; r < 16; ++r) v[r] = (in0 ? acc[ph][0][r] : 0.f) + (in1 ? acc[ph][1][r] : 0.f);
                            st = fold16_half(v, l32);
                        }
                        {
                            float v[16];
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float a = in0 ? acc[ph][0][r] : 0.f, b = in1 ? acc[ph][1][r] : 0.f;
                                v[r] = a * a + b * b;
                            }
                            qt = fold16_half(v, l32);
                        }
                        if ((l32 & 1) == 0) {
                            float* d = sred + ((ph * C::WPX + wpx) * CO_TILE + my_row) * 2;
                            d[0] = st;
                            d[1] = qt;
                        }
                    }
                }
            }
            if (want) {
                __syncthreads();
                if (tid < 4 * CO_TILE) {
                    const int ph = tid >> 5, c = tid & 31;
                    const int co = cot * CO_TILE + c;
                    if (co < p.Cout) {
                        float s = 0.f, q2 = 0.f;
#pragma unroll
                        for (int w = 0; w < C::WPX; ++w) {
                            s += sred[((ph * C::WPX + w) * CO_TILE + c) * 2];
                            q2 += sred[((ph * C::WPX + w) * CO_TILE + c) * 2 + 1];
                        }
                        float* d = p.stats + (((long long)n * p.Cout + co) * p.stat_tiles + p.ph_stat[ph] + cur.ty * p.tiles_x + cur.tx) * 2;
                        d[0] = s;
                        d[1] = q2;
                    }
                }
            }
        };
        // (wave-uniform) whole pixel pairs, 16-byte aligned rows, no activation, no bias
        if (p.act == 0 && p.bias == nullptr && (p.OW & 1) == 0 && (p.o_rstride & 3) == 0 && (p.o_cstride & 3) == 0 && (p.o_nstride & 3) == 0 &&
            (reinterpret_cast<uintptr_t>(p.y) & 15) == 0)
            epilogue(std::true_type{});
        else
            epilogue(std::false_type{});
        if (!has_next) break;
        cur = nxt;
#pragma unroll
        for (int k = 0; k < NXW; ++k) cgoff[k] = ngoff[k];
        tile += tile_step;
    }
    dma_wait_all();                                                // (the last tile's last chunk staged a chunk nobody reads)
}

}  // namespace apamd
