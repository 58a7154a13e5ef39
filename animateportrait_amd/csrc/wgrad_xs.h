// wgrad_xs.h -- the weight gradient of wgrad_bf16x3.h with BOTH operands read in the layout the convolutions stage
// (conv_prepass.h: XS[n][part][C/8][HW + 1] slots of 8 channels of one pixel), so that no operand is prepared for it:
// the shifted operand is the split copy the forward pass staged of the layer's sources, the M-role operand the split copy
// of dy that ap_instnorm_bwd_split writes for the data-gradient convolution anyway.
//
// An MFMA operand fragment is 8 consecutive K elements per lane -- K is the pixel index here -- while a slot holds 8
// CHANNELS of one pixel: the fragments are read with ds_read_b64_tr_b16, which hands each lane of a 16-lane group the
// column it needs of a 4 (pixels) x 16 (channels) block whose rows the 16 lanes address (lane 4 j + q: pixel j, channel
// quad q; measured semantics: tools/probe/tr_probe.hip).  Two such reads make a fragment (pixels 0..3 and 4..7 of the
// half-wave's octet); a tap (ky, kx) is an immediate offset on the same base address: no funnel shifts, no raw-octet ring.
//
// LDS image of a stage (2 output rows x 32 columns), per part:
//   G: [channel octet (M_TILE / 8)][row 2][pixel 32]          slots, octet stride G_OS
//   A: [channel octet 8][row 2 + K - 1][pixel 32 + K - 1]     slots, octet stride A_OS
// octet strides = 4 (mod 16) slots: the four channel octets a 32-lane group reads land on disjoint 16-bank groups.
// Staging is a per-lane gather (one 16-byte slot per lane and LDS-DMA piece): padding is index arithmetic (reflection) or
// the copy's all-zero slot HW.
//
// Served (wgrad_host.hip, ap_conv2d_wgrad_xs): 3x3 stride-1 layers; 4x4 stride-1 layers (split bf16; 16 accumulator tiles: the
// 4-wave workgroup); the 2x2 space-to-depth forms of stride-2 layers, from the forward pass's space-to-depth copy or with the
// view gathered from its plain copy (WgradXsParams::s2d_c).  Measurements and what was tried: profiles/r05_wgrad_routes.md.
#pragma once
#include "wgrad_bf16x3.h"

namespace apamd {

struct WgradXsParams {
    const uint4* g_xs;            // split copy of dy: [N][2][M / 8][GH * GW + 1]
    const uint4* a_xs[kMaxSeg];   // split copy of each source segment: [N][2][C_s / 8][H * W + 1]
    int a_cg_begin[kMaxSeg + 1];  // first channel octet of each segment; [nseg] = Cin / 8
    int nseg;
    int N, M, GH, GW, H, W, pad, pad_mode;
    int s2d_c;                    // C0 > 0: the shifted operand is the space-to-depth VIEW (4 C0 channels, H x W = H0/2 + 1 x W0/2 + 1, pad 0)
                                  // gathered from the PLAIN copy of the (C0, 2 (H - 1), 2 (W - 1)) source: view channel r * C0 + c, pixel
                                  // (y, x) = pad1(source)[c][2 y + (r >> 1)][2 x + (r & 1)]  (a stride-2 layer whose forward pass staged that copy)
    int tiles_x, tiles_y, nstages, P, m_tiles, c_tiles;
    float* partial;               // [P][tile][wave][tap][16][64]
};

template <int K_, int PARTS_, int WM_>
struct WgradXsCfg {
    static constexpr int K = K_, T = K * K, PARTS = PARTS_, WM = WM_, PR = 2;
    static constexpr int NWAVES = 2 * WM, NT = NWAVES * 64, M_TILE = 32 * WM, GOCT = M_TILE / 8, AOCT = 8;
    static constexpr int ROWS = PR + K - 1, AW = 32 + K - 1;
    static constexpr int pad4(int v) { return v + ((4 - v % 16) + 16) % 16; }          // -> 4 (mod 16)
    static constexpr int G_OS = pad4(PR * 32), A_OS = pad4(ROWS * AW);
    static constexpr int G_PART = GOCT * G_OS, A_PART = AOCT * A_OS, PART = G_PART + A_PART;
    static constexpr int STAGE = PARTS * PART;                                         // slots
    static constexpr int NPT = (PART + NT - 1) / NT;                                   // DMA pieces per thread, part and stage
    static constexpr size_t lds_bytes() { return (size_t)2 * STAGE * 16; }
    static constexpr int WG_PER_CU = (2 * lds_bytes() <= 160 * 1024 && T * 16 <= 160) ? 2 : 1;
};

typedef short v4i16 __attribute__((ext_vector_type(4)));

// LDS-DMA with a per-lane 64-bit source address (lane i lands at lds_addr + 16 i)
__device__ __forceinline__ void glds16_vv(const void* src, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}

template <class C>
__global__ __launch_bounds__(C::NT, C::WG_PER_CU) void wgrad_xs_kernel(const WgradXsParams p) {
    constexpr int K = C::K, T = C::T, PARTS = C::PARTS, PROD = PARTS == 1 ? 1 : 3, WM = C::WM, NT = C::NT;
    constexpr int ROWS = C::ROWS, AW = C::AW, G_OS = C::G_OS, A_OS = C::A_OS, G_PART = C::G_PART, PART = C::PART, STAGE = C::STAGE;
    constexpr int NPT = C::NPT, GOCT = C::GOCT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave % WM, wq = wave / WM;
    int b = blockIdx.x;
    const int split = b % p.P; b /= p.P;
    const int ct = b % p.c_tiles, mt = b / p.c_tiles;
    const int st0 = (int)((long long)p.nstages * split / p.P), st1 = (int)((long long)p.nstages * (split + 1) / p.P);
    const int GHW = p.GH * p.GW, HW = p.H * p.W, MG = p.M >> 3;

    // ---- what this thread stages: slot tid + NT i of a part's image, i < NPT -- decoded once
    //   code = kind << 28 | octet << 20 | row << 12 | pixel      (kind 0: G, 1: A, 3: padding slot, not staged)
    int code[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int s = tid + NT * i;
        int c = 3 << 28;
        if (s < G_PART) {
            const int o = s / G_OS, r = s - o * G_OS;
            if (r < C::PR * 32) c = (0 << 28) | (o << 20) | ((r >> 5) << 12) | (r & 31);
        } else if (s < PART) {
            const int sa = s - G_PART, o = sa / A_OS, r = sa - o * A_OS;
            if (r < ROWS * AW) { const int row = r / AW; c = (1 << 28) | (o << 20) | (row << 12) | (r - row * AW); }
        }
        code[i] = c;
    }
    // per-octet plane bases (slots, part 0) that do not change with the stage: the image index is added per stage
    // (a thread's pieces cover at most NPT different octets: kept as the octet's offset inside an image)
    int obase[NPT];                // slots from the copy's start to (image 0, part 0, this octet's plane)   (the host checks that a
    int ostride[NPT];              // slots per (image, part) of that copy: CG * (HW + 1)                      copy has < 2^31 slots)
    int which[NPT];                // 0: g_xs, 1..3: a_xs[which - 1]
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int kind = code[i] >> 28, o = (code[i] >> 20) & 0x3f;
        obase[i] = 0; ostride[i] = 0; which[i] = -1;
        if (kind == 0) {
            const int go = mt * GOCT + o;
            if (go < MG) { obase[i] = go * (GHW + 1); ostride[i] = MG * (GHW + 1); which[i] = 0; }
        } else if (kind == 1) {
            const int ao = ct * 8 + o;
            if (p.s2d_c > 0) {
                const int CG0 = p.s2d_c >> 3, H0 = 2 * (p.H - 1), W0 = 2 * (p.W - 1);
                if (ao < 4 * CG0) {
                    const int r = ao / CG0;
                    obase[i] = (ao - r * CG0) * (H0 * W0 + 1);
                    ostride[i] = CG0 * (H0 * W0 + 1);
                    which[i] = 1;
                    code[i] |= r << 26;                              // the phase of this octet (bits 26-27)
                }
            } else if (ao < p.a_cg_begin[p.nseg]) {
                int sg = 0;
                if (p.nseg > 1 && ao >= p.a_cg_begin[1]) sg = 1;
                if (p.nseg > 2 && ao >= p.a_cg_begin[2]) sg = 2;
                const int cgs = p.a_cg_begin[sg + 1] - p.a_cg_begin[sg];
                obase[i] = (ao - p.a_cg_begin[sg]) * (HW + 1);
                ostride[i] = cgs * (HW + 1);
                which[i] = 1 + sg;
            }
        }
    }
    // channel octets beyond the operands (M or Cin not a multiple of the tile): their slots must hold zeros -- cleared once,
    // never staged (both stage buffers)
    {
        uint4* z = reinterpret_cast<uint4*>(smem_raw);
        for (int i = tid; i < 2 * STAGE; i += NT) z[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }

    // piece i of stage st (its position decoded by the caller, once per stage)
    auto issue_piece = [&](int i, int n, int ity, int itx, int buf) __attribute__((always_inline)) {
        if (which[i] < 0) return;
        const int kind = code[i] >> 28, row = (code[i] >> 12) & 0xff, px = code[i] & 0xfff;
        int pix;
        if (kind == 0) {
            const int y = ity * C::PR + row, x = itx * 32 + px;
            pix = (y < p.GH && x < p.GW) ? y * p.GW + x : GHW;
        } else {
            int y = ity * C::PR + row - p.pad, x = itx * 32 + px - p.pad;
            bool ok = true;
            if (p.s2d_c > 0) {
                const int r = (code[i] >> 26) & 3, H0 = 2 * (p.H - 1), W0 = 2 * (p.W - 1);
                const int yy = 2 * y + (r >> 1) - 1, xx = 2 * x + (r & 1) - 1;
                ok = y < p.H && x < p.W && yy >= 0 && yy < H0 && xx >= 0 && xx < W0;
                pix = ok ? yy * W0 + xx : H0 * W0;
            } else if (p.pad_mode == 1) {
                // (rows / columns of tiles beyond the output grid: their G slots are zero, any finite value will do)
                y = reflect_clamp(y < p.H + p.pad ? y : p.H - 1, p.H);
                x = reflect_clamp(x < p.W + p.pad ? x : p.W - 1, p.W);
            } else {
                ok = y >= 0 && y < p.H && x >= 0 && x < p.W;
            }
            if (p.s2d_c == 0) pix = ok ? y * p.W + x : HW;
        }
        const uint4* base = which[i] == 0 ? p.g_xs : (which[i] == 1 ? p.a_xs[0] : (which[i] == 2 ? p.a_xs[1] : p.a_xs[2]));
#pragma unroll
        for (int part = 0; part < PARTS; ++part) {
            const uint4* src = base + ((n * 2 + part) * ostride[i] + obase[i] + pix);
            glds16_vv(src, lds0 + (unsigned)((buf * STAGE + part * PART + NT * i + wave * 64) * 16));
        }
    };
    auto issue_stage = [&](int st, int buf) __attribute__((always_inline)) {
        const int itx = st % p.tiles_x, t2 = st / p.tiles_x, ity = t2 % p.tiles_y, n = t2 / p.tiles_y;
#pragma unroll
        for (int i = 0; i < NPT; ++i) issue_piece(i, n, ity, itx, buf);
    };

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- fragment addresses: lane L: group g = L >> 4 (half = g >> 1: pixels 8 half .., cgrp = g & 1: channels 16 cgrp ..),
    // i = L & 15: pixel j = i >> 2 of the read's four, channel quad q = i & 3
    const int half = lane >> 5, cgrp = (lane >> 4) & 1, jq = lane & 15, pj = jq >> 2, q = jq & 3;
    // G: channels wm * 32 + 16 cgrp + 4 q ..: octet wm * 4 + cgrp * 2 + (q >> 1), byte (q & 1) * 8; pixel half * 8 + pj (+ 4 for the second read)
    const unsigned g_addr = lds0 + (unsigned)(((wm * 4 + cgrp * 2 + (q >> 1)) * G_OS + half * 8 + pj) * 16 + (q & 1) * 8);
    const unsigned a_addr = lds0 + (unsigned)((G_PART + (wq * 4 + cgrp * 2 + (q >> 1)) * A_OS + half * 8 + pj) * 16 + (q & 1) * 8);
    auto tr = [&](unsigned addr) __attribute__((always_inline)) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(uintptr_t)addr);
    };
    auto frag = [&](unsigned addr) __attribute__((always_inline)) {          // pixels 0..3 and 4..7 of the lane's octet
        const v4i16 lo = tr(addr), hi = tr(addr + 4 * 16);
        typedef short v8i16 __attribute__((ext_vector_type(8)));
        const v8i16 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    if (st0 < st1) issue_stage(st0, 0);
    dma_wait_all();
    __syncthreads();
    // A stage = 2 output rows x 32 columns.  The B fragment of tap (ky, kx) for output row py is the one of tap (ky + 1, kx) for row
    // py - 1: walked by STAGED row r = py + ky, every fragment is read once and multiplies for both output rows (24 fragment sets
    // per stage and part instead of 36).  Groups g = (column half xh, staged row r), 2 x ROWS of them; the fragments of group
    // g + 1 are requested before the products of group g, and the next stage's DMA pieces go out one per group.
    constexpr int NGRP = 2 * ROWS;
    static_assert(NPT <= NGRP, "one DMA piece per group");
    for (int st = st0; st < st1; ++st) {
        const int buf = (st - st0) & 1;
        const bool more = st + 1 < st1;
        const int nst = st + 1;
        const int nitx = nst % p.tiles_x, nt2 = nst / p.tiles_x, nity = nt2 % p.tiles_y, nn = nt2 / p.tiles_y;
        const unsigned sb = (unsigned)(buf * STAGE * 16);
        bf16x8 ah[2], al[2], bh[2][K], bl[2][K];
        auto fetch_b = [&](int g, int slot) __attribute__((always_inline)) {
            const int xh = g / ROWS, r = g % ROWS;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const unsigned off = (unsigned)((r * AW + xh * 16 + kx) * 16);
                bh[slot][kx] = frag(a_addr + sb + off);
                if constexpr (PARTS == 2) bl[slot][kx] = frag(a_addr + sb + (unsigned)(PART * 16) + off);
            }
        };
        auto fetch_g = [&](int xh) __attribute__((always_inline)) {
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                ah[py] = frag(g_addr + sb + (unsigned)((py * 32 + xh * 16) * 16));
                if constexpr (PARTS == 2) al[py] = frag(g_addr + sb + (unsigned)((PART + py * 32 + xh * 16) * 16));
            }
        };
        fetch_g(0);
        fetch_b(0, 0);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int r = g % ROWS, slot = g & 1;
            if (g + 1 < NGRP) fetch_b(g + 1, slot ^ 1);
            if (g < NPT && more) issue_piece(g, nn, nity, nitx, buf ^ 1);
#pragma unroll
            for (int pr = (PROD == 1 ? 2 : 0); pr < 3; ++pr)
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int ky = r - py;
                    if (ky < 0 || ky >= K) continue;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        f32x16& a = acc[ky * K + kx];
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pr == 0 ? al[py] : ah[py], pr == 1 ? bl[slot][kx] : bh[slot][kx], a, 0, 0, 0);
                    }
                }
            if (r == ROWS - 1 && g + 1 < NGRP) fetch_g(1);       // (the G fragments of the second column half: after their last use)
        }
        dma_wait_all();
        __syncthreads();
    }

    const long long tile_floats = (long long)C::NWAVES * T * 1024;
    float* out = p.partial + ((long long)split * p.m_tiles * p.c_tiles + (long long)mt * p.c_tiles + ct) * tile_floats +
                 (long long)wave * T * 1024 + lane;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64] = acc[t][r];
}

}  // namespace apamd
