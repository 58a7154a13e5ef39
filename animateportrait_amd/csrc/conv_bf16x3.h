// conv_bf16x3.h -- the implicit-GEMM convolution of conv_igemm.h on the bf16 matrix pipe, at fp32-class accuracy.
//
// gfx950 has no TF32-like mode and its exact-fp32 MFMA runs at 1/16 of the bf16 rate.  This path splits
// every fp32 operand into a bf16 head and a bf16 tail (x = xh + xl, |xl| <= 2^-9 |x|) and evaluates
//     x * w  ~=  xh*wh + xh*wl + xl*wh            (the dropped xl*wl term is <= 2^-18 |x w|)
// with three v_mfma_f32_32x32x16_bf16 per tile and fp32 accumulation: 3/16 of the fp32-MFMA time per FLOP.
// bf16 keeps the fp32 exponent range, so there is no overflow / subnormal hazard (an fp16 split would need
// denormal-preserving MFMA inputs).  Measured on the full generator (ngf=64): L-inf 1.4e-4 vs the fp32
// reference, against the 1e-3 budget of BASELINE.json and 7.8e-2 for plain bf16.
//
// Data flow (same im2col-free scheme as conv_igemm.h; reference layers Module2/models/networks.py:1251, 2329-2421):
//   * split_prepass_kernel (one streaming pass per activation tensor, shared by all its consumers) applies the
//     producer's InstanceNorm + activation and writes the split tensor XS[n][head|tail][C/8][H*W + 1][8 x bf16]
//     (16-byte slots; the extra slot of every plane is all-zero and serves the zero-padding taps);
//   * the convolution stages BOTH operands with global_load_lds_dwordx4 only -- no staging registers, no VALU:
//     activation tile [head|tail][k-group][IH*IW px] slots (reflection / zero padding = per-lane source address),
//     weights pre-packed as the LDS image [head|tail][tap][k-group][cout] slots;
//   * channel chunk = 16 (one MFMA K); an A / B fragment is ONE ds_read_b128 per lane (lane = cout / pixel,
//     half-wave = k-group), conflict-free; fragments of tap t+1 are fetched while tap t multiplies;
//   * the workgroups are PERSISTENT: one per CU, each walking its share of the (image, pixel tile, cout tile)
//     list.  The (tile, chunk) stages form one flat software pipeline -- stage g+2 is issued to the LDS-DMA
//     while stage g multiplies, across tile boundaries -- so the first-stage latency of a tile and the drain of
//     the previous tile's output stores hide under MFMA work; the one barrier per stage sits in front of the
//     LAST tap of a stage (whose fragments are already in registers), so the matrix pipe never idles at it;
//   * epilogue identical to the fp32 kernel (bias, activation, InstanceNorm partial statistics).
#pragma once
#include <type_traits>
#include <utility>

#include "conv_igemm.h"

namespace apamd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// K_ > 0: dense K x K taps.  K_ == 0: NTAP_ (1, 2 or 4) taps at run-time offsets inside a 2 x 2 window (the
// sub-pixel phases of a stride-2 transposed convolution); the count is a template parameter so that the tap
// loop unrolls with compile-time register buffers.
// ROW_ = 1: a 1 x K kernel (K horizontal taps).  The 7x7 stems (3 input channels) run this way: their input is
// expanded to "row channels" (ky, c) -- 21 of 32 channels of a split tensor -- so that the 49 x 3 products of a
// pixel become 2 channel chunks x 7 taps (ap_split_prepass_rows).
// PARTS_ = 2: split-bf16 arithmetic (head + tail of both operands staged, three MFMAs per product).
// PARTS_ = 1: plain bf16 arithmetic (AP_PRECISION_BF16, the training configurations): only the head parts are
// staged -- half the LDS image and half the LDS-DMA traffic -- and a product is ONE MFMA (fp32 accumulation).  The
// global operand layouts (XS tensors, packed weights) are the same; the kernel just skips the tail planes.
// S2D3_ = 1 (K_ == 0, NTAP_ == 4): the space-to-depth form of a 3x3 stride-2 layer with an even chunk count per input phase --
// the four phases' tap sets (4 / 2 / 2 / 1 taps) are compile-time: four chunk loops in sequence, see `stage`.
// FNORM_ = 1 (inference, stride-1 3x3 tall tile): InstanceNorm (+ activation, + residual) of the layer's OWN output inside the
// epilogue -- the workgroups that hold the tiles of one (image, cout tile) exchange their per-channel sums through global
// memory (two rounds: mean, then centred squares), so the raw fp32 output never travels to HBM and the norm_split pass
// between two convolutions disappears.  See fused_norm_epilogue in the kernel and ap_conv2d_fwd_norm.
// OB16_ = 1 (plain bf16 arithmetic, dense 3x3 stride 1): the OUTPUT is stored as bf16 (round to nearest even; statistics from the
// fp32 accumulators) -- the raw outputs of the ResNet trunk and the gradients that leave its data-gradient convolutions in the
// plain-bf16 train step (ap_conv2d_fwd_bf16out): every reader of those tensors rounds them to bf16 anyway.
template <int S_, int K_, int WCO_, int MT_, int WPX_, int NT_, int NTAP_ = 0, int ROW_ = 0, int PARTS_ = 2, int S2D3_ = 0, int FNORM_ = 0,
          int OB16_ = 0>
struct Bf3Cfg {
    static constexpr int FNORM = FNORM_;
    static constexpr int OB16 = OB16_;
    static_assert(!OB16_ || (K_ == 3 && S_ == 1 && PARTS_ == 1 && !FNORM_ && !ROW_), "bf16 output: the plain-bf16 3x3 stride-1 tiles");
    static_assert(!FNORM_ || (K_ == 3 && S_ == 1 && WCO_ == 1 && PARTS_ == 2), "fused normalisation: the 3x3 stride-1 split-bf16 tile");
    static constexpr int CI = 16, S = S_, K = K_, WCO = WCO_, MT = MT_, WPX = WPX_, NT = NT_, ROW = ROW_, PARTS = PARTS_;
    static constexpr int S2D3 = S2D3_;
    static_assert(!S2D3_ || (K_ == 0 && NTAP_ == 4), "compile-time s2d tap sets: the 4-tap run-time-tap family");
    static_assert(PARTS == 1 || PARTS == 2, "head only, or head + tail");
    static constexpr int TMAX = K > 0 ? (ROW ? K : K * K) : NTAP_;
    static_assert(TMAX >= 1, "K == 0 needs a tap count");
    static constexpr int EXT = K > 0 ? K - 1 : 1;
    static constexpr int EXTY = (K > 0 && ROW) ? 0 : EXT;
    static constexpr int TH = WPX * NT;
    static constexpr int CO_TILE = WCO * MT * 32;
    static constexpr int IH = (TH - 1) * S + EXTY + 1;
    static constexpr int IW = 31 * S + EXT + 1;
    static constexpr int PLANE = IH * IW;                          // pixels of the staged tile
    static constexpr int XP = (2 * PLANE + 63) / 64 * 64;          // slots per part: [kgroup][pixel], padded to whole DMA pieces
    static constexpr int X_SLOTS = PARTS * XP;                     // [part][kgroup][pixel]
    static int w_slots(int ntaps) { return PARTS * ntaps * 2 * CO_TILE; }   // LDS image [part][tap][kgroup][cout], multiple of 64
    static constexpr int NIT = XP / 256 + (XP % 256 ? 1 : 0);      // DMA pieces per thread and part
    // epilogue: 32 x 36-float transposition patches (NPATCH per wave, so that two accumulator tiles are in flight
    // between the LDS write and read phases) + statistics; they live in the free stage buffer of the tile's last chunk
    static constexpr int W_BYTES_MAX = PARTS * TMAX * 2 * CO_TILE * 16;
    static constexpr int NPATCH = (NT % 2 == 0 && (8 * 32 * 36 + WPX * CO_TILE * 2) * 4 <= PARTS * XP * 16 + W_BYTES_MAX) ? 2 : 1;
    static constexpr int EPI_FLOATS = 4 * NPATCH * 32 * 36 + WPX * CO_TILE * 2;
    static_assert(WCO * WPX == 4, "4 waves per workgroup");
    static_assert(CO_TILE % 16 == 0, "weight image must be whole wave-wide LDS-DMA pieces");
    // the epilogue patches live in the (free) stage buffer of the tile's last chunk -- the LAST stage in memory, each
    // stage being one contiguous [weight image | activation image] block -- and, where a head-only stage is smaller
    // than the patches, in EPI_EXTRA bytes behind it
    static_assert(IH < 128 && IW < 256, "piece geometry is packed into 15 bits");
    // two workgroups share a CU when two stage pairs fit its 160 KB of LDS: the kernel is then compiled for <= 256 registers
    static constexpr int WG_PER_CU = (PARTS == 1 || 2 * (X_SLOTS + PARTS * TMAX * 2 * CO_TILE) * 16 + 1024 <= 80 * 1024) ? 2 : 1;
    static int wfloats(int ntaps) { return 2 * ntaps * 2 * CO_TILE * 4; }   // floats per packed (cout tile, chunk) weight block: always both parts
    static size_t lds_bytes(int ntaps) {                                                          // two stages (+ patches)
        const size_t stage = (size_t)(X_SLOTS + w_slots(ntaps)) * 16, epi = (size_t)EPI_FLOATS * 4;
        return 2 * stage + (epi > stage ? (epi - stage + 15) / 16 * 16 : 0);
    }
};

__device__ __forceinline__ void split_bf16(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

struct Bf3Tile {
    int n, cot, ty, tx;
};

// LDS-DMA issued behind the compiler's back.  With the builtin, the compiler books every in-flight
// global_load_lds as a "flat access that may touch LDS" and degrades each later s_waitcnt on an LDS read to
// lgkmcnt(0) for as long as the DMA is pending -- i.e. through the whole MFMA stage this kernel overlaps it with.
// Raw issue keeps the fragment-read waits exact; the price is that completion must be awaited explicitly
// (dma_wait_all) before the barrier that publishes the stage.  lds_addr: wave-uniform LDS byte address of
// lane 0's 16 bytes (lane i lands at +16 i).
// Source address = scalar base + per-lane unsigned 32-bit byte offset.
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory", "m0");
}
// The same with the LDS address as scalar base + compile-time offset: one s_add into M0 per piece.  (Passing each piece's
// LDS address as a value kept 19 scalars alive per stage; with ~100 SGPRs in use the compiler spilled them into VGPR lanes
// and read them back with v_readlane in front of every piece: profiles/r05_dominant_cycle_account.md.)
template <int IMM>
__device__ __forceinline__ void glds16_si(const void* sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base), "n"(IMM)
                 : "memory", "m0", "scc");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <class F, int... J>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, J...>) {
    (f(std::integral_constant<int, J>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// seg[s].data of the bf16x3 kernel points to an XS tensor (see split_prepass_kernel); seg[s].C = channels.
// Launch with min(#tiles, #CUs) workgroups of 256 threads; needs nchunks >= 2.
// (head-only arithmetic halves the LDS image: two workgroups per CU, i.e. two waves per SIMD and <= 256 registers each)
template <class C>
__global__ __launch_bounds__(256, C::WG_PER_CU) void conv_bf16x3(const ConvKParams p) {
    constexpr int S = C::S, K = C::K, TMAX = C::TMAX, MT = C::MT, NT = C::NT, WCO = C::WCO;
    constexpr int IW = C::IW, PLANE = C::PLANE, NIT = C::NIT, CO_TILE = C::CO_TILE, XP = C::XP;
    constexpr int IWE = (IW + 1) / 2;      // even columns of a row of the LDS image (stride-2 layout, see pgeo)
    constexpr bool XPF = true;             // fragments of the next stage's tap 0 are fetched during this stage's last tap
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* const smem = reinterpret_cast<uint4*>(smem_raw);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA destinations stay in SGPRs
    const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wco = wave % WCO, wpx = wave / WCO;
    const int H = p.H, W = p.W, HW = H * W;
    const int nchunks = p.nchunks;         // even: the host pads an odd count with one all-zero weight chunk
    const int nreal = p.cin_pad >> 4;      // chunks that exist in the sources

    constexpr int T = TMAX;
    constexpr int PARTS = C::PARTS;
    constexpr int W_SLOTS = PARTS * T * 2 * CO_TILE;
    constexpr int STAGE = W_SLOTS + C::X_SLOTS;                // stage b = smem + b * STAGE: [W_SLOTS weights | X_SLOTS activations]
    uint4* const wbuf = smem;                                  // + b * STAGE
    uint4* const xbuf = smem + W_SLOTS;                        // + b * STAGE
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

#ifdef APAMD_ABLATION
    // ---- cycle account (experiment build; tools/cycle_account.py): lane 0 of every wave records (code, s_memtime) pairs in LDS.
    // code = kind | chunk << 8 | tile ordinal << 16.  An s_memtime result is awaited with lgkmcnt(0) inside the same asm
    // statement (the compiler does not know the instruction is a scalar memory read).
    const bool stamping = p.stamps != nullptr;
    unsigned* const stamp_lds = reinterpret_cast<unsigned*>(smem_raw + p.stamp_lds_off) + wave * p.stamp_words;
    int stamp_n = 0, stamp_tile = 0;
    auto stamp_put = [&](unsigned kind, int c, unsigned long long t) __attribute__((always_inline)) {
        if (lane == 0 && stamp_n + 2 <= p.stamp_words) {
            stamp_lds[stamp_n] = kind | ((unsigned)c << 8) | ((unsigned)stamp_tile << 16);
            stamp_lds[stamp_n + 1] = (unsigned)t;
        }
        stamp_n += 2;
    };
    auto stamp_now = [&](unsigned kind, int c) __attribute__((always_inline)) {
        if (!stamping) return;
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        stamp_put(kind, c, t);
    };
    // the stage's one synchronisation point, stamped: arrival (0), DMA of the next stage landed (1), barrier passed (2)
    auto stage_sync = [&](int c) __attribute__((always_inline)) {
        if (!stamping) {
            dma_wait_all();
            if (!AP_ABLATE(p, 2)) __syncthreads();
            return;
        }
        unsigned long long ta, tb, tc;
        asm volatile("s_memtime %0\n\ts_waitcnt vmcnt(0)\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"
                     "s_memtime %2\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(ta), "=&s"(tb), "=&s"(tc)::"memory");
        stamp_put(0, c, ta);
        stamp_put(1, c, tb);
        stamp_put(2, c, tc);
    };
#define AP_STAMP(kind, c) stamp_now(kind, c)
#else
    auto stage_sync = [&](int) __attribute__((always_inline)) {
        dma_wait_all();
        __syncthreads();
    };
#define AP_STAMP(kind, c) ((void)0)
#endif

    // ---- this workgroup's tiles.  Workgroup b runs on XCD b % 8; each XCD owns a contiguous range of the tile
    // list (cout tile fastest, so the workgroups of an XCD share activation tiles through its L2) and its
    // workgroups walk that range in lock step.
    int tile, tile_end, tile_step;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int nx = G < 8 ? G : 8;                              // XCDs that received workgroups
        const int xcd = b % nx, idx = b / nx;
        const int ntl = p.N * p.tiles_y * p.tiles_x * p.co_tiles;
        const int q = ntl / nx, r = ntl % nx;
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile_step = (G - xcd + nx - 1) / nx;                       // workgroups on this XCD
        tile = base + idx;
        tile_end = base + q + (xcd < r ? 1 : 0);
    }
    if (tile >= tile_end) return;

    auto seg_of = [&](int chunk) {
        int s = 0;
        if (p.nseg > 1 && chunk >= p.seg[1].chunk_begin) s = 1;
        if (p.nseg > 2 && chunk >= p.seg[2].chunk_begin) s = 2;
        return s;
    };

    // ---- DMA geometry: piece k of this thread covers slot it = tid + k*256 of a part = (k-group, tile pixel);
    // source = that pixel of the image (reflected) or the all-zero slot (zero padding, tile padding)
    int pgeo[NIT];      // (k-group << 15) | (ly << 8) | lx of the piece, or -1 beyond the tile
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = tid + k * 256;
        const int kg = it >= PLANE ? 1 : 0;
        const int pix = it - kg * PLANE;
        const int ly = pix / IW;
        int lx = pix - ly * IW;
        // stride 2: a row of the LDS image holds the even image columns first, then the odd ones, so that the 32
        // pixels of a fragment (image columns 2 l + kx) are CONSECUTIVE slots -- stride-2 slots are a 2-way bank
        // conflict on every ds_read_b128 (8.2 M conflict cycles per launch, profiles/r01y_pmc_mfma.md)
        if constexpr (S == 2) lx = lx < IWE ? 2 * lx : 2 * (lx - IWE) + 1;
        pgeo[k] = it < 2 * PLANE ? ((kg << 15) | (ly << 8) | lx) : -1;
    }
    auto locate = [&](int logical, Bf3Tile& t, int (&goff)[NIT]) __attribute__((always_inline)) {
        t.cot = logical % p.co_tiles;
        int t_ = logical / p.co_tiles;
        t.tx = t_ % p.tiles_x;
        t_ /= p.tiles_x;
        t.ty = t_ % p.tiles_y;
        t.n = t_ / p.tiles_y;
        int dy0 = p.dy0, dx0 = p.dx0;
        if (p.nphase > 1) {                                    // fused sub-pixel phases: the phase sets the window origin
            const int ph = t.cot / p.co_tiles_phase;
            dy0 = p.ph_dy0[ph];
            dx0 = p.ph_dx0[ph];
        }
        const int iy0 = t.ty * C::TH * S + dy0, ix0 = t.tx * 32 * S + dx0;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            int gy = iy0 + ((pgeo[k] >> 8) & 127), gx = ix0 + (pgeo[k] & 255);
            bool ok = pgeo[k] >= 0;
            if (p.pad_mode == 1) {
                gy = reflect_clamp(gy, H);
                gx = reflect_clamp(gx, W);
            } else {
                ok = ok && gy >= 0 && gy < H && gx >= 0 && gx < W;
            }
            // byte offset from the chunk's first channel-group plane; out-of-image pixels read that plane's zero slot
            goff[k] = (ok ? ((pgeo[k] >> 15) & 1) * (HW + 1) + gy * W + gx : HW) * 16;
        }
    };
    // ---- LDS-DMA of one stage = NPIECE wave-wide 1 KiB pieces per wave: 2 * NIT activation pieces (head, tail) and
    // the wave's share of the weight image.  DmaCtx holds the scalar part of the addresses.
    // A piece is {s_mov m0; global_load_lds v_off, s[base]}: the per-lane 32-bit byte offsets are constants of the
    // tile (goff) or of the kernel (woff), everything that changes per stage is scalar -- so a piece costs no
    // vector ALU work and slots between two MFMAs of the stage's last tap.
    struct DmaCtx {
        const unsigned char* xh;   // the chunk's first head plane of image n (scalar)
        const unsigned char* xl;   // ... and tail plane
        const unsigned char* wsrc; // the (cout tile, chunk) weight block (scalar)
        unsigned xdst, wdst;       // LDS byte addresses of the stage's activation / weight images (this wave's lane 0)
        unsigned wmask;            // S2D3: taps of the chunk's input phase -- the weight pieces of the others are not staged (the
                                   // stage never reads them).  With the fragment reads of absent taps gone the kernel is bound by
                                   // the LDS-DMA (1.15 MB per CU against 27 k MFMA cycles, profiles/r04p_pmc_*): 19 % fewer bytes
    };
    constexpr int NWP = (W_SLOTS / 64 + 3) / 4, NPIECE = PARTS * NIT + NWP;
    unsigned woff[NWP];            // byte offset of this lane's slot in weight piece j
#pragma unroll
    for (int j = 0; j < NWP; ++j) woff[j] = ((j * 4 + wave) * 64 + lane) * 16;
    // The scalar part of a stage's addresses, computed ONCE per stage and pinned in SGPRs (the empty asm): left to itself the
    // compiler re-derived the 64-bit plane address in front of every piece (ten scalar multiplies / adds per piece).
    // The padding chunk of an odd count (chunk_ >= nreal: its weights are all zero) stages the last real chunk's activations
    // again -- finite data, so the products are exact zeros -- instead of skipping its pieces under a per-piece branch.
    auto dma_setup = [&](int n, int cot, int chunk_, int buf) __attribute__((always_inline)) {
        const int chunk = chunk_ < nreal ? chunk_ : nreal - 1;
        const int s = seg_of(chunk);
        const int cg0 = (chunk - p.seg[s].chunk_begin) * 2;          // first channel group of the chunk
        const int CG = p.seg[s].C >> 3;
        const unsigned char* xs = reinterpret_cast<const unsigned char*>(p.seg[s].data);
        const unsigned char* xh = xs + ((long long)(n * 2 + 0) * CG + cg0) * (HW + 1) * 16;
        const unsigned char* xl = xs + ((long long)(n * 2 + 1) * CG + cg0) * (HW + 1) * 16;
        unsigned xdst = lds0 + (buf * STAGE + W_SLOTS + wave * 64) * 16;
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp + ((long long)cot * nchunks + chunk_) * p.wfloats);
        unsigned wdst = lds0 + (buf * STAGE + wave * 64) * 16;
        unsigned wmask = 0xFu;
        if constexpr (C::S2D3) wmask = p.s2d_mask[chunk_ / p.s2d_div < 3 ? chunk_ / p.s2d_div : 3];
        asm volatile("" : "+s"(xh), "+s"(xl), "+s"(wsrc), "+s"(xdst), "+s"(wdst));
        return DmaCtx{xh, xl, wsrc, xdst, wdst, wmask};
    };
    // piece j (a compile-time index): {s_add m0; global_load_lds v_off, s[base]} -- nothing else
    // (alt / use_alt: the tile's tail stage takes the NEXT tile's offsets; selected here, per piece, on registers -- a
    // selected copy of the array was kept in scratch memory)
    auto dma_piece = [&](const DmaCtx& d, const int (&goff)[NIT], const int (&alt)[NIT], bool use_alt, auto jt) __attribute__((always_inline)) {
        constexpr int j = decltype(jt)::value;
        if constexpr (j < PARTS * NIT) {
            constexpr int part = j / NIT, k = j % NIT;
            constexpr bool whole = k * 256 + 3 * 64 < XP;           // every wave's slice of the piece lies inside the part
            if (whole || k * 256 + wave * 64 < XP)                  // (wave-uniform)
                glds16_si<(part * XP + k * 256) * 16>(part ? d.xl : d.xh, (unsigned)(use_alt ? alt[k] : goff[k]), d.xdst);
        } else {
            constexpr int jj = j - PARTS * NIT;
            constexpr bool whole = jj * 4 + 3 < W_SLOTS / 64;
            bool want = whole || jj * 4 + wave < W_SLOTS / 64;
            // weight image [part][tap][k-group][CO_TILE] in 64-slot pieces: piece -> tap (wave-uniform)
            if constexpr (C::S2D3) want = want && ((d.wmask >> (((jj * 4 + wave) % (T * 2 * CO_TILE / 64)) / (2 * CO_TILE / 64))) & 1u);
            if (want) glds16_si<jj * 4096>(d.wsrc, woff[jj], d.wdst);
        }
    };
    auto issue = [&](const Bf3Tile& t, const int (&goff)[NIT], int chunk_, int buf) __attribute__((always_inline)) {
        const DmaCtx d = dma_setup(t.n, t.cot, chunk_, buf);
        static_for<NPIECE>([&](auto jt) __attribute__((always_inline)) { dma_piece(d, goff, goff, false, jt); });
    };

    // fragment addresses (16-byte slots)
    const int a_slot = half * CO_TILE + wco * MT * 32 + l32;                       // + ((part*T + t)*2) * CO_TILE + m*32
    const int b_slot = half * PLANE + (wpx * NT) * S * IW + (S == 2 ? l32 : l32 * S);   // + part*XP + toff + q*S*IW

    f32x16 acc[MT][NT];
    bf16x8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
    // Dense stride-1 taps share activation fragments vertically: the fragment of pixel row q at tap (ky+1, kx) is the
    // one of row q+1 at tap (ky, kx).  With the taps walked column by column (kx outer, ky inner) a wave keeps a sliding
    // window of NT + K - 1 row fragments per kx and reads ONE new row per tap (NT at a column change) instead of NT.
    // Used for the 4x4 layers (MT = 1: 22 instead of 40 ds_read_b128 per tap column, 196 -> 170..185 us on the
    // PatchGAN's 256 -> 512 layer).  The split 3x3 kernel (MT = 2: 72 instead of 108 per chunk) is no faster per launch with it
    // -- its fragment reads already hide under the MFMAs -- but a third fewer LDS reads is power the chip gives back as clock:
    // 2257 / 2258 / 2259 -> 2272 / 2275 / 2263 frames/s, same box, interleaved (round 5, gpurun_out/r05n; the kernel runs at the
    // board's power limit, profiles/r05e_power_gen.md), so it takes the window as well.
    // ... with ONE product per tap (plain bf16, PARTS = 1) it is the other way round: 6 fragment reads per 8 MFMAs and 8 waves on
    // a CU's LDS port keep that port ~75 % busy, so the head-only 3x3 kernel takes the window too (36 instead of 54 reads per chunk):
    // 132.8-134.8 -> 130.3 us per launch in the plain-bf16 train step (gpurun_out/r04ah_train_bf16_kernel_stats.md), i.e. the
    // port was not the bound either; kept because it is not slower.
    constexpr bool WIN = (K == 4 || (K == 3 && NT > 1)) && !C::ROW && S == 1;
    constexpr int WR = WIN ? NT + K - 1 : 1;
    bf16x8 wh[2][WR], wl[2][WR];                                   // [kx parity][window row]
    auto fetch_a = [&](int stage_buf, int t, int buf) __attribute__((always_inline)) {
        const uint4* Wc = wbuf + stage_buf * STAGE + a_slot;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((0 * T + t) * 2) * CO_TILE + m * 32);
            if constexpr (PARTS == 2) al[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((1 * T + t) * 2) * CO_TILE + m * 32);
        }
    };
    auto fetch_brow = [&](int stage_buf, int kx, int j, int wb) __attribute__((always_inline)) {
        const uint4* Xc = xbuf + stage_buf * STAGE + b_slot + j * IW + kx;
        wh[wb][j] = *reinterpret_cast<const bf16x8*>(Xc);
        if constexpr (PARTS == 2) wl[wb][j] = *reinterpret_cast<const bf16x8*>(Xc + XP);
    };
    auto fetch = [&](int stage_buf, int t, int buf) __attribute__((always_inline)) {
        const uint4* Wc = wbuf + stage_buf * STAGE + a_slot;
        const uint4* Xc = xbuf + stage_buf * STAGE + b_slot;
        int toff;
        if constexpr (K > 0) toff = C::ROW ? t : (t / K) * IW + (S == 2 ? ((t % K) & 1) * IWE + ((t % K) >> 1) : (t % K));
        else toff = (int)((p.tap_bits >> (2 * t)) & 1u) * IW + (int)((p.tap_bits >> (2 * t + 1)) & 1u);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((0 * T + t) * 2) * CO_TILE + m * 32);
            if constexpr (PARTS == 2) al[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((1 * T + t) * 2) * CO_TILE + m * 32);
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            bh[buf][q] = *reinterpret_cast<const bf16x8*>(Xc + toff + q * S * IW);
            if constexpr (PARTS == 2) bl[buf][q] = *reinterpret_cast<const bf16x8*>(Xc + XP + toff + q * S * IW);
        }
    };

    // fragment r of tap t (r < 2 MT: weights head / tail alternating; then activations)
    auto fetch_one = [&](int stage_buf, int t, int buf, int r) __attribute__((always_inline)) {
        const uint4* Wc = wbuf + stage_buf * STAGE + a_slot;
        const uint4* Xc = xbuf + stage_buf * STAGE + b_slot;
        int toff;
        if constexpr (K > 0) toff = C::ROW ? t : (t / K) * IW + (S == 2 ? ((t % K) & 1) * IWE + ((t % K) >> 1) : (t % K));
        else toff = (int)((p.tap_bits >> (2 * t)) & 1u) * IW + (int)((p.tap_bits >> (2 * t + 1)) & 1u);
        if (r < PARTS * MT) {
            const int m = r / PARTS;
            if (PARTS == 2 && (r & 1)) al[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((1 * T + t) * 2) * CO_TILE + m * 32);
            else ah[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((0 * T + t) * 2) * CO_TILE + m * 32);
        } else {
            const int q = (r - PARTS * MT) / PARTS;
            if (PARTS == 2 && (r & 1)) bl[buf][q] = *reinterpret_cast<const bf16x8*>(Xc + XP + toff + q * S * IW);
            else bh[buf][q] = *reinterpret_cast<const bf16x8*>(Xc + toff + q * S * IW);
        }
    };
    // the same for the windowed form: fragment r of tap (0, 0) -- weights into buffer `buf`, window rows into `wb`
    auto fetch_one_win = [&](int stage_buf, int buf, int wb, int r) __attribute__((always_inline)) {
        const uint4* Wc = wbuf + stage_buf * STAGE + a_slot;
        const uint4* Xc = xbuf + stage_buf * STAGE + b_slot;
        if (r < PARTS * MT) {
            const int m = r / PARTS;
            if (PARTS == 2 && (r & 1)) al[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((1 * T + 0) * 2) * CO_TILE + m * 32);
            else ah[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((0 * T + 0) * 2) * CO_TILE + m * 32);
        } else {
            const int q = (r - PARTS * MT) / PARTS;
            if (PARTS == 2 && (r & 1)) wl[wb][q] = *reinterpret_cast<const bf16x8*>(Xc + XP + q * IW);
            else wh[wb][q] = *reinterpret_cast<const bf16x8*>(Xc + q * IW);
        }
    };

    Bf3Tile cur, nxt;
    int cgoff[NIT], ngoff[NIT];
    locate(tile, cur, cgoff);
    AP_STAMP(7, 0);                                                // kernel entry (tile geometry done)
    issue(cur, cgoff, 0, 0);
    issue(cur, cgoff, 1, 1);
    dma_wait_all();
    __syncthreads();
    AP_STAMP(8, 0);                                                // first two stages landed

    // taps of the 2 x 2 window that exist for the current tile (fused sub-pixel phases; wave-uniform)
    auto tapmask_of = [&](const Bf3Tile& t) -> unsigned {
        if (K != 0 || p.nphase <= 1) return 0xFu;
        return (unsigned)__builtin_amdgcn_readfirstlane((int)p.ph_tapmask[t.cot / p.co_tiles_phase]);
    };
    unsigned tmask = tapmask_of(cur);
    for (;;) {
        const bool has_next = tile + tile_step < tile_end;
        locate(has_next ? tile + tile_step : tile, nxt, ngoff);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

        // one pipeline stage = one 16-channel chunk; P = its LDS stage buffer (compile-time in each copy).
        // MASK != 0 (space-to-depth 3x3 layers, K == 0 with 4 taps): the stage's taps are known at COMPILE time -- the set bits of
        // MASK -- so an absent tap costs neither MFMAs nor fragment reads.  With the run-time mask only the MFMAs were skipped:
        // a stage then issued 32 ds_read_b128 per wave for 27 MFMAs on average (4 / 2 / 2 / 1 taps of the four input phases),
        // and the LDS read port, not the matrix pipe, bounded the kernel (8 waves per CU, profiles/r03_ablate_layers.md).
        auto stage = [&](auto ptag, auto mtag, int c) __attribute__((always_inline)) {
            constexpr int P = decltype(ptag)::value;
            constexpr unsigned MASK = decltype(mtag)::value;
            static_assert(MASK == 0 || (K == 0 && (MASK & 1u)), "compile-time tap sets: run-time-tap family, tap 0 always present");
            constexpr int NREAL = MASK ? __builtin_popcount(MASK) : TMAX;
            auto tap_of = [](int i) constexpr {                   // i-th tap of the stage
                if (!MASK) return i;
                int n = 0;
                for (int b = 0; b < TMAX; ++b)
                    if ((MASK >> b) & 1u) {
                        if (n == i) return b;
                        ++n;
                    }
                return 0;
            };
            // taps of this stage: of the tile's phase, or (space-to-depth 3x3) of the chunk's input phase
            const unsigned tm = MASK ? MASK : ((K == 0 && p.s2d_div > 0) ? p.s2d_mask[c / p.s2d_div < 3 ? c / p.s2d_div : 3] : tmask);
            constexpr int FP = (XPF && (NREAL & 1)) ? P : 0;       // register buffer of tap 0
            constexpr int WP = (XPF && (K & 1)) ? P : 0;           // window buffer of kx = 0
            if (!XPF || c == 0) {
                if constexpr (WIN) {
                    fetch_a(P, 0, FP);
#pragma unroll
                    for (int q = 0; q < NT; ++q) fetch_brow(P, 0, q, WP & 1);
                } else {
                    fetch(P, 0, FP);
                }
            }
#pragma unroll
            for (int ti = 0; ti < NREAL; ++ti) {
                const int tp = tap_of(ti);
                // windowed form: taps column by column; t = the tap's index in the weight image
                constexpr int KD = K > 0 ? K : 1;                    // (K == 0 instantiations never take the windowed form)
                const int kx = WIN ? tp / KD : 0, ky = WIN ? tp % KD : 0;
                const int t = WIN ? ky * K + kx : tp;
                const int cb = (ti + FP) & 1, wb = (kx + WP) & 1;
                const bool last = ti == NREAL - 1;
                auto mfma_one = [&](int i) __attribute__((always_inline)) {
                    // the three partial products go round all MT*NT accumulators in turn, so consecutive MFMAs
                    // never wait on each other's result (small terms first)
                    const int g = PARTS == 2 ? i / (MT * NT) : 2, m = (i % (MT * NT)) / NT, q = i % NT;
                    const bf16x8 xh = WIN ? wh[wb][WIN ? ky + q : 0] : bh[cb][q];
                    const bf16x8 xl = WIN ? wl[wb][WIN ? ky + q : 0] : bl[cb][q];
                    if (g == 0) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb][m], xh, acc[m][q], 0, 0, 0);
                    else if (g == 1) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][m], xl, acc[m][q], 0, 0, 0);
                    else acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][m], xh, acc[m][q], 0, 0, 0);
                };
                constexpr int NM = (PARTS == 2 ? 3 : 1) * MT * NT, NRD = PARTS * (MT + NT);
                if (!last) {
                    int nrd = NRD;                                  // LDS reads issued for the next tap
                    if constexpr (WIN) {
                        const int kx2 = (tp + 1) / KD, ky2 = (tp + 1) % KD;
                        fetch_a(P, ky2 * K + kx2, cb ^ 1);
                        if (kx2 == kx) {
                            fetch_brow(P, kx, ky2 + NT - 1, wb);
                            nrd = PARTS * (MT + 1);
                        } else {
#pragma unroll
                            for (int q = 0; q < NT; ++q) fetch_brow(P, kx2, q, wb ^ 1);
                        }
                    } else {
                        fetch(P, MASK ? tap_of(ti + 1) : t + 1, cb ^ 1);
                    }
                    if (K != 0 || ((tm >> tp) & 1u)) {       // (an absent tap of a fused phase: zero weights, nothing to add)
#pragma unroll
                    for (int i = 0; i < NM; ++i) mfma_one(i);
                    }
                    // pin the schedule: the next tap's fragment reads are spread evenly between this tap's MFMAs
                    // (left alone, the scheduler sinks every read to just before its first use and stalls on it)
                    constexpr int PER = (NM + NRD - 1) / NRD, NRD1 = PARTS * (MT + 1), PER1 = (NM + NRD1 - 1) / NRD1;
                    if (nrd == NRD) {
#pragma unroll
                        for (int i = 0; i < NRD; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
                            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);    // PER MFMAs
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NRD1; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, PER1, 0);
                        }
                    }
                } else {
                    // every fragment of this stage is in registers: the stage buffer can be refilled, and the
                    // next stage (issued one stage ago) has landed once everybody is past the barrier.  The DMA
                    // pieces of stage c+2 (past the tile's end: chunk 0 of the next tile) and the fragment reads of
                    // the next stage's tap 0 are issued one per MFMA slot, so the matrix pipe keeps running.
                    // The last stage of a tile issues nothing (its buffer becomes the epilogue's patch area); the last but one
                    // stages chunk 0 of the NEXT tile -- without a next tile this tile's own chunk 0 again, which nobody reads:
                    // an unconditional issue keeps the pieces free of branches.  What a piece may cost is decided here:
                    // the scalar addresses are pinned in SGPRs by dma_setup, the per-lane offset is one select,
                    // and the `more` test in front of a piece is a bare scalar branch.  (Before, the compiler put the branch, the
                    // offset select and the re-derivation of the 64-bit source address -- ten scalar multiplies / adds and a
                    // v_readlane of a spilled LDS address -- in front of EVERY piece: 74 cycles per MFMA slot of this tap
                    // instead of 32, ~1000 cycles per stage, profiles/r05_dominant_cycle_account.md.)
                    const bool more = c + 1 < nchunks;
                    const bool tail = c + 2 >= nchunks;
                    const DmaCtx d = dma_setup(tail ? nxt.n : cur.n, tail ? nxt.cot : cur.cot, tail ? 0 : c + 2, P);
                    // (the offset selects sit HERE, in front of the barrier: vector selects on SGPR-pair masks between LDS-DMA
                    // pieces and MFMAs are what makes a kernel disturb a co-resident one, profiles/r04_cohazard.md)
                    int ig[NIT];
#pragma unroll
                    for (int k = 0; k < NIT; ++k) {
                        ig[k] = tail ? ngoff[k] : cgoff[k];
                        asm volatile("" : "+v"(ig[k]));                 // (pinned here: the compiler sinks the select to its use)
                    }
                    stage_sync(c);
                    constexpr int PPS = (NPIECE + NM - 1) / NM;                 // DMA pieces per MFMA slot
                    constexpr int RPS = (NRD + NM - 1) / NM;                    // fragment reads per MFMA slot
                    constexpr int R0 = NM - (NRD + RPS - 1) / RPS;              // first slot that carries reads
                    const bool real_tap = K != 0 || ((tm >> tp) & 1u);
                    // (the MFMAs stay in straight-line code: with them inside the two arms of a branch the accumulators became phi
                    // nodes and the allocator moved all 128 between AGPRs and VGPRs every stage)
                    // (as scalar integers: left as `bool`s the compiler parks the condition in a VGPR and tests it from there)
                    const int dma = __builtin_amdgcn_readfirstlane((more && !AP_ABLATE(p, 1)) ? 1 : 0);
                    static_for<NM>([&](auto it) __attribute__((always_inline)) {
                        constexpr int i = decltype(it)::value;
                        if (real_tap) mfma_one(i);
                        static_for<PPS>([&](auto ppt) __attribute__((always_inline)) {
                            constexpr int j = i * PPS + decltype(ppt)::value;
                            if constexpr (j < NPIECE) {
                                if (dma) dma_piece(d, ig, ig, false, std::integral_constant<int, j>{});
                            }
                        });
                        if constexpr (XPF && i >= R0) {
                            if (more) {
#pragma unroll
                                for (int rr = 0; rr < RPS; ++rr)
                                    if ((i - R0) * RPS + rr < NRD) {
                                        if constexpr (WIN) fetch_one_win(P ^ 1, cb ^ 1, ((K & 1) ? (P ^ 1) : 0) & 1, (i - R0) * RPS + rr);
                                        else fetch_one(P ^ 1, 0, cb ^ 1, (i - R0) * RPS + rr);
                                    }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
            }
        };
        // nchunks is even: chunk pairs run straight-line through stage buffers 0 and 1 (a run-time choice of the
        // buffer per stage makes the register allocator shuttle all accumulators between VGPRs and AGPRs)
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        if constexpr (C::S2D3) {
            // chunks [r * s2d_div, (r + 1) * s2d_div) belong to input phase r = (ry, rx), whose taps inside the 2 x 2 window are
            // all four / the left column / the top row / the corner: one straight-line chunk loop per phase (s2d_div is even)
            const int cpp = p.s2d_div;
            int c = 0;
            for (; c < cpp; c += 2) {
                stage(P0{}, std::integral_constant<unsigned, 0xFu>{}, c);
                stage(P1{}, std::integral_constant<unsigned, 0xFu>{}, c + 1);
            }
            for (; c < 2 * cpp; c += 2) {
                stage(P0{}, std::integral_constant<unsigned, 0x5u>{}, c);
                stage(P1{}, std::integral_constant<unsigned, 0x5u>{}, c + 1);
            }
            for (; c < 3 * cpp; c += 2) {
                stage(P0{}, std::integral_constant<unsigned, 0x3u>{}, c);
                stage(P1{}, std::integral_constant<unsigned, 0x3u>{}, c + 1);
            }
            for (; c < nchunks; c += 2) {
                stage(P0{}, std::integral_constant<unsigned, 0x1u>{}, c);
                stage(P1{}, std::integral_constant<unsigned, 0x1u>{}, c + 1);
            }
        } else {
            for (int c = 0; c < nchunks; c += 2) {
                stage(P0{}, std::integral_constant<unsigned, 0u>{}, c);
                stage(P1{}, std::integral_constant<unsigned, 0u>{}, c + 1);
            }
        }
        AP_STAMP(3, 0);                                            // last MFMA issued
        constexpr int pl = 1;                                      // stage buffer of the last chunk: free now; the
                                                                   // next tile's chunk 0 sits in buffer 0 again

        // ACT >= 0: activation known at compile time (a run-time switch per element costs scalar branches)
        auto epilogue = [&](auto atag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(atag)::value;
            auto actf = [&](float v) { return ACT >= 0 ? apply_act(v, ACT) : apply_act(v, p.act); };
            // ---- epilogue.  MFMA C/D layout: column j = lane & 31 (pixel), row i = (r & 3) + 8 * (r >> 2) + 4 * half
            // (cout).  Each 32 x 32 tile is transposed through a private LDS patch so that a lane owns 4 consecutive
            // pixels of one cout row: 16-byte global stores (the dword-per-lane form is store-issue bound) and a
            // 3-step row reduction for the InstanceNorm statistics instead of a 5-step one per accumulator register.
            constexpr int TS = 36;                                            // patch row stride (floats, 16-B aligned)
            float* const epi = reinterpret_cast<float*>(smem + pl * STAGE);
            constexpr int NP = C::NPATCH;
            float* const patch0 = epi + wave * (NP * 32 * TS);
            float* const sred = epi + 4 * NP * 32 * TS;                        // [WPX][CO_TILE][2]
            const int n = cur.n;
            int cot = cur.cot, oy_off = p.oy_off, ox_off = p.ox_off, stat_off = p.stat_tile_off;
            if (p.nphase > 1) {                                // fused sub-pixel phases: (phase, cout tile of the phase)
                const int ph = cot / p.co_tiles_phase;
                cot -= ph * p.co_tiles_phase;
                oy_off = p.ph_oy[ph];
                ox_off = p.ph_ox[ph];
                stat_off = p.ph_stat[ph];
            }
            const int oy0 = cur.ty * C::TH, ox0 = cur.tx * 32;
            const int co_base = cot * CO_TILE + wco * MT * 32;
            const bool want_stats = p.stats != nullptr;
            const bool vec_ok = (p.o_rstride & 3) == 0 && p.osx == 1 && ox_off == 0;
            const bool full = vec_ok && ox0 + 32 <= p.OW;                      // whole 32-pixel rows: no per-element edge tests
            const int prow = lane >> 3, pcol = (lane & 7) * 4;
            const int oxv = ox0 + pcol;
            bool octet = false;
            if constexpr (C::K == 0 || C::ROW || (C::K == 3 && C::S == 1 && C::PARTS == 2 && !C::OB16 && !C::FNORM)) {
                // ---- channel-octet output y[n][Cout/8][OH*OW][8] (ap_conv2d_fwd_octet; the run-time-tap and row families
                // only: the producers of the warp kernel's input).  The MFMA layout already holds 4 consecutive couts of
                // one pixel per lane and register quad (row = (r & 3) + 8 (r >> 2) + 4 half), and the two halves of a wave
                // hold the two halves of an octet: 16-byte stores straight from the accumulators, a wave covers 32 pixels
                // x 32 bytes = 1 KiB contiguous; no LDS patch.  Row sums by a 5-step butterfly per accumulator register.
                octet = p.o_octet != 0;
                if (octet) {
                    const int ox = ox0 + l32;
                    const long long ohw = (long long)p.OH * p.OW;
                    // (round 6) every cout, row and column of the tile inside the tensor, no bias, no activation -- tested once, wave-uniform:
                    // the stores and sums below then run without the per-lane test (see fast32)
                    const bool allv = !(C::WG_PER_CU == 2 && (C::K == 0 || (C::ROW && C::MT == 2))) &&
                                      __builtin_amdgcn_readfirstlane((int)(p.bias == nullptr && ACT == 0 && co_base + MT * 32 <= p.Cout &&
                                                                           oy0 + C::TH <= p.OH && ox0 + 32 <= p.OW)) != 0;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        float s16[16], q16[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) { s16[r] = 0.f; q16[r] = 0.f; }
                        if (allv) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int co = co_base + m * 32 + g * 8 + 4 * half;
                                float* const ob = p.y + (long long)n * p.o_nstride + (long long)(co >> 3) * ohw * 8 + (co & 7) +
                                                  ((long long)(oy0 + wpx * NT) * p.OW + ox) * 8;
#pragma unroll
                                for (int q = 0; q < NT; ++q) {
                                    float vv[4];
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        vv[j] = acc[m][q][g * 4 + j];
                                        s16[g * 4 + j] += vv[j];
                                        q16[g * 4 + j] += vv[j] * vv[j];
                                    }
                                    *reinterpret_cast<float4*>(ob + (long long)q * p.OW * 8) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                                }
                            }
                        } else
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int co = co_base + m * 32 + g * 8 + 4 * half;          // 4 consecutive couts (Cout % 8 == 0)
                            const bool cok = co < p.Cout;
                            float bv[4] = {0.f, 0.f, 0.f, 0.f};
                            if (p.bias != nullptr && cok) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) bv[j] = p.bias[co + j];
                            }
                            float* const obase = p.y + (long long)n * p.o_nstride + (long long)(co >> 3) * ohw * 8 + (co & 7);
#pragma unroll
                            for (int q = 0; q < NT; ++q) {
                                const int oy = oy0 + wpx * NT + q;
                                if (cok && oy < p.OH && ox < p.OW) {
                                    float vv[4];
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        vv[j] = acc[m][q][g * 4 + j] + bv[j];
                                        s16[g * 4 + j] += vv[j];
                                        q16[g * 4 + j] += vv[j] * vv[j];
                                    }
                                    *reinterpret_cast<float4*>(obase + ((long long)oy * p.OW + ox) * 8) =
                                        make_float4(actf(vv[0]), actf(vv[1]), actf(vv[2]), actf(vv[3]));
                                }
                            }
                        }
                        if (want_stats) {
                            // 16 row sums over the 32 lanes of a half-wave as a transpose-reduce: at every step a lane
                            // hands half of its rows to the partner and keeps the sums of the other half (8 + 4 + 2 + 1
                            // exchanges, then one plain step) -- 16 shuffles per quantity instead of 16 x 5
                            auto fold = [&](float (&v)[16]) {
                                const bool b16 = l32 & 16, b8 = l32 & 8, b4 = l32 & 4, b2 = l32 & 2;
                                float w8[8], w4[4], w2[2];
#pragma unroll
                                for (int i = 0; i < 8; ++i)
                                    w8[i] = (b16 ? v[i + 8] : v[i]) + __shfl_xor(b16 ? v[i] : v[i + 8], 16, 64);
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    w4[i] = (b8 ? w8[i + 4] : w8[i]) + __shfl_xor(b8 ? w8[i] : w8[i + 4], 8, 64);
#pragma unroll
                                for (int i = 0; i < 2; ++i)
                                    w2[i] = (b4 ? w4[i + 2] : w4[i]) + __shfl_xor(b4 ? w4[i] : w4[i + 2], 4, 64);
                                float w1 = (b2 ? w2[1] : w2[0]) + __shfl_xor(b2 ? w2[0] : w2[1], 2, 64);
                                return w1 + __shfl_xor(w1, 1, 64);
                            };
                            const float s = fold(s16), q2 = fold(q16);
                            if ((l32 & 1) == 0) {
                                const int r = ((l32 >> 4) & 1) * 8 + ((l32 >> 3) & 1) * 4 + ((l32 >> 2) & 1) * 2 + ((l32 >> 1) & 1);
                                float* d = sred + ((wpx * CO_TILE) + wco * MT * 32 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 2;
                                d[0] = s;
                                d[1] = q2;
                            }
                        }
                    }
                }
            }
            bool wide16 = false;
            if constexpr (C::OB16) {
                // ---- bf16 output, whole 32-pixel rows at 16-byte aligned addresses (the forward outputs of the trunk): a lane takes
                // EIGHT consecutive pixels of a cout row from the patch (two ds_read_b128) and stores them as ONE 16-byte word --
                // 2 store instructions per 32 x 32 tile where the 4-pixel form below needs 4.  The epilogue is bound by the number
                // of store instructions a CU can retire (~115 cycles each whatever their width: MI355X_MICROARCH.md "store tail",
                // profiles/r05_dominant_cycle_account.md), so the 8-byte form was no faster than the fp32 output it replaced.
                // (every cout and row of the tile inside the tensor as well: the path below is straight-line code -- see fast32)
                wide16 = __builtin_amdgcn_readfirstlane((int)(p.osx == 1 && ox_off == 0 && ox0 + 32 <= p.OW && co_base + MT * 32 <= p.Cout &&
                         oy0 + C::TH <= p.OH && ((p.o_rstride | (int)p.o_cstride | (int)p.o_nstride) & 7) == 0 &&
                         (reinterpret_cast<uintptr_t>(p.y) & 15) == 0)) != 0;
                if (wide16) {
                    const int prow8 = lane >> 2, pcol8 = (lane & 3) * 8;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        float s2[2] = {0.f, 0.f}, q2s[2] = {0.f, 0.f};
                        float bv2[2];
                        long long cb2[2];
                        bool cok2[2];
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            const int co = co_base + m * 32 + ps * 16 + prow8;
                            cok2[ps] = co < p.Cout;
                            bv2[ps] = (p.bias != nullptr && cok2[ps]) ? p.bias[co] : 0.f;
                            cb2[ps] = (long long)n * p.o_nstride + (long long)co * p.o_cstride + ox0 + pcol8;
                        }
#pragma unroll
                        for (int q0 = 0; q0 < NT; q0 += NP) {
#pragma unroll
                            for (int b = 0; b < NP; ++b)
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    patch0[b * (32 * TS) + ((r & 3) + 8 * (r >> 2) + 4 * half) * TS + l32] = acc[m][q0 + b][r];
#pragma unroll
                            for (int b = 0; b < NP; ++b) {
                                const int oy = oy0 + wpx * NT + q0 + b;
                                const long long rowoff = (long long)(oy * p.osy + oy_off) * p.o_rstride;
#pragma unroll
                                for (int ps = 0; ps < 2; ++ps) {
                                    const float* src = patch0 + b * (32 * TS) + (ps * 16 + prow8) * TS + pcol8;
                                    const float4 va = *reinterpret_cast<const float4*>(src), vb = *reinterpret_cast<const float4*>(src + 4);
                                    const float bv = bv2[ps];
                                    const float vv[8] = {va.x + bv, va.y + bv, va.z + bv, va.w + bv, vb.x + bv, vb.y + bv, vb.z + bv, vb.w + bv};
                                    if (cok2[ps] && oy < p.OH) {   // (always true here.  Without the lane test the compiler hoists every patch read of the
                                                                   // round -- and this instantiation, 128 accumulators in VGPRs under a 256-register
                                                                   // cap for two workgroups per CU, spills 175 registers: measured, kept as is)
                                        s2[ps] += ((vv[0] + vv[1]) + (vv[2] + vv[3])) + ((vv[4] + vv[5]) + (vv[6] + vv[7]));
                                        q2s[ps] += ((vv[0] * vv[0] + vv[1] * vv[1]) + (vv[2] * vv[2] + vv[3] * vv[3])) +
                                                   ((vv[4] * vv[4] + vv[5] * vv[5]) + (vv[6] * vv[6] + vv[7] * vv[7]));
                                        uint4 pk;
                                        auto pack2 = [&](float a, float b2) {
                                            return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)actf(a)) |
                                                   ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)actf(b2)) << 16);
                                        };
                                        pk.x = pack2(vv[0], vv[1]); pk.y = pack2(vv[2], vv[3]); pk.z = pack2(vv[4], vv[5]); pk.w = pack2(vv[6], vv[7]);
                                        *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(p.y) + cb2[ps] + rowoff) = pk;
                                    }
                                }
                            }
                        }
                        if (want_stats) {
#pragma unroll
                            for (int ps = 0; ps < 2; ++ps) {
                                float s = s2[ps], q2 = q2s[ps];
#pragma unroll
                                for (int sh = 1; sh < 4; sh <<= 1) {
                                    s += __shfl_xor(s, sh, 64);
                                    q2 += __shfl_xor(q2, sh, 64);
                                }
                                if ((lane & 3) == 0) {
                                    float* d = sred + ((wpx * CO_TILE) + wco * MT * 32 + m * 32 + ps * 16 + prow8) * 2;
                                    d[0] = s;
                                    d[1] = q2;
                                }
                            }
                        }
                    }
                }
            }
            // ---- fp32 output, the common case as STRAIGHT-LINE code (round 6): whole 32-pixel rows, every cout and every row of the
            // tile inside the tensor, no bias -- wave-uniform, tested once.  The general loop below keeps a lane-divergent
            // `cout valid && row valid` test around every 16-byte group; inside it the compiler waits for each ds_read_b128 of the
            // transposition before it issues the next one: 32 serial LDS round trips + ~130 branches per wave and tile, 14.9 k of
            // the tile's ~150 k cycles WITH OR WITHOUT the global stores (profiles/r06_epilogue_account.md: APAMD_ABLATE=16 removes
            // the stores and the segment stays at 14.9 k) -- the epilogue was bound by its own control flow, not by store issue.
            bool fast32 = false;
            // (not where two workgroups share a CU's registers and the instantiation is already at its 256-register cap: the
            // run-time-tap family and the tall row tile of plain-bf16 arithmetic spilled 230-400 bytes with the extra path --
            // tests/test_isa_cpu.py)
            constexpr bool FAST_OK = !C::OB16 && !(C::WG_PER_CU == 2 && (C::K == 0 || (C::ROW && C::MT == 2)));
            if constexpr (FAST_OK) {
                fast32 = __builtin_amdgcn_readfirstlane((int)(!octet && full && p.bias == nullptr && ACT == 0 && p.osy == 1 &&
                                                              co_base + MT * 32 <= p.Cout && oy0 + C::TH <= p.OH)) != 0;
                if (fast32) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
                        float* dst0[4];
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps)
                            dst0[ps] = p.y + (long long)n * p.o_nstride + (long long)(co_base + m * 32 + ps * 8 + prow) * p.o_cstride + oxv +
                                       (long long)(oy0 + wpx * NT + oy_off) * p.o_rstride;
#pragma unroll
                        for (int q0 = 0; q0 < NT; q0 += NP) {
#pragma unroll
                            for (int b = 0; b < NP; ++b)
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    patch0[b * (32 * TS) + ((r & 3) + 8 * (r >> 2) + 4 * half) * TS + l32] = acc[m][q0 + b][r];
                            float4 v[NP][4];
#pragma unroll
                            for (int b = 0; b < NP; ++b)
#pragma unroll
                                for (int ps = 0; ps < 4; ++ps)
                                    v[b][ps] = *reinterpret_cast<const float4*>(patch0 + b * (32 * TS) + (ps * 8 + prow) * TS + pcol);
#pragma unroll
                            for (int b = 0; b < NP; ++b)
#pragma unroll
                                for (int ps = 0; ps < 4; ++ps) {
                                    const float4 t = v[b][ps];
                                    s4[ps] += (t.x + t.y) + (t.z + t.w);
                                    q4[ps] += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
                                    if (!AP_ABLATE(p, 16)) *reinterpret_cast<float4*>(dst0[ps] + (long long)(q0 + b) * p.o_rstride) = t;
                                }
                        }
                        if (want_stats) {
#pragma unroll
                            for (int ps = 0; ps < 4; ++ps) {
                                float s = s4[ps], q2 = q4[ps];
#pragma unroll
                                for (int sh = 1; sh < 8; sh <<= 1) {
                                    s += __shfl_xor(s, sh, 64);
                                    q2 += __shfl_xor(q2, sh, 64);
                                }
                                if ((lane & 7) == 0) {
                                    float* d = sred + ((wpx * CO_TILE) + wco * MT * 32 + m * 32 + ps * 8 + prow) * 2;
                                    d[0] = s;
                                    d[1] = q2;
                                }
                            }
                        }
                    }
                }
            }
            if (!octet && !wide16 && !fast32) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
                float bvv[4];
                long long cbase[4];
                bool cokc[4];
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int co = co_base + m * 32 + ps * 8 + prow;
                    cokc[ps] = co < p.Cout;
                    bvv[ps] = (p.bias != nullptr && cokc[ps]) ? p.bias[co] : 0.f;
                    cbase[ps] = (long long)n * p.o_nstride + (long long)co * p.o_cstride + oxv * p.osx + ox_off;
                }
#pragma unroll
                for (int q0 = 0; q0 < NT; q0 += NP) {
#pragma unroll
                    for (int b = 0; b < NP; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            patch0[b * (32 * TS) + ((r & 3) + 8 * (r >> 2) + 4 * half) * TS + l32] = acc[m][q0 + b][r];
#pragma unroll
                    for (int b = 0; b < NP; ++b) {
                        const int oy = oy0 + wpx * NT + q0 + b;
                        const long long rowoff = (long long)(oy * p.osy + oy_off) * p.o_rstride;
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const float4 v = *reinterpret_cast<const float4*>(patch0 + b * (32 * TS) + (ps * 8 + prow) * TS + pcol);
                            const float bv = bvv[ps];
                            const float vv[4] = {v.x + bv, v.y + bv, v.z + bv, v.w + bv};
                            if (cokc[ps] && oy < p.OH) {
                                float* dst = p.y + cbase[ps] + rowoff;
                                if constexpr (C::OB16) {
                                    // bf16 output: the same element offsets on a 2-byte element; 8-byte stores of 4 pixels (4-byte
                                    // alignment suffices: the padded rows of a data gradient start at odd pixel pairs)
                                    typedef unsigned u32x2 __attribute__((ext_vector_type(2), aligned(4)));
                                    __bf16* d16 = reinterpret_cast<__bf16*>(p.y) + cbase[ps] + rowoff;
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
                                        if (full || oxv + j < p.OW) { s4[ps] += vv[j]; q4[ps] += vv[j] * vv[j]; }
                                    // (an 8-byte store of four bf16 needs even element strides only: the 66-element rows of a padded
                                    // data gradient qualify, which the 16-byte fp32 form does not)
                                    const bool v16 = p.osx == 1 && ((p.o_rstride | (int)p.o_cstride | (int)p.o_nstride | ox_off) & 1) == 0;
                                    if (v16 && oxv + 3 < p.OW) {
                                        u32x2 pk;
                                        pk[0] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)actf(vv[0])) |
                                                ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)actf(vv[1])) << 16);
                                        pk[1] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)actf(vv[2])) |
                                                ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)actf(vv[3])) << 16);
                                        *reinterpret_cast<u32x2*>(d16) = pk;
                                    } else {
#pragma unroll
                                        for (int j = 0; j < 4; ++j)
                                            if (oxv + j < p.OW) d16[j * p.osx] = (__bf16)actf(vv[j]);
                                    }
                                } else if (full) {
                                    s4[ps] += (vv[0] + vv[1]) + (vv[2] + vv[3]);
                                    q4[ps] += (vv[0] * vv[0] + vv[1] * vv[1]) + (vv[2] * vv[2] + vv[3] * vv[3]);
                                    // (experiment build, APAMD_ABLATE bit 16: the epilogue without its global stores -- what is left is
                                    // the LDS transposition + statistics; profiles/r06_epilogue_account.md)
                                    if (!AP_ABLATE(p, 16))
                                    *reinterpret_cast<float4*>(dst) =
                                        make_float4(actf(vv[0]), actf(vv[1]), actf(vv[2]), actf(vv[3]));
                                } else {
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
                                        if (oxv + j < p.OW) { s4[ps] += vv[j]; q4[ps] += vv[j] * vv[j]; }
                                    if (vec_ok && oxv + 3 < p.OW) {
                                        *reinterpret_cast<float4*>(dst) =
                                            make_float4(actf(vv[0]), actf(vv[1]), actf(vv[2]), actf(vv[3]));
                                    } else {
#pragma unroll
                                        for (int j = 0; j < 4; ++j)
                                            if (oxv + j < p.OW) dst[j * p.osx] = actf(vv[j]);
                                    }
                                }
                            }
                        }
                    }
                }
                if (want_stats) {
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        float s = s4[ps], q2 = q4[ps];
#pragma unroll
                        for (int sh = 1; sh < 8; sh <<= 1) {
                            s += __shfl_xor(s, sh, 64);
                            q2 += __shfl_xor(q2, sh, 64);
                        }
                        if ((lane & 7) == 0) {
                            float* d = sred + ((wpx * CO_TILE) + wco * MT * 32 + m * 32 + ps * 8 + prow) * 2;
                            d[0] = s;
                            d[1] = q2;
                        }
                    }
                }
            }
            }   // !octet
            AP_STAMP(9, 0);                                        // rows transposed, stored, row sums formed
            if (want_stats) {
                __syncthreads();
                if (tid < CO_TILE) {
                    const int co = cot * CO_TILE + tid;
                    if (co < p.Cout) {
                        float s = 0.f, q2 = 0.f;
#pragma unroll
                        for (int w = 0; w < C::WPX; ++w) {
                            s += sred[(w * CO_TILE + tid) * 2];
                            q2 += sred[(w * CO_TILE + tid) * 2 + 1];
                        }
                        float* d = p.stats + (((long long)n * p.Cout + co) * p.stat_tiles + stat_off +
                                              cur.ty * p.tiles_x + cur.tx) * 2;
                        d[0] = s;
                        d[1] = q2;
                    }
                }
            }
        };
        // ---- FNORM: out = act((conv - mean) * rstd) [+ residual], mean / rstd over the whole (n, c) plane.  The plane is
        // spread over tiles_y * tiles_x tiles held by as many workgroups that run CONCURRENTLY (the host checks that all tiles of
        // an image fall into one round of one XCD's workgroups).  Each workgroup forms its tile's own per-channel mean and centred
        // sum of squares from the accumulators (two passes over registers, no cancellation), publishes the pair in p.stats, bumps
        // the group's counter and waits for the others -- ONE exchange; the pairs combine by Chan's formula
        // (M2 = sum M2_i + n_i sum (mean_i - mean)^2), in fixed order and fp64, to the same bits in every workgroup.  The
        // accumulators stay in registers meanwhile; the result leaves as the split-bf16 copy the next convolution stages and / or
        // as channel-octet fp32 (the residual stream).  MFMA C/D layout: lane (half, l32) holds couts (r & 3) + 8 (r >> 2) + 4 half
        // of pixel column l32, i.e. 4 consecutive channels per register quad: 8-byte bf16 half-slots, 16-byte fp32 octet halves.
        auto fused_norm_epilogue = [&]() __attribute__((always_inline)) {
            float* const epi = reinterpret_cast<float*>(smem + pl * STAGE);
            float* const sred = epi;                          // [WPX][CO_TILE] row sums; later the [group <= 32][CO_TILE] float2 table
            float* const smean = epi + 32 * CO_TILE * 2;      // [CO_TILE]
            float* const srstd = smean + CO_TILE;             // [CO_TILE]
            const int n = cur.n, cot = cur.cot;
            const int co_base = cot * CO_TILE;
            const int tile_in_plane = cur.ty * p.tiles_x + cur.tx;
            unsigned* const ctr = p.fn_counters + ((long long)n * p.co_tiles + cot) * 2;
            const unsigned group = (unsigned)(p.tiles_y * p.tiles_x);
            // 16 row sums over the 32 lanes of a half-wave as a transpose-reduce (see the octet epilogue)
            auto fold = [&](float (&v)[16]) {
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
            };
            const int rr = ((l32 >> 4) & 1) * 8 + ((l32 >> 3) & 1) * 4 + ((l32 >> 2) & 1) * 2 + ((l32 >> 1) & 1);
            const int my_row = (rr & 3) + 8 * (rr >> 2) + 4 * half;       // the row this lane's fold result belongs to
            // per-workgroup channel totals of per-lane values (fold over the half-wave's pixels, then over the four waves' rows)
            auto wg_total = [&](float (&part)[MT]) -> float {
                if ((l32 & 1) == 0) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) sred[wpx * CO_TILE + m * 32 + my_row] = part[m];
                }
                __syncthreads();
                float t = 0.f;
                if (tid < CO_TILE) {
#pragma unroll
                    for (int w = 0; w < C::WPX; ++w) t += sred[w * CO_TILE + tid];
                }
                __syncthreads();
                return t;                                                   // valid in threads tid < CO_TILE
            };
            constexpr float kTilePix = (float)(C::TH * 32);
            // ---- this tile's own mean and centred sum of squares (no exchange yet: Chan's pairwise form combines them later)
            float part[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = acc[m][0][r];
#pragma unroll
                    for (int q = 1; q < NT; ++q) v[r] += acc[m][q][r];
                }
                part[m] = fold(v);
            }
            const float lsum = wg_total(part);
            if (tid < CO_TILE) smean[tid] = lsum * (1.0f / kTilePix);
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mu = smean[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                    v[r] = 0.f;
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const float dlt = acc[m][q][r] - mu;
                        v[r] += dlt * dlt;
                    }
                }
                part[m] = fold(v);
            }
            const float lm2 = wg_total(part);
            // ---- ONE exchange: (mean_i, M2_i) of every tile of the plane
            // Memory ordering WITHOUT agent-scope fences: on gfx950 a release / acquire at agent scope writes back / invalidates the
            // XCD's whole L2 (buffer_wbl2 / buffer_inv sc1) -- with megabytes of dirty output lines in it that cost ~35 us per
            // tile (first version: 292 us per launch).  Instead every access to the exchange data is itself an agent-scope
            // RELAXED atomic (sc1: performed at the coherence point, no cache maintenance), and program order is kept by waiting
            // for the stores' acknowledgement (s_waitcnt vmcnt(0)) before the workgroup barrier that precedes the counter bump.
            float* const slots = p.stats;
            if (tid < CO_TILE && !(p.fn_debug & 4)) {
                float* sl = slots + (((long long)n * p.Cout + co_base + tid) * p.stat_tiles + tile_in_plane) * 2;
                __hip_atomic_store(sl, smean[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sl + 1, lm2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !(p.fn_debug & 5)) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // bounded wait (~0.2 s): if the group's other workgroups never arrive -- the device is shared with work that
                // keeps them from being scheduled -- give up, raise the launch's error flag and let the host fail loudly
                unsigned spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < group) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) {
                        __hip_atomic_store(p.fn_counters + (long long)p.N * p.co_tiles * 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                // The launch leaves its counters as it found them (zero): the second word of the pair counts departures, and the
                // workgroup that leaves last -- everybody is past the wait by then -- clears both.  A replay of a captured HIP
                // graph runs this launch again on the SAME counters; with arrivals left at `group` its wait passed at once and
                // the workgroups combined stale pairs.
                if (__hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == group - 1) {
                    __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            // all 256 threads fetch the group's pairs (thread = channel x tile residue, loads independent), LDS table, then
            // channel threads combine in fixed tile order and fp64: every workgroup of the group computes the same bits
            float2* const tab = reinterpret_cast<float2*>(sred);              // [group <= 32][CO_TILE]
            {
                const int c = tid & (CO_TILE - 1);
                const float* src = slots + ((long long)n * p.Cout + co_base + c) * p.stat_tiles * 2;
                for (unsigned k = tid / CO_TILE; k < group && !(p.fn_debug & 4); k += 256 / CO_TILE)
                    tab[k * CO_TILE + c] = make_float2(__hip_atomic_load(src + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                                       __hip_atomic_load(src + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            __syncthreads();
            if (tid < CO_TILE) {
                double ms = 0.0, m2 = 0.0;
                for (unsigned k = 0; k < group; ++k) ms += (double)tab[k * CO_TILE + tid].x;
                const double mean = ms / (double)group;
                for (unsigned k = 0; k < group; ++k) {
                    const double dm = (double)tab[k * CO_TILE + tid].x - mean;
                    m2 += (double)tab[k * CO_TILE + tid].y + (double)kTilePix * dm * dm;
                }
                const float mu = (float)mean, rs = (float)(1.0 / sqrt(m2 * p.fn_inv_count + (double)p.fn_eps));
                if (tile_in_plane == 0) {                                  // the finished statistics, for whoever reads them later
                    p.fn_mean[(long long)n * p.Cout + co_base + tid] = mu;
                    p.fn_rstd[(long long)n * p.Cout + co_base + tid] = rs;
                }
                smean[tid] = mu;      // (smean / srstd lie behind the table)
                srstd[tid] = rs;
            }
            __syncthreads();
            // ---- output
            const float slope = p.fn_act == 1 ? 0.f : (p.fn_act == 2 ? 0.2f : 1.f);
            const long long ohw = (long long)p.OH * p.OW;
            const int CG = p.Cout >> 3;
            const int ox = cur.tx * 32 + l32;
            unsigned char* const xsb = reinterpret_cast<unsigned char*>(p.fn_xs);
            if (xsb != nullptr && tile_in_plane == 0 && tid < 2 * (CO_TILE / 8)) {      // the all-zero slot that closes every plane
                const int part_ = tid & 1, cg = (co_base >> 3) + (tid >> 1);
                *reinterpret_cast<uint4*>(xsb + (((long long)(n * 2 + part_) * CG + cg) * (ohw + 1) + ohw) * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = m * 32 + g * 8 + 4 * half;              // 4 consecutive channels of the tile
                    const int co = co_base + cl;
                    float mu[4], rs[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { mu[j] = smean[cl + j]; rs[j] = srstd[cl + j]; }
                    const long long obase = ((long long)n * CG + (co >> 3)) * ohw;
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const long long pix = (long long)(cur.ty * C::TH + wpx * NT + q) * p.OW + ox;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = (acc[m][q][g * 4 + j] - mu[j]) * rs[j];
                            v[j] = fmaxf(v[j], slope * v[j]);              // slope 1 / 0 / 0.2: none / ReLU / LeakyReLU, branch-free
                        }
                        if (p.fn_debug & 2) {                                  // (timing experiments only: no loads / stores)
                            if (v[0] == 123.456f) p.fn_mean[0] = v[1] + v[2] + v[3];
                            continue;
                        }
                        if (p.fn_res_oct != nullptr) {
                            const float4 rv = *reinterpret_cast<const float4*>(p.fn_res_oct + (obase + pix) * 8 + 4 * half);
                            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                        } else if (p.fn_res_nchw != nullptr) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += p.fn_res_nchw[((long long)n * p.Cout + co + j) * ohw + pix];
                        }
                        if (p.fn_y_oct != nullptr)
                            *reinterpret_cast<float4*>(p.fn_y_oct + (obase + pix) * 8 + 4 * half) = make_float4(v[0], v[1], v[2], v[3]);
                        if (xsb != nullptr) {
                            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                            bf16x4 hv, lv;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const __bf16 h = (__bf16)v[j];
                                hv[j] = h;
                                lv[j] = (__bf16)(v[j] - (float)h);
                            }
                            const long long slot = ((long long)(n * 2) * CG + (co >> 3)) * (ohw + 1) + pix;
                            *reinterpret_cast<bf16x4*>(xsb + slot * 16 + 8 * half) = hv;
                            *reinterpret_cast<bf16x4*>(xsb + (slot + (long long)CG * (ohw + 1)) * 16 + 8 * half) = lv;
                        }
                    }
                }
        };
        if constexpr (C::FNORM) {
            fused_norm_epilogue();
        } else if (AP_ABLATE(p, 8)) {
            if (acc[0][0][0] == 123.456f) p.y[0] = 1.f;
        } else if (p.act == 0) {
            epilogue(std::integral_constant<int, 0>{});
        } else {
            epilogue(std::integral_constant<int, -1>{});
        }
        AP_STAMP(4, 0);                                            // epilogue done: every output store issued
        if (!has_next) break;
        __syncthreads();                                           // patches and statistics consumed: refill that stage
        AP_STAMP(5, 0);
        if (!AP_ABLATE(p, 1)) issue(nxt, ngoff, 1, pl);
        AP_STAMP(6, 0);                                            // the next tile's chunk 1 is on its way
#ifdef APAMD_ABLATION
        ++stamp_tile;
#endif
        cur = nxt;
        tmask = tapmask_of(cur);
#pragma unroll
        for (int k = 0; k < NIT; ++k) cgoff[k] = ngoff[k];
        tile += tile_step;
    }
    dma_wait_all();                                                // (the last tile's tail stage staged a chunk nobody reads)
#ifdef APAMD_ABLATION
    if (stamping) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        AP_STAMP(10, 0);                                           // every store acknowledged
        __syncthreads();
        const unsigned* src = reinterpret_cast<const unsigned*>(smem_raw + p.stamp_lds_off);
        for (int i = tid; i < 4 * p.stamp_words; i += 256) p.stamps[(long long)blockIdx.x * 4 * p.stamp_words + i] = src[i];
    }
#endif
#undef AP_STAMP
}

}  // namespace apamd
