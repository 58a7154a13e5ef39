// wgrad_host.hip -- C ABI of the weight-gradient operator (wgrad_igemm.h).
//
// ap_conv2d_wgrad = [pad_materialize A] (+ [pad_materialize G] unless g is plain and tile-aligned)
//                   -> wgrad_igemm_f32 (split over pixels) -> wgrad_reduce_kernel.
// The caller provides ONE workspace; its layout is  [A padded][G padded (optional)][partials].
#include "common.h"
#include "conv_head.h"
#include "wgrad_bf16x3.h"
#include "wgrad_xs.h"
#include "wgrad_igemm.h"
#include "wgrad_narrow.h"
#include "wgrad_final.h"
#include "wgrad_k7.h"
#include "dgrad_k7.h"
#include "conv_d0.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace apamd {

// A/B switches are integers on both sides of the boundary: APAMD_X=0 means "off", as ops.py reads it
static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}

struct WgradKernel {
    int S, K, M_TILE, Q_TILE, PR;
    const void* fn;
    size_t lds_bytes;
};

template <class C>
static WgradKernel wk() {
    return WgradKernel{C::S, C::K, C::M_TILE, C::Q_TILE, C::PR, reinterpret_cast<const void*>(&wgrad_igemm_f32<C>),
                       4 * C::lds_floats()};
}

static const std::vector<WgradKernel>& wgrad_registry() {
    static std::vector<WgradKernel> v = {
        wk<WgradCfg<1, 3, 2, 4, 2>>(), wk<WgradCfg<1, 4, 2, 4, 2>>(), wk<WgradCfg<1, 7, 2, 4, 2>>(),
        wk<WgradCfg<2, 3, 2, 4, 1>>(), wk<WgradCfg<2, 4, 2, 4, 1>>(),
        wk<WgradCfg<1, 7, 1, 3, 2>>(),                 // 64 x 192 tiles: the 7x7 stems (Cin = 3: Q = 147)
        wk<WgradCfg<2, 3, 1, 1, 1>>(),                 // 64 x 64 tiles: the landmark encoder's 8 -> 16 -> 16 layers
        wk<WgradCfg<2, 3, 1, 4, 1>>(),                 // 64 x 256 tiles: 64-output stride-2 layers
    };
    return v;
}

struct WgradPlan {
    const WgradKernel* k = nullptr;
    int Cin = 0, Q = 0, tiles_x = 0, tiles_y = 0, nstages = 0, P = 0, m_tiles = 0, q_tiles = 0;
    int GHp = 0, GWp = 0, Hp = 0, Wp = 0;
    bool g_direct = false;
    long long a_floats = 0, g_floats = 0, part_floats = 0;
    // split-bf16 kernel (wgrad_bf16x3.h): operands as [n][part][row][x/8][channel] pixel-octet slots
    bool bf3 = false;
    int c_tiles = 0, Mp = 0, Cp = 0, GX8 = 0, AX8 = 0;
    // ... of the space-to-depth form of a stride-2 layer: a 2 x 2 stride-1 layer over 4 Cin channels
    bool s2d = false;
    // ... of the row form of a 7x7 stem: a 1 x 7 layer over 7 Cin channels (channel ci * 7 + ky = the input shifted by ky rows)
    bool rows = false;
    bool wide = false;                      // the 8-wave workgroup: 128-channel M tiles (split-bf16 arithmetic, K <= 3, M % 128 == 0)
    int Kb = 0, Cb = 0, Hb = 0, Wb = 0;     // kernel size, channels and operand size the bf16 GEMM kernel sees
    // streaming kernel for 1..2 input channels (wgrad_narrow.h): > 0 = output channels per workgroup
    int narrow_cob = 0, narrow_ppt = 0, gwc = 0, gwc_shift = 0, rpi = 0, rows_per_block = 0;
};

// split-bf16 instantiations by kernel size (stride 1)
struct WgradBf3Kernel {
    int K;
    const void* fn;        // head + tail staged, three products per tap (AP_PRECISION_BF16X3)
    size_t lds_bytes;
    const void* fn1;       // head planes only, one product per tap (AP_PRECISION_BF16): two workgroups per CU
    size_t lds_bytes1;
    const void* fn_wide = nullptr;   // the 8-wave workgroup (128 x 64 channel tile, WgradBf3Cfg WM = 4) of the split-bf16 form
    size_t lds_bytes_wide = 0;
};
static const std::vector<WgradBf3Kernel>& wgrad_bf3_registry() {
    static std::vector<WgradBf3Kernel> v = {
#define APAMD_WBF3(K)                                                                                         \
    {K, reinterpret_cast<const void*>(&wgrad_bf16x3<WgradBf3Cfg<K, 2>>), WgradBf3Cfg<K, 2>::lds_bytes(),     \
     reinterpret_cast<const void*>(&wgrad_bf16x3<WgradBf3Cfg<K, 1>>), WgradBf3Cfg<K, 1>::lds_bytes()}
#define APAMD_WBF3W(K)                                                                                        \
    {K, reinterpret_cast<const void*>(&wgrad_bf16x3<WgradBf3Cfg<K, 2>>), WgradBf3Cfg<K, 2>::lds_bytes(),     \
     reinterpret_cast<const void*>(&wgrad_bf16x3<WgradBf3Cfg<K, 1>>), WgradBf3Cfg<K, 1>::lds_bytes(),         \
     reinterpret_cast<const void*>(&wgrad_bf16x3<WgradBf3Cfg<K, 2, K, 4>>), WgradBf3Cfg<K, 2, K, 4>::lds_bytes()}
        APAMD_WBF3W(3), APAMD_WBF3(4), APAMD_WBF3W(2),   // (K = 2: the space-to-depth forms of stride-2 layers)
#undef APAMD_WBF3W
#undef APAMD_WBF3
        // K = 7: the ROW form of the 7x7 stems (1 x 7 taps over 7 Cin row channels: WgradBf3Cfg KY = 1)
        {7, reinterpret_cast<const void*>(&wgrad_bf16x3<WgradBf3Cfg<7, 2, 1>>), WgradBf3Cfg<7, 2, 1>::lds_bytes(),
         reinterpret_cast<const void*>(&wgrad_bf16x3<WgradBf3Cfg<7, 1, 1>>), WgradBf3Cfg<7, 1, 1>::lds_bytes()},
    };
    return v;
}

static int num_cus_w() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) return 256;
    return n;
}

static long long round4(long long x) { return (x + 3) / 4 * 4; }

static int make_wgrad_plan(const ap_wgrad_desc* d, WgradPlan& pl) {
    if (!d) return fail(AP_ERR_INVALID, "wgrad: null descriptor");
    if (d->nsrc < 1 || d->nsrc > kMaxSeg) return fail(AP_ERR_INVALID, "wgrad: nsrc=%d", d->nsrc);
    if (d->N < 1 || d->M < 1 || d->GH < 1 || d->GW < 1 || d->H < 1 || d->W < 1) return fail(AP_ERR_INVALID, "wgrad: bad dims");
    {
        // several tile shapes of a family: the one whose padded (M x Q) tile grid wastes least (the 7x7 stems are
        // M = 32..64 by Q = 147: 64 x 192 tiles are 77 % full where 128 x 256 ones are 29 %)
        int cin = 0;
        for (int s = 0; s < d->nsrc; ++s) cin += d->src[s].C;
        const long long Q = (long long)cin * d->K * d->K;
        long long best = -1;
        for (const auto& k : wgrad_registry())
            if (k.S == d->stride && k.K == d->K) {
                const long long padded = ((d->M + k.M_TILE - 1) / k.M_TILE) * (long long)k.M_TILE *
                                         (((Q + k.Q_TILE - 1) / k.Q_TILE) * k.Q_TILE);
                // (a smaller tile re-reads its operands more often: it must save at least 30 % of the padded work)
                if (best < 0 || padded * 10 < best * 7) { best = padded; pl.k = &k; }
            }
    }
    if (!pl.k) return fail(AP_ERR_UNSUPPORTED, "wgrad: no kernel for stride %d, k %d", d->stride, d->K);
    if (d->pad_mode == AP_PAD_REFLECT && (d->pad >= d->H || d->pad >= d->W))
        return fail(AP_ERR_INVALID, "wgrad: reflection pad %d >= input size", d->pad);
    // the iterated grid must be the conv output grid of the shifted tensor
    const int oh = (d->H + 2 * d->pad - d->K) / d->stride + 1, ow = (d->W + 2 * d->pad - d->K) / d->stride + 1;
    if (oh != d->GH || ow != d->GW)
        return fail(AP_ERR_INVALID, "wgrad: grid %dx%d does not match conv output %dx%d", d->GH, d->GW, oh, ow);
    pl.Cin = 0;
    for (int s = 0; s < d->nsrc; ++s) {
        if (d->src[s].C < 1) return fail(AP_ERR_INVALID, "wgrad: segment %d has C=%d", s, d->src[s].C);
        pl.Cin += d->src[s].C;
    }
    const int S = d->stride, K = d->K;
    pl.Q = pl.Cin * K * K;
    // 1..2 input channels (landmark encoder / PatchGAN first layer): one streaming pass, no operand copies (wgrad_narrow.h)
    {
        int cob = 0;
        if (d->nsrc == 1 && d->g.mean == nullptr && d->g.act == AP_ACT_NONE) {
            if (K == 3 && S == 1 && pl.Cin == 1) cob = 8;
            else if (K == 4 && S == 2 && pl.Cin == 1) cob = 4;
            else if (K == 4 && S == 2 && pl.Cin == 2) cob = 4;
        }
        if (cob) {
            pl.narrow_cob = cob;
            const int groups = (d->M + cob - 1) / cob;
            int gwc = 1, sh = 0;
            while (gwc < d->GW && gwc < 256) { gwc <<= 1; ++sh; }
            pl.gwc = gwc; pl.gwc_shift = sh; pl.rpi = 256 / gwc;
            const long long total_rows = (long long)d->N * d->GH;
            long long P = std::max<long long>(1, 1024 / groups);
            long long rpb = (total_rows + P - 1) / P;
            // K = 4 stride 2 with zero pad 1 on rows a workgroup spans exactly (the PatchGAN first layer): the LDS-staged
            // form (one output pixel per thread and iteration: 2 and 4 measured slower, 277 registers) (wgrad_narrow_s2k4_kernel)
            constexpr int kNarrowPPT = 1;
            if (K == 4 && S == 2 && d->pad == 1 && d->pad_mode == AP_PAD_ZERO && gwc == d->GW && d->W == 2 * d->GW &&
                d->H == 2 * d->GH && (d->GH % (pl.rpi * kNarrowPPT)) == 0)
                pl.narrow_ppt = kNarrowPPT;
            const int rit = pl.rpi * (pl.narrow_ppt ? pl.narrow_ppt : 1);
            rpb = (rpb + rit - 1) / rit * rit;
            pl.rows_per_block = (int)rpb;
            pl.P = (int)((total_rows + rpb - 1) / rpb);
            pl.part_floats = (long long)pl.P * d->M * pl.Q;
            return AP_OK;
        }
    }
    const char* nob = getenv("APAMD_NO_BF16X3");
    const bool bf_ok = d->precision != AP_PRECISION_FP32 && d->M >= 48 && !(nob && atoi(nob));
    // stride-2 3x3 / 4x4 pad-1 layers (the generator's encoder, the PatchGAN body): the space-to-depth form -- a 2 x 2
    // stride-1 layer over 4 Cin channels -- runs on the same bf16 GEMM kernel (16x the fp32 MFMA rate per product; a
    // 3x3 layer carries 7 of 16 all-zero taps along)
    const char* nos = getenv("APAMD_NO_S2D_WGRAD");
    const bool s2d = bf_ok && S == 2 && (K == 3 || K == 4) && d->pad == 1 && d->pad_mode == AP_PAD_ZERO && pl.Cin >= 8 &&
                     d->H % 2 == 0 && d->W % 2 == 0 && !(nos && atoi(nos));
    pl.Kb = K; pl.Cb = pl.Cin; pl.Hb = d->H; pl.Wb = d->W;
    if (s2d) { pl.s2d = true; pl.Kb = 2; pl.Cb = 4 * pl.Cin; pl.Hb = d->H / 2 + 1; pl.Wb = d->W / 2 + 1; }
    // 7x7 stems on a few channels (the generator's three input layers): the row form -- a 1 x 7 layer over 7 Cin channels -- as the
    // forward pass runs them (ap_split_prepass_rows); one 64-channel tile holds up to 9 input channels
    // OPT-IN (APAMD_ROWS_WGRAD=1), plain-bf16 arithmetic only.  Measured in the train step at 2B = 32 (profiles/r05_wgrad_routes.md):
    // the kernel itself takes 125 us, but both operands have to be prepared for this one launch -- the 64-channel gradient at
    // 256 x 256 into pixel-octet slots (~140 us) and the row view padded from 21 to the kernel's 64 channels (~210 us, 277 MB
    // written) -- so the whole operator is ~470 us and 1.1 GB more HBM traffic per stem against 433 us on the fp32 kernel (and ~680 us
    // with three products and both parts).  Kept for the tests and as the starting point of a 32-channel tile.
    const char* wr = getenv("APAMD_ROWS_WGRAD");
    const bool rows_m = d->precision == AP_PRECISION_BF16 && d->M >= 24 && !(nob && atoi(nob)) && wr && atoi(wr);
    const bool rows = rows_m && S == 1 && K == 7 && d->pad == 3 && d->nsrc == 1 && pl.Cin * 7 <= 64;
    if (rows) { pl.rows = true; pl.Cb = 7 * pl.Cin; }
    if (s2d || rows || (bf_ok && S == 1 && (K == 3 || K == 4) && pl.Cin >= 32)) {
        // wide layer: operands split into bf16 head + tail, bf16 matrix pipe (wgrad_bf16x3.h)
        pl.bf3 = true;
        pl.tiles_x = (d->GW + 31) / 32;
        pl.tiles_y = (d->GH + 1) / 2;
        pl.nstages = d->N * pl.tiles_y * pl.tiles_x;
        // split-bf16 arithmetic on 128-output multiples: the 8-wave workgroup (two waves per SIMD where the 4-wave one, whose
        // head + tail stages fill the LDS, has one)
        // (plain bf16 keeps two 4-wave workgroups per CU: the 8-wave form measured the same, 160.9 against 160.0 us per 3x3 layer)
        pl.wide = d->precision == AP_PRECISION_BF16X3 && !pl.rows && pl.Kb <= 3 && d->M % 128 == 0;
        const int mtile = pl.wide ? 128 : 64;
        pl.m_tiles = (d->M + mtile - 1) / mtile;
        pl.c_tiles = (pl.Cb + 63) / 64;
        const char* e = getenv("APAMD_WGRAD_BLOCKS");
        if (e) {                                                  // tuning / test knob: never silent
            static bool told = false;
            if (!told) fprintf(stderr, "libapamd: APAMD_WGRAD_BLOCKS=%s overrides the workgroup count\n", e);
            told = true;
        }
        // one workgroup per CU (its LDS stages fill a CU); two with head-only staging
        const int target = e ? atoi(e) : num_cus_w() * (d->precision == AP_PRECISION_BF16 && (pl.Kb <= 3 || pl.rows) && !pl.wide ? 2 : 1);
        int P = target / (pl.m_tiles * pl.c_tiles);
        if (P > pl.nstages / 2) P = pl.nstages / 2;
        if (P < 1) P = 1;
        pl.P = P;
        pl.Mp = pl.m_tiles * mtile;
        pl.Cp = pl.c_tiles * 64;
        pl.GHp = pl.tiles_y * 2;
        pl.GX8 = pl.tiles_x * 4;
        pl.Hp = pl.rows ? pl.GHp : pl.GHp + pl.Kb - 1;
        pl.AX8 = pl.tiles_x * 4 + 1;
        pl.a_floats = (long long)d->N * 2 * pl.Hp * pl.AX8 * pl.Cp * 4;      // 16-byte slots -> floats
        pl.g_floats = (long long)d->N * 2 * pl.GHp * pl.GX8 * pl.Mp * 4;
        pl.part_floats = (long long)pl.P * pl.m_tiles * pl.c_tiles * (pl.wide ? 8 : 4) * (pl.rows ? pl.Kb : pl.Kb * pl.Kb) * 1024;   // accumulator-order tiles
        return AP_OK;
    }
    const int PR = pl.k->PR;
    pl.tiles_x = (d->GW + 31) / 32;
    pl.tiles_y = (d->GH + PR - 1) / PR;
    pl.nstages = d->N * pl.tiles_y * pl.tiles_x;
    pl.m_tiles = (d->M + pl.k->M_TILE - 1) / pl.k->M_TILE;
    pl.q_tiles = (pl.Q + pl.k->Q_TILE - 1) / pl.k->Q_TILE;
    const char* e = getenv("APAMD_WGRAD_BLOCKS");
    const int target = e ? atoi(e) : 1024;
    int P = (target + pl.m_tiles * pl.q_tiles - 1) / (pl.m_tiles * pl.q_tiles);
    if (P > pl.nstages) P = pl.nstages;
    if (P < 1) P = 1;
    pl.P = P;
    pl.GHp = pl.tiles_y * PR;
    pl.GWp = pl.tiles_x * 32;
    pl.Hp = (pl.GHp - 1) * S + K;
    pl.Wp = (int)round4((long long)(pl.GWp - 1) * S + K);
    // every padded row/plane the kernel can touch must exist
    if (pl.Hp < d->H + 2 * d->pad) pl.Hp = d->H + 2 * d->pad;
    if (pl.Wp < d->W + 2 * d->pad) pl.Wp = (int)round4(d->W + 2 * d->pad);
    pl.g_direct = d->g.mean == nullptr && d->g.act == AP_ACT_NONE && pl.GHp == d->GH && pl.GWp == d->GW;
    pl.a_floats = round4((long long)d->N * pl.Cin * pl.Hp * pl.Wp);
    pl.g_floats = pl.g_direct ? 0 : round4((long long)d->N * d->M * pl.GHp * pl.GWp);
    pl.part_floats = (long long)pl.P * d->M * pl.Q;
    return AP_OK;
}

static int launch_pad(const ap_src* segs, int nseg, int N, int C, int H, int W, int pad, int pad_mode, int Hp, int Wp,
                      float* out, hipStream_t stream) {
    PadParams p;
    memset(&p, 0, sizeof(p));
    p.nseg = nseg;
    int cbeg = 0;
    for (int s = 0; s < nseg; ++s) {
        p.seg[s].data = segs[s].data; p.seg[s].mean = segs[s].mean; p.seg[s].rstd = segs[s].rstd;
        p.seg[s].C = segs[s].C; p.seg[s].act = segs[s].act; p.seg[s].chunk_begin = cbeg;
        cbeg += segs[s].C;
    }
    p.N = N; p.C = C; p.H = H; p.W = W; p.pad = pad; p.pad_mode = pad_mode; p.Hp = Hp; p.Wp = Wp; p.out = out;
    if (N > 65535 || C > 65535) return fail(AP_ERR_UNSUPPORTED, "pad_materialize: N=%d C=%d", N, C);
    int bx = (Hp * Wp + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(pad_materialize_kernel, dim3(bx, C, N), dim3(256), 0, stream, p);
    return check_launch("pad_materialize_kernel");
}

// raises a kernel's dynamic-LDS limit once per process
static int set_dyn_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::vector<const void*> done;
    std::lock_guard<std::mutex> lk(mu);
    for (const void* f : done) if (f == fn) return AP_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    done.push_back(fn);
    return AP_OK;
}

static int launch_split_transpose(const ap_src* segs, int nseg, int N, int C, int H, int W, int pad, int pad_mode,
                                  int Hp, int X8, int Cp, uint4* out, hipStream_t stream, int s2d_c, int heads_only, int rows_k = 0) {
    SplitTParams p;
    memset(&p, 0, sizeof(p));
    p.nseg = nseg;
    int cbeg = 0;
    bool any_b16 = false;
    for (int s = 0; s < nseg; ++s) {
        p.seg[s].data = segs[s].data; p.seg[s].mean = segs[s].mean; p.seg[s].rstd = segs[s].rstd;
        // ap_src.act bit 8: the segment's data are bf16 values (a raw output of ap_conv2d_fwd_bf16out); only the padded-row kernel reads those
        p.seg[s].C = segs[s].C; p.seg[s].act = segs[s].act & 0xff; p.seg[s].chunk_begin = cbeg;
        p.seg[s].pad_ = (segs[s].act >> 8) & 1;
        any_b16 = any_b16 || p.seg[s].pad_;
        cbeg += segs[s].C;
    }
    p.N = N; p.C = C; p.H = H; p.W = W; p.pad = pad; p.pad_mode = pad_mode; p.Hp = Hp; p.X8 = X8; p.Cp = Cp; p.out = out;
    p.s2d_c = s2d_c;
    p.heads_only = heads_only;
    p.rows_k = rows_k;
    if (rows_k > 0) {                                 // the row view exists in the general kernel only
        if (any_b16 || s2d_c) return fail(AP_ERR_UNSUPPORTED, "split_transpose: row view of a bf16 / space-to-depth source");
        if (N > 65535 || Cp / 64 > 65535) return fail(AP_ERR_UNSUPPORTED, "split_transpose: N=%d C=%d", N, C);
        hipLaunchKernelGGL(split_transpose_kernel, dim3((Hp * X8 + 7) / 8, Cp / 64, N), dim3(256), 0, stream, p);
        return check_launch("split_transpose_kernel");
    }
    if (any_b16 && !((W == 64 || W == 128 || W == 256 || W == 32) && s2d_c == 0 && pad == 1))
        return fail(AP_ERR_UNSUPPORTED, "split_transpose: a bf16 source needs the padded-row form (pad 1, W in {32, 64, 128, 256})");
    if (N > 65535 || Cp / 64 > 65535) return fail(AP_ERR_UNSUPPORTED, "split_transpose: N=%d C=%d", N, C);
    if (nseg == 1 && pad == 0 && s2d_c == 0 && X8 * 8 == W) {
        // unpadded operand with whole octet rows: 16-byte loads, 1 KiB per wave (split_transpose_vec_kernel)
        static bool attr = false;
        const size_t lds = 64 * 257 * sizeof(float);
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&split_transpose_vec_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr = true;
        }
        hipLaunchKernelGGL(split_transpose_vec_kernel, dim3((Hp * X8 + 31) / 32, Cp / 64, N), dim3(256), lds, stream, p);
        return check_launch("split_transpose_vec_kernel");
    }
    const bool rows_ok = (W == 64 || W == 128 || W == 256 || (W == 32 && s2d_c == 0));
    if (rows_ok && s2d_c == 0 && pad == 1) {
        // padded rows, 16-byte loads (split_transpose_pad_kernel)
        const int R = 256 / W;
        const size_t lds = (size_t)64 * (R * X8 * 8 + 1) * sizeof(float);
        int rc = set_dyn_lds(reinterpret_cast<const void*>(&split_transpose_pad_kernel), 96 * 1024);
        if (rc != AP_OK) return rc;
        hipLaunchKernelGGL(split_transpose_pad_kernel, dim3((Hp + R - 1) / R, Cp / 64, N), dim3(256), lds, stream, p);
        return check_launch("split_transpose_pad_kernel");
    }
    if (rows_ok && s2d_c > 0 && s2d_c % 32 == 0 && nseg == 1 && C == 4 * s2d_c && Cp == C) {
        // space-to-depth view, whole source rows (split_transpose_s2d_kernel)
        const int R = 256 / W;
        const size_t lds = (size_t)64 * (R * X8 * 8 + 1) * sizeof(float);
        int rc = set_dyn_lds(reinterpret_cast<const void*>(&split_transpose_s2d_kernel), 96 * 1024);
        if (rc != AP_OK) return rc;
        hipLaunchKernelGGL(split_transpose_s2d_kernel, dim3(((Hp + R - 1) / R) * 2, s2d_c / 32, N), dim3(256), lds, stream, p);
        return check_launch("split_transpose_s2d_kernel");
    }
    hipLaunchKernelGGL(split_transpose_kernel, dim3((Hp * X8 + 7) / 8, Cp / 64, N), dim3(256), 0, stream, p);
    return check_launch("split_transpose_kernel");
}

// the shifted operand re-tiled from the forward pass's split copies (xs_transpose_kernel); C = channels of the view
static int launch_xs_transpose(const void* const* xs, const int* seg_c, int nseg, int N, int H, int W, int pad, int pad_mode,
                               int Hp, int X8, int Cp, int parts, uint4* out, hipStream_t stream, int s2d_c = 0, int H0 = 0, int W0 = 0) {
    XsTParams p;
    memset(&p, 0, sizeof(p));
    p.nseg = nseg;
    int cg = 0;
    for (int s = 0; s < nseg; ++s) {
        p.xs[s] = reinterpret_cast<const uint4*>(xs[s]);
        p.cg_begin[s] = cg;
        cg += seg_c[s] / 8;
    }
    p.cg_begin[nseg] = cg;
    p.N = N; p.H = H; p.W = W; p.pad = pad; p.pad_mode = pad_mode; p.Hp = Hp; p.X8 = X8; p.Cp = Cp; p.parts = parts; p.out = out;
    p.s2d_c = s2d_c; p.H0 = H0; p.W0 = W0;
    if (N > 65535 || Cp / 8 > 65535) return fail(AP_ERR_UNSUPPORTED, "xs_transpose: N=%d Cp=%d", N, Cp);
    hipLaunchKernelGGL(xs_transpose_kernel, dim3((Hp * X8 + 31) / 32, (Cp / 8 + kXsCgPerThread - 1) / kXsCgPerThread, N), dim3(256), 0,
                       stream, p);
    return check_launch("xs_transpose_kernel");
}

// can the shifted operand of this plan come from the forward split copies the descriptor carries?
static bool wgrad_xs_route(const ap_wgrad_desc* d, const WgradPlan& pl) {
    if (!pl.bf3 || pl.rows || env_int("APAMD_NO_XS_WGRAD", 0)) return false;
    if (d->precision != AP_PRECISION_BF16 && d->xs_parts != 2) return false;      // a split-bf16 product reads the tail planes
    if (d->xs_parts != 1 && d->xs_parts != 2) return false;
    if (pl.s2d) return (d->src_xs_s2d != nullptr || d->src_xs[0] != nullptr) && d->nsrc == 1 && pl.Cin % 8 == 0;
    for (int s = 0; s < d->nsrc; ++s)
        if (!d->src_xs[s] || d->src[s].C % 8 != 0) return false;
    return true;
}

static std::mutex g_wattr_mu;
static std::vector<const void*> g_wattr_done;

static int ensure_wattr(const void* fn) {
    std::lock_guard<std::mutex> lk(g_wattr_mu);
    for (auto f : g_wattr_done)
        if (f == fn) return AP_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    g_wattr_done.push_back(fn);
    return AP_OK;
}

}  // namespace apamd

using namespace apamd;

// ---- both operands straight from the convolutions' split copies (wgrad_xs.h): no operand preparation at all
static bool wgrad_xs_direct_ok(const ap_wgrad_desc* d, const WgradPlan& pl) {
    if (!pl.bf3 || pl.rows || env_int("APAMD_NO_XS_DIRECT", 0)) return false;
    if (d->M % 8 != 0) return false;
    if (pl.s2d) {
        // the space-to-depth form of a stride-2 layer: the 2 x 2 layer over the forward pass's space-to-depth copy (split-bf16 only)
        // (that copy, or the plain one: the view is then gathered from it)
        if (d->precision != AP_PRECISION_BF16X3 || d->xs_parts != 2 || !(d->src_xs_s2d || d->src_xs[0]) || d->nsrc != 1 || pl.Cin % 8 != 0) return false;
        if ((long long)d->N * 2 * (d->M / 8) * ((long long)d->GH * d->GW + 1) >= (1LL << 31)) return false;
        return (long long)d->N * 2 * (pl.Cb / 8) * ((long long)pl.Hb * pl.Wb + 1) < (1LL << 31);
    }
    // 3x3, and (split bf16) the PatchGAN's 4x4 stride-1 layers: 16 accumulator tiles, the 4-wave workgroup, one per CU
    if (d->stride != 1 || !(d->K == 3 || (d->K == 4 && d->precision == AP_PRECISION_BF16X3))) return false;
    // (the kernel indexes a copy's 16-byte slots with 32 bits)
    if ((long long)d->N * 2 * (d->M / 8) * ((long long)d->GH * d->GW + 1) >= (1LL << 31)) return false;
    for (int s = 0; s < d->nsrc; ++s)
        if ((long long)d->N * 2 * (d->src[s].C / 8) * ((long long)d->H * d->W + 1) >= (1LL << 31)) return false;
    if (d->precision != AP_PRECISION_BF16 && d->xs_parts != 2) return false;
    if (d->xs_parts != 1 && d->xs_parts != 2) return false;
    for (int s = 0; s < d->nsrc; ++s)
        if (!d->src_xs[s] || d->src[s].C % 8 != 0) return false;
    return true;
}

template <class C>
static int launch_wgrad_xs(const WgradXsParams& p, unsigned nblk, hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(&wgrad_xs_kernel<C>);
    int rc = ensure_wattr(fn);
    if (rc) return rc;
    WgradXsParams q = p;
    void* args[] = {&q};
    hipError_t e = hipLaunchKernel(fn, dim3(nblk), dim3(C::NT), args, C::lds_bytes(), stream);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "wgrad_xs launch: %s", hipGetErrorString(e));
    return AP_OK;
}

extern "C" int32_t ap_conv2d_wgrad_xs_ok(const ap_wgrad_desc* d) {
    WgradPlan pl;
    if (make_wgrad_plan(d, pl)) return 0;
    return wgrad_xs_direct_ok(d, pl) ? 1 : 0;
}

extern "C" int ap_conv2d_wgrad_xs(const ap_wgrad_desc* d, const void* g_xs, float* workspace, float* dw, ap_stream_t stream_) {
    WgradPlan pl;
    int rc = make_wgrad_plan(d, pl);
    if (rc) return rc;
    if (!g_xs || !workspace || !dw) return fail(AP_ERR_INVALID, "wgrad_xs: null pointer");
    if (!wgrad_xs_direct_ok(d, pl)) return fail(AP_ERR_UNSUPPORTED, "wgrad_xs: the layer is not on this route (ap_conv2d_wgrad_xs_ok)");
    hipStream_t stream = (hipStream_t)stream_;
    const bool b16 = d->precision == AP_PRECISION_BF16;
    WgradXsParams p;
    memset(&p, 0, sizeof(p));
    p.g_xs = reinterpret_cast<const uint4*>(g_xs);
    p.N = d->N; p.M = d->M; p.GH = d->GH; p.GW = d->GW;
    if (pl.s2d) {
        p.nseg = 1;
        p.a_xs[0] = reinterpret_cast<const uint4*>(d->src_xs_s2d ? d->src_xs_s2d : d->src_xs[0]);
        p.a_cg_begin[0] = 0; p.a_cg_begin[1] = pl.Cb / 8;
        p.H = pl.Hb; p.W = pl.Wb; p.pad = 0; p.pad_mode = AP_PAD_ZERO;
        p.s2d_c = d->src_xs_s2d ? 0 : pl.Cin;
    } else {
        p.nseg = d->nsrc;
        int cg = 0;
        for (int s = 0; s < d->nsrc; ++s) {
            p.a_xs[s] = reinterpret_cast<const uint4*>(d->src_xs[s]);
            p.a_cg_begin[s] = cg;
            cg += d->src[s].C / 8;
        }
        p.a_cg_begin[d->nsrc] = cg;
        p.H = d->H; p.W = d->W; p.pad = d->pad; p.pad_mode = d->pad_mode;
    }
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.nstages = pl.nstages; p.P = pl.P; p.m_tiles = pl.m_tiles; p.c_tiles = pl.c_tiles;
    p.partial = workspace;
    const unsigned nblk = (unsigned)(pl.m_tiles * pl.c_tiles * pl.P);
    if (pl.s2d) rc = pl.wide ? launch_wgrad_xs<WgradXsCfg<2, 2, 4>>(p, nblk, stream) : launch_wgrad_xs<WgradXsCfg<2, 2, 2>>(p, nblk, stream);
    else if (d->K == 4) rc = launch_wgrad_xs<WgradXsCfg<4, 2, 2>>(p, nblk, stream);
    else if (pl.wide) rc = launch_wgrad_xs<WgradXsCfg<3, 2, 4>>(p, nblk, stream);
    else if (b16) rc = launch_wgrad_xs<WgradXsCfg<3, 1, 2>>(p, nblk, stream);
    else rc = launch_wgrad_xs<WgradXsCfg<3, 2, 2>>(p, nblk, stream);
    if (rc) return rc;
    const int T = pl.Kb * pl.Kb;
    const long long total = (long long)pl.m_tiles * pl.c_tiles * (pl.wide ? 8 : 4) * T * 1024;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(wgrad_bf3_reduce_kernel, dim3(blocks), dim3(256), 0, stream, workspace, pl.P, d->M, pl.Cb, T, pl.c_tiles, total,
                       pl.s2d ? pl.Cin : 0, d->K, dw, pl.wide ? 4 : 2);
    return check_launch("wgrad_bf3_reduce_kernel");
}


extern "C" {

int ap_pad_materialize(const ap_src* src, int32_t nsrc, int32_t N, int32_t H, int32_t W, int32_t pad, int32_t pad_mode,
                       int32_t Hp, int32_t Wp, float* out, ap_stream_t stream) {
    if (!src || !out || nsrc < 1 || nsrc > kMaxSeg) return fail(AP_ERR_INVALID, "pad_materialize: bad arguments");
    if (Hp < H + 2 * pad || Wp < W + 2 * pad) return fail(AP_ERR_INVALID, "pad_materialize: output smaller than padded input");
    if (pad_mode == AP_PAD_REFLECT && (pad >= H || pad >= W)) return fail(AP_ERR_INVALID, "pad_materialize: reflection pad too large");
    int C = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (!src[s].data || src[s].C < 1) return fail(AP_ERR_INVALID, "pad_materialize: segment %d", s);
        if ((src[s].mean == nullptr) != (src[s].rstd == nullptr)) return fail(AP_ERR_INVALID, "pad_materialize: mean/rstd mismatch");
        C += src[s].C;
    }
    return launch_pad(src, nsrc, N, C, H, W, pad, pad_mode, Hp, Wp, out, (hipStream_t)stream);
}

int ap_conv_head_wgrad(const ap_src* src, const float* g, int32_t N, int32_t H, int32_t W, int32_t K, int32_t pad,
                       float* dw, ap_stream_t stream) {
    if (!src || !src->data || !g || !dw) return fail(AP_ERR_INVALID, "conv_head_wgrad: null pointer");
    if ((src->mean == nullptr) != (src->rstd == nullptr)) return fail(AP_ERR_INVALID, "conv_head_wgrad: mean/rstd mismatch");
    const int OH = H + 2 * pad - K + 1, OW = W + 2 * pad - K + 1;
    if (K != 4 || pad != 1 || N < 1 || src->C < 1 || src->C > 65535 || OH < 1 || OW < 1 || W > kHeadMaxW || H > kHeadMaxH)
        return fail(AP_ERR_UNSUPPORTED, "conv_head_wgrad: built for 4x4 pad-1 heads on maps up to %dx%d (K=%d, pad=%d, %dx%d)",
                    kHeadMaxH, kHeadMaxW, K, pad, H, W);
    HeadParams p;
    memset(&p, 0, sizeof(p));
    p.src.data = src->data; p.src.mean = src->mean; p.src.rstd = src->rstd; p.src.C = src->C; p.src.act = src->act;
    p.N = N; p.C = src->C; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
    p.g = g; p.dw = dw;
    hipLaunchKernelGGL(conv_head_wgrad_kernel, dim3(src->C), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("conv_head_wgrad_kernel");
}

// pixel-tile split of wgrad_final_kernel: ~1024 workgroups in all
static void final_split(int N, int C, int H, int W, int& tiles_x, int& tiles_y, int& tpb, int& P) {
    tiles_x = (W + 63) / 64;
    tiles_y = (H + 16 * kFinalStrips - 1) / (16 * kFinalStrips);
    const long long total = (long long)N * tiles_x * tiles_y;
    long long want = std::max<long long>(1, 1024 / std::max(C, 1));
    if (want > total) want = total;
    tpb = (int)((total + want - 1) / want);
    P = (int)((total + tpb - 1) / tpb);
}

int64_t ap_conv_final_wgrad_workspace_floats(int32_t N, int32_t C, int32_t H, int32_t W) {
    if (N < 1 || C < 1 || H < 1 || W < 1) return fail(AP_ERR_INVALID, "conv_final_wgrad: bad sizes");
    int tx, ty, tpb, P;
    final_split(N, C, H, W, tx, ty, tpb, P);
    return (int64_t)P * C * 49;
}

int ap_conv_final_wgrad(const ap_src* src, const float* g, int32_t N, int32_t H, int32_t W, int32_t K, int32_t pad,
                        int32_t pad_mode, float* workspace, float* dw, ap_stream_t stream_) {
    if (!src || !src->data || !g || !workspace || !dw) return fail(AP_ERR_INVALID, "conv_final_wgrad: null pointer");
    if ((src->mean == nullptr) != (src->rstd == nullptr)) return fail(AP_ERR_INVALID, "conv_final_wgrad: mean/rstd mismatch");
    if (K != 7 || pad != 3 || N < 1 || src->C < 1 || src->C > 65535 || H < 1 || W < 1)
        return fail(AP_ERR_UNSUPPORTED, "conv_final_wgrad: built for 7x7 pad-3 layers with one output channel (K=%d, pad=%d)", K, pad);
    if (pad_mode == AP_PAD_REFLECT && (pad >= H || pad >= W)) return fail(AP_ERR_INVALID, "conv_final_wgrad: reflection pad too large");
    if (src->act < 0 || src->act > 2) return fail(AP_ERR_INVALID, "conv_final_wgrad: act %d", src->act);
    hipStream_t stream = (hipStream_t)stream_;
    WgradFinalParams p;
    memset(&p, 0, sizeof(p));
    p.src.data = src->data; p.src.mean = src->mean; p.src.rstd = src->rstd; p.src.C = src->C; p.src.act = src->act;
    p.g = g; p.N = N; p.C = src->C; p.H = H; p.W = W; p.pad_mode = pad_mode;
    int P;
    final_split(N, src->C, H, W, p.tiles_x, p.tiles_y, p.tiles_per_block, P);
    p.partial = workspace;
    if ((W & 3) == 0) hipLaunchKernelGGL(wgrad_final_kernel<true>, dim3(P, src->C), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(wgrad_final_kernel<false>, dim3(P, src->C), dim3(256), 0, stream, p);
    int rc = check_launch("wgrad_final_kernel");
    if (rc) return rc;
    const long long n = (long long)src->C * 49;
    launch_wgrad_reduce(stream, workspace, P, n, dw);
    return check_launch("wgrad_reduce_kernel");
}

// ---- the 7x7 edge layers at full resolution on the bf16 matrix pipe (wgrad_k7.h; plain-bf16 arithmetic)
struct K7Plan {
    int MT, NT, R, A, NW, RB, bpi, grid;
    long long narrow_floats, part_floats;
    size_t lds;
};
static bool k7_plan(int N, int MW, int CN, int H, int W, int final_form, K7Plan& k) {
    if (N < 1 || H < 4 || W < 16 || W > 256 || (W & 15) || (MW != 32 && MW != 64)) return false;
    if (final_form ? CN != 1 : (CN != 1 && CN != 3)) return false;
    k.MT = MW / 32;
    k.NT = (CN * 49 + 31) / 32;
    k.R = final_form ? H + 6 : H;
    k.A = k.R + 6;
    k.NW = W + 16;
    int bpi = std::max(1, (num_cus_w() + N - 1) / N);
    if (bpi > k.R / 2) bpi = std::max(1, k.R / 2);
    k.RB = (k.R + bpi - 1) / bpi;
    k.RB += k.RB & 1;
    k.bpi = (k.R + k.RB - 1) / k.RB;
    k.grid = N * k.bpi;
    k.narrow_floats = round4(((long long)N * CN * k.A * 2 * k.NW + 1) / 2);
    k.part_floats = (long long)k.grid * k.MT * k.NT * 1024;
    const size_t tiles = (size_t)k.MT * 32 * ((W + 8) * 2 + 16) + (size_t)8 * CN * 2 * (k.NW + 8) * 2;
    const size_t red = (size_t)k.MT * k.NT * 16 * 64 * 4;
    k.lds = std::max(tiles, red);
    return true;
}

int32_t ap_wgrad_k7_bf16_ok(int32_t N, int32_t MW, int32_t CN, int32_t H, int32_t W, int32_t final_form) {
    K7Plan k;
    return k7_plan(N, MW, CN, H, W, final_form, k) ? 1 : 0;
}

int64_t ap_wgrad_k7_bf16_workspace_floats(int32_t N, int32_t MW, int32_t CN, int32_t H, int32_t W, int32_t final_form) {
    K7Plan k;
    if (!k7_plan(N, MW, CN, H, W, final_form, k)) return fail(AP_ERR_UNSUPPORTED, "wgrad_k7_bf16: shape not served (see ap_wgrad_k7_bf16_ok)");
    return k.narrow_floats + k.part_floats;
}

int ap_wgrad_k7_bf16(const ap_src* wide, const ap_src* narrow, int32_t N, int32_t H, int32_t W, int32_t final_form, float* workspace,
                     float* dw, ap_stream_t stream_) {
    if (!wide || !wide->data || !narrow || !narrow->data || !workspace || !dw) return fail(AP_ERR_INVALID, "wgrad_k7_bf16: null pointer");
    if ((wide->mean == nullptr) != (wide->rstd == nullptr)) return fail(AP_ERR_INVALID, "wgrad_k7_bf16: mean/rstd mismatch");
    if (narrow->mean || narrow->rstd || narrow->act != AP_ACT_NONE)
        return fail(AP_ERR_UNSUPPORTED, "wgrad_k7_bf16: the narrow operand must be a plain tensor");
    const bool wb16 = (wide->act & 0x100) != 0;                  // ap_src.act bit 8: the tensor holds bf16 values
    const int wact = wide->act & 0xff;
    if (wact < 0 || wact > 2) return fail(AP_ERR_INVALID, "wgrad_k7_bf16: act %d", wact);
    if (!final_form && (wide->mean || wact != AP_ACT_NONE))
        return fail(AP_ERR_UNSUPPORTED, "wgrad_k7_bf16: the stem form takes a plain gradient");
    if (final_form && wb16) return fail(AP_ERR_UNSUPPORTED, "wgrad_k7_bf16: a bf16-stored wide operand is read in the stem form only");
    K7Plan k;
    if (!k7_plan(N, wide->C, narrow->C, H, W, final_form, k))
        return fail(AP_ERR_UNSUPPORTED, "wgrad_k7_bf16: N=%d wide C=%d narrow C=%d %dx%d form %d not served", N, wide->C, narrow->C, H, W, final_form);
    hipStream_t stream = (hipStream_t)stream_;
    K7NarrowParams np;
    np.src = narrow->data; np.dst = reinterpret_cast<unsigned*>(workspace);
    np.N = N; np.CN = narrow->C; np.H = H; np.W = W; np.A = k.A; np.NW = k.NW; np.final_form = final_form;
    const long long ndw = (long long)N * narrow->C * k.A * (k.NW / 2);
    hipLaunchKernelGGL(wgrad_k7_narrow_kernel, dim3((unsigned)std::min<long long>((ndw + 255) / 256, 4096)), dim3(256), 0, stream, np);
    int rc = check_launch("wgrad_k7_narrow_kernel");
    if (rc) return rc;
    WgradK7Params p;
    memset(&p, 0, sizeof(p));
    p.wide = wide->data; p.wmean = wide->mean; p.wrstd = wide->rstd; p.wact = wact;
    p.narrow = reinterpret_cast<const unsigned short*>(workspace);
    p.N = N; p.MW = wide->C; p.CN = narrow->C; p.H = H; p.W = W; p.R = k.R; p.A = k.A; p.NW = k.NW; p.RB = k.RB; p.blocks_per_img = k.bpi;
    p.partial = workspace + k.narrow_floats;
    const void* fn = nullptr;
    if (final_form) fn = reinterpret_cast<const void*>(k.MT == 2 ? &wgrad_k7_kernel<2, 2, true> : &wgrad_k7_kernel<1, 2, true>);
    else if (k.NT == 5 && wb16) fn = reinterpret_cast<const void*>(k.MT == 2 ? &wgrad_k7_kernel<2, 5, false, true> : &wgrad_k7_kernel<1, 5, false, true>);
    else if (k.NT == 5) fn = reinterpret_cast<const void*>(k.MT == 2 ? &wgrad_k7_kernel<2, 5, false> : &wgrad_k7_kernel<1, 5, false>);
    else if (wb16) fn = reinterpret_cast<const void*>(k.MT == 2 ? &wgrad_k7_kernel<2, 2, false, true> : &wgrad_k7_kernel<1, 2, false, true>);
    else fn = reinterpret_cast<const void*>(k.MT == 2 ? &wgrad_k7_kernel<2, 2, false> : &wgrad_k7_kernel<1, 2, false>);
    rc = ensure_wattr(fn);
    if (rc) return rc;
    void* args[] = {&p};
    hipError_t e = hipLaunchKernel(fn, dim3(k.grid), dim3(256), args, k.lds, stream);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "wgrad_k7 launch: %s", hipGetErrorString(e));
    const int total = k.MT * k.NT * 1024;
    hipLaunchKernelGGL(wgrad_k7_reduce_kernel, dim3(total / 64), dim3(256), 0, stream, p.partial, k.grid, total, k.NT, wide->C, narrow->C,
                       final_form, dw);
    return check_launch("wgrad_k7_reduce_kernel");
}

// ---- weight gradient of the PatchGAN's first layer: form 2 of the same kernel (wide = the output gradient [N][64][H/2][W/2])
static bool d0w_plan(int N, int M, int Cin, int H, int W, K7Plan& k) {
    if (N < 1 || M != 64 || (Cin != 1 && Cin != 2) || H < 2 || (H & 1) || W < 32 || W > 512 || (W & 31)) return false;
    const int OH = H / 2, OW = W / 2;                              // OW: a multiple of 16 in 16..256
    // the ring's two new rows per tile are fetched with at most two 16-byte loads per thread (wgrad_k7.h, NQ)
    if (Cin * 4 * ((OW + 16) / 8) * 2 > 512) return false;
    k.MT = 2;
    k.NT = 1;
    k.R = OH;
    k.A = H + 2;
    k.NW = OW + 16;
    int bpi = std::max(1, (num_cus_w() + N - 1) / N);
    if (bpi > k.R / 2) bpi = std::max(1, k.R / 2);
    k.RB = (k.R + bpi - 1) / bpi;
    k.RB += k.RB & 1;
    k.bpi = (k.R + k.RB - 1) / k.RB;
    k.grid = N * k.bpi;
    k.narrow_floats = round4(((long long)N * Cin * k.A * 4 * k.NW + 1) / 2);
    k.part_floats = (long long)k.grid * k.MT * k.NT * 1024;
    const size_t tiles = (size_t)k.MT * 32 * ((OW + 8) * 2 + 16) + (size_t)8 * Cin * 4 * (k.NW + 8) * 2;
    const size_t red = (size_t)k.MT * k.NT * 16 * 64 * 4;
    k.lds = std::max(tiles, red);
    return true;
}

int32_t ap_wgrad_d0_bf16_ok(int32_t N, int32_t M, int32_t Cin, int32_t H, int32_t W) {
    K7Plan k;
    return d0w_plan(N, M, Cin, H, W, k) ? 1 : 0;
}

int64_t ap_wgrad_d0_bf16_workspace_floats(int32_t N, int32_t M, int32_t Cin, int32_t H, int32_t W) {
    K7Plan k;
    if (!d0w_plan(N, M, Cin, H, W, k)) return fail(AP_ERR_UNSUPPORTED, "wgrad_d0_bf16: shape not served (see ap_wgrad_d0_bf16_ok)");
    return k.narrow_floats + k.part_floats;
}

int ap_wgrad_d0_bf16(const float* g, const float* x, int32_t N, int32_t M, int32_t Cin, int32_t H, int32_t W, float* workspace, float* dw,
                     ap_stream_t stream_) {
    if (!g || !x || !workspace || !dw) return fail(AP_ERR_INVALID, "wgrad_d0_bf16: null pointer");
    K7Plan k;
    if (!d0w_plan(N, M, Cin, H, W, k))
        return fail(AP_ERR_UNSUPPORTED, "wgrad_d0_bf16: N=%d %d <- %d channels %dx%d not served (1 | 2 -> 64, even H, W a multiple of 32 up to 512)", N, M, Cin, H, W);
    hipStream_t stream = (hipStream_t)stream_;
    K7NarrowParams np;
    np.src = x; np.dst = reinterpret_cast<unsigned*>(workspace);
    np.N = N; np.CN = Cin; np.H = H; np.W = W; np.A = k.A; np.NW = k.NW; np.final_form = 2;
    const long long ndw = (long long)N * Cin * k.A * 2 * (k.NW / 2);
    hipLaunchKernelGGL(wgrad_d0_narrow_kernel, dim3((unsigned)std::min<long long>((ndw + 255) / 256, 4096)), dim3(256), 0, stream, np);
    int rc = check_launch("wgrad_d0_narrow_kernel");
    if (rc) return rc;
    WgradK7Params p;
    memset(&p, 0, sizeof(p));
    p.wide = g;
    p.narrow = reinterpret_cast<const unsigned short*>(workspace);
    p.N = N; p.MW = M; p.CN = Cin; p.H = H / 2; p.W = W / 2; p.R = k.R; p.A = k.A; p.NW = k.NW; p.RB = k.RB; p.blocks_per_img = k.bpi;
    p.partial = workspace + k.narrow_floats;
    const void* fn = reinterpret_cast<const void*>(&wgrad_k7_kernel<2, 1, 2>);
    rc = ensure_wattr(fn);
    if (rc) return rc;
    void* args[] = {&p};
    hipError_t e = hipLaunchKernel(fn, dim3(k.grid), dim3(256), args, k.lds, stream);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "wgrad_d0 launch: %s", hipGetErrorString(e));
    const int total = k.MT * k.NT * 1024;
    hipLaunchKernelGGL(wgrad_k7_reduce_kernel, dim3(total / 64), dim3(256), 0, stream, p.partial, k.grid, total, k.NT, M, Cin, 0, dw, 16);
    return check_launch("wgrad_k7_reduce_kernel");
}

// ---- data gradient of the last layer on the bf16 matrix pipe (dgrad_k7.h)
int32_t ap_conv_final_dgrad_bf16_ok(int32_t N, int32_t C, int32_t H, int32_t W) {
    return (N >= 1 && (C == 32 || C == 64) && H >= 1 && W >= 16 && W <= 256 && (W & 15) == 0) ? 1 : 0;
}

int64_t ap_conv_final_dgrad_bf16_workspace_floats(int32_t N, int32_t C, int32_t H, int32_t W) {
    if (!ap_conv_final_dgrad_bf16_ok(N, C, H, W)) return fail(AP_ERR_UNSUPPORTED, "conv_final_dgrad_bf16: shape not served");
    return round4(((long long)N * (H + 12) * 2 * (W + 16) + 1) / 2);
}

int ap_conv_final_dgrad_bf16(const float* g, const float* w, int32_t N, int32_t C, int32_t H, int32_t W, float* workspace, float* gp,
                             ap_stream_t stream_) {
    if (!g || !w || !workspace || !gp) return fail(AP_ERR_INVALID, "conv_final_dgrad_bf16: null pointer");
    if (!ap_conv_final_dgrad_bf16_ok(N, C, H, W))
        return fail(AP_ERR_UNSUPPORTED, "conv_final_dgrad_bf16: N=%d C=%d %dx%d not served (C 32 / 64, W a multiple of 16 in 16..256)", N, C, H, W);
    hipStream_t stream = (hipStream_t)stream_;
    const int A = H + 12, NW = W + 16;
    K7NarrowParams np;
    np.src = g; np.dst = reinterpret_cast<unsigned*>(workspace);
    np.N = N; np.CN = 1; np.H = H; np.W = W; np.A = A; np.NW = NW; np.final_form = 1;
    const long long ndw = (long long)N * A * (NW / 2);
    hipLaunchKernelGGL(wgrad_k7_narrow_kernel, dim3((unsigned)std::min<long long>((ndw + 255) / 256, 4096)), dim3(256), 0, stream, np);
    int rc = check_launch("wgrad_k7_narrow_kernel");
    if (rc) return rc;
    DgradK7Params p;
    memset(&p, 0, sizeof(p));
    p.narrow = reinterpret_cast<const unsigned short*>(workspace);
    p.w = w; p.gp = gp; p.N = N; p.C = C; p.H = H; p.W = W; p.HP = H + 6; p.WP = W + 6; p.A = A; p.NW = NW;
    // one workgroup per CU (its LDS row buffers fill one): whole rounds of workgroups where the row count allows
    int bpi = std::max(1, (num_cus_w() + N - 1) / N);
    p.RB = (p.HP + bpi - 1) / bpi;
    p.blocks_per_img = (p.HP + p.RB - 1) / p.RB;
    const size_t lds = (size_t)2 * C * ((p.WP + 7) & ~7) * 4 + (size_t)(kDgradK7Rows + 7) * 2 * (NW + 8) * 2;
    const void* fn = C == 64 ? reinterpret_cast<const void*>(&dgrad_k7_final_kernel<2>) : reinterpret_cast<const void*>(&dgrad_k7_final_kernel<1>);
    rc = ensure_wattr(fn);
    if (rc) return rc;
    void* args[] = {&p};
    hipError_t e = hipLaunchKernel(fn, dim3(N * p.blocks_per_img), dim3(512), args, lds, stream);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "dgrad_k7 launch: %s", hipGetErrorString(e));
    return AP_OK;
}

// ---- data gradient of the PatchGAN's output layer (dgrad_k7.h: dgrad_head_kernel)
int32_t ap_conv_head_dgrad_bf16_ok(int32_t N, int32_t C, int32_t H, int32_t W) {
    return (N >= 1 && C >= 32 && (C & 31) == 0 && H >= 2 && W >= 2 && H * W <= 1156 && (long long)N * (C / 32) < 2147483647LL) ? 1 : 0;
}

int ap_conv_head_dgrad_bf16(const float* g, const float* w, int32_t N, int32_t C, int32_t H, int32_t W, float* gx, ap_stream_t stream_) {
    if (!g || !w || !gx) return fail(AP_ERR_INVALID, "conv_head_dgrad_bf16: null pointer");
    if (!ap_conv_head_dgrad_bf16_ok(N, C, H, W))
        return fail(AP_ERR_UNSUPPORTED, "conv_head_dgrad_bf16: N=%d C=%d %dx%d not served (C a multiple of 32, H W <= 1156)", N, C, H, W);
    DgradHeadParams p;
    p.g = g; p.w = w; p.gx = gx; p.N = N; p.C = C; p.H = H; p.W = W;
    const size_t lds = (size_t)32 * H * W * 4 + (size_t)(H + 3) * 2 * (W + 8) * 2;
    const void* fn = reinterpret_cast<const void*>(&dgrad_head_kernel);
    int rc = ensure_wattr(fn);
    if (rc) return rc;
    void* args[] = {&p};
    hipError_t e = hipLaunchKernel(fn, dim3(N * (C / 32)), dim3(256), args, lds, (hipStream_t)stream_);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "dgrad_head launch: %s", hipGetErrorString(e));
    return AP_OK;
}

// ---- the PatchGAN's first layer as an output stream on the bf16 matrix pipe (conv_d0.h)
int32_t ap_conv_d0_fwd_bf16_ok(int32_t N, int32_t Cin, int32_t Cout, int32_t H, int32_t W) {
    return (N >= 1 && (Cin == 1 || Cin == 2) && Cout == 64 && H >= 2 && (H & 1) == 0 && W >= 8 && W <= 256 && (W & 3) == 0) ? 1 : 0;
}

int ap_conv_d0_fwd_bf16(const float* x, const float* w, const float* bias, int32_t N, int32_t Cin, int32_t Cout, int32_t H, int32_t W,
                        int32_t act, float* y, ap_stream_t stream_) {
    if (!x || !w || !y) return fail(AP_ERR_INVALID, "conv_d0_fwd_bf16: null pointer");
    if (!ap_conv_d0_fwd_bf16_ok(N, Cin, Cout, H, W))
        return fail(AP_ERR_UNSUPPORTED, "conv_d0_fwd_bf16: N=%d %d -> %d channels %dx%d not served (1 | 2 -> 64, even H, W %% 4 == 0, W <= 256)",
                    N, Cin, Cout, H, W);
    if (act < 0 || act > 2) return fail(AP_ERR_UNSUPPORTED, "conv_d0_fwd_bf16: act %d", act);
    ConvD0Params p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.N = N; p.H = H; p.W = W; p.OH = H / 2; p.OW = W / 2; p.act = act;
    p.blocks_per_img = (p.OH + kConvD0Rows - 1) / kConvD0Rows;
    const size_t lds = (size_t)Cin * (2 * kConvD0Rows + 2) * (W + 8) * 2;
    hipStream_t stream = (hipStream_t)stream_;
    if (Cin == 1) hipLaunchKernelGGL(conv_d0_kernel<1>, dim3(N * p.blocks_per_img), dim3(512), lds, stream, p);
    else hipLaunchKernelGGL(conv_d0_kernel<2>, dim3(N * p.blocks_per_img), dim3(512), lds, stream, p);
    return check_launch("conv_d0_kernel");
}

int64_t ap_conv2d_wgrad_workspace_floats(const ap_wgrad_desc* d) {
    WgradPlan pl;
    int rc = make_wgrad_plan(d, pl);
    if (rc) return rc;
    return pl.a_floats + pl.g_floats + pl.part_floats;
}

int ap_conv2d_wgrad_gt_dims(const ap_wgrad_desc* d, int32_t* dims) {
    WgradPlan pl;
    int rc = make_wgrad_plan(d, pl);
    if (rc) return rc;
    if (!dims) return fail(AP_ERR_INVALID, "wgrad_gt_dims: null pointer");
    if (!pl.bf3) return 0;
    dims[0] = pl.GHp; dims[1] = pl.GX8; dims[2] = pl.Mp;
    return 1;
}

}  // extern "C"

// g_t: the M-role operand already in the kernel's layout (ap_conv2d_wgrad_gt_dims), written by its producer
// (ap_instnorm_bwd_split) -- the transposition pass over d->g is skipped and d->g.data is not read
static int wgrad_impl(const ap_wgrad_desc* d, const void* g_t, float* workspace, float* dw, ap_stream_t stream_) {
    WgradPlan pl;
    int rc = make_wgrad_plan(d, pl);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (g_t && !pl.bf3) return fail(AP_ERR_UNSUPPORTED, "wgrad: a prepared operand needs the bf16 matrix plan (ap_conv2d_wgrad_gt_dims)");
    if (!workspace || !dw || (!d->g.data && !g_t)) return fail(AP_ERR_INVALID, "wgrad: null pointer");
    if ((d->g.mean == nullptr) != (d->g.rstd == nullptr)) return fail(AP_ERR_INVALID, "wgrad: g mean/rstd mismatch");
    const bool from_xs = wgrad_xs_route(d, pl);
    for (int s = 0; s < d->nsrc; ++s) {
        if (!d->src[s].data && !from_xs) return fail(AP_ERR_INVALID, "wgrad: segment %d: null data", s);
        if ((d->src[s].mean == nullptr) != (d->src[s].rstd == nullptr))
            return fail(AP_ERR_INVALID, "wgrad: segment %d mean/rstd mismatch", s);
        if ((d->src[s].act & 0x100) && !pl.bf3)      // bit 8: bf16 data (launch_split_transpose)
            return fail(AP_ERR_UNSUPPORTED, "wgrad: segment %d holds bf16 values but the layer is not on the bf16 matrix plan", s);
    }
    if (pl.narrow_cob) {
        WgradNarrowParams p;
        memset(&p, 0, sizeof(p));
        p.src.data = d->src[0].data; p.src.mean = d->src[0].mean; p.src.rstd = d->src[0].rstd;
        p.src.C = d->src[0].C; p.src.act = d->src[0].act;
        p.g = d->g.data;
        p.N = d->N; p.M = d->M; p.GH = d->GH; p.GW = d->GW; p.H = d->H; p.W = d->W; p.pad = d->pad; p.pad_mode = d->pad_mode;
        p.rows_per_block = pl.rows_per_block; p.gwc = pl.gwc; p.gwc_shift = pl.gwc_shift; p.rpi = pl.rpi;
        p.partial = workspace;
        const dim3 grid(pl.P, (d->M + pl.narrow_cob - 1) / pl.narrow_cob);
        const size_t nlds = (size_t)2 * pl.Cin * (2 * pl.rpi * pl.narrow_ppt + 2) * (d->W + 2) * sizeof(float);
        if (d->K == 3) hipLaunchKernelGGL((wgrad_narrow_kernel<3, 1, 1, 8>), grid, dim3(256), 0, stream, p);
        else if (pl.narrow_ppt && pl.Cin == 1) hipLaunchKernelGGL((wgrad_narrow_s2k4_kernel<1, 4, 1>), grid, dim3(256), nlds, stream, p);
        else if (pl.narrow_ppt) hipLaunchKernelGGL((wgrad_narrow_s2k4_kernel<2, 4, 1>), grid, dim3(256), nlds, stream, p);
        else if (pl.Cin == 1) hipLaunchKernelGGL((wgrad_narrow_kernel<4, 2, 1, 4>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((wgrad_narrow_kernel<4, 2, 2, 4>), grid, dim3(256), 0, stream, p);
        rc = check_launch("wgrad_narrow_kernel");
        if (rc) return rc;
        const long long n = (long long)d->M * pl.Q;
        launch_wgrad_reduce(stream, workspace, pl.P, n, dw);
        return check_launch("wgrad_reduce_kernel");
    }
    if (pl.bf3) {
        const WgradBf3Kernel* bk = nullptr;
        for (const auto& k : wgrad_bf3_registry())
            if (k.K == pl.Kb) bk = &k;
        if (!bk) return fail(AP_ERR_UNSUPPORTED, "wgrad: no split-bf16 kernel for k=%d", pl.Kb);
        if (pl.wide && !bk->fn_wide) return fail(AP_ERR_UNSUPPORTED, "wgrad: no 8-wave kernel for k=%d", pl.Kb);
        const bool b16 = d->precision == AP_PRECISION_BF16;
        const void* wfn = pl.wide ? bk->fn_wide : (b16 ? bk->fn1 : bk->fn);
        rc = ensure_wattr(wfn);
        if (rc) return rc;
        uint4* at = reinterpret_cast<uint4*>(workspace);
        uint4* gt = reinterpret_cast<uint4*>(workspace + pl.a_floats);
        float* partial = workspace + pl.a_floats + pl.g_floats;
        if (from_xs) {
            const int parts = d->precision == AP_PRECISION_BF16 ? 1 : 2;
            if (pl.s2d) {
                // the forward pass staged the space-to-depth copy (2 x 2 form) or, where it ran the stride-2 kernel, the plain one
                const void* xs[1] = {d->src_xs_s2d ? d->src_xs_s2d : d->src_xs[0]};
                const int cs[1] = {pl.Cb};
                rc = launch_xs_transpose(xs, cs, 1, d->N, pl.Hb, pl.Wb, 0, AP_PAD_ZERO, pl.Hp, pl.AX8, pl.Cp, parts, at, stream,
                                         d->src_xs_s2d ? 0 : pl.Cin, d->H, d->W);
            } else {
                int cs[kMaxSeg];
                for (int s = 0; s < d->nsrc; ++s) cs[s] = d->src[s].C;
                rc = launch_xs_transpose(d->src_xs, cs, d->nsrc, d->N, d->H, d->W, d->pad, d->pad_mode, pl.Hp, pl.AX8, pl.Cp, parts,
                                         at, stream);
            }
        } else {
            rc = launch_split_transpose(d->src, d->nsrc, d->N, pl.Cb, d->H, d->W, pl.s2d ? 0 : d->pad, d->pad_mode, pl.Hp, pl.AX8,
                                        pl.Cp, at, stream, pl.s2d ? pl.Cin : 0, d->precision == AP_PRECISION_BF16, pl.rows ? pl.Kb : 0);
        }
        if (rc) return rc;
        if (g_t) {
            gt = reinterpret_cast<uint4*>(const_cast<void*>(g_t));
        } else {
            ap_src g = d->g;
            g.C = d->M;
            rc = launch_split_transpose(&g, 1, d->N, d->M, d->GH, d->GW, 0, AP_PAD_ZERO, pl.GHp, pl.GX8, pl.Mp, gt, stream, 0, d->precision == AP_PRECISION_BF16);
            if (rc) return rc;
        }
        WgradBf3Params p;
        memset(&p, 0, sizeof(p));
        p.gt = gt; p.at = at;
        p.N = d->N; p.M = d->M; p.Cin = pl.Cb; p.Q = pl.Cb * (pl.rows ? pl.Kb : pl.Kb * pl.Kb);
        p.GHp = pl.GHp; p.GX8 = pl.GX8; p.Mp = pl.Mp; p.Hp = pl.Hp; p.AX8 = pl.AX8; p.Cp = pl.Cp;
        p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.nstages = pl.nstages; p.P = pl.P;
        p.m_tiles = pl.m_tiles; p.c_tiles = pl.c_tiles;
        p.partial = partial;
        {
#ifdef APAMD_ABLATION
            const char* ab = getenv("APAMD_ABLATE");
            p.ablate = ab ? atoi(ab) : 0;
#endif
        }
        void* args[] = {&p};
        const unsigned nblk = (unsigned)(pl.m_tiles * pl.c_tiles * pl.P);
        hipError_t e = hipLaunchKernel(wfn, dim3(nblk), dim3(pl.wide ? 512 : 256), args,
                                       pl.wide ? bk->lds_bytes_wide : (b16 ? bk->lds_bytes1 : bk->lds_bytes), stream);
        if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "wgrad_bf16x3 launch: %s", hipGetErrorString(e));
        const int T = pl.rows ? pl.Kb : pl.Kb * pl.Kb;
        const long long total = (long long)pl.m_tiles * pl.c_tiles * (pl.wide ? 8 : 4) * T * 1024;
        const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(wgrad_bf3_reduce_kernel, dim3(blocks), dim3(256), 0, stream, partial, pl.P, d->M, pl.Cb, T,
                           pl.c_tiles, total, pl.s2d ? pl.Cin : 0, d->K, dw, pl.wide ? 4 : 2);
        return check_launch("wgrad_bf3_reduce_kernel");
    }
    rc = ensure_wattr(pl.k->fn);
    if (rc) return rc;
    float* a_pad = workspace;
    float* g_pad = workspace + pl.a_floats;
    float* partial = g_pad + pl.g_floats;
    rc = launch_pad(d->src, d->nsrc, d->N, pl.Cin, d->H, d->W, d->pad, d->pad_mode, pl.Hp, pl.Wp, a_pad, stream);
    if (rc) return rc;
    if (!pl.g_direct) {
        ap_src g = d->g;
        g.C = d->M;
        rc = launch_pad(&g, 1, d->N, d->M, d->GH, d->GW, 0, AP_PAD_ZERO, pl.GHp, pl.GWp, g_pad, stream);
        if (rc) return rc;
    }
    WgradKParams p;
    memset(&p, 0, sizeof(p));
    p.g = pl.g_direct ? d->g.data : g_pad;
    p.a = a_pad;
    p.N = d->N; p.M = d->M; p.Cin = pl.Cin; p.Q = pl.Q;
    p.GHp = pl.GHp; p.GWp = pl.GWp; p.Hp = pl.Hp; p.Wp = pl.Wp;
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.nstages = pl.nstages; p.P = pl.P;
    p.m_tiles = pl.m_tiles; p.q_tiles = pl.q_tiles;
    p.partial = partial;
    {
#ifdef APAMD_ABLATION
        const char* ab = getenv("APAMD_ABLATE");
        p.ablate = ab ? atoi(ab) : 0;
#endif
    }
    void* args[] = {&p};
    const unsigned nblk = (unsigned)(pl.m_tiles * pl.q_tiles * pl.P);
    hipError_t e = hipLaunchKernel(pl.k->fn, dim3(nblk), dim3(256), args, pl.k->lds_bytes, stream);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "wgrad_igemm_f32 launch: %s", hipGetErrorString(e));
    const long long n = (long long)d->M * pl.Q;
    launch_wgrad_reduce(stream, partial, pl.P, n, dw);
    return check_launch("wgrad_reduce_kernel");
}

extern "C" {

int ap_conv2d_wgrad(const ap_wgrad_desc* d, float* workspace, float* dw, ap_stream_t stream) {
    return wgrad_impl(d, nullptr, workspace, dw, stream);
}

int ap_conv2d_wgrad_pre(const ap_wgrad_desc* d, const void* g_t, float* workspace, float* dw, ap_stream_t stream) {
    if (!g_t) return fail(AP_ERR_INVALID, "wgrad_pre: null operand");
    return wgrad_impl(d, g_t, workspace, dw, stream);
}

}  // extern "C"
