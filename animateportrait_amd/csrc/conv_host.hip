// conv_host.hip -- host-side planning and the C ABI of the convolution operator.
//
// A descriptor (ap_conv_desc) is turned into one launch (Conv2d) or four launches
// (ConvTranspose2d stride 2 as sub-pixel phases) of conv_igemm_f32; the same plan drives
// the weight packer so that the packed image and the kernel always agree.
#include "common.h"
#include "conv_registry.h"
#include "conv_direct.h"
#include "conv_bf3_registry.h"
#include "conv_prepass.h"
#include "conv_small.h"
#include "conv_head.h"
#include "conv_tsmall.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace apamd {

#ifdef APAMD_VARIANTS
const void* bf3_sb_kernel(size_t* lds_bytes);     // tools/variants/conv_bf3_sb.hip
#endif

static thread_local char g_err[512];
char* last_error_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static const std::vector<ConvKernelInfo>& registry() {
    static std::vector<ConvKernelInfo> v;
    static std::once_flag once;
    std::call_once(once, [] {
        register_s1k0(v);
        register_s1k3(v);
        register_s1k4(v);
        register_s1k7(v);
        register_s2k3(v);
        register_s2k4(v);
    });
    return v;
}

static const ConvKernelInfo* find_kernel(int CI, int S, int K, int CO_TILE) {
    for (const auto& k : registry())
        if (k.CI == CI && k.S == S && k.K == K && k.CO_TILE == CO_TILE) return &k;
    return nullptr;
}

// split-bf16 kernels (conv_bf16x3.h, conv_ph4.h): instantiated per tile family in conv_bf3_inst_*.hip / conv_ph4_inst.hip
static const std::vector<Bf3Kernel>& bf3_registry() {
    static std::vector<Bf3Kernel> v;
    static std::once_flag once;
    std::call_once(once, [] {
        bf3_register_k3_short(v);
        bf3_register_k3_tall(v);
        bf3_register_s2k3(v);
        bf3_register_k4(v);
        bf3_register_row_tall(v);
        bf3_register_row_half(v);
        bf3_register_taps12(v);        // sub-pixel phases of ConvTranspose2d(s=2): 1, 2 or 4 taps (same tile geometry; the plan
        bf3_register_taps4(v);         // keeps the last)
        bf3_register_taps4_half(v);
    });
    return v;
}
// the kernel of one launch: K == 0 kernels are instantiated per tap count
static const Bf3Kernel* bf3_for_taps(const Bf3Kernel* k, int ntaps) {
    if (k->K != 0) return k;
    for (const auto& e : bf3_registry())
        if (e.K == 0 && e.S == k->S && e.TMAX == ntaps && e.TH == k->TH) return &e;
    return nullptr;
}

struct Tap { int ky, kx, ly, lx; };  // (ky,kx): index into the caller's weight; (ly,lx): LDS tile offset

struct Launch {
    std::vector<Tap> taps;
    int OH, OW, dy0, dx0;
    int osy, osx, oy_off, ox_off;
    int tiles_x, tiles_y;
    long long wp_off;      // float offset of this launch's weights in the packed buffer
    int stat_tile_off;
};

struct Plan {
    const ConvKernelInfo* k = nullptr;
    int direct_cop = 0;            // > 0: conv_direct_f32<K, direct_cop> instead of the implicit-GEMM kernel
    bool small = false;            // conv_small_f32: narrow 3x3 layers on the vector ALUs
    bool head = false;             // conv_head_fwd_kernel: one output channel, 4x4, narrow map
    bool tsmall = false;           // conv_tsmall_f32: transposed 4x4 stride 2 with 1..4 output channels
    long long head_w_off = -1;     // >= 0: plain copy of the caller's weights at this offset of the packed image
    bool bf3 = false;              // split-bf16 matrix path
    bool fused_phases = false;     // transposed, split-bf16: the four sub-pixel phases are one launch
    int ph4 = 0;                   // 3 / 4: ... and one TILE (conv_ph4.h)
    bool k0_small = false;         // run-time-tap family: the half-height 4-tap instantiation (two workgroups per CU)
    const Bf3Kernel* bk = nullptr;
    std::vector<Launch> launches;
    int Cin = 0, nchunks = 0, cin_pad = 0, co_tiles = 0;
    int chunk_begin[kMaxSeg] = {0, 0, 0};
    int Hout = 0, Wout = 0, stat_tiles = 0;
    long long packed_floats = 0;
};

static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}

static int num_cus();

static int make_plan(const ap_conv_desc* d, Plan& pl) {
    if (!d) return fail(AP_ERR_INVALID, "null descriptor");
    if (d->nsrc < 1 || d->nsrc > kMaxSeg) return fail(AP_ERR_INVALID, "nsrc=%d out of range", d->nsrc);
    if (d->N < 1 || d->H < 1 || d->W < 1 || d->Cout < 1) return fail(AP_ERR_INVALID, "bad dims");
    // 1 x 7 over the row channels of a 7x7 stem (ap_split_prepass_rows): split-bf16 path only
    const bool rowk = d->KH == 1 && d->KW == 7;
    if (rowk) {
        if (d->transposed || d->stride != 1 || d->pad != 3 || d->nsrc != 1 || d->src[0].C != 32 ||
            d->precision == AP_PRECISION_FP32 || d->w_layout != AP_W_OIHW || d->w_flip)
            return fail(AP_ERR_UNSUPPORTED, "1x7 kernels exist only as the row form of a 7x7 stem (one 32-channel "
                                            "split source, stride 1, pad 3, split-bf16 precision)");
        if (d->pad_mode == AP_PAD_REFLECT && d->pad >= d->W) return fail(AP_ERR_INVALID, "reflection pad %d >= width", d->pad);
    } else if (d->KH != d->KW || d->KH < 1 || d->KH > 7) {
        return fail(AP_ERR_UNSUPPORTED, "kernel %dx%d", d->KH, d->KW);
    }
    const int K = d->KW;
    int S, KT;   // kernel family: stride and dense tap count (0 = run-time taps)
    if (rowk) {
        S = 1;
        KT = K;
        pl.Hout = d->H;
        pl.Wout = d->W;
    } else if (!d->transposed) {
        if (d->stride != 1 && d->stride != 2) return fail(AP_ERR_UNSUPPORTED, "stride %d", d->stride);
        S = d->stride;
        KT = K;
        if (K == 2) {
            // the space-to-depth form of a 4x4 stride-2 layer (ap_split_prepass_s2d): 4 run-time taps, split-bf16 only
            if (d->stride != 1 || d->pad != 0 || d->precision == AP_PRECISION_FP32 || d->w_layout != AP_W_OIHW || d->w_flip)
                return fail(AP_ERR_UNSUPPORTED, "2x2 kernels exist only as the space-to-depth form of a 4x4 stride-2 layer "
                                                "(stride 1, pad 0, split-bf16 precision)");
            KT = 0;
        } else if (K == 1) {
            // 1x1 convolution (channel_mapping of the intrinsic-flow regressor, intrinsic_flow_models/networks.py:23-24):
            // one run-time tap at offset (0, 0)
            if (d->stride != 1 || d->pad != 0) return fail(AP_ERR_UNSUPPORTED, "1x1 kernels: stride 1, pad 0 only");
            KT = 0;
        } else if (K != 3 && K != 4 && K != 7) {
            return fail(AP_ERR_UNSUPPORTED, "kernel size %d (built: 1, 3, 4, 7)", K);
        }
        pl.Hout = (d->H + 2 * d->pad - K) / S + 1;
        pl.Wout = (d->W + 2 * d->pad - K) / S + 1;
        if (d->pad_mode == AP_PAD_REFLECT && (d->pad >= d->H || d->pad >= d->W))
            return fail(AP_ERR_INVALID, "reflection pad %d >= input size", d->pad);
    } else {
        if (d->stride != 2) return fail(AP_ERR_UNSUPPORTED, "transposed conv needs stride 2");
        if (d->pad_mode != AP_PAD_ZERO) return fail(AP_ERR_UNSUPPORTED, "transposed conv with reflection pad");
        S = 1;
        KT = 0;
        pl.Hout = (d->H - 1) * 2 - 2 * d->pad + K + d->output_padding;
        pl.Wout = (d->W - 1) * 2 - 2 * d->pad + K + d->output_padding;
    }
    if (pl.Hout < 1 || pl.Wout < 1) return fail(AP_ERR_INVALID, "empty output");

    int minC = 1 << 30;
    pl.Cin = 0;
    for (int s = 0; s < d->nsrc; ++s) {
        if (d->src[s].C < 1) return fail(AP_ERR_INVALID, "segment %d has C=%d", s, d->src[s].C);
        pl.Cin += d->src[s].C;
        if (d->src[s].C < minC) minC = d->src[s].C;
    }
    // 1..4 output channels, 7x7 'same' convolution: vector-ALU direct kernel (conv_direct.h)
    if (rowk) {
        const bool two_per_cu = true;
        for (const auto& k : bf3_registry())
            if (k.ROW && k.K == K && (!pl.bk || (two_per_cu ? k.TH < pl.bk->TH : k.TH > pl.bk->TH))) pl.bk = &k;
        if (!pl.bk) return fail(AP_ERR_UNSUPPORTED, "no 1x%d row kernel", K);
    }
    // (the PatchGAN's 4x4 pad-1 head, 512 -> 1 on 30 x 30 outputs, was tried here too: 160 workgroups of 128 chunks
    // each run 0.60 ms against 0.39 ms on the MFMA kernel with a 1-of-32 filled tile)
    if (!rowk && !d->transposed && d->stride == 1 && K == 7 && d->pad == 3 && d->Cout <= 4) {
        pl.direct_cop = d->Cout == 1 ? 1 : 4;
        const int ci = 4;
        pl.nchunks = 0;
        for (int s = 0; s < d->nsrc; ++s) {
            pl.chunk_begin[s] = pl.nchunks;
            pl.nchunks += (d->src[s].C + ci - 1) / ci;
        }
        pl.cin_pad = pl.nchunks * ci;
        pl.packed_floats = (long long)pl.cin_pad * K * K * pl.direct_cop;
        pl.stat_tiles = ((pl.Hout + 15) / 16) * ((pl.Wout + 63) / 64);
        return AP_OK;
    }
    // PatchGAN output layer (one output channel, 4x4 pad 1; conv_head.h).  The weight image must not depend on the map
    // size (weights are packed once per layer): layers of this shape carry their OIHW weights behind the regular image,
    // and maps narrow enough for the half-wave row kernel use those.
    const bool head_w = !rowk && !d->transposed && d->stride == 1 && K == 4 && d->Cout == 1 && d->nsrc == 1 && pl.Cin >= 64 &&
                        d->w_layout == AP_W_OIHW && !d->w_flip && d->pad == 1;
    pl.head = head_w && d->W <= kHeadMaxW;
    // ConvTranspose2d(C, 1..4, 4, 2, 1): the data gradient of the PatchGAN's first layer w.r.t. the frame (conv_tsmall.h);
    // same arrangement -- the IOHW weights ride behind the regular image
    pl.tsmall = !rowk && d->transposed && d->stride == 2 && K == 4 && d->pad == 1 && d->output_padding == 0 && d->Cout <= 4 &&
                d->nsrc == 1 && d->w_layout == AP_W_IOHW && !d->w_flip;
    // narrow 3x3 layers (landmark encoder): memory streams, one lane per output pixel (conv_small.h)
    if (!rowk && !d->transposed && K == 3 && d->nsrc == 1 && pl.Cin <= 16 && (d->Cout == 8 || d->Cout == 16) &&
        d->w_layout == AP_W_OIHW && !d->w_flip && !env_int("APAMD_NO_SMALL", 0)) {
        pl.small = true;
        pl.nchunks = 1;
        pl.cin_pad = pl.Cin;
        pl.packed_floats = (long long)d->Cout * pl.Cin * 9;       // the OIHW weights as they are
        pl.stat_tiles = ((pl.Hout + 7) / 8) * ((pl.Wout + 31) / 32);
        return AP_OK;
    }
    // split-bf16 matrix path (conv_bf16x3.h): wide 3x3 / transposed layers when the caller allows ~1e-4 relative error
    // (>= 48 outputs fill most of a 64-cout tile; with >= 128 inputs even a 16-output layer -- the data gradient of a
    // ResnetBlock2 convolution w.r.t. its 16-channel landmark segments -- is 3x faster here than on the fp32 pipe)
    if (!rowk && d->precision != AP_PRECISION_FP32 && (d->Cout >= 48 || (d->Cout >= 16 && pl.Cin >= 128)) && pl.Cin >= 32 &&
        !env_int("APAMD_NO_BF16X3", 0)) {
        bool seg_ok = true;
        for (int s = 0; s < d->nsrc; ++s) seg_ok = seg_ok && d->src[s].C % 16 == 0;
        if (seg_ok && (KT == 0 || K == 3 || (K == 4 && S == 1))) {
            // several tile heights of a family: the tallest one (best operand reuse) unless its tile list leaves
            // most CUs idle (streaming inference at batch 1..4), then the shortest
            const Bf3Kernel* tall = nullptr;
            const Bf3Kernel* small = nullptr;
            for (const auto& k : bf3_registry())
                if (k.S == S && k.K == KT && !k.ROW) {
                    if (!tall || k.TH >= tall->TH) tall = &k;
                    if (!small || k.TH < small->TH) small = &k;
                }
            pl.bk = tall;
            if (KT == 0) {
                // every launch of these plans has 4 taps (2x2 layers, 4x4 phases, fused 3x3 phases of an even output)
                const bool all4 = K == 2 || K == 4 || (d->transposed && K == 3 && pl.Hout % 2 == 0 && pl.Wout % 2 == 0);
                pl.k0_small = all4;
                if (pl.k0_small) pl.bk = small;
                // stride-2 transposed 3x3 / 4x4 with pad 1 and an even output: all four phases in one tile
                if (d->transposed && (K == 3 || K == 4) && d->pad == 1 && d->stride == 2 && pl.Hout % 2 == 0 && pl.Wout % 2 == 0 &&
                    true) {
                    pl.ph4 = K;
                    pl.bk = ph4_kernel(K);
                }
            }
            if (tall && small != tall && KT != 0) {
                const long long tiles = (long long)d->N * ((pl.Hout + tall->TH - 1) / tall->TH) * ((pl.Wout + 31) / 32) *
                                        ((d->Cout + tall->CO_TILE - 1) / tall->CO_TILE);
                // (also for maps no taller than the short tile: the column strip of a reflection-padded data gradient)
                if ((tiles * 2 <= num_cus() || pl.Hout <= small->TH) && !env_int("APAMD_NO_SMALL_TILES", 0)) pl.bk = small;
            }
        }
        if (KT == 0 && K != 1 && K != 2 && K != 3 && K != 4) pl.bk = nullptr;
    }
    if (pl.bk) {
        pl.bf3 = true;
        pl.nchunks = 0;
        for (int s = 0; s < d->nsrc; ++s) {
            pl.chunk_begin[s] = pl.nchunks;
            pl.nchunks += d->src[s].C / 16;
        }
        pl.cin_pad = pl.nchunks * 16;              // channels that exist in the sources
        pl.nchunks = (pl.nchunks + 1) & ~1;        // the kernel's pipeline runs chunk pairs: pad with an all-zero chunk
        pl.co_tiles = (d->Cout + pl.bk->CO_TILE - 1) / pl.bk->CO_TILE;
    } else {
    // tile configuration by output width
    int co_tile = d->Cout >= 96 ? 128 : (d->Cout >= 48 ? 64 : 32);
    co_tile = env_int("APAMD_CONV_COTILE", co_tile);
    // channel chunk: largest of {8,4,2} that does not over-pad the narrowest segment, then shrink
    // until two workgroups fit a CU's LDS
    int ci = minC <= 2 ? 2 : (minC <= 4 ? 4 : 8);
    for (int s = 0; s < d->nsrc; ++s)
        while (ci > 2 && d->src[s].C % ci != 0 && d->src[s].C > ci) ci >>= 1;
    int ntaps_max = d->transposed ? ((K + 1) / 2) * ((K + 1) / 2) : K * K;
    const size_t lds_target = (size_t)env_int("APAMD_CONV_LDS_TARGET", 72 * 1024);
    for (;;) {
        const ConvKernelInfo* k = find_kernel(ci, S, KT, co_tile);
        if (!k) return fail(AP_ERR_UNSUPPORTED, "no kernel for CI=%d S=%d K=%d CO_TILE=%d", ci, S, KT, co_tile);
        int nch = 0;
        for (int s = 0; s < d->nsrc; ++s) nch += (d->src[s].C + ci - 1) / ci;
        size_t bytes = 4 * k->lds_floats(ntaps_max, nch > 1 ? 2 : 1, nch * ci);
        if (bytes <= lds_target || ci == 2) {
            if (bytes > 160 * 1024) return fail(AP_ERR_UNSUPPORTED, "LDS tile of %zu bytes does not fit", bytes);
            pl.k = k;
            break;
        }
        ci >>= 1;
    }
    {
        int forced = env_int("APAMD_CONV_CI", 0);
        if (forced) {
            const ConvKernelInfo* k = find_kernel(forced, S, KT, co_tile);
            if (k) { pl.k = k; ci = forced; }
        }
    }
    pl.nchunks = 0;
    for (int s = 0; s < d->nsrc; ++s) {
        pl.chunk_begin[s] = pl.nchunks;
        pl.nchunks += (d->src[s].C + ci - 1) / ci;
    }
    pl.cin_pad = pl.nchunks * ci;
    pl.co_tiles = (d->Cout + co_tile - 1) / co_tile;
    }

    auto finish = [&](Launch& L) {
        L.tiles_x = (L.OW + 31) / 32;
        const int th = pl.bk ? pl.bk->TH : pl.k->TH;
        const int wf = pl.bk ? pl.bk->wfloats((int)L.taps.size()) : pl.k->wfloats((int)L.taps.size());
        L.tiles_y = (L.OH + th - 1) / th;
        L.wp_off = pl.packed_floats;
        L.stat_tile_off = pl.stat_tiles;
        pl.packed_floats += (long long)pl.co_tiles * pl.nchunks * wf;
        pl.stat_tiles += L.tiles_x * L.tiles_y;
    };
    if (!d->transposed) {
        Launch L;
        for (int ky = 0; ky < (rowk ? 1 : K); ++ky)
            for (int kx = 0; kx < K; ++kx) {
                Tap t;
                t.ly = ky;
                t.lx = kx;
                t.ky = d->w_flip ? K - 1 - ky : ky;
                t.kx = d->w_flip ? K - 1 - kx : kx;
                L.taps.push_back(t);
            }
        L.OH = pl.Hout; L.OW = pl.Wout;
        L.dy0 = rowk ? 0 : -d->pad; L.dx0 = -d->pad;
        L.osy = L.osx = 1; L.oy_off = L.ox_off = 0;
        finish(L);
        pl.launches.push_back(L);
    } else {
        // y[co, 2q+ph] = sum over taps k with (ph + pad - k) even of x[ci, q + (ph + pad - k)/2] * w[ci,co,k]
        // Split-bf16 path, even output size: the four phases share one pixel-tile grid and run as ONE launch whose
        // cout tiles enumerate (phase, cout tile) -- the activation tile is fetched from HBM once instead of four
        // times and the interleaved output lines of the phases meet in the XCD's L2 (conv_bf16x3.h, ConvKParams.nphase).
        const bool fuse = pl.bk != nullptr && pl.Hout % 2 == 0 && pl.Wout % 2 == 0;
        for (int phy = 0; phy < 2; ++phy)
            for (int phx = 0; phx < 2; ++phx) {
                Launch L;
                L.OH = (pl.Hout - phy + 1) / 2;
                L.OW = (pl.Wout - phx + 1) / 2;
                if (L.OH < 1 || L.OW < 1) continue;
                std::vector<std::pair<int, int>> ys, xs;  // (k, shift)
                for (int k = 0; k < K; ++k) {
                    if (((phy + d->pad - k) % 2 + 2) % 2 == 0) ys.push_back({k, (phy + d->pad - k) / 2});
                    if (((phx + d->pad - k) % 2 + 2) % 2 == 0) xs.push_back({k, (phx + d->pad - k) / 2});
                }
                int miny = 1 << 30, minx = 1 << 30, maxy = -(1 << 30), maxx = -(1 << 30);
                for (auto& a : ys) { miny = std::min(miny, a.second); maxy = std::max(maxy, a.second); }
                for (auto& a : xs) { minx = std::min(minx, a.second); maxx = std::max(maxx, a.second); }
                if (ys.empty() || xs.empty() || maxy - miny > 1 || maxx - minx > 1)
                    return fail(AP_ERR_UNSUPPORTED, "transposed conv k=%d pad=%d not decomposable", K, d->pad);
                for (auto& a : ys)
                    for (auto& b : xs) {
                        Tap t;
                        t.ky = d->w_flip ? K - 1 - a.first : a.first;
                        t.kx = d->w_flip ? K - 1 - b.first : b.first;
                        t.ly = a.second - miny;
                        t.lx = b.second - minx;
                        L.taps.push_back(t);
                    }
                L.dy0 = miny; L.dx0 = minx;
                L.osy = L.osx = 2; L.oy_off = phy; L.ox_off = phx;
                if (fuse) {
                    // one launch for the four phases: every phase carries the whole 2 x 2 window (positions it does
                    // not have get zero weights), in the fixed order t = ly * 2 + lx
                    std::vector<Tap> win(4, Tap{-1, -1, 0, 0});
                    for (int t = 0; t < 4; ++t) { win[t].ly = t >> 1; win[t].lx = t & 1; }
                    for (const auto& t : L.taps) { win[t.ly * 2 + t.lx].ky = t.ky; win[t.ly * 2 + t.lx].kx = t.kx; }
                    L.taps = win;
                }
                finish(L);
                pl.launches.push_back(L);
            }
        pl.fused_phases = fuse && pl.launches.size() == 4;
    }
    if (head_w || pl.tsmall) {
        pl.head_w_off = (pl.packed_floats + 3) & ~3LL;            // float4 loads
        pl.packed_floats = pl.head_w_off + (long long)pl.Cin * d->Cout * K * K;
    }
    return AP_OK;
}

// ------------------------------------------------------------------ weight packer
struct PackParams {
    const float* w;
    float* out;
    int Cin, Cout, K, layout;
    int nseg, segC[kMaxSeg], chunk_begin[kMaxSeg];
    int CI, CO_TILE, nchunks, co_tiles, ntaps, wfloats;
    int tap_ky[kMaxTaps], tap_kx[kMaxTaps];
};

__global__ void pack_weights_kernel(const PackParams p) {
    const long long total = (long long)p.co_tiles * p.nchunks * p.wfloats;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int within = (int)(idx % p.wfloats);
        const long long blk = idx / p.wfloats;
        const int chunk = (int)(blk % p.nchunks);
        const int cot = (int)(blk / p.nchunks);
        float v = 0.f;
        if (within < p.ntaps * p.CI * p.CO_TILE) {
            const int col = within % p.CO_TILE;
            const int ci = (within / p.CO_TILE) % p.CI;
            const int t = within / (p.CO_TILE * p.CI);
            const int co = cot * p.CO_TILE + col;
            int s = 0;
            if (p.nseg > 1 && chunk >= p.chunk_begin[1]) s = 1;
            if (p.nseg > 2 && chunk >= p.chunk_begin[2]) s = 2;
            const int cs = (chunk - p.chunk_begin[s]) * p.CI + ci;
            if (cs < p.segC[s] && co < p.Cout) {
                int cin = cs;
                for (int j = 0; j < s; ++j) cin += p.segC[j];
                const int ky = p.tap_ky[t], kx = p.tap_kx[t];
                const long long off = p.layout == AP_W_OIHW
                                          ? (((long long)co * p.Cin + cin) * p.K + ky) * p.K + kx
                                          : (((long long)cin * p.Cout + co) * p.K + ky) * p.K + kx;
                v = p.w[off];
            }
        }
        p.out[idx] = v;
    }
}

struct PackDirectParams {
    const float* w;
    float* out;
    int Cin, Cout, K, layout, flip, COP, CI, nchunks;
    int nseg, segC[kMaxSeg], chunk_begin[kMaxSeg];
};

// out[cin_pad][K][K][COP]
__global__ void pack_direct_kernel(const PackDirectParams p) {
    const int total = p.nchunks * p.CI * p.K * p.K * p.COP;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int co = idx % p.COP;
        int t = idx / p.COP;
        int kx = t % p.K; t /= p.K;
        int ky = t % p.K; t /= p.K;
        const int ci = t % p.CI;
        const int chunk = t / p.CI;
        int s = 0;
        if (p.nseg > 1 && chunk >= p.chunk_begin[1]) s = 1;
        if (p.nseg > 2 && chunk >= p.chunk_begin[2]) s = 2;
        const int cs = (chunk - p.chunk_begin[s]) * p.CI + ci;
        float v = 0.f;
        if (cs < p.segC[s] && co < p.Cout) {
            int cin = cs;
            for (int j = 0; j < s; ++j) cin += p.segC[j];
            if (p.flip) { ky = p.K - 1 - ky; kx = p.K - 1 - kx; }
            const long long off = p.layout == AP_W_OIHW ? (((long long)co * p.Cin + cin) * p.K + ky) * p.K + kx
                                                        : (((long long)cin * p.Cout + co) * p.K + ky) * p.K + kx;
            v = p.w[off];
        }
        p.out[idx] = v;
    }
}

static const void* direct_fn(int K, int cop) {
    if (K == 7 && cop == 1) return reinterpret_cast<const void*>(&conv_direct_f32<DirectCfg<7, 1>>);
    if (K == 7 && cop == 4) return reinterpret_cast<const void*>(&conv_direct_f32<DirectCfg<7, 4>>);
    return nullptr;
}
static size_t direct_lds_bytes(int K, int cop, int nbuf, int cin_pad) {
    if (cop == 1) return 4 * DirectCfg<7, 1>::lds_floats(nbuf, cin_pad);
    return 4 * DirectCfg<7, 4>::lds_floats(nbuf, cin_pad);
}

static std::mutex g_attr_mu;
static std::vector<const void*> g_attr_done;

static int ensure_lds_attr(const void* fn) {
    std::lock_guard<std::mutex> lk(g_attr_mu);
    for (auto f : g_attr_done)
        if (f == fn) return AP_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    g_attr_done.push_back(fn);
    return AP_OK;
}

// compute units of the current device (cached per device ordinal)
static int num_cus() {
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lk(mu);
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// the packer's parameters for one launch of a split-bf16 plan; `v` (nullable): the weight as a strided / derived view
static PackBf3Params bf3_pack_params(const ap_conv_desc* d, const Plan& pl, const Launch& L, const float* weight,
                                     const ap_weight_view* v, float* packed) {
    PackBf3Params p;
    memset(&p, 0, sizeof(p));
    p.w = v ? v->w : weight;
    p.out = reinterpret_cast<unsigned short*>(packed + L.wp_off);
    p.Cin = pl.Cin; p.Cout = d->Cout; p.K = d->KW; p.layout = d->w_layout; p.flip = 0;
    p.KH = d->KH != d->KW ? d->KH : 0;
    p.nseg = d->nsrc;
    for (int s = 0; s < d->nsrc; ++s) { p.segC[s] = d->src[s].C; p.chunk_begin[s] = pl.chunk_begin[s]; }
    p.CO_TILE = pl.bk->CO_TILE; p.nchunks = pl.nchunks; p.co_tiles = pl.co_tiles;
    p.ntaps = (int)L.taps.size();
    for (int t = 0; t < p.ntaps; ++t) { p.tap_ky[t] = L.taps[t].ky; p.tap_kx[t] = L.taps[t].kx; }
    if (v) {
        p.view = 1;
        p.s_co = v->s_co; p.s_ci = v->s_ci; p.s_ky = v->s_ky; p.s_kx = v->s_kx;
        p.s2d_c = v->s2d_c; p.rows_c = v->rows_c; p.ksrc = v->ksrc;
    }
    return p;
}

}  // namespace apamd

using namespace apamd;

extern "C" {

const char* ap_version(void) { return "animateportrait_amd 0.3 (gfx950)"; }
int32_t ap_abi_version(void) { return AP_ABI_VERSION; }
const char* ap_last_error(void) { return last_error_buf(); }

int ap_conv2d_out_size(const ap_conv_desc* d, int32_t* Hout, int32_t* Wout) {
    Plan pl;
    int rc = make_plan(d, pl);
    if (rc) return rc;
    if (Hout) *Hout = pl.Hout;
    if (Wout) *Wout = pl.Wout;
    return AP_OK;
}

int64_t ap_conv2d_packed_floats(const ap_conv_desc* d) {
    Plan pl;
    int rc = make_plan(d, pl);
    return rc ? rc : pl.packed_floats;
}

int32_t ap_conv2d_stat_tiles(const ap_conv_desc* d) {
    Plan pl;
    int rc = make_plan(d, pl);
    return rc ? rc : pl.stat_tiles;
}

int32_t ap_conv2d_wants_presplit(const ap_conv_desc* d) {
    Plan pl;
    int rc = make_plan(d, pl);
    return rc ? rc : (pl.bf3 ? 1 : 0);
}

int64_t ap_split_prepass_bytes(int32_t N, int32_t C, int32_t H, int32_t W) {
    if (N < 1 || C < 8 || (C & 7) || H < 1 || W < 1) return fail(AP_ERR_INVALID, "split_prepass: C=%d must be a multiple of 8", C);
    return (int64_t)N * 2 * (C / 8) * ((int64_t)H * W + 1) * 16;     // one all-zero slot closes every plane
}

int ap_norm_apply_split(const ap_src* src, const float* stat_partials, int32_t tiles, float eps, float* mean_out,
                        float* rstd_out, const ap_src* residual, int32_t N, int32_t H, int32_t W, float* y, void* xs,
                        ap_stream_t stream) {
    return ap_norm_apply_split_ex(src, stat_partials, tiles, eps, mean_out, rstd_out, residual, N, H, W, y, xs, 0, stream);
}

int ap_norm_apply_split_ex(const ap_src* src, const float* stat_partials, int32_t tiles, float eps, float* mean_out,
                           float* rstd_out, const ap_src* residual, int32_t N, int32_t H, int32_t W, float* y, void* xs,
                           int32_t flags, ap_stream_t stream) {
    if (!src || !src->data) return fail(AP_ERR_INVALID, "norm_apply_split: null source");
    if (!y && !xs && !stat_partials) return fail(AP_ERR_INVALID, "norm_apply_split: nothing to produce");
    if (N < 1 || src->C < 8 || (src->C & 7) || H < 1 || W < 1)
        return fail(AP_ERR_INVALID, "norm_apply_split: C=%d must be a multiple of 8", src->C);
    if ((src->mean == nullptr) != (src->rstd == nullptr)) return fail(AP_ERR_INVALID, "norm_apply_split: mean/rstd mismatch");
    if (src->act < 0 || src->act > 2) return fail(AP_ERR_INVALID, "norm_apply_split: act %d", src->act);
    if (stat_partials) {
        if (src->mean) return fail(AP_ERR_INVALID, "norm_apply_split: pass finished statistics OR partial tiles");
        if (tiles < 1 || !mean_out || !rstd_out) return fail(AP_ERR_INVALID, "norm_apply_split: partial tiles need tiles >= 1 and mean_out / rstd_out");
    }
    if (residual) {
        if (!residual->data || residual->C != src->C) return fail(AP_ERR_INVALID, "norm_apply_split: residual mismatch");
        if ((residual->mean == nullptr) != (residual->rstd == nullptr) || residual->act != AP_ACT_NONE)
            return fail(AP_ERR_INVALID, "norm_apply_split: a residual may be normalised but not activated");
    }
    if (N > 65535 || src->C / 8 > 65535) return fail(AP_ERR_UNSUPPORTED, "norm_apply_split: grid too large");
    NormSplitParams p;
    memset(&p, 0, sizeof(p));
    p.x = src->data; p.mean = src->mean; p.rstd = src->rstd;
    p.partials = stat_partials; p.tiles = tiles; p.inv_count = 1.0 / ((double)H * W); p.eps = eps;
    p.mean_out = mean_out; p.rstd_out = rstd_out;
    p.act = src->act;
    if (residual) { p.res = residual->data; p.res_mean = residual->mean; p.res_rstd = residual->rstd; }
    if (flags & 4) {            // the residual is handed over as its split copy (head + tail planes)
        if (!residual || residual->mean) return fail(AP_ERR_INVALID, "norm_apply_split: a split-copy residual must be a plain feature");
        p.res_xs = reinterpret_cast<const uint4*>(residual->data);
        p.res = nullptr;
    }
    p.y = y; p.xs = reinterpret_cast<uint4*>(xs);
    p.heads_only = (flags & 1) ? 1 : 0;
    p.xs_relu = (flags & 2) ? 1 : 0;
    p.N = N; p.C = src->C; p.HW = H * W;
    const bool xb16 = (flags & 8) != 0;       // src->data holds bf16 values (a raw output of ap_conv2d_fwd_bf16out)
    const int res_kind = p.res_xs ? 2 : (p.res ? 1 : 0);
    if (flags & 16) {           // src->data is the channel-octet raw output of ap_conv2d_fwd_octet; the only output is the split copy
        if (y || !xs || res_kind == 1 || xb16)
            return fail(AP_ERR_UNSUPPORTED, "norm_apply_split: a channel-octet source gives a split copy only, with no residual or a split-copy one");
        dim3 grid((p.HW + 511) / 512, src->C / 8, N);
        if (res_kind == 2) hipLaunchKernelGGL(norm_split_oct_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(norm_split_oct_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, p);
        return check_launch("norm_split_oct_kernel");
    }
    auto launch = [&](auto vt, auto xt, auto rt, dim3 grid) {
        hipLaunchKernelGGL((norm_split_kernel<decltype(vt)::value, decltype(xt)::value, decltype(rt)::value>), grid, dim3(256), 0, (hipStream_t)stream, p);
    };
    auto by_res = [&](auto vt, auto xt, dim3 grid) {
        if (res_kind == 0) launch(vt, xt, std::integral_constant<int, 0>{}, grid);
        else if (res_kind == 1) launch(vt, xt, std::integral_constant<int, 1>{}, grid);
        else launch(vt, xt, std::integral_constant<int, 2>{}, grid);
    };
    if ((p.HW & 3) == 0) {
        dim3 grid((p.HW / 4 + 255) / 256, src->C / 8, N);
        if (xb16) by_res(std::integral_constant<int, 4>{}, std::true_type{}, grid);
        else by_res(std::integral_constant<int, 4>{}, std::false_type{}, grid);
    } else {
        dim3 grid((p.HW + 255) / 256, src->C / 8, N);
        if (xb16) by_res(std::integral_constant<int, 1>{}, std::true_type{}, grid);
        else by_res(std::integral_constant<int, 1>{}, std::false_type{}, grid);
    }
    return check_launch("norm_split_kernel");
}

int ap_split_prepass_rows(const ap_src* src, int32_t N, int32_t H, int32_t W, int32_t K, int32_t pad, int32_t pad_mode,
                          void* out, ap_stream_t stream) {
    if (!src || !src->data || !out) return fail(AP_ERR_INVALID, "split_prepass_rows: null pointer");
    if (K != 7 || pad != 3 || src->C < 1 || src->C > 4)
        return fail(AP_ERR_UNSUPPORTED, "split_prepass_rows: built for 7x7 stems with 1..4 input channels (K=%d, C=%d)", K, src->C);
    if (N < 1 || N > 65535 || H < 1 || W < 1) return fail(AP_ERR_INVALID, "split_prepass_rows: bad sizes");
    if ((src->mean == nullptr) != (src->rstd == nullptr)) return fail(AP_ERR_INVALID, "split_prepass_rows: mean/rstd mismatch");
    if (pad_mode == AP_PAD_REFLECT && pad >= H) return fail(AP_ERR_INVALID, "split_prepass_rows: reflection pad %d >= height", pad);
    SplitRowsParams p;
    memset(&p, 0, sizeof(p));
    p.x = src->data; p.mean = src->mean; p.rstd = src->rstd; p.act = src->act;
    p.N = N; p.C = src->C; p.H = H; p.W = W; p.K = K; p.pad = pad; p.pad_mode = pad_mode;
    p.out = reinterpret_cast<uint4*>(out);
    const dim3 grid((H * W + 255) / 256, N);
    switch (src->C) {
        case 1: hipLaunchKernelGGL((split_rows_kernel<7, 1>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
        case 2: hipLaunchKernelGGL((split_rows_kernel<7, 2>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
        case 3: hipLaunchKernelGGL((split_rows_kernel<7, 3>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
        default: hipLaunchKernelGGL((split_rows_kernel<7, 4>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
    }
    return check_launch("split_rows_kernel");
}

int ap_split_prepass_s2d(const ap_src* src, int32_t N, int32_t H, int32_t W, void* out, ap_stream_t stream) {
    if (!src || !src->data || !out) return fail(AP_ERR_INVALID, "split_prepass_s2d: null pointer");
    if (src->C < 8 || src->C % 8 != 0 || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(AP_ERR_UNSUPPORTED, "split_prepass_s2d: needs channels in multiples of 8 and an even map (C=%d, %dx%d)",
                    src->C, H, W);
    if (N < 1 || N > 65535 || src->C / 2 > 65535) return fail(AP_ERR_INVALID, "split_prepass_s2d: bad sizes");
    if ((src->mean == nullptr) != (src->rstd == nullptr)) return fail(AP_ERR_INVALID, "split_prepass_s2d: mean/rstd mismatch");
    if (src->act < 0 || src->act > 2) return fail(AP_ERR_INVALID, "split_prepass_s2d: act %d", src->act);
    SplitS2dParams p;
    memset(&p, 0, sizeof(p));
    p.x = src->data; p.mean = src->mean; p.rstd = src->rstd; p.act = src->act;
    p.N = N; p.C = src->C; p.H = H; p.W = W;
    p.out = reinterpret_cast<uint4*>(out);
    const int HW2 = (H / 2 + 1) * (W / 2 + 1);
    hipLaunchKernelGGL(split_s2d_kernel, dim3((HW2 + 255) / 256, src->C / 2, N), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("split_s2d_kernel");
}

int ap_split_prepass(const ap_src* src, int32_t N, int32_t H, int32_t W, void* out, ap_stream_t stream) {
    if (!src || !out) return fail(AP_ERR_INVALID, "split_prepass: null pointer");
    return ap_norm_apply_split(src, nullptr, 0, 0.f, nullptr, nullptr, nullptr, N, H, W, nullptr, out, stream);
}

int ap_conv2d_kernel_name(const ap_conv_desc* d, char* buf, int32_t buflen) {
    Plan pl;
    int rc = make_plan(d, pl);
    if (rc) return rc;
    if (!buf || buflen < 1) return fail(AP_ERR_INVALID, "kernel_name: bad buffer");
    if (pl.direct_cop) {
        snprintf(buf, buflen, "DirectCfg<%d, %d>", d->KH, pl.direct_cop);
        return AP_OK;
    }
    if (pl.small) {
        snprintf(buf, buflen, "SmallCfg<%d, %d>", d->stride, d->Cout);
        return AP_OK;
    }
    if (pl.head) {
        snprintf(buf, buflen, "HeadCfg<%d>", d->KH);
        return AP_OK;
    }
    if (pl.tsmall) {
        snprintf(buf, buflen, "TSmallCfg<%d>", d->Cout);
        return AP_OK;
    }
    if (pl.bf3) {
        snprintf(buf, buflen, "%s%s", pl.bk->name, d->precision == AP_PRECISION_BF16 ? " bf16" : "");
        return AP_OK;
    }
    snprintf(buf, buflen, "ConvCfg<%d, %d, %d, %d, %d, %d, %d>", pl.k->CI, pl.k->S, pl.k->K, pl.k->WCO, pl.k->MT,
             pl.k->WPX, pl.k->NT);
    return AP_OK;
}

int ap_conv2d_pack_weights(const ap_conv_desc* d, const float* weight, float* packed, ap_stream_t stream) {
    Plan pl;
    int rc = make_plan(d, pl);
    if (rc) return rc;
    if (!weight || !packed) return fail(AP_ERR_INVALID, "null weight/packed pointer");
    if (pl.small) {
        hipError_t e = hipMemcpyAsync(packed, weight, (size_t)pl.packed_floats * sizeof(float), hipMemcpyDeviceToDevice,
                                      (hipStream_t)stream);
        if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "pack (copy) weights: %s", hipGetErrorString(e));
        return AP_OK;
    }
    if (pl.direct_cop) {
        PackDirectParams p;
        memset(&p, 0, sizeof(p));
        p.w = weight; p.out = packed;
        p.Cin = pl.Cin; p.Cout = d->Cout; p.K = d->KH; p.layout = d->w_layout; p.flip = d->w_flip;
        p.COP = pl.direct_cop; p.CI = 4; p.nchunks = pl.nchunks; p.nseg = d->nsrc;
        for (int s = 0; s < d->nsrc; ++s) { p.segC[s] = d->src[s].C; p.chunk_begin[s] = pl.chunk_begin[s]; }
        hipLaunchKernelGGL(pack_direct_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, p);
        return check_launch("pack_direct_kernel");
    }
    if (pl.bf3) {
        for (const auto& L : pl.launches) {
            const PackBf3Params p = bf3_pack_params(d, pl, L, weight, nullptr, packed);
            hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, p);
            rc = check_launch("pack_bf16x3_kernel");
            if (rc) return rc;
        }
        return AP_OK;
    }
    for (const auto& L : pl.launches) {
        PackParams p;
        memset(&p, 0, sizeof(p));
        p.w = weight;
        p.out = packed + L.wp_off;
        p.Cin = pl.Cin; p.Cout = d->Cout; p.K = d->KH; p.layout = d->w_layout;
        p.nseg = d->nsrc;
        for (int s = 0; s < d->nsrc; ++s) { p.segC[s] = d->src[s].C; p.chunk_begin[s] = pl.chunk_begin[s]; }
        p.CI = pl.k->CI; p.CO_TILE = pl.k->CO_TILE; p.nchunks = pl.nchunks; p.co_tiles = pl.co_tiles;
        p.ntaps = (int)L.taps.size();
        p.wfloats = pl.k->wfloats(p.ntaps);
        for (int t = 0; t < p.ntaps; ++t) { p.tap_ky[t] = L.taps[t].ky; p.tap_kx[t] = L.taps[t].kx; }
        const long long total = (long long)p.co_tiles * p.nchunks * p.wfloats;
        int blocks = (int)std::min<long long>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
        rc = check_launch("pack_weights_kernel");
        if (rc) return rc;
    }
    if (pl.head_w_off >= 0) {
        hipError_t e = hipMemcpyAsync(packed + pl.head_w_off, weight, (size_t)pl.Cin * d->Cout * d->KH * d->KW * sizeof(float),
                                      hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "pack (head copy) weights: %s", hipGetErrorString(e));
    }
    return AP_OK;
}

int32_t ap_conv2d_pack_entry_bytes(void) { return (int32_t)sizeof(PackBf3Params); }

int32_t ap_conv2d_pack_entries(const ap_conv_desc* d, const ap_weight_view* v, float* packed, void* entries, int32_t max_entries) {
    Plan pl;
    int rc = make_plan(d, pl);
    if (rc) return rc;
    if (!v || !v->w || !packed) return fail(AP_ERR_INVALID, "pack_entries: null view / packed pointer");
    if (!pl.bf3) return 0;                                   // not a split-bf16 plan: use ap_conv2d_pack_weights
    if ((int)pl.launches.size() > max_entries || !entries) return fail(AP_ERR_INVALID, "pack_entries: room for %d entries, plan has %d", max_entries, (int)pl.launches.size());
    if ((v->s2d_c > 0 || v->rows_c > 0) && v->ksrc < 1) return fail(AP_ERR_INVALID, "pack_entries: derived views need ksrc");
    PackBf3Params* out = reinterpret_cast<PackBf3Params*>(entries);
    int n = 0;
    for (const auto& L : pl.launches) out[n++] = bf3_pack_params(d, pl, L, nullptr, v, packed);
    return n;
}

int ap_conv2d_pack_run(const void* entries_dev, int32_t count, ap_stream_t stream) {
    if (!entries_dev || count < 1 || count > 65535) return fail(AP_ERR_INVALID, "pack_run: bad table");
    hipLaunchKernelGGL(pack_bf16x3_table_kernel, dim3(48, count), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const PackBf3Params*>(entries_dev));
    return check_launch("pack_bf16x3_table_kernel");
}

static int conv2d_fwd_impl(const ap_conv_desc* d, const ap_out_view* view, const float* packed, const float* bias, float* y,
                           float* stat_partials, ap_stream_t stream, bool octet = false, const ap_fused_norm* fn = nullptr,
                           bool ob16 = false);

int ap_conv2d_fwd(const ap_conv_desc* d, const float* packed, const float* bias, float* y,
                  float* stat_partials, ap_stream_t stream) {
    return conv2d_fwd_impl(d, nullptr, packed, bias, y, stat_partials, stream);
}

int ap_conv2d_fwd_view(const ap_conv_desc* d, const ap_out_view* view, const float* packed, const float* bias, float* y,
                       ap_stream_t stream) {
    if (!view) return fail(AP_ERR_INVALID, "conv2d_fwd_view: null view");
    return conv2d_fwd_impl(d, view, packed, bias, y, nullptr, stream);
}

// the channel-octet output form exists in the run-time-tap and row families of the split-bf16 kernel (conv_bf16x3.h epilogue)
static bool octet_plan_ok(const ap_conv_desc* d, const Plan& pl) {
    if (!pl.bf3 || pl.fused_phases || pl.ph4 || pl.launches.size() != 1 || (d->Cout & 7)) return false;
    const Bf3Kernel* kern = bf3_for_taps(pl.bk, (int)pl.launches[0].taps.size());
    return kern && (kern->K == 0 || kern->ROW || (kern->K == 3 && kern->S == 1 && d->precision == AP_PRECISION_BF16X3));
}

int32_t ap_conv2d_octet_ok(const ap_conv_desc* d) {
    Plan pl;
    if (make_plan(d, pl)) return 0;
    return octet_plan_ok(d, pl) ? 1 : 0;
}

int ap_conv2d_fwd_octet(const ap_conv_desc* d, const float* packed, const float* bias, float* y, float* stat_partials,
                        ap_stream_t stream) {
    return conv2d_fwd_impl(d, nullptr, packed, bias, y, stat_partials, stream, true);
}

// the bf16-output form exists for the plain-bf16 instantiations of the dense 3x3 stride-1 tiles (Bf3Cfg::OB16)
static bool ob16_plan_ok(const ap_conv_desc* d, const Plan& pl) {
    if (!pl.bf3 || pl.fused_phases || pl.ph4 || pl.launches.size() != 1 || d->precision != AP_PRECISION_BF16) return false;
    return pl.bk && pl.bk->fn1_ob16 != nullptr;
}

int32_t ap_conv2d_bf16out_ok(const ap_conv_desc* d) {
    Plan pl;
    if (make_plan(d, pl)) return 0;
    return ob16_plan_ok(d, pl) ? 1 : 0;
}

int ap_conv2d_fwd_bf16out(const ap_conv_desc* d, const float* packed, const float* bias, void* y_bf16, float* stat_partials,
                          ap_stream_t stream) {
    return conv2d_fwd_impl(d, nullptr, packed, bias, reinterpret_cast<float*>(y_bf16), stat_partials, stream, false, nullptr, true);
}

int ap_conv2d_fwd_view_bf16out(const ap_conv_desc* d, const ap_out_view* view, const float* packed, const float* bias, void* y_bf16,
                               ap_stream_t stream) {
    if (!view) return fail(AP_ERR_INVALID, "conv2d_fwd_view_bf16out: null view");
    return conv2d_fwd_impl(d, view, packed, bias, reinterpret_cast<float*>(y_bf16), nullptr, stream, false, nullptr, true);
}

// ---- convolution + InstanceNorm in one launch (conv_bf16x3<..., FNORM>): which plans qualify, and is the launch deadlock-free?
// The workgroups holding the tiles of one (image, cout tile) wait for each other inside the kernel, so they must all be running
// at the same time: the persistent grid puts at most one workgroup on a CU and walks the tile list in rounds (workgroup b of an XCD
// takes tiles base + idx, base + idx + step, ...); every group has to lie inside ONE round of ONE XCD's range.
static bool fnorm_plan_ok(const ap_conv_desc* d, const Plan& pl) {
    if (!pl.bf3 || pl.fused_phases || pl.ph4 || pl.launches.size() != 1 || d->transposed || d->precision != AP_PRECISION_BF16X3 ||
        env_int("APAMD_NO_FUSED_NORM", 0))
        return false;
    const Bf3Kernel* k = pl.bk;
    if (!k || k->S != 1 || k->K != 3 || k->ROW || k->TH != 16 || k->CO_TILE != 64) return false;
    if ((d->Cout % 64) || (pl.Hout % 16) || (pl.Wout % 32)) return false;           // whole tiles only: every lane holds real pixels
    if ((pl.Hout / 16) * (pl.Wout / 32) > 32) return false;                          // the exchange table holds 32 tiles per plane
    const Launch& L = pl.launches[0];
    const long long per_image = (long long)L.tiles_y * L.tiles_x * pl.co_tiles, ntl = d->N * per_image;
    const long long G = ntl < num_cus() ? ntl : num_cus();
    const long long nx = G < 8 ? G : 8, q = ntl / nx, r = ntl % nx;
    for (int n = 0; n < d->N; ++n) {
        // the tiles of image n are contiguous in the list ((n, ty, tx) major, cout tile fastest): all of them in one round
        const long long t0 = n * per_image, t1 = t0 + per_image - 1;
        auto where = [&](long long t, long long& xcd, long long& round) {
            for (xcd = 0; xcd < nx; ++xcd) {
                const long long base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1 : 0);
                if (t >= base && t < base + cnt) {
                    const long long step = (G - xcd + nx - 1) / nx;
                    round = (t - base) / step;
                    return;
                }
            }
        };
        long long x0 = -1, r0 = -1, x1 = -2, r1 = -2;
        where(t0, x0, r0);
        where(t1, x1, r1);
        if (x0 != x1 || r0 != r1) return false;
    }
    return true;
}

int32_t ap_conv2d_fused_norm_ok(const ap_conv_desc* d) {
    Plan pl;
    if (make_plan(d, pl)) return 0;
    return fnorm_plan_ok(d, pl) ? 1 : 0;
}

int32_t ap_conv2d_fused_norm_counters(const ap_conv_desc* d) {
    Plan pl;
    int rc = make_plan(d, pl);
    return rc ? rc : d->N * pl.co_tiles * 2 + 1;        // + the launch's error flag (set when a workgroup gave up waiting)
}

int ap_conv2d_fwd_norm(const ap_conv_desc* d, const float* packed, const ap_fused_norm* fn, ap_stream_t stream) {
    if (!fn || !fn->partials || !fn->counters || !fn->mean || !fn->rstd) return fail(AP_ERR_INVALID, "conv2d_fwd_norm: null workspace");
    if (!fn->xs && !fn->y_oct) return fail(AP_ERR_INVALID, "conv2d_fwd_norm: neither a split nor a channel-octet output");
    if (fn->res_oct && fn->res_nchw) return fail(AP_ERR_INVALID, "conv2d_fwd_norm: two residuals");
    if (fn->act < 0 || fn->act > 2) return fail(AP_ERR_INVALID, "conv2d_fwd_norm: act %d", fn->act);
    return conv2d_fwd_impl(d, nullptr, packed, nullptr, fn->partials, fn->partials, stream, false, fn);
}

static int conv2d_fwd_impl(const ap_conv_desc* d, const ap_out_view* view, const float* packed, const float* bias, float* y,
                           float* stat_partials, ap_stream_t stream, bool octet, const ap_fused_norm* fn, bool ob16) {
    Plan pl;
    int rc = make_plan(d, pl);
    if (rc) return rc;
    if (ob16 && (!ob16_plan_ok(d, pl) || octet || fn))
        return fail(AP_ERR_UNSUPPORTED, "conv2d_fwd_bf16out: only the plain-bf16 dense 3x3 stride-1 layers store bf16 (ap_conv2d_bf16out_ok)");
    if (fn && !fnorm_plan_ok(d, pl))
        return fail(AP_ERR_UNSUPPORTED, "conv2d_fwd_norm: this layer / shape cannot normalise in its epilogue (ap_conv2d_fused_norm_ok)");
    if (!packed || !y) return fail(AP_ERR_INVALID, "null packed/y pointer");
    if (octet && !octet_plan_ok(d, pl))
        return fail(AP_ERR_UNSUPPORTED, "conv2d_fwd_octet: only single-launch split-bf16 layers of the run-time-tap / row kernel "
                                        "families with Cout %% 8 == 0 write the channel-octet layout (ap_conv2d_octet_ok)");
    if (view) {
        if (!pl.bf3 || pl.fused_phases || pl.launches.size() != 1)
            return fail(AP_ERR_UNSUPPORTED, "conv2d_fwd_view: only single-launch split-bf16 plans take an output window");
        if (view->OH < 1 || view->OW < 1 || view->OH > pl.Hout || view->OW > pl.Wout)
            return fail(AP_ERR_INVALID, "conv2d_fwd_view: window %dx%d outside the %dx%d output", view->OH, view->OW, pl.Hout, pl.Wout);
    }
    if (d->presplit && !pl.bf3) return fail(AP_ERR_INVALID, "desc.presplit set for a layer that does not take split sources");
    for (int s = 0; s < d->nsrc; ++s) {
        if (!d->src[s].data) return fail(AP_ERR_INVALID, "segment %d: null data", s);
        if ((d->src[s].mean == nullptr) != (d->src[s].rstd == nullptr))
            return fail(AP_ERR_INVALID, "segment %d: mean and rstd must be given together", s);
        if (d->src[s].act < 0 || d->src[s].act > 2) return fail(AP_ERR_INVALID, "segment %d: act %d", s, d->src[s].act);
    }
    if (pl.tsmall && !stat_partials) {
        TSmallParams p;
        memset(&p, 0, sizeof(p));
        p.src.data = d->src[0].data; p.src.mean = d->src[0].mean; p.src.rstd = d->src[0].rstd;
        p.src.C = d->src[0].C; p.src.act = d->src[0].act;
        p.N = d->N; p.C = pl.Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
        p.w = packed + pl.head_w_off; p.bias = bias; p.act = d->act; p.y = y;
        if (d->N > 65535 || (d->H + 7) / 8 > 65535) return fail(AP_ERR_UNSUPPORTED, "conv_tsmall: grid too large");
        const dim3 grid((d->W + 31) / 32, (d->H + 7) / 8, d->N);
        if (d->Cout == 1) hipLaunchKernelGGL((conv_tsmall_f32<1>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (d->Cout == 2) hipLaunchKernelGGL((conv_tsmall_f32<2>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv_tsmall_f32<4>), grid, dim3(256), 0, (hipStream_t)stream, p);
        return check_launch("conv_tsmall_f32");
    }
    if (pl.head && !stat_partials) {      // (with a statistics epilogue wanted the layer stays on the general kernel)
        HeadParams p;
        memset(&p, 0, sizeof(p));
        p.src.data = d->src[0].data; p.src.mean = d->src[0].mean; p.src.rstd = d->src[0].rstd;
        p.src.C = d->src[0].C; p.src.act = d->src[0].act;
        p.N = d->N; p.C = pl.Cin; p.H = d->H; p.W = d->W; p.OH = pl.Hout; p.OW = pl.Wout;
        p.w = packed + pl.head_w_off; p.bias = bias; p.act = d->act; p.y = y;
        // Rows per workgroup.  The kernel is bound by its vector ALUs (normalisation, activation, three wave shifts and 16 FMAs per
        // staged row element, on RB + 3 rows for RB outputs) and a launch is a few hundred 1024-thread workgroups: what counts is
        // the number of workgroup ROUNDS on the CUs times the rows a workgroup stages.  (Round 6; before: 3-row bands unless taller
        // ones still gave >= 200 workgroups -- 320 workgroups = two rounds at 2B = 32, 160 = 5/8 of the chip at B = 16.)
        const int cus = num_cus();
        int best = 3;
        long long best_cost = -1;
        for (int rb : {2, 3, 4, 5, 6, 10}) {
            const long long wgs = (long long)d->N * ((pl.Hout + rb - 1) / rb);
            const long long cost = ((wgs + cus - 1) / cus) * (rb + 3);
            if (best_cost < 0 || cost < best_cost) { best = rb; best_cost = cost; }
        }
        const dim3 hg(d->N, (pl.Hout + best - 1) / best);
        switch (best) {
            case 2: hipLaunchKernelGGL(conv_head_fwd_kernel<2>, hg, dim3(1024), 0, (hipStream_t)stream, p); break;
            case 4: hipLaunchKernelGGL(conv_head_fwd_kernel<4>, hg, dim3(1024), 0, (hipStream_t)stream, p); break;
            case 5: hipLaunchKernelGGL(conv_head_fwd_kernel<5>, hg, dim3(1024), 0, (hipStream_t)stream, p); break;
            case 6: hipLaunchKernelGGL(conv_head_fwd_kernel<6>, hg, dim3(1024), 0, (hipStream_t)stream, p); break;
            case 10: hipLaunchKernelGGL(conv_head_fwd_kernel<10>, hg, dim3(1024), 0, (hipStream_t)stream, p); break;
            default: hipLaunchKernelGGL(conv_head_fwd_kernel<3>, hg, dim3(1024), 0, (hipStream_t)stream, p); break;
        }
        return check_launch("conv_head_fwd_kernel");
    }
    if (pl.small) {
        SmallKParams p;
        memset(&p, 0, sizeof(p));
        p.src.data = d->src[0].data; p.src.mean = d->src[0].mean; p.src.rstd = d->src[0].rstd;
        p.src.C = d->src[0].C; p.src.act = d->src[0].act;
        p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = pl.Cin; p.Cout = d->Cout; p.OH = pl.Hout; p.OW = pl.Wout;
        p.pad = d->pad; p.pad_mode = d->pad_mode;
        p.y = y; p.w = packed; p.bias = bias; p.act = d->act;
        p.stats = stat_partials; p.stat_tiles = pl.stat_tiles;
        p.tiles_x = (pl.Wout + 31) / 32; p.tiles_y = (pl.Hout + 7) / 8;
        const dim3 grid((unsigned)((long long)d->N * p.tiles_y * p.tiles_x));
        if (d->stride == 1 && d->Cout == 8) hipLaunchKernelGGL((conv_small_f32<SmallCfg<1, 8>>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (d->stride == 1) hipLaunchKernelGGL((conv_small_f32<SmallCfg<1, 16>>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (d->Cout == 8) hipLaunchKernelGGL((conv_small_gather_f32<SmallCfg<2, 8>>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv_small_gather_f32<SmallCfg<2, 16>>), grid, dim3(256), 0, (hipStream_t)stream, p);
        return check_launch("conv_small_f32");
    }
    if (pl.bf3) {
        if (!d->presplit)
            return fail(AP_ERR_INVALID, "this layer runs on the split-bf16 path: pass sources prepared by "
                                        "ap_split_prepass and set desc.presplit (see ap_conv2d_wants_presplit)");
        for (const auto& L : pl.launches) {
            const Bf3Kernel* kern = pl.ph4 ? pl.bk : bf3_for_taps(pl.bk, (int)L.taps.size());
            if (!kern) return fail(AP_ERR_UNSUPPORTED, "no split-bf16 kernel for a phase with %d taps", (int)L.taps.size());
            const void* kfn = ob16 ? kern->fn1_ob16 : kern->kernel(d->precision);
            rc = ensure_lds_attr(kfn);
            if (rc) return rc;
            ConvKParams p;
            memset(&p, 0, sizeof(p));
            p.nseg = d->nsrc;
            for (int s = 0; s < d->nsrc; ++s) {
                p.seg[s].data = d->src[s].data; p.seg[s].C = d->src[s].C; p.seg[s].chunk_begin = pl.chunk_begin[s];
            }
            p.N = d->N; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
            p.OH = L.OH; p.OW = L.OW; p.dy0 = L.dy0; p.dx0 = L.dx0;
            p.pad_mode = d->pad_mode;
            p.y = y;
            p.o_nstride = (long long)d->Cout * pl.Hout * pl.Wout;
            p.o_cstride = (long long)pl.Hout * pl.Wout;
            p.o_rstride = pl.Wout;
            p.osy = L.osy; p.osx = L.osx; p.oy_off = L.oy_off; p.ox_off = L.ox_off;
            p.wp = packed + L.wp_off; p.bias = bias; p.act = d->act;
            p.stats = stat_partials; p.stat_tiles = pl.stat_tiles; p.stat_tile_off = L.stat_tile_off;
            p.ntaps = (int)L.taps.size(); p.nchunks = pl.nchunks;
            p.tiles_x = L.tiles_x; p.tiles_y = L.tiles_y; p.co_tiles = pl.co_tiles;
            p.cin_pad = pl.cin_pad;
            p.wfloats = pl.bk->wfloats(p.ntaps);
#ifdef APAMD_ABLATION
            p.ablate = env_int("APAMD_ABLATE", 0);
#endif
            p.o_octet = octet ? 1 : 0;
            if (view) {
                // output window: a sub-grid of the output, stored with the caller's strides (ap_out_view)
                p.OH = view->OH; p.OW = view->OW;
                p.tiles_x = (view->OW + 31) / 32;
                p.tiles_y = (view->OH + pl.bk->TH - 1) / pl.bk->TH;
                p.o_nstride = view->nstride; p.o_cstride = view->cstride; p.o_rstride = view->rstride;
                p.osy = 1; p.oy_off = view->y_off;
                p.osx = view->xstride; p.ox_off = view->x_off * view->xstride;
            }
            if (pl.fused_phases) {
                // this launch (the geometry of phase 0, shared by all) covers the four phases: their weight blocks
                // follow each other in the packed image, so the virtual cout tile indexes them directly
                p.nphase = 4;
                p.co_tiles_phase = pl.co_tiles;
                p.co_tiles = 4 * pl.co_tiles;
                for (int ph = 0; ph < 4; ++ph) {
                    const Launch& Lp = pl.launches[ph];
                    p.ph_dy0[ph] = Lp.dy0; p.ph_dx0[ph] = Lp.dx0; p.ph_oy[ph] = Lp.oy_off; p.ph_ox[ph] = Lp.ox_off;
                    p.ph_stat[ph] = Lp.stat_tile_off;
                    p.ph_tapmask[ph] = 0;
                    for (size_t t = 0; t < Lp.taps.size() && t < 4; ++t)
                        if (Lp.taps[t].ky >= 0) p.ph_tapmask[ph] |= 1u << t;
                }
                if (pl.ph4) {
                    // conv_ph4 walks cout tiles only and derives the phase geometry from K: check the plan agrees
                    if (view) return fail(AP_ERR_UNSUPPORTED, "conv_ph4: output views are not supported");
                    // (the kernel folds the phase into 32-bit per-lane byte offsets of the packed weights)
                    if (3LL * pl.co_tiles * pl.nchunks * p.wfloats * 4 >= (1LL << 31))
                        return fail(AP_ERR_UNSUPPORTED, "conv_ph4: packed phase blocks of %d cout tiles x %d chunks are too large", pl.co_tiles, pl.nchunks);
                    p.co_tiles = pl.co_tiles;
                    const int base = pl.ph4 == 3 ? 0 : -1;
                    for (int ph = 0; ph < 4; ++ph) {
                        const Launch& Lp = pl.launches[ph];
                        const unsigned want = ((pl.ph4 == 3 && !(ph >> 1)) ? 1u : 3u) * 1u;          // taps along y
                        const unsigned wanx = (pl.ph4 == 3 && !(ph & 1)) ? 1u : 3u;
                        unsigned m = 0;
                        for (int t = 0; t < 4; ++t) if (((want >> (t >> 1)) & 1u) && ((wanx >> (t & 1)) & 1u)) m |= 1u << t;
                        const int oy = pl.ph4 == 3 ? 0 : (ph >> 1), ox = pl.ph4 == 3 ? 0 : (ph & 1);
                        unsigned have = 0;
                        for (size_t t = 0; t < Lp.taps.size() && t < 4; ++t) if (Lp.taps[t].ky >= 0) have |= 1u << t;
                        if (have != m || Lp.dy0 != base + oy || Lp.dx0 != base + ox || Lp.oy_off != (ph >> 1) || Lp.ox_off != (ph & 1) ||
                            Lp.osy != 2 || Lp.osx != 2 || Lp.OH != pl.launches[0].OH || Lp.OW != pl.launches[0].OW)
                            return fail(AP_ERR_UNSUPPORTED, "conv_ph4: phase %d geometry (taps %x/%x, origin %d,%d)", ph, have, m, Lp.dy0, Lp.dx0);
                    }
                }
            }
            p.tap_bits = 0;
            if (d->s2d_k == 3 && !d->transposed && pl.bk->K == 0 && d->KW == 2 && d->nsrc == 1 && d->src[0].C % 64 == 0) {
                // space-to-depth form of a 3x3 stride-2 layer: input phase (ry, rx) = 16-channel chunks [r C/16, (r+1) C/16)
                p.s2d_div = d->src[0].C / 64;
                for (int r = 0; r < 4; ++r) {
                    p.s2d_mask[r] = 0;
                    for (int t = 0; t < 4; ++t)
                        if (2 * (t >> 1) + (r >> 1) <= 2 && 2 * (t & 1) + (r & 1) <= 2) p.s2d_mask[r] |= 1u << t;
                }
            }
            if (pl.bk->K == 0) {
                if (p.ntaps > 4) return fail(AP_ERR_UNSUPPORTED, "phase with %d taps", p.ntaps);
                for (int t = 0; t < p.ntaps; ++t)
                    p.tap_bits |= (unsigned)((L.taps[t].ly & 1) | ((L.taps[t].lx & 1) << 1)) << (2 * t);
            }
            if (p.s2d_div > 0 && (p.s2d_div & 1) == 0 && p.nchunks == 4 * p.s2d_div && kern->kernel_s2d3(d->precision)) {
                // even chunk count per input phase: the instantiation with compile-time tap sets (no fragment reads for absent taps)
                kfn = kern->kernel_s2d3(d->precision);
                rc = ensure_lds_attr(kfn);
                if (rc) return rc;
            }
            if (fn) {
                kfn = bf3_fnorm_kernel();
                rc = ensure_lds_attr(kfn);
                if (rc) return rc;
                p.fn_act = fn->act; p.fn_eps = fn->eps; p.fn_inv_count = 1.0 / ((double)pl.Hout * pl.Wout);
                p.fn_res_oct = fn->res_oct; p.fn_res_nchw = fn->res_nchw; p.fn_y_oct = fn->y_oct; p.fn_xs = fn->xs;
                p.fn_mean = fn->mean; p.fn_rstd = fn->rstd; p.fn_counters = fn->counters;
                p.fn_debug = env_int("APAMD_FNORM_DEBUG", 0);
            }
            size_t lds = kern->lds(d->precision, p.ntaps);
            bool sb = false;
#ifdef APAMD_VARIANTS
            if (!fn && !view && !octet && kern->K == 3 && kern->S == 1 && kern->TH == 16 && !kern->ROW && d->precision != AP_PRECISION_BF16 &&
                p.osx == 1 && p.osy == 1 && p.oy_off == 0 && p.ox_off == 0 && env_int("APAMD_CONV_SB", 0)) {
                // rejected experiment kept for A/B (tools/variants/conv_bf16x3_sb.h, `make variants`): one LDS stage per
                // workgroup, two workgroups per CU
                kfn = bf3_sb_kernel(&lds);
                rc = ensure_lds_attr(kfn);
                if (rc) return rc;
                p.fn_debug = env_int("APAMD_CONV_SB_SKEW", 0);
                sb = true;
            }
#endif
#ifdef APAMD_ABLATION
            // cycle account (tools/cycle_account.py): APAMD_STAMP_BUF = device address of the stamp buffer, APAMD_STAMP_WORDS =
            // dwords per wave; the stamps live in LDS behind the stage buffers, so only kernels that leave that room take them
            if (const char* sb_ = getenv("APAMD_STAMP_BUF")) {
                const int words = env_int("APAMD_STAMP_WORDS", 512);
                if (!pl.ph4 && lds + (size_t)16 * words <= 160 * 1024) {
                    p.stamps = reinterpret_cast<unsigned*>(strtoull(sb_, nullptr, 0));
                    p.stamp_lds_off = (int)lds;
                    p.stamp_words = words;
                    lds += (size_t)16 * words;
                }
            }
#endif
            if (lds > 160 * 1024) return fail(AP_ERR_UNSUPPORTED, "bf16x3 LDS tile of %zu bytes does not fit", lds);
            if (p.nchunks < 2) return fail(AP_ERR_UNSUPPORTED, "bf16x3 pipeline needs >= 32 input channels");
            if (((long long)d->H * d->W + 1) * 32 >= (1LL << 31))   // per-lane DMA offsets span two channel-group planes
                return fail(AP_ERR_UNSUPPORTED, "split-bf16 path: %d x %d planes are too large", d->H, d->W);
            // persistent workgroups: one per CU (the two LDS stages fill a CU), each walks its share of the tiles
            long long nblk = (long long)d->N * p.tiles_y * p.tiles_x * p.co_tiles;
            // (two per CU when two stage sets fit its 160 KB of LDS)
            int cus = num_cus() * ((2 * lds <= 160 * 1024 || sb) ? 2 : 1);
            if (const int forced = env_int("APAMD_BF3_BLOCKS", 0)) {      // tuning / test knob: never silent
                static bool told = false;
                if (!told) fprintf(stderr, "libapamd: APAMD_BF3_BLOCKS=%d overrides the persistent workgroup count\n", forced);
                told = true;
                cus = forced;
            }
            if (nblk > cus) nblk = cus;
            void* args[] = {&p};
            hipError_t e = hipLaunchKernel(kfn, dim3((unsigned)nblk), dim3(256), args, lds, (hipStream_t)stream);
            if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "conv_bf16x3 launch: %s", hipGetErrorString(e));
            if (pl.fused_phases) break;
        }
        return AP_OK;
    }
    rc = ensure_lds_attr(pl.direct_cop ? direct_fn(d->KH, pl.direct_cop) : pl.k->fn);
    if (rc) return rc;
    if (pl.direct_cop) {
        DirectKParams p;
        memset(&p, 0, sizeof(p));
        p.nseg = d->nsrc;
        for (int s = 0; s < d->nsrc; ++s) {
            p.seg[s].data = d->src[s].data; p.seg[s].mean = d->src[s].mean; p.seg[s].rstd = d->src[s].rstd;
            p.seg[s].C = d->src[s].C; p.seg[s].act = d->src[s].act; p.seg[s].chunk_begin = pl.chunk_begin[s];
        }
        p.N = d->N; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.OH = pl.Hout; p.OW = pl.Wout;
        p.pad = d->pad; p.pad_mode = d->pad_mode;
        p.y = y; p.wp = packed; p.bias = bias; p.act = d->act;
        p.stats = stat_partials; p.stat_tiles = pl.stat_tiles;
        p.nchunks = pl.nchunks; p.cin_pad = pl.cin_pad;
        p.tiles_x = (pl.Wout + 63) / 64; p.tiles_y = (pl.Hout + 15) / 16;
        const size_t lds = direct_lds_bytes(d->KH, pl.direct_cop, p.nchunks > 1 ? 2 : 1, p.cin_pad);
        void* args[] = {&p};
        hipError_t e = hipLaunchKernel(direct_fn(d->KH, pl.direct_cop), dim3((unsigned)(d->N * p.tiles_y * p.tiles_x)),
                                       dim3(256), args, lds, (hipStream_t)stream);
        if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "conv_direct_f32 launch: %s", hipGetErrorString(e));
        return AP_OK;
    }
    for (const auto& L : pl.launches) {
        ConvKParams p;
        memset(&p, 0, sizeof(p));
        p.nseg = d->nsrc;
        for (int s = 0; s < d->nsrc; ++s) {
            p.seg[s].data = d->src[s].data;
            p.seg[s].mean = d->src[s].mean;
            p.seg[s].rstd = d->src[s].rstd;
            p.seg[s].C = d->src[s].C;
            p.seg[s].act = d->src[s].act;
            p.seg[s].chunk_begin = pl.chunk_begin[s];
        }
        p.N = d->N; p.H = d->H; p.W = d->W; p.Cout = d->Cout;
        p.OH = L.OH; p.OW = L.OW; p.dy0 = L.dy0; p.dx0 = L.dx0;
        p.pad_mode = d->pad_mode;
        p.y = y;
        p.o_nstride = (long long)d->Cout * pl.Hout * pl.Wout;
        p.o_cstride = (long long)pl.Hout * pl.Wout;
        p.o_rstride = pl.Wout;
        p.osy = L.osy; p.osx = L.osx; p.oy_off = L.oy_off; p.ox_off = L.ox_off;
        p.wp = packed + L.wp_off;
        p.bias = bias;
        p.act = d->act;
        p.stats = stat_partials;
        p.stat_tiles = pl.stat_tiles;
        p.stat_tile_off = L.stat_tile_off;
        p.ntaps = (int)L.taps.size();
        p.nchunks = pl.nchunks;
        p.tiles_x = L.tiles_x; p.tiles_y = L.tiles_y; p.co_tiles = pl.co_tiles;
        p.cin_pad = pl.cin_pad;
        p.wfloats = pl.k->wfloats(p.ntaps);
#ifdef APAMD_ABLATION
        p.ablate = env_int("APAMD_ABLATE", 0);
#endif
        p.tap_bits = 0;
        if (pl.k->K == 0) {
            if (p.ntaps > 4) return fail(AP_ERR_UNSUPPORTED, "phase with %d taps", p.ntaps);
            for (int t = 0; t < p.ntaps; ++t)
                p.tap_bits |= (unsigned)((L.taps[t].ly & 1) | ((L.taps[t].lx & 1) << 1)) << (2 * t);
        }
        const size_t lds = 4 * pl.k->lds_floats(p.ntaps, p.nchunks > 1 ? 2 : 1, p.cin_pad);
        const long long nblk = (long long)d->N * L.tiles_y * L.tiles_x * pl.co_tiles;
        if (nblk > 0x7fffffffLL) return fail(AP_ERR_UNSUPPORTED, "grid too large");
        void* args[] = {&p};
        hipError_t e = hipLaunchKernel(pl.k->fn, dim3((unsigned)nblk), dim3(256), args, lds, (hipStream_t)stream);
        if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "conv_igemm_f32 launch: %s", hipGetErrorString(e));
    }
    return AP_OK;
}

}  // extern "C"
