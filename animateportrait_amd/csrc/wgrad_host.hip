// wgrad_host.hip -- C ABI of the weight-gradient kernel (wgrad_igemm.h).
#include "common.h"
#include "wgrad_igemm.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace apamd {

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
        wk<WgradCfg<1, 3, 2, 4, 1>>(), wk<WgradCfg<1, 4, 2, 4, 1>>(), wk<WgradCfg<1, 7, 2, 4, 1>>(),
        wk<WgradCfg<2, 3, 2, 4, 1>>(), wk<WgradCfg<2, 4, 2, 4, 1>>(),
    };
    return v;
}

struct WgradPlan {
    const WgradKernel* k = nullptr;
    int Cin = 0, Q = 0, tiles_x = 0, tiles_y = 0, nstages = 0, P = 0, m_tiles = 0, q_tiles = 0;
};

static int make_wgrad_plan(const ap_wgrad_desc* d, WgradPlan& pl) {
    if (!d) return fail(AP_ERR_INVALID, "wgrad: null descriptor");
    if (d->nsrc < 1 || d->nsrc > kMaxSeg) return fail(AP_ERR_INVALID, "wgrad: nsrc=%d", d->nsrc);
    if (d->N < 1 || d->M < 1 || d->GH < 1 || d->GW < 1 || d->H < 1 || d->W < 1) return fail(AP_ERR_INVALID, "wgrad: bad dims");
    for (const auto& k : wgrad_registry())
        if (k.S == d->stride && k.K == d->K) pl.k = &k;
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
    pl.Q = pl.Cin * d->K * d->K;
    pl.tiles_x = (d->GW + 31) / 32;
    pl.tiles_y = (d->GH + pl.k->PR - 1) / pl.k->PR;
    pl.nstages = d->N * pl.tiles_y * pl.tiles_x;
    pl.m_tiles = (d->M + pl.k->M_TILE - 1) / pl.k->M_TILE;
    pl.q_tiles = (pl.Q + pl.k->Q_TILE - 1) / pl.k->Q_TILE;
    const char* e = getenv("APAMD_WGRAD_BLOCKS");
    const int target = e ? atoi(e) : 1024;
    int P = (target + pl.m_tiles * pl.q_tiles - 1) / (pl.m_tiles * pl.q_tiles);
    if (P > pl.nstages) P = pl.nstages;
    if (P < 1) P = 1;
    pl.P = P;
    return AP_OK;
}

static std::mutex g_wattr_mu;
static std::vector<const void*> g_wattr_done;

}  // namespace apamd

using namespace apamd;

extern "C" {

int64_t ap_conv2d_wgrad_workspace_floats(const ap_wgrad_desc* d) {
    WgradPlan pl;
    int rc = make_wgrad_plan(d, pl);
    if (rc) return rc;
    return (int64_t)pl.P * d->M * pl.Q;
}

int ap_conv2d_wgrad(const ap_wgrad_desc* d, float* workspace, float* dw, ap_stream_t stream) {
    WgradPlan pl;
    int rc = make_wgrad_plan(d, pl);
    if (rc) return rc;
    if (!workspace || !dw || !d->g.data) return fail(AP_ERR_INVALID, "wgrad: null pointer");
    if ((d->g.mean == nullptr) != (d->g.rstd == nullptr)) return fail(AP_ERR_INVALID, "wgrad: g mean/rstd mismatch");
    {
        std::lock_guard<std::mutex> lk(g_wattr_mu);
        bool done = false;
        for (auto f : g_wattr_done) done = done || f == pl.k->fn;
        if (!done) {
            hipError_t e = hipFuncSetAttribute(pl.k->fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            g_wattr_done.push_back(pl.k->fn);
        }
    }
    WgradKParams p;
    memset(&p, 0, sizeof(p));
    p.g.data = d->g.data; p.g.mean = d->g.mean; p.g.rstd = d->g.rstd; p.g.C = d->M; p.g.act = d->g.act;
    p.nseg = d->nsrc;
    int cbeg = 0;
    for (int s = 0; s < d->nsrc; ++s) {
        if (!d->src[s].data) return fail(AP_ERR_INVALID, "wgrad: segment %d: null data", s);
        if ((d->src[s].mean == nullptr) != (d->src[s].rstd == nullptr))
            return fail(AP_ERR_INVALID, "wgrad: segment %d mean/rstd mismatch", s);
        p.seg[s].data = d->src[s].data; p.seg[s].mean = d->src[s].mean; p.seg[s].rstd = d->src[s].rstd;
        p.seg[s].C = d->src[s].C; p.seg[s].act = d->src[s].act; p.seg[s].chunk_begin = cbeg;
        cbeg += d->src[s].C;
    }
    p.N = d->N; p.M = d->M; p.Cin = pl.Cin; p.Q = pl.Q; p.GH = d->GH; p.GW = d->GW; p.H = d->H; p.W = d->W;
    p.pad = d->pad; p.pad_mode = d->pad_mode;
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.nstages = pl.nstages; p.P = pl.P;
    p.m_tiles = pl.m_tiles; p.q_tiles = pl.q_tiles;
    p.partial = workspace;
    void* args[] = {&p};
    const unsigned nblk = (unsigned)(pl.m_tiles * pl.q_tiles * pl.P);
    hipError_t e = hipLaunchKernel(pl.k->fn, dim3(nblk), dim3(256), args, pl.k->lds_bytes, (hipStream_t)stream);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "wgrad_igemm_f32 launch: %s", hipGetErrorString(e));
    const long long n = (long long)d->M * pl.Q;
    int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, workspace, pl.P, n, dw);
    return check_launch("wgrad_reduce_kernel");
}

}  // extern "C"
