// wino_bench.hip -- go / no-go micro-benchmark of the Winograd F(2x2,3x3) split-bf16 path (tools/conv_wino.h) at the
// generator's trunk shapes, with a CPU spot check of the result.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ianimateportrait_amd/csrc -Itools tools/wino_bench.hip -o tools/wino_bench.bin
// Run on the GPU box: tools/wino_bench.bin [N=16]
#include "conv_wino.h"

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>

namespace apamd {
static char g_err[512];
char* last_error_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace apamd
using namespace apamd;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

static int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

template <int ABL>
static const void* wino_fn() { return reinterpret_cast<const void*>(&conv_wino<ABL>); }

static void run_case(int N, std::vector<int> segC, int Cout, int H, int W, int pad_mode, bool stats, int iters, bool ablate = false) {
    const int nseg = (int)segC.size();
    int Cin = 0;
    for (int c : segC) Cin += c;
    const int TW = W / 2, T = (H / 2) * TW;
    const int kst = (Cin + 31) / 32;
    const int co_tiles = (Cout + 127) / 128, px_tiles = T / 128;
    printf("---- N=%d Cin=%d (", N, Cin);
    for (int c : segC) printf("%d ", c);
    printf(") Cout=%d %dx%d pad=%s  kstages=%d jobs=%d\n", Cout, H, W, pad_mode ? "reflect" : "zero", kst, N * px_tiles * co_tiles);
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<std::vector<float>> hx(nseg);
    std::vector<float*> dx(nseg);
    std::vector<uint4*> dvs(nseg);
    for (int s = 0; s < nseg; ++s) {
        hx[s].resize((size_t)N * segC[s] * H * W);
        for (auto& v : hx[s]) { v = nd(rng); v = v > 0.f ? v : 0.f; }
        CK(hipMalloc(&dx[s], hx[s].size() * 4));
        CK(hipMemcpy(dx[s], hx[s].data(), hx[s].size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dvs[s], (size_t)N * 16 * 2 * (segC[s] / 8) * T * 16));
    }
    std::vector<float> hw((size_t)Cout * Cin * 9), hb(Cout);
    for (auto& v : hw) v = 0.02f * nd(rng);
    for (auto& v : hb) v = 0.1f * nd(rng);
    float *dw, *db, *dy, *dstats = nullptr;
    unsigned char* dup;
    CK(hipMalloc(&dw, hw.size() * 4));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dy, (size_t)N * Cout * H * W * 4));
    const size_t up_bytes = (size_t)co_tiles * 16 * kst * WinoCfg::W_BYTES;
    CK(hipMalloc(&dup, up_bytes));
    const int stat_tiles = px_tiles * 2;
    if (stats) CK(hipMalloc(&dstats, (size_t)N * Cout * stat_tiles * 2 * 4));

    WinoPackParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.w = dw; pp.out = reinterpret_cast<unsigned short*>(dup);
    pp.Cin = Cin; pp.Cout = Cout; pp.layout = 0; pp.flip = 0; pp.kstages = kst; pp.co_tiles = co_tiles;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(1024), dim3(256), 0, 0, pp);
    CK(hipGetLastError());

    auto run_input = [&](int s) {
        WinoInParams ip;
        memset(&ip, 0, sizeof(ip));
        ip.x = dx[s]; ip.vs = dvs[s]; ip.pad_mode = pad_mode;
        ip.N = N; ip.C = segC[s]; ip.H = H; ip.W = W; ip.TW = TW; ip.T = T;
        const int R = 256 / TW;
        const size_t lds = (size_t)8 * (2 * R + 2) * (W + 4) * 4;
        hipLaunchKernelGGL(wino_input_kernel, dim3(T / 256, segC[s] / 8, N), dim3(256), lds, 0, ip);
    };
    WinoParams p;
    memset(&p, 0, sizeof(p));
    p.nseg = nseg;
    int cgb = 0;
    for (int s = 0; s < nseg; ++s) {
        p.seg[s].vs = reinterpret_cast<const unsigned char*>(dvs[s]);
        p.seg[s].CG = segC[s] / 8;
        p.seg[s].cg_begin = cgb;
        cgb += segC[s] / 8;
    }
    p.cg_real = cgb;
    p.N = N; p.H = H; p.W = W; p.TW = TW; p.T = T; p.Cout = Cout; p.kstages = kst;
    p.up = dup; p.bias = stats ? nullptr : db; p.act = 0; p.y = dy; p.stats = dstats; p.stat_tiles = stat_tiles;
    p.co_tiles = co_tiles; p.px_tiles = px_tiles;
    const void* fns[] = {wino_fn<0>(), wino_fn<1>(), wino_fn<2>(), wino_fn<4>(), wino_fn<8>(), wino_fn<16>(), wino_fn<32>(), wino_fn<1 | 4>(), wino_fn<2 | 4 | 8>()};
    const char* fnames[] = {"full", "no LDS-DMA", "no MFMA", "no fold", "no output stores", "weight stream only", "activation stream only",
                            "no DMA, no fold (MFMA + fragment reads)", "DMA only (no MFMA / fold / stores)"};
    for (const void* f : fns) CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const void* fn = fns[0];
    int ncu = 256;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    int nblk = N * px_tiles * co_tiles;
    if (nblk > ncu) nblk = ncu;
    auto run_conv = [&]() {
        void* args[] = {&p};
        CK(hipLaunchKernel(fn, dim3(nblk), dim3(256), args, WinoCfg::lds_bytes(), 0));
    };
    for (int s = 0; s < nseg; ++s) run_input(s);
    CK(hipGetLastError());
    run_conv();
    CK(hipDeviceSynchronize());

    // ---- spot check against a double-precision direct convolution
    std::vector<float> hy((size_t)N * Cout * H * W);
    CK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
    std::mt19937 prng(99);
    double max_err = 0, max_ref = 0;
    for (int k = 0; k < 4000; ++k) {
        const int n = prng() % N, co = prng() % Cout;
        int oy = prng() % H, ox = prng() % W;
        if (k % 4 == 0) oy = (k / 4) % 2 ? H - 1 : 0;       // borders
        if (k % 4 == 1) ox = (k / 4) % 2 ? W - 1 : 0;
        double acc = stats ? 0.0 : hb[co];
        int c0 = 0;
        for (int s = 0; s < nseg; ++s) {
            for (int c = 0; c < segC[s]; ++c)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        int iy = oy + ky - 1, ix = ox + kx - 1;
                        double xv;
                        if (pad_mode) { iy = reflect(iy, H); ix = reflect(ix, W); xv = hx[s][(((size_t)n * segC[s] + c) * H + iy) * W + ix]; }
                        else xv = (iy < 0 || iy >= H || ix < 0 || ix >= W) ? 0.0 : hx[s][(((size_t)n * segC[s] + c) * H + iy) * W + ix];
                        acc += xv * hw[(((size_t)co * Cin + c0 + c) * 3 + ky) * 3 + kx];
                    }
            c0 += segC[s];
        }
        const double got = hy[(((size_t)n * Cout + co) * H + oy) * W + ox];
        max_err = std::max(max_err, std::fabs(got - acc));
        max_ref = std::max(max_ref, std::fabs(acc));
    }
    printf("spot check (4000 outputs): max |err| %.3e   max |ref| %.3f   -> rel %.3e  %s\n", max_err, max_ref, max_err / max_ref,
           max_err / max_ref < 2e-4 ? "OK" : "FAIL");
    if (stats) {
        std::vector<float> hs((size_t)N * Cout * stat_tiles * 2);
        CK(hipMemcpy(hs.data(), dstats, hs.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int k = 0; k < 200; ++k) {
            const int n = prng() % N, co = prng() % Cout;
            double s = 0, q = 0, rs = 0, rq = 0;
            for (int t = 0; t < stat_tiles; ++t) { s += hs[(((size_t)n * Cout + co) * stat_tiles + t) * 2]; q += hs[(((size_t)n * Cout + co) * stat_tiles + t) * 2 + 1]; }
            for (int i = 0; i < H * W; ++i) { const double v = hy[((size_t)n * Cout + co) * H * W + i]; rs += v; rq += v * v; }
            worst = std::max(worst, std::fabs(s - rs) / (std::fabs(rs) + 1.0));
            worst = std::max(worst, std::fabs(q - rq) / (std::fabs(rq) + 1.0));
        }
        printf("statistics partials vs the output: worst relative difference %.3e  %s\n", worst, worst < 1e-4 ? "OK" : "FAIL");
    }

    // ---- timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) run_input(0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double in_us = ms * 1e3 / iters;
    const double in_bytes = (double)N * segC[0] * H * W * 4 + (double)N * 16 * 2 * (segC[0] / 8) * T * 16;
    printf("wino_input_kernel (segment 0, %d ch): %.1f us  (%.2f TB/s: %.0f MB read + written)\n", segC[0], in_us,
           in_bytes / in_us * 1e-6, in_bytes * 1e-6);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) run_conv();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    const double alg = 2.0 * N * H * W * (double)Cout * Cin * 9;
    const double exe = 2.0 * N * T * (double)co_tiles * 128 * kst * 32 * 16 * 3;
    printf("conv_wino: %.1f us   algorithmic %.1f TFLOP/s   executed %.1f TFLOP/s (bf16 MFMA)   L2->LDS %.2f TB/s\n", us,
           alg / us * 1e-6, exe / us * 1e-6, (double)N * px_tiles * co_tiles * 16 * kst * 32768.0 / us * 1e-6);
    if (ablate) {
        for (int v = 1; v < 9; ++v) {
            fn = fns[v];
            run_conv();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) run_conv();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("   ablation %-44s %.1f us\n", fnames[v], ms * 1e3 / iters);
        }
        fn = fns[0];
    }
    for (int s = 0; s < nseg; ++s) { CK(hipFree(dx[s])); CK(hipFree(dvs[s])); }
    CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy)); CK(hipFree(dup));
    if (dstats) CK(hipFree(dstats));
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16;
    if (argc > 2) {                                           // profiling mode: the ResnetBlock layer only
        run_case(N, {256}, 256, 64, 64, 1, true, 20, atoi(argv[2]) > 1);
        return 0;
    }
    run_case(2, {64}, 128, 32, 64, 1, true, 3);              // small: correctness of every path
    run_case(2, {64, 16, 16}, 128, 32, 64, 0, false, 3);
    run_case(N, {256}, 256, 64, 64, 1, true, 20);            // ResnetBlock conv (15 of the 22 launches)
    run_case(N, {256, 16, 16}, 256, 64, 64, 1, true, 20);    // ResnetBlock2 conv_block.1
    run_case(N, {256, 16, 16}, 256, 64, 64, 0, true, 20);    // ResnetBlock2 shortcut
    run_case(N, {256, 256, 256}, 256, 64, 64, 0, false, 20); // merge conv (bias, no norm)
    return 0;
}
