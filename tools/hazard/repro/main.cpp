// main.cpp -- library-free reproducer of the co-residency hazard (profiles/r04_cohazard.md), part 3 of 3: the host program.
// Uses hip_runtime.h only (no libapamd.so, no torch, no rocBLAS):
//     cd tools/hazard/repro && make && ./cohazard.bin [launches per cell, default 200]
// For every victim reduction level and every stream arrangement it runs the victim `launches` times beside the looping
// aggressor (another stream) and compares each output BITWISE with the victim's output when it ran alone.
// Stream arrangements:
//   shared            two plain streams: the kernels share compute units
//   control           two plain streams, the aggressor WITHOUT its run-time-indexed private array (clean in the lab)
//   masked:half       hipExtStreamCreateWithCUMask, aggressor on mask bits 0..127, victim on bits 128..255 (disjoint CUs)
//   masked:xcd        ... aggressor on even mask bits, victim on odd ones (disjoint XCDs if bits go round-robin over the XCDs)
//   masked:cu         ... aggressor on bits with (i / 8) even, victim on the others (alternating CUs of every XCD)
//   masked:same       both streams on the SAME half of the mask (control of the masking itself: co-resident again)
// Exit code = number of (level, arrangement) cells with at least one wrong launch among the `shared` and `masked:*` rows.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" hipError_t launch_aggressor(int which, const void* src, void* sink, int iters, hipStream_t stream);
extern "C" hipError_t launch_victim(int level, const float* x, const float* mean, const float* rstd, const float* motion, const float* flow,
                                    const float* ifmask, float* out, int N, int C, int H, int W, int S, float flow_scale, hipStream_t stream);
extern "C" size_t victim_out_floats(int level, int N, int C, int H, int W);
extern "C" hipError_t launch_count_diff(const void* a, const void* ref, size_t n_words, unsigned long long* count, unsigned* lane_hist, hipStream_t stream);

// LEVEL 3 = the product's own translation unit (animateportrait_amd/csrc/warp.hip compiled into this program, `make PRODUCT=1`
// is the default): the kernel that fails in the lab, through its C entry point.  Its two external helpers are defined here.
extern "C" int ap_warp_concat_fwd_ex(const float* x, const float* x_mean, const float* x_rstd, int32_t x_act, const float* motion, const float* flow,
                                     const float* ifmask, float* out, void* xs, int32_t N, int32_t C, int32_t H, int32_t W, int32_t S,
                                     float flow_scale, int32_t flags, void* stream);
#define DECL_K(n) extern "C" hipError_t launch_victim_k##n(const float*, const float*, const float*, const float*, const float*, const float*, float*, int, int, int, int, int, float, hipStream_t);
DECL_K(0) DECL_K(1) DECL_K(2) DECL_K(3) DECL_K(4) DECL_K(5) DECL_K(6)
namespace apamd {
static char g_err[512];
char* last_error_buf() { return g_err; }
int fail(int code, const char* fmt, ...) { snprintf(g_err, sizeof(g_err), "%s", fmt); return code; }
}  // namespace apamd

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
            exit(99);                                                                                  \
        }                                                                                              \
    } while (0)

static unsigned rng_state = 12345u;
static float urand() {   // uniform in [0, 1)
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)(rng_state >> 8) * (1.f / 16777216.f);
}
static float* upload(const std::vector<float>& h) {
    float* d;
    CK(hipMalloc(&d, h.size() * sizeof(float)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

struct Masks { uint32_t a[8], v[8]; };
static Masks make_masks(const std::string& kind) {
    Masks m;
    for (int w = 0; w < 8; ++w) {
        if (kind == "half") { m.a[w] = w < 4 ? 0xFFFFFFFFu : 0u; m.v[w] = w < 4 ? 0u : 0xFFFFFFFFu; }
        else if (kind == "xcd") { m.a[w] = 0x55555555u; m.v[w] = 0xAAAAAAAAu; }
        else if (kind == "cu") { m.a[w] = 0x00FF00FFu; m.v[w] = 0xFF00FF00u; }
        else { m.a[w] = w < 4 ? 0xFFFFFFFFu : 0u; m.v[w] = m.a[w]; }     // "same"
    }
    return m;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int iters = argc > 2 ? atoi(argv[2]) : 60;        // aggressor loop iterations per launch
    const int N = 8, C = 128, H = 64, W = 64, S = 256, K = 40;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), %d CUs\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);

    std::vector<float> hx((size_t)N * C * H * W), hm((size_t)N * C), hr((size_t)N * C), hmo((size_t)N * S * S * 2), hf((size_t)N * 2 * S * S), hk((size_t)N * S * S);
    for (auto& v : hx) v = (urand() + urand() + urand() + urand() - 2.f) * 1.7f;
    for (auto& v : hm) v = (urand() - 0.5f) * 0.2f;
    for (auto& v : hr) v = 0.5f + urand();
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x) {
                const size_t i = ((size_t)n * S + y) * S + x;
                hmo[2 * i] = (2.f * x + 1.f) / S - 1.f + (urand() - 0.5f) * 0.2f;
                hmo[2 * i + 1] = (2.f * y + 1.f) / S - 1.f + (urand() - 0.5f) * 0.2f;
            }
    for (auto& v : hf) v = (urand() - 0.5f) * 16.f;
    for (auto& v : hk) v = urand();
    float *dx = upload(hx), *dm = upload(hm), *dr = upload(hr), *dmo = upload(hmo), *df = upload(hf), *dk = upload(hk);
    std::vector<float> hbig((size_t)256 * 24 * 4096 / 4);
    for (auto& v : hbig) v = urand() - 0.5f;
    float* dbig = upload(hbig);
    float* dsink;
    CK(hipMalloc(&dsink, 64));
    unsigned long long* dcount;
    unsigned* dhist;
    CK(hipMalloc(&dcount, 8));
    CK(hipMalloc(&dhist, 64 * 4));

    hipStream_t plainA, plainB;
    CK(hipStreamCreate(&plainA));
    CK(hipStreamCreate(&plainB));

    // how long the aggressor takes alone on a plain stream and on half of the CUs: shows whether the CU mask is honoured
    auto time_aggr = [&](hipStream_t s) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(launch_aggressor(1, dbig, dsink, 60, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(launch_aggressor(1, dbig, dsink, 60, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / 10;
    };
    {
        const Masks m = make_masks("half");
        hipStream_t half;
        CK(hipExtStreamCreateWithCUMask(&half, 8, m.a));
        printf("aggressor alone: %.3f ms per launch on a plain stream, %.3f ms on a stream masked to 128 of the 256 CUs\n", time_aggr(plainA), time_aggr(half));
        CK(hipStreamDestroy(half));
    }

    int bad_cells = 0;
    printf("\n| victim level | arrangement | wrong victim launches | of | wrong 32-bit words | lanes (index mod 64) with wrong words | ms: aggressor batches / victim batches / overlap |\n|---|---|---|---|---|---|---|\n");
    const char* arrangements[] = {"shared", "control", "masked:half", "masked:xcd", "masked:cu", "masked:same"};
    // victims: id 3 = the product's translation unit; 10 + k = its kernel copied into victim_product.hip with -DKNOB=k (pieces
    // removed step by step); 0 / 1 / 2 = the hand-reduced kernel of victim.hip
    auto run_victim = [&](int level, float* out, hipStream_t st) -> hipError_t {
        if (level < 3 || level >= 20) return launch_victim(level, dx, dm, dr, dmo, df, dk, out, N, C, H, W, S, 0.25f, st);
        if (level == 3) return ap_warp_concat_fwd_ex(dx, dm, dr, 1, dmo, df, dk, out, nullptr, N, C, H, W, S, 0.25f, 0, (void*)st) == 0 ? hipSuccess : hipErrorUnknown;
        decltype(&launch_victim_k0) ks[] = {launch_victim_k0, launch_victim_k1, launch_victim_k2, launch_victim_k3, launch_victim_k4, launch_victim_k5, launch_victim_k6};
        return ks[level - 10](dx, dm, dr, dmo, df, dk, out, N, C, H, W, S, 0.25f, st);
    };
    auto victim_name = [](int level) -> std::string {
        if (level == 3) return "product warp.hip (its own translation unit)";
        if (level == 20) return "SGPR-pair lane mask alone: v_cmp_e64 -> gather chain -> v_cndmask_e64";
        if (level == 21) return "SGPR-pair lane mask alone: v_cmp_e64 -> v_cndmask_e64 back to back";
        if (level >= 10) return "product kernel copy, KNOB=" + std::to_string(level - 10);
        return "hand-reduced, level " + std::to_string(level);
    };
    hipEvent_t ev[4];
    for (auto& e : ev) CK(hipEventCreate(&e));
    const int order[] = {3, 10, 13, 16, 20, 21, 0, 1, 2};
    for (int level : order) {
        const size_t nout = victim_out_floats((level >= 3 && level < 20) ? 0 : level, N, C, H, W);
        float* dref;
        CK(hipMalloc(&dref, nout * 4));
        std::vector<float*> douts(K);
        for (auto& p : douts) CK(hipMalloc(&p, nout * 4));
        // reference: the victim alone (twice: it must agree with itself)
        CK(run_victim(level, dref, plainA));
        CK(run_victim(level, douts[0], plainA));
        CK(hipMemsetAsync(dcount, 0, 8, plainA));
        CK(hipMemsetAsync(dhist, 0, 256, plainA));
        CK(launch_count_diff(douts[0], dref, nout, dcount, dhist, plainA));
        unsigned long long self = 0;
        CK(hipMemcpy(&self, dcount, 8, hipMemcpyDeviceToHost));
        if (self) { printf("victim %s does not agree with itself when run alone (%llu words)\n", victim_name(level).c_str(), self); return 98; }
        for (const char* arr : arrangements) {
            const std::string a(arr);
            if (level != 3 && level != 16 && a.rfind("masked:", 0) == 0) continue;       // the CU-mask arrangements: the product victim only
            hipStream_t sv = plainA, sa = plainB;
            const bool masked = a.rfind("masked:", 0) == 0;
            if (masked) {
                const Masks m = make_masks(a.substr(7));
                CK(hipExtStreamCreateWithCUMask(&sa, 8, m.a));
                CK(hipExtStreamCreateWithCUMask(&sv, 8, m.v));
            }
            const int which = a == "control" ? 0 : 1;
            int wrong = 0, total = 0;
            unsigned long long words = 0;
            unsigned hist[64] = {0};
            float aggr_ms = 0.f, vict_ms = 0.f, overlap_ms = 0.f;
            while (total < launches) {
                CK(hipEventRecord(ev[0], sa));
                for (int i = 0; i < K; ++i) CK(launch_aggressor(which, dbig, dsink, iters, sa));
                CK(hipEventRecord(ev[1], sa));
                CK(hipEventRecord(ev[2], sv));
                for (int i = 0; i < K; ++i) CK(run_victim(level, douts[i], sv));
                CK(hipEventRecord(ev[3], sv));
                CK(hipDeviceSynchronize());
                {   // did the two batches overlap in time?  (victim batch inside the aggressor batch's interval)
                    float a_ms, v_ms, v_start;
                    CK(hipEventElapsedTime(&a_ms, ev[0], ev[1]));
                    CK(hipEventElapsedTime(&v_ms, ev[2], ev[3]));
                    CK(hipEventElapsedTime(&v_start, ev[0], ev[2]));
                    aggr_ms += a_ms; vict_ms += v_ms;
                    overlap_ms += fmaxf(0.f, fminf(a_ms, v_start + v_ms) - fmaxf(0.f, v_start));
                }
                for (int i = 0; i < K; ++i) {
                    CK(hipMemset(dcount, 0, 8));
                    CK(launch_count_diff(douts[i], dref, nout, dcount, dhist, plainA));
                    unsigned long long c = 0;
                    CK(hipMemcpy(&c, dcount, 8, hipMemcpyDeviceToHost));
                    wrong += c != 0;
                    words += c;
                }
                total += K;
            }
            CK(hipMemcpy(hist, dhist, 256, hipMemcpyDeviceToHost));
            CK(hipMemset(dhist, 0, 256));
            std::string lanes;
            int lo = -1;
            for (int l = 0; l <= 64; ++l) {
                const bool on = l < 64 && hist[l];
                if (on && lo < 0) lo = l;
                if (!on && lo >= 0) { lanes += (lanes.empty() ? "" : ", ") + std::to_string(lo) + "-" + std::to_string(l - 1); lo = -1; }
            }
            printf("| %s | %s | %d | %d | %llu | %s | %.1f / %.1f / %.1f |\n", victim_name(level).c_str(), arr, wrong, total, words,
                   lanes.empty() ? "-" : lanes.c_str(), aggr_ms, vict_ms, overlap_ms);
            fflush(stdout);
            if (wrong && a != "control" && a != "masked:same") ++bad_cells;
            if (masked) { CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sv)); }
        }
        for (auto p : douts) CK(hipFree(p));
        CK(hipFree(dref));
    }
    return bad_cells;
}
