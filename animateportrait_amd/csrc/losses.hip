// losses.hip -- the memory-stream kernels around the networks of the geomgm_ifw_fore train step and of the
// streaming-inference caller: loss reductions, mask compositing, the window-crop + resize glue in front of the
// frozen auxiliary nets, and the landmark rasterisers of the data side.
//
// Reference code these replace (all ATen elementwise / numpy / cv2 work there):
//   GANLoss lsgan                       Module2/models/networks.py:429-430, 455-473
//   L1 / lip-line / warp losses         Module2/models/geomgm_ifw_fore_model.py:715-739
//   foreground / background compositing Module2/models/geomgm_ifw_fore_model.py:523-543
//   BaseModel.masked                    Module2/models/base_model.py:238-247
//   get_lm crop + bicubic               Module2/models/geomgm_ifw_fore_model.py:390-407
//   FaceLoss.crop_head_bbox             Module2/models/networks.py:2946-2966
//   kp_to_map / flow_network_warp       Module2/models/geomgm_ifw_fore_model.py:19-51, 69-84
//   draw2 op=0 (cv2.circle, filled)     Module2/data/umlvdfw_test_dataset.py:34-41
//
// Everything here is HBM-bound (or trivially small); the compositing formulas keep the reference's operation order
// (this file is compiled with -ffp-contract=off) so that their results are bit-identical to the ATen chain.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace apamd {

// ------------------------------------------------------------------------------------------------ reductions
enum { RED_SQDIFF = 0, RED_L1 = 1, RED_WMEAN = 2 };

template <int OP>
__device__ __forceinline__ float red_term(float a, float b, float c) {
    if (OP == RED_SQDIFF) { const float d = a - c; return d * d; }      // (pred - target)^2, target scalar c
    if (OP == RED_L1) return fabsf(a - b);                              // |a - b|
    return (a + c) * b;                                                 // (a + add) * w
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) v += __shfl_xor(v, sh, 64);
    return v;
}

// stage 1: block partial sums (fixed grid-stride order -> deterministic)
template <int OP>
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float c, long long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        s += red_term<OP>(a[i], b ? b[i] : 0.f, c);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// stage 2: one wave adds the partials in a fixed order; out = weight * sum / n
__global__ __launch_bounds__(64) void reduce_final_kernel(const float* __restrict__ partial, int blocks, float scale,
                                                          float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < blocks; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s * scale;
}

// backward of out = weight * mean(term):  ga[i] = gout * weight / n * d term / d a
template <int OP>
__global__ __launch_bounds__(256) void reduce_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float c,
                                                         long long n, const float* __restrict__ gout, float scale,
                                                         float* __restrict__ ga) {
    const float g = gout[0] * scale;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float d;
        if (OP == RED_SQDIFF) d = 2.f * (a[i] - c);
        else if (OP == RED_L1) { const float t = a[i] - b[i]; d = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f); }
        else d = b[i];
        ga[i] = g * d;
    }
}

// ------------------------------------------------------------------------------------------------ compositing
// mode 0: (a/2+.5)*m*2-1                                  masked() type 0
// mode 1: ((a/2+.5)*m + 1 - m)*2-1                        masked() type 1 / 3, foreground-on-white
// mode 2: ((a/2+.5)*m + (s/2+.5)*(1-m))*2-1               background blend with the static drawing
// mode 3: a                                               masked() type 2 (only the mask channel is appended)
// a: N x C x HW, m: N x 1 x HW (broadcast over C), s: N x Cs x HW with Cs in {1, C}; out: N x (C + append) x HW,
// channel C (if append) = m.  Operation order is the reference's.
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ a, const float* __restrict__ m,
                                                        const float* __restrict__ s, int C, int Cs, int HW, int mode,
                                                        int append, float* __restrict__ out) {
    const int n = blockIdx.z, c = blockIdx.y;
    const int OC = C + append;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const float mv = m[(long long)n * HW + i];
        float r;
        if (c == C) r = mv;
        else {
            const float av = a[((long long)n * C + c) * HW + i];
            if (mode == 3) r = av;
            else {
                const float t = (av / 2.f + 0.5f) * mv;
                if (mode == 0) r = t * 2.f - 1.f;
                else if (mode == 1) r = ((t + 1.f) - mv) * 2.f - 1.f;
                else {
                    const float sv = s[((long long)n * Cs + (Cs == 1 ? 0 : c)) * HW + i];
                    r = (t + (sv / 2.f + 0.5f) * (1.f - mv)) * 2.f - 1.f;
                }
            }
        }
        out[((long long)n * OC + c) * HW + i] = r;
    }
}

// backward w.r.t. a of every mode: ga = g[:, :C] * m (modes 0-2), g[:, :C] (mode 3); g has GC >= C channels
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ g, const float* __restrict__ m, int C,
                                                            int GC, int HW, int mode, float* __restrict__ ga) {
    const int n = blockIdx.z, c = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const float gv = g[((long long)n * GC + c) * HW + i];
        ga[((long long)n * C + c) * HW + i] = mode == 3 ? gv : gv * m[(long long)n * HW + i];
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n,
                                                   float alpha) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] += alpha * src[i];
}

// ------------------------------------------------------------------------------------------------ crop + resize
// The box of get_lm / crop_head_bbox: a (x2-x1)^2 square of ones; rows/cols of the window that lie inside the image
// (and inside [y1, y2) for rows) hold the image.  Returns the source pixel index or -1 for "one".
struct Box {
    int x1, x2, y1, y2, H, W, size;
    __device__ __forceinline__ int src(int by, int bx) const {
        by = min(max(by, 0), size - 1);          // border-replicated taps of the resize
        bx = min(max(bx, 0), size - 1);
        const int iy = y1 + by, ix = x1 + bx;
        if (iy < max(0, y1) || iy >= min(y2, H) || ix < max(0, x1) || ix >= min(W, x2)) return -1;
        return iy * W + ix;
    }
};

__device__ __forceinline__ void cubic_coeffs(float t, float* c) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
    c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    c[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// taps of one output pixel: up to 4 x 4 (bicubic, align_corners=False) or 2 x 2 (bilinear, align_corners=True)
struct Taps {
    int iy[4], ix[4], n;
    float wy[4], wx[4];
};

__device__ __forceinline__ Taps make_taps(int oy, int ox, int OH, int OW, int size, int mode) {
    Taps t;
    if (mode == 1) {
        t.n = 4;
        const float sy = (float)size / (float)OH, sx = (float)size / (float)OW;
        const float ry = sy * ((float)oy + 0.5f) - 0.5f, rx = sx * ((float)ox + 0.5f) - 0.5f;
        const float fy = floorf(ry), fx = floorf(rx);
        cubic_coeffs(ry - fy, t.wy);
        cubic_coeffs(rx - fx, t.wx);
        for (int i = 0; i < 4; ++i) { t.iy[i] = (int)fy - 1 + i; t.ix[i] = (int)fx - 1 + i; }
    } else {
        t.n = 2;
        const float sy = OH > 1 ? (float)(size - 1) / (float)(OH - 1) : 0.f;
        const float sx = OW > 1 ? (float)(size - 1) / (float)(OW - 1) : 0.f;
        const float ry = sy * (float)oy, rx = sx * (float)ox;
        const int y0 = (int)ry, x0 = (int)rx;
        t.iy[0] = y0; t.iy[1] = y0 + (y0 < size - 1 ? 1 : 0);
        t.ix[0] = x0; t.ix[1] = x0 + (x0 < size - 1 ? 1 : 0);
        t.wy[1] = ry - (float)y0; t.wy[0] = 1.f - t.wy[1];
        t.wx[1] = rx - (float)x0; t.wx[0] = 1.f - t.wx[1];
    }
    return t;
}

// grid: (ceil(OH*OW/256), OC, N)
__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ x, const int* __restrict__ win, int C,
                                                          int H, int W, int chmap, int OH, int OW, int mode, float scale,
                                                          float shift, float* __restrict__ out) {
    const int n = blockIdx.z, k = blockIdx.y, OC = gridDim.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= OH * OW) return;
    const int* wn = win + 4 * n;
    Box b{wn[0], wn[1], wn[2], wn[3], H, W, wn[1] - wn[0]};
    const float* xp = x + ((long long)n * C + ((chmap >> (8 * k)) & 255)) * H * W;
    const Taps t = make_taps(p / OW, p % OW, OH, OW, b.size, mode);
    float acc = 0.f;
    for (int i = 0; i < t.n; ++i) {
        float row = 0.f;
        for (int j = 0; j < t.n; ++j) {
            const int s = b.src(t.iy[i], t.ix[j]);
            row += t.wx[j] * (s < 0 ? 1.f : xp[s]);
        }
        acc += t.wy[i] * row;
    }
    out[((long long)n * OC + k) * OH * OW + p] = acc * scale + shift;
}

// gx (zeroed by the host wrapper) += scale * sum over taps; atomics: the maps are tiny (OH*OW*OC*16 taps per sample)
__global__ __launch_bounds__(256) void crop_resize_bwd_kernel(const float* __restrict__ gout, const int* __restrict__ win,
                                                              int C, int H, int W, int chmap, int OH, int OW, int mode,
                                                              float scale, float* __restrict__ gx) {
    const int n = blockIdx.z, k = blockIdx.y, OC = gridDim.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= OH * OW) return;
    const int* wn = win + 4 * n;
    Box b{wn[0], wn[1], wn[2], wn[3], H, W, wn[1] - wn[0]};
    float* gp = gx + ((long long)n * C + ((chmap >> (8 * k)) & 255)) * H * W;
    const Taps t = make_taps(p / OW, p % OW, OH, OW, b.size, mode);
    const float g = gout[((long long)n * OC + k) * OH * OW + p] * scale;
    for (int i = 0; i < t.n; ++i)
        for (int j = 0; j < t.n; ++j) {
            const int s = b.src(t.iy[i], t.ix[j]);
            if (s >= 0) atomicAdd(gp + s, g * t.wy[i] * t.wx[j]);
        }
}

// ------------------------------------------------------------------------------------------------ rasterisers
// kp_to_map(mode='binary'): out[n][p][y][x] = ((x - kx)^2 + (y - ky)^2 <= r^2), kx = lm_x * num / den as numpy forms it
// (float32 product, float32 quotient), the test itself in float64 (int64 grid minus float32 scalar promotes); a
// coordinate equal to -1 blanks the map.  grid: (ceil(S*S/256), P, N)
__global__ __launch_bounds__(256) void kp_to_map_kernel(const float* __restrict__ lm, int P, int S, float num, float den,
                                                        float radius, float* __restrict__ out) {
    const int n = blockIdx.z, p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * S) return;
    const float* q = lm + ((long long)n * P + p) * 2;
    const float kx = (q[0] * num) / den, ky = (q[1] * num) / den;
    float v = 0.f;
    if (!(kx == -1.f || ky == -1.f)) {
        const double dx = (double)(i % S) - (double)kx, dy = (double)(i / S) - (double)ky;
        v = (dx * dx + dy * dy <= (double)radius * (double)radius) ? 1.f : 0.f;
    }
    out[((long long)n * P + p) * S * S + i] = v;
}

// flow_network_warp tail: mask = (argmax_c vis < 2); flow' = flow * 20 * mask / 7 * 8; both resized S -> OS with
// bilinear align_corners=True.  grid: (ceil(OS*OS/256), N).  flow (N,2,S,S), vis (N,VC,S,S)
__global__ __launch_bounds__(256) void flow_post_kernel(const float* __restrict__ flow, const float* __restrict__ vis,
                                                        int VC, int S, int OS, float gain, float num, float den,
                                                        float* __restrict__ flow_out, float* __restrict__ mask_out) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= OS * OS) return;
    const float sc = OS > 1 ? (float)(S - 1) / (float)(OS - 1) : 0.f;
    const float ry = sc * (float)(p / OS), rx = sc * (float)(p % OS);
    const int y0 = (int)ry, x0 = (int)rx;
    const int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
    const float ly = ry - (float)y0, lx = rx - (float)x0;
    const int idx[4] = {y0 * S + x0, y0 * S + x1, y1 * S + x0, y1 * S + x1};
    const float w[4] = {(1.f - ly) * (1.f - lx), (1.f - ly) * lx, ly * (1.f - lx), ly * lx};
    float fm = 0.f, f0 = 0.f, f1 = 0.f;
    for (int t = 0; t < 4; ++t) {
        const float* vp = vis + (long long)n * VC * S * S + idx[t];
        int best = 0;                      // torch.argmax: first index of the maximum
        float bv = vp[0];
        for (int c = 1; c < VC; ++c) {
            const float v = vp[(long long)c * S * S];
            if (v > bv) { bv = v; best = c; }
        }
        const float m = best < 2 ? 1.f : 0.f;
        const float* fp = flow + (long long)n * 2 * S * S + idx[t];
        fm += w[t] * m;
        f0 += w[t] * (((fp[0] * gain) * m) / den * num);
        f1 += w[t] * (((fp[(long long)S * S] * gain) * m) / den * num);
    }
    flow_out[((long long)n * 2 + 0) * OS * OS + p] = f0;
    flow_out[((long long)n * 2 + 1) * OS * OS + p] = f1;
    mask_out[(long long)n * OS * OS + p] = fm;
}

// draw2(op=0): union of filled cv2.circle(radius) discs at np.round(lands), value +1 on a -1 canvas.
// halfw[d] (d = |dy| <= radius) is the half width of OpenCV's filled circle on that row (see circle_rows()).
// grid: (ceil(H*W/256), N); lm: N x P x 2 (x, y)
constexpr int kMaxDiscRadius = 31;
struct DiscRows { int hw[kMaxDiscRadius + 1]; };

__global__ __launch_bounds__(256) void landmark_discs_kernel(const float* __restrict__ lm, int P, int H, int W, int radius,
                                                             DiscRows rows, float lo, float hi, float* __restrict__ out) {
    extern __shared__ int pts[];           // P x 2 rounded coordinates
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * P; i += 256) pts[i] = (int)rintf(lm[(long long)n * P * 2 + i]);   // np.round: half to even
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p % W;
    bool hit = false;
    for (int i = 0; i < P && !hit; ++i) {
        const int dy = abs(y - pts[2 * i + 1]);
        if (dy <= radius) hit = abs(x - pts[2 * i]) <= rows.hw[dy];
    }
    out[(long long)n * H * W + p] = hit ? hi : lo;
}

// Row half-widths of OpenCV's filled circle: the octant walk of Circle() in modules/imgproc/src/drawing.cpp
// (opencv-python 4.2.0.34 is what the reference pins, requirements.txt:2): err = 0, dx = r, dy = 0, plus = 1,
// minus = 2r - 1; each step fills rows cy +- dy over [cx - dx, cx + dx] and rows cy +- dx over [cx - dy, cx + dy],
// then dy++, err += plus, plus += 2, and when err > 0: err -= minus, dx--, minus -= 2.
// ---- getlipline (Module2/models/geomgm_ifw_fore_model.py:507-515): the union of P0-P1 segments drawn by
// cv2.line(mask, p0, p1, 255, thickness), thickness >= 2, LINE_8.  One lane per (sample, segment) runs OpenCV's own
// integer algorithm (modules/imgproc/src/drawing.cpp, 4.2.0): ThickLine -> the quad p +- dp through FillConvexPoly at
// 16.16 fixed point (outline by Line2, scanlines with the rounded edge slopes) + a filled Circle at both end points.
// Restated in oracle/cv_raster.py with the source rules quoted; literal fixtures tests/golden/opencv_rules.json.
// Lanes store the same value (1.0f), so overlapping segments need no ordering.  out must be zero on entry.
struct LipPut {
    float* img;
    int H, W;
    __device__ __forceinline__ void put(int x, int y) const {
        if (x >= 0 && x < W && y >= 0 && y < H) img[y * W + x] = 1.f;
    }
    __device__ __forceinline__ void hline(int y, int x0, int x1) const {
        if (y < 0 || y >= H) return;
        x0 = x0 < 0 ? 0 : x0;
        x1 = x1 >= W ? W - 1 : x1;
        for (int x = x0; x <= x1; ++x) img[y * W + x] = 1.f;
    }
};

__device__ static void lip_circle(const LipPut& o, int cx, int cy, int radius) {
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        o.hline(cy - dy, cx - dx, cx + dx);
        o.hline(cy + dy, cx - dx, cx + dx);
        o.hline(cy - dx, cx - dy, cx + dy);
        o.hline(cy + dx, cx - dy, cx + dy);
        ++dy;
        err += plus;
        plus += 2;
        if (err > 0) { err -= minus; --dx; minus -= 2; }
    }
}

__device__ static void lip_line2(const LipPut& o, long long x1, long long y1, long long x2, long long y2) {
    constexpr int SH = 16;
    constexpr long long ONE = 1 << SH;
    long long dx = x2 - x1, dy = y2 - y1;
    const long long ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    if (ax > ay) {
        if (dx < 0) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; dy = -dy; }
        const long long y_step = (dy * ONE) / (ax | 1);            // C division: truncation toward zero
        long long ecount = (x2 - x1) >> SH;
        o.put((int)((x2 + (ONE >> 1)) >> SH), (int)((y2 + (ONE >> 1)) >> SH));
        long long x = (x1 + (ONE >> 1)) >> SH, y = y1 + (ONE >> 1);
        for (; ecount >= 0; --ecount) { o.put((int)x, (int)(y >> SH)); ++x; y += y_step; }
    } else {
        if (dy < 0) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; dx = -dx; }
        const long long x_step = (dx * ONE) / (ay | 1);
        long long ecount = (y2 - y1) >> SH;
        o.put((int)((x2 + (ONE >> 1)) >> SH), (int)((y2 + (ONE >> 1)) >> SH));
        long long x = x1 + (ONE >> 1), y = (y1 + (ONE >> 1)) >> SH;
        for (; ecount >= 0; --ecount) { o.put((int)(x >> SH), (int)y); x += x_step; ++y; }
    }
}

__device__ static void lip_fill_quad(const LipPut& o, const long long (&vx)[4], const long long (&vy)[4]) {
    constexpr int SH = 16, NP = 4;
    constexpr long long ONE = 1 << SH, delta = ONE >> 1;
    int imin = 0;
    long long ymin = vy[0], ymax = vy[0], xmin = vx[0], xmax = vx[0];
    long long px = vx[NP - 1], py = vy[NP - 1];
    for (int i = 0; i < NP; ++i) {
        if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
        ymax = vy[i] > ymax ? vy[i] : ymax;
        xmax = vx[i] > xmax ? vx[i] : xmax;
        xmin = vx[i] < xmin ? vx[i] : xmin;
        lip_line2(o, px, py, vx[i], vy[i]);
        px = vx[i]; py = vy[i];
    }
    xmin = (xmin + delta) >> SH; xmax = (xmax + delta) >> SH;
    ymin = (ymin + delta) >> SH; ymax = (ymax + delta) >> SH;
    if (xmax < 0 || ymax < 0 || xmin >= o.W || ymin >= o.H) return;
    if (ymax > o.H - 1) ymax = o.H - 1;
    int e_idx[2] = {imin, imin}, e_di[2] = {1, NP - 1};
    long long e_x[2] = {-ONE, -ONE}, e_dx[2] = {0, 0}, e_ye[2] = {ymin, ymin};
    int edges = NP;
    long long y = ymin;
    for (;;) {
        for (int i = 0; i < 2; ++i) {
            if (y >= e_ye[i]) {
                int idx0 = e_idx[i];
                const int di = e_di[i];
                int idx = (idx0 + di) % NP;
                for (;;) {
                    if (--edges < 0) break;
                    const long long ty = (vy[idx] + delta) >> SH;
                    if (ty > y) {
                        const long long xs = vx[idx0], xe = vx[idx];
                        e_ye[i] = ty;
                        e_dx[i] = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        e_x[i] = xs;
                        e_idx[i] = idx;
                        break;
                    }
                    idx0 = idx;
                    idx = (idx + di) % NP;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            const int l = e_x[0] > e_x[1] ? 1 : 0, r = 1 - l;
            const long long xx1 = (e_x[l] + delta) >> SH, xx2 = (e_x[r] + delta) >> SH;
            if (xx2 >= 0 && xx1 < o.W) o.hline((int)y, (int)xx1, (int)xx2);
        }
        e_x[0] += e_dx[0];
        e_x[1] += e_dx[1];
        if (++y > ymax) break;
    }
}

struct LipSegs { int a[32], b[32]; };

__global__ void lip_line_kernel(const float* __restrict__ lands, int P, LipSegs segs, int nseg, int N, int S, int thickness,
                                float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * nseg) return;
    const int n = t / nseg, s = t - n * nseg;
    const float* l = lands + (long long)n * P * 2;
    // the Python binding truncates the float coordinates to int
    const long long x0 = (long long)(int)l[segs.a[s] * 2] << 16, y0 = (long long)(int)l[segs.a[s] * 2 + 1] << 16;
    const long long x1 = (long long)(int)l[segs.b[s] * 2] << 16, y1 = (long long)(int)l[segs.b[s] * 2 + 1] << 16;
    LipPut o{out + (long long)n * S * S, S, S};
    const double dx = (double)(x0 - x1) * (1.0 / 65536.0), dy = (double)(y1 - y0) * (1.0 / 65536.0);
    double r = dx * dx + dy * dy;
    const int odd = thickness & 1;
    const long long tf = (long long)thickness << 15;
    if (fabs(r) > 2.220446049250313e-16) {
        r = ((double)tf + odd * 65536.0 * 0.5) / sqrt(r);
        const long long dpx = (long long)rint(dy * r), dpy = (long long)rint(dx * r);       // cvRound
        const long long vx[4] = {x0 + dpx, x0 - dpx, x1 - dpx, x1 + dpx};
        const long long vy[4] = {y0 + dpy, y0 - dpy, y1 - dpy, y1 + dpy};
        lip_fill_quad(o, vx, vy);
    }
    const int rad = (int)((tf + 32768) >> 16);
    lip_circle(o, (int)((x0 + 32768) >> 16), (int)((y0 + 32768) >> 16), rad);
    lip_circle(o, (int)((x1 + 32768) >> 16), (int)((y1 + 32768) >> 16), rad);
}


static DiscRows circle_rows(int r) {
    DiscRows t;
    for (int i = 0; i <= kMaxDiscRadius; ++i) t.hw[i] = -1;
    int err = 0, dx = r, dy = 0, plus = 1, minus = (r << 1) - 1;
    while (dx >= dy) {
        t.hw[dy] = std::max(t.hw[dy], dx);
        t.hw[dx] = std::max(t.hw[dx], dy);
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
    return t;
}

static inline int stream_blocks(long long n) { return (int)std::min<long long>((n + 255) / 256, 2048); }

}  // namespace apamd

using namespace apamd;

extern "C" {

int64_t ap_reduce_workspace_floats(void) { return 2048; }

#define AP_REDUCE_LAUNCH(OP)                                                                                          \
    hipLaunchKernelGGL(reduce_partial_kernel<OP>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, c,           \
                       (long long)n, workspace)

int ap_reduce_mean(int32_t op, const float* a, const float* b, float c, int64_t n, float weight, float* workspace,
                   float* out, ap_stream_t stream) {
    if (!a || !workspace || !out || n < 1) return fail(AP_ERR_INVALID, "reduce_mean: bad arguments");
    if (op < 0 || op > 2 || (op != RED_SQDIFF && !b)) return fail(AP_ERR_INVALID, "reduce_mean: op %d / missing operand", op);
    const int blocks = stream_blocks(n);
    if (op == RED_SQDIFF) AP_REDUCE_LAUNCH(RED_SQDIFF);
    else if (op == RED_L1) AP_REDUCE_LAUNCH(RED_L1);
    else AP_REDUCE_LAUNCH(RED_WMEAN);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, blocks,
                       weight / (float)n, out);
    return check_launch("reduce_mean");
}

int ap_reduce_mean_bwd(int32_t op, const float* a, const float* b, float c, int64_t n, float weight, const float* gout,
                       float* ga, ap_stream_t stream) {
    if (!a || !gout || !ga || n < 1) return fail(AP_ERR_INVALID, "reduce_mean_bwd: bad arguments");
    if (op < 0 || op > 2 || (op != RED_SQDIFF && !b)) return fail(AP_ERR_INVALID, "reduce_mean_bwd: op %d", op);
    const int blocks = stream_blocks(n);
    const float scale = weight / (float)n;
#define AP_BWD(OP) hipLaunchKernelGGL(reduce_bwd_kernel<OP>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, \
                                      (long long)n, gout, scale, ga)
    if (op == RED_SQDIFF) AP_BWD(RED_SQDIFF);
    else if (op == RED_L1) AP_BWD(RED_L1);
    else AP_BWD(RED_WMEAN);
    return check_launch("reduce_mean_bwd");
}

int ap_mask_composite(const float* a, const float* m, const float* s, int32_t N, int32_t C, int32_t Cs, int32_t HW,
                      int32_t mode, int32_t append_mask, float* out, ap_stream_t stream) {
    if (!a || !m || !out) return fail(AP_ERR_INVALID, "mask_composite: null pointer");
    if (N < 1 || N > 65535 || C < 1 || C > 4096 || HW < 1 || mode < 0 || mode > 3)
        return fail(AP_ERR_INVALID, "mask_composite: bad sizes / mode");
    if (mode == 2 && (!s || (Cs != 1 && Cs != C))) return fail(AP_ERR_INVALID, "mask_composite: blend needs s with 1 or C channels");
    append_mask = append_mask ? 1 : 0;
    hipLaunchKernelGGL(composite_kernel, dim3(std::min((HW + 255) / 256, 1024), C + append_mask, N), dim3(256), 0,
                       (hipStream_t)stream, a, m, s, C, Cs, HW, mode, append_mask, out);
    return check_launch("composite_kernel");
}

int ap_mask_composite_bwd(const float* g, const float* m, int32_t N, int32_t C, int32_t GC, int32_t HW, int32_t mode,
                          float* ga, ap_stream_t stream) {
    if (!g || !m || !ga) return fail(AP_ERR_INVALID, "mask_composite_bwd: null pointer");
    if (N < 1 || N > 65535 || C < 1 || GC < C || HW < 1 || mode < 0 || mode > 3)
        return fail(AP_ERR_INVALID, "mask_composite_bwd: bad sizes / mode");
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(std::min((HW + 255) / 256, 1024), C, N), dim3(256), 0,
                       (hipStream_t)stream, g, m, C, GC, HW, mode, ga);
    return check_launch("composite_bwd_kernel");
}

int ap_axpy(float* dst, const float* src, int64_t n, float alpha, ap_stream_t stream) {
    if (!dst || !src || n < 1) return fail(AP_ERR_INVALID, "axpy: bad arguments");
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_blocks(n) * 4), dim3(256), 0, (hipStream_t)stream, dst, src, (long long)n,
                       alpha);
    return check_launch("axpy_kernel");
}

static int crop_resize_check(const void* x, const void* win, const void* out, int N, int C, int H, int W, int OC, int chmap,
                             int OH, int OW, int mode) {
    if (!x || !win || !out) return fail(AP_ERR_INVALID, "crop_resize: null pointer");
    if (N < 1 || N > 65535 || C < 1 || H < 1 || W < 1 || OC < 1 || OC > 4 || OH < 1 || OW < 1 || mode < 0 || mode > 1)
        return fail(AP_ERR_INVALID, "crop_resize: bad sizes / mode");
    for (int k = 0; k < OC; ++k)
        if (((chmap >> (8 * k)) & 255) >= C) return fail(AP_ERR_INVALID, "crop_resize: channel map entry %d >= C", k);
    return AP_OK;
}

int ap_crop_resize_fwd(const float* x, const int32_t* win, int32_t N, int32_t C, int32_t H, int32_t W, int32_t OC,
                       int32_t chmap, int32_t OH, int32_t OW, int32_t mode, float scale, float shift, float* out,
                       ap_stream_t stream) {
    if (int rc = crop_resize_check(x, win, out, N, C, H, W, OC, chmap, OH, OW, mode)) return rc;
    hipLaunchKernelGGL(crop_resize_kernel, dim3((OH * OW + 255) / 256, OC, N), dim3(256), 0, (hipStream_t)stream, x, win, C,
                       H, W, chmap, OH, OW, mode, scale, shift, out);
    return check_launch("crop_resize_kernel");
}

int ap_crop_resize_bwd(const float* gout, const int32_t* win, int32_t N, int32_t C, int32_t H, int32_t W, int32_t OC,
                       int32_t chmap, int32_t OH, int32_t OW, int32_t mode, float scale, float* gx, ap_stream_t stream) {
    if (int rc = crop_resize_check(gout, win, gx, N, C, H, W, OC, chmap, OH, OW, mode)) return rc;
    hipError_t e = hipMemsetAsync(gx, 0, (size_t)N * C * H * W * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "crop_resize_bwd memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(crop_resize_bwd_kernel, dim3((OH * OW + 255) / 256, OC, N), dim3(256), 0, (hipStream_t)stream, gout,
                       win, C, H, W, chmap, OH, OW, mode, scale, gx);
    return check_launch("crop_resize_bwd_kernel");
}

int ap_kp_to_map(const float* lm, int32_t N, int32_t P, int32_t S, float num, float den, float radius, float* out,
                 ap_stream_t stream) {
    if (!lm || !out) return fail(AP_ERR_INVALID, "kp_to_map: null pointer");
    if (N < 1 || N > 65535 || P < 1 || P > 65535 || S < 1 || den == 0.f) return fail(AP_ERR_INVALID, "kp_to_map: bad sizes");
    hipLaunchKernelGGL(kp_to_map_kernel, dim3((S * S + 255) / 256, P, N), dim3(256), 0, (hipStream_t)stream, lm, P, S, num,
                       den, radius, out);
    return check_launch("kp_to_map_kernel");
}

int ap_flow_post(const float* flow, const float* vis, int32_t N, int32_t VC, int32_t S, int32_t OS, float gain, float num,
                 float den, float* flow_out, float* mask_out, ap_stream_t stream) {
    if (!flow || !vis || !flow_out || !mask_out) return fail(AP_ERR_INVALID, "flow_post: null pointer");
    if (N < 1 || N > 65535 || VC < 1 || S < 1 || OS < 1 || den == 0.f) return fail(AP_ERR_INVALID, "flow_post: bad sizes");
    hipLaunchKernelGGL(flow_post_kernel, dim3((OS * OS + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, flow, vis, VC, S,
                       OS, gain, num, den, flow_out, mask_out);
    return check_launch("flow_post_kernel");
}

int ap_landmark_discs(const float* lm, int32_t N, int32_t P, int32_t H, int32_t W, int32_t radius, float lo, float hi,
                      float* out, ap_stream_t stream) {
    if (!lm || !out) return fail(AP_ERR_INVALID, "landmark_discs: null pointer");
    if (N < 1 || N > 65535 || P < 1 || P > 4096 || H < 1 || W < 1 || radius < 0 || radius > kMaxDiscRadius)
        return fail(AP_ERR_INVALID, "landmark_discs: bad sizes (radius <= %d)", kMaxDiscRadius);
    hipLaunchKernelGGL(landmark_discs_kernel, dim3((H * W + 255) / 256, N), dim3(256), 2 * P * sizeof(int),
                       (hipStream_t)stream, lm, P, H, W, radius, circle_rows(radius), lo, hi, out);
    return check_launch("landmark_discs_kernel");
}

int ap_lip_line_mask(const float* lands, int32_t N, int32_t P, const int32_t* seg_a, const int32_t* seg_b, int32_t nseg,
                     int32_t size, int32_t thickness, float* out, ap_stream_t stream) {
    if (!lands || !seg_a || !seg_b || !out) return fail(AP_ERR_INVALID, "lip_line_mask: null pointer");
    if (N < 1 || P < 1 || nseg < 1 || nseg > 32 || size < 1 || thickness < 2 || thickness > 16)
        return fail(AP_ERR_INVALID, "lip_line_mask: bad sizes (1..32 segments, thickness 2..16)");
    LipSegs sg;
    for (int i = 0; i < nseg; ++i) {
        if (seg_a[i] < 0 || seg_a[i] >= P || seg_b[i] < 0 || seg_b[i] >= P) return fail(AP_ERR_INVALID, "lip_line_mask: segment %d out of range", i);
        sg.a[i] = seg_a[i];
        sg.b[i] = seg_b[i];
    }
    hipError_t e = hipMemsetAsync(out, 0, (size_t)N * size * size * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "lip_line_mask: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(lip_line_kernel, dim3((N * nseg + 63) / 64), dim3(64), 0, (hipStream_t)stream, lands, P, sg, nseg, N, size,
                       thickness, out);
    return check_launch("lip_line_kernel");
}

/* host-side table of the filled-circle rows (what ap_landmark_discs rasterises): hw[d], d = 0..radius */
int ap_circle_rows(int32_t radius, int32_t* hw) {
    if (!hw || radius < 0 || radius > kMaxDiscRadius) return fail(AP_ERR_INVALID, "circle_rows: bad radius");
    const DiscRows t = circle_rows(radius);
    for (int i = 0; i <= radius; ++i) hw[i] = t.hw[i];
    return AP_OK;
}

}  // extern "C"
