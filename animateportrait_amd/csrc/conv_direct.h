// conv_direct.h -- direct (VALU) convolution for layers with 1..4 output channels.
//
// The generator ends in ReflectionPad2d(3) + Conv2d(ngf, output_nc, 7) + Tanh with output_nc = 1
// (drawing) or 3 (cartoon) (reference: Module2/models/networks.py:1277-1279).  With a single output
// channel an MFMA tile would be 31/32 padding, so this layer runs on the vector ALUs instead: every
// lane owns a 1 x 4 strip of output pixels, the input tile (with halo) is staged through LDS once per
// channel chunk with the same fused loader as the implicit-GEMM kernel (concat segments,
// InstanceNorm + ReLU of the producer, reflection / zero padding), window rows are read as aligned
// ds_read_b128 and the weights are wave-uniform scalars (s_load -> SGPR operands of v_fmac).
#pragma once
#include "conv_igemm.h"

namespace apamd {

struct DirectKParams {
    SrcSeg seg[kMaxSeg];
    int nseg;
    int N, H, W, Cout, OH, OW, pad, pad_mode;
    float* y;
    const float* wp;     // [cin_pad][K][K][COP]
    const float* bias;
    int act;
    float* stats;        // [N][Cout][stat_tiles][2] or null
    int stat_tiles;
    int nchunks, cin_pad, tiles_x, tiles_y;
};

template <int K_, int COP_>
struct DirectCfg {
    static constexpr int K = K_, COP = COP_;
    static constexpr int CI = 4;
    static constexpr int WPE = COP_ == 1 ? 3 : 2;        // waves per SIMD the register budget is set for (three 53 KB tiles fit a CU)
    static constexpr int TH = 16, TW = 64;
    static constexpr int R = K / 2;                      // halo (pad == R is required: 'same' convolution)
    static constexpr int LPAD = (4 - R % 4) % 4;         // left padding so every strip window is 16-B aligned
    static constexpr int IH = TH + K - 1;
    static constexpr int IWS = ((LPAD + TW + K - 1) + 3) / 4 * 4;   // LDS row stride (floats)
    static constexpr int PLANE = IH * IWS;
    static constexpr int XE = CI * PLANE;
    static constexpr int NV = (LPAD + 4 + K - 1 + 3) / 4;           // float4 reads per window row
    static size_t lds_floats(int nbuf, int cin_pad) { return (size_t)nbuf * XE + 2 * (size_t)cin_pad + 64 * COP; }
};

template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C::WPE, C::WPE))) void conv_direct_f32(const DirectKParams p) {
    constexpr int K = C::K, COP = C::COP, CI = C::CI, IWS = C::IWS, PLANE = C::PLANE, XE = C::XE, NV = C::NV;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    int b = (int)xcd_logical_block(gridDim.x, blockIdx.x);   // neighbouring tiles (shared halo rows) on one XCD's L2
    const int tix = b % p.tiles_x; b /= p.tiles_x;
    const int tiy = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int oy0 = tiy * C::TH, ox0 = tix * C::TW;
    const int H = p.H, W = p.W, HW = H * W;
    const int nbuf = p.nchunks > 1 ? 2 : 1;
    float* const xbuf = smem;
    float* const s_mean = smem + nbuf * XE;
    float* const s_rstd = s_mean + p.cin_pad;
    float* const s_red = s_rstd + p.cin_pad;

    auto seg_of = [&](int chunk) {
        int s = 0;
        if (p.nseg > 1 && chunk >= p.seg[1].chunk_begin) s = 1;
        if (p.nseg > 2 && chunk >= p.seg[2].chunk_begin) s = 2;
        return s;
    };
    for (int c = tid; c < p.cin_pad; c += 256) {
        const int s = seg_of(c / CI);
        const int cs = c - p.seg[s].chunk_begin * CI;
        float m = 0.f, r = 1.f;
        if (p.seg[s].mean != nullptr && cs < p.seg[s].C) {
            m = p.seg[s].mean[n * p.seg[s].C + cs];
            r = p.seg[s].rstd[n * p.seg[s].C + cs];
        }
        s_mean[c] = m;
        s_rstd[c] = r;
    }
    __syncthreads();

    // loader geometry, identical for every chunk.  Wave w stages channel w of the chunk (CI = 4 waves): the plane pointer,
    // the InstanceNorm constants and the activation are then wave-uniform, and an element costs a load, four vector
    // instructions and an LDS store.  (Round 2 spread the [CI][IH][IWS] elements over the 256 threads: channel, validity,
    // norm and activation were decided per element with run-time branches -- ~25 instructions per staged element, more
    // than the 98 packed FMAs per channel the kernel exists for.)  Column c <-> x = ox0 - pad - LPAD + c.
    static_assert(CI == 4, "one staging wave per channel of the chunk");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int NE = (PLANE + 63) / 64;
    int goff[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = lane + k * 64;
        const int ly = e / IWS, lx = e - ly * IWS;
        int gy = oy0 - p.pad + ly, gx = ox0 - p.pad - C::LPAD + lx;
        bool ok = e < PLANE;
        if (p.pad_mode == 1) {
            gy = reflect_clamp(gy, H);
            gx = reflect_clamp(gx, W);
        } else {
            ok = ok && gy >= 0 && gy < H && gx >= 0 && gx < W;
        }
        goff[k] = ok ? gy * W + gx : -1;
    }
    float xr[NE];
    auto issue = [&](int chunk) {
        const int s = seg_of(chunk);
        const int cbase = (chunk - p.seg[s].chunk_begin) * CI;
        if (cbase + wave < p.seg[s].C) {                       // wave-uniform
            const float* base = p.seg[s].data + ((long long)n * p.seg[s].C + cbase + wave) * HW;
#pragma unroll
            for (int k = 0; k < NE; ++k) xr[k] = base[goff[k] >= 0 ? goff[k] : 0];   // clamped (always legal) address
        }
    };
    auto commit = [&](int chunk, float* dst) {
        const int s = seg_of(chunk);
        const int cbase = (chunk - p.seg[s].chunk_begin) * CI;
        const bool on = cbase + wave < p.seg[s].C;
        const int act = p.seg[s].act;
        const float slope = act == 1 ? 0.f : (act == 2 ? 0.2f : 1.f);      // act(t) = max(t, slope * t)
        const float m = s_mean[chunk * CI + wave], r = s_rstd[chunk * CI + wave];   // (0, 1) for a plain segment
        float* const d = dst + wave * PLANE + lane;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const float t = (xr[k] - m) * r;
            float v = fmaxf(t, slope * t);
            v = (on && goff[k] >= 0) ? v : 0.f;
            if (k + 1 < NE || lane + k * 64 < PLANE) d[k * 64] = v;
        }
    };

    float acc[COP][4];
#pragma unroll
    for (int c = 0; c < COP; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v accA01 = {0.f, 0.f}, accA23 = {0.f, 0.f}, accB12 = {0.f, 0.f};      // COP == 1: see the tap loop
    float accB0 = 0.f, accB3 = 0.f;

    issue(0);
    commit(0, xbuf);
    __syncthreads();
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < p.nchunks;
        if (more) issue(chunk + 1);
        const float* X = xbuf + cur * XE + ty * IWS + tx * 4;
        // constant address space: the weights stay scalar loads (s_load -> SGPR operands) although the hand-issued
        // LDS reads below count as memory-clobbering statements for the compiler
        typedef const float __attribute__((address_space(4))) cfloat;
        const cfloat* wc = (const cfloat*)(uintptr_t)(p.wp + (long long)chunk * CI * K * K * COP);
#pragma unroll 1
        for (int ci = 0; ci < CI; ++ci) {   // rolled: 49 weights per channel fit the SGPR file, 196 do not
            // one output channel: the 49 weights of the channel are fetched (scalar loads) ahead of the window reads, so
            // that one wait covers both
            float wreg[COP == 1 ? K * K : 1];
            if constexpr (COP == 1) {
#pragma unroll
                for (int t = 0; t < K * K; ++t) wreg[t] = wc[ci * K * K + t];
            }
            // The window rows are read as whole 16-byte lanes, issued here by hand: left to itself the compiler drops
            // the unused edge elements and splits every float4 into b64 / b32 pieces (48 LDS instructions per channel
            // instead of 21, at a 16-byte lane stride that those narrower reads serve with bank conflicts): 396 us
            // against 240 us for this form at B=16.  All K rows of a channel are in flight before the single wait.
            // (No "memory" clobber: the statements are ordered against the barriers that publish / recycle the
            // stage buffer as side-effecting asm; a clobber would turn the scalar weight loads into vector loads.)
            static_assert(NV == 3 && K == 7, "hand-issued window reads are written for 7x7");
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v q[K][NV];
            {
                const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)(X + ci * PLANE);
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const unsigned a = a0 + ky * IWS * 4;
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32"
                                 : "=&v"(q[ky][0]), "=&v"(q[ky][1]), "=&v"(q[ky][2]) : "v"(a));
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]),
                               "+v"(q[2][0]), "+v"(q[2][1]), "+v"(q[2][2]), "+v"(q[3][0]), "+v"(q[3][1]), "+v"(q[3][2]),
                               "+v"(q[4][0]), "+v"(q[4][1]), "+v"(q[4][2]), "+v"(q[5][0]), "+v"(q[5][1]), "+v"(q[5][2]),
                               "+v"(q[6][0]), "+v"(q[6][1]), "+v"(q[6][2]));
            }
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float win[NV * 4];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    win[v * 4 + 0] = q[ky][v].x; win[v * 4 + 1] = q[ky][v].y; win[v * 4 + 2] = q[ky][v].z; win[v * 4 + 3] = q[ky][v].w;
                }
                if constexpr (COP == 1) {
                    // One output channel: packed FMAs want their two window elements in an EVEN-aligned register pair.  The four
                    // outputs of a strip are therefore summed in two groupings -- taps whose window starts on an even register
                    // add the pairs (0,1), (2,3); the others add the pair (1,2) and outputs 0 and 3 -- and the two sets are added
                    // once at the end.  (Accumulated as (0,1), (2,3) for every tap, the odd-aligned taps cost one or two register
                    // moves per packed FMA: ~70 moves around the 98 FMAs of a channel; now ~28.  Same box, round 6: 175-181 ->
                    // 169-174 us for the B = 16 layer alone.)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const float w = wreg[ky * K + kx];                      // wave-uniform -> SGPR
                        constexpr int LP = C::LPAD;
                        const int b = LP + kx;
                        if ((LP + kx) % 2 == 0) {
                            const f2v p01 = {win[b], win[b + 1]}, p23 = {win[b + 2], win[b + 3]};
                            const f2v w2 = {w, w};
                            accA01 = w2 * p01 + accA01;
                            accA23 = w2 * p23 + accA23;
                        } else {
                            const f2v p12 = {win[b + 1], win[b + 2]};
                            const f2v w2 = {w, w};
                            accB0 = fmaf(w, win[b], accB0);
                            accB12 = w2 * p12 + accB12;
                            accB3 = fmaf(w, win[b + 3], accB3);
                        }
                    }
                } else {
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
#pragma unroll
                        for (int c = 0; c < COP; ++c) {
                            const float w = wc[((ci * K + ky) * K + kx) * COP + c];   // wave-uniform -> SGPR
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[c][j] = fmaf(w, win[C::LPAD + j + kx], acc[c][j]);
                        }
                    }
                }
            }
        }
        if (more) commit(chunk + 1, xbuf + (cur ^ 1) * XE);
        __syncthreads();
    }

    if constexpr (COP == 1) {
        acc[0][0] = accA01.x + accB0;
        acc[0][1] = accA01.y + accB12.x;
        acc[0][2] = accA23.x + accB12.y;
        acc[0][3] = accA23.y + accB3;
    }
    const int oy = oy0 + ty, ox = ox0 + tx * 4;
#pragma unroll
    for (int c = 0; c < COP; ++c) {
        float s = 0.f, q2 = 0.f;
        if (c < p.Cout) {
            const float bv = p.bias ? p.bias[c] : 0.f;
            float* dst = p.y + (((long long)n * p.Cout + c) * p.OH + oy) * p.OW + ox;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[c][j] + bv;
                if (oy < p.OH && ox + j < p.OW) {
                    s += v;
                    q2 += v * v;
                    dst[j] = apply_act(v, p.act);
                }
            }
        }
        if (p.stats != nullptr) {
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                s += __shfl_xor(s, sh, 64);
                q2 += __shfl_xor(q2, sh, 64);
            }
            if ((tid & 63) == 0) {
                s_red[((tid >> 6) * COP + c) * 2] = s;
                s_red[((tid >> 6) * COP + c) * 2 + 1] = q2;
            }
        }
    }
    if (p.stats != nullptr) {
        __syncthreads();
        if (tid < p.Cout) {
            float s = 0.f, q2 = 0.f;
            for (int w = 0; w < 4; ++w) {
                s += s_red[(w * COP + tid) * 2];
                q2 += s_red[(w * COP + tid) * 2 + 1];
            }
            float* d = p.stats + (((long long)n * p.Cout + tid) * p.stat_tiles + tiy * p.tiles_x + tix) * 2;
            d[0] = s;
            d[1] = q2;
        }
    }
}

}  // namespace apamd
