// conv_bf16x3_sb.h -- the split-bf16 3x3 stride-1 convolution with ONE LDS stage per workgroup and TWO workgroups per CU.
//
// conv_bf16x3 keeps one workgroup per CU (two 76 KB stages) and hides the LDS-DMA of stage g + 2 under the MFMAs of stage g;
// what it cannot hide is its own epilogue -- all persistent workgroups finish a tile together, 33 MB leave at once and the
// matrix pipe idles (19 of ~195 us per launch, APAMD_ABLATE = 8) -- nor the prologue of a launch.  Here a workgroup owns a
// single stage (DMA, wait, multiply, repeat: nothing overlaps INSIDE it) and the CU's second workgroup fills its gaps: while
// one waits for its DMA or writes its output, the other multiplies.  Same operands, same tile (64 couts x 16 rows x 32
// columns), same arithmetic and accumulation order per tile as conv_bf16x3 -- the results are bit-identical.
// Experiment of round 4 (HISTORY.md section 3.12); selected by APAMD_CONV_SB=1.
#pragma once
#include "conv_bf16x3.h"

namespace apamd {

template <class C>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_sb(const ConvKParams p) {
    static_assert(C::K == 3 && C::S == 1 && C::PARTS == 2 && !C::ROW && C::WCO == 1, "3x3 stride-1 split-bf16 tile");
    constexpr int K = 3, T = 9, MT = C::MT, NT = C::NT, IW = C::IW, PLANE = C::PLANE, NIT = C::NIT, CO_TILE = C::CO_TILE, XP = C::XP;
    constexpr int W_SLOTS = 2 * T * 2 * CO_TILE, STAGE = W_SLOTS + C::X_SLOTS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* const smem = reinterpret_cast<uint4*>(smem_raw);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wpx = wave;
    const int H = p.H, W = p.W, HW = H * W;
    const int nreal = p.cin_pad >> 4;

    int tile, tile_end, tile_step;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int nx = G < 8 ? G : 8;
        const int xcd = b % nx, idx = b / nx;
        const int ntl = p.N * p.tiles_y * p.tiles_x * p.co_tiles;
        const int q = ntl / nx, r = ntl % nx;
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile_step = (G - xcd + nx - 1) / nx;
        tile = base + idx;
        tile_end = base + q + (xcd < r ? 1 : 0);
    }
    if (tile >= tile_end) return;
    // the second half of the grid (the workgroups that land beside an already running one) starts half a stage late, so that the
    // two workgroups of a CU alternate between DMA wait and multiplication instead of marching in step
    if ((int)blockIdx.x >= (int)gridDim.x / 2) {
        for (int i = 0; i < p.fn_debug; ++i) __builtin_amdgcn_s_sleep(64);
    }

    int pgeo[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = tid + k * 256;
        const int kg = it >= PLANE ? 1 : 0;
        const int pix = it - kg * PLANE;
        const int ly = pix / IW, lx = pix - ly * IW;
        pgeo[k] = it < 2 * PLANE ? ((kg << 15) | (ly << 8) | lx) : -1;
    }
    constexpr int NWP = (W_SLOTS / 64 + 3) / 4;
    unsigned woff[NWP];
#pragma unroll
    for (int j = 0; j < NWP; ++j) woff[j] = ((j * 4 + wave) * 64 + lane) * 16;

    const int a_slot = half * CO_TILE + l32;
    const int b_slot = W_SLOTS + half * PLANE + (wpx * NT) * IW + l32;

    for (; tile < tile_end; tile += tile_step) {
        const int cot = tile % p.co_tiles;
        int t_ = tile / p.co_tiles;
        const int tx = t_ % p.tiles_x;
        t_ /= p.tiles_x;
        const int ty = t_ % p.tiles_y, n = t_ / p.tiles_y;
        int goff[NIT];
        {
            const int iy0 = ty * C::TH + p.dy0, ix0 = tx * 32 + p.dx0;
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
                goff[k] = (ok ? ((pgeo[k] >> 15) & 1) * (HW + 1) + gy * W + gx : HW) * 16;
            }
        }
        f32x16 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

        for (int c = 0; c < nreal; ++c) {
            __syncthreads();                    // every wave is done with the previous stage (or the epilogue patches)
            {
                int s = 0;
                if (p.nseg > 1 && c >= p.seg[1].chunk_begin) s = 1;
                if (p.nseg > 2 && c >= p.seg[2].chunk_begin) s = 2;
                const int cg0 = (c - p.seg[s].chunk_begin) * 2, CG = p.seg[s].C >> 3;
                const unsigned char* xs = reinterpret_cast<const unsigned char*>(p.seg[s].data);
                const unsigned char* xh = xs + ((long long)(n * 2 + 0) * CG + cg0) * (HW + 1) * 16;
                const unsigned char* xl = xs + ((long long)(n * 2 + 1) * CG + cg0) * (HW + 1) * 16;
                const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wp + ((long long)cot * p.nchunks + c) * p.wfloats);
                const unsigned xdst = lds0 + (W_SLOTS + wave * 64) * 16, wdst = lds0 + (wave * 64) * 16;
#pragma unroll
                for (int part = 0; part < 2; ++part)
#pragma unroll
                    for (int k = 0; k < NIT; ++k)
                        if (k * 256 + 3 * 64 < XP || k * 256 + wave * 64 < XP)
                            glds16_sv(part ? xl : xh, (unsigned)goff[k], xdst + (part * XP + k * 256) * 16);
#pragma unroll
                for (int jj = 0; jj < NWP; ++jj)
                    if (jj * 4 + 3 < W_SLOTS / 64 || jj * 4 + wave < W_SLOTS / 64) glds16_sv(wsrc, woff[jj], wdst + jj * 4096);
            }
            dma_wait_all();
            __syncthreads();
            bf16x8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
            auto fetch = [&](int t, int buf) __attribute__((always_inline)) {
                const uint4* Wc = smem + a_slot;
                const uint4* Xc = smem + b_slot + (t / K) * IW + (t % K);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((0 * T + t) * 2) * CO_TILE + m * 32);
                    al[buf][m] = *reinterpret_cast<const bf16x8*>(Wc + ((1 * T + t) * 2) * CO_TILE + m * 32);
                }
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    bh[buf][q] = *reinterpret_cast<const bf16x8*>(Xc + q * IW);
                    bl[buf][q] = *reinterpret_cast<const bf16x8*>(Xc + XP + q * IW);
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int cb = t & 1;
                if (t + 1 < T) fetch(t + 1, cb ^ 1);
                // the same product order as conv_bf16x3 (small terms first, round-robin over the accumulators)
#pragma unroll
                for (int i = 0; i < 3 * MT * NT; ++i) {
                    const int g = i / (MT * NT), m = (i % (MT * NT)) / NT, q = i % NT;
                    if (g == 0) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb][m], bh[cb][q], acc[m][q], 0, 0, 0);
                    else if (g == 1) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][m], bl[cb][q], acc[m][q], 0, 0, 0);
                    else acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][m], bh[cb][q], acc[m][q], 0, 0, 0);
                }
                if (t + 1 < T) {
                    constexpr int NM = 3 * MT * NT, NRD = 2 * (MT + NT), PER = (NM + NRD - 1) / NRD;
#pragma unroll
                    for (int i = 0; i < NRD; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                    }
                }
            }
        }
        __syncthreads();                        // the stage is free: the transposition patches and statistics slots overlay it

        // ---- epilogue (conv_bf16x3's plain form): 32 x 32 tiles through private LDS patches, 16-byte stores, row sums
        constexpr int TS = 36, NP = 2;
        float* const epi = reinterpret_cast<float*>(smem);
        float* const patch0 = epi + wave * (NP * 32 * TS);
        float* const sred = epi + 4 * NP * 32 * TS;
        const int oy0 = ty * C::TH, ox0 = tx * 32;
        const int co_base = cot * CO_TILE;
        const bool want_stats = p.stats != nullptr;
        const int prow = lane >> 3, pcol = (lane & 7) * 4;
        const int oxv = ox0 + pcol;
        const bool full = (p.o_rstride & 3) == 0 && ox0 + 32 <= p.OW;
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
                cbase[ps] = (long long)n * p.o_nstride + (long long)co * p.o_cstride + oxv;
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
                    const long long rowoff = (long long)oy * p.o_rstride;
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const float4 v = *reinterpret_cast<const float4*>(patch0 + b * (32 * TS) + (ps * 8 + prow) * TS + pcol);
                        const float bv = bvv[ps];
                        const float vv[4] = {v.x + bv, v.y + bv, v.z + bv, v.w + bv};
                        if (cokc[ps] && oy < p.OH) {
                            float* dst = p.y + cbase[ps] + rowoff;
                            if (full) {
                                s4[ps] += (vv[0] + vv[1]) + (vv[2] + vv[3]);
                                q4[ps] += (vv[0] * vv[0] + vv[1] * vv[1]) + (vv[2] * vv[2] + vv[3] * vv[3]);
                                *reinterpret_cast<float4*>(dst) = make_float4(apply_act(vv[0], p.act), apply_act(vv[1], p.act),
                                                                               apply_act(vv[2], p.act), apply_act(vv[3], p.act));
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (oxv + j < p.OW) {
                                        s4[ps] += vv[j];
                                        q4[ps] += vv[j] * vv[j];
                                        dst[j] = apply_act(vv[j], p.act);
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
                        float* d = sred + ((wpx * CO_TILE) + m * 32 + ps * 8 + prow) * 2;
                        d[0] = s;
                        d[1] = q2;
                    }
                }
            }
        }
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
                    float* d = p.stats + (((long long)n * p.Cout + co) * p.stat_tiles + p.stat_tile_off + ty * p.tiles_x + tx) * 2;
                    d[0] = s;
                    d[1] = q2;
                }
            }
        }
    }
}

}  // namespace apamd
