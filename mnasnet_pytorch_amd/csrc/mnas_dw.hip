// Depthwise kxk convolution (k in {3,5}, stride 1, pad k/2) forward and backward, NHWC bf16, fp32 math.
// Replaces ATen's grouped conv2d fwd/bwd for ConvBlock(groups=C) (mnasnet.py:76-81,122-125).
//
// Mapping ("lanes = channel pairs"): a thread owns one channel PAIR (one dword of the NHWC row) and a
// BH x BW block of output pixels; consecutive lanes own consecutive channel pairs, so LDS reads are
// consecutive dwords and global stores are contiguous runs of 4*CPW bytes per pixel.  The k*k*2 filter
// taps of the thread's channel pair live in VGPRs for the whole (persistent) kernel.  The input tile with
// its halo is staged through registers (act-on-load / dy-on-load applied once per element) into LDS in the
// same [y][x][c] order as global memory, so staging is plain 16-byte copies.
// Roofline: HBM (AI 4.5 flop/B for 3x3, 12.5 for 5x5); the fp32 FMA work is ~half the HBM time at peak.
#include "mnas_common.h"

#define DW_BH 4
#define DW_BW 4

struct DwCfg {
    int cpw;        // channel pairs per workgroup (multiple of 4)
    int ni, sy, sx; // strips: images x rows x cols of BHxBW blocks
    int tiles_y, tiles_x, groups;
    int ih, iw;     // LDS tile dims (with halo)
    int cblocks;
    size_t lds_tile_bytes;
};

static bool dw_pick_config(int N, int H, int W, int C, int k, int ntiles_lds, DwCfg* out) {
    const int cps = C / 2;
    double best = -1.0;
    for (int cpw = 4; cpw <= 128 && cpw <= ((cps + 3) / 4) * 4; cpw += 4) {
        const int S = 256 / cpw;
        if (S < 1) break;
        const int cblocks = (cps + cpw - 1) / cpw;
        const int maxsy = (H + DW_BH - 1) / DW_BH, maxsx = (W + DW_BW - 1) / DW_BW;
        for (int sy = 1; sy <= maxsy && sy <= S; ++sy) {
            for (int sx = 1; sx <= maxsx && sy * sx <= S; ++sx) {
                int ni = 1;
                if (sy == maxsy && sx == maxsx) ni = S / (sy * sx);
                if (ni > N) ni = N;
                if (ni < 1) ni = 1;
                const int th = sy * DW_BH, tw = sx * DW_BW;
                const int ih = th + k - 1, iw = tw + k - 1;
                const size_t lds = (size_t)ni * ih * iw * cpw * 2 * 2;
                if (lds * ntiles_lds + 8192 > 64 * 1024) continue;
                const int ty = (H + th - 1) / th, tx = (W + tw - 1) / tw, groups = (N + ni - 1) / ni;
                const double work = (double)ty * tx * groups * cblocks * 256.0 * DW_BH * DW_BW;
                const double useful = (double)N * H * W * cps;
                const double halo = (double)(ih * iw) / (double)(th * tw);
                const double score = useful / work / (0.75 + 0.25 * halo);
                if (score > best) {
                    best = score;
                    out->cpw = cpw; out->ni = ni; out->sy = sy; out->sx = sx;
                    out->tiles_y = ty; out->tiles_x = tx; out->groups = groups;
                    out->ih = ih; out->iw = iw; out->cblocks = cblocks; out->lds_tile_bytes = lds;
                }
            }
        }
    }
    return best > 0.0;
}

struct DwGeom {
    int N, H, W, C;
    int cpw, ni, sy, sx, tiles_y, tiles_x, groups, ih, iw;
};

// ---- staging -------------------------------------------------------------------------------------
// tile layout in LDS: [ni][ih][iw][cblk] bf16, cblk = 2*cpw channels (cblk % 8 == 0)
template <int KS>
__device__ __forceinline__ void dw_stage_act(const DwGeom& g, const MnasActIn& in, const float* lds_coef /*[2][cblk]*/,
                                             uint4* tile, int n0, int y0, int x0, int c0) {
    constexpr int PAD = KS / 2;
    const int cg_n = g.cpw >> 2;
    const int chunks = g.ni * g.ih * g.iw * cg_n;
    const uint4* src = (const uint4*)in.data;
    const bool has = in.scale != nullptr;
    for (int q = threadIdx.x; q < chunks; q += 256) {
        const int cgl = q % cg_n;
        int pix = q / cg_n;
        const int ix = pix % g.iw; pix /= g.iw;
        const int iy = pix % g.ih;
        const int ni = pix / g.ih;
        const int gy = y0 + iy - PAD, gx = x0 + ix - PAD, n = n0 + ni, c = c0 + cgl * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < g.N && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W && c < g.C) {
            v = src[(((size_t)n * g.H + gy) * g.W + gx) * (g.C >> 3) + (c >> 3)];
            if (has) {
                const float4* cf = (const float4*)(lds_coef + cgl * 8);
                const float4* ct = (const float4*)(lds_coef + 2 * g.cpw + cgl * 8);
                float s[8], t[8];
                *(float4*)&s[0] = cf[0]; *(float4*)&s[4] = cf[1];
                *(float4*)&t[0] = ct[0]; *(float4*)&t[4] = ct[1];
                v = act8(v, s, t);
            }
        }
        tile[q] = v;
    }
}

template <int KS>
__device__ __forceinline__ void dw_stage_dy(const DwGeom& g, const MnasGradIn& d, const float* lds_coef /*[5][cblk]*/,
                                            uint4* tile, int n0, int y0, int x0, int c0) {
    constexpr int PAD = KS / 2;
    const int cg_n = g.cpw >> 2;
    const int cblk = 2 * g.cpw;
    const int chunks = g.ni * g.ih * g.iw * cg_n;
    const uint4* gs = (const uint4*)d.g;
    const uint4* ys = (const uint4*)d.y;
    for (int q = threadIdx.x; q < chunks; q += 256) {
        const int cgl = q % cg_n;
        int pix = q / cg_n;
        const int ix = pix % g.iw; pix /= g.iw;
        const int iy = pix % g.ih;
        const int ni = pix / g.ih;
        const int gy = y0 + iy - PAD, gx = x0 + ix - PAD, n = n0 + ni, c = c0 + cgl * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < g.N && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W && c < g.C) {
            const size_t off = (((size_t)n * g.H + gy) * g.W + gx) * (g.C >> 3) + (c >> 3);
            const uint4 gv = gs[off], yv = ys[off];
            float cf[5][8];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float4* p = (const float4*)(lds_coef + r * cblk + cgl * 8);
                *(float4*)&cf[r][0] = p[0];
                *(float4*)&cf[r][4] = p[1];
            }
            float o[8];
            dy8(gv, yv, cf[0], cf[1], cf[2], cf[3], cf[4], o);
            v = pack8(o);
        }
        tile[q] = v;
    }
}

__device__ __forceinline__ void dw_load_coefs(float* lds_coef, const float* src, int rows, int C, int c0, int cblk) {
    for (int i = threadIdx.x; i < rows * cblk; i += 256) {
        const int r = i / cblk, c = c0 + i % cblk;
        lds_coef[i] = (src && c < C) ? src[(size_t)r * C + c] : 0.f;
    }
}

// ---- forward ---------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void k_dw_fwd(DwGeom g, MnasActIn in, const float* __restrict__ w,
                                                const float* __restrict__ bias, uint32_t* __restrict__ out,
                                                float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WIN_W = DW_BW + KS - 1, WIN_H = DW_BH + KS - 1;
    const int cblk = 2 * g.cpw;
    float* lds_coef = (float*)smem;                                   // [2][cblk] scale/shift
    float* lds_red = lds_coef + 2 * cblk;                              // [2][cblk] stats
    uint4* tile = (uint4*)(smem + 4 * cblk * sizeof(float));
    const uint32_t* tile32 = (const uint32_t*)tile;

    const int c0 = blockIdx.y * cblk;
    const int tid = threadIdx.x;
    const int cp = tid % g.cpw;
    const int strip = tid / g.cpw;
    const int nstrips = g.ni * g.sy * g.sx;
    const bool active = strip < nstrips;
    const int sx = strip % g.sx, sy = (strip / g.sx) % g.sy, sni = strip / (g.sx * g.sy);
    const int ch = c0 + 2 * cp;
    const bool ch_ok = ch < g.C;

    // scale/shift for the staging pass (rows [scale | shift])
    if (in.scale) {
        dw_load_coefs(lds_coef, in.scale, 1, g.C, c0, cblk);
        dw_load_coefs(lds_coef + cblk, in.shift, 1, g.C, c0, cblk);
    }
    for (int i = tid; i < 2 * cblk; i += 256) lds_red[i] = 0.f;

    float wt[KS * KS][2];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) {
        wt[t][0] = ch_ok ? w[(size_t)t * g.C + ch] : 0.f;
        wt[t][1] = ch_ok ? w[(size_t)t * g.C + ch + 1] : 0.f;
    }
    const float b0 = (bias && ch_ok) ? bias[ch] : 0.f, b1 = (bias && ch_ok) ? bias[ch + 1] : 0.f;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};

    const int th = g.sy * DW_BH, tw = g.sx * DW_BW;
    const int ntiles = g.groups * g.tiles_y * g.tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % g.tiles_x, ty = (t / g.tiles_x) % g.tiles_y, grp = t / (g.tiles_x * g.tiles_y);
        const int n0 = grp * g.ni, y0 = ty * th, x0 = tx * tw;
        __syncthreads();   // previous tile fully consumed (also orders the coef/red init on the first pass)
        dw_stage_act<KS>(g, in, lds_coef, tile, n0, y0, x0, c0);
        __syncthreads();
        if (active) {
            float acc[DW_BH][DW_BW][2];
#pragma unroll
            for (int i = 0; i < DW_BH; ++i)
#pragma unroll
                for (int j = 0; j < DW_BW; ++j) { acc[i][j][0] = b0; acc[i][j][1] = b1; }
            const int base = ((sni * g.ih + sy * DW_BH) * g.iw + sx * DW_BW) * g.cpw + cp;
#pragma unroll
            for (int r = 0; r < WIN_H; ++r) {
                float xr[WIN_W][2];
#pragma unroll
                for (int x = 0; x < WIN_W; ++x) {
                    const uint32_t u = tile32[base + (r * g.iw + x) * g.cpw];
                    xr[x][0] = bf_lo(u);
                    xr[x][1] = bf_hi(u);
                }
#pragma unroll
                for (int oy = 0; oy < DW_BH; ++oy) {
                    const int ky = r - oy;
                    if (ky < 0 || ky >= KS) continue;
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            acc[oy][ox][0] = fmaf(wt[ky * KS + kx][0], xr[ox + kx][0], acc[oy][ox][0]);
                            acc[oy][ox][1] = fmaf(wt[ky * KS + kx][1], xr[ox + kx][1], acc[oy][ox][1]);
                        }
                }
            }
            const int n = n0 + sni;
            if (ch_ok && n < g.N) {
#pragma unroll
                for (int oy = 0; oy < DW_BH; ++oy) {
                    const int gy = y0 + sy * DW_BH + oy;
                    if (gy >= g.H) continue;
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) {
                        const int gx = x0 + sx * DW_BW + ox;
                        if (gx >= g.W) continue;
                        const float v0 = acc[oy][ox][0], v1 = acc[oy][ox][1];
                        s1[0] += v0; s2[0] = fmaf(v0, v0, s2[0]);
                        s1[1] += v1; s2[1] = fmaf(v1, v1, s2[1]);
                        out[((((size_t)n * g.H + gy) * g.W + gx) * g.C + ch) >> 1] = pack_bf16(v0, v1);
                    }
                }
            }
        }
    }
    if (stats) {
        __syncthreads();
        if (active && ch_ok) {
            atomicAdd(&lds_red[2 * cp], s1[0]);
            atomicAdd(&lds_red[2 * cp + 1], s1[1]);
            atomicAdd(&lds_red[cblk + 2 * cp], s2[0]);
            atomicAdd(&lds_red[cblk + 2 * cp + 1], s2[1]);
        }
        __syncthreads();
        for (int i = tid; i < 2 * cblk; i += 256) {
            const int r = i / cblk, c = c0 + i % cblk;
            if (c < g.C) stats[((size_t)blockIdx.x * 2 + r) * g.C + c] = lds_red[i];
        }
    }
}

extern "C" int mnas_dw_fwd(const MnasDwFwd* a, void* stream) {
    if (!a || (a->k != 3 && a->k != 5) || (a->C & 7) || a->nparts < 1) return MNAS_EINVAL;
    DwCfg cfg;
    if (!dw_pick_config(a->N, a->H, a->W, a->C, a->k, 1, &cfg)) return MNAS_EINVAL;
    DwGeom g = {a->N, a->H, a->W, a->C, cfg.cpw, cfg.ni, cfg.sy, cfg.sx, cfg.tiles_y, cfg.tiles_x, cfg.groups, cfg.ih, cfg.iw};
    const size_t lds = 4 * 2 * cfg.cpw * sizeof(float) + cfg.lds_tile_bytes;
    dim3 grid(a->nparts, cfg.cblocks);
    if (a->k == 3)
        hipLaunchKernelGGL(k_dw_fwd<3>, grid, dim3(256), lds, (hipStream_t)stream, g, a->in, a->w, a->bias, (uint32_t*)a->out, a->stats);
    else
        hipLaunchKernelGGL(k_dw_fwd<5>, grid, dim3(256), lds, (hipStream_t)stream, g, a->in, a->w, a->bias, (uint32_t*)a->out, a->stats);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ---- backward: dgrad (flipped filter over dy) + wgrad (per-channel k*k reduction) in one pass -----------
template <int KS>
__global__ __launch_bounds__(256) void k_dw_bwd(DwGeom g, MnasActIn x, MnasGradIn d, const float* __restrict__ w,
                                                uint32_t* __restrict__ gin, float* __restrict__ wpartial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WIN_W = DW_BW + KS - 1, WIN_H = DW_BH + KS - 1, PAD = KS / 2;
    const int cblk = 2 * g.cpw;
    float* lds_cx = (float*)smem;                 // [2][cblk] scale/shift of x
    float* lds_cd = lds_cx + 2 * cblk;            // [5][cblk] dy coefficients
    float* lds_red = lds_cd + 5 * cblk;           // [KS*KS][cblk]
    const size_t hdr = (size_t)(7 + KS * KS) * cblk * sizeof(float);
    const size_t tile_bytes = (size_t)g.ni * g.ih * g.iw * cblk * 2;
    uint4* tile_d = (uint4*)(smem + hdr);
    uint4* tile_x = (uint4*)(smem + hdr + tile_bytes);
    const uint32_t* td32 = (const uint32_t*)tile_d;
    const uint32_t* tx32 = (const uint32_t*)tile_x;

    const int c0 = blockIdx.y * cblk;
    const int tid = threadIdx.x;
    const int cp = tid % g.cpw;
    const int strip = tid / g.cpw;
    const int nstrips = g.ni * g.sy * g.sx;
    const bool active = strip < nstrips;
    const int sx = strip % g.sx, sy = (strip / g.sx) % g.sy, sni = strip / (g.sx * g.sy);
    const int ch = c0 + 2 * cp;
    const bool ch_ok = ch < g.C;

    if (x.scale) {
        dw_load_coefs(lds_cx, x.scale, 1, g.C, c0, cblk);
        dw_load_coefs(lds_cx + cblk, x.shift, 1, g.C, c0, cblk);
    }
    dw_load_coefs(lds_cd, d.coef, 5, g.C, c0, cblk);
    for (int i = tid; i < KS * KS * cblk; i += 256) lds_red[i] = 0.f;

    float wacc[KS * KS][2];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) { wacc[t][0] = 0.f; wacc[t][1] = 0.f; }

    const int th = g.sy * DW_BH, tw = g.sx * DW_BW;
    const int ntiles = g.groups * g.tiles_y * g.tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % g.tiles_x, ty = (t / g.tiles_x) % g.tiles_y, grp = t / (g.tiles_x * g.tiles_y);
        const int n0 = grp * g.ni, y0 = ty * th, x0 = tx * tw;
        __syncthreads();
        dw_stage_dy<KS>(g, d, lds_cd, tile_d, n0, y0, x0, c0);
        dw_stage_act<KS>(g, x, lds_cx, tile_x, n0, y0, x0, c0);
        __syncthreads();
        if (!active) continue;
        const int base = ((sni * g.ih + sy * DW_BH) * g.iw + sx * DW_BW) * g.cpw + cp;
        // ---- phase 1: dgrad.  gin[p] = sum_j dy_win[p + j] * w[KS-1-j]
        {
            float wt[KS * KS][2];
#pragma unroll
            for (int k = 0; k < KS * KS; ++k) {
                wt[k][0] = ch_ok ? w[(size_t)k * g.C + ch] : 0.f;
                wt[k][1] = ch_ok ? w[(size_t)k * g.C + ch + 1] : 0.f;
            }
            float acc[DW_BH][DW_BW][2];
#pragma unroll
            for (int i = 0; i < DW_BH; ++i)
#pragma unroll
                for (int j = 0; j < DW_BW; ++j) { acc[i][j][0] = 0.f; acc[i][j][1] = 0.f; }
#pragma unroll
            for (int r = 0; r < WIN_H; ++r) {
                float xr[WIN_W][2];
#pragma unroll
                for (int xx = 0; xx < WIN_W; ++xx) {
                    const uint32_t u = td32[base + (r * g.iw + xx) * g.cpw];
                    xr[xx][0] = bf_lo(u);
                    xr[xx][1] = bf_hi(u);
                }
#pragma unroll
                for (int oy = 0; oy < DW_BH; ++oy) {
                    const int jy = r - oy;
                    if (jy < 0 || jy >= KS) continue;
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                        for (int jx = 0; jx < KS; ++jx) {
                            const int k = (KS - 1 - jy) * KS + (KS - 1 - jx);
                            acc[oy][ox][0] = fmaf(wt[k][0], xr[ox + jx][0], acc[oy][ox][0]);
                            acc[oy][ox][1] = fmaf(wt[k][1], xr[ox + jx][1], acc[oy][ox][1]);
                        }
                }
            }
            const int n = n0 + sni;
            if (ch_ok && n < g.N) {
#pragma unroll
                for (int oy = 0; oy < DW_BH; ++oy) {
                    const int gy = y0 + sy * DW_BH + oy;
                    if (gy >= g.H) continue;
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) {
                        const int gx = x0 + sx * DW_BW + ox;
                        if (gx >= g.W) continue;
                        gin[((((size_t)n * g.H + gy) * g.W + gx) * g.C + ch) >> 1] = pack_bf16(acc[oy][ox][0], acc[oy][ox][1]);
                    }
                }
            }
        }
        // ---- phase 2: wgrad.  dW[ky][kx] += sum_p dy[p] * a_win[p + (ky,kx)]
        {
            float dyc[DW_BH][DW_BW][2];
#pragma unroll
            for (int oy = 0; oy < DW_BH; ++oy)
#pragma unroll
                for (int ox = 0; ox < DW_BW; ++ox) {
                    const uint32_t u = td32[base + ((oy + PAD) * g.iw + ox + PAD) * g.cpw];
                    dyc[oy][ox][0] = bf_lo(u);
                    dyc[oy][ox][1] = bf_hi(u);
                }
#pragma unroll
            for (int r = 0; r < WIN_H; ++r) {
                float xr[WIN_W][2];
#pragma unroll
                for (int xx = 0; xx < WIN_W; ++xx) {
                    const uint32_t u = tx32[base + (r * g.iw + xx) * g.cpw];
                    xr[xx][0] = bf_lo(u);
                    xr[xx][1] = bf_hi(u);
                }
#pragma unroll
                for (int oy = 0; oy < DW_BH; ++oy) {
                    const int ky = r - oy;
                    if (ky < 0 || ky >= KS) continue;
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            wacc[ky * KS + kx][0] = fmaf(dyc[oy][ox][0], xr[ox + kx][0], wacc[ky * KS + kx][0]);
                            wacc[ky * KS + kx][1] = fmaf(dyc[oy][ox][1], xr[ox + kx][1], wacc[ky * KS + kx][1]);
                        }
                }
            }
        }
    }
    __syncthreads();
    if (active && ch_ok) {
#pragma unroll
        for (int k = 0; k < KS * KS; ++k) {
            atomicAdd(&lds_red[k * cblk + 2 * cp], wacc[k][0]);
            atomicAdd(&lds_red[k * cblk + 2 * cp + 1], wacc[k][1]);
        }
    }
    __syncthreads();
    for (int i = tid; i < KS * KS * cblk; i += 256) {
        const int k = i / cblk, c = c0 + i % cblk;
        if (c < g.C) wpartial[((size_t)blockIdx.x * KS * KS + k) * g.C + c] = lds_red[i];
    }
}

extern "C" int mnas_dw_bwd(const MnasDwBwd* a, void* stream) {
    if (!a || (a->k != 3 && a->k != 5) || (a->C & 7) || a->nparts < 1) return MNAS_EINVAL;
    DwCfg cfg;
    if (!dw_pick_config(a->N, a->H, a->W, a->C, a->k, 2, &cfg)) return MNAS_EINVAL;
    DwGeom g = {a->N, a->H, a->W, a->C, cfg.cpw, cfg.ni, cfg.sy, cfg.sx, cfg.tiles_y, cfg.tiles_x, cfg.groups, cfg.ih, cfg.iw};
    const size_t lds = (size_t)(7 + a->k * a->k) * 2 * cfg.cpw * sizeof(float) + 2 * cfg.lds_tile_bytes;
    dim3 grid(a->nparts, cfg.cblocks);
    if (a->k == 3)
        hipLaunchKernelGGL(k_dw_bwd<3>, grid, dim3(256), lds, (hipStream_t)stream, g, a->x, a->dy, a->w, (uint32_t*)a->gin, a->wpartial);
    else
        hipLaunchKernelGGL(k_dw_bwd<5>, grid, dim3(256), lds, (hipStream_t)stream, g, a->x, a->dy, a->w, (uint32_t*)a->gin, a->wpartial);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
