// Depthwise kxk convolution (k in {3,5}, stride 1, pad k/2) forward and backward, NHWC bf16, fp32 math.
// Replaces ATen's grouped conv2d fwd/bwd for ConvBlock(groups=C) (mnasnet.py:76-81,122-125).
//
// Structure ("vertical sweep over an LDS row ring"):
//   * a workgroup owns ONE image, a column strip of TW = 4*sx output columns and a block of 2*cpw channels
//     (all channels of the pixel when C <= 144, otherwise >= 64 channels = 128-byte runs), and sweeps the
//     strip top to bottom G = 4 output rows at a time;
//   * the input rows it needs live in an 8-row ring in LDS, in the same [row][x][c] order as HBM, so every
//     input element is fetched from HBM ONCE per strip (halo only horizontally: (TW+k-1)/TW) and staged with
//     plain 16-byte loads/stores; BatchNorm+ReLU of the producer ("act-on-load") or BatchNorm/ReLU backward
//     ("dy-on-load") is applied once per element on the way into LDS;
//   * staging is division-free in steady state: each thread's (row, x, channel-group) slots are computed once
//     per kernel, its channel group never changes, so the per-channel coefficients sit in registers, and the
//     <= 6 loads of a step are issued back to back before any of them is consumed (bytes in flight);
//   * compute: thread = (channel pair, 4-column strip), consecutive lanes = consecutive channel pairs
//     (conflict-free dword LDS reads, contiguous global stores); the k*k*2 taps of its channel pair are in VGPRs;
//     rows are STREAMED: each input row is read from LDS once (k+3 dwords) and scattered into a register ring of
//     k partial output rows; the oldest ring entry is complete after every input row and is emitted;
//   * BatchNorm partial statistics (forward) and the k*k weight-gradient sums (backward) are accumulated in
//     registers over the whole sweep and written once per workgroup (no atomics in HBM, deterministic);
//   * backward = two launches of this structure: dgrad (the forward kernel with dy-on-load staging and the
//     flipped filter) and wgrad.  Fusing them needs 2 x k*k*2 persistent VGPRs per thread (filter + gradient
//     sums) and halves occupancy; measured trade-off recorded in DESIGN.md.
// Roofline: HBM (AI 4.5 flop/B for 3x3, 12.5 for 5x5 forward; backward moves 4 tensors for 2x the FMAs).
#include "mnas_common.h"

#define DW_G 4          // output rows per sweep step
#define DW_BW 4         // output columns per thread
#define DW_RR 8         // ring rows (>= G + k - 1)
#define DW_MAXCOL 2     // column positions (x, channel group) a thread stages per row

struct DwArgs {
    int N, H, W, C;
    int cpw, sx, nthreads, cgn, iw, ps;     // ps: LDS pixel stride (dwords), multiple of 4
    int strips_x, cblocks, items, geff;     // geff: workgroups that take items (multiple of cblocks)
    int rc, ncol, tcol, rpp;                // staging: chunks per row, columns per thread, threads per column pass, rows per pass
};

static bool dw_pick(int N, int H, int W, int C, int k, int nrings, DwArgs* a) {
    const int cps = C / 2;
    // Search (channel pairs per workgroup, column strips).  Whole pixel when it fits (cps <= 72), otherwise channel
    // blocks of >= 32 pairs (>= 128-byte runs per pixel).  Score = lane utilisation x occupancy / halo.
    int best_sx = 0, best_cpw = 0; double best = -1.0;
    const int maxsx = (W + DW_BW - 1) / DW_BW;
    for (int cpw = 4; cpw <= 128 && cpw <= cps; cpw += 4) {
        if (cps <= 72) { if (cpw != cps) continue; }
        else if (cpw < 32) continue;
        const int cblocks = (cps + cpw - 1) / cpw;
        const int cgn = cpw / 4;
        const int ps = (cpw % 32 == 0) ? cpw : cpw + 4;
        for (int sx = 1; sx <= maxsx && sx * cpw <= 256; ++sx) {
            const int tw = sx * DW_BW, iw = tw + k - 1;
            const size_t lds = (size_t)nrings * DW_RR * iw * ps * 4;
            if (lds > 56 * 1024) continue;
            const int nth = ((sx * cpw + 63) / 64) * 64;
            const int rc = iw * cgn;
            const int tcol = (nth / cgn) * cgn;
            if (rc > DW_MAXCOL * tcol) continue;
            const int strips = (W + tw - 1) / tw;
            const double util = (double)W / (strips * tw) * (double)(sx * cpw) / nth * (double)cps / (cblocks * cpw);
            const double halo = (double)iw / tw;
            const double occ = 0.55 + 0.45 * nth / 256.0;        // tiny workgroups starve the CU of waves
            const double score = util * occ / (0.6 + 0.4 * halo);
            if (score > best) { best = score; best_sx = sx; best_cpw = cpw; }
        }
    }
    if (best_sx == 0) return false;
    const int cpw = best_cpw, cgn = cpw / 4;
    a->N = N; a->H = H; a->W = W; a->C = C;
    a->cpw = cpw; a->sx = best_sx; a->cgn = cgn; a->ps = (cpw % 32 == 0) ? cpw : cpw + 4;
    a->nthreads = ((best_sx * cpw + 63) / 64) * 64;
    a->iw = best_sx * DW_BW + k - 1;
    a->strips_x = (W + best_sx * DW_BW - 1) / (best_sx * DW_BW);
    a->cblocks = (cps + cpw - 1) / cpw;
    a->items = N * a->strips_x * a->cblocks;
    a->rc = a->iw * cgn;
    if (a->rc <= a->nthreads) {          // several rows per pass, one column position per thread
        a->ncol = 1; a->tcol = a->rc; a->rpp = a->nthreads / a->rc;
        if (a->rpp > DW_G) a->rpp = DW_G;
    } else {                             // one row per pass, up to DW_MAXCOL column positions per thread
        a->tcol = (a->nthreads / cgn) * cgn; a->ncol = (a->rc + a->tcol - 1) / a->tcol; a->rpp = 1;
    }
    return true;
}

__device__ __forceinline__ int dw_slot(int image_row) { return (image_row + DW_RR) & (DW_RR - 1); }   // rows >= -RR

// per-thread staging plan (tile-invariant): its column position(s) and row lane
struct DwStagePlan {
    int goff[DW_MAXCOL];     // element offset (uint4 units) of the chunk inside an image row, relative to x0-PAD: ix*C8 + c8
    int loff[DW_MAXCOL];     // dword offset inside a ring row: ix*ps + cgl*4
    int ix[DW_MAXCOL];
    int rl;                  // row lane (0..rpp-1), or -1 if the thread does not stage
    int cgl;
};

__device__ __forceinline__ void dw_make_plan(const DwArgs& a, DwStagePlan& p) {
    const int tid = threadIdx.x;
    int col0;
    if (a.ncol == 1) { p.rl = tid / a.rc; col0 = tid - p.rl * a.rc; if (p.rl >= a.rpp) p.rl = -1; }
    else { p.rl = (tid < a.tcol) ? 0 : -1; col0 = tid; }
    p.cgl = col0 % a.cgn;
#pragma unroll
    for (int j = 0; j < DW_MAXCOL; ++j) {
        const int col = col0 + j * a.tcol;
        const int ix = (col < a.rc && j < a.ncol) ? col / a.cgn : -1;
        p.ix[j] = ix;
        p.goff[j] = ix * (a.C >> 3);
        p.loff[j] = ix * a.ps + p.cgl * 4;
    }
}

// Stage `nrows` (<= G) image rows starting at image row `row0` (may be negative / beyond H: zero rows) into
// the ring.  MODE 0: act-on-load, coefficients lds_c = [2][cblk] (scale, shift).  MODE 1: dy-on-load,
// lds_c = [5][cblk] (s,t,c1,c2,c3).  Slots t = 0..G*ncol-1 -> (row group t/ncol, column t%ncol) are processed
// in batches of 4: the batch's loads are issued back to back, then transformed and written to LDS.
template <int KS, int MODE, int BATCH = 4>
__device__ __forceinline__ void dw_stage(const DwArgs& a, const DwStagePlan& p, uint32_t* ring, const uint4* __restrict__ src0,
                                         const uint4* __restrict__ src1, const float* lds_c, bool has_coef, int n, int row0,
                                         int nrows, int x0, int c0) {
    constexpr int PAD = KS / 2;
    if (p.rl < 0) return;
    const int cblk = 2 * a.cpw;
    const int c = c0 + p.cgl * 8;
    const bool c_ok = c < a.C;
    const int C8 = a.C >> 3;
    const int rowstride = a.W * C8;
    const int xbase = (x0 - PAD) * C8 + (c >> 3);
    const int nslots = DW_G * a.ncol;
    const int jshift = a.ncol - 1;           // ncol in {1,2}
#pragma unroll
    for (int tb = 0; tb < DW_G * DW_MAXCOL; tb += BATCH) {
        if (tb >= nslots) break;
        uint4 v0[BATCH], v1[BATCH];
        bool inb[BATCH], st[BATCH];
        int lofs[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int t = tb + u;
            const int q = t >> jshift, j = t & jshift;
            const int row = q * a.rpp + p.rl;
            const int gy = row0 + row;
            const int ixj = j ? p.ix[1] : p.ix[0];
            const int gx = x0 - PAD + ixj;
            st[u] = (q * a.rpp < nrows) && (row < nrows) && ixj >= 0;
            inb[u] = st[u] && c_ok && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
            lofs[u] = (dw_slot(gy) * a.iw + ixj) * a.ps + p.cgl * 4;
            v0[u] = make_uint4(0, 0, 0, 0);
            if (MODE == 1) v1[u] = make_uint4(0, 0, 0, 0);
            if (inb[u]) {
                const size_t off = ((size_t)n * a.H + gy) * rowstride + xbase + ixj * C8;
                v0[u] = src0[off];
                if (MODE == 1) v1[u] = src1[off];
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            if (!st[u]) continue;
            uint4 v = v0[u];
            if (MODE == 0) {
                if (has_coef && inb[u]) {
                    float cs[8], ct[8];
                    *(float4*)&cs[0] = *(const float4*)(lds_c + p.cgl * 8);
                    *(float4*)&cs[4] = *(const float4*)(lds_c + p.cgl * 8 + 4);
                    *(float4*)&ct[0] = *(const float4*)(lds_c + cblk + p.cgl * 8);
                    *(float4*)&ct[4] = *(const float4*)(lds_c + cblk + p.cgl * 8 + 4);
                    v = act8(v, cs, ct);
                }
            } else if (inb[u]) {
                float cf[5][8];
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    *(float4*)&cf[r][0] = *(const float4*)(lds_c + r * cblk + p.cgl * 8);
                    *(float4*)&cf[r][4] = *(const float4*)(lds_c + r * cblk + p.cgl * 8 + 4);
                }
                float o[8];
                dy8(v, v1[u], cf[0], cf[1], cf[2], cf[3], cf[4], o);
                v = pack8(o);
            }
            *(uint4*)(ring + lofs[u]) = v;
        }
    }
}

__device__ __forceinline__ void dw_item(const DwArgs& a, int item, int& n, int& x0, int& c0) {
    const int cb = item % a.cblocks;
    const int r = item / a.cblocks;
    const int sxi = r % a.strips_x;
    n = r / a.strips_x;
    x0 = sxi * a.sx * DW_BW;
    c0 = cb * 2 * a.cpw;
}

__device__ __forceinline__ void dw_load_coefs(float* lds_c, const float* r0, const float* r1, const float* rows5, int nrows,
                                              int C, int c0, int cblk) {
    for (int i = threadIdx.x; i < nrows * cblk; i += blockDim.x) {
        const int r = i / cblk, c = c0 + i % cblk;
        float v = 0.f;
        if (c < C) v = rows5 ? rows5[(size_t)r * C + c] : (r == 0 ? (r0 ? r0[c] : 1.f) : (r1 ? r1[c] : 0.f));
        lds_c[i] = v;
    }
}

// ---- forward (MODE 0) and input gradient (MODE 1: dy-on-load staging, flipped filter, no bias/stats) --------
template <int KS, int MODE>
__global__ __launch_bounds__(256, ((KS == 3 && MODE == 0) ? 3 : 2)) void k_dw_conv(DwArgs a, MnasActIn in, MnasGradIn d,
                                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                                    uint32_t* __restrict__ out, float* __restrict__ stats,
                                                                    const uint32_t* __restrict__ red_y, const float* __restrict__ red_bn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PAD = KS / 2, WIN_W = DW_BW + KS - 1, CROWS = (MODE == 0) ? 2 : 5;
    const int cblk = 2 * a.cpw;
    float* lds_c = (float*)smem;                         // [CROWS][cblk] staging coefficients
    float* lds_red = lds_c + CROWS * cblk;               // [2][cblk]
    uint32_t* ring = (uint32_t*)(lds_red + 2 * cblk);    // [RR][iw][ps]
    const int tid = threadIdx.x;
    const int cp = tid % a.cpw, sxi = tid / a.cpw;
    const bool active = sxi < a.sx;
    DwStagePlan plan;
    dw_make_plan(a, plan);
    const bool has_coef = (MODE == 1) || in.scale != nullptr;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    int cur_c0 = -1;
    float wt[KS * KS][2], b0 = 0.f, b1 = 0.f;
    float rs[2] = {0.f, 0.f}, rt[2] = {0.f, 0.f}, ris[2] = {0.f, 0.f}, rmu[2] = {0.f, 0.f};   // fused BN-bwd reduce coefs
    const int nsteps = (a.H + 2 * PAD + DW_G - 1) / DW_G;
    const uint4* src0 = (const uint4*)(MODE == 0 ? in.data : d.g);
    const uint4* src1 = (const uint4*)(MODE == 0 ? nullptr : d.y);

    for (int item = blockIdx.x; item < a.items && (int)blockIdx.x < a.geff; item += a.geff) {
        int n, x0, c0;
        dw_item(a, item, n, x0, c0);
        const int ch = c0 + 2 * cp;
        const bool ch_ok = ch < a.C;
        if (c0 != cur_c0) {       // first item: a workgroup only ever sees ONE channel block (see dw_setup)
            cur_c0 = c0;
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) {
                // MODE 1: gin[p] = sum_k dy[p + PAD - k] * w[k]  ==  a forward conv with the filter flipped in y and x
                const int ts = (MODE == 0) ? t : (KS * KS - 1 - t);
                wt[t][0] = ch_ok ? w[(size_t)ts * a.C + ch] : 0.f;
                wt[t][1] = ch_ok ? w[(size_t)ts * a.C + ch + 1] : 0.f;
            }
            b0 = (MODE == 0 && bias && ch_ok) ? bias[ch] : 0.f;
            b1 = (MODE == 0 && bias && ch_ok) ? bias[ch + 1] : 0.f;
            if (MODE == 1 && red_y && ch_ok) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    rs[e] = red_bn[0 * a.C + ch + e];
                    rt[e] = red_bn[1 * a.C + ch + e];
                    ris[e] = red_bn[6 * a.C + ch + e];
                    rmu[e] = -red_bn[5 * a.C + ch + e] * ris[e];
                }
            }
            __syncthreads();
            if (MODE == 0) dw_load_coefs(lds_c, in.scale, in.shift, nullptr, 2, a.C, c0, cblk);
            else dw_load_coefs(lds_c, nullptr, nullptr, d.coef, 5, a.C, c0, cblk);
        }
        // register ring of KS partial output rows: A[i] = output row (iy - PAD + i) while input row iy is processed
        float A[KS][DW_BW][2];
#pragma unroll
        for (int i = 0; i < KS; ++i)
#pragma unroll
            for (int j = 0; j < DW_BW; ++j) { A[i][j][0] = b0; A[i][j][1] = b1; }
        const int gx0 = x0 + sxi * DW_BW;
        uint32_t* outp = out + (((size_t)n * a.H * a.W + gx0) * a.C + ch) / 2;
        const uint32_t* colp = ring + (size_t)sxi * DW_BW * a.ps + cp;

        for (int s = 0; s < nsteps; ++s) {
            const int r0 = -PAD + s * DW_G;
            __syncthreads();                         // previous group fully consumed (and coefficients visible)
            dw_stage<KS, MODE>(a, plan, ring, src0, src1, lds_c, has_coef, n, r0, DW_G, x0, c0);
            __syncthreads();
            if (!active) continue;
#pragma unroll 1
            for (int j = 0; j < DW_G; ++j) {
                const int iy = r0 + j;
                if (iy >= a.H + PAD) break;
                const uint32_t* rowp = colp + (size_t)dw_slot(iy) * a.iw * a.ps;
                const int oy = iy - PAD;             // A[0] is complete after this row
                // fused reduce: fetch the target's raw outputs for the row we are about to emit; lands under the FMAs
                uint32_t ypre[DW_BW];
                if (MODE == 1 && red_y && oy >= 0 && ch_ok) {
                    const uint32_t* yp = red_y + (((size_t)n * a.H * a.W + gx0) * a.C + ch) / 2 + (size_t)oy * a.W * a.C / 2;
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) ypre[ox] = (gx0 + ox < a.W) ? yp[(size_t)ox * a.C / 2] : 0u;
                }
                float xr[WIN_W][2];
#pragma unroll
                for (int x = 0; x < WIN_W; ++x) {
                    const uint32_t u = rowp[x * a.ps];
                    xr[x][0] = bf_lo(u);
                    xr[x][1] = bf_hi(u);
                }
#pragma unroll
                for (int i = 0; i < KS; ++i)
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            A[i][ox][0] = fmaf(wt[(KS - 1 - i) * KS + kx][0], xr[ox + kx][0], A[i][ox][0]);
                            A[i][ox][1] = fmaf(wt[(KS - 1 - i) * KS + kx][1], xr[ox + kx][1], A[i][ox][1]);
                        }
                if (oy >= 0 && ch_ok) {
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) {
                        if (gx0 + ox < a.W) {
                            const float v0 = A[0][ox][0], v1 = A[0][ox][1];
                            if (MODE == 0) {
                                s1[0] += v0; s2[0] = fmaf(v0, v0, s2[0]);
                                s1[1] += v1; s2[1] = fmaf(v1, v1, s2[1]);
                            }
                            const uint32_t pk = pack_bf16(v0, v1);
                            outp[((size_t)oy * a.W + ox) * a.C / 2] = pk;
                            if (MODE == 1 && red_y) {      // fused BN-backward reduce for the producer of x
                                const uint32_t yv = ypre[ox];
                                const float g0 = bf_lo(pk), g1 = bf_hi(pk), y0 = bf_lo(yv), y1 = bf_hi(yv);
                                const float dz0 = (fmaf(y0, rs[0], rt[0]) > 0.f) ? g0 : 0.f;
                                const float dz1 = (fmaf(y1, rs[1], rt[1]) > 0.f) ? g1 : 0.f;
                                s1[0] += dz0; s2[0] = fmaf(dz0, fmaf(y0, ris[0], rmu[0]), s2[0]);
                                s1[1] += dz1; s2[1] = fmaf(dz1, fmaf(y1, ris[1], rmu[1]), s2[1]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i + 1 < KS; ++i)
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) { A[i][ox][0] = A[i + 1][ox][0]; A[i][ox][1] = A[i + 1][ox][1]; }
#pragma unroll
                for (int ox = 0; ox < DW_BW; ++ox) { A[KS - 1][ox][0] = b0; A[KS - 1][ox][1] = b1; }
            }
        }
    }
    if ((MODE == 0 || red_y) && stats) {
        // stats table is float[2][C][rows], rows = geff / cblocks: workgroup b owns column b / cblocks of its own
        // channel block (b % cblocks), so every (channel, column) is written exactly once -- no zero fill, no atomics
        __syncthreads();
        for (int i = tid; i < 2 * cblk; i += blockDim.x) lds_red[i] = 0.f;
        __syncthreads();
        if (active && cur_c0 >= 0 && cur_c0 + 2 * cp < a.C) {
            atomicAdd(&lds_red[2 * cp], s1[0]); atomicAdd(&lds_red[2 * cp + 1], s1[1]);
            atomicAdd(&lds_red[cblk + 2 * cp], s2[0]); atomicAdd(&lds_red[cblk + 2 * cp + 1], s2[1]);
        }
        __syncthreads();
        if ((int)blockIdx.x < a.geff) {
            const int rows = a.geff / a.cblocks, col = blockIdx.x / a.cblocks;
            const int cb0 = (blockIdx.x % a.cblocks) * cblk;
            for (int i = tid; i < 2 * cblk; i += blockDim.x) {
                const int r = i / cblk, c = cb0 + i % cblk;
                if (c < a.C) stats[((size_t)r * a.C + c) * rows + col] = (cur_c0 >= 0) ? lds_red[i] : 0.f;
            }
        }
    }
}

// ---- weight gradient: dW[ky][kx][c] = sum_p dy[p][c] * act(x)[p + (ky-PAD, kx-PAD)][c] --------------------------
// Streamed: at image row iy the dy row iy (centre columns) enters a register ring D[0..KS-1] (D[q] = dy row iy-q) and
// the activation row r = iy-PAD is read once; it pairs with dy rows r-ky+PAD = iy-ky = D[ky].  12 LDS dwords per row.
template <int KS>
__global__ __launch_bounds__(256, 2) void k_dw_wgrad(DwArgs a, MnasActIn x, MnasGradIn d,
                                                                     float* __restrict__ wpartial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PAD = KS / 2, WIN_W = DW_BW + KS - 1;
    const int cblk = 2 * a.cpw;
    float* lds_cx = (float*)smem;                            // [2][cblk]
    float* lds_cd = lds_cx + 2 * cblk;                       // [5][cblk]
    float* lds_red = lds_cd + 5 * cblk;                      // [KS*KS][cblk]
    uint32_t* ring_d = (uint32_t*)(lds_red + KS * KS * cblk);
    uint32_t* ring_x = ring_d + (size_t)DW_RR * a.iw * a.ps;
    const int tid = threadIdx.x;
    const int cp = tid % a.cpw, sxi = tid / a.cpw;
    const bool active = sxi < a.sx;
    DwStagePlan plan;
    dw_make_plan(a, plan);
    const bool has_coef = x.scale != nullptr;
    int cur_c0 = -1;
    float wacc[KS * KS][2];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) { wacc[t][0] = 0.f; wacc[t][1] = 0.f; }
    const int nsteps = (a.H + 2 * PAD + DW_G - 1) / DW_G;

    for (int item = blockIdx.x; item < a.items && (int)blockIdx.x < a.geff; item += a.geff) {
        int n, x0, c0;
        dw_item(a, item, n, x0, c0);
        if (c0 != cur_c0) {
            cur_c0 = c0;
            __syncthreads();
            dw_load_coefs(lds_cx, x.scale, x.shift, nullptr, 2, a.C, c0, cblk);
            dw_load_coefs(lds_cd, nullptr, nullptr, d.coef, 5, a.C, c0, cblk);
        }
        const size_t coloff = (size_t)sxi * DW_BW * a.ps + cp;
        float D[KS][DW_BW][2];          // register ring of the last KS dy rows (centre columns)
#pragma unroll
        for (int q = 0; q < KS; ++q)
#pragma unroll
            for (int ox = 0; ox < DW_BW; ++ox) { D[q][ox][0] = 0.f; D[q][ox][1] = 0.f; }
        for (int s = 0; s < nsteps; ++s) {
            const int r0 = -PAD + s * DW_G;
            __syncthreads();
            dw_stage<KS, 1>(a, plan, ring_d, (const uint4*)d.g, (const uint4*)d.y, lds_cd, true, n, r0, DW_G, x0, c0);
            dw_stage<KS, 0>(a, plan, ring_x, (const uint4*)x.data, nullptr, lds_cx, has_coef, n, r0, DW_G, x0, c0);
            __syncthreads();
            if (!active) continue;
#pragma unroll 1
            for (int j = 0; j < DW_G; ++j) {
                const int iy = r0 + j;
                if (iy >= a.H + PAD) break;
                // D[q] = centre columns of dy row (iy - q); rotate the ring and read the new row iy
#pragma unroll
                for (int q = KS - 1; q > 0; --q)
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) { D[q][ox][0] = D[q - 1][ox][0]; D[q][ox][1] = D[q - 1][ox][1]; }
                {
                    const uint32_t* rowp = ring_d + (size_t)dw_slot(iy) * a.iw * a.ps + coloff;
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) {
                        const uint32_t u = rowp[(ox + PAD) * a.ps];    // rows/columns outside the image were staged as zeros
                        D[0][ox][0] = bf_lo(u);
                        D[0][ox][1] = bf_hi(u);
                    }
                }
                // activation row r = iy - PAD pairs with dy rows o = r - ky + PAD = iy - ky  (ky = 0..KS-1) -> D[ky]
                const int r = iy - PAD;
                if (r < -PAD) continue;
                const uint32_t* rowp = ring_x + (size_t)dw_slot(r) * a.iw * a.ps + coloff;
                float xr[WIN_W][2];
#pragma unroll
                for (int xx = 0; xx < WIN_W; ++xx) {
                    const uint32_t u = rowp[xx * a.ps];
                    xr[xx][0] = bf_lo(u);
                    xr[xx][1] = bf_hi(u);
                }
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            wacc[ky * KS + kx][0] = fmaf(D[ky][ox][0], xr[ox + kx][0], wacc[ky * KS + kx][0]);
                            wacc[ky * KS + kx][1] = fmaf(D[ky][ox][1], xr[ox + kx][1], wacc[ky * KS + kx][1]);
                        }
            }
        }
    }
    // wpartial is float[rows][k*k][C], rows = geff / cblocks: workgroup b writes row b / cblocks, its own channel block only
    __syncthreads();
    for (int i = tid; i < KS * KS * cblk; i += blockDim.x) lds_red[i] = 0.f;
    __syncthreads();
    if (active && cur_c0 >= 0 && cur_c0 + 2 * cp < a.C) {
#pragma unroll
        for (int k = 0; k < KS * KS; ++k) {
            atomicAdd(&lds_red[k * cblk + 2 * cp], wacc[k][0]);
            atomicAdd(&lds_red[k * cblk + 2 * cp + 1], wacc[k][1]);
        }
    }
    __syncthreads();
    if ((int)blockIdx.x < a.geff) {
        const int row = blockIdx.x / a.cblocks;
        const int cb0 = (blockIdx.x % a.cblocks) * cblk;
        for (int i = tid; i < KS * KS * cblk; i += blockDim.x) {
            const int k = i / cblk, c = cb0 + i % cblk;
            if (c < a.C) wpartial[((size_t)row * KS * KS + k) * a.C + c] = (cur_c0 >= 0) ? lds_red[i] : 0.f;
        }
    }
}

// ---- fused backward: input gradient + weight gradient (+ BN-backward reduce of the producer of x) in ONE sweep ----
// Per image row iy (dy row iy and activation row iy-PAD are in the rings):
//   xr  = dy row iy, k+3 columns        -> scattered into the register ring A of KS partial gin rows (flipped filter);
//                                          its centre 4 columns become D[0] of the dy ring (D[q] = dy row iy-q)
//   xa  = activation row r = iy-PAD     -> wacc[ky][kx] += D[ky][ox] * xa[ox+kx]   (dy rows r-ky+PAD = iy-ky)
// 16 LDS dwords and 2*k*k*4*2 FMAs per row; g and y are read from HBM once, dy-on-load is computed once.
template <int KS, bool RED>
__global__ __launch_bounds__(256, 2) void k_dw_bwd_fused(DwArgs a, MnasActIn x, MnasGradIn d, const float* __restrict__ w,
                                                         uint32_t* __restrict__ gin, float* __restrict__ wpartial,
                                                         float* __restrict__ red_partial, const float* __restrict__ red_bn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PAD = KS / 2, WIN_W = DW_BW + KS - 1;
    const int cblk = 2 * a.cpw;
    float* lds_cx = (float*)smem;                            // [2][cblk]
    float* lds_cd = lds_cx + 2 * cblk;                       // [5][cblk]
    float* lds_red = lds_cd + 5 * cblk;                      // [KS*KS][cblk]  (also reused for the [2][cblk] reduce)
    uint32_t* ring_d = (uint32_t*)(lds_red + KS * KS * cblk);
    uint32_t* ring_x = ring_d + (size_t)DW_RR * a.iw * a.ps;
    const int tid = threadIdx.x;
    const int cp = tid % a.cpw, sxi = tid / a.cpw;
    const bool active = sxi < a.sx;
    DwStagePlan plan;
    dw_make_plan(a, plan);
    const bool has_coef = x.scale != nullptr;
    constexpr bool do_red = RED;
    const uint32_t* red_y = (const uint32_t*)x.data;
    // rings start zeroed: rows that are never staged (activation rows < -PAD of the first step) must read as finite
    for (int i = tid; i < 2 * DW_RR * a.iw * a.ps; i += blockDim.x) ring_d[i] = 0u;
    int cur_c0 = -1;
    float wt[KS * KS][2], wacc[KS * KS][2];
    float rs[2] = {0.f, 0.f}, rt[2] = {0.f, 0.f}, ris[2] = {0.f, 0.f}, rmu[2] = {0.f, 0.f};
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) { wacc[t][0] = 0.f; wacc[t][1] = 0.f; }
    const int nsteps = (a.H + 2 * PAD + DW_G - 1) / DW_G;

    for (int item = blockIdx.x; item < a.items && (int)blockIdx.x < a.geff; item += a.geff) {
        int n, x0, c0;
        dw_item(a, item, n, x0, c0);
        const int ch = c0 + 2 * cp;
        const bool ch_ok = ch < a.C;
        if (c0 != cur_c0) {
            cur_c0 = c0;
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) {      // flipped filter for the input gradient
                wt[t][0] = ch_ok ? w[(size_t)(KS * KS - 1 - t) * a.C + ch] : 0.f;
                wt[t][1] = ch_ok ? w[(size_t)(KS * KS - 1 - t) * a.C + ch + 1] : 0.f;
            }
            if (do_red && ch_ok) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    rs[e] = red_bn[0 * a.C + ch + e];
                    rt[e] = red_bn[1 * a.C + ch + e];
                    ris[e] = red_bn[6 * a.C + ch + e];
                    rmu[e] = -red_bn[5 * a.C + ch + e] * ris[e];
                }
            }
            __syncthreads();
            dw_load_coefs(lds_cx, x.scale, x.shift, nullptr, 2, a.C, c0, cblk);
            dw_load_coefs(lds_cd, nullptr, nullptr, d.coef, 5, a.C, c0, cblk);
        }
        float A[KS][DW_BW][2], D[KS][DW_BW][2];
#pragma unroll
        for (int i = 0; i < KS; ++i)
#pragma unroll
            for (int j = 0; j < DW_BW; ++j) { A[i][j][0] = 0.f; A[i][j][1] = 0.f; D[i][j][0] = 0.f; D[i][j][1] = 0.f; }
        const int gx0 = x0 + sxi * DW_BW;
        const size_t obase = (((size_t)n * a.H * a.W + gx0) * a.C + ch) / 2;
        const size_t coloff = (size_t)sxi * DW_BW * a.ps + cp;

        for (int s = 0; s < nsteps; ++s) {
            const int r0 = -PAD + s * DW_G;
            __syncthreads();
            dw_stage<KS, 1, (KS == 3 ? 4 : 2)>(a, plan, ring_d, (const uint4*)d.g, (const uint4*)d.y, lds_cd, true, n, r0, DW_G, x0, c0);
            dw_stage<KS, 0, (KS == 3 ? 4 : 2)>(a, plan, ring_x, (const uint4*)x.data, nullptr, lds_cx, has_coef, n, r0, DW_G, x0, c0);
            __syncthreads();
            if (!active) continue;
#pragma unroll 1
            for (int j = 0; j < DW_G; ++j) {
                const int iy = r0 + j;
                const int oy = iy - PAD;
                const bool emit = oy >= 0 && oy < a.H && ch_ok;
                uint32_t ypre[DW_BW];
                if (do_red && emit) {
                    const uint32_t* yp = red_y + obase + (size_t)oy * a.W * a.C / 2;
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) ypre[ox] = (gx0 + ox < a.W) ? yp[(size_t)ox * a.C / 2] : 0u;
                }
                // ---- dy row iy: input-gradient scatter + dy ring
                {
                    const uint32_t* rowp = ring_d + (size_t)dw_slot(iy) * a.iw * a.ps + coloff;
                    float xr[WIN_W][2];
#pragma unroll
                    for (int xx = 0; xx < WIN_W; ++xx) {
                        const uint32_t u = rowp[xx * a.ps];
                        xr[xx][0] = bf_lo(u);
                        xr[xx][1] = bf_hi(u);
                    }
#pragma unroll
                    for (int i = 0; i < KS; ++i)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                            for (int kx = 0; kx < KS; ++kx) {
                                A[i][ox][0] = fmaf(wt[(KS - 1 - i) * KS + kx][0], xr[ox + kx][0], A[i][ox][0]);
                                A[i][ox][1] = fmaf(wt[(KS - 1 - i) * KS + kx][1], xr[ox + kx][1], A[i][ox][1]);
                            }
#pragma unroll
                    for (int q = KS - 1; q > 0; --q)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox) { D[q][ox][0] = D[q - 1][ox][0]; D[q][ox][1] = D[q - 1][ox][1]; }
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) { D[0][ox][0] = xr[ox + PAD][0]; D[0][ox][1] = xr[ox + PAD][1]; }
                }
                // ---- emit gin row oy (+ fused BN-backward reduce of the producer of x)
                if (emit) {
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) {
                        if (gx0 + ox < a.W) {
                            const uint32_t pk = pack_bf16(A[0][ox][0], A[0][ox][1]);
                            gin[obase + ((size_t)oy * a.W + ox) * a.C / 2] = pk;
                            if (do_red) {
                                const uint32_t yv = ypre[ox];
                                const float g0 = bf_lo(pk), g1 = bf_hi(pk), y0 = bf_lo(yv), y1 = bf_hi(yv);
                                const float dz0 = (fmaf(y0, rs[0], rt[0]) > 0.f) ? g0 : 0.f;
                                const float dz1 = (fmaf(y1, rs[1], rt[1]) > 0.f) ? g1 : 0.f;
                                s1[0] += dz0; s2[0] = fmaf(dz0, fmaf(y0, ris[0], rmu[0]), s2[0]);
                                s1[1] += dz1; s2[1] = fmaf(dz1, fmaf(y1, ris[1], rmu[1]), s2[1]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i + 1 < KS; ++i)
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) { A[i][ox][0] = A[i + 1][ox][0]; A[i][ox][1] = A[i + 1][ox][1]; }
#pragma unroll
                for (int ox = 0; ox < DW_BW; ++ox) { A[KS - 1][ox][0] = 0.f; A[KS - 1][ox][1] = 0.f; }
                // ---- activation row r = iy - PAD: weight gradient against the dy ring (rows < -PAD: zero-initialised ring)
                {
                    const uint32_t* rowp = ring_x + (size_t)dw_slot(oy) * a.iw * a.ps + coloff;
                    float xa[WIN_W][2];
#pragma unroll
                    for (int xx = 0; xx < WIN_W; ++xx) {
                        const uint32_t u = rowp[xx * a.ps];
                        xa[xx][0] = bf_lo(u);
                        xa[xx][1] = bf_hi(u);
                    }
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                            for (int kx = 0; kx < KS; ++kx) {
                                wacc[ky * KS + kx][0] = fmaf(D[ky][ox][0], xa[ox + kx][0], wacc[ky * KS + kx][0]);
                                wacc[ky * KS + kx][1] = fmaf(D[ky][ox][1], xa[ox + kx][1], wacc[ky * KS + kx][1]);
                            }
                }
            }
        }
    }
    const int row = blockIdx.x / a.cblocks, rows = a.geff / a.cblocks;
    const int cb0 = (blockIdx.x % a.cblocks) * cblk;
    // ---- fused-reduce table float[2][C][rows]
    if (do_red) {
        __syncthreads();
        for (int i = tid; i < 2 * cblk; i += blockDim.x) lds_red[i] = 0.f;
        __syncthreads();
        if (active && cur_c0 >= 0 && cur_c0 + 2 * cp < a.C) {
            atomicAdd(&lds_red[2 * cp], s1[0]); atomicAdd(&lds_red[2 * cp + 1], s1[1]);
            atomicAdd(&lds_red[cblk + 2 * cp], s2[0]); atomicAdd(&lds_red[cblk + 2 * cp + 1], s2[1]);
        }
        __syncthreads();
        if ((int)blockIdx.x < a.geff)
            for (int i = tid; i < 2 * cblk; i += blockDim.x) {
                const int r = i / cblk, c = cb0 + i % cblk;
                if (c < a.C) red_partial[((size_t)r * a.C + c) * rows + row] = (cur_c0 >= 0) ? lds_red[i] : 0.f;
            }
    }
    // ---- wpartial float[rows][k*k][C]
    __syncthreads();
    for (int i = tid; i < KS * KS * cblk; i += blockDim.x) lds_red[i] = 0.f;
    __syncthreads();
    if (active && cur_c0 >= 0 && cur_c0 + 2 * cp < a.C) {
#pragma unroll
        for (int k = 0; k < KS * KS; ++k) {
            atomicAdd(&lds_red[k * cblk + 2 * cp], wacc[k][0]);
            atomicAdd(&lds_red[k * cblk + 2 * cp + 1], wacc[k][1]);
        }
    }
    __syncthreads();
    if ((int)blockIdx.x < a.geff)
        for (int i = tid; i < KS * KS * cblk; i += blockDim.x) {
            const int k = i / cblk, c = cb0 + i % cblk;
            if (c < a.C) wpartial[((size_t)row * KS * KS + k) * a.C + c] = (cur_c0 >= 0) ? lds_red[i] : 0.f;
        }
}

static bool dw_setup(DwArgs* a, int N, int H, int W, int C, int k, int nrings, int nparts) {
    if (!dw_pick(N, H, W, C, k, nrings, a)) return false;
    if (nparts < a->cblocks) return false;
    int g = nparts < a->items ? nparts : a->items;
    a->geff = g / a->cblocks * a->cblocks;          // multiple of cblocks: item % cblocks is constant per workgroup
    return a->geff >= a->cblocks;
}

// rows of the partial tables a launch with `nparts` writes.  which = 0: forward statistics / the input-gradient
// launch's fused-reduce table, float[2][C][rows]; which = 1: the weight-gradient launch, float[rows][k*k][C]
// (it stages two rings and may pick a narrower strip, hence its own count).
extern "C" int mnas_dw_rows(int N, int H, int W, int C, int k, int nparts, int which) {
    DwArgs a;
    if (!dw_setup(&a, N, H, W, C, k, which ? 2 : 1, nparts)) return -1;
    return a.geff / a.cblocks;
}

extern "C" int mnas_dw_fwd(const MnasDwFwd* c, void* stream) {
    if (!c || (c->k != 3 && c->k != 5) || (c->C & 7) || c->nparts < 1) return MNAS_EINVAL;
    DwArgs a;
    if (!dw_setup(&a, c->N, c->H, c->W, c->C, c->k, 1, c->nparts)) return MNAS_EINVAL;
    const size_t lds = (size_t)4 * 2 * a.cpw * sizeof(float) + (size_t)DW_RR * a.iw * a.ps * 4;
    hipStream_t s = (hipStream_t)stream;
    MnasGradIn nod = {nullptr, nullptr, nullptr};
    if (c->k == 3)
        hipLaunchKernelGGL((k_dw_conv<3, 0>), dim3(a.geff), dim3(a.nthreads), lds, s, a, c->in, nod, c->w, c->bias, (uint32_t*)c->out, c->stats, nullptr, nullptr);
    else
        hipLaunchKernelGGL((k_dw_conv<5, 0>), dim3(a.geff), dim3(a.nthreads), lds, s, a, c->in, nod, c->w, c->bias, (uint32_t*)c->out, c->stats, nullptr, nullptr);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_dw_bwd(const MnasDwBwd* c, void* stream) {
    if (!c || (c->k != 3 && c->k != 5) || (c->C & 7) || c->nparts < 1) return MNAS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    MnasActIn noa = {nullptr, nullptr, nullptr};
    const bool red = c->red_bn != nullptr && c->red_partial != nullptr;
    if (c->phase == 0) {   // fused: one sweep, both partial tables have mnas_dw_rows(..., 1) rows
        DwArgs a;
        if (!dw_setup(&a, c->N, c->H, c->W, c->C, c->k, 2, c->nparts)) return MNAS_EINVAL;
        const size_t lds = (size_t)(7 + c->k * c->k) * 2 * a.cpw * sizeof(float) + (size_t)2 * DW_RR * a.iw * a.ps * 4;
#define MNAS_DWF(K_, R_) hipLaunchKernelGGL((k_dw_bwd_fused<K_, R_>), dim3(a.geff), dim3(a.nthreads), lds, s, a, c->x, c->dy, c->w, \
                                            (uint32_t*)c->gin, c->wpartial, c->red_partial, c->red_bn)
        if (c->k == 3) { if (red) MNAS_DWF(3, true); else MNAS_DWF(3, false); }
        else { if (red) MNAS_DWF(5, true); else MNAS_DWF(5, false); }
#undef MNAS_DWF
        MNAS_CHECK_LAUNCH();
        return MNAS_OK;
    }
    if (c->phase != 2) {   // input gradient
        DwArgs a;
        if (!dw_setup(&a, c->N, c->H, c->W, c->C, c->k, 1, c->nparts)) return MNAS_EINVAL;
        const size_t lds = (size_t)7 * 2 * a.cpw * sizeof(float) + (size_t)DW_RR * a.iw * a.ps * 4;
        if (c->k == 3)
            hipLaunchKernelGGL((k_dw_conv<3, 1>), dim3(a.geff), dim3(a.nthreads), lds, s, a, noa, c->dy, c->w, nullptr, (uint32_t*)c->gin, red ? c->red_partial : nullptr, red ? (const uint32_t*)c->x.data : nullptr, c->red_bn);
        else
            hipLaunchKernelGGL((k_dw_conv<5, 1>), dim3(a.geff), dim3(a.nthreads), lds, s, a, noa, c->dy, c->w, nullptr, (uint32_t*)c->gin, red ? c->red_partial : nullptr, red ? (const uint32_t*)c->x.data : nullptr, c->red_bn);
        MNAS_CHECK_LAUNCH();
    }
    if (c->phase != 1) {   // weight gradient
        DwArgs a;
        if (!dw_setup(&a, c->N, c->H, c->W, c->C, c->k, 2, c->nparts)) return MNAS_EINVAL;
        const size_t lds = (size_t)(7 + c->k * c->k) * 2 * a.cpw * sizeof(float) + (size_t)2 * DW_RR * a.iw * a.ps * 4;
        if (c->k == 3)
            hipLaunchKernelGGL(k_dw_wgrad<3>, dim3(a.geff), dim3(a.nthreads), lds, s, a, c->x, c->dy, c->wpartial);
        else
            hipLaunchKernelGGL(k_dw_wgrad<5>, dim3(a.geff), dim3(a.nthreads), lds, s, a, c->x, c->dy, c->wpartial);
        MNAS_CHECK_LAUNCH();
    }
    return MNAS_OK;
}
