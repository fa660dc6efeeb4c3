// Depthwise kxk convolution (k in {3,5}, stride 1, pad k/2) forward and backward, NHWC bf16, fp32 math.
// Replaces ATen's grouped conv2d fwd/bwd for ConvBlock(groups=C) (mnasnet.py:76-81,122-125).
//
// Structure ("vertical sweep over LDS row rings filled by DMA"):
//   * a workgroup owns ONE image, a column strip of TW = 4*sx output columns and a block of 2*cpw channels
//     (all channels of the pixel when C <= 144, otherwise >= 64 channels = 128-byte runs), and sweeps the strip
//     top to bottom G = 4 rows at a time;
//   * the RAW tensors (forward: the producer's raw output y; backward: g, y and the forward input) are copied
//     HBM -> LDS with `global_load_lds` (16 B per lane, no VGPR round trip, no staging ALU work), into 8-row rings
//     laid out exactly like the HBM row segment ([x][c], 16-byte chunks in (x, channel-group) order); every input
//     element is fetched from HBM once per strip (halo only horizontally); the ring is two buffers of G rows, so the
//     DMA of the next row group overlaps the arithmetic of the current one;
//   * the producer's BatchNorm+ReLU ("act-on-load") and the BatchNorm/ReLU backward ("dy-on-load") are applied
//     when a thread READS its window from LDS: a thread always works on the same channel pair, so the
//     coefficients are 4 (resp. 10) registers; padding is handled by a per-thread column mask and a uniform row test;
//   * compute: thread = (channel pair, 4-column strip), consecutive lanes = consecutive channel pairs
//     (conflict-free dword LDS reads, contiguous global stores); the k*k*2 taps of its channel pair are in VGPRs;
//     rows are STREAMED: each input row is read from LDS once (k+3 dwords) and scattered into a register ring of
//     k partial output rows; the oldest ring entry is complete after every input row and is emitted;
//   * BatchNorm partial statistics (forward), the fused BatchNorm-backward reduction of the producer of x and the
//     k*k weight-gradient sums (backward) are accumulated in registers over the whole sweep and written once
//     per workgroup (no atomics in HBM, deterministic);
//   * backward is ONE launch (phase 0: input gradient + weight gradient + reduce from the same three rings) for both
//     kernel sizes; the 5x5 form holds 2*k*k*2 persistent accumulators on top of two register rings (256 VGPRs, 12
//     spilled) and still beats the two-launch form (phase 1 + phase 2, kept for callers that want the weight gradient
//     on another stream) once 2-row DMA groups give it full-width strips: 155 vs 118 + 92 us at 56x56x72, bs 256.
// Roofline: HBM (AI 4.5 flop/B for 3x3, 12.5 for 5x5 forward; backward moves 4 tensors for 2x the FMAs).
#include "mnas_common.h"
#ifndef MNAS_DW_XFILL
#define MNAS_DW_XFILL 1      // backward: out-of-image ring columns of x pre-filled, window read without per-column selects (as the forward)
#endif

#define DW_G 4          // rows per sweep step (default; the 5x5 weight-gradient sweep uses 2, see mnas_dw_bwd)
#define DW_BW 4         // output columns per thread
#define DW_RR 8         // ring rows = 2 * G (two buffers of G rows)

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

struct DwArgs {
    int N, H, W, C;
    int cpw, sx, nthreads, cgn, iw;
    int strips_x, cblocks, items, geff;     // geff: workgroups that take items (multiple of cblocks)
    int rc, nb;                             // 16-byte chunks per ring row; DMA blocks (64 chunks) per row
    float score;                            // geometry score of dw_pick (host only)
    int nt;                                 // nontemporal output stores
};

// exp_kpad > 0: geometry for the fused expand+depthwise forms (the workgroup also holds the expand conv's weight block
// [2*cpw rounded to 16][exp_kpad+8] bf16, bias and input coefficients in LDS; channel blocks of at most 128 channels = 8 MFMA
// tiles; at most DW_EXP_MAXPG 16-pixel groups of the G x iw ring patch per wave)
#define DW_EXP_MAXPG 4
#ifndef DW_SRC_LDS_CAP
#define DW_SRC_LDS_CAP (52 * 1024)      // SRC backward: three workgroups per CU
#endif
static size_t dw_exp_lds(int cpw, int exp_kpad) {
    const int rows = (2 * cpw + 15) / 16 * 16;
    return (size_t)rows * (exp_kpad + 8) * 2 + (size_t)rows * 4 + (size_t)2 * exp_kpad * 4;
}
// src_cin > 0: geometry for the SRC backward (k_dw_bwd<.., SRC>): two weight blocks (expand conv, project conv^T) + two staging
// rings of the narrow tensors (block input x, project-conv dy: src_cin channels) next to the three row rings
static size_t dw_src_lds(int cpw, int kpad, int rr, int iw, int src_cin) {
    return 2 * dw_exp_lds(cpw, kpad) + (size_t)2 * rr * iw * src_cin * 2;
}
static bool dw_pick(int N, int H, int W, int C, int k, int nrings, int rr, DwArgs* a, int exp_kpad = 0, int src_cin = 0) {
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
        for (int sx = 1; sx <= maxsx && sx * cpw <= 256; ++sx) {
            const int tw = sx * DW_BW, iw = tw + k - 1;
            size_t lds = (size_t)nrings * rr * iw * cpw * 4;
            if (exp_kpad > 0) {
                if (cpw > 64) continue;
                lds += src_cin > 0 ? dw_src_lds(cpw, exp_kpad, rr, iw, src_cin) : dw_exp_lds(cpw, exp_kpad);
                const int nth_ = ((sx * cpw + 63) / 64) * 64;
                const int npg = ((rr / 2) * iw + 15) / 16;
                if (src_cin == 0 && (npg + nth_ / 64 - 1) / (nth_ / 64) > DW_EXP_MAXPG) continue;
                if (src_cin > 0 && nth_ > 192) continue;                       // + one producer wave = 256 threads
            }
            // two workgroups per CU either way (160 KB LDS): wide strips (78 KB) measured 8-10 % faster than 60 KB for
            // every launch form except the 5x5 weight-gradient sweep (3 rings), which is 14 % slower with them
            const size_t cap = src_cin > 0 ? (size_t)DW_SRC_LDS_CAP : ((k == 5 && nrings == 3) ? 60 * 1024 : 78 * 1024);
            if (lds > cap) continue;
            const int nth = ((sx * cpw + 63) / 64) * 64;
            const int rc = iw * cgn;
            if ((rc + 63) / 64 > 2 * (nth / 64)) continue;          // <= 2 DMA blocks per wave per row
            const int strips = (W + tw - 1) / tw;
            const double util = (double)W / (strips * tw) * (double)(sx * cpw) / nth * (double)cps / (cblocks * cpw);
            const double halo = (double)iw / tw;
            const double occ = 0.55 + 0.45 * nth / 256.0;        // tiny workgroups starve the CU of waves
            const double score = util * occ / (0.6 + 0.4 * halo);
            if (score > best) { best = score; best_sx = sx; best_cpw = cpw; }
        }
    }
    if (best_sx == 0) return false;
    a->score = (float)best;
    const int cpw = best_cpw, cgn = cpw / 4;
    a->N = N; a->H = H; a->W = W; a->C = C;
    a->cpw = cpw; a->sx = best_sx; a->cgn = cgn;
    a->nthreads = ((best_sx * cpw + 63) / 64) * 64;
    a->iw = best_sx * DW_BW + k - 1;
    a->strips_x = (W + best_sx * DW_BW - 1) / (best_sx * DW_BW);
    a->cblocks = (cps + cpw - 1) / cpw;
    a->items = N * a->strips_x * a->cblocks;
    a->rc = a->iw * cgn;
    a->nb = (a->rc + 63) / 64;
    return true;
}

template <int RR>
__device__ __forceinline__ int dw_slot(int image_row) { return (image_row + RR) & (RR - 1); }   // rows >= -RR

__device__ __forceinline__ void dw_item(const DwArgs& a, int item, int& n, int& x0, int& c0) {
    const int cb = item % a.cblocks;
    const int r = item / a.cblocks;
    const int sxi = r % a.strips_x;
    n = r / a.strips_x;
    x0 = sxi * a.sx * DW_BW;
    c0 = cb * 2 * a.cpw;
}

// ---- DMA staging -------------------------------------------------------------------------------------------
// A ring row holds rc 16-byte chunks in (x, channel-group) order.  Wave w copies DMA blocks b = w and w + nwaves
// (64 chunks each) of every row; a lane's chunk within a block never changes, so its source offset and validity
// (column inside the image, channel group inside C) are computed once per item.
struct DwDma {
    int goff[2];        // uint4 offset of the lane's chunk relative to the row segment start
    bool ok[2];
};

template <int KS>
__device__ __forceinline__ void dw_dma_plan(const DwArgs& a, DwDma& p, int wave, int nwaves, int lane, int x0, int c0) {
    constexpr int PAD = KS / 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int b = wave + j * nwaves;
        const int q = b * 64 + lane;
        const int x = q / a.cgn, cgl = q - x * a.cgn;
        const int gx = x0 - PAD + x;
        p.ok[j] = (b < a.nb) && (q < a.rc) && gx >= 0 && gx < a.W && (c0 + cgl * 8) < a.C;
        p.goff[j] = x * (a.C >> 3) + cgl;
    }
}

// copy image rows [row0, row0+G) of one tensor (rows outside the image are skipped: readers test the row themselves)
template <int KS, int G>
__device__ __forceinline__ void dw_dma_rows(const DwArgs& a, const DwDma& p, uint32_t* ring, const uint4* __restrict__ src,
                                            int n, int row0, int x0, int c0, int wave, int nwaves) {
    constexpr int PAD = KS / 2;
    const int C8 = a.C >> 3;
#pragma unroll
    for (int r = 0; r < G; ++r) {
        const int gy = row0 + r;
        if (gy < 0 || gy >= a.H) continue;                                   // uniform
        const uint4* rowsrc = src + ((size_t)n * a.H + gy) * a.W * C8 + (ptrdiff_t)(x0 - PAD) * C8 + (c0 >> 3);
        uint32_t* rowdst = ring + (size_t)dw_slot<2 * G>(gy) * a.rc * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int b = wave + j * nwaves;
            if (b >= a.nb) continue;                                          // uniform
            if (p.ok[j])
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(rowsrc + p.goff[j]), (lds_void_ptr)(rowdst + b * 256), 16, 0, 0);
        }
    }
}

// A thread works on ONE channel pair: every per-element quantity is a float2 (lo, hi channel) so that the multiply-adds
// are v_pk_fma_f32 -- two FMAs per issue slot, 2x the plain-f32 VALU rate (these kernels carry no MFMAs and their inner
// loops are VALU-issue bound: 25 FMAs per 5x5 output element against ~4.5 B of HBM traffic).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 f2fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 f2bf(uint32_t u) { f2 r; r.x = bf_lo(u); r.y = bf_hi(u); return r; }   // bf16 pair -> f32 pair

// window row of activations: relu(s*x+t) (or x), zero outside the image columns (mz[xx] = 1.0 inside, 0.0 outside)
template <int WIN_W>
__device__ __forceinline__ void dw_read_act(const uint32_t* rowp, int ps, bool has_coef, f2 s, f2 t, unsigned colmask,
                                            f2 (&xr)[WIN_W]) {
#pragma unroll
    for (int xx = 0; xx < WIN_W; ++xx) {
        f2 v = f2bf(rowp[xx * ps]);
        if (has_coef) {
            v = f2fma(v, s, t);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
        }
        const bool in = (colmask >> xx) & 1u;
        xr[xx].x = in ? v.x : 0.f;
        xr[xx].y = in ? v.y : 0.f;
    }
}

// The same without the per-column select: the FORWARD sweep fills the out-of-image columns of its ring with a value that
// activates to zero (dw_fill_edges), so the row body carries no selects (the sweeps are VALU-issue bound: SQ_ACTIVE_INST_VALU x
// waves per SIMD = 75-100 %, profiles/r02_sq_counters.txt; the selects were 12-16 % of the 5x5 row body).
template <int WIN_W>
__device__ __forceinline__ void dw_read_act_nm(const uint32_t* rowp, int ps, bool has_coef, f2 s, f2 t, f2 (&xr)[WIN_W]) {
#pragma unroll
    for (int xx = 0; xx < WIN_W; ++xx) {
        f2 v = f2bf(rowp[xx * ps]);
        if (has_coef) {
            v = f2fma(v, s, t);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);       // v_max_f32(qNaN, 0) = 0 (IEEE maxNum)
        }
        xr[xx] = v;
    }
}
// Ring chunks of image columns outside [0, W) are never written by the DMA: give them, in every ring row, the bf16 pattern
// that reads as 0 after act-on-read -- a quiet NaN when relu(s*x+t) is applied (max(fma(NaN,s,t), 0) = 0 whatever the
// sign of s), plain zero for a materialised input.  Once per item (the columns depend on the strip only).
template <int RR>
__device__ __forceinline__ void dw_fill_edges(const DwArgs& a, const DwDma& p, uint32_t* ring, int wave, int nwaves, int lane,
                                              bool has_coef) {
    const uint32_t f = has_coef ? 0x7fc07fc0u : 0u;
    const uint4 f4 = make_uint4(f, f, f, f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int b = wave + j * nwaves, q = b * 64 + lane;
        if (b < a.nb && q < a.rc && !p.ok[j]) {
#pragma unroll
            for (int r = 0; r < RR; ++r) ((uint4*)ring)[(size_t)r * a.rc + q] = f4;
        }
    }
}

// window row of dy = c1*(g*[s*y+t>0]) + c2*y + c3, zero outside the image columns.  cf = {s,t,c1,c2,c3}
// GM ("g is masked", round 4): the producer of g -- mnas_pw_bwd's out-stage forms with gin_masked, which compute the mask for
// their fused reduce anyway -- stored dz = g*[s*y+t>0] instead of g: the window read drops the mask (one packed FMA, two
// compares, two selects per channel pair and column: 40 of the 5x5 row body's ~530 vector instructions)
// GA ("g affine", round 4): the stored gradient is read as g*ge + gz with per-(image, channel) ge, gz (the squeeze-excite backward
// g*sigmoid(u) + dz/HW formed on read instead of materialised by k_se_bwd_apply: csrc/mnas_se.hip)
template <int WIN_W, bool GM = false, bool GA = false>
__device__ __forceinline__ void dw_read_dy(const uint32_t* growp, const uint32_t* yrowp, int ps, const f2 (&cf)[5],
                                           unsigned colmask, f2 (&xr)[WIN_W], f2 ge = {1.f, 1.f}, f2 gz = {0.f, 0.f}) {
#pragma unroll
    for (int xx = 0; xx < WIN_W; ++xx) {
        const f2 g = f2bf(growp[xx * ps]), y = f2bf(yrowp[xx * ps]);
        f2 d;
        if constexpr (GA) {
            // ge / gz arrive pre-multiplied by c1 (the caller folds them once per item: one register pair more than the plain form
            // instead of two -- this kernel sits at 256 VGPRs and every further live value is reloaded from scratch inside the row
            // loop, each reload draining the ring DMA with it): dy = [s*y+t>0] * (g*c1*e + c1*z) + c2*y + c3
            const f2 z = f2fma(y, cf[0], cf[1]);
            const f2 t1 = f2fma(g, ge, gz);
            f2 dz;
            dz.x = (z.x > 0.f) ? t1.x : 0.f;
            dz.y = (z.y > 0.f) ? t1.y : 0.f;
            d = dz + f2fma(cf[3], y, cf[4]);
        } else {
            f2 dz = g;
            if constexpr (!GM) {
                const f2 z = f2fma(y, cf[0], cf[1]);
                dz.x = (z.x > 0.f) ? g.x : 0.f;
                dz.y = (z.y > 0.f) ? g.y : 0.f;
            }
            d = f2fma(cf[2], dz, f2fma(cf[3], y, cf[4]));
        }
        const bool in = (colmask >> xx) & 1u;
        xr[xx].x = in ? d.x : 0.f;
        xr[xx].y = in ? d.y : 0.f;
    }
}

// Deterministic workgroup reduction of per-thread accumulators v[NV] (channel pairs) (thread = channel pair cp x column strip sxi)
// over the column strips: every thread parks its values in LDS (the row rings are dead once the sweep is over; they
// are always large enough: sx*NV*cblk*4 B <= nrings*RR*iw*cpw*4 B), then NV*cblk threads add the sx copies in strip
// order.  No float atomics, so the partial tables are bit-reproducible run to run.
template <int NV, typename F>
__device__ __forceinline__ void dw_block_reduce(float* scratch, const f2 (&v)[NV], int cp, int sxi, int sx, int cblk,
                                                bool active, F&& emit) {
    __syncthreads();                                   // all ring reads done
    if (active) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            scratch[(sxi * NV + k) * cblk + 2 * cp] = v[k].x;
            scratch[(sxi * NV + k) * cblk + 2 * cp + 1] = v[k].y;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV * cblk; i += blockDim.x) {
        float acc = scratch[i];
        for (int sidx = 1; sidx < sx; ++sidx) acc += scratch[sidx * NV * cblk + i];
        emit(i / cblk, i % cblk, acc);
    }
}

// ---- forward -----------------------------------------------------------------------------------------------------
template <int KS, int G>
__global__ __launch_bounds__(256, (KS == 3 ? 4 : 3)) void k_dw_fwd(DwArgs a, MnasActIn in, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, uint32_t* __restrict__ out,
                                                                   float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PAD = KS / 2, WIN_W = DW_BW + KS - 1;
    const int cblk = 2 * a.cpw;
    uint32_t* ring = (uint32_t*)smem;                    // [RR][rc*4 dwords]; reused as reduction scratch at the end
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
    const int cp = tid % a.cpw, sxi = tid / a.cpw;
    const bool active = sxi < a.sx;
    const bool has_coef = in.scale != nullptr;
    const f2 zero2 = {0.f, 0.f};
    f2 s1 = zero2, s2 = zero2;
    int cur_c0 = -1;
    f2 wt[KS * KS], b2 = zero2, cs = {1.f, 1.f}, ct = zero2;
    const int nsteps = (a.H + 2 * PAD + G - 1) / G;
    const int ps = a.cpw;

    for (int item = blockIdx.x; item < a.items; item += a.geff) {
        int n, x0, c0;
        dw_item(a, item, n, x0, c0);
        const int ch = c0 + 2 * cp;
        const bool ch_ok = ch < a.C;
        if (c0 != cur_c0) {       // first item: a workgroup only ever sees ONE channel block (see dw_setup)
            cur_c0 = c0;
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) {
                wt[t].x = ch_ok ? w[(size_t)t * a.C + ch] : 0.f;
                wt[t].y = ch_ok ? w[(size_t)t * a.C + ch + 1] : 0.f;
            }
            b2.x = (bias && ch_ok) ? bias[ch] : 0.f;
            b2.y = (bias && ch_ok) ? bias[ch + 1] : 0.f;
            if (has_coef && ch_ok) { cs.x = in.scale[ch]; cs.y = in.scale[ch + 1]; ct.x = in.shift[ch]; ct.y = in.shift[ch + 1]; }
        }
        DwDma plan;
        dw_dma_plan<KS>(a, plan, wave, nwaves, lane, x0, c0);
        const int gx0 = x0 + sxi * DW_BW;
        // register ring of KS partial output rows: A[i] = output row (iy - PAD + i) while input row iy is processed
        f2 A[KS][DW_BW];
#pragma unroll
        for (int i = 0; i < KS; ++i)
#pragma unroll
            for (int j = 0; j < DW_BW; ++j) A[i][j] = b2;
        uint32_t* outp = out + (((size_t)n * a.H * a.W + gx0) * a.C + ch) / 2;
        const uint32_t* colp = ring + (size_t)sxi * DW_BW * ps + cp;

        // Every input row is read from LDS exactly once (the vertical window lives in the register ring A), so the 8 ring
        // rows are two buffers of G = 4: the DMA of group s+1 is in flight while group s is computed.  One barrier per
        // step (dma_barrier: vmcnt(0) then s_barrier): it publishes group s and retires the readers of group s-1, whose
        // buffer the next DMA overwrites.
        __syncthreads();                             // previous item's last group consumed
        dw_fill_edges<2 * G>(a, plan, ring, wave, nwaves, lane, has_coef);
        dw_dma_rows<KS, G>(a, plan, ring, (const uint4*)in.data, n, -PAD, x0, c0, wave, nwaves);
        for (int s = 0; s < nsteps; ++s) {
            const int r0 = -PAD + s * G;
            dma_barrier();                           // group s landed (every wave's own DMA) + readers of group s-1 retired
            if (s + 1 < nsteps) dw_dma_rows<KS, G>(a, plan, ring, (const uint4*)in.data, n, r0 + G, x0, c0, wave, nwaves);
            if (!active) continue;
            // unrolled over the row group (3x3: all G rows, 5x5: two -- more spills past 168 VGPRs): the register-ring shift of
            // consecutive rows becomes renaming instead of KS*DW_BW 64-bit moves per row (3x3 at 112x112: 146 -> 132 us)
#pragma unroll (KS == 3 ? G : (G == 4 ? 2 : 1))
            for (int j = 0; j < G; ++j) {
                const int iy = r0 + j;
                const int oy = iy - PAD;             // A[0] is complete after this row
                if (iy >= 0 && iy < a.H) {           // uniform: rows outside the image contribute nothing
                    f2 xr[WIN_W];
                    dw_read_act_nm<WIN_W>(colp + (size_t)dw_slot<(2 * G)>(iy) * a.rc * 4, ps, has_coef, cs, ct, xr);
#pragma unroll
                    for (int i = 0; i < KS; ++i)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                            for (int kx = 0; kx < KS; ++kx)
                                A[i][ox] = f2fma(wt[(KS - 1 - i) * KS + kx], xr[ox + kx], A[i][ox]);
                }
                if (oy >= 0 && oy < a.H && ch_ok) {
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) {
                        if (gx0 + ox < a.W) {
                            const f2 v = A[0][ox];
                            s1 += v;
                            s2 = f2fma(v, v, s2);
                            st_u1(outp + ((size_t)oy * a.W + ox) * a.C / 2, pack_bf16(v.x, v.y), a.nt);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i + 1 < KS; ++i)
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) A[i][ox] = A[i + 1][ox];
#pragma unroll
                for (int ox = 0; ox < DW_BW; ++ox) A[KS - 1][ox] = b2;
            }
        }
    }
    if (stats) {
        // stats table is float[2][C][rows], rows = geff / cblocks: workgroup b owns column b / cblocks of its own
        // channel block (b % cblocks), so every (channel, column) is written exactly once -- no zero fill, no atomics
        const int rows = a.geff / a.cblocks, col = blockIdx.x / a.cblocks;
        const int cb0 = (blockIdx.x % a.cblocks) * cblk;
        const f2 sv[2] = {s1, s2};
        const bool any = cur_c0 >= 0;
        dw_block_reduce<2>((float*)ring, sv, cp, sxi, a.sx, cblk, active, [&](int r, int cl, float v) {
            const int c = cb0 + cl;
            if (c < a.C) stats[((size_t)r * a.C + c) * rows + col] = any ? v : 0.f;
        });
    }
}


// ---- forward, fused with the preceding 1x1 expand conv (MBConv_block, mnasnet.py:116-125) ------------------------------
// The ring rows are not copied from HBM: they are COMPUTED from the block input x (Cin channels, the small tensor) on the
// matrix cores: y1[pixel][c0..c0+cblk) = W1 act(x)[pixel] + b1 for the G x iw patch of the next row group, written to the
// ring as the same bf16 image the DMA would have delivered.  The expand conv's BatchNorm coefficients (`in.scale/shift`,
// applied when the depthwise window is read) are known before this kernel runs: mnas_gram + mnas_gram_bn_finalize derive
// them from the covariance of x.  The x fragments (8 consecutive input channels of one pixel = 16 contiguous bytes in
// NHWC) go straight from global memory to registers one row group ahead; W1's block is LDS-resident.  With e.y1 == NULL the
// expanded tensor never reaches HBM; otherwise the workgroup also stores the rows it owns (backward of the unfused form).
struct DwExp {
    const uint16_t* x;       // block input (N,H,W,Cin) bf16
    const float* xs;         // act-on-load coefficients of x ([Cin]) or NULL
    const float* xt;
    const uint16_t* w1;      // MNAS_PACK_FWD weights of the expand conv: [C rounded to 16][Kpad]
    const float* b1;         // [C] or NULL
    uint32_t* y1;            // raw expand output (N,H,W,C) or NULL
    int Cin, Kpad;
};

template <int KS, int G, int KST>
__global__ __launch_bounds__(256, 2) void k_dw_fwd_exp(DwArgs a, DwExp e, MnasActIn in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, uint32_t* __restrict__ out,
                                                       float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PAD = KS / 2, WIN_W = DW_BW + KS - 1;
    const int cblk = 2 * a.cpw;
    const int ntb = (cblk + 15) >> 4;                     // MFMA cout tiles of the channel block
    const int ldw = e.Kpad + 8;
    uint32_t* ring = (uint32_t*)smem;                     // [2G][rc*4 dwords]
    uint16_t* lds_w1 = (uint16_t*)(ring + (size_t)(2 * G) * a.rc * 4);      // [ntb*16][ldw]
    float* lds_b1 = (float*)(lds_w1 + (size_t)ntb * 16 * ldw);             // [ntb*16]
    float* lds_xc = lds_b1 + ntb * 16;                                      // [2][Kpad]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int cp = tid % a.cpw, sxi = tid / a.cpw;
    const bool active = sxi < a.sx;
    const bool has_coef = in.scale != nullptr, has_xc = e.xs != nullptr;
    const f2 zero2 = {0.f, 0.f};
    f2 s1 = zero2, s2 = zero2;
    f2 wt[KS * KS], b2 = zero2, cs = {1.f, 1.f}, ct = zero2;
    const int nsteps = (a.H + 2 * PAD + G - 1) / G;
    const int ps = a.cpw;
    const int npg = (G * a.iw + 15) >> 4;                 // 16-pixel groups of the G x iw patch

    // ---- per-workgroup constants: its channel block never changes (geff is a multiple of cblocks)
    const int c0 = ((int)blockIdx.x % a.cblocks) * cblk;
    {
        const int ch = c0 + 2 * cp;
        const bool ch_ok = ch < a.C;
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) {
            wt[t].x = ch_ok ? w[(size_t)t * a.C + ch] : 0.f;
            wt[t].y = ch_ok ? w[(size_t)t * a.C + ch + 1] : 0.f;
        }
        b2.x = (bias && ch_ok) ? bias[ch] : 0.f;
        b2.y = (bias && ch_ok) ? bias[ch + 1] : 0.f;
        if (has_coef && ch_ok) { cs.x = in.scale[ch]; cs.y = in.scale[ch + 1]; ct.x = in.shift[ch]; ct.y = in.shift[ch + 1]; }
        const int kc8n = e.Kpad >> 3, cpad16 = (a.C + 15) / 16 * 16;
        for (int q = tid; q < ntb * 16 * kc8n; q += blockDim.x) {
            const int r = q / kc8n, k8 = q - r * kc8n;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < cblk && c0 + r < cpad16) v = *(const uint4*)(e.w1 + (size_t)(c0 + r) * e.Kpad + k8 * 8);
            *(uint4*)(lds_w1 + r * ldw + k8 * 8) = v;
        }
        for (int i = tid; i < ntb * 16; i += blockDim.x) lds_b1[i] = (e.b1 && i < cblk && c0 + i < a.C) ? e.b1[c0 + i] : 0.f;
        for (int i = tid; i < 2 * e.Kpad; i += blockDim.x) {
            const int r = i / e.Kpad, c = i - r * e.Kpad;
            lds_xc[i] = (has_xc && c < e.Cin) ? (r == 0 ? e.xs[c] : e.xt[c]) : 0.f;
        }
    }
    // ---- expand plan: this wave's pixel groups pg = wave + nwaves*i; lane l15 <-> patch pixel pi = pg*16 + l15 = (r, xx)
    int pr_[DW_EXP_MAXPG], px_[DW_EXP_MAXPG];
#pragma unroll
    for (int i = 0; i < DW_EXP_MAXPG; ++i) {
        const int pi = (wave + nwaves * i) * 16 + l15;
        pr_[i] = pi / a.iw; px_[i] = pi - pr_[i] * a.iw;
        if (wave + nwaves * i >= npg || pi >= G * a.iw) pr_[i] = -1;
    }
    uint4 xf[DW_EXP_MAXPG][KST];
    __syncthreads();

    for (int item = blockIdx.x; item < a.items; item += a.geff) {
        int n, x0, c0i;
        dw_item(a, item, n, x0, c0i);
        const int ch = c0 + 2 * cp;
        const bool ch_ok = ch < a.C;
        const int gx0 = x0 + sxi * DW_BW;
        unsigned colmask = 0;
#pragma unroll
        for (int xx = 0; xx < WIN_W; ++xx) { const int gx = gx0 - PAD + xx; colmask |= (gx >= 0 && gx < a.W) ? (1u << xx) : 0u; }
        f2 A[KS][DW_BW];
#pragma unroll
        for (int i = 0; i < KS; ++i)
#pragma unroll
            for (int j = 0; j < DW_BW; ++j) A[i][j] = b2;
        uint32_t* outp = out + (((size_t)n * a.H * a.W + gx0) * a.C + ch) / 2;
        const uint32_t* colp = ring + (size_t)sxi * DW_BW * ps + cp;

        // x fragments of the row group starting at image row r0 (registers; consumed by expand_rows one step later)
        auto load_x = [&](int r0) {
#pragma unroll
            for (int i = 0; i < DW_EXP_MAXPG; ++i) {
                const int gy = r0 + pr_[i], gx = x0 - PAD + px_[i];
                const bool ok = pr_[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const uint16_t* src = e.x + (((size_t)n * a.H + gy) * a.W + gx) * e.Cin + lg * 8;
#pragma unroll
                for (int ks = 0; ks < KST; ++ks)
                    xf[i][ks] = (ok && ks * 32 + lg * 8 < e.Cin) ? *(const uint4*)(src + ks * 32) : make_uint4(0, 0, 0, 0);
            }
        };
        // y1 rows [r0, r0+G) of the patch -> ring (and, for the pixels this workgroup owns, HBM)
        auto expand_rows = [&](int r0) {
#pragma unroll
            for (int i = 0; i < DW_EXP_MAXPG; ++i) {
                if (wave + nwaves * i >= npg) break;                      // uniform per wave
                const int gy = r0 + pr_[i], gx = x0 - PAD + px_[i];
                const bool ok = pr_[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                bf16x8_t bfrag[KST];
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    uint4 v = xf[i][ks];
                    if (has_xc && ok) {
                        const int k = ks * 32 + lg * 8;
                        float sc[8], sh[8];
                        *(float4*)&sc[0] = *(const float4*)(lds_xc + k);
                        *(float4*)&sc[4] = *(const float4*)(lds_xc + k + 4);
                        *(float4*)&sh[0] = *(const float4*)(lds_xc + e.Kpad + k);
                        *(float4*)&sh[4] = *(const float4*)(lds_xc + e.Kpad + k + 4);
                        v = act8(v, sc, sh);
                        if (k >= e.Cin) v = make_uint4(0, 0, 0, 0);
                    }
                    bfrag[ks] = *(const bf16x8_t*)&v;
                }
                const bool own = ok && px_[i] >= PAD && px_[i] < PAD + a.sx * DW_BW;
                uint32_t* ringp = ring + (size_t)dw_slot<2 * G>(gy) * a.rc * 4 + (size_t)px_[i] * a.cgn * 4 + (lg >> 1) * 4 + (lg & 1) * 2;
                for (int nt = 0; nt < ntb; ++nt) {
                    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) {
                        const bf16x8_t afrag = *(const bf16x8_t*)(lds_w1 + (nt * 16 + l15) * ldw + ks * 32 + lg * 8);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[ks], acc, 0, 0, 0);
                    }
                    const int cl = nt * 16 + lg * 4;                      // first of this lane's 4 channels inside the block
                    const float4 bb = *(const float4*)(lds_b1 + cl);
                    uint2 pk;
                    pk.x = pack_bf16(acc[0] + bb.x, acc[1] + bb.y);
                    pk.y = pack_bf16(acc[2] + bb.z, acc[3] + bb.w);
                    if (ok && cl < cblk) {
                        *(uint2*)(ringp + nt * 8) = pk;
                        if (e.y1 && own && c0 + cl < a.C)
                            st_u2(e.y1 + (((size_t)n * a.H + gy) * a.W + gx) * a.C / 2 + (c0 + cl) / 2, pk, a.nt);
                    }
                }
            }
        };

        __syncthreads();                             // previous item's last group consumed
        load_x(-PAD);
        expand_rows(-PAD);
        if (nsteps > 1) load_x(-PAD + G);
        for (int s = 0; s < nsteps; ++s) {
            const int r0 = -PAD + s * G;
            __syncthreads();                         // group s is in the ring; readers of group s-1 are done with the other buffer
            if (s + 1 < nsteps) {
                expand_rows(r0 + G);
                if (s + 2 < nsteps) load_x(r0 + 2 * G);
            }
            if (!active) continue;
#pragma unroll 1
            for (int j = 0; j < G; ++j) {
                const int iy = r0 + j;
                const int oy = iy - PAD;
                if (iy >= 0 && iy < a.H) {
                    f2 xr[WIN_W];
                    dw_read_act<WIN_W>(colp + (size_t)dw_slot<(2 * G)>(iy) * a.rc * 4, ps, has_coef, cs, ct, colmask, xr);
#pragma unroll
                    for (int i = 0; i < KS; ++i)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                            for (int kx = 0; kx < KS; ++kx)
                                A[i][ox] = f2fma(wt[(KS - 1 - i) * KS + kx], xr[ox + kx], A[i][ox]);
                }
                if (oy >= 0 && oy < a.H && ch_ok) {
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) {
                        if (gx0 + ox < a.W) {
                            const f2 v = A[0][ox];
                            s1 += v;
                            s2 = f2fma(v, v, s2);
                            st_u1(outp + ((size_t)oy * a.W + ox) * a.C / 2, pack_bf16(v.x, v.y), a.nt);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i + 1 < KS; ++i)
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) A[i][ox] = A[i + 1][ox];
#pragma unroll
                for (int ox = 0; ox < DW_BW; ++ox) A[KS - 1][ox] = b2;
            }
        }
    }
    if (stats) {
        const int rows = a.geff / a.cblocks, col = blockIdx.x / a.cblocks;
        const f2 sv[2] = {s1, s2};
        const bool any = (int)blockIdx.x < a.items;
        dw_block_reduce<2>((float*)ring, sv, cp, sxi, a.sx, cblk, active, [&](int r, int cl, float v) {
            const int c = c0 + cl;
            if (c < a.C) stats[((size_t)r * a.C + c) * rows + col] = any ? v : 0.f;
        });
    }
}

// ---- backward: input gradient (DG), weight gradient (WG), fused BN-backward reduce of the producer of x (RED) ----
// Per image row iy (rows iy of g,y and iy-PAD.. of x are in the rings):
//   xr = dy row iy, k+3 columns  -> DG: scattered into the register ring A of KS partial gin rows (flipped filter);
//                                   its centre 4 columns become D[0] of the dy ring (D[q] = dy row iy-q)
//   xa = act(x) row r = iy-PAD   -> WG: wacc[ky][kx] += D[ky][ox] * xa[ox+kx]      (dy rows r-ky+PAD = iy-ky)
//   RED: the raw x centre values of row iy-PAD are in the x ring too: sum dz, sum dz*xhat for the emitted gin row.
// SRC (round 4, the spatially tiled fused inverted-residual block): the g ring and the x ring are not copied from HBM -- the
// tensors they would be copied from (g2 = dy3 . W3, the project conv's input gradient, and y1 = W1 act(x) + b1, the expand
// conv's raw output: both t times wider than the block) were never written.  Their rows are COMPUTED on the matrix cores, one
// row group ahead of the sweep, from the two NARROW tensors of the block (dy3 and the block input x, Cin channels), whose row
// segments are staged in LDS by DMA one further group ahead.  The epilogues round exactly as the kernels that used to store
// the tensors did (same MFMA, same k order), so everything downstream is bit-identical to the per-layer path.
// WAVE SPECIALISATION: the workgroup carries ONE extra wave (the last one) that does nothing but this: it issues the staging
// DMA and runs the MFMAs + epilogues of the next group while the sweep waves work the current group on the vector ALUs -- the
// matrix pipe and the vector pipe of a CU are busy at the same time, from different waves, and the producer's dependent
// LDS -> MFMA -> pack -> LDS chains sit on nobody's critical path (folded into the sweep waves, the same work made the launch
// 2x slower: 614 vs 315 us at 112x112, 43 spilled VGPRs).  The sweep waves are exactly those of the plain kernel.
struct DwSrc {
    const uint16_t* x;       // block input (N,H,W,Cin) bf16
    const float* xs;         // its act-on-load coefficients ([Cin]) or NULL
    const float* xt;
    const uint16_t* w1;      // MNAS_PACK_FWD weights of the expand conv [C rounded to 16][Kpad]
    const float* b1;         // [C] or NULL
    const uint16_t* dy;      // materialised dy of the project conv (N,H,W,Cin) bf16
    const uint16_t* w3t;     // MNAS_PACK_DGRAD weights of the project conv [C rounded to 16][Kpad]
    int Cin, Kpad;
};

template <int KS, bool DG, bool WG, bool RED, int G, bool SRC = false, int NTB = 1, bool GM = false, bool GA = false>
__global__ __launch_bounds__(256, ((KS == 5 && DG && (WG || RED)) ? 2 : 3)) void k_dw_bwd(
    DwArgs a, MnasActIn x, MnasGradIn d, const float* __restrict__ w, uint32_t* __restrict__ gin, float* __restrict__ wpartial,
    float* __restrict__ red_partial, const float* __restrict__ red_bn, DwSrc e, const float* __restrict__ g_gate = nullptr,
    const float* __restrict__ g_bias = nullptr) {
    static_assert(!SRC || (DG && WG), "the SRC form is the fused sweep");
    static_assert(!GA || (!SRC && !GM), "g affine: plain fused sweep");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PAD = KS / 2, WIN_W = DW_BW + KS - 1;
    constexpr bool NEEDX = WG;                 // x ring only when the weight gradient is computed here
    constexpr bool REDG = RED && !WG;          // input-gradient-only launch: the reduce reads raw x from global (2 rings)
    const int cblk = 2 * a.cpw;
    uint32_t* ring_g = (uint32_t*)smem;                      // rings; reused as reduction scratch at the end
    uint32_t* ring_y = ring_g + (size_t)(2 * G) * a.rc * 4;
    uint32_t* ring_x = ring_y + (size_t)(2 * G) * a.rc * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = (blockDim.x >> 6) - (SRC ? 1 : 0);   // sweep waves
    const bool is_prod = SRC && wave == nwaves;           // SRC: the extra (last) wave produces the g / x ring rows
    const int cp = tid % a.cpw, sxi = tid / a.cpw;
    const bool active = sxi < a.sx;                       // (false for the producer wave: tid >= sx * cpw)
    const bool has_coef = x.scale != nullptr;
    const uint32_t* xglob = (const uint32_t*)x.data;
    // ---- SRC: weight blocks, coefficient tables and the staging rings of the two narrow tensors
    const int ntb = (cblk + 15) >> 4;                     // MFMA cout tiles of the channel block
    const int ldw = e.Kpad + 8;
    const int cgs = e.Cin >> 3, rcs = a.iw * cgs;         // staging: 16-byte chunks per pixel / per ring row
    float* lds_ga = (float*)(ring_x + (size_t)(2 * G) * a.rc * 4);            // GA: [2][cblk] (c1*gate, c1*bias) of the current item
    uint16_t* lds_w1 = (uint16_t*)(ring_x + (size_t)(2 * G) * a.rc * 4);      // [ntb*16][ldw]
    uint16_t* lds_w3 = lds_w1 + (size_t)ntb * 16 * ldw;                       // [ntb*16][ldw]
    float* lds_b1 = (float*)(lds_w3 + (size_t)ntb * 16 * ldw);               // [ntb*16]
    float* lds_xc = lds_b1 + ntb * 16;                                        // [2][Kpad]
    uint16_t* stage_d = (uint16_t*)(lds_xc + 2 * e.Kpad);                     // [2G][rcs*8]
    uint16_t* stage_x = stage_d + (size_t)(2 * G) * rcs * 8;                  // [2G][rcs*8]
    const int l15 = lane & 15, lg = lane >> 4;
    const bool has_xc = SRC && e.xs != nullptr;
    const int npg = (G * a.iw + 15) >> 4;                 // 16-pixel groups of the G x iw patch
    if constexpr (SRC) {
        const int c0w = ((int)blockIdx.x % a.cblocks) * cblk;          // a workgroup only ever sees one channel block
        const int kc8n = e.Kpad >> 3, cpad16 = (a.C + 15) / 16 * 16;
        for (int q = tid; q < ntb * 16 * kc8n; q += blockDim.x) {
            const int r = q / kc8n, k8 = q - r * kc8n;
            uint4 v1 = make_uint4(0, 0, 0, 0), v3 = make_uint4(0, 0, 0, 0);
            if (r < cblk && c0w + r < cpad16) {
                v1 = *(const uint4*)(e.w1 + (size_t)(c0w + r) * e.Kpad + k8 * 8);
                v3 = *(const uint4*)(e.w3t + (size_t)(c0w + r) * e.Kpad + k8 * 8);
            }
            *(uint4*)(lds_w1 + r * ldw + k8 * 8) = v1;
            *(uint4*)(lds_w3 + r * ldw + k8 * 8) = v3;
        }
        for (int i = tid; i < ntb * 16; i += blockDim.x) lds_b1[i] = (e.b1 && i < cblk && c0w + i < a.C) ? e.b1[c0w + i] : 0.f;
        for (int i = tid; i < 2 * e.Kpad; i += blockDim.x) {
            const int r = i / e.Kpad, c = i - r * e.Kpad;
            lds_xc[i] = (has_xc && c < e.Cin) ? (r == 0 ? e.xs[c] : e.xt[c]) : 0.f;
        }
    }
    const f2 zero2 = {0.f, 0.f};
    int cur_c0 = -1;
    f2 wt[DG ? KS * KS : 1], wacc[WG ? KS * KS : 1];
    f2 cf[5], cs = {1.f, 1.f}, ct = zero2;
    f2 ris = zero2, rmu = zero2;
    f2 s1 = zero2, s2 = zero2;
#pragma unroll
    for (int t = 0; t < (WG ? KS * KS : 1); ++t) wacc[t] = zero2;
    const int nsteps = (a.H + 2 * PAD + G - 1) / G;
    const int ps = a.cpw;

    if (SRC && is_prod) {
      if constexpr (SRC) {
        // ================= producer wave: its own loop nest (disjoint live ranges: the kernel's register count is the larger of
        // the two roles, not their sum), meeting the sweep waves at the same barriers =================
        __syncthreads();                                  // weight blocks / tables are in LDS
        bf16x8_t af1[NTB], af3[NTB];                      // expand / project^T weight fragments, bias: registers for the whole kernel
        float4 bb1[NTB];
        float xsc[8], xsh[8];
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) {
            af1[nt] = *(const bf16x8_t*)(lds_w1 + (nt * 16 + l15) * ldw + lg * 8);
            af3[nt] = *(const bf16x8_t*)(lds_w3 + (nt * 16 + l15) * ldw + lg * 8);
            bb1[nt] = *(const float4*)(lds_b1 + nt * 16 + lg * 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { xsc[j] = lds_xc[(lg * 8 + j) & (e.Kpad - 1)]; xsh[j] = lds_xc[e.Kpad + ((lg * 8 + j) & (e.Kpad - 1))]; }
        for (int item = blockIdx.x; item < a.items; item += a.geff) {
            int n, x0, c0;
            dw_item(a, item, n, x0, c0);
            cur_c0 = c0;
            // ---- staging: staging DMA of the narrow tensors' row segments (chunk q of a staging row = chunk q of the HBM
            // row segment: all Cin channels of iw pixels, contiguous)
            auto stage_rows = [&](uint16_t* stage, const uint16_t* src, int row0) {
#pragma unroll
                for (int r = 0; r < G; ++r) {
                    const int gy = row0 + r;
                    if (gy < 0 || gy >= a.H) continue;                                   // uniform
                    const uint4* rowsrc = (const uint4*)src + (((size_t)n * a.H + gy) * a.W + (ptrdiff_t)(x0 - PAD)) * cgs;
                    uint32_t* rowdst = (uint32_t*)stage + (size_t)dw_slot<2 * G>(gy) * rcs * 4;
                    for (int b = 0; b * 64 < rcs; ++b) {
                        const int q = b * 64 + lane, gx = x0 - PAD + q / cgs;
                        if (q < rcs && gx >= 0 && gx < a.W)
                            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(rowsrc + q), (lds_void_ptr)(rowdst + b * 256), 16, 0, 0);
                    }
                }
            };
            auto stage_group = [&](int r0) {          // group r0: dy rows [r0, r0+G) (-> ring_g), x rows [r0-PAD, r0-PAD+G) (-> ring_x)
                stage_rows(stage_d, e.dy, r0);
                stage_rows(stage_x, e.x, r0 - PAD);
            };
            // Ring rows of group r0 computed from the staged rows: D[c][pixel] = W[c][:] . in[pixel][:] (+ bias), rounded to bf16.
            // Lane l15 <-> pixel pg*16 + l15 of the G x iw patch; a lane's 4 accumulators are 4 consecutive channels of that pixel.
            // Both producers of a pixel group run together: 2 LDS reads, 2*NTB independent MFMAs back to back (weight fragments and
            // bias live in registers for the whole kernel), then the 2*NTB pack + store epilogues.
            auto produce_group = [&](int r0) {
                for (int pg = 0; pg < npg; ++pg) {
                    const int pi = pg * 16 + l15;
                    const int pr = pi / a.iw, px = pi - pr * a.iw;
                    const int gx = x0 - PAD + px;
                    const bool cok = pi < G * a.iw && gx >= 0 && gx < a.W;
                    const int gyd = r0 + pr, gyx = r0 - PAD + pr;
                    const bool okd = cok && gyd >= 0 && gyd < a.H, okx = cok && gyx >= 0 && gyx < a.H;
                    const int sd = dw_slot<2 * G>(okd ? gyd : 0), sx_ = dw_slot<2 * G>(okx ? gyx : 0);
                    uint4 vd = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
                    if (lg * 8 < e.Cin) {
                        if (okd) vd = *(const uint4*)(stage_d + (size_t)sd * rcs * 8 + px * e.Cin + lg * 8);
                        if (okx) vx = *(const uint4*)(stage_x + (size_t)sx_ * rcs * 8 + px * e.Cin + lg * 8);
                        if (has_xc && okx) vx = act8(vx, xsc, xsh);
                    }
                    const bf16x8_t bd = *(const bf16x8_t*)&vd, bx = *(const bf16x8_t*)&vx;
                    f32x4_t accd[NTB], accx[NTB];
#pragma unroll
                    for (int nt = 0; nt < NTB; ++nt) {
                        accd[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af3[nt], bd, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1[nt], bx, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    }
                    uint32_t* rpd = ring_g + (size_t)sd * a.rc * 4 + (size_t)px * a.cgn * 4 + lg * 2;
                    uint32_t* rpx = ring_x + (size_t)sx_ * a.rc * 4 + (size_t)px * a.cgn * 4 + lg * 2;
#pragma unroll
                    for (int nt = 0; nt < NTB; ++nt) {
                        const bool cin_blk = nt * 16 + lg * 4 < cblk;
                        uint2 pk;
                        pk.x = pack_bf16(accd[nt][0], accd[nt][1]);
                        pk.y = pack_bf16(accd[nt][2], accd[nt][3]);
                        if (okd && cin_blk) *(uint2*)(rpd + nt * 8) = pk;
                        pk.x = pack_bf16(accx[nt][0] + bb1[nt].x, accx[nt][1] + bb1[nt].y);
                        pk.y = pack_bf16(accx[nt][2] + bb1[nt].z, accx[nt][3] + bb1[nt].w);
                        if (okx && cin_blk) *(uint2*)(rpx + nt * 8) = pk;
                    }
                }
            };

            __syncthreads();                             // (sweep waves: previous item's last group consumed)
            stage_group(-PAD);
            dma_barrier();                               // staging of group 0 landed
            if (nsteps > 1) stage_group(-PAD + G);
            produce_group(-PAD);
            for (int s = 0; s < nsteps; ++s) {
                const int r0 = -PAD + s * G;
                dma_barrier();                           // staging of group s+1 landed; readers of ring buffer (s+1)%2 retired
                if (s + 1 < nsteps) {
                    if (s + 2 < nsteps) stage_group(r0 + 2 * G);
                    produce_group(r0 + G);
                }
            }
        }
      }
    } else {
    if constexpr (SRC) __syncthreads();                   // (producer wave: weight blocks visible)
    for (int item = blockIdx.x; item < a.items; item += a.geff) {
        int n, x0, c0;
        dw_item(a, item, n, x0, c0);
        const int ch = c0 + 2 * cp;
        const bool ch_ok = ch < a.C;
        if (c0 != cur_c0) {
            cur_c0 = c0;
            if (DG) {
#pragma unroll
                for (int t = 0; t < KS * KS; ++t) {      // flipped filter for the input gradient
                    wt[t].x = ch_ok ? w[(size_t)(KS * KS - 1 - t) * a.C + ch] : 0.f;
                    wt[t].y = ch_ok ? w[(size_t)(KS * KS - 1 - t) * a.C + ch + 1] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                cf[r].x = ch_ok ? d.coef[(size_t)r * a.C + ch] : 0.f;
                cf[r].y = ch_ok ? d.coef[(size_t)r * a.C + ch + 1] : 0.f;
            }
            if (has_coef && ch_ok) { cs.x = x.scale[ch]; cs.y = x.scale[ch + 1]; ct.x = x.shift[ch]; ct.y = x.shift[ch + 1]; }
            if (RED && ch_ok) {
                ris.x = red_bn[6 * a.C + ch]; ris.y = red_bn[6 * a.C + ch + 1];
                rmu.x = -red_bn[5 * a.C + ch] * ris.x; rmu.y = -red_bn[5 * a.C + ch + 1] * ris.y;
            }
        }
        f2 ge = {1.f, 1.f}, gz = zero2;

        DwDma plan;
        dw_dma_plan<KS>(a, plan, wave, nwaves, lane, x0, c0);
        const int gx0 = x0 + sxi * DW_BW;
        unsigned colmask = 0;
#pragma unroll
        for (int xx = 0; xx < WIN_W; ++xx) { const int gx = gx0 - PAD + xx; colmask |= (gx >= 0 && gx < a.W) ? (1u << xx) : 0u; }
        f2 A[DG ? KS : 1][DW_BW], D[WG ? KS : 1][DW_BW];
#pragma unroll
        for (int i = 0; i < (DG ? KS : 1); ++i)
#pragma unroll
            for (int j = 0; j < DW_BW; ++j) A[i][j] = zero2;
#pragma unroll
        for (int i = 0; i < (WG ? KS : 1); ++i)
#pragma unroll
            for (int j = 0; j < DW_BW; ++j) D[i][j] = zero2;
        const size_t obase = (((size_t)n * a.H * a.W + gx0) * a.C + ch) / 2;
        const size_t coloff = (size_t)sxi * DW_BW * ps + cp;

        // double-buffered groups of G rows (see k_dw_fwd); the x ring runs PAD rows behind g/y: row iy of dy meets row
        // oy = iy - PAD of x, and each row of either is read from LDS exactly once
        auto dma_group = [&](int r0) {
            if constexpr (!SRC) dw_dma_rows<KS, G>(a, plan, ring_g, (const uint4*)d.g, n, r0, x0, c0, wave, nwaves);
            dw_dma_rows<KS, G>(a, plan, ring_y, (const uint4*)d.y, n, r0, x0, c0, wave, nwaves);
            if constexpr (!SRC) { if (NEEDX) dw_dma_rows<KS, G>(a, plan, ring_x, (const uint4*)x.data, n, r0 - PAD, x0, c0, wave, nwaves); }
        };
        // input-gradient-only launch with the fused reduce: the raw x values of the rows it emits come from global memory,
        // fetched one group AHEAD (with the DMA, before the barrier that drains vmcnt) so that no load issued inside the
        // compute phase has to wait behind the in-flight DMA of the next group (vmcnt retires in order)
        uint32_t xq[REDG ? G : 1][DW_BW], xn[REDG ? G : 1][DW_BW];
        auto load_xn = [&](int r0) {
            if constexpr (REDG) {
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int oy = r0 - PAD + j;
                    const bool ok = oy >= 0 && oy < a.H && ch_ok && active;
                    const uint32_t* yp = xglob + obase + (size_t)oy * a.W * a.C / 2;
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) xn[j][ox] = (ok && gx0 + ox < a.W) ? yp[(size_t)ox * a.C / 2] : 0u;
                }
            }
        };
        __syncthreads();                             // previous item's last group consumed
        if constexpr (GA) {
            // per-item table (c1*gate, c1*bias) of this image's channel block, in LDS: the row loop fetches the pair it needs with
            // the window reads.  As registers the two pairs pushed this 256-VGPR kernel's spill reloads (scratch = vector memory)
            // behind the DMA issue of every row group -- a reload waits for everything older, so each group's DMA was drained
            // before its predecessor was computed: 1.9x slower at 112x112
            if (sxi == 0) {
                const size_t o = (size_t)n * a.C + (ch_ok ? ch : 0);
                const float e0 = g_gate[o], e1 = g_gate[o + 1], z0 = g_bias[o], z1 = g_bias[o + 1];
                f2 ce, cz;
                ce.x = ch_ok ? e0 : 1.f; ce.y = ch_ok ? e1 : 1.f;
                cz.x = ch_ok ? z0 : 0.f; cz.y = ch_ok ? z1 : 0.f;
                *(f2*)(lds_ga + 2 * cp) = ce * cf[2];
                *(f2*)(lds_ga + cblk + 2 * cp) = cz * cf[2];
            }
        }
#if MNAS_DW_XFILL
        if (NEEDX) dw_fill_edges<2 * G>(a, plan, ring_x, wave, nwaves, lane, has_coef);   // out-of-image columns of x act to 0
#endif
        dma_group(-PAD);
        if constexpr (SRC) dma_barrier();            // (producer wave: staging of group 0 landed -> it produces the rings of group 0)
        load_xn(-PAD);
        for (int s = 0; s < nsteps; ++s) {
            const int r0 = -PAD + s * G;
            dma_barrier();                           // group s landed (every wave's own DMA) + readers of group s-1 retired
            // (SRC: the rings of group s are complete -- y by DMA, g / x produced by the producer wave during step s-1)
            if constexpr (REDG) {
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) xq[j][ox] = xn[j][ox];
            }
            if (s + 1 < nsteps) { dma_group(r0 + G); load_xn(r0 + G); }
            if (!active) continue;
#pragma unroll 1       // unrolling the group would drop the ring-shift moves (as in k_dw_fwd), but needs > 168 VGPRs: measured 266 -> 300 us at 2 waves/SIMD
            for (int j = 0; j < G; ++j) {
                const int iy = r0 + j;
                const int oy = iy - PAD;
                const bool row_in = iy >= 0 && iy < a.H;
                const bool orow_in = oy >= 0 && oy < a.H;
                uint32_t xraw[DW_BW];
                if constexpr (REDG) {                // row j of the group fetched one step ago; rotate the register rows
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) xraw[ox] = xq[0][ox];
#pragma unroll
                    for (int q = 0; q + 1 < G; ++q)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox) xq[q][ox] = xq[q + 1][ox];
                }
                // ---- dy row iy
                f2 xr[WIN_W];
                if (row_in) {
                    const size_t ro = (size_t)dw_slot<2 * G>(iy) * a.rc * 4 + coloff;
                    if constexpr (GA) {
                        const f2 ge = *(const f2*)(lds_ga + 2 * cp), gz = *(const f2*)(lds_ga + cblk + 2 * cp);
                        dw_read_dy<WIN_W, GM, GA>(ring_g + ro, ring_y + ro, ps, cf, colmask, xr, ge, gz);
                    } else {
                        dw_read_dy<WIN_W, GM, GA>(ring_g + ro, ring_y + ro, ps, cf, colmask, xr);
                    }
                } else {
#pragma unroll
                    for (int xx = 0; xx < WIN_W; ++xx) xr[xx] = zero2;
                }
                if (DG && row_in) {
#pragma unroll
                    for (int i = 0; i < KS; ++i)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                            for (int kx = 0; kx < KS; ++kx)
                                A[i][ox] = f2fma(wt[(KS - 1 - i) * KS + kx], xr[ox + kx], A[i][ox]);
                }
                if (WG) {
#pragma unroll
                    for (int q = KS - 1; q > 0; --q)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox) D[q][ox] = D[q - 1][ox];
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) D[0][ox] = xr[ox + PAD];
                }
                // ---- x row oy = iy - PAD: activation window (WG) and raw centre (RED)
                f2 xa[WIN_W];
                if (NEEDX && orow_in) {
                    const uint32_t* rowp = ring_x + (size_t)dw_slot<2 * G>(oy) * a.rc * 4 + coloff;
#if MNAS_DW_XFILL
                    if (WG) dw_read_act_nm<WIN_W>(rowp, ps, has_coef, cs, ct, xa);
#else
                    if (WG) dw_read_act<WIN_W>(rowp, ps, has_coef, cs, ct, colmask, xa);
#endif
                    if (RED && !REDG) {
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox) xraw[ox] = rowp[(ox + PAD) * ps];
                    }
                }
                // ---- emit gin row oy (+ fused BN-backward reduce of the producer of x)
                if (DG && orow_in && ch_ok) {
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) {
                        if (gx0 + ox < a.W) {
                            const uint32_t pk = pack_bf16(A[0][ox].x, A[0][ox].y);
                            st_u1(gin + obase + ((size_t)oy * a.W + ox) * a.C / 2, pk, a.nt);
                            if (RED) {
                                const f2 gq = f2bf(pk), yq = f2bf(xraw[ox]);
                                const f2 z = f2fma(yq, cs, ct);
                                f2 dz;
                                dz.x = (z.x > 0.f) ? gq.x : 0.f;
                                dz.y = (z.y > 0.f) ? gq.y : 0.f;
                                s1 += dz;
                                s2 = f2fma(dz, f2fma(yq, ris, rmu), s2);
                            }
                        }
                    }
                }
                if (DG) {
#pragma unroll
                    for (int i = 0; i + 1 < KS; ++i)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox) A[i][ox] = A[i + 1][ox];
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) A[KS - 1][ox] = zero2;
                }
                // ---- weight gradient: activation row oy against the dy ring (dy rows oy-ky+PAD = iy-ky = D[ky])
                if (WG && orow_in) {
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int ox = 0; ox < DW_BW; ++ox)
#pragma unroll
                            for (int kx = 0; kx < KS; ++kx)
                                wacc[ky * KS + kx] = f2fma(D[ky][ox], xa[ox + kx], wacc[ky * KS + kx]);
                }
            }
        }
    }
    }       // sweep waves
    const int row = blockIdx.x / a.cblocks, rows = a.geff / a.cblocks;
    const int cb0 = (blockIdx.x % a.cblocks) * cblk;
    const bool any = cur_c0 >= 0;
    if constexpr (RED) {      // fused-reduce table float[2][C][rows]
        const f2 sv[2] = {s1, s2};
        dw_block_reduce<2>((float*)ring_g, sv, cp, sxi, a.sx, cblk, active, [&](int r, int cl, float v) {
            const int c = cb0 + cl;
            if (c < a.C) red_partial[((size_t)r * a.C + c) * rows + row] = any ? v : 0.f;
        });
    }
    if constexpr (WG) {       // wpartial float[rows][k*k][C]
        dw_block_reduce<KS * KS>((float*)ring_g, wacc, cp, sxi, a.sx, cblk, active, [&](int k, int cl, float v) {
            const int c = cb0 + cl;
            if (c < a.C) wpartial[((size_t)row * KS * KS + k) * a.C + c] = any ? v : 0.f;
        });
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
// Launch forms: forward (1 ring, x); phase 1 = input gradient (+reduce) (2 rings: g, y); phase 0 = fused and phase 2 =
// weight gradient only (3 rings: g, y, x).  Rows per DMA group G = 4 (ring = 8 rows) except the 5x5 weight-gradient-only
// sweep, G = 2: its three rings are what limits the strip width, and with 4-row rings it gets 2 full-width strips on a
// 56-wide image where 8-row rings allow only 5 narrow ones (halo 1.33, 84 % of the lanes busy).
static int dw_rings(int form) { return form < 0 ? 1 : (form == 1 ? 2 : 3); }      // form: -1 forward, else phase
// Pick rows-per-group G in {4, 2} and the strip geometry for a launch form: 2-row groups halve the ring footprint (wider
// strips / more channels per workgroup) at the price of a barrier every 2 rows instead of 4 (priced at 7 %).
static bool dw_choose(int N, int H, int W, int C, int k, int form, DwArgs* a, int* g, int exp_kpad = 0, int src_cin = 0) {
    DwArgs a4, a2;
    const int nrings = dw_rings(form);
    const bool ok4 = dw_pick(N, H, W, C, k, nrings, 8, &a4, exp_kpad, src_cin), ok2 = dw_pick(N, H, W, C, k, nrings, 4, &a2, exp_kpad, src_cin);
    if (!ok4 && !ok2) return false;
    if (ok2 && (!ok4 || 0.93f * a2.score > a4.score)) { *a = a2; *g = 2; }
    else { *a = a4; *g = 4; }
    return true;
}
static bool dw_finish(DwArgs* a, int nparts) {
    if (nparts < a->cblocks) return false;
    int g = nparts < a->items ? nparts : a->items;
    a->geff = g / a->cblocks * a->cblocks;          // multiple of cblocks: item % cblocks is constant per workgroup
    return a->geff >= a->cblocks;
}
// Partial-table rows for a launch with `nparts`:  which = 0 forward statistics float[2][C][rows];  which = 1 both tables of
// a phase-0 launch (wpartial float[rows][k*k][C], reduce float[2][C][rows]);  which = 2 the reduce table of a phase-1
// launch;  which = 3 wpartial of a phase-2 launch.
static const int dw_form_of[4] = {-1, 0, 1, 2};
extern "C" int mnas_dw_rows(int N, int H, int W, int C, int k, int nparts, int which) {
    if (which < 0 || which > 3) return -1;
    DwArgs a;
    int g;
    if (!dw_choose(N, H, W, C, k, dw_form_of[which], &a, &g) || !dw_finish(&a, nparts)) return -1;
    return a.geff / a.cblocks;
}

// Geometry chosen for a launch form (diagnostics / DESIGN.md tables): out = {channel pairs per workgroup, 4-column strips
// per workgroup, threads, strips per image row, channel blocks, LDS bytes, rows per DMA group}
extern "C" int mnas_dw_geometry(int N, int H, int W, int C, int k, int which, int* out) {
    if (which < 0 || which > 3 || !out) return MNAS_EINVAL;
    DwArgs a;
    int g;
    if (!dw_choose(N, H, W, C, k, dw_form_of[which], &a, &g)) return MNAS_EINVAL;
    out[0] = a.cpw; out[1] = a.sx; out[2] = a.nthreads; out[3] = a.strips_x; out[4] = a.cblocks;
    out[5] = dw_rings(dw_form_of[which]) * 2 * g * a.rc * 16; out[6] = g;
    return MNAS_OK;
}

extern "C" int mnas_dw_fwd(const MnasDwFwd* c, void* stream) {
    if (!c || (c->k != 3 && c->k != 5) || (c->C & 7) || c->nparts < 1) return MNAS_EINVAL;
    DwArgs a;
    int g;
    if (!dw_choose(c->N, c->H, c->W, c->C, c->k, -1, &a, &g) || !dw_finish(&a, c->nparts)) return MNAS_EINVAL;
    a.nt = (mnas_nt_mask() & MNAS_NT_DW_FWD) ? 1 : 0;
    size_t lds = (size_t)2 * g * a.rc * 16;
    const size_t red_need = (size_t)a.sx * 2 * 2 * a.cpw * sizeof(float);          // dw_block_reduce scratch
    if (lds < red_need) lds = red_need;
    hipStream_t s = (hipStream_t)stream;
#define MNAS_DWF(K_, G_) hipLaunchKernelGGL((k_dw_fwd<K_, G_>), dim3(a.geff), dim3(a.nthreads), lds, s, a, c->in, c->w, c->bias, \
                                            (uint32_t*)c->out, c->stats)
    if (c->k == 3) { if (g == 4) MNAS_DWF(3, 4); else MNAS_DWF(3, 2); }
    else { if (g == 4) MNAS_DWF(5, 4); else MNAS_DWF(5, 2); }
#undef MNAS_DWF
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// Fused expand (1x1, Cin -> C) + depthwise forward.  Geometry differs from the plain forward (LDS also holds the expand
// weights): rows of the statistics table = mnas_dw_exp_rows(...).  Supported: Cin <= 96 (three MFMA k-steps); returns
// MNAS_EINVAL otherwise (callers fall back to mnas_conv_gemm + mnas_dw_fwd).
extern "C" int mnas_dw_exp_rows(int N, int H, int W, int C, int k, int Cin, int nparts) {
    if (Cin < 8 || (Cin & 7) || Cin > 96) return -1;
    DwArgs a;
    int g;
    if (!dw_choose(N, H, W, C, k, -1, &a, &g, (Cin + 31) / 32 * 32) || !dw_finish(&a, nparts)) return -1;
    return a.geff / a.cblocks;
}

extern "C" int mnas_dw_exp_fwd(const MnasDwExpFwd* c, void* stream) {
    if (!c || (c->k != 3 && c->k != 5) || (c->C & 7) || c->nparts < 1 || c->Cin < 8 || (c->Cin & 7) || c->Cin > 96) return MNAS_EINVAL;
    if (!c->x.data || !c->w1 || !c->w || !c->out) return MNAS_EINVAL;
    DwArgs a;
    int g;
    const int kpad = (c->Cin + 31) / 32 * 32;
    if (!dw_choose(c->N, c->H, c->W, c->C, c->k, -1, &a, &g, kpad) || !dw_finish(&a, c->nparts)) return MNAS_EINVAL;
    a.nt = (mnas_nt_mask() & MNAS_NT_DW_FWD) ? 1 : 0;
    DwExp e;
    e.x = (const uint16_t*)c->x.data; e.xs = c->x.scale; e.xt = c->x.shift;
    e.w1 = (const uint16_t*)c->w1; e.b1 = c->b1; e.y1 = (uint32_t*)c->y1; e.Cin = c->Cin; e.Kpad = kpad;
    MnasActIn in = {nullptr, c->bn1_scale, c->bn1_shift};
    size_t lds = (size_t)2 * g * a.rc * 16 + dw_exp_lds(a.cpw, kpad);
    const size_t red_need = (size_t)a.sx * 2 * 2 * a.cpw * sizeof(float);
    if (lds < red_need) lds = red_need;
    hipStream_t s = (hipStream_t)stream;
    const int kst = kpad / 32;
#define MNAS_DWE(K_, G_, T_) hipLaunchKernelGGL((k_dw_fwd_exp<K_, G_, T_>), dim3(a.geff), dim3(a.nthreads), lds, s, a, e, in, c->w, \
                                               c->bias, (uint32_t*)c->out, c->stats)
#define MNAS_DWE_T(K_, G_) do { if (kst == 1) MNAS_DWE(K_, G_, 1); else if (kst == 2) MNAS_DWE(K_, G_, 2); else MNAS_DWE(K_, G_, 3); } while (0)
    if (c->k == 3) { if (g == 4) MNAS_DWE_T(3, 4); else MNAS_DWE_T(3, 2); }
    else { if (g == 4) MNAS_DWE_T(5, 4); else MNAS_DWE_T(5, 2); }
#undef MNAS_DWE_T
#undef MNAS_DWE
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// SRC form: geometry / partial-table rows (both tables of the launch, like which = 1 of mnas_dw_rows)
static bool dw_src_ok(int C, int cin) { return cin >= 8 && !(cin & 7) && cin <= 32 && !(C & 7); }
extern "C" int mnas_dw_src_rows(int N, int H, int W, int C, int k, int cin, int nparts) {
    if (!dw_src_ok(C, cin) || (k != 3 && k != 5)) return -1;
    DwArgs a;
    int g;
    if (!dw_choose(N, H, W, C, k, 0, &a, &g, 32, cin) || !dw_finish(&a, nparts)) return -1;
    const int ntb = (2 * a.cpw + 15) / 16;
    if (ntb != 3 && ntb != 5) return -1;
    return a.geff / a.cblocks;
}
static int dw_bwd_src(const MnasDwBwd* c, hipStream_t s) {
    if (c->phase != 0 || !c->red_bn || !c->red_partial || !dw_src_ok(c->C, c->src_cin)) return MNAS_EINVAL;
    if (!c->src_x.data || !c->src_w1 || !c->src_w3t || !c->dy.y || !c->dy.coef || !c->w || !c->gin || !c->wpartial) return MNAS_EINVAL;
    DwArgs a;
    int g;
    if (!dw_choose(c->N, c->H, c->W, c->C, c->k, 0, &a, &g, 32, c->src_cin) || !dw_finish(&a, c->nparts)) return MNAS_EINVAL;
    a.nt = (mnas_nt_mask() & MNAS_NT_DW_BWD) ? 1 : 0;
    DwSrc e;
    e.x = (const uint16_t*)c->src_x.data; e.xs = c->src_x.scale; e.xt = c->src_x.shift;
    e.w1 = (const uint16_t*)c->src_w1; e.b1 = c->src_b1; e.dy = (const uint16_t*)c->src_dy; e.w3t = (const uint16_t*)c->src_w3t;
    e.Cin = c->src_cin; e.Kpad = 32;
    size_t lds = (size_t)3 * 2 * g * a.rc * 16 + dw_src_lds(a.cpw, 32, 2 * g, a.iw, c->src_cin);
    const size_t red_need = (size_t)a.sx * c->k * c->k * 2 * a.cpw * sizeof(float);
    if (lds < red_need) lds = red_need;
    const int ntb = (2 * a.cpw + 15) / 16;
    if (ntb != 3 && ntb != 5) return MNAS_EINVAL;          // 48- and 72-channel blocks (the 112x112 / 56x56 stages)
#define MNAS_DWS(K_, G_, T_) hipLaunchKernelGGL((k_dw_bwd<K_, true, true, true, G_, true, T_>), dim3(a.geff), dim3(a.nthreads + 64), lds, s, a, \
                                                c->x, c->dy, c->w, (uint32_t*)c->gin, c->wpartial, c->red_partial, c->red_bn, e)
#define MNAS_DWS_T(K_, G_) do { if (ntb == 3) MNAS_DWS(K_, G_, 3); else MNAS_DWS(K_, G_, 5); } while (0)
    if (c->k == 3) { if (g == 4) MNAS_DWS_T(3, 4); else MNAS_DWS_T(3, 2); }
    else { if (g == 4) MNAS_DWS_T(5, 4); else MNAS_DWS_T(5, 2); }
#undef MNAS_DWS_T
#undef MNAS_DWS
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_dw_bwd(const MnasDwBwd* c, void* stream) {
    if (!c || (c->k != 3 && c->k != 5) || (c->C & 7) || c->nparts < 1 || c->phase < 0 || c->phase > 2) return MNAS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (c->src_dy) return dw_bwd_src(c, s);
    if (!c->x.data || !c->dy.g) return MNAS_EINVAL;
    const bool red = c->red_bn != nullptr && c->red_partial != nullptr;
    // phase 0: everything in one fused sweep.  phase 1: input gradient (+reduce).  phase 2: weight gradient.
    const bool want_dg = c->phase != 2, want_wg = c->phase != 1;
    const int nrings = dw_rings(c->phase);
    DwArgs a;
    int g;
    if (!dw_choose(c->N, c->H, c->W, c->C, c->k, c->phase, &a, &g) || !dw_finish(&a, c->nparts)) return MNAS_EINVAL;
    a.nt = (mnas_nt_mask() & MNAS_NT_DW_BWD) ? 1 : 0;
    size_t lds = (size_t)nrings * 2 * g * a.rc * 16;
    const size_t red_need = (size_t)a.sx * (want_wg ? c->k * c->k : 2) * 2 * a.cpw * sizeof(float);   // dw_block_reduce scratch
    if (lds < red_need) lds = red_need;
#define MNAS_DWB(K_, DG_, WG_, R_, G_) hipLaunchKernelGGL((k_dw_bwd<K_, DG_, WG_, R_, G_>), dim3(a.geff), dim3(a.nthreads), lds, s, a, \
                                                         c->x, c->dy, c->w, (uint32_t*)c->gin, c->wpartial, c->red_partial, c->red_bn, DwSrc{})
#define MNAS_DWB_G(K_, DG_, WG_, R_) do { if (g == 4) MNAS_DWB(K_, DG_, WG_, R_, 4); else MNAS_DWB(K_, DG_, WG_, R_, 2); } while (0)
    if (c->g_gate) {                         // g read as g*g_gate[n][c] + g_bias[n][c]: fused sweep only
        if (!(want_dg && want_wg && red) || c->g_masked || !c->g_bias) return MNAS_EINVAL;
        lds = (size_t)nrings * 2 * g * a.rc * 16 + (size_t)4 * a.cpw * sizeof(float);      // + the per-item (gate, bias) table
        if (lds < red_need) lds = red_need;
#define MNAS_DWA(K_, G_) hipLaunchKernelGGL((k_dw_bwd<K_, true, true, true, G_, false, 1, false, true>), dim3(a.geff), dim3(a.nthreads), lds, s, a, \
                                            c->x, c->dy, c->w, (uint32_t*)c->gin, c->wpartial, c->red_partial, c->red_bn, DwSrc{}, c->g_gate, c->g_bias)
        if (c->k == 3) { if (g == 4) MNAS_DWA(3, 4); else MNAS_DWA(3, 2); }
        else { if (g == 4) MNAS_DWA(5, 4); else MNAS_DWA(5, 2); }
#undef MNAS_DWA
        MNAS_CHECK_LAUNCH();
        return MNAS_OK;
    }
    if (c->g_masked) {                       // fused sweep only (what the engine runs behind a project conv's masked gradient)
        if (!(want_dg && want_wg && red)) return MNAS_EINVAL;
#define MNAS_DWM(K_, G_) hipLaunchKernelGGL((k_dw_bwd<K_, true, true, true, G_, false, 1, true>), dim3(a.geff), dim3(a.nthreads), lds, s, a, \
                                            c->x, c->dy, c->w, (uint32_t*)c->gin, c->wpartial, c->red_partial, c->red_bn, DwSrc{})
        if (c->k == 3) { if (g == 4) MNAS_DWM(3, 4); else MNAS_DWM(3, 2); }
        else { if (g == 4) MNAS_DWM(5, 4); else MNAS_DWM(5, 2); }
#undef MNAS_DWM
        MNAS_CHECK_LAUNCH();
        return MNAS_OK;
    }
    if (c->k == 3) {
        if (want_dg && want_wg) { if (red) MNAS_DWB_G(3, true, true, true); else MNAS_DWB_G(3, true, true, false); }
        else if (want_dg) { if (red) MNAS_DWB_G(3, true, false, true); else MNAS_DWB_G(3, true, false, false); }
        else MNAS_DWB_G(3, false, true, false);
    } else {
        if (want_dg && want_wg) { if (red) MNAS_DWB_G(5, true, true, true); else MNAS_DWB_G(5, true, true, false); }
        else if (want_dg) { if (red) MNAS_DWB_G(5, true, false, true); else MNAS_DWB_G(5, true, false, false); }
        else MNAS_DWB_G(5, false, true, false);
    }
#undef MNAS_DWB_G
#undef MNAS_DWB
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
