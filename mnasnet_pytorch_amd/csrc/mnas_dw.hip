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

// DIAGNOSIS builds only (tools/build_alt.sh mnas_dw.hip -D...): the sweep without its DMA (compute on whatever the LDS holds) and
// without its arithmetic (ring fills and barriers only, nothing stored) -- the compute-only / memory-only decomposition of DESIGN.md
#ifndef MNAS_DW_NODMA
#define MNAS_DW_NODMA 0
#endif
#ifndef MNAS_DW_NOCOMP
#define MNAS_DW_NOCOMP 0
#endif
#ifndef MNAS_DW_NOBARRIER
#define MNAS_DW_NOBARRIER 0      // 1: the per-step workgroup barrier is dropped (WRONG results: an upper bound of what wave-private rings could gain)
#endif
__device__ __forceinline__ void dw_step_barrier() {
    if (MNAS_DW_NOBARRIER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else dma_barrier();
}

#ifndef MNAS_DW_XCD
#define MNAS_DW_XCD 1        // 1: consecutive items (the channel blocks / neighbouring strips of one image: shared lines, shared halos) run
#endif                       //    on ONE XCD, i.e. behind one L2 (hardware places workgroup b on XCD b % 8); 0: A/B builds
#ifndef MNAS_DW_RA
#define MNAS_DW_RA 1         // 1: the window of row j+1 is read from LDS before row j is computed (0: A/B builds)
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

static bool dw_pick(int N, int H, int W, int C, int k, int nrings, int rr, DwArgs* a) {
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
            // two workgroups per CU either way (160 KB LDS): wide strips (78 KB) measured 8-10 % faster than 60 KB for
            // every launch form except the 5x5 weight-gradient sweep (3 rings), which is 14 % slower with them
            const size_t cap = (k == 5 && nrings == 3) ? 60 * 1024 : 78 * 1024;
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
    if (MNAS_DW_NODMA) return;
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
// The two halves of dw_read_act_nm for the software-pipelined sweeps: the RAW window of row j+1 is read from LDS before the
// arithmetic of row j starts (the row body used to open with WIN_W ds_reads and a full lgkmcnt(0) -- at 3 waves per SIMD the LDS
// latency of every row was exposed), the activation is applied when the row is consumed.
template <int WIN_W>
__device__ __forceinline__ void dw_ld_raw(const uint32_t* rowp, int ps, uint32_t (&r)[WIN_W]) {
#pragma unroll
    for (int xx = 0; xx < WIN_W; ++xx) r[xx] = rowp[xx * ps];
}
template <int WIN_W>
__device__ __forceinline__ void dw_act_raw(const uint32_t (&r)[WIN_W], bool has_coef, f2 s, f2 t, f2 (&xr)[WIN_W]) {
#pragma unroll
    for (int xx = 0; xx < WIN_W; ++xx) {
        f2 v = f2bf(r[xx]);
        if (has_coef) {
            v = f2fma(v, s, t);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
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
template <int WIN_W, bool GM = false>
__device__ __forceinline__ void dw_read_dy(const uint32_t* growp, const uint32_t* yrowp, int ps, const f2 (&cf)[5],
                                           unsigned colmask, f2 (&xr)[WIN_W]) {
#pragma unroll
    for (int xx = 0; xx < WIN_W; ++xx) {
        const f2 g = f2bf(growp[xx * ps]), y = f2bf(yrowp[xx * ps]);
        f2 d;
        f2 dz = g;
        if constexpr (!GM) {
            const f2 z = f2fma(y, cf[0], cf[1]);
            dz.x = (z.x > 0.f) ? g.x : 0.f;
            dz.y = (z.y > 0.f) ? g.y : 0.f;
        }
        d = f2fma(cf[2], dz, f2fma(cf[3], y, cf[4]));
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
// HC ("has coefficients", round 6): whether the input is a virtual activation is a template parameter -- as a run-time flag
// the window read computed both forms and selected (16 v_cndmask per row on top of the 8 packed FMAs + 16 v_max of the
// activation itself: 9 % of the 5x5 row body's ~185 vector instructions)
template <int KS, int G, bool HC>
__global__ __launch_bounds__(256, (KS == 3 ? 4 : 3)) void k_dw_fwd(DwArgs a, MnasActIn in, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, uint32_t* __restrict__ out,
                                                                   float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PAD = KS / 2, WIN_W = DW_BW + KS - 1, RR = 2 * G;
    const int cblk = 2 * a.cpw;
    uint32_t* ring = (uint32_t*)smem;                    // [RR][rc*4 dwords]; reused as reduction scratch at the end
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
    const int cp = tid % a.cpw, sxi = tid / a.cpw;
    const bool active = sxi < a.sx;
    constexpr bool has_coef = HC;
    const f2 zero2 = {0.f, 0.f};
    f2 s1 = zero2, s2 = zero2;
    int cur_c0 = -1;
    f2 wt[KS * KS], b2 = zero2, cs = {1.f, 1.f}, ct = zero2;
    const int nsteps = (a.H + 2 * PAD + G - 1) / G;
    const int ps = a.cpw;

    const int bid = MNAS_DW_XCD ? (int)xcd_remap(blockIdx.x, a.geff) : (int)blockIdx.x;     // logical workgroup id
    for (int item = bid; item < a.items; item += a.geff) {
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
        __syncthreads();                             // previous item's last group consumed (nothing is in flight here)
        dw_fill_edges<RR>(a, plan, ring, wave, nwaves, lane, has_coef);
        dw_dma_rows<KS, G>(a, plan, ring, (const uint4*)in.data, n, -PAD, x0, c0, wave, nwaves);
        for (int s = 0; s < nsteps; ++s) {
            const int r0 = -PAD + s * G;
            dw_step_barrier();                       // group s landed (every wave's own DMA) + readers of group s-1 retired
            if (s + 1 < nsteps) dw_dma_rows<KS, G>(a, plan, ring, (const uint4*)in.data, n, r0 + G, x0, c0, wave, nwaves);
            if (!active || MNAS_DW_NOCOMP) continue;
            // One image row: scatter its window into the register ring, emit the completed output row, shift the ring.
            auto row = [&](int iy, const f2 (&xr)[WIN_W], bool have) {
                const int oy = iy - PAD;             // A[0] is complete after this row
                if (have) {                          // uniform: rows outside the image contribute nothing
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
            };
            // rows in PAIRS (3x3: the whole group unrolled, 5x5: one pair per trip -- more spills past 168 VGPRs): the register-
            // ring shift of consecutive rows becomes renaming instead of KS*DW_BW 64-bit moves per row (3x3 at 112x112: 146 ->
            // 132 us), and the raw window of the next row is in flight while a row is computed (MNAS_DW_RA; reads of rows outside
            // the image land on some ring row and are dropped)
            const auto rowp = [&](int iy) { return colp + (size_t)dw_slot<RR>(iy) * a.rc * 4; };
            constexpr bool RA = MNAS_DW_RA && KS == 3;            // 5x5: the second raw window spills (168 VGPRs = 3 waves per SIMD)
            constexpr int STEP = (KS == 5 && G == 2) ? 1 : 2;     // 5x5 with 2-row groups: single rows (the pair spills as well)
            uint32_t wa[WIN_W], wb[WIN_W];
            if (RA) dw_ld_raw<WIN_W>(rowp(r0), ps, wa);
#pragma unroll (KS == 3 ? G / 2 : 1)
            for (int j = 0; j < G; j += STEP) {
                const int iy = r0 + j;
                const bool in0 = iy >= 0 && iy < a.H, in1 = iy + 1 >= 0 && iy + 1 < a.H;
                f2 xr[WIN_W];
                if constexpr (RA) {
                    dw_ld_raw<WIN_W>(rowp(iy + 1), ps, wb);
                    dw_act_raw<WIN_W>(wa, has_coef, cs, ct, xr);
                    row(iy, xr, in0);
                    if (j + 2 < G) dw_ld_raw<WIN_W>(rowp(iy + 2), ps, wa);
                    dw_act_raw<WIN_W>(wb, has_coef, cs, ct, xr);
                    row(iy + 1, xr, in1);
                } else {
                    if (in0) dw_read_act_nm<WIN_W>(rowp(iy), ps, has_coef, cs, ct, xr);
                    row(iy, xr, in0);
                    if constexpr (STEP == 2) {
                        if (in1) dw_read_act_nm<WIN_W>(rowp(iy + 1), ps, has_coef, cs, ct, xr);
                        row(iy + 1, xr, in1);
                    }
                }
            }
        }
    }
    if (stats) {
        // stats table is float[2][C][rows], rows = geff / cblocks: workgroup b owns column b / cblocks of its own
        // channel block (b % cblocks), so every (channel, column) is written exactly once -- no zero fill, no atomics
        const int rows = a.geff / a.cblocks, col = bid / a.cblocks;
        const int cb0 = (bid % a.cblocks) * cblk;
        const f2 sv[2] = {s1, s2};
        const bool any = cur_c0 >= 0;
        dw_block_reduce<2>((float*)ring, sv, cp, sxi, a.sx, cblk, active, [&](int r, int cl, float v) {
            const int c = cb0 + cl;
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
template <int KS, bool DG, bool WG, bool RED, int G, bool GM = false, bool HC = true>
__global__ __launch_bounds__(256, ((KS == 5 && DG && (WG || RED)) ? 2 : 3)) void k_dw_bwd(
    DwArgs a, MnasActIn x, MnasGradIn d, const float* __restrict__ w, uint32_t* __restrict__ gin, float* __restrict__ wpartial,
    float* __restrict__ red_partial, const float* __restrict__ red_bn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PAD = KS / 2, WIN_W = DW_BW + KS - 1;
    constexpr bool NEEDX = WG;                 // x ring only when the weight gradient is computed here
    constexpr bool REDG = RED && !WG;          // input-gradient-only launch: the reduce reads raw x from global (2 rings)
    const int cblk = 2 * a.cpw;
    uint32_t* ring_g = (uint32_t*)smem;                      // rings; reused as reduction scratch at the end
    uint32_t* ring_y = ring_g + (size_t)(2 * G) * a.rc * 4;
    uint32_t* ring_x = ring_y + (size_t)(2 * G) * a.rc * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
    const int cp = tid % a.cpw, sxi = tid / a.cpw;
    const bool active = sxi < a.sx;
    constexpr bool has_coef = HC || !WG;       // only the activation window of the weight gradient depends on it (HC = false: raw x)
    const uint32_t* xglob = (const uint32_t*)x.data;
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

    const int bid = MNAS_DW_XCD ? (int)xcd_remap(blockIdx.x, a.geff) : (int)blockIdx.x;     // logical workgroup id
    for (int item = bid; item < a.items; item += a.geff) {
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
            if (x.scale != nullptr && ch_ok) { cs.x = x.scale[ch]; cs.y = x.scale[ch + 1]; ct.x = x.shift[ch]; ct.y = x.shift[ch + 1]; }
            if (RED && ch_ok) {
                ris.x = red_bn[6 * a.C + ch]; ris.y = red_bn[6 * a.C + ch + 1];
                rmu.x = -red_bn[5 * a.C + ch] * ris.x; rmu.y = -red_bn[5 * a.C + ch + 1] * ris.y;
            }
        }
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
            dw_dma_rows<KS, G>(a, plan, ring_g, (const uint4*)d.g, n, r0, x0, c0, wave, nwaves);
            dw_dma_rows<KS, G>(a, plan, ring_y, (const uint4*)d.y, n, r0, x0, c0, wave, nwaves);
            if (NEEDX) dw_dma_rows<KS, G>(a, plan, ring_x, (const uint4*)x.data, n, r0 - PAD, x0, c0, wave, nwaves);
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
#if MNAS_DW_XFILL
        if (NEEDX) dw_fill_edges<2 * G>(a, plan, ring_x, wave, nwaves, lane, has_coef);   // out-of-image columns of x act to 0
#endif
        dma_group(-PAD);
        load_xn(-PAD);
        for (int s = 0; s < nsteps; ++s) {
            const int r0 = -PAD + s * G;
            dw_step_barrier();                       // group s landed (every wave's own DMA) + readers of group s-1 retired
            if constexpr (REDG) {
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int ox = 0; ox < DW_BW; ++ox) xq[j][ox] = xn[j][ox];
            }
            if (s + 1 < nsteps) { dma_group(r0 + G); load_xn(r0 + G); }
            if (!active || MNAS_DW_NOCOMP) continue;
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
                    dw_read_dy<WIN_W, GM>(ring_g + ro, ring_y + ro, ps, cf, colmask, xr);
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
    const int row = bid / a.cblocks, rows = a.geff / a.cblocks;
    const int cb0 = (bid % a.cblocks) * cblk;
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
// DIAGNOSIS build only (tools/build_alt.sh): MNAS_DW_CPW / MNAS_DW_SX / MNAS_DW_G force the geometry (tools/kbench_dw.py sweeps)
static bool dw_force(int N, int H, int W, int C, int k, int nrings, DwArgs* a, int* g) {
    const int cpw = mnas_diag_env("MNAS_DW_CPW", 0), sx = mnas_diag_env("MNAS_DW_SX", 0), gg = mnas_diag_env("MNAS_DW_G", 0);
    if (cpw <= 0 || sx <= 0 || (gg != 2 && gg != 4) || (cpw & 3) || cpw * sx > 256 || cpw > C / 2) return false;
    const int cps = C / 2, cgn = cpw / 4, iw = sx * DW_BW + k - 1;
    if ((size_t)nrings * 2 * gg * iw * cpw * 4 > 160 * 1024) return false;
    a->N = N; a->H = H; a->W = W; a->C = C; a->score = 0.f;
    a->cpw = cpw; a->sx = sx; a->cgn = cgn;
    a->nthreads = ((sx * cpw + 63) / 64) * 64;
    a->iw = iw;
    a->strips_x = (W + sx * DW_BW - 1) / (sx * DW_BW);
    a->cblocks = (cps + cpw - 1) / cpw;
    a->items = N * a->strips_x * a->cblocks;
    a->rc = iw * cgn;
    a->nb = (a->rc + 63) / 64;
    if (a->nb > 2 * (a->nthreads / 64)) return false;
    *g = gg;
    return true;
}
static bool dw_choose(int N, int H, int W, int C, int k, int form, DwArgs* a, int* g) {
    DwArgs a4, a2;
    const int nrings = dw_rings(form);
    // (two ring buffers of G rows; three -- the DMA two groups ahead behind a counted vmcnt -- measured 1-18 % slower in round 6)
    // (4 output columns per thread; a 2-column forward form -- seven strips cover a 14-wide row exactly, 125 VGPRs = four waves per
    // SIMD for the 5x5 -- measured equal at 14x14x576 and 5-18 % slower everywhere else, round 6)
    if (mnas_diag_env("MNAS_DW_CPW", 0) > 0 && dw_force(N, H, W, C, k, nrings, a, g)) return true;
    const bool ok4 = dw_pick(N, H, W, C, k, nrings, 8, &a4), ok2 = dw_pick(N, H, W, C, k, nrings, 4, &a2);
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
    if (which == 4 || which == 7) return mnas_dw2_rows(N, H, W, C, k, nparts, which - 4);      // stride 2 (csrc/mnas_dw2.hip)
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
    if (!c || (c->k != 3 && c->k != 5) || (c->C & 7) || c->nparts < 1 || c->stride < 0 || c->stride > 2) return MNAS_EINVAL;
    if (c->stride == 2) return mnas_dw2_fwd(c, stream);
    DwArgs a;
    int g;
    if (!dw_choose(c->N, c->H, c->W, c->C, c->k, -1, &a, &g) || !dw_finish(&a, c->nparts)) return MNAS_EINVAL;
    a.nt = (mnas_nt_mask() & MNAS_NT_DW_FWD) ? 1 : 0;
    size_t lds = (size_t)2 * g * a.rc * 16;
    const size_t red_need = (size_t)a.sx * 2 * 2 * a.cpw * sizeof(float);          // dw_block_reduce scratch
    if (lds < red_need) lds = red_need;
    hipStream_t s = (hipStream_t)stream;
#define MNAS_DWF(K_, G_) do { if (c->in.scale) hipLaunchKernelGGL((k_dw_fwd<K_, G_, true>), dim3(a.geff), dim3(a.nthreads), lds, s, a, c->in, c->w, c->bias, \
                                            (uint32_t*)c->out, c->stats); \
                              else hipLaunchKernelGGL((k_dw_fwd<K_, G_, false>), dim3(a.geff), dim3(a.nthreads), lds, s, a, c->in, c->w, c->bias, \
                                            (uint32_t*)c->out, c->stats); } while (0)
    if (c->k == 3) { if (g == 4) MNAS_DWF(3, 4); else MNAS_DWF(3, 2); }
    else { if (g == 4) MNAS_DWF(5, 4); else MNAS_DWF(5, 2); }
#undef MNAS_DWF
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// HC = false (a materialised x: only direct callers of the C ABI; the network's depthwise inputs are virtual activations) exists
// for the forms that read the activation window (WG) only
template <int KS, bool DG, bool WG, bool RED, int G, bool GM>
static void dw_bwd_launch(const DwArgs& a, size_t lds, hipStream_t s, const MnasDwBwd* c) {
    if constexpr (WG) {
        if (!c->x.scale) {
            hipLaunchKernelGGL((k_dw_bwd<KS, DG, WG, RED, G, GM, false>), dim3(a.geff), dim3(a.nthreads), lds, s, a, c->x, c->dy, c->w,
                               (uint32_t*)c->gin, c->wpartial, c->red_partial, c->red_bn);
            return;
        }
    }
    hipLaunchKernelGGL((k_dw_bwd<KS, DG, WG, RED, G, GM, true>), dim3(a.geff), dim3(a.nthreads), lds, s, a, c->x, c->dy, c->w,
                       (uint32_t*)c->gin, c->wpartial, c->red_partial, c->red_bn);
}

extern "C" int mnas_dw_bwd(const MnasDwBwd* c, void* stream) {
    if (!c || (c->k != 3 && c->k != 5) || (c->C & 7) || c->nparts < 1 || c->phase < 0 || c->phase > 2) return MNAS_EINVAL;
    if (c->stride < 0 || c->stride > 2) return MNAS_EINVAL;
    if (c->stride == 2) return mnas_dw2_bwd(c, stream);
    hipStream_t s = (hipStream_t)stream;
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
#define MNAS_DWB(K_, DG_, WG_, R_, G_) dw_bwd_launch<K_, DG_, WG_, R_, G_, false>(a, lds, s, c)
#define MNAS_DWB_G(K_, DG_, WG_, R_) do { if (g == 4) MNAS_DWB(K_, DG_, WG_, R_, 4); else MNAS_DWB(K_, DG_, WG_, R_, 2); } while (0)
    if (c->g_masked) {                       // fused sweep only (what the engine runs behind a project conv's masked gradient)
        if (!(want_dg && want_wg && red)) return MNAS_EINVAL;
        if (c->k == 3) { if (g == 4) dw_bwd_launch<3, true, true, true, 4, true>(a, lds, s, c); else dw_bwd_launch<3, true, true, true, 2, true>(a, lds, s, c); }
        else { if (g == 4) dw_bwd_launch<5, true, true, true, 4, true>(a, lds, s, c); else dw_bwd_launch<5, true, true, true, 2, true>(a, lds, s, c); }
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
