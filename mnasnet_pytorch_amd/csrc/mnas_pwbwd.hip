// Fused backward of a 1x1 convolution for the large-pixel-count stages (112x112 / 56x56 / 28x28):
//     gin[pix][ci]  = sum_co dy[pix][co] * W[co][ci]  (+ residual gradient)        -- ATen conv2d input gradient
//     dW[co][ci]    = sum_pix dy[pix][co] * act(x)[pix][ci]                         -- ATen conv2d weight gradient
//     (sum dz, sum dz*xhat) of the ConvBlock whose activated output gin belongs to     -- batch-norm backward, pass 1
// in ONE sweep over the pixels.  Replaces the pair k_igemm<dgrad> + k_wgrad for ConvBlock(kernel_size=1)
// (autograd mirror of mnasnet.py:48-62).  Why: both kernels stream the same (g, y) pair -- the BatchNorm/ReLU backward
// "dy-on-load" operand -- and at these sizes both are HBM-bound; sharing the staged dy tile removes a third of their
// combined traffic (e.g. 112x112 expand: 2*48 + 16 + 16 + 16 channels per pixel instead of (2*48 + 16 + 16) + (2*48 + 16)).
//
// One workgroup (4 waves) walks 64*PT-pixel tiles persistently:
//   stage   dy tile [pix][co] (dy-on-load from g, y) and act(x) tile [pix][ci] (act-on-load), both pixel-major bf16 in LDS;
//   dgrad   MFMA A = W^T rows [ci16][co32] (LDS, resident for the whole kernel), B = dy rows [pix16][co32]
//           -> D[ci][pix]: a lane holds 4 consecutive ci of one pixel (lane-local epilogue, 8-byte stores);
//   wgrad   MFMA A = dy^T [co16][pix32], B = act(x)^T [ci16][pix32], both read with the LDS transpose read
//           (ds_read_b64_tr_b16) from the SAME pixel-major tiles; the (co16 x ci16) accumulator tiles are split
//           over the 4 waves (each wave owns whole tiles: no cross-wave reduction) and live in registers for the
//           whole kernel; written once per workgroup to wpartial[workgroup][co][ci] (mnas_wgrad_finalize sums them);
//   reduce  per tile: 16-lane shuffle tree, then one LDS slot per (wave, channel) -- fixed order, no atomics.
// Roofline: HBM.  MFMA work per tile is ~0.3 us against ~5 us of memory time.
//
// FORM (round 4; forms 1 "NOGIN" and 3 "RED4" lost their A/B and were removed in round 5 -- DESIGN_HISTORY.md):
//   2 "RECOMP" expand conv: dy-on-load's raw forward output y1 is not read: after the x tile is staged, y1 = bf16(W1 act(x) + b1)
//              is recomputed on the matrix cores (same MFMA, same k order as the forward -> the same bits) and dy is formed
//              in place over the raw g tile.  One more barrier per tile; a third of the launch's reads gone.
#include "mnas_common.h"
#ifndef MNAS_PWB_PF
#define MNAS_PWB_PF 1        // 0: no form prefetches (A/B builds)
#endif

typedef __attribute__((ext_vector_type(4))) short pw_s4_t;
typedef __attribute__((address_space(3))) pw_s4_t* pw_lds_s4_ptr;

struct PwBwdArgs {
    int M, Ci, Co, Kd;       // Kd = Co rounded up to 32 (dgrad reduction length, row length of the packed weights)
    MnasActIn x;
    MnasGradIn dy;
    const uint16_t* w;       // [round16(Ci)][Kd]
    const void* resid;
    void* gin;
    float* wpartial;
    float* red_partial;
    const void* red_y;
    const float* red_bn;
    int nt;                  // nontemporal gin stores
    const uint16_t* w_fwd;   // FORM 2: MNAS_PACK_FWD [round16(Co)][Kf]
    const float* b_fwd;      // FORM 2: [Co] or NULL
    int Kf;                  // FORM 2: Ci rounded up to 32
    int gin_masked;          // out-stage forms: store dz = gin*[s*x+t>0] (the fused reduce's mask) instead of gin
    int seg_px;              // > 0: segment mode (see the tile walk)
};

__device__ __forceinline__ bf16x8_t pw_tr_frag(const uint16_t* tile, int ld, int row0, int col0, int lane) {
    // rows row0 + (lane>>4)*8 + {0..7}, column col0 + (lane&15)   (see mnas_wgrad.hip)
    const int i = lane & 15, g = lane >> 4;
    const uint16_t* p = tile + (row0 + g * 8 + (i >> 2)) * ld + col0 + (i & 3) * 4;
    const pw_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s4_ptr)p);
    const pw_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s4_ptr)(p + 4 * ld));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// OS (out-stage; the narrowing convs, where gin is the WIDE tensor): the input-gradient tile goes through LDS (the act(x) tile
// is dead after the weight-gradient MFMAs) and leaves as 16-byte nontemporal stores of whole row segments -- the tile's
// pixels are contiguous in memory, so these are full-line writes -- with the fused reduce done on that copy-out path (its
// operand red_y loaded as 16-byte chunks before the MFMA phases).  The 8-byte-per-lane stores of the MFMA epilogue
// (partial lines) sustained 2-2.4 TB/s of writes; the same change took the widening forward convs from 2.4 to 4.7 TB/s.
template <int NTO, int NTI, int PT, bool OS, int FORM, bool PF>
__global__ __launch_bounds__(256, 2) void k_pw_bwd(PwBwdArgs a) {
    static_assert(FORM == 0 || (FORM == 2 && !OS), "RECOMP rides on the plain epilogue");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BP = 64 * PT;
    constexpr int COP = NTO * 16, CIP = NTI * 16;
    constexpr bool OWN_O = NTO >= NTI;                       // waves split the cout tiles (else the cin tiles)
    constexpr int NOWN = OWN_O ? (NTO + 3) / 4 : (NTI + 3) / 4;
    constexpr int NOTH = OWN_O ? NTI : NTO;
    constexpr int ND = (BP * (COP / 8) + 255) / 256;         // dy staging slots per thread (upper bound)
    // OS copy-out role: thread -> fixed 16-byte channel column oc8 of the tile rows orow0 + k*OROWS.  The out-stage forms STAGE x
    // with the same mapping: the thread that stages chunk (row, oc8) of x is the one that later copies out -- and reduces --
    // chunk (row, oc8) of the input gradient, whose reduce operand (the raw output of x's producer = x.data itself for a
    // project conv) it therefore already holds in a register.  Round 3 re-read that operand from global memory (red_y): the
    // project convs moved 2L + 2S + L bytes, a third of them the same tensor twice.
    constexpr int NCH8 = CIP / 8, TCOLS = (256 / NCH8) * NCH8, OROWS = TCOLS / NCH8, MAXR = OS ? (BP + OROWS - 1) / OROWS : 1;
    constexpr int NX = OS ? MAXR : (BP * (CIP / 8) + 255) / 256;         // x staging slots per thread
    const int ldd = a.Kd + 8, ldw = a.Kd + 8, lda = CIP + 8;
    float* lds_cd = (float*)smem;                            // [5][COP]
    float* lds_cx = lds_cd + 5 * COP;                        // [2][CIP]
    float* lds_rc = lds_cx + 2 * CIP;                        // [4][CIP]
    float* lds_st = lds_rc + 4 * CIP;                        // [4 waves][2][CIP]
    uint16_t* lds_w = (uint16_t*)(lds_st + 8 * CIP);         // [CIP][ldw]
    uint16_t* tile_d = lds_w + CIP * ldw;                    // [BP][ldd]
    uint16_t* tile_a = tile_d + BP * ldd;                    // [BP][lda]
    constexpr int KSF = (CIP + 31) / 32;                     // FORM 2: k-steps of the recompute GEMM
    const int ldf = a.Kf + 8;
    uint16_t* lds_wf = tile_a + BP * lda;                    // FORM 2: [COP][ldf] forward weights
    float* lds_bf = (float*)(lds_wf + (FORM == 2 ? COP * ldf : 0));      // FORM 2: [COP] forward bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const bool hasx = a.x.scale != nullptr;
    const bool do_red = a.red_partial != nullptr;
    // wide inputs (Ci > NTI*16) are cut into channel slices over grid.y: a workgroup stages the whole dy tile but only its
    // slice of x, and produces that slice of gin / dW / the reduce (dy is the narrow tensor for these layers)
    const int ci0 = blockIdx.y * CIP;
    const int cis = min(CIP, a.Ci - ci0);                    // valid channels of this slice (multiple of 8)

    // ---- staging plan (tile-invariant): slot -> (pixel in tile, 16-byte channel chunk)
    const int cwd = a.Co >> 3, cwa = cis >> 3;
    int pd[ND], cd8[ND], pa[NX], ca8[NX];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int q = tid + 256 * i;
        pd[i] = q / cwd; cd8[i] = q - pd[i] * cwd;
        if (pd[i] >= BP) pd[i] = -1;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        if constexpr (OS) {                                  // copy-out mapping (see above)
            pa[i] = (tid < TCOLS && (tid % NCH8) < cwa && tid / NCH8 + i * OROWS < BP) ? tid / NCH8 + i * OROWS : -1;
            ca8[i] = tid % NCH8;
        } else {
            const int q = tid + 256 * i;
            pa[i] = q / cwa; ca8[i] = q - pa[i] * cwa;
            if (pa[i] >= BP) pa[i] = -1;
        }
    }

    // Tile walk: workgroups stride over the 64*PT-pixel tiles of the whole tensor, or (seg_px > 0, "segment mode") workgroup b
    // owns the contiguous pixels [b*seg_px, (b+1)*seg_px) -- its weight-gradient partial is then the sum over a known pixel
    // range (a fraction of ONE image for the squeeze-excite project conv: csrc/mnas_se.hip k_se_proj_du)
    const int mend = a.seg_px ? min(a.M, ((int)blockIdx.x + 1) * a.seg_px) : a.M;
    const int tstep = a.seg_px ? BP : (int)gridDim.x * BP;
    // The tile's global loads (g, y of dy-on-load; the x chunks) are issued ONE TILE AHEAD (PF): right after the staging barrier of
    // tile t the loads of tile t+1 go out into a second register set and land under the MFMA phases and the epilogue / copy-out
    // of tile t, instead of standing exposed at the top of every tile.  Per-launch, round 5 (us without -> with): 576->96 72.6 ->
    // 65.9, 480->80 64 -> 58, 240->40 98 -> 89, 72->24 88 -> 81, 24->72 75 -> 70, 48->16 199 -> 172, 16->48 161 -> 148.  Not for
    // 40->240 (16 more uint4 per thread: 255 VGPRs + spills, 74 -> 80) and 32->16 (131 -> 140): launch_pw_bwd picks.
    // (the zero fills below are redundant for the results -- the staging zeroes what was not loaded by its own predicates -- but not
    // for the schedule: without them the registers are undefined on the not-loaded path and the compiler sinks the loads to their
    // use in the next tile's staging, i.e. un-does the prefetch: class 2.36 -> 2.67 ms, round 6; the same in k_wgrad_t)
    uint4 ng[ND], ny[ND], nx[NX];
    auto issue = [&](int t0) {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            ng[i] = make_uint4(0, 0, 0, 0); ny[i] = make_uint4(0, 0, 0, 0);
            if (pd[i] >= 0 && t0 + pd[i] < mend) {
                const size_t off = (size_t)(t0 + pd[i]) * a.Co + cd8[i] * 8;
                ng[i] = *(const uint4*)((const uint16_t*)a.dy.g + off);
                if constexpr (FORM != 2) ny[i] = *(const uint4*)((const uint16_t*)a.dy.y + off);
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            nx[i] = make_uint4(0, 0, 0, 0);
            if (pa[i] >= 0 && t0 + pa[i] < mend)
                nx[i] = *(const uint4*)((const uint16_t*)a.x.data + (size_t)(t0 + pa[i]) * a.Ci + ci0 + ca8[i] * 8);
        }
    };
    const int tfirst = a.seg_px ? (int)blockIdx.x * a.seg_px : (int)blockIdx.x * BP;
    if (PF && MNAS_EARLY) issue(tfirst);                     // ahead of the setup: the first tile's loads share its round trip
    // ---- one-time setup: coefficient tables, resident weights, zeroed tiles / statistics (every table / copy with four loads in
    // flight: as plain loops each iteration was a memory round trip of its own, ~10 in a row before the first tile)
    {
        const int kc8n = a.Kd >> 3;
        mnas_copy_items(CIP * kc8n, tid, 256,
                        [&](int q) { const int r = q / kc8n, kc8 = q - r * kc8n; return *(const uint4*)(a.w + (size_t)(ci0 + r) * a.Kd + kc8 * 8); },
                        [&](int q, const uint4& v) { const int r = q / kc8n, kc8 = q - r * kc8n; *(uint4*)(lds_w + r * ldw + kc8 * 8) = v; });
        if constexpr (FORM == 2) {
            const int kf8 = a.Kf >> 3, cop16 = (a.Co + 15) / 16 * 16;
            mnas_copy_items(COP * kf8, tid, 256,
                            [&](int q) { const int r = q / kf8, k8 = q - r * kf8;
                                         return r < cop16 ? *(const uint4*)(a.w_fwd + (size_t)r * a.Kf + k8 * 8) : make_uint4(0, 0, 0, 0); },
                            [&](int q, const uint4& v) { const int r = q / kf8, k8 = q - r * kf8; *(uint4*)(lds_wf + r * ldf + k8 * 8) = v; });
            mnas_fill_table(lds_bf, COP, tid, 256, [&](int i) { return (a.b_fwd && i < a.Co) ? a.b_fwd[i] : 0.f; });
        }
    }
    mnas_fill_table(lds_cd, 5 * COP, tid, 256, [&](int i) {
        const int r = i / COP, c = i % COP;
        return (c < a.Co) ? a.dy.coef[(size_t)r * a.Co + c] : 0.f;
    });
    mnas_fill_table(lds_cx, 2 * CIP, tid, 256, [&](int i) {
        const int r = i / CIP, c = i % CIP;
        return (hasx && c < cis) ? (r == 0 ? a.x.scale[ci0 + c] : a.x.shift[ci0 + c]) : 0.f;
    });
    mnas_fill_table(lds_rc, 4 * CIP, tid, 256, [&](int i) {
        const int r = i / CIP, c = i % CIP;
        return (do_red && c < cis) ? mnas_red_coef(a.red_bn, a.Ci, r, ci0 + c) : 0.f;
    });
    for (int i = tid; i < 8 * CIP; i += 256) lds_st[i] = 0.f;
    {
        // (the dy tile rows past the staged slots and the pad columns are read by the MFMA fragments: zero once.  The first tile's
        // staging writes come after the barrier at the top of the tile loop, i.e. after these)
        const int nz = (BP * ldd + BP * lda) / 8;            // both tiles are contiguous; row strides are multiples of 8
        for (int i = tid; i < nz; i += 256) ((uint4*)tile_d)[i] = make_uint4(0, 0, 0, 0);
    }

    f32x4_t acc_w[NOWN][NOTH];
#pragma unroll
    for (int i = 0; i < NOWN; ++i)
#pragma unroll
        for (int j = 0; j < NOTH; ++j) acc_w[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // BatchNorm-backward partial sums: per lane in registers across all tiles when there are few cin tiles (one shuffle tree
    // at the very end), per tile through LDS otherwise (15 cin tiles would need 120 registers)
    // OS copy-out role: thread -> fixed 16-byte channel column c8 of the out-stage rows orow0 + k*OROWS
    const int oc8 = tid % NCH8, orow0 = tid / NCH8;
    const bool ocol_ok = tid < TCOLS && oc8 * 8 < cis;
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { r1[j] = 0.f; r2[j] = 0.f; }
    constexpr bool REGSTAT = NTI <= 6;
    float rs1[REGSTAT ? NTI : 1][4], rs2[REGSTAT ? NTI : 1][4];
#pragma unroll
    for (int i = 0; i < (REGSTAT ? NTI : 1); ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { rs1[i][r] = 0.f; rs2[i][r] = 0.f; }

    if (PF && !MNAS_EARLY) issue(tfirst);
    for (int tile0 = tfirst; tile0 < mend; tile0 += tstep) {
        __syncthreads();                                     // previous tile's fragments consumed (first pass: setup visible)
        if (!PF) issue(tile0);
        uint4 vg[ND], vy[ND], vx[NX];
#pragma unroll
        for (int i = 0; i < ND; ++i) { vg[i] = ng[i]; vy[i] = ny[i]; }
#pragma unroll
        for (int i = 0; i < NX; ++i) vx[i] = nx[i];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            if (pd[i] < 0) continue;
            uint4 v = make_uint4(0, 0, 0, 0);
            if constexpr (FORM == 2) {
                if (tile0 + pd[i] < mend) v = vg[i];          // raw g: dy is formed after the recompute below
            } else if (tile0 + pd[i] < mend) {
                float cf[5][8];
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    *(float4*)&cf[r][0] = *(const float4*)(lds_cd + r * COP + cd8[i] * 8);
                    *(float4*)&cf[r][4] = *(const float4*)(lds_cd + r * COP + cd8[i] * 8 + 4);
                }
                float o[8];
                dy8(vg[i], vy[i], cf[0], cf[1], cf[2], cf[3], cf[4], o);
                v = pack8(o);
            }
            *(uint4*)(tile_d + pd[i] * ldd + cd8[i] * 8) = v;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (pa[i] < 0) continue;
            uint4 v = vx[i];
            if (tile0 + pa[i] >= mend) v = make_uint4(0, 0, 0, 0);
            else if (hasx) {
                float s[8], sh[8];
                *(float4*)&s[0] = *(const float4*)(lds_cx + ca8[i] * 8);
                *(float4*)&s[4] = *(const float4*)(lds_cx + ca8[i] * 8 + 4);
                *(float4*)&sh[0] = *(const float4*)(lds_cx + CIP + ca8[i] * 8);
                *(float4*)&sh[4] = *(const float4*)(lds_cx + CIP + ca8[i] * 8 + 4);
                v = act8(v, s, sh);
            }
            *(uint4*)(tile_a + pa[i] * lda + ca8[i] * 8) = v;
        }
        __syncthreads();
        if (PF) issue(tile0 + tstep);                        // past the end: every predicate false, nothing issued
        if constexpr (FORM == 2) {
            // y1 = bf16(W act(x) + b) for this wave's pixels (D[co][pix]: a lane holds 4 consecutive co of one pixel), then
            // dy = c1*(g*[s*y+t>0]) + c2*y + c3 in place over the raw g values of the same (pixel, 4 channels)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int p = (wave * PT + pt) * 16 + l15;
                bf16x8_t xb[KSF];
#pragma unroll
                for (int ks = 0; ks < KSF; ++ks) {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (ks * 32 + lg * 8 < CIP) v = *(const uint4*)(tile_a + p * lda + ks * 32 + lg * 8);
                    xb[ks] = *(const bf16x8_t*)&v;
                }
                const bool pok = tile0 + p < mend;              // rows past the end stay zero (they feed the weight gradient)
#pragma unroll
                for (int to = 0; to < NTO; ++to) {
                    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KSF; ++ks) {
                        const bf16x8_t afrag = *(const bf16x8_t*)(lds_wf + (to * 16 + l15) * ldf + ks * 32 + lg * 8);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, xb[ks], acc, 0, 0, 0);
                    }
                    const int co = to * 16 + lg * 4;
                    const float4 bb = *(const float4*)(lds_bf + co);
                    uint2 yp;
                    yp.x = pack_bf16(acc[0] + bb.x, acc[1] + bb.y);
                    yp.y = pack_bf16(acc[2] + bb.z, acc[3] + bb.w);
                    const uint2 gp = *(const uint2*)(tile_d + p * ldd + co);
                    const float4 cs = *(const float4*)(lds_cd + co), ct = *(const float4*)(lds_cd + COP + co);
                    const float4 c1 = *(const float4*)(lds_cd + 2 * COP + co), c2 = *(const float4*)(lds_cd + 3 * COP + co);
                    const float4 c3 = *(const float4*)(lds_cd + 4 * COP + co);
                    const uint32_t yu[2] = {yp.x, yp.y}, gu[2] = {gp.x, gp.y};
                    const mnas_f2 s2[2] = {{cs.x, cs.y}, {cs.z, cs.w}}, t2[2] = {{ct.x, ct.y}, {ct.z, ct.w}};
                    const mnas_f2 a2[2] = {{c1.x, c1.y}, {c1.z, c1.w}}, b2[2] = {{c2.x, c2.y}, {c2.z, c2.w}}, e2[2] = {{c3.x, c3.y}, {c3.z, c3.w}};
                    uint32_t dp[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {            // channel pairs in float2 (v_pk_fma_f32)
                        const mnas_f2 yq = mnas_bf2(yu[h]), gq = mnas_bf2(gu[h]);
                        const mnas_f2 z = mnas_f2fma(yq, s2[h], t2[h]);
                        mnas_f2 dz;
                        dz.x = (z.x > 0.f) ? gq.x : 0.f;
                        dz.y = (z.y > 0.f) ? gq.y : 0.f;
                        const mnas_f2 d = mnas_f2fma(a2[h], dz, mnas_f2fma(b2[h], yq, e2[h]));
                        dp[h] = pack_bf16(d.x, d.y);
                    }
                    if (pok) *(uint2*)(tile_d + p * ldd + co) = make_uint2(dp[0], dp[1]);
                }
            }
            __syncthreads();
        }

        // the epilogue's global operands (raw output of the reduce target, residual gradient) for this lane's fragments:
        // issued now, they land under the MFMA phases
        uint2 ypre[PT][NTI], rpre[PT][NTI];
        // (out-stage forms: the reduce operand of copy-out slot k is vx[k], the raw x chunk this thread staged)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int m = tile0 + (wave * PT + pt) * 16 + l15;
#pragma unroll
            for (int nt = 0; nt < NTI; ++nt) {
                const int ci = nt * 16 + lg * 4;
                ypre[pt][nt] = make_uint2(0, 0); rpre[pt][nt] = make_uint2(0, 0);
                if (m < mend && ci < cis) {
                    const size_t o = (size_t)m * a.Ci + ci0 + ci;
                    if (!OS && do_red) ypre[pt][nt] = *(const uint2*)((const uint16_t*)a.red_y + o);
                    if (!OS && a.resid) rpre[pt][nt] = *(const uint2*)((const uint16_t*)a.resid + o);
                }
            }
        }
        // ---- input gradient: D[ci][pix] = sum_co W^T[ci][co] * dy[pix][co]
        f32x4_t acc_g[PT][NTI];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int nt = 0; nt < NTI; ++nt) acc_g[pt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int ksteps = a.Kd >> 5;
        for (int ks = 0; ks < ksteps; ++ks) {
            bf16x8_t bfrag[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                bfrag[pt] = *(const bf16x8_t*)(tile_d + ((wave * PT + pt) * 16 + l15) * ldd + ks * 32 + lg * 8);
#pragma unroll
            for (int nt = 0; nt < NTI; ++nt) {
                const bf16x8_t afrag = *(const bf16x8_t*)(lds_w + (nt * 16 + l15) * ldw + ks * 32 + lg * 8);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc_g[pt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[pt], acc_g[pt][nt], 0, 0, 0);
            }
        }
        // ---- weight gradient: D[co][ci] += sum_pix dy^T[co][pix] * act(x)^T[ci][pix]; this wave's tiles only
#pragma unroll
        for (int ks = 0; ks < BP / 32; ++ks) {
            bf16x8_t oth[NOTH];
#pragma unroll
            for (int j = 0; j < NOTH; ++j)
                oth[j] = OWN_O ? pw_tr_frag(tile_a, lda, ks * 32, j * 16, lane) : pw_tr_frag(tile_d, ldd, ks * 32, j * 16, lane);
#pragma unroll
            for (int i = 0; i < NOWN; ++i) {
                const int own = wave + 4 * i;
                if (own >= (OWN_O ? NTO : NTI)) continue;        // uniform per wave
                const bf16x8_t mine = OWN_O ? pw_tr_frag(tile_d, ldd, ks * 32, own * 16, lane)
                                            : pw_tr_frag(tile_a, lda, ks * 32, own * 16, lane);
#pragma unroll
                for (int j = 0; j < NOTH; ++j)
                    acc_w[i][j] = OWN_O ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine, oth[j], acc_w[i][j], 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_16x16x32_bf16(oth[j], mine, acc_w[i][j], 0, 0, 0);
            }
        }
        if constexpr (OS) {
            // ---- out-stage: bf16 tile [pixel][ci] over the dead act(x) tile, then whole row segments leave as 16-byte stores
            __syncthreads();                                 // every wave is done reading tile_a / tile_d fragments
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int p = (wave * PT + pt) * 16 + l15;
#pragma unroll
                for (int nt = 0; nt < NTI; ++nt) {
                    const float v[4] = {acc_g[pt][nt][0], acc_g[pt][nt][1], acc_g[pt][nt][2], acc_g[pt][nt][3]};   // (no residual in the out-stage forms)
                    uint2 pk;
                    pk.x = pack_bf16(v[0], v[1]);
                    pk.y = pack_bf16(v[2], v[3]);
                    *(uint2*)(tile_a + p * lda + nt * 16 + lg * 4) = pk;
                }
            }
            __syncthreads();
            // the reduce coefficients of this thread's channel column: fetched from LDS per tile, so that they are not live
            // across the MFMA phases (32 registers: the out-stage forms sat at 175-250 VGPRs = two workgroups per CU)
            float rcs[8], rct[8], rci[8], rcm[8];
            if (do_red) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    *(float4*)&rcs[4 * h] = *(const float4*)(lds_rc + oc8 * 8 + 4 * h);
                    *(float4*)&rct[4 * h] = *(const float4*)(lds_rc + CIP + oc8 * 8 + 4 * h);
                    *(float4*)&rci[4 * h] = *(const float4*)(lds_rc + 2 * CIP + oc8 * 8 + 4 * h);
                    *(float4*)&rcm[4 * h] = *(const float4*)(lds_rc + 3 * CIP + oc8 * 8 + 4 * h);
                }
            }
#pragma unroll
            for (int k = 0; k < MAXR; ++k) {
                const int p = orow0 + k * OROWS;
                if (!(ocol_ok && p < BP && tile0 + p < mend)) continue;
                const uint4 pk = *(const uint4*)(tile_a + p * lda + oc8 * 8);
                uint16_t* gdst = (uint16_t*)a.gin + ((size_t)(tile0 + p) * a.Ci + ci0 + oc8 * 8);
                if (!a.gin_masked) st_u4(gdst, pk, true);
                if (do_red) {                                // channel pairs in float2 (v_pk_fma_f32); same operations as the scalar form
                    const uint32_t gu[4] = {pk.x, pk.y, pk.z, pk.w}, yu[4] = {vx[k].x, vx[k].y, vx[k].z, vx[k].w};
                    uint32_t mz[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const mnas_f2 gq = mnas_bf2(gu[j]), yq = mnas_bf2(yu[j]);
                        const mnas_f2 z = mnas_f2fma(yq, mnas_ld2(rcs + 2 * j), mnas_ld2(rct + 2 * j));
                        mnas_f2 dz;
                        dz.x = (z.x > 0.f) ? gq.x : 0.f;
                        dz.y = (z.y > 0.f) ? gq.y : 0.f;
                        mz[j] = (__float_as_uint(dz.x) >> 16) | (__float_as_uint(dz.y) & 0xffff0000u);   // dz is g or 0: exact in bf16
                        const mnas_f2 xh = mnas_f2fma(yq, mnas_ld2(rci + 2 * j), mnas_ld2(rcm + 2 * j));
                        const mnas_f2 a1 = mnas_ld2(r1 + 2 * j) + dz, a2 = mnas_f2fma(dz, xh, mnas_ld2(r2 + 2 * j));
                        r1[2 * j] = a1.x; r1[2 * j + 1] = a1.y;
                        r2[2 * j] = a2.x; r2[2 * j + 1] = a2.y;
                    }
                    if (a.gin_masked) st_u4(gdst, make_uint4(mz[0], mz[1], mz[2], mz[3]), true);
                }
            }
            continue;
        }
        // ---- input-gradient epilogue: lane holds ci = nt*16 + lg*4 + {0..3} of pixel tile0 + (wave*PT+pt)*16 + l15
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int m = tile0 + (wave * PT + pt) * 16 + l15;
            const bool mok = m < mend;
#pragma unroll
            for (int nt = 0; nt < NTI; ++nt) {
                const int ci = nt * 16 + lg * 4;
                const bool ok = mok && ci < cis;
                float v[4] = {acc_g[pt][nt][0], acc_g[pt][nt][1], acc_g[pt][nt][2], acc_g[pt][nt][3]};
                float d1[4] = {0.f, 0.f, 0.f, 0.f}, d2[4] = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const size_t o = (size_t)m * a.Ci + ci0 + ci;
                    if (a.resid) {
                        const uint2 rv = rpre[pt][nt];
                        v[0] += bf_lo(rv.x); v[1] += bf_hi(rv.x); v[2] += bf_lo(rv.y); v[3] += bf_hi(rv.y);
                    }
                    uint2 pk;
                    pk.x = pack_bf16(v[0], v[1]);
                    pk.y = pack_bf16(v[2], v[3]);
                    st_u2((uint16_t*)a.gin + o, pk, a.nt);
                    if (do_red) {
                        // dz = g*[s*y+t>0] with g as stored (bf16), y = raw output of the reduce target; xhat = y*invstd - mean*invstd
                        const uint2 yv = ypre[pt][nt];
                        const float4 cs = *(const float4*)(lds_rc + ci), ct = *(const float4*)(lds_rc + CIP + ci);
                        const float4 cI = *(const float4*)(lds_rc + 2 * CIP + ci), cm = *(const float4*)(lds_rc + 3 * CIP + ci);
                        const uint32_t gu[2] = {pk.x, pk.y}, yu[2] = {yv.x, yv.y};
                        const mnas_f2 s2[2] = {{cs.x, cs.y}, {cs.z, cs.w}}, t2[2] = {{ct.x, ct.y}, {ct.z, ct.w}};
                        const mnas_f2 i2[2] = {{cI.x, cI.y}, {cI.z, cI.w}}, m2[2] = {{cm.x, cm.y}, {cm.z, cm.w}};
#pragma unroll
                        for (int h = 0; h < 2; ++h) {        // channel pairs in float2 (v_pk_fma_f32)
                            const mnas_f2 gq = mnas_bf2(gu[h]), yq = mnas_bf2(yu[h]);
                            const mnas_f2 z = mnas_f2fma(yq, s2[h], t2[h]);
                            mnas_f2 dz;
                            dz.x = (z.x > 0.f) ? gq.x : 0.f;
                            dz.y = (z.y > 0.f) ? gq.y : 0.f;
                            const mnas_f2 pr = dz * mnas_f2fma(yq, i2[h], m2[h]);
                            d1[2 * h] = dz.x; d1[2 * h + 1] = dz.y;
                            d2[2 * h] = pr.x; d2[2 * h + 1] = pr.y;
                        }
                    }
                }
                if (do_red && REGSTAT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { rs1[REGSTAT ? nt : 0][r] += d1[r]; rs2[REGSTAT ? nt : 0][r] += d2[r]; }
                }
                if (do_red && !REGSTAT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x1 = d1[r], x2 = d2[r];
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
                        if (l15 == 0) {                      // this lane is the only writer of (wave, channel)
                            lds_st[(wave * 2 + 0) * CIP + ci + r] += x1;
                            lds_st[(wave * 2 + 1) * CIP + ci + r] += x2;
                        }
                    }
                }
            }
        }
    }

    // ---- weight-gradient partial of this workgroup: wpartial[blockIdx.x][co][ci]
    float* wp = a.wpartial + (size_t)blockIdx.x * a.Co * a.Ci;
#pragma unroll
    for (int i = 0; i < NOWN; ++i) {
        const int own = wave + 4 * i;
        if (own >= (OWN_O ? NTO : NTI)) continue;
#pragma unroll
        for (int j = 0; j < NOTH; ++j) {
            const int to = OWN_O ? own : j, ti = OWN_O ? j : own;
            const int ci = ti * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = to * 16 + lg * 4 + r;
                if (co < a.Co && ci < cis) wp[(size_t)co * a.Ci + ci0 + ci] = acc_w[i][j][r];
            }
        }
    }
    if (OS && do_red) {
        // per-thread sums -> per-channel: the threads of one channel column added in thread order (deterministic)
        float* fin = (float*)tile_d;                         // [256][16] (tile_d + tile_a hold >= 16 KB)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) { fin[tid * 16 + j] = r1[j]; fin[tid * 16 + 8 + j] = r2[j]; }
        __syncthreads();
        for (int i = tid; i < 2 * CIP; i += 256) {
            const int r = i / CIP, c = i % CIP;
            float v = 0.f;
            for (int th = c >> 3; th < TCOLS; th += NCH8) v += fin[th * 16 + r * 8 + (c & 7)];
            if (c < cis) a.red_partial[((size_t)r * a.Ci + ci0 + c) * gridDim.x + blockIdx.x] = v;   // [2][Ci][P]
        }
    } else if (do_red) {
        if (REGSTAT) {
#pragma unroll
            for (int nt = 0; nt < (REGSTAT ? NTI : 1); ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x1 = rs1[nt][r], x2 = rs2[nt][r];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
                    if (l15 == 0) {
                        lds_st[(wave * 2 + 0) * CIP + nt * 16 + lg * 4 + r] = x1;
                        lds_st[(wave * 2 + 1) * CIP + nt * 16 + lg * 4 + r] = x2;
                    }
                }
        }
        __syncthreads();
        for (int i = tid; i < 2 * CIP; i += 256) {
            const int r = i / CIP, c = i % CIP;
            const float v = ((lds_st[(0 * 2 + r) * CIP + c] + lds_st[(1 * 2 + r) * CIP + c]) + lds_st[(2 * 2 + r) * CIP + c]) +
                            lds_st[(3 * 2 + r) * CIP + c];
            if (c < cis) a.red_partial[((size_t)r * a.Ci + ci0 + c) * gridDim.x + blockIdx.x] = v;   // [2][Ci][P]
        }
    }
}

static int pw_bwd_outstage() {
    static int on = -1;
    if (on < 0) on = mnas_diag_env("MNAS_PWB_OS", 2);      // 0 off, 1 narrowing convs, 2 + the 14x14 channel slices
    return on;
}
template <int NTO, int NTI, int PT>
static int launch_pw_bwd(const PwBwdArgs& a, int nparts, hipStream_t stream, int nslices = 1) {
    constexpr int BP = 64 * PT, COP = NTO * 16, CIP = NTI * 16;
    constexpr bool PF = MNAS_PWB_PF && NTO != 15 && !(NTO == 1 && NTI == 2);    // loads one tile ahead (see the tile walk)
    const bool recomp = a.dy.y == nullptr;
    size_t lds = (size_t)(5 * COP + 14 * CIP) * sizeof(float) +
                 ((size_t)CIP * (a.Kd + 8) + (size_t)BP * (a.Kd + 8) + (size_t)BP * (CIP + 8)) * 2;
    if (recomp) lds += (size_t)COP * (a.Kf + 8) * 2 + (size_t)COP * sizeof(float);
    if (lds > 160 * 1024) return MNAS_EINVAL;
    // narrowing conv with >= 48 result channels: gin is the wide tensor (32 -> 16 at 112x112 is 124 us without and 184 us with
    // the out-stage: two more barriers per tile and nothing to win on 64-byte rows)
    if constexpr ((NTI > NTO && NTI >= 3) || (NTI == NTO && NTI >= 5)) {       // (the 14x14 channel slices: 5/5, 6/6)
        if (recomp) return MNAS_EINVAL;
        // the out-stage forms reduce against the x chunks they staged: red_y must BE x.data (it is for a project conv: the
        // reduce target is the depthwise conv that produced x); anything else takes the plain epilogue below
        const bool redx = !a.red_partial || a.red_y == a.x.data;
        if (pw_bwd_outstage() >= (NTI == NTO ? 2 : 1) && !a.resid && redx) {
            hipLaunchKernelGGL((k_pw_bwd<NTO, NTI, PT, true, 0, PF>), dim3(nparts, nslices), dim3(256), lds, stream, a);
            MNAS_CHECK_LAUNCH();
            return MNAS_OK;
        }
    }
    if (a.gin_masked) return MNAS_EINVAL;                     // only the out-stage kernel masks (a -DMNAS_DIAG build can switch it off)
    if constexpr (NTO > NTI && NTI <= 2) {                    // the expand convs of the 112x112 / 56x56 stages
        if (recomp) {
            hipLaunchKernelGGL((k_pw_bwd<NTO, NTI, PT, false, 2, PF>), dim3(nparts, nslices), dim3(256), lds, stream, a);
            MNAS_CHECK_LAUNCH();
            return MNAS_OK;
        }
    }
    if (recomp) return MNAS_EINVAL;
    hipLaunchKernelGGL((k_pw_bwd<NTO, NTI, PT, false, 0, PF>), dim3(nparts, nslices), dim3(256), lds, stream, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// Supported channel pairs: the pointwise convs of the 112x112 / 56x56 / 28x28 stages of MNASNet-1.0 (16<->32/48, 24<->72,
// 40<->240 channels; 40<->120 for the cut_channels_first=True topology) and the narrowing convs of the 14x14 stage (480->80, 576->96; channel slices).  Other shapes:
// MNAS_EINVAL (use mnas_conv_gemm + mnas_conv_wgrad).
struct PwCfg { int nto, nti_total, nti_slice, nslices, pt; };
static const PwCfg* pw_cfg(int Ci, int Co) {
    if ((Ci & 7) || (Co & 7) || Ci < 8 || Co < 8) return nullptr;
    const int nto = (Co + 15) / 16, nti = (Ci + 15) / 16;
    // pixel tiles: 128 (pt 2) on the 112x112 layers, 64 (pt 1) elsewhere.  24 -> 72 at 56x56 moved to 64-pixel tiles in round 4:
    // 185 -> 143 VGPRs (RECOMP form 149 -> 108) = 3-4 resident workgroups per CU instead of 2, step 10.98 -> 10.83 ms in one
    // call; the same change is neutral for 72 -> 24 and loses on the 112x112 layers (11.07 ms with all five on 64-pixel tiles)
    static const PwCfg cfgs[] = {{1, 2, 2, 1, 2}, {1, 3, 3, 1, 2}, {3, 1, 1, 1, 2}, {5, 2, 2, 1, 1}, {2, 5, 5, 1, 2}, {15, 3, 3, 1, 1},
                                 {8, 3, 3, 1, 1},       // 40 -> 120 and
                                 {3, 8, 4, 2, 1},       // 120 -> 40 (two 64-channel slices): the 28x28 blocks of Mnasnet(cut_channels_first=True)
                                 {3, 15, 5, 3, 1},      // 240 -> 40: three 80-channel slices
                                 {5, 30, 5, 6, 1},      // 480 -> 80: six 80-channel slices
                                 {6, 36, 6, 6, 1}};     // 576 -> 96: six 96-channel slices
    for (auto& c : cfgs) if (c.nto == nto && c.nti_total == nti) return &c;
    return nullptr;
}
extern "C" int mnas_pw_bwd_supported(int Ci, int Co) { return pw_cfg(Ci, Co) ? 1 : 0; }
// pixels per tile / channel slices (grid.y) of the launch for this channel pair: what a caller sizing MnasPwBwd.seg_px needs
extern "C" int mnas_pw_bwd_tile_pixels(int Ci, int Co) { const PwCfg* c = pw_cfg(Ci, Co); return c ? 64 * c->pt : -1; }
extern "C" int mnas_pw_bwd_slices(int Ci, int Co) { const PwCfg* c = pw_cfg(Ci, Co); return c ? c->nslices : -1; }
// bit 1: RECOMP (dy.y == NULL + w_fwd) available, bit 2: out-stage form (gin_masked) -- mirrors launch_pw_bwd's dispatch (bit 0 was
// the NOGIN form, removed in round 5)
extern "C" int mnas_pw_bwd_forms(int Ci, int Co) {
    const PwCfg* c = pw_cfg(Ci, Co);
    if (!c) return 0;
    int f = 0;
    if (c->nto > c->nti_slice && c->nti_slice <= 2 && c->nslices == 1) f |= 2;
    if ((c->nti_slice > c->nto && c->nti_slice >= 3) || (c->nti_slice == c->nto && c->nti_slice >= 5)) f |= 4;     // out-stage form
    return f;
}

extern "C" int mnas_pw_bwd(const MnasPwBwd* c, void* stream) {
    if (!c || c->M < 1 || c->nparts < 1 || c->nparts > 65535 || !mnas_pw_bwd_supported(c->Ci, c->Co)) return MNAS_EINVAL;
    if (!c->x.data || !c->dy.g || !c->dy.coef || !c->w || !c->wpartial || !c->gin) return MNAS_EINVAL;
    if (!c->dy.y && !c->w_fwd) return MNAS_EINVAL;           // RECOMP needs the forward weights
    if (c->red_partial && (!c->red_bn || !c->red_y)) return MNAS_EINVAL;
    PwBwdArgs a;
    a.M = c->M; a.Ci = c->Ci; a.Co = c->Co; a.Kd = (c->Co + 31) / 32 * 32;
    a.x = c->x; a.dy = c->dy; a.w = (const uint16_t*)c->w; a.resid = c->resid; a.gin = c->gin;
    a.wpartial = c->wpartial; a.red_partial = c->red_partial; a.red_y = c->red_y; a.red_bn = c->red_bn;
    a.nt = (mnas_nt_mask() & MNAS_NT_PW_BWD) ? 1 : 0;
    a.w_fwd = (const uint16_t*)c->w_fwd; a.b_fwd = c->b_fwd; a.Kf = (c->Ci + 31) / 32 * 32;
    a.gin_masked = c->gin_masked;
    a.seg_px = c->seg_px;
    if (a.seg_px < 0 || (a.seg_px > 0 && ((int64_t)a.seg_px * c->nparts < c->M || (int64_t)a.seg_px * (c->nparts - 1) >= c->M))) return MNAS_EINVAL;
    if (a.gin_masked && (!c->red_partial || c->resid || c->red_y != c->x.data || !(mnas_pw_bwd_forms(c->Ci, c->Co) & 4))) return MNAS_EINVAL;
    const PwCfg* cfg = pw_cfg(c->Ci, c->Co);
    hipStream_t s = (hipStream_t)stream;
#define MNAS_PWB(O_, I_, P_) if (cfg->nto == O_ && cfg->nti_slice == I_ && cfg->pt == P_) return launch_pw_bwd<O_, I_, P_>(a, c->nparts, s, cfg->nslices);
    MNAS_PWB(1, 2, 2) MNAS_PWB(1, 3, 2) MNAS_PWB(3, 1, 2) MNAS_PWB(5, 2, 1) MNAS_PWB(2, 5, 2) MNAS_PWB(15, 3, 1)
    MNAS_PWB(3, 5, 1) MNAS_PWB(5, 5, 1) MNAS_PWB(6, 6, 1) MNAS_PWB(8, 3, 1) MNAS_PWB(3, 4, 1)
#undef MNAS_PWB
    return MNAS_EINVAL;
}
