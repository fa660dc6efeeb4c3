// Weight gradient of the 1x1 / dense kxk convolutions on the matrix cores:
//     dW[co][tap*Ci+ci] = sum over output pixels of dy[pix][co] * act(x)[src(pix,tap)][ci]
// Replaces ATen's conv2d weight-gradient for ConvBlock(groups=1) (autograd mirror of mnasnet.py:48-54).
//
// GEMM view: D[co][k] (16x16 MFMA tiles) with the reduction running over PIXELS.  Both operands are stored
// pixel-major in HBM/LDS (NHWC), i.e. transposed with respect to what the MFMA fragments want (8 consecutive
// reduction indices per lane), so the fragments are read with gfx950's LDS transpose-read
// ds_read_b64_tr_b16: lane i of a 16-lane group points at row (i>>2), 4 columns starting at (i&3)*4 and
// receives column i of 4 consecutive rows (verified by tools/probe).  Two such reads = one bf16x8 fragment.
//
// A workgroup owns a (<=64 cout) x (<=64 k) slab of dW and a contiguous range of pixels; its 4 waves split
// each 128-pixel chunk (32 pixels = one MFMA k-step each) and hold the whole slab in accumulators; the
// four copies are summed through LDS in wave order once at the end and written to partial[split][co][k] (plain
// stores, no atomics, deterministic); mnas_wgrad_finalize reduces the splits.
// dy-on-load and act-on-load are applied while staging, exactly as in mnas_gemm.hip.
#include "mnas_common.h"

typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((address_space(3))) s4_t* lds_s4_ptr;

struct WgradArgs {
    int M, Hi, Wi, Ci, Ho, Wo, Co;
    int kh, kw, stride, pad;
    int Ktot, taps, is_pw, chunk;   // chunk = pixels per split (multiple of 128)
    MnasActIn x;
    MnasGradIn dy;
    float* partial;
};

#define WG_BPIX 128
#define WG_T 4            // 4x4 tiles of 16x16 per workgroup slab

__device__ __forceinline__ bf16x8_t tr_frag(const uint16_t* tile, int ld, int row0, int col0, int lane) {
    // rows row0 + (lane>>4)*8 + {0..7}, column col0 + (lane&15)
    const int i = lane & 15, g = lane >> 4;
    const uint16_t* p = tile + (row0 + g * 8 + (i >> 2)) * ld + col0 + (i & 3) * 4;
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)p);
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p + 4 * ld));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool STEM>
__global__ __launch_bounds__(256, 2) void k_wgrad(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int co0 = blockIdx.y * 64, kc0 = blockIdx.z * 64;
    const int con = min(64, a.Co - co0);                  // valid couts in the slab (multiple of 8)
    const int kn = min(64, a.Ktot - kc0);                 // valid k columns (multiple of 8)
    const int ctn = (con + 15) >> 4, ktn = (kn + 15) >> 4;
    const int ldd = 64 + 8, lda = 64 + 8;                 // LDS row strides (elements)
    const int xcols = (a.taps == 1) ? 64 : a.Ci;
    float* lds_cd = (float*)smem;                         // [5][64]   dy coefficients for couts co0..
    float* lds_cx = lds_cd + 5 * 64;                      // [2][xcols] scale/shift of x
    uint16_t* tile_d = (uint16_t*)(lds_cx + 2 * xcols);   // [128][ldd]
    uint16_t* tile_a = tile_d + WG_BPIX * ldd;            // [128][lda]
    float* lds_out = (float*)tile_d;                      // reused at the end: [64][65]

    const bool hasdy = a.dy.coef != nullptr;              // false: dy.g is a materialised dy
    for (int i = tid; i < 5 * 64; i += 256) {
        const int r = i >> 6, c = co0 + (i & 63);
        lds_cd[i] = (hasdy && c < a.Co) ? a.dy.coef[(size_t)r * a.Co + c] : 0.f;
    }
    const bool hasx = !STEM && a.x.scale != nullptr;
    if (hasx)
        for (int i = tid; i < 2 * xcols; i += 256) {
            const int r = i / xcols, c = (a.taps == 1 ? kc0 : 0) + i % xcols;
            lds_cx[i] = (c < a.Ci) ? (r == 0 ? a.x.scale[c] : a.x.shift[c]) : 0.f;
        }
    // both tiles start zeroed: staging only ever writes the VALID 16-byte chunks of a row (cwd / cwa per row), so
    // the padding columns of a partially filled 16-wide MFMA tile stay zero for the whole kernel
    for (int i = tid; i < 2 * WG_BPIX * 72 / 8; i += 256) ((uint4*)tile_d)[i] = make_uint4(0, 0, 0, 0);

    // ---- per-thread staging plan (tile-invariant): up to 4 (pixel, chunk) slots per tile for each operand
    const int cwd = (con + 7) >> 3, cwa = STEM ? 4 : ((kn + 7) >> 3);
    int pd[4], cd8[4], pa[4], ca8[4], aci[4], ath[4], atw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        pd[i] = q / cwd; cd8[i] = q - pd[i] * cwd;
        if (pd[i] >= WG_BPIX) pd[i] = -1;
        pa[i] = q / cwa; ca8[i] = q - pa[i] * cwa;
        if (pa[i] >= WG_BPIX) pa[i] = -1;
        const int k = kc0 + ca8[i] * 8;
        aci[i] = k; ath[i] = 0; atw[i] = 0;
        if (!STEM && !a.is_pw) {
            const int tap = k / a.Ci;
            aci[i] = k - tap * a.Ci;
            ath[i] = tap / a.kw; atw[i] = tap - ath[i] * a.kw;
        }
    }

    f32x4_t acc[WG_T][WG_T];
#pragma unroll
    for (int i = 0; i < WG_T; ++i)
#pragma unroll
        for (int j = 0; j < WG_T; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int p_begin = blockIdx.x * a.chunk;
    const int p_end = min(a.M, p_begin + a.chunk);
    // software pipeline: the global loads of chunk pc+128 are issued before the MFMAs of chunk pc and consumed (transform
    // + LDS write) one iteration later
    uint4 vg[4], vy[4], vx[4];
    bool okd[4], oka[4];
    auto issue = [&](int pc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            okd[i] = false;
            vg[i] = make_uint4(0, 0, 0, 0); vy[i] = make_uint4(0, 0, 0, 0);
            if (pd[i] >= 0) {
                const int m = pc + pd[i];
                if (m < p_end) {
                    okd[i] = true;
                    const size_t off = (size_t)m * a.Co + co0 + cd8[i] * 8;
                    vg[i] = *(const uint4*)((const uint16_t*)a.dy.g + off);
                    if (hasdy) vy[i] = *(const uint4*)((const uint16_t*)a.dy.y + off);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            oka[i] = false;
            vx[i] = make_uint4(0, 0, 0, 0);
            if (pa[i] < 0) continue;
            const int m = pc + pa[i];
            const int k = kc0 + ca8[i] * 8;
            if (STEM) {
                // im2col of the fp32 NCHW image: k = ci*9 + kh*3 + kw (reference weight order)
                if (m < p_end && k < 32) {
                    const int hw = a.Ho * a.Wo;
                    const int n = m / hw, rem = m - n * hw;
                    const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
                    const float* x = (const float*)a.x.data;
                    float f[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kk = k + j;
                        const int c3 = kk / 9, r9 = kk - c3 * 9, t3 = r9 / 3, u3 = r9 - t3 * 3;
                        const int ih = oh * 2 + t3 - 1, iw = ow * 2 + u3 - 1;
                        const bool okj = kk < 27 && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
                        f[j] = okj ? x[(((size_t)n * 3 + c3) * a.Hi + ih) * a.Wi + iw] : 0.f;
                    }
                    vx[i] = pack8(f);
                }
            } else if (m < p_end && k < a.Ktot) {
                size_t src;
                bool inb = true;
                if (a.is_pw) {
                    src = (size_t)m * a.Ci + k;
                } else {
                    const int hw = a.Ho * a.Wo;
                    const int n = m / hw, rem = m - n * hw;
                    const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
                    const int ih = oh * a.stride + ath[i] - a.pad, iw = ow * a.stride + atw[i] - a.pad;
                    inb = ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
                    src = (((size_t)n * a.Hi + ih) * a.Wi + iw) * a.Ci + aci[i];
                }
                if (inb) {
                    oka[i] = true;
                    vx[i] = *(const uint4*)((const uint16_t*)a.x.data + src);
                }
            }
        }
    };
    if (p_begin < p_end) issue(p_begin);
    for (int pc = p_begin; pc < p_end; pc += WG_BPIX) {
        __syncthreads();
        // ---- transform + write
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (pd[i] < 0) continue;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (okd[i] && !hasdy) v = vg[i];           // materialised dy (mnas_dy_materialize): no transform
            else if (okd[i]) {
                float cf[5][8];
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    *(float4*)&cf[r][0] = *(const float4*)(lds_cd + r * 64 + cd8[i] * 8);
                    *(float4*)&cf[r][4] = *(const float4*)(lds_cd + r * 64 + cd8[i] * 8 + 4);
                }
                float o[8];
                dy8(vg[i], vy[i], cf[0], cf[1], cf[2], cf[3], cf[4], o);
                v = pack8(o);
            }
            *(uint4*)(tile_d + pd[i] * ldd + cd8[i] * 8) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (pa[i] < 0) continue;
            uint4 v = vx[i];
            if (hasx && oka[i]) {
                const int cc = (a.taps == 1) ? ca8[i] * 8 : aci[i];
                float s[8], t[8];
                *(float4*)&s[0] = *(const float4*)(lds_cx + cc);
                *(float4*)&s[4] = *(const float4*)(lds_cx + cc + 4);
                *(float4*)&t[0] = *(const float4*)(lds_cx + xcols + cc);
                *(float4*)&t[4] = *(const float4*)(lds_cx + xcols + cc + 4);
                v = act8(v, s, t);
            }
            *(uint4*)(tile_a + pa[i] * lda + ca8[i] * 8) = v;
        }
        __syncthreads();
        if (pc + WG_BPIX < p_end) issue(pc + WG_BPIX);
        // ---- each wave: its 32 pixels, all slab tiles
        bf16x8_t bf[WG_T];
#pragma unroll
        for (int kt = 0; kt < WG_T; ++kt)
            if (kt < ktn) bf[kt] = tr_frag(tile_a, lda, wave * 32, kt * 16, lane);
#pragma unroll
        for (int ct = 0; ct < WG_T; ++ct) {
            if (ct >= ctn) continue;
            const bf16x8_t af = tr_frag(tile_d, ldd, wave * 32, ct * 16, lane);
#pragma unroll
            for (int kt = 0; kt < WG_T; ++kt)
                if (kt < ktn) acc[ct][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[kt], acc[ct][kt], 0, 0, 0);
        }
    }
    // ---- sum the 4 waves' slabs through LDS in wave order (each lane owns its elements: no atomics, the sum is
    // bit-reproducible), then write partial[split][co][k]
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave != w) continue;
#pragma unroll
        for (int ct = 0; ct < WG_T; ++ct)
#pragma unroll
            for (int kt = 0; kt < WG_T; ++kt) {
                if (ct >= ctn || kt >= ktn) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* d = &lds_out[(ct * 16 + lg * 4 + r) * 65 + kt * 16 + l15];
                    *d = (w == 0 ? 0.f : *d) + acc[ct][kt][r];
                }
            }
    }
    __syncthreads();
    float* dst = a.partial + (size_t)blockIdx.x * a.Co * a.Ktot;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        if (r < con && c < kn) dst[(size_t)(co0 + r) * a.Ktot + kc0 + c] = lds_out[r * 65 + c];
    }
}

// ------------------------------------------------------------------------------------------------
// k_wgrad_t<CT,KT>: the same GEMM with a (32*CT cout) x (32*KT k) slab per workgroup.  The 4 waves tile the SLAB 2x2 (a wave
// owns CT x KT MFMA tiles for ALL pixels of the workgroup's range) instead of splitting the pixels of a 64x64 slab four ways:
// every staged 16 bytes feed (CT*KT)/4 times the MFMAs of k_wgrad<false>'s layout, each operand is re-read from L2
// Co/(32 CT) resp. K/(32 KT) times instead of Co/64, K/64 times, and there is no cross-wave reduction at the end (the
// accumulators go straight to partial[split]).  Pixel chunks of 64 (two MFMA k-steps), LDS tiles double-buffered: ONE barrier
// per chunk; the global loads of chunk c+2 are in flight while chunk c is multiplied.  Row pitch = tile + 16 elements: the four
// rows a 16-lane group of ds_read_b64_tr_b16 touches land in four disjoint 8-bank windows.
// ------------------------------------------------------------------------------------------------
#define WT_BP 64
#ifndef MNAS_WT_LEAN
#define MNAS_WT_LEAN 1       // 1: wave index / valid-tile counts in scalar registers, the k x k gather's pixel decode carried from
#endif                       //    chunk to chunk instead of two divisions per slot and chunk (0: A/B builds)
template <int CT, int KT>
__global__ __launch_bounds__(256, 2) void k_wgrad_t(WgradArgs a) {
    constexpr int COT = 32 * CT, KTT = 32 * KT, LDD = COT + 16, LDA = KTT + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = MNAS_WT_LEAN ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);   // scalar: the valid-tile tests below become scalar branches
    const int l15 = lane & 15, lg = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int co0 = blockIdx.y * COT, kc0 = blockIdx.z * KTT;
    const int xcols = (a.taps == 1) ? KTT : a.Ci;
    float* lds_cd = (float*)smem;                         // [5][COT]   dy coefficients for couts co0..
    float* lds_cx = lds_cd + 5 * COT;                     // [2][xcols] scale/shift of x
    uint16_t* tile_d = (uint16_t*)(lds_cx + 2 * xcols);   // [2][64][LDD]
    uint16_t* tile_a = tile_d + 2 * WT_BP * LDD;          // [2][64][LDA]

    const bool hasdy = a.dy.coef != nullptr;              // false: dy.g is a materialised dy
    const bool hasx = a.x.scale != nullptr;

    // ---- per-thread staging plan (chunk-invariant): CT dy slots and KT x slots of 16 bytes; a slot outside the tensor's
    // channels writes zeros, so the MFMA tiles past the slab's valid edge multiply zeros
    int pd[CT], cd8[CT], pa[KT], ca8[KT], aci[KT], ath[KT], atw[KT];
    bool vd[CT], va[KT];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int q = tid + 256 * i;
        pd[i] = q / (4 * CT); cd8[i] = q - pd[i] * (4 * CT);
        vd[i] = co0 + cd8[i] * 8 < a.Co;
    }
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        const int q = tid + 256 * i;
        pa[i] = q / (4 * KT); ca8[i] = q - pa[i] * (4 * KT);
        const int k = kc0 + ca8[i] * 8;
        va[i] = k < a.Ktot;
        aci[i] = k; ath[i] = 0; atw[i] = 0;
        if (!a.is_pw) {
            const int tap = k / a.Ci;
            aci[i] = k - tap * a.Ci;
            ath[i] = tap / a.kw; atw[i] = tap - ath[i] * a.kw;
        }
    }

    f32x4_t acc[CT][KT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // valid 16-wide tiles of this wave's part of the slab
    const int ctn = max(0, min(CT, (min(COT, a.Co - co0) - wr * CT * 16 + 15) >> 4));
    const int ktn = max(0, min(KT, (min(KTT, a.Ktot - kc0) - wc * KT * 16 + 15) >> 4));

    const int p_begin = blockIdx.x * a.chunk;
    const int p_end = min(a.M, p_begin + a.chunk);
    const int nch = (p_end - p_begin + WT_BP - 1) / WT_BP;
    uint4 vg[CT], vy[CT], vx[KT];
    unsigned okd = 0, oka = 0;
    const int hw = a.Ho * a.Wo;
#if MNAS_WT_LEAN
    // k x k gather: (image, row, column) of each x slot's pixel, decoded once and advanced by one chunk (WT_BP pixels) per issue()
    // -- issue() is always called for consecutive chunks starting at p_begin
    int sn[KT], soh[KT], sow[KT];
    const int adv_n = WT_BP / hw, adv_oh = (WT_BP - adv_n * hw) / a.Wo, adv_ow = WT_BP - adv_n * hw - adv_oh * a.Wo;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        sn[i] = 0; soh[i] = 0; sow[i] = 0;
        if (!a.is_pw) {
            const int m = p_begin + pa[i];
            sn[i] = m / hw;
            const int rem = m - sn[i] * hw;
            soh[i] = rem / a.Wo; sow[i] = rem - soh[i] * a.Wo;
        }
    }
#endif
    // (a slot that is not loaded goes to the tile as zeros from zero-initialised registers: leaving the registers alone and
    // branching in stage() measured 25 % slower -- undefined on the not-loaded path, the loads may be sunk to their use = no prefetch; one 8-channel group per thread for all its dy slots -- the coefficient rows read
    // from LDS once per chunk instead of once per slot, 240 threads x 6 slots for the 160-channel slab -- measured equal: round 6)
    auto issue = [&](int pc) {
        okd = 0; oka = 0;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            vg[i] = make_uint4(0, 0, 0, 0); vy[i] = make_uint4(0, 0, 0, 0);
            const int m = pc + pd[i];
            if (vd[i] && m < p_end) {
                okd |= 1u << i;
                const size_t off = (size_t)m * a.Co + co0 + cd8[i] * 8;
                vg[i] = *(const uint4*)((const uint16_t*)a.dy.g + off);
                if (hasdy) vy[i] = *(const uint4*)((const uint16_t*)a.dy.y + off);
            }
        }
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            vx[i] = make_uint4(0, 0, 0, 0);
            const int m = pc + pa[i];
            if (!va[i] || m >= p_end) continue;
            size_t src;
            bool inb = true;
            if (a.is_pw) {
                src = (size_t)m * a.Ci + aci[i];
            } else {
#if MNAS_WT_LEAN
                const int n = sn[i], oh = soh[i], ow = sow[i];
#else
                const int n = m / hw, rem = m - n * hw;
                const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
#endif
                const int ih = oh * a.stride + ath[i] - a.pad, iw = ow * a.stride + atw[i] - a.pad;
                inb = ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
                src = (((size_t)n * a.Hi + ih) * a.Wi + iw) * a.Ci + aci[i];
            }
            if (inb) {
                oka |= 1u << i;
                vx[i] = *(const uint4*)((const uint16_t*)a.x.data + src);
            }
        }
#if MNAS_WT_LEAN
        if (!a.is_pw) {
#pragma unroll
            for (int i = 0; i < KT; ++i) {
                int ow = sow[i] + adv_ow, oh = soh[i] + adv_oh, n = sn[i] + adv_n;
                if (ow >= a.Wo) { ow -= a.Wo; ++oh; }
                if (oh >= a.Ho) { oh -= a.Ho; ++n; }
                sow[i] = ow; soh[i] = oh; sn[i] = n;
            }
        }
#endif
    };
    auto stage = [&](int buf) {       // transform + LDS write of the chunk held in vg / vy / vx
        uint16_t* td = tile_d + buf * WT_BP * LDD;
        uint16_t* ta = tile_a + buf * WT_BP * LDA;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            uint4 v = vg[i];                                  // materialised dy (mnas_dy_materialize): no transform; zeros when masked
            if (hasdy && ((okd >> i) & 1)) {
                float cf[5][8];
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    *(float4*)&cf[r][0] = *(const float4*)(lds_cd + r * COT + cd8[i] * 8);
                    *(float4*)&cf[r][4] = *(const float4*)(lds_cd + r * COT + cd8[i] * 8 + 4);
                }
                float o[8];
                dy8(vg[i], vy[i], cf[0], cf[1], cf[2], cf[3], cf[4], o);
                v = pack8(o);
            }
            *(uint4*)(td + pd[i] * LDD + cd8[i] * 8) = v;
        }
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            uint4 v = vx[i];
            if (hasx && ((oka >> i) & 1)) {
                const int cc = (a.taps == 1) ? ca8[i] * 8 : aci[i];
                float s[8], t[8];
                *(float4*)&s[0] = *(const float4*)(lds_cx + cc);
                *(float4*)&s[4] = *(const float4*)(lds_cx + cc + 4);
                *(float4*)&t[0] = *(const float4*)(lds_cx + xcols + cc);
                *(float4*)&t[4] = *(const float4*)(lds_cx + xcols + cc + 4);
                v = act8(v, s, t);
            }
            *(uint4*)(ta + pa[i] * LDA + ca8[i] * 8) = v;
        }
    };
    if (MNAS_EARLY && nch > 0) issue(p_begin);                // the first chunk's loads and the coefficient tables share one round trip
    mnas_fill_table(lds_cd, 5 * COT, tid, 256, [&](int i) {
        const int r = i / COT, c = co0 + i % COT;
        return (hasdy && c < a.Co) ? a.dy.coef[(size_t)r * a.Co + c] : 0.f;
    });
    if (hasx)
        mnas_fill_table(lds_cx, 2 * xcols, tid, 256, [&](int i) {
            const int r = i / xcols, c = (a.taps == 1 ? kc0 : 0) + i % xcols;
            return (c < a.Ci) ? (r == 0 ? a.x.scale[c] : a.x.shift[c]) : 0.f;
        });
    __syncthreads();                                          // coefficient tables
    if (nch > 0) {
        if (!MNAS_EARLY) issue(p_begin);
        stage(0);
        if (nch > 1) issue(p_begin + WT_BP);
    }
    for (int c = 0; c < nch; ++c) {
        __syncthreads();      // tile c complete; everyone is done multiplying tile c-1, whose buffer the next stage() overwrites
        if (c + 1 < nch) {
            stage((c + 1) & 1);
            if (c + 2 < nch) issue(p_begin + (c + 2) * WT_BP);
        }
        const uint16_t* td = tile_d + (c & 1) * WT_BP * LDD;
        const uint16_t* ta = tile_a + (c & 1) * WT_BP * LDA;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t bf[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                if (kt < ktn) bf[kt] = tr_frag(ta, LDA, ks * 32, (wc * KT + kt) * 16, lane);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (ct >= ctn) continue;
                const bf16x8_t af = tr_frag(td, LDD, ks * 32, (wr * CT + ct) * 16, lane);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    if (kt < ktn) acc[ct][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[kt], acc[ct][kt], 0, 0, 0);
            }
        }
    }
    // ---- partial[split][co][k] straight from the accumulators (D: row = lg*4 + r -> cout, column = l15 -> k)
    float* dst = a.partial + (size_t)blockIdx.x * a.Co * a.Ktot;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (ct >= ctn || kt >= ktn) continue;
            const int col = kc0 + (wc * KT + kt) * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = co0 + (wr * CT + ct) * 16 + lg * 4 + r;
                if (row < a.Co && col < a.Ktot) dst[(size_t)row * a.Ktot + col] = acc[ct][kt][r];
            }
        }
}

// Slab shape per launch: the (CT, KT) pair that minimises padded MFMA work plus staged bytes (both operands are re-read once
// per slab of the OTHER dimension); CT <= 5, KT <= 4.
static void wgrad_pick_tiles(int Co, int K, int* ct, int* kt) {
#ifdef MNAS_DIAG
    {   // diagnosis build only: MNAS_WT = 10*CT + KT forces the slab shape (tools/kbench_wgrad.py sweeps)
        const int f = mnas_diag_env("MNAS_WT", 0);
        if (f > 0) { *ct = f / 10; *kt = f % 10; return; }
    }
#endif
    double best = 1e300;
    for (int c = 5; c >= 1; --c)
        for (int k = 4; k >= 1; --k) {
            if (c == 5 && k == 4) continue;          // 80 accumulator registers + the staging registers do not fit 256 VGPRs
            const double cop = (double)((Co + 32 * c - 1) / (32 * c)) * 32 * c, kp = (double)((K + 32 * k - 1) / (32 * k)) * 32 * k;
            const double cost = cop * kp * (1.0 + 48.0 / (32 * c) + 48.0 / (32 * k));
            if (cost < best) { best = cost; *ct = c; *kt = k; }
        }
}
extern "C" int mnas_conv_wgrad_slabs(int Co, int Ci, int taps) {
    if (Co < 1 || Ci < 1 || taps < 1) return 0;
    int ct, kt;
    wgrad_pick_tiles(Co, taps * Ci, &ct, &kt);
    return ((Co + 32 * ct - 1) / (32 * ct)) * ((taps * Ci + 32 * kt - 1) / (32 * kt));
}

template <int CT, int KT>
static void wgrad_t_launch(const WgradArgs& a, int nsplit, hipStream_t s) {
    const int xcols = (a.taps == 1) ? 32 * KT : a.Ci;
    const size_t lds = (size_t)(5 * 32 * CT + 2 * xcols) * sizeof(float) + (size_t)2 * WT_BP * (32 * CT + 16 + 32 * KT + 16) * 2;
    dim3 grid(nsplit, (a.Co + 32 * CT - 1) / (32 * CT), (a.Ktot + 32 * KT - 1) / (32 * KT));
    hipLaunchKernelGGL((k_wgrad_t<CT, KT>), grid, dim3(256), lds, s, a);
}

extern "C" int mnas_conv_wgrad(const MnasConvWgrad* c, void* stream) {
    if (!c || (c->Ci & 7) || (c->Co & 7) || c->nsplit < 1) return MNAS_EINVAL;
    WgradArgs a;
    a.M = c->N * c->Ho * c->Wo;
    a.Hi = c->Hi; a.Wi = c->Wi; a.Ci = c->Ci; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co;
    a.kh = c->kh; a.kw = c->kw; a.stride = c->stride; a.pad = c->pad;
    a.taps = c->kh * c->kw;
    a.Ktot = a.taps * c->Ci;
    a.is_pw = (a.taps == 1 && c->stride == 1 && c->pad == 0) ? 1 : 0;
    if (a.taps == 1 && !a.is_pw) return MNAS_EINVAL;
    if (a.taps != 1 && c->Ci > 1024) return MNAS_EINVAL;
    const int per = (a.M + c->nsplit - 1) / c->nsplit;
    a.chunk = (per + WT_BP - 1) / WT_BP * WT_BP;
    a.x = c->x; a.dy = c->dy; a.partial = c->partial;
    int ct, kt;
    wgrad_pick_tiles(a.Co, a.Ktot, &ct, &kt);
    hipStream_t s = (hipStream_t)stream;
#define MNAS_WT(C_, K_) if (ct == C_ && kt == K_) wgrad_t_launch<C_, K_>(a, c->nsplit, s);
#define MNAS_WTC(C_) MNAS_WT(C_, 1) MNAS_WT(C_, 2) MNAS_WT(C_, 3) MNAS_WT(C_, 4)
    MNAS_WTC(1) MNAS_WTC(2) MNAS_WTC(3) MNAS_WTC(4) MNAS_WT(5, 1) MNAS_WT(5, 2) MNAS_WT(5, 3)
#undef MNAS_WTC
#undef MNAS_WT
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// Stem weight gradient: partial[split][co][27] already in the reference's [co][ci][kh][kw] order
// (finalize with mnas_wgrad_finalize(partial, nsplit, Co, 27, 1, grad, ...)).
extern "C" int mnas_stem_wgrad(const MnasStemWgrad* c, void* stream) {
    if (!c || (c->Co & 7) || c->nparts < 1) return MNAS_EINVAL;
    {   // 32 couts, W % 4 == 0, Wo % 8 == 0: band kernel (csrc/mnas_stem.hip)
        const int rc = mnas_stem_wgrad_band(c, stream);
        if (rc != MNAS_EINVAL) return rc;
    }
    if (c->in_affine || c->in_u8) return MNAS_EINVAL;               // the fused input pipeline exists in the band kernels only
    WgradArgs a;
    a.M = c->N * c->Ho * c->Wo;
    a.Hi = c->H; a.Wi = c->W; a.Ci = 27; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co;
    a.kh = 3; a.kw = 3; a.stride = 2; a.pad = 1;
    a.taps = 1; a.Ktot = 27; a.is_pw = 0;
    const int per = (a.M + c->nparts - 1) / c->nparts;
    a.chunk = (per + WG_BPIX - 1) / WG_BPIX * WG_BPIX;
    a.x.data = c->x; a.x.scale = nullptr; a.x.shift = nullptr;
    a.dy = c->dy; a.partial = c->partial;
    const size_t lds = (size_t)(5 * 64 + 2 * 64) * sizeof(float) + (size_t)2 * WG_BPIX * 72 * 2;
    dim3 grid(c->nparts, (c->Co + 63) / 64, 1);
    hipLaunchKernelGGL(k_wgrad<true>, grid, dim3(256), lds, (hipStream_t)stream, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
