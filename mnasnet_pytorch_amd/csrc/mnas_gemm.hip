// 1x1 and dense kxk convolution forward / input-gradient as an im2col-free implicit GEMM on the gfx950
// matrix cores (v_mfma_f32_16x16x32_bf16), NHWC bf16 activations, fp32 accumulate.
// Replaces ATen conv2d forward + dgrad for ConvBlock(groups=1) (mnasnet.py:48-54; kernel_size 1 and 3).
//
// Orientation: the MFMA "A" operand (D rows) is the WEIGHT tile [16 cout][32 k], the "B" operand (D cols)
// is the ACTIVATION tile [16 pixels][32 k]; both are k-contiguous 16-byte fragments read from LDS with
// ds_read_b128.  D[cout][pixel] then leaves every lane with 4 CONSECUTIVE output channels of ONE pixel, so
// the epilogue (bias, BatchNorm partial statistics, residual add, bf16 pack) is lane-local and the store is
// one 8-byte write per fragment -- no LDS transpose on the way out.
//
// Prologue fusion: the activation tile is staged global -> registers -> LDS, applying on the way either
//   relu(scale*x+shift)                      (forward: the producer's BatchNorm+ReLU, "act-on-load"), or
//   c1*(g*[s*y+t>0]) + c2*y + c3             (dgrad: BatchNorm/ReLU backward, "dy-on-load").
// Epilogue fusion: per-workgroup partial (sum, sumsq) of the fp32 output for the NEXT BatchNorm.
//
// Work split: grid.x persistent workgroups stride over pixel tiles (BP = 64*PT pixels; each of the 4 waves
// owns PT groups of 16 pixels), grid.y over blocks of NT*16 output channels.  K = taps*Ci is walked in LDS
// chunks of <= 64.  Roofline: unfused these layers sit left of the bf16 ridge (AI 11-165 flop/B < 312), so
// HBM bounds them; MFMA utilisation only matters for the late, small-M layers.
#include "mnas_common.h"

struct IgemmArgs {
    int M, Hi, Wi, Ci, Ho, Wo, Co;
    int kh, kw, stride, pad;
    int Ktot, Kpad, kch, taps, is_pw, co_pad16;
    int s2, Mc, tpc;         // MODE 1, 3x3 stride 2: output-parity classes (pixels per class, tiles per class)
    MnasActIn act;
    MnasGradIn grad;
    const uint16_t* w;
    const float* bias;
    const void* resid;
    void* out;
    float* stats;
    const void* red_y;       // MODE 1: fused BN-backward reduce target (raw output of the ConvBlock whose g we produce)
    const float* red_bn;
    int nt;                  // nontemporal output stores
    float rcp_hw, rcp_wo, rcp_ci;   // reciprocals for igemm_fdiv (0 = use the exact integer division)
    const float* gate;       // MODE 0, 1x1 (GATE instantiations): per-(image, input channel) multiplier [N][Ci] applied after the activation
};

// floor(n / d) for 0 <= n < 2^24, d > 0 with a float reciprocal and one correction step (7 instructions instead of the ~25
// of the integer division sequence): the im2col address decode of the dense 3x3 / stride-2 forms runs it twice per staged
// 16-byte slot; k_igemm<dgrad, stride 2> at 112x112 spent 26 k VALU instructions per wave, most of them here.
__device__ __forceinline__ int igemm_fdiv(int n, int d, float rcp) {
    if (rcp == 0.f) return n / d;
    int q = (int)((float)n * rcp);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

template <int MODE, int NT, int PT, int KCH, bool PIPE, bool PAR2, bool GATE = false>
__global__ __launch_bounds__(256) void k_igemm(IgemmArgs a) {
    static_assert(!GATE || (MODE == 0 && !PAR2), "the gate is an act-on-load extension of the 1x1 forward");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BP = 64 * PT;
    constexpr int CROWS = (MODE == 1) ? 5 : 2;
    constexpr int kch = KCH;                       // K elements staged per LDS chunk (32 or 64)
    constexpr int ldk = KCH + 8;                   // LDS row stride in bf16 elements (16-byte padded)
    constexpr int kc8n = KCH >> 3;                 // 16-byte chunks per row
    constexpr int KSH = (KCH == 128) ? 4 : (KCH == 64) ? 3 : 2;       // log2(kc8n)
    const int ccols = (a.taps == 1) ? a.Kpad : a.Ci;   // coefficient columns kept in LDS (whole K: no per-chunk reload)
    float* lds_coef = (float*)smem;                                        // [CROWS][ccols]
    float* lds_red = lds_coef + CROWS * ccols;                             // [2][NT*16]
    float* lds_redc = lds_red + 2 * NT * 16;                               // [4][NT*16] fused-reduce coefficients
    uint16_t* lds_w = (uint16_t*)(lds_redc + 4 * NT * 16);                 // [NT*16][ldk]
    uint16_t* lds_a = lds_w + NT * 16 * ldk;                               // [BP][ldk]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.y * NT * 16;
    const int nkc = (a.Kpad + kch - 1) / kch;
    const bool has_coef = (MODE == 1 && a.grad.coef != nullptr) || (MODE == 0 && a.act.scale != nullptr);
    const bool has_y = MODE == 1 && a.grad.y != nullptr;        // false: grad.g is a materialised dy (mnas_dy_materialize)
    const float* coef_src[CROWS];
    if (MODE != 1) { coef_src[0] = a.act.scale; coef_src[1] = a.act.shift; }
    else if (has_coef) {
#pragma unroll
        for (int r = 0; r < CROWS; ++r) coef_src[r] = a.grad.coef + (size_t)r * a.Ci;
    }

    auto load_coefs = [&]() {
        if (!has_coef) return;
        for (int i = tid; i < CROWS * ccols; i += 256) {
            const int r = i / ccols, c = i % ccols;
            lds_coef[i] = (c < a.Ci) ? coef_src[r][c] : 0.f;
        }
    };
    constexpr int NW = (NT * 16 * kc8n + 255) / 256;
    uint4 wv[NW];
    // Stride-2 3x3 input gradient: an output pixel (oh, ow) only receives taps th = oh+1 (mod 2), tw = ow+1 (mod 2), i.e.
    // 1, 2, 2 or 4 of the 9 taps depending on its parity class.  Tiles are formed inside one class (cls = 2*ph + pw) and
    // walk only that class's taps: K shrinks from 9*Ci to {1,2,2,4}*Ci (2.25*Ci on average), no MFMA work on zeros.
    constexpr bool par2 = PAR2;            // host sets it for MODE 1 && a.s2 only
    int c_ph = 0, c_pw = 0, c_ntw = 1, c_K = a.Ktot;      // class of the tile being STAGED (set by decode_tile)
    auto decode_tile = [&](int t) -> int {                // returns the tile's first local pixel index
        if constexpr (!par2) return t * BP;
        const int cls = t / a.tpc;
        c_ph = cls >> 1; c_pw = cls & 1;
        c_ntw = c_pw ? 2 : 1;
        c_K = (c_ph ? 2 : 1) * c_ntw * a.Ci;
        return (t - cls * a.tpc) * BP;
    };
    auto load_w = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int q = tid + 256 * i;
            const int r = q >> KSH, kc8 = q & (kc8n - 1);
            const int k = k0 + kc8 * 8;
            wv[i] = make_uint4(0, 0, 0, 0);
            if (!(r < NT * 16 && n0 + r < a.co_pad16)) continue;
            if (par2) {
                if (k < c_K) {
                    const int tj = k / a.Ci, ci = k - tj * a.Ci;
                    const int thj = tj / c_ntw, twj = tj - thj * c_ntw;
                    const int th = c_ph ? 2 * thj : 1, tw = c_pw ? 2 * twj : 1;
                    wv[i] = *(const uint4*)(a.w + (size_t)(n0 + r) * a.Kpad + (th * 3 + tw) * a.Ci + ci);
                }
            } else if (k < a.Kpad) {
                wv[i] = *(const uint4*)(a.w + (size_t)(n0 + r) * a.Kpad + k);
            }
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int q = tid + 256 * i;
            const int r = q >> KSH, kc8 = q & (kc8n - 1);
            if (r < NT * 16) *(uint4*)(lds_w + r * ldk + kc8 * 8) = wv[i];
        }
    };
    // activation / gradient tile: each thread owns NA (pixel, k-chunk) slots; its k-chunk column is the same for all of
    // them (256 % kc8n == 0), so the per-channel coefficients are fetched once per chunk.  Software pipeline: load_a
    // issues the global loads of the NEXT (tile, chunk) before the MFMA phase of the current one; store_a applies the
    // prologue transform and writes LDS one phase later, when they have landed.
    constexpr int NA = BP * kc8n / 256;
    uint4 v0[NA], v1[MODE == 1 ? NA : 1];
    // GATE: a thread's slots share the channel chunk and a tile spans at most two images (host: H*W >= tile pixels), so two gate
    // vectors per load phase serve all of them (per-slot vectors cost 32 VGPRs and 2x the activation bytes in L1 traffic: the
    // 72 -> 24 project conv at 56x56 went 39 -> 78 us)
    float gq0[8], gq1[8];                          // (plain floats: pointer arithmetic across float4 members lands in scratch)
    unsigned okm = 0, gsel = 0;
    auto load_a = [&](int t, int k0) {
        const int tile0 = decode_tile(t);
        const int kc8 = tid & (kc8n - 1);
        const int k = k0 + kc8 * 8;
        const bool kok = k < (par2 ? c_K : a.Ktot);
        int ci = k, th = 0, tw = 0;
        if (par2) {
            const int tj = igemm_fdiv(k, a.Ci, a.rcp_ci);
            ci = k - tj * a.Ci;
            const int thj = tj / c_ntw, twj = tj - thj * c_ntw;
            th = c_ph ? 2 * thj : 1; tw = c_pw ? 2 * twj : 1;
        } else if (!a.is_pw && MODE != 2) {
            const int tap = igemm_fdiv(k, a.Ci, a.rcp_ci);
            ci = k - tap * a.Ci;
            th = tap / a.kw; tw = tap - th * a.kw;
        }
        okm = 0;
        int m_next = 0;                            // GATE: first pixel of the image after the tile's first one
        if constexpr (GATE) {
            const int hw = a.Ho * a.Wo;
            const int nA = igemm_fdiv(tile0 < a.M ? tile0 : a.M - 1, hw, a.rcp_hw);
            m_next = (nA + 1) * hw;
            gsel = 0;
            if (kok) {
                const float* gp = a.gate + (size_t)nA * a.Ci + k;
                const float* gn = gp + (m_next < a.M ? a.Ci : 0);
                const float4 a0 = *(const float4*)gp, a1 = *(const float4*)(gp + 4);
                const float4 b0 = *(const float4*)gn, b1 = *(const float4*)(gn + 4);
                gq0[0] = a0.x; gq0[1] = a0.y; gq0[2] = a0.z; gq0[3] = a0.w; gq0[4] = a1.x; gq0[5] = a1.y; gq0[6] = a1.z; gq0[7] = a1.w;
                gq1[0] = b0.x; gq1[1] = b0.y; gq1[2] = b0.z; gq1[3] = b0.w; gq1[4] = b1.x; gq1[5] = b1.y; gq1[6] = b1.z; gq1[7] = b1.w;
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int p = (tid >> KSH) + (256 >> KSH) * i;
            const int m = tile0 + p;
            if constexpr (GATE) gsel |= (m >= m_next ? 1u : 0u) << i;
            v0[i] = make_uint4(0, 0, 0, 0);
            if (MODE == 1) v1[i] = make_uint4(0, 0, 0, 0);
            if (MODE == 2) {
                // stem (mnasnet.py:179): im2col of the fp32 NCHW image, k = ci*9 + kh*3 + kw (reference
                // weight order), 3x3 stride 2 pad 1; Hi,Wi = image dims
                if (m < a.M) {
                    const int hw = a.Ho * a.Wo;
                    const int n = igemm_fdiv(m, hw, a.rcp_hw), rem = m - n * hw;
                    const int oh = igemm_fdiv(rem, a.Wo, a.rcp_wo), ow = rem - oh * a.Wo;
                    const float* x = (const float*)a.act.data;
                    float f[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kk = k + j;
                        const int c3 = kk / 9, r9 = kk - c3 * 9, t3 = r9 / 3, u3 = r9 - t3 * 3;
                        const int ih = oh * 2 + t3 - 1, iw = ow * 2 + u3 - 1;
                        const bool okj = kk < 27 && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
                        f[j] = okj ? x[(((size_t)n * 3 + c3) * a.Hi + ih) * a.Wi + iw] : 0.f;
                    }
                    v0[i] = pack8(f);
                }
                continue;
            }
            if (par2) {
                if (m < a.Mc && kok) {
                    const int w2 = a.Wo >> 1, hw2 = (a.Ho >> 1) * w2;
                    const int n = igemm_fdiv(m, hw2, a.rcp_hw), rem = m - n * hw2;
                    const int oh2 = igemm_fdiv(rem, w2, a.rcp_wo), ow2 = rem - oh2 * w2;
                    const int ih = (2 * oh2 + c_ph + 1 - th) >> 1, iw = (2 * ow2 + c_pw + 1 - tw) >> 1;   // exact: parity matches
                    if (ih < a.Hi && iw < a.Wi) {
                        const size_t src = (((size_t)n * a.Hi + ih) * a.Wi + iw) * a.Ci + ci;
                        okm |= 1u << i;
                        v0[i] = *(const uint4*)((const uint16_t*)a.grad.g + src);
                        if (has_y) v1[i] = *(const uint4*)((const uint16_t*)a.grad.y + src);
                    }
                }
                continue;
            }
            if (m < a.M && kok) {
                size_t src;
                bool inb = true;
                if (a.is_pw) {
                    src = (size_t)m * a.Ci + k;
                } else {
                    const int hw = a.Ho * a.Wo;
                    const int n = igemm_fdiv(m, hw, a.rcp_hw), rem = m - n * hw;
                    const int oh = igemm_fdiv(rem, a.Wo, a.rcp_wo), ow = rem - oh * a.Wo;
                    int ih, iw;
                    if (MODE == 0) {
                        ih = oh * a.stride + th - a.pad;
                        iw = ow * a.stride + tw - a.pad;
                    } else {   // transposed geometry: dy pixel that this forward-input pixel fed through tap
                        const int yh = oh + a.pad - th, yw = ow + a.pad - tw;
                        inb = (yh >= 0) && (yw >= 0) && (yh % a.stride == 0) && (yw % a.stride == 0);
                        ih = yh / a.stride;
                        iw = yw / a.stride;
                    }
                    inb = inb && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
                    src = (((size_t)n * a.Hi + ih) * a.Wi + iw) * a.Ci + ci;
                }
                if (inb) {
                    okm |= 1u << i;
                    if (MODE != 1) {
                        v0[i] = *(const uint4*)((const uint16_t*)a.act.data + src);
                    } else {
                        v0[i] = *(const uint4*)((const uint16_t*)a.grad.g + src);
                        if (has_y) v1[i] = *(const uint4*)((const uint16_t*)a.grad.y + src);
                    }
                }
            }
        }
    };
    auto store_a = [&](int k0) {
        const int kc8 = tid & (kc8n - 1);
        if (MODE != 2 && has_coef) {
            const int k = k0 + kc8 * 8;
            int cc = k;
            if (a.taps != 1) cc = k - (k / a.Ci) * a.Ci;       // (class-space k for stride-2 dgrad: same formula)
            float cf[CROWS][8];
#pragma unroll
            for (int r = 0; r < CROWS; ++r) {
                *(float4*)&cf[r][0] = *(const float4*)(lds_coef + r * ccols + cc);
                *(float4*)&cf[r][4] = *(const float4*)(lds_coef + r * ccols + cc + 4);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (!((okm >> i) & 1u)) continue;
                if (MODE == 0) {
                    if constexpr (GATE) {
                        const bool hi = (gsel >> i) & 1u;
                        float gg[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) gg[j] = hi ? gq1[j] : gq0[j];
                        v0[i] = act8g(v0[i], cf[0], cf[1], gg);
                    }
                    else v0[i] = act8(v0[i], cf[0], cf[1]);
                } else if (MODE == 1) {
                    float o[8];
                    dy8(v0[i], v1[i], cf[0], cf[1], cf[2 % CROWS], cf[3 % CROWS], cf[4 % CROWS], o);
                    v0[i] = pack8(o);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int p = (tid >> KSH) + (256 >> KSH) * i;
            *(uint4*)(lds_a + p * ldk + kc8 * 8) = v0[i];
        }
    };

    // per-lane epilogue constants: the 4 consecutive output channels this lane owns in each cout tile
    float bias_r[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = n0 + nt * 16 + lg * 4 + r;
            bias_r[nt][r] = (MODE != 1 && a.bias && co < a.Co) ? a.bias[co] : 0.f;   // dgrad has no bias term
        }
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    // fused BN-backward reduce (MODE 1): per-channel (s, t, invstd, -mean*invstd) of the target layer in LDS
    const bool do_red = (MODE == 1) && a.red_y != nullptr;
    if (do_red) {
        for (int i = tid; i < 4 * NT * 16; i += 256) {
            const int r = i / (NT * 16), co = n0 + i % (NT * 16);
            float v = 0.f;
            if (co < a.Co) {
                if (r == 0) v = a.red_bn[0 * a.Co + co];
                else if (r == 1) v = a.red_bn[1 * a.Co + co];
                else if (r == 2) v = a.red_bn[6 * a.Co + co];
                else v = -a.red_bn[5 * a.Co + co] * a.red_bn[6 * a.Co + co];
            }
            lds_redc[i] = v;
        }
    }

    load_coefs();
    const bool wloop = nkc > 1 || par2;      // weights re-staged inside the K loop (several chunks, or per-class taps)
    if (!wloop) {         // weights are tile-invariant: stage once
        load_w(0);
        store_w();
    }

    const int ntiles = par2 ? 4 * a.tpc : (a.M + BP - 1) / BP;
    // local pixel index (inside the tile's parity class for stride-2 dgrad) -> output pixel, or -1
    int e_ph = 0, e_pw = 0;
    auto out_pixel = [&](int ml) -> int {
        if constexpr (!par2) return ml < a.M ? ml : -1;
        if (ml >= a.Mc) return -1;
        const int w2 = a.Wo >> 1, hw2 = (a.Ho >> 1) * w2;
        const int n = igemm_fdiv(ml, hw2, a.rcp_hw), rem = ml - n * hw2;
        const int oh2 = igemm_fdiv(rem, w2, a.rcp_wo), ow2 = rem - oh2 * w2;
        return (n * a.Ho + 2 * oh2 + e_ph) * a.Wo + 2 * ow2 + e_pw;
    };
    if (PIPE && (int)blockIdx.x < ntiles) {
        load_a(blockIdx.x, 0);
        if (wloop) load_w(0);
    }
    // tile order of this workgroup.  Parity-class form: the four classes of the SAME pixel block back to back (class tile q of
    // class c covers output rows 2*oh2+ph, columns 2*ow2+pw of one region), so the 32-byte pieces the classes interleave into
    // each output line (and read from the fused-reduce target) meet in this CU's L2 within microseconds instead of being
    // written by four workgroups on four XCDs at four different times.
    auto tile_at = [&](int j) -> int {
        if constexpr (par2) {
            const int q = blockIdx.x + (j >> 2) * gridDim.x;
            return q < a.tpc ? (j & 3) * a.tpc + q : ntiles;
        } else {
            return blockIdx.x + j * gridDim.x;
        }
    };
    for (int jt = 0, t = tile_at(0); t < ntiles; t = tile_at(++jt)) {
        int tile0_ = t * BP, t_nkc_ = nkc, t_kpad_ = a.Kpad;
        if constexpr (par2) {
            const int cls = t / a.tpc;
            e_ph = cls >> 1; e_pw = cls & 1;
            tile0_ = (t - cls * a.tpc) * BP;
            t_kpad_ = ((e_ph ? 2 : 1) * (e_pw ? 2 : 1) * a.Ci + 31) & ~31;
            t_nkc_ = (t_kpad_ + kch - 1) / kch;
        }
        const int tile0 = tile0_;
        const int t_nkc = par2 ? t_nkc_ : nkc, t_kpad = par2 ? t_kpad_ : a.Kpad;
        f32x4_t acc[PT][NT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[pt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // fused reduce: fetch the target layer's raw outputs for this tile's fragments now; they land under the K loop
        uint2 ypre[PT][NT];
        if (do_red) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                int m = tile0 + (wave * PT + pt) * 16 + l15;
                if constexpr (par2) m = out_pixel(m);
                else if (m >= a.M) m = -1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int co = n0 + nt * 16 + lg * 4;
                    ypre[pt][nt] = make_uint2(0, 0);
                    if (m >= 0 && co < a.Co) ypre[pt][nt] = *(const uint2*)((const uint16_t*)a.red_y + (size_t)m * a.Co + co);
                }
            }
        }
        for (int kc = 0; kc < t_nkc; ++kc) {
            const int k0 = kc * kch;
            __syncthreads();                       // previous chunk's fragments consumed (first pass: coefficients visible)
            if (!PIPE) {
                load_a(t, k0);
                if (wloop) load_w(k0);
            }
            store_a(k0);
            if (wloop) store_w();
            __syncthreads();
            // next (tile, chunk) of this workgroup: its loads fly under the MFMAs (and the epilogue / next ypre fetch)
            if (PIPE) {
                int nt_ = t, nk_ = kc + 1;
                if (nk_ == t_nkc) { nk_ = 0; nt_ = tile_at(jt + 1); }
                if (nt_ < ntiles) {
                    load_a(nt_, nk_ * kch);
                    if (wloop) load_w(nk_ * kch);
                }
            }
            const int ksteps = min(kch, t_kpad - k0) >> 5;
            for (int ks = 0; ks < ksteps; ++ks) {
                bf16x8_t bfrag[PT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    bfrag[pt] = *(const bf16x8_t*)(lds_a + ((wave * PT + pt) * 16 + l15) * ldk + ks * 32 + lg * 8);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8_t afrag = *(const bf16x8_t*)(lds_w + (nt * 16 + l15) * ldk + ks * 32 + lg * 8);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        acc[pt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[pt], acc[pt][nt], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: lane holds couts n0+nt*16+lg*4+{0..3} of pixel tile0+(wave*PT+pt)*16+l15
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            int m = tile0 + (wave * PT + pt) * 16 + l15;
            if constexpr (par2) {
                m = out_pixel(m);
                if (m < 0) continue;
            } else {
                if (m >= a.M) continue;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = n0 + nt * 16 + lg * 4;
                if (co >= a.Co) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[pt][nt][r] + bias_r[nt][r];
                const size_t o = (size_t)m * a.Co + co;
                if (a.resid) {
                    const uint2 rv = *(const uint2*)((const uint16_t*)a.resid + o);
                    v[0] += bf_lo(rv.x); v[1] += bf_hi(rv.x); v[2] += bf_lo(rv.y); v[3] += bf_hi(rv.y);
                }
                if (MODE != 1) {                             // channel pairs in float2 (v_pk_fma_f32)
                    mnas_stat2((mnas_f2){v[0], v[1]}, &s1[nt][0], &s2[nt][0]);
                    mnas_stat2((mnas_f2){v[2], v[3]}, &s1[nt][2], &s2[nt][2]);
                }
                uint2 pk;
                pk.x = pack_bf16(v[0], v[1]);
                pk.y = pack_bf16(v[2], v[3]);
                st_u2((uint16_t*)a.out + o, pk, a.nt);
                if (do_red) {
                    // fused BN-backward reduce for the layer whose activated output this gradient belongs to:
                    // dz = g*[s*y+t>0] (g as stored, i.e. bf16-rounded), xhat = y*invstd - mean*invstd
                    const uint2 yv = ypre[pt][nt];
                    const int cl = nt * 16 + lg * 4;
                    const float4 cs = *(const float4*)(lds_redc + cl), ct = *(const float4*)(lds_redc + NT * 16 + cl);
                    const float4 ci = *(const float4*)(lds_redc + 2 * NT * 16 + cl), cm = *(const float4*)(lds_redc + 3 * NT * 16 + cl);
                    mnas_red2(pk.x, yv.x, (mnas_f2){cs.x, cs.y}, (mnas_f2){ct.x, ct.y}, (mnas_f2){ci.x, ci.y}, (mnas_f2){cm.x, cm.y},
                              &s1[nt][0], &s2[nt][0]);
                    mnas_red2(pk.y, yv.y, (mnas_f2){cs.z, cs.w}, (mnas_f2){ct.z, ct.w}, (mnas_f2){ci.z, ci.w}, (mnas_f2){cm.z, cm.w},
                              &s1[nt][2], &s2[nt][2]);
                }
            }
        }
    }

    if ((MODE != 1 || do_red) && a.stats) {
        // deterministic workgroup reduction: 16-lane shuffle tree, one LDS slot per (wave, channel), waves summed in
        // fixed order (no float atomics: results are bit-reproducible run to run)
        float* red4 = (float*)lds_a;                    // [4 waves][2][NT*16], the activation tile is dead by now
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x1 = s1[nt][r], x2 = s2[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
                if (l15 == 0) {
                    red4[(wave * 2 + 0) * NT * 16 + nt * 16 + lg * 4 + r] = x1;
                    red4[(wave * 2 + 1) * NT * 16 + nt * 16 + lg * 4 + r] = x2;
                }
            }
        __syncthreads();
        for (int i = tid; i < 2 * NT * 16; i += 256) {
            const int r = i / (NT * 16), cl = i % (NT * 16), c = n0 + cl;
            const float v = ((red4[(0 * 2 + r) * NT * 16 + cl] + red4[(1 * 2 + r) * NT * 16 + cl]) +
                             red4[(2 * 2 + r) * NT * 16 + cl]) + red4[(3 * 2 + r) * NT * 16 + cl];
            if (c < a.Co) a.stats[((size_t)r * a.Co + c) * gridDim.x + blockIdx.x] = v;   // [2][Co][P]
        }
    }
}

template <int MODE, int NT, int PT, int KCH>
static int launch_igemm_k(const IgemmArgs& a, int nparts, int nblocks, bool pipe, hipStream_t stream) {
    constexpr int CROWS = (MODE == 1) ? 5 : 2;
    const int ccols = (a.taps == 1) ? a.Kpad : a.Ci;
    const size_t lds = (size_t)(CROWS * ccols + 6 * NT * 16) * sizeof(float) +
                       (size_t)(NT * 16 + 64 * PT) * (a.kch + 8) * 2;
    if (lds > 160 * 1024) return MNAS_EINVAL;
    if constexpr (MODE == 1) {
        if (a.s2) {       // parity-class tiling: never pipelined (launch_igemm), own instantiation
            hipLaunchKernelGGL((k_igemm<MODE, NT, PT, KCH, false, true>), dim3(nparts, nblocks), dim3(256), lds, stream, a);
            MNAS_CHECK_LAUNCH();
            return MNAS_OK;
        }
    }
    if (a.gate) {       // gated act-on-load: the narrow 1x1 forward shapes only (the squeeze-excite project convs, K < 192)
        if constexpr (MODE == 0 && NT <= 3 && KCH <= 64) {
            if (a.Ho * a.Wo < 64 * PT) return MNAS_EINVAL;       // a tile may span two images, not three
            if (pipe) hipLaunchKernelGGL((k_igemm<MODE, NT, PT, KCH, true, false, true>), dim3(nparts, nblocks), dim3(256), lds, stream, a);
            else hipLaunchKernelGGL((k_igemm<MODE, NT, PT, KCH, false, false, true>), dim3(nparts, nblocks), dim3(256), lds, stream, a);
            MNAS_CHECK_LAUNCH();
            return MNAS_OK;
        }
        return MNAS_EINVAL;
    }
    if (pipe) hipLaunchKernelGGL((k_igemm<MODE, NT, PT, KCH, true, false>), dim3(nparts, nblocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((k_igemm<MODE, NT, PT, KCH, false, false>), dim3(nparts, nblocks), dim3(256), lds, stream, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
template <int MODE, int NT, int PT>
static int launch_igemm(const IgemmArgs& a_in, int nparts, int nblocks, hipStream_t stream) {
    IgemmArgs a = a_in;
    if (a.s2) a.tpc = (a.Mc + 64 * PT - 1) / (64 * PT);
    // Software pipelining (global loads of the next (tile, chunk) in flight under the MFMA phase) costs 30-50 VGPRs.
    // It pays when a workgroup walks several load phases (long K, persistent tile loops) and the extra registers do
    // not halve the occupancy; measured per layer shape on MI355X (igemm table).
    const int ntiles = a.s2 ? 4 * a.tpc : (a.M + 64 * PT - 1) / (64 * PT);
    const int nkc = (a.Kpad + a.kch - 1) / a.kch;
    const int phases = nkc * ((ntiles + nparts - 1) / nparts);
    const bool s2 = a.taps > 1 && a.stride == 2;
    bool pipe;
    if (MODE == 0) pipe = phases >= 4 && !(s2 && PT == 2) && !(NT >= 6 && PT == 2);
    else pipe = phases >= 2 && nkc > 1 && !s2;
    // long reductions on small pixel counts (the 14x14 / 7x7 stages, dense 3x3): 128-wide K chunks put 4x more
    // bytes in flight per staging pass (these launches are latency-bound, not bandwidth-bound)
    if constexpr (PT == 1 && (NT == 2 || NT == 3 || NT == 4 || NT == 6)) {
        if (a.kch == 128) return launch_igemm_k<MODE, NT, PT, 128>(a, nparts, nblocks, pipe, stream);
    }
    return a.kch >= 64 ? launch_igemm_k<MODE, NT, PT, 64>(a, nparts, nblocks, pipe, stream)
                       : launch_igemm_k<MODE, NT, PT, 32>(a, nparts, nblocks, pipe, stream);
}

// cout tiling: NT*16 channels per workgroup column.  Up to 8 tiles: one block (the activation tile is staged
// once).  Wider outputs: several blocks of 4 or 3 tiles, fewest padded tiles first (NT = 6/8 blocks need 150-250
// VGPRs; measured 15-25 % slower on the 14x14 / 7x7 expand-type layers than re-staging the small-K activation tile;
// with K >= 512 re-staging costs more than the registers and 6-tile blocks win by 5-20 %).
// Pixel tiling: PT=2 (128-pixel tiles) when there are enough tiles to fill the chip twice over.
static void igemm_tiling(int Co, int Kpad, int M, int* nt_out, int* nblocks_out, int* pt_out) {
    const int tiles = (Co + 15) / 16;
    int best_nt = 1, best_waste = 1 << 30, best_blocks = 1 << 30;
    if (tiles <= 8) {
        static const int opts1[] = {1, 2, 3, 4, 6, 8};
        for (int nt : opts1) if (nt >= tiles) { best_nt = nt; best_blocks = 1; break; }
    } else {
        // long reductions (K >= 512: the activation tile is the expensive operand) keep 6-tile blocks
        static const int opts_small_k[3] = {4, 3, 0}, opts_large_k[3] = {6, 4, 3};
        for (int nt : (Kpad >= 512 ? opts_large_k : opts_small_k)) {
            if (nt == 0) continue;
            const int blocks = (tiles + nt - 1) / nt;
            const int waste = blocks * nt - tiles;
            if (waste < best_waste || (waste == best_waste && blocks < best_blocks)) {
                best_nt = nt; best_waste = waste; best_blocks = blocks;
            }
        }
    }
    *nt_out = best_nt;
    *nblocks_out = best_blocks;
    *pt_out = ((int64_t)M * best_blocks >= (int64_t)128 * 1024) ? 2 : 1;
}

// Pixels per tile (64 or 128) mnas_conv_gemm will use for this problem: lets the caller size `nparts` (persistent
// workgroups along the pixel dimension) in whole tiles.  M = output pixels, K = kh*kw*Ci of the mode's reduction.
extern "C" int mnas_conv_gemm_tile_pixels(int M, int Co, int K) {
    if (M < 1 || Co < 1 || K < 1) return -1;
    int nt, nblocks, pt;
    igemm_tiling(Co, (K + 31) / 32 * 32, M, &nt, &nblocks, &pt);
    return 64 * pt;
}

// Preferred number of pixel-workgroups (nparts) for a launch: > 0 for the kernels that size their own persistent grid
// (the DMA-pipelined 1x1 forward), -1 = "caller's choice" (k_igemm: whole tiles, see mnas_conv_gemm_tile_pixels).
extern "C" int mnas_conv_gemm_parts(int mode, int M, int Ci, int Co, int taps) {
    if (mode == 0 && taps == 1) {
        const int p = mnas_pwx_parts(M, Ci, Co);
        if (p > 0) return p;
    }
    if (taps == 1) {
        const int p = mnas_pws_parts(mode, M, Ci, Co);      // Ci = reduction length of the mode (dy channels for mode 1)
        if (p > 0) return p;
    }
    if (mode == 0 && taps == 1 && mnas_pwf_enabled()) return mnas_pwf_parts(M, Ci, Co);
    if (mode == 1 && taps == 1 && mnas_pwd_enabled()) return mnas_pwd_parts(M, Ci, Co);
    return -1;
}

// Shapes whose 1x1 forward takes MnasConvGemm.gate: the streaming kernel with short per-wave K slices (K >= 192), or k_igemm
// with at most three output tiles.
extern "C" int mnas_conv_gemm_gate_ok(int N, int HW, int Ci, int Co) {
    if (N < 1 || HW < 1 || (Ci & 7) || (Co & 7) || Ci < 8 || Co < 8) return 0;
    const int M = N * HW;
    if (mnas_pws_parts(0, M, Ci, Co) > 0) return mnas_pws_gate_ok(M, Ci, Co);
    int nt, nblocks, pt;
    igemm_tiling(Co, (Ci + 31) / 32 * 32, M, &nt, &nblocks, &pt);
    return (nt <= 3 && nblocks == 1 && !(pt == 1 && Ci >= 256) && HW >= 64 * pt) ? 1 : 0;
}

extern "C" int mnas_conv_gemm(const MnasConvGemm* c, void* stream) {
    if (!c || (c->mode != 0 && c->mode != 1)) return MNAS_EINVAL;
    if ((c->Ci & 7) || (c->Co & 7) || c->nparts < 1 || c->nparts > 65535) return MNAS_EINVAL;
    IgemmArgs a;
    a.M = c->N * c->Ho * c->Wo;
    a.Hi = c->Hi; a.Wi = c->Wi; a.Ci = c->Ci; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co;
    a.kh = c->kh; a.kw = c->kw; a.stride = c->stride; a.pad = c->pad;
    a.taps = c->kh * c->kw;
    a.Ktot = a.taps * c->Ci;
    a.Kpad = (a.Ktot + 31) / 32 * 32;
    a.kch = a.Kpad >= 64 ? 64 : 32;
    a.is_pw = (a.taps == 1 && c->stride == 1 && c->pad == 0) ? 1 : 0;
    a.co_pad16 = (c->Co + 15) / 16 * 16;
    a.act = c->act; a.grad = c->grad;
    a.w = (const uint16_t*)c->w; a.bias = c->bias; a.resid = c->resid; a.out = c->out; a.stats = c->stats;
    a.red_y = (c->mode == 1) ? c->red_y : nullptr; a.red_bn = c->red_bn;
    a.nt = (mnas_nt_mask() & (c->mode == 1 ? MNAS_NT_IGEMM_DGRAD : MNAS_NT_IGEMM_FWD)) ? 1 : 0;
    a.gate = c->gate;
    if (a.gate && (c->mode != 0 || !a.is_pw || !c->act.scale || c->resid)) return MNAS_EINVAL;
    {   // the decode divides by the OUTPUT plane (half plane for the parity-class form); exact division beyond 2^24 pixels
        const int s2f = (c->mode == 1 && c->kh == 3 && c->kw == 3 && c->stride == 2 && c->pad == 1 && !(c->Ho & 1) && !(c->Wo & 1));
        const int wd = s2f ? c->Wo / 2 : c->Wo, hwd = s2f ? (c->Ho / 2) * wd : c->Ho * c->Wo;
        const bool small = (int64_t)a.M < (1 << 24);
        a.rcp_hw = small ? 1.0f / (float)hwd : 0.f;
        a.rcp_wo = small ? 1.0f / (float)wd : 0.f;
        a.rcp_ci = 1.0f / (float)c->Ci;
    }
    // stride-2 3x3 input gradient over even output sizes: parity-class tiling (see k_igemm)
    a.s2 = (c->mode == 1 && c->kh == 3 && c->kw == 3 && c->stride == 2 && c->pad == 1 && !(c->Ho & 1) && !(c->Wo & 1)) ? 1 : 0;
    a.Mc = c->N * (c->Ho / 2) * (c->Wo / 2);
    a.tpc = 0;
    if (a.red_y && (!a.red_bn || !a.stats)) return MNAS_EINVAL;
    if (a.taps != 1 && c->Ci > 1024) return MNAS_EINVAL;
    if (a.taps == 1 && !a.is_pw) return MNAS_EINVAL;   // strided / padded 1x1 does not occur in this network
    if (c->mode == 0 && !c->act.data) return MNAS_EINVAL;
    if (c->mode == 1 && (!c->grad.g || (!c->grad.y) != (!c->grad.coef))) return MNAS_EINVAL;    // (y, coef) both or neither
    if (c->mode == 0 && a.is_pw && !c->resid && !c->gate && mnas_pwx_parts(a.M, c->Ci, c->Co) > 0)
        return mnas_pwx_forward(c, stream);
    if (a.is_pw && (c->mode == 1 || !c->resid) && mnas_pws_parts(c->mode, a.M, c->Ci, c->Co) > 0) {
        const int rc = mnas_pws_run(c, stream);          // MNAS_EINVAL: not that kernel's case (a materialised dy): fall through
        if (rc != MNAS_EINVAL) return rc;
    }
    if (c->mode == 0 && a.is_pw && !c->resid && !c->gate && mnas_pwf_enabled() && mnas_pwf_parts(a.M, c->Ci, c->Co) > 0)
        return mnas_pwf_forward(c, stream);
    if (c->mode == 1 && a.is_pw && !c->resid && !c->bias && mnas_pwd_enabled() && mnas_pwd_parts(a.M, c->Ci, c->Co) > 0)
        return mnas_pwd_dgrad(c, stream);

    // weight-heavy dense 3x3 on the 7x7 planes: weight slices register-resident, images streamed (csrc/mnas_c3r.hip)
    if (a.taps == 9 && !c->resid && !(c->mode == 1 && c->grad.y) &&
        mnas_c3r_parts(c->mode, c->N, c->Hi, c->Wi, c->Ci, c->Ho, c->Wo, c->Co, c->kh, c->kw, c->stride, c->pad) > 0)
        return mnas_c3r_run(c, stream);
    // dense 3x3 on the 14x14 / 7x7 maps: whole image per workgroup (csrc/mnas_dimg.hip); MODE 1 there takes a materialised dy
    if (a.taps == 9 && !(c->mode == 0 && c->resid) && !(c->mode == 1 && (c->grad.y || c->resid)) &&
        mnas_dimg_parts(c->mode, c->N, c->Hi, c->Wi, c->Ci, c->Ho, c->Wo, c->Co, c->kh, c->kw, c->stride, c->pad) > 0)
        return mnas_dimg_run(c, stream);

    // stride-2 3x3 forward on the large maps: every wave weight-stationary, fragments gathered from global memory (csrc/mnas_c3x.hip)
    if (c->mode == 0 && a.taps == 9 && !c->resid && !c->gate &&
        mnas_c3x_ok(c->N, c->Hi, c->Wi, c->Ci, c->Ho, c->Wo, c->Co, c->kh, c->kw, c->stride, c->pad))
        return mnas_c3x_run(c, stream);

    int best_nt, nblocks, pt;
    igemm_tiling(c->Co, a.Kpad, a.M, &best_nt, &nblocks, &pt);
    if (pt == 1 && a.Kpad >= 256 && (best_nt == 2 || best_nt == 3 || best_nt == 6 || (best_nt == 4 && mnas_diag_env("MNAS_IG_K128_NT4", 1)))) a.kch = 128;
    hipStream_t s = (hipStream_t)stream;
#define MNAS_IG(MODE_, NT_) \
    if (c->mode == MODE_ && best_nt == NT_) return pt == 2 ? launch_igemm<MODE_, NT_, 2>(a, c->nparts, nblocks, s) \
                                                           : launch_igemm<MODE_, NT_, 1>(a, c->nparts, nblocks, s);
    MNAS_IG(0, 1) MNAS_IG(0, 2) MNAS_IG(0, 3) MNAS_IG(0, 4) MNAS_IG(0, 6) MNAS_IG(0, 8)
    MNAS_IG(1, 1) MNAS_IG(1, 2) MNAS_IG(1, 3) MNAS_IG(1, 4) MNAS_IG(1, 6) MNAS_IG(1, 8)
#undef MNAS_IG
    return MNAS_EINVAL;
}

// Stem: dense 3x3 stride-2 conv on the fp32 NCHW image (mnasnet.py:179) = the same GEMM with an im2col
// staging mode (K = 27 padded to 32, image values rounded to bf16 like every other activation).
extern "C" int mnas_stem_fwd(const MnasStemFwd* c, void* stream) {
    if (!c || (c->Co & 7) || c->Co > 128 || c->nparts < 1) return MNAS_EINVAL;
    {   // 32 couts, W % 4 == 0: band kernel (csrc/mnas_stem.hip); anything else: im2col staging below
        const int rc = mnas_stem_fwd_band(c, stream);
        if (rc != MNAS_EINVAL) return rc;
    }
    if (c->in_affine || c->in_u8) return MNAS_EINVAL;               // the fused input pipeline exists in the band kernels only
    IgemmArgs a;
    a.M = c->N * c->Ho * c->Wo;
    a.Hi = c->H; a.Wi = c->W; a.Ci = 27; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co;
    a.kh = 3; a.kw = 3; a.stride = 2; a.pad = 1;
    a.taps = 1; a.Ktot = 27; a.Kpad = 32; a.kch = 32; a.is_pw = 0;
    a.co_pad16 = (c->Co + 15) / 16 * 16;
    a.act.data = c->x; a.act.scale = nullptr; a.act.shift = nullptr;
    a.grad.g = nullptr; a.grad.y = nullptr; a.grad.coef = nullptr;
    a.w = (const uint16_t*)c->w; a.bias = c->bias; a.resid = nullptr; a.out = c->out; a.stats = c->stats;
    a.red_y = nullptr; a.red_bn = nullptr; a.gate = nullptr;
    a.nt = (mnas_nt_mask() & MNAS_NT_STEM) ? 1 : 0;
    a.rcp_hw = (int64_t)a.M < (1 << 24) ? 1.0f / (float)(c->Ho * c->Wo) : 0.f;
    a.rcp_wo = (int64_t)a.M < (1 << 24) ? 1.0f / (float)c->Wo : 0.f;
    a.rcp_ci = 0.f;
    a.s2 = 0; a.Mc = 0; a.tpc = 0;
    const int tiles = (c->Co + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    if (tiles == 1) return launch_igemm_k<2, 1, 2, 32>(a, c->nparts, 1, true, s);
    if (tiles == 2) return launch_igemm_k<2, 2, 2, 32>(a, c->nparts, 1, true, s);
    if (tiles <= 4) return launch_igemm_k<2, 4, 2, 32>(a, c->nparts, 1, true, s);
    return launch_igemm_k<2, 8, 2, 32>(a, c->nparts, 1, true, s);
}
