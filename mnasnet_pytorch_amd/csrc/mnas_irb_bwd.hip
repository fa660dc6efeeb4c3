// Backward of the fused inverted-residual block (MBConv_block, mnasnet.py:105-137; autograd mirror, train.py:439) on the small
// feature maps.  Forward and the shape rules: mnas_irb.hip.  Three launches, separated by the BatchNorm-backward reductions
// (each needs sums over the whole batch before its dy can be formed):
//
//   k_irb_bwd_proj  G, y3, y2            -> dy3 (materialised, small), dW3 partials, BatchNorm2-backward sums
//                   the project conv's input gradient da2 = dy3 . W3 is formed on the matrix cores ONLY to reduce it
//                   (sum dz2, sum dz2*xhat2): the t-times expanded gradient g2 is never written;
//   k_irb_bwd_dw    dy3, y2, x           -> g1 (= dz1: the masked gradient of the expand conv's activated output), depthwise dW
//                   partials, P = dz1^T x partials, BatchNorm1-backward sums
//                   RECOMPUTES da2 = dy3 . W3 and y1 = W1 . act(x) on the matrix cores instead of reading g2 / y1; the depthwise
//                   input- and weight-gradient run as one sweep over two zero-padded LDS images (dy2, a1);
//   k_irb_bwd_exp   g1, x, G             -> dx = dy1 . W1 + G with dy1 = c1*dz1 + c2*y1 + c3, y1 recomputed; the two GEMMs are
//                   chained in registers (the D layout of the first is the B operand of the second under a row permutation
//                   of the streamed weight chunk).
//   k_irb_w1_fin    the expand conv's weight gradient WITHOUT a pass over dy1: dy1 is affine in (dz1, y1) per channel, so
//                   dW1[e][c] = c1[e]*P[e][c] + c2[e]*(W1 G + b1 Sx^T)[e][c] + c3[e]*Sx[c], with G = sum a a^T and Sx = sum a of the
//                   block input, already known from the forward's statistics pass (mnas_gram).
// HBM traffic of the wide tensors per block application: y2 read twice, g1 written once and read once (unfused: nine passes).
// Bit-reproducible: every partial table is written by exactly one workgroup and reduced in a fixed order.
#include "mnas_common.h"

typedef __attribute__((ext_vector_type(4))) short ib_s4_t;
typedef __attribute__((address_space(3))) ib_s4_t* ib_lds_s4_ptr;

// 16 (column col0 + lane&15) x 32 (rows row0 + (lane>>4)*8 + 0..7) fragment of a row-major bf16 LDS tile: the MFMA operand whose
// reduction index runs over the ROWS (pixels) -- gfx950 LDS transpose read, as in mnas_wgrad.hip
__device__ __forceinline__ bf16x8_t ib_tr_frag(const uint16_t* tile, int ld, int row0, int col0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const uint16_t* p = tile + (row0 + g * 8 + (i >> 2)) * ld + col0 + (i & 3) * 4;
    const ib_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ib_lds_s4_ptr)p);
    const ib_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ib_lds_s4_ptr)(p + 4 * ld));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float ib_bf16r(float v) { return bf_lo(pack_bf16(v, 0.f)); }

struct IrbGeom {
    int N, H, W, C, E;
    int HW, NI;                 // pixels per image, images per pass
    int Kpad;                   // C rounded up to 32
    int npt, npk;               // 16-pixel tiles / 32-pixel reduction steps per pass
    int ipg;                    // passes per workgroup
};

// =====================================================================================================================
// k_irb_bwd_proj
// =====================================================================================================================
struct IrbProjArgs {
    IrbGeom g;
    const uint16_t* gout;       // G: gradient of the block's activated project output (M,C)
    const uint16_t* y3;         // raw project output (M,C)
    const float* bn3;           // bnbuf rows 0..4 (s, t, c1, c2, c3)
    const uint16_t* y2;         // raw depthwise output (M,E)
    const float* bn2;           // bnbuf rows 0,1,5,6
    const uint16_t* w3t;        // MNAS_PACK_DGRAD of the project conv: [E_pad16][Kpad], k = c
    uint16_t* dy3;              // out (M,C)
    float* wpartial;            // out [gridDim.y][C][E]
    float* red2;                // out [2][E][gridDim.y]
};

// workgroup = 256 threads = one 32-channel slice of E x a group of image passes
template <int KST>
__global__ __launch_bounds__(256, 2) void k_irb_bwd_proj(IrbProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IrbGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int e0 = blockIdx.x * 32;
    const int CP = g.Kpad + 8, KP = g.npk * 32;
    constexpr int YP = 40;
    uint16_t* t_dy3 = (uint16_t*)smem;                              // [KP][CP]
    uint16_t* t_y2 = t_dy3 + KP * CP;                               // [KP][YP] raw
    uint16_t* t_w = t_y2 + KP * YP;                                 // [32][CP]
    float* tab3 = (float*)(t_w + 32 * CP);                          // [5][Kpad]
    float* tab2 = tab3 + 5 * g.Kpad;                                // [4][32]: s2, t2, invstd2, -mean2*invstd2
    float* lds_red = tab2 + 4 * 32;                                 // [4 waves][2][32]
    const int c8n = g.Kpad >> 3;

    for (int i = tid; i < 5 * g.Kpad; i += 256) {
        const int r = i / g.Kpad, c = i - r * g.Kpad;
        tab3[i] = c < g.C ? a.bn3[(size_t)r * g.C + c] : 0.f;
    }
    if (tid < 128) {
        const int r = tid >> 5, e = e0 + (tid & 31);
        float v;
        if (r == 0) v = a.bn2[e];
        else if (r == 1) v = a.bn2[g.E + e];
        else if (r == 2) v = a.bn2[6 * g.E + e];
        else v = -a.bn2[5 * g.E + e] * a.bn2[6 * g.E + e];
        tab2[tid] = v;
    }
    for (int i = tid; i < 32 * c8n; i += 256) {
        const int r = i / c8n, c8 = i - r * c8n;
        *(uint4*)(t_w + r * CP + c8 * 8) = *(const uint4*)(a.w3t + (size_t)(e0 + r) * g.Kpad + c8 * 8);
    }
    float r1[2][4], r2[2][4];
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int r = 0; r < 4; ++r) { r1[et][r] = 0.f; r2[et][r] = 0.f; }
    // weight-gradient tile pairs (ct, et) of this wave: et = wave & 1, ct = (wave >> 1) + 2j
    constexpr int NQ = KST;                                         // ct tiles per wave: Kpad/16/2 = KST
    const int wet = wave & 1;
    f32x4_t acc3[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) acc3[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const float ws2 = a.bn2[e0 + wet * 16 + l15], wt2 = a.bn2[g.E + e0 + wet * 16 + l15];
    __syncthreads();

    const int npass = (g.N + g.NI - 1) / g.NI;
    for (int pi = 0; pi < g.ipg; ++pi) {
        const int pass = blockIdx.y * g.ipg + pi;
        if (pass >= npass) break;
        const int n0 = pass * g.NI;
        const int npx = min(g.NI, g.N - n0) * g.HW;                  // valid pixels of this pass
        const size_t m0 = (size_t)n0 * g.HW;
        // ---- stage dy3 (dy-on-load of G, y3) and the raw y2 slice: every chunk of both tiles is written (zeros outside the
        // image / channels); ALL global loads of the pass are issued before the first use (one memory round trip per pass)
        {
            constexpr int NCH = 12;                                 // chunks per thread and batch (one batch for C <= 96 at 14x14, C <= 192 at 7x7)
            uint4 gv[NCH], yv[NCH], y2c[4];
            const int nchunk = KP * c8n;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = tid + 256 * j;
                const int row = q >> 2, c8 = q & 3;
                y2c[j] = make_uint4(0, 0, 0, 0);
                if (q < KP * 4 && row < npx) y2c[j] = *(const uint4*)(a.y2 + (m0 + row) * g.E + e0 + c8 * 8);
            }
          for (int q0 = 0; q0 < nchunk; q0 += 256 * NCH) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int q = q0 + tid + 256 * j;
                const int row = q / c8n, c8 = q - row * c8n;
                gv[j] = make_uint4(0, 0, 0, 0); yv[j] = make_uint4(0, 0, 0, 0);
                if (q < nchunk && row < npx && c8 * 8 < g.C) {
                    gv[j] = *(const uint4*)(a.gout + (m0 + row) * g.C + c8 * 8);
                    yv[j] = *(const uint4*)(a.y3 + (m0 + row) * g.C + c8 * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int q = q0 + tid + 256 * j;
                if (q >= nchunk) continue;
                const int row = q / c8n, c8 = q - row * c8n;
                uint4 o = make_uint4(0, 0, 0, 0);
                if (row < npx && c8 * 8 < g.C) {
                    float cf[5][8], d[8];
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        *(float4*)&cf[r][0] = *(const float4*)(tab3 + r * g.Kpad + c8 * 8);
                        *(float4*)&cf[r][4] = *(const float4*)(tab3 + r * g.Kpad + c8 * 8 + 4);
                    }
                    dy8(gv[j], yv[j], cf[0], cf[1], cf[2], cf[3], cf[4], d);
                    o = pack8(d);
                    if (blockIdx.x == 0) *(uint4*)(a.dy3 + (m0 + row) * g.C + c8 * 8) = o;
                }
                *(uint4*)(t_dy3 + row * CP + c8 * 8) = o;
            }
          }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = tid + 256 * j;
                if (q < KP * 4) *(uint4*)(t_y2 + (q >> 2) * YP + (q & 3) * 8) = y2c[j];
            }
        }
        __syncthreads();
        // ---- da2 = dy3 . W3 for the slice, reduced on the spot: D[e][pix]
        {
            bf16x8_t bfr[4][KST];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pt = wave + 4 * i;
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    bfr[i][ks] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
                    if (pt < g.npt && ks * 32 < g.Kpad) bfr[i][ks] = *(const bf16x8_t*)(t_dy3 + (pt * 16 + l15) * CP + ks * 32 + lg * 8);
                }
            }
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                bf16x8_t af[KST];
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    af[ks] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
                    if (ks * 32 < g.Kpad) af[ks] = *(const bf16x8_t*)(t_w + (et * 16 + l15) * CP + ks * 32 + lg * 8);
                }
                const float4 s2 = *(const float4*)(tab2 + et * 16 + lg * 4), t2 = *(const float4*)(tab2 + 32 + et * 16 + lg * 4);
                const float4 i2 = *(const float4*)(tab2 + 64 + et * 16 + lg * 4), m2 = *(const float4*)(tab2 + 96 + et * 16 + lg * 4);
                const float s2a[4] = {s2.x, s2.y, s2.z, s2.w}, t2a[4] = {t2.x, t2.y, t2.z, t2.w};
                const float i2a[4] = {i2.x, i2.y, i2.z, i2.w}, m2a[4] = {m2.x, m2.y, m2.z, m2.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pt = wave + 4 * i;
                    if (pt >= g.npt) break;
                    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks)
                        if (ks * 32 < g.Kpad) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bfr[i][ks], acc, 0, 0, 0);
                    const uint2 yv = *(const uint2*)(t_y2 + (pt * 16 + l15) * YP + et * 16 + lg * 4);
                    const float yq[4] = {bf_lo(yv.x), bf_hi(yv.x), bf_lo(yv.y), bf_hi(yv.y)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dz = (fmaf(yq[r], s2a[r], t2a[r]) > 0.f) ? ib_bf16r(acc[r]) : 0.f;
                        r1[et][r] += dz;
                        r2[et][r] = fmaf(dz, fmaf(yq[r], i2a[r], m2a[r]), r2[et][r]);
                    }
                }
            }
        }
        // ---- dW3[c][e] += sum_pix dy3[pix][c] * a2[pix][e]
        for (int ks = 0; ks < g.npk; ++ks) {
            const bf16x8_t yraw = ib_tr_frag(t_y2, YP, ks * 32, wet * 16, lane);
            uint4 u = *(const uint4*)&yraw;
            float f[8];
            unpack8(u, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], ws2, wt2), 0.f);
            u = pack8(f);
            const bf16x8_t bfrag = *(const bf16x8_t*)&u;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int ct = (wave >> 1) + 2 * j;
                if (ct * 16 >= g.Kpad) break;
                const bf16x8_t afrag = ib_tr_frag(t_dy3, CP, ks * 32, ct * 16, lane);
                acc3[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag, acc3[j], 0, 0, 0);
            }
        }
        __syncthreads();                                            // tiles consumed
    }
    // ---- outputs
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int ct = (wave >> 1) + 2 * j;
        if (ct * 16 >= g.Kpad) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = ct * 16 + lg * 4 + r;
            if (c < g.C) a.wpartial[((size_t)blockIdx.y * g.C + c) * g.E + e0 + wet * 16 + l15] = acc3[j][r];
        }
    }
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x1 = r1[et][r], x2 = r2[et][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
            if (l15 == 0) {
                lds_red[(wave * 2 + 0) * 32 + et * 16 + lg * 4 + r] = x1;
                lds_red[(wave * 2 + 1) * 32 + et * 16 + lg * 4 + r] = x2;
            }
        }
    __syncthreads();
    if (tid < 64) {
        const int r = tid >> 5, c = tid & 31;
        const float v = ((lds_red[(0 * 2 + r) * 32 + c] + lds_red[(1 * 2 + r) * 32 + c]) + lds_red[(2 * 2 + r) * 32 + c]) +
                        lds_red[(3 * 2 + r) * 32 + c];
        a.red2[((size_t)r * g.E + e0 + c) * gridDim.y + blockIdx.y] = v;
    }
}

// =====================================================================================================================
// k_irb_bwd_dw
// =====================================================================================================================
struct IrbDwArgs {
    IrbGeom g;
    MnasActIn x;
    const uint16_t* dy3;        // materialised by k_irb_bwd_proj (M,C)
    const uint16_t* y2;
    const uint16_t* w1;         // MNAS_PACK_FWD expand weights [E_pad16][Kpad]
    const uint16_t* w3t;        // MNAS_PACK_DGRAD project weights [E_pad16][Kpad]
    const float* b1;
    const float* bn1;           // rows 0,1 (s,t), 5,6 (mean, invstd)
    const float* bn2;           // rows 0..4 (s,t,c1,c2,c3)
    const float* wdw;           // [k*k][E]
    uint16_t* g1;               // out (M,E): dz1
    float* dwpartial;           // out [gridDim.y][k*k][E]
    float* ppartial;            // out [gridDim.y][E][C]
    float* red1;                // out [2][E][gridDim.y]
    int lds_bytes;              // dynamic LDS of the launch (zeroed once: see the x tile note in the kernel)
};

template <int KS, int WW, int KST>
__global__ __launch_bounds__(256, 2) void k_irb_bwd_dw(IrbDwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IrbGeom& g = a.g;
    constexpr int P = KS / 2;
    constexpr int RW = (WW + 2 * P) | 1;
    constexpr int YP = 40;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int e0 = blockIdx.x * 32;
    const int RH = g.H + 2 * P;
    const int CP = g.Kpad + 8, KP = g.npk * 32;
    const int img_dw = g.NI * RH * RW * 16;                          // dwords of one padded image set
    // region R: the two padded images, re-used after the depthwise sweep as the x tile [NI*HW rows][CP] of the P GEMM.  The
    // GEMM's last 32-pixel step reads up to KP rows: the rows beyond the written ones hold other FINITE bf16 data of this
    // workgroup (end of R, start of t_y1) and meet zero rows of dz1, i.e. contribute exactly 0.
    const int xs_dw = (g.NI * g.HW * CP + 1) >> 1;
    const int reg_dw = (max(2 * img_dw, xs_dw) + 3) & ~3;
    uint32_t* a1p = (uint32_t*)smem;                                // [img_dw]   } region R
    uint32_t* dy2p = a1p + img_dw;                                  // [img_dw]   }
    uint16_t* xs = (uint16_t*)smem;
    uint16_t* t_y1 = (uint16_t*)(a1p + reg_dw);                     // [KP][YP]: y1 (bf16), then dz1 in place
    uint16_t* t_w1 = t_y1 + KP * YP;                                // [32][CP]
    uint16_t* t_w3 = t_w1 + 32 * CP;                                // [32][CP]
    float* tabx = (float*)(t_w3 + 32 * CP);                         // [2][Kpad]
    float* tab1 = tabx + 2 * g.Kpad;                                // [5][32]: b1, s1, t1, invstd1, -mean1*invstd1
    float* tab2 = tab1 + 5 * 32;                                    // [5][32]: s2, t2, c1, c2, c3
    float* tabw = tab2 + 5 * 32;                                    // [KS*KS][32] depthwise weights
    float* lds_red = (float*)smem;                                  // [16][KS][32]: over region R, after the last pass
    const int c8n = g.Kpad >> 3;
    const bool hasx = a.x.scale != nullptr;

    // every byte of the workgroup's LDS starts as zero: the P GEMM's last 32-pixel step reads x-tile rows that alias whatever
    // follows region R (t_y1, the PAD columns of the weight tiles): they meet zero rows of dz1, so they only have to be finite --
    // uninitialised LDS (another kernel's leftovers) is not
    for (int i = tid; i < (a.lds_bytes >> 4); i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int i = tid; i < 2 * g.Kpad; i += 256) {
        const int c = i < g.Kpad ? i : i - g.Kpad;
        tabx[i] = (hasx && c < g.C) ? (i < g.Kpad ? a.x.scale[c] : a.x.shift[c]) : 0.f;
    }
    for (int i = tid; i < 5 * 32; i += 256) {
        const int r = i >> 5, e = e0 + (i & 31);
        float v;
        if (r == 0) v = a.b1 ? a.b1[e] : 0.f;
        else if (r == 1) v = a.bn1[e];
        else if (r == 2) v = a.bn1[g.E + e];
        else if (r == 3) v = a.bn1[6 * g.E + e];
        else v = -a.bn1[5 * g.E + e] * a.bn1[6 * g.E + e];
        tab1[i] = v;
        tab2[i] = a.bn2[(size_t)r * g.E + e];
    }
    for (int i = tid; i < KS * KS * 32; i += 256) tabw[i] = a.wdw[(size_t)(i >> 5) * g.E + e0 + (i & 31)];
    for (int i = tid; i < 32 * c8n; i += 256) {
        const int r = i / c8n, c8 = i - r * c8n;
        *(uint4*)(t_w1 + r * CP + c8 * 8) = *(const uint4*)(a.w1 + (size_t)(e0 + r) * g.Kpad + c8 * 8);
        *(uint4*)(t_w3 + r * CP + c8 * 8) = *(const uint4*)(a.w3t + (size_t)(e0 + r) * g.Kpad + c8 * 8);
    }
    // (t_y1 rows beyond the image stay zero for the whole kernel; the x tile's rows of a SHORT last pass are not rewritten and
    // keep finite values)

    // MFMA-phase pixels of this lane
    int poff[4], ppix[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pt = wave + 4 * i;
        const int p = pt * 16 + l15;
        const bool ok = pt < g.npt && p < g.NI * g.HW;
        const int im = ok ? p / g.HW : 0, pp = ok ? p - im * g.HW : 0;
        const int py = pp / g.W, px = pp - py * g.W;
        ppix[i] = ok ? p : -1;
        poff[i] = ((im * RH + py + P) * RW + px + P) * 16;
    }
    // depthwise task
    const int pair = tid & 15, slot = tid >> 4;
    const bool task = slot < g.NI * g.H;
    const int t_im = task ? slot / g.H : 0, t_row = task ? slot - t_im * g.H : 0;
    float2 wacc[KS * KS];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) wacc[t] = make_float2(0.f, 0.f);
    float2 q1 = make_float2(0.f, 0.f), q2 = make_float2(0.f, 0.f);
    // P = dz1^T x tile pairs of this wave: et = wave & 1, ct = (wave >> 1) + 2j
    const int wet = wave & 1;
    f32x4_t accp[KST];
#pragma unroll
    for (int j = 0; j < KST; ++j) accp[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int npass = (g.N + g.NI - 1) / g.NI;
    for (int pi = 0; pi < g.ipg; ++pi) {
        const int pass = blockIdx.y * g.ipg + pi;
        if (pass >= npass) break;
        const int n0 = pass * g.NI;
        const int nimg = min(g.NI, g.N - n0);
        const int npx = nimg * g.HW;
        const size_t m0 = (size_t)n0 * g.HW;
        __syncthreads();                                            // previous pass done with region R / first pass: tables visible
        for (int i = tid; i < (2 * img_dw + 3) >> 2; i += 256) ((uint4*)a1p)[i] = make_uint4(0, 0, 0, 0);     // zero borders
        __syncthreads();
        // ---- y1 = W1 act(x) + b1 -> t_y1 (bf16) and a1 -> padded image;   da2 = dy3 W3 -> dy2 -> padded image
        // all global loads of the pass (x, dy3 fragments and the y2 values of this lane's pixels) are issued up front
        constexpr int IC = KST <= 3 ? 4 : 2;                        // pixel tiles whose loads are in flight together
#pragma unroll
      for (int ib = 0; ib < 4; ib += IC) {
        if (wave + 4 * ib >= g.npt) break;
        uint4 xv[4][KST], dv[4][KST];
        uint2 y2v[4][2];
#pragma unroll
        for (int i = ib; i < ib + IC; ++i) {
            const bool live = ppix[i] >= 0 && ppix[i] < npx;
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                xv[i][ks] = make_uint4(0, 0, 0, 0); dv[i][ks] = make_uint4(0, 0, 0, 0);
                const int c = ks * 32 + lg * 8;
                if (live && c < g.C) {
                    xv[i][ks] = *(const uint4*)((const uint16_t*)a.x.data + (m0 + ppix[i]) * g.C + c);
                    dv[i][ks] = *(const uint4*)(a.dy3 + (m0 + ppix[i]) * g.C + c);
                }
            }
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                y2v[i][et] = make_uint2(0, 0);
                if (live) y2v[i][et] = *(const uint2*)(a.y2 + (m0 + ppix[i]) * g.E + e0 + et * 16 + lg * 4);
            }
        }
#pragma unroll
        for (int i = ib; i < ib + IC; ++i) {
            if (wave + 4 * i >= g.npt) break;
            const bool live = ppix[i] >= 0 && ppix[i] < npx;
            if (hasx) {
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    const int c = ks * 32 + lg * 8;
                    if (c >= g.Kpad) break;
                    float s[8], t[8];
                    *(float4*)&s[0] = *(const float4*)(tabx + c); *(float4*)&s[4] = *(const float4*)(tabx + c + 4);
                    *(float4*)&t[0] = *(const float4*)(tabx + g.Kpad + c); *(float4*)&t[4] = *(const float4*)(tabx + g.Kpad + c + 4);
                    xv[i][ks] = live ? act8(xv[i][ks], s, t) : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                f32x4_t acc1 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc2 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    if (ks * 32 >= g.Kpad) break;
                    const bf16x8_t w1f = *(const bf16x8_t*)(t_w1 + (et * 16 + l15) * CP + ks * 32 + lg * 8);
                    const bf16x8_t w3f = *(const bf16x8_t*)(t_w3 + (et * 16 + l15) * CP + ks * 32 + lg * 8);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f, *(const bf16x8_t*)&xv[i][ks], acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f, *(const bf16x8_t*)&dv[i][ks], acc2, 0, 0, 0);
                }
                if (live) {
                    const int el = et * 16 + lg * 4;
                    const float4 b1v = *(const float4*)(tab1 + el), s1v = *(const float4*)(tab1 + 32 + el), t1v = *(const float4*)(tab1 + 64 + el);
                    const float b1a[4] = {b1v.x, b1v.y, b1v.z, b1v.w}, s1a[4] = {s1v.x, s1v.y, s1v.z, s1v.w}, t1a[4] = {t1v.x, t1v.y, t1v.z, t1v.w};
                    uint2 yr;
                    yr.x = pack_bf16(acc1[0] + b1a[0], acc1[1] + b1a[1]); yr.y = pack_bf16(acc1[2] + b1a[2], acc1[3] + b1a[3]);
                    *(uint2*)(t_y1 + ppix[i] * YP + el) = yr;
                    const float yq[4] = {bf_lo(yr.x), bf_hi(yr.x), bf_lo(yr.y), bf_hi(yr.y)};
                    uint2 ar;
                    ar.x = pack_bf16(fmaxf(fmaf(yq[0], s1a[0], t1a[0]), 0.f), fmaxf(fmaf(yq[1], s1a[1], t1a[1]), 0.f));
                    ar.y = pack_bf16(fmaxf(fmaf(yq[2], s1a[2], t1a[2]), 0.f), fmaxf(fmaf(yq[3], s1a[3], t1a[3]), 0.f));
                    *(uint2*)(a1p + poff[i] + et * 8 + lg * 2) = ar;
                    // dy2 = c1 * (da2 * [s2 y2 + t2 > 0]) + c2 * y2 + c3
                    const float4 s2v = *(const float4*)(tab2 + el), t2v = *(const float4*)(tab2 + 32 + el), c1v = *(const float4*)(tab2 + 64 + el);
                    const float4 c2v = *(const float4*)(tab2 + 96 + el), c3v = *(const float4*)(tab2 + 128 + el);
                    const float s2a[4] = {s2v.x, s2v.y, s2v.z, s2v.w}, t2a[4] = {t2v.x, t2v.y, t2v.z, t2v.w}, c1a[4] = {c1v.x, c1v.y, c1v.z, c1v.w};
                    const float c2a[4] = {c2v.x, c2v.y, c2v.z, c2v.w}, c3a[4] = {c3v.x, c3v.y, c3v.z, c3v.w};
                    const float y2q[4] = {bf_lo(y2v[i][et].x), bf_hi(y2v[i][et].x), bf_lo(y2v[i][et].y), bf_hi(y2v[i][et].y)};
                    float d[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dz = (fmaf(y2q[r], s2a[r], t2a[r]) > 0.f) ? ib_bf16r(acc2[r]) : 0.f;
                        d[r] = fmaf(c1a[r], dz, fmaf(c2a[r], y2q[r], c3a[r]));
                    }
                    uint2 dr_;
                    dr_.x = pack_bf16(d[0], d[1]); dr_.y = pack_bf16(d[2], d[3]);
                    *(uint2*)(dy2p + poff[i] + et * 8 + lg * 2) = dr_;
                }
            }
        }
      }
        __syncthreads();
        // ---- depthwise backward: input gradient (flipped filter over dy2) + weight gradient (dy2 row x a1 window), one sweep
        if (task && t_im < nimg) {
            const int rbase = ((t_im * RH + t_row) * RW) * 16 + pair;
            float2 dyc[WW];
#pragma unroll
            for (int j = 0; j < WW; ++j) {
                const uint32_t u = dy2p[rbase + ((P * RW) + P + j) * 16];
                dyc[j] = make_float2(bf_lo(u), bf_hi(u));
            }
            float2 acc[WW];
#pragma unroll
            for (int j = 0; j < WW; ++j) acc[j] = make_float2(0.f, 0.f);
            uint32_t mlo = 0, mhi = 0;                              // ReLU masks of the centre a1 row (channel pair)
#pragma unroll
            for (int dr = 0; dr < KS; ++dr) {
                float2 wf[KS];                                      // flipped filter row: w[KS-1-dr][KS-1-dc]
#pragma unroll
                for (int dc = 0; dc < KS; ++dc) wf[dc] = *(const float2*)(tabw + ((KS - 1 - dr) * KS + (KS - 1 - dc)) * 32 + pair * 2);
                float2 in[WW + 2 * P];
#pragma unroll
                for (int c = 0; c < WW + 2 * P; ++c) {
                    const uint32_t u = dy2p[rbase + (dr * RW + c) * 16];
                    in[c] = make_float2(bf_lo(u), bf_hi(u));
                }
#pragma unroll
                for (int j = 0; j < WW; ++j)
#pragma unroll
                    for (int dc = 0; dc < KS; ++dc) {
                        acc[j].x = fmaf(wf[dc].x, in[j + dc].x, acc[j].x);
                        acc[j].y = fmaf(wf[dc].y, in[j + dc].y, acc[j].y);
                    }
#pragma unroll
                for (int c = 0; c < WW + 2 * P; ++c) {
                    const uint32_t u = a1p[rbase + (dr * RW + c) * 16];
                    in[c] = make_float2(bf_lo(u), bf_hi(u));
                }
                if (dr == P) {
#pragma unroll
                    for (int j = 0; j < WW; ++j) {
                        if (in[P + j].x > 0.f) mlo |= 1u << j;
                        if (in[P + j].y > 0.f) mhi |= 1u << j;
                    }
                }
#pragma unroll
                for (int dc = 0; dc < KS; ++dc) {
                    float2 s = wacc[dr * KS + dc];
#pragma unroll
                    for (int j = 0; j < WW; ++j) {
                        s.x = fmaf(dyc[j].x, in[j + dc].x, s.x);
                        s.y = fmaf(dyc[j].y, in[j + dc].y, s.y);
                    }
                    wacc[dr * KS + dc] = s;
                }
                __builtin_amdgcn_sched_barrier(0);                  // keep the next row's LDS reads from being hoisted (registers)
            }
            // dz1 = da1 * mask1 (da1 rounded to bf16: it is what the unfused path stores), BatchNorm1-backward sums, g1
            const float2 iv = *(const float2*)(tab1 + 96 + pair * 2), nm = *(const float2*)(tab1 + 128 + pair * 2);
            const int prow = t_im * g.HW + t_row * g.W;
            uint16_t* go = a.g1 + (m0 + prow) * g.E + e0 + pair * 2;
#pragma unroll
            for (int j = 0; j < WW; ++j) {
                const uint32_t pk = pack_bf16(((mlo >> j) & 1u) ? acc[j].x : 0.f, ((mhi >> j) & 1u) ? acc[j].y : 0.f);
                const float dzx = bf_lo(pk), dzy = bf_hi(pk);
                const uint32_t yu = *(const uint32_t*)(t_y1 + (prow + j) * YP + pair * 2);
                q1.x += dzx; q1.y += dzy;
                q2.x = fmaf(dzx, fmaf(bf_lo(yu), iv.x, nm.x), q2.x);
                q2.y = fmaf(dzy, fmaf(bf_hi(yu), iv.y, nm.y), q2.y);
                *(uint32_t*)(go + (size_t)j * g.E) = pk;
                *(uint32_t*)(t_y1 + (prow + j) * YP + pair * 2) = pk;      // t_y1 becomes dz1 (this thread owns these elements)
            }
        }
        __syncthreads();
        // ---- x tile (act-on-load) over region R: rows of the pass only, all loads in flight
        {
            constexpr int NCH = 10;                                 // one batch for C <= 96 at 14x14 / C <= 192 at 7x7
            uint4 xc[NCH];
            const int nchunk = npx * c8n;
          for (int q0 = 0; q0 < nchunk; q0 += 256 * NCH) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int q = q0 + tid + 256 * j;
                const int row = q / c8n, c8 = q - row * c8n;
                xc[j] = make_uint4(0, 0, 0, 0);
                if (q < nchunk && c8 * 8 < g.C) xc[j] = *(const uint4*)((const uint16_t*)a.x.data + (m0 + row) * g.C + c8 * 8);
            }
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int q = q0 + tid + 256 * j;
                if (q >= nchunk) continue;
                const int row = q / c8n, c8 = q - row * c8n;
                uint4 o = xc[j];
                if (hasx) {
                    float s[8], t[8];
                    *(float4*)&s[0] = *(const float4*)(tabx + c8 * 8); *(float4*)&s[4] = *(const float4*)(tabx + c8 * 8 + 4);
                    *(float4*)&t[0] = *(const float4*)(tabx + g.Kpad + c8 * 8); *(float4*)&t[4] = *(const float4*)(tabx + g.Kpad + c8 * 8 + 4);
                    o = (c8 * 8 < g.C) ? act8(o, s, t) : make_uint4(0, 0, 0, 0);
                }
                *(uint4*)(xs + row * CP + c8 * 8) = o;
            }
          }
        }
        // rows of t_y1 of images missing from a short last pass must not carry stale dz1
        if (npx < g.NI * g.HW)
            for (int q = tid; q < (g.NI * g.HW - npx) * 4; q += 256)
                *(uint4*)(t_y1 + (npx + (q >> 2)) * YP + (q & 3) * 8) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        // ---- P[e][c] += sum_pix dz1[pix][e] * act(x)[pix][c]
        for (int ks = 0; ks < g.npk; ++ks) {
            const bf16x8_t afrag = ib_tr_frag(t_y1, YP, ks * 32, wet * 16, lane);
#pragma unroll
            for (int j = 0; j < KST; ++j) {
                const int ct = (wave >> 1) + 2 * j;
                if (ct * 16 >= g.Kpad) break;
                const bf16x8_t bfrag = ib_tr_frag(xs, CP, ks * 32, ct * 16, lane);
                accp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag, accp[j], 0, 0, 0);
            }
        }
    }
    // ---- outputs: P partial slab, depthwise weight-gradient partials, BatchNorm1-backward sums
#pragma unroll
    for (int j = 0; j < KST; ++j) {
        const int ct = (wave >> 1) + 2 * j;
        if (ct * 16 >= g.Kpad) break;
        const int c = ct * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c < g.C) a.ppartial[((size_t)blockIdx.y * g.E + e0 + wet * 16 + lg * 4 + r) * g.C + c] = accp[j][r];
    }
    for (int dr = 0; dr < KS; ++dr) {
        __syncthreads();
#pragma unroll
        for (int dc = 0; dc < KS; ++dc) {
            // constant index after full unrolling of the dr loop is not guaranteed: select with a switch-free copy
            float2 v = make_float2(0.f, 0.f);
#pragma unroll
            for (int d2 = 0; d2 < KS; ++d2) if (d2 == dr) v = wacc[d2 * KS + dc];
            *(float2*)(lds_red + (slot * KS + dc) * 32 + pair * 2) = v;
        }
        __syncthreads();
        if (tid < KS * 32) {
            const int dc = tid >> 5, c = tid & 31;
            float v = 0.f;
            for (int s = 0; s < 16; ++s) v += lds_red[(s * KS + dc) * 32 + c];
            a.dwpartial[((size_t)blockIdx.y * KS * KS + dr * KS + dc) * g.E + e0 + c] = v;
        }
    }
    __syncthreads();
    *(float2*)(lds_red + (slot * 2 + 0) * 32 + pair * 2) = q1;
    *(float2*)(lds_red + (slot * 2 + 1) * 32 + pair * 2) = q2;
    __syncthreads();
    if (tid < 64) {
        const int r = tid >> 5, c = tid & 31;
        float v = 0.f;
        for (int s = 0; s < 16; ++s) v += lds_red[(s * 2 + r) * 32 + c];
        a.red1[((size_t)r * g.E + e0 + c) * gridDim.y + blockIdx.y] = v;
    }
}

// =====================================================================================================================
// k_irb_bwd_exp
// =====================================================================================================================
struct IrbExpArgs {
    IrbGeom g;
    MnasActIn x;
    const uint16_t* g1;         // dz1 (M,E)
    const uint16_t* w1;         // MNAS_PACK_FWD [E_pad16][Kpad]
    const float* b1;
    const float* bn1;           // rows 2,3,4 (c1,c2,c3)
    const uint16_t* resid;      // G (M,C) or NULL
    uint16_t* dx;               // out (M,C)
};

// workgroup = 512 threads (8 waves) = one image pass at a time; wave w owns the 16-pixel tiles w and w + 8
template <int KST>
__global__ __launch_bounds__(512) void k_irb_bwd_exp(IrbExpArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const IrbGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int CP = g.Kpad + 8;
    constexpr int CTN = 2 * KST;                                    // 16-channel tiles of C
    uint16_t* wbuf = (uint16_t*)smem;                               // [2][32][CP]: expand-weight chunks (32 rows of E)
    float* tabx = (float*)(wbuf + 2 * 32 * CP);                     // [2][Kpad]
    float* tab1 = tabx + 2 * g.Kpad;                                // [4][E]: b1, c1, c2, c3
    const int c8n = g.Kpad >> 3;
    const bool hasx = a.x.scale != nullptr;
    const int nes = g.E >> 5;

    for (int i = tid; i < 2 * g.Kpad; i += 512) {
        const int c = i < g.Kpad ? i : i - g.Kpad;
        tabx[i] = (hasx && c < g.C) ? (i < g.Kpad ? a.x.scale[c] : a.x.shift[c]) : 0.f;
    }
    for (int i = tid; i < 4 * g.E; i += 512) {
        const int r = i / g.E, e = i - r * g.E;
        tab1[i] = r == 0 ? (a.b1 ? a.b1[e] : 0.f) : a.bn1[(size_t)(r + 1) * g.E + e];
    }
    // weight-chunk staging plan: 32 x c8n 16-byte pieces per chunk, <= 2 per thread (Kpad <= 192: 768 pieces)
    const int wq0 = tid, wq1 = tid + 512;
    const int npieces = 32 * c8n;
    auto wload = [&](uint4 (&wr)[2], int es) {
        wr[0] = make_uint4(0, 0, 0, 0); wr[1] = make_uint4(0, 0, 0, 0);
        if (es >= nes) return;
        if (wq0 < npieces) wr[0] = *(const uint4*)(a.w1 + ((size_t)es * 32 + wq0 / c8n) * g.Kpad + (wq0 % c8n) * 8);
        if (wq1 < npieces) wr[1] = *(const uint4*)(a.w1 + ((size_t)es * 32 + wq1 / c8n) * g.Kpad + (wq1 % c8n) * 8);
    };
    auto wstore = [&](const uint4 (&wr)[2], int buf) {
        uint16_t* d = wbuf + buf * 32 * CP;
        if (wq0 < npieces) *(uint4*)(d + (wq0 / c8n) * CP + (wq0 % c8n) * 8) = wr[0];
        if (wq1 < npieces) *(uint4*)(d + (wq1 / c8n) * CP + (wq1 % c8n) * 8) = wr[1];
    };
    // row of the weight chunk that feeds MFMA row m of e-tile et: e = (m/4)*8 + et*4 + (m%4)  -> the lane that holds D rows
    // lg*4 + r of both tiles holds e = lg*8 + 0..7 in order: exactly the B operand of the second GEMM
    const int wrow0 = (l15 >> 2) * 8 + (l15 & 3);

    const int npass = (g.N + g.NI - 1) / g.NI;
    for (int pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        const int n0 = pass * g.NI;
        const int npx = min(g.NI, g.N - n0) * g.HW;
        const size_t m0 = (size_t)n0 * g.HW;
        int pix[2];
        bool live[2];
        bf16x8_t xfr[2][KST];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pt = wave + 8 * i;
            pix[i] = pt * 16 + l15;
            live[i] = pt < g.npt && pix[i] < npx;
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                uint4 u = make_uint4(0, 0, 0, 0);
                const int c = ks * 32 + lg * 8;
                if (live[i] && c < g.C) {
                    u = *(const uint4*)((const uint16_t*)a.x.data + (m0 + pix[i]) * g.C + c);
                    if (hasx) {
                        float s[8], t[8];
                        *(float4*)&s[0] = *(const float4*)(tabx + c); *(float4*)&s[4] = *(const float4*)(tabx + c + 4);
                        *(float4*)&t[0] = *(const float4*)(tabx + g.Kpad + c); *(float4*)&t[4] = *(const float4*)(tabx + g.Kpad + c + 4);
                        u = act8(u, s, t);
                    }
                }
                xfr[i][ks] = *(const bf16x8_t*)&u;
            }
        }
        f32x4_t dacc[2][CTN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct) dacc[i][ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        uint4 wr[2];
        wload(wr, 0);
        __syncthreads();                                            // previous pass finished with wbuf (first pass: tables visible)
        wstore(wr, 0);
        wload(wr, 1);
        uint2 dzn[2][2];                                            // dz1 of the NEXT chunk, loaded one step ahead
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                dzn[i][et] = make_uint2(0, 0);
                if (live[i]) dzn[i][et] = *(const uint2*)(a.g1 + (m0 + pix[i]) * g.E + lg * 8 + et * 4);
            }
        for (int es = 0; es < nes; ++es) {
            __syncthreads();                                        // chunk es published; buffer (es+1)&1 free
            if (es + 1 < nes) wstore(wr, (es + 1) & 1);
            wload(wr, es + 2);
            uint2 dz[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    dz[i][et] = dzn[i][et];
                    dzn[i][et] = make_uint2(0, 0);
                    if (live[i] && es + 1 < nes) dzn[i][et] = *(const uint2*)(a.g1 + (m0 + pix[i]) * g.E + (es + 1) * 32 + lg * 8 + et * 4);
                }
            const uint16_t* wb = wbuf + (es & 1) * 32 * CP;
            // ---- y1 chunk: D[e][pix], e-tile rows permuted (see wrow0)
            f32x4_t acc1[2][2];
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                bf16x8_t af[KST];
#pragma unroll
                for (int ks = 0; ks < KST; ++ks)
                    af[ks] = *(const bf16x8_t*)(wb + (wrow0 + et * 4) * CP + ks * 32 + lg * 8);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc1[et][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks)
                        if (ks * 32 < g.Kpad) acc1[et][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], xfr[i][ks], acc1[et][i], 0, 0, 0);
                }
            }
            // ---- dy1 = c1*dz1 + c2*y1 + c3 for e = es*32 + lg*8 + 0..7: the second GEMM's B fragment, in registers
            bf16x8_t bfr[2];
            {
                const int eb = es * 32 + lg * 8;
                float b1a[8], c1a[8], c2a[8], c3a[8];
                *(float4*)&b1a[0] = *(const float4*)(tab1 + eb); *(float4*)&b1a[4] = *(const float4*)(tab1 + eb + 4);
                *(float4*)&c1a[0] = *(const float4*)(tab1 + g.E + eb); *(float4*)&c1a[4] = *(const float4*)(tab1 + g.E + eb + 4);
                *(float4*)&c2a[0] = *(const float4*)(tab1 + 2 * g.E + eb); *(float4*)&c2a[4] = *(const float4*)(tab1 + 2 * g.E + eb + 4);
                *(float4*)&c3a[0] = *(const float4*)(tab1 + 3 * g.E + eb); *(float4*)&c3a[4] = *(const float4*)(tab1 + 3 * g.E + eb + 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float d[8];
#pragma unroll
                    for (int et = 0; et < 2; ++et) {
                        const float dzq[4] = {bf_lo(dz[i][et].x), bf_hi(dz[i][et].x), bf_lo(dz[i][et].y), bf_hi(dz[i][et].y)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = et * 4 + r;
                            const float y1 = ib_bf16r(acc1[et][i][r] + b1a[j]);
                            d[j] = live[i] ? fmaf(c1a[j], dzq[r], fmaf(c2a[j], y1, c3a[j])) : 0.f;
                        }
                    }
                    const uint4 u = pack8(d);
                    bfr[i] = *(const bf16x8_t*)&u;
                }
            }
            // ---- dx[c][pix] += W1[e][c]^T dy1: A = transpose read of the same chunk (rows = e in natural order)
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct) {
                if (ct * 16 >= g.Kpad) break;
                const bf16x8_t af = ib_tr_frag(wb, CP, 0, ct * 16, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i) dacc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[i], dacc[i][ct], 0, 0, 0);
            }
        }
        // ---- epilogue: + G, store
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!live[i]) continue;
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct) {
                const int c = ct * 16 + lg * 4;
                if (c >= g.C) break;
                float v[4] = {dacc[i][ct][0], dacc[i][ct][1], dacc[i][ct][2], dacc[i][ct][3]};
                if (a.resid) {
                    const uint2 rv = *(const uint2*)(a.resid + (m0 + pix[i]) * g.C + c);
                    v[0] += bf_lo(rv.x); v[1] += bf_hi(rv.x); v[2] += bf_lo(rv.y); v[3] += bf_hi(rv.y);
                }
                uint2 pk;
                pk.x = pack_bf16(v[0], v[1]); pk.y = pack_bf16(v[2], v[3]);
                *(uint2*)(a.dx + (m0 + pix[i]) * g.C + c) = pk;
            }
        }
    }
}

// =====================================================================================================================
// k_irb_w1_fin: dW1[e][c] += c1[e] * sum_p P[p][e][c] + c2[e] * (sum_j W1b[e][j] G[j][c] + b1[e] Sx[c]) + c3[e] * Sx[c]
// =====================================================================================================================
__global__ __launch_bounds__(256) void k_irb_w1_fin(const float* __restrict__ ppartial, int nparts, int E, int C,
                                                    const double* __restrict__ gsum, const float* __restrict__ w1,
                                                    const float* __restrict__ b1, const float* __restrict__ bn1,
                                                    float* __restrict__ grad, int accumulate) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= E * C) return;
    const int e = idx / C, c = idx - e * C;
    double p = 0.0;
    for (int s = 0; s < nparts; ++s) p += (double)ppartial[((size_t)s * E + e) * C + c];
    double q = 0.0;
    for (int j = 0; j < C; ++j) q += (double)bf_to_f(f_to_bf(w1[(size_t)e * C + j])) * gsum[(size_t)j * C + c];
    const double sx = gsum[(size_t)C * C + c];
    q += (b1 ? (double)b1[e] : 0.0) * sx;
    const double v = (double)bn1[2 * E + e] * p + (double)bn1[3 * E + e] * q + (double)bn1[4 * E + e] * sx;
    grad[idx] = (accumulate ? grad[idx] : 0.f) + (float)v;
}

// ---- host side ---------------------------------------------------------------------------------------------------------
extern "C" int mnas_irb_supported(int N, int H, int W, int C, int E, int k);

static bool irb_geom(int N, int H, int W, int C, int E, int k, int nparts, IrbGeom* g) {
    if (!mnas_irb_supported(N, H, W, C, E, k) || nparts < 1) return false;
    g->N = N; g->H = H; g->W = W; g->C = C; g->E = E;
    g->HW = H * W; g->NI = W == 7 ? 2 : 1;
    g->Kpad = (C + 31) / 32 * 32;
    g->npt = (g->NI * g->HW + 15) / 16;
    g->npk = (g->NI * g->HW + 31) / 32;
    const int npass = (N + g->NI - 1) / g->NI;
    g->ipg = (npass + nparts - 1) / nparts;
    return (npass + g->ipg - 1) / g->ipg == nparts;                 // nparts must come from mnas_irb_fwd_parts
}

extern "C" int mnas_irb_bwd_proj(const MnasIrbBwd* c, void* stream) {
    IrbProjArgs a;
    if (!c || !irb_geom(c->N, c->H, c->W, c->C, c->E, c->k, c->nparts, &a.g)) return MNAS_EINVAL;
    if (!c->gout.g || !c->gout.y || !c->gout.coef || !c->y2 || !c->bn2 || !c->w3t || !c->dy3 || !c->w3partial || !c->red2) return MNAS_EINVAL;
    a.gout = (const uint16_t*)c->gout.g; a.y3 = (const uint16_t*)c->gout.y; a.bn3 = c->gout.coef;
    a.y2 = (const uint16_t*)c->y2; a.bn2 = c->bn2; a.w3t = (const uint16_t*)c->w3t;
    a.dy3 = (uint16_t*)c->dy3; a.wpartial = c->w3partial; a.red2 = c->red2;
    const int CP = a.g.Kpad + 8, KP = a.g.npk * 32;
    const size_t lds = (size_t)KP * CP * 2 + (size_t)KP * 40 * 2 + (size_t)32 * CP * 2 + (size_t)5 * a.g.Kpad * 4 + 128 * 4 + 256 * 4;
    const dim3 grid(c->E / 32, c->nparts);
    hipStream_t s = (hipStream_t)stream;
    if (a.g.Kpad <= 96) hipLaunchKernelGGL((k_irb_bwd_proj<3>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((k_irb_bwd_proj<6>), grid, dim3(256), lds, s, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_irb_bwd_dw(const MnasIrbBwd* c, void* stream) {
    IrbDwArgs a;
    if (!c || !irb_geom(c->N, c->H, c->W, c->C, c->E, c->k, c->nparts, &a.g)) return MNAS_EINVAL;
    if (!c->x.data || !c->dy3 || !c->y2 || !c->w1 || !c->w3t || !c->bn1 || !c->bn2 || !c->wdw || !c->g1 || !c->dwpartial ||
        !c->ppartial || !c->red1) return MNAS_EINVAL;
    if ((c->x.scale == nullptr) != (c->x.shift == nullptr)) return MNAS_EINVAL;
    a.x = c->x; a.dy3 = (const uint16_t*)c->dy3; a.y2 = (const uint16_t*)c->y2; a.w1 = (const uint16_t*)c->w1;
    a.w3t = (const uint16_t*)c->w3t; a.b1 = c->b1; a.bn1 = c->bn1; a.bn2 = c->bn2; a.wdw = c->wdw;
    a.g1 = (uint16_t*)c->g1; a.dwpartial = c->dwpartial; a.ppartial = c->ppartial; a.red1 = c->red1;
    const int P = c->k / 2, RW = (c->W + 2 * P) | 1, RH = c->H + 2 * P;
    const int CP = a.g.Kpad + 8, KP = a.g.npk * 32;
    const size_t img = (size_t)a.g.NI * RH * RW * 16 * 4;
    size_t region = 2 * img;
    const size_t xs_b = (((size_t)a.g.NI * a.g.HW * CP + 1) / 2) * 4;
    if (xs_b > region) region = xs_b;
    if ((size_t)16 * c->k * 32 * 4 > region) region = (size_t)16 * c->k * 32 * 4;      // the final reductions alias region R
    region = (region + 15) & ~(size_t)15;
    size_t lds = region + (size_t)KP * 40 * 2 + (size_t)2 * 32 * CP * 2 + (size_t)2 * a.g.Kpad * 4 + (size_t)10 * 32 * 4 +
                 (size_t)c->k * c->k * 32 * 4;
    lds = (lds + 15) & ~(size_t)15;
    a.lds_bytes = (int)lds;
    const dim3 grid(c->E / 32, c->nparts);
    hipStream_t s = (hipStream_t)stream;
    const int kst = a.g.Kpad <= 96 ? 3 : 6;
#define MNAS_IRB_DW(K_, W_, T_) \
    if (c->k == K_ && c->W == W_ && kst == T_) { hipLaunchKernelGGL((k_irb_bwd_dw<K_, W_, T_>), grid, dim3(256), lds, s, a); MNAS_CHECK_LAUNCH(); return MNAS_OK; }
    MNAS_IRB_DW(3, 14, 3) MNAS_IRB_DW(5, 14, 3) MNAS_IRB_DW(3, 7, 3) MNAS_IRB_DW(5, 7, 3)
    MNAS_IRB_DW(3, 14, 6) MNAS_IRB_DW(5, 14, 6) MNAS_IRB_DW(3, 7, 6) MNAS_IRB_DW(5, 7, 6)
#undef MNAS_IRB_DW
    return MNAS_EINVAL;
}

extern "C" int mnas_irb_bwd_exp(const MnasIrbBwd* c, void* stream) {
    IrbExpArgs a;
    if (!c || !irb_geom(c->N, c->H, c->W, c->C, c->E, c->k, 1, &a.g)) return MNAS_EINVAL;
    if (!c->x.data || !c->g1 || !c->w1 || !c->bn1 || !c->dx) return MNAS_EINVAL;
    if ((c->x.scale == nullptr) != (c->x.shift == nullptr)) return MNAS_EINVAL;
    a.x = c->x; a.g1 = (const uint16_t*)c->g1; a.w1 = (const uint16_t*)c->w1; a.b1 = c->b1; a.bn1 = c->bn1;
    a.resid = (const uint16_t*)c->gout.g; a.dx = (uint16_t*)c->dx;
    const int CP = a.g.Kpad + 8;
    const size_t lds = (size_t)2 * 32 * CP * 2 + (size_t)2 * a.g.Kpad * 4 + (size_t)4 * c->E * 4;
    const int npass = (c->N + a.g.NI - 1) / a.g.NI;
    const dim3 grid(npass);
    hipStream_t s = (hipStream_t)stream;
    if (a.g.Kpad <= 96) hipLaunchKernelGGL((k_irb_bwd_exp<3>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((k_irb_bwd_exp<6>), grid, dim3(512), lds, s, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_irb_w1_finalize(const float* ppartial, int nparts, int E, int C, const double* gsum, const float* w1,
                                    const float* b1, const float* bn1, float* grad, int accumulate, void* stream) {
    if (!ppartial || nparts < 1 || E < 1 || C < 1 || !gsum || !w1 || !bn1 || !grad) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_irb_w1_fin, dim3((E * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ppartial, nparts, E, C, gsum, w1,
                       b1, bn1, grad, accumulate);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
