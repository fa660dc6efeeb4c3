// Stem convolution (dense 3x3, stride 2, pad 1, 3 -> 32 channels on the fp32 NCHW network input; mnasnet.py:179 /
// ConvBlock :48-62): forward and weight gradient as BAND kernels.  Replaces, for Co == 32, the im2col staging modes of
// k_igemm / k_wgrad (csrc/mnas_gemm.hip MODE 2, csrc/mnas_wgrad.hip STEM), which build every 27-value patch with 27
// scalar fp32 global loads and an index decode per value (measured 153 us forward / 217 us weight gradient at
// 256 x 3 x 224 x 224 against 76 / 120 us of memory time).
//
// A workgroup walks bands of RB output rows of one image persistently.  Per band the 2*RB+1 input rows of the three
// planes are read ONCE with 16-byte row-contiguous loads, rounded to bf16 (as every activation) and kept in LDS
// ([plane][row][W + 4], zero columns left and right, zero rows above / below the image); the MFMA fragments gather
// their patch values from that tile with 2-byte LDS reads at per-lane constant offsets:
//   forward  D[co][pix] = W[co16][k32] * patch[pix16][k32]      k = ci*9 + kh*3 + kw (27 used, reference weight order)
//            lane (pix = l15, k = 8*lg + j): LDS element (ci, 2*oy_l + kh, 2*ox + kw); epilogue as k_igemm: bias,
//            BatchNorm partial statistics per workgroup, bf16 NHWC store;
//   wgrad    D[co][k]   = dy^T[co16][pix32] * patch^T[k16][pix32]  over all pixels of the workgroup's bands
//            dy = dy-on-load of (g, y) staged pixel-major in LDS and read with ds_read_b64_tr_b16 (as k_wgrad);
//            lane (k = l15, pixels 8*lg + j): 8 consecutive output pixels never cross a row (Wo % 8 == 0).
// Roofline: HBM (reads 154 MB image + writes 205 MB / reads 154 + 411 MB at batch 256).
#include "mnas_common.h"

typedef __attribute__((ext_vector_type(4))) short st_s4_t;
typedef __attribute__((address_space(3))) st_s4_t* st_lds_s4_ptr;

struct StemArgs {
    int N, H, W, Ho, Wo;
    int RB, nbh;             // output rows per band, bands per image
    int LW;                  // LDS row pitch in elements (W + 4, even)
    const float* x;
    const uint16_t* w;       // forward: packed [32][32] bf16
    const float* bias;
    void* out;
    float* stats;            // forward: [2][32][gridDim.x]
    MnasGradIn dy;           // wgrad
    float* partial;          // wgrad: [gridDim.x][32][27]
    int nt;
    const float* in_affine;  // [2][3] or NULL: x -> scale[c] * x + shift[c] (fused input normalisation)
    int in_u8;               // x is uint8 NCHW
};

// input rows 2*oy0-1 .. 2*oy0+2*RB-1 of the three planes of image n -> tile[(c*R + r)*LW + iw + 2] (bf16)
__device__ __forceinline__ void stem_stage_input(const StemArgs& a, uint16_t* tile, int n, int oy0) {
    const int R = 2 * a.RB + 1, w4 = a.W >> 2;
    const int total = 3 * R * w4;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int row = i / w4, q = i - row * w4;
        const int c = row / R, r = row - c * R;
        const int ih = 2 * oy0 - 1 + r;
        uint2 pk = make_uint2(0, 0);
        if (ih >= 0 && ih < a.H) {
            const size_t e = (((size_t)n * 3 + c) * a.H + ih) * a.W + 4 * q;
            float4 v;
            if (a.in_u8) {
                const uint32_t u = *(const uint32_t*)((const uint8_t*)a.x + e);
                v = make_float4((float)(u & 0xffu), (float)((u >> 8) & 0xffu), (float)((u >> 16) & 0xffu), (float)(u >> 24));
            } else {
                v = *(const float4*)(a.x + e);
            }
            if (a.in_affine) {                                      // zero padding stays zero: it pads the NORMALISED image
                const float sc = a.in_affine[c], sh = a.in_affine[3 + c];
                v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
            }
            pk.x = pack_bf16(v.x, v.y);
            pk.y = pack_bf16(v.z, v.w);
        }
        uint32_t* d = (uint32_t*)(tile + row * a.LW + 4 * q + 2);
        d[0] = pk.x; d[1] = pk.y;
    }
}
__device__ __forceinline__ void stem_zero_pads(const StemArgs& a, uint16_t* tile) {
    const int rows = 3 * (2 * a.RB + 1);
    for (int i = threadIdx.x; i < rows * 4; i += 256) {
        const int row = i >> 2, j = i & 3;
        tile[row * a.LW + (j < 2 ? j : a.W + j)] = 0;
    }
}

// ---- forward ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stem_fwd(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* tile = (uint16_t*)smem;                       // [3][2RB+1][LW]
    float* lds_red = (float*)(tile + 3 * (2 * a.RB + 1) * a.LW + 8);
    lds_red = (float*)(((uintptr_t)lds_red + 15) & ~(uintptr_t)15);      // [4][2][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int R = 2 * a.RB + 1;
    stem_zero_pads(a, tile);
    // weights: A fragments of the two cout tiles, resident in registers; bias of this lane's couts
    bf16x8_t afrag[2];
    float bias_r[2][4], s1[2][4], s2[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        afrag[ct] = *(const bf16x8_t*)(a.w + (ct * 16 + l15) * 32 + lg * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bias_r[ct][r] = a.bias ? a.bias[ct * 16 + lg * 4 + r] : 0.f;
            s1[ct][r] = 0.f; s2[ct][r] = 0.f;
        }
    }
    // this lane's 8 patch offsets: k = 8*lg + j = ci*9 + kh*3 + kw -> (ci*R + kh)*LW + kw + 1; k >= 27 reads element 0 (a zero pad)
    int off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = lg * 8 + j, ci = k / 9, r9 = k - ci * 9, kh = r9 / 3, kw = r9 - kh * 3;
        off[j] = k < 27 ? (ci * R + kh) * a.LW + kw + 1 : -1;
    }
    const int nbands = a.N * a.nbh;
    for (int b = blockIdx.x; b < nbands; b += gridDim.x) {
        const int n = b / a.nbh, oy0 = (b - n * a.nbh) * a.RB;
        const int rbv = min(a.RB, a.Ho - oy0), P = rbv * a.Wo;
        __syncthreads();                                   // previous band consumed (first pass: pads visible)
        stem_stage_input(a, tile, n, oy0);
        __syncthreads();
        const int ntile = (P + 15) >> 4;
        for (int t = wave; t < ntile; t += 4) {
            const int p = t * 16 + l15;
            const bool ok = p < P;
            const int oyl = ok ? p / a.Wo : 0, ox = ok ? p - oyl * a.Wo : 0;
            const uint16_t* base = tile + 2 * oyl * a.LW + 2 * ox;
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = off[j] >= 0 ? base[off[j]] : 0u;
            uint4 pk;
            pk.x = v[0] | (v[1] << 16); pk.y = v[2] | (v[3] << 16); pk.z = v[4] | (v[5] << 16); pk.w = v[6] | (v[7] << 16);
            const bf16x8_t bfrag = *(const bf16x8_t*)&pk;
            uint16_t* o = (uint16_t*)a.out + ((size_t)n * a.Ho * a.Wo + (size_t)oy0 * a.Wo + p) * 32;
            f32x4_t accs[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)       // both MFMAs under uniform control flow; the masked epilogue comes after
                accs[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[ct], bfrag, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            if (!ok) continue;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const f32x4_t acc = accs[ct];
                float r4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    r4[r] = acc[r] + bias_r[ct][r];
                    s1[ct][r] += r4[r];
                    s2[ct][r] = fmaf(r4[r], r4[r], s2[ct][r]);
                }
                uint2 st;
                st.x = pack_bf16(r4[0], r4[1]);
                st.y = pack_bf16(r4[2], r4[3]);
                st_u2(o + ct * 16 + lg * 4, st, a.nt);
            }
        }
    }
    if (a.stats) {
        // deterministic workgroup reduction (as k_igemm): 16-lane shuffle tree, one LDS slot per (wave, channel), waves in order
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x1 = s1[ct][r], x2 = s2[ct][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
                if (l15 == 0) {
                    lds_red[(wave * 2 + 0) * 32 + ct * 16 + lg * 4 + r] = x1;
                    lds_red[(wave * 2 + 1) * 32 + ct * 16 + lg * 4 + r] = x2;
                }
            }
        __syncthreads();
        if (tid < 64) {
            const int r = tid >> 5, c = tid & 31;
            const float v = ((lds_red[(0 * 2 + r) * 32 + c] + lds_red[(1 * 2 + r) * 32 + c]) + lds_red[(2 * 2 + r) * 32 + c]) +
                            lds_red[(3 * 2 + r) * 32 + c];
            a.stats[((size_t)r * 32 + c) * gridDim.x + blockIdx.x] = v;
        }
    }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t stem_tr_frag(const uint16_t* tile, int ld, int row0, int col0, int lane) {
    // rows row0 + (lane>>4)*8 + {0..7}, column col0 + (lane&15)   (see mnas_wgrad.hip)
    const int i = lane & 15, g = lane >> 4;
    const uint16_t* p = tile + (row0 + g * 8 + (i >> 2)) * ld + col0 + (i & 3) * 4;
    const st_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s4_ptr)p);
    const st_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s4_ptr)(p + 4 * ld));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256) void k_stem_wgrad(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LDD = 40;                                  // dy tile row pitch (32 couts + 8 pad)
    const int R = 2 * a.RB + 1;
    const int Pmax = (a.RB * a.Wo + 31) & ~31;
    float* lds_cd = (float*)smem;                            // [5][32]
    uint16_t* tile_d = (uint16_t*)(lds_cd + 5 * 32);         // [Pmax][LDD]
    uint16_t* tile = tile_d + Pmax * LDD;                    // [3][R][LW]
    float* lds_out = (float*)tile_d;                         // reused at the end: [32][33]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const bool hasdy = a.dy.coef != nullptr;
    for (int i = tid; i < 5 * 32; i += 256) lds_cd[i] = hasdy ? a.dy.coef[i] : 0.f;
    for (int i = tid; i < Pmax * LDD / 8; i += 256) ((uint4*)tile_d)[i] = make_uint4(0, 0, 0, 0);
    stem_zero_pads(a, tile);
    // this lane's patch column of the two k tiles: k = kt*16 + l15 -> (ci*R + kh)*LW + kw + 1
    int koff[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int k = kt * 16 + l15, ci = k / 9, r9 = k - ci * 9, kh = r9 / 3, kw = r9 - kh * 3;
        koff[kt] = k < 27 ? (ci * R + kh) * a.LW + kw + 1 : -1;
    }
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nbands = a.N * a.nbh;
    for (int b = blockIdx.x; b < nbands; b += gridDim.x) {
        const int n = b / a.nbh, oy0 = (b - n * a.nbh) * a.RB;
        const int rbv = min(a.RB, a.Ho - oy0), P = rbv * a.Wo;
        __syncthreads();
        stem_stage_input(a, tile, n, oy0);
        {   // dy tile: P pixels x 4 chunks of 8 couts; rows P..Pmax stay/are zero
            const size_t pix0 = (size_t)n * a.Ho * a.Wo + (size_t)oy0 * a.Wo;
            const int slots = Pmax * 4;
            for (int i = tid; i < slots; i += 256) {
                const int p = i >> 2, c8 = i & 3;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (p < P) {
                    const size_t o = (pix0 + p) * 32 + c8 * 8;
                    const uint4 vg = *(const uint4*)((const uint16_t*)a.dy.g + o);
                    if (hasdy) {
                        const uint4 vy = *(const uint4*)((const uint16_t*)a.dy.y + o);
                        float cf[5][8], d[8];
#pragma unroll
                        for (int r = 0; r < 5; ++r) {
                            *(float4*)&cf[r][0] = *(const float4*)(lds_cd + r * 32 + c8 * 8);
                            *(float4*)&cf[r][4] = *(const float4*)(lds_cd + r * 32 + c8 * 8 + 4);
                        }
                        dy8(vg, vy, cf[0], cf[1], cf[2], cf[3], cf[4], d);
                        v = pack8(d);
                    } else {
                        v = vg;
                    }
                }
                *(uint4*)(tile_d + p * LDD + c8 * 8) = v;
            }
        }
        __syncthreads();
        const int nstep = (P + 31) >> 5;
        for (int s = wave; s < nstep; s += 4) {
            // B fragments: patch^T[k][8 pixels p0 .. p0+7 of one output row]
            const int p0 = s * 32 + lg * 8;
            const int oyl = p0 / a.Wo, ox0 = p0 - oyl * a.Wo;
            const uint16_t* base = tile + 2 * oyl * a.LW + 2 * ox0;
            bf16x8_t bf[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                uint32_t v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (koff[kt] >= 0 && p0 < P) ? base[koff[kt] + 2 * j] : 0u;
                uint4 pk;
                pk.x = v[0] | (v[1] << 16); pk.y = v[2] | (v[3] << 16); pk.z = v[4] | (v[5] << 16); pk.w = v[6] | (v[7] << 16);
                bf[kt] = *(const bf16x8_t*)&pk;
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const bf16x8_t af = stem_tr_frag(tile_d, LDD, s * 32, ct * 16, lane);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) acc[ct][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[kt], acc[ct][kt], 0, 0, 0);
            }
        }
    }
    // ---- the 4 waves' slabs summed through LDS in wave order, then partial[workgroup][co][27]
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave != w) continue;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* d = &lds_out[(ct * 16 + lg * 4 + r) * 33 + kt * 16 + l15];
                    *d = (w == 0 ? 0.f : *d) + acc[ct][kt][r];
                }
    }
    __syncthreads();
    float* dst = a.partial + (size_t)blockIdx.x * 32 * 27;
    for (int i = tid; i < 32 * 27; i += 256) {
        const int r = i / 27, c = i - r * 27;
        dst[i] = lds_out[r * 33 + c];
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// preferred persistent grid (host-side, no launch): which = 0 forward, 1 weight gradient; -1 = not a band shape
extern "C" int mnas_stem_parts(int which, int N, int H, int W, int Co);
int mnas_stem_band_enabled() {
    static int on = -1;
    if (on < 0) on = mnas_diag_env("MNAS_STEM_BAND", 1);
    return on;
}
static bool stem_band_ok(int N, int H, int W, int Ho, int Wo, int Co, bool wgrad) {
    if (!mnas_stem_band_enabled() || Co != 32 || (W & 3) || W < 8 || H < 2 || N < 1) return false;
    if (Ho != (H - 1) / 2 + 1 || Wo != (W - 1) / 2 + 1) return false;
    if (wgrad && (Wo & 7)) return false;
    return true;
}
static size_t stem_lds(int RB, int W, int Wo, bool wgrad) {
    const int LW = (W + 4 + 1) & ~1, R = 2 * RB + 1;
    size_t b = (size_t)3 * R * LW * 2;
    if (wgrad) b += (size_t)5 * 32 * 4 + (size_t)((RB * Wo + 31) & ~31) * 40 * 2;
    else b += 16 + 16 + (size_t)8 * 32 * 4;
    return b;
}

extern "C" int mnas_stem_parts(int which, int N, int H, int W, int Co) {
    if (!stem_band_ok(N, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, Co, which == 1)) return -1;
    const int Ho = (H - 1) / 2 + 1;
    const int rb = which == 1 ? 4 : 8;
    const long long bands = (long long)N * ((Ho + rb - 1) / rb);
    const int want = which == 1 ? 768 : 1024;          // 3 x 48 KB / 4+ x 24 KB workgroups per CU
    return (int)(bands < want ? bands : want);
}

// returns MNAS_EINVAL when the shape is not a band shape (caller falls back to the im2col path)
int mnas_stem_fwd_band(const MnasStemFwd* c, void* stream) {
    if (!stem_band_ok(c->N, c->H, c->W, c->Ho, c->Wo, c->Co, false)) return MNAS_EINVAL;
    StemArgs a = {};
    a.N = c->N; a.H = c->H; a.W = c->W; a.Ho = c->Ho; a.Wo = c->Wo;
    a.RB = 8;
    while (a.RB > 1 && stem_lds(a.RB, c->W, c->Wo, false) > 48 * 1024) a.RB >>= 1;
    const size_t lds = stem_lds(a.RB, c->W, c->Wo, false);
    if (lds > 64 * 1024) return MNAS_EINVAL;
    a.nbh = (c->Ho + a.RB - 1) / a.RB;
    a.LW = (c->W + 4 + 1) & ~1;
    a.x = c->x; a.w = (const uint16_t*)c->w; a.bias = c->bias; a.out = c->out; a.stats = c->stats;
    a.in_affine = c->in_affine; a.in_u8 = c->in_u8;
    a.nt = (mnas_nt_mask() & MNAS_NT_STEM) ? 1 : 0;
    hipLaunchKernelGGL(k_stem_fwd, dim3(c->nparts), dim3(256), lds, (hipStream_t)stream, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

int mnas_stem_wgrad_band(const MnasStemWgrad* c, void* stream) {
    if (!stem_band_ok(c->N, c->H, c->W, c->Ho, c->Wo, c->Co, true)) return MNAS_EINVAL;
    if ((c->dy.coef == nullptr) != (c->dy.y == nullptr)) return MNAS_EINVAL;
    StemArgs a = {};
    a.N = c->N; a.H = c->H; a.W = c->W; a.Ho = c->Ho; a.Wo = c->Wo;
    a.RB = 4;
    while (a.RB > 1 && stem_lds(a.RB, c->W, c->Wo, true) > 52 * 1024) a.RB >>= 1;
    const size_t lds = stem_lds(a.RB, c->W, c->Wo, true);
    if (lds > 64 * 1024) return MNAS_EINVAL;
    a.nbh = (c->Ho + a.RB - 1) / a.RB;
    a.LW = (c->W + 4 + 1) & ~1;
    a.x = c->x; a.dy = c->dy; a.partial = c->partial;
    a.in_affine = c->in_affine; a.in_u8 = c->in_u8;
    hipLaunchKernelGGL(k_stem_wgrad, dim3(c->nparts), dim3(256), lds, (hipStream_t)stream, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ---- input gradient of the stem: dL/d image (round 5) --------------------------------------------------------------------------
// Not on the training path (train.py:427: images do not require grad); it exists so that autograd through the drop-in module is
// complete (saliency maps, adversarial examples): ATen's conv2d input gradient of ConvBlock(3, Co, 3, stride 2) (mnasnet.py:179)
// with dy-on-load, as a plain gather -- thread = one image pixel, its 1 / 2 / 4 contributing output pixels (by the parity of the
// row and the column) x Co channels x 3 planes.  dy and the weights are rounded to bf16 like every other input gradient of the
// path; fp32 accumulate, fp32 NCHW result; with the fused input normalisation the result is scaled by in_affine[0][c].
__global__ __launch_bounds__(256) void k_stem_dgrad(MnasGradIn d, const float* __restrict__ w, int N, int H, int W, int Ho, int Wo, int Co,
                                                    const float* __restrict__ in_affine, float* __restrict__ dx) {
    extern __shared__ float sd_s[];                                  // [27][Co] weights (bf16-rounded), [5][Co] dy coefficients
    float* w_s = sd_s;
    float* c_s = sd_s + 27 * Co;
    for (int i = threadIdx.x; i < 27 * Co; i += 256) {
        const int k = i / Co, co = i - k * Co;
        w_s[i] = bf_lo(pack_bf16(w[(size_t)co * 27 + k], 0.f));
    }
    for (int i = threadIdx.x; i < 5 * Co; i += 256) c_s[i] = d.coef[i];
    __syncthreads();
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * H * W) return;
    const int iw = (int)(idx % W), ih = (int)((idx / W) % H), n = (int)(idx / ((long long)W * H));
    float acc[3] = {0.f, 0.f, 0.f};
    for (int kh = 0; kh < 3; ++kh) {
        const int t = ih + 1 - kh;
        if (t < 0 || (t & 1) || (t >> 1) >= Ho) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int u = iw + 1 - kw;
            if (u < 0 || (u & 1) || (u >> 1) >= Wo) continue;
            const size_t base = (((size_t)n * Ho + (t >> 1)) * Wo + (u >> 1)) * Co;
            const int tap = kh * 3 + kw;
            for (int c8 = 0; c8 < Co; c8 += 8) {
                const uint4 g8 = *(const uint4*)((const uint16_t*)d.g + base + c8);
                const uint4 y8 = *(const uint4*)((const uint16_t*)d.y + base + c8);
                float o[8];
                dy8(g8, y8, c_s + c8, c_s + Co + c8, c_s + 2 * Co + c8, c_s + 3 * Co + c8, c_s + 4 * Co + c8, o);
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const uint32_t pk = pack_bf16(o[j], o[j + 1]);
                    const float d0 = bf_lo(pk), d1 = bf_hi(pk);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        acc[c] = fmaf(d1, w_s[(c * 9 + tap) * Co + c8 + j + 1], fmaf(d0, w_s[(c * 9 + tap) * Co + c8 + j], acc[c]));
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        dx[(((size_t)n * 3 + c) * H + ih) * W + iw] = acc[c] * (in_affine ? in_affine[c] : 1.f);
}
extern "C" int mnas_stem_dgrad(const MnasGradIn* dy, const float* w, int N, int H, int W, int Ho, int Wo, int Co, const float* in_affine,
                               float* dx, void* stream) {
    if (!dy || !dy->g || !dy->y || !dy->coef || !w || !dx || N < 1 || H < 1 || W < 1 || Co < 8 || (Co & 7) || Co > 128) return MNAS_EINVAL;
    if (Ho != (H - 1) / 2 + 1 || Wo != (W - 1) / 2 + 1) return MNAS_EINVAL;           // 3x3, stride 2, pad 1
    const long long total = (long long)N * H * W;
    hipLaunchKernelGGL(k_stem_dgrad, dim3((unsigned)((total + 255) / 256)), dim3(256), (size_t)32 * Co * sizeof(float), (hipStream_t)stream,
                       *dy, w, N, H, W, Ho, Wo, Co, in_affine, dx);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
