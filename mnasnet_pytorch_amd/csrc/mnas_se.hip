// Squeeze-and-excitation for the MBConv block (BASELINE config 4: "5x5-depthwise + SE-block variant").  BUILD-DEFINED: the
// reference has no SE block anywhere, so the definition is this repo's (MnasNet-A1 form), restated in
// oracle/mnasnet_oracle.py::se_apply -- "parity unpinned by the reference":
//     a2 = relu(bn2(y2))                       the activated depthwise output (N,H,W,E)
//     z  = mean_hw a2                          squeeze          -> mnas_pool_act (existing)
//     h  = relu(fc1 z + b1),  u = fc2 h + b2   excite (R << E)  -> mnas_head_linear_fwd (existing GEMM)
//     a2s = a2 * sigmoid(u)[n][e]              scale            -> k_se_scale: the project conv reads a2s as a plain activation
// Backward (gs = dL/da2s from the project conv's input gradient):
//     du[n][e] = (sum_hw gs * a2) * s (1 - s)                   -> k_se_bwd_reduce
//     MLP backward                                              -> mnas_head_linear_bwd_w / _bwd_x (existing)
//     g_a2 = gs * s + dz[n][e] / HW                             -> k_se_bwd_apply  (dz = dL/dz from the MLP backward)
// All three kernels are elementwise / per-image reductions over the E-wide tensor: HBM-bound (read y2 [+ gs], write one tensor).
#include "mnas_common.h"

__device__ __forceinline__ float se_sigmoid(float u) { return 1.f / (1.f + __expf(-u)); }

// Common geometry: a workgroup = (image n, pixel range); a thread keeps ONE 16-byte channel group for its whole range, so the
// BatchNorm coefficients and sigmoid(u[n][c]) of its 8 channels are registers computed once (the naive form re-evaluated 8
// exponentials per pixel and channel group).  R = 256 / G pixel rows are in flight per workgroup step (G = C / 8).
struct SeGeom { int G, R, HW, C, chunk; };      // chunk = pixels per workgroup (multiple of R)

__global__ __launch_bounds__(256) void k_se_scale(MnasActIn a, const float* __restrict__ u, SeGeom g, uint4* __restrict__ out) {
    const int n = blockIdx.x, cg = threadIdx.x % g.G, pr = threadIdx.x / g.G;
    if (pr >= g.R) return;
    float s[8], t[8], sg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s[j] = a.scale ? a.scale[cg * 8 + j] : 1.f;
        t[j] = a.scale ? a.shift[cg * 8 + j] : 0.f;
        sg[j] = se_sigmoid(u[(size_t)n * g.C + cg * 8 + j]);
    }
    const int p0 = blockIdx.y * g.chunk, p1 = min(g.HW, p0 + g.chunk);
    const uint4* src = (const uint4*)a.data + (size_t)n * g.HW * g.G + cg;
    uint4* dst = out + (size_t)n * g.HW * g.G + cg;
    for (int p = p0 + pr; p < p1; p += g.R) {
        float f[8];
        unpack8(src[(size_t)p * g.G], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = a.scale ? fmaxf(fmaf(f[j], s[j], t[j]), 0.f) : f[j];
            f[j] = v * sg[j];
        }
        dst[(size_t)p * g.G] = pack8(f);
    }
}

// du partial sums: every workgroup reduces its pixel range in registers, its R rows through LDS in fixed order, and writes
// dupart[split][n][c]; k_se_bwd_finish adds the splits in order and applies s (1 - s).  Deterministic.
__global__ __launch_bounds__(256) void k_se_bwd_reduce(const uint4* __restrict__ gs, MnasActIn a, SeGeom g, float* __restrict__ dupart,
                                                       int N) {
    extern __shared__ float red[];                                  // [R][C]
    const int n = blockIdx.x, cg = threadIdx.x % g.G, pr = threadIdx.x / g.G;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pr < g.R) {
        float s[8], t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = a.scale ? a.scale[cg * 8 + j] : 1.f; t[j] = a.scale ? a.shift[cg * 8 + j] : 0.f; }
        const int p0 = blockIdx.y * g.chunk, p1 = min(g.HW, p0 + g.chunk);
        const size_t base = (size_t)n * g.HW * g.G + cg;
        for (int p = p0 + pr; p < p1; p += g.R) {
            float gq[8], y[8];
            unpack8(gs[base + (size_t)p * g.G], gq);
            unpack8(((const uint4*)a.data)[base + (size_t)p * g.G], y);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = a.scale ? fmaxf(fmaf(y[j], s[j], t[j]), 0.f) : y[j];
                acc[j] = fmaf(gq[j], v, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[pr * g.C + cg * 8 + j] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < g.C; c += 256) {
        float v = 0.f;
        for (int r = 0; r < g.R; ++r) v += red[r * g.C + c];
        dupart[((size_t)blockIdx.y * N + n) * g.C + c] = v;
    }
}
__global__ __launch_bounds__(256) void k_se_bwd_finish(const float* __restrict__ dupart, const float* __restrict__ u, int splits, int NC,
                                                       float* __restrict__ du) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NC) return;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += dupart[(size_t)s * NC + i];
    const float sg = se_sigmoid(u[i]);
    du[i] = v * sg * (1.f - sg);
}

// RED: also the BatchNorm-backward reduce of the depthwise conv whose activated output `out` is the gradient of (red_y = its raw
// output, red_bn = its bnbuf): partial (sum dz, sum dz*xhat) per workgroup, dz = out * [s*y+t > 0] with out AS STORED (bf16) --
// exactly what mnas_bn_bwd_reduce would compute from (out, red_y, red_bn) in a separate pass over both tensors.
template <bool RED>
__global__ __launch_bounds__(256) void k_se_bwd_apply(const uint4* __restrict__ gs, const float* __restrict__ u, const float* __restrict__ dz,
                                                      SeGeom g, uint4* __restrict__ out, const uint4* __restrict__ red_y,
                                                      const float* __restrict__ red_bn, float* __restrict__ red_partial) {
    extern __shared__ float red[];                                  // RED: [R][2][C]
    const int n = blockIdx.x, cg = threadIdx.x % g.G, pr = threadIdx.x / g.G;
    float r1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, r2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pr < g.R) {
        float sg[8], zz[8], bs[8], bt[8], bi[8], bm[8];
        const float inv = 1.f / (float)g.HW;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            sg[j] = se_sigmoid(u[(size_t)n * g.C + c]);
            zz[j] = dz[(size_t)n * g.C + c] * inv;
            if (RED) {
                bs[j] = red_bn[c]; bt[j] = red_bn[g.C + c]; bi[j] = red_bn[6 * g.C + c];
                bm[j] = -red_bn[5 * g.C + c] * red_bn[6 * g.C + c];
            }
        }
        const int p0 = blockIdx.y * g.chunk, p1 = min(g.HW, p0 + g.chunk);
        const size_t base = (size_t)n * g.HW * g.G + cg;
        for (int p = p0 + pr; p < p1; p += g.R) {
            float gq[8];
            unpack8(gs[base + (size_t)p * g.G], gq);
#pragma unroll
            for (int j = 0; j < 8; ++j) gq[j] = fmaf(gq[j], sg[j], zz[j]);
            const uint4 o = pack8(gq);
            out[base + (size_t)p * g.G] = o;
            if (RED) {
                float y[8], q[8];
                unpack8(red_y[base + (size_t)p * g.G], y);
                unpack8(o, q);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = (fmaf(y[j], bs[j], bt[j]) > 0.f) ? q[j] : 0.f;
                    r1[j] += d;
                    r2[j] = fmaf(d, fmaf(y[j], bi[j], bm[j]), r2[j]);
                }
            }
        }
    }
    if (RED) {
        if (pr < g.R) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[(pr * 2 + 0) * g.C + cg * 8 + j] = r1[j]; red[(pr * 2 + 1) * g.C + cg * 8 + j] = r2[j]; }
        }
        __syncthreads();
        const int col = blockIdx.x * gridDim.y + blockIdx.y, ncols = gridDim.x * gridDim.y;
        for (int i = threadIdx.x; i < 2 * g.C; i += 256) {
            const int r = i / g.C, c = i - r * g.C;
            float v = 0.f;
            for (int k = 0; k < g.R; ++k) v += red[(k * 2 + r) * g.C + c];           // fixed order: deterministic
            red_partial[((size_t)r * g.C + c) * ncols + col] = v;
        }
    }
}

// ---- excitation applied ON LOAD (round 4): the project conv reads relu(bn2(y2)) * sigmoid(u)[n][c] through MnasConvGemm.gate
// instead of a materialised a2s (k_se_scale's read + write and the project conv's second read of an E-wide tensor go away).
__global__ __launch_bounds__(256) void k_se_gate(const float* __restrict__ u, int NC, float* __restrict__ gate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NC) gate[i] = se_sigmoid(u[i]);
}
extern "C" int mnas_se_gate(const float* u, int N, int C, float* gate, void* stream) {
    if (!u || !gate || N < 1 || C < 1) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_gate, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, u, N * C, gate);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// Backward of the gated project conv WITHOUT a pass over the E-wide tensors for du.  mnas_pw_bwd ran in segment mode on the
// UNGATED activation a2 = relu(bn2(y2)), `kseg` workgroups per image: wpartial[n*kseg + j][o][c] = sum over that pixel range of
// dy3[pix][o] * a2[pix][c].  With gs = dy3 . W3 (the project conv's input gradient, = dL/d(a2*s)):
//     du_pre[n][c] = sum_pix gs[pix][c] * a2[pix][c] = sum_o W3[o][c] * (sum_j wpartial[n*kseg+j][o][c])
//     dW3[o][c]    = sum_n s[n][c] * (sum_j wpartial[n*kseg+j][o][c])                      (a2s = a2 * s)
// k_se_proj_du: one workgroup per image; folds the image's kseg slabs into slab n*kseg scaled by s (what k_se_proj_dw then adds
// over the images) and writes du = du_pre * s (1 - s).  Both sums run in a fixed order: deterministic.
__global__ __launch_bounds__(256) void k_se_proj_du(float* __restrict__ wpartial, int kseg, int Co, int Ci, const float* __restrict__ u,
                                                    const float* __restrict__ W, float* __restrict__ du) {
    // thread = (input channel c, output-channel group og): all 256 threads work whatever Ci is (the squeeze-excite project convs
    // have Ci = 48 / 72 / 240: one thread per channel left 19-94 % of the workgroup idle on a launch that sits on the critical
    // path); the og partial sums of a channel meet in LDS and are added in group order: deterministic.  W is the fp32 master
    // weight (gs was formed with its bf16 rounding: the difference is inside the gradient tolerances, include/mnas.h says so).
    __shared__ float part[256];
    const int n = blockIdx.x;
    const size_t slab = (size_t)Co * Ci;
    float* base = wpartial + (size_t)n * kseg * slab;
    if (Ci <= 256) {
        const int G = 256 / Ci;
        const int c = threadIdx.x % Ci, og = threadIdx.x / Ci;
        const float sg = se_sigmoid(u[(size_t)n * Ci + c]);
        float acc = 0.f;
        if (og < G) {
            for (int o = og; o < Co; o += G) {
                float p = base[(size_t)o * Ci + c];
                for (int j = 1; j < kseg; ++j) p += base[j * slab + (size_t)o * Ci + c];
                acc = fmaf(W[(size_t)o * Ci + c], p, acc);
                base[(size_t)o * Ci + c] = p * sg;
            }
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        if (og == 0) {
            float a = part[c];
            for (int g = 1; g < G; ++g) a += part[g * Ci + c];
            du[(size_t)n * Ci + c] = a * sg * (1.f - sg);
        }
        return;
    }
    for (int c = threadIdx.x; c < Ci; c += 256) {
        const float sg = se_sigmoid(u[(size_t)n * Ci + c]);
        float acc = 0.f;
        for (int o = 0; o < Co; ++o) {
            float p = base[(size_t)o * Ci + c];
            for (int j = 1; j < kseg; ++j) p += base[j * slab + (size_t)o * Ci + c];
            acc = fmaf(W[(size_t)o * Ci + c], p, acc);
            base[(size_t)o * Ci + c] = p * sg;
        }
        du[(size_t)n * Ci + c] = acc * sg * (1.f - sg);
    }
}
// dW[i] (+)= sum_n wpartial[n*kseg][i]: thread column tx = element, 8 row groups ty combined through LDS in fixed order
__global__ __launch_bounds__(256) void k_se_proj_dw(const float* __restrict__ wpartial, int N, size_t nstride, int total,
                                                    float* __restrict__ dW, int accumulate) {
    __shared__ float red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (i < total) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int n = ty; n < N; n += 32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += (n + 8 * j < N) ? wpartial[(size_t)(n + 8 * j) * nstride + i] : 0.f;
        }
        s = (a[0] + a[1]) + (a[2] + a[3]);
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < total) {
#pragma unroll
        for (int j = 1; j < 8; ++j) s += red[j][tx];
        dW[i] = (accumulate ? dW[i] : 0.f) + s;
    }
}
extern "C" int mnas_se_proj_finalize(float* wpartial, int N, int kseg, int Co, int Ci, const float* u, const float* W, float* dW,
                                     int accumulate, float* du, void* stream) {
    if (!wpartial || !u || !W || !dW || !du || N < 1 || kseg < 1 || Co < 1 || Ci < 1) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_proj_du, dim3(N), dim3(256), 0, (hipStream_t)stream, wpartial, kseg, Co, Ci, u, W, du);
    const int total = Co * Ci;
    hipLaunchKernelGGL(k_se_proj_dw, dim3((total + 31) / 32), dim3(256), 0, (hipStream_t)stream, wpartial,
                       N, (size_t)kseg * total, total, dW, accumulate);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// pixel splits per image: enough workgroups to fill the chip (>= ~2048), whole multiples of R pixels each
static bool se_geom(int N, int HW, int C, SeGeom* g, int* splits) {
    if (N < 1 || HW < 1 || C < 8 || (C & 7) || C > 2048) return false;
    g->G = C >> 3; g->R = 256 / g->G; g->HW = HW; g->C = C;
    if (g->R < 1) return false;
    int sp = 2048 / N;                                               // N * splits <= 2048: the fused reduce's table has one column per workgroup
    const int maxsp = (HW + g->R * 4 - 1) / (g->R * 4);              // at least 4 pixels per thread
    if (sp > maxsp) sp = maxsp;
    if (sp < 1) sp = 1;
    int chunk = (HW + sp - 1) / sp;
    chunk = (chunk + g->R - 1) / g->R * g->R;
    g->chunk = chunk;
    *splits = (HW + chunk - 1) / chunk;
    return true;
}
// bytes of scratch mnas_se_bwd_reduce needs (partial sums float[splits][N][C])
extern "C" int64_t mnas_se_scratch_bytes(int N, int HW, int C) {
    SeGeom g; int sp;
    if (!se_geom(N, HW, C, &g, &sp)) return -1;
    return (int64_t)sp * N * C * sizeof(float);
}
extern "C" int mnas_se_scale(const MnasActIn* a, const float* u, int N, int HW, int C, void* out_bf16, void* stream) {
    SeGeom g; int sp;
    if (!a || !a->data || !u || !out_bf16 || !se_geom(N, HW, C, &g, &sp)) return MNAS_EINVAL;
    if ((a->scale == nullptr) != (a->shift == nullptr)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_scale, dim3(N, sp), dim3(256), 0, (hipStream_t)stream, *a, u, g, (uint4*)out_bf16);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
extern "C" int mnas_se_bwd_reduce(const void* gs, const MnasActIn* a, const float* u, int N, int HW, int C, float* du, float* scratch,
                                  void* stream) {
    SeGeom g; int sp;
    if (!gs || !a || !a->data || !u || !du || !scratch || !se_geom(N, HW, C, &g, &sp)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_bwd_reduce, dim3(N, sp), dim3(256), (size_t)g.R * C * sizeof(float), (hipStream_t)stream, (const uint4*)gs, *a, g,
                       scratch, N);
    hipLaunchKernelGGL(k_se_bwd_finish, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, u, sp, N * C, du);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
extern "C" int mnas_se_bwd_apply(const void* gs, const float* u, const float* dz, int N, int HW, int C, void* out_bf16,
                                 const void* red_y, const float* red_bn, float* red_partial, void* stream) {
    SeGeom g; int sp;
    if (!gs || !u || !dz || !out_bf16 || !se_geom(N, HW, C, &g, &sp)) return MNAS_EINVAL;
    if (red_partial) {
        if (!red_y || !red_bn) return MNAS_EINVAL;
        hipLaunchKernelGGL(k_se_bwd_apply<true>, dim3(N, sp), dim3(256), (size_t)g.R * 2 * C * sizeof(float), (hipStream_t)stream,
                           (const uint4*)gs, u, dz, g, (uint4*)out_bf16, (const uint4*)red_y, red_bn, red_partial);
    } else {
        hipLaunchKernelGGL(k_se_bwd_apply<false>, dim3(N, sp), dim3(256), 0, (hipStream_t)stream, (const uint4*)gs, u, dz, g,
                           (uint4*)out_bf16, (const uint4*)nullptr, (const float*)nullptr, (float*)nullptr);
    }
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
// columns of the fused reduce table mnas_se_bwd_apply writes (= N * pixel splits), or -1
extern "C" int mnas_se_bwd_apply_cols(int N, int HW, int C) {
    SeGeom g; int sp;
    if (!se_geom(N, HW, C, &g, &sp)) return -1;
    return N * sp;
}
