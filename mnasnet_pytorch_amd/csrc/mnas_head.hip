// Classifier head and loss of FineTuneModelPool: the nn.Sequential of Dropout / Linear / ReLU over the pooled features
// (classifiers.py:56-89, forward :107-111) and nn.CrossEntropyLoss (train.py:277), forward and backward, fp32.
// Replaces ATen's dropout / addmm / relu / log_softmax / nll_loss kernels (about 25 launches of 3-12 us per step) with
// 2 launches per Linear and direction plus 2 for the loss.
//
// One Linear "layer" here = the Dropout in front of it + the Linear + the optional ReLU behind it:
//     xd = u * keep(seed, idx) / (1 - p)        u: [N][I] the layer's input (pooled features or the previous layer's output)
//     z  = xd W^T + b                            W: [O][I]
//     u' = relu(z) or z
// Only u' is stored; the dropout mask is a counter-based hash of (seed, element index), recomputed wherever it is needed
// (forward operand load, weight-gradient operand load, input-gradient epilogue), never written to memory.
// Backward of a layer, given dz = dL/dz:  dW = dz^T xd,  db = sum_n dz,  du = (dz W) * keep/(1-p),  and for the layer in
// front (if it ends in a ReLU)  dz_prev = du * [u > 0].
//
// The three products are one tiled fp32 FMA GEMM with run-time strides (32x32 tile per workgroup, K split over its 8
// waves); at N = 256 these are 0.1-0.3 GFLOP launches of 80-512 workgroups; fp32 FMA (not bf16 MFMA) keeps the head
// comparable with the reference's fp32 head at 2e-5.  Deterministic:
// fixed summation order, no atomics.
#include "mnas_common.h"

// splitmix64 of (seed, index): P(keep) = 1 - p with thresh = p * 2^32  (oracle/mnasnet_oracle.py: head_dropout_keep)
__device__ __forceinline__ bool head_keep(unsigned long long seed, unsigned long long idx, unsigned int thresh) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned int)(z >> 32) >= thresh;
}

struct HeadGemmArgs {
    int M, N, K;
    const float* A; long long sam, sak;      // A(m,k) = A[m*sam + k*sak]
    const float* B; long long sbk, sbn;      // B(k,n) = B[k*sbk + n*sbn]
    float* C; long long ldc;                 // C[m*ldc + n]
    const float* bias;                       // [N] or NULL
    int relu, accumulate;
    int drop_where;                          // 0 none, 1 on A (element m*K+k), 2 on B (element k*N+n), 3 on C (element m*N+n)
    unsigned int drop_thresh; float drop_scale; unsigned long long seed;
    const float* relu_mask; long long ldm;   // epilogue: C *= [relu_mask[m*ldm+n] > 0], or NULL
    float* rowsum; int rowsum_acc;           // rowsum[m] (+)= sum_k A(m,k)  (written by the n-tile-0 workgroups), or NULL
};

// 32x32 output tile per workgroup of 8 waves, K in steps of 128 staged through LDS (k-contiguous rows; the next step's
// operands are loaded into registers while the current one is multiplied).  Each wave multiplies an EIGHTH of every K step
// for the whole tile (4x4 outputs per lane, rows ty+8i / columns tx+8j so that the 16-byte LDS reads of a wave fall into
// distinct banks) and the eight partial tiles are added in wave order at the end: a single wave walking K = 1000 alone is
// bound by its own instruction stream (40-60 us measured), not by memory.  VEC: 16-byte global loads along whichever
// index is contiguous in memory (needs that extent and the leading strides to be multiples of 4).
template <bool VEC>
__global__ __launch_bounds__(512) void k_head_gemm(HeadGemmArgs a) {
    constexpr int BM = 32, BK = 128, LDK = 132, NV = BM * BK / 4 / 512, KW = BK / 8;
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDK];
    float* As = smem;                         // [32 m][LDK]
    float* Bs = smem + BM * LDK;              // [32 n][LDK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = lane & 7, ty = lane >> 3;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BM;
    float acc[4][4], asum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    }
    const bool a_kfast = a.sak == 1, b_kfast = a.sbk == 1;
    const bool do_rowsum = a.rowsum != nullptr && blockIdx.x == 0;
    float4 ra[NV], rb[NV];
    // one operand: X(r, k) = P[r*sr + k*sk], r in [r0, r0+32) (bounded by R), dropout element index r*er + k*ek
    auto load_op = [&](float4* reg, const float* P, long long sr, long long sk, bool kfast, int r0, int R, int k0, bool drop,
                       long long er, long long ek) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int q = tid + 512 * e;                     // float4 index in the 32 x 128 tile
            // kfast: 4 consecutive k of one row; else: 4 consecutive rows at one k
            const int rr = kfast ? q / (BK / 4) : (q % (BM / 4)) * 4, kk = kfast ? (q % (BK / 4)) * 4 : q / (BM / 4);
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            const int r = r0 + rr, k = k0 + kk;
            const bool full = kfast ? (r < R && k + 3 < a.K) : (r + 3 < R && k < a.K);
            if (VEC && full) {
                const float4 t = *(const float4*)(P + r * sr + k * sk);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ri = kfast ? r : r + i, ki = kfast ? k + i : k;
                    if (ri < R && ki < a.K) v[i] = P[ri * sr + ki * sk];
                }
            }
            if (drop) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ri = kfast ? r : r + i, ki = kfast ? k + i : k;
                    v[i] = head_keep(a.seed, (unsigned long long)(ri * er + ki * ek), a.drop_thresh) ? v[i] * a.drop_scale : 0.f;
                }
            }
            reg[e] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    auto store_op = [&](float* T, const float4* reg, bool kfast) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int q = tid + 512 * e;
            const int rr = kfast ? q / (BK / 4) : (q % (BM / 4)) * 4, kk = kfast ? (q % (BK / 4)) * 4 : q / (BM / 4);
            if (kfast) {
                *(float4*)(T + rr * LDK + kk) = reg[e];
            } else {
                T[(rr + 0) * LDK + kk] = reg[e].x; T[(rr + 1) * LDK + kk] = reg[e].y;
                T[(rr + 2) * LDK + kk] = reg[e].z; T[(rr + 3) * LDK + kk] = reg[e].w;
            }
        }
    };
    auto load = [&](int k0) {
        load_op(ra, a.A, a.sam, a.sak, a_kfast, m0, a.M, k0, a.drop_where == 1, a.K, 1);
        load_op(rb, a.B, a.sbn, a.sbk, b_kfast, n0, a.N, k0, a.drop_where == 2, 1, a.N);
    };
    load(0);
    for (int k0 = 0; k0 < a.K; k0 += BK) {
        store_op(As, ra, a_kfast);
        store_op(Bs, rb, b_kfast);
        __syncthreads();
        if (k0 + BK < a.K) load(k0 + BK);
#pragma unroll
        for (int kq = 0; kq < KW; kq += 4) {
            const int kk = wave * KW + kq;
            float4 av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = *(const float4*)(As + (ty + 8 * i) * LDK + kk);
                bv[i] = *(const float4*)(Bs + (tx + 8 * i) * LDK + kk);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ar[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
                if (do_rowsum) asum[i] += (ar[0] + ar[1]) + (ar[2] + ar[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float br[4] = {bv[j].x, bv[j].y, bv[j].z, bv[j].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j] = fmaf(ar[c], br[c], acc[i][j]);
                }
            }
        }
        __syncthreads();
    }
    // ---- the eight waves' partial tiles -> LDS, summed in wave order (the operand tiles are free: the loop ended with a barrier)
    float* part = smem;                      // [8][32][33] = 8448 floats (the operand tiles hold 8448)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(wave * 32 + ty + 8 * i) * 33 + tx + 8 * j] = acc[i][j];
    __syncthreads();
    float outv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int q = tid + 512 * e, mm = q >> 5, nn = q & 31;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += part[(w * 32 + mm) * 33 + nn];
        outv[e] = v;
    }
    if (do_rowsum) {                          // uniform per workgroup
        __syncthreads();
        float* psum = smem;                  // [8][32]
        if (tx == 0)
#pragma unroll
            for (int i = 0; i < 4; ++i) psum[wave * 32 + ty + 8 * i] = asum[i];
        __syncthreads();
        if (tid < 32 && m0 + tid < a.M) {
            float rs = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) rs += psum[w * 32 + tid];
            a.rowsum[m0 + tid] = a.rowsum_acc ? a.rowsum[m0 + tid] + rs : rs;
        }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int q = tid + 512 * e, mm = q >> 5, nn = q & 31;
        const int m = m0 + mm, n = n0 + nn;
        if (m >= a.M || n >= a.N) continue;
        float v = outv[e];
        if (a.bias) v += a.bias[n];
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.drop_where == 3)
            v = head_keep(a.seed, (unsigned long long)m * a.N + n, a.drop_thresh) ? v * a.drop_scale : 0.f;
        if (a.relu_mask && !(a.relu_mask[m * a.ldm + n] > 0.f)) v = 0.f;
        float* c = a.C + m * a.ldc + n;
        *c = a.accumulate ? *c + v : v;
    }
}

static int head_drop(float p, unsigned int* thresh, float* scale) {
    if (!(p >= 0.f) || p >= 1.f) return MNAS_EINVAL;
    double t = (double)p * 4294967296.0;
    *thresh = t >= 4294967295.0 ? 4294967295u : (unsigned int)t;
    *scale = (float)(1.0 / (1.0 - (double)p));
    return MNAS_OK;
}

static int head_launch(const HeadGemmArgs& a, hipStream_t s) {
    if (a.M < 1 || a.N < 1 || a.K < 1) return MNAS_EINVAL;
    // 16-byte loads: the contiguous index of either operand must cover whole float4s at aligned addresses
    auto vec_ok = [&](const float* P, long long sr, long long sk, int R) {
        const long long lead = sk == 1 ? sr : sk;            // stride of the non-contiguous index
        const int extent = sk == 1 ? a.K : R;                // extent of the contiguous one
        return (sr == 1 || sk == 1) && (lead & 3) == 0 && (extent & 3) == 0 && ((uintptr_t)P & 15) == 0;
    };
    const dim3 grid((a.N + 31) / 32, (a.M + 31) / 32);
    if (vec_ok(a.A, a.sam, a.sak, a.M) && vec_ok(a.B, a.sbn, a.sbk, a.N)) hipLaunchKernelGGL(k_head_gemm<true>, grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL(k_head_gemm<false>, grid, dim3(512), 0, s, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_head_linear_fwd(const MnasHeadLinear* c, void* stream) {
    if (!c || !c->x || !c->w || !c->y || c->N < 1 || c->I < 1 || c->O < 1) return MNAS_EINVAL;
    HeadGemmArgs a = {};
    a.M = c->N; a.N = c->O; a.K = c->I;
    a.A = (const float*)c->x; a.sam = c->I; a.sak = 1;
    a.B = (const float*)c->w; a.sbk = 1; a.sbn = c->I;
    a.C = (float*)c->y; a.ldc = c->O;
    a.bias = (const float*)c->b; a.relu = c->relu ? 1 : 0;
    a.seed = c->seed;
    if (c->drop_p > 0.f) {
        if (head_drop(c->drop_p, &a.drop_thresh, &a.drop_scale) != MNAS_OK) return MNAS_EINVAL;
        a.drop_where = 1;
    }
    return head_launch(a, (hipStream_t)stream);
}

// dW[O][I] (+)= dz^T (x * keep/(1-p)),  db[O] (+)= sum_n dz
extern "C" int mnas_head_linear_bwd_w(const MnasHeadLinear* c, void* stream) {
    if (!c || !c->x || !c->dz || !c->dw || c->N < 1 || c->I < 1 || c->O < 1) return MNAS_EINVAL;
    HeadGemmArgs a = {};
    a.M = c->O; a.N = c->I; a.K = c->N;
    a.A = (const float*)c->dz; a.sam = 1; a.sak = c->O;
    a.B = (const float*)c->x; a.sbk = c->I; a.sbn = 1;
    a.C = (float*)c->dw; a.ldc = c->I; a.accumulate = c->accumulate ? 1 : 0;
    a.rowsum = (float*)c->db; a.rowsum_acc = a.accumulate;
    a.seed = c->seed;
    if (c->drop_p > 0.f) {
        if (head_drop(c->drop_p, &a.drop_thresh, &a.drop_scale) != MNAS_OK) return MNAS_EINVAL;
        a.drop_where = 2;
    }
    return head_launch(a, (hipStream_t)stream);
}

// dx[N][I] = (dz W) * keep/(1-p) * [relu_mask > 0]   (relu_mask: this layer's input u when the layer in front ends in a ReLU)
extern "C" int mnas_head_linear_bwd_x(const MnasHeadLinear* c, void* stream) {
    if (!c || !c->dz || !c->w || !c->dx || c->N < 1 || c->I < 1 || c->O < 1) return MNAS_EINVAL;
    HeadGemmArgs a = {};
    a.M = c->N; a.N = c->I; a.K = c->O;
    a.A = (const float*)c->dz; a.sam = c->O; a.sak = 1;
    a.B = (const float*)c->w; a.sbk = c->I; a.sbn = 1;
    a.C = (float*)c->dx; a.ldc = c->I;
    a.relu_mask = (const float*)c->relu_mask; a.ldm = c->I;
    a.seed = c->seed;
    if (c->drop_p > 0.f) {
        if (head_drop(c->drop_p, &a.drop_thresh, &a.drop_scale) != MNAS_OK) return MNAS_EINVAL;
        a.drop_where = 3;
    }
    return head_launch(a, (hipStream_t)stream);
}

// keep mask of a layer as bytes (tests / oracle cross-check only)
__global__ void k_head_mask(unsigned char* out, long long n, unsigned long long seed, unsigned int thresh) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = head_keep(seed, (unsigned long long)i, thresh) ? 1 : 0;
}
extern "C" int mnas_head_dropout_mask(void* out, int64_t n, float p, uint64_t seed, void* stream) {
    unsigned int thresh; float scale;
    if (!out || n < 1 || head_drop(p, &thresh, &scale) != MNAS_OK) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_head_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)out,
                       (long long)n, (unsigned long long)seed, thresh);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ---- nn.CrossEntropyLoss(reduction='mean', ignore_index): one workgroup per row --------------------------------------
// loss_rows[n] = logsumexp(x[n]) - x[n][t],  dlogits[n][c] = (softmax(x[n])[c] - [c == t]) / count   (0 for ignored rows)
// count = rows whose target != ignore_index: every workgroup counts the (few hundred) targets itself.
__device__ __forceinline__ float head_wg_reduce(float v, float* red, bool is_max) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(256) void k_head_ce(const float* x, const long long* target, int N, int C, long long ignore_index,
                                                 float* loss_rows, float* dlogits, int* bad) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* row = x + (size_t)n * C;
    float cnt = 0.f;
    for (int i = tid; i < N; i += 256) cnt += target[i] != ignore_index ? 1.f : 0.f;
    cnt = head_wg_reduce(cnt, red, false);
    const long long t = target[n];
    const bool ignored = t == ignore_index;
    if (!ignored && (t < 0 || t >= C)) {            // ATen asserts on the device; here: flag it, treat the row as ignored
        if (tid == 0) atomicExch(bad, 1);
    }
    const bool valid = !ignored && t >= 0 && t < C;
    float mx = -INFINITY;
    for (int c = tid; c < C; c += 256) mx = fmaxf(mx, row[c]);
    mx = head_wg_reduce(mx, red, true);
    float se = 0.f;
    for (int c = tid; c < C; c += 256) se += expf(row[c] - mx);
    se = head_wg_reduce(se, red, false);
    const float inv = valid && cnt > 0.f ? 1.f / cnt : 0.f, rse = 1.f / se;
    if (dlogits)
        for (int c = tid; c < C; c += 256) {
            const float p = expf(row[c] - mx) * rse;
            dlogits[(size_t)n * C + c] = valid ? (p - (c == (int)t ? 1.f : 0.f)) * inv : 0.f;
        }
    if (tid == 0) loss_rows[n] = valid ? (logf(se) + mx - row[t]) : (ignored ? 0.f : NAN);   // out-of-range target: poison the loss
}

// loss = sum(loss_rows) / count, fixed order (thread-strided partials, then a tree)
__global__ __launch_bounds__(256) void k_head_loss_mean(const float* loss_rows, const long long* target, int N, long long ignore_index,
                                                        float* loss) {
    __shared__ float red[4];
    float s = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) { s += loss_rows[i]; cnt += target[i] != ignore_index ? 1.f : 0.f; }
    s = head_wg_reduce(s, red, false);
    cnt = head_wg_reduce(cnt, red, false);
    if (threadIdx.x == 0) *loss = s / cnt;       // 0/0 = nan when every row is ignored, as ATen
}

extern "C" int mnas_head_cross_entropy(const void* logits, const void* target, int N, int C, int64_t ignore_index,
                                       void* loss_rows, void* loss, void* dlogits, void* bad_flag, void* stream) {
    if (!logits || !target || !loss_rows || !loss || !bad_flag || N < 1 || C < 1) return MNAS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_head_ce, dim3(N), dim3(256), 0, s, (const float*)logits, (const long long*)target, N, C,
                       (long long)ignore_index, (float*)loss_rows, (float*)dlogits, (int*)bad_flag);
    MNAS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_head_loss_mean, dim3(1), dim3(256), 0, s, (const float*)loss_rows, (const long long*)target, N,
                       (long long)ignore_index, (float*)loss);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
