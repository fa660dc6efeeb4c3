// probe: __builtin_amdgcn_global_load_lds (16 B/lane), masked lanes, ordering vs __syncthreads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const uint4* __restrict__ src, uint4* __restrict__ out, int nvalid) {
    __shared__ __attribute__((aligned(16))) uint4 lds[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 256; i += blockDim.x) lds[i] = make_uint4(0xdeadbeef, 0, 0, 0);
    __syncthreads();
    // each wave copies 64 chunks: source is per-lane (reversed order to prove per-lane addressing), dest = base + lane*16
    const uint4* g = src + wave * 64 + (63 - lane);
    if (lane < nvalid)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_ptr_t)(lds + wave * 64), 16, 0, 0);
    __syncthreads();
    out[tid] = lds[tid];
}
int main() {
    const int n = 256;
    std::vector<uint4> h(n);
    for (int i = 0; i < n; ++i) h[i] = make_uint4(i, i * 2, i * 3, i * 4);
    uint4 *d, *o; hipMalloc(&d, n * 16); hipMalloc(&o, n * 16);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int nvalid : {64, 40}) {
        k<<<1, 256>>>(d, o, nvalid);
        std::vector<uint4> r(n); hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
        int bad = 0, stale = 0;
        for (int i = 0; i < n; ++i) {
            int lane = i & 63, wave = i >> 6; int srci = wave * 64 + 63 - lane;
            if (lane < nvalid) { if (r[i].x != (unsigned)srci || r[i].w != (unsigned)srci * 4) bad++; }
            else { if (r[i].x != 0xdeadbeef) stale++; }
        }
        printf("nvalid=%d: wrong=%d, masked-lanes-overwritten=%d  (r[0]=%u r[1]=%u r[63]=%u)\n", nvalid, bad, stale, r[0].x, r[1].x, r[63].x);
    }
    return 0;
}
