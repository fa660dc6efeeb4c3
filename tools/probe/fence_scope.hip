// Probe (compile only): what a cross-workgroup "last arriver finalizes" ticket costs on gfx950.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o - tools/probe/fence_scope.hip | grep -E "global_|buffer_|s_waitcnt"
// k_fence:  plain stores + __threadfence() + atomicAdd  ->  buffer_wbl2 sc1 / buffer_inv sc1 around the atomic (L2 write-back +
//           invalidate per workgroup: the L2s of the 8 XCDs are not coherent with each other for ordinary device memory);
// k_scoped: agent-scope relaxed atomic stores / loads   ->  global_store_dword ... sc1, global_load_dword ... sc1 and no fence
//           (the stores still need an explicit s_waitcnt vmcnt(0) before the ticket atomic).
// See DESIGN.md section 8 item 4 for why neither form was built into the BatchNorm finalizes.
#include <hip/hip_runtime.h>
__global__ void k_scoped(float* p, unsigned* t, float* out) {
    __hip_atomic_store(p + blockIdx.x, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    unsigned v = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == gridDim.x - 1) {
        float s = 0;
        for (int i = 0; i < gridDim.x; ++i) s += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *out = s;
    }
}
__global__ void k_fence(float* p, unsigned* t, float* out) {
    p[blockIdx.x] = 1.0f;
    __threadfence();
    unsigned v = atomicAdd(t, 1u);
    if (v == gridDim.x - 1) {
        __threadfence();
        float s = 0;
        for (int i = 0; i < gridDim.x; ++i) s += p[i];
        *out = s;
    }
}
