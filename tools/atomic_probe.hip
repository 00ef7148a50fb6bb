// What does an atomic on ONE address cost when every wave of a launch issues one?  (11.3 ns device-wide per wave-instruction on MI355X, with or without a
// returned value, 1 or 8 lanes: profiles/r06_atomic_probe.txt — the reason k_bvb_children numbers a level's children with one atomic per 1 024 nodes.)
// build + run:  hipcc --offload-arch=gfx950 -O2 tools/atomic_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_same(unsigned *c, unsigned *out, int per_wave_lanes) {
    unsigned lane = threadIdx.x & 63u;
    unsigned v = 0;
    if ((int)lane < per_wave_lanes) v = atomicAdd(c, 2u);
    if (v == 0xffffffffu) out[0] = v;
}
__global__ void k_noret(unsigned *c, int per_wave_lanes) {
    unsigned lane = threadIdx.x & 63u;
    if ((int)lane < per_wave_lanes) atomicMax(c + 1, blockIdx.x & 7u);
}
int main() {
    unsigned *c, *o; hipMalloc(&c, 64); hipMalloc(&o, 64); hipMemset(c, 0, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int lanes : {1, 1, 8}) for (int blocks : {51000, 411000}) {
        hipEventRecord(a); k_same<<<blocks, 64>>>(c, o, lanes); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("returning add: %d waves x %d lanes: %.3f ms = %.1f ns per atomic\n", blocks, lanes, ms, ms * 1e6 / ((double)blocks * lanes));
        hipEventRecord(a); k_noret<<<blocks, 64>>>(c, lanes); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); printf("no-return max: %d waves x %d lanes: %.3f ms = %.1f ns per atomic\n", blocks, lanes, ms, ms * 1e6 / ((double)blocks * lanes));
    }
    hipEventRecord(a); k_noret<<<411000, 64>>>(c, 0); hipEventRecord(b); hipEventSynchronize(b);
    float ms0; hipEventElapsedTime(&ms0, a, b); printf("empty 411000 waves: %.3f ms\n", ms0);
    return 0;
}
