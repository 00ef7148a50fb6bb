// What does a lane-divergent load cost the texture-address path of a gfx950 CU?  (round 4: the streamed global-memory walks keep that unit 83-92 % busy.)
// Build: hipcc --offload-arch=gfx950 -O3 -o ta_rates ta_rates.hip ; run on the GPU box: ta_rates [waves per SIMD]
// Every lane reads its own 16-byte-aligned address inside a 16 KB window (L1 hits after the first trips), 16 independent loads per trip; per width
// (4 / 8 / 16 bytes per lane) and EXEC mask (64, 32 contiguous, 32 alternating, 16 lanes).  Output: CU cycles per wave load instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 4096
#define WINDOW_BYTES 16384

template <typename T> __device__ __forceinline__ uint32_t fold(T v);
template <> __device__ __forceinline__ uint32_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ __forceinline__ uint32_t fold<uint2>(uint2 v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ uint32_t fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <typename T, int MASK_KIND>
__global__ __launch_bounds__(256) void k_loads(const unsigned char *buf, uint32_t *out, uint32_t salt) {
    const uint32_t lane = threadIdx.x & 63u;
    bool on = true;
    if (MASK_KIND == 1) on = lane < 32u;
    if (MASK_KIND == 2) on = (lane & 1u) == 0u;
    if (MASK_KIND == 3) on = lane < 16u;
    uint32_t acc = 0u;
    if (on) {
        uint32_t h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u) ^ salt;
        for (int i = 0; i < ITER; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                h = h * 1664525u + 1013904223u;                       // a new divergent address per load, independent of the loaded data
                const uint32_t off = (h >> 8) & (WINDOW_BYTES - 16u);
                acc ^= fold<T>(*reinterpret_cast<const T *>(buf + off));
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the same address for every lane (what a wave-uniform node visit looks like to the vector path)
template <typename T>
__global__ __launch_bounds__(256) void k_loads_uniform(const unsigned char *buf, uint32_t *out, uint32_t salt) {
    uint32_t acc = 0u;
    uint32_t h = (blockIdx.x * 40503u) ^ salt ^ ((threadIdx.x >> 6) * 977u);
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            h = h * 1664525u + 1013904223u;
            const uint32_t off = (h >> 8) & (WINDOW_BYTES - 16u);
            acc ^= fold<T>(*reinterpret_cast<const T *>(buf + off + (threadIdx.x & 0u)));
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * wps;          // 256 threads = one wave per SIMD per block
    unsigned char *buf; uint32_t *out;
    (void)hipMalloc(&buf, WINDOW_BYTES); (void)hipMemset(buf, 1, WINDOW_BYTES); (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    struct Case { const char *name; void (*fn)(const unsigned char *, uint32_t *, uint32_t); double lanes; };
    Case cases[] = {
        {"dword  exec 64", k_loads<uint32_t, 0>, 64}, {"dword  exec 32 (low half)", k_loads<uint32_t, 1>, 32}, {"dword  exec 32 (every other)", k_loads<uint32_t, 2>, 32}, {"dword  exec 16", k_loads<uint32_t, 3>, 16},
        {"dwordx2 exec 64", k_loads<uint2, 0>, 64}, {"dwordx2 exec 32 (low half)", k_loads<uint2, 1>, 32}, {"dwordx2 exec 16", k_loads<uint2, 3>, 16},
        {"dwordx4 exec 64", k_loads<uint4, 0>, 64}, {"dwordx4 exec 32 (low half)", k_loads<uint4, 1>, 32}, {"dwordx4 exec 32 (every other)", k_loads<uint4, 2>, 32}, {"dwordx4 exec 16", k_loads<uint4, 3>, 16},
        {"dword   one address per wave", k_loads_uniform<uint32_t>, 64}, {"dwordx4 one address per wave", k_loads_uniform<uint4>, 64},
    };
    printf("waves/SIMD %d, %d CUs, clock %d MHz\n", wps, cus, khz / 1000);
    for (auto &c : cases) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0); c.fn<<<blocks, 256>>>(buf, out, (uint32_t)r); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // wave load instructions per CU = 4 SIMDs x wps waves x ITER x 16
        const double per_cu = 4.0 * wps * (double)ITER * 16.0;
        printf("%-34s %8.3f ms   %6.2f CU cycles per wave load instruction\n", c.name, best, best * 1e-3 * khz * 1e3 / per_cu);
    }
    return 0;
}
