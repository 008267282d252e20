// Calibration of rocprofv3 FETCH_SIZE for the seed kernel's access pattern (MI355X_MICROARCH.md, HBM section:
// "Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Three kernels over a table far larger than the 256 MiB Infinity Cache:
//   k_rand8   : every lane reads ONE 8-byte word at an independent random address  (dir / hs lookups)
//   k_rand4   : every lane reads ONE 4-byte word at an independent random address  (bitmap lookups)
//   k_stream16: every lane reads 16 B, fully coalesced                              (the guide's calibrated case)
// Run under `rocprofv3 --pmc FETCH_SIZE`; FETCH_SIZE(KB)*1024 / accesses = counted bytes per random access.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void k_rand8(const uint64_t *t, uint64_t words, uint64_t n, uint64_t *out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; if (i >= n) return;
    uint64_t v = t[mix(i + 1) % words]; if (v == 0x1234567) out[0] = v;
}
__global__ void k_rand4(const uint32_t *t, uint64_t words, uint64_t n, uint64_t *out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; if (i >= n) return;
    uint32_t v = t[mix(i + 7) % words]; if (v == 0x1234567) out[0] = v;
}
__global__ void k_stream16(const uint4 *t, uint64_t n, uint64_t *out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; if (i >= n) return;
    uint4 v = t[i]; if (v.x == 0x1234567 && v.y == 1) out[0] = v.z;
}
int main() {
    const uint64_t bytes = 4ull << 30, n = 1ull << 26;
    void *t; uint64_t *out; hipMalloc(&t, bytes); hipMalloc(&out, 8); hipMemset(t, 0, bytes);
    for (int r = 0; r < 2; r++) {
        k_rand8<<<n / 256, 256>>>((const uint64_t *)t, bytes / 8, n, out);
        k_rand4<<<n / 256, 256>>>((const uint32_t *)t, bytes / 4, n, out);
        k_stream16<<<n / 256, 256>>>((const uint4 *)t, n, out);
    }
    hipDeviceSynchronize();
    printf("accesses per launch: %llu (rand8 %llu B, rand4 %llu B, stream16 %llu B algorithmic)\n", (unsigned long long)n,
           (unsigned long long)n * 8, (unsigned long long)n * 4, (unsigned long long)n * 16);
    return 0;
}
