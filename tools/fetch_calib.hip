// fetch_calib.hip — what one unit of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ means on gfx950 for the access shapes
// band_run_kernel uses (8-byte loads at per-lane addresses), measured against byte counts known by construction.
// MI355X_MICROARCH.md calibrates FETCH_SIZE for ONE shape only (16 B/lane coalesced streaming: counter = half the bytes);
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".
//
// Every pattern is its own kernel (the counter CSV is per kernel name) over a fresh region of a buffer far larger
// than the 256 MiB Infinity Cache, so every line comes from HBM exactly once:
//   calib_stream16     16 B / lane, coalesced (the guide's calibrated case)
//   calib_stream8      8 B / lane, coalesced (a wavefront reads 512 contiguous bytes per instruction)
//   calib_lane_seq8    every lane walks its OWN 160-byte record in 8-byte steps, records back to back (band_run_kernel's
//                      read windows: a wave instruction touches 64 different 160-byte records)
//   calib_sector64     8 B out of every 64 B   (one load per 64-byte sector)
//   calib_sector128    8 B out of every 128 B  (one load per 128-byte line)
//   calib_sector256    8 B out of every 256 B  (every other line untouched)
// Output: one line per kernel with the bytes the lanes asked for and the bytes of the 64-B sectors / 128-B lines they
// touched; tools/pmc_summarize.py --calib joins it with the counter CSV.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void calib_stream16(const uint4* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
__global__ void calib_stream8(const uint2* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 v = p[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) *out = acc;
}
// record r (160 bytes) belongs to lane r; 20 loads of 8 bytes each, one per loop trip
__global__ void calib_lane_seq8(const uint8_t* __restrict__ p, size_t n_rec, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += (size_t)gridDim.x * blockDim.x) {
        const uint2* q = (const uint2*)(p + r * 160);
#pragma unroll 1
        for (int t = 0; t < 20; ++t) { const uint2 v = q[t]; acc ^= v.x ^ v.y; asm volatile("" ::"v"(acc)); }
    }
    if (acc == 0x12345678u) *out = acc;
}
template <int STRIDE>
__global__ void calib_sector(const uint8_t* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 v = *(const uint2*)(p + i * STRIDE);
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) *out = acc;
}
__global__ void calib_fill(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (uint32_t)(i * 2654435761u);
}

int main() {
    const size_t region = (size_t)1 << 30;            // 1 GiB per pattern: 4x the Infinity Cache
    const int n_pat = 6;
    uint8_t* buf = nullptr;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&buf, region * n_pat));
    CHECK(hipMalloc(&out, 4));
    hipLaunchKernelGGL(calib_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)buf, region * n_pat / 4);
    CHECK(hipDeviceSynchronize());
    const dim3 g(256 * 8), b(256);
    uint8_t* r = buf;
    hipLaunchKernelGGL(calib_stream16, g, b, 0, 0, (const uint4*)r, region / 16, out); r += region;
    hipLaunchKernelGGL(calib_stream8, g, b, 0, 0, (const uint2*)r, region / 8, out); r += region;
    const size_t n_rec = region / 160;
    hipLaunchKernelGGL(calib_lane_seq8, g, b, 0, 0, r, n_rec, out); r += region;
    hipLaunchKernelGGL(calib_sector<64>, g, b, 0, 0, r, region / 64, out); r += region;
    hipLaunchKernelGGL(calib_sector<128>, g, b, 0, 0, r, region / 128, out); r += region;
    hipLaunchKernelGGL(calib_sector<256>, g, b, 0, 0, r, region / 256, out); r += region;
    CHECK(hipDeviceSynchronize());
    // kernel, bytes requested by the lanes, bytes of the 64-B sectors touched, bytes of the 128-B lines touched
    printf("CALIB calib_stream16 %zu %zu %zu\n", region, region, region);
    printf("CALIB calib_stream8 %zu %zu %zu\n", region, region, region);
    printf("CALIB calib_lane_seq8 %zu %zu %zu\n", n_rec * 160, n_rec * 160, n_rec * 160);
    printf("CALIB calib_sector<64> %zu %zu %zu\n", region / 8, region, region);
    printf("CALIB calib_sector<128> %zu %zu %zu\n", region / 16, region / 2, region);
    printf("CALIB calib_sector<256> %zu %zu %zu\n", region / 32, region / 4, region / 2);
    return 0;
}
