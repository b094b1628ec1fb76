// residency_census.hip — how many one-wavefront workgroups of a given dynamic-LDS size are resident on the chip at once
// (tools/gpu_campaign.sh does not run it; build + run by hand on the GPU box:
//     hipcc -O2 --offload-arch=gfx950 -o /tmp/census tools/residency_census.hip && /tmp/census 165 19584 20480 24576 32768 38528)
// Every workgroup notes the wall clock (100 MHz) when it starts and after spinning for 2 ms; the host counts how many intervals
// overlap.  Written to find out why band_sweep_kernel (19 584 B of LDS per workgroup: 8 per CU by the 160 KiB rule) ran with the
// residency of round 4's kernel (38 528 B: 4 per CU).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

// V: a register the kernel claims (next_free_vgpr = V + 1), to give it the register footprint of the kernel under study
template <int V>
__global__ __launch_bounds__(64) void census_kernel(unsigned long long* t, unsigned int* xcc) {
    extern __shared__ unsigned int lds[];
    if (V == 100) asm volatile("v_mov_b32 v100, 0" ::: "v100");
    if (V == 130) asm volatile("v_mov_b32 v130, 0" ::: "v130");
    if (V == 165) asm volatile("v_mov_b32 v165, 0" ::: "v165");
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x] = (unsigned int)t0;
    while (wall_clock64() - t0 < 200000ull) { }
    if (threadIdx.x == 0) {
        t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = wall_clock64() + (lds[1] & 0u);
        unsigned int id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[blockIdx.x] = id;
    }
}

int main(int argc, char** argv) {
    const int grid = 4096;
    unsigned long long* d_t; unsigned int* d_x;
    hipMalloc(&d_t, grid * 2 * sizeof(unsigned long long)); hipMalloc(&d_x, grid * sizeof(unsigned int));
    // usage: census VGPRS(0 | 100 | 130 | 165) LDS_BYTES...
    const int vg = argc > 1 ? atoi(argv[1]) : 0;
    for (int a = 2; a < argc; ++a) {
        const int lds = atoi(argv[a]);
#define LAUNCH(V) { if (lds > 48 * 1024) hipFuncSetAttribute((const void*)census_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
                    hipLaunchKernelGGL(census_kernel<V>, dim3(grid), dim3(64), lds, 0, d_t, d_x); }
        if (vg == 100) LAUNCH(100) else if (vg == 130) LAUNCH(130) else if (vg == 165) LAUNCH(165) else LAUNCH(0)
        if (hipDeviceSynchronize() != hipSuccess) { printf("lds %d: launch failed\n", lds); continue; }
        std::vector<unsigned long long> t(2 * grid);
        hipMemcpy(t.data(), d_t, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<std::pair<unsigned long long, int>> ev;
        for (int b = 0; b < grid; ++b) { ev.push_back({t[2 * b], 1}); ev.push_back({t[2 * b + 1], -1}); }
        std::sort(ev.begin(), ev.end());
        int cur = 0, best = 0;
        for (auto& e : ev) { cur += e.second; best = std::max(best, cur); }
        printf("vgpr > %d, lds %6d B per workgroup: at most %d workgroups resident (%.2f per CU; 160 KiB rule: %d)\n", vg, lds, best, best / 256.0, 163840 / std::max(lds, 1));
    }
    return 0;
}
