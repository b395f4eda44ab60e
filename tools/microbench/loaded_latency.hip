// loaded_latency.hip -- dependent-load latency of one lane per wave while N waves do the same thing, each in a private
// region (the situation of the codec's per-frame match tables): 16 waves per CU, region 32 KiB (level-1 table) or 384 KiB
// (level-3 tables), regions 2.6 MiB apart as in the library's scratch.  Optionally one store per step to another cell.
// Build: hipcc -O3 --offload-arch=gfx950 -o loaded_latency loaded_latency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void __launch_bounds__(64) chase(unsigned* base, size_t strideWords, unsigned cells, int n, int withStore, unsigned long long* out)
{
    if (threadIdx.x != 0) return;
    unsigned* p = base + (size_t)blockIdx.x * strideWords;
    unsigned i = (blockIdx.x * 977u) % cells;
    for (int k = 0; k < 32; k++) i = p[i] % cells;
    unsigned long long t0 = clock64();
    for (int k = 0; k < n; k++) {
        unsigned const v = p[i];
        if (withStore) p[(i ^ 0x155u) % cells + cells] = v;   // second half of the region: never read
        i = v % cells;
    }
    unsigned long long t1 = clock64();
    out[blockIdx.x] = t1 - t0;
    if (i == 0xFFFFFFFFu) out[0] = 0;
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int const cus = prop.multiProcessorCount;
    size_t const strideWords = ((size_t)2600 << 10) / 4;
    int const maxWaves = cus * 16;
    unsigned* d; hipMalloc(&d, strideWords * 4 * maxWaves);
    unsigned long long* d_out; hipMalloc(&d_out, 8 * maxWaves);
    std::vector<unsigned long long> h(maxWaves);
    for (unsigned regionKiB : { 32u, 384u }) {
        unsigned const cells = regionKiB * 1024 / 4 / 2;          // first half chased, second half stored to
        std::vector<unsigned> perm(cells);
        for (unsigned k = 0; k < cells; k++) perm[k] = k;
        srand(1); for (unsigned k = cells - 1; k > 0; k--) { unsigned j = rand() % k; unsigned t = perm[k]; perm[k] = perm[j]; perm[j] = t; }
        for (int w = 0; w < maxWaves; w++) hipMemcpy(d + (size_t)w * strideWords, perm.data(), cells * 4, hipMemcpyHostToDevice);
        for (int wpc : { 1, 4, 16 })
            for (int ws = 0; ws < 2; ws++) {
                int const waves = cus * wpc, n = 2000;
                hipLaunchKernelGGL(chase, dim3(waves), dim3(64), 0, 0, d, strideWords, cells, n, ws, d_out);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), d_out, 8 * waves, hipMemcpyDeviceToHost);
                double s = 0; for (int w = 0; w < waves; w++) s += (double)h[w];
                printf("region %3u KiB, %2d waves/CU (%5d waves), store=%d : %.0f cycles per dependent load\n", regionKiB, wpc, waves, ws, s / waves / n);
                fflush(stdout);
            }
    }
    return 0;
}
