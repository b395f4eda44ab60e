// latency.hip -- single-wave latency probes on MI355X: dependent loads from LDS / global (various footprints),
// with and without an interleaved global store.  Build: hipcc -O3 --offload-arch=gfx950 -o latency latency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void chase_global(const unsigned* __restrict__ p, int n, unsigned long long* out, unsigned* sink, int withStore)
{
    if (threadIdx.x != 0) return;
    unsigned i = 0;
    for (int k = 0; k < 64; k++) i = p[i];           // warm
    unsigned long long t0 = clock64();
    for (int k = 0; k < n; k++) { i = p[i]; if (withStore) sink[k & 1023] = i; }
    unsigned long long t1 = clock64();
    out[0] = t1 - t0; sink[2048] = i;
}
__global__ void chase_lds(int n, unsigned long long* out, unsigned* sink)
{
    __shared__ unsigned l[8192];
    for (int k = threadIdx.x; k < 8192; k += 64) l[k] = (k * 1237 + 17) & 8191;
    __syncthreads();
    if (threadIdx.x != 0) return;
    unsigned i = 0;
    unsigned long long t0 = clock64();
    for (int k = 0; k < n; k++) i = l[i];
    unsigned long long t1 = clock64();
    out[0] = t1 - t0; sink[0] = i;
}
__global__ void wave_round(const unsigned* __restrict__ p, int n, unsigned long long* out, unsigned* sink)
{   // all 64 lanes load (random), ballot, next address depends on the ballot: one "cooperative step"
    unsigned i = threadIdx.x * 977u;
    unsigned long long t0 = clock64();
    for (int k = 0; k < n; k++) {
        unsigned v = p[(i + k) & 0xFFFFF];
        unsigned long long b = __ballot(v & 1);
        i = (unsigned)(b >> 7) + threadIdx.x * 977u + v;
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = i;
}
int main()
{
    unsigned long long* d_out; unsigned* d_sink; hipMalloc(&d_out, 64); hipMalloc(&d_sink, 1 << 16);
    int n = 20000;
    for (size_t words : { (size_t)2048, (size_t)8192, (size_t)32768, (size_t)262144, (size_t)4 << 20, (size_t)64 << 20 }) {
        std::vector<unsigned> h(words);
        // random cyclic permutation (Sattolo)
        for (size_t k = 0; k < words; k++) h[k] = (unsigned)k;
        srand(1); for (size_t k = words - 1; k > 0; k--) { size_t j = rand() % k; unsigned t = h[k]; h[k] = h[j]; h[j] = t; }
        unsigned* d; hipMalloc(&d, words * 4); hipMemcpy(d, h.data(), words * 4, hipMemcpyHostToDevice);
        for (int ws = 0; ws < 2; ws++) {
            unsigned long long c = 0;
            hipLaunchKernelGGL(chase_global, dim3(1), dim3(64), 0, 0, d, n, d_out, d_sink, ws);
            hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
            printf("global chase footprint %8zu KiB store=%d : %.1f cycles/load\n", words * 4 / 1024, ws, (double)c / n);
        }
        if (words == ((size_t)4 << 20)) {
            unsigned long long c = 0;
            hipLaunchKernelGGL(wave_round, dim3(1), dim3(64), 0, 0, d, n, d_out, d_sink);
            hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
            printf("wave round (64 random loads + ballot), 4 MiB window : %.1f cycles/round\n", (double)c / n);
        }
        hipFree(d);
    }
    unsigned long long c = 0;
    hipLaunchKernelGGL(chase_lds, dim3(1), dim3(64), 0, 0, n, d_out, d_sink);
    hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
    printf("LDS chase : %.1f cycles/load\n", (double)c / n);
    return 0;
}
