// issue_rate.hip -- instruction issue throughput per CU on MI355X for scalar-ALU, vector-ALU and 2:1 mixed streams at
// 1/2/4/8 waves per SIMD (one wave per workgroup, as the codec kernels run).  Diagnostics for DESIGN.md section 6.
// Build: hipcc -O3 --offload-arch=gfx950 -o issue_rate issue_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE> __global__ void __launch_bounds__(64) spin(int iters, unsigned* sink)
{
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {        // 64 scalar adds, 4 independent chains
            REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
        } else if (MODE == 1) { // 64 vector adds
            REP16(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
        } else {                // 2 scalar : 1 vector, 96 instructions
            REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %4, %4, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n v_add_u32 %5, %5, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1) : : "scc");)
        }
    }
    if (s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3 == 0xFFFFFFFFu) sink[0] = 1;
}
int main()
{
    unsigned* d_sink; hipMalloc(&d_sink, 64);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int const cus = prop.multiProcessorCount; double const ghz = prop.clockRate / 1e6;
    printf("CUs %d, clock %.2f GHz\n", cus, ghz);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = { "scalar", "vector", "2s:1v" };
    int const per[3] = { 64, 64, 96 };
    for (int mode = 0; mode < 3; mode++)
        for (int wpc : { 4, 8, 16, 32 }) {
            int const iters = 20000;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(spin<0>, dim3(cus * wpc), dim3(64), 0, 0, iters, d_sink);
                else if (mode == 1) hipLaunchKernelGGL(spin<1>, dim3(cus * wpc), dim3(64), 0, 0, iters, d_sink);
                else hipLaunchKernelGGL(spin<2>, dim3(cus * wpc), dim3(64), 0, 0, iters, d_sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double const instr = (double)iters * per[mode] * wpc;       // per CU
            double const cyc = ms * 1e-3 * ghz * 1e9;
            printf("%-7s %2d waves/CU: %.3f instr/cycle/CU, %.2f cycles/instr/wave\n", names[mode], wpc, instr / cyc, cyc / ((double)iters * per[mode])); fflush(stdout);
        }
    return 0;
}
