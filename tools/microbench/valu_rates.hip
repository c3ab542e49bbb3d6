// valu_rates.hip -- GPU-box microbenchmark: sustained issue cost of the VALU instructions the raster kernels are
// made of, per wave64 instruction per SIMD, at 1 and 8 resident waves per SIMD.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(name, asmtext)                                                                         \
    __global__ void name(float* out, int iters)                                                     \
    {                                                                                               \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = 1.0001f, b1 = 0.9999f;                                                           \
        for (int i = 0; i < iters; i++) {                                                           \
            asm volatile(asmtext : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1)); \
        }                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;         \
    }

// 16 instructions per asm block, 8 independent chains (dependent distance 8)
BODY(k_mul, REP8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n") )
BODY(k_mul8, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %9\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %9\n"
             "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %9\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %9\n")
BODY(k_fma8, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %9, %8\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %9, %8\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %9, %8\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %9, %8\n"
             "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %9, %8\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %9, %8\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %9, %8\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %9, %8\n")
BODY(k_dpp8, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
             "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n"
             "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
             "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n")
BODY(k_swap8, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7\n"
              "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7\n")
BODY(k_rcp8, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
             "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
BODY(k_cnd8, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
             "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n")

// packed: 8 register pairs
__global__ void k_pk8(float* out, int iters, int mode)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b0 = {1.0001f, 0.9999f}, b1 = {0.9999f, 1.0001f};
    for (int i = 0; i < iters; i++) {
        if (mode == 0)
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %9\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %9\n"
                         "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %9\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        else if (mode == 1)
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                         "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        else
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %9, %8\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %9, %8\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %9, %8\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %9, %8\n"
                         "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %9, %8\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %9, %8\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %9, %8\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %9, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <typename F>
static double time_ms(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, %.2f GHz nominal\n", p.gcnArchName, cus, ghz);
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) { // waves per SIMD
        const int blocks = cus * wps; // 256 threads = 4 waves = one per SIMD
        auto report = [&](const char* name, double ms, int flops_per_lane_instr) {
            const double insts = (double)iters * 16;            // per wave
            const double cyc = ms * 1e-3 * ghz * 1e9;           // nominal cycles
            printf("  %-22s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (%.2f per wave), %6.1f Tlane-op/s\n", name, wps,
                   cyc / (insts * wps), cyc / insts, insts * wps * 4 * cus * 64.0 * flops_per_lane_instr / (ms * 1e-3) / 1e12);
        };
        report("v_mul_f32", time_ms([&] { hipLaunchKernelGGL(k_mul8, dim3(blocks), dim3(256), 0, 0, out, iters); }), 1);
        report("v_fma_f32", time_ms([&] { hipLaunchKernelGGL(k_fma8, dim3(blocks), dim3(256), 0, 0, out, iters); }), 1);
        report("v_pk_mul_f32", time_ms([&] { hipLaunchKernelGGL(k_pk8, dim3(blocks), dim3(256), 0, 0, out, iters, 0); }), 2);
        report("v_pk_add_f32", time_ms([&] { hipLaunchKernelGGL(k_pk8, dim3(blocks), dim3(256), 0, 0, out, iters, 1); }), 2);
        report("v_pk_fma_f32", time_ms([&] { hipLaunchKernelGGL(k_pk8, dim3(blocks), dim3(256), 0, 0, out, iters, 2); }), 2);
        report("v_add_f32_dpp", time_ms([&] { hipLaunchKernelGGL(k_dpp8, dim3(blocks), dim3(256), 0, 0, out, iters); }), 1);
        report("v_permlane*_swap", time_ms([&] { hipLaunchKernelGGL(k_swap8, dim3(blocks), dim3(256), 0, 0, out, iters); }), 1);
        report("v_rcp_f32", time_ms([&] { hipLaunchKernelGGL(k_rcp8, dim3(blocks), dim3(256), 0, 0, out, iters); }), 1);
        report("v_cmp+v_cndmask", time_ms([&] { hipLaunchKernelGGL(k_cnd8, dim3(blocks), dim3(256), 0, 0, out, iters); }), 1);
    }
    hipFree(out);
    return 0;
}
