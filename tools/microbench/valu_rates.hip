// valu_rates.hip -- GPU-box microbenchmark: what one wave64 VALU instruction costs a gfx950 SIMD, for the opcodes
// the raster kernels are made of.  For each opcode: 16 instructions per loop trip over 8 independent register chains
// ("x8") or one dependent chain ("x1"), at 1..8 resident waves per SIMD.  Reported in cycles of the shader clock
// measured inside the kernel (s_memtime against the 100 MHz s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_rates.hip -o tools/microbench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ALL8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define X8(OP) ALL8(OP) ALL8(OP)
#define X1(OP) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)

#define KERNEL(name, asmtext)                                                                                        \
    __global__ void name(float* out, unsigned long long* clk, int iters)                                             \
    {                                                                                                                \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = 1.0001f, b1 = 0.9999f;                                                                            \
        const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();                             \
        for (int i = 0; i < iters; i++) {                                                                            \
            asm volatile(asmtext                                                                                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)            \
                         : "v"(b0), "v"(b1)                                                                          \
                         : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");                                         \
        }                                                                                                            \
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();                             \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                          \
    }

#define OP_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define OP_ADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define OP_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define OP_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define OP_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define OP_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define OP_LSHL_ADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define OP_MAX(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define OP_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define OP_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define OP_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define OP_CVT(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define OP_DPP_QUAD(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_DPP_ROWSHR(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_shr:4 row_mask:0xf bank_mask:0xf\n"
#define OP_DPP_MIRROR(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_half_mirror row_mask:0xf bank_mask:0xf\n"
#define OP_DPP_BCAST(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define OP_MOV_DPP(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_CMP_VCC(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define OP_CMP_SGPR(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\n"
#define OP_CND_VCC(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define OP_CND_SGPR(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[22:23]\n"
#define OP_READLANE(n) "v_readlane_b32 s24, %" #n ", 3\n"
#define OP_READFIRST(n) "v_readfirstlane_b32 s24, %" #n "\n"
#define OP_MUL_SGPR(n) "v_mul_f32 %" #n ", s25, %" #n "\n"
#define OP_MUL_LIT(n) "v_mul_f32 %" #n ", 0x3f800347, %" #n "\n"
#define OP_MAD64(n) "v_mad_u64_u32 v[40:41], s[20:21], %" #n ", %8, v[40:41]\n"
#define OP_MUL_THEN_S(n) "v_mul_f32 %" #n ", %" #n ", %8\n s_and_b32 s24, s24, s25\n"
#define OP_MUL_THEN_NOP(n) "v_mul_f32 %" #n ", %" #n ", %8\n s_nop 0\n"
#define OP_MUL_THEN_2S(n) "v_mul_f32 %" #n ", %" #n ", %8\n s_and_b32 s24, s24, s25\n s_and_b32 s20, s20, s25\n"
#define OP_S_AND(n) "s_and_b32 s24, s24, s25\n"
#define OP_CMP_CND_VCC(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define OP_CMP_CND_SGPR(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, s[20:21]\n"
#define OP_CMP_CND_SGPR_ROT(n) "v_cmp_lt_f32 s[22:23], %" #n ", %8\n v_cmp_lt_f32 s[20:21], %" #n ", %9\n v_cndmask_b32 %" #n ", %" #n ", %9, s[22:23]\n v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
#define OP_S_AND64(n) "s_and_b64 s[20:21], s[20:21], s[22:23]\n"
#define OP_S_FF1(n) "s_ff1_i32_b64 s24, s[20:21]\n"
#define OP_S_BRANCH(n) "s_cmp_eq_u32 s25, 12345\n s_cbranch_scc1 1\n s_nop 0\n"

KERNEL(k_mul_x8, X8(OP_MUL))
KERNEL(k_mul_x1, X1(OP_MUL))
KERNEL(k_add_x8, X8(OP_ADD))
KERNEL(k_fma_x8, X8(OP_FMA))
KERNEL(k_fma_x1, X1(OP_FMA))
KERNEL(k_mov_x8, X8(OP_MOV))
KERNEL(k_addu_x8, X8(OP_ADDU))
KERNEL(k_and_x8, X8(OP_AND))
KERNEL(k_lshladd_x8, X8(OP_LSHL_ADD))
KERNEL(k_max_x8, X8(OP_MAX))
KERNEL(k_rcp_x8, X8(OP_RCP))
KERNEL(k_exp_x8, X8(OP_EXP))
KERNEL(k_sqrt_x8, X8(OP_SQRT))
KERNEL(k_cvt_x8, X8(OP_CVT))
KERNEL(k_dppquad_x8, X8(OP_DPP_QUAD))
KERNEL(k_dppquad_x1, X1(OP_DPP_QUAD))
KERNEL(k_dpprowshr_x8, X8(OP_DPP_ROWSHR))
KERNEL(k_dppmirror_x8, X8(OP_DPP_MIRROR))
KERNEL(k_dppbcast_x8, X8(OP_DPP_BCAST))
KERNEL(k_movdpp_x8, X8(OP_MOV_DPP))
KERNEL(k_cmpvcc_x8, X8(OP_CMP_VCC))
KERNEL(k_cmpsgpr_x8, X8(OP_CMP_SGPR))
KERNEL(k_cndvcc_x8, X8(OP_CND_VCC))
KERNEL(k_cndsgpr_x8, X8(OP_CND_SGPR))
KERNEL(k_readlane_x8, X8(OP_READLANE))
KERNEL(k_readfirst_x8, X8(OP_READFIRST))
KERNEL(k_mulsgpr_x8, X8(OP_MUL_SGPR))
KERNEL(k_mullit_x8, X8(OP_MUL_LIT))
KERNEL(k_mul_s_x8, ALL8(OP_MUL_THEN_S))
KERNEL(k_mul_nop_x8, ALL8(OP_MUL_THEN_NOP))
KERNEL(k_mul_2s_x8, ALL8(OP_MUL_THEN_2S))
// two / four v_mul per s_and
KERNEL(k_mul2_s, OP_MUL(0) OP_MUL(1) OP_S_AND(0) OP_MUL(2) OP_MUL(3) OP_S_AND(0) OP_MUL(4) OP_MUL(5) OP_S_AND(0) OP_MUL(6) OP_MUL(7) OP_S_AND(0))
KERNEL(k_mul4_s, OP_MUL(0) OP_MUL(1) OP_MUL(2) OP_MUL(3) OP_S_AND(0) OP_MUL(4) OP_MUL(5) OP_MUL(6) OP_MUL(7) OP_S_AND(0))
KERNEL(k_s_and, X8(OP_S_AND))
KERNEL(k_cmpcnd_vcc, ALL8(OP_CMP_CND_VCC))
KERNEL(k_cmpcnd_sgpr, ALL8(OP_CMP_CND_SGPR))
KERNEL(k_cmpcnd_sgpr_rot, ALL8(OP_CMP_CND_SGPR_ROT))
KERNEL(k_s_and64, X8(OP_S_AND64))
KERNEL(k_s_ff1, X8(OP_S_FF1))
KERNEL(k_s_branch, ALL8(OP_S_BRANCH))
KERNEL(k_swap32_x8, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                    "v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n v_permlane32_swap_b32 %4, %6\n v_permlane32_swap_b32 %5, %7\n"
                    "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                    "v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n v_permlane32_swap_b32 %4, %6\n v_permlane32_swap_b32 %5, %7\n")
KERNEL(k_swap16_x8, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                    "v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7\n"
                    "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                    "v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7\n")

// Does the SIMD skip the half (or quarter) of a wave64 whose EXEC bits are all zero?  The same v_fma stream with
// only some lanes enabled (the loop's trip count is uniform, so EXEC can simply be narrowed around it).
#define KERNEL_EXEC(name, lo, hi)                                                                                    \
    __global__ void name(float* out, unsigned long long* clk, int iters)                                             \
    {                                                                                                                \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = 1.0001f, b1 = 0.9999f;                                                                            \
        const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();                             \
        asm volatile("s_mov_b32 exec_lo, " lo "\n s_mov_b32 exec_hi, " hi "\n" ::: "memory");                        \
        for (int i = 0; i < iters; i++) {                                                                            \
            asm volatile(X8(OP_FMA)                                                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)            \
                         : "v"(b0), "v"(b1));                                                                        \
        }                                                                                                            \
        asm volatile("s_mov_b64 exec, -1\n" ::: "memory");                                                           \
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();                             \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                          \
    }
KERNEL_EXEC(k_fma_exec_lo32, "0xffffffff", "0")
KERNEL_EXEC(k_fma_exec_hi32, "0", "0xffffffff")
KERNEL_EXEC(k_fma_exec_lo16, "0xffff", "0")
KERNEL_EXEC(k_fma_exec_even, "0x55555555", "0x55555555")
KERNEL_EXEC(k_fma_exec_rows02, "0x0000ffff", "0x0000ffff")
KERNEL_EXEC(k_fma_exec_one, "1", "0")

// packed fp32: 8 independent register pairs
typedef float pk2 __attribute__((ext_vector_type(2)));
#define PK_MUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define PK_ADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define PK_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define PK_MOV(n) "v_pk_mov_b32 %" #n ", %8, %9\n"
#define KERNEL_PK(name, asmtext)                                                                                     \
    __global__ void name(float* out, unsigned long long* clk, int iters)                                             \
    {                                                                                                                \
        pk2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        pk2 b0 = {1.0001f, 0.9999f}, b1 = {0.9999f, 1.0001f};                                                        \
        const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();                             \
        for (int i = 0; i < iters; i++) {                                                                            \
            asm volatile(asmtext                                                                                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)            \
                         : "v"(b0), "v"(b1));                                                                        \
        }                                                                                                            \
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();                             \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }                             \
        pk2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                                      \
    }
KERNEL_PK(k_pkmul_x8, X8(PK_MUL))
KERNEL_PK(k_pkadd_x8, X8(PK_ADD))
KERNEL_PK(k_pkfma_x8, X8(PK_FMA))
KERNEL_PK(k_pkmov_x8, X8(PK_MOV))

// LDS: ds_read_b128 of a wave-uniform address (the raster kernels' per-entry record reads) and ds_bpermute
__global__ void k_lds(float* out, unsigned long long* clk, int iters, int mode)
{
    __shared__ float4 s[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float acc = 0;
    int idx = threadIdx.x >> 6;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            float4 v;
            if (mode == 0) v = s[(idx + k * 5) & 1023];                    // uniform address: broadcast read
            else v = s[(threadIdx.x + k * 64 + idx) & 1023];               // per-lane 16-byte reads
            acc += v.x + v.w;
            idx = (idx + (int)v.y) & 1023;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

typedef void (*kern_t)(float*, unsigned long long*, int);

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs, %.2f GHz nominal\n", p.gcnArchName, cus, p.clockRate / 1e6);
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    hipMalloc(&clk, 16);
    const int iters = 20000;
    struct K { const char* name; kern_t fn; };
    const std::vector<K> kernels = {
        {"v_mul_f32 x8", k_mul_x8}, {"v_mul_f32 x1 (dependent)", k_mul_x1}, {"v_add_f32 x8", k_add_x8}, {"v_fma_f32 x8", k_fma_x8},
        {"v_fma_f32 x1 (dependent)", k_fma_x1}, {"v_mov_b32 x8", k_mov_x8}, {"v_add_u32 x8", k_addu_x8}, {"v_and_b32 x8", k_and_x8},
        {"v_lshl_add_u32 x8", k_lshladd_x8}, {"v_max_f32 x8", k_max_x8}, {"v_mul_f32 sgpr src", k_mulsgpr_x8}, {"v_mul_f32 literal src", k_mullit_x8},
        {"v_rcp_f32 x8", k_rcp_x8}, {"v_exp_f32 x8", k_exp_x8}, {"v_sqrt_f32 x8", k_sqrt_x8}, {"v_cvt_i32_f32 x8", k_cvt_x8},
        {"v_add_f32_dpp quad_perm x8", k_dppquad_x8}, {"v_add_f32_dpp quad_perm x1", k_dppquad_x1}, {"v_add_f32_dpp row_shr x8", k_dpprowshr_x8},
        {"v_add_f32_dpp row_half_mirror", k_dppmirror_x8}, {"v_add_f32_dpp row_bcast15", k_dppbcast_x8}, {"v_mov_b32_dpp quad_perm", k_movdpp_x8},
        {"v_cmp_lt_f32 -> vcc", k_cmpvcc_x8}, {"v_cmp_lt_f32 -> sgpr pair", k_cmpsgpr_x8}, {"v_cndmask_b32 vcc", k_cndvcc_x8},
        {"v_cndmask_b32 sgpr pair", k_cndsgpr_x8}, {"v_readlane_b32", k_readlane_x8}, {"v_readfirstlane_b32", k_readfirst_x8},
        {"v_permlane32_swap", k_swap32_x8}, {"v_permlane16_swap", k_swap16_x8},
        {"v_fma_f32 x8, EXEC = lanes 0-31", k_fma_exec_lo32}, {"v_fma_f32 x8, EXEC = lanes 32-63", k_fma_exec_hi32},
        {"v_fma_f32 x8, EXEC = lanes 0-15", k_fma_exec_lo16}, {"v_fma_f32 x8, EXEC = even lanes", k_fma_exec_even},
        {"v_fma_f32 x8, EXEC = rows 0 and 2", k_fma_exec_rows02}, {"v_fma_f32 x8, EXEC = lane 0", k_fma_exec_one},
        {"v_pk_mul_f32 x8", k_pkmul_x8}, {"v_pk_add_f32 x8", k_pkadd_x8}, {"v_pk_fma_f32 x8", k_pkfma_x8}, {"v_pk_mov_b32 x8", k_pkmov_x8},
    };
    const std::vector<K> mixed = {{"(v_mul_f32 + s_and_b32) pairs", k_mul_s_x8}, {"(v_mul_f32 + s_nop) pairs", k_mul_nop_x8},
                                  {"(v_mul_f32 + 2 s_and_b32), per v_mul", k_mul_2s_x8}, {"(2 v_mul_f32 + s_and_b32), per v_mul", k_mul2_s},
                                  {"(4 v_mul_f32 + s_and_b32), per v_mul", k_mul4_s}, {"(s_cmp + s_cbranch not taken), per pair", k_s_branch},
                                  {"(v_cmp -> vcc, s_nop 1, v_cndmask vcc), per pair", k_cmpcnd_vcc},
                                  {"(v_cmp -> sgpr pair, s_nop 1, v_cndmask sgpr), per pair", k_cmpcnd_sgpr}};
    const std::vector<K> mixed2 = {{"2 x (v_cmp -> sgpr, v_cndmask sgpr) interleaved, per pair", k_cmpcnd_sgpr_rot}};
    const std::vector<K> scalar = {{"s_and_b32 alone", k_s_and}, {"s_and_b64 alone", k_s_and64}, {"s_ff1_i32_b64 alone", k_s_ff1}};
    printf("left: issue interval seen by the oldest wave of a SIMD (it wins arbitration); right: what the SIMD spends per\n"
           "wave-instruction when W waves per SIMD run the same stream (whole-kernel time)\n");
    printf("%-32s %8s | %s\n", "opcode", "MHz", "oldest wave: cycles/instr at W=1,2,4,8  ||  SIMD cycles/instr (kernel time x clock / instrs / W) at W=1,2,4,8");
    hipEvent_t ev0, ev1;
    hipEventCreate(&ev0);
    hipEventCreate(&ev1);
    auto run = [&](const K& k, int instr_per_trip) {
        double wave_cyc[4], simd_cyc[4], mhz = 0;
        int wi = 0;
        for (int wps = 1; wps <= 8; wps *= 2, wi++) {
            const int blocks = cus * wps;
            unsigned long long h[2] = {0, 0};
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
            hipEventRecord(ev0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
            hipEventRecord(ev1);
            hipEventSynchronize(ev1);
            float ms = 0;
            hipEventElapsedTime(&ms, ev0, ev1);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            wave_cyc[wi] = (double)h[0] / ((double)iters * instr_per_trip);
            mhz = (double)h[0] / (double)h[1] * 100.0;
            simd_cyc[wi] = ms * 1e-3 * mhz * 1e6 / ((double)iters * instr_per_trip * wps);
        }
        printf("%-32s %8.0f | %6.2f %6.2f %6.2f %6.2f  || %6.2f %6.2f %6.2f %6.2f\n", k.name, mhz, wave_cyc[0], wave_cyc[1], wave_cyc[2], wave_cyc[3],
               simd_cyc[0], simd_cyc[1], simd_cyc[2], simd_cyc[3]);
    };
    for (const K& k : kernels) run(k, 16);
    for (const K& k : mixed) run(k, 8); // per v_mul (each followed by one SALU instruction)
    for (const K& k : mixed2) run(k, 16);
    for (const K& k : scalar) run(k, 16);
    for (int mode = 0; mode < 2; mode++) {
        double wave_cyc[4], mhz = 0;
        int wi = 0;
        for (int wps = 1; wps <= 8; wps *= 2, wi++) {
            unsigned long long h[2];
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_lds, dim3(cus * wps), dim3(256), 0, 0, out, clk, iters / 8, mode);
            hipDeviceSynchronize();
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            wave_cyc[wi] = (double)h[0] / ((double)(iters / 8) * 16);
            mhz = (double)h[0] / (double)h[1] * 100.0;
        }
        printf("%-32s %8.0f | %6.1f %6.1f %6.1f %6.1f  || %6.1f %6.1f %6.1f %6.1f   (dependent ds_read_b128 + 3 VALU)\n",
               mode == 0 ? "ds_read_b128 uniform addr chain" : "ds_read_b128 per-lane chain", mhz, wave_cyc[0], wave_cyc[1], wave_cyc[2], wave_cyc[3],
               wave_cyc[0], wave_cyc[1] / 2, wave_cyc[2] / 4, wave_cyc[3] / 8);
    }
    hipFree(out);
    hipFree(clk);
    return 0;
}
