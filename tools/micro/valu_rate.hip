// Issue rate of the VALU instructions the front-end kernels are made of (gfx950): cycles per wave64 instruction per
// SIMD with 1, 2, 4, 8 waves per SIMD, 8 independent chains per wave.  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DEF_KERNEL(NAME, ASM)                                                                                    \
    __global__ void NAME(unsigned *out, unsigned b, unsigned c, int iters, unsigned long long *cyc) {            \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7;                                                                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int i = 0; i < iters; i++) {                                                                        \
            _Pragma("unroll") for (int u = 0; u < 16; u++) {                                                     \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c));                                                   \
                asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c));                                                   \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c));                                                   \
                asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c));                                                   \
                asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c));                                                   \
                asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c));                                                   \
                asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c));                                                   \
                asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c));                                                   \
            }                                                                                                    \
        }                                                                                                        \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                      \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                         \
    }

DEF_KERNEL(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEF_KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEF_KERNEL(k_mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %0")
DEF_KERNEL(k_and_b32, "v_and_b32 %0, %1, %0")
DEF_KERNEL(k_and_or_b32, "v_and_or_b32 %0, %1, %2, %0")
DEF_KERNEL(k_bitop3, "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x80")
DEF_KERNEL(k_or3, "v_or3_b32 %0, %1, %2, %0")
DEF_KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %1, %0")
DEF_KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %1, %0")
DEF_KERNEL(k_add_u32, "v_add_u32 %0, %1, %0")
DEF_KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
DEF_KERNEL(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
DEF_KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %1, %0, 1")
DEF_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %1, 3, %0")
DEF_KERNEL(k_med3_u32, "v_med3_u32 %0, %1, %2, %0")
DEF_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
DEF_KERNEL(k_xor, "v_xor_b32 %0, %1, %0")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %1, %0, %2")
DEF_KERNEL(k_bfe_u32, "v_bfe_u32 %0, %0, 8, 8")
DEF_KERNEL(k_cvt_f32_ubyte, "v_cvt_f32_ubyte1 %0, %0")
DEF_KERNEL(k_sad_u8, "v_sad_u8 %0, %1, %2, %0")
DEF_KERNEL(k_max3_u32, "v_max3_u32 %0, %1, %2, %0")
DEF_KERNEL(k_min_u16, "v_min_u16 %0, %1, %0")
DEF_KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %1, %0")
DEF_KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %1, %2, %0")
DEF_KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %0, %1, %0")

typedef void (*kern_t)(unsigned *, unsigned, unsigned, int, unsigned long long *);
struct K { const char *name; kern_t f; };

int main() {
    K ks[] = {{"v_fma_f32", k_fma_f32}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mad_i32_i24", k_mad_i32_i24}, {"v_and_b32", k_and_b32},
              {"v_and_or_b32", k_and_or_b32}, {"v_bitop3_b32", k_bitop3}, {"v_or3_b32", k_or3}, {"v_pk_min_i16", k_pk_min_i16},
              {"v_pk_max_i16", k_pk_max_i16}, {"v_add_u32", k_add_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_dot4_u32_u8", k_dot4_u32_u8},
              {"v_alignbyte_b32", k_alignbyte}, {"v_lshl_or_b32", k_lshl_or}, {"v_med3_u32", k_med3_u32}, {"v_bcnt_u32_b32", k_bcnt},
              {"v_xor_b32", k_xor}, {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe_u32}, {"v_cvt_f32_ubyte1", k_cvt_f32_ubyte},
              {"v_sad_u8", k_sad_u8}, {"v_max3_u32", k_max3_u32}, {"v_min_u16", k_min_u16}, {"v_pk_add_u16", k_pk_add_u16},
              {"v_pk_mad_u16", k_pk_mad_u16}, {"v_pk_sub_i16", k_pk_sub_i16}};
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, (size_t)cus * 8 * 4 * 64 * 4);
    hipMalloc(&cyc, (size_t)cus * 8 * 8);
    const int iters = 256; /* x 128 instructions */
    printf("instruction,waves_per_simd,cycles_per_wave64_inst_per_simd(s_memtime 100MHz-based wall),wall_ms\n");
    for (auto &k : ks)
        for (int wps : {1, 2, 4, 8}) {
            /* one workgroup per CU with 4 * wps waves: wps waves on each of the 4 SIMDs */
            const int threads = 64 * 4 * wps;
            dim3 grid(cus), block(threads);
            if (threads > 1024) { grid = dim3(cus * 2); block = dim3(1024); }
            hipLaunchKernelGGL(k.f, grid, block, 0, 0, out, 3u, 5u, 4, cyc);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.f, grid, block, 0, 0, out, 3u, 5u, iters, cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            /* instructions per SIMD = wps * iters * 128; cycles at the clock the chip held = wall * f; report per 2.4 GHz too */
            const double inst_per_simd = (double)wps * iters * 128.0;
            printf("%s,%d,%.3f,%.4f\n", k.name, wps, ms * 1e-3 * 2.4e9 / inst_per_simd, ms);
        }
    return 0;
}
