// Microbenchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU instructions the
// orbit loop is made of, on gfx950.  One workgroup of 64*W threads per CU-SIMD slot is not
// controllable from HIP, so we launch 256 CUs x 4 SIMDs x W waves as blocks of 256*W threads
// (W waves land on each SIMD) and time with s_memtime inside the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// each BODY issues 32 instructions (4 x 8 independent chains)
#define DEF_KERNEL(NAME, ...) DEF_KERNEL_(NAME, __VA_ARGS__)
#define DEF_KERNEL_(NAME, I0, I1, I2, I3, I4, I5, I6, I7) \
    __global__ void NAME(unsigned long long *out, int iters, double seed) {                     \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4,           \
               a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, k = 1.0000001;                        \
        unsigned b0 = iters, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7, kk = 3;                        \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int i = 0; i < iters; i++) {                                                       \
            asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"          \
                         I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"          \
                         I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"          \
                         I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"          \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),  \
                           "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) \
                         : "v"(k), "v"(kk)                                                               \
                         : "vcc");                                                              \
        }                                                                                       \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678 || (b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7) == 0x12345) out[0] = 0;                      \
    }

#define OP3(op, n) op " %" #n ", %" #n ", %16"
#define ALL8(op) OP3(op, 0), OP3(op, 1), OP3(op, 2), OP3(op, 3), OP3(op, 4), OP3(op, 5), OP3(op, 6), OP3(op, 7)

DEF_KERNEL(k_mul_f64, ALL8("v_mul_f64"))
DEF_KERNEL(k_add_f64, ALL8("v_add_f64"))
#define FMA(n) "v_fma_f64 %" #n ", %" #n ", %16, %" #n
DEF_KERNEL(k_fma_f64, FMA(0), FMA(1), FMA(2), FMA(3), FMA(4), FMA(5), FMA(6), FMA(7))
#define CMP(n) "v_cmp_gt_f64 vcc, %" #n ", %16"
DEF_KERNEL(k_cmp_f64, CMP(0), CMP(1), CMP(2), CMP(3), CMP(4), CMP(5), CMP(6), CMP(7))
#define U0 "%8"
#define U1 "%9"
#define U2 "%10"
#define U3 "%11"
#define U4 "%12"
#define U5 "%13"
#define U6 "%14"
#define U7 "%15"
#define CMPU(n) "v_cmp_gt_u32 vcc, " U##n ", %17"
DEF_KERNEL(k_cmp_u32, CMPU(0), CMPU(1), CMPU(2), CMPU(3), CMPU(4), CMPU(5), CMPU(6), CMPU(7))
#define MAXU(n) "v_max_u32 " U##n ", " U##n ", %17"
DEF_KERNEL(k_max_u32, MAXU(0), MAXU(1), MAXU(2), MAXU(3), MAXU(4), MAXU(5), MAXU(6), MAXU(7))
#define MOV(n) "v_mov_b32 " U##n ", %17"
DEF_KERNEL(k_mov_b32, MOV(0), MOV(1), MOV(2), MOV(3), MOV(4), MOV(5), MOV(6), MOV(7))
#define MOV64(n) "v_mov_b64 %" #n ", %16"
DEF_KERNEL(k_mov_b64, MOV64(0), MOV64(1), MOV64(2), MOV64(3), MOV64(4), MOV64(5), MOV64(6), MOV64(7))
#define MULF(n) "v_mul_f32 " U##n ", " U##n ", %17"
DEF_KERNEL(k_mul_f32, MULF(0), MULF(1), MULF(2), MULF(3), MULF(4), MULF(5), MULF(6), MULF(7))
#define PKMULF(n) "v_pk_mul_f32 %" #n ", %" #n ", %16"
DEF_KERNEL(k_pk_mul_f32, PKMULF(0), PKMULF(1), PKMULF(2), PKMULF(3), PKMULF(4), PKMULF(5), PKMULF(6), PKMULF(7))
#define PKADDF(n) "v_pk_add_f32 %" #n ", %" #n ", %16"
DEF_KERNEL(k_pk_add_f32, PKADDF(0), PKADDF(1), PKADDF(2), PKADDF(3), PKADDF(4), PKADDF(5), PKADDF(6), PKADDF(7))
#define MAX64(n) "v_max_f64 %" #n ", %" #n ", %16"
DEF_KERNEL(k_max_f64, MAX64(0), MAX64(1), MAX64(2), MAX64(3), MAX64(4), MAX64(5), MAX64(6), MAX64(7))
// f32 compares (VOPC writing vcc / exec): f32 rate or the 32-bit integer rate?  (round 3: can the block test of an
// f64 orbit loop compare the HIGH dword of |z|^2 at the f32 rate?)
#define CMPF(n) "v_cmp_gt_f32 vcc, " U##n ", %17"
DEF_KERNEL(k_cmp_f32, CMPF(0), CMPF(1), CMPF(2), CMPF(3), CMPF(4), CMPF(5), CMPF(6), CMPF(7))
#define CMPXF(n) "v_cmpx_le_f32 " U##n ", " U##n
DEF_KERNEL(k_cmpx_f32, CMPXF(0), CMPXF(1), CMPXF(2), CMPXF(3), CMPXF(4), CMPXF(5), CMPXF(6), CMPXF(7))
#define CMPXD(n) "v_cmpx_le_f64 %" #n ", %" #n
DEF_KERNEL(k_cmpx_f64, CMPXD(0), CMPXD(1), CMPXD(2), CMPXD(3), CMPXD(4), CMPXD(5), CMPXD(6), CMPXD(7))
// seven f64 ops + one compare: f64 compare against f32 compare
DEF_KERNEL(k_mix_7d_cmpd, OP3("v_mul_f64", 0), OP3("v_add_f64", 1), OP3("v_add_f64", 2), OP3("v_mul_f64", 3),
           OP3("v_add_f64", 4), OP3("v_add_f64", 5), OP3("v_mul_f64", 6), CMPXD(7))
DEF_KERNEL(k_mix_7d_cmpf, OP3("v_mul_f64", 0), OP3("v_add_f64", 1), OP3("v_add_f64", 2), OP3("v_mul_f64", 3),
           OP3("v_add_f64", 4), OP3("v_add_f64", 5), OP3("v_mul_f64", 6), CMPXF(7))
// mixed: 3 mul + 4 add + 1 cmp (the minimal orbit iteration)
DEF_KERNEL(k_mix_orbit, OP3("v_mul_f64", 0), OP3("v_add_f64", 1), OP3("v_add_f64", 2), OP3("v_mul_f64", 3),
           OP3("v_add_f64", 4), OP3("v_add_f64", 5), OP3("v_mul_f64", 6), CMP(7))
// mixed: f64 op alternating with a 32-bit op (does the 32-bit op hide behind the f64 op?)
DEF_KERNEL(k_mix_f64_u32, OP3("v_mul_f64", 0), MAXU(1), OP3("v_add_f64", 2), MAXU(3), OP3("v_mul_f64", 4), MAXU(5),
           OP3("v_add_f64", 6), MAXU(7))

// scalar ALU: does a wave's SALU work hide behind the other waves' VALU work, and at what rate does a SIMD
// take scalar instructions at all?  (Bodies of 32 instructions like the others.)
#define SALU_KERNEL(NAME, BODY)                                                                   \
    __global__ void NAME(unsigned long long *out, int iters, double seed) {                      \
        float a0 = (float)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, k = 1.0000001f;           \
        for (int i = 0; i < iters; i++) {                                                        \
            asm volatile(BODY BODY BODY BODY                                                      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                 \
                         : "v"(k)                                                                 \
                         : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");        \
        }                                                                                        \
        if (a0 + a1 + a2 + a3 == 12345.678f) out[0] = 0;                                          \
    }
#define S4 "s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n"
#define V4 "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
SALU_KERNEL(k_salu, S4 S4)                   // 8 scalar
SALU_KERNEL(k_valu32, V4 V4)                 // 8 f32 vector
SALU_KERNEL(k_v4s4, V4 S4)                   // 4 vector + 4 scalar
SALU_KERNEL(k_v6s2, V4 "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n")
SALU_KERNEL(k_v7s1, V4 "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n s_add_u32 s40, s40, 1\n")

typedef void (*kern_t)(unsigned long long *, int, double);

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int iters = 4000;
    struct K { const char *name; kern_t fn; } ks[] = {
        {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_fma_f64", k_fma_f64}, {"v_max_f64", k_max_f64},
        {"v_cmp_gt_f64", k_cmp_f64}, {"v_cmp_gt_u32", k_cmp_u32}, {"v_max_u32", k_max_u32},
        {"v_cmp_gt_f32", k_cmp_f32}, {"v_cmpx_le_f32", k_cmpx_f32}, {"v_cmpx_le_f64", k_cmpx_f64},
        {"7 f64 + v_cmpx_le_f64", k_mix_7d_cmpd}, {"7 f64 + v_cmpx_le_f32", k_mix_7d_cmpf},
        {"v_mov_b32", k_mov_b32}, {"v_mov_b64", k_mov_b64}, {"v_mul_f32", k_mul_f32},
        {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32},
        {"mix 3mul+4add+cmp f64", k_mix_orbit}, {"mix f64/u32 alternating", k_mix_f64_u32},
        {"s_add_u32", k_salu}, {"v_mul_f32 (4 chains)", k_valu32}, {"4 v_mul_f32 + 4 s_add_u32", k_v4s4},
        {"6 v_mul_f32 + 2 s_add_u32", k_v6s2}, {"7 v_mul_f32 + 1 s_add_u32", k_v7s1},
    };
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * 256 * 4 * 8 * 4);
    printf("%-26s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD; waves/SIMD = 1,2,4,8)\n", "instruction", "w=1", "w=2", "w=4", "w=8");
    for (auto &k : ks) {
        printf("%-26s", k.name);
        for (int w : {1, 2, 4, 8}) {
            // blocks of 256 threads = 4 waves, one per SIMD; w blocks per CU
            int blocks = prop.multiProcessorCount * w;
            int nw = blocks * 4;
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0);  // warm
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(nw);
            hipMemcpy(h.data(), d, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            double med = (double)h[nw / 2];
            // s_memtime counts at a fixed 100 MHz reference on gfx9; convert using wall time instead:
            // cycles/instr/SIMD = wall_s * f_clk / (iters*32*w); report both tick-based and wall-based @2.4GHz
            double wall_cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 32 * w);
            printf(" %8.2f", wall_cyc);
            (void)med;
        }
        printf("\n");
    }
    hipFree(d);
    return 0;
}
