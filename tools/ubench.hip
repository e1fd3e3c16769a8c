// ubench.hip -- VALU instruction-rate micro-benchmarks for gfx950 (feeds DESIGN.md's cost model of k2).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
// Each kernel issues ITER x 8 x UNROLL independent instances of one instruction on 8 register chains.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                            \
    do {                                                                                    \
        hipError_t e = (x);                                                                 \
        if (e != hipSuccess) {                                                              \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);    \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

constexpr int ITER = 2000;
constexpr int REP = 8; // instructions per chain per iteration

#define DEF_KERNEL32(NAME, ASM)                                                             \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {             \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
        unsigned a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;    \
        unsigned b = seed | 0x9E3779B1u, c = seed ^ 0x85EBCA6Bu;                            \
        for (int i = 0; i < ITER; ++i) {                                                    \
            _Pragma("unroll") for (int r = 0; r < REP; ++r) {                               \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)        \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),  \
                               "+v"(a6), "+v"(a7)                                           \
                             : "v"(b), "v"(c));                                             \
            }                                                                               \
        }                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
    }

#define A_MUL_LO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_MUL_HI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define A_MUL_U24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define A_MAD_U24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 13\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define A_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 9\n"
#define A_LSHL_OR(n) "v_lshl_or_b32 %" #n ", %" #n ", 2, %8\n"
#define A_AND_OR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define A_XAD(n) "v_xad_u32 %" #n ", %" #n ", %8, %9\n"
#define A_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x96\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_BFREV(n) "v_bfrev_b32 %" #n ", %" #n "\n"
#define A_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define A_ADDC(n) "v_addc_co_u32 %" #n ", vcc, %" #n ", %8, vcc\n"

#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n"
#define A_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define A_LSHL(n) "v_lshlrev_b32 %" #n ", 5, %" #n "\n"
#define A_LSHR(n) "v_lshrrev_b32 %" #n ", 5, %" #n "\n"
#define A_LSHLV(n) "v_lshlrev_b32 %" #n ", %8, %" #n "\n"
#define A_NOT(n) "v_not_b32 %" #n ", %" #n "\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define A_XORLIT(n) "v_xor_b32 %" #n ", 0x12345678, %" #n "\n"
#define A_ANDLIT(n) "v_and_b32 %" #n ", 0x0f0f0f0f, %" #n "\n"
#define A_MIN(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define A_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define A_MULF(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define A_ADDF(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define A_CMPCND(n) "v_cmp_lt_u32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define A_CMP32(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define A_XOR_MUL(n) "v_xor_b32 %" #n ", %" #n ", %8\nv_mul_lo_u32 %" #n ", %" #n ", %9\n"
#define A_XOR_ALIGN(n) "v_xor_b32 %" #n ", %" #n ", %8\nv_alignbit_b32 %" #n ", %" #n ", %9, 7\n"
#define A_MUL_ALIGN(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\nv_alignbit_b32 %" #n ", %" #n ", %9, 7\n"
#define A_XOR_ADD(n) "v_xor_b32 %" #n ", %" #n ", %8\nv_add_u32 %" #n ", %" #n ", %9\n"
#define A_SDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define A_DPP(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_MINF32(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define A_MINF32ABS(n) "v_min_f32_e64 %" #n ", |%" #n "|, %8\n"
#define A_CMPNGTF32(n) "v_cmp_ngt_f32 vcc, %" #n ", %8\n"
#define A_CMPNGTF32ABS(n) "v_cmp_ngt_f32_e64 vcc, |%" #n "|, %8\n"
#define A_CMPGEU32(n) "v_cmp_ge_u32 vcc, %" #n ", %8\n"
#define A_LSHLADD32(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define A_DOT4(n) "v_dot4_u32_u8 %" #n ", %" #n ", %8, %9\n"
#define A_PKMULLO16(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define A_PKADD16(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define A_MAXU32(n) "v_max_u32 %" #n ", %" #n ", %8\n"
#define A_LSHRV(n) "v_lshrrev_b32 %" #n ", %8, %" #n "\n"
#define A_ASHR(n) "v_ashrrev_i32 %" #n ", 5, %" #n "\n"
#define A_MIN3U32(n) "v_min3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_SUBREV(n) "v_subrev_u32 %" #n ", %8, %" #n "\n"
#define A_CVTF32U32(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %" #n ", %8, %9\n"
#define A_ADDU32_X2(n) "v_add_u32 %" #n ", %" #n ", %" #n "\n"
DEF_KERNEL32(k_minf32, A_MINF32)
DEF_KERNEL32(k_minf32abs, A_MINF32ABS)
DEF_KERNEL32(k_cmpngtf32, A_CMPNGTF32)
DEF_KERNEL32(k_cmpngtf32abs, A_CMPNGTF32ABS)
DEF_KERNEL32(k_cmpgeu32, A_CMPGEU32)
DEF_KERNEL32(k_lshladd32, A_LSHLADD32)
DEF_KERNEL32(k_dot4, A_DOT4)
DEF_KERNEL32(k_pkmullo16, A_PKMULLO16)
DEF_KERNEL32(k_pkadd16, A_PKADD16)
DEF_KERNEL32(k_maxu32, A_MAXU32)
DEF_KERNEL32(k_lshrv, A_LSHRV)
DEF_KERNEL32(k_ashr, A_ASHR)
DEF_KERNEL32(k_min3u32, A_MIN3U32)
DEF_KERNEL32(k_subrev, A_SUBREV)
DEF_KERNEL32(k_cvtf32u32, A_CVTF32U32)
DEF_KERNEL32(k_bfi, A_BFI)
DEF_KERNEL32(k_addx2, A_ADDU32_X2)
DEF_KERNEL32(k_and, A_AND)
DEF_KERNEL32(k_or, A_OR)
DEF_KERNEL32(k_sub, A_SUB)
DEF_KERNEL32(k_lshl, A_LSHL)
DEF_KERNEL32(k_lshr, A_LSHR)
DEF_KERNEL32(k_lshlv, A_LSHLV)
DEF_KERNEL32(k_not, A_NOT)
DEF_KERNEL32(k_mov, A_MOV)
DEF_KERNEL32(k_xorlit, A_XORLIT)
DEF_KERNEL32(k_andlit, A_ANDLIT)
DEF_KERNEL32(k_min, A_MIN)
DEF_KERNEL32(k_fma, A_FMA)
DEF_KERNEL32(k_fmac, A_FMAC)
DEF_KERNEL32(k_mulf, A_MULF)
DEF_KERNEL32(k_addf, A_ADDF)
DEF_KERNEL32(k_cmpcnd, A_CMPCND)
DEF_KERNEL32(k_cmp32, A_CMP32)
DEF_KERNEL32(k_xor_mul, A_XOR_MUL)
DEF_KERNEL32(k_xor_align, A_XOR_ALIGN)
DEF_KERNEL32(k_mul_align, A_MUL_ALIGN)
DEF_KERNEL32(k_xor_add, A_XOR_ADD)
DEF_KERNEL32(k_sdwa, A_SDWA)
DEF_KERNEL32(k_dpp, A_DPP)
DEF_KERNEL32(k_mul_lo, A_MUL_LO)
DEF_KERNEL32(k_mul_hi, A_MUL_HI)
DEF_KERNEL32(k_mul_u24, A_MUL_U24)
DEF_KERNEL32(k_mad_u24, A_MAD_U24)
DEF_KERNEL32(k_xor, A_XOR)
DEF_KERNEL32(k_add, A_ADD)
DEF_KERNEL32(k_add3, A_ADD3)
DEF_KERNEL32(k_alignbit, A_ALIGNBIT)
DEF_KERNEL32(k_perm, A_PERM)
DEF_KERNEL32(k_bfe, A_BFE)
DEF_KERNEL32(k_lshl_or, A_LSHL_OR)
DEF_KERNEL32(k_and_or, A_AND_OR)
DEF_KERNEL32(k_xad, A_XAD)
DEF_KERNEL32(k_bitop3, A_BITOP3)
DEF_KERNEL32(k_cndmask, A_CNDMASK)
DEF_KERNEL32(k_bfrev, A_BFREV)
DEF_KERNEL32(k_addco, A_ADDCO)
DEF_KERNEL32(k_addc, A_ADDC)

// 64-bit register-pair instructions
#define DEF_KERNEL64(NAME, ASM)                                                             \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {             \
        unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
        unsigned long long a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;    \
        unsigned b = seed | 0x9E3779B1u;                                                    \
        unsigned long long c = ((unsigned long long)seed << 32) ^ 0x85EBCA6B12345ull;       \
        for (int i = 0; i < ITER; ++i) {                                                    \
            _Pragma("unroll") for (int r = 0; r < REP; ++r) {                               \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)        \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),  \
                               "+v"(a6), "+v"(a7)                                           \
                             : "v"(b), "v"(c));                                             \
            }                                                                               \
        }                                                                                   \
        unsigned long long x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)x ^ (unsigned)(x >> 32);     \
    }

// v_mad_u64_u32 D(64), vcc-like sdst, S0(32), S1(32), S2(64):  D = S0*S1 + S2
#define A_MAD64(n) "v_mad_u64_u32 %" #n ", s[4:5], %8, %8, %" #n "\n"
#define A_LSHL64(n) "v_lshlrev_b64 %" #n ", 3, %" #n "\n"
#define A_LSHR64(n) "v_lshrrev_b64 %" #n ", 3, %" #n "\n"
#define A_LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 2, %9\n"
#define A_CMPLT64(n) "v_cmp_lt_u64 vcc, %" #n ", %9\n"
#define A_MULF64(n) "v_mul_f64 %" #n ", %" #n ", %9\n"
#define A_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %9\n"

#define A_MINF64(n) "v_min_f64 %" #n ", %" #n ", %9\n"
#define A_MAXF64(n) "v_max_f64 %" #n ", %" #n ", %9\n"
#define A_FMAF64(n) "v_fma_f64 %" #n ", %" #n ", %9, %9\n"
#define A_ADDF64(n) "v_add_f64 %" #n ", %" #n ", %9\n"
DEF_KERNEL64(k_minf64, A_MINF64)
DEF_KERNEL64(k_maxf64, A_MAXF64)
DEF_KERNEL64(k_fmaf64, A_FMAF64)
DEF_KERNEL64(k_addf64, A_ADDF64)
DEF_KERNEL64(k_mad64, A_MAD64)
DEF_KERNEL64(k_lshl64, A_LSHL64)
DEF_KERNEL64(k_lshr64, A_LSHR64)
DEF_KERNEL64(k_lshladd64, A_LSHLADD64)
DEF_KERNEL64(k_cmplt64, A_CMPLT64)

// compiler-generated 64-bit multiply by a constant, 4 chains
__global__ __launch_bounds__(256) void k_mul64c(unsigned *out, unsigned seed) {
    unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    unsigned long long a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            a0 *= 0x87c37b91114253d5ULL; a1 *= 0x87c37b91114253d5ULL; a2 *= 0x87c37b91114253d5ULL; a3 *= 0x87c37b91114253d5ULL;
            a4 *= 0x87c37b91114253d5ULL; a5 *= 0x87c37b91114253d5ULL; a6 *= 0x87c37b91114253d5ULL; a7 *= 0x87c37b91114253d5ULL;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    unsigned long long x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)x ^ (unsigned)(x >> 32);
}

// random ds_read_b64 / b32 from a 2 KiB table
template <int WIDE>
__global__ __launch_bounds__(256) void k_lds_lut(unsigned *out, unsigned seed) {
    __shared__ unsigned long long T[256];
    T[threadIdx.x] = threadIdx.x * 0x9E3779B97F4A7C15ull + seed;
    __syncthreads();
    unsigned idx0 = threadIdx.x * 2654435761u + seed, idx1 = idx0 * 3 + 1, idx2 = idx0 * 5 + 7, idx3 = idx0 * 7 + 11;
    unsigned long long acc = 0;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (WIDE) {
                acc += T[(idx0 >> 7) & 255]; acc ^= T[(idx1 >> 9) & 255]; acc += T[(idx2 >> 11) & 255]; acc ^= T[(idx3 >> 13) & 255];
            } else {
                const unsigned *T32 = (const unsigned *)T;
                acc += T32[2 * ((idx0 >> 7) & 255)]; acc ^= T32[2 * ((idx1 >> 9) & 255)];
                acc += T32[2 * ((idx2 >> 11) & 255)]; acc ^= T32[2 * ((idx3 >> 13) & 255)];
            }
            idx0 = idx0 * 1664525u + 1013904223u + (unsigned)acc; idx1 += idx0; idx2 ^= idx1; idx3 += idx2;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)acc ^ (unsigned)(acc >> 32);
}

// wall clock vs shader clock: a kernel that spins on s_memtime for a fixed number of ticks
__global__ void k_clock(unsigned long long *out, unsigned long long ticks) {
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t - t0;
}

// Do v_min_f64 / v_cmp_ngt_f32 order integer bit patterns the way the sketch kernel wants to use them?
//   min64: for a, b < 2^62 (sign 0, exponent never all ones; denormals included) v_min_f64 must return min(a, b) bit for bit
//   cmp32: "not (|X| > B)" on the bit patterns must be true whenever (X & 0x7fffffff) <= B  (NaN patterns of X count as true)
__global__ void k_semantics(unsigned long long *bad, unsigned seed) {
    unsigned long long x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + seed;
    unsigned long long nb64 = 0, nb32 = 0, nfalsepos = 0;
    for (int i = 0; i < 4096; ++i) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        unsigned long long a = x >> 2, b = (x * 0x94D049BB133111EBull) >> 2;
        const int sh = (i & 63);
        if (i & 64) { a >>= sh; b >>= (sh ^ 5) & 63; }          // small values: denormals and zero
        if ((i & 255) == 7) b = a;                               // ties
        if ((i & 255) == 9) b = a ^ 1ull;                        // differ in the last bit
        unsigned long long m;
        asm volatile("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
        if (m != (a < b ? a : b)) ++nb64;
        unsigned X = (unsigned)(x >> 17), B = (unsigned)((x >> 40) & 0xFFFFF) >> (i & 15);
        if ((i & 3) == 1) X >>= (i >> 2) & 31;
        unsigned long long vc;
        asm volatile("v_cmp_ngt_f32_e64 %0, |%1|, %2" : "=s"(vc) : "v"(X), "v"(B));
        const bool f = (vc >> (threadIdx.x & 63)) & 1ull;
        const bool want = (X & 0x7FFFFFFFu) <= B;
        if (want && !f) ++nb32;
        if (f && !want && ((X & 0x7F800000u) != 0x7F800000u)) ++nfalsepos;
    }
    atomicAdd(&bad[0], nb64); atomicAdd(&bad[1], nb32); atomicAdd(&bad[2], nfalsepos);
}

typedef void (*kern_t)(unsigned *, unsigned);

static void run(const char *name, kern_t k, double inst_per_thread, unsigned *d_out, int blocks, double clk_ghz) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 2u + t);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waves = (double)blocks * 4;
    const double wave_inst = waves * inst_per_thread;
    const double per_s = wave_inst / (best * 1e-3);
    // cycles per wave-instruction per SIMD at the given clock (1024 SIMDs)
    const double cyc = clk_ghz * 1e9 * 1024.0 / per_s;
    printf("%-14s %8.3f ms  %8.2f G wave-inst/s  ~%5.2f cyc/inst/SIMD @%.2f GHz\n", name, best, per_s * 1e-9, cyc, clk_ghz);
}

int main(int argc, char **argv) {
    double clk = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int blocks = prop.multiProcessorCount * 8; // 8 blocks x 4 waves = 32 waves/CU
    unsigned *d_out;
    CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    {
        unsigned long long *d_t;
        CHECK(hipMalloc(&d_t, 8));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d_t, 100000000ull);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long t;
            CHECK(hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost));
            printf("s_memtime: %llu ticks in %.3f ms => %.1f MHz tick rate\n", t, ms, t / (ms * 1e3));
        }
    }
    const double n32 = (double)ITER * REP * 8;
    {
        unsigned long long *d_bad, h_bad[3] = {0, 0, 0};
        CHECK(hipMalloc(&d_bad, 24));
        CHECK(hipMemset(d_bad, 0, 24));
        hipLaunchKernelGGL(k_semantics, dim3(1024), dim3(256), 0, 0, d_bad, 12345u);
        CHECK(hipMemcpy(h_bad, d_bad, 24, hipMemcpyDeviceToHost));
        printf("semantics over %llu cases: v_min_f64 != integer min: %llu   v_cmp_ngt_f32(|X|, B) missed (X & 0x7fffffff) <= B: %llu   (extra non-NaN trues: %llu)\n",
               1024ull * 256 * 4096, h_bad[0], h_bad[1], h_bad[2]);
    }
    run("v_min_f64", k_minf64, n32, d_out, blocks, clk);
    run("v_max_f64", k_maxf64, n32, d_out, blocks, clk);
    run("v_fma_f64", k_fmaf64, n32, d_out, blocks, clk);
    run("v_add_f64", k_addf64, n32, d_out, blocks, clk);
    run("v_min_f32", k_minf32, n32, d_out, blocks, clk);
    run("v_min_f32 |x|", k_minf32abs, n32, d_out, blocks, clk);
    run("v_cmp_ngt_f32", k_cmpngtf32, n32, d_out, blocks, clk);
    run("v_cmp_ngt_f32 |x|", k_cmpngtf32abs, n32, d_out, blocks, clk);
    run("v_cmp_ge_u32", k_cmpgeu32, n32, d_out, blocks, clk);
    run("v_lshl_add_u32", k_lshladd32, n32, d_out, blocks, clk);
    run("v_dot4_u32_u8", k_dot4, n32, d_out, blocks, clk);
    run("v_pk_mul_lo_u16", k_pkmullo16, n32, d_out, blocks, clk);
    run("v_pk_add_u16", k_pkadd16, n32, d_out, blocks, clk);
    run("v_max_u32", k_maxu32, n32, d_out, blocks, clk);
    run("v_lshrrev_b32 v", k_lshrv, n32, d_out, blocks, clk);
    run("v_ashrrev_i32", k_ashr, n32, d_out, blocks, clk);
    run("v_min3_u32", k_min3u32, n32, d_out, blocks, clk);
    run("v_subrev_u32", k_subrev, n32, d_out, blocks, clk);
    run("v_cvt_f32_u32", k_cvtf32u32, n32, d_out, blocks, clk);
    run("v_bfi_b32", k_bfi, n32, d_out, blocks, clk);
    run("v_add_u32 x,x", k_addx2, n32, d_out, blocks, clk);
    if (argc > 2) return 0; // (a second argument: only the round-5 additions above)
    run("v_and_b32", k_and, n32, d_out, blocks, clk);
    run("v_or_b32", k_or, n32, d_out, blocks, clk);
    run("v_sub_u32", k_sub, n32, d_out, blocks, clk);
    run("v_lshlrev_b32 i", k_lshl, n32, d_out, blocks, clk);
    run("v_lshrrev_b32 i", k_lshr, n32, d_out, blocks, clk);
    run("v_lshlrev_b32 v", k_lshlv, n32, d_out, blocks, clk);
    run("v_not_b32", k_not, n32, d_out, blocks, clk);
    run("v_mov_b32", k_mov, n32, d_out, blocks, clk);
    run("v_xor lit", k_xorlit, n32, d_out, blocks, clk);
    run("v_and lit", k_andlit, n32, d_out, blocks, clk);
    run("v_min_u32", k_min, n32, d_out, blocks, clk);
    run("v_fma_f32", k_fma, n32, d_out, blocks, clk);
    run("v_fmac_f32", k_fmac, n32, d_out, blocks, clk);
    run("v_mul_f32", k_mulf, n32, d_out, blocks, clk);
    run("v_add_f32", k_addf, n32, d_out, blocks, clk);
    run("cmp+cndmask x2", k_cmpcnd, 2 * n32, d_out, blocks, clk);
    run("v_cmp_lt_u32", k_cmp32, n32, d_out, blocks, clk);
    run("xor+mul_lo x2", k_xor_mul, 2 * n32, d_out, blocks, clk);
    run("xor+alignbit x2", k_xor_align, 2 * n32, d_out, blocks, clk);
    run("mul+alignbit x2", k_mul_align, 2 * n32, d_out, blocks, clk);
    run("xor+add x2", k_xor_add, 2 * n32, d_out, blocks, clk);
    run("v_add_sdwa", k_sdwa, n32, d_out, blocks, clk);
    run("v_mov_dpp", k_dpp, n32, d_out, blocks, clk);
    // occupancy sweep on two representative instructions
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        char nm[64];
        snprintf(nm, sizeof nm, "xor @%d blk/CU", bpc);
        run(nm, k_xor, n32, d_out, prop.multiProcessorCount * bpc, clk);
        snprintf(nm, sizeof nm, "mul_lo @%d blk/CU", bpc);
        run(nm, k_mul_lo, n32, d_out, prop.multiProcessorCount * bpc, clk);
    }
    run("v_xor_b32", k_xor, n32, d_out, blocks, clk);
    run("v_add_u32", k_add, n32, d_out, blocks, clk);
    run("v_add3_u32", k_add3, n32, d_out, blocks, clk);
    run("v_alignbit", k_alignbit, n32, d_out, blocks, clk);
    run("v_perm_b32", k_perm, n32, d_out, blocks, clk);
    run("v_bfe_u32", k_bfe, n32, d_out, blocks, clk);
    run("v_lshl_or", k_lshl_or, n32, d_out, blocks, clk);
    run("v_and_or", k_and_or, n32, d_out, blocks, clk);
    run("v_xad_u32", k_xad, n32, d_out, blocks, clk);
    run("v_bitop3", k_bitop3, n32, d_out, blocks, clk);
    run("v_cndmask", k_cndmask, n32, d_out, blocks, clk);
    run("v_bfrev", k_bfrev, n32, d_out, blocks, clk);
    run("v_add_co", k_addco, n32, d_out, blocks, clk);
    run("v_addc_co", k_addc, n32, d_out, blocks, clk);
    run("v_mul_u32_u24", k_mul_u24, n32, d_out, blocks, clk);
    run("v_mad_u32_u24", k_mad_u24, n32, d_out, blocks, clk);
    run("v_mul_lo_u32", k_mul_lo, n32, d_out, blocks, clk);
    run("v_mul_hi_u32", k_mul_hi, n32, d_out, blocks, clk);
    run("v_mad_u64_u32", k_mad64, n32, d_out, blocks, clk);
    run("v_lshlrev_b64", k_lshl64, n32, d_out, blocks, clk);
    run("v_lshrrev_b64", k_lshr64, n32, d_out, blocks, clk);
    run("v_lshl_add_u64", k_lshladd64, n32, d_out, blocks, clk);
    run("v_cmp_lt_u64", k_cmplt64, n32, d_out, blocks, clk);
    run("mul64 by const", k_mul64c, n32, d_out, blocks, clk);
    run("lds lut b64 x4", k_lds_lut<1>, (double)ITER * REP * 4, d_out, blocks, clk);
    run("lds lut b32 x4", k_lds_lut<0>, (double)ITER * REP * 4, d_out, blocks, clk);
    return 0;
}
