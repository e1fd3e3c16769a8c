// LDS table-lookup cost on gfx950 by access width: random per-lane indices into small tables, the pattern of the
// sketch kernel's murmur3 lookup tables.  Reports LDS cycles per wave-level lookup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_lds tools/ubench_lds.hip ; run: ./tools/ubench_lds [GHz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 1000, REP = 8;

// MODE 0: b32, 256-entry dword table      1: two b32 reads, split lo/hi tables (read2st64 form)
//      2: b64, 256 x 8 B                  4: b128 of 256 x 16 B records
//      5: b64, 1024 x 8 B                 6: b32, 1024-entry dword table      7: b128 of 1024 x 16 B
//      8: b96 of 256 x 16 B records (three of the four dwords used)
template <int MODE>
__global__ __launch_bounds__(256) void k_lut(unsigned *out, unsigned seed) {
    __shared__ __attribute__((aligned(16))) unsigned T[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) T[i] = i * 2654435761u + seed;
    __syncthreads();
    constexpr unsigned ENT = (MODE >= 5 && MODE != 8) ? 1024u : 256u;
    constexpr unsigned RB = (MODE == 0 || MODE == 1 || MODE == 6) ? 4u : (MODE == 2 || MODE == 5) ? 8u : 16u; // record bytes
    unsigned lane_h = (threadIdx.x + blockIdx.x * 256u) * 2654435761u + seed;
    unsigned idx[4], inc[4];
    for (int q = 0; q < 4; ++q) {
        idx[q] = lane_h >> (3 + 5 * q);
        inc[q] = ((lane_h >> (2 * q + 1)) | 1u) * 7u;
    }
    unsigned acc = 0;
    const char *base = (const char *)T;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                idx[q] += inc[q];
                const unsigned off = (idx[q] & (ENT - 1u)) * RB;
                if (MODE == 0 || MODE == 6) acc ^= *(const unsigned *)(base + off);
                else if (MODE == 1) {
                    acc ^= *(const unsigned *)(base + off);
                    acc ^= *(const unsigned *)(base + 1024 + off);
                } else if (MODE == 2 || MODE == 5) {
                    const uint2 v = *(const uint2 *)(base + off);
                    acc ^= v.x ^ v.y;
                } else if (MODE == 8) {
                    const uint4 v = *(const uint4 *)(base + off);
                    acc ^= v.x ^ v.y ^ v.z;
                } else {
                    const uint4 v = *(const uint4 *)(base + off);
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kern_t)(unsigned *, unsigned);
static void run(const char *name, kern_t k, unsigned *d_out, int blocks, int cus, double clk_ghz) {
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
    const double wave_lookups = (double)blocks * 4.0 * ITER * REP * 4.0;
    const double cyc_per_cu = best * 1e-3 * clk_ghz * 1e9 * cus / wave_lookups;
    printf("%-34s %8.3f ms   %6.2f LDS cycles per wave-lookup per CU @%.2f GHz\n", name, best, cyc_per_cu, clk_ghz);
}

int main(int argc, char **argv) {
    const double clk = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *d_out;
    for (int bpc = 4; bpc <= 8; bpc *= 2) {
        const int blocks = cus * bpc;
        CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
        printf("-- %d blocks (x4 waves) per CU\n", bpc);
        run("b32   256-entry", k_lut<0>, d_out, blocks, cus, clk);
        run("2xb32 256-entry split lo/hi", k_lut<1>, d_out, blocks, cus, clk);
        run("b64   256 x 8 B", k_lut<2>, d_out, blocks, cus, clk);
        run("b128  256 x 16 B", k_lut<4>, d_out, blocks, cus, clk);
        run("b96   256 x 16 B", k_lut<8>, d_out, blocks, cus, clk);
        run("b32   1024-entry", k_lut<6>, d_out, blocks, cus, clk);
        run("b64   1024 x 8 B", k_lut<5>, d_out, blocks, cus, clk);
        run("b128  1024 x 16 B", k_lut<7>, d_out, blocks, cus, clk);
        CHECK(hipFree(d_out));
    }
    return 0;
}
