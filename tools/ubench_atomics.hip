// Random read-modify-write rate of the device memory system on gfx950 as the sketch kernel's admit path uses it:
// every lane updates one 40-byte entry at a random index of a table of a given size.  Reports G entry-updates/s by
// table size and by the number / kind of atomics per update.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_atomics tools/ubench_atomics.hip ; run: ./tools/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned long long ull;
struct Entry { ull hash, kmer, pos, count, extra; };

__device__ inline uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// MODE 0: the admit path's five atomics (CAS key, add count, add extra, min pos, CAS kmer)
//      1: CAS key only          2: one no-return add on the entry       3: plain 8-byte load of the key
//      4: CAS key + no-return add count (two atomics, one returning)
//      5: load key, then the four other atomics (load instead of CAS when the key is there)
template <int MODE>
__global__ __launch_bounds__(256) void k_upd(Entry *t, uint64_t n_entries, int per_lane, uint64_t seed, ull *sink) {
    const uint64_t gid = blockIdx.x * 256ull + threadIdx.x;
    ull acc = 0;
    for (int i = 0; i < per_lane; ++i) {
        const uint64_t r = mix(seed + gid * 1315423911ull + (uint64_t)i * 0x9E3779B97F4A7C15ull);
        const uint64_t slot = (uint64_t)(((unsigned __int128)r * n_entries) >> 64);
        Entry *e = &t[slot];
        const ull h = (ull)slot + 1;
        if (MODE == 0) {
            ull old = atomicCAS(&e->hash, ~0ull, h);
            acc += old;
            atomicAdd(&e->count, 1ull);
            if (r & 1) atomicAdd(&e->extra, 1ull);
            atomicMin(&e->pos, (ull)(r >> 8));
            ull ok = atomicCAS(&e->kmer, ~0ull, h * 3);
            acc += ok;
        } else if (MODE == 1) {
            acc += atomicCAS(&e->hash, ~0ull, h);
        } else if (MODE == 2) {
            atomicAdd(&e->count, 1ull);
        } else if (MODE == 3) {
            acc += __hip_atomic_load(&e->hash, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 4) {
            acc += atomicCAS(&e->hash, ~0ull, h);
            atomicAdd(&e->count, 1ull);
        } else if (MODE == 5) {
            ull old = __hip_atomic_load(&e->hash, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old != h) old = atomicCAS(&e->hash, ~0ull, h);
            acc += old;
            atomicAdd(&e->count, 1ull);
            if (r & 1) atomicAdd(&e->extra, 1ull);
            atomicMin(&e->pos, (ull)(r >> 8));
            ull ok = __hip_atomic_load(&e->kmer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ok != h * 3) ok = atomicCAS(&e->kmer, ~0ull, h * 3);
            acc += ok;
        }
    }
    if (acc == 0x1234567) sink[0] = acc;
}

// the form the kernel uses: the entry's key, k-mer and first position are loaded (one round trip); an occurrence
// of a hash that is already there then only needs its counter adds.  PACKED: both counters in one add.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_upd_loads(Entry *t, uint64_t n_entries, int per_lane, uint64_t seed, ull *sink) {
    const uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    ull acc = 0;
    for (int i = 0; i < per_lane; ++i) {
        const uint64_t r = mix(seed + gid * 1315423911ull + (uint64_t)i * 0x9E3779B97F4A7C15ull);
        const uint64_t slot = (uint64_t)(((unsigned __int128)r * n_entries) >> 64);
        Entry *e = &t[slot];
        const ull h = (ull)slot + 1, pos = (ull)(r >> 8) | (1ull << 60);
        ull old = __hip_atomic_load(&e->hash, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ull ok = __hip_atomic_load(&e->kmer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ull op = __hip_atomic_load(&e->pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old != h) old = atomicCAS(&e->hash, ~0ull, h);
        if (PACKED) atomicAdd(&e->count, 1ull | ((r & 1) << 32));
        else {
            atomicAdd(&e->count, 1ull);
            if (r & 1) atomicAdd(&e->extra, 1ull);
        }
        if (op > pos) atomicMin(&e->pos, pos);
        if (ok != h * 3) ok = atomicCAS(&e->kmer, ~0ull, h * 3);
        acc += ok + old;
    }
    if (acc == 0x1234567) sink[0] = acc;
}

template <bool PACKED>
static double run_loads(Entry *t, uint64_t n_entries, ull *sink) {
    const int blocks = 256 * 4 * 4, per_lane = 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) k_upd_loads<PACKED><<<blocks, 256>>>(t, n_entries, per_lane, 77 + r, sink); // same seeds: the timed runs find their entries
    k_upd_loads<PACKED><<<blocks, 256>>>(t, n_entries, per_lane, 77 + 3, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) k_upd_loads<PACKED><<<blocks, 256>>>(t, n_entries, per_lane, 77 + r, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return 4.0 * blocks * 256.0 * per_lane / (ms * 1e-3) / 1e9;
}

template <int MODE>
static double run(Entry *t, uint64_t n_entries, ull *sink) {
    const int blocks = 256 * 4 * 4, per_lane = 16; // 16 waves per CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_upd<MODE><<<blocks, 256>>>(t, n_entries, per_lane, 1, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) k_upd<MODE><<<blocks, 256>>>(t, n_entries, per_lane, 77 + r, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return 4.0 * blocks * 256.0 * per_lane / (ms * 1e-3) / 1e9;
}

// latency under light load: the read-first update with 1..16 waves per CU (the sketch kernel keeps only a few
// percent of its waves in the admit path at any time)
static void occupancy_sweep(Entry *t, uint64_t n_entries, ull *sink) {
    printf("waves/CU   G updates/s   us per update per lane   (3 loads + adds, %llu MiB table)\n", (ull)(n_entries * sizeof(Entry) >> 20));
    for (int wpc : {1, 2, 4, 8, 16}) {
        const int blocks = 256 * wpc, per_lane = 64; // one wave per block
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_upd_loads<false>, dim3(blocks), dim3(64), 0, 0, t, n_entries, per_lane, 5ull, sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_upd_loads<false>, dim3(blocks), dim3(64), 0, 0, t, n_entries, per_lane, 5ull, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double upd = 4.0 * blocks * 64.0 * per_lane;
        printf("%5d      %8.2f      %8.2f\n", wpc, upd / (ms * 1e-3) / 1e9, (ms * 1e3 / 4.0) / per_lane);
    }
}

int main() {
    ull *sink; CHECK(hipMalloc(&sink, 8));
    const uint64_t sizes_mb[] = {8, 64, 192, 512, 2048, 8192};
    printf("table      G updates/s:  5 atomics   CAS only   1 add   load only  CAS+add  load-first 5   3 loads+adds   3 loads+1 add\n");
    for (uint64_t mb : sizes_mb) {
        const uint64_t n = (mb << 20) / sizeof(Entry);
        Entry *t; CHECK(hipMalloc(&t, n * sizeof(Entry)));
        CHECK(hipMemset(t, 0xFF, n * sizeof(Entry)));
        CHECK(hipDeviceSynchronize());
        double a = run<0>(t, n, sink), b = run<1>(t, n, sink), c = run<2>(t, n, sink), d = run<3>(t, n, sink), e = run<4>(t, n, sink), f = run<5>(t, n, sink), g = run_loads<false>(t, n, sink), h2 = run_loads<true>(t, n, sink);
        printf("%5llu MiB              %8.2f   %8.2f  %8.2f  %8.2f  %8.2f  %8.2f  %8.2f  %8.2f\n", (ull)mb, a, b, c, d, e, f, g, h2);
        if (mb == 2048) occupancy_sweep(t, n, sink);
        CHECK(hipFree(t));
    }
    return 0;
}
