// fh_kernels.h -- launchers of the gfx950 kernels (fh_kernels.hip), used by the C ABI (fh_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "fh_device.h"

#define FH_NPARTS 4 // fh_k2.hip is compiled this many times, each part instantiating 32/FH_NPARTS values of K

namespace fh {

hipError_t launch_k2(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2_part0(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2_part1(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2_part2(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_k2_part3(int k, const SketchArgs &a, int blocks, hipStream_t st);
hipError_t launch_prune_small(Entry *table, uint32_t *live, uint32_t *dead, uint32_t dead_cap, Ctl *ctl, uint32_t kind,
                              uint64_t size, uint64_t max_hash, uint32_t trigger, uint32_t force, hipStream_t st);
hipError_t launch_clear_slots(Entry *table, uint64_t cap, const uint32_t *live, const uint32_t *dead, const Ctl *ctl,
                              hipStream_t st);
hipError_t launch_gather(const Entry *table, const uint32_t *live, const Ctl *ctl, int k, uint64_t *o_hash,
                         uint32_t *o_count, uint32_t *o_extra, uint64_t *o_kmer, uint64_t *o_pos, uint32_t cap_out,
                         hipStream_t st);
hipError_t launch_fill_table(Entry *table, uint64_t cap, hipStream_t st);
hipError_t launch_init_ctl(Ctl *ctl, uint64_t tau0, hipStream_t st);
hipError_t launch_synth_genome(uint8_t *out, uint64_t len, uint64_t seed, hipStream_t st);
hipError_t launch_synth_reads(uint8_t *out, const uint8_t *genome, uint64_t genome_len, uint64_t first_read,
                              uint64_t n_reads, uint32_t read_len, uint64_t seed, uint32_t sub_ppm, uint32_t n_ppm,
                              hipStream_t st);

} // namespace fh
