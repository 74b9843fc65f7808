// Micro-benchmark: random bucket gathers from an HBM / Infinity-Cache resident table.
// Measures the request-rate ceiling of the probe pattern used by gf_k_map_reads:
//   ./mb_gather <table_MiB> <bytes_per_probe 16|32|64> <probes_in_flight 1..8> [iters]
// Each lane issues `inflight` independent probes per iteration at hashed addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h;
}

template <int BYTES, int INFLIGHT>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ tab, uint32_t nbuckets, int iters,
                                                uint32_t* __restrict__ out) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  uint32_t seed = tid * 0x9E3779B1u + 12345u;
  for (int it = 0; it < iters; ++it) {
    uint4 v[INFLIGHT][BYTES / 16];
#pragma unroll
    for (int p = 0; p < INFLIGHT; ++p) {
      seed = mix32(seed + 0x7F4A7C15u);
      uint32_t b = (uint32_t)(((uint64_t)seed * nbuckets) >> 32);
      const uint4* q = tab + (size_t)b * 4;  // 64-byte buckets
#pragma unroll
      for (int j = 0; j < BYTES / 16; ++j) v[p][j] = q[j];
    }
#pragma unroll
    for (int p = 0; p < INFLIGHT; ++p)
#pragma unroll
      for (int j = 0; j < BYTES / 16; ++j) acc += v[p][j].x ^ v[p][j].y ^ v[p][j].z ^ v[p][j].w;
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

template <int BYTES, int INFLIGHT>
double run(const uint4* tab, uint32_t nb, int iters, uint32_t* out, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k_gather<BYTES, INFLIGHT>), dim3(grid), dim3(256), 0, 0, tab, nb, iters / 4, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k_gather<BYTES, INFLIGHT>), dim3(grid), dim3(256), 0, 0, tab, nb, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double probes = (double)grid * 256 * iters * INFLIGHT;
  return probes / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  int mib = argc > 1 ? atoi(argv[1]) : 128;
  int iters = argc > 2 ? atoi(argv[2]) : 64;
  size_t bytes = (size_t)mib << 20;
  uint32_t nb = (uint32_t)(bytes / 64);
  uint4* tab; uint32_t* out;
  hipMalloc((void**)&tab, bytes); hipMemset(tab, 1, bytes);
  hipMalloc((void**)&out, 64u << 20);
  int grids_default[] = {256 * 4, 256 * 8, 256 * 16};
  int grids_small[] = {32, 64, 128, 256, 512, 1024};
  bool small = argc > 3;
  int* grids_p = small ? grids_small : grids_default;
  int ng = small ? 6 : 3;
  std::vector<int> grids(grids_p, grids_p + ng);
  printf("table %d MiB, %u buckets\n", mib, nb);
  for (int g : grids) {
    printf("grid %5d:", g);
    printf("  16B x1 %.1f x4 %.1f x8 %.1f |", run<16, 1>(tab, nb, iters, out, g), run<16, 4>(tab, nb, iters, out, g), run<16, 8>(tab, nb, iters, out, g));
    printf("  32B x1 %.1f x4 %.1f |", run<32, 1>(tab, nb, iters, out, g), run<32, 4>(tab, nb, iters, out, g));
    printf("  64B x1 %.1f x2 %.1f x4 %.1f  Gprobes/s\n", run<64, 1>(tab, nb, iters, out, g), run<64, 2>(tab, nb, iters, out, g), run<64, 4>(tab, nb, iters, out, g));
  }
  return 0;
}
