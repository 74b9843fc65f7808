// Micro-benchmark: can the scalar data path carry scattered 4-byte look-ups of an L2-resident table beside the
// vector path?  gf_k_seedverify_stream is bound by the L1's miss queue (NOTEBOOK.md §5, r01-r03's reading: ~86 requests in flight per
// CU x 351 cycles), at 56 % of the L2's look-up rate; scalar loads have a queue of their own.
//   ./mb_scalar_gather [table_KiB=3072] [iters=2000]
// mode V: 4 vector look-ups per lane and iteration; mode S: SPER scalar look-ups per wave and iteration;
// mode M: both.  Prints G look-ups/s of each kind.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef const __attribute__((address_space(4))) uint32_t* cptr32;

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h;
}

template <int VPER, int SPER>
__global__ __launch_bounds__(256) void k_mix(const uint32_t* __restrict__ tab, uint32_t nwords, int iters,
                                             uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t seed = tid * 0x9E3779B1u + 12345u;
  uint32_t sseed = __builtin_amdgcn_readfirstlane((tid >> 6) * 0x7F4A7C15u + 999u);
  uint32_t acc = 0, sacc = 0;
  cptr32 ctab = (cptr32)(uintptr_t)tab;
  for (int it = 0; it < iters; ++it) {
    uint32_t v[VPER > 0 ? VPER : 1];
#pragma unroll
    for (int p = 0; p < VPER; ++p) {
      seed = mix32(seed + 0x7F4A7C15u);
      v[p] = tab[(uint32_t)(((uint64_t)seed * nwords) >> 32)];
    }
    uint32_t s[SPER > 0 ? SPER : 1];
#pragma unroll
    for (int p = 0; p < SPER; ++p) {
      sseed = sseed * 1664525u + 1013904223u;   // (scalar ALU: the seed is uniform)
      const uint32_t w = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)(sseed ^ (sseed >> 15)) * nwords) >> 32));
      s[p] = ctab[w];
    }
#pragma unroll
    for (int p = 0; p < VPER; ++p) acc += v[p];
#pragma unroll
    for (int p = 0; p < SPER; ++p) sacc += s[p];
  }
  if ((acc ^ sacc) == 0x12345678u) out[tid] = acc;
}

template <int VPER, int SPER>
void run(const char* name, const uint32_t* tab, uint32_t nwords, int iters, uint32_t* out, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k_mix<VPER, SPER>), dim3(grid), dim3(256), 0, 0, tab, nwords, iters / 4, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k_mix<VPER, SPER>), dim3(grid), dim3(256), 0, 0, tab, nwords, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double vl = (double)grid * 256 * iters * VPER / (ms * 1e-3) / 1e9;
  const double sl = (double)grid * 4 * iters * SPER / (ms * 1e-3) / 1e9;
  printf("  %-22s grid %5d: %8.3f ms  vector %7.1f G/s  scalar %7.2f G/s\n", name, grid, ms, vl, sl);
}

int main(int argc, char** argv) {
  const int kib = argc > 1 ? atoi(argv[1]) : 3072;
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  const size_t bytes = (size_t)kib << 10;
  const uint32_t nwords = (uint32_t)(bytes / 4);
  uint32_t* tab; uint32_t* out;
  hipMalloc((void**)&tab, bytes); hipMemset(tab, 1, bytes);
  hipMalloc((void**)&out, 64u << 20);
  printf("table %d KiB\n", kib);
  for (int grid : {256 * 4, 256 * 8}) {
    run<4, 0>("V4", tab, nwords, iters, out, grid);
    run<0, 8>("S8", tab, nwords, iters, out, grid);
    run<0, 14>("S14", tab, nwords, iters, out, grid);
    run<4, 8>("V4+S8", tab, nwords, iters, out, grid);
    run<4, 14>("V4+S14", tab, nwords, iters, out, grid);
    run<2, 14>("V2+S14", tab, nwords, iters, out, grid);
  }
  return 0;
}
