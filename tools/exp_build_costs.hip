// Micro-benchmarks behind NOTEBOOK.md §4 K1's pricing of a partitioned index build (r03): what the pieces a build is
// made of cost on this chip, each over N = 30 M items (a cancer-sized gene set's sites).
//   0  atomicCAS on a random 8-byte slot of a 467 MB table        (today's gf_k_index_insert, minus everything else)
//   1  atomicOr on a random word of a 15 MB array                  (unique flags, one atomic per site)
//   2  byte store at a random byte of a 30 MB array                (unique flags as bytes, no atomics)
//   3  look-then-atomicOr on a random word of a 7.7 MiB array, N/2 (the presence filter's fill)
//   4  8-byte store at a random 8-byte slot of a 240 MB array      (one-level scatter, no coalescing)
//   5  8-byte stores, 64 consecutive per random 512-byte chunk     (block-binned scatter: a partition pass's writes)
//   6  stream copy of 240 MB in 8-byte items                       (a partition pass's reads, for scale)
//   7  LDS build: 4096 items per block into an 8192-slot LDS table with 64-bit CAS, table written out (467 MB)
// Build: hipcc -O3 --offload-arch=gfx950 -o exp_build_costs exp_build_costs.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t i) {
  uint64_t h = i * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  return h ^ (h >> 32);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_exp(uint8_t* buf, uint64_t n_items, uint64_t span_bytes, const uint64_t* src) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthr = (uint64_t)gridDim.x * 256;
  for (uint64_t i = tid; i < n_items; i += nthr) {
    const uint64_t h = mix(i);
    if (MODE == 0) {
      unsigned long long* p = (unsigned long long*)buf + (h >> 8) % (span_bytes / 8);
      atomicCAS(p, 0ull, (unsigned long long)(h | 1));
    } else if (MODE == 1) {
      atomicOr((unsigned int*)buf + (h >> 8) % (span_bytes / 4), 1u << (h & 31));
    } else if (MODE == 2) {
      buf[(h >> 8) % span_bytes] = 1;
    } else if (MODE == 3) {
      unsigned int* p = (unsigned int*)buf + (h >> 8) % (span_bytes / 4);
      const uint32_t b = (1u << (h & 31)) | (1u << ((h >> 5) & 31));
      if ((__builtin_nontemporal_load(p) & b) != b) atomicOr(p, b);
    } else if (MODE == 4) {
      ((uint64_t*)buf)[(h >> 8) % (span_bytes / 8)] = h;
    } else if (MODE == 5) {
      const uint64_t chunk = mix(i >> 6) % (span_bytes / 512);
      ((uint64_t*)buf)[chunk * 64 + (i & 63)] = h;
    } else if (MODE == 6) {
      ((uint64_t*)buf)[i] = src[i] + 1;
    }
  }
}

// 7: one block per 64 KB slice of the table
__global__ __launch_bounds__(256) void k_lds_build(uint64_t* table, const uint64_t* __restrict__ items, int per_block) {
  __shared__ unsigned long long s_tab[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) s_tab[i] = 0;
  __syncthreads();
  const uint64_t* mine = items + (size_t)blockIdx.x * per_block;
  for (int i = threadIdx.x; i < per_block; i += 256) {
    const uint64_t it = mine[i];
    const uint32_t key = (uint32_t)(it >> 32) | 1u;
    uint32_t b = (uint32_t)(mix(key) >> 40) & 1023u;
    for (int guard = 0; guard < 1024; ++guard) {
      bool done = false;
      for (int j = 0; j < 8 && !done; ++j) {
        unsigned long long cur = s_tab[b * 8 + j];
        if (cur == 0) {
          const unsigned long long prev = atomicCAS(&s_tab[b * 8 + j], 0ull, ((unsigned long long)key << 32) | 1ull);
          if (prev == 0) { done = true; break; }
          cur = prev;
        }
        if ((uint32_t)(cur >> 32) == key) { atomicAdd((unsigned int*)&s_tab[b * 8 + j], 1u); done = true; }
      }
      if (done) break;
      b = (b + 1) & 1023u;
    }
  }
  __syncthreads();
  uint64_t* out = table + (size_t)blockIdx.x * 8192;
  for (int i = threadIdx.x; i < 8192; i += 256) out[i] = s_tab[i];
}

__global__ void k_fill(uint64_t* src, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) src[i] = mix(i ^ 0x1234567ull);
}

int main() {
  const uint64_t N = 30000000ull;
  const size_t big = 467ull << 20;
  uint8_t* buf;
  uint64_t* src;
  CK(hipMalloc(&buf, big));
  CK(hipMalloc(&src, N * 8));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, src, N);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const dim3 grid(256 * 16), blk(256);
  const char* names[] = {"atomicCAS 8B random in 467 MB", "atomicOr random in 15 MB", "byte store random in 30 MB",
                         "look+atomicOr random in 7.7 MiB (N/2)", "8B store random in 240 MB", "8B stores, 512B random chunks, 240 MB",
                         "stream copy 240 MB", "LDS build + table write (467 MB)"};
  for (int mode = 0; mode < 8; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(buf, 0, big));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_exp<0>, grid, blk, 0, 0, buf, N, (uint64_t)big, src); break;
        case 1: hipLaunchKernelGGL(k_exp<1>, grid, blk, 0, 0, buf, N, 15ull << 20, src); break;
        case 2: hipLaunchKernelGGL(k_exp<2>, grid, blk, 0, 0, buf, N, 30ull << 20, src); break;
        case 3: hipLaunchKernelGGL(k_exp<3>, grid, blk, 0, 0, buf, N / 2, 7700ull << 10, src); break;
        case 4: hipLaunchKernelGGL(k_exp<4>, grid, blk, 0, 0, buf, N, 240ull << 20, src); break;
        case 5: hipLaunchKernelGGL(k_exp<5>, grid, blk, 0, 0, buf, N, 240ull << 20, src); break;
        case 6: hipLaunchKernelGGL(k_exp<6>, grid, blk, 0, 0, buf, N, 240ull << 20, src); break;
        case 7: hipLaunchKernelGGL(k_lds_build, dim3((unsigned)(big / 65536)), blk, 0, 0, (uint64_t*)buf, src, (int)(N / (big / 65536))); break;
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("mode %d  %-42s %8.3f ms  (%.1f ps per item)\n", mode, names[mode], best, best * 1e9 / (mode == 3 ? N / 2 : N));
  }
  return 0;
}
