// Counter calibration for profiles/hbm_traffic.json: kernels whose byte counts are
// KNOWN, launched by bench.py --calib on the very buffer gf_k_seedverify_stream streams, so that their
// FETCH_SIZE / TCC_EA0_RDREQ land in the same rocprofv3 pass as the mapping kernels'.
//   mode 0  16 B per lane, non-temporal, into LDS   (the ASCII staging loads of seed+verify)
//   mode 1  16 B per lane, plain
//   mode 2  4 B per lane, non-temporal              (the packed hand-over's g_pk loads)
//   mode 3  one dword per lane at a hashed 64-byte line of the buffer (scattered L2-missing requests:
//           bucket / gdu / filter-miss traffic); n_bytes / 64 loads in all
// Built as tools/libgfcalib.so by tools/Makefile; not part of libgfmatch.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void gf_k_calib_stream(const uint8_t* __restrict__ buf, int64_t n_bytes,
                                                         uint32_t* __restrict__ sink) {
  __shared__ volatile uint32_t s[4 * 256];  // (volatile: every iteration's LDS writes stay)
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthr = (int64_t)gridDim.x * 256;
  u32x4 acc = {0, 0, 0, 0};
  if (MODE <= 1) {
    const u32x4* p = (const u32x4*)buf;
    for (int64_t i = tid; i < n_bytes / 16; i += nthr) {
      const u32x4 v = MODE == 0 ? __builtin_nontemporal_load(p + i) : p[i];
      s[threadIdx.x] = v.x; s[256 + threadIdx.x] = v.y; s[512 + threadIdx.x] = v.z; s[768 + threadIdx.x] = v.w;  // into LDS, as the staging does
      acc.x ^= s[threadIdx.x ^ 1];
    }
  } else if (MODE == 2) {
    const uint32_t* p = (const uint32_t*)buf;
    for (int64_t i = tid; i < n_bytes / 4; i += nthr) acc.x ^= __builtin_nontemporal_load(p + i);
  } else {
    const uint32_t* p = (const uint32_t*)buf;
    const uint64_t lines = (uint64_t)n_bytes / 64;
    for (int64_t i = tid; i < (int64_t)lines; i += nthr) {
      uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
      h ^= h >> 29;
      acc.x ^= p[16 * ((h * 0xBF58476D1CE4E5B9ull >> 11) % lines)];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = 1;  // (keeps the loads alive)
}

extern "C" int gf_calib_stream(const void* d_buf, int64_t n_bytes, int mode, void* d_sink, void* stream) {
  const dim3 grid(256 * 16), blk(256);
  hipStream_t st = (hipStream_t)stream;
  const uint8_t* b = (const uint8_t*)d_buf;
  uint32_t* k = (uint32_t*)d_sink;
  if (mode == 0) hipLaunchKernelGGL(gf_k_calib_stream<0>, grid, blk, 0, st, b, n_bytes, k);
  else if (mode == 1) hipLaunchKernelGGL(gf_k_calib_stream<1>, grid, blk, 0, st, b, n_bytes, k);
  else if (mode == 2) hipLaunchKernelGGL(gf_k_calib_stream<2>, grid, blk, 0, st, b, n_bytes, k);
  else hipLaunchKernelGGL(gf_k_calib_stream<3>, grid, blk, 0, st, b, n_bytes, k);
  return (int)hipGetLastError();
}
