// Measurement harness (not part of the product): what do rocprofv3's memory-side counters report for access patterns whose byte
// count is known?  Four kernels over a 1 GiB buffer (far beyond the 32 MiB of L2; beyond the 256 MiB Infinity Cache too):
//   k_read16    16 bytes per lane, coalesced (1 KiB per wave instruction)            reads exactly `bytes`
//   k_read4     4 bytes per lane, coalesced (256 B per wave instruction), 8 in flight  reads exactly `bytes`
//   k_rows4     like the masked SpGEMM's B-row streams: "rows" of 64..8192 words at arbitrary 4-byte offsets, a wave per row
//   k_gather8   one random 8-byte gather per lane (like the SpMV's cold gathers)        useful bytes = 8 per gather
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/fetch_calib tools/probes/fetch_calib.hip
// Run under: rocprofv3 --pmc FETCH_SIZE -- tools/probes/fetch_calib    (and TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum ... in other passes)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read16(const v4u* __restrict__ p, uint64_t n16, unsigned long long* sink) {
  v4u acc = {0, 0, 0, 0};
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) { const v4u v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) atomicAdd(sink, 1ull);
}
__global__ __launch_bounds__(256) void k_read4(const uint32_t* __restrict__ p, uint64_t n4, unsigned long long* sink) {
  uint32_t acc = 0;
  const uint64_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6, nwaves = gridDim.x * 4ull; const uint32_t lane = threadIdx.x & 63;
  for (uint64_t b = wave * 512; b + 512 <= n4; b += nwaves * 512) {
    uint32_t j[8];
#pragma unroll
    for (int u = 0; u < 8; u++) j[u] = p[b + 64 * u + lane];
#pragma unroll
    for (int u = 0; u < 8; u++) acc ^= j[u];
  }
  if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}
__global__ __launch_bounds__(256) void k_rows4(const uint32_t* __restrict__ p, const uint32_t* __restrict__ rb, const uint32_t* __restrict__ rl, uint32_t nrows, unsigned long long* sink) {
  uint32_t acc = 0;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u, lane = threadIdx.x & 63;
  for (uint32_t r = wave; r < nrows; r += nwaves) {
    const uint32_t bb = rb[r], be = bb + rl[r];
    uint32_t pb0 = bb + lane;
    for (; pb0 + 448 < be; pb0 += 512) {
      uint32_t j[8];
#pragma unroll
      for (int u = 0; u < 8; u++) j[u] = p[pb0 + 64 * u];
#pragma unroll
      for (int u = 0; u < 8; u++) acc ^= j[u];
    }
    for (; pb0 < be; pb0 += 64) acc ^= p[pb0];
  }
  if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}
// the same rows with 16-byte loads per lane at the rows' own (4-byte) alignment: 1 KiB per wave instruction, DEPTH in flight
struct __attribute__((packed, aligned(4))) u4x { uint32_t x, y, z, w; };
template <int DEPTH, int LDSKB>
__global__ __launch_bounds__(256) void k_rows16(const uint32_t* __restrict__ p, const uint32_t* __restrict__ rb, const uint32_t* __restrict__ rl, uint32_t nrows, unsigned long long* sink) {
  __shared__ uint32_t pad[LDSKB * 256]; if (threadIdx.x == 0) pad[0] = 0;       // LDS footprint of the SpGEMM kernels (limits the waves per CU)
  uint32_t acc = 0;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u, lane = threadIdx.x & 63;
  for (uint32_t r = wave; r < nrows; r += nwaves) {
    const uint32_t bb = rb[r], be = bb + rl[r];
    uint32_t base = bb;
    for (; base + 256 * DEPTH <= be; base += 256 * DEPTH) {
      u4x j[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH; u++) j[u] = *(const u4x*)(p + base + 256 * u + 4 * lane);
#pragma unroll
      for (int u = 0; u < DEPTH; u++) acc ^= j[u].x ^ j[u].y ^ j[u].z ^ j[u].w;
    }
    for (uint32_t q = base + lane; q < be; q += 64) acc ^= p[q];
  }
  if (acc == 0x12345678u) atomicAdd(sink, 1ull);
  if (pad[0] == 77) atomicAdd(sink, 1ull);
}
template <int DEPTH, int LDSKB>
__global__ __launch_bounds__(256) void k_rows4d(const uint32_t* __restrict__ p, const uint32_t* __restrict__ rb, const uint32_t* __restrict__ rl, uint32_t nrows, unsigned long long* sink) {
  __shared__ uint32_t pad[LDSKB * 256]; if (threadIdx.x == 0) pad[0] = 0;
  uint32_t acc = 0;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u, lane = threadIdx.x & 63;
  for (uint32_t r = wave; r < nrows; r += nwaves) {
    const uint32_t bb = rb[r], be = bb + rl[r];
    uint32_t pb0 = bb + lane;
    for (; pb0 - lane + 64 * DEPTH <= be; pb0 += 64 * DEPTH) {
      uint32_t j[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH; u++) j[u] = p[pb0 + 64 * u];
#pragma unroll
      for (int u = 0; u < DEPTH; u++) acc ^= j[u];
    }
    for (; pb0 < be; pb0 += 64) acc ^= p[pb0];
  }
  if (acc == 0x12345678u) atomicAdd(sink, 1ull);
  if (pad[0] == 77) atomicAdd(sink, 1ull);
}
__global__ __launch_bounds__(256) void k_gather8(const unsigned long long* __restrict__ p, uint64_t n8, uint32_t per_lane, unsigned long long* sink) {
  unsigned long long acc = 0; uint64_t s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  for (uint32_t it = 0; it < per_lane; it += 4) {
    unsigned long long v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { s = s * 6364136223846793005ull + 1442695040888963407ull; v[u] = p[(s >> 20) % n8]; }
#pragma unroll
    for (int u = 0; u < 4; u++) acc ^= v[u];
  }
  if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

int main(int argc, char** argv) {
  const uint64_t bytes = 1ull << 30;
  void* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
  unsigned long long* sink; CK(hipMalloc(&sink, 8)); CK(hipMemset(sink, 0, 8));
  // rows: lengths 64..8192 words (log-uniform), back to back with a random 1..15-word gap so the starts are not line-aligned
  const uint32_t maxrows = 1u << 20; uint32_t* hb = (uint32_t*)malloc(maxrows * 4); uint32_t* hl = (uint32_t*)malloc(maxrows * 4);
  uint64_t pos = 0, words = 0; uint32_t nrows = 0; uint64_t s = 777;
  while (nrows < maxrows) {
    s = s * 6364136223846793005ull + 1442695040888963407ull; const uint32_t e = 6 + (uint32_t)((s >> 33) % 8); 
    s = s * 6364136223846793005ull + 1442695040888963407ull; const uint32_t len = (1u << e) + (uint32_t)((s >> 33) % (1u << e));
    s = s * 6364136223846793005ull + 1442695040888963407ull; const uint32_t gap = 1 + (uint32_t)((s >> 33) % 15);
    if ((pos + len + gap) * 4 > bytes) break;
    hb[nrows] = (uint32_t)pos; hl[nrows] = len; nrows++; pos += len + gap; words += len;
  }
  uint32_t *db, *dl; CK(hipMalloc(&db, nrows * 4)); CK(hipMalloc(&dl, nrows * 4));
  CK(hipMemcpy(db, hb, nrows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, hl, nrows * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
  const uint32_t per_lane = 64; const uint64_t ngather = 4096ull * 256 * per_lane;
  for (int rep = 0; rep < 2; rep++) {      // (second round: the numbers printed; under --pmc every launch is listed anyway)
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const v4u*)buf, bytes / 16, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("k_read16   bytes %.6g  %.3f ms  %.0f GB/s\n", (double)bytes, ms, bytes / ms / 1e6);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("k_read4    bytes %.6g  %.3f ms  %.0f GB/s\n", (double)bytes, ms, bytes / ms / 1e6);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_rows4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)buf, db, dl, nrows, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("k_rows4    bytes %.6g (%u rows)  %.3f ms  %.0f GB/s\n", (double)words * 4, nrows, ms, words * 4 / ms / 1e6);
#define ROWS(K, NAME, GRID) CK(hipEventRecord(e0)); hipLaunchKernelGGL(K, dim3(GRID), dim3(256), 0, 0, (const uint32_t*)buf, db, dl, nrows, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
    if (rep) printf("%-28s grid %5d  %.3f ms  %.0f GB/s\n", NAME, GRID, ms, words * 4 / ms / 1e6);
    for (int grid : {1024, 2048, 4096}) {
      ROWS((k_rows4d<4, 1>), "rows 4B x4 deep, 1K LDS", grid) ROWS((k_rows4d<8, 1>), "rows 4B x8 deep, 1K LDS", grid) ROWS((k_rows4d<16, 1>), "rows 4B x16 deep, 1K LDS", grid)
      ROWS((k_rows4d<8, 20>), "rows 4B x8 deep, 20K LDS", grid) ROWS((k_rows4d<8, 40>), "rows 4B x8 deep, 40K LDS", grid)
      ROWS((k_rows16<2, 1>), "rows 16B x2 deep, 1K LDS", grid) ROWS((k_rows16<4, 1>), "rows 16B x4 deep, 1K LDS", grid) ROWS((k_rows16<8, 1>), "rows 16B x8 deep, 1K LDS", grid)
      ROWS((k_rows16<4, 20>), "rows 16B x4 deep, 20K LDS", grid) ROWS((k_rows16<4, 40>), "rows 16B x4 deep, 40K LDS", grid)
    }
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_gather8, dim3(4096), dim3(256), 0, 0, (const unsigned long long*)buf, bytes / 8, per_lane, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("k_gather8  gathers %.6g = %.6g useful bytes, %.6g bytes of 64-byte sectors, %.6g of 128-byte lines  %.3f ms\n", (double)ngather, ngather * 8.0, ngather * 64.0, ngather * 128.0, ms);
  }
  return 0;
}
