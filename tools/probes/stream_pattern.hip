// Measurement harness (not part of the product): what read bandwidth does the PANEL PIPELINE's access pattern get?
// 256 persistent workgroups of 16 waves; every wave owns contiguous chunks of `chunk_kb` KiB (chunks dealt like kernel X's: interleaved over
// the workgroups, or contiguous per workgroup) and reads them `burst_kb` KiB at a time with `depth` bursts in flight, through buffer
// descriptors with the nt policy or plain global loads.  Prints GB/s.  Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/probes/libstream_pattern.so tools/probes/stream_pattern.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int BURST /* 16-byte loads per lane per burst: 1 = 1 KiB per wave */, int DEPTH, bool NT>
__global__ __launch_bounds__(1024, 1) void k_stream(const uint8_t* __restrict__ base, uint64_t bytes, uint32_t chunk_bytes, int contiguous, unsigned long long* sink) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t nchunks = bytes / chunk_bytes;
  const uint32_t nwaves = gridDim.x * 16, wid = contiguous ? blockIdx.x * 16 + wv : wv * gridDim.x + blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)0x7FFFFFF0, 0x00020000);
  v4u acc = {0, 0, 0, 0};
  const uint32_t bursts_per_chunk = chunk_bytes / (1024 * BURST);
  // the wave's sequence of bursts: chunk c = wid, wid + nwaves, ...; inside a chunk burst b = 0 .. bursts_per_chunk - 1
  uint64_t c = wid; uint32_t b = 0;
  v4u st[DEPTH][BURST];
  auto issue = [&](v4u (&dst)[BURST]) {
    const bool ok = c < nchunks;
    const uint64_t off = ok ? c * chunk_bytes + (uint64_t)b * 1024 * BURST + lane * 16u * BURST : 0;
#pragma unroll
    for (int j = 0; j < BURST; j++) {
      if (NT) dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? (uint32_t)(off + 16 * j) : 0xFFFFFFFFu), 0, 2);
      else dst[j] = ok ? *(const v4u*)(base + off + 16 * j) : v4u{0, 0, 0, 0};
    }
    if (++b == bursts_per_chunk) { b = 0; c += nwaves; }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; d++) issue(st[d]);
  for (;;) {
    bool more = false;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
      for (int j = 0; j < BURST; j++) acc += st[d][j];
      more = c < nchunks;
      issue(st[d]);
    }
    if (!more) break;
  }
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
#pragma unroll
    for (int j = 0; j < BURST; j++) acc += st[d][j];
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc.z + acc.w;
}

// the masked SpGEMM's pattern: a wave walks a "row" of `row_entries` 4-byte entries, 64 entries (256 B) per load instruction, INFL loads in
// flight per lane (plain global loads) — or 4 entries (16 B) per lane per instruction when VEC
template <int INFL, bool VEC>
__global__ __launch_bounds__(512, 2) void k_rows(const uint32_t* __restrict__ base, uint64_t entries, uint32_t row_entries, unsigned long long* sink) {
  const int lane = threadIdx.x & 63;
  const uint64_t nrowsT = entries / row_entries, nwaves = (uint64_t)gridDim.x * 8, wid = (uint64_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  uint32_t acc = 0;
  for (uint64_t r = wid; r < nrowsT; r += nwaves) {
    const uint32_t* row = base + r * row_entries;
    if (VEC) {
      for (uint32_t p = lane * 4; p < row_entries; p += 256 * INFL) {
        uint4 v[INFL];
#pragma unroll
        for (int u = 0; u < INFL; u++) { const uint32_t q = p + 256 * u; v[u] = q < row_entries ? *(const uint4*)(row + q) : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < INFL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
      }
    } else {
      for (uint32_t p = lane; p < row_entries; p += 64 * INFL) {
        uint32_t v[INFL];
#pragma unroll
        for (int u = 0; u < INFL; u++) { const uint32_t q = p + 64 * u; v[u] = row[q < row_entries ? q : row_entries - 1]; }
#pragma unroll
        for (int u = 0; u < INFL; u++) acc += v[u];
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int INFL, bool VEC> static float run_rows(const uint8_t* buf, uint64_t bytes, uint32_t row_entries, int reps, unsigned long long* sink) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_rows<INFL, VEC>), dim3(1024), dim3(512), 0, 0, (const uint32_t*)buf, bytes / 4, row_entries, sink);
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_rows<INFL, VEC>), dim3(1024), dim3(512), 0, 0, (const uint32_t*)buf, bytes / 4, row_entries, sink);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b); hipEventDestroy(a); hipEventDestroy(b);
  return ms / reps;
}
extern "C" int row_pattern_probe(uint64_t bytes) {
  uint8_t* buf; unsigned long long* sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  hipMemset(buf, 1, bytes); hipDeviceSynchronize();
  printf("%-12s %-6s %-5s %10s\n", "row entries", "infl", "vec", "GB/s");
  for (uint32_t re : {512u, 2048u, 8192u, 32768u}) {
#define RR(I, V) { const float ms = run_rows<I, V>(buf, bytes, re, 10, sink); printf("%-12u %-6d %-5d %10.0f\n", re, I, (int)V, bytes / (ms * 1e-3) / 1e9); }
    RR(4, false) RR(8, false) RR(16, false) RR(1, true) RR(2, true) RR(4, true)
#undef RR
  }
  fflush(stdout); hipFree(buf); hipFree(sink); return 0;
}

template <int BURST, int DEPTH, bool NT> static float run(const uint8_t* buf, uint64_t bytes, uint32_t chunk_bytes, int contiguous, int reps, unsigned long long* sink) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_stream<BURST, DEPTH, NT>), dim3(256), dim3(1024), 0, 0, buf, bytes, chunk_bytes, contiguous, sink);
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream<BURST, DEPTH, NT>), dim3(256), dim3(1024), 0, 0, buf, bytes, chunk_bytes, contiguous, sink);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b); hipEventDestroy(a); hipEventDestroy(b);
  return ms / reps;
}

extern "C" int stream_pattern_probe(uint64_t bytes) {
  uint8_t* buf; unsigned long long* sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  hipMemset(buf, 1, bytes); hipDeviceSynchronize();
  printf("%-10s %-6s %-5s %-4s %-12s %10s\n", "chunk KiB", "burst", "depth", "nt", "deal", "GB/s");
  for (uint32_t chunk_kb : {4u, 16u, 64u, 1024u}) {
    for (int contiguous = 0; contiguous < 2; contiguous++) {
#define RUN(BU, DE, NT_) if ((chunk_kb * 1024u) % (1024u * BU) == 0) { const float ms = run<BU, DE, NT_>(buf, bytes, chunk_kb * 1024u, contiguous, 10, sink); \
      printf("%-10u %-6d %-5d %-4d %-12s %10.0f\n", chunk_kb, BU, DE, (int)NT_, contiguous ? "contiguous" : "interleaved", bytes / (ms * 1e-3) / 1e9); }
      RUN(1, 2, true) RUN(1, 4, true) RUN(1, 8, true) RUN(2, 2, true) RUN(2, 4, true) RUN(4, 2, true) RUN(4, 4, true) RUN(1, 4, false) RUN(4, 2, false)
#undef RUN
    }
  }
  fflush(stdout);
  hipFree(buf); hipFree(sink); return 0;
}
