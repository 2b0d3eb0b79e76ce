// Measurement harness (not part of the product), round 6: what does the INSTRUCTION MIX of a lane-per-piece step cost?
// 256 persistent workgroups of 16 waves; a wave walks its share of a stream of "steps" (64 column words + 64 eight-byte values) with DEPTH
// macro-steps in flight.  G = steps per macro-step: 1 -> a 4-byte and an 8-byte load per lane and step (grb_spmv_sell.hpp as first built),
// 4 -> one 16-byte and two 16-byte loads per lane for four steps.  GATHER: one gather per step from a 32 MiB vector, `cold_pct` % of the lanes
// in range (random 8-byte reads inside the workgroup's eighth of the vector), the others out of range (no memory request).  PID: one more
// load per step whose lanes are all out of range.  Prints GB/s of stream bytes.
// Build + run: hipcc -O3 --offload-arch=gfx950 -o /tmp/step_pattern tools/probes/step_pattern.hip && /tmp/step_pattern
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

template <int G, int DEPTH, bool GATHER, bool PID, bool VALS, int GAUX = 0>
__global__ __launch_bounds__(1024, 1) void k_steps(const uint32_t* __restrict__ cols, const uint8_t* __restrict__ vals, const double* __restrict__ u, uint32_t ulen, uint32_t nmacro, uint32_t cold_thresh,
                                                   unsigned long long* sink) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t nwaves = gridDim.x * 16, wid = wv * gridDim.x + blockIdx.x;
  const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)cols, (short)0, (int)(nmacro * G * 256u), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)vals, (short)0, (int)(nmacro * G * 512u), 0x00020000);
  const __amdgpu_buffer_rsrc_t u_rs = __builtin_amdgcn_make_buffer_rsrc((void*)u, (short)0, (int)(ulen * 8u), 0x00020000);
  // items of 32 macro-steps dealt round-robin to the waves
  const uint32_t ITEM = 32;
  uint32_t item = wid, k = 0;
  struct St { uint32_t c[G]; v2u v[G]; v2u g[G]; uint32_t p[G]; };
  St st[DEPTH];
  uint32_t acc = 0;
  auto issue = [&](St& s) {
    const uint32_t m = item * ITEM + k; const bool ok = m < nmacro;
    if (G == 1) {
      s.c[0] = __builtin_amdgcn_raw_buffer_load_b32(c_rs, (int)(ok ? m * 256u + lane * 4u : 0xFFFFFFFFu), 0, 2);
      if (VALS) s.v[0] = __builtin_amdgcn_raw_buffer_load_b64(v_rs, (int)(ok ? m * 512u + lane * 8u : 0xFFFFFFFFu), 0, 2);
    } else {
      const v4u c = __builtin_amdgcn_raw_buffer_load_b128(c_rs, (int)(ok ? m * 1024u + lane * 16u : 0xFFFFFFFFu), 0, 2);
      s.c[0] = c.x; s.c[1 % G] = c.y; s.c[2 % G] = c.z; s.c[3 % G] = c.w;
      if (VALS) {
        const v4u a = __builtin_amdgcn_raw_buffer_load_b128(v_rs, (int)(ok ? m * 2048u + lane * 32u : 0xFFFFFFFFu), 0, 2);
        const v4u b = __builtin_amdgcn_raw_buffer_load_b128(v_rs, (int)(ok ? m * 2048u + lane * 32u + 16u : 0xFFFFFFFFu), 0, 2);
        s.v[0] = v2u{a.x, a.y}; s.v[1 % G] = v2u{a.z, a.w}; s.v[2 % G] = v2u{b.x, b.y}; s.v[3 % G] = v2u{b.z, b.w};
      }
    }
    if (++k == ITEM) { k = 0; item += nwaves; }
  };
  auto second = [&](St& s) {       // what depends on the column words: gathers, the partial-id load
#pragma unroll
    for (int j = 0; j < G; j++) {
      const uint32_t w = s.c[j];
      if (GATHER && (GAUX == 300 || GAUX == 301)) {       // the opposite: a step's cold lanes split over TWO (300) or FOUR (301) gather instructions by lane number — fewer active lanes per instruction
        const bool cold = (w & 0xFFFFu) < cold_thresh; const uint32_t off = (w >> 3) << 3;
        const int parts = GAUX == 300 ? 2 : 4;
        v2u acc2 = v2u{0, 0};
#pragma unroll
        for (int q = 0; q < parts; q++) { const v2u t = __builtin_amdgcn_raw_buffer_load_b64(u_rs, (int)((cold && (lane % parts) == q) ? off : 0xFFFFFFFFu), 0, 0); acc2.x |= t.x; acc2.y |= t.y; }
        s.g[j] = acc2;
      } else
      if (GATHER && GAUX == 200) {       // what compacting the cold lanes of G steps into ONE gather instruction would buy (upper bound: no shuffle cost): step 0 gathers with G x the probability, the others issue nothing
        if (j == 0) s.g[j] = __builtin_amdgcn_raw_buffer_load_b64(u_rs, (int)((w & 0xFFFFu) < cold_thresh * (uint32_t)G ? (w >> 3) << 3 : 0xFFFFFFFFu), 0, 0); else s.g[j] = v2u{0, 0};
      } else
      if (GATHER && GAUX != 100) s.g[j] = __builtin_amdgcn_raw_buffer_load_b64(u_rs, (int)((w & 0xFFFFu) < cold_thresh ? (w >> 3) << 3 : 0xFFFFFFFFu), 0, GAUX);
      if (GATHER && GAUX == 100) {
        // the cold lanes one by one through the SCALAR data cache: readlane -> s_load_dwordx2 -> writelane, eight loads in flight
        typedef __attribute__((address_space(4))) const unsigned long long* kptr;
        const kptr uk = (kptr)(uintptr_t)u;
        const bool cold = (w & 0xFFFFu) < cold_thresh;
        const uint32_t off = (w >> 3);                  // index of the 8-byte value
        unsigned long long m = __ballot(cold);
        uint32_t glo = 0, ghi = 0;
        while (m) {
          int l[8]; bool ok[8]; unsigned long long x[8];
#pragma unroll
          for (int q = 0; q < 8; q++) { ok[q] = m != 0; l[q] = ok[q] ? __builtin_ctzll(m) : 0; m &= m - 1; }
#pragma unroll
          for (int q = 0; q < 8; q++) { const uint32_t o = ok[q] ? (uint32_t)__builtin_amdgcn_readlane((int)off, l[q]) : 0u; x[q] = uk[o]; }
#pragma unroll
          for (int q = 0; q < 8; q++) if (ok[q]) { const uint32_t xl = (uint32_t)x[q], xh = (uint32_t)(x[q] >> 32);
            asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(glo), "+v"(ghi) : "s"(xl), "s"(xh), "s"(l[q]) : "m0"); }
        }
        s.g[j] = v2u{glo, ghi};
      }
      if (PID) s.p[j] = __builtin_amdgcn_raw_buffer_load_b32(c_rs, (int)((w == 0x12345u) ? lane * 4u : 0xFFFFFFFFu), 0, 0);
    }
  };
  auto consume = [&](St& s) {
#pragma unroll
    for (int j = 0; j < G; j++) { acc += s.c[j]; if (VALS) acc += s.v[j].x ^ s.v[j].y; if (GATHER) acc += s.g[j].x + s.g[j].y; if (PID) acc += s.p[j]; }
  };
  constexpr int H = DEPTH / 2;        // column words DEPTH ahead, dependants H ahead
#pragma unroll
  for (int d = 0; d < DEPTH; d++) issue(st[d]);
#pragma unroll
  for (int d = 0; d < H; d++) second(st[d]);
  const uint32_t rounds = (nmacro + nwaves * DEPTH - 1) / (nwaves * DEPTH) + 1;
  for (uint32_t r = 0; r < rounds; r++) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      second(st[(d + H) % DEPTH]);
      consume(st[d]);
      issue(st[d]);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int G, int DEPTH, bool GATHER, bool PID, bool VALS, int GAUX = 0>
static void run(const char* name, const uint32_t* cols, const uint8_t* vals, const double* u, uint32_t ulen, uint32_t nsteps, uint32_t cold_pct, unsigned long long* sink) {
  const uint32_t nmacro = nsteps / G, thresh = 65536u * cold_pct / 100u;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_steps<G, DEPTH, GATHER, PID, VALS, GAUX>), dim3(256), dim3(1024), 0, 0, cols, vals, u, ulen, nmacro, thresh, sink);
  hipEventRecord(e0, 0);
  const int reps = 10;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_steps<G, DEPTH, GATHER, PID, VALS, GAUX>), dim3(256), dim3(1024), 0, 0, cols, vals, u, ulen, nmacro, thresh, sink);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double bytes = (double)nmacro * G * (256.0 + (VALS ? 512.0 : 0.0));
  printf("%-44s G=%d depth=%2d gather=%d pid=%d vals=%d cold=%2u%%: %.3f ms  %.0f GB/s of stream bytes (%.1f M steps)\n", name, G, DEPTH, (int)GATHER, (int)PID, (int)VALS, cold_pct, ms, bytes / ms / 1e6, nmacro * G / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint32_t nsteps = 1u << 20;               // 64 M entries, as R-MAT-22
  const uint32_t ulen = 1u << 22;
  uint32_t* cols; uint8_t* vals; double* u; unsigned long long* sink;
  hipMalloc(&cols, (size_t)nsteps * 256 + 64); hipMalloc(&vals, (size_t)nsteps * 512 + 64); hipMalloc(&u, (size_t)ulen * 8); hipMalloc(&sink, 64);
  // column words: workgroup b & 7 reads inside its eighth of u: the word's low 16 bits decide "cold", the offset is anywhere in the eighth.
  // Macro-steps are dealt round-robin to waves whose workgroup is b: fill by position so that a workgroup's steps point into its own eighth.
  std::vector<uint32_t> h((size_t)nsteps * 64);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < h.size(); i++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const uint32_t item = (uint32_t)(i / (64u * 32u)), wid = item % 4096u, wg = wid % 256u, xcd = wg & 7u;     // (G = 1 layout; for G = 4 the mapping is approximate — the eighths still see ~1/8 each)
    const uint32_t within = (uint32_t)(x >> 20) % (ulen / 8u);
    const uint32_t off = (xcd * (ulen / 8u) + within) * 8u;                                               // byte offset, 8-aligned
    h[i] = (off & ~0xFFFFu) | (uint32_t)(x & 0xFFFFu);                                                     // low 16 bits random (cold test), offset quantised to 64 KiB + the low bits
  }
  hipMemcpy(cols, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(vals, 1, (size_t)nsteps * 512); hipMemset(u, 0, (size_t)ulen * 8);
  const uint32_t cp = argc > 1 ? (uint32_t)atoi(argv[1]) : 23;
  run<1, 8, false, false, true>("stream only, 4B + 8B per lane", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, false, false, true>("stream only, 4B + 8B per lane", cols, vals, u, ulen, nsteps, cp, sink);
  run<4, 4, false, false, true>("stream only, 16B + 2x16B per lane", cols, vals, u, ulen, nsteps, cp, sink);
  run<4, 8, false, false, true>("stream only, 16B + 2x16B per lane", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, false, true, true>("+ an all-out-of-range load per step", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true>("+ gather per step", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, true, true>("+ gather + out-of-range load per step", cols, vals, u, ulen, nsteps, cp, sink);
  run<4, 4, true, false, true>("16B loads + 4 gathers per macro-step", cols, vals, u, ulen, nsteps, cp, sink);
  run<4, 8, true, false, true>("16B loads + 4 gathers per macro-step", cols, vals, u, ulen, nsteps, cp, sink);
  run<4, 8, true, true, true>("16B loads + 4 gathers + 4 oor per macro-step", cols, vals, u, ulen, nsteps, cp, sink);
  // round 6, second look: the cache-policy bits of the GATHER instruction (aux: 1 = sc0, 2 = nt, 16 = sc1) — does any of them make a cold gather cheaper than a 128-byte L2->L1 line?
  run<1, 16, true, false, true, 1>("+ gather per step, sc0", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true, 2>("+ gather per step, nt", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true, 3>("+ gather per step, sc0 nt", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true, 16>("+ gather per step, sc1", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true, 17>("+ gather per step, sc0 sc1", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true, 18>("+ gather per step, sc1 nt", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true, 19>("+ gather per step, sc0 sc1 nt", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true, 300>("+ gather per step in two half-lane instructions", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true, 301>("+ gather per step in four quarter-lane instructions", cols, vals, u, ulen, nsteps, cp, sink);
  run<4, 4, true, false, true, 200>("16B loads + ONE gather for 4 steps' cold lanes", cols, vals, u, ulen, nsteps, cp, sink);
  run<4, 8, true, false, true, 200>("16B loads + ONE gather for 4 steps' cold lanes", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, true, 100>("+ cold lanes through the scalar cache", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 8, true, false, true, 100>("+ cold lanes through the scalar cache", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, false, 100>("column words + cold lanes through the scalar cache", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, false, false, false>("column words only, 4B per lane", cols, vals, u, ulen, nsteps, cp, sink);
  run<4, 8, false, false, false>("column words only, 16B per lane", cols, vals, u, ulen, nsteps, cp, sink);
  run<1, 16, true, false, false>("column words 4B + gather", cols, vals, u, ulen, nsteps, cp, sink);
  run<4, 8, true, false, false>("column words 16B + 4 gathers", cols, vals, u, ulen, nsteps, cp, sink);
  return 0;
}
