// Probe (gfx950): what does ONE CU sustain when it streams operands from L2 / Infinity Cache / HBM, by load path, waves per CU and bytes in flight?
// the "operand-delivery roofline" of DESIGN 3.5 (~43 KB/us per CU) had only been measured by the GEMM kernels themselves.
// This is a standalone stream: one workgroup per CU (LDS > 80 KiB), every wave keeps D pieces of 1 KiB (64 lanes x 16 B) in flight behind a counted
// s_waitcnt vmcnt(D-1), nothing else in the loop - optionally 4 or 8 v_mfma_f32_32x32x16_bf16 per piece on random operands (config 11's ratio is 4).
//   mode 0  buffer_load_dwordx4 ... offen lds      (LDS-DMA, buffer addressing)
//   mode 1  global_load_lds_dwordx4                 (LDS-DMA, flat addressing - what sf_gemm_pp.hip issues)
//   mode 2  global_load_dwordx4 -> VGPR             (consumed by one v_xor per dword)
//   mode 3  buffer_load_dwordx4 -> VGPR
//   residency 0: 1 MiB per XCD, shared by the XCD's 32 workgroups (L2 hits) | 1: 128 MiB, 512 KiB per workgroup (thrashes the 4 MiB L2s, fits the 256 MiB
//   Infinity Cache) | 2: 2 GiB, 8 MiB per workgroup (HBM)
//   shape 0: a piece = 1 KiB contiguous | 1: 8 rows x 128 B, rows 1536 B apart (an A operand piece of K = 768) | 2: 16 rows x 64 B, rows 1536 B apart
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/l2_stream.hip -o tools/probe/l2_stream.bin ; run: tools/probe/l2_stream.bin [csv|quick] [stag]   ("stag": only the MFMA-only baseline and
// the staggered arrangement - phases of 2 pieces | barrier | 8 MFMAs | barrier, a SIMD's second wave one barrier behind: config 11 without its fragment reads)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  const char* base;
  uint64_t xcd_stride, blk_stride;   // byte offsets of a workgroup's stream: (blk % 8) * xcd_stride + (blk / 8) * blk_stride
  uint32_t span_mask;                // the stream wraps inside span = span_mask + 1 bytes (power of two)
  uint32_t pieces;                   // pieces per wave
  uint32_t shape;
  uint64_t* out;                     // per workgroup: shader cycles, 100 MHz ticks
  float* sink;
};

template <int N> __device__ __forceinline__ void wait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}

__device__ __forceinline__ uint32_t lane_off(uint32_t shape, uint32_t q, uint32_t lane) {
  // byte offset of this lane's 16 bytes of piece q of the workgroup's stream (before wrapping)
  if (shape == 0) return q * 1024u + lane * 16u;
  if (shape == 1) {   // pieces walk 12 k-slabs of 128 B along a block of 8 rows (pitch 1536 B), then the next 8 rows
    const uint32_t slab = q % 12u, rb = q / 12u;
    return (rb * 8u + (lane >> 3)) * 1536u + slab * 128u + (lane & 7u) * 16u;
  }
  const uint32_t slab = q % 24u, rb = q / 24u;   // 16 rows x 64 B
  return (rb * 16u + (lane >> 2)) * 1536u + slab * 64u + (lane & 3u) * 16u;
}

template <int MODE, int D, int MFMA, int STAG = 0>
__global__ __launch_bounds__((MFMA || D >= 8) ? 512 : 1024) void stream_kernel(Args a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const uint64_t wg_off = (uint64_t)(blockIdx.x & 7u) * a.xcd_stride + (uint64_t)(blockIdx.x >> 3) * a.blk_stride;
  const char* wg_base = a.base + wg_off;
  const uint32_t start = ((blockIdx.x >> 3) * 32768u) & a.span_mask;   // the workgroups of an XCD start at different places of a shared span
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wg_base), (short)0, (int)(a.span_mask + 1u), 0x00020000);
  const uint32_t lds_wave = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + wave * (uint32_t)(D * 1024);

  bf16x8 fa[2], fb[2];
  f32x16 acc[4];
  if (MFMA) {
    uint32_t s = 0x9e3779b9u * (threadIdx.x + 1u) + blockIdx.x;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u; fa[i][j] = (__bf16)(((int)(s >> 20) - 2048) * (1.0f / 2048.0f));
        s = s * 1664525u + 1013904223u; fb[i][j] = (__bf16)(((int)(s >> 20) - 2048) * (1.0f / 2048.0f));
      }
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  }
  u32x4 r[(MODE == 2 || MODE == 3) ? D : 1];
  uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;

  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  const uint32_t total = a.pieces;
  // prologue: D - 1 pieces in flight
  auto issue = [&](uint32_t p, int slot) {
    const uint32_t q = p * nw + wave;
    const uint32_t off = (start + lane_off(a.shape, q, lane)) & a.span_mask;
    if (MODE == 0) {
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(off), "s"(rsrc), "s"(lds_wave + (uint32_t)slot * 1024u) : "memory");
    } else if (MODE == 1) {
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(off), "s"(wg_base), "s"(lds_wave + (uint32_t)slot * 1024u) : "memory");
    } else if (MODE == 4) {
      (void)off;                                  // MFMA-only baseline: no memory instruction at all
    } else if (MODE == 2) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[slot]) : "v"(off), "s"(wg_base) : "memory");
    } else {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[slot]) : "v"(off), "s"(rsrc) : "memory");
    }
  };
  auto retire = [&](int slot) {   // the oldest piece has landed: VGPR modes consume it
    if (MODE == 2 || MODE == 3) {
      asm volatile("" : "+v"(r[slot]));   // ties the use to the position behind the counted wait
      x0 ^= r[slot].x; x1 ^= r[slot].y; x2 ^= r[slot].z; x3 ^= r[slot].w;
    }
  };
  auto mfmas = [&]() {
#pragma unroll
    for (int m = 0; m < MFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m & 1], fb[(m >> 1) & 1], acc[m & 3], 0, 0, 0);
  };
#pragma unroll
  for (int d = 0; d < D - 1 - STAG; ++d) issue((uint32_t)d, d);     // prologue: D - 1 pieces in flight (STAG: D - 2, it issues two per phase)
  uint32_t p = D - 1 - STAG;
  if constexpr (STAG != 0) {
    // config 11's arrangement without its LDS fragment reads: a phase = { 2 pieces + counted wait | s_barrier | 2 x MFMA matrix ops | s_barrier }, the waves of the
    // second half (one per SIMD) run ONE barrier behind the first half: a SIMD's two waves alternate between the matrix segment and the issue segment
    const bool late = wave >= (nw >> 1);
    if (late) __builtin_amdgcn_s_barrier();
    for (; p + D <= total; p += D) {
#pragma unroll
      for (int d = 0; d < D; d += 2) {
        issue(p + d, (d + D - 2) % D);
        issue(p + d + 1, (d + D - 1) % D);
        wait_vm<D - 2>();
        retire(d);
        retire(d + 1);
        __builtin_amdgcn_s_barrier();
        mfmas(); mfmas();
        __builtin_amdgcn_s_barrier();
      }
    }
    if (!late) __builtin_amdgcn_s_barrier();
  } else
  for (; p + D <= total; p += D) {       // D pieces per trip: slots are compile-time
#pragma unroll
    for (int d = 0; d < D; ++d) {
      issue(p + d, (d + D - 1) % D);
      wait_vm<D - 1>();
      retire(d);
      if (MFMA) mfmas();
    }
  }
  wait_vm<0>();
  if (MODE == 2 || MODE == 3) {
#pragma unroll
    for (int d = 0; d < D; ++d) asm volatile("" : "+v"(r[d]));
  }
  __syncthreads();
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  if (threadIdx.x == 0) { a.out[blockIdx.x * 2] = t1 - t0; a.out[blockIdx.x * 2 + 1] = w1 - w0; }
  float s = __uint_as_float((x0 ^ x1 ^ x2 ^ x3) & 0x007fffffu);
  if (MFMA) for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
  if (MODE < 2) s += (float)lds[threadIdx.x * 16];
  if (s == 123.456f) a.sink[0] = s;
}

typedef void (*kern_t)(Args);
template <int MODE, int MFMA> static kern_t pick_d(int d) {
  switch (d) {
    case 1: return stream_kernel<MODE, 1, MFMA>;   case 2: return stream_kernel<MODE, 2, MFMA>;   case 3: return stream_kernel<MODE, 3, MFMA>;
    case 4: return stream_kernel<MODE, 4, MFMA>;   case 6: return stream_kernel<MODE, 6, MFMA>;   case 8: return stream_kernel<MODE, 8, MFMA>;
    case 12: return stream_kernel<MODE, 12, MFMA>; case 16: return stream_kernel<MODE, 16, MFMA>; case 24: return stream_kernel<MODE, 24, MFMA>;
  }
  return nullptr;
}
template <int MFMA> static kern_t pick_m(int mode, int d) {
  switch (mode) { case 0: return pick_d<0, MFMA>(d); case 1: return pick_d<1, MFMA>(d); case 2: return pick_d<2, MFMA>(d); case 3: return pick_d<3, MFMA>(d); }
  return nullptr;
}
static kern_t pick_stag(int mode, int d) {
  switch (mode * 100 + d) {
    case 2: return stream_kernel<0, 2, 4, 1>;    case 4: return stream_kernel<0, 4, 4, 1>;    case 8: return stream_kernel<0, 8, 4, 1>;    case 12: return stream_kernel<0, 12, 4, 1>;
    case 102: return stream_kernel<1, 2, 4, 1>;  case 104: return stream_kernel<1, 4, 4, 1>;  case 108: return stream_kernel<1, 8, 4, 1>;  case 112: return stream_kernel<1, 12, 4, 1>;
    case 202: return stream_kernel<2, 2, 4, 1>;  case 204: return stream_kernel<2, 4, 4, 1>;  case 208: return stream_kernel<2, 8, 4, 1>;  case 212: return stream_kernel<2, 12, 4, 1>;
    case 402: return stream_kernel<4, 2, 4, 1>;  case 404: return stream_kernel<4, 4, 4, 1>;
  }
  return nullptr;
}
static kern_t pick(int mode, int d, int mfma) { return mfma == 0 ? pick_m<0>(mode, d) : mfma == 4 ? pick_m<4>(mode, d) : pick_m<8>(mode, d); }

int main(int argc, char** argv) {
  const bool csv = argc > 1 && !strcmp(argv[1], "csv");
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const size_t pool = (size_t)2 << 30;
  char* buf; CK(hipMalloc(&buf, pool + (1 << 20)));
  {   // random bytes: the switching power of the data path is part of the measurement
    std::vector<uint32_t> h(1 << 22);
    uint32_t s = 12345u; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    for (size_t o = 0; o < pool; o += h.size() * 4) CK(hipMemcpy(buf + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  uint64_t* out; CK(hipMalloc(&out, ncu * 16)); float* sink; CK(hipMalloc(&sink, 4));
  std::vector<uint64_t> hout(ncu * 2);
  const char* mode_name[5] = {"buffer_load_x4_lds", "global_load_lds_x4", "global_load_x4_vgpr", "buffer_load_x4_vgpr", "no_load"};
  const char* res_name[3] = {"L2", "MALL", "HBM"};
  const char* shape_name[3] = {"1KiB", "8x128B", "16x64B"};
  const bool only_stag = argc > 2 && !strcmp(argv[2], "stag");
  printf("%s\n", csv ? "mode,residency,shape,waves,inflight_KiB,mfma_per_piece,us,KB_per_us_per_CU,B_per_clk_per_CU,TB_per_s_chip,GHz,arrangement"
                     : "# mode residency shape waves inflight_KiB mfma/piece | us  KB/us/CU  B/clk/CU  TB/s(chip)  GHz  arrangement");
  auto run_one = [&](kern_t k, int mode, int res, int shape, int waves, int d, int mfma, const char* arrangement) {
    Args a;
    a.base = buf; a.out = out; a.sink = sink; a.shape = (uint32_t)shape;
    if (res == 0) { a.xcd_stride = 1 << 20; a.blk_stride = 0; a.span_mask = (1u << 20) - 1; }
    else if (res == 1) { a.xcd_stride = (uint64_t)(512 << 10) * (ncu / 8); a.blk_stride = 512 << 10; a.span_mask = (512u << 10) - 1; }
    else { a.xcd_stride = (uint64_t)(8 << 20) * (ncu / 8); a.blk_stride = 8 << 20; a.span_mask = (8u << 20) - 1; }
    const uint32_t kib_per_cu = (res == 2 ? 8u : 16u) << 10;   // 8 or 16 MiB per CU and run
    a.pieces = (kib_per_cu / waves / d) * d + (d - 1);
    const size_t lds = std::max<size_t>(96 << 10, (size_t)waves * d * 1024);
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double best_us = 1e30, best_cyc = 0;
    for (int rep = 0; rep < 4; ++rep) {
      hipLaunchKernelGGL(k, dim3(ncu), dim3(waves * 64), lds, 0, a);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hout.data(), out, ncu * 16, hipMemcpyDeviceToHost));
      uint64_t cyc = 0, wall = 0;
      for (int b = 0; b < ncu; ++b) { cyc = std::max(cyc, hout[b * 2]); wall = std::max(wall, hout[b * 2 + 1]); }
      const double us = wall * 0.01;
      if (rep > 0 && us < best_us) { best_us = us; best_cyc = (double)cyc; }
    }
    const double bytes_cu = (double)(a.pieces - (d - 1)) * waves * 1024.0;   // (mode no_load: the bytes the same number of pieces WOULD have carried)
    const double kbus = bytes_cu / 1e3 / best_us, bclk = bytes_cu / best_cyc, tbs = bytes_cu * ncu / best_us / 1e6, ghz = best_cyc / best_us / 1e3;
    printf(csv ? "%s,%s,%s,%d,%d,%d,%.1f,%.1f,%.2f,%.2f,%.3f,%s\n" : "%-20s %-4s %-7s %2d %3d %d | %8.1f %7.1f %6.2f %6.2f %5.3f %s\n",
           mode_name[mode], res_name[res], shape_name[shape], waves, d * waves, mfma, best_us, kbus, bclk, tbs, ghz, arrangement);
    fflush(stdout);
  };
  // (1) free-running waves: every wave issues a piece, waits for its oldest, runs its MFMAs
  for (int mfma : {0, 4, 8})
    for (int res = 0; res < 3; ++res)
      for (int shape = 0; shape < 3; ++shape)
        for (int mode = 0; mode < 4; ++mode)
          for (int waves : {4, 8, 16})
            for (int kib : {8, 16, 32, 64, 96}) {
              if (only_stag) continue;
              if (quick && (shape != 0 || mfma == 8 || res == 1)) continue;
              if (shape != 0 && (res == 2 || mfma == 8)) continue;
              if (mfma && waves == 16) continue;     // 16 waves of an MFMA kernel do not exist in this repo (>= 128 accumulator registers)
              const int d = kib / waves;
              if (d * waves != kib || d < 1) continue;
              kern_t k = pick(mode, d, mfma);
              if (!k || (mfma && d == 24 && mode >= 2)) continue;   // (that instantiation spills)
              run_one(k, mode, res, shape, waves, d, mfma, "free");
            }
  // (2) the matrix pipe alone (no memory instruction): what 4 / 8 MFMAs per "piece" cost at the clock random operands allow - the ceiling of the rows with MFMAs
  for (int waves : {4, 8}) {
    run_one(stream_kernel<4, 4, 4, 0>, 4, 0, 0, waves, 4, 4, "free");
    run_one(stream_kernel<4, 4, 8, 0>, 4, 0, 0, waves, 4, 8, "free");
  }
  run_one(stream_kernel<4, 4, 4, 1>, 4, 0, 0, 8, 4, 4, "staggered");
  // (3) config 11's arrangement without its fragment reads: 8 waves, phases of 2 pieces | barrier | 8 MFMAs | barrier, the second wave of every SIMD one barrier behind
  for (int res = 0; res < 2; ++res)
    for (int shape = 0; shape < 2; ++shape)
      for (int mode = 0; mode < 3; ++mode)
        for (int d : {2, 4, 8, 12}) {
          kern_t k = pick_stag(mode, d);
          if (k) run_one(k, mode, res, shape, 8, d, 4, "staggered");
        }
  return 0;
}
