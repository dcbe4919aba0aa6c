// Probe (gfx950): what MFMA rate does the 1400 W package cap leave?  A register-only loop of back-to-back MFMAs (no LDS, no memory), with random
// or all-zero operands, timed with HIP events while tools/power_probe.sh samples rocm-smi.  The dense peaks of MI355X_MICROARCH.md are quoted
// at 2.4 GHz; under the cap the clock drops, and this is the number a GEMM on this box can at best approach.
//   mfma_power.bin <bf16|mx> <rand|zero> <waves per SIMD and block> <seconds> [blocks per CU]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <bool MX>
__global__ void __launch_bounds__(1024) spin(const int* __restrict__ src, float* __restrict__ sink, int iters, int prio_mode) {
  const int lane = threadIdx.x & 63;
  // prio_mode 1: waves 4.. of a block (the second wave of every SIMD, if waves go round-robin over the SIMDs) run at a lower priority; 2: odd waves do
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (prio_mode == 1) { if (wv & 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(3); }
  if (prio_mode == 2) { if (wv & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(3); }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  if constexpr (!MX) {
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = *reinterpret_cast<const bf16x8*>(src + (i * 64 + lane) * 4);
      b[i] = *reinterpret_cast<const bf16x8*>(src + ((i + 4) * 64 + lane) * 4);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + j) & 3], b[i], acc[j], 0, 0, 0);
    }
  } else {
    i32x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *reinterpret_cast<const i32x8*>(src + (i * 64 + lane) * 8);
      b[i] = *reinterpret_cast<const i32x8*>(src + ((i + 2) * 64 + lane) * 8);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(i + j) & 1], b[i], acc[j], 0, 0, 0, 127, 0, 127);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
  const bool mx = argc > 1 && !strcmp(argv[1], "mx");
  const bool zero = argc > 2 && !strcmp(argv[2], "zero");
  const int wps = argc > 3 ? atoi(argv[3]) : 2;
  const double secs = argc > 4 ? atof(argv[4]) : 4.0;
  const int wg_per_cu = argc > 5 ? atoi(argv[5]) : 1;
  const int prio_mode = argc > 6 ? atoi(argv[6]) : 0;   // (waves per SIMD = wps x wg_per_cu, as wg_per_cu blocks of 256 x wps threads)
  std::vector<int> h(8 * 64 * 8);
  srand(7);
  for (auto& v : h) {
    if (zero) { v = 0; continue; }
    if (mx) {   // four e4m3 bytes: sign random, exponent 5..8 (|x| in [1/4, 4)), mantissa random
      int w = 0;
      for (int q = 0; q < 4; ++q) w |= (((rand() & 1) << 7) | ((5 + (rand() & 3)) << 3) | (rand() & 7)) << (8 * q);
      v = w;
    } else {    // two bf16: sign random, |x| in [0.5, 2)
      int w = 0;
      for (int q = 0; q < 2; ++q) w |= (((rand() & 1) << 15) | ((126 + (rand() & 1)) << 7) | (rand() & 127)) << (16 * q);
      v = w;
    }
  }
  int* src; float* sink;
  hipMalloc(&src, h.size() * 4); hipMalloc(&sink, 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int block = 256 * wps, grid = 256 * wg_per_cu, iters = 200000;
  const double flop_per_launch = double(grid) * (block / 64) * iters * (mx ? 8 * 2.0 * 32 * 32 * 64 : 16 * 2.0 * 32 * 32 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double total_ms = 0; int n = 0;
  while (total_ms < secs * 1e3) {
    hipEventRecord(e0);
    if (mx) hipLaunchKernelGGL(spin<true>, dim3(grid), dim3(block), 0, 0, src, sink, iters, prio_mode);
    else hipLaunchKernelGGL(spin<false>, dim3(grid), dim3(block), 0, 0, src, sink, iters, prio_mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    total_ms += ms; ++n;
    printf("%s %s waves/SIMD %d x %d prio %d: launch %d  %.1f ms  %.1f TFLOP/s\n", mx ? "mxfp8" : "bf16", zero ? "zero" : "rand", wps, wg_per_cu, prio_mode, n, ms, flop_per_launch / ms * 1e-9);
  }
  return 0;
}
