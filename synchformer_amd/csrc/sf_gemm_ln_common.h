// Shared by sf_gemm_ln.hip (bf16 operands) and sf_gemm_ln_mx.hip (MXFP8 operands): the LDS layout of the 128 x 768 full-row tile, the LDS-DMA
// issue helpers (inline asm: the compiler neither counts these loads nor guards later ds_reads with vmcnt(0)) and the counted waits.
#pragma once
#include "sf_common.h"

#define RL_BM 128
#define RL_N 768
#define RL_BK 32
#define RL_A_BYTES (RL_BM * RL_BK * 2)                 // 8 KiB
#define RL_B_BYTES (RL_N * RL_BK * 2)                  // 48 KiB
#define RL_STAGE (RL_A_BYTES + RL_B_BYTES)             // 56 KiB
#define RL_SLAB_OFF (2 * RL_STAGE)                     // per-wave 16 x 64 fp32 transposition slabs
#define RL_SLAB_BYTES 4096
#define RL_STAT_OFF (RL_SLAB_OFF + 8 * RL_SLAB_BYTES)  // [128 rows][4 column waves] partial sums, twice (mean pass, variance pass)
#define RL_STAT_BYTES (RL_BM * 4 * 4)
#define RL_PARAM_OFF (RL_STAT_OFF + 2 * RL_STAT_BYTES) // bias | gamma | beta (768 floats each), staged once per workgroup
#define RL_LDS (RL_PARAM_OFF + 3 * RL_N * 4)           // 157 KiB
#define RL_RING_WAVE 12288                             // residual landing ring inside the (idle) operand slots: 3 steps x 4 KiB per wave

// ---- round 3: the quadrant-phased schedule of sf_gemm_pp.hip on this tile (template parameter PP) --------------------------------------------
// A k-step (32 deep) is THREE phases, one per third of the wave's 192 columns (2 x 2 blocks x 2 k-halves = 8 MFMAs each); the stage is cut into
// A (8 KiB) | W0 | W1 | W2 (16 KiB each: for every column wave its 64 columns c*64 .. c*64+63); A's fragments stay in registers for the three phases.
// Every part is refilled (for k-step kt+2) two phases after its last fragment read, one W third (2 LDS-DMA pieces per wave, + the A piece with W0)
// per phase, so that one whole stage (7 pieces per wave = 56 KiB per CU) is in flight behind every counted wait (vmcnt 7), and the wm = 1 waves run
// one barrier behind the wm = 0 waves: on every SIMD one wave multiplies while the other reads fragments and issues loads.  The stage stride is
// 64 KiB (bit 16 of the four fragment-address registers is flipped once per k-step; the 8 KiB between the stages hold the row statistics and the bias).
// The load stream is per tile (the residual of the epilogue lands in the idle operand slots, as before); same products in the same order as the
// round-2 loop, so the two schedules are bit-identical (tests/test_kernels_gpu.py::test_gemm_res_ln_schedules_bitwise).
#define RP_STRIDE 65536
#define RP_W_OFF 8192
#define RP_THIRD 16384
#define RP_STAT_OFF 57344                              // 4 KiB of row statistics + 3 KiB of bias in the gap between the stages
#define RP_BIAS_OFF (RP_STAT_OFF + 2 * RL_STAT_BYTES)
#define RP_SLAB_OFF (RP_STRIDE + RL_STAGE)             // 122880
#define RP_GB_OFF (RP_SLAB_OFF + 8 * RL_SLAB_BYTES)    // gamma | beta
#define RP_LDS (RP_GB_OFF + 2 * RL_N * 4)              // 161792 B
typedef __attribute__((ext_vector_type(4))) unsigned int rl_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int rl_u32x2;

// One stage of LDS-DMA for this wave: its A piece (16 rows) and its six W pieces (96 rows), 1 KiB each.  SGPR base + 32-bit VGPR byte offset
// (zero-extended); M0 = LDS destination of the piece.  Issued from inline asm so that hipcc neither counts these loads nor guards later
// ds_reads with vmcnt(0); every wait for them is a hand-placed counted s_waitcnt.  M0 is saved / restored inside the statement.
__device__ __forceinline__ void rl_dma7(uint32_t voff_a, const void* sa, uint32_t voff_b, const void* sb0, const void* sb1, const void* sb2,
                                        const void* sb3, const void* sb4, const void* sb5, uint32_t lds_a, uint32_t lds_b) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b32 m0, %11\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %8\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %9\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_a), "v"(voff_b), "s"(sa), "s"(sb0), "s"(sb1), "s"(sb2), "s"(sb3), "s"(sb4), "s"(sb5), "s"(lds_a), "s"(lds_b)
      : "memory", "scc");
}

// One piece (SF_RL_SPREAD: the refill's seven pieces are issued one at a time between MFMAs instead of back to back).
__device__ __forceinline__ void rl_dma1(uint32_t voff, const void* sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}

// two consecutive pieces (a W third of this wave)
__device__ __forceinline__ void rl_dma2(uint32_t v0, uint32_t v1, const void* sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sbase), "s"(lds) : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void rl_wait_vmcnt() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void rl_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ void rl_dma1_nt(uint32_t voff, const void* sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}

// Four 1-KiB pieces of the fp32 residual (4 rows x 64 columns each) -> consecutive KiB of this wave's landing ring; `nt`: read once.
__device__ __forceinline__ void rl_dma_r4(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, const void* sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5 nt\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5 nt\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5 nt\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds)
      : "memory", "scc");
}

__device__ __forceinline__ uint32_t rl_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

template <int N>
__device__ __forceinline__ void rl_wait_vmcnt_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4); every lane of the row ends with the total
__device__ __forceinline__ float rl_row16_sum(float v) {
  int t;
  t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);  v += __int_as_float(t);   // quad_perm [1,0,3,2]
  t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);  v += __int_as_float(t);   // quad_perm [2,3,0,1]
  t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true); v += __int_as_float(t);   // row_half_mirror
  t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true); v += __int_as_float(t);   // row_mirror
  return v;
}

