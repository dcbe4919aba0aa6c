// Shared pieces of the bf16 GEMM kernels (sf_gemm.hip, sf_gemm_pp.hip): argument block, LDS-DMA issue helpers, the branch-free
// buffer-op epilogue tail and the counted-wait + barrier primitive.
#pragma once
#include "sf_common.h"
#include <stdlib.h>
#include "../../include/synchformer_hip.h"

#ifndef SF_ABL
#define SF_ABL 0   // tools/ablate_gemm.sh builds throwaway variants with -DSF_ABL=mask; the product build is 0
#endif
#define EPI_LD 68                          // fp32 row stride of the epilogue staging slab (64 cols + pad, 16-B aligned)

struct GemmArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  void* C; int64_t ldc;
  void* C2 = nullptr;  // config 11, GELU epilogue: also write the pre-activation here (same dtype and row stride as C)
  const float* R; int64_t ldr;
  RowMap cmap, rmap;
  int64_t M;
  int N, K;
  uint32_t tiles_n, tiles_total;
  uint32_t nchunk;   // persistent kernel: column tiles per sweep (0 = all)
  int64_t wk;        // persistent kernel: elements between consecutive 64-deep k-tiles of a W row (64 row-major, N * 64 k-tile-major)
  // strided batch (blockIdx.y = b0 * batch_inner + b1): element offsets added to A / W / C per batch index
  int batch_inner;
  int64_t sA0, sA1, sW0, sW1, sC0, sC1;
  uint32_t stagger = 0;   // config 11: the workgroups that own one tile fewer than their XCD's first start this many x ~1.2 us late (SF_PP_STAGGER)
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) {   // LDS byte address of a __shared__ pointer
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// Four LDS-DMA pieces (1 KiB each, consecutive in LDS from wave-uniform address `l0`) issued from inline asm, so
// hipcc neither counts them nor guards later ds_reads with vmcnt(0) (it does for the builtin once the loop gets
// complicated - that wait serialised prefetch and compute in the first persistent kernel).  Every wait for these
// loads is a hand-placed counted s_waitcnt.  M0 (LDS destination base) is saved/restored inside the statement.
__device__ __forceinline__ void dma4(const void* g0, const void* g1, const void* g2, const void* g3, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(l0)
      : "memory", "scc");
}

// The same four pieces addressed as SGPR base + zero-extended 32-bit lane offsets (operands below 4 GiB): half the address registers per piece.
__device__ __forceinline__ void dma4s(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, const void* sbase, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(l0)
      : "memory", "scc");
}

__device__ __forceinline__ void dma4_nt(const void* g0, const void* g1, const void* g2, const void* g3, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off nt\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(l0)
      : "memory", "scc");
}

// cache-policy bits of the epilogue's buffer ops (aux operand: bit 0 sc0/glc, bit 1 nt/slc, bit 4 sc1)
// Outputs are written once and the fp32 residual is read once: marked non-temporal (nt) so that they do not evict the weight matrix
// (3.5-4.7 MB against a 4 MB L2 per XCD) and the A panels the other column tiles of the same rows are about to read.  Measured on
// M = 175,728: qkv 891 -> 950 TFLOP/s, fc1+GELU 717 -> 744, proj+residual 563 -> 618 (profiles/r01_gemm_configs.md).
#ifndef SF_EPI_STORE_AUX
#define SF_EPI_STORE_AUX 2
#endif
#ifndef SF_EPI_LOAD_AUX
#define SF_EPI_LOAD_AUX 2
#endif
#ifndef SF_DMA_SPREAD
#define SF_DMA_SPREAD 1   // 1 (measured +1-2.6 %): issue the next stage's LDS-DMA behind the first two MFMA clusters instead of right after the barrier
#endif
#ifndef SF_EPI_WIDE
#define SF_EPI_WIDE 1   // persistent kernel, bf16 output without residual: transposed accumulator blocks + 8-byte slab writes + 16-byte row stores
#endif
#ifndef SF_KROT
#define SF_KROT 0      // (measured neutral on every shape: profiles/r02_gemm_ln.md) persistent kernel: workgroup i of an XCD walks its k-loop starting at k-tile (i * SF_KROT) % nk (0 = every workgroup starts at 0)
#endif
#ifndef SF_A_NT
#define SF_A_NT 0      // 1: stream the A operand (activations) through L2 with the nt hint as well
#endif
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Branch-free epilogue tail for one group of 16 rows x 64 cols per wave (4 float4 per lane, rows 4 apart):
// + bias, erf-GELU, + fp32 residual, convert, store.  Row bounds come for free from the buffer descriptors
// (num_records = M * ld * esz: rows >= M are dropped by the hardware range check), so there is no exec-mask
// branching and hipcc keeps counted vmcnt waits: stores never wait for earlier stores, and the residual rows of
// the NEXT group are already in flight (loaded before this group's stores were issued).
template <bool OUT_BF16, bool GELU, bool HAS_RES>
__device__ __forceinline__ void epi_group_store(float4 (&v)[4], const float4& bias4, const float4 (&res)[4],
                                                __amdgpu_buffer_rsrc_t rc, uint32_t coff, uint32_t cstep) {
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    float4 x = v[ps];
    x.x += bias4.x; x.y += bias4.y; x.z += bias4.z; x.w += bias4.w;
    if (GELU) {
      sf_f32x2_t g0 = {x.x, x.y}, g1 = {x.z, x.w};
      gelu_erf4(g0, g1);
      x.x = g0.x; x.y = g0.y; x.z = g1.x; x.w = g1.y;
    }
    if (HAS_RES) { x.x += res[ps].x; x.y += res[ps].y; x.z += res[ps].z; x.w += res[ps].w; }
    if (OUT_BF16) {
      u32x2 o; o.x = pack_bf2(x.x, x.y); o.y = pack_bf2(x.z, x.w);
      __builtin_amdgcn_raw_buffer_store_b64(o, rc, coff + ps * cstep, 0, SF_EPI_STORE_AUX);
    } else {
      u32x4 o;
      o.x = __float_as_uint(x.x); o.y = __float_as_uint(x.y); o.z = __float_as_uint(x.z); o.w = __float_as_uint(x.w);
      __builtin_amdgcn_raw_buffer_store_b128(o, rc, coff + ps * cstep, 0, SF_EPI_STORE_AUX);
    }
  }
}
__device__ __forceinline__ void epi_group_load_res(float4 (&res)[4], __amdgpu_buffer_rsrc_t rr, uint32_t roff, uint32_t rstep) {
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rr, roff + ps * rstep, 0, SF_EPI_LOAD_AUX);
    res[ps] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
  // Counted wait for this wave's own LDS-DMA pieces, then the workgroup barrier.  The wait is the BUILTIN (gfx9
  // encoding: vmcnt[3:0] | expcnt<<4 | lgkmcnt<<8 | vmcnt[5:4]<<14, other counters left at max) so that hipcc's
  // waitcnt pass sees it and stops inserting its own conservative vmcnt(0) in the loop; the empty asm statements are
  // compiler memory fences (the raw s_barrier builtin alone does not order LDS accesses for the compiler).
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// config 11 (sf_gemm_pp.hip): the quadrant-phased persistent 256 x 256 x 64 kernel
bool sf_gemm_pp_supported(const GemmArgs& a);
int sf_gemm_pp_dispatch(const GemmArgs& a, bool out_bf16, bool gelu, bool res, hipStream_t s);
// config 12 (sf_gemm_w4.hip): the same tile on four waves with register-resident fragments (bf16 output, optional GELU, no residual)
bool sf_gemm_r4_supported(const GemmArgs& a);
int sf_gemm_r4_dispatch(const GemmArgs& a, bool gelu, hipStream_t s);
