// Persistent 256 x 256 x 64 bf16 GEMM on FOUR waves (one per SIMD, the whole 512-register file each) with REGISTER-RESIDENT operand fragments - config 12 of
// sf_gemm_bf16 (bf16 output, optional exact-erf GELU; Mlp.forward vit_helper.py:379-398 is the launch it was built for).  NOT the default: measured level with
// config 11 on the main loop and behind it on the epilogue (profiles/r05_gemm_w4.md); kept, bit-identical to config 11, as the measured alternative.
//
// Why it was built: config 11 (sf_gemm_pp.hip) gives every SIMD two waves that take turns on the matrix pipe, and its epilogue - the VALU-bound erf
// GELU above all: 287 us of fc1's 1670 us - runs with the matrix pipe idle because the accumulators fill the registers.  Config 10 (sf_gemm.hip) had shown ONE wave per
// SIMD with 128 x 128 accumulators running its pinned MFMA stream at 94 % of the pipe's rate - and losing that again to the 16 LDS-DMA issues per k-tile it packed behind
// 16 MFMAs.  This kernel keeps the wave shape and changes the schedule:
//
//   * fragments live in REGISTERS for a whole k-tile: 4 sets (A0 | A1 | B0 | B1, the halves of the wave's 128 rows / 128 columns: 2 blocks x 4 k-steps x 4 registers = 32
//     registers each, 128 VGPRs beside the 256 accumulators in the AGPRs - the MFMAs are asm statements with "+a" operands, with the builtin hipcc spills 200+).  A
//     k-tile is four PHASES of 16 MFMAs - even k-tiles (A0,B0) (A0,B1) (A1,B1) (A1,B0), odd ones (A0,B1) (A0,B0) (A1,B0) (A1,B1) - and in every phase the wave reads
//     ONE set for a later phase into the registers that fell free (8 ds_read_b128, a full phase before their first use: the MFMA stream never waits for LDS).
//   * LDS is a pure LANDING ring of eight 16-KiB parts: a part is dead once every wave has read it and is refilled two phases later with the same part two k-tiles
//     ahead - 4 LDS-DMA pieces per wave and phase - so the load stream runs 6 phases (1.5 k-tiles, ~100 KiB per CU) ahead of the reads, continuously across k-tiles
//     and output tiles; every wait is a counted vmcnt(16), one s_barrier per TWO phases.
//   * one MFMA per issue slot group: what rides behind an MFMA (a fragment read, or the 4 instructions of an LDS-DMA piece) fits its 32 cycles on the pipe.
//   * LDS-DMA through raw buffers (`buffer_load_dwordx4 ... offen lds`): one lane-offset register per operand, rows beyond M (and weight rows beyond N) arrive as zeros
//     from the range check, a dry load iterator points beyond the buffer.
// Part sequence (P_n = the n-th part in read order; read in phase n - 2, issued in phase n - 8, into the buffer of P_(n-8)):
//   A0(0) B0(0) | B1(0) A1(0) A0(1) B1(1) B0(1) A1(1) A0(2) B0(2) | B1(2) ...      (k-tile kt lives in stage kt & 1; K % 128 == 0: every tile starts in stage 0)
// Measured (M = 351,456; config 11 beside it on the same box): qkv-shaped 1105-1140 us against 1070-1085; fc1 + GELU 1750-1775 against 1665-1690; main loop alone
// (no epilogue) 1048-1079 / 1372-1387 us.  The loop is bound by L2 -> LDS delivery (no-MFMA build: 865 us = 45 KB/us per CU, the same wall every GEMM of this library
// sits on) next to an MFMA stream of the same length, and one wave per SIMD cannot run the VALU-bound GELU (256 elements per lane) under its own MFMAs: 4,500
// instructions per tile against ~5 issue slots per MFMA gap x 768 MFMAs, and packed-fp32 VALU beside MFMAs is an anti-lever (MI355X_MICROARCH.md).
#ifdef SF_ABLATION   // config 12: a measured-slower alternative, compiled into the ablation build only
#include "sf_gemm_common.h"
#include <type_traits>

#define R4_PART (128 * 128)            // 16 KiB part: 128 rows x 64 k (bf16)
#define R4_STAGE (4 * R4_PART)         // A0 | A1 | B0 | B1
#define R4_SLAB 4096                   // per wave: the epilogue's transposing slab (32 token rows x 128 B)
#define R4_BIASB 512                   // per wave and tile parity: the wave's 128 bias floats
#define R4_WAVE_AREA (R4_SLAB + 2 * R4_BIASB)
#define R4_LDS (2 * R4_STAGE + 4 * R4_WAVE_AREA)   // 148 KiB
#define R4_DRY 0xF0000000u             // scalar offset of a dry iterator: beyond every buffer this kernel accepts

#ifndef R4_ABL
#define R4_ABL 0                       // measurement builds: 1 no epilogue, 2 no LDS-DMA in the loop (wrong results), 4 no MFMA, 8 no fragment reads in the loop, 16 no barriers in the loop (racy)
#endif
#ifndef R4_BAR2
#define R4_BAR2 1                      // 1: one s_barrier per TWO phases (after the odd ones); 0: after every phase
#endif
#ifndef R4_RD2
#define R4_RD2 0                       // 0: the phase's 8 fragment reads one behind each of its first eight MFMAs; 1: two behind each of the first four
#endif
#ifndef R4_DMA_LATE
#define R4_DMA_LATE 0                  // 0: LDS-DMA pieces behind MFMAs 5 7 9 11; 1: behind 8 10 12 14
#endif
#ifndef R4_DMA_ROT
#define R4_DMA_ROT 0                   // 1: wave w issues piece j behind MFMA 4 j + w (one piece per MFMA slot CU-wide; measured SLOWER: 1234 vs 1140 us, the branches cost more than the burst)
#endif
#ifndef R4_EPI_STORES
#define R4_EPI_STORES 32
#endif

// one LDS-DMA piece (1 KiB: 64 lanes x 16 B, lane-linear in LDS from the wave-uniform byte address l0) through a raw buffer: lane offset v + scalar offset so
// (M0 is written and NOT restored: hipcc has no use of M0 in this kernel - gfx9 LDS instructions do not read it - and two scalar moves per piece are two issue slots
//  of an MFMA gap; `grep m0` over the kernel's ISA shows only these moves)
__device__ __forceinline__ void r4_dma1(uint32_t v, __amdgpu_buffer_rsrc_t r, uint32_t so, uint32_t l0) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(v), "s"(r), "s"(so), "s"(l0) : "memory");
}
// 256 bytes: 64 lanes x 4 B from per-lane addresses
__device__ __forceinline__ void r4_dma_row256(const void* gsrc, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(l0)
      : "memory");
}
template <int N>
__device__ __forceinline__ void r4_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void r4_wait_lgkm0() {               // the builtin (not asm): hipcc's own counter tracking sees it
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xF | (0x7 << 4) | (0x0 << 8) | (0x3 << 14));
}
__device__ __forceinline__ void r4_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
#define R4_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// The MFMA as an asm statement with the accumulator block PINNED to the accumulation registers ("+a"): with the builtin hipcc spreads the 256 accumulators over both
// halves of the unified file and spills 200+ registers; volatile also keeps the stream in program order.  Hazards the assembler cannot see: a block is touched by every
// fourth MFMA only (no back-to-back dependent pair), r4_acc_fence() separates the stream from compiler-generated v_accvgpr_write / _read of the same registers.
__device__ __forceinline__ void r4_mfma(f32x16& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// first touch of a tile's accumulator block: C = 0 (no accumulator initialisation pass: 256 v_accvgpr_write per tile and, with hipcc, a detour through VGPRs that spills)
__device__ __forceinline__ void r4_mfma0(f32x16& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void r4_acc_fence() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
template <int V> using r4c = std::integral_constant<int, V>;
struct R4Kt { uint32_t a, w; };        // scalar byte offsets of a k-tile's A rows / W rows (tile origin + 128 kt)

template <bool GELU>
__global__ __launch_bounds__(256, 1) void gemm_bf16_r4_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;                       // 2 x 2 waves, wave tile 128 x 128

  // persistent schedule (as config 11): block b sits on XCD b % 8; every XCD owns a contiguous range of 256-row panels and sweeps it once per chunk of column tiles
  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t tiles_m = p.tiles_total / p.tiles_n;
  const uint32_t mp8 = (tiles_m + 7u) >> 3;
  const uint32_t mp0 = min(xcd * mp8, tiles_m), mp1 = min(mp0 + mp8, tiles_m), n_mp = mp1 - mp0;
  const uint32_t gchunk = p.nchunk ? min(p.nchunk, p.tiles_n) : p.tiles_n;
  const uint32_t n_chunks = (p.tiles_n + gchunk - 1) / gchunk, chunk_tiles = n_mp * gchunk;
  const uint32_t t_end = n_mp * p.tiles_n;
  if (li >= t_end) return;
  auto tile_origin = [&](uint32_t t, int64_t& m0, int& n0) {
    const uint32_t c = min(t / chunk_tiles, n_chunks - 1), r = t - c * chunk_tiles;
    const uint32_t gw = (c == n_chunks - 1) ? p.tiles_n - c * gchunk : gchunk;
    const uint32_t tm = mp0 + r / gw, tn = c * gchunk + r % gw;
    m0 = (int64_t)tm * 256; n0 = (int)tn * 256;
  };

  // ---- lane-derived offsets, RE-DERIVED at the top of every tile from an asm mbcnt (nothing lane-derived stays live across the epilogue: hipcc spilled them otherwise,
  // and a scratch reload drags a vmcnt(0) through the operand stream) ----
  //   fragment reads inside a part: row r is 128 B, 16-byte chunk c at slot c ^ ((r >> 1) & 7);  a_off / b_off: + i * 4096 for the second 32-row block
  //   LDS-DMA: piece j (0..3) of part h of this wave = part rows (4 wave + j) * 8 .. + 7 = tile rows (wave >> 1) * 128 + h * 64 + (4 (wave & 1) + j) * 8 + (lane >> 3);
  //   lane slot lane & 7 holds source chunk slot ^ ((row >> 1) & 7) = slot ^ (lane >> 4) ^ 4 (j & 1): odd pieces flip bit 6 of the lane offset (ld % 64 == 0)
  const uint32_t lda2 = (uint32_t)(p.lda * 2), ldw2 = (uint32_t)(p.ldw * 2);
  int a_off[4], b_off[4];
  uint32_t vA, vW;
  auto derive = [&]() {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int dl31 = ln & 31, dhi = ln >> 5, sw = (dl31 >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      a_off[kk] = (wm * 64 + dl31) * 128 + (((kk * 2 + dhi) ^ sw) << 4);
      b_off[kk] = (wn * 64 + dl31) * 128 + (((kk * 2 + dhi) ^ sw) << 4);
    }
    vA = (uint32_t)(ln >> 3) * lda2 + (uint32_t)(((ln & 7) ^ (ln >> 4)) << 4);
    vW = (uint32_t)(ln >> 3) * ldw2 + (uint32_t)(((ln & 7) ^ (ln >> 4)) << 4);
  };
  derive();

  // ---- load side --------------------------------------------------------------------------------------------------------------------------------
  const int nk = p.K / 64;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), (short)0, (int)(uint32_t)(p.M * p.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), (short)0, (int)(uint32_t)((int64_t)p.N * p.ldw * 2), 0x00020000);
  const uint32_t rowb = (uint32_t)((wave >> 1) * 128 + (wave & 1) * 32);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const uint32_t lds_wave = lds0 + (uint32_t)wave * 4096u;
  uint32_t it_t = li, it_k = 0, it_a, it_w;
  {
    int64_t m0; int n0;
    tile_origin(it_t, m0, n0);
    it_a = (uint32_t)(m0 * p.lda * 2); it_w = (uint32_t)((int64_t)n0 * p.ldw * 2);
  }
  auto it_next = [&]() -> R4Kt {
    R4Kt r = {it_a, it_w};
    if (++it_k == (uint32_t)nk) {
      it_k = 0;
      it_t += per_xcd_blocks;
      if (it_t < t_end) {
        int64_t m0; int n0;
        tile_origin(it_t, m0, n0);
        it_a = (uint32_t)(m0 * p.lda * 2); it_w = (uint32_t)((int64_t)n0 * p.ldw * 2);
      } else { it_a = R4_DRY; it_w = R4_DRY; it_t = t_end; }
    } else if (it_t < t_end) { it_a += 128u; it_w += 128u; }
    return r;
  };
  // PART: 0 A0 | 1 A1 | 2 B0 | 3 B1
  auto issue_piece = [&](auto PARTc, auto Sc, auto Jc, const R4Kt& d) {
    constexpr int PART = decltype(PARTc)::value, S = decltype(Sc)::value, J = decltype(Jc)::value;
    constexpr bool isA = PART < 2;
    constexpr int h = PART & 1;
    if (R4_ABL & 2) return;
    const uint32_t l = lds_wave + S * R4_STAGE + PART * R4_PART + J * 1024;
    const uint32_t row = rowb + h * 64 + J * 8;
    // (the whole offset rides in the LANE offset: the scalar offset of a buffer instruction is not part of the range check)
    if (isA) r4_dma1(((J & 1) ? (vA ^ 64u) : vA) + (d.a + row * lda2), ra, 0u, l);
    else r4_dma1(((J & 1) ? (vW ^ 64u) : vW) + (d.w + row * ldw2), rw, 0u, l);
  };
  auto issue_part = [&](auto PARTc, auto Sc, const R4Kt& d) {
    issue_piece(PARTc, Sc, r4c<0>{}, d); issue_piece(PARTc, Sc, r4c<1>{}, d); issue_piece(PARTc, Sc, r4c<2>{}, d); issue_piece(PARTc, Sc, r4c<3>{}, d);
  };

  // ---- compute-side state -----------------------------------------------------------------------------------------------------------------------
  uint32_t t = li;
  int64_t m0; int n0;
  tile_origin(t, m0, n0);
  char* wslab = smem + 2 * R4_STAGE + wave * R4_WAVE_AREA;
  const uint32_t bias_lds = __builtin_amdgcn_readfirstlane(lds_addr(wslab) + R4_SLAB);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)(uint32_t)(p.M * p.ldc * 2), 0x00020000);
  const bool has_bias = p.bias != nullptr;
  uint32_t tpar = 0;                                             // tile parity: which bias buffer holds this tile's values
  auto issue_bias = [&](int n0_, uint32_t par) {                 // the wave's 128 bias floats of a tile -> LDS (2 pieces of 256 B)
    if (has_bias) {
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      r4_dma_row256(p.bias + min(n0_ + wn * 128 + ln, p.N - 1), bias_lds + par * R4_BIASB);
      r4_dma_row256(p.bias + min(n0_ + wn * 128 + 64 + ln, p.N - 1), bias_lds + par * R4_BIASB + 256);
    }
  };

  bf16x8 fA[2][2][4], fB[2][2][4];                               // [half][32-row block][k-step]
  f32x16 acc[4][4];
  bool had_stores = false;                                       // the previous tile's epilogue left its stores in flight behind the loads

  // ---- prologue: P_0 .. P_7 in flight (the whole ring), A0(0) | B0(0) into registers --------------------------------------------------------------------------
  R4Kt D0, D1;
  {
    const R4Kt d0 = it_next(), d1 = it_next();
    issue_bias(n0, 0);
    issue_part(r4c<0>{}, r4c<0>{}, d0); issue_part(r4c<2>{}, r4c<0>{}, d0); issue_part(r4c<3>{}, r4c<0>{}, d0); issue_part(r4c<1>{}, r4c<0>{}, d0);
    issue_part(r4c<0>{}, r4c<1>{}, d1); issue_part(r4c<3>{}, r4c<1>{}, d1); issue_part(r4c<2>{}, r4c<1>{}, d1); issue_part(r4c<1>{}, r4c<1>{}, d1);
    asm volatile("" ::: "memory");
    if (R4_ABL & 2) r4_wait_vmcnt<0>(); else r4_wait_vmcnt<28>();   // P_0 = A0(0) (and the bias) landed
    r4_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) fA[0][r >> 2][r & 3] = *reinterpret_cast<const bf16x8*>(smem + (r >> 2) * 4096 + a_off[r & 3]);
    r4_wait_lgkm0();
    if (R4_ABL & 2) r4_wait_vmcnt<0>(); else r4_wait_vmcnt<24>();   // P_1 = B0(0)
    r4_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) fB[0][r >> 2][r & 3] = *reinterpret_cast<const bf16x8*>(smem + 2 * R4_PART + (r >> 2) * 4096 + b_off[r & 3]);
    r4_wait_lgkm0();
    if (R4_ABL & 2) r4_wait_vmcnt<0>(); else r4_wait_vmcnt<16>();   // P_2 = B1(0), P_3 = A1(0)
    r4_barrier();
    D0 = it_next(); D1 = it_next();
  }

  for (;;) {
    // the NEXT tile's bias goes in flight now (into the other buffer)
    {
      const uint32_t tn_ = t + per_xcd_blocks;
      if (tn_ < t_end) { int64_t m1; int n1; tile_origin(tn_, m1, n1); issue_bias(n1, tpar ^ 1u); }
    }

    // one phase: 16 MFMAs (halves HA x HB) | 8 fragment reads of part RP of stage RS | 4 LDS-DMA pieces refilling part FP of stage FS from k-tile descriptor d
    auto phase = [&](auto HAc, auto HBc, auto RPc, auto RSc, auto FPc, auto FSc, const R4Kt& d, bool relaxed, auto ZEROc, auto ODDc) {
      constexpr int HA = decltype(HAc)::value, HB = decltype(HBc)::value, RP = decltype(RPc)::value, RS = decltype(RSc)::value;
      const char* rbase = smem + RS * R4_STAGE + RP * R4_PART;
      auto rd = [&](auto Rc) {
        constexpr int r = decltype(Rc)::value;
        if (R4_ABL & 8) return;
        if (RP < 2) fA[RP & 1][r >> 2][r & 3] = *reinterpret_cast<const bf16x8*>(rbase + (r >> 2) * 4096 + a_off[r & 3]);
        else fB[RP & 1][r >> 2][r & 3] = *reinterpret_cast<const bf16x8*>(rbase + (r >> 2) * 4096 + b_off[r & 3]);
      };
      // 16 slots of { one MFMA | at most ~4 other instructions }: what rides behind an MFMA has to fit its 32 cycles on the pipe (8 issue slots, ~5 usable), or the next
      // MFMA issues late - with the fragment reads and the LDS-DMA sequence of a PAIR of MFMAs in one gap the pipe idled a quarter of the time (profiles/r05_gemm_w4.md)
      auto slot = [&](auto Mc) {
        constexpr int m = decltype(Mc)::value;
        constexpr int kk = m >> 2, i = (m >> 1) & 1, j = m & 1;
        if (!(R4_ABL & 4)) {
          if (decltype(ZEROc)::value && kk == 0) r4_mfma0(acc[HA * 2 + i][HB * 2 + j], fB[HB][j][kk], fA[HA][i][kk]);
          else r4_mfma(acc[HA * 2 + i][HB * 2 + j], fB[HB][j][kk], fA[HA][i][kk]);
        } else if (m == 0) asm volatile("" :: "v"(fA[HA][0][0]), "v"(fA[HA][1][3]), "v"(fB[HB][0][0]), "v"(fB[HB][1][3]));
        if (R4_RD2) { if (m < 4) { rd(r4c<2 * (m & 3)>{}); rd(r4c<2 * (m & 3) + 1>{}); } }
        else if (m < 8) rd(r4c<(m & 7)>{});
        if (R4_DMA_ROT) {                                          // wave w issues piece j behind MFMA 4 j + w: the CU's vector-memory queue sees ONE piece per MFMA slot
          if (wave == (m & 3)) issue_piece(FPc, FSc, r4c<(m >> 2) & 3>{}, d);
        } else {
          constexpr int d0 = R4_DMA_LATE ? 8 : 5;                  // LDS-DMA pieces behind MFMAs d0, d0 + 2, d0 + 4, d0 + 6 (all four waves at once)
          if (m >= d0 && ((m - d0) & 1) == 0 && (m - d0) / 2 < 4) issue_piece(FPc, FSc, r4c<((m - d0) / 2) & 3>{}, d);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      slot(r4c<0>{}); slot(r4c<1>{}); slot(r4c<2>{}); slot(r4c<3>{}); slot(r4c<4>{}); slot(r4c<5>{}); slot(r4c<6>{}); slot(r4c<7>{});
      slot(r4c<8>{}); slot(r4c<9>{}); slot(r4c<10>{}); slot(r4c<11>{}); slot(r4c<12>{}); slot(r4c<13>{}); slot(r4c<14>{}); slot(r4c<15>{});
      if (R4_BAR2 && !decltype(ODDc)::value) return;               // (barrier after the odd phases only)
      constexpr int YOUNG = R4_BAR2 ? 16 : 20;                     // pieces issued after the part(s) that must have landed: 4 per phase over the last 4 (5) phases
      r4_wait_lgkm0();                                             // the fragment reads since the last barrier are complete: their parts may be refilled after this one
      if (R4_ABL & 2) {}
      else if (relaxed) r4_wait_vmcnt<(YOUNG + R4_EPI_STORES > 63 ? 63 : YOUNG + R4_EPI_STORES)>();   // the epilogue's stores sit between the parts read next and the younger loads
      else r4_wait_vmcnt<YOUNG>();                                 // the parts read in the next (two) phases have landed (this wave's pieces)
      if (!(R4_ABL & 16)) r4_barrier();
    };
    // a pair of k-tiles (even in stage 0, odd in stage 1)
    // (a part is refilled TWO phases after the phase that read it - the barrier in between, also with R4_BAR2, proves every wave is done with it - by the same part two
    //  k-tiles ahead: D0 | D1 = the k-tiles 2 and 3 ahead of the pair's even one; phase f issues P_(f+8), the waits leave the 16 (20) youngest pieces in flight)
    auto pair = [&](auto FIRSTc, bool rel) {
      constexpr int F = decltype(FIRSTc)::value;                   // first pair of a tile: every block's first MFMA (k-step 0 of phases E0 .. E3) starts from zero
      phase(r4c<0>{}, r4c<0>{}, r4c<3>{}, r4c<0>{}, r4c<0>{}, r4c<0>{}, D0, rel, r4c<F>{}, r4c<0>{});      // E0: (A0,B0) | read B1(kt)   | refill A0 s0 <- A0(kt+2)
      phase(r4c<0>{}, r4c<1>{}, r4c<1>{}, r4c<0>{}, r4c<2>{}, r4c<0>{}, D0, rel, r4c<F>{}, r4c<1>{});      // E1: (A0,B1) | read A1(kt)   | refill B0 s0 <- B0(kt+2)
      phase(r4c<1>{}, r4c<1>{}, r4c<0>{}, r4c<1>{}, r4c<3>{}, r4c<0>{}, D0, rel, r4c<F>{}, r4c<0>{});      // E2: (A1,B1) | read A0(kt+1) | refill B1 s0 <- B1(kt+2)
      phase(r4c<1>{}, r4c<0>{}, r4c<3>{}, r4c<1>{}, r4c<1>{}, r4c<0>{}, D0, rel, r4c<F>{}, r4c<1>{});      // E3: (A1,B0) | read B1(kt+1) | refill A1 s0 <- A1(kt+2)
      phase(r4c<0>{}, r4c<1>{}, r4c<2>{}, r4c<1>{}, r4c<0>{}, r4c<1>{}, D1, false, r4c<0>{}, r4c<0>{});    // O0: (A0,B1) | read B0(kt+1) | refill A0 s1 <- A0(kt+3)
      phase(r4c<0>{}, r4c<0>{}, r4c<1>{}, r4c<1>{}, r4c<3>{}, r4c<1>{}, D1, false, r4c<0>{}, r4c<1>{});    // O1: (A0,B0) | read A1(kt+1) | refill B1 s1 <- B1(kt+3)
      phase(r4c<1>{}, r4c<0>{}, r4c<0>{}, r4c<0>{}, r4c<2>{}, r4c<1>{}, D1, false, r4c<0>{}, r4c<0>{});    // O2: (A1,B0) | read A0(kt+2) | refill B0 s1 <- B0(kt+3)
      phase(r4c<1>{}, r4c<1>{}, r4c<2>{}, r4c<0>{}, r4c<1>{}, r4c<1>{}, D1, false, r4c<0>{}, r4c<1>{});    // O3: (A1,B1) | read B0(kt+2) | refill A1 s1 <- A1(kt+3)
      D0 = it_next(); D1 = it_next();
    };
    pair(r4c<1>{}, had_stores);
#pragma unroll 1
    for (int kt = 2; kt < nk; kt += 2) pair(r4c<0>{}, false);

    // ---- epilogue: bf16 (GELU) rows through the wave's 4-KiB slab, 16-byte row stores (the transposed-block epilogue of config 11, twice 64 features) ----
    r4_acc_fence();
    const int64_t em0 = m0; const int en0 = n0;
    had_stores = false;
    if (!(R4_ABL & 1) && en0 + wn * 128 < p.N) {                  // wave-uniform (N % 128 == 0)
      had_stores = true;
      int el;                                                     // lane id from mbcnt, as asm: not hoisted out of the tile loop (nothing lane-derived stays live across the k-loop)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
      const int el31 = el & 31, ehi = el >> 5;
      const int wr_off = el31 * 128 + ((ehi ^ ((el31 >> 3) & 1)) << 3), sw7 = el31 & 7;
      const int tr0 = el >> 3, ch = el & 7;
      const int rd_off = tr0 * 128 + ((ch ^ (tr0 & 7)) << 4);
      const float* bs = reinterpret_cast<const float*>(wslab + R4_SLAB + tpar * R4_BIASB);
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        float4 bia[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) bia[j][g] = has_bias ? *reinterpret_cast<const float4*>(bs + jh * 64 + j * 32 + g * 8 + ehi * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const uint32_t cbase = (uint32_t)((em0 + wm * 128 + tr0) * p.ldc + en0 + wn * 128 + jh * 64 + ch * 8) * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float4 x = make_float4(acc[i][jh * 2 + j][g * 4 + 0] + bia[j][g].x, acc[i][jh * 2 + j][g * 4 + 1] + bia[j][g].y, acc[i][jh * 2 + j][g * 4 + 2] + bia[j][g].z,
                                     acc[i][jh * 2 + j][g * 4 + 3] + bia[j][g].w);
              if (GELU) {
                sf_f32x2_t g0 = {x.x, x.y}, g1 = {x.z, x.w};
                gelu_erf4(g0, g1);
                x.x = g0.x; x.y = g0.y; x.z = g1.x; x.w = g1.y;
              }
              u32x2 w; w.x = pack_bf2(x.x, x.y); w.y = pack_bf2(x.z, x.w);
              *reinterpret_cast<u32x2*>(wslab + wr_off + (((j * 4 + g) ^ sw7) << 4)) = w;
            }
          R4_WAVE_SYNC();
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(wslab + rd_off + q * 8 * 128);
            u32x4 o;
            if (q & 1) { o.x = v.z; o.y = v.w; o.z = v.x; o.w = v.y; } else { o = v; }
            __builtin_amdgcn_raw_buffer_store_b128(o, rc, cbase + (uint32_t)((i * 32 + q * 8) * p.ldc) * 2u, 0, SF_EPI_STORE_AUX);
          }
          R4_WAVE_SYNC();
          __builtin_amdgcn_sched_barrier(0);                       // one 32-token block at a time: its 32 accumulator registers + the GELU's temporaries, not the whole tile's
        }
      }
    } else if (R4_ABL & 1) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
      if (sum == 1.2345e30f) reinterpret_cast<float*>(p.C)[0] = sum;
    }
    t += per_xcd_blocks;
    if (t >= t_end) break;
    tile_origin(t, m0, n0);
    tpar ^= 1u;
    derive();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the dry iterator's last pieces (zeros into parts nobody reads) must not outlive the workgroup
}

template <bool GELU>
static int launch_gemm_r4(GemmArgs a, hipStream_t s) {
  auto kern = gemm_bf16_r4_kernel<GELU>;
  if (int rc = sf_prepare_kernel((const void*)kern, R4_LDS, "sf_gemm_bf16")) return rc;
  const int n_cus = sf_cu_count("sf_gemm_bf16");
  if (n_cus <= 0) return -1;
  const int64_t tiles_m = (a.M + 255) / 256;
  a.tiles_n = (uint32_t)((a.N + 255) / 256);
  const int64_t total = tiles_m * a.tiles_n;
  if (total >= ((int64_t)1 << 31)) { sf_set_error("sf_gemm_bf16: too many tiles"); return -1; }
  a.tiles_total = (uint32_t)total;
  static int env_chunk = -2;
  if (env_chunk == -2) { const char* e = getenv("SF_GEMM_NCHUNK"); env_chunk = e ? atoi(e) : -1; }
  if (env_chunk >= 0) a.nchunk = (uint32_t)env_chunk;
  else a.nchunk = a.K <= 1024 ? (uint32_t)(2400000 / (512 * a.K) > 0 ? 2400000 / (512 * a.K) : 1) : 0u;
  int64_t blocks = (n_cus / 8) * 8;                              // one workgroup per CU, a multiple of the 8 XCDs
  if (blocks < 8) blocks = 8;
  const int64_t need = ((total + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), R4_LDS, s, a);
  SF_LAUNCH_CHECK();
  return 0;
}

// bf16 output, no residual, row-major W; K % 128 == 0 (whole k-tile pairs), N % 128 == 0 (a wave stores all of its 128 columns or none), row strides multiples of 64 elements
// (the chunk swizzle lives in bits 4-6 of a lane offset), everything the 32-bit buffer offsets address below R4_DRY
bool sf_gemm_r4_supported(const GemmArgs& a) {
  return (a.K % 128) == 0 && a.K >= 256 && (a.N % 128) == 0 && (a.lda % 64) == 0 && (a.ldw % 64) == 0 && a.lda < ((int64_t)1 << 19) && a.ldw < ((int64_t)1 << 19) &&
         a.wk == 64 && a.M * a.lda * 2 < (int64_t)R4_DRY && (int64_t)a.N * a.ldw * 2 < (int64_t)R4_DRY && !a.R && !a.C2;
}

int sf_gemm_r4_dispatch(const GemmArgs& a, bool gelu, hipStream_t s) {
  return gelu ? launch_gemm_r4<true>(a, s) : launch_gemm_r4<false>(a, s);
}

#endif  // SF_ABLATION
