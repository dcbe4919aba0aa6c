// Persistent 256 x 256 x 64 bf16 GEMM, quadrant-phased ("ping-pong") main loop - config 11 of sf_gemm_bf16.
//
// Round 2's persistent kernel (config 7, sf_gemm.hip) waits vmcnt(0) + barrier at the top of every k-tile and only then refills the
// other ring slot: its prefetch distance is ONE 64-KiB stage and a k-tile costs the load time (~1.5-1.9 us against 0.86 us of matrix
// work), profiles/r02_gemm_ln.md.  This kernel keeps the tile, the operand bytes and the epilogues and replaces the schedule:
//
//   * the two 64-KiB stages are split into HALF-TILES of 16 KiB (128 rows x 64 k): per stage A0 | A1 | B0 | B1.  Half Ah holds, for both
//     wave rows wm, the 64 token rows wm*128 + h*64 .. +63 of the tile; half Bh holds, for the four wave columns wn, the 32 weight rows
//     wn*64 + h*32 .. +31.  A wave's 128 x 64 output block is four QUADRANTS (Aha x Bhb = 64 x 32 outputs, 8 MFMAs of 32x32x16 per k-tile).
//   * a k-tile is four PHASES, one quadrant each, in the order (A0,B0) (A0,B1) (A1,B1) (A1,B0): A0 is dead after its first fragment
//     read (phase 0), B1 after phase 1, A1 after phase 2, B0 after phase 0 (its fragments stay in registers for phase 3) - so every half-tile
//     buffer is free for its refill (k-tile kt+2) two phases after its last read, and ONE half-tile (2 LDS-DMA pieces per wave) is issued
//     per phase: the load stream runs 6 half-tiles = 1.5 stages ahead of the reads, continuously, across k-tiles AND across output tiles.
//   * waits are COUNTED: s_waitcnt vmcnt(8) leaves the four youngest half-tiles (64 KiB per CU) in flight across the barriers; vmcnt(0)
//     only in the last two k-tiles of a workgroup's last tile.
//   * every phase is { fragment reads + DMA issue + counted wait | s_barrier | 8 MFMAs under s_setprio 1 | s_barrier }, and the wm = 1
//     waves run ONE barrier behind the wm = 0 waves: on every SIMD (which hosts one wave of each group) one wave is in its matrix
//     segment while the other reads fragments and issues loads.  The groups are re-aligned around the epilogue (both run it
//     concurrently).
//   * LDS ordering rules used (cdna_hip_programming.md, "256^2 8-phase template"): data landed by LDS-DMA is read one phase AFTER the
//     counted wait that retires it, with the wait in front of the phase's FIRST barrier (so that the lagging group's wait also precedes
//     the leading group's read); a buffer is refilled two phases after its last read.
// Fragment registers: a[2][4] (one A half, 2 row blocks x 4 k-steps), b0[4], b1[4]: 64 VGPRs beside the 128 accumulators.
// Schedules measured against this one and dropped (profiles/r03_gemm_pp.md): the phase's LDS-DMA issue behind the first MFMAs of the matrix segment
// (-5 %) or in front of the fragment reads (neutral); the NEXT phase's fragment reads interleaved with the MFMAs into the registers they free
// (-3 %: any instruction between the MFMAs costs the matrix pipe more than the read segment gains); reads balanced 8 / 4 / 8 / 4 (with the first:
// -5 %); two 16-MFMA phases per k-tile, i.e. half the barriers (+1 %, within noise).  The matrix segment has to stay pure MFMA.
#include "sf_gemm_common.h"
#include <type_traits>

#define Q_HALF (128 * 128)             // 16 KiB half-tile: 128 rows x 64 k (bf16)
#define Q_STAGE (4 * Q_HALF)           // A0 | A1 | B0 | B1
#define Q_SLAB_BYTES 4096
#define Q_LDS (2 * Q_STAGE + 8 * Q_SLAB_BYTES)   // 160 KiB

#ifndef SF_PP_PRIO
#define SF_PP_PRIO 1                   // s_setprio 1 around the MFMA segments
#endif
#ifndef SF_PP_SLIM
#define SF_PP_SLIM 1                   // 1: branch-free read segments - running k-tile base pointers, and a dry load iterator (last k-tiles of a workgroup's last
#endif                                 //    tile) issues DUMMY pieces into the idle epilogue slab, so that every counted wait keeps its count (no `issued` branch)
#ifndef SF_PP_NOPEEL
#define SF_PP_NOPEEL 0                 // 1: the first k-tile pair of a tile is not recognisable to the optimiser (no peeled second copy of the loop body)
#endif
#ifndef SF_PP_ABL
#define SF_PP_ABL 0                    // throw-away ablation builds (tools/ab_pp.sh): 1 no epilogue, 2 no LDS-DMA in the loop, 4 no MFMA, 8 no stagger, 16 no fragment reads in the loop, 64 no B1 reads, 128 no A1 reads (wrong results: what does a lighter LDS read load buy?)
#endif
#ifndef SF_PP_STORECNT
#define SF_PP_STORECNT 1               // first k-tile after an epilogue: allow the epilogue's stores to stay in flight (vmcnt 8 + stores)
#endif

// two LDS-DMA pieces (1 KiB each, consecutive in LDS from the wave-uniform byte address l0): SGPR base + 32-bit lane offsets
__device__ __forceinline__ void pp_dma2(uint32_t v0, uint32_t v1, const void* sbase, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "s"(sbase), "s"(l0)
      : "memory", "scc");
}
// the same with a cache-policy suffix on the two loads (measurement builds: -DSF_PP_A_POLN=1 streams the A panels through L2 without displacing the
// chunk's weight slice; " sc1" / " sc0 sc1" = the other policies of the instruction)
#ifndef SF_PP_A_POLN
#define SF_PP_A_POLN 0                 // 0 default policy | 1 nt | 2 sc1 | 3 sc0 sc1
#endif
#if SF_PP_A_POLN == 1
#define SF_PP_A_POL " nt"
#elif SF_PP_A_POLN == 2
#define SF_PP_A_POL " sc1"
#elif SF_PP_A_POLN == 3
#define SF_PP_A_POL " sc0 sc1"
#else
#define SF_PP_A_POL ""
#endif
__device__ __forceinline__ void pp_dma2_a(uint32_t v0, uint32_t v1, const void* sbase, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %3" SF_PP_A_POL "\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" SF_PP_A_POL "\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "s"(sbase), "s"(l0)
      : "memory", "scc");
}
// one 256-byte piece: 64 lanes x 4 bytes (the wave's 64 bias values) to the wave-uniform LDS byte address l0
__device__ __forceinline__ void pp_dma_row256(const void* gsrc, uint32_t l0) {
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
__device__ __forceinline__ void pp_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void pp_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
#define PP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

template <int V> using ic = std::integral_constant<int, V>;

#if SF_PP_ABL & 32
// tracing build (SF_PP_ABL & 32): per wave, s_memtime cycles spent in [read segment | wait at the first barrier | matrix segment | wait at the second
// barrier | epilogue], summed over the launch: g_pp_trace[block][wave][5]
__device__ unsigned long long g_pp_trace[256 * 8 * 5];
#define PP_T(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); var = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#endif

// DUAL (with GELU, bf16 output, no residual): the pre-activation goes to p.C2 as well - the forward of a TRAINED MLP keeps both (fc1 of the Stage-1 towers:
// the backward needs gelu'(pre), fc2 reads gelu(pre)); was sf_gemm_bf16 -> sf_gelu_fwd, a second pass over the (rows, 3072) activations.
template <bool OUT_BF16, bool GELU, bool HAS_RES, bool DUAL = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_kernel(GemmArgs p) {
  constexpr bool WIDE = OUT_BF16 && !HAS_RES;                    // transposed accumulator blocks + 16-byte row stores (as config 7)
  static_assert(!DUAL || (WIDE && GELU), "DUAL is the bf16 GELU epilogue with a second output");
  constexpr int EPI_STORES = WIDE ? (DUAL ? 32 : 16) : 32;       // buffer stores per wave and epilogue
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                       // 2 x 4 waves, wave tile 128 x 64
  const int l31 = lane & 31, hi = lane >> 5;

  // persistent schedule (as config 7): block b sits on XCD b % 8; every XCD owns a contiguous range of 256-row panels and sweeps it
  // once per chunk of `nchunk` column tiles, N fastest inside a chunk
  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t tiles_m = p.tiles_total / p.tiles_n;
  const uint32_t mp8 = (tiles_m + 7u) >> 3;
  const uint32_t mp0 = min(xcd * mp8, tiles_m), mp1 = min(mp0 + mp8, tiles_m), n_mp = mp1 - mp0;
  const uint32_t gchunk = p.nchunk ? min(p.nchunk, p.tiles_n) : p.tiles_n;
  const uint32_t n_chunks = (p.tiles_n + gchunk - 1) / gchunk, chunk_tiles = n_mp * gchunk;
  const uint32_t t_end = n_mp * p.tiles_n;
  auto tile_origin = [&](uint32_t t, int64_t& m0, int& n0) {
    const uint32_t c = min(t / chunk_tiles, n_chunks - 1), r = t - c * chunk_tiles;
    const uint32_t gw = (c == n_chunks - 1) ? p.tiles_n - c * gchunk : gchunk;
    const uint32_t tm = mp0 + r / gw, tn = c * gchunk + r % gw;
    m0 = (int64_t)tm * 256; n0 = (int)tn * 256;
  };

  // ---- fragment read offsets: buffer row r of a half-tile is 128 B, 16-byte chunk c at slot c ^ ((r >> 1) & 7) -------------------------
  const int sw = (l31 >> 1) & 7;
  int a_off[4], b_off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    a_off[kk] = (wm * 64 + l31) * 128 + (((kk * 2 + hi) ^ sw) << 4);                 // + ha * Q_HALF + i * 4096
    b_off[kk] = 2 * Q_HALF + (wn * 32 + l31) * 128 + (((kk * 2 + hi) ^ sw) << 4);    // + hb * Q_HALF
  }

  // ---- load iterator: runs 6 half-tiles ahead of the reads over the workgroup's whole k-tile sequence ---------------------------
  const int nk = p.K / 64;
  if (li >= t_end) return;
  if (t_end && (t_end - 1u - li) / per_xcd_blocks < (t_end - 1u) / per_xcd_blocks) for (uint32_t i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(32);
  uint32_t ld_t = li;
  bool ld_ok = ld_t < t_end;
  if (!ld_ok) return;
  int ld_kt = 0;
  const char* ldA = nullptr;
  const char* ldW = nullptr;
  uint32_t oA[2][2], oW[2][2];                                   // byte offsets of this lane's pieces from ldA / ldW (tile-relative)
  auto ld_set = [&](uint32_t t) {
    int64_t m0; int n0;
    tile_origin(t, m0, n0);
    ldA = reinterpret_cast<const char*>(p.A + m0 * p.lda);
    ldW = reinterpret_cast<const char*>(p.W + (int64_t)n0 * p.ldw);
    const int64_t mleft = p.M - 1 - m0;
    const int mrem = mleft < 255 ? (int)mleft : 255, nrem = min(p.N - 1 - n0, 255);
    int ltid = threadIdx.x;
    asm volatile("" : "+v"(ltid));                               // recompute the lane's row / chunk here (a few VALU operations per tile) instead of
    const int lr = (ltid & 63) >> 3, lc = ltid & 7;              // keeping them in registers - spilled, with a vmcnt(0) at the reload - across the k-loop
    const uint32_t lda2 = (uint32_t)(p.lda * 2), ldw2 = (uint32_t)(p.ldw * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wave * 16 + j * 8 + lr;                      // buffer row this lane fills
      const uint32_t gch = (uint32_t)((lc ^ ((r >> 1) & 7)) << 4);   // source-side swizzle (bytes)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tr = min((r >> 6) * 128 + h * 64 + (r & 63), mrem);        // rows beyond M re-read the last valid row
        const int tc = min((r >> 5) * 64 + h * 32 + (r & 31), nrem);
        oA[h][j] = __umul24((uint32_t)tr, lda2) + gch;          // 24-bit operands (launcher: ld < 2^22): one v_mad_u32_u24, a 32-bit result
        oW[h][j] = __umul24((uint32_t)tc, ldw2) + gch;
      }
    }
  };
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 2048);
  const uint32_t slab_dummy = __builtin_amdgcn_readfirstlane(lds_addr(smem) + 2 * Q_STAGE + wave * Q_SLAB_BYTES + 1024);   // 2 KiB behind the slab's bias row
  const char* curA = nullptr; const char* curW = nullptr;
  const int64_t wk2 = p.wk * 2;
  // PART: 0 = A0, 1 = B0, 2 = B1, 3 = A1 (the order in which a k-tile's halves are read); returns whether a load was issued
  // SURE: the caller has established that the iterator cannot run dry at this point (no branch)
  auto issue = [&](auto PARTc, auto STc, auto SUREc) -> bool {
    constexpr int PART = decltype(PARTc)::value, ST = decltype(STc)::value;
    constexpr bool isA = PART == 0 || PART == 3;
    constexpr int h = PART >= 2 ? 1 : 0;
    if (SF_PP_SLIM) {
      // branch-free: the k-tile bases run along (curA / curW); when the iterator is dry they stay on the last valid k-tile and the pieces go to this
      // wave's (idle) epilogue slab instead of the ring - same instruction count, same vmcnt bookkeeping, nobody reads them
      const uint32_t real = lds_wave + ST * Q_STAGE + (isA ? h : 2 + h) * Q_HALF;
      const uint32_t l = ld_ok ? real : slab_dummy;
      if ((SF_PP_ABL & 2) && !decltype(SUREc)::value) {}
      else if (isA) pp_dma2_a(oA[h][0], oA[h][1], curA, l);
      else pp_dma2(oW[h][0], oW[h][1], curW, l);
      if (PART == 3 && ld_ok) {
        if (++ld_kt == nk) {
          ld_kt = 0;
          ld_t += per_xcd_blocks;
          ld_ok = ld_t < t_end;
          if (ld_ok) { ld_set(ld_t); curA = ldA; curW = ldW; }
        } else {
          curA += 128; curW += wk2;
        }
      }
      return true;
    }
    const bool did = decltype(SUREc)::value ? true : ld_ok;
    if (did) {
      const uint32_t l = lds_wave + ST * Q_STAGE + (isA ? h : 2 + h) * Q_HALF;
      if ((SF_PP_ABL & 2) && !decltype(SUREc)::value) {}
      else if (isA) pp_dma2(oA[h][0], oA[h][1], ldA + (int64_t)ld_kt * 128, l);
      else pp_dma2(oW[h][0], oW[h][1], ldW + (int64_t)ld_kt * p.wk * 2, l);
      if (PART == 3) {                                           // k-tile complete: advance (to the next tile after the last k-tile)
        if (++ld_kt == nk) {
          ld_kt = 0;
          ld_t += per_xcd_blocks;
          ld_ok = ld_t < t_end;
          if (ld_ok) ld_set(ld_t);
        }
      }
    }
    return did;
  };

  // ---- compute-side state ------------------------------------------------------------------------------------------------------------
  uint32_t t = li;
  int64_t m0; int n0;
  tile_origin(t, m0, n0);
  char* bslab = smem + 2 * Q_STAGE + wave * Q_SLAB_BYTES;
  const uint32_t slab_lds = __builtin_amdgcn_readfirstlane(lds_addr(bslab));
  const uint32_t esz = OUT_BF16 ? 2u : 4u;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)(uint32_t)(p.M * p.ldc * esz), 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), (short)0, HAS_RES ? (int)(uint32_t)(p.M * p.ldr * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc2 = __builtin_amdgcn_make_buffer_rsrc(DUAL ? p.C2 : p.C, (short)0, DUAL ? (int)(uint32_t)(p.M * p.ldc * esz) : 0, 0x00020000);
  const uint32_t cstep = (uint32_t)(4 * p.ldc) * esz, rstep = (uint32_t)(4 * p.ldr) * 4u;
  const bool has_bias = p.bias != nullptr;
  // WIDE: the wave's 64 bias values travel by one LDS-DMA piece into the (idle) epilogue slab at the top of the tile - no
  // compiler-visible load in the kernel, so hipcc never inserts a vmcnt(0) that would drain the operand stream
  auto issue_bias = [&](int n0_) {
    if (WIDE && has_bias) pp_dma_row256(p.bias + min(n0_ + wn * 64 + lane, p.N - 1), slab_lds);
  };

  // ---- prologue: k-tile 0 and A0 | B0 of k-tile 1 in flight -------------------------------------------------------------------
  ld_set(ld_t);
  curA = ldA; curW = ldW;
  issue_bias(n0);
  issue(ic<0>{}, ic<0>{}, ic<1>{}); issue(ic<1>{}, ic<0>{}, ic<1>{}); issue(ic<2>{}, ic<0>{}, ic<1>{}); issue(ic<3>{}, ic<0>{}, ic<1>{});
  issue(ic<0>{}, ic<1>{}, ic<1>{}); issue(ic<1>{}, ic<1>{}, ic<1>{});
  asm volatile("" ::: "memory");
  pp_wait_vmcnt<8>();                                            // A0 | B0 of k-tile 0 have landed (this wave's pieces)   // A0 | B0 (schedule 3: and B1) of k-tile 0 have landed (this wave's pieces)
  pp_barrier();
  int extra = 0;                                                 // epilogue stores that may still be in flight behind the loads (first k-tile of a tile)

  for (;;) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[2][4], b0[4], b1[4];

    auto wait_loads = [&](bool issued, bool first) {
      asm volatile("" ::: "memory");
      if (!SF_PP_SLIM && !issued) pp_wait_vmcnt<0>();
      else if (SF_PP_STORECNT && first && extra) pp_wait_vmcnt<8 + EPI_STORES>();
      else pp_wait_vmcnt<8>();
    };
    auto mma = [&](auto HAc, auto HBc, const bf16x8 (&bf)[4]) {
      constexpr int HA = decltype(HAc)::value, HB = decltype(HBc)::value;
      if (SF_PP_PRIO) __builtin_amdgcn_s_setprio(1);
      if (SF_PP_ABL & 4) { asm volatile("" :: "v"(a[0][0]), "v"(a[1][3]), "v"(bf[0]), "v"(bf[3])); if (SF_PP_PRIO) __builtin_amdgcn_s_setprio(0); return; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[HA * 2 + i][HB] = WIDE ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[kk], a[i][kk], acc[HA * 2 + i][HB], 0, 0, 0)   // C^T block: lanes = tokens
                                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kk], bf[kk], acc[HA * 2 + i][HB], 0, 0, 0);
      // an opaque use pins the (pure) MFMAs inside their matrix segment: LLVM otherwise may sink them towards their next use across the run-time
      // branches of the read segments (it did in the MXFP8 fork of this loop: 400 spilled registers)
      asm volatile("" : "+v"(acc[HA * 2][HB]), "+v"(acc[HA * 2 + 1][HB]));
      if (SF_PP_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    // one k-tile held in stage S; `first` = first k-tile after an epilogue
#if SF_PP_ABL & 32
    unsigned long long tr_t0, tr_t1, tr_t2, tr_t3, tr_acc[5] = {0, 0, 0, 0, 0};
    PP_T(tr_t0);
    auto bar_a = [&]() { PP_T(tr_t1); pp_barrier(); PP_T(tr_t2); tr_acc[0] += tr_t1 - tr_t0; tr_acc[1] += tr_t2 - tr_t1; };
    auto bar_b = [&]() { PP_T(tr_t3); pp_barrier(); PP_T(tr_t0); tr_acc[2] += tr_t3 - tr_t2; tr_acc[3] += tr_t0 - tr_t3; };
#else
    auto bar_a = [&]() { pp_barrier(); };
    auto bar_b = [&]() { pp_barrier(); };
#endif
    auto ktile = [&](auto Sc, auto SURE, bool first) {
      constexpr int S = decltype(Sc)::value;
      const char* st = smem + S * Q_STAGE;
      const bool rd = !(SF_PP_ABL & 16) || first;
      // ---- phase 0: (A0, B0) ----
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) if (rd) b0[kk] = *reinterpret_cast<const bf16x8*>(st + b_off[kk]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) if (rd) a[i][kk] = *reinterpret_cast<const bf16x8*>(st + i * 4096 + a_off[kk]);
      __builtin_amdgcn_sched_barrier(0);
      wait_loads(issue(ic<2>{}, ic<S ^ 1>{}, SURE), first);            // B1(kt+1) issued; B1(kt) landed
      bar_a();
      __builtin_amdgcn_sched_barrier(0);
      mma(ic<0>{}, ic<0>{}, b0);
      __builtin_amdgcn_sched_barrier(0);
      bar_b();
      // ---- phase 1: (A0, B1) ----
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { if (SF_PP_ABL & 64) b1[kk] = b0[kk]; else if (rd) b1[kk] = *reinterpret_cast<const bf16x8*>(st + Q_HALF + b_off[kk]); }   // (ABL 64: no B1 fragment reads - a sixth of the LDS read bytes)
      __builtin_amdgcn_sched_barrier(0);
      wait_loads(issue(ic<3>{}, ic<S ^ 1>{}, SURE), first);            // A1(kt+1) issued; A1(kt) landed
      bar_a();
      __builtin_amdgcn_sched_barrier(0);
      mma(ic<0>{}, ic<1>{}, b1);
      __builtin_amdgcn_sched_barrier(0);
      bar_b();
      // ---- phase 2: (A1, B1) ----
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) if (rd && !(SF_PP_ABL & 128)) a[i][kk] = *reinterpret_cast<const bf16x8*>(st + Q_HALF + i * 4096 + a_off[kk]);   // (ABL 128: no A1 reads - a third)
      __builtin_amdgcn_sched_barrier(0);
      issue(ic<0>{}, ic<S>{}, SURE);                             // A0(kt+2); phase 3 reads nothing: no wait
      bar_a();
      __builtin_amdgcn_sched_barrier(0);
      mma(ic<1>{}, ic<1>{}, b1);
      __builtin_amdgcn_sched_barrier(0);
      bar_b();
      // ---- phase 3: (A1, B0) ----
      wait_loads(issue(ic<1>{}, ic<S>{}, SURE), first);                // B0(kt+2); A0 | B0 of k-tile kt+1 landed
      bar_a();
      __builtin_amdgcn_sched_barrier(0);
      mma(ic<1>{}, ic<0>{}, b0);
      __builtin_amdgcn_sched_barrier(0);
      bar_b();
    };

    if (wm == 1 && !(SF_PP_ABL & 8)) pp_barrier();                                   // the wm = 1 waves run one barrier behind
    int kt_first = 0;
    if (SF_PP_NOPEEL) asm volatile("" : "+s"(kt_first));           // opaque zero: with a literal `kt == 0` hipcc peels the first iteration (a second copy of the body)
    for (int kt = 0; kt < nk; kt += 2) {       // (two copies of the k-tile body - a branch-free one for the steady state - cost 250+ spilled registers)
      ktile(ic<0>{}, ic<0>{}, kt == kt_first);
      ktile(ic<1>{}, ic<0>{}, false);
    }
    if (wm == 0 && !(SF_PP_ABL & 8)) pp_barrier();                                   // re-align: both groups run the epilogue concurrently

    if (SF_PP_SLIM && !ld_ok) pp_wait_vmcnt<0>();                  // dummy pieces of a dry iterator land in the slab: all of them before the epilogue uses it
    const int64_t em0 = m0; const int en0 = n0;
    extra = 0;
#if SF_PP_ABL & 32
    unsigned long long tr_e0; PP_T(tr_e0);
#endif
    if (SF_PP_ABL & 1) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
      if (sum == 1.2345e30f) reinterpret_cast<float*>(p.C)[0] = sum;
    } else if (WIDE) {
      // ---- wide bf16 epilogue (config 7's): 4 passes of 32 tokens x 64 features through the wave's 4 KiB slab (rows of 128 B, 16-byte chunk c of
      // row t at slot c ^ (t & 7), the two 8-byte halves of a chunk swapped in rows with bit 3 set) ----
      int etid = threadIdx.x;
      asm volatile("" : "+v"(etid));                              // lane-derived epilogue values must not be hoisted across the k-loop
      const int el = etid & 63, el31 = el & 31, ehi = el >> 5;
      if (en0 + wn * 64 < p.N) {                                  // wave-uniform
        extra = EPI_STORES;
        const int colbase = en0 + wn * 64;
        if (!has_bias) {
          if (el < 16) *reinterpret_cast<float4*>(bslab + el * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
          PP_WAVE_SYNC();
        }
        float4 bia[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) bia[j][g] = *reinterpret_cast<const float4*>(bslab + (j * 32 + g * 8 + ehi * 4) * 4);
        PP_WAVE_SYNC();
        const int wr_off = el31 * 128 + ((ehi ^ ((el31 >> 3) & 1)) << 3), sw7 = el31 & 7;
        const int tr0 = el >> 3, ch = el & 7;
        const int rd_off = tr0 * 128 + ((ch ^ (tr0 & 7)) << 4);
        const uint32_t cbase = (uint32_t)((em0 + wm * 128 + tr0) * p.ldc + colbase + ch * 8) * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (DUAL) {                                            // the pre-activation first: same slab round trip, stores to C2
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w.x = pack_bf2(acc[i][j][g * 4 + 0] + bia[j][g].x, acc[i][j][g * 4 + 1] + bia[j][g].y);
                w.y = pack_bf2(acc[i][j][g * 4 + 2] + bia[j][g].z, acc[i][j][g * 4 + 3] + bia[j][g].w);
                *reinterpret_cast<u32x2*>(bslab + wr_off + (((j * 4 + g) ^ sw7) << 4)) = w;
              }
            PP_WAVE_SYNC();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const u32x4 v = *reinterpret_cast<const u32x4*>(bslab + rd_off + q * 8 * 128);
              u32x4 o;
              if (q & 1) { o.x = v.z; o.y = v.w; o.z = v.x; o.w = v.y; } else { o = v; }
              __builtin_amdgcn_raw_buffer_store_b128(o, rc2, cbase + (uint32_t)((i * 32 + q * 8) * p.ldc) * 2u, 0, SF_EPI_STORE_AUX);
            }
            PP_WAVE_SYNC();
          }
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float4 x = make_float4(acc[i][j][g * 4 + 0] + bia[j][g].x, acc[i][j][g * 4 + 1] + bia[j][g].y, acc[i][j][g * 4 + 2] + bia[j][g].z,
                                     acc[i][j][g * 4 + 3] + bia[j][g].w);
              if (GELU) {
                sf_f32x2_t g0 = {x.x, x.y}, g1 = {x.z, x.w};
                gelu_erf4(g0, g1);
                x.x = g0.x; x.y = g0.y; x.z = g1.x; x.w = g1.y;
              }
              u32x2 w; w.x = pack_bf2(x.x, x.y); w.y = pack_bf2(x.z, x.w);
              *reinterpret_cast<u32x2*>(bslab + wr_off + (((j * 4 + g) ^ sw7) << 4)) = w;
            }
          PP_WAVE_SYNC();
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(bslab + rd_off + q * 8 * 128);
            u32x4 o;
            if (q & 1) { o.x = v.z; o.y = v.w; o.z = v.x; o.w = v.y; } else { o = v; }
            __builtin_amdgcn_raw_buffer_store_b128(o, rc, cbase + (uint32_t)((i * 32 + q * 8) * p.ldc) * 2u, 0, SF_EPI_STORE_AUX);
          }
          PP_WAVE_SYNC();
        }
      }
    } else if (en0 + wn * 64 < p.N) {
      // ---- general epilogue (fp32 output and / or fp32 residual): 8 branch-free groups of 16 rows x 64 cols through the wave's slab ----
      extra = EPI_STORES;
      float* slab = reinterpret_cast<float*>(bslab);
      const int ecol = (lane & 15) * 4;
      const int gcol = en0 + wn * 64 + ecol;
      const int64_t row0 = em0 + wm * 128 + (lane >> 4);
      const uint32_t coff0 = (uint32_t)(row0 * p.ldc + gcol) * esz, roff0 = (uint32_t)(row0 * p.ldr + gcol) * 4u;
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_bias) bias4 = *reinterpret_cast<const float4*>(p.bias + gcol);
      float4 res[2][4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) res[0][ps] = res[1][ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS_RES) epi_group_load_res(res[0], rr, roff0, rstep);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int i = g >> 1, q2 = g & 1;
        if (HAS_RES && g + 1 < 8) epi_group_load_res(res[(g + 1) & 1], rr, roff0 + (g + 1) * 4 * rstep, rstep);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              slab[(qq * 8 + hi * 4 + r) * 64 + j * 32 + l31] = acc[i][j][(q2 * 2 + qq) * 4 + r];
        float4 v[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) v[ps] = *reinterpret_cast<const float4*>(slab + (ps * 4 + (lane >> 4)) * 64 + ecol);
        epi_group_store<OUT_BF16, GELU, HAS_RES>(v, bias4, res[g & 1], rc, coff0 + g * 4 * cstep, cstep);
      }
    }
#if SF_PP_ABL & 32
    {
      unsigned long long tr_e1; PP_T(tr_e1);
      tr_acc[4] += tr_e1 - tr_e0;
      if (lane == 0 && blockIdx.x < 256) for (int i = 0; i < 5; ++i) g_pp_trace[(blockIdx.x * 8 + wave) * 5 + i] += tr_acc[i];
    }
#endif
    t += per_xcd_blocks;
    if (t >= t_end) break;
    tile_origin(t, m0, n0);
    issue_bias(n0);
  }
}

template <bool OUT_BF16, bool GELU, bool HAS_RES, bool DUAL = false>
static int launch_gemm_pp(GemmArgs a, hipStream_t s) {
  auto kern = gemm_bf16_pp_kernel<OUT_BF16, GELU, HAS_RES, DUAL>;
  if (int rc = sf_prepare_kernel((const void*)kern, Q_LDS, "sf_gemm_bf16")) return rc;
  const int n_cus = sf_cu_count("sf_gemm_bf16");
  if (n_cus <= 0) return -1;
  const int64_t tiles_m = (a.M + 255) / 256;
  a.tiles_n = (uint32_t)((a.N + 255) / 256);
  const int64_t total = tiles_m * a.tiles_n;
  if (total >= ((int64_t)1 << 31)) { sf_set_error("sf_gemm_bf16: too many tiles"); return -1; }
  a.tiles_total = (uint32_t)total;
  static int env_chunk = -2;
  if (env_chunk == -2) { const char* e = getenv("SF_GEMM_NCHUNK"); env_chunk = e ? atoi(e) : -1; }
  { static int st = -1; if (st < 0) { const char* e = getenv("SF_PP_STAGGER"); st = e ? atoi(e) : 0; if (st < 0) st = 0; } a.stagger = (uint32_t)st; }
  if (env_chunk >= 0) a.nchunk = (uint32_t)env_chunk;
  else a.nchunk = a.K <= 1024 ? (uint32_t)(2400000 / (512 * a.K) > 0 ? 2400000 / (512 * a.K) : 1) : 0u;
  int64_t blocks = (n_cus / 8) * 8;                          // one workgroup per CU, a multiple of the 8 XCDs
  if (blocks < 8) blocks = 8;                                    // (a device / partition with fewer than 8 CUs: never an empty grid)
  const int64_t need = ((total + 7) / 8) * 8;
  if (blocks > need) blocks = need;
#if SF_PP_ABL & 32
  static unsigned long long zero[256 * 8 * 5] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pp_trace), zero, sizeof(zero));
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), Q_LDS, s, a);
  SF_LAUNCH_CHECK();
#if SF_PP_ABL & 32
  {
    static unsigned long long host[256 * 8 * 5];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_trace), sizeof(host));
    static int calls = 0;
    if ((calls++ % 16) == 0) {
      double g[2][5] = {{0}};
      for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) for (int i = 0; i < 5; ++i) g[w >> 2][i] += (double)host[(b * 8 + w) * 5 + i] / (256.0 * 4.0);
      const char* nm[5] = {"read segment", "wait barrier A", "matrix segment", "wait barrier B", "epilogue"};
      fprintf(stderr, "[pp trace] M %lld N %d K %d obf %d gelu %d res %d: cycles per wave\n", (long long)a.M, a.N, a.K, (int)OUT_BF16, (int)GELU, (int)HAS_RES);
      for (int i = 0; i < 5; ++i) fprintf(stderr, "    %-16s waves 0-3: %12.0f   waves 4-7: %12.0f\n", nm[i], g[0][i], g[1][i]);
    }
  }
#endif
  return 0;
}

// K % 128 == 0 and K >= 256 (whole pairs of k-tiles, static stage indices), 256 * ld * 2 < 4 GiB (32-bit tile-relative lane offsets)
bool sf_gemm_pp_supported(const GemmArgs& a) {
  return (a.K % 128) == 0 && a.K >= 256 && a.lda < ((int64_t)1 << 22) && a.ldw < ((int64_t)1 << 22);
}

int sf_gemm_pp_dispatch(const GemmArgs& a, bool out_bf16, bool gelu, bool res, hipStream_t s) {
  if (a.C2) {
    if (!(out_bf16 && gelu && !res)) { sf_set_error("sf_gemm_bf16: a second (pre-activation) output needs the bf16 GELU epilogue without residual"); return -1; }
    return launch_gemm_pp<true, true, false, true>(a, s);
  }
  if (out_bf16) {
    if (gelu) return res ? launch_gemm_pp<true, true, true>(a, s) : launch_gemm_pp<true, true, false>(a, s);
    return res ? launch_gemm_pp<true, false, true>(a, s) : launch_gemm_pp<true, false, false>(a, s);
  }
  if (gelu) return res ? launch_gemm_pp<false, true, true>(a, s) : launch_gemm_pp<false, true, false>(a, s);
  return res ? launch_gemm_pp<false, false, true>(a, s) : launch_gemm_pp<false, false, false>(a, s);
}
