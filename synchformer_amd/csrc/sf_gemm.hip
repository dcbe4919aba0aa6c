// bf16 GEMM with fused epilogues for the Synchformer hot path on gfx950 (MI355X):
//     C[map(m), n] = epi( sum_k A[m, k] * W[n, k] + bias[n] ) (+ R[rmap(m), n])
// A (activations, M x K) and W (nn.Linear weight, N x K) are both K-contiguous, so both MFMA operands are
// "row, 8 consecutive k" fragments and neither needs a transpose.  This one kernel carries >93 % of the
// model's FLOPs: the fused qkv / proj / fc1 / fc2 Linears of every DividedSpaceTimeBlock
// (vit_helper.py:103,155,392-396), ASTLayer (modeling_ast.py:142-146,199,263,274), sync Block
// (modules/transformer.py:59-61,74,86-91), the aggregator layers, the two patch-embedding convolutions
// (as patch-gather GEMMs), vproj/aproj and the offset head.
//
// Structure (per workgroup = one BM x BN output tile):
//   * operand tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave-instruction) into an
//     NS-deep ring of BK-wide stages; NS-1 stages are kept in flight with COUNTED s_waitcnt vmcnt(N) and one raw
//     s_barrier per K-step (never vmcnt(0) in the steady state) - with K as short as 768 the loop is bound by
//     operand-delivery latency, not MFMA rate, unless several stages are outstanding (profiles/r01 notes).
//   * LDS rows are BK*2 bytes; the 16-byte chunk index is XOR-swizzled so the ds_read_b128 fragment reads are
//     bank-conflict-free.  LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane SOURCE address
//     and to the read address (same involution), never to the LDS destination.
//   * every wave owns a (BM/WM) x 64 block of 16 x 16 fragments of v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
//   * epilogue: accumulators are transposed through per-wave private LDS so bias / erf-GELU / fp32 residual /
//     stores run on 16-byte row-contiguous pieces (full 128/256-byte row segments per 16 lanes).
//   * workgroup -> tile mapping is XCD-aware (block b runs on XCD b % 8): each XCD walks a contiguous range of
//     tiles, N fastest, so an A row-panel is fetched once per XCD-local L2 and W stays L2/MALL-resident.
#include "sf_gemm_common.h"


// Tile configuration: BM x BN x BK block tile, NS-stage LDS ring, WM x WN waves, each wave (BM/WM) x 64 outputs.
template <int BM_, int BN_, int WM_, int WN_, int BK_, int NS_, int WG_PER_CU_, bool PIPE_ = false>
struct GemmCfg {
  static constexpr bool PIPE = PIPE_;      // fragment register double-buffering one K-step ahead (needs BK 32, NS >= 4)
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_, NS = NS_;
  static constexpr int NWAVES = WM * WN, THREADS = NWAVES * 64;
  static constexpr int WT_M = BM / WM, WT_N = BN / WN;           // wave tile
  static constexpr int FI = WT_M / 16, FJ = WT_N / 16;           // 16x16 fragments per wave
  static constexpr int ROW_B = BK * 2;                            // bytes per LDS row
  static constexpr int PIECE_ROWS = 1024 / ROW_B;                 // rows per 1-KiB LDS-DMA piece
  static constexpr int A_BYTES = BM * ROW_B, B_BYTES = BN * ROW_B, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = BM / PIECE_ROWS / NWAVES, B_PIECES = BN / PIECE_ROWS / NWAVES;
  static constexpr int PIECES = A_PIECES + B_PIECES;              // LDS-DMA instructions per thread per stage
  static constexpr int LDS = NS * STAGE;
  static constexpr int MIN_WAVES_PER_SIMD = WG_PER_CU_ * NWAVES / 4;
  static_assert(BK == 32 || BK == 64, "BK must be 32 or 64");
  static_assert(WT_N == 64, "epilogue assumes 64-column wave tiles");
  static_assert(NWAVES * 32 * EPI_LD * 4 <= LDS, "epilogue slabs must fit in the operand LDS");
  static_assert(FI % 2 == 0, "general epilogue walks 32-row halves");
  static_assert(A_PIECES >= 1 && B_PIECES >= 1 && (NS - 2) * PIECES <= 63, "piece / vmcnt budget");
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(!PIPE_ || (BK_ == 32 && NS_ >= 4), "PIPE needs one k-step per stage and a 4-deep ring");
};

// byte offset inside a tile of the 16-byte chunk `c` (0 .. BK/8-1) of row `r`, XOR-swizzled.
template <int BK>
__device__ __forceinline__ int lds_chunk_off(int r, int c) {
  if (BK == 64) return r * 128 + ((c ^ (r & 7)) << 4);          // 8 chunks/row, 2 rows per 256-B bank row
  return r * 64 + ((c ^ ((r >> 2) & 3)) << 4);                   // 4 chunks/row, 4 rows per 256-B bank row
}

template <class Cfg, bool OUT_BF16, bool GELU, bool HAS_RES, bool FAST>
__global__ __launch_bounds__(Cfg::THREADS, Cfg::MIN_WAVES_PER_SIMD) void gemm_bf16_kernel(GemmArgs p) {
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NS = Cfg::NS, FI = Cfg::FI, FJ = Cfg::FJ, P = Cfg::PIECES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (gridDim.y > 1) {                                          // strided batch: shift the operand bases (wave-uniform)
    const int b0 = blockIdx.y / p.batch_inner, b1 = blockIdx.y - b0 * p.batch_inner;
    p.A += b0 * p.sA0 + b1 * p.sA1;
    p.W += b0 * p.sW0 + b1 * p.sW1;
    const int64_t co = b0 * p.sC0 + b1 * p.sC1;
    p.C = OUT_BF16 ? (void*)(reinterpret_cast<bf16_t*>(p.C) + co) : (void*)(reinterpret_cast<float*>(p.C) + co);
  }

  // ---- XCD-aware, bijective block -> tile remap ---------------------------------------------------------
  uint32_t vb;
  {
    const uint32_t nb = p.tiles_total, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const uint32_t tm = vb / p.tiles_n, tn = vb - tm * p.tiles_n;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = (int)tn * BN;

  // ---- per-lane source pointers for the LDS-DMA pieces this wave issues per stage -------------------------
  constexpr int CPR = BK / 8;                                   // 16-byte chunks per row
  const int piece_row = lane / CPR, pos = lane % CPR;           // LDS position (row, chunk slot) this lane fills
  const int gchunk = (BK == 64) ? (pos ^ (piece_row & 7)) : (pos ^ ((piece_row >> 2) & 3));   // source-side swizzle
  const bf16_t* a_src[Cfg::A_PIECES];
  const bf16_t* b_src[Cfg::B_PIECES];
#pragma unroll
  for (int i = 0; i < Cfg::A_PIECES; ++i) {
    int64_t ar = m0 + (wave * Cfg::A_PIECES + i) * Cfg::PIECE_ROWS + piece_row;
    if (ar > p.M - 1) ar = p.M - 1;                             // clamp: tail rows re-read the last valid row
    a_src[i] = p.A + ar * p.lda + gchunk * 8;
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_PIECES; ++i) {
    int br = n0 + (wave * Cfg::B_PIECES + i) * Cfg::PIECE_ROWS + piece_row;
    if (br > p.N - 1) br = p.N - 1;
    b_src[i] = p.W + (int64_t)br * p.ldw + gchunk * 8;
  }
  auto stage = [&](int s, int kt) {
    if ((SF_ABL & 2) && kt >= NS - 1) return;
    char* abase = smem + s * Cfg::STAGE + (wave * Cfg::A_PIECES) * 1024;
    char* bbase = smem + s * Cfg::STAGE + Cfg::A_BYTES + (wave * Cfg::B_PIECES) * 1024;
#pragma unroll
    for (int i = 0; i < Cfg::A_PIECES; ++i) glds16(a_src[i] + kt * BK, abase + i * 1024);
#pragma unroll
    for (int i = 0; i < Cfg::B_PIECES; ++i) glds16(b_src[i] + kt * BK, bbase + i * 1024);
  };

  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int fr = lane & 15, fg = lane >> 4;
  int a_off[BK / 32], b_off[BK / 32];                           // swizzled fragment offsets per 32-deep k-step
#pragma unroll
  for (int ks = 0; ks < BK / 32; ++ks) {
    a_off[ks] = lds_chunk_off<BK>(wm * Cfg::WT_M + fr, ks * 4 + fg);   // (+ i*16 rows keeps row&7 / (row>>2)&3)
    b_off[ks] = lds_chunk_off<BK>(wn * Cfg::WT_N + fr, ks * 4 + fg);
  }

  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  // ---- prologue: NS-1 stages in flight ---------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) stage(s, s);
  if constexpr (!Cfg::PIPE) {
    int cur = 0, fill = NS - 1;                                 // ring slots: being computed / next to refill
    for (int kt = 0; kt < nk; ++kt) {
      // wait until tile kt has landed: only the (<= NS-2) younger stages may still be outstanding
      const int ahead = min(NS - 2, nk - 1 - kt);
      if (NS >= 4 && ahead >= 2) wait_vmcnt_barrier<2 * P>();
      else if (NS >= 3 && ahead == 1) wait_vmcnt_barrier<P>();
      else wait_vmcnt_barrier<0>();
      // every wave has passed compute(kt-1): slot `fill` (== slot of tile kt-1) is free -> refill with tile kt+NS-1
      if (kt + NS - 1 < nk) stage(fill, kt + NS - 1);
      const char* sa = smem + cur * Cfg::STAGE;
      const char* sb = sa + Cfg::A_BYTES;
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 a[FI], b[FJ];
#pragma unroll
        for (int j = 0; j < FJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[ks] + j * 16 * Cfg::ROW_B);
#pragma unroll
        for (int i = 0; i < FI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sa + a_off[ks] + i * 16 * Cfg::ROW_B);
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
          for (int j = 0; j < FJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      fill = (fill + 1 == NS) ? 0 : fill + 1;
    }
  } else {
    // Fragment-pipelined loop (BK = 32: one k-step per stage).  Iteration kt multiplies the fragments of tile kt that
    // were read from LDS during iteration kt-1, while this iteration's ds_reads fetch tile kt+1: LDS latency hides
    // under the MFMAs instead of alternating with them (the two waves of a SIMD run in lockstep behind the barrier).
    // Barrier(kt) publishes tile kt+1 (each wave waited for its own DMA pieces) and retires every read of tile kt-1
    // (consumed by the MFMAs of iteration kt-1), whose slot is refilled with tile kt+NS-1.
    auto load_frags = [&](int slot, bf16x8 (&a)[FI], bf16x8 (&b)[FJ]) {
      const char* sa = smem + slot * Cfg::STAGE;
      const char* sb = sa + Cfg::A_BYTES;
#pragma unroll
      for (int j = 0; j < FJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[0] + j * 16 * Cfg::ROW_B);
#pragma unroll
      for (int i = 0; i < FI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sa + a_off[0] + i * 16 * Cfg::ROW_B);
    };
    auto mma = [&](const bf16x8 (&a)[FI], const bf16x8 (&b)[FJ]) {
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    auto sync_for = [&](int tile) {   // make `tile` visible: younger stages (<= NS-3 of them) may stay in flight
      const int ahead = min(NS - 3, nk - 1 - tile);
      if (ahead >= 1) wait_vmcnt_barrier<P>(); else wait_vmcnt_barrier<0>();
    };
    static_assert(NS == 4, "sync_for() is written for a 4-deep ring");
    bf16x8 a0[FI], b0[FJ], a1[FI], b1[FJ];
    // tile 0
    { const int ahead = min(NS - 2, nk - 1); if (ahead >= 2) wait_vmcnt_barrier<2 * P>(); else if (ahead == 1) wait_vmcnt_barrier<P>(); else wait_vmcnt_barrier<0>(); }
    load_frags(0, a0, b0);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {             // two tiles per trip so fragment buffers are statically named
      sync_for(kt + 1);
      if (kt + NS - 1 < nk) stage((kt + NS - 1) & (NS - 1), kt + NS - 1);
      load_frags((kt + 1) & (NS - 1), a1, b1);
      mma(a0, b0);
      if (kt + 2 < nk) {
        sync_for(kt + 2);
        if (kt + NS < nk) stage((kt + NS) & (NS - 1), kt + NS);
        load_frags((kt + 2) & (NS - 1), a0, b0);
      }
      mma(a1, b1);
    }
    if (kt < nk) mma(a0, b0);                  // odd tile count: last tile's fragments are already in a0/b0
  }
  __syncthreads();   // every wave is done reading operand tiles; LDS becomes per-wave epilogue scratch

  if (SF_ABL & 8) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 1.2345e30f) reinterpret_cast<float*>(p.C)[0] = sum;
    return;
  }
  // ---- epilogue ----------------------------------------------------------------------------------------------
  const int ecol = (lane & 15) * 4;                 // 4 consecutive output columns per lane
  const int gcol = n0 + wn * Cfg::WT_N + ecol;
  if constexpr (FAST) {
    // identity row maps, N % 64 == 0, buffers < 4 GiB: branch-free groups of 16 rows through a 16-row LDS slab
    if (n0 + wn * Cfg::WT_N < p.N) {               // wave-uniform (N % 64 == 0)
      float* slab = reinterpret_cast<float*>(smem + wave * (16 * EPI_LD * 4));
      const uint32_t esz = OUT_BF16 ? 2u : 4u;
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)(uint32_t)(p.M * p.ldc * esz), 0x00020000);
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), (short)0, HAS_RES ? (int)(uint32_t)(p.M * p.ldr * 4) : 0, 0x00020000);
      const int64_t row0 = m0 + wm * Cfg::WT_M + (lane >> 4);
      const uint32_t coff0 = (uint32_t)(row0 * p.ldc + gcol) * esz, cstep = (uint32_t)(4 * p.ldc) * esz;
      const uint32_t roff0 = (uint32_t)(row0 * p.ldr + gcol) * 4u, rstep = (uint32_t)(4 * p.ldr) * 4u;
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + gcol);
      float4 res[2][4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) res[0][ps] = res[1][ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS_RES) epi_group_load_res(res[0], rr, roff0, rstep);
#pragma unroll
      for (int g = 0; g < FI; ++g) {
        if (HAS_RES && g + 1 < FI) epi_group_load_res(res[(g + 1) & 1], rr, roff0 + (g + 1) * 4 * rstep, rstep);
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[(fg * 4 + r) * EPI_LD + j * 16 + fr] = acc[g][j][r];
        float4 v[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) v[ps] = *reinterpret_cast<const float4*>(slab + (ps * 4 + (lane >> 4)) * EPI_LD + ecol);
        if (!(SF_ABL & 1)) epi_group_store<OUT_BF16, GELU, HAS_RES>(v, bias4, res[g & 1], rc, coff0 + g * 4 * cstep, cstep);
        else if (v[0].x == 1.2345e30f) epi_group_store<OUT_BF16, GELU, HAS_RES>(v, bias4, res[g & 1], rc, coff0, cstep);
      }
    }
  } else {
    // general path (row maps, ragged N such as the 21-way / 2-way heads): per-row predicates, element-wise tails
    float* slab = reinterpret_cast<float*>(smem + wave * (32 * EPI_LD * 4));
    const bool vec = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && (!HAS_RES || (p.ldr & 3) == 0);
    float bias_e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) if (p.bias && gcol + e < p.N) bias_e[e] = p.bias[gcol + e];
#pragma unroll
    for (int half = 0; half < FI / 2; ++half) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[(ii * 16 + fg * 4 + r) * EPI_LD + j * 16 + fr] = acc[half * 2 + ii][j][r];
      for (int pass = 0; pass < 8; ++pass) {
        const int lrow = pass * 4 + (lane >> 4);
        const float4 t = *reinterpret_cast<const float4*>(slab + lrow * EPI_LD + ecol);
        const int64_t grow = m0 + wm * Cfg::WT_M + half * 32 + lrow;
        if (grow < p.M && gcol < p.N) {
          float vv[4] = {t.x + bias_e[0], t.y + bias_e[1], t.z + bias_e[2], t.w + bias_e[3]};
          if (GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = gelu_erf(vv[e]);
          }
          const int64_t crow = map_row(p.cmap, grow);
          const int64_t rrow = HAS_RES ? map_row(p.rmap, grow) : 0;
          if (vec && gcol + 3 < p.N) {
            if (HAS_RES) {
              const float4 rr4 = *reinterpret_cast<const float4*>(p.R + rrow * p.ldr + gcol);
              vv[0] += rr4.x; vv[1] += rr4.y; vv[2] += rr4.z; vv[3] += rr4.w;
            }
            if (OUT_BF16) {
              uint2 o; o.x = pack_bf2(vv[0], vv[1]); o.y = pack_bf2(vv[2], vv[3]);
              *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + gcol) = o;
            } else {
              *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + crow * p.ldc + gcol) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gcol + e < p.N) {
                float o = vv[e];
                if (HAS_RES) o += p.R[rrow * p.ldr + gcol + e];
                if (OUT_BF16) reinterpret_cast<bf16_t*>(p.C)[crow * p.ldc + gcol + e] = f2bf(o);
                else reinterpret_cast<float*>(p.C)[crow * p.ldc + gcol + e] = o;
              }
          }
        }
      }
    }
  }
}

// =========================================================================================================
// Persistent 256 x 256 x 64 kernel on v_mfma_f32_32x32x16_bf16 - the big token GEMMs.
// The ablation in profiles/r01_gemm_ablation.md shows a fixed ~7 us per output tile (workgroup launch, first-load
// latency, epilogue) on top of ~15 us of MFMA work at K = 768.  So: ONE workgroup per CU walks many tiles, and the
// first K-stage of the NEXT tile is put in flight (LDS-DMA into ring slot 0) before the current tile's epilogue
// runs (which stages through slot 1's memory) - launch cost is paid once, load latency hides under the epilogue.
// 32x32x16 fragments: same LDS bytes per FLOP as 16x16x32, 15 % higher MFMA peak (2.38 vs 2.08 PFLOP/s).
//   A fragment: lane l holds row (l & 31), k = (l >> 5) * 8 .. +7 of the 16-deep step;  B likewise with n.
//   C fragment: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
// LDS rows are 128 B; chunk c of row r lives at slot c ^ ((r >> 1) & 7): 16 consecutive rows (and the 16-lane
// groups ds_read_b128 is served in, for 32-row fragments) hit 16 distinct 16-byte slots of the 256-B bank row.
// =========================================================================================================
#define PBM 256
#define PBN 256
#define PBK 64
#define P_STAGE (2 * PBM * PBK * 2)       // 64 KiB: A tile + B tile
#define P_EPI_LD 64                       // unpadded slab rows: conflict-free for the 32x32 C layout (32 consecutive lanes = 32 consecutive columns)
#define P_SLAB_ROWS 16
#define P_SLAB_BYTES (P_SLAB_ROWS * P_EPI_LD * 4)   // 4 KiB per wave
#define P_LDS (2 * P_STAGE + 8 * P_SLAB_BYTES)      // 160 KiB: two ring slots + a private epilogue region, so BOTH slots of the
                                                    // next tile can be in flight while this tile's epilogue runs

template <bool OUT_BF16, bool GELU, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void gemm_bf16_persistent_kernel(GemmArgs p) {
  // bf16 output without residual (qkv, fc1 + GELU): the accumulator blocks are computed TRANSPOSED (operands swapped), so a lane holds four
  // consecutive features of one token per register group; bias / GELU are applied in registers, the bf16 results go to the wave's slab as 8-byte
  // writes (32 instead of 128 ds_write_b32 per tile) and leave as 16-byte row segments (16 dwordx4 stores instead of 32 dwordx2 - the store tail
  // of a bf16 epilogue is store-ISSUE bound, MI355X_MICROARCH.md)
  constexpr bool WIDE = OUT_BF16 && !HAS_RES && SF_EPI_WIDE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;                      // 2 x 4 waves, wave tile 128 x 64
  const int l31 = lane & 31, hi = lane >> 5;

  // persistent schedule: block b sits on XCD b % 8; every XCD owns a contiguous tile range, N fastest
  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  // every XCD owns a contiguous range of 256-row panels and sweeps it once per CHUNK of `nchunk` column tiles (N fastest inside
  // a chunk): the weight slice of a chunk (nchunk * 256 rows of W) stays L2-resident for the whole sweep
  const uint32_t tiles_m = p.tiles_total / p.tiles_n;
  const uint32_t mp8 = (tiles_m + 7u) >> 3;
  const uint32_t mp0 = min(xcd * mp8, tiles_m), mp1 = min(mp0 + mp8, tiles_m), n_mp = mp1 - mp0;
  const uint32_t gchunk = p.nchunk ? min(p.nchunk, p.tiles_n) : p.tiles_n;
  const uint32_t n_chunks = (p.tiles_n + gchunk - 1) / gchunk, chunk_tiles = n_mp * gchunk;
  const uint32_t t_begin = 0, t_end = n_mp * p.tiles_n;           // local tile index inside this XCD's range

  const int piece_row = lane >> 3, slot = lane & 7;
  // fragment read offsets (bytes) inside a tile for k-step kk: row-dependent swizzle is lane-constant
  const int sw = (l31 >> 1) & 7;
  int frag_off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) frag_off[kk] = l31 * 128 + (((kk * 2 + hi) ^ sw) << 4);
  const int a_base = wm * 128 * 128, b_base = PBM * PBK * 2 + wn * 64 * 128;

  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
  auto set_tile = [&](uint32_t t, int64_t& m0, int& n0) {
    const uint32_t c = min(t / chunk_tiles, n_chunks - 1), r = t - c * chunk_tiles;
    const uint32_t gw = (c == n_chunks - 1) ? p.tiles_n - c * gchunk : gchunk;
    const uint32_t tm = mp0 + r / gw, tn = c * gchunk + r % gw;
    m0 = (int64_t)tm * PBM; n0 = (int)tn * PBN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 8 + piece_row;            // tile row this lane fills
      const int gch = slot ^ ((row >> 1) & 7);                   // source-side swizzle
      int64_t ar = m0 + row; if (ar > p.M - 1) ar = p.M - 1;
      int br = n0 + row; if (br > p.N - 1) br = p.N - 1;
      a_src[i] = p.A + ar * p.lda + gch * 8;
      b_src[i] = p.W + (int64_t)br * p.ldw + gch * 8;
    }
  };
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(lds_addr(smem) + (wave * 4) * 1024);
  // k-loop rotation: the 32 workgroups of an XCD run in near lockstep (same tile shape, same start), so without it they all ask the XCD's L2
  // for the SAME operand lines at the same moment (the A panel shared by the column tiles of a row panel, the W rows shared by everything) and
  // queue on those lines' channels.  Starting every workgroup at a different k-tile spreads the requests over the whole k-extent of the
  // operands; the sum over k is the same set of products in a rotated order (fp32 accumulation: last-bit differences between tilings).
  const int nk = p.K / PBK;
  const int krot = SF_KROT ? (int)((li * (uint32_t)SF_KROT) % (uint32_t)nk) : 0;
  auto kmap = [&](int kt) { const int k = kt + krot; return k >= nk ? k - nk : k; };
  auto stage = [&](int s, int kt_) {
    const int kt = kmap(kt_);
    const uint32_t l = lds_wave + s * P_STAGE;
    if (SF_A_NT) dma4_nt(a_src[0] + kt * PBK, a_src[1] + kt * PBK, a_src[2] + kt * PBK, a_src[3] + kt * PBK, l);
    else dma4(a_src[0] + kt * PBK, a_src[1] + kt * PBK, a_src[2] + kt * PBK, a_src[3] + kt * PBK, l);
    dma4(b_src[0] + kt * p.wk, b_src[1] + kt * p.wk, b_src[2] + kt * p.wk, b_src[3] + kt * p.wk, l + PBM * PBK * 2);
  };

  uint32_t t = t_begin + li;
  if (t >= t_end) return;
  int64_t m0; int n0;
  set_tile(t, m0, n0);
  stage(0, 0);
  float* slab = reinterpret_cast<float*>(smem + 2 * P_STAGE + wave * P_SLAB_BYTES);
  const int ecol = (lane & 15) * 4;
  const uint32_t esz = OUT_BF16 ? 2u : 4u;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)(uint32_t)(p.M * p.ldc * esz), 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), (short)0, HAS_RES ? (int)(uint32_t)(p.M * p.ldr * 4) : 0, 0x00020000);
  const uint32_t cstep = (uint32_t)(4 * p.ldc) * esz, rstep = (uint32_t)(4 * p.ldr) * 4u;
  bool stage1_in_flight = false;
  if (nk > 1) { stage(1, 1); stage1_in_flight = true; }

  for (;;) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // WIDE: the wave's 64 bias values as ONE load (lanes 0-15) here, at the top of the tile, handed round through the slab in the epilogue.  Loaded in
    // the epilogue (8 float4 per lane) they sat BEHIND the next tile's 16 LDS-DMA pieces in the in-order vector-memory queue: the epilogue's first
    // use waited for both prefetched stages to land.
    float4 bias_raw = make_float4(0.f, 0.f, 0.f, 0.f);
    if (WIDE && p.bias && lane < 16 && n0 + wn * 64 < p.N) bias_raw = *reinterpret_cast<const float4*>(p.bias + n0 + wn * 64 + lane * 4);

    for (int kt = 0; kt < nk; ++kt) {
      wait_vmcnt_barrier<0>();                                   // tile kt landed everywhere; slot (kt+1)&1 is free
      const bool refill = kt + 1 < nk && !(kt == 0 && stage1_in_flight);
      if (!SF_DMA_SPREAD && refill) stage((kt + 1) & 1, kt + 1);
      const char* sa = smem + (kt & 1) * P_STAGE + a_base;
      const char* sb = smem + (kt & 1) * P_STAGE + b_base;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 a[4], b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * 128 + frag_off[kk]);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sa + i * 32 * 128 + frag_off[kk]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = WIDE ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0)      // C^T block: lanes = tokens, registers = features
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (SF_DMA_SPREAD && refill && kk < 2) {                 // the refill's 8 LDS-DMA issues ride behind the first two MFMA clusters
          __builtin_amdgcn_sched_barrier(0);
          const uint32_t l = lds_wave + ((kt + 1) & 1) * P_STAGE;
          const int ko = kmap(kt + 1) * PBK;
          const int64_t kow = kmap(kt + 1) * p.wk;
          if (kk == 0) dma4(a_src[0] + ko, a_src[1] + ko, a_src[2] + ko, a_src[3] + ko, l);
          else dma4(b_src[0] + kow, b_src[1] + kow, b_src[2] + kow, b_src[3] + kow, l + PBM * PBK * 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // all waves finished reading both slots -> slot 0 can take the next tile's first stage, slot 1 is epilogue scratch
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int64_t em0 = m0; const int en0 = n0;
    const uint32_t tnext = t + per_xcd_blocks;
    const bool more = tnext < t_end;
    stage1_in_flight = false;
    if (more) { set_tile(tnext, m0, n0); stage(0, 0); if (nk > 1) { stage(1, 1); stage1_in_flight = true; } }

    if (WIDE) {
      // ---- wide bf16 epilogue: 4 passes of 32 tokens x 64 features through the wave's 4 KiB slab (rows of 128 B, 16-byte chunk c of row t at slot
      // c ^ (t & 7), the two 8-byte halves of a chunk swapped in rows with bit 3 set: conflict-free for the 8-byte writes and the 16-byte reads) ----
      int etid = threadIdx.x;
      asm volatile("" : "+v"(etid));                              // lane-derived epilogue values must not be hoisted across the k-loop
      const int el = etid & 63, el31 = el & 31, ehi = el >> 5;
      if (en0 + wn * 64 < p.N) {                                  // wave-uniform
        char* bslab = reinterpret_cast<char*>(slab);
        const int colbase = en0 + wn * 64;
        if (el < 16) *reinterpret_cast<float4*>(bslab + el * 16) = bias_raw;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 bia[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) bia[j][g] = *reinterpret_cast<const float4*>(bslab + (j * 32 + g * 8 + ehi * 4) * 4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int wr_off = el31 * 128 + ((ehi ^ ((el31 >> 3) & 1)) << 3), sw7 = el31 & 7;
        const int tr0 = el >> 3, ch = el & 7;
        const int rd_off = tr0 * 128 + ((ch ^ (tr0 & 7)) << 4);
        const uint32_t cbase = (uint32_t)((em0 + wm * 128 + tr0) * p.ldc + colbase + ch * 8) * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
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
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(bslab + rd_off + rr * 8 * 128);
            u32x4 o;
            if (rr & 1) { o.x = v.z; o.y = v.w; o.z = v.x; o.w = v.y; } else { o = v; }
            if (!(SF_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(o, rc, cbase + (uint32_t)((i * 32 + rr * 8) * p.ldc) * 2u, 0, SF_EPI_STORE_AUX);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    } else
    // ---- epilogue of tile (em0, en0): 8 branch-free groups of 16 rows x 64 cols through this wave's slab -------
    if (en0 + wn * 64 < p.N) {                                   // wave-uniform (N % 64 == 0 on this path)
      const int gcol = en0 + wn * 64 + ecol;
      const int64_t row0 = em0 + wm * 128 + (lane >> 4);
      const uint32_t coff0 = (uint32_t)(row0 * p.ldc + gcol) * esz, roff0 = (uint32_t)(row0 * p.ldr + gcol) * 4u;
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + gcol);
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
              slab[(qq * 8 + hi * 4 + r) * P_EPI_LD + j * 32 + l31] = acc[i][j][(q2 * 2 + qq) * 4 + r];
        float4 v[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) v[ps] = *reinterpret_cast<const float4*>(slab + (ps * 4 + (lane >> 4)) * P_EPI_LD + ecol);
        if (!(SF_ABL & 1)) epi_group_store<OUT_BF16, GELU, HAS_RES>(v, bias4, res[g & 1], rc, coff0 + g * 4 * cstep, cstep);
        else if (v[0].x == 1.2345e30f) epi_group_store<OUT_BF16, GELU, HAS_RES>(v, bias4, res[g & 1], rc, coff0, cstep);
      }
    }
    if (!more) break;
    t = tnext;
  }
}

template <bool OUT_BF16, bool GELU, bool HAS_RES>
static int launch_gemm_persistent(GemmArgs a, hipStream_t s) {
  auto kern = gemm_bf16_persistent_kernel<OUT_BF16, GELU, HAS_RES>;
  if (int rc = sf_prepare_kernel((const void*)kern, P_LDS, "sf_gemm_bf16")) return rc;
  const int n_cu = sf_cu_count("sf_gemm_bf16");
  if (n_cu <= 0) return -1;
  const int64_t tiles_m = (a.M + PBM - 1) / PBM;
  a.tiles_n = (uint32_t)((a.N + PBN - 1) / PBN);
  const int64_t total = tiles_m * a.tiles_n;
  if (total >= ((int64_t)1 << 31)) { sf_set_error("sf_gemm_bf16: too many tiles"); return -1; }
  a.tiles_total = (uint32_t)total;
  // Column-chunked sweeps keep the weight slice of a sweep (nchunk * 256 * K bf16) within ~2.4 MB of the XCD's 4 MB L2, so that the
  // streamed A panels and outputs stop evicting it: PMC on M = 351,456 - qkv 1912 -> 1132 MiB fetched per launch (518 algorithmic),
  // fc1 5604 -> 1167, and +2-4 % speed (profiles/r01_gemm_configs.md).  Only for short K: with K = 3072 an A panel is 1.5 MB and
  // re-reading it per sweep costs more than it saves (fc2 already fetches its algorithmic bytes).  SF_GEMM_NCHUNK overrides (experiments).
  static int env_chunk = -2;
  if (env_chunk == -2) { const char* e = getenv("SF_GEMM_NCHUNK"); env_chunk = e ? atoi(e) : -1; }
  if (env_chunk >= 0) a.nchunk = (uint32_t)env_chunk;
  else a.nchunk = a.K <= 1024 ? (uint32_t)(2400000 / (512 * a.K) > 0 ? 2400000 / (512 * a.K) : 1) : 0u;
  int64_t blocks = (n_cu / 8) * 8;                               // one workgroup per CU, a multiple of the 8 XCDs
  if (blocks < 8) blocks = 8;                                    // (a device / partition with fewer than 8 CUs: never an empty grid)
  const int64_t need = ((total + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), P_LDS, s, a);
  SF_LAUNCH_CHECK();
  return 0;
}

static int dispatch_gemm_persistent(const GemmArgs& a, bool out_bf16, bool gelu, bool res, hipStream_t s) {
  if (out_bf16) {
    if (gelu) return res ? launch_gemm_persistent<true, true, true>(a, s) : launch_gemm_persistent<true, true, false>(a, s);
    return res ? launch_gemm_persistent<true, false, true>(a, s) : launch_gemm_persistent<true, false, false>(a, s);
  }
  if (gelu) return res ? launch_gemm_persistent<false, true, true>(a, s) : launch_gemm_persistent<false, true, false>(a, s);
  return res ? launch_gemm_persistent<false, false, true>(a, s) : launch_gemm_persistent<false, false, false>(a, s);
}

#ifdef SF_ABLATION   // measured-slower alternatives live in the ablation build only (synchformer_amd/build.py::build_ablation, tests load it explicitly)
// =========================================================================================================
// Config 10: the same persistent 256 x 256 x 64 tile on FOUR waves (one per SIMD), 128 x 128 accumulators each (256 registers: the
// accumulator half of the unified file) - the wave shape the vendor library uses on these shapes.  Per k-tile a wave reads (128 + 128) x 64
// operand elements from LDS where the 8-wave kernel's waves read (128 + 64) x 64 each: 128 KiB instead of 196 KiB of fragment traffic per CU
// and k-tile - the 8-wave kernel's LDS pipe is as busy as its matrix pipe (profiles/r01_gemm_configs.md).  With one wave per SIMD nothing
// covers a stall, so the loop is software-pipelined by hand and PINNED with sched_barrier(0) fences:
//   * fragments are double-buffered per 16-deep step: while the 16 MFMAs of step kk run, the 8 fragment reads of step kk + 1 are issued, one
//     behind every second MFMA;
//   * the k-tile barrier sits in front of the LAST step of a k-tile: its fragments are already in registers, so the slot is free from there on
//     (refilled with k-tile kt + 2 by 16 LDS-DMA pieces per wave, one behind each MFMA of that step) and the first fragments of k-tile kt + 1
//     are read under those MFMAs.
// =========================================================================================================
#define W4_SLAB_BYTES (16 * P_EPI_LD * 4)
#define W4_LDS (2 * P_STAGE + 4 * W4_SLAB_BYTES)

template <bool OUT_BF16, bool GELU, bool HAS_RES>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;                      // 2 x 2 waves, wave tile 128 x 128
  const int l31 = lane & 31, hi = lane >> 5;

  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t tiles_m = p.tiles_total / p.tiles_n;
  const uint32_t mp8 = (tiles_m + 7u) >> 3;
  const uint32_t mp0 = min(xcd * mp8, tiles_m), mp1 = min(mp0 + mp8, tiles_m), n_mp = mp1 - mp0;
  const uint32_t gchunk = p.nchunk ? min(p.nchunk, p.tiles_n) : p.tiles_n;
  const uint32_t n_chunks = (p.tiles_n + gchunk - 1) / gchunk, chunk_tiles = n_mp * gchunk;
  const uint32_t t_end = n_mp * p.tiles_n;

  const int piece_row = lane >> 3, slot = lane & 7;
  const int sw = (l31 >> 1) & 7;
  int frag_off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) frag_off[kk] = l31 * 128 + (((kk * 2 + hi) ^ sw) << 4);
  const int a_base = wm * 128 * 128, b_base = PBM * PBK * 2 + wn * 128 * 128;

  uint32_t a_src[8], b_src[8];                                  // byte offsets from p.A / p.W (both below 4 GiB: checked by the launcher)
  auto set_tile = [&](uint32_t t, int64_t& m0, int& n0) {
    const uint32_t c = min(t / chunk_tiles, n_chunks - 1), r = t - c * chunk_tiles;
    const uint32_t gw = (c == n_chunks - 1) ? p.tiles_n - c * gchunk : gchunk;
    const uint32_t tm = mp0 + r / gw, tn = c * gchunk + r % gw;
    m0 = (int64_t)tm * PBM; n0 = (int)tn * PBN;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (wave * 8 + i) * 8 + piece_row;            // tile row this lane fills (8 pieces of A and of B per wave)
      const int gch = slot ^ ((row >> 1) & 7);
      int64_t ar = m0 + row; if (ar > p.M - 1) ar = p.M - 1;
      int br = n0 + row; if (br > p.N - 1) br = p.N - 1;
      a_src[i] = (uint32_t)((ar * p.lda + gch * 8) * 2);
      b_src[i] = (uint32_t)(((int64_t)br * p.ldw + gch * 8) * 2);
    }
  };
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(lds_addr(smem) + (wave * 8) * 1024);
  auto dma_quad = [&](int s, int kt, int q) {                     // q = 0, 1: A pieces 0-3 / 4-7;  q = 2, 3: B pieces 0-3 / 4-7
    const uint32_t l = lds_wave + s * P_STAGE + (q >= 2 ? PBM * PBK * 2 : 0) + (q & 1) * 4096;
    const uint32_t ko = (uint32_t)kt * (PBK * 2);
    const uint32_t* src = q >= 2 ? b_src : a_src;
    const int o = (q & 1) * 4;
    dma4s(src[o] + ko, src[o + 1] + ko, src[o + 2] + ko, src[o + 3] + ko, q >= 2 ? (const void*)p.W : (const void*)p.A, l);
  };
  auto stage = [&](int s, int kt) { dma_quad(s, kt, 0); dma_quad(s, kt, 1); dma_quad(s, kt, 2); dma_quad(s, kt, 3); };

  const int nk = p.K / PBK;
  uint32_t t = li;
  if (t >= t_end) return;
  int64_t m0; int n0;
  set_tile(t, m0, n0);
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  float* slab = reinterpret_cast<float*>(smem + 2 * P_STAGE + wave * W4_SLAB_BYTES);
  const int ecol = (lane & 15) * 4;
  const uint32_t esz = OUT_BF16 ? 2u : 4u;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)(uint32_t)(p.M * p.ldc * esz), 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), (short)0, HAS_RES ? (int)(uint32_t)(p.M * p.ldr * 4) : 0, 0x00020000);
  const uint32_t cstep = (uint32_t)(4 * p.ldc) * esz, rstep = (uint32_t)(4 * p.ldr) * 4u;

  for (;;) {
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8 f0[8], f1[8];                                          // fragment sets: [0..3] = A row blocks, [4..7] = B column blocks
    auto read_frag = [&](bf16x8& dst, const char* st, int idx, int kk) {
      const int base = idx < 4 ? a_base + idx * 32 * 128 : b_base + (idx - 4) * 32 * 128;
      dst = *reinterpret_cast<const bf16x8*>(st + base + frag_off[kk]);
    };
    // one 16-deep step: 16 MFMAs on `cur`; behind every second one a fragment of (nst, nkk) goes into `nxt`; DMA: 0 = none, else the quad
    // of k-tile `dkt` for slot `dslot` behind MFMAs 1, 5, 9, 13
    auto step = [&](bf16x8 (&cur)[8], bf16x8 (&nxt)[8], const char* nst, int nkk, bool do_reads, bool do_dma, int dslot, int dkt) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int m0_ = 2 * b, m1_ = 2 * b + 1;
        if (!(SF_ABL & 4)) {
          acc[m0_ >> 2][m0_ & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m0_ >> 2], cur[4 + (m0_ & 3)], acc[m0_ >> 2][m0_ & 3], 0, 0, 0);
          acc[m1_ >> 2][m1_ & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m1_ >> 2], cur[4 + (m1_ & 3)], acc[m1_ >> 2][m1_ & 3], 0, 0, 0);
        } else if (b == 0) {
          acc[0][0][0] += (float)cur[0][0] + (float)cur[7][0];       // ablation: keep the fragments live, no matrix work
        }
        if (do_reads) read_frag(nxt[b], nst, b, nkk);
        if (do_dma && (b & 1) == 0 && !(SF_ABL & 2)) dma_quad(dslot, dkt, b >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    wait_vmcnt_barrier<0>();                                      // k-tile 0 of this tile is visible
#pragma unroll
    for (int b = 0; b < 8; ++b) read_frag(f0[b], smem, b, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const char* st = smem + (kt & 1) * P_STAGE;
      const char* stn = smem + ((kt + 1) & 1) * P_STAGE;
      step(f0, f1, st, 1, true, false, 0, 0);
      step(f1, f0, st, 2, true, false, 0, 0);
      step(f0, f1, st, 3, true, false, 0, 0);
      const bool more_k = kt + 1 < nk;
      if (more_k) {
        // every fragment of this k-tile is in registers once f1 has landed: publish k-tile kt + 1 and free this slot
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt_barrier<0>();
      }
      step(f1, f0, stn, 0, more_k, kt + 2 < nk, kt & 1, kt + 2);
    }
    // all waves are done with both slots -> next tile's first two k-tiles go in flight under the epilogue
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int64_t em0 = m0; const int en0 = n0;
    const uint32_t tnext = t + per_xcd_blocks;
    const bool more = tnext < t_end;
    if (more) { set_tile(tnext, m0, n0); stage(0, 0); if (nk > 1) stage(1, 1); }

    // ---- epilogue: 8 row groups x 2 column halves of (16 rows x 64 cols) through this wave's slab -------------------
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      if (en0 + wn * 128 + ch * 64 < p.N) {
        const int gcol = en0 + wn * 128 + ch * 64 + ecol;
        const int64_t row0 = em0 + wm * 128 + (lane >> 4);
        const uint32_t coff0 = (uint32_t)(row0 * p.ldc + gcol) * esz, roff0 = (uint32_t)(row0 * p.ldr + gcol) * 4u;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + gcol);
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
                slab[(qq * 8 + hi * 4 + r) * P_EPI_LD + j * 32 + l31] = acc[i][ch * 2 + j][(q2 * 2 + qq) * 4 + r];
          float4 v[4];
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) v[ps] = *reinterpret_cast<const float4*>(slab + (ps * 4 + (lane >> 4)) * P_EPI_LD + ecol);
          epi_group_store<OUT_BF16, GELU, HAS_RES>(v, bias4, res[g & 1], rc, coff0 + g * 4 * cstep, cstep);
        }
      }
    }
    if (!more) break;
    t = tnext;
  }
}

template <bool OUT_BF16, bool GELU, bool HAS_RES>
static int launch_gemm_w4(GemmArgs a, hipStream_t s) {
  auto kern = gemm_bf16_w4_kernel<OUT_BF16, GELU, HAS_RES>;
  if (int rc = sf_prepare_kernel((const void*)kern, W4_LDS, "sf_gemm_bf16")) return rc;
  const int n_cu = sf_cu_count("sf_gemm_bf16");
  if (n_cu <= 0) return -1;
  const int64_t tiles_m = (a.M + PBM - 1) / PBM;
  a.tiles_n = (uint32_t)((a.N + PBN - 1) / PBN);
  const int64_t total = tiles_m * a.tiles_n;
  if (total >= ((int64_t)1 << 31)) { sf_set_error("sf_gemm_bf16: too many tiles"); return -1; }
  a.tiles_total = (uint32_t)total;
  a.nchunk = a.K <= 1024 ? (uint32_t)(2400000 / (512 * a.K) > 0 ? 2400000 / (512 * a.K) : 1) : 0u;
  int64_t blocks = (n_cu / 8) * 8;
  if (blocks < 8) blocks = 8;                                    // (a device / partition with fewer than 8 CUs: never an empty grid)
  const int64_t need = ((total + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), W4_LDS, s, a);
  SF_LAUNCH_CHECK();
  return 0;
}

static int dispatch_gemm_w4(const GemmArgs& a, bool out_bf16, bool gelu, bool res, hipStream_t s) {
  if (out_bf16) {
    if (gelu) return res ? launch_gemm_w4<true, true, true>(a, s) : launch_gemm_w4<true, true, false>(a, s);
    return res ? launch_gemm_w4<true, false, true>(a, s) : launch_gemm_w4<true, false, false>(a, s);
  }
  if (gelu) return res ? launch_gemm_w4<false, true, true>(a, s) : launch_gemm_w4<false, true, false>(a, s);
  return res ? launch_gemm_w4<false, false, true>(a, s) : launch_gemm_w4<false, false, false>(a, s);
}

#endif  // SF_ABLATION

//                 BM   BN  WM WN BK NS wg/CU
typedef GemmCfg<128, 128, 2, 2, 64, 2, 2> Cfg0;   // 4 waves,  64 KiB: small-M GEMMs (AST, aggregators, sync, heads)
typedef GemmCfg<256, 256, 2, 4, 64, 2, 1> Cfg1;   // 8 waves, 128 KiB, 2-stage
typedef GemmCfg<256, 256, 2, 4, 32, 4, 1> Cfg2;   // 8 waves, 128 KiB, 4-stage ring of 32-deep steps (3 in flight)
typedef GemmCfg<256, 128, 4, 2, 64, 3, 1> Cfg3;   // 8 waves, 144 KiB, 3-stage ring of 64-deep steps (2 in flight)
typedef GemmCfg<128, 128, 2, 2, 32, 4, 2> Cfg4;   // 4 waves,  64 KiB, 4-stage ring, 2 workgroups/CU
typedef GemmCfg<256, 256, 2, 4, 32, 4, 1, true> Cfg5;   // Cfg2 + fragment register double-buffering
typedef GemmCfg<128, 128, 2, 2, 32, 4, 2, true> Cfg6;   // Cfg4 + fragment register double-buffering
typedef GemmCfg<256, 128, 2, 2, 32, 3, 2> Cfg8;   // 4 waves x (128 x 64), 72 KiB, 2 workgroups/CU: epilogue overlaps the other WG's loop
typedef GemmCfg<128, 256, 1, 4, 32, 3, 2> Cfg9;   // same, transposed block shape

static thread_local int g_force_cfg = -1;   // tuning / test hook: per calling thread, so that a benchmark thread cannot change another thread's launches
static thread_local int64_t g_batch_count = 1;   // set by sf_gemm_bf16_batched around its launch
extern "C" void sf_gemm_force_config(int cfg) { g_force_cfg = cfg; }

template <class Cfg, bool OUT_BF16, bool GELU, bool HAS_RES, bool FAST>
static int launch_gemm(GemmArgs a, hipStream_t s) {
  auto kern = gemm_bf16_kernel<Cfg, OUT_BF16, GELU, HAS_RES, FAST>;
  if (int rc = sf_prepare_kernel((const void*)kern, Cfg::LDS, "sf_gemm_bf16")) return rc;
  const int64_t tiles_m = (a.M + Cfg::BM - 1) / Cfg::BM;
  a.tiles_n = (uint32_t)((a.N + Cfg::BN - 1) / Cfg::BN);
  const int64_t total = tiles_m * a.tiles_n;
  if (total >= ((int64_t)1 << 31)) { sf_set_error("sf_gemm_bf16: too many tiles"); return -1; }
  a.tiles_total = (uint32_t)total;
  hipLaunchKernelGGL(kern, dim3(a.tiles_total, a.batch_inner > 0 ? (unsigned)g_batch_count : 1u), dim3(Cfg::THREADS), Cfg::LDS, s, a);
  SF_LAUNCH_CHECK();
  return 0;
}

template <class Cfg>
static int dispatch_gemm(const GemmArgs& a, bool out_bf16, bool gelu, bool res, bool fast, hipStream_t s) {
  if (!fast) {   // general path (row maps / ragged N): the variants the model needs
    if (gelu) { sf_set_error("sf_gemm_bf16: the GELU epilogue needs identity row maps and N %% 64 == 0"); return -1; }
    if (out_bf16) return res ? launch_gemm<Cfg, true, false, true, false>(a, s) : launch_gemm<Cfg, true, false, false, false>(a, s);
    return res ? launch_gemm<Cfg, false, false, true, false>(a, s) : launch_gemm<Cfg, false, false, false, false>(a, s);
  }
  if (out_bf16) {
    if (gelu) return res ? launch_gemm<Cfg, true, true, true, true>(a, s) : launch_gemm<Cfg, true, true, false, true>(a, s);
    return res ? launch_gemm<Cfg, true, false, true, true>(a, s) : launch_gemm<Cfg, true, false, false, true>(a, s);
  }
  if (gelu) return res ? launch_gemm<Cfg, false, true, true, true>(a, s) : launch_gemm<Cfg, false, true, false, true>(a, s);
  return res ? launch_gemm<Cfg, false, false, true, true>(a, s) : launch_gemm<Cfg, false, false, false, true>(a, s);
}

// The automatic tile-configuration choice of sf_gemm_bf16 (-1 = no configuration serves these arguments: a k-tile-major weight outside the persistent kernels' range).
static int gemm_auto_config(int64_t M, int64_t N, int64_t K, bool fast, bool res, bool w_kmajor, bool pp_ok) {
  // Measured on MI355X (tools/bench_gemm.py, profiles/r01_gemm_configs.md): the persistent 256x256 kernel wins on the
  // big token GEMMs; the short-K fp32-residual projection (N = K = 768) is HBM-bound and prefers 2 workgroups/CU;
  // everything small (AST, aggregators, sync transformer, heads) and every mapped/ragged GEMM takes the 128x128 kernel.
  const bool big = fast && M >= 8192 && N >= 512;
  int cfg = 0;
  if (w_kmajor) { if (!big) return -1; cfg = 7; }
  const bool pp = pp_ok;                       // round 3: the quadrant-phased kernel replaces config 7 wherever K is a whole number of k-tile pairs
  if (!w_kmajor && big && !(res && K <= 1024)) {
    // Tile-round quantisation decides between the persistent 256x256 kernel (one workgroup per CU, ~8 % faster per tile pair when
    // the chip is full) and the 128x128 kernel (two per CU): e.g. fc2 of a single clip is 86 x 3 = 258 big tiles = TWO rounds of
    // 256 CUs at 50 % fill, but 1032 small tiles = 2.02 rounds of 512 slots.  Pick the better filled one (measured, M = 21,966 /
    // 43,932: fc2 600 -> 803 / 766 -> 917 TFLOP/s; the 16-clip batch keeps the persistent kernel everywhere).
    int n_cu = sf_cu_count("sf_gemm_bf16");
    if (n_cu <= 0) n_cu = 256;
    const double t7 = (double)((M + 255) / 256) * (double)((N + 255) / 256), t0 = (double)((M + 127) / 128) * (double)((N + 127) / 128);
    const double r7 = (double)(int64_t)((t7 + n_cu - 1) / n_cu), r0 = (double)(int64_t)((t0 + 2 * n_cu - 1) / (2 * n_cu));
    const double e7 = t7 / (r7 * n_cu), e0 = t0 / (r0 * 2 * n_cu);
    // (config 11 is another ~12 % ahead of config 7 per filled round: tools/bench_gemm.py 14 28 56.  Two Stage-1 clips, N = 768, are the tie - 516 big tiles
    // against 2064 small ones, fill 0.672 vs 0.806 = 1 : 1.2: measured there config 11 wins without a residual - dproj 58 vs 63 us, dqkv 160 vs 165, dfc1 206
    // vs 217 - and loses with the fp32 residual epilogue, fc2 237 vs 228)
    cfg = (e7 * (pp ? (res ? 1.18 : 1.24) : 1.08) >= e0) ? 7 : 0;
  }
  if (cfg == 7 && pp) cfg = 11;
  return cfg;
}

extern "C" int sf_gemm_bf16(const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, void* C,
                            int c_dtype, int64_t ldc, const int64_t* c_map, const float* R, int64_t ldr,
                            const int64_t* r_map, int epilogue, int64_t M, int64_t N, int64_t K, void* stream) {
  SF_CHECK_ARG(A && W && C, "sf_gemm_bf16: null pointer");
  SF_CHECK_ARG(c_dtype == SF_BF16 || c_dtype == SF_F32, "sf_gemm_bf16: c_dtype must be bf16 or f32");
  SF_CHECK_ARG(epilogue == SF_EPI_NONE || epilogue == SF_EPI_GELU, "sf_gemm_bf16: bad epilogue %d", epilogue);
  SF_CHECK_ARG(K > 0 && (K % 64) == 0, "sf_gemm_bf16: K=%lld must be a positive multiple of 64", (long long)K);
  SF_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0, "sf_gemm_bf16: lda/ldw must be multiples of 8 elements (16 B)");
  // ldw == 64 with K > 64: the weight is given k-tile-major, [K/64][N][64] (the 32 KiB slice of every 64-deep k-tile of a 256-row W tile is
  // contiguous); only the persistent 256x256 kernel reads that layout
  const bool w_kmajor = ldw == 64 && K > 64;
  SF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "sf_gemm_bf16: A/W must be 16-byte aligned");
  SF_CHECK_ARG(M < ((int64_t)1 << 31) && N < ((int64_t)1 << 31), "sf_gemm_bf16: M, N must be < 2^31");
  if (M <= 0 || N <= 0) return 0;
  // FAST epilogue: identity row maps, whole 64-column wave tiles, 16-byte aligned rows, and every byte offset of the
  // (tile-padded) output / residual below 4 GiB (32-bit buffer offsets; rows >= M are dropped by the range check).
  const int64_t m_pad = ((M + 255) / 256) * 256;
  const bool fast = !c_map && !r_map && (N % 64) == 0 && (ldc % 4) == 0 && (!R || (ldr % 4) == 0) &&
                    ((uintptr_t)C % 16) == 0 && (!R || ((uintptr_t)R % 16) == 0) && (!bias || ((uintptr_t)bias % 16) == 0) &&
                    m_pad * ldc * (c_dtype == SF_BF16 ? 2 : 4) < ((int64_t)1 << 32) && (!R || m_pad * ldr * 4 < ((int64_t)1 << 32));
  GemmArgs a;
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.C = C; a.ldc = ldc; a.R = R; a.ldr = ldr;
  a.cmap = sf_rowmap(c_map); a.rmap = sf_rowmap(r_map);
  a.M = M; a.N = (int)N; a.K = (int)K;
  a.tiles_n = 0; a.tiles_total = 0; a.nchunk = 0; a.wk = w_kmajor ? N * 64 : 64;
  a.batch_inner = 0; a.sA0 = a.sA1 = a.sW0 = a.sW1 = a.sC0 = a.sC1 = 0;
  hipStream_t s = (hipStream_t)stream;
  const bool gelu = epilogue == SF_EPI_GELU, res = R != nullptr, obf = c_dtype == SF_BF16;
  int cfg = g_force_cfg;
  if (cfg < 0) {
    cfg = gemm_auto_config(M, N, K, fast, res, w_kmajor, sf_gemm_pp_supported(a));
    SF_CHECK_ARG(cfg >= 0, "sf_gemm_bf16: a k-tile-major weight needs the persistent kernel (identity maps, N %% 64 == 0, M >= 8192, N >= 512)");
  }
  if (w_kmajor && cfg != 7 && cfg != 11) { sf_set_error("sf_gemm_bf16: only the persistent kernels (configs 7, 11) read a k-tile-major weight (forced config %d)", cfg); return -1; }
  switch (cfg) {
    case 0: return dispatch_gemm<Cfg0>(a, obf, gelu, res, fast, s);
#ifdef SF_ABLATION
    case 1: return dispatch_gemm<Cfg1>(a, obf, gelu, res, fast, s);
    case 2: return dispatch_gemm<Cfg2>(a, obf, gelu, res, fast, s);
    case 3: return dispatch_gemm<Cfg3>(a, obf, gelu, res, fast, s);
    case 5: return dispatch_gemm<Cfg5>(a, obf, gelu, res, fast, s);
    case 6: return dispatch_gemm<Cfg6>(a, obf, gelu, res, fast, s);
    case 8: return dispatch_gemm<Cfg8>(a, obf, gelu, res, fast, s);
    case 9: return dispatch_gemm<Cfg9>(a, obf, gelu, res, fast, s);
#else
    case 1: case 2: case 3: case 5: case 6: case 8: case 9: case 10: case 12:
      sf_set_error("sf_gemm_bf16: tile config %d is a measured-slower alternative that only the ablation build carries (libsynchformer_hip_ablation.so, -DSF_ABLATION); "
                   "the product library holds configs 0 (128 x 128), 4 (batched), 7 and 11 (persistent 256 x 256)", cfg);
      return -1;
#endif
    case 4: return dispatch_gemm<Cfg4>(a, obf, gelu, res, fast, s);
    case 7: if (!fast) { sf_set_error("sf_gemm_bf16: config 7 needs N %% 64 == 0"); return -1; }
            return dispatch_gemm_persistent(a, obf, gelu, res, s);
    case 11: if (!fast || !sf_gemm_pp_supported(a)) { sf_set_error("sf_gemm_bf16: config 11 needs N %% 64 == 0, K %% 128 == 0, K >= 256"); return -1; }
             return sf_gemm_pp_dispatch(a, obf, gelu, res, s);
#ifdef SF_ABLATION
    case 10: if (!fast || w_kmajor || M * lda * 2 >= ((int64_t)1 << 32) || N * ldw * 2 >= ((int64_t)1 << 32)) {
              sf_set_error("sf_gemm_bf16: config 10 needs N %% 64 == 0, a row-major weight and operands below 4 GiB"); return -1; }
            return dispatch_gemm_w4(a, obf, gelu, res, s);
    case 12: if (!fast || !obf || res || !sf_gemm_r4_supported(a)) {
              sf_set_error("sf_gemm_bf16: config 12 needs a bf16 output without residual, N %% 128 == 0, K %% 128 == 0, K >= 256, row strides %% 64 == 0, a row-major weight"); return -1; }
            return sf_gemm_r4_dispatch(a, gelu, s);
#endif
    default: sf_set_error("sf_gemm_bf16: unknown tile config %d", cfg); return -1;
  }
}

// fc1 of a TRAINED MLP in one launch: pre = A W^T + bias (bf16) AND act = gelu(pre) (bf16), both kept for the backward (was sf_gemm_bf16 -> sf_gelu_fwd).
// Config 11 only: K % 128 == 0, K >= 256, N % 64 == 0, 16-byte aligned rows, outputs below 4 GiB; returns SF_NOT_APPLICABLE (-2, nothing launched; cannot collide with a
// hipError_t such as hipErrorInvalidValue = 1 coming back from the launch) when the shape is outside that range, so the caller can take the two-launch path.
extern "C" int sf_gemm_bf16_gelu_dual(const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, bf16_t* pre, bf16_t* act, int64_t ldc,
                                      int64_t M, int64_t N, int64_t K, void* stream) {
  SF_CHECK_ARG(A && W && pre && act, "sf_gemm_bf16_gelu_dual: null pointer");
  if (M <= 0 || N <= 0) return 0;
  const int64_t m_pad = ((M + 255) / 256) * 256;
  const bool ok = K > 0 && (K % 128) == 0 && K >= 256 && (N % 64) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && (ldc % 4) == 0 && ((uintptr_t)A % 16) == 0 &&
                  ((uintptr_t)W % 16) == 0 && ((uintptr_t)pre % 16) == 0 && ((uintptr_t)act % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0) &&
                  m_pad * ldc * 2 < ((int64_t)1 << 32) && M >= 256 && N >= 256 && M < ((int64_t)1 << 31) && N < ((int64_t)1 << 31);
  GemmArgs a;
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.C = act; a.C2 = pre; a.ldc = ldc; a.R = nullptr; a.ldr = 0;
  a.cmap = sf_rowmap(nullptr); a.rmap = sf_rowmap(nullptr);
  a.M = M; a.N = (int)N; a.K = (int)K;
  a.tiles_n = 0; a.tiles_total = 0; a.nchunk = 0; a.wk = 64;
  a.batch_inner = 0; a.sA0 = a.sA1 = a.sW0 = a.sW1 = a.sC0 = a.sC1 = 0;
  if (!ok || !sf_gemm_pp_supported(a)) return SF_NOT_APPLICABLE;
  return sf_gemm_pp_dispatch(a, true, true, false, (hipStream_t)stream);
}

// Which tile configuration the automatic choice takes for a row-major, identity-mapped, 16-byte aligned GEMM of this shape (bench.py files its launch timings under
// the kernel symbol rocprofv3 will report): 0 = 128 x 128, 7 / 11 = persistent 256 x 256 (round-2 / quadrant-phased schedule).
extern "C" int sf_gemm_bf16_auto_config(int64_t M, int64_t N, int64_t K, int has_residual) {
  GemmArgs a;
  a.lda = K; a.ldw = K; a.K = (int)K;
  return gemm_auto_config(M, N, K, (N % 64) == 0, has_residual != 0, false, sf_gemm_pp_supported(a));
}

// Strided-batched small GEMM: for b0 < batch_outer, b1 < batch_inner
//     C[b0,b1] (M x N) = A[b0,b1] (M x K) * W[b0,b1]^T (N x K) (+ bias)
// with X[b0,b1] = X + b0 * sX0 + b1 * sX1 (element strides).  K % 32 == 0 (callers zero-pad the contraction dimension),
// any M / N.  128 x 128 x 32 tiles, 4-stage ring.  Used by the attention backward of the Stage-2 train step: per
// (clip, head) Q K^T, dO V^T, dS K, dS^T Q, P^T dO are 198 x 198 x 96-class products inside the packed (rows, 2304)
// projection buffers.
extern "C" int sf_gemm_bf16_batched(const bf16_t* A, int64_t lda, int64_t sA0, int64_t sA1, const bf16_t* W, int64_t ldw,
                                    int64_t sW0, int64_t sW1, const float* bias, void* C, int c_dtype, int64_t ldc, int64_t sC0,
                                    int64_t sC1, int64_t M, int64_t N, int64_t K, int batch_outer, int batch_inner,
                                    void* stream) {
  SF_CHECK_ARG(A && W && C, "sf_gemm_bf16_batched: null pointer");
  SF_CHECK_ARG(c_dtype == SF_BF16 || c_dtype == SF_F32, "sf_gemm_bf16_batched: c_dtype must be bf16 or f32");
  SF_CHECK_ARG(K > 0 && (K % 32) == 0, "sf_gemm_bf16_batched: K=%lld must be a positive multiple of 32", (long long)K);
  SF_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0 && (sA0 % 8) == 0 && (sA1 % 8) == 0 && (sW0 % 8) == 0 && (sW1 % 8) == 0,
               "sf_gemm_bf16_batched: lda/ldw/batch strides of A and W must be multiples of 8 elements (16 B)");
  SF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "sf_gemm_bf16_batched: A/W must be 16-byte aligned");
  SF_CHECK_ARG(batch_outer >= 1 && batch_inner >= 1 && (int64_t)batch_outer * batch_inner < 65536, "sf_gemm_bf16_batched: bad batch");
  if (M <= 0 || N <= 0) return 0;
  GemmArgs a;
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.C = C; a.ldc = ldc; a.R = nullptr; a.ldr = 0;
  a.cmap = sf_rowmap(nullptr); a.rmap = sf_rowmap(nullptr);
  a.M = M; a.N = (int)N; a.K = (int)K; a.tiles_n = 0; a.tiles_total = 0; a.nchunk = 0; a.wk = 64;
  a.batch_inner = batch_inner; a.sA0 = sA0; a.sA1 = sA1; a.sW0 = sW0; a.sW1 = sW1; a.sC0 = sC0; a.sC1 = sC1;
  g_batch_count = (int64_t)batch_outer * batch_inner;
  // 64-deep stages when the contraction allows it (the split-K weight-gradient products of the train steps: K = chunks of 64 rows)
  // the branch-free buffer-op epilogue also serves batches whose per-batch output is a whole number of 64-column wave tiles (the split-K
  // weight-gradient partials): the batch offset is folded into the base pointer inside the kernel
  const int64_t m_pad = ((M + 255) / 256) * 256;
  const int esz = c_dtype == SF_BF16 ? 2 : 4;
  const bool fast = (N % 64) == 0 && (ldc % 4) == 0 && ((uintptr_t)C % 16) == 0 && (sC0 * esz) % 16 == 0 && (sC1 * esz) % 16 == 0 &&
                    (!bias || ((uintptr_t)bias % 16) == 0) && m_pad * ldc * esz < ((int64_t)1 << 32);
  const int rc = (K % 64 == 0 && K >= 512) ? dispatch_gemm<Cfg0>(a, c_dtype == SF_BF16, false, false, fast, (hipStream_t)stream)
                                           : dispatch_gemm<Cfg4>(a, c_dtype == SF_BF16, false, false, /*fast=*/false, (hipStream_t)stream);
  g_batch_count = 1;
  return rc;
}

// =========================================================================================================
// Split-K "TN" weight-gradient GEMM (train steps):   part[s][i][j] = sum_{m in chunk s} dY[m][i] * X[m][j]
// Both operands are the ROW-MAJOR activations as the backward pass holds them (rows = tokens = the contraction index), so no
// transposed copies are made (the first version transposed dY and X per linear layer: 12 % of the Stage-1 step).  A stage is 32 token
// rows x 128 columns of each operand, LDS-DMA'd as whole 256-byte rows; the MFMA operands - 8 consecutive tokens of one column per
// lane - are gathered with ds_read_b64_tr_b16, the 4 x 16 transposing LDS read: lane (fr, fg) points at the 8 bytes
// T[k0 + 8 fg + 4 h + (fr >> 2)][c0 + 4 (fr & 3) .. +3] and receives T[k0 + 8 fg + 4 h + 0..3][c0 + fr] (h = 0, 1 -> the 8 k values).
// 16-byte chunk c of token row k sits at chunk c ^ swz(k), swz(k) = 2 ((k & 3) | ((k >> 3) & 1) << 2): the 8 rows a 32-lane half
// reads land on 8 different 32-byte bank groups, and the permutation is applied on the SOURCE side of the lane-linear LDS-DMA.
// 128 x 128 output tile, 4 waves (64 x 64 each), a 4-slot ring of 32-row stages (three in flight), 2 workgroups per CU; rows beyond M read a
// zero page.
// =========================================================================================================
struct TnArgs {
  const bf16_t* A; int64_t lda;        // dY (M, N): output rows i index its columns
  const bf16_t* B; int64_t ldb;        // X  (M, K): output columns j index its columns
  float* C; int64_t ldc, sC;           // part (split, N, K) fp32
  int64_t M;                           // valid token rows
  int kc;                              // token rows per split chunk (multiple of 64)
  uint32_t tiles_n;                    // column tiles (K / 128)
  int64_t ldc_n;                       // N (row length of bias_part)
  float* bias_part;                    // optional (split, N) fp32: per-chunk column sums of dY (the bias gradient's partials)
};
#define TN_BK 32                       // token rows per stage: one 32-deep MFMA k-step
#define TN_NS 4                        // ring slots: three stages in flight while one is being multiplied
#define TN_STAGE (2 * TN_BK * 256)     // 16 KiB: A rows + B rows
#define TN_LDS (TN_NS * TN_STAGE)      // 64 KiB -> two workgroups per CU
__device__ __attribute__((aligned(256))) const uint32_t g_tn_zero_page[64] = {0};

typedef short tn_s4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tn_s4 tn_tr_read(const char* lds) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s4*)lds);
}
__device__ __forceinline__ int tn_swz(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

__global__ __launch_bounds__(256, 2) void gemm_tn_splitk_kernel(TnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware bijective remap (as in gemm_bf16_kernel): the workgroups an XCD receives (ids x, x+8, ...) cover a CONTIGUOUS range of
  // tiles, so neighbouring tiles - which share their dY columns (same tm) - hit the same L2
  uint32_t vb;
  {
    const uint32_t nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const uint32_t tm = vb / p.tiles_n, tn = vb - tm * p.tiles_n;
  const int i0 = (int)tm * 128, j0 = (int)tn * 128;
  const int64_t row_base = (int64_t)blockIdx.y * p.kc;
  const int64_t left = p.M - row_base;
  const int valid = left < p.kc ? (int)(left > 0 ? left : 0) : p.kc;      // token rows of this chunk that exist
  const int nk = (valid + TN_BK - 1) / TN_BK;

  // LDS-DMA pieces: piece q = 4 token rows x 256 B; wave w issues pieces 2w, 2w+1 of A and of B per stage.  The loop is bound by the
  // latency of these loads (a 64-row, 2-slot version spent ~60 % of its time in the vmcnt(0) before the barrier), hence the 4-slot ring.
  const int prow = lane >> 4, pos = lane & 15;
  const bf16_t* a_src[2];
  const bf16_t* b_src[2];
  int krow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    krow[i] = (wave * 2 + i) * 4 + prow;
    const int gch = pos ^ tn_swz(krow[i]);
    a_src[i] = p.A + (row_base + krow[i]) * p.lda + i0 + gch * 8;
    b_src[i] = p.B + (row_base + krow[i]) * p.ldb + j0 + gch * 8;
  }
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_tn_zero_page) + pos * 8;
  const int64_t a_step = (int64_t)TN_BK * p.lda, b_step = (int64_t)TN_BK * p.ldb;
  // stage() is called with kt = 0, 1, 2, ... in order: the source pointers advance by 32 token rows per call; rows beyond `valid`
  // (last k-tile of a ragged chunk) read the zero page.  The four pieces of a wave (A rows 8w..8w+7, then B rows 8w..8w+7) are
  // consecutive in LDS and issued from inline asm (dma4): hipcc puts a vmcnt(0) in front of the fragment reads when it sees the
  // global_load_lds builtin in this loop, which serialises prefetch and compute; the waits below are the hand-counted ones.
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 4096);
  auto stage = [&](int slot, int kt) {
    const int lim = valid - kt * TN_BK;
    dma4(krow[0] < lim ? a_src[0] : zero, krow[1] < lim ? a_src[1] : zero, krow[0] < lim ? b_src[0] : zero, krow[1] < lim ? b_src[1] : zero,
         lds_wave + slot * TN_STAGE);
#pragma unroll
    for (int i = 0; i < 2; ++i) { a_src[i] += a_step; b_src[i] += b_step; }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  int a_off[4], b_off[4];                                         // token row k of a stage lives at (k >> 3) * 4096 + (k & 7) * 256 (A) / + 2048 (B)
  {
    const int sw = tn_swz(fg * 8 + (fr >> 2)), c1 = (fr & 3) >> 1, hb = (fr & 1) * 8, rb = fg * 4096 + (fr >> 2) * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a_off[i] = rb + (((wm * 8 + 2 * i + c1) ^ sw) << 4) + hb;
      b_off[i] = rb + (((wn * 8 + 2 * i + c1) ^ sw) << 4) + hb + 2048;
    }
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.bias_part != nullptr && tn == 0 && wn == 0;
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s_ = 0; s_ < TN_NS - 1; ++s_)
    if (s_ < nk) stage(s_, s_);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed when at most the (<= 2) younger stages are outstanding (4 LDS-DMA instructions per thread per stage)
    const int ahead = min(TN_NS - 2, nk - 1 - kt);
    if (ahead >= 2) wait_vmcnt_barrier<8>(); else if (ahead == 1) wait_vmcnt_barrier<4>(); else wait_vmcnt_barrier<0>();
    // every wave is past compute(kt-1): its slot is free -> refill with tile kt+3
    if (kt + TN_NS - 1 < nk) stage((kt + TN_NS - 1) & (TN_NS - 1), kt + TN_NS - 1);
    const char* st = smem + (kt & (TN_NS - 1)) * TN_STAGE;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      union { bf16x8 v; tn_s4 h[2]; } ua, ub;
      ua.h[0] = tn_tr_read(st + a_off[i]);
      ua.h[1] = tn_tr_read(st + a_off[i] + 4 * 256);
      ub.h[0] = tn_tr_read(st + b_off[i]);
      ub.h[1] = tn_tr_read(st + b_off[i] + 4 * 256);
      a[i] = ua.v; b[i] = ub.v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if (do_bias) {                                                // wave-uniform: the column-tile-0 workgroups' wn == 0 waves
      const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
#pragma unroll
      for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
    }
  }
  // bias gradient partials: against an all-ones B fragment every column of the 16 x 16 block is the row sum of the dY^T fragment, i.e. the
  // column sum of dY over this chunk's token rows - four extra MFMAs per k-step in 1 / (2 K/128) of the waves instead of a separate pass over dY
  if (do_bias && fr == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(p.bias_part + (int64_t)blockIdx.y * (p.ldc_n) + i0 + wm * 64 + i * 16 + fg * 4) = make_float4(accb[i][0], accb[i][1], accb[i][2], accb[i][3]);
  }
  __syncthreads();                                               // operand LDS becomes per-wave epilogue scratch

  // epilogue: 16-row groups through a per-wave slab -> 16-byte row stores
  float* slab = reinterpret_cast<float*>(smem + wave * (16 * EPI_LD * 4));
  float* cbase = p.C + (int64_t)blockIdx.y * p.sC + (int64_t)(i0 + wm * 64) * p.ldc + j0 + wn * 64;
  const int ecol = (lane & 15) * 4;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(fg * 4 + r) * EPI_LD + j * 16 + fr] = acc[g][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int lrow = ps * 4 + (lane >> 4);
      const float4 v = *reinterpret_cast<const float4*>(slab + lrow * EPI_LD + ecol);
      *reinterpret_cast<float4*>(cbase + (int64_t)(g * 16 + lrow) * p.ldc + ecol) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// part (split, N, K) fp32 <- per-chunk dY^T X; chunk s covers token rows [s*kc, min((s+1)*kc, M)).  The caller sums the chunks (sf_seqsum).
// bias_part (split, N) fp32 or NULL <- per-chunk column sums of dY (the bias gradient; summed by sf_seqsum(bias_part, N, split, 1, N, ...)).
extern "C" int sf_gemm_tn_splitk(const bf16_t* dY, int64_t ldy, const bf16_t* X, int64_t ldx, float* part, float* bias_part, int64_t M, int64_t N,
                                 int64_t K, int split, int64_t kc, void* stream) {
  SF_CHECK_ARG(dY && X && part, "sf_gemm_tn_splitk: null pointer");
  SF_CHECK_ARG(M >= 1 && N >= 128 && K >= 128 && (N % 128) == 0 && (K % 128) == 0, "sf_gemm_tn_splitk: N=%lld and K=%lld must be multiples of 128",
               (long long)N, (long long)K);
  SF_CHECK_ARG(split >= 1 && split < 65536 && kc >= 64 && (kc % 64) == 0 && kc < ((int64_t)1 << 30) && (int64_t)split * kc >= M,
               "sf_gemm_tn_splitk: split * kc must cover M, kc %% 64 == 0");
  SF_CHECK_ARG((ldy % 8) == 0 && (ldx % 8) == 0 && ((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)part % 16) == 0,
               "sf_gemm_tn_splitk: operands must be 16-byte aligned with row strides %% 8 == 0");
  if (int rc = sf_prepare_kernel((const void*)gemm_tn_splitk_kernel, TN_LDS, "sf_gemm_tn_splitk")) return rc;
  TnArgs a;
  a.A = dY; a.lda = ldy; a.B = X; a.ldb = ldx; a.C = part; a.ldc = K; a.sC = N * K; a.M = M; a.kc = (int)kc;
  a.tiles_n = (uint32_t)(K / 128);
  a.bias_part = bias_part; a.ldc_n = N;
  SF_CHECK_ARG(!bias_part || ((uintptr_t)bias_part % 16) == 0, "sf_gemm_tn_splitk: bias_part must be 16-byte aligned");
  const int64_t tiles = (N / 128) * (K / 128);
  SF_CHECK_ARG(tiles < ((int64_t)1 << 31), "sf_gemm_tn_splitk: too many tiles");
  hipLaunchKernelGGL(gemm_tn_splitk_kernel, dim3((unsigned)tiles, (unsigned)split), dim3(256), TN_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
