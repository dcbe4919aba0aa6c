// Split-K "TN" weight-gradient GEMM on the quadrant-phased schedule of sf_gemm_pp.hip (gfx950):
//     part[s][i][j] = sum_{m in chunk s} dY[m][i] * X[m][j]            (fp32; the caller sums the chunks)
// Both operands are the ROW-MAJOR activations of the backward pass (rows = tokens = the contraction index), as in gemm_tn_splitk_kernel (sf_gemm.hip),
// which this kernel replaces for the big weight gradients of the Stage-1 train step: that one has 64 x 64 wave tiles (1 KiB of fragment reads per 32
// cycles of MFMA = the LDS port saturated before the matrix pipe), one barrier + vmcnt wait per 32-row stage, and reached 0.69 PFLOP/s.
//
// Here: 256 x 256 output tile, 8 waves as 2 x 4 with 128 x 64 wave tiles (0.75 KiB of fragment reads per 32 MFMA cycles), a k-tile = 64 token rows
// of dY (256 columns) and of X (256 columns) = two 32-KiB images, cut into half-tiles A0 | A1 | B0 | B1 of 16 KiB = [64 token rows][256 B]: half Ah holds,
// for both wave rows wm, the dY columns wm*128 + h*64 .. +63; half Bh holds, for the four wave columns wn, the X columns wn*64 + h*32 .. +31.  The k-tile's
// four phases (A0,B0) (A0,B1) (A1,B1) (A1,B0), the one-half-tile-per-phase LDS-DMA stream 6 half-tiles ahead of the reads, the counted vmcnt(8) waits,
// the two barriers per phase and the wm = 1 waves running one barrier behind are exactly sf_gemm_pp.hip's (its header has the ordering rules).
// What differs:
//   * the LDS image is TOKEN-major, so the MFMA operands (8 consecutive tokens of one column per lane) are gathered with ds_read_b64_tr_b16, the 4 x 16
//     transposing read: for a 32-column block and a 16-token k-step, 16-lane group g = (column half g & 1, token half g >> 1) points lane i at the 8 bytes
//     T[k0 + 8 (g >> 1) + 4 r + (i >> 2)][c0 + 16 (g & 1) + 4 (i & 3) .. +3] and receives column c0 + 16 (g & 1) + i, tokens .. + 4 r + 0..3 (r = 0, 1: the two
//     reads of an operand).  16-byte chunk c of token row k sits at chunk c ^ swz(k), swz(k) = 2 ((k & 3) | ((k >> 3) & 1) << 2): the 8 rows x 4 chunks one
//     read instruction touches cover every one of the 16 chunk positions (= bank groups) exactly twice;
//   * the LDS-DMA is `buffer_load_dwordx4 ... offen lds` (raw buffer over the operand, num_records = M rows): token rows beyond M arrive as ZEROS from the
//     range check (probed on the hardware: tools/probe/buf_lds_dma.hip), so a ragged last chunk needs no second code path and no zero page; the lane
//     offsets are absolute (32-bit: operands below 4 GiB) and advance by 64 rows per k-tile;
//   * work items are (chunk s, tile): consecutive workgroups of an XCD take the tiles of ONE chunk, whose dY / X rows they share through that XCD's L2.
//   * the bias gradient (column sums of dY over the chunk) rides along in the items of column tile 0: a lane's A fragment IS 8 tokens of one dY column, so
//     the sum is 4 v_dot2_f32_bf16 against ones per fragment, issued behind the phase's MFMAs - by all four wave columns wn (sharing the k-steps between them
//     needs a run-time selection of the fragment, i.e. branches in the matrix segment: measured +10 %); wn = 0 writes bias_part (split, N).
#include "sf_gemm_common.h"
#include <type_traits>

#define TQ_HALF (64 * 256)             // 16 KiB half-tile: 64 token rows x 128 columns (bf16)
#define TQ_STAGE (4 * TQ_HALF)         // A0 | A1 | B0 | B1
#define TQ_SLAB_BYTES 4096
#define TQ_LDS (2 * TQ_STAGE + 8 * TQ_SLAB_BYTES)   // 160 KiB
#define TQ_EPI_STORES 32
#ifndef SF_TQ_STORECNT
#define SF_TQ_STORECNT 1               // first k-tile after an epilogue: its 32 stores may stay in flight behind the loads (vmcnt 8 + 32)
#endif

struct TnPpArgs {
  const bf16_t* A; int64_t lda;        // dY (M x N)
  const bf16_t* B; int64_t ldb;        // X  (M x K)
  float* C;                            // part (split, N, K)
  float* bias_part;                    // (split, N) or NULL
  int64_t M;
  int N, K;
  int kc;                              // token rows per chunk (a multiple of 128)
  uint32_t tiles_n, tiles, items;      // column tiles, tiles per chunk, tiles * split
};

typedef short tq_s4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tq_s4 tq_tr_read(const char* lds) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tq_s4*)lds);
}

// two LDS-DMA pieces (1 KiB each, consecutive in LDS from the wave-uniform byte address l0) through a raw buffer: lanes beyond num_records write zeros
__device__ __forceinline__ void tq_dma2(uint32_t v0, uint32_t v1, __amdgpu_buffer_rsrc_t r, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "s"(r), "s"(l0)
      : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void tq_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void tq_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int V> using tq_ic = std::integral_constant<int, V>;
// The lane index, re-derived where it is needed (mbcnt over an opaque zero: neither hoisted nor kept): at 256 registers a thread id kept live across the
// k-loop is spilled, and its reload sits behind a compiler vmcnt(0) - i.e. behind the whole operand stream in flight.
__device__ __forceinline__ int tq_lane() {
  int z = 0;
  asm volatile("" : "+v"(z));
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

__global__ __launch_bounds__(512, 2) void gemm_tn_pp_kernel(TnPpArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                       // 2 x 4 waves, wave tile 128 x 64

  // work items u = s * tiles + tile; XCD x (blocks x, x + 8, ...) owns the contiguous range [x * per, (x + 1) * per) of them
  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t per = (p.items + 7u) >> 3;
  const uint32_t u0 = min(xcd * per, p.items), u1 = min(u0 + per, p.items), t_end = u1 - u0;
  auto item_origin = [&](uint32_t t, uint32_t& s, int& i0, int& j0) {
    const uint32_t u = u0 + t;
    s = u / p.tiles;
    const uint32_t tile = u - s * p.tiles, tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    i0 = (int)tm * 256; j0 = (int)tn * 256;
  };

  // ---- fragment read offsets inside a half-tile ([64 rows][256 B], chunk c of row k at c ^ swz(k)) ---------------------------------------
  int a_off[2][4], b_off[4];
  {
    const int g = lane >> 4, li4 = lane & 15;
    const int cg = g & 1, kg = g >> 1;
    const int swz = ((li4 >> 2) | (kg << 2)) << 1;
    const int rowb = (kg * 8 + (li4 >> 2)) * 256, sub = ((li4 & 3) >> 1), byt = (li4 & 1) * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a_off[i][kk] = kk * 16 * 256 + rowb + (((wm * 8 + i * 4 + cg * 2 + sub) ^ swz) << 4) + byt;          // + ha * TQ_HALF; second read + 4 * 256
      b_off[kk] = 2 * TQ_HALF + kk * 16 * 256 + rowb + (((wn * 4 + cg * 2 + sub) ^ swz) << 4) + byt;                                      // + hb * TQ_HALF
    }
  }

  // ---- load iterator: runs 6 half-tiles ahead of the reads over the workgroup's whole k-tile sequence ---------------------------
  const int nk = p.kc / 64;
  uint32_t ld_t = li;
  bool ld_ok = ld_t < t_end;
  if (!ld_ok) return;
  int ld_kt = 0;
  uint32_t oA[2][2], oB[2][2];                                   // absolute byte offsets of this lane's pieces of the CURRENT k-tile (advance by 64 rows per k-tile)
  const uint32_t stepA = (uint32_t)(64 * p.lda * 2), stepB = (uint32_t)(64 * p.ldb * 2);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), (short)0, (int)(uint32_t)(p.M * p.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B), (short)0, (int)(uint32_t)(p.M * p.ldb * 2), 0x00020000);
  auto ld_set = [&](uint32_t t) {
    uint32_t s; int i0, j0;
    item_origin(t, s, i0, j0);
    const int ll = tq_lane();                                    // lane quantities are re-derived per item, not kept across the k-loop
    const int kr = ll >> 4, slot = ll & 15;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = (wave * 2 + j) * 4 + kr;                     // token row of the k-tile this lane fills
      const int c = slot ^ ((((k & 3) | (((k >> 3) & 1) << 2))) << 1);   // source chunk of the lane's (physical) slot
      const uint32_t row = s * (uint32_t)p.kc + (uint32_t)k;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        oA[h][j] = (row * (uint32_t)p.lda + (uint32_t)(i0 + (c >> 3) * 128 + h * 64 + (c & 7) * 8)) * 2u;
        oB[h][j] = (row * (uint32_t)p.ldb + (uint32_t)(j0 + (c >> 2) * 64 + h * 32 + (c & 3) * 8)) * 2u;
      }
    }
  };
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 2048);
  const uint32_t slab_dummy = __builtin_amdgcn_readfirstlane(lds_addr(smem) + 2 * TQ_STAGE + wave * TQ_SLAB_BYTES + 1024);
  // PART: 0 = A0, 1 = B0, 2 = B1, 3 = A1 (the order in which a k-tile's halves are read).  Branch-free: a dry iterator keeps issuing (dummy pieces into this
  // wave's idle epilogue slab), so every counted wait keeps its count
  auto issue = [&](auto PARTc, auto STc) {
    constexpr int PART = decltype(PARTc)::value, ST = decltype(STc)::value;
    constexpr bool isA = PART == 0 || PART == 3;
    constexpr int h = PART >= 2 ? 1 : 0;
    const uint32_t real = lds_wave + ST * TQ_STAGE + (isA ? h : 2 + h) * TQ_HALF;
    const uint32_t l = ld_ok ? real : slab_dummy;
    if (isA) tq_dma2(oA[h][0], oA[h][1], ra, l);
    else tq_dma2(oB[h][0], oB[h][1], rb, l);
    if (PART == 3 && ld_ok) {
      if (++ld_kt == nk) {
        ld_kt = 0;
        ld_t += per_xcd_blocks;
        ld_ok = ld_t < t_end;
        if (ld_ok) ld_set(ld_t);
      } else {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int j = 0; j < 2; ++j) { oA[hh][j] += stepA; oB[hh][j] += stepB; }
      }
    }
  };

  // ---- compute-side state ------------------------------------------------------------------------------------------------------------
  uint32_t t = li;
  uint32_t cs; int i0, j0;
  item_origin(t, cs, i0, j0);
  char* bslab = smem + 2 * TQ_STAGE + wave * TQ_SLAB_BYTES;
  const int64_t plane = (int64_t)p.N * p.K;                      // elements per chunk plane
  const uint32_t cstep = (uint32_t)(4 * p.K) * 4u;

  // ---- prologue: k-tile 0 and A0 | B0 of k-tile 1 in flight -------------------------------------------------------------------
  ld_set(ld_t);
  issue(tq_ic<0>{}, tq_ic<0>{}); issue(tq_ic<1>{}, tq_ic<0>{}); issue(tq_ic<2>{}, tq_ic<0>{}); issue(tq_ic<3>{}, tq_ic<0>{});
  issue(tq_ic<0>{}, tq_ic<1>{}); issue(tq_ic<1>{}, tq_ic<1>{});
  asm volatile("" ::: "memory");
  tq_wait_vmcnt<8>();                                            // A0 | B0 of k-tile 0 have landed (this wave's pieces)
  tq_barrier();
  int extra = 0;                                                 // epilogue stores that may still be in flight behind the loads (first k-tile of an item)
  typedef __attribute__((ext_vector_type(2))) __bf16 tq_bf2;
  const tq_bf2 ones2 = __builtin_bit_cast(tq_bf2, 0x3F803F80u);

  for (;;) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[2][4], b0[4], b1[4];
    const bool do_bias = p.bias_part != nullptr && j0 == 0;       // wave-uniform, per item
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};                         // this lane's dY column of row block b: sum over the tokens of k-half hi
    auto bias_acc = [&](auto HAc) {
      constexpr int HA = decltype(HAc)::value;
      if (!do_bias) return;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 w = __builtin_bit_cast(uint4, a[i][kk]);
          float d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tq_bf2, w.x), ones2, bsum[HA * 2 + i], false);
          d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tq_bf2, w.y), ones2, d, false);
          d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tq_bf2, w.z), ones2, d, false);
          bsum[HA * 2 + i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tq_bf2, w.w), ones2, d, false);
        }
    };

    auto wait_loads = [&](bool first) {
      asm volatile("" ::: "memory");
      if (SF_TQ_STORECNT && first && extra) { if (extra > TQ_EPI_STORES) tq_wait_vmcnt<8 + TQ_EPI_STORES + 4>(); else tq_wait_vmcnt<8 + TQ_EPI_STORES>(); }
      else tq_wait_vmcnt<8>();
    };
    auto frag = [&](const char* base, int off) -> bf16x8 {
      union { bf16x8 v; tq_s4 h[2]; } u;
      u.h[0] = tq_tr_read(base + off);
      u.h[1] = tq_tr_read(base + off + 4 * 256);
      return u.v;
    };
    auto mma = [&](auto HAc, auto HBc, const bf16x8 (&bf)[4]) {
      constexpr int HA = decltype(HAc)::value, HB = decltype(HBc)::value;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[HA * 2 + i][HB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kk], bf[kk], acc[HA * 2 + i][HB], 0, 0, 0);
      asm volatile("" : "+v"(acc[HA * 2][HB]), "+v"(acc[HA * 2 + 1][HB]));   // pins the (pure) MFMAs inside their matrix segment
      __builtin_amdgcn_s_setprio(0);
    };
    // one k-tile held in stage S; `first` = first k-tile after an epilogue
    auto ktile = [&](auto Sc, bool first) {
      constexpr int S = decltype(Sc)::value;
      const char* st = smem + S * TQ_STAGE;
      // ---- phase 0: (A0, B0) ----
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) b0[kk] = frag(st, b_off[kk]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i][kk] = frag(st, a_off[i][kk]);
      __builtin_amdgcn_sched_barrier(0);
      issue(tq_ic<2>{}, tq_ic<S ^ 1>{});                          // B1(kt+1) issued; B1(kt) landed
      wait_loads(first);
      tq_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(tq_ic<0>{}, tq_ic<0>{}, b0);
      __builtin_amdgcn_sched_barrier(0);
      bias_acc(tq_ic<0>{});                                       // behind the phase's MFMAs (the fragments are in registers by now; in the read segment the
      __builtin_amdgcn_sched_barrier(0);                          // sums would wait for the LDS reads in front of the DMA issue and the barrier)
      tq_barrier();
      // ---- phase 1: (A0, B1) ----
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) b1[kk] = frag(st + TQ_HALF, b_off[kk]);
      __builtin_amdgcn_sched_barrier(0);
      issue(tq_ic<3>{}, tq_ic<S ^ 1>{});                          // A1(kt+1) issued; A1(kt) landed
      wait_loads(first);
      tq_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(tq_ic<0>{}, tq_ic<1>{}, b1);
      __builtin_amdgcn_sched_barrier(0);
      tq_barrier();
      // ---- phase 2: (A1, B1) ----
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i][kk] = frag(st + TQ_HALF, a_off[i][kk]);
      __builtin_amdgcn_sched_barrier(0);
      issue(tq_ic<0>{}, tq_ic<S>{});                              // A0(kt+2); phase 3 reads nothing: no wait
      tq_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(tq_ic<1>{}, tq_ic<1>{}, b1);
      __builtin_amdgcn_sched_barrier(0);
      bias_acc(tq_ic<1>{});
      __builtin_amdgcn_sched_barrier(0);
      tq_barrier();
      // ---- phase 3: (A1, B0) ----
      issue(tq_ic<1>{}, tq_ic<S>{});                              // B0(kt+2); A0 | B0 of k-tile kt+1 landed
      wait_loads(first);
      tq_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(tq_ic<1>{}, tq_ic<0>{}, b0);
      __builtin_amdgcn_sched_barrier(0);
      tq_barrier();
    };

    if (wm == 1) tq_barrier();                                   // the wm = 1 waves run one barrier behind
    for (int kt = 0; kt < nk; kt += 2) {
      ktile(tq_ic<0>{}, kt == 0);
      ktile(tq_ic<1>{}, false);
    }
    if (wm == 0) tq_barrier();                                   // re-align: both groups run the epilogue concurrently

    if (!ld_ok) tq_wait_vmcnt<0>();                              // dummy pieces of a dry iterator land in the slab: all of them before the epilogue uses it
    // ---- epilogue: 8 branch-free groups of 16 rows x 64 cols through the wave's slab, fp32 row stores into this chunk's plane ----
    {
      const int el = tq_lane(), el31 = el & 31, ehi = el >> 5;    // (re-derived: see tq_lane)
      extra = TQ_EPI_STORES;
      if (do_bias) {                                             // 4 more stores per wave in front of the tile's 32
        extra = TQ_EPI_STORES + 4;
        float* bp = p.bias_part + (int64_t)cs * p.N + i0 + wm * 128 + el31;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float tot = bsum[b] + __shfl_xor(bsum[b], 32, 64);
          if (ehi == 0 && wn == 0) bp[b * 32] = tot;
        }
      }
      float* slab = reinterpret_cast<float*>(bslab);
      const int ecol = (el & 15) * 4;
      const int gcol = j0 + wn * 64 + ecol;
      const int row0 = i0 + wm * 128 + (el >> 4);
      float* cplane = p.C + (int64_t)cs * plane;
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(cplane, (short)0, (int)(uint32_t)(plane * 4), 0x00020000);
      const uint32_t coff0 = (uint32_t)(row0 * p.K + gcol) * 4u;
      const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 nores[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int i = g >> 1, q2 = g & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              slab[(qq * 8 + ehi * 4 + r) * 64 + j * 32 + el31] = acc[i][j][(q2 * 2 + qq) * 4 + r];
        float4 v[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) v[ps] = *reinterpret_cast<const float4*>(slab + (ps * 4 + (el >> 4)) * 64 + ecol);
        epi_group_store<false, false, false>(v, zero4, nores, rc, coff0 + g * 4 * cstep, cstep);
      }
    }
    t += per_xcd_blocks;
    if (t >= t_end) break;
    item_origin(t, cs, i0, j0);
  }
}

// part (split, N, K) fp32 <- per-chunk dY^T X; chunk s covers token rows [s * kc, min((s + 1) * kc, M)).  kc % 128 == 0, N % 256 == 0, K % 256 == 0.
// bias_part (split, N) fp32 or NULL <- per-chunk column sums of dY (the bias gradient: sf_seqsum(bias_part, N, split, 1, N, ...)).
extern "C" int sf_gemm_tn_pp(const bf16_t* dY, int64_t ldy, const bf16_t* X, int64_t ldx, float* part, float* bias_part, int64_t M, int64_t N, int64_t K,
                             int split, int64_t kc, void* stream) {
  SF_CHECK_ARG(dY && X && part, "sf_gemm_tn_pp: null pointer");
  SF_CHECK_ARG(M >= 1 && N >= 256 && K >= 256 && (N % 256) == 0 && (K % 256) == 0, "sf_gemm_tn_pp: N=%lld and K=%lld must be multiples of 256", (long long)N,
               (long long)K);
  SF_CHECK_ARG(split >= 1 && split < 65536 && kc >= 128 && (kc % 128) == 0 && (int64_t)split * kc >= M && (int64_t)(split - 1) * kc < M,
               "sf_gemm_tn_pp: split * kc must cover M with no empty chunk, kc %% 128 == 0");
  SF_CHECK_ARG((ldy % 8) == 0 && (ldx % 8) == 0 && ldy >= N && ldx >= K && ((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)part % 16) == 0,
               "sf_gemm_tn_pp: operands must be 16-byte aligned with row strides %% 8 == 0");
  // 32-bit byte offsets: the operands (buffer descriptors + lane offsets, up to split * kc rows) and one chunk plane of the output
  SF_CHECK_ARG((int64_t)split * kc * ldy * 2 < ((int64_t)1 << 32) && (int64_t)split * kc * ldx * 2 < ((int64_t)1 << 32) && N * K * 4 < ((int64_t)1 << 32),
               "sf_gemm_tn_pp: operands and one output plane must stay below 4 GiB");
  const int n_cu = sf_cu_count("sf_gemm_tn_pp");
  if (n_cu <= 0) return -1;
  if (int rc = sf_prepare_kernel((const void*)gemm_tn_pp_kernel, TQ_LDS, "sf_gemm_tn_pp")) return rc;
  TnPpArgs a;
  a.A = dY; a.lda = ldy; a.B = X; a.ldb = ldx; a.C = part; a.bias_part = bias_part; a.M = M; a.N = (int)N; a.K = (int)K; a.kc = (int)kc;
  a.tiles_n = (uint32_t)(K / 256);
  a.tiles = (uint32_t)((N / 256) * (K / 256));
  const int64_t items = (int64_t)a.tiles * split;
  SF_CHECK_ARG(items < ((int64_t)1 << 30), "sf_gemm_tn_pp: too many work items");
  a.items = (uint32_t)items;
  int64_t blocks = n_cu & ~7;                                      // a whole number of workgroups per XCD
  if (blocks < 8) blocks = 8;
  hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3((unsigned)blocks), dim3(512), TQ_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
