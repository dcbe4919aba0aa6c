// Full-row projection GEMM fused with the residual add AND the next LayerNorm, for gfx950 (MI355X):
//     X[m, :] = A[m, :] * W^T + bias + R[m, :]        (fp32 residual stream, 768 columns; X may alias R)
//     Y[m, :] = LayerNorm(X[m, :]) * gamma + beta     (bf16: the A operand of the NEXT sub-layer's GEMM; Y may alias A)
// This is `x = x + proj(attn(norm(x)))` / `x = x + mlp(norm(x))` followed by the norm that opens the next sub-layer of a
// DividedSpaceTimeBlock (vit_helper.py:364-376: temporal proj -> norm1, spatial proj -> norm2, fc2 -> norm3 of the next block).
//
// Why a kernel of its own: with K = N = 768 the un-fused pair is HBM-bound twice - the GEMM epilogue reads and writes the fp32
// stream (2 x 1.08 GB per launch at 224 segments), then sf_layernorm768 reads the same 1.08 GB again to write 0.54 GB of bf16
// (profiles/r01_bench_summary.md: 13 % + 8 % of the inference step).  LayerNorm needs whole rows, so one workgroup owns BM = 128
// complete rows: a 128 x 768 fp32 accumulator tile = 393 KB, i.e. three quarters of the CU's register file - 8 waves as 2 (M) x 4 (N),
// each holding a 64 x 192 block as 2 x 6 fragments of v_mfma_f32_32x32x16_bf16 (192 accumulator registers per lane).  The epilogue adds
// bias + residual, writes X once, reduces the row statistics in registers (16-lane DPP row sums, then 4 partials per row through LDS)
// and writes the normalised bf16 rows: the stream is read once and written once, and the separate LayerNorm launch disappears.
//
// Main loop: K-steps of 32 (a stage = 128 x 32 of A + 768 x 32 of W = 56 KiB; two ring slots = 112 KiB), operands HBM/L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4 from inline asm, SGPR base + 32-bit lane offset: 7 pieces of 1 KiB per wave per stage), one counted wait +
// raw s_barrier per K-step, the refill of the other slot issued behind the first MFMA cluster.  LDS rows are 64 B; 16-byte chunk c of row r
// sits at slot c ^ ((r >> 2) & 3) (applied to the SOURCE address of the lane-linear DMA and to the fragment reads): the 16-lane groups a
// ds_read_b128 is served in then touch 16 distinct slots of the 256-byte bank row.
// The workgroups are persistent (one per CU, tile t = block + k * grid): the next tile's first two stages are put in flight before the
// epilogue runs, so the epilogue's HBM traffic and the next main loop's first operand loads overlap.
#include "sf_gemm_ln_common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/synchformer_hip.h"

#ifndef SF_RL_PP
#define SF_RL_PP 1            // 1: quadrant-phased schedule (round 3); 0: round 2's loop (SF_RL_SCHED=0 in the environment selects it at run time, for tests)
#endif
#ifndef SF_RL_STORECNT
#define SF_RL_STORECNT 1      // first k-step of a tile: the 48 Y stores of the previous epilogue may stay in flight behind the loads (vmcnt 7 + 48)
#endif

#ifndef SF_RL_STORE_AUX
#define SF_RL_STORE_AUX 2     // nt: written once, consumed by a later launch
#endif
#ifndef SF_RL_SPREAD
#define SF_RL_SPREAD 1        // 1: one LDS-DMA piece behind every third MFMA of the k-step; 0: all seven behind the first MFMA cluster
#endif
#ifndef SF_RL_A_NT
#define SF_RL_A_NT 0         // experiment: the A rows (read by exactly one workgroup) with the nt hint
#endif
#ifndef SF_RL_KROT
#define SF_RL_KROT 1          // workgroup i of an XCD starts its k-loop at k-step (i * SF_RL_KROT) % nk instead of 0 (see below); 0 = off
#endif
#ifndef SF_RL_LOAD_AUX
#define SF_RL_LOAD_AUX 2
#endif

struct ResLnArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  uint32_t wk;                                                    // bytes between consecutive 32-deep k-steps of a W row (64 row-major, 768 * 64 k-step-major)
  const float* bias;
  const float* R; int64_t ldr;
  float* X; int64_t ldx;
  const float* gamma; const float* beta;
  bf16_t* Y; int64_t ldy;
  int64_t M;
  int K;
  float eps;
  uint32_t tiles;
  uint32_t phases;
  uint32_t stagger;                                               // measurement hook (SF_RL_STAGGER): every other workgroup of an XCD starts stagger x ~1.2 us late
};

// ABL: ablation mask of the measurement builds (tools/bench_gemm_ln.py, SF_RL_ABL): 1 = no residual loads, 2 = no X stores, 4 = no Y stores,
// 8 = no operand refills after the first two stages, 16 = no MFMAs, 32 = only half of the W pieces are refilled.  The product instantiation is ABL = 0.
template <int ABL, bool PP>
__global__ __launch_bounds__(512, 2) void gemm_res_ln768_kernel(ResLnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                        // 2 x 4 waves, wave tile 64 x 192
  // ---- LDS-DMA addressing: lane i of a piece fills (row i >> 2, slot i & 3) with source chunk slot ^ ((row >> 2) & 3) ----
  const int prow = lane >> 2, pslot = lane & 3;
  const int pchunk = pslot ^ ((prow >> 2) & 3);
  const uint32_t voff_b0 = (uint32_t)(prow * p.ldw * 2 + pchunk * 16);
  const char* wbase = reinterpret_cast<const char*>(p.W) + (int64_t)(wave * 96) * p.ldw * 2;
  const int64_t wstep = (int64_t)16 * p.ldw * 2;
  const void* sb0 = wbase; const void* sb1 = wbase + wstep; const void* sb2 = wbase + 2 * wstep;
  const void* sb3 = wbase + 3 * wstep; const void* sb4 = wbase + 4 * wstep; const void* sb5 = wbase + 5 * wstep;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(rl_lds_addr(smem));
  const uint32_t lds_a_w = lds0 + wave * 1024, lds_b_w = lds0 + RL_A_BYTES + wave * 6 * 1024;

  // fragment read offsets (bytes) inside an operand tile for the two 16-deep k-steps of a stage
  int frag_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) frag_off[ks] = (lane & 31) * 64 + (((ks * 2 + (lane >> 5)) ^ (((lane & 31) >> 2) & 3)) << 4);
  const int a_base = wm * 64 * 64, b_base = RL_A_BYTES + wn * 192 * 64;

  const int nk = p.K / RL_BK;
  uint32_t t = blockIdx.x;
  if (t >= p.tiles) return;
  // the workgroups that own one tile fewer than workgroup 0 (tiles % gridDim.x != 0) start late: their epilogues then fall into the others' main loops
  if (p.phases <= 1u) {
    if ((p.tiles - 1u - blockIdx.x) / gridDim.x < (p.tiles - 1u) / gridDim.x) for (uint32_t i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(32);
  } else {
    for (uint32_t i = 0, n = ((blockIdx.x >> 3) % p.phases) * p.stagger; i < n; ++i) __builtin_amdgcn_s_sleep(32);
  }

  int64_t m0 = (int64_t)t * RL_BM;
  const void* sa; uint32_t voff_a0;
  auto set_tile = [&](int64_t mm) {
    sa = reinterpret_cast<const char*>(p.A) + mm * p.lda * 2;
    int64_t r = wave * 16 + prow;
    const int64_t last = p.M - 1 - mm;                           // tail tile: rows beyond M re-read the last valid row (their outputs are dropped)
    if (r > last) r = last;
    voff_a0 = (uint32_t)(r * p.lda * 2 + pchunk * 16);
  };
  // k-loop rotation: the 32 workgroups of an XCD (block b sits on XCD b % 8) run in near lockstep, so without it they all ask the XCD's L2 for
  // the SAME 48 KiB of W at the same moment and queue on those lines; starting every workgroup at a different k-step spreads the requests
  // over all of W (proj 815 -> 724 us, fc2 2050 -> 1940 us at 224 segments).  Same products, rotated fp32 summation order.
  const int krot = SF_RL_KROT ? (int)(((blockIdx.x >> 3) * (uint32_t)SF_RL_KROT) % (uint32_t)nk) : 0;
  auto kmap = [&](int kt) { int k = kt + krot; return k >= nk ? k - nk : k; };
  auto stage = [&](int slot, int kt_) {
    const int kt = kmap(kt_);
    rl_dma7(voff_a0 + kt * (RL_BK * 2), sa, voff_b0 + kt * p.wk, sb0, sb1, sb2, sb3, sb4, sb5, lds_a_w + slot * RL_STAGE,
            lds_b_w + slot * RL_STAGE);
  };
  // ---- PP: lane offsets of this wave's two pieces of a W third (buffer row r = wave * 32 + j * 16 + prow <-> W row (r >> 6) * 192 + c * 64 + (r & 63)),
  // the third's 64-row shift and the k-step go into the scalar base ----
  uint32_t voff_w[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) voff_w[j] = (uint32_t)(((wave >> 1) * 192 + (wave & 1) * 32 + j * 16 + prow) * (int)p.ldw * 2 + pchunk * 16);
  const char* w0 = reinterpret_cast<const char*>(p.W);
  const int64_t third_b = (int64_t)64 * p.ldw * 2;
  // running offsets of the load stream (scalar; no per-issue multiplies in the read segments): wo1 = rotated k-step kt+1 (W thirds 1 and 2 are issued
  // in phases 0 / 1), wo2 / ao2 = rotated k-step kt+2 (A and W third 0, phase 2)
  const char* wb1 = w0 + third_b; const char* wb2 = w0 + 2 * third_b;
  uint32_t wo1 = 0, wo2 = 0, ao2 = 0;
  int kq2 = 0;
  auto pp_issue_w = [&](int c, int slot) {
    if (ABL & 8) return;
    rl_dma2(voff_w[0], voff_w[1], c == 0 ? w0 + wo2 : (c == 1 ? wb1 + wo1 : wb2 + wo1), lds0 + slot * RP_STRIDE + RP_W_OFF + c * RP_THIRD + wave * 2048);
  };
  auto pp_issue_a = [&](int slot) {
    if (ABL & 8) return;
    rl_dma1(voff_a0, reinterpret_cast<const char*>(sa) + ao2, lds0 + slot * RP_STRIDE + wave * 1024);
  };
  auto pp_advance = [&]() {                                        // end of a k-step: kt+2 becomes kt+1, the next rotated k-step becomes kt+2
    wo1 = wo2;
    if (++kq2 == nk) { kq2 = 0; wo2 = 0; ao2 = 0; } else { wo2 += p.wk; ao2 += RL_BK * 2; }
  };
  auto pp_prologue = [&]() {                                       // A | W0, W1, W2 of k-step 0 and A | W0 of k-step 1: 10 pieces per wave
    const int k0 = kmap(0), k1 = kmap(1);
    rl_dma1(voff_a0 + k0 * (RL_BK * 2), sa, lds0 + wave * 1024);
    rl_dma2(voff_w[0], voff_w[1], w0 + (int64_t)k0 * p.wk, lds0 + RP_W_OFF + wave * 2048);
    rl_dma2(voff_w[0], voff_w[1], w0 + third_b + (int64_t)k0 * p.wk, lds0 + RP_W_OFF + RP_THIRD + wave * 2048);
    rl_dma2(voff_w[0], voff_w[1], w0 + 2 * third_b + (int64_t)k0 * p.wk, lds0 + RP_W_OFF + 2 * RP_THIRD + wave * 2048);
    rl_dma1(voff_a0 + k1 * (RL_BK * 2), sa, lds0 + RP_STRIDE + wave * 1024);
    rl_dma2(voff_w[0], voff_w[1], w0 + (int64_t)k1 * p.wk, lds0 + RP_STRIDE + RP_W_OFF + wave * 2048);
    wo1 = (uint32_t)k1 * p.wk;
    kq2 = nk > 2 ? kmap(2) : 0;
    wo2 = (uint32_t)kq2 * p.wk; ao2 = (uint32_t)kq2 * (RL_BK * 2);
  };
  set_tile(m0);
  bool stage1_in_flight = false;
  if (PP) pp_prologue();
  else {
    stage(0, 0);
    if (nk > 1) { stage(1, 1); stage1_in_flight = true; }
  }

  float* stat = reinterpret_cast<float*>(smem + (PP ? RP_STAT_OFF : RL_STAT_OFF));
  float* pbias = reinterpret_cast<float*>(smem + (PP ? RP_BIAS_OFF : RL_PARAM_OFF));     // the epilogue reads bias / gamma / beta from LDS: a global load there
  float* pgamma = reinterpret_cast<float*>(smem + (PP ? RP_GB_OFF : RL_PARAM_OFF + RL_N * 4));   // would make hipcc wait vmcnt(0), i.e. for every residual piece
  float* pbeta = pgamma + RL_N;                                                              // still in flight
  for (int i = tid; i < RL_N; i += 512) {
    pbias[i] = p.bias ? p.bias[i] : 0.f;
    pgamma[i] = p.gamma[i];
    pbeta[i] = p.beta[i];
  }
  __syncthreads();
  bool y_stores_behind = false;                                    // PP: the previous epilogue's 48 Y stores were issued after this tile's first loads
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.X, (short)0, (int)(uint32_t)(p.M * p.ldx * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.Y, (short)0, (int)(uint32_t)(p.M * p.ldy * 2), 0x00020000);

  for (;;) {
    f32x16 acc[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (PP) {
      int fa[2], fw[2];                                            // fragment addresses in the CURRENT stage (bit 16 flipped once per k-step)
      {
        int ptid = threadIdx.x;
        asm volatile("" : "+v"(ptid));
        const int pl31 = ptid & 31, phi = (ptid & 63) >> 5;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int sw = ((ks * 2 + phi) ^ ((pl31 >> 2) & 3)) << 4;
          fa[ks] = (wm * 64 + pl31) * 64 + sw;                     // + i * 2048
          fw[ks] = RP_W_OFF + (wn * 64 + pl31) * 64 + sw;          // + c * RP_THIRD + jj * 2048
        }
      }
      bf16x8 a[2][2], w[2][2];
      auto mma = [&](auto Cc) {
        constexpr int C = decltype(Cc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              if (!(ABL & 16)) acc[i][2 * C + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], w[jj][ks], acc[i][2 * C + jj], 0, 0, 0);
              else if (ks == 0 && i == 0 && jj == 0) asm volatile("" :: "v"(a[0][0]), "v"(a[1][1]), "v"(w[0][0]), "v"(w[1][1]));
            }
        __builtin_amdgcn_s_setprio(0);
      };
      auto read_w = [&](auto Cc) {
        constexpr int C = decltype(Cc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) w[jj][ks] = *reinterpret_cast<const bf16x8*>(smem + C * RP_THIRD + jj * 2048 + fw[ks]);
      };
      // one k-step in stage S; `more1` = k-step kt+1 exists, `more2` = k-step kt+2 exists, `first` = first k-step of the tile
      auto kstep = [&](auto Sc, int kt, bool more1, bool more2, bool first) {
        constexpr int S = decltype(Sc)::value;
        const bool behind = SF_RL_STORECNT && first && y_stores_behind;
        // ---- phase 0: A x W third 0 ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) a[i][ks] = *reinterpret_cast<const bf16x8*>(smem + i * 2048 + fa[ks]);
        read_w(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        if (more1) pp_issue_w(1, S ^ 1);
        if (ABL & 8) rl_wait_vmcnt<0>();                           // W third 1 of this k-step has landed; the youngest stage stays in flight
        else if (!more1) rl_wait_vmcnt<2>();
        else if (behind) rl_wait_vmcnt<7 + 48>();
        else rl_wait_vmcnt<7>();
        rl_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        rl_barrier();
        // ---- phase 1: third 1 ----
        read_w(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (more1) pp_issue_w(2, S ^ 1);
#ifndef SF_RL_A_EARLY
#define SF_RL_A_EARLY 1   // the A piece of k-step kt+2 (the only piece of a stage that comes from HBM: ~2 us; W hits L2) is issued HERE, a phase earlier than its stage's W third 0: its slot - A of
#endif                    // this k-step - is free since phase 0 (fragments in registers, the wm = 1 waves one barrier behind have read them too), and the pieces issued behind it are needed late enough
        if (SF_RL_A_EARLY && more2) pp_issue_a(S);
        if (!more1 || (ABL & 8)) rl_wait_vmcnt<0>();               // W third 2 has landed
        else if (SF_RL_A_EARLY && more2) { if (behind) rl_wait_vmcnt<8 + 48>(); else rl_wait_vmcnt<8>(); }
        else if (behind) rl_wait_vmcnt<7 + 48>();
        else rl_wait_vmcnt<7>();
        rl_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        rl_barrier();
        // ---- phase 2: third 2; the fragment addresses move on to the other stage ----
        read_w(std::integral_constant<int, 2>{});
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(fa[ks]));
          asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(fw[ks]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more2) { if (!SF_RL_A_EARLY) pp_issue_a(S); pp_issue_w(0, S); }
        pp_advance();
        if (more1) {                                               // A | W third 0 of the next k-step have landed
          if (ABL & 8) rl_wait_vmcnt<0>();
          else if (!more2) rl_wait_vmcnt<4>();
          else if (behind) rl_wait_vmcnt<7 + 48>();
          else rl_wait_vmcnt<7>();
        }
        rl_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        rl_barrier();
      };
      // A | W third 0 of k-step 0 have landed (this wave's pieces), then everybody's
      if (ABL & 8) rl_wait_vmcnt<0>();
      else if (SF_RL_STORECNT && y_stores_behind) rl_wait_vmcnt<7 + 48>();
      else rl_wait_vmcnt<7>();
      rl_barrier();
      if (wm == 1) rl_barrier();                                   // the wm = 1 waves run one barrier behind
      for (int kt = 0; kt < nk; kt += 2) {
        const bool more = kt + 2 < nk;
        kstep(std::integral_constant<int, 0>{}, kt, true, more, kt == 0);
        kstep(std::integral_constant<int, 1>{}, kt + 1, more, more, false);
      }
      if (wm == 0) rl_barrier();                                   // re-align; also: every wave is done with both stages
    } else {
    for (int kt = 0; kt < nk; ++kt) {
      rl_wait_vmcnt_barrier<0>();                                  // stage kt landed everywhere; slot (kt+1)&1 is free
      const bool refill = kt + 1 < nk && !(kt == 0 && stage1_in_flight) && !(ABL & 8);
      const char* st = smem + (kt & 1) * RL_STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[2], b[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) b[j] = *reinterpret_cast<const bf16x8*>(st + b_base + j * 32 * 64 + frag_off[ks]);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 32 * 64 + frag_off[ks]);
        if (SF_RL_SPREAD) {
          // the vector-memory path takes ~16 cycles per 1-KiB piece and the issuing wave is held while it queues: seven back-to-back issues by
          // all eight waves right after the barrier leave the matrix pipe idle, one piece every third MFMA does not
          const uint32_t ko = kmap(kt + 1) * (RL_BK * 2), kow = kmap(kt + 1) * p.wk, la = lds_a_w + ((kt + 1) & 1) * RL_STAGE, lb = lds_b_w + ((kt + 1) & 1) * RL_STAGE;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
              if (!(ABL & 16)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
              else if (i == 0 && j == 0) acc[0][0][0] += (float)a[0][0] + (float)b[0][0];   // ablation: operands stay live, no matrix work
              const bool slot_now = SF_RL_SPREAD == 2 ? ((ks == 0 && (j == 0 || j == 2 || j == 4)) || (ks == 1 && i == 0 && j == 0))
                                                      : ((j == 1 || j == 4) && !(ks == 1 && i == 1 && j == 4));
              if (slot_now) {
                const int piece = SF_RL_SPREAD == 2 ? (ks == 1 ? 6 : i * 3 + (j >> 1)) : ks * 4 + i * 2 + (j == 4);   // 0 .. 6: A, W0 .. W5
                __builtin_amdgcn_sched_barrier(0);
                if (refill && !((ABL & 32) && (piece & 1) == 0 && piece > 0)) {
                  if (piece == 0) { if (SF_RL_A_NT) rl_dma1_nt(voff_a0 + ko, sa, la); else rl_dma1(voff_a0 + ko, sa, la); }
                  else rl_dma1(voff_b0 + kow, piece == 1 ? sb0 : piece == 2 ? sb1 : piece == 3 ? sb2 : piece == 4 ? sb3 : piece == 5 ? sb4 : sb5,
                               lb + (piece - 1) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
          if (ks == 0 && refill) {                                   // the refill's 7 LDS-DMA issues ride behind the first MFMA cluster
            __builtin_amdgcn_sched_barrier(0);
            stage((kt + 1) & 1, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // every wave is done with both slots -> they take the next tile's first two stages while this tile's epilogue runs
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    }
    const int64_t em0 = m0;
    const uint32_t tnext = t + gridDim.x;
    const bool more = tnext < p.tiles;

    // ---- epilogue, pass 1: x = acc + bias + residual, kept in registers in row-contiguous form ----------------------------------------
    // step s = c * 4 + g: 16-row group g (of 4) x 64-column chunk c (of 3) goes through this wave's slab; afterwards lane (lr, ecol) holds
    // rows g*16 + ps*4 + lr (ps < 4), columns c*64 + ecol .. +3 of the wave tile.  The residual rows arrive by LDS-DMA in exactly that
    // lane order (piece ps of a step = 4 rows x 256 B, lane i <- row i >> 4, 16 bytes at column (i & 15) * 4), three steps = 12 KiB per wave
    // in flight in the idle operand slots: no registers are spent on prefetch depth, and this pass issues no other vector-memory
    // operation, so the counted vmcnt waits below cover exactly these pieces (the X stores come after the pass: loads and stores complete
    // out of order with respect to each other, a counted wait over a mix would not be safe).
    // Every lane-derived quantity is re-derived here from an opaque copy of the thread id: computed up front, hipcc keeps ~10 of them
    // live across the main loop, where the 192 accumulators + 32 fragment registers leave no room (they spilled into the k-loop).
    int etid = threadIdx.x;
    asm volatile("" : "+v"(etid));
    const int elane = etid & 63, l31 = elane & 31, hi = elane >> 5, lr = elane >> 4, ecol = (elane & 15) * 4;
    const int gcol0 = wn * 192 + ecol;                             // + c * 64
    float* slab = reinterpret_cast<float*>(smem + (PP ? RP_SLAB_OFF : RL_SLAB_OFF) + wave * RL_SLAB_BYTES);
    const int ring_off = PP ? (wave >> 2) * RP_STRIDE + (wave & 3) * RL_RING_WAVE : wave * RL_RING_WAVE;   // PP: not across the statistics between the stages
    const char* ring = smem + ring_off + elane * 16;
    const uint32_t ring_lds = lds0 + ring_off;
    int64_t rrow0 = em0 + wm * 64;                                  // tail tile: rows beyond M re-read row M - 1 (their outputs are dropped)
    if (rrow0 > p.M - 1) rrow0 = p.M - 1;
    const int64_t rleft = p.M - 1 - rrow0;
    const int rlast = rleft < 63 ? (int)rleft : 63;
    const void* rbase = reinterpret_cast<const char*>(p.R) + rrow0 * p.ldr * 4;
    auto issue_res = [&](int s) {
      if (ABL & 1) return;
      const int c = s >> 2, g = s & 3;
      uint32_t vo[4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        int rl = g * 16 + ps * 4 + lr;
        if (rl > rlast) rl = rlast;
        vo[ps] = (uint32_t)(rl * (int)p.ldr + gcol0 + c * 64) * 4u;
      }
      rl_dma_r4(vo[0], vo[1], vo[2], vo[3], rbase, ring_lds + (s % 3) * 4096);
    };
    issue_res(0); issue_res(1); issue_res(2);
    float4 xr[4][3][4];
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const int c = s >> 2, g = s & 3, i = g >> 1, q2 = g & 1;
      const float4 bias4 = *reinterpret_cast<const float4*>(pbias + gcol0 + c * 64);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[(qq * 8 + hi * 4 + r) * 64 + jj * 32 + l31] = acc[i][2 * c + jj][(q2 * 2 + qq) * 4 + r];
      if (!(ABL & 1)) {                                             // step s has landed: only the (<= 2) younger steps may be outstanding
        if (s <= 9) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (s == 10) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        float4 v = *reinterpret_cast<const float4*>(slab + (ps * 4 + lr) * 64 + ecol);
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(ABL & 1)) rv = *reinterpret_cast<const float4*>(ring + (s % 3) * 4096 + ps * 1024);
        v.x = (v.x + bias4.x) + rv.x; v.y = (v.y + bias4.y) + rv.y; v.z = (v.z + bias4.z) + rv.z; v.w = (v.w + bias4.w) + rv.w;
        xr[g][c][ps] = v;
      }
      if (s + 3 < 12) {                                             // this step's ring slot has been read back: refill it with step s + 3
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_res(s + 3);
      }
    }
    // ---- X (fp32 stream) out: 48 row-contiguous 16-byte stores per lane, nothing waits for them --------------------------------------
    const int64_t row0 = em0 + wm * 64 + lr;
    const uint32_t xoff0 = (uint32_t)(row0 * p.ldx + gcol0) * 4u, xstep = (uint32_t)(4 * p.ldx) * 4u;
    if (!(ABL & 2)) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float4 v = xr[g][c][ps];
            rl_u32x4 o;
            o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
            __builtin_amdgcn_raw_buffer_store_b128(o, rx, xoff0 + (uint32_t)(g * 4 + ps) * xstep + (uint32_t)c * 256u, 0, SF_RL_STORE_AUX);
          }
    }

    // ---- pass 2: row sums (16-lane DPP sums, then the 4 column waves' partials through LDS) --------------------------------------
    const int srow0 = (wm * 64 + lr) * 4;                           // stat index of (g = 0, ps = 0); + (g * 16 + ps * 4) * 4
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        float sres = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) sres += (xr[g][c][ps].x + xr[g][c][ps].y) + (xr[g][c][ps].z + xr[g][c][ps].w);
        sres = rl_row16_sum(sres);
        if ((elane & 15) == 0) stat[srow0 + (g * 16 + ps * 4) * 4 + wn] = sres;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // raw barrier: __syncthreads() would also drain the 48 stores in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // every wave is past its residual ring: the operand slots take the next tile's first two stages while the rest of the epilogue runs
    stage1_in_flight = false;
    if (more) {
      m0 = (int64_t)tnext * RL_BM;
      set_tile(m0);
      if (PP) { pp_prologue(); y_stores_behind = true; }
      else {
        stage(0, 0);
        if (nk > 1) { stage(1, 1); stage1_in_flight = true; }
      }
    }
    // ---- pass 3: centred second moments (the mean of a row is re-derived from its 4 partials where it is needed: no 32 live registers) ----
    float* stat2 = stat + RL_BM * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const float4 q = *reinterpret_cast<const float4*>(stat + srow0 + (g * 16 + ps * 4) * 4);
        const float mu = ((q.x + q.y) + (q.z + q.w)) * (1.0f / RL_N);
        float sres = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float4 v = xr[g][c][ps];
          const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
          sres += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        sres = rl_row16_sum(sres);
        if ((elane & 15) == 0) stat2[srow0 + (g * 16 + ps * 4) * 4 + wn] = sres;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- pass 4: normalise, scale, shift, bf16, store ------------------------------------------------------------------------------
    const uint32_t yoff0 = (uint32_t)(row0 * p.ldy + gcol0) * 2u, ystep = (uint32_t)(4 * p.ldy) * 2u;
    float4 gm[3], bt[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gm[c] = *reinterpret_cast<const float4*>(pgamma + gcol0 + c * 64);
      bt[c] = *reinterpret_cast<const float4*>(pbeta + gcol0 + c * 64);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const float4 q = *reinterpret_cast<const float4*>(stat + srow0 + (g * 16 + ps * 4) * 4);
        const float4 q2 = *reinterpret_cast<const float4*>(stat2 + srow0 + (g * 16 + ps * 4) * 4);
        const float mu = ((q.x + q.y) + (q.z + q.w)) * (1.0f / RL_N);
        const float rs = rsqrtf(((q2.x + q2.y) + (q2.z + q2.w)) * (1.0f / RL_N) + p.eps);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float4 v = xr[g][c][ps];
          rl_u32x2 o;
          o.x = pack_bf2((v.x - mu) * rs * gm[c].x + bt[c].x, (v.y - mu) * rs * gm[c].y + bt[c].y);
          o.y = pack_bf2((v.z - mu) * rs * gm[c].z + bt[c].z, (v.w - mu) * rs * gm[c].w + bt[c].w);
          if (!(ABL & 4) || o.x == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b64(o, ry, yoff0 + (uint32_t)(g * 4 + ps) * ystep + (uint32_t)c * 128u, 0, SF_RL_STORE_AUX);
        }
      }
    if (!more) break;
    t = tnext;
  }
}

static thread_local int g_rl_force_sched = -1;   // test hook (per calling thread): -1 default, 0 round 2's loop, 1 quadrant-phased
extern "C" void sf_gemm_res_ln_force_schedule(int sched) { g_rl_force_sched = sched; }

int sf_gemm_res_ln768_v2_launch(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, const float* R, int64_t ldr, float* X, int64_t ldx,
                                const float* gamma, const float* beta, float eps, uint16_t* Y, int64_t ldy, int64_t M, int64_t K, void* stream);
#ifndef SF_RL_DEFAULT_SCHED
#define SF_RL_DEFAULT_SCHED SF_RL_PP
#endif
extern "C" int sf_gemm_res_ln768(const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, const float* R, int64_t ldr,
                                 float* X, int64_t ldx, const float* gamma, const float* beta, float eps, bf16_t* Y, int64_t ldy, int64_t M,
                                 int64_t K, void* stream) {
  SF_CHECK_ARG(A && W && R && X && gamma && beta && Y, "sf_gemm_res_ln768: null pointer");
  SF_CHECK_ARG(K > 0 && (K % RL_BK) == 0 && K < (1 << 20), "sf_gemm_res_ln768: K=%lld must be a positive multiple of 32", (long long)K);
  const bool w_kmajor = ldw == RL_BK && K > RL_BK;                 // W given k-step-major: [K / 32][768][32] (every stage's 48 KiB slice contiguous)
  SF_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0 && lda >= K && (ldw >= K || w_kmajor), "sf_gemm_res_ln768: lda/ldw must be >= K and multiples of 8 elements");
  SF_CHECK_ARG((ldr % 4) == 0 && (ldx % 4) == 0 && (ldy % 4) == 0 && ldr >= RL_N && ldx >= RL_N && ldy >= RL_N,
               "sf_gemm_res_ln768: ldr/ldx/ldy must be >= 768 and multiples of 4 elements");
  SF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)R % 16) == 0 && ((uintptr_t)X % 16) == 0 &&
                   ((uintptr_t)Y % 8) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0),
               "sf_gemm_res_ln768: operands must be 16-byte aligned");
  if (M <= 0) return 0;
  // Round 4, schedule 2 (sf_gemm_ln2.hip: 192 complete rows per workgroup in two column passes on the 192 x 384 main loop of sf_qkv_space_attention): taken for a
  // ROW-MAJOR weight when selected (sf_gemm_res_ln_force_schedule(2), or SF_RL_SCHED=2) and the shape fits (K % 128 == 0, lda / ldw multiples of 64)
  {
    static int sched2 = -1;
    if (sched2 < 0) { const char* e = getenv("SF_RL_SCHED"); sched2 = (e ? atoi(e) : SF_RL_DEFAULT_SCHED) == 2 ? 1 : 0; }
    const bool want2 = g_rl_force_sched == 2 || (g_rl_force_sched < 0 && sched2);
#ifdef SF_ABLATION
    if (want2 && !w_kmajor && (K % 128) == 0 && K >= 128 && (lda % 64) == 0 && (ldw % 64) == 0 && ldw >= K)
      return sf_gemm_res_ln768_v2_launch(A, lda, W, ldw, bias, R, ldr, X, ldx, gamma, beta, eps, Y, ldy, M, K, stream);
#else
    if (want2) { sf_set_error("sf_gemm_res_ln768: schedule 2 is a measured-slower alternative that only the ablation build carries (libsynchformer_hip_ablation.so, -DSF_ABLATION)"); return -1; }
#endif
  }
  const int64_t m_pad = ((M + RL_BM - 1) / RL_BM) * RL_BM;
  // 32-bit byte offsets: buffer descriptors for R / X / Y (rows >= M are dropped by the hardware range check) and the lane offsets of the
  // LDS-DMA (16 rows of A or W plus the k offset)
  SF_CHECK_ARG(m_pad * ldr * 4 < ((int64_t)1 << 32) && m_pad * ldx * 4 < ((int64_t)1 << 32) && m_pad * ldy * 2 < ((int64_t)1 << 32),
               "sf_gemm_res_ln768: R / X / Y must stay below 4 GiB");
  SF_CHECK_ARG(128 * lda * 2 + K * 2 < ((int64_t)1 << 31) && 16 * ldw * 2 + K * 2 < ((int64_t)1 << 31), "sf_gemm_res_ln768: row strides too large");
  const int n_cu = sf_cu_count("sf_gemm_res_ln768");
  if (n_cu <= 0) return -1;
  ResLnArgs a;
  a.wk = w_kmajor ? (uint32_t)(RL_N * RL_BK * 2) : (uint32_t)(RL_BK * 2);
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.R = R; a.ldr = ldr; a.X = X; a.ldx = ldx; a.gamma = gamma; a.beta = beta;
  a.Y = Y; a.ldy = ldy; a.M = M; a.K = (int)K; a.eps = eps;
  const int64_t tiles = m_pad / RL_BM;
  SF_CHECK_ARG(tiles < ((int64_t)1 << 31), "sf_gemm_res_ln768: too many tiles");
  a.tiles = (uint32_t)tiles;
  { static int st = -1; if (st < 0) { const char* e = getenv("SF_RL_STAGGER"); st = e ? atoi(e) : 12; if (st < 0) st = 0; } a.stagger = K >= 768 ? (uint32_t)st : 0u; }   // (a tile of a shorter k-loop is shorter than the delay)
  { static int ph = -1; if (ph < 0) { const char* e = getenv("SF_RL_PHASES"); ph = e ? atoi(e) : 1; if (ph < 1) ph = 1; } a.phases = (uint32_t)ph; }
  int64_t blocks = tiles < n_cu ? tiles : n_cu;                    // one persistent workgroup per CU
  static int abl = -1, max_blocks = -1, sched = -1;
  if (max_blocks < 0) { const char* e = getenv("SF_RL_BLOCKS"); max_blocks = e ? atoi(e) : 0; }   // measurement hook: fewer resident workgroups
  if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
  if (abl < 0) { const char* e = getenv("SF_RL_ABL"); abl = e ? atoi(e) : 0; }
  if (sched < 0) { const char* e = getenv("SF_RL_SCHED"); sched = e ? atoi(e) : SF_RL_PP; }       // 0: round 2's loop, 1: quadrant-phased (needs K % 64 == 0)
  const bool pp = (sched != 0 || g_rl_force_sched == 1) && g_rl_force_sched != 0 && (K % 64) == 0 && K >= 128;
  const dim3 grid((unsigned)blocks), blk(512);
  hipStream_t st = (hipStream_t)stream;
#define RL_LAUNCH(ABL_, PP_, LDS_) do { if (int rc_ = sf_prepare_kernel((const void*)gemm_res_ln768_kernel<ABL_, PP_>, LDS_, "sf_gemm_res_ln768")) return rc_; \
    hipLaunchKernelGGL((gemm_res_ln768_kernel<ABL_, PP_>), grid, blk, LDS_, st, a); } while (0)
  if (pp) {
    switch (abl) {     // ablations: 1 no residual, 2 no X stores, 4 no Y stores, 8 no refills after the first two stages, 16 no MFMA
      case 0: RL_LAUNCH(0, true, RP_LDS); break;
#ifdef SF_ABLATION
      case 1: RL_LAUNCH(1, true, RP_LDS); break;     // round 6: which part of the epilogue's memory traffic costs what (profiles/r06_gemm_ln_epilogue.md)
      case 4: RL_LAUNCH(4, true, RP_LDS); break;
#endif
      case 7: RL_LAUNCH(7, true, RP_LDS); break;
      case 15: RL_LAUNCH(15, true, RP_LDS); break;
      case 23: RL_LAUNCH(23, true, RP_LDS); break;
      default: sf_set_error("sf_gemm_res_ln768: unknown SF_RL_ABL %d for the quadrant-phased schedule", abl); return -1;
    }
  } else {
    switch (abl) {
      case 0: RL_LAUNCH(0, false, RL_LDS); break;
      case 1: RL_LAUNCH(1, false, RL_LDS); break;
      case 15: RL_LAUNCH(15, false, RL_LDS); break;
      case 17: RL_LAUNCH(17, false, RL_LDS); break;
      default: sf_set_error("sf_gemm_res_ln768: unknown SF_RL_ABL %d", abl); return -1;
    }
  }
  SF_LAUNCH_CHECK();
  return 0;
}
