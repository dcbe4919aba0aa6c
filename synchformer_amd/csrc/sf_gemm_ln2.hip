// sf_gemm_res_ln768, schedule 2 (round 4): the full-row projection + bias + fp32 residual + the NEXT LayerNorm on the main loop of sf_qkv_space.hip.
//
// The round-2/3 kernel (sf_gemm_ln.hip) gives one workgroup 128 COMPLETE rows (a 128 x 768 tile: 48 KiB of W per 32-deep k-step for 128 rows, one 56-KiB stage in
// flight) so that the LayerNorm statistics stay in registers; it is bound by operand delivery (K = 3072: 1469 of 1557 us remain with the MFMAs removed,
// profiles/r03_gemm_ln_ablation.md).  This schedule gives a workgroup 192 complete rows in TWO COLUMN PASSES of 384: each pass is the 192 x 384 tile of
// sf_qkv_space_attention (2 x 4 waves of 96 x 96, three phases per 64-deep k-tile with 12 MFMAs per wave and phase, counted vmcnt waits, staggered wave groups: 1.3 PFLOP/s
// there), the operand bytes per MAC of the 256 x 256 tile.  The price is the row statistics across the passes:
//   pass 0: X[:, 0:384] = acc + bias + R is written (fp32);
//   pass 1: X[:, 384:768] likewise; then per row the first half is read back (the same wave wrote it: 1.5 KB out of L2), mean and CENTERED variance over all 768 values,
//           and Y = LayerNorm(X) * gamma + beta leaves as bf16.
// Epilogue data path: the accumulators are transposed (a lane holds one token's features), so they go through LDS in six chunks of 32 rows x 384 fp32 (row stride 1552 B,
// two buffers; the fp32 residual of chunk n+1 lands by LDS-DMA while chunk n is processed and the four waves that hold the chunk ADD their accumulators to it in LDS) and come
// back ROW-wise - a wave owns 4 rows of a chunk, a lane 4 (+4) consecutive floats of a row - so that the residual loads and the X / Y stores are whole row segments.  No
// cross-workgroup exchange, no atomics.
// MEASURED SLOWER than schedule 1 (proj 991 us against 672 us, fc2 2003 us against 1551 us at 224 segments: the epilogue is not hidden under a main loop, 7.15 rounds of
// tiles; profiles/r04_experiments.md section 6): kept as the tested alternative behind sf_gemm_res_ln_force_schedule(2) / SF_RL_SCHED=2, not used by the engine.
#ifdef SF_ABLATION   // schedule 2 of sf_gemm_res_ln768: a measured-slower alternative, compiled into the ablation build only
#include "sf_common.h"
#include <type_traits>
#include <stdlib.h>
#include "../../include/synchformer_hip.h"

#define G2_ROWS 192
#define G2_COLS 384
#define G2_N 768
#define G2_A_BYTES (G2_ROWS * 128)      // 24 KiB: 192 rows x 64 k (bf16)
#define G2_W_PART (128 * 128)           // 16 KiB: 4 wave columns x 32 features x 64 k
#define G2_STAGE (G2_A_BYTES + 3 * G2_W_PART)   // 72 KiB
#define G2_CH_LD 1552                   // bytes per staged fp32 row: 384 floats + 16
#define G2_BIAS_OFF (2 * G2_STAGE + 1024) // 145 KiB: the pass's 384 bias floats
#define G2_GB_OFF (G2_BIAS_OFF + 1536)  // gamma | beta (768 floats each), staged once per workgroup
#define G2_LDS (160 * 1024)
#ifndef G2_ABL
#define G2_ABL 0                        // measurement builds: 1 no epilogue memory traffic (no residual loads, no stores), 2 no MFMAs
#endif

struct G2Args {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  const float* R; int64_t ldr;
  float* X; int64_t ldx;
  const float* gamma; const float* beta;
  bf16_t* Y; int64_t ldy;
  int64_t M; int K;
  float eps;
  uint32_t tiles;
};

__device__ __forceinline__ void g2_dma1(uint32_t voff, const void* sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}
__device__ __forceinline__ void g2_dma_dword_addr(const void* gaddr, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gaddr), "s"(lds) : "memory");
}
template <int N>
__device__ __forceinline__ void g2_wait_vmcnt() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void g2_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint32_t g2_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
// sum over the 64 lanes of a wave, every lane ends with the total: 4 DPP steps inside the 16-lane rows, then the two cross-row exchanges
__device__ __forceinline__ float g2_wave_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <int V> using g2_ic = std::integral_constant<int, V>;

__global__ __launch_bounds__(512, 2) void gemm_res_ln768_v2_kernel(G2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                         // 2 x 4 waves, wave tile 96 rows x 96 features
  const int hi = lane >> 5;

  // persistent schedule: block b sits on XCD b % 8; every XCD owns a contiguous range of 192-row tiles
  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t t8 = (p.tiles + 7u) >> 3;
  const uint32_t t0 = min(xcd * t8, p.tiles), t1 = min(t0 + t8, p.tiles);
  uint32_t t = t0 + li;
  if (t >= t1) return;

  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(g2_lds_addr(smem));
  const uint32_t lds_a_w = __builtin_amdgcn_readfirstlane(lds0 + wave * 3072);
  const uint32_t lds_w_w = __builtin_amdgcn_readfirstlane(lds0 + G2_A_BYTES + wave * 2048);
  const uint32_t w8 = (uint32_t)(8 * p.ldw * 2);
  const int nk = p.K >> 6;                                          // 64-deep k-tiles (K % 128 == 0: an even number)
  if (tid < 384) {                                                  // gamma | beta -> LDS (read per row group in the pass-1 epilogue)
    const int w_ = tid < 192 ? 0 : 1, c_ = (tid - w_ * 192) * 4;
    *reinterpret_cast<float4*>(smem + G2_GB_OFF + (w_ * G2_N + c_) * 4) = *reinterpret_cast<const float4*>((w_ ? p.beta : p.gamma) + c_);
  }

  for (;;) {
    const int64_t row0 = (int64_t)t * G2_ROWS;
    const int64_t rows_left = p.M - row0;                           // >= 1
    const char* xbase = reinterpret_cast<const char*>(p.A + row0 * p.lda);
    // lane offsets of the LDS-DMA pieces, re-derived per tile (rows beyond M re-read the last valid row: their results are never stored)
    uint32_t voff_a[3], voff_w[3];
    {
      int dtid = threadIdx.x;
      asm volatile("" : "+v"(dtid));
      const int dl = dtid & 63;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const int r = (wave * 3 + pc) * 8 + (dl >> 3);
        const int64_t rc = r < rows_left ? r : rows_left - 1;
        voff_a[pc] = (uint32_t)(rc * p.lda * 2 + ((((dl & 7) ^ ((r >> 1) & 7))) << 4));
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int pr = wave * 2 * 8 + (dl >> 3);
        const int c = (pr >> 5) * 96 + j * 32 + (pr & 31);        // tile column = row of the pass's 384-row W slice
        voff_w[j] = (uint32_t)((int64_t)c * p.ldw * 2 + ((((dl & 7) ^ ((pr >> 1) & 7))) << 4));
      }
    }
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const char* wbase = reinterpret_cast<const char*>(p.W + (int64_t)pass * G2_COLS * p.ldw);
      auto issue_a = [&](int pc, int S, int kt) { g2_dma1(voff_a[pc], xbase + kt * 128, lds_a_w + S * G2_STAGE + pc * 1024); };
      auto issue_w = [&](int j, int S, int kt) {
        g2_dma1(voff_w[j], wbase + kt * 128, lds_w_w + S * G2_STAGE + j * G2_W_PART);
        g2_dma1((voff_w[j] ^ 64u) + w8, wbase + kt * 128, lds_w_w + S * G2_STAGE + j * G2_W_PART + 1024);
      };
      // ---- prologue (every wave is out of the staging chunks: barrier at the bottom of the previous epilogue) ----
      if (wave < 6 && p.bias) g2_dma_dword_addr(p.bias + pass * G2_COLS + wave * 64 + lane, lds0 + G2_BIAS_OFF + wave * 256);
      issue_w(0, 0, 0); issue_a(0, 0, 0);
      issue_w(1, 0, 0); issue_a(1, 0, 0); issue_a(2, 0, 0);
      issue_w(2, 0, 0);
      issue_w(0, 1, 1); issue_a(0, 1, 1);
      g2_wait_vmcnt<5>();
      g2_barrier();
      f32x16 acc[3][3];
      {
        const float* bs = reinterpret_cast<const float*>(smem + G2_BIAS_OFF);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(bs + wn * 96 + j * 32 + g * 8 + hi * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 3; ++i) { acc[j][i][g * 4 + 0] = b4.x; acc[j][i][g * 4 + 1] = b4.y; acc[j][i][g * 4 + 2] = b4.z; acc[j][i][g * 4 + 3] = b4.w; }
          }
      }
      {
        int fo[4];
        {
          int ptid = threadIdx.x;
          asm volatile("" : "+v"(ptid));
          const int pl31 = ptid & 31, phi = (ptid & 63) >> 5;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) fo[kk] = pl31 * 128 + (((kk * 2 + phi) ^ ((pl31 >> 1) & 7)) << 4);
        }
        const int a_base = wm * 96 * 128, w_base = G2_A_BYTES + wn * 32 * 128;
        bf16x8 xf[3][4], wf[4];
        auto read_w = [&](const char* st, int j) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) wf[kk] = *reinterpret_cast<const bf16x8*>(st + w_base + j * G2_W_PART + fo[kk]);
        };
        auto mma = [&](auto Jc) {
          constexpr int J = decltype(Jc)::value;
          __builtin_amdgcn_s_setprio(1);
          if (!(G2_ABL & 2)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
              for (int i = 0; i < 3; ++i) acc[J][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk], xf[i][kk], acc[J][i], 0, 0, 0);
          } else asm volatile("" :: "v"(wf[0]), "v"(wf[3]), "v"(xf[0][0]), "v"(xf[2][3]));
          asm volatile("" : "+v"(acc[J][0]), "+v"(acc[J][1]), "+v"(acc[J][2]));
          __builtin_amdgcn_s_setprio(0);
        };
        // one k-tile held in stage S; ld1 / ld2: k-tiles kt+1 / kt+2 exist (the schedule of sf_qkv_space.hip)
        auto ktile = [&](auto Sc, int kt, bool ld1, bool ld2) {
          constexpr int S = decltype(Sc)::value;
          const char* st = smem + S * G2_STAGE;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xf[i][kk] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 4096 + fo[kk]);
          read_w(st, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (ld1) { issue_w(1, S ^ 1, kt + 1); issue_a(1, S ^ 1, kt + 1); issue_a(2, S ^ 1, kt + 1); }
          g2_barrier();
          __builtin_amdgcn_sched_barrier(0);
          mma(g2_ic<0>{});
          __builtin_amdgcn_sched_barrier(0);
          g2_barrier();
          read_w(st, 1);
          __builtin_amdgcn_sched_barrier(0);
          if (ld1) { issue_w(2, S ^ 1, kt + 1); g2_wait_vmcnt<9>(); } else g2_wait_vmcnt<0>();
          g2_barrier();
          __builtin_amdgcn_sched_barrier(0);
          mma(g2_ic<1>{});
          __builtin_amdgcn_sched_barrier(0);
          g2_barrier();
          read_w(st, 2);
          __builtin_amdgcn_sched_barrier(0);
          if (ld2) { issue_w(0, S, kt + 2); issue_a(0, S, kt + 2); g2_wait_vmcnt<5>(); }
          else if (ld1) g2_wait_vmcnt<2>();
          else g2_wait_vmcnt<0>();
          g2_barrier();
          __builtin_amdgcn_sched_barrier(0);
          mma(g2_ic<2>{});
          __builtin_amdgcn_sched_barrier(0);
          g2_barrier();
        };
        if (wm == 1) g2_barrier();                                  // waves 4-7 run one barrier behind waves 0-3
#pragma unroll 1
        for (int kt = 0; kt < nk; kt += 2) {
          ktile(g2_ic<0>{}, kt, true, kt + 2 < nk);
          ktile(g2_ic<1>{}, kt + 1, kt + 2 < nk, kt + 3 < nk);
        }
        if (wm == 0) g2_barrier();                                  // re-align; every wave is done with both stages
      }

      // ---- epilogue of the pass: six chunks of 32 rows (row block i of wave row wmc) through two LDS buffers --------------------------------------------------
      // The fp32 residual of chunk n+1 lands by LDS-DMA (whole 1.5-KiB row segments, no registers) while chunk n is processed; the four waves that hold the chunk ADD their
      // accumulators to it in LDS; then every wave takes 4 whole rows: X store, row statistics, (pass 1) the first half back from L2, LayerNorm, Y store.
      {
        int etid = threadIdx.x;
        asm volatile("" : "+v"(etid));
        const int el = etid & 63, el31 = etid & 31, ehi = el >> 5;
        const int c1 = el * 4, c2 = 256 + el31 * 4;                 // this lane's columns of a row half: 4 floats at c1 (all lanes), 4 at c2 (lanes 0-31)
        const bool has2 = el < 32;
        const int col0 = pass * G2_COLS;
        const float* gbl = reinterpret_cast<const float*>(smem + G2_GB_OFF);
        auto issue_r = [&](int n) {                                 // residual rows 4 wave .. + 3 of chunk n -> buffer n & 1 (rows beyond M re-read the last valid row)
          const int wmc = n >= 3 ? 1 : 0, i = n - 3 * wmc;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = wave * 4 + q;
            int64_t gr = row0 + wmc * 96 + i * 32 + r;
            if (gr > p.M - 1) gr = p.M - 1;
            const char* src = reinterpret_cast<const char*>(p.R + gr * p.ldr + col0);
            const uint32_t dst = lds0 + (n & 1) * (32 * G2_CH_LD) + r * G2_CH_LD;
            g2_dma1((uint32_t)el * 16u, src, dst);
            if (has2) g2_dma1((uint32_t)el * 16u, src + 1024, dst + 1024);
          }
        };
        if (!(G2_ABL & 1)) issue_r(0);
#pragma unroll 1
        for (int n = 0; n < 6; ++n) {
          const int wmc = n >= 3 ? 1 : 0, i = n - 3 * wmc;
          char* buf = smem + (n & 1) * (32 * G2_CH_LD);
          if (!(G2_ABL & 1)) {
            if (n + 1 < 6) { issue_r(n + 1); g2_wait_vmcnt<8>(); } else g2_wait_vmcnt<0>();     // the residual of chunk n has landed (and the stores of chunk n-1 have retired)
          }
          g2_barrier();
          // (a) the four waves of wave row wmc add their row block i: chunk row l31, columns wn * 96 + 32 j + 8 g + 4 hi
          if (wm == wmc) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                float4 v;
                if (i == 0) v = make_float4(acc[j][0][g * 4], acc[j][0][g * 4 + 1], acc[j][0][g * 4 + 2], acc[j][0][g * 4 + 3]);
                else if (i == 1) v = make_float4(acc[j][1][g * 4], acc[j][1][g * 4 + 1], acc[j][1][g * 4 + 2], acc[j][1][g * 4 + 3]);
                else v = make_float4(acc[j][2][g * 4], acc[j][2][g * 4 + 1], acc[j][2][g * 4 + 2], acc[j][2][g * 4 + 3]);
                float4* dst = reinterpret_cast<float4*>(buf + el31 * G2_CH_LD + (wn * 96 + j * 32 + g * 8 + ehi * 4) * 4);
                if (!(G2_ABL & 1)) { const float4 r4 = *dst; v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
                *dst = v;
              }
          }
          g2_barrier();
          // (b) rows 4 wave .. + 3 of the chunk, whole rows per wave
          {
            float y1[4][4], y2[4][4];
            int64_t grow[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int64_t gr = row0 + wmc * 96 + i * 32 + wave * 4 + q;
              grow[q] = gr < p.M ? gr : -1;
              float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f), r2 = r1;
              if (pass == 1 && grow[q] >= 0 && !(G2_ABL & 1)) {       // the first half of the row back from L2 (this wave wrote it in pass 0)
                r1 = *reinterpret_cast<const float4*>(p.X + gr * p.ldx + c1);
                if (has2) r2 = *reinterpret_cast<const float4*>(p.X + gr * p.ldx + c2);
              }
              y1[q][0] = r1.x; y1[q][1] = r1.y; y1[q][2] = r1.z; y1[q][3] = r1.w;
              y2[q][0] = r2.x; y2[q][1] = r2.y; y2[q][2] = r2.z; y2[q][3] = r2.w;
            }
            // the x values of a row are read from the chunk every time they are needed (sums, centered squares, LayerNorm): LDS reads are cheap here, registers are
            // not - the accumulators of the row blocks still to come stay live next to this
            float s[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = wave * 4 + q;
              const float4 a1 = *reinterpret_cast<const float4*>(buf + r * G2_CH_LD + c1 * 4);
              float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f);
              if (has2) a2 = *reinterpret_cast<const float4*>(buf + r * G2_CH_LD + c2 * 4);
              if (grow[q] >= 0 && !(G2_ABL & 1)) {
                *reinterpret_cast<float4*>(p.X + grow[q] * p.ldx + col0 + c1) = a1;
                if (has2) *reinterpret_cast<float4*>(p.X + grow[q] * p.ldx + col0 + c2) = a2;
              }
              s[q] = (a1.x + a1.y) + (a1.z + a1.w) + (has2 ? (a2.x + a2.y) + (a2.z + a2.w) : 0.f);
              if (pass == 1) s[q] += (y1[q][0] + y1[q][1]) + (y1[q][2] + y1[q][3]) + (has2 ? (y2[q][0] + y2[q][1]) + (y2[q][2] + y2[q][3]) : 0.f);
            }
            if (pass == 1) {                                        // (pass 0 only stores its half of X: the statistics need the whole row)
#pragma unroll
              for (int q = 0; q < 4; ++q) s[q] = g2_wave_sum(s[q]);
              float vs[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int r = wave * 4 + q;
                s[q] *= (1.0f / 768.0f);                            // the mean
                const float4 a1 = *reinterpret_cast<const float4*>(buf + r * G2_CH_LD + c1 * 4);
                const float xa[4] = {a1.x, a1.y, a1.z, a1.w};
                float v = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d1 = xa[e] - s[q], d3 = y1[q][e] - s[q]; v += d1 * d1 + d3 * d3; }
                if (has2) {
                  const float4 a2 = *reinterpret_cast<const float4*>(buf + r * G2_CH_LD + c2 * 4);
                  const float xb[4] = {a2.x, a2.y, a2.z, a2.w};
#pragma unroll
                  for (int e = 0; e < 4; ++e) { const float d2 = xb[e] - s[q], d4 = y2[q][e] - s[q]; v += d2 * d2 + d4 * d4; }
                }
                vs[q] = v;
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) vs[q] = g2_wave_sum(vs[q]);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                if (grow[q] < 0 || (G2_ABL & 1)) continue;
                const int r = wave * 4 + q;
                const float rstd = rsqrtf(vs[q] * (1.0f / 768.0f) + p.eps), m = s[q];
                bf16_t* yrow = p.Y + grow[q] * p.ldy;
                {
                  const float4 a1 = *reinterpret_cast<const float4*>(buf + r * G2_CH_LD + c1 * 4);
                  const float4 ga = *reinterpret_cast<const float4*>(gbl + c1), ba = *reinterpret_cast<const float4*>(gbl + G2_N + c1);
                  const float4 gb = *reinterpret_cast<const float4*>(gbl + G2_COLS + c1), bb = *reinterpret_cast<const float4*>(gbl + G2_N + G2_COLS + c1);
                  *reinterpret_cast<uint2*>(yrow + c1) = make_uint2(pack_bf2((y1[q][0] - m) * rstd * ga.x + ba.x, (y1[q][1] - m) * rstd * ga.y + ba.y),
                                                                    pack_bf2((y1[q][2] - m) * rstd * ga.z + ba.z, (y1[q][3] - m) * rstd * ga.w + ba.w));
                  *reinterpret_cast<uint2*>(yrow + G2_COLS + c1) = make_uint2(pack_bf2((a1.x - m) * rstd * gb.x + bb.x, (a1.y - m) * rstd * gb.y + bb.y),
                                                                              pack_bf2((a1.z - m) * rstd * gb.z + bb.z, (a1.w - m) * rstd * gb.w + bb.w));
                }
                if (has2) {
                  const float4 a2 = *reinterpret_cast<const float4*>(buf + r * G2_CH_LD + c2 * 4);
                  const float4 ga = *reinterpret_cast<const float4*>(gbl + c2), ba = *reinterpret_cast<const float4*>(gbl + G2_N + c2);
                  const float4 gb = *reinterpret_cast<const float4*>(gbl + G2_COLS + c2), bb = *reinterpret_cast<const float4*>(gbl + G2_N + G2_COLS + c2);
                  *reinterpret_cast<uint2*>(yrow + c2) = make_uint2(pack_bf2((y2[q][0] - m) * rstd * ga.x + ba.x, (y2[q][1] - m) * rstd * ga.y + ba.y),
                                                                    pack_bf2((y2[q][2] - m) * rstd * ga.z + ba.z, (y2[q][3] - m) * rstd * ga.w + ba.w));
                  *reinterpret_cast<uint2*>(yrow + G2_COLS + c2) = make_uint2(pack_bf2((a2.x - m) * rstd * gb.x + bb.x, (a2.y - m) * rstd * gb.y + bb.y),
                                                                              pack_bf2((a2.z - m) * rstd * gb.z + bb.z, (a2.w - m) * rstd * gb.w + bb.w));
                }
              }
            }
          }
          g2_barrier();                                             // the chunk is consumed: its buffer may take the residual of chunk n + 2
        }
      }
    }
    t += per_xcd_blocks;
    if (t >= t1) break;
  }
}

// The launcher of schedule 2 (see sf_gemm_res_ln768 in sf_gemm_ln.hip, which dispatches here for a row-major W when the schedule is selected): W (768, K) bf16 row-major,
// K % 128 == 0, row strides multiples of 64 elements (the chunk slot lives in the low 7 bits of a piece's byte offset), R / X / Y below 4 GiB is NOT required (64-bit row
// addressing in the epilogue); A may be any readable (M, K) matrix: the ragged last tile re-reads its last valid row.
int sf_gemm_res_ln768_v2_launch(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, const float* R, int64_t ldr, float* X, int64_t ldx,
                                const float* gamma, const float* beta, float eps, uint16_t* Y, int64_t ldy, int64_t M, int64_t K, void* stream) {
  SF_CHECK_ARG(A && W && R && X && gamma && beta && Y, "sf_gemm_res_ln768 (schedule 2): null pointer");
  SF_CHECK_ARG(K >= 128 && (K % 128) == 0 && (lda % 64) == 0 && (ldw % 64) == 0 && lda >= K && ldw >= K, "sf_gemm_res_ln768 (schedule 2): K %% 128 == 0, lda / ldw multiples of 64");
  SF_CHECK_ARG((ldr % 4) == 0 && (ldx % 4) == 0 && (ldy % 4) == 0 && ldr >= G2_N && ldx >= G2_N && ldy >= G2_N, "sf_gemm_res_ln768 (schedule 2): bad row strides of R / X / Y");
  SF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)R % 16) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 8) == 0 &&
                   ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0), "sf_gemm_res_ln768 (schedule 2): operands must be 16-byte aligned");
  SF_CHECK_ARG((int64_t)G2_ROWS * lda * 2 < ((int64_t)1 << 32) && (int64_t)G2_COLS * ldw * 2 < ((int64_t)1 << 32), "sf_gemm_res_ln768 (schedule 2): a tile of A / a slice of W must stay below 4 GiB");
  if (M <= 0) return 0;
  const int64_t tiles = (M + G2_ROWS - 1) / G2_ROWS;
  SF_CHECK_ARG(tiles < ((int64_t)1 << 31), "sf_gemm_res_ln768 (schedule 2): too many tiles");
  if (int rc = sf_prepare_kernel((const void*)gemm_res_ln768_v2_kernel, G2_LDS, "sf_gemm_res_ln768")) return rc;
  const int n_cu = sf_cu_count("sf_gemm_res_ln768");
  if (n_cu <= 0) return -1;
  G2Args a;
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.R = R; a.ldr = ldr; a.X = X; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.Y = Y; a.ldy = ldy;
  a.M = M; a.K = (int)K; a.eps = eps; a.tiles = (uint32_t)tiles;
  int64_t blocks = (n_cu / 8) * 8;
  if (blocks < 8) blocks = 8;                                    // (a device / partition with fewer than 8 CUs: never an empty grid)
  const int64_t need = ((tiles + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL(gemm_res_ln768_v2_kernel, dim3((unsigned)blocks), dim3(512), G2_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}

#endif  // SF_ABLATION
