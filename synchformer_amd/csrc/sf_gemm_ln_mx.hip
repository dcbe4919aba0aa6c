// sf_gemm_res_ln768 on MXFP8 operands, with an MXFP8 output (gfx950 / MI355X):
//     X[m, :] = dq(A[m, :]) * dq(W)^T + bias + R[m, :]        (fp32 residual stream, 768 columns; X may alias R)
//     Y[m, :] = mxfp8( LayerNorm(X[m, :]) * gamma + beta )     (e4m3 bytes + one E8M0 scale per 32 columns: the A operand of the NEXT MX GEMM; Y / sY may
//                                                               alias A / sA when K == 768 - a workgroup owns whole rows and has read them before it writes)
// the `x = x + proj(attn)` / `x = x + fc2(gelu(fc1))` -> next LayerNorm step of a DividedSpaceTimeBlock (vit_helper.py:364-376) in the frozen fp8 towers of
// the synchronizability fine-tune (BASELINE configs[4]).  Un-fused (round 2) this was sf_gemm_mxfp8 with the residual epilogue + sf_layernorm768_mxfp8:
// the second launch re-read the whole fp32 stream (1.0 GB at 13 x 16 segments, 236 us x 36 launches = 8 % of the step).
//
// Structure = the quadrant-phased gemm_res_ln768_kernel (sf_gemm_ln.hip) with the SAME BYTE geometry: a k-step is 64 fp8 = 64 BYTES per row (there: 32 bf16),
// a stage 128 x 64 B of A + 768 x 64 B of W, three phases per k-step (one per third of the wave's 192 columns), 7 operand pieces per wave per k-step,
// wm = 1 waves one barrier behind, stage stride 64 KiB.  A lane's two 16-byte fragment reads of a row (chunks hi and 2 + hi: the two k-halves of the bf16
// step) are exactly the 2 x 16 bytes it supplies to one v_mfma_scale_f32_32x32x64_f8f6f4 (16 bytes of each of the row's two 32-k MX blocks, see
// sf_gemm_mx.hip): 4 scaled MFMAs per phase (64 cycles each) where the bf16 kernel has 8 (32 cycles each) - twice the k per byte and per cycle.
// Scales: the stage-major planes hold one dword per row per 128 k = per PAIR of k-steps; lanes < 32 supply the scale of the step's first block, lanes >= 32 of
// its second (ds_read_u8 of byte 2 * (step & 1) + hi of the dword).  A pair's 768 + 128 dwords travel as an EIGHTH piece per wave per k-step (global_load_lds_dword, 64 dwords):
// with the A | W0 group of k-step k comes half (k & 1 ? 0 : 1) of pair (k + 1) >> 1 - half 0 = W rows 0-511, half 1 = W rows 512-767 | the tile's 128 A rows |
// 2 dummy pieces (every wave issues the same number of loads, so every counted wait is vmcnt(8)).  Three 4-KiB scale slots (pair % 3) live in the
// epilogue's transposition slabs, which are idle while the main loop runs.
// Epilogue = passes 1-3 of the bf16 kernel; pass 4 quantises like sf_layernorm768_mxfp8 (bf16 rounding first, so the quantiser sees the value the
// un-fused pair gives it; a 32-column block = 8 consecutive lanes, amax on DPP) and stages the scale bytes in LDS for two coalesced dword stores per lane.
#include "sf_gemm_ln_common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/synchformer_hip.h"

#define XL_BK 64                                        // fp8 elements = bytes per row per k-step
#define XL_SC_OFF RP_SLAB_OFF                           // scale slots 0-2 (the slabs of waves 0-2)
#define XL_SC_SLOT 4096                                 // [W rows 0-767 | A rows 0-127 | 128 dummy] dwords
#define XL_SC_A 3072
#define XL_SC_DUMMY 3584
#define XL_YS_OFF (RP_SLAB_OFF + 4 * RL_SLAB_BYTES)     // the tile's output scale bytes [6 planes][128 rows][4] (the slab of wave 4), written in pass 4
#ifndef SF_XL_STORECNT
#define SF_XL_STORECNT 1                                // first k-step of a tile: the previous epilogue's 50 Y / scale stores may stay in flight (vmcnt 8 + 50)
#endif
#define XL_STORES 50

typedef __attribute__((ext_vector_type(8))) int xl_i32x8;
typedef __attribute__((ext_vector_type(4))) int xl_i32x4;

struct MxResLnArgs {
  const uint8_t* A; int64_t lda; const uint8_t* sA; int64_t ldsa;
  const uint8_t* W; int64_t ldw; const uint8_t* sW; int64_t ldsw;
  const float* bias;
  const float* R; int64_t ldr;
  float* X; int64_t ldx;
  const float* gamma; const float* beta;
  uint8_t* Y; int64_t ldy; uint8_t* sY; int64_t ldsy;
  int64_t M;
  int K;
  float eps;
  uint32_t tiles;
  uint32_t stagger;                                               // as in sf_gemm_ln.hip: the workgroups with one tile fewer start stagger x ~1.2 us late (SF_RL_STAGGER)
};

// 64 dwords (one per lane) -> 256 consecutive bytes of LDS
__device__ __forceinline__ void xl_dma_dword(uint32_t v0, const void* sbase, uint32_t l0) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "s"(sbase), "s"(l0) : "memory");
}

// The lane index, re-derived where it is needed (mbcnt over an opaque zero: not hoisted, not kept): the kernel runs at the 256-register limit, and a thread
// id kept live across the tile loop was spilled - its reload in the epilogue sat behind a compiler vmcnt(0), i.e. behind the 48 X stores in flight.
__device__ __forceinline__ int xl_lane() {
  int z = 0;
  asm volatile("" : "+v"(z));
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

__global__ __launch_bounds__(512, 2) void gemm_mx_res_ln768_kernel(MxResLnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                        // 2 x 4 waves, wave tile 64 x 192
  // ---- LDS-DMA addressing: lane i of a piece fills (row i >> 2, slot i & 3) with source chunk slot ^ ((row >> 2) & 3) ----
  const int prow = lane >> 2, pslot = lane & 3;
  const int pchunk = pslot ^ ((prow >> 2) & 3);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(rl_lds_addr(smem));
  const uint32_t lane4 = (uint32_t)lane * 4u;

  const int nk = p.K / XL_BK;                                      // even (K % 128 == 0)
  uint32_t t = blockIdx.x;
  if (t >= p.tiles) return;
  if ((p.tiles - 1u - blockIdx.x) / gridDim.x < (p.tiles - 1u) / gridDim.x) for (uint32_t i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(32);

  int64_t m0 = (int64_t)t * RL_BM;
  const void* sa; uint32_t voff_a0;
  const char* sa_sc;                                               // the tile's first A scale dword in plane 0
  auto set_tile = [&](int64_t mm) {                                // (lane quantities re-derived from an opaque thread id in 32-bit arithmetic: kept live across
    sa = reinterpret_cast<const char*>(p.A) + mm * p.lda;          // the main loop they were spilled, and the reload sat behind a vmcnt(0) on the X stores)
    sa_sc = reinterpret_cast<const char*>(p.sA) + mm * 4;
    const int sl = xl_lane();
    const int srow = sl >> 2, schunk = (sl & 3) ^ ((srow >> 2) & 3);
    const int64_t left = p.M - 1 - mm;                             // tail tile: rows beyond M re-read the last valid row (their outputs are dropped)
    const int last = left < 127 ? (int)left : 127;
    int r = wave * 16 + srow;
    if (r > last) r = last;
    voff_a0 = __umul24((uint32_t)r, (uint32_t)p.lda) + (uint32_t)schunk * 16u;
  };
  // k-loop rotation (see sf_gemm_ln.hip), by whole pairs of k-steps: a pair shares its scale dwords
  const int krot = (int)((blockIdx.x >> 3) % (uint32_t)nk) & ~1;
  auto kmap = [&](int kt) { int k = kt + krot; return k >= nk ? k - nk : k; };
  // lane offsets of this wave's two pieces of a W third (buffer row r = wave * 32 + j * 16 + prow <-> W row (r >> 6) * 192 + c * 64 + (r & 63))
  uint32_t voff_w[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) voff_w[j] = (uint32_t)(((wave >> 1) * 192 + (wave & 1) * 32 + j * 16 + prow) * (int)p.ldw + pchunk * 16);
  const char* w0 = reinterpret_cast<const char*>(p.W);
  const int64_t third_b = (int64_t)64 * p.ldw;
  const char* wb1 = w0 + third_b; const char* wb2 = w0 + 2 * third_b;
  // running state of the load stream: wo1 = byte offset of rotated k-step kt+1 (W thirds 1, 2), wo2 = of k-step kt+2 (A, W third 0, scale piece)
  uint32_t wo1 = 0, wo2 = 0;
  int kq2 = 0;
  uint32_t wslot = 0;                                              // LDS byte offset of the scale slot of the pair being loaded
  // this wave's scale piece of the two halves of a pair: source offset inside the plane and destination inside the slot
  const bool h1_is_a = wave == 4 || wave == 5;
  const uint32_t sc_src1 = wave < 4 ? 2048u + wave * 256u : (wave < 6 ? (wave - 4) * 256u : (wave - 6) * 256u);
  const uint32_t sc_dst1 = wave < 4 ? 2048u + wave * 256u : (wave < 6 ? XL_SC_A + (wave - 4) * 256u : XL_SC_DUMMY + (wave - 6) * 256u);
  // scale piece that travels with the A | W0 group of the k-step whose rotated index is kq (parity known at compile time)
  auto issue_scale = [&](auto ODDc, int kq) {
    constexpr bool ODD = decltype(ODDc)::value;
    if (ODD) {                                                     // half 0 of the NEXT pair (the one after kq's): W rows wave * 64 .. + 63
      wslot = wslot == 2 * XL_SC_SLOT ? 0u : wslot + XL_SC_SLOT;
      const int pq = (kq + 1 == nk ? 0 : kq + 1) >> 1;
      xl_dma_dword(lane4, reinterpret_cast<const char*>(p.sW) + (int64_t)pq * p.ldsw + wave * 256, lds0 + XL_SC_OFF + wslot + wave * 256);
    } else {                                                       // half 1 of kq's own pair
      const int pq = kq >> 1;
      const char* src = h1_is_a ? sa_sc + (int64_t)pq * p.ldsa : reinterpret_cast<const char*>(p.sW) + (int64_t)pq * p.ldsw;
      xl_dma_dword(lane4, src + sc_src1, lds0 + XL_SC_OFF + wslot + sc_dst1);
    }
  };
  auto pp_issue_w = [&](int c, int slot) {
    rl_dma2(voff_w[0], voff_w[1], c == 0 ? w0 + wo2 : (c == 1 ? wb1 + wo1 : wb2 + wo1), lds0 + slot * RP_STRIDE + RP_W_OFF + c * RP_THIRD + wave * 2048);
  };
  auto pp_issue_a = [&](int slot) { rl_dma1(voff_a0, reinterpret_cast<const char*>(sa) + wo2, lds0 + slot * RP_STRIDE + wave * 1024); };
  auto pp_advance = [&]() {                                        // end of a k-step: kt+2 becomes kt+1, the next rotated k-step becomes kt+2
    wo1 = wo2;
    if (++kq2 == nk) { kq2 = 0; wo2 = 0; } else wo2 += XL_BK;
  };
  auto pp_prologue = [&]() {   // pair 0's scales | A | W0 of k-step 0, then W1, W2 of k-step 0, A | W0 of k-step 1 + half 0 of pair 1: 13 pieces per wave
    const int k0 = kmap(0), k1 = kmap(1);
    wslot = 0;
    xl_dma_dword(lane4, reinterpret_cast<const char*>(p.sW) + (int64_t)(k0 >> 1) * p.ldsw + wave * 256, lds0 + XL_SC_OFF + wave * 256);
    issue_scale(std::false_type{}, k0);
    rl_dma1(voff_a0, reinterpret_cast<const char*>(sa) + k0 * XL_BK, lds0 + wave * 1024);
    rl_dma2(voff_w[0], voff_w[1], w0 + k0 * XL_BK, lds0 + RP_W_OFF + wave * 2048);
    rl_dma2(voff_w[0], voff_w[1], wb1 + k0 * XL_BK, lds0 + RP_W_OFF + RP_THIRD + wave * 2048);
    rl_dma2(voff_w[0], voff_w[1], wb2 + k0 * XL_BK, lds0 + RP_W_OFF + 2 * RP_THIRD + wave * 2048);
    rl_dma1(voff_a0, reinterpret_cast<const char*>(sa) + k1 * XL_BK, lds0 + RP_STRIDE + wave * 1024);
    rl_dma2(voff_w[0], voff_w[1], w0 + k1 * XL_BK, lds0 + RP_STRIDE + RP_W_OFF + wave * 2048);
    issue_scale(std::true_type{}, k1);
    wo1 = (uint32_t)k1 * XL_BK;
    kq2 = nk > 2 ? kmap(2) : 0;
    wo2 = (uint32_t)kq2 * XL_BK;
  };
  set_tile(m0);
  pp_prologue();

  float* stat = reinterpret_cast<float*>(smem + RP_STAT_OFF);
  float* pbias = reinterpret_cast<float*>(smem + RP_BIAS_OFF);     // the epilogue reads bias / gamma / beta from LDS: a global load there would make hipcc
  float* pgamma = reinterpret_cast<float*>(smem + RP_GB_OFF);      // wait vmcnt(0), i.e. for every residual piece still in flight
  float* pbeta = pgamma + RL_N;
  for (int i = tid; i < RL_N; i += 512) {
    pbias[i] = p.bias ? p.bias[i] : 0.f;
    pgamma[i] = p.gamma[i];
    pbeta[i] = p.beta[i];
  }
  __syncthreads();                                                 // (drains the prologue too: first tile only)
  bool stores_behind = false;                                      // the previous epilogue's 50 stores were issued after this tile's first loads
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.X, (short)0, (int)(uint32_t)(p.M * p.ldx * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.Y, (short)0, (int)(uint32_t)(p.M * p.ldy), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.sY, (short)0, (int)(uint32_t)(6 * p.ldsy), 0x00020000);

  for (;;) {
    f32x16 acc[2][6];
    // zeroed by explicit moves: a `= 0.f` lets hipcc keep ONE zero vector live across the tile loop - in scratch, at this register count - and makes the
    // accumulators' first use wait for its reload with vmcnt(0); opaque zeros also keep it from peeling the first k-step pair (see sf_gemm_mx.hip)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { float z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); acc[i][j][r] = z; }

    {
      int fa[2], fw[2];                                            // fragment addresses in the CURRENT stage (bit 16 flipped once per k-step)
      int fsa, fsw;                                                // this half-wave's scale BYTE (block hi of step 0) in the current slot: A rows wm*64 + l31 (+ i*32), W rows wn*192 + l31 (+ c*64 + jj*32)
      {
        const int pl = xl_lane();
        const int pl31 = pl & 31, phi = pl >> 5;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int sw = ((ks * 2 + phi) ^ ((pl31 >> 2) & 3)) << 4;
          fa[ks] = (wm * 64 + pl31) * 64 + sw;                     // + i * 2048
          fw[ks] = RP_W_OFF + (wn * 64 + pl31) * 64 + sw;          // + c * RP_THIRD + jj * 2048
        }
        fsa = XL_SC_OFF + XL_SC_A + (wm * 64 + pl31) * 4 + phi;
        fsw = XL_SC_OFF + (wn * 192 + pl31) * 4 + phi;
      }
      int rslot = 0;                                               // scale slot of the current pair (scalar copy of what fsa / fsw carry)
      xl_i32x8 a[2], w[2];
      int sav[2], sbv[2];
      auto frag = [&](const char* base, const int (&off)[2]) -> xl_i32x8 {
        const xl_i32x4 lo = *reinterpret_cast<const xl_i32x4*>(base + off[0]);
        const xl_i32x4 hi4 = *reinterpret_cast<const xl_i32x4*>(base + off[1]);
        return xl_i32x8{lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
      };
      auto mma = [&](auto Cc) {
        constexpr int C = decltype(Cc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            acc[i][2 * C + jj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], w[jj], acc[i][2 * C + jj], 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, sav[i], 0, sbv[jj]);
        asm volatile("" : "+v"(acc[0][2 * C]), "+v"(acc[0][2 * C + 1]), "+v"(acc[1][2 * C]), "+v"(acc[1][2 * C + 1]));   // pins the MFMAs in the matrix segment
        __builtin_amdgcn_s_setprio(0);
      };
      // fragments + this half-wave's scale bytes of W third C in k-step parity S (byte 2 S + hi of the pair's dword)
      auto read_w = [&](auto Cc, auto Sc) {
        constexpr int C = decltype(Cc)::value, S = decltype(Sc)::value;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          w[jj] = frag(smem + C * RP_THIRD + jj * 2048, fw);
          sbv[jj] = (int)*reinterpret_cast<const uint8_t*>(smem + fsw + (C * 64 + jj * 32) * 4 + S * 2);
        }
      };
      // one k-step in stage S (= its parity inside the pair); `more1` = k-step kt+1 exists, `more2` = k-step kt+2 exists, `first` = first k-step of the tile
      auto kstep = [&](auto Sc, bool more1, bool more2, bool first) {
        constexpr int S = decltype(Sc)::value;
        const bool behind = SF_XL_STORECNT && first && stores_behind;
        // ---- phase 0: A x W third 0 ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = frag(smem + i * 2048, fa);
          sav[i] = (int)*reinterpret_cast<const uint8_t*>(smem + fsa + i * 128 + S * 2);
        }
        read_w(std::integral_constant<int, 0>{}, Sc);
        __builtin_amdgcn_sched_barrier(0);
        if (more1) pp_issue_w(1, S ^ 1);
        if (!more1) rl_wait_vmcnt<2>();                            // W third 1 of this k-step has landed; the youngest stage stays in flight
        else if (behind) rl_wait_vmcnt<8 + XL_STORES>();
        else rl_wait_vmcnt<8>();
        rl_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        rl_barrier();
        // ---- phase 1: third 1 ----
        read_w(std::integral_constant<int, 1>{}, Sc);
        __builtin_amdgcn_sched_barrier(0);
        if (more1) pp_issue_w(2, S ^ 1);
        if (!more1) rl_wait_vmcnt<0>();                            // W third 2 has landed
        else if (behind) rl_wait_vmcnt<8 + XL_STORES>();
        else rl_wait_vmcnt<8>();
        rl_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        rl_barrier();
        // ---- phase 2: third 2; the fragment addresses move on to the other stage ----
        read_w(std::integral_constant<int, 2>{}, Sc);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(fa[ks]));
          asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(fw[ks]));
        }
        if (S == 1) {                                              // the next k-step opens the next pair: its slot
          const int d = rslot == 2 * XL_SC_SLOT ? -2 * XL_SC_SLOT : XL_SC_SLOT;
          rslot += d; fsa += d; fsw += d;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more2) { pp_issue_a(S); pp_issue_w(0, S); issue_scale(std::integral_constant<bool, S == 1>{}, kq2); }
        pp_advance();
        if (more1) {                                               // A | W third 0 (| scale piece) of the next k-step have landed
          if (!more2) rl_wait_vmcnt<4>();
          else if (behind) rl_wait_vmcnt<8 + XL_STORES>();
          else rl_wait_vmcnt<8>();
        }
        rl_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        rl_barrier();
      };
      // pair 0's scales | A | W third 0 of k-step 0 have landed (this wave's pieces), then everybody's
      if (SF_XL_STORECNT && stores_behind) rl_wait_vmcnt<8 + XL_STORES>();
      else rl_wait_vmcnt<8>();
      rl_barrier();
      if (wm == 1) rl_barrier();                                   // the wm = 1 waves run one barrier behind
      int kt_first = 0;
      asm volatile("" : "+s"(kt_first));
      for (int kt = 0; kt < nk; kt += 2) {
        const bool more = kt + 2 < nk;
        kstep(std::integral_constant<int, 0>{}, true, more, kt == kt_first);
        kstep(std::integral_constant<int, 1>{}, more, more, false);
      }
      if (wm == 0) rl_barrier();                                   // re-align; also: every wave is done with both stages and the scale slots
    }
    const int64_t em0 = m0;
    const uint32_t tnext = t + gridDim.x;
    const bool more = tnext < p.tiles;

    // ---- epilogue, passes 1-3: as in gemm_res_ln768_kernel (sf_gemm_ln.hip has the commentary) ----------------------------------------------
    const int elane = xl_lane(), l31 = elane & 31, hi = elane >> 5, lr = elane >> 4, ecol = (elane & 15) * 4;
    const int gcol0 = wn * 192 + ecol;                             // + c * 64
    float* slab = reinterpret_cast<float*>(smem + RP_SLAB_OFF + wave * RL_SLAB_BYTES);
    const int ring_off = (wave >> 2) * RP_STRIDE + (wave & 3) * RL_RING_WAVE;
    const char* ring = smem + ring_off + elane * 16;
    const uint32_t ring_lds = lds0 + ring_off;
    int64_t rrow0 = em0 + wm * 64;                                  // tail tile: rows beyond M re-read row M - 1 (their outputs are dropped)
    if (rrow0 > p.M - 1) rrow0 = p.M - 1;
    const int64_t rleft = p.M - 1 - rrow0;
    const int rlast = rleft < 63 ? (int)rleft : 63;
    const void* rbase = reinterpret_cast<const char*>(p.R) + rrow0 * p.ldr * 4;
    auto issue_res = [&](int s) {
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
      if (s <= 9) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // step s has landed: only the (<= 2) younger steps may be outstanding
      else if (s == 10) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        float4 v = *reinterpret_cast<const float4*>(slab + (ps * 4 + lr) * 64 + ecol);
        const float4 rv = *reinterpret_cast<const float4*>(ring + (s % 3) * 4096 + ps * 1024);
        v.x = (v.x + bias4.x) + rv.x; v.y = (v.y + bias4.y) + rv.y; v.z = (v.z + bias4.z) + rv.z; v.w = (v.w + bias4.w) + rv.w;
        xr[g][c][ps] = v;
      }
      if (s + 3 < 12) {                                             // this step's ring slot has been read back: refill it with step s + 3
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_res(s + 3);
      }
    }
    const int64_t row0 = em0 + wm * 64 + lr;
    const uint32_t xoff0 = (uint32_t)(row0 * p.ldx + gcol0) * 4u, xstep = (uint32_t)(4 * p.ldx) * 4u;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float4 v = xr[g][c][ps];
          rl_u32x4 o;
          o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
          __builtin_amdgcn_raw_buffer_store_b128(o, rx, xoff0 + (uint32_t)(g * 4 + ps) * xstep + (uint32_t)c * 256u, 0, 2);
        }
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
    // every wave is past its residual ring and its slab: the operand slots and scale slots 0 / 1 take the next tile's first loads
    if (more) {
      m0 = (int64_t)tnext * RL_BM;
      set_tile(m0);
      pp_prologue();
      stores_behind = true;
    }
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
    // ---- pass 4: normalise, scale, shift, bf16-round, quantise to e4m3 with the block's E8M0 scale, store ------------------------------------
    const uint32_t yoff0 = (uint32_t)(row0 * p.ldy + gcol0), ystep = (uint32_t)(4 * p.ldy);
    uint8_t* ys = reinterpret_cast<uint8_t*>(smem + XL_YS_OFF);
    const int blk0 = wn * 6 + ((elane >> 3) & 1);                   // + c * 2: the 32-column block of this lane's four columns
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
        const int trow = wm * 64 + g * 16 + ps * 4 + lr;            // row inside the tile
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float4 v = xr[g][c][ps];
          const uint32_t p01 = pack_bf2((v.x - mu) * rs * gm[c].x + bt[c].x, (v.y - mu) * rs * gm[c].y + bt[c].y);
          const uint32_t p23 = pack_bf2((v.z - mu) * rs * gm[c].z + bt[c].z, (v.w - mu) * rs * gm[c].w + bt[c].w);
          const float f0 = __uint_as_float(p01 << 16), f1 = __uint_as_float(p01 & 0xffff0000u), f2 = __uint_as_float(p23 << 16), f3 = __uint_as_float(p23 & 0xffff0000u);
          float amax = fmaxf(fmaxf(fabsf(f0), fabsf(f1)), fmaxf(fabsf(f2), fabsf(f3)));
          // max over the block's eight lanes on DPP: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
          amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0xB1, 0xF, 0xF, true)));
          amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0x4E, 0xF, 0xF, true)));
          amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0x141, 0xF, 0xF, true)));
          int be = sf_mx_be(amax);
          be = be < 1 ? 1 : (be > 254 ? 254 : be);
          const float inv = __uint_as_float((uint32_t)(254 - be) << 23);
          int wq = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f0 * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(f1 * inv, 448.f, -448.f), 0, false);
          wq = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f2 * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(f3 * inv, 448.f, -448.f), wq, true);
          __builtin_amdgcn_raw_buffer_store_b32((uint32_t)wq, ry, yoff0 + (uint32_t)(g * 4 + ps) * ystep + (uint32_t)c * 64u, 0, 2);
          const int blk = blk0 + c * 2;
          if ((elane & 7) == 0) ys[(blk >> 2) * 512 + trow * 4 + (blk & 3)] = (uint8_t)be;
        }
      }
    // the tile's 768 scale dwords (6 planes x 128 rows), coalesced: two stores per lane (the second only in waves 0-3); rows >= M get an offset the range check drops
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int d = wave * 64 + elane + k * 512;
      const int pl = d >> 7, rw = d & 127;
      const uint32_t word = *reinterpret_cast<const uint32_t*>(ys + (d < 768 ? d : 0) * 4);
      const uint32_t off = (d < 768 && em0 + rw < p.M) ? (uint32_t)((int64_t)pl * p.ldsy + (em0 + rw) * 4) : 0xffffffffu;
      __builtin_amdgcn_raw_buffer_store_b32(word, rsy, off, 0, 0);
    }
    if (!more) break;
    t = tnext;
  }
}

extern "C" int sf_gemm_mx_res_ln768(const uint8_t* A, int64_t lda, const uint8_t* sA, int64_t ldsa, const uint8_t* W, int64_t ldw, const uint8_t* sW,
                                    int64_t ldsw, const float* bias, const float* R, int64_t ldr, float* X, int64_t ldx, const float* gamma,
                                    const float* beta, float eps, uint8_t* Y, int64_t ldy, uint8_t* sY, int64_t ldsy, int64_t M, int64_t K, void* stream) {
  SF_CHECK_ARG(A && sA && W && sW && R && X && gamma && beta && Y && sY, "sf_gemm_mx_res_ln768: null pointer");
  SF_CHECK_ARG(K >= 128 && (K % 128) == 0 && K < (1 << 20), "sf_gemm_mx_res_ln768: K=%lld must be a positive multiple of 128", (long long)K);
  SF_CHECK_ARG((lda % 16) == 0 && (ldw % 16) == 0 && lda >= K && ldw >= K, "sf_gemm_mx_res_ln768: lda/ldw must be >= K and multiples of 16 bytes");
  SF_CHECK_ARG((ldr % 4) == 0 && (ldx % 4) == 0 && (ldy % 4) == 0 && ldr >= RL_N && ldx >= RL_N && ldy >= RL_N,
               "sf_gemm_mx_res_ln768: ldr/ldx/ldy must be >= 768 and multiples of 4 elements");
  SF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)R % 16) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 4) == 0 &&
                   ((uintptr_t)sA % 4) == 0 && ((uintptr_t)sW % 4) == 0 && ((uintptr_t)sY % 4) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                   ((uintptr_t)beta % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0),
               "sf_gemm_mx_res_ln768: operands must be 16-byte aligned (scale planes 4-byte)");
  if (M <= 0) return 0;
  const int64_t m_pad = ((M + RL_BM - 1) / RL_BM) * RL_BM;
  // scale planes: one dword per row per 128 k; the A planes are read for whole 128-row tiles, the W planes for all 768 rows
  SF_CHECK_ARG((ldsa % 4) == 0 && (ldsw % 4) == 0 && (ldsy % 4) == 0 && ldsa >= m_pad * 4 && ldsw >= RL_N * 4 && ldsy >= M * 4,
               "sf_gemm_mx_res_ln768: scale planes must hold >= ceil(M / 128) * 128 (A), 768 (W), M (Y) dwords");
  SF_CHECK_ARG(m_pad * ldr * 4 < ((int64_t)1 << 32) && m_pad * ldx * 4 < ((int64_t)1 << 32) && m_pad * ldy < ((int64_t)1 << 32) && 6 * ldsy < ((int64_t)1 << 32),
               "sf_gemm_mx_res_ln768: R / X / Y / sY must stay below 4 GiB");
  SF_CHECK_ARG(128 * lda + K < ((int64_t)1 << 31) && RL_N * ldw + K < ((int64_t)1 << 31), "sf_gemm_mx_res_ln768: row strides too large");
  const int n_cu = sf_cu_count("sf_gemm_mx_res_ln768");
  if (n_cu <= 0) return -1;
  MxResLnArgs a;
  a.A = A; a.lda = lda; a.sA = sA; a.ldsa = ldsa; a.W = W; a.ldw = ldw; a.sW = sW; a.ldsw = ldsw; a.bias = bias; a.R = R; a.ldr = ldr; a.X = X; a.ldx = ldx;
  a.gamma = gamma; a.beta = beta; a.Y = Y; a.ldy = ldy; a.sY = sY; a.ldsy = ldsy; a.M = M; a.K = (int)K; a.eps = eps;
  const int64_t tiles = m_pad / RL_BM;
  SF_CHECK_ARG(tiles < ((int64_t)1 << 31), "sf_gemm_mx_res_ln768: too many tiles");
  a.tiles = (uint32_t)tiles;
  { static int st = -1; if (st < 0) { const char* e = getenv("SF_RL_STAGGER"); st = e ? atoi(e) : 12; if (st < 0) st = 0; } a.stagger = K >= 768 ? (uint32_t)st : 0u; }   // (a tile of a shorter k-loop is shorter than the delay)
  const int64_t blocks = tiles < n_cu ? tiles : n_cu;              // one persistent workgroup per CU
  if (int rc = sf_prepare_kernel((const void*)gemm_mx_res_ln768_kernel, RP_LDS, "sf_gemm_mx_res_ln768")) return rc;
  hipLaunchKernelGGL(gemm_mx_res_ln768_kernel, dim3((unsigned)blocks), dim3(512), RP_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
