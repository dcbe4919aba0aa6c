// Motionformer SPACE attention fused into its qkv projection (gfx950): one launch computes, for every patch token, the spatial q | k | v of
// DividedSpaceTimeBlock (vit_helper.py:370 -> DividedAttention.forward, vit_helper.py:97-150 with the '(b f) n d' regrouping of :341-342) and the
// per-frame attention over [CLS key; the frame's 196 patches] in the GEMM's epilogue.  Un-fused (sf_gemm_bf16 -> sf_attention_cls_partial) the 2304-wide
// projection goes to HBM (1.62 GB at 224 segments) and is read back; fused, only the 768-wide attention output is written.
//
//   * Work item = (frame, HEAD PAIR).  A frame's group is 196 tokens = 6 x 32 + 4: the GEMM tile is the frame's first 192 token rows (contiguous in X) x the
//     q | k | v of two heads (384 features) = 6 x 12 = 72 blocks of 32 x 32, nine per wave (2 x 4 waves of 96 rows x 96 features) - no padded rows, the operand
//     bytes per MAC of the 256 x 256 tile.  The four left-over tokens of every frame and the CLS row are projected up front by a small GEMM of the caller
//     (`side`: 33 rows per sequence); they join as queries 192-195 / keys 193-196 / key 0 in the epilogue.
//   * Main loop: 64-deep k-tiles in two 72-KiB stages (A 24 KiB | W0 | W1 | W2 of 16 KiB, W part j = feature block j of each of the four wave columns), THREE phases per
//     k-tile - one feature block each, 12 MFMAs of 32x32x16 per wave and phase with the wave's 12 A fragments held in registers - accumulators transposed
//     (C^T = W X^T: a lane ends with one token's features).  LDS-DMA pieces (9 per wave and k-tile) run about one stage ahead behind COUNTED waits, the wm = 1
//     waves one barrier behind the wm = 0 waves - the schedule of sf_qkv_time.hip / sf_gemm_pp.hip:
//         phase 0 of k-tile kt: issue W1, A1, A2 of kt+1 (no wait) | phase 1: issue W2 of kt+1, vmcnt(9): W2 of kt landed | phase 2: issue W0, A0 of kt+2,
//         vmcnt(5): A, W0, W1 of kt+1 landed;   every part is refilled two phases after its last fragment read, data is read one phase after its wait.
//   * Epilogue: accumulators (+ bias) -> bf16 -> LDS as K | V | Q of both heads in the row-major, XOR-swizzled layout of attn_mfma_kernel (sf_attention.hip); the side rows
//     arrive by LDS-DMA during the main loop; then the arithmetic of that kernel (S^T = K Q^T on v_mfma_f32_16x16x32_bf16, in-lane base-2 softmax, P V with
//     ds_read_b64_tr_b16 V fragments) on PAIRS of 16-query tiles - every K / V fragment read from LDS feeds two MFMAs: 14 pair units (2 heads x 7) over the 8 waves -, the
//     CLS query's softmax partial of the frame in the free 197th query slot ([seq][head][8][66] records for sf_attention_cls_combine).  The attention arrays take 156
//     of the 160 KiB: the next tile's operands are NOT prefetched under it.
//   * Measured (M = 351,456, profiles/r04_qkv_space.md): 1330 us against 1050 (sf_gemm_bf16 qkv) + 483 (attention) un-fused; the main loop + accumulator hand-over
//     alone run at 1.31 PFLOP/s (947 us), the attention epilogue costs ~9 us per work item (softmax VALU + LDS fragment reads + MFMAs of ONE workgroup per CU).
//   * persistent, one workgroup per CU; every XCD owns a contiguous range of frames and sweeps it once per chunk of head pairs (pair fastest inside a chunk).
#include "sf_common.h"
#include <type_traits>
#include <stdlib.h>
#include "../../include/synchformer_hip.h"

#define QS_TOK 196                     // tokens of a space group (one frame)
#define QS_ROWS 192                    // ... of which the GEMM tile computes the first 192
#define QS_D 768
#define QS_A_BYTES (QS_ROWS * 128)     // 24 KiB: 192 rows x 64 k (bf16)
#define QS_W_PART (128 * 128)          // 16 KiB: 4 wave columns x 32 features x 64 k
#define QS_STAGE (QS_A_BYTES + 3 * QS_W_PART)   // 72 KiB
#define QS_SIDE_OFF (2 * QS_STAGE)     // 144 KiB: landing area of the side rows (240 x 16 B) during the main loop
#define QS_ARR (208 * 128)             // one K / V / Q array of the attention: 208 rows of 128 B
#define QS_BIAS_OFF (6 * QS_ARR)       // 156 KiB: the tile's 384 bias floats (behind the attention arrays: lives through main loop AND epilogue)
#define QS_MASK_OFF (QS_BIAS_OFF + 1536) // MASK variant: one float per key of the frame's group (0 = kept, -inf = masked), 208 floats; key 0 = the CLS key
#define QS_LDS (160 * 1024)
#ifndef QS_ABL
#define QS_ABL 0                       // measurement builds: 1 no attention (epilogue part 2 skipped), 2 no MFMAs in the main loop, 4 no softmax arithmetic, 8 no S phase, 16 one of the seven P V steps
#endif

struct QsArgs {
  const bf16_t* X; int64_t ldx;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  const bf16_t* side; int64_t lds_;    // (n_seq * 33, 2304) bf16: row seq * 33 = the CLS row's q | k | v, row seq * 33 + 1 + 4 f + i = token 192 + i of frame f
  bf16_t* out; int64_t ldo;
  float* cls_part;                     // [n_seq][12][8][66]
  int64_t seq_rows;
  uint32_t n_frames;                   // n_seq * 8
  uint32_t pair_chunk;                 // head pairs per sweep over an XCD's frames (divides 6)
  float scale;
  // MX (template parameter): X / W are e4m3 BYTES (ldx / ldw in bytes) with stage-major E8M0 scale planes (one dword per row per 128 k, ldsx / ldsw bytes between planes)
  const uint8_t* sX = nullptr; int64_t ldsx = 0;
  const uint8_t* sW = nullptr; int64_t ldsw = 0;
  // MX, optional: the attention output as MXFP8 (e4m3 bytes, row stride ldq, + E8M0 bytes in the scale planes [6][rows][4], splane bytes apart) instead of bf16 `out`
  uint8_t* out_q = nullptr; int64_t ldq = 0; uint8_t* out_s = nullptr; int64_t splane = 0;
  // MASK (template parameter): token flags, one byte per row of X; a row with flag 0 is a masked KEY (score -inf) for every query of its frame and for the CLS query
  const uint8_t* key_keep = nullptr;
};
#define QS_SC_OFF (148 * 1024)         // MX: the k-tile's scale dwords, two parities x (192 token rows | 3 x 128 part rows of W) = 2 x 2304 B
#define QS_SC_BYTES 2304
#ifndef QS_MXOUT
#define QS_MXOUT 1
#endif
typedef __attribute__((ext_vector_type(8))) int qs_i32x8;

typedef short qs_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) qs_s4 qs_lds_s4;

__device__ __forceinline__ void qs_dma1(uint32_t voff, const void* sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}
__device__ __forceinline__ void qs_dma_dword_addr(const void* gaddr, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gaddr), "s"(lds) : "memory");
}
template <int N>
__device__ __forceinline__ void qs_wait_vmcnt() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void qs_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint32_t qs_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ int qs_arr_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }   // = k_lds_off<64> of sf_attention.hip

template <int V> using qs_ic = std::integral_constant<int, V>;

template <bool MX, bool MASK = false>
__device__ __forceinline__ void qkv_space_attn_body(const QsArgs& p) {
  constexpr int ESZ = MX ? 1 : 2;                                 // bytes per operand element
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                         // 2 x 4 waves, wave tile 96 token rows x 96 features
  const int hi = lane >> 5;

  // persistent schedule: block b sits on XCD b % 8; every XCD owns a contiguous range of frames and walks (frame, head pair) in chunks of `hc` pairs
  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t f8 = (p.n_frames + 7u) >> 3;
  const uint32_t fr0 = min(xcd * f8, p.n_frames), fr1 = min(fr0 + f8, p.n_frames);
  const uint32_t hc = p.pair_chunk, chunk_tiles = (fr1 - fr0) * hc, t_end = (fr1 - fr0) * 6u;
  uint32_t t = li;
  if (t >= t_end) return;

  // ---- tile-invariant lane offsets of the LDS-DMA pieces -----------------------------------------------------------------------------
  // A piece pc of this wave = tile rows (3 wave + pc) * 8 .. + 7; lane (r = lane >> 3, chunk slot = lane & 7) -> LDS row-linear, source chunk XOR-swizzled
  uint32_t voff_a = 0, voff_w[3] = {0, 0, 0}, sc_voff = 0;         // lane offsets of the LDS-DMA pieces: re-derived at the top of every tile (not kept live across the attention)
  const uint32_t a8 = (uint32_t)(8 * p.ldx * ESZ), w8 = (uint32_t)(8 * p.ldw * ESZ);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(qs_lds_addr(smem));
  const uint32_t lds_a_w = __builtin_amdgcn_readfirstlane(lds0 + wave * 3072);
  const uint32_t lds_w_w = __builtin_amdgcn_readfirstlane(lds0 + QS_A_BYTES + wave * 2048);

  const char* xbase; const char* wbase;
  uint32_t fr; int hp; int64_t xrow0;
  auto set_tile = [&](uint32_t tt) {
    const uint32_t c = tt / chunk_tiles, r = tt - c * chunk_tiles;
    fr = fr0 + r / hc; hp = (int)(c * hc + r % hc);
    const int64_t seq = fr >> 3; const int f = (int)(fr & 7u);
    xrow0 = seq * p.seq_rows + 1 + (int64_t)f * QS_TOK;
    xbase = reinterpret_cast<const char*>(p.X) + xrow0 * p.ldx * ESZ;
    wbase = reinterpret_cast<const char*>(p.W) + (int64_t)hp * 128 * p.ldw * ESZ;
  };
  // MX: the scale dwords of a k-tile - 192 token rows (768 contiguous bytes of plane kt of sX) and the tile's 384 W rows (12 runs of 32 rows of plane kt of sW, in
  // part order: area index j * 128 + 32 wn' + rr) - as 16-byte LDS-DMA lanes: 48 + 96 lanes = three pieces.  EVERY wave issues exactly one of them (wave % 3; the
  // copies land the same bytes on the same addresses), so that all waves count the same number of vector-memory operations per k-tile
  const int sc_seg = wave % 3;
  const int sc_lanes = sc_seg == 0 ? 48 : (sc_seg == 1 ? 64 : 32);
  // tile-invariant lane offsets (row strides are multiples of 128 bytes - launcher check -, so the low 7 bits of an offset are its chunk slot: the piece 8 rows
  // further down is (offset ^ 64) + 8 rows, computed at issue time instead of held in a register per piece).
  // A piece pc of this wave = tile rows (3 wave + pc) * 8 .. + 7; lane (r = lane >> 3, chunk slot = lane & 7) -> LDS row-linear, source chunk XOR-swizzled.
  // W part j, piece pc of this wave = part rows (2 wave + pc) * 8 .. + 7; part row pr = 32 wn' + rr <-> tile column wn' * 96 + 32 j + rr <-> row
  // which * 768 + hd * 64 + feat of W (the head pair's offset rides in the SGPR base)
  auto derive_offsets = [&]() {
    int dtid = threadIdx.x;
    asm volatile("" : "+v"(dtid));
    const int dl = dtid & 63;
    {
      const int r = wave * 3 * 8 + (dl >> 3);
      voff_a = (uint32_t)((int64_t)r * p.ldx * ESZ + ((((dl & 7) ^ ((r >> 1) & 7))) << 4));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pr = wave * 2 * 8 + (dl >> 3);
      const int c = (pr >> 5) * 96 + j * 32 + (pr & 31);
      const int hd = c / 192, within = c - hd * 192;
      const int grow = (within >> 6) * QS_D + hd * 64 + (within & 63);
      voff_w[j] = (uint32_t)((int64_t)grow * p.ldw * ESZ + ((((dl & 7) ^ ((pr >> 1) & 7))) << 4));
    }
    if (MX) {
      if (sc_seg == 0) sc_voff = (uint32_t)dl * 16u;
      else {
        const int L = (sc_seg == 1 ? 0 : 64) + dl, run = L >> 3, j = run >> 2, wnp = run & 3;
        const int c = wnp * 96 + j * 32 + (L & 7) * 4;
        const int hd = c / 192, within = c - hd * 192;
        sc_voff = (uint32_t)(((within >> 6) * QS_D + hd * 64 + (within & 63)) * 4);
      }
    }
  };
  const uint32_t sc_lds = __builtin_amdgcn_readfirstlane(lds0 + QS_SC_OFF + (sc_seg == 0 ? 0 : (sc_seg == 1 ? 768 : 768 + 1024)));
  auto issue_sc = [&](int kt) {
    const char* base = sc_seg == 0 ? reinterpret_cast<const char*>(p.sX) + (int64_t)kt * p.ldsx + xrow0 * 4
                                   : reinterpret_cast<const char*>(p.sW) + (int64_t)kt * p.ldsw + (int64_t)hp * 512;
    if (lane < sc_lanes) qs_dma1(sc_voff, base, sc_lds + (kt & 1) * QS_SC_BYTES);
  };
  auto issue_a = [&](int pc, int S, int kt) {                       // piece pc: 8 pc rows further down; the chunk swizzle flips bit 2 with every 8 rows
    qs_dma1(((pc & 1) ? (voff_a ^ 64u) : voff_a) + (uint32_t)pc * a8, xbase + kt * 128, lds_a_w + S * QS_STAGE + pc * 1024);
  };
  auto issue_w = [&](int j, int S, int kt) {
    qs_dma1(voff_w[j], wbase + kt * 128, lds_w_w + S * QS_STAGE + j * QS_W_PART);
    qs_dma1((voff_w[j] ^ 64u) + w8, wbase + kt * 128, lds_w_w + S * QS_STAGE + j * QS_W_PART + 1024);
  };

  constexpr int nk = MX ? QS_D / 128 : QS_D / 64;                   // 12 k-tiles of 64 bf16 / 6 of 128 fp8: 128 bytes per row either way
  const float sc2 = p.scale * 1.44269504088896f;                   // softmax in base 2
  uint32_t tcount = 0;

  for (;;) {
    set_tile(t);
    derive_offsets();
    const int64_t seq = fr >> 3; const int f = (int)(fr & 7u);
    // ---- prologue: bias, side rows, k-tile 0 and W0 | A0 of k-tile 1 (the previous tile's attention is over: barrier at the bottom of the loop) ----------
    if (wave < 6) {                                                 // 6 x 64 bias floats: tile columns 64 wave .. + 63 = (head hd = wave / 3, q | k | v = wave % 3)
      if (p.bias) qs_dma_dword_addr(p.bias + (wave % 3) * QS_D + (hp * 2 + wave / 3) * 64 + lane, lds0 + QS_BIAS_OFF + wave * 256);
    }
    if (wave < 4) {                                                 // side rows: chunk x = ((hd * 5 + srow) * 3 + which) * 8 + ch, 240 chunks of 16 B
      const int x = wave * 64 + lane;
      if (x < 240) {
        const int ch = x & 7, w3 = (x >> 3) % 3, hs = (x >> 3) / 3, srow = hs % 5, hd = hs / 5;
        const uint32_t voff = (uint32_t)(((srow == 0 ? 0 : 1 + f * 4 + (srow - 1)) * p.lds_ + w3 * QS_D + hd * 64 + ch * 8) * 2);
        qs_dma1(voff, reinterpret_cast<const char*>(p.side + seq * 33 * p.lds_ + hp * 128), lds0 + QS_SIDE_OFF + wave * 1024);
      }
    }
    if (MX) issue_sc(0);
    issue_w(0, 0, 0); issue_a(0, 0, 0);
    issue_w(1, 0, 0); issue_a(1, 0, 0); issue_a(2, 0, 0);
    issue_w(2, 0, 0);
    issue_w(0, 1, 1); issue_a(0, 1, 1);
    if (!(QS_ABL & 32)) qs_wait_vmcnt<5>();                         // bias, side rows, A | W0 | W1 of k-tile 0 (this wave's pieces) have landed   (ABL 32: WRONG results - what does this wait cost?)
    qs_barrier();

    // accumulators start at the bias: block (j, i) = features 96 wn + 32 j + 8 g + 4 hi + r of the tile, tokens 96 wm + 32 i + l31
    f32x16 acc[3][3];
    {
      const float* bs = reinterpret_cast<const float*>(smem + QS_BIAS_OFF);
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
      int fo[4], fs_x = 0, fs_w = 0, shi = 0;                       // (MX: addresses of this lane's scale dwords - its token row of block 0, row l31 of its W part 0 - and 8 * (lane >> 5))
      {
        int ptid = threadIdx.x;
        asm volatile("" : "+v"(ptid));
        const int pl31 = ptid & 31, phi = (ptid & 63) >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fo[kk] = pl31 * 128 + (((kk * 2 + phi) ^ ((pl31 >> 1) & 7)) << 4);
        if (MX) { fs_x = QS_SC_OFF + (wm * 96 + pl31) * 4; fs_w = QS_SC_OFF + 768 + (wn * 32 + pl31) * 4; shi = phi * 8; }
      }
      const int a_base = wm * 96 * 128, w_base = QS_A_BYTES + wn * 32 * 128;
      // fragments as 32-byte pairs (kk = 2 k2, 2 k2 + 1): the two 16-byte reads a lane supplies to ONE 64-deep scaled MFMA sit in eight consecutive registers, no copies
      union QsFrag { bf16x8 h[2]; qs_i32x8 v; };
      QsFrag xf[3][2], wf[2];
      uint32_t sx[3] = {0u, 0u, 0u}, sw = 0u;                        // MX: this lane's scale dwords of the k-tile (token row of block i; W row of the current part), >> shi
      auto read_w = [&](const char* st, int j, int par) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wf[kk >> 1].h[kk & 1] = *reinterpret_cast<const bf16x8*>(st + w_base + j * QS_W_PART + fo[kk]);
        if (MX) sw = *reinterpret_cast<const uint32_t*>(smem + fs_w + j * 512 + par * QS_SC_BYTES) >> shi;
      };
      auto mma = [&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MX) {
          // a 128-byte LDS row = 128 fp8 k: fragments 2 k2 and 2 k2 + 1 are the 2 x 16 bytes a lane supplies to ONE 64-deep scaled MFMA (as qkv_time_attn_kernel<true, true>)
#pragma unroll
          // scale operands: the k-tile's dword of the row, shifted by 8 * (lane >> 5) so that BYTE 2 k2 is this half-wave's 32-k block of MFMA k2 - selected by the
          // instruction's op_sel (byte index of the scale register), no extraction arithmetic inside the matrix segment
          for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              if (k2 == 0) acc[J][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[0].v, xf[i][0].v, acc[J][i], 0 /* e4m3 */, 0 /* e4m3 */, 0, (int)sw, 0, (int)sx[i]);
              else acc[J][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[1].v, xf[i][1].v, acc[J][i], 0, 0, 2, (int)sw, 2, (int)sx[i]);
            }
          }
        } else
        if (!(QS_ABL & 2)) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[J][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk >> 1].h[kk & 1], xf[i][kk >> 1].h[kk & 1], acc[J][i], 0, 0, 0);
        } else asm volatile("" :: "v"(wf[0].v), "v"(wf[1].v), "v"(xf[0][0].v), "v"(xf[2][1].v));
        asm volatile("" : "+v"(acc[J][0]), "+v"(acc[J][1]), "+v"(acc[J][2]));   // pins the (pure) MFMAs inside their matrix segment
        __builtin_amdgcn_s_setprio(0);
      };
      // one k-tile held in stage S; ld1 / ld2: k-tiles kt+1 / kt+2 exist
      auto ktile = [&](auto Sc, int kt, bool ld1, bool ld2) {
        constexpr int S = decltype(Sc)::value;
        const char* st = smem + S * QS_STAGE;
        // ---- phase 0: feature block 0 ----
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) xf[i][kk >> 1].h[kk & 1] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 4096 + fo[kk]);
        read_w(st, 0, S);                                            // (nk is even: the parity of k-tile kt is the stage S)
        if (MX) {
#pragma unroll
          for (int i = 0; i < 3; ++i) sx[i] = *reinterpret_cast<const uint32_t*>(smem + fs_x + i * 128 + S * QS_SC_BYTES) >> shi;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ld1) { issue_w(1, S ^ 1, kt + 1); issue_a(1, S ^ 1, kt + 1); issue_a(2, S ^ 1, kt + 1); }      // W1, A1, A2 of k-tile kt+1
        qs_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(qs_ic<0>{});
        __builtin_amdgcn_sched_barrier(0);
        qs_barrier();
        // ---- phase 1: feature block 1 ----
        read_w(st, 1, S);
        __builtin_amdgcn_sched_barrier(0);
        // (MX: the scale piece of k-tile kt+1 goes in FRONT of W2 - the phase-2 wait below then retires it with A | W0 | W1 - into the other parity's area, last read in phase 0 of kt-1)
        if (ld1) { if (MX) issue_sc(kt + 1); issue_w(2, S ^ 1, kt + 1); if (MX) qs_wait_vmcnt<10>(); else qs_wait_vmcnt<9>(); } else qs_wait_vmcnt<0>();   // W2 of k-tile kt+1 issued; W2 of this k-tile has landed
        qs_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(qs_ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
        qs_barrier();
        // ---- phase 2: feature block 2 ----
        read_w(st, 2, S);
        __builtin_amdgcn_sched_barrier(0);
        if (ld2) { issue_w(0, S, kt + 2); issue_a(0, S, kt + 2); qs_wait_vmcnt<5>(); }                     // W0, A0 of k-tile kt+2; A | W0 | W1 of k-tile kt+1 have landed
        else if (ld1) qs_wait_vmcnt<2>();
        else qs_wait_vmcnt<0>();
        qs_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(qs_ic<2>{});
        __builtin_amdgcn_sched_barrier(0);
        qs_barrier();
      };
      if (wm == 1) qs_barrier();                                    // waves 4-7 run one barrier behind waves 0-3
#pragma unroll 1
      for (int kt = 0; kt < nk; kt += 2) {
        ktile(qs_ic<0>{}, kt, true, kt + 2 < nk);
        ktile(qs_ic<1>{}, kt + 1, kt + 2 < nk, kt + 3 < nk);
      }
      if (wm == 0) qs_barrier();                                    // re-align; every wave is done with both stages
    }

    // ---- epilogue (1): accumulators -> bf16 -> the attention's K | V | Q arrays (they overlay the operand stages) ---------------------------------------
    {
      int etid = threadIdx.x;
      asm volatile("" : "+v"(etid));
      const int el31 = etid & 31, ehi = (etid & 63) >> 5;
      // the side rows first: they sit at 144 KiB, inside the second head's Q array - copy them out before anything is written there
      uint4 sv = make_uint4(0u, 0u, 0u, 0u);
      if (etid < 240) sv = *reinterpret_cast<const uint4*>(smem + QS_SIDE_OFF + etid * 16);
      // MASK: the group's key flags -> additive score terms (the S accumulators START at them: 0 or -inf, no arithmetic in the softmax); the byte loads fly under the hand-over
      uint8_t kflag = 1;
      if (MASK && etid >= 256 && etid < 256 + QS_TOK + 1) { const int k = etid - 256; kflag = k == 0 ? p.key_keep[seq * p.seq_rows] : p.key_keep[xrow0 + k - 1]; }
      qs_barrier();
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int cb = (wn * 3 + j) * 32;                           // first tile column of the block (wave-uniform)
        const int hd = cb / 192, within = cb - hd * 192, which = within >> 6, feat0 = within & 63;
        const int arr = hd * 3 * QS_ARR + (which == 1 ? 0 : (which == 2 ? QS_ARR : 2 * QS_ARR));          // K | V | Q per head
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int row = wm * 96 + i * 32 + el31 + (which ? 1 : 0);                                      // key 0 is the CLS row
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack_bf2(acc[j][i][g * 4 + 0], acc[j][i][g * 4 + 1]);
            w.y = pack_bf2(acc[j][i][g * 4 + 2], acc[j][i][g * 4 + 3]);
            *reinterpret_cast<uint2*>(smem + arr + qs_arr_off(row, (feat0 >> 3) + g) + ehi * 8) = w;
          }
        }
      }
      if (etid < 240) {                                             // side rows: CLS -> key 0 / the free query slot 196; left-over token i -> query 192 + i, key 193 + i
        const int ch = etid & 7, w3 = (etid >> 3) % 3, hs = (etid >> 3) / 3, srow = hs % 5, hd = hs / 5;
        const int arr = hd * 3 * QS_ARR + (w3 == 1 ? 0 : (w3 == 2 ? QS_ARR : 2 * QS_ARR));
        const int row = w3 == 0 ? (srow == 0 ? QS_TOK : QS_ROWS + srow - 1) : (srow == 0 ? 0 : QS_ROWS + srow);
        *reinterpret_cast<uint4*>(smem + arr + qs_arr_off(row, ch)) = sv;
      }
      if (MASK && etid >= 256 && etid < 256 + 208) {
        const int k = etid - 256;
        reinterpret_cast<float*>(smem + QS_MASK_OFF)[k] = (k <= QS_TOK && kflag) ? 0.f : -INFINITY;
      }
      if (etid >= 256 && etid < 256 + 176) {                        // V rows 197 .. 207 of both heads = 0: P is zero there and must meet finite values
        const int x = etid - 256, ch = x & 7, rr = (x >> 3) % 11, hd = (x >> 3) / 11;
        *reinterpret_cast<uint4*>(smem + hd * 3 * QS_ARR + QS_ARR + qs_arr_off(QS_TOK + 1 + rr, ch)) = make_uint4(0u, 0u, 0u, 0u);
      }
      qs_barrier();
    }

    // ---- epilogue (2): the space attention of (frame, head), the arithmetic of attn_mfma_kernel<64, 13> -----------------------------------------------------
    // A work unit is a PAIR of 16-query tiles of one head: every K fragment (S^T = K Q^T) and every V^T fragment (P V) read from LDS feeds two MFMAs - the
    // attention is bound by the LDS port (a 16-query tile reads all of K and V: 56 KB; 26 tiles per work item = 1.4 MB against 0.15 MB of arrays), pairs halve it.
    // 13 tiles per head = 6 pairs + tile 12 alone (its partner slot repeats tile 12, nothing of it is stored): 14 units over the 8 waves, rotating from tile to tile.
    if (!(QS_ABL & 1)) {
      int atid = threadIdx.x;
      asm volatile("" : "+v"(atid));
      const int alane = atid & 63, fr_ = alane & 15, fg = alane >> 4;
      constexpr int NKT = 13, nq = QS_TOK, nkeys = QS_TOK + 1;
      int v_off[4];
      {
        const int krow = fg * 4 + (fr_ >> 2), c1 = (fr_ & 3) >> 1, hb = (fr_ & 1) * 8;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) v_off[dt] = qs_arr_off(krow, dt * 2 + c1) + hb;
      }
#ifndef QS_SKEW
#define QS_SKEW 0
#endif
      if (QS_SKEW && wave >= 4) __builtin_amdgcn_s_sleep(QS_SKEW);   // (experiment: the two waves of a SIMD out of phase - one's softmax under the other's MFMAs)
      // one work unit: NT = 2 query tiles (qt0, qt1) of head h sharing every K / V fragment, or tile 12 alone (NT = 1)
      auto unit = [&](auto NTc, const int h, const int qt0, const int qt1) {
        constexpr int NT = decltype(NTc)::value;
        const int head = hp * 2 + h;
        const char* k_lds = smem + h * 3 * QS_ARR;
        const char* v_lds = k_lds + QS_ARR;
        const char* q_lds = k_lds + 2 * QS_ARR;
        bf16_t* obase = p.out + (seq * p.seq_rows + 1 + (int64_t)f * QS_TOK) * p.ldo + head * 64;
        bf16x8 qf[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          qf[0][ks] = *reinterpret_cast<const bf16x8*>(q_lds + qs_arr_off(qt0 * 16 + fr_, ks * 4 + fg));
          if (NT == 2) qf[1][ks] = *reinterpret_cast<const bf16x8*>(q_lds + qs_arr_off(qt1 * 16 + fr_, ks * 4 + fg));
        }
        f32x4 s[2][NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          s[0][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (MASK) { const float4 mk = *reinterpret_cast<const float4*>(smem + QS_MASK_OFF + (kt * 16 + fg * 4) * 4); s[0][kt] = f32x4{mk.x, mk.y, mk.z, mk.w}; }   // keys kt * 16 + fg * 4 + r
          s[1][kt] = s[0][kt];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            if (QS_ABL & 8) continue;
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + qs_arr_off(kt * 16 + fr_, ks * 4 + fg));
            s[0][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[0][kt], 0, 0, 0);
            if (NT == 2) s[1][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[1][kt], 0, 0, 0);
          }
        }
        float msc_[2], l_[2], linv_[2];
        bool cls_[2];
#pragma unroll
        for (int e = 0; e < NT; ++e) {
          const int qt = e ? qt1 : qt0;
          const bool cls_slot = qt * 16 + fr_ == nq;                // this lane's query column is the CLS query
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((NKT - 1) * 16 + fg * 4 + r >= nkeys) s[e][NKT - 1][r] = -INFINITY;
          if (cls_slot && f != 0 && fg == 0) s[e][0][0] = -INFINITY;   // the CLS key itself is counted by frame 0's record only
          float m = -INFINITY;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[e][kt][r]);
          m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
          const float msc = m * sc2;
          const float msafe = m == -INFINITY ? 0.f : msc;
          // exp2(s * sc2 - m * sc2) on float2 (v_pk_fma_f32 / v_pk_add_f32: the softmax arithmetic is what bounds this epilogue next to the quarter-rate v_exp_f32)
#ifndef QS_PK
#define QS_PK 1
#endif
          const sf_f32x2_t sc2v = {sc2, sc2}, mneg = {-msafe, -msafe};
          sf_f32x2_t l2 = {0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              if (QS_ABL & 4) { l2.x += s[e][kt][r]; l2.y += s[e][kt][r + 1]; }
              else if (QS_PK) {
                const sf_f32x2_t a2 = sf_f32x2_t{s[e][kt][r], s[e][kt][r + 1]} * sc2v + mneg;
                const sf_f32x2_t ex = {__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                s[e][kt][r] = ex.x; s[e][kt][r + 1] = ex.y;
                l2 = l2 + ex;
              } else {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(s[e][kt][r], sc2, -msafe)), e1 = __builtin_amdgcn_exp2f(fmaf(s[e][kt][r + 1], sc2, -msafe));
                s[e][kt][r] = e0; s[e][kt][r + 1] = e1; l2.x += e0; l2.y += e1;
              }
            }
          float l = l2.x + l2.y;
          l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
          msc_[e] = msc; l_[e] = l; linv_[e] = 1.0f / l; cls_[e] = cls_slot;
        }
        f32x4 o[2][4];
#pragma unroll
        for (int e = 0; e < NT; ++e)
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[e][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < ((QS_ABL & 16) ? 1 : (NKT + 1) / 2); ++kk) {
          union { bf16x8 v; uint32_t u[4]; } pa[2];
#pragma unroll
          for (int e = 0; e < NT; ++e) {
            pa[e].u[0] = pack_bf2(s[e][2 * kk][0], s[e][2 * kk][1]);
            pa[e].u[1] = pack_bf2(s[e][2 * kk][2], s[e][2 * kk][3]);
            if (2 * kk + 1 < NKT) {
              const int t1 = 2 * kk + 1 < NKT ? 2 * kk + 1 : 0;
              pa[e].u[2] = pack_bf2(s[e][t1][0], s[e][t1][1]);
              pa[e].u[3] = pack_bf2(s[e][t1][2], s[e][t1][3]);
            } else { pa[e].u[2] = 0; pa[e].u[3] = 0; }
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            union { bf16x8 v; qs_s4 hh[2]; } vb;
            vb.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((qs_lds_s4*)(v_lds + v_off[dt] + kk * 32 * 128));
            vb.hh[1] = qs_s4{0, 0, 0, 0};
            if (2 * kk + 1 < NKT) vb.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((qs_lds_s4*)(v_lds + v_off[dt] + (kk * 32 + 16) * 128));
            o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb.v, pa[0].v, o[0][dt], 0, 0, 0);
            if (NT == 2) o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb.v, pa[1].v, o[1][dt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int e = 0; e < NT; ++e) {
          const int qo = (e ? qt1 : qt0) * 16 + fr_;
          if (cls_[e]) {                                            // unnormalised partial of the CLS query over this frame's keys
            float* part = p.cls_part + ((seq * 12 + head) * 8 + f) * 66;
            if (fg == 0) { part[0] = msc_[e]; part[1] = l_[e]; }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
              for (int r = 0; r < 4; ++r) part[2 + dt * 16 + fg * 4 + r] = o[e][dt][r];
          }
          if (MX && QS_MXOUT && p.out_q) {
            // MXFP8 output (the A operand of the MX projection that follows), exactly as attn_mfma_kernel<64, 13, true>: the head's 64 dims are two scale blocks; block b =
            // dim tiles 2b, 2b + 1, 8 values in this lane and 8 in each of the lanes fg' != fg of the same query (16 and 32 lanes away); quantised from the bf16-rounded value
            const float linv = linv_[e];
            const int64_t row = xrow0 + (qo < nq ? qo : nq - 1);
            uint8_t* qrow = p.out_q + row * p.ldq + head * 64 + (fg & 1) * 16 + (fg >> 1) * 8;
            uint32_t be2 = 0;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              float fv[8];
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) {
                const uint32_t p01 = pack_bf2(o[e][2 * b + hh][0] * linv, o[e][2 * b + hh][1] * linv), p23 = pack_bf2(o[e][2 * b + hh][2] * linv, o[e][2 * b + hh][3] * linv);
                fv[hh * 4 + 0] = __uint_as_float(p01 << 16); fv[hh * 4 + 1] = __uint_as_float(p01 & 0xffff0000u);
                fv[hh * 4 + 2] = __uint_as_float(p23 << 16); fv[hh * 4 + 3] = __uint_as_float(p23 & 0xffff0000u);
              }
              float amax = fmaxf(fmaxf(fmaxf(fabsf(fv[0]), fabsf(fv[1])), fmaxf(fabsf(fv[2]), fabsf(fv[3]))), fmaxf(fmaxf(fabsf(fv[4]), fabsf(fv[5])), fmaxf(fabsf(fv[6]), fabsf(fv[7]))));
              amax = fmaxf(amax, __shfl_xor(amax, 16, 64)); amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
              const int be = sf_mx_exp(amax);
              const float inv = sf_mx_inv(be);
              be2 |= (uint32_t)be << (8 * b);
              const uint32_t d_lo = sf_fp8x4(fv, inv), d_hi = sf_fp8x4(fv + 4, inv);      // dims 32 b + fg * 4 + 0..3 and 32 b + 16 + fg * 4 + 0..3
              const uint32_t got = (uint32_t)__shfl_xor((int)((fg & 1) ? d_lo : d_hi), 16, 64);
              uint2 w;
              if (fg & 1) { w.x = got; w.y = d_hi; } else { w.x = d_lo; w.y = got; }
              if (qo < nq) *reinterpret_cast<uint2*>(qrow + b * 32) = w;
            }
            if (qo < nq && fg == 0) *reinterpret_cast<uint16_t*>(p.out_s + (int64_t)(head >> 1) * p.splane + row * 4 + (head & 1) * 2) = (uint16_t)be2;
          } else
#ifndef QS_OUT_LINES
#define QS_OUT_LINES 0   // (measured: 1319 us against 1312 us - the space attention is long enough to retire its 8-byte stores; sf_qkv_time2.hip gains 28 us from the same change)
#endif
          if (QS_OUT_LINES) {
            // through the tile's own 16 Q rows (dead: only this unit read them): the accumulator layout would store 8 bytes per lane, 32 bytes apart - every 128-byte line of
            // `out` (one token, one head) in 16 pieces over 4 instructions; read back row-wise below, 8 lanes x 16 bytes write one complete line
            const float linv = linv_[e];
            char* qrows = const_cast<char*>(q_lds);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              uint2 w;
              w.x = pack_bf2(o[e][dt][0] * linv, o[e][dt][1] * linv);
              w.y = pack_bf2(o[e][dt][2] * linv, o[e][dt][3] * linv);
              *reinterpret_cast<uint2*>(qrows + qs_arr_off(qo, dt * 2 + (fg >> 1)) + (fg & 1) * 8) = w;
            }
          } else
          if (qo < nq) {
            bf16_t* orow = obase + (int64_t)qo * p.ldo + fg * 4;
            const float linv = linv_[e];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              uint2 w;
              w.x = pack_bf2(o[e][dt][0] * linv, o[e][dt][1] * linv);
              w.y = pack_bf2(o[e][dt][2] * linv, o[e][dt][3] * linv);
#ifndef QS_OUT_NT
#define QS_OUT_NT 0   // the nt hint on the output stores: measured 1345 us against 1300 us on the 8-byte stores of the accumulator layout (partial lines), see QS_OUT_LINES for whole lines
#endif
              typedef unsigned int qs_u2 __attribute__((ext_vector_type(2)));
              const qs_u2 wv = {w.x, w.y};
              if (QS_OUT_NT) __builtin_nontemporal_store(wv, reinterpret_cast<qs_u2*>(orow + dt * 16)); else *reinterpret_cast<qs_u2*>(orow + dt * 16) = wv;
            }
          }
        }
        if (QS_OUT_LINES && !(MX && QS_MXOUT && p.out_q)) {
#pragma unroll
          for (int e = 0; e < NT; ++e)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int row = (e ? qt1 : qt0) * 16 + (alane >> 3) + 8 * j, ch = alane & 7;
              const uint4 w = *reinterpret_cast<const uint4*>(q_lds + qs_arr_off(row, ch));
              if (row < nq) {
                typedef unsigned int qs_u4 __attribute__((ext_vector_type(4)));
                const qs_u4 wv = {w.x, w.y, w.z, w.w};
                if (QS_OUT_NT) __builtin_nontemporal_store(wv, reinterpret_cast<qs_u4*>(obase + (int64_t)row * p.ldo + ch * 8));
                else *reinterpret_cast<qs_u4*>(obase + (int64_t)row * p.ldo + ch * 8) = wv;
              }
            }
        }
      };
#pragma unroll 1
      for (int uu = 0; uu < 2; ++uu) {
        // 12 pair units (6 per head) + the two single tiles (12 of each head) as units 12, 13; wave w takes units (w + tile count) mod 8 and that + 8: the
        // four units u, u + 4, u + 8, u + 12 share a SIMD (waves w, w + 4), so the two SIMDs with four units are the ones that hold a single: 7 | 7 | 6 | 6 tiles
        const int u = ((wave + (int)tcount) & 7) + 8 * uu;         // wave-uniform
        if (u >= 14) break;
        if (u < 12) { const int h = u >= 6 ? 1 : 0, pu = u - 6 * h; unit(qs_ic<2>{}, h, 2 * pu, 2 * pu + 1); }
        else unit(qs_ic<1>{}, u - 12, 12, 12);
      }
    }
    qs_barrier();                                                   // every wave is out of the attention arrays: the next tile's operands may land
    t += per_xcd_blocks;
    ++tcount;
    if (t >= t_end) break;
  }
}

__global__ __launch_bounds__(512, 2) void qkv_space_attn_kernel(QsArgs p) { qkv_space_attn_body<false>(p); }
__global__ __launch_bounds__(512, 2) void qkv_space_attn_mx_kernel(QsArgs p) { qkv_space_attn_body<true>(p); }
__global__ __launch_bounds__(512, 2) void qkv_space_attn_masked_kernel(QsArgs p) { qkv_space_attn_body<false, true>(p); }

// X (n_seq * seq_rows, 768) bf16 = norm1(x), seq_rows = 1 + 8 * 196 rows [CLS; frame-major patches] per sequence; W (2304, 768) bf16 = attn.qkv.weight, bias 2304 fp32 or
// NULL; side (n_seq * 33, 2304) bf16 = the same projection of [the CLS row; per frame f its tokens 192 .. 195] (row seq * 33, rows seq * 33 + 1 + 4 f + i), computed by the
// caller with sf_gemm_bf16 on a gathered copy of those rows; out (rows as X, 768) bf16: the PATCH rows are written (row 0 of every sequence comes from
// sf_attention_cls_combine on cls_partial [n_seq][12][8][66] fp32, one record per frame as sf_attention_cls_partial writes them).  out must not alias X (other workgroups
// still read X).  Reference: vit_helper.py:97-150 with the '(b f) n d' groups of :341-342, heads = 12, head dim 64, q scaled by `scale` (vit_helper.py:113).
static int qs_launch(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                     uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, const uint8_t* key_keep, void* stream) {
  SF_CHECK_ARG(X && W && side && out && cls_partial, "sf_qkv_space_attention: null pointer");
  SF_CHECK_ARG(n_tok == QS_TOK, "sf_qkv_space_attention: built for 196-token frames (8 frames per sequence), got %d", n_tok);
  SF_CHECK_ARG((ldx % 64) == 0 && (ldw % 64) == 0 && (lds_ % 8) == 0 && (ldo % 8) == 0 && ldx >= QS_D && ldw >= QS_D && lds_ >= 3 * QS_D && ldo >= QS_D,
               "sf_qkv_space_attention: bad row strides (ldx / ldw multiples of 64 elements: the chunk slot lives in the low 7 bits of a piece's byte offset)");
  SF_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)side % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0) &&
                   ((uintptr_t)cls_partial % 8) == 0, "sf_qkv_space_attention: operands must be 16-byte aligned");
  SF_CHECK_ARG((const void*)X != (const void*)out, "sf_qkv_space_attention: out must not alias X");
  if (n_seq <= 0) return 0;
  const int64_t seq_rows = 1 + 8 * (int64_t)QS_TOK;
  SF_CHECK_ARG((int64_t)QS_TOK * ldx * 2 < ((int64_t)1 << 32) && (int64_t)3 * QS_D * ldw * 2 < ((int64_t)1 << 32) && (int64_t)33 * lds_ * 2 < ((int64_t)1 << 32),
               "sf_qkv_space_attention: a frame of X, W and a sequence's side rows must stay below 4 GiB (32-bit lane offsets)");
  SF_CHECK_ARG(n_seq * 8 * 6 < ((int64_t)1 << 31), "sf_qkv_space_attention: too many tiles");
  if (int rc = sf_prepare_kernel(key_keep ? (const void*)qkv_space_attn_masked_kernel : (const void*)qkv_space_attn_kernel, QS_LDS, "sf_qkv_space_attention")) return rc;
  const int n_cu = sf_cu_count("sf_qkv_space_attention");
  if (n_cu <= 0) return -1;
  QsArgs a;
  a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.bias = bias; a.side = side; a.lds_ = lds_; a.out = out; a.ldo = ldo; a.cls_part = cls_partial;
  a.seq_rows = seq_rows; a.n_frames = (uint32_t)(n_seq * 8); a.scale = scale; a.key_keep = key_keep;
  static int env_hc = -1;
  if (env_hc < 0) { const char* e = getenv("SF_QS_PAIR_CHUNK"); env_hc = e ? atoi(e) : 6; if (env_hc < 1 || 6 % env_hc) env_hc = 6; }
  a.pair_chunk = (uint32_t)env_hc;
  int64_t blocks = (n_cu / 8) * 8;                               // one workgroup per CU, a multiple of the 8 XCD slots of the tile schedule (blockIdx % 8) ...
  if (blocks < 8) blocks = 8;                                    // ... and never an empty grid on a device / partition with fewer than 8 CUs
  const int64_t need = ((n_seq * 8 * 6 + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  if (key_keep) hipLaunchKernelGGL(qkv_space_attn_masked_kernel, dim3((unsigned)blocks), dim3(512), QS_LDS, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(qkv_space_attn_kernel, dim3((unsigned)blocks), dim3(512), QS_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
extern "C" int sf_qkv_space_attention(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                                      uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, void* stream) {
  return qs_launch(X, ldx, W, ldw, bias, side, lds_, out, ldo, cls_partial, n_seq, n_tok, scale, nullptr, stream);
}
// The same launch with TOKEN MASKS (Synchformer.forward(vis_mask=...), sync_model.py:72-80 -> the -inf key masks of vit_helper.py:107-141): key_keep holds one byte per
// row of X; a row with flag 0 is a masked KEY for every query of its frame's group and for the CLS query (its own output row is still computed, as in the reference).
// The flags become additive terms (0 / -inf) the score accumulators start from: an all-ones mask is bit-identical to sf_qkv_space_attention.
extern "C" int sf_qkv_space_attention_masked(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                                             uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, const uint8_t* key_keep, void* stream) {
  SF_CHECK_ARG(key_keep, "sf_qkv_space_attention_masked: null key_keep (call sf_qkv_space_attention for an unmasked forward)");
  return qs_launch(X, ldx, W, ldw, bias, side, lds_, out, ldo, cls_partial, n_seq, n_tok, scale, key_keep, stream);
}

// The same launch on MXFP8 operands (fp8 towers of the synchronizability fine-tune): X (rows, 768) e4m3 bytes with its stage-major scale planes sX (6 planes, ldsx bytes
// apart, one dword per row) - what sf_gemm_mx_res_ln768 writes -, W (2304, 768) e4m3 + sW (6 planes of 2304 dwords); side (n_seq * 33, 2304) bf16 as in
// sf_qkv_space_attention (from sf_gemm_mxfp8 on gathered copies of the rows and of their scale dwords).  The attention runs on the bf16-rounded projection, exactly as on the
// un-fused MX path (sf_gemm_mxfp8 with a bf16 output, then sf_attention_cls_partial(_mx)).  Output: EITHER out (bf16, patch rows) OR out_q / out_s (e4m3 bytes (rows, 768) +
// the scale planes [6][rows][4], splane bytes apart: byte for byte sf_quantize_mxfp8 of the bf16 output - the A operand of the MX projection that follows; buffers of
// their own, not X / sX).  Replaces sf_gemm_mxfp8 (spatial qkv) + sf_attention_cls_partial_mx.
extern "C" int sf_qkv_space_attention_mx(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW, int64_t ldsw,
                                         const float* bias, const uint16_t* side, int64_t lds_, uint16_t* out, int64_t ldo, uint8_t* out_q, int64_t ldq, uint8_t* out_s,
                                         int64_t splane, float* cls_partial, int64_t n_seq, int n_tok, float scale, void* stream) {
  SF_CHECK_ARG(X && sX && W && sW && side && cls_partial && ((out != nullptr) != (out_q != nullptr)), "sf_qkv_space_attention_mx: null pointer (exactly one of out / out_q)");
  SF_CHECK_ARG(n_tok == QS_TOK, "sf_qkv_space_attention_mx: built for 196-token frames (8 frames per sequence), got %d", n_tok);
  SF_CHECK_ARG((ldx % 128) == 0 && (ldw % 128) == 0 && ldx >= QS_D && ldw >= QS_D && (lds_ % 8) == 0 && lds_ >= 3 * QS_D, "sf_qkv_space_attention_mx: bad row strides (ldx / ldw multiples of 128 bytes)");
  SF_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)sX % 16) == 0 && ((uintptr_t)sW % 16) == 0 && ((uintptr_t)side % 16) == 0 &&
                   (!bias || ((uintptr_t)bias % 16) == 0) && ((uintptr_t)cls_partial % 8) == 0, "sf_qkv_space_attention_mx: operands must be 16-byte aligned");
  if (out) SF_CHECK_ARG(((uintptr_t)out % 16) == 0 && (ldo % 8) == 0 && ldo >= QS_D, "sf_qkv_space_attention_mx: out must be a 16-byte aligned bf16 buffer");
  if (out_q) SF_CHECK_ARG(out_s && ((uintptr_t)out_q % 8) == 0 && ((uintptr_t)out_s % 2) == 0 && (ldq % 8) == 0 && ldq >= QS_D && out_q != X && out_s != sX,
                          "sf_qkv_space_attention_mx: out_q (8-byte aligned, ldq %% 8 == 0) / out_s must be buffers of their own");
  if (n_seq <= 0) return 0;
  const int64_t seq_rows = 1 + 8 * (int64_t)QS_TOK;
  if (out_q) SF_CHECK_ARG(splane >= n_seq * seq_rows * 4, "sf_qkv_space_attention_mx: a scale plane holds 4 bytes per row");
  SF_CHECK_ARG((ldsx % 16) == 0 && (ldsw % 16) == 0 && ldsx >= n_seq * seq_rows * 4 && ldsw >= 3 * QS_D * 4, "sf_qkv_space_attention_mx: scale planes must hold one dword per row of X / W");
  SF_CHECK_ARG((int64_t)QS_TOK * ldx < ((int64_t)1 << 32) && (int64_t)3 * QS_D * ldw < ((int64_t)1 << 32) && (int64_t)33 * lds_ * 2 < ((int64_t)1 << 32),
               "sf_qkv_space_attention_mx: a frame of X, W and a sequence's side rows must stay below 4 GiB (32-bit lane offsets)");
  SF_CHECK_ARG(n_seq * 8 * 6 < ((int64_t)1 << 31), "sf_qkv_space_attention_mx: too many tiles");
  if (int rc = sf_prepare_kernel((const void*)qkv_space_attn_mx_kernel, QS_LDS, "sf_qkv_space_attention_mx")) return rc;
  const int n_cu = sf_cu_count("sf_qkv_space_attention_mx");
  if (n_cu <= 0) return -1;
  QsArgs a;
  a.X = reinterpret_cast<const bf16_t*>(X); a.ldx = ldx; a.W = reinterpret_cast<const bf16_t*>(W); a.ldw = ldw; a.bias = bias; a.side = side; a.lds_ = lds_;
  a.out = out; a.ldo = ldo; a.cls_part = cls_partial; a.seq_rows = seq_rows; a.n_frames = (uint32_t)(n_seq * 8); a.scale = scale; a.pair_chunk = 6;
  a.sX = sX; a.ldsx = ldsx; a.sW = sW; a.ldsw = ldsw; a.out_q = out_q; a.ldq = ldq; a.out_s = out_s; a.splane = splane;
  int64_t blocks = (n_cu / 8) * 8;                               // one workgroup per CU, a multiple of the 8 XCD slots of the tile schedule (blockIdx % 8) ...
  if (blocks < 8) blocks = 8;                                    // ... and never an empty grid on a device / partition with fewer than 8 CUs
  const int64_t need = ((n_seq * 8 * 6 + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL(qkv_space_attn_mx_kernel, dim3((unsigned)blocks), dim3(512), QS_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
