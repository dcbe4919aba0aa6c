// Motionformer TIME attention fused into its qkv projection, round 4 (gfx950): sf_qkv_time_attention on the main loop of sf_qkv_space_attention.
//
// sf_qkv_time.hip (round 2/3) gives a workgroup 256 token rows x ONE head's q | k | v (192 features) with every wave on 32 rows x all 192 features, so that a wave ends
// with everything its 4 patches need: 28 KiB of LDS fragment reads per wave and 64-deep k-tile for 0.79 MFLOP - 224 KiB per k-tile and CU = 1792 cycles of the LDS port
// against 1546 cycles of MFMAs: the launch is bound by fragment reads (893 TFLOP/s).  This schedule takes the 192 x 384 tile of sf_qkv_space.hip - 2 x 4 waves of
// 96 x 96, 24 KiB of fragment reads per wave for 1.18 MFLOP, three phases per k-tile, two 72-KiB stages, counted vmcnt waits, staggered wave groups: 1.3 PFLOP/s there -
// and pays with an exchange of q | k | v through LDS before the attention, as the spatial kernel does.
//
//   * Work item = (sequence, PATCH BLOCK tb, HEAD PAIR).  A sequence's 196 patches are 8 blocks of 24 + 4 left over.  The GEMM tile is the block's 24 patches x their 8
//     frames = 192 token rows, frame-major (tile row R = 24 fr + pl <-> X row seq * 1569 + 1 + 196 fr + 24 tb + pl: eight runs of 24 contiguous rows, gathered by the
//     LDS-DMA source addresses - wave w loads frame w), x the q | k | v of two heads (384 features).  DividedAttention.forward with the '(b n) f d' groups
//     (vit_helper.py:97-150, :343-344): every patch attends [CLS key; its own 8 frames].
//   * Epilogue (1): accumulators (+ bias) -> bf16 -> LDS arrays K | V | Q per head, rows PATCH-major (row' = 8 pl + fr), 128-byte rows with the 16-byte chunks XOR-swizzled
//     by (pl + fr) & 7 - the lanes of a store (consecutive pl of one frame) and the rows of a 16-row fragment read (two patches x 8 frames) both spread over the banks.
//     The sequence's CLS q | k | v (from `side`, landed by LDS-DMA during the main loop) go to row 0 of three 16-row blocks per head (the rest of them zero).
//   * Epilogue (2): a 16-row tile of the arrays = 2 patches x 8 frames is one attention unit on v_mfma_f32_16x16x32_bf16, the arithmetic of attn_mfma_kernel
//     (sf_attention.hip): S^T = K Q^T against key tile {the unit's own 16 rows} and key tile {CLS block}, scores of the other patch and of the block's empty rows
//     masked to -inf, in-lane base-2 softmax, P V with ds_read_b64_tr_b16 V fragments.  24 units per item (2 heads x 12), three per wave, run stage by stage so that
//     their dependency chains overlap.  The CLS QUERY of the sequence (vit_helper.py:126: it attends every token) meets the item's 192 keys in eight pieces - wave w: head
//     w & 1, key tiles 3 (w >> 1) .. + 2 - each writing its softmax partial as record 4 tb + (w >> 1) of cls_partial [seq][head][33][66] (sf_attention_cls_partial's
//     records; sf_attention_cls_combine with n_part = 33 writes the CLS row of `out`).
//   * The 4 left-over patches of a sequence (tokens 192..195 of every frame: rows seq * 33 + 1 + 4 f + i of `side`, the buffer sf_qkv_space_attention takes) are a ninth,
//     GEMM-less item per (sequence, head pair): 32 array rows loaded from `side`, 4 patch units, and the CLS query's record 32 over these 32 keys AND the CLS key itself.
//   * persistent, one workgroup per CU; every XCD owns a contiguous range of (sequence, block) row tiles and sweeps it once per chunk of head pairs.
// Token masks: sf_qkv_time_attention2_masked (template parameter MASK: the key flags become the starting values of the score accumulators).
#include "sf_common.h"
#include <type_traits>
#include <stdlib.h>
#include "../../include/synchformer_hip.h"

#define QT2_NP 196                       // patches per frame
#define QT2_TP 24                        // patches per GEMM tile
#define QT2_ROWS 192                     // 24 patches x 8 frames
#define QT2_D 768
#define QT2_A_BYTES (QT2_ROWS * 128)     // 24 KiB: 192 rows x 64 k (bf16)
#define QT2_W_PART (128 * 128)           // 16 KiB: 4 wave columns x 32 features x 64 k
#define QT2_STAGE (QT2_A_BYTES + 3 * QT2_W_PART)   // 72 KiB
#ifndef QT2_ABL_HALF
#define QT2_ABL_HALF 0
#endif
#define QT2_ARR (QT2_ROWS * 128)         // one K / V / Q array: 192 rows of 128 B (24 KiB); the six arrays overlay the two operand stages exactly
#define QT2_CB_OFF (6 * QT2_ARR)         // 144 KiB: CLS blocks Kc[2] | Vc[2] | Qc[2], 16 rows of 128 B each, row 0 = the sequence's CLS k / v / q of the head, rows 1..15 zero
#define QT2_BIAS_OFF (QT2_CB_OFF + 6 * 2048)       // 156 KiB: the tile's 384 bias floats
#define QT2_LAND_OFF (QT2_BIAS_OFF + 1536)         // 768 B: landing area of the CLS row's q | k | v of both heads (48 x 16 B, lane-linear)
// MX variant (fp8 towers): the k-tile's scale dwords need 2 x 2304 B next to two full operand stages - the six 2-KiB CLS blocks shrink to their only non-zero row each
// (CBX: 6 x 128 B) plus ONE shared zero row, rows 1..15 of a block being an address select, and the scales land where the blocks were
#define QT2_CBX_OFF QT2_CB_OFF                      // MX: row 0 of block b (K h0, K h1, V h0, V h1, Q h0, Q h1) at + b * 128
#define QT2_ZROW_OFF (QT2_CB_OFF + 6 * 128)         // MX: 128 zero bytes
#define QT2_SC_OFF (QT2_CB_OFF + 1024)              // MX: two parities x (192 token rows | 3 x 128 part rows of W) scale dwords = 2 x 2304 B
#define QT2_SC_BYTES 2304
#define QT2_MASK_OFF (QT2_LAND_OFF + 768)           // MASK variant: one float per array row (patch-major, 192) + the CLS key at index 192: 0 = kept, -inf = masked
#define QT2_LDS (160 * 1024)
#define QT2_NPART 33                     // CLS-query records per (sequence, head): 4 per block (one per wave pair: 3 key tiles each) + 1 of the left-over item
#ifndef QT2_ABL
#define QT2_ABL 0                        // measurement builds (WRONG results): 1 no attention units, 2 no MFMAs in the main loop, 4 no CLS-query units, 8 no patch units, 16 no units in the left-over items, 32 no output stores
#endif

struct Qt2Args {
  const bf16_t* X; int64_t ldx;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  const bf16_t* side; int64_t lds_;      // (n_seq * 33, 2304) bf16: row seq * 33 = the CLS row's q | k | v, row seq * 33 + 1 + 4 f + i = token 192 + i of frame f
  bf16_t* out; int64_t ldo;
  float* cls_part;                       // [n_seq][12][33][66]
  int64_t seq_rows;
  uint32_t n_rt;                         // n_seq * 9 row tiles: (sequence, block 0..7) and the left-over item 8
  uint32_t pair_chunk;                   // head pairs per sweep over an XCD's row tiles (divides 6)
  float scale;
  uint32_t stagger;
  const uint8_t* key_keep = nullptr;     // MASK (template parameter): token flags, one byte per row of X; flag 0 = a masked KEY (for its patch's group and for the CLS query)
  // MX (template parameter): X / W are e4m3 BYTES (ldx / ldw in bytes) with stage-major E8M0 scale planes (one dword per row per 128 k, ldsx / ldsw bytes between planes);
  // the attention output leaves as bf16 `out` OR as MXFP8 (e4m3 bytes, row stride ldq, + E8M0 bytes in the scale planes [6][rows][4], splane bytes apart)
  const uint8_t* sX = nullptr; int64_t ldsx = 0;
  const uint8_t* sW = nullptr; int64_t ldsw = 0;
  uint8_t* out_q = nullptr; int64_t ldq = 0; uint8_t* out_s = nullptr; int64_t splane = 0;
};
typedef __attribute__((ext_vector_type(8))) int qt2_i32x8;

typedef short qt2_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) qt2_s4 qt2_lds_s4;

__device__ __forceinline__ void qt2_dma1(uint32_t voff, const void* sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}
__device__ __forceinline__ void qt2_dma_dword_addr(const void* gaddr, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gaddr), "s"(lds) : "memory");
}
template <int N>
__device__ __forceinline__ void qt2_wait_vmcnt() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void qt2_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint32_t qt2_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
// array row (patch-major: row = 8 pl + fr; also a CLS block's row 0..15), 16-byte chunk 0..7 -> byte offset; the chunk slot is XOR-swizzled by (pl + fr) & 7
__device__ __forceinline__ int qt2_arr_off(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 3) + row) & 7)) << 4); }

template <int V> using qt2_ic = std::integral_constant<int, V>;

template <bool MASK, bool MX = false>
__device__ __forceinline__ void qkv_time2_attn_body(const Qt2Args& p) {
  constexpr int ESZ = MX ? 1 : 2;                                  // bytes per operand element
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                         // 2 x 4 waves, wave tile 96 token rows x 96 features
  const int hi = lane >> 5;

  // persistent schedule: block b sits on XCD b % 8; every XCD owns a contiguous range of row tiles and walks (row tile, head pair) in chunks of `hc` pairs
  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t r8 = (p.n_rt + 7u) >> 3;
  const uint32_t rt0 = min(xcd * r8, p.n_rt), rt1 = min(rt0 + r8, p.n_rt);
  const uint32_t hc = p.pair_chunk, chunk_tiles = (rt1 - rt0) * hc, t_end = (rt1 - rt0) * 6u;
  uint32_t t = li;
  if (t >= t_end) return;
  // optional start stagger (measurement: do the output-store bursts of 256 workgroups in lockstep cost more than a delayed start?): the groups of `hc` workgroups that share a
  // row tile start (group * 8 + xcd) * stagger * ~1.2 us apart
  for (uint32_t i = 0, n = ((li / hc) * 8u + xcd) * p.stagger; i < n; ++i) __builtin_amdgcn_s_sleep(32);

  uint32_t voff_a = 0, voff_w[3] = {0, 0, 0}, sc_voff = 0, cls_voff = 0;   // lane offsets of the LDS-DMA pieces: re-derived at the top of every tile (not kept live across the attention)
  const int sc_seg = wave % 3;
  const int sc_lanes = sc_seg == 0 ? 48 : (sc_seg == 1 ? 64 : 32);
  const uint32_t a8 = (uint32_t)(8 * p.ldx * ESZ), w8 = (uint32_t)(8 * p.ldw * ESZ);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(qt2_lds_addr(smem));
  const uint32_t lds_a_w = __builtin_amdgcn_readfirstlane(lds0 + wave * 3072);
  const uint32_t lds_w_w = __builtin_amdgcn_readfirstlane(lds0 + QT2_A_BYTES + wave * 2048);

  // the CLS blocks start as zeros; only their row 0 is ever written again
  for (int x = tid; x < (MX ? 1024 : 6 * 2048) / 16; x += 512) *reinterpret_cast<uint4*>(smem + QT2_CB_OFF + x * 16) = make_uint4(0u, 0u, 0u, 0u);
  qt2_barrier();                                                   // (a left-over item writes row 0 of the blocks before its first barrier)

  const char* xbase; const char* wbase;
  int64_t seq; int tb, hp;
  auto set_tile = [&](uint32_t tt) {
    const uint32_t c = tt / chunk_tiles, r = tt - c * chunk_tiles;
    const uint32_t rt = rt0 + r / hc;
    hp = (int)(c * hc + r % hc);
    seq = rt / 9u; tb = (int)(rt - (uint32_t)seq * 9u);
    xbase = reinterpret_cast<const char*>(p.X) + (seq * p.seq_rows + 1 + (int64_t)tb * QT2_TP) * p.ldx * ESZ;
    wbase = reinterpret_cast<const char*>(p.W) + (int64_t)hp * 128 * p.ldw * ESZ;
  };
  // A piece pc of this wave = tile rows (3 wave + pc) * 8 .. + 7 = patches 8 pc .. + 7 of the block in frame `wave`: source row 196 wave + 8 pc + (lane >> 3) past the block's
  // first row; lane (r = lane >> 3, chunk slot = lane & 7) -> LDS row-linear, source chunk XOR-swizzled by the TILE row (row strides are multiples of 128 bytes - launcher
  // check -, so the low 7 bits of an offset are its chunk slot: the piece 8 rows further down is (offset ^ 64) + 8 rows).
  // W part j, piece pc of this wave = part rows (2 wave + pc) * 8 .. + 7; part row pr = 32 wn' + rr <-> tile column wn' * 96 + 32 j + rr <-> row
  // which * 768 + hd * 64 + feat of W (the head pair's offset rides in the SGPR base)
  auto derive_offsets = [&]() {
    int dtid = threadIdx.x;
    asm volatile("" : "+v"(dtid));
    const int dl = dtid & 63;
    {
      const int r = wave * 3 * 8 + (dl >> 3);
      voff_a = (uint32_t)((int64_t)(wave * QT2_NP + (dl >> 3)) * p.ldx * ESZ + ((((dl & 7) ^ ((r >> 1) & 7))) << 4));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pr = wave * 2 * 8 + (dl >> 3);
      const int c = (pr >> 5) * 96 + j * 32 + (pr & 31);
      const int hd = c / 192, within = c - hd * 192;
      const int grow = (within >> 6) * QT2_D + hd * 64 + (within & 63);
      voff_w[j] = (uint32_t)((int64_t)grow * p.ldw * ESZ + ((((dl & 7) ^ ((pr >> 1) & 7))) << 4));
    }
    { const int ch = dl & 7, w3 = (dl >> 3) % 3, hd = (dl >> 3) / 3; cls_voff = (uint32_t)((w3 * QT2_D + hd * 64 + ch * 8) * 2); }   // wave 7, lanes 0..47: the CLS row's q | k | v chunk
    if (MX) {
      // the k-tile's scale dwords as 16-byte LDS-DMA lanes, three pieces; EVERY wave issues one (wave % 3; copies land the same bytes) so that all waves count alike:
      // piece 0 = the tile's 192 token rows in tile order (frame fr: 24 rows = 6 lanes, source rows 196 fr .. past the block's first row), pieces 1 | 2 = the 384 W rows
      // in part order (area index j * 128 + 32 wn' + rr), exactly sf_qkv_space_attention_mx's
      if (sc_seg == 0) sc_voff = (uint32_t)((dl / 6) * QT2_NP * 4 + (dl % 6) * 16);
      else {
        const int L = (sc_seg == 1 ? 0 : 64) + dl, run = L >> 3, j = run >> 2, wnp = run & 3;
        const int c = wnp * 96 + j * 32 + (L & 7) * 4;
        const int hd = c / 192, within = c - hd * 192;
        sc_voff = (uint32_t)(((within >> 6) * QT2_D + hd * 64 + (within & 63)) * 4);
      }
    }
  };
  const uint32_t sc_lds = __builtin_amdgcn_readfirstlane(lds0 + QT2_SC_OFF + (sc_seg == 0 ? 0 : (sc_seg == 1 ? 768 : 768 + 1024)));
  auto issue_sc = [&](int kt) {
    const char* base = sc_seg == 0 ? reinterpret_cast<const char*>(p.sX) + (int64_t)kt * p.ldsx + (seq * p.seq_rows + 1 + (int64_t)tb * QT2_TP) * 4
                                   : reinterpret_cast<const char*>(p.sW) + (int64_t)kt * p.ldsw + (int64_t)hp * 512;
    if (lane < sc_lanes) qt2_dma1(sc_voff, base, sc_lds + (kt & 1) * QT2_SC_BYTES);
  };
  auto issue_a = [&](int pc, int S, int kt) {                       // piece pc: 8 rows further down; the chunk swizzle flips bit 2 with every 8 rows
    qt2_dma1(((pc & 1) ? (voff_a ^ 64u) : voff_a) + (uint32_t)pc * a8, xbase + kt * 128, lds_a_w + S * QT2_STAGE + pc * 1024);
  };
  auto issue_w = [&](int j, int S, int kt) {
    qt2_dma1(voff_w[j], wbase + kt * 128, lds_w_w + S * QT2_STAGE + j * QT2_W_PART);
    qt2_dma1((voff_w[j] ^ 64u) + w8, wbase + kt * 128, lds_w_w + S * QT2_STAGE + j * QT2_W_PART + 1024);
  };

  constexpr int nk = MX ? QT2_D / 128 : QT2_D / 64;                 // 12 k-tiles of 64 bf16 / 6 of 128 fp8: 128 bytes per row either way
  const float sc2 = p.scale * 1.44269504088896f;                   // softmax in base 2
  uint32_t tcount = 0;
  // a CLS block's 16-byte chunk: block blk (0 K | 2 V | 4 Q) of head h, row 0..15 (only row 0 is ever non-zero).  bf16 kernel: six 2-KiB blocks in the arrays' swizzled
  // layout; MX kernel: row 0 of every block in CBX, rows 1..15 all the ONE zero row (an address select per lane)
  auto cb_ptr = [&](int blk, int h, int row, int chunk) -> const char* {
    if (!MX) return smem + QT2_CB_OFF + (blk + h) * 2048 + qt2_arr_off(row, chunk);
    return smem + (row == 0 ? QT2_CBX_OFF + (blk + h) * 128 : QT2_ZROW_OFF) + (chunk << 4);
  };

  for (;;) {
    set_tile(t);
    const bool main_item = tb < 8;                                  // wave-uniform (workgroup-uniform)
    if (main_item) {
      derive_offsets();
      // ---- prologue: bias, the CLS row's q | k | v, k-tile 0 and W0 | A0 of k-tile 1 (the previous item's attention is over: barrier at the bottom of the loop) ----
      if (wave < 6) {                                               // 6 x 64 bias floats: tile columns 64 wave .. + 63 = (head hd = wave / 3, q | k | v = wave % 3)
        if (p.bias) qt2_dma_dword_addr(p.bias + (wave % 3) * QT2_D + (hp * 2 + wave / 3) * 64 + lane, lds0 + QT2_BIAS_OFF + wave * 256);
      } else if (wave == 7) {                                       // CLS row: chunk x = (hd * 3 + which) * 8 + ch, 48 chunks of 16 B
        if (lane < 48) qt2_dma1(cls_voff, reinterpret_cast<const char*>(p.side + seq * 33 * p.lds_ + hp * 128), lds0 + QT2_LAND_OFF);
      }
      if (MX) issue_sc(0);
      issue_w(0, 0, 0); issue_a(0, 0, 0);
      issue_w(1, 0, 0); issue_a(1, 0, 0); issue_a(2, 0, 0);
      issue_w(2, 0, 0);
      issue_w(0, 1, 1); issue_a(0, 1, 1);
      qt2_wait_vmcnt<5>();                                          // bias, CLS row, (scales,) A | W0 | W1 of k-tile 0 (this wave's pieces) have landed
      qt2_barrier();

      // accumulators start at the bias: block (j, i) = features 96 wn + 32 j + 8 g + 4 hi + r of the tile, tokens 96 wm + 32 i + l31
      f32x16 acc[3][3];
      {
        const float* bs = reinterpret_cast<const float*>(smem + QT2_BIAS_OFF);
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
        int fo[4], fs_x = 0, fs_w = 0, shi = 0;                     // (MX: addresses of this lane's scale dwords - its token row of block 0, row l31 of its W part 0 - and 8 * (lane >> 5))
        {
          int ptid = threadIdx.x;
          asm volatile("" : "+v"(ptid));
          const int pl31 = ptid & 31, phi = (ptid & 63) >> 5;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) fo[kk] = pl31 * 128 + (((kk * 2 + phi) ^ ((pl31 >> 1) & 7)) << 4);
          if (MX) { fs_x = QT2_SC_OFF + (wm * 96 + pl31) * 4; fs_w = QT2_SC_OFF + 768 + (wn * 32 + pl31) * 4; shi = phi * 8; }
        }
        const int a_base = wm * 96 * 128, w_base = QT2_A_BYTES + wn * 32 * 128;
        // fragments as 32-byte pairs (kk = 2 k2, 2 k2 + 1): the two 16-byte reads a lane supplies to ONE 64-deep scaled MFMA sit in eight consecutive registers
        union Qt2Frag { bf16x8 h[2]; qt2_i32x8 v; };
        Qt2Frag xf[3][2], wf[2];
        uint32_t sx[3] = {0u, 0u, 0u}, sw = 0u;                      // MX: this lane's scale dwords of the k-tile (token row of block i; W row of the current part), >> shi
        auto read_w = [&](const char* st, int j, int par) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) wf[kk >> 1].h[kk & 1] = *reinterpret_cast<const bf16x8*>(st + w_base + j * QT2_W_PART + fo[kk]);
          if (MX) sw = *reinterpret_cast<const uint32_t*>(smem + fs_w + j * 512 + par * QT2_SC_BYTES) >> shi;
        };
        auto mma = [&](auto Jc) {
          constexpr int J = decltype(Jc)::value;
          __builtin_amdgcn_s_setprio(1);
          if constexpr (MX) {
            // a 128-byte LDS row = 128 fp8 k: fragments 2 k2 and 2 k2 + 1 are the 2 x 16 bytes a lane supplies to ONE 64-deep scaled MFMA; scale operands: the k-tile's dword of
            // the row shifted by 8 * (lane >> 5), BYTE 2 k2 selected by the instruction's op_sel (as sf_qkv_space_attention_mx)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
              for (int i = 0; i < 3; ++i) {
                if (k2 == 0) acc[J][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[0].v, xf[i][0].v, acc[J][i], 0 /* e4m3 */, 0 /* e4m3 */, 0, (int)sw, 0, (int)sx[i]);
                else acc[J][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[1].v, xf[i][1].v, acc[J][i], 0, 0, 2, (int)sw, 2, (int)sx[i]);
              }
            }
          } else
          if (!(QT2_ABL & 2)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
              for (int i = 0; i < 3; ++i) acc[J][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk >> 1].h[kk & 1], xf[i][kk >> 1].h[kk & 1], acc[J][i], 0, 0, 0);
          } else asm volatile("" :: "v"(wf[0].v), "v"(wf[1].v), "v"(xf[0][0].v), "v"(xf[2][1].v));
          asm volatile("" : "+v"(acc[J][0]), "+v"(acc[J][1]), "+v"(acc[J][2]));   // pins the (pure) MFMAs inside their matrix segment
          __builtin_amdgcn_s_setprio(0);
        };
        // one k-tile held in stage S; ld1 / ld2: k-tiles kt+1 / kt+2 exist (the schedule of sf_qkv_space.hip:
        //   phase 0: issue W1, A1, A2 of kt+1 (no wait) | phase 1: issue (MX: the scale piece and) W2 of kt+1, vmcnt(9 | MX 10): W2 of kt landed | phase 2: issue W0, A0 of kt+2,
        //   vmcnt(5): A, W0, W1 (and the scales) of kt+1 landed)
        auto ktile = [&](auto Sc, int kt, bool ld1, bool ld2) {
          constexpr int S = decltype(Sc)::value;
          const char* st = smem + S * QT2_STAGE;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xf[i][kk >> 1].h[kk & 1] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 4096 + fo[kk]);
          read_w(st, 0, S);                                          // (nk is even: the parity of k-tile kt is the stage S)
          if (MX) {
#pragma unroll
            for (int i = 0; i < 3; ++i) sx[i] = *reinterpret_cast<const uint32_t*>(smem + fs_x + i * 128 + S * QT2_SC_BYTES) >> shi;
          }
          __builtin_amdgcn_sched_barrier(0);
          if (ld1) { issue_w(1, S ^ 1, kt + 1); issue_a(1, S ^ 1, kt + 1); issue_a(2, S ^ 1, kt + 1); }
          qt2_barrier();
          __builtin_amdgcn_sched_barrier(0);
          mma(qt2_ic<0>{});
          __builtin_amdgcn_sched_barrier(0);
          qt2_barrier();
          read_w(st, 1, S);
          __builtin_amdgcn_sched_barrier(0);
          if (ld1) { if (MX) issue_sc(kt + 1); issue_w(2, S ^ 1, kt + 1); if (MX) qt2_wait_vmcnt<10>(); else qt2_wait_vmcnt<9>(); } else qt2_wait_vmcnt<0>();
          qt2_barrier();
          __builtin_amdgcn_sched_barrier(0);
          mma(qt2_ic<1>{});
          __builtin_amdgcn_sched_barrier(0);
          qt2_barrier();
          read_w(st, 2, S);
          __builtin_amdgcn_sched_barrier(0);
          if (ld2) { issue_w(0, S, kt + 2); issue_a(0, S, kt + 2); qt2_wait_vmcnt<5>(); }
          else if (ld1) qt2_wait_vmcnt<2>();
          else qt2_wait_vmcnt<0>();
          qt2_barrier();
          __builtin_amdgcn_sched_barrier(0);
          mma(qt2_ic<2>{});
          __builtin_amdgcn_sched_barrier(0);
          qt2_barrier();
        };
        if (wm == 1) qt2_barrier();                                 // waves 4-7 run one barrier behind waves 0-3
#pragma unroll 1
        for (int kt = 0; kt < nk; kt += 2) {
          ktile(qt2_ic<0>{}, kt, true, kt + 2 < nk);
          ktile(qt2_ic<1>{}, kt + 1, kt + 2 < nk, kt + 3 < nk);
        }
        if (wm == 0) qt2_barrier();                                 // re-align; every wave is done with both stages
      }

      // ---- epilogue (1): accumulators -> bf16 -> the K | V | Q arrays (they overlay the operand stages), rows patch-major ------------------------------------
      {
        int etid = threadIdx.x;
        asm volatile("" : "+v"(etid));
        const int el31 = etid & 31, ehi = (etid & 63) >> 5;
        // MASK: the item's key flags -> additive score terms (0 / -inf) the S accumulators start from; the byte loads fly under the hand-over stores
        uint8_t kflag = 1;
        if (MASK && etid < QT2_ROWS + 1) {
          const int fr = etid / QT2_TP, pl = etid - fr * QT2_TP;
          kflag = etid == QT2_ROWS ? p.key_keep[seq * p.seq_rows] : p.key_keep[seq * p.seq_rows + 1 + (int64_t)fr * QT2_NP + tb * QT2_TP + pl];
        }
        int rowp[3];                                                // array row of this lane's token of row block i: tile row R = 24 fr + pl -> 8 pl + fr
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int R = wm * 96 + i * 32 + el31, fr = R / QT2_TP; rowp[i] = (R - fr * QT2_TP) * 8 + fr; }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int cb = (wn * 3 + j) * 32;                         // first tile column of the block (wave-uniform)
          const int hd = cb / 192, within = cb - hd * 192, which = within >> 6, feat0 = within & 63;
          const int arr = hd * 3 * QT2_ARR + (which == 1 ? 0 : (which == 2 ? QT2_ARR : 2 * QT2_ARR));      // K | V | Q per head
#pragma unroll
          for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              uint2 w;
              w.x = pack_bf2(acc[j][i][g * 4 + 0], acc[j][i][g * 4 + 1]);
              w.y = pack_bf2(acc[j][i][g * 4 + 2], acc[j][i][g * 4 + 3]);
#if QT2_ABL_HALF   // measurement only (WRONG results): lanes l and l + 8 of a 16-lane store group share a 16-byte chunk slot - here they take different halves of it, which
                   // makes the hand-over stores conflict-free: what do the 2-way conflicts of the product layout cost?  (profiles/r06_attention_fabric.md)
              *reinterpret_cast<uint2*>(smem + arr + qt2_arr_off(rowp[i], (feat0 >> 3) + g) + (ehi ^ ((el31 >> 3) & 1)) * 8) = w;
#else
              *reinterpret_cast<uint2*>(smem + arr + qt2_arr_off(rowp[i], (feat0 >> 3) + g) + ehi * 8) = w;
#endif
            }
          }
        }
        if (etid < 48) {                                            // the CLS row: q -> Qc, k -> Kc, v -> Vc, row 0 of the head's block (row 0: chunk slot = chunk)
          const int ch = etid & 7, w3 = (etid >> 3) % 3, hd = (etid >> 3) / 3;
          const uint4 sv = *reinterpret_cast<const uint4*>(smem + QT2_LAND_OFF + etid * 16);
          *reinterpret_cast<uint4*>(const_cast<char*>(cb_ptr(w3 == 1 ? 0 : (w3 == 2 ? 2 : 4), hd, 0, ch))) = sv;
        }
        if (MASK && etid < QT2_ROWS + 1) {
          const int fr = etid / QT2_TP, pl = etid - fr * QT2_TP;
          reinterpret_cast<float*>(smem + QT2_MASK_OFF)[etid == QT2_ROWS ? QT2_ROWS : pl * 8 + fr] = kflag ? 0.f : -INFINITY;
        }
        qt2_barrier();
      }
    } else {
      // ---- the left-over item: 4 patches x 8 frames of q | k | v of both heads from `side` -> array rows 0..31 (row = 8 i + f); the CLS row -> the blocks --------
      int stid = threadIdx.x;
      asm volatile("" : "+v"(stid));
      uint4 sv[3], cv = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int x = stid + e * 512, ch = x & 7, rr = (x >> 3) & 31, w3 = (x >> 8) % 3, hd = (x >> 8) / 3;
        sv[e] = *reinterpret_cast<const uint4*>(p.side + (seq * 33 + 1 + (rr & 7) * 4 + (rr >> 3)) * p.lds_ + w3 * QT2_D + (hp * 2 + hd) * 64 + ch * 8);
      }
      if (stid < 48) {
        const int ch = stid & 7, w3 = (stid >> 3) % 3, hd = (stid >> 3) / 3;
        cv = *reinterpret_cast<const uint4*>(p.side + seq * 33 * p.lds_ + w3 * QT2_D + (hp * 2 + hd) * 64 + ch * 8);
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int x = stid + e * 512, ch = x & 7, rr = (x >> 3) & 31, w3 = (x >> 8) % 3, hd = (x >> 8) / 3;
        *reinterpret_cast<uint4*>(smem + hd * 3 * QT2_ARR + (w3 == 1 ? 0 : (w3 == 2 ? QT2_ARR : 2 * QT2_ARR)) + qt2_arr_off(rr, ch)) = sv[e];
      }
      if (stid < 48) {
        const int ch = stid & 7, w3 = (stid >> 3) % 3, hd = (stid >> 3) / 3;
        *reinterpret_cast<uint4*>(const_cast<char*>(cb_ptr(w3 == 1 ? 0 : (w3 == 2 ? 2 : 4), hd, 0, ch))) = cv;
      }
      if (MASK && stid >= 64 && stid < 64 + 33) {                   // rows 8 i + f = token 192 + i of frame f, and the CLS key
        const int r = stid - 64;
        const uint8_t kf = r == 32 ? p.key_keep[seq * p.seq_rows] : p.key_keep[seq * p.seq_rows + 1 + (int64_t)(r & 7) * QT2_NP + QT2_ROWS + (r >> 3)];
        reinterpret_cast<float*>(smem + QT2_MASK_OFF)[r == 32 ? QT2_ROWS : r] = kf ? 0.f : -INFINITY;
      }
      qt2_barrier();
    }

    // ---- epilogue (2): the time attention of the item's patches and the CLS query's partial over the item's keys ------------------------------------------------
    if (!(QT2_ABL & 1)) {
      int atid = threadIdx.x;
      asm volatile("" : "+v"(atid));
      const int alane = atid & 63, fr_ = alane & 15, fg = alane >> 4;
      const int krow = fg * 4 + (fr_ >> 2), c1 = (fr_ & 3) >> 1, hb = (fr_ & 1) * 8;      // this lane's row / chunk / half of a ds_read_b64_tr_b16 V fragment
      const int64_t orow0 = seq * p.seq_rows + 1 + (main_item ? tb * QT2_TP : QT2_ROWS);  // out row of (frame 0, patch 0 of the item)
      // NU 16-row tiles (unit u: head u / 12, tile u % 12; a tile = 2 patches x 8 frames; keys = the same 16 rows, own patch only, + the CLS key), stage by stage over
      // the units: NU independent dependency chains (LDS read -> MFMA -> cross-lane max -> exp2 -> cross-lane sum -> MFMA -> store) overlap in one wave
      auto patch_units = [&](auto NUc, const int u0, const int ustep) {
        constexpr int NU = decltype(NUc)::value;
        const char* k_lds[NU];
        int qt[NU], hh_[NU];
        f32x4 s0[NU], s1[NU];
#pragma unroll
        for (int e = 0; e < NU; ++e) {
          const int u = u0 + e * ustep, h = u >= 12 ? 1 : 0;
          qt[e] = u - 12 * h;
          k_lds[e] = smem + h * 3 * QT2_ARR;
          hh_[e] = h;
          s0[e] = f32x4{0.f, 0.f, 0.f, 0.f}; s1[e] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (MASK) {                                               // rows fg * 4 + r of the unit's tile; row 0 of the CLS block = the CLS key
            const float4 mk = *reinterpret_cast<const float4*>(smem + QT2_MASK_OFF + (qt[e] * 16 + fg * 4) * 4);
            s0[e] = f32x4{mk.x, mk.y, mk.z, mk.w};
            s1[e][0] = reinterpret_cast<const float*>(smem + QT2_MASK_OFF)[QT2_ROWS];
          }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < NU; ++e) {
            const bf16x8 qf = *reinterpret_cast<const bf16x8*>(k_lds[e] + 2 * QT2_ARR + qt2_arr_off(qt[e] * 16 + fr_, ks * 4 + fg));
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds[e] + qt2_arr_off(qt[e] * 16 + fr_, ks * 4 + fg));
            const bf16x8 cf = *reinterpret_cast<const bf16x8*>(cb_ptr(0, hh_[e], fr_, ks * 4 + fg));
            s0[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, s0[e], 0, 0, 0);       // S^T: rows = keys fg * 4 + r of the tile, column = query fr_
            s1[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf, qf, s1[e], 0, 0, 0);       // row 0 (fg = 0, r = 0) = the CLS key
          }
        const bool own = (fg >> 1) == (fr_ >> 3);                   // key rows 4 fg .. + 3 belong to patch fg >> 1 of the tile, the query to patch fr_ >> 3
        float m[NU], l[NU], pc[NU];
#pragma unroll
        for (int e = 0; e < NU; ++e) {
          m[e] = fg == 0 ? s1[e][0] : -INFINITY;
#pragma unroll
          for (int r = 0; r < 4; ++r) { s0[e][r] = own ? s0[e][r] : -INFINITY; m[e] = fmaxf(m[e], s0[e][r]); }
        }
#pragma unroll
        for (int e = 0; e < NU; ++e) m[e] = fmaxf(m[e], __shfl_xor(m[e], 16, 64));
#pragma unroll
        for (int e = 0; e < NU; ++e) {                              // (finite without masks: every query sees its own 8 frames and the CLS key)
          m[e] = fmaxf(m[e], __shfl_xor(m[e], 32, 64)) * sc2;
          if (MASK && m[e] == -INFINITY) m[e] = 0.f;                // every key of the group masked: exp2(-inf) = 0 everywhere, l = 0 (a NaN row, as softmax over an all-masked row gives)
        }
#pragma unroll
        for (int e = 0; e < NU; ++e) {
          l[e] = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) { s0[e][r] = __builtin_amdgcn_exp2f(fmaf(s0[e][r], sc2, -m[e])); l[e] += s0[e][r]; }
          pc[e] = fg == 0 ? __builtin_amdgcn_exp2f(fmaf(s1[e][0], sc2, -m[e])) : 0.f;
          l[e] += pc[e];
        }
#pragma unroll
        for (int e = 0; e < NU; ++e) l[e] += __shfl_xor(l[e], 16, 64);
#pragma unroll
        for (int e = 0; e < NU; ++e) l[e] += __shfl_xor(l[e], 32, 64);
        f32x4 o[NU][4];
#pragma unroll
        for (int e = 0; e < NU; ++e) {
          union { bf16x8 v; uint32_t u[4]; } pa;                     // P^T fragment: k slots 8 fg .. + 3 = tile rows 4 fg .. + 3, + 4 .. + 7 = CLS block rows 4 fg .. + 3
          pa.u[0] = pack_bf2(s0[e][0], s0[e][1]); pa.u[1] = pack_bf2(s0[e][2], s0[e][3]);
          pa.u[2] = pack_bf2(pc[e], 0.f); pa.u[3] = 0u;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            union { bf16x8 v; qt2_s4 hh[2]; } vb;
            vb.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((qt2_lds_s4*)(k_lds[e] + QT2_ARR + qt2_arr_off(qt[e] * 16 + krow, dt * 2 + c1) + hb));
            vb.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((qt2_lds_s4*)(cb_ptr(2, hh_[e], krow, dt * 2 + c1) + hb));
            o[e][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb.v, pa.v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          }
        }
#ifndef QT2_OUT_LINES
#define QT2_OUT_LINES 1
#endif
#pragma unroll
        for (int e = 0; e < NU; ++e) {
          // query fr_ of the tile = (patch 2 qt + (fr_ >> 3), frame fr_ & 7); this lane holds dims dt * 16 + fg * 4 + r
          const int u = u0 + e * ustep, head = hp * 2 + (u >= 12 ? 1 : 0);
          const float linv = 1.0f / l[e];
          if (QT2_OUT_LINES) {
            // through the unit's own 16 Q rows (dead: only this unit read them): the accumulator layout would store 8 bytes per lane, 32 bytes apart - every 128-byte line
            // of `out` (one token, one head) in 16 pieces over 4 instructions; read back row-wise, 8 lanes x 16 bytes write one complete line
            char* qrows = const_cast<char*>(k_lds[e]) + 2 * QT2_ARR;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              uint2 w;
              w.x = pack_bf2(o[e][dt][0] * linv, o[e][dt][1] * linv);
              w.y = pack_bf2(o[e][dt][2] * linv, o[e][dt][3] * linv);
              *reinterpret_cast<uint2*>(qrows + qt2_arr_off(qt[e] * 16 + fr_, dt * 2 + (fg >> 1)) + (fg & 1) * 8) = w;
            }
          } else {
            bf16_t* orow = p.out + (orow0 + (int64_t)(fr_ & 7) * QT2_NP + 2 * qt[e] + (fr_ >> 3)) * p.ldo + head * 64 + fg * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              uint2 w;
              w.x = pack_bf2(o[e][dt][0] * linv, o[e][dt][1] * linv);
              w.y = pack_bf2(o[e][dt][2] * linv, o[e][dt][3] * linv);
              if (!(QT2_ABL & 32) || linv == 12345.678f) *reinterpret_cast<uint2*>(orow + dt * 16) = w;
            }
          }
        }
        if (QT2_OUT_LINES) {
#pragma unroll
          for (int e = 0; e < NU; ++e) {
            const int u = u0 + e * ustep, head = hp * 2 + (u >= 12 ? 1 : 0);
            const char* qrows = k_lds[e] + 2 * QT2_ARR;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int rr = (alane >> 3) + 8 * j, ch = alane & 7;    // tile row rr = (patch 2 qt + j, frame alane >> 3)
              const uint4 w = *reinterpret_cast<const uint4*>(qrows + qt2_arr_off(qt[e] * 16 + rr, ch));
              if (MX && p.out_q) {
                // MXFP8 output (the A operand of the MX projection that follows), byte for byte sf_quantize_mxfp8 of the bf16 row: the head's 64 dims are two scale blocks,
                // block b = this lane's quad (chunks 4 b .. 4 b + 3); quantised from the bf16-rounded values the bf16 launch would have stored
                const int64_t row = orow0 + (int64_t)(alane >> 3) * QT2_NP + 2 * qt[e] + j;
                float fv[8];
                fv[0] = __uint_as_float(w.x << 16); fv[1] = __uint_as_float(w.x & 0xffff0000u); fv[2] = __uint_as_float(w.y << 16); fv[3] = __uint_as_float(w.y & 0xffff0000u);
                fv[4] = __uint_as_float(w.z << 16); fv[5] = __uint_as_float(w.z & 0xffff0000u); fv[6] = __uint_as_float(w.w << 16); fv[7] = __uint_as_float(w.w & 0xffff0000u);
                float amax = fmaxf(fmaxf(fmaxf(fabsf(fv[0]), fabsf(fv[1])), fmaxf(fabsf(fv[2]), fabsf(fv[3]))), fmaxf(fmaxf(fabsf(fv[4]), fabsf(fv[5])), fmaxf(fabsf(fv[6]), fabsf(fv[7]))));
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64)); amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                const int be = sf_mx_exp(amax);
                const float inv = sf_mx_inv(be);
                uint2 d; d.x = sf_fp8x4(fv, inv); d.y = sf_fp8x4(fv + 4, inv);
                const int be_hi = __shfl_xor(be, 4, 64);              // the head's other block (lanes ch and ch ^ 4 of the row)
                if (!(QT2_ABL & 32) || l[e] == 12345.678f) {
                  *reinterpret_cast<uint2*>(p.out_q + row * p.ldq + head * 64 + ch * 8) = d;
                  if (ch == 0) *reinterpret_cast<uint16_t*>(p.out_s + (int64_t)(head >> 1) * p.splane + row * 4 + (head & 1) * 2) = (uint16_t)(be | (be_hi << 8));
                }
                continue;
              }
              bf16_t* dst = p.out + (orow0 + (int64_t)(alane >> 3) * QT2_NP + 2 * qt[e] + j) * p.ldo + head * 64 + ch * 8;
#ifndef QT2_OUT_NT
#define QT2_OUT_NT 1   // the attention output with the nt hint: written once, read once by the next launch
#endif
              if (!(QT2_ABL & 32) || l[e] == 12345.678f) {
                typedef unsigned int qt2_u4 __attribute__((ext_vector_type(4)));
                const qt2_u4 wv = {w.x, w.y, w.z, w.w};
                if (QT2_OUT_NT) __builtin_nontemporal_store(wv, reinterpret_cast<qt2_u4*>(dst)); else *reinterpret_cast<qt2_u4*>(dst) = wv;
              }
            }
          }
        }
      };
      // the CLS query of the sequence against NK key tiles of the item from tile kt0 on (+ the CLS key itself when WITH_CLS): unnormalised record `rec` of head h
      auto cls_unit = [&](auto NKc, auto WCc, const int h, const int kt0, const int rec) {
        constexpr int NK = decltype(NKc)::value;
        constexpr bool WITH_CLS = decltype(WCc)::value != 0;
        constexpr int NKT = NK + (WITH_CLS ? 1 : 0);
        const int head = hp * 2 + h;
        const char* k_lds = smem + h * 3 * QT2_ARR;
        const char* v_lds = k_lds + QT2_ARR;
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(cb_ptr(4, h, fr_, ks * 4 + fg));      // column 0 = the CLS query (columns 1..15: zero rows)
        f32x4 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (MASK) {
            if (kt < NK) { const float4 mk = *reinterpret_cast<const float4*>(smem + QT2_MASK_OFF + ((kt0 + kt) * 16 + fg * 4) * 4); s[kt] = f32x4{mk.x, mk.y, mk.z, mk.w}; }
            else s[kt][0] = reinterpret_cast<const float*>(smem + QT2_MASK_OFF)[QT2_ROWS];
          }
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 kf = kt < NK ? *reinterpret_cast<const bf16x8*>(k_lds + qt2_arr_off((kt0 + kt) * 16 + fr_, ks * 4 + fg))
                                      : *reinterpret_cast<const bf16x8*>(cb_ptr(0, h, fr_, ks * 4 + fg));
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
          }
        }
        if (WITH_CLS) {                                             // of the CLS block only row 0 is a key
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (fg != 0 || r != 0) s[NKT - 1][r] = -INFINITY;
        }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float msc = m * sc2;
        const float msafe = (MASK && m == -INFINITY) ? 0.f : msc;   // every key of the record masked: m = -inf, l = 0 goes to the combine (as sf_attention_cls_partial_masked writes it)
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sc2, -msafe)); s[kt][r] = e; l += e; }
        l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < (NKT + 1) / 2; ++kk) {
          union { bf16x8 v; uint32_t u[4]; } pa;
          pa.u[0] = pack_bf2(s[2 * kk][0], s[2 * kk][1]);
          pa.u[1] = pack_bf2(s[2 * kk][2], s[2 * kk][3]);
          if (2 * kk + 1 < NKT) {
            const int t1 = 2 * kk + 1 < NKT ? 2 * kk + 1 : 0;
            pa.u[2] = pack_bf2(s[t1][0], s[t1][1]);
            pa.u[3] = pack_bf2(s[t1][2], s[t1][3]);
          } else { pa.u[2] = 0u; pa.u[3] = 0u; }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            union { bf16x8 v; qt2_s4 hh[2]; } vb;
            const int ta = 2 * kk, tb_ = 2 * kk + 1;
            vb.hh[0] = ta < NK ? __builtin_amdgcn_ds_read_tr16_b64_v4i16((qt2_lds_s4*)(v_lds + qt2_arr_off((kt0 + ta) * 16 + krow, dt * 2 + c1) + hb))
                               : __builtin_amdgcn_ds_read_tr16_b64_v4i16((qt2_lds_s4*)(cb_ptr(2, h, krow, dt * 2 + c1) + hb));
            vb.hh[1] = qt2_s4{0, 0, 0, 0};
            if (tb_ < NKT) vb.hh[1] = tb_ < NK ? __builtin_amdgcn_ds_read_tr16_b64_v4i16((qt2_lds_s4*)(v_lds + qt2_arr_off((kt0 + tb_) * 16 + krow, dt * 2 + c1) + hb))
                                               : __builtin_amdgcn_ds_read_tr16_b64_v4i16((qt2_lds_s4*)(cb_ptr(2, h, krow, dt * 2 + c1) + hb));
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb.v, pa.v, o[dt], 0, 0, 0);
          }
        }
        if (fr_ == 0) {                                             // column 0 = the CLS query
          float* part = p.cls_part + ((seq * 12 + head) * QT2_NPART + rec) * 66;
          if (fg == 0) { part[0] = msc; part[1] = l; }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[2 + dt * 16 + fg * 4 + r] = o[dt][r];
        }
      };
      const int wr = (wave + (int)tcount) & 7;                      // rotate the roles over the SIMDs from item to item (wave-uniform)
      if (main_item) {
        // every wave: three of the 24 patch units (wr, wr + 8, wr + 16) and a quarter of one head's CLS-query work (key tiles 3 q .. 3 q + 2 -> record 4 tb + q)
        if (!(QT2_ABL & 8)) patch_units(qt2_ic<3>{}, wr, 8);
        if (!(QT2_ABL & 4)) cls_unit(qt2_ic<3>{}, qt2_ic<0>{}, wr & 1, 3 * (wr >> 1), tb * 4 + (wr >> 1));
      } else if (!(QT2_ABL & 16)) {
        if (wr < 2) cls_unit(qt2_ic<2>{}, qt2_ic<1>{}, wr, 0, 32);
        else if (wr < 6) { const int u = (wr - 2) & 1; patch_units(qt2_ic<1>{}, ((wr - 2) >> 1) * 12 + u, 0); }
      }
    }
    qt2_barrier();                                                  // every wave is out of the attention arrays: the next item's operands may land
    t += per_xcd_blocks;
    ++tcount;
    if (t >= t_end) break;
  }
}

__global__ __launch_bounds__(512, 2) void qkv_time2_attn_kernel(Qt2Args p) { qkv_time2_attn_body<false>(p); }
__global__ __launch_bounds__(512, 2) void qkv_time2_attn_masked_kernel(Qt2Args p) { qkv_time2_attn_body<true>(p); }
__global__ __launch_bounds__(512, 2) void qkv_time2_attn_mx_kernel(Qt2Args p) { qkv_time2_attn_body<false, true>(p); }

// X (n_seq * 1569, 768) bf16 = norm3(x), rows [CLS; frame-major patches] per sequence; W (2304, 768) bf16 = timeattn.qkv.weight, bias 2304 fp32 or NULL; side
// (n_seq * 33, 2304) bf16 = the same projection of [the CLS row; per frame f its tokens 192 .. 195] (row seq * 33, rows seq * 33 + 1 + 4 f + i) - the buffer layout of
// sf_qkv_space_attention; out (rows as X, 768) bf16: the PATCH rows are written (row 0 of every sequence comes from sf_attention_cls_combine on cls_partial
// [n_seq][12][33][66] fp32, n_part = 33).  out must not alias X.  Reference: vit_helper.py:97-150 with the '(b n) f d' groups of :343-344, 12 heads x 64, q scaled by `scale`.
static int qt2_launch(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                      uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, const uint8_t* key_keep, void* stream) {
  SF_CHECK_ARG(X && W && side && out && cls_partial, "sf_qkv_time_attention2: null pointer");
  SF_CHECK_ARG(n_tok == QT2_NP, "sf_qkv_time_attention2: built for 196 patches per frame (8 frames per sequence), got %d", n_tok);
  SF_CHECK_ARG((ldx % 64) == 0 && (ldw % 64) == 0 && (lds_ % 8) == 0 && (ldo % 8) == 0 && ldx >= QT2_D && ldw >= QT2_D && lds_ >= 3 * QT2_D && ldo >= QT2_D,
               "sf_qkv_time_attention2: bad row strides (ldx / ldw multiples of 64 elements)");
  SF_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)side % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0) &&
                   ((uintptr_t)cls_partial % 8) == 0, "sf_qkv_time_attention2: operands must be 16-byte aligned");
  SF_CHECK_ARG((const void*)X != (const void*)out, "sf_qkv_time_attention2: out must not alias X");
  if (n_seq <= 0) return 0;
  const int64_t seq_rows = 1 + 8 * (int64_t)QT2_NP;
  SF_CHECK_ARG(seq_rows * ldx * 2 < ((int64_t)1 << 32) && (int64_t)3 * QT2_D * ldw * 2 < ((int64_t)1 << 32) && (int64_t)33 * lds_ * 2 < ((int64_t)1 << 32),
               "sf_qkv_time_attention2: a sequence of X, W and a sequence's side rows must stay below 4 GiB (32-bit lane offsets)");
  SF_CHECK_ARG(n_seq * 9 * 6 < ((int64_t)1 << 31), "sf_qkv_time_attention2: too many tiles");
  if (int rc = sf_prepare_kernel(key_keep ? (const void*)qkv_time2_attn_masked_kernel : (const void*)qkv_time2_attn_kernel, QT2_LDS, "sf_qkv_time_attention2")) return rc;
  const int n_cu = sf_cu_count("sf_qkv_time_attention2");
  if (n_cu <= 0) return -1;
  Qt2Args a;
  a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.bias = bias; a.side = side; a.lds_ = lds_; a.out = out; a.ldo = ldo; a.cls_part = cls_partial;
  a.seq_rows = seq_rows; a.n_rt = (uint32_t)(n_seq * 9); a.scale = scale; a.key_keep = key_keep;
  static int env_hc = -1;
  if (env_hc < 0) { const char* e = getenv("SF_QT2_PAIR_CHUNK"); env_hc = e ? atoi(e) : 6; if (env_hc < 1 || 6 % env_hc) env_hc = 6; }
  a.pair_chunk = (uint32_t)env_hc;
  static int env_st = -1;
  if (env_st < 0) { const char* e = getenv("SF_QT2_STAGGER"); env_st = e ? atoi(e) : 0; if (env_st < 0) env_st = 0; }
  a.stagger = (uint32_t)env_st;
  int64_t blocks = (n_cu / 8) * 8;
  if (blocks < 8) blocks = 8;                                    // (a device / partition with fewer than 8 CUs: never an empty grid)
  const int64_t need = ((n_seq * 9 * 6 + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  if (key_keep) hipLaunchKernelGGL(qkv_time2_attn_masked_kernel, dim3((unsigned)blocks), dim3(512), QT2_LDS, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(qkv_time2_attn_kernel, dim3((unsigned)blocks), dim3(512), QT2_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
extern "C" int sf_qkv_time_attention2(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                                      uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, void* stream) {
  return qt2_launch(X, ldx, W, ldw, bias, side, lds_, out, ldo, cls_partial, n_seq, n_tok, scale, nullptr, stream);
}
// The same launch with TOKEN MASKS (Synchformer.forward(vis_mask=...), sync_model.py:72-80 -> vit_helper.py:107-141): key_keep holds one byte per row of X; a row with flag
// 0 is a masked KEY for the queries of its patch's 8-frame group and for the CLS query.  The flags become additive terms (0 / -inf) the score accumulators start from:
// an all-ones mask is bit-identical to sf_qkv_time_attention2.
extern "C" int sf_qkv_time_attention2_masked(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                                             uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, const uint8_t* key_keep, void* stream) {
  SF_CHECK_ARG(key_keep, "sf_qkv_time_attention2_masked: null key_keep (call sf_qkv_time_attention2 for an unmasked forward)");
  return qt2_launch(X, ldx, W, ldw, bias, side, lds_, out, ldo, cls_partial, n_seq, n_tok, scale, key_keep, stream);
}

// The same launch on MXFP8 operands (fp8 towers of the synchronizability fine-tune, round 5: the temporal half gets the schedule the spatial half got in round 4): X (rows,
// 768) e4m3 bytes with its stage-major scale planes sX (6 planes, ldsx bytes apart, one dword per row) - what sf_gemm_mx_res_ln768 / sf_layernorm768_mxfp8 write -, W (2304,
// 768) e4m3 + sW (6 planes of 2304 dwords); side (n_seq * 33, 2304) bf16 as in sf_qkv_time_attention2 (from sf_gemm_mxfp8 on gathered copies of the rows and of their scale
// dwords: sf_side_rows).  The attention runs on the bf16-rounded projection, exactly as sf_qkv_time_attention_mx does.  Output: EITHER out (bf16, patch rows) OR out_q / out_s
// (e4m3 bytes (rows, 768) + the scale planes [6][rows][4], splane bytes apart: byte for byte sf_quantize_mxfp8 of the bf16 output - the A operand of the MX projection that
// follows; buffers of their own, not X / sX).  cls_partial [n_seq][12][33][66] (merge with sf_attention_cls_combine(_mx), n_part = 33).  Replaces sf_gemm_mxfp8 (CLS rows) +
// sf_qkv_time_attention_mx(_q).
extern "C" int sf_qkv_time_attention2_mx(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW, int64_t ldsw,
                                         const float* bias, const uint16_t* side, int64_t lds_, uint16_t* out, int64_t ldo, uint8_t* out_q, int64_t ldq, uint8_t* out_s,
                                         int64_t splane, float* cls_partial, int64_t n_seq, int n_tok, float scale, void* stream) {
  SF_CHECK_ARG(X && sX && W && sW && side && cls_partial && ((out != nullptr) != (out_q != nullptr)), "sf_qkv_time_attention2_mx: null pointer (exactly one of out / out_q)");
  SF_CHECK_ARG(n_tok == QT2_NP, "sf_qkv_time_attention2_mx: built for 196 patches per frame (8 frames per sequence), got %d", n_tok);
  SF_CHECK_ARG((ldx % 128) == 0 && (ldw % 128) == 0 && ldx >= QT2_D && ldw >= QT2_D && (lds_ % 8) == 0 && lds_ >= 3 * QT2_D, "sf_qkv_time_attention2_mx: bad row strides (ldx / ldw multiples of 128 bytes)");
  SF_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)sX % 16) == 0 && ((uintptr_t)sW % 16) == 0 && ((uintptr_t)side % 16) == 0 &&
                   (!bias || ((uintptr_t)bias % 16) == 0) && ((uintptr_t)cls_partial % 8) == 0, "sf_qkv_time_attention2_mx: operands must be 16-byte aligned");
  if (out) SF_CHECK_ARG(((uintptr_t)out % 16) == 0 && (ldo % 8) == 0 && ldo >= QT2_D, "sf_qkv_time_attention2_mx: out must be a 16-byte aligned bf16 buffer");
  if (out_q) SF_CHECK_ARG(out_s && ((uintptr_t)out_q % 8) == 0 && ((uintptr_t)out_s % 2) == 0 && (ldq % 8) == 0 && ldq >= QT2_D && out_q != X && out_s != sX,
                          "sf_qkv_time_attention2_mx: out_q (8-byte aligned, ldq %% 8 == 0) / out_s must be buffers of their own");
  if (n_seq <= 0) return 0;
  const int64_t seq_rows = 1 + 8 * (int64_t)QT2_NP;
  if (out_q) SF_CHECK_ARG(splane >= n_seq * seq_rows * 4, "sf_qkv_time_attention2_mx: a scale plane holds 4 bytes per row");
  SF_CHECK_ARG((ldsx % 16) == 0 && (ldsw % 16) == 0 && ldsx >= n_seq * seq_rows * 4 && ldsw >= 3 * QT2_D * 4, "sf_qkv_time_attention2_mx: scale planes must hold one dword per row of X / W");
  SF_CHECK_ARG(seq_rows * ldx < ((int64_t)1 << 32) && (int64_t)3 * QT2_D * ldw < ((int64_t)1 << 32) && (int64_t)33 * lds_ * 2 < ((int64_t)1 << 32),
               "sf_qkv_time_attention2_mx: a sequence of X, W and a sequence's side rows must stay below 4 GiB (32-bit lane offsets)");
  SF_CHECK_ARG(n_seq * 9 * 6 < ((int64_t)1 << 31), "sf_qkv_time_attention2_mx: too many tiles");
  if (int rc = sf_prepare_kernel((const void*)qkv_time2_attn_mx_kernel, QT2_LDS, "sf_qkv_time_attention2_mx")) return rc;
  const int n_cu = sf_cu_count("sf_qkv_time_attention2_mx");
  if (n_cu <= 0) return -1;
  Qt2Args a;
  a.X = reinterpret_cast<const bf16_t*>(X); a.ldx = ldx; a.W = reinterpret_cast<const bf16_t*>(W); a.ldw = ldw; a.bias = bias; a.side = side; a.lds_ = lds_;
  a.out = out; a.ldo = ldo; a.cls_part = cls_partial; a.seq_rows = seq_rows; a.n_rt = (uint32_t)(n_seq * 9); a.scale = scale; a.pair_chunk = 6; a.stagger = 0;
  a.sX = sX; a.ldsx = ldsx; a.sW = sW; a.ldsw = ldsw; a.out_q = out_q; a.ldq = ldq; a.out_s = out_s; a.splane = splane;
  int64_t blocks = (n_cu / 8) * 8;
  if (blocks < 8) blocks = 8;
  const int64_t need = ((n_seq * 9 * 6 + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL(qkv_time2_attn_mx_kernel, dim3((unsigned)blocks), dim3(512), QT2_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
