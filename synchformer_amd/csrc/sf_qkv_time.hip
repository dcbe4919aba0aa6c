// Motionformer TIME attention fused into its qkv projection (gfx950): one launch computes, for every patch token, the temporal q | k | v of
// DividedSpaceTimeBlock (vit_helper.py:366 -> DividedAttention.forward, vit_helper.py:97-150 with the '(b n) f d' regrouping of :343-344)
// and the 8-frame attention over [CLS key; the patch's 8 frames] right in the GEMM's epilogue.  Un-fused (sf_gemm_bf16 -> sf_attention +
// sf_attention_cls) the 2304-wide projection output goes to HBM (1.6 GB at 224 segments) and is read back twice; fused, only the 768-wide
// attention output (0.54 GB) is written.
//
//   * GEMM tile = 256 token rows x 192 output features: the rows are 32 consecutive PATCHES x their 8 frames, GATHERED by the LDS-DMA source
//     addresses (token (seg, f, p) lives at row seg * seq_rows + 1 + f * n_groups + p; the LDS image is lane-linear, the global side is per lane),
//     the features are ONE head's q | k | v (W rows h*64.. of each of the three 768-row blocks).  So a workgroup ends its k-loop holding
//     everything the time attention of 32 patches x one head needs, and nothing else.
//   * 8 waves, wave w owns rows 32w .. 32w+31 (4 patches) x all 192 features: 6 accumulator blocks of v_mfma_f32_32x32x16_bf16 with the
//     operands SWAPPED (C^T = W X^T), so a lane ends with ONE token's features (4 consecutive per register group) - packed to bf16 and stored to
//     a wave-private LDS slab as rows [token][q 64 | k 64 | v 64] with 8-byte writes.
//   * epilogue, per wave (its 4 patches = 32 tokens, all of one sequence): the arithmetic of attn_tiny64_kernel (bf16 q, k, v - the accumulators are
//     rounded to bf16 first, as the un-fused projection writes them - v_dot2 scores, base-2 softmax in fp32, fp32 P V) in two lane layouts:
//       SCORE phase, lane (patch, frame, half head): 32 head dims of one query against [CLS key; the patch's 8 frames]; the normalised probabilities
//       go to the query's own (by then dead) q row of the slab; the CLS QUERY's score of the lane's token is reduced over the wave (DPP row
//       operations + two cross-row shuffles) into the softmax state of one `cls_partial` record per wave ([seq][head][n_groups / 4][66] fp32 as in
//       sf_attention's cls_partial mode, merged by sf_attention_cls_combine);
//       P V phase, lane (patch, query half, 8-dim slice): 8 head dims of 4 queries, so every V element is unpacked twice per patch instead of once
//       per query lane and a query row leaves as 16-byte stores; the CLS query's weighted values ride on the same unpacked V.
//     The CLS key / value / query of the sequence come from a small (n_seq, 2304) buffer the caller fills with the same projection of the CLS rows
//     (one 16-byte load per lane 0-23 at the top of the tile, parked in 384 B of wave-private LDS); the head's bias arrives as one LDS-DMA piece per tile.
//   * persistent, one workgroup per CU; inside an XCD's range of row tiles the heads go in chunks of 6 (head fastest), two 56 KiB operand slots;
//     the slabs (8 x 12.5 KiB) overlay slot 1, so only the next tile's FIRST k-tile is prefetched under the epilogue.  160 KiB of LDS exactly.
#include "sf_common.h"
#include <type_traits>
#include <stdlib.h>
#include "../../include/synchformer_hip.h"

#define QT_BM 256
#define QT_BN 192
#define QT_BK 64
#define QT_A_BYTES (QT_BM * 128)                 // 32 KiB
#define QT_B_BYTES (QT_BN * 128)                 // 24 KiB
#define QT_STAGE (QT_A_BYTES + QT_B_BYTES)       // 56 KiB
#define QT_SLAB_LD 400                           // bytes per token row of a slab: 384 + 16 (the 16 lanes of a ds_write_b64 group hit 16 distinct bank pairs)
#define QT_SLAB_BYTES (32 * QT_SLAB_LD)          // 12.5 KiB per wave
#define QT_BIAS_OFF (QT_STAGE + 8 * QT_SLAB_BYTES) // 1 KiB behind the slabs: the head's q | k | v bias (192 floats), one LDS-DMA piece per tile
#define QT_CLS_OFF (QT_BIAS_OFF + 1024)          // 8 x 384 B: the CLS q | k | v of every wave's sequence and head
#define QT_LDS (QT_CLS_OFF + 8 * 384)            // 160 KiB: slot 0 | slot 1 = the first 56 KiB of the slab area | bias | CLS
// MXFP8 variant (template parameter MX): the same 4 KiB behind the slabs hold bias (768 B) | the k-tile's scale dwords (8 waves x 224 B: this wave's 32 token
// rows | 24 of the head's 192 W rows) | FOUR shared CLS landing slots (tile parity x the at most two sequences a 32-patch tile touches; the bf16 variant's eight
// private slots do not leave the 1792 B the scales need)
#define QT_SC_OFF (QT_BIAS_OFF + 768)
#define QT_SC_WAVE 224
#define QT_CLSX_OFF (QT_SC_OFF + 8 * QT_SC_WAVE)  // + 4 x 384 B = QT_LDS exactly
#ifndef QT_OUT_NT
#define QT_OUT_NT 1   // attention output with the nt hint: read once, by the next launch (+0.4 % on the step, interleaved A/B)
#endif
#ifndef QT_ABL
#define QT_ABL 0   // measurement builds: 1 = one of the four patch passes per wave, 2 = no operand refills, 4 = no MFMAs
#endif
#define QT_D 768
#define QT_HEADS 12
// Round 3, template parameter PP: the quadrant-phased schedule of sf_gemm_pp.hip / sf_gemm_res_ln768 on this tile.  A k-tile (64 deep, 24 MFMAs per
// wave) is THREE phases - the q, k and v thirds of the head's 192 W rows, 2 blocks x 4 k-steps = 8 MFMAs each; the wave's own A fragments stay in
// registers for the three phases.  Parts of a stage: A (4 pieces per wave: the wave's own 32 rows) | W0 | W1 | W2 (1 piece per wave each); every part is
// refilled for k-tile kt+2 two phases after its last fragment read, 3 / 3 / 1 pieces per phase:
//     phase 0 of k-tile kt: W1, A2, A3 of kt+1 | phase 1: W2 of kt+1 | phase 2: W0, A0, A1 of kt+2
// behind counted waits (vmcnt 9 / 7 / 4: the youngest 4-9 pieces per wave stay in flight), and waves 4-7 run one barrier behind waves 0-3, so that on
// every SIMD one wave multiplies while the other reads fragments and issues loads.  Per tile: k-tile 0 arrives under the previous epilogue (stage 0),
// as before; the CLS q | k | v of the tile now arrive by an LDS-DMA piece issued in the previous epilogue (no compiler-visible load is left in the
// kernel, so hipcc inserts no vmcnt(0) of its own).  Same products in the same order as the round-2 loop: bit-identical outputs.
#ifndef QT_PP
#define QT_PP 1
#endif

struct QtArgs {
  const bf16_t* X; int64_t ldx;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  const bf16_t* qkv_cls; int64_t ldc;
  bf16_t* out; int64_t ldo;
  float* cls_part;
  int64_t n_seq, seq_rows;
  int n_groups;
  float scale;
  uint32_t tiles_m;
  uint32_t head_chunk;                                           // heads per sweep over an XCD's row tiles (divides 12)
  const uint8_t* key_keep;                                       // optional token keep flags, one byte per row of X (0 = masked key: -inf before the softmax)
  // MX: X / W are e4m3 BYTES (ldx / ldw in bytes) with stage-major E8M0 scale planes (one dword per row per 128 k, ldsx / ldsw bytes between planes)
  const uint8_t* sX; int64_t ldsx;
  const uint8_t* sW; int64_t ldsw;
  // MX, optional: the attention output as MXFP8 (e4m3 bytes, row stride ldq, + E8M0 bytes in the scale planes [6][rows][4], splane bytes apart) instead of bf16 `out`
  uint8_t* out_q = nullptr; int64_t ldq = 0; uint8_t* out_s = nullptr; int64_t splane = 0;
};

typedef __attribute__((ext_vector_type(2))) __bf16 qt_bf2;
typedef __attribute__((ext_vector_type(2))) unsigned int qt_u32x2;

__device__ __forceinline__ void qt_dma1(uint32_t voff, const void* sbase, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}
// one dword per lane (64-bit lane addresses) -> 4 consecutive bytes per lane from the wave-uniform LDS address lds; inactive lanes write nothing
__device__ __forceinline__ void qt_dma_dword_addr(const void* gaddr, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gaddr), "s"(lds) : "memory");
}
typedef __attribute__((ext_vector_type(8))) int qt_i32x8;
template <int N>
__device__ __forceinline__ void qt_wait_vmcnt() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void qt_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint32_t qt_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ void qt_wait_vmcnt0_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((0 & 0xF) | (0x7 << 4) | (0xF << 8) | (0 << 14));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ float qt_dot8(const uint4& a, const uint4& b) {
  float d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(qt_bf2, a.x), __builtin_bit_cast(qt_bf2, b.x), 0.f, false);
  d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(qt_bf2, a.y), __builtin_bit_cast(qt_bf2, b.y), d, false);
  d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(qt_bf2, a.z), __builtin_bit_cast(qt_bf2, b.z), d, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(qt_bf2, a.w), __builtin_bit_cast(qt_bf2, b.w), d, false);
}
__device__ __forceinline__ void qt_axpy8(sf_f32x2_t (&o)[4], float e, const uint4& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  const sf_f32x2_t e2 = {e, e};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const sf_f32x2_t vf = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
    o[i] = e2 * vf + o[i];
  }
}
// sum / max over the eight lanes sharing lane >> 3 (DPP: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror)
__device__ __forceinline__ float qt_sum8(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  return v;
}

// sum over the lane pair (lane ^ 1); max / sum over the 16 lanes of a DPP row (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror)
__device__ __forceinline__ float qt_sum2(float v) { return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float qt_max_row16(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false)));
  return v;
}
__device__ __forceinline__ float qt_sum_row16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  return v;
}

template <bool PP, bool MX = false>
__global__ __launch_bounds__(512, 2) void qkv_time_attn_kernel(QtArgs p) {
  static_assert(PP || !MX, "the MXFP8 variant exists on the quadrant-phased schedule only");
  constexpr int ESZ = MX ? 1 : 2;                                 // bytes per operand element
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // persistent schedule: block b sits on XCD b % 8; every XCD owns a contiguous range of row tiles and walks (row tile, head) with the head fastest
  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t mp8 = (p.tiles_m + 7u) >> 3;
  const uint32_t mp0 = min(xcd * mp8, p.tiles_m), mp1 = min(mp0 + mp8, p.tiles_m);
  const uint32_t t_end = (mp1 - mp0) * QT_HEADS;
  uint32_t t = li;
  if (t >= t_end) return;

  const uint32_t n_patches = (uint32_t)(p.n_seq * p.n_groups);     // < 2^31 (host check): 32-bit divisions below
  const int sw = (l31 >> 1) & 7;
  int frag_off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) frag_off[kk] = l31 * 128 + (((kk * 2 + hi) ^ sw) << 4);
  const int a_base = wave * 32 * 128;

  // LDS-DMA sources.  A: piece i of this wave = patch 4 * wave + i of the tile, lane (f = lane >> 3, chunk = lane & 7) -> LDS row 32 wave + 8 i + f.
  // B: piece j = rows 24 wave + 8 j + (lane >> 3) of the head's 192 W rows (q | k | v blocks 768 rows apart; the head offset rides in the SGPR base).
  uint32_t voff_a[4], voff_b[3];
  const uint8_t* sc_addr = nullptr;                               // MX: see set_tile
  const uint32_t sc_step = MX ? (uint32_t)(lane < 32 ? p.ldsx : p.ldsw) : 0u;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int br = PP ? j * 64 + wave * 8 + (lane >> 3) : wave * 24 + j * 8 + (lane >> 3);   // PP: piece j = this wave's 8 rows of W third j
    const int gch = (lane & 7) ^ ((br >> 1) & 7);
    voff_b[j] = (uint32_t)(((int64_t)(br >> 6) * QT_D + (br & 63)) * p.ldw * ESZ + gch * 16);
  }
  // Inside an XCD's row-tile range the heads go in CHUNKS of `hc`: all row tiles x the chunk's heads (head fastest), then the next chunk.  With every
  // head in flight at once (hc = 12) the XCD's L2 has to hold all of W (3.5 MB of its 4 MB) next to the streamed A tiles and thrashes: 2.4 GB
  // fetched per launch at 224 segments for 0.54 GB of A (PMC, profiles/r02_bench_summary.md); a chunk of 6 heads keeps 1.8 MB of W resident
  // and re-reads A once more.
  const uint32_t hc = p.head_chunk, chunk_tiles = (mp1 - mp0) * hc;
  const char* wbase; uint32_t tm; int head;
  auto set_tile = [&](uint32_t tt) {
    const uint32_t c = tt / chunk_tiles, r = tt - c * chunk_tiles;
    tm = mp0 + r / hc; head = (int)(c * hc + r % hc);
    wbase = reinterpret_cast<const char*>(p.W) + (int64_t)head * 64 * p.ldw * ESZ;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t g = tm * 32u + wave * 4 + i;                        // global patch index (seq, patch); the ragged last tile re-reads the last patch
      if (g > n_patches - 1) g = n_patches - 1;
      const uint32_t seq = g / (uint32_t)p.n_groups;
      const int pp = (int)(g - seq * (uint32_t)p.n_groups);
      const int f = lane >> 3;
      const int64_t row = (int64_t)seq * p.seq_rows + 1 + (int64_t)f * p.n_groups + pp;
      const int r = wave * 32 + i * 8 + f;
      voff_a[i] = (uint32_t)(row * p.ldx * ESZ + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
    }
    if (MX) {                                                      // this lane's dword of the k-tile's scale piece (plane 0; + kt * step per k-tile)
      if (lane < 32) {                                             // the wave's token row lane: LDS row 8 i + f  <->  patch i = lane >> 3, frame f = lane & 7
        uint32_t g = tm * 32u + wave * 4 + (lane >> 3);
        if (g > n_patches - 1) g = n_patches - 1;
        const uint32_t seq = g / (uint32_t)p.n_groups;
        const int pp = (int)(g - seq * (uint32_t)p.n_groups);
        sc_addr = p.sX + ((int64_t)seq * p.seq_rows + 1 + (int64_t)(lane & 7) * p.n_groups + pp) * 4;
      } else {                                                     // lanes 32-55: rows 4 wave .. + 3 of the head's six 32-row W blocks (q | k | v thirds x 2)
        const int q = (lane - 32) < 24 ? lane - 32 : 23, grp = q >> 2;
        sc_addr = p.sW + ((int64_t)(grp >> 1) * QT_D + head * 64 + (grp & 1) * 32 + wave * 4 + (q & 3)) * 4;
      }
    }
  };
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(qt_lds_addr(smem));
  const uint32_t lds_a_w = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096), lds_b_w = __builtin_amdgcn_readfirstlane(lds0 + QT_A_BYTES + wave * 3072);
  auto piece = [&](int pc, int slot, int kt) {                     // pc = 0 .. 3: A pieces, 4 .. 6: B pieces of k-tile kt
    if (pc < 4) qt_dma1(voff_a[pc], reinterpret_cast<const char*>(p.X) + kt * 128, lds_a_w + slot * QT_STAGE + pc * 1024);
    else if (PP) qt_dma1(voff_b[pc - 4], wbase + kt * 128, lds0 + QT_A_BYTES + slot * QT_STAGE + (pc - 4) * 8192 + wave * 1024);
    else qt_dma1(voff_b[pc - 4], wbase + kt * 128, lds_b_w + slot * QT_STAGE + (pc - 4) * 1024);
  };
  // PP: the CLS q | k | v (3 x 128 B) of this wave's sequence and head as one masked LDS-DMA piece (lanes 0-23, 16 bytes each) into the wave's 384 B
  const uint32_t lds_cls = __builtin_amdgcn_readfirstlane(lds0 + QT_CLS_OFF + wave * 384);
  // MX: the landing slot is shared by the waves of one sequence (every one of them writes the same 384 bytes) and alternates with the tile parity, so that a
  // fast wave's piece for the next tile cannot land on what a slow wave has not read yet
  auto cls_slot_mx = [&](uint32_t tile_m, int par) -> uint32_t {
    uint32_t g0 = tile_m * 32u + wave * 4, g00 = tile_m * 32u;
    if (g0 > n_patches - 1) g0 = n_patches - 1;
    if (g00 > n_patches - 1) g00 = n_patches - 1;
    return (uint32_t)(QT_CLSX_OFF + (par * 2 + (int)(g0 / (uint32_t)p.n_groups - g00 / (uint32_t)p.n_groups)) * 384);
  };
  auto cls_piece = [&](int par) {
    uint32_t g0 = tm * 32u + wave * 4;
    if (g0 > n_patches - 1) g0 = n_patches - 1;
    const char* cls = reinterpret_cast<const char*>(p.qkv_cls + (int64_t)(g0 / (uint32_t)p.n_groups) * p.ldc + head * 64);
    const uint32_t dst = MX ? __builtin_amdgcn_readfirstlane(lds0 + cls_slot_mx(tm, par)) : lds_cls;
    if (lane < 24) qt_dma1((uint32_t)(((lane >> 3) * QT_D + (lane & 7) * 8) * 2), cls, dst);
  };
  // MX: the scale dwords of k-tile kt - this wave's 32 token rows (lanes 0-31) and its 24 of the head's W rows (lanes 32-55) - as ONE piece of 56 dwords
  const uint32_t lds_sc = __builtin_amdgcn_readfirstlane(lds0 + QT_SC_OFF + wave * QT_SC_WAVE);
  auto scale_piece = [&](int kt) {
    if (lane < 56) qt_dma_dword_addr(sc_addr + (uint64_t)(uint32_t)kt * sc_step, lds_sc);
  };
  // the head's bias: lanes 0-47 of wave 0 fetch 16 bytes each of the q | k | v thirds (768 floats apart), the other lanes re-read lane 0's
  const uint32_t voff_bias = lane < 48 ? (uint32_t)(((lane >> 4) * QT_D + (lane & 15) * 4) * 4) : 0u;
  const uint32_t lds_bias = lds0 + QT_BIAS_OFF;
  auto bias_piece = [&]() {                                        // wave-uniform branch; after set_tile (uses `head`)
    if (wave == 0 && p.bias && (!MX || lane < 48)) qt_dma1(voff_bias, reinterpret_cast<const char*>(p.bias) + head * 256, lds_bias);   // MX: the 256 B behind the 768 are the scale area
  };
  set_tile(t);
#pragma unroll
  for (int pc = 0; pc < 7; ++pc) piece(pc, 0, 0);
  bias_piece();
  if (MX) scale_piece(0);
  int tpar = 0;                                                    // MX: parity of the current tile (CLS landing slots)
  if (PP) { cls_piece(0); qt_wait_vmcnt<0>(); }

  constexpr int nk = MX ? QT_D / 128 : QT_D / QT_BK;                // 12 k-tiles of 64 bf16 / 6 of 128 fp8: 128 bytes per row either way
  const float sc = p.scale * 1.44269504088896f;                    // softmax in base 2

  for (;;) {
    // accumulators start at the bias (block j = (q, k, v)[j >> 1], features (j & 1) * 32 + 8 g + 4 hi + i of head `head`), read from LDS behind the
    // first k-step's barrier: the 192 floats arrive as an eighth LDS-DMA piece of wave 0 with the tile's first stage - 24 float4 global loads per
    // lane and tile were 192 vector-memory instructions per workgroup and tile on a path that is the bottleneck of the k-loop
    f32x16 acc[6];

    // the CLS q | k | v (3 x 128 B) of this wave's sequence and head: ONE load instruction per wave (lanes 0-23, 16 bytes each) instead of nine
    // sliced ones, parked in 384 B of wave-private LDS behind the first k-step's wait and read back in the epilogue's two lane layouts
    uint4 cls_raw = make_uint4(0u, 0u, 0u, 0u);
    if (!PP) {
      uint32_t g0 = tm * 32u + wave * 4;
      if (g0 > n_patches - 1) g0 = n_patches - 1;
      const bf16_t* cls = p.qkv_cls + (int64_t)(g0 / (uint32_t)p.n_groups) * p.ldc + head * 64;
      if (lane < 24) cls_raw = *reinterpret_cast<const uint4*>(cls + (lane >> 3) * QT_D + (lane & 7) * 8);
    }

    auto kstep = [&](int kt, auto refill_tag, auto first_tag) {
      constexpr bool REFILL = decltype(refill_tag)::value;
      constexpr bool FIRST = decltype(first_tag)::value;
      qt_wait_vmcnt0_barrier();                                    // k-tile kt landed everywhere; the other slot is free (kt = 0: every wave is out of the slabs)
      if (FIRST) {
        if (lane < 24) *reinterpret_cast<uint4*>(smem + QT_CLS_OFF + wave * 384 + lane * 16) = cls_raw;     // the explicit wait above covered the load
        const float* bs = reinterpret_cast<const float*>(smem + QT_BIAS_OFF);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(bs + (j >> 1) * 64 + (j & 1) * 32 + g * 8 + hi * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[j][g * 4 + 0] = b4.x; acc[j][g * 4 + 1] = b4.y; acc[j][g * 4 + 2] = b4.z; acc[j][g * 4 + 3] = b4.w;
          }
      }
      const char* st = smem + (kt & 1) * QT_STAGE;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(st + a_base + frag_off[kk]);
        bf16x8 wf[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(st + QT_A_BYTES + j * 32 * 128 + frag_off[kk]);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (!(QT_ABL & 4)) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf, acc[j], 0, 0, 0);
          else if (j == 0) acc[0][0] += (float)xf[0] + (float)wf[0][0] + (float)wf[5][0];
          if (REFILL && !(QT_ABL & 2) && (j == 2 || j == 5) && kk * 2 + (j == 5) < 7) {   // one LDS-DMA piece of k-tile kt + 1 behind every third MFMA
            __builtin_amdgcn_sched_barrier(0);
            piece(kk * 2 + (j == 5), (kt + 1) & 1, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    if constexpr (PP) {
      const int wg = wave >> 2;                                     // waves 4-7 run one barrier behind waves 0-3
      int fo[4];                                                   // fragment offsets, re-derived here (not kept live across the epilogue)
      int fsx = 0, fsw = 0, shi = 0;                               // MX: addresses of this lane's scale dwords (its token row; row l31 of W block 0), 8 * (lane >> 5)
      {
        int ptid = threadIdx.x;
        asm volatile("" : "+v"(ptid));
        const int pl31 = ptid & 31, phi = (ptid & 63) >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fo[kk] = pl31 * 128 + (((kk * 2 + phi) ^ ((pl31 >> 1) & 7)) << 4);
        if (MX) {
          fsx = QT_SC_OFF + wave * QT_SC_WAVE + pl31 * 4;
          fsw = QT_SC_OFF + (pl31 >> 2) * QT_SC_WAVE + 128 + (pl31 & 3) * 4;      // + (2 C + jj) * 16
          shi = phi * 8;
        }
      }
      uint32_t sxd = 0, swd[6] = {0, 0, 0, 0, 0, 0};               // MX: the k-tile's scale dwords, shifted so that byte 2 kk is this half-wave's block of MFMA kk
      int sxv[2] = {0, 0}, swv[2][2] = {{0, 0}, {0, 0}};
      // every wave is out of its slab (they overlay stage 1) and has waited for its pieces of k-tile 0 (vmcnt 0 in front of the epilogue's stores)
      qt_barrier();
      if (!(QT_ABL & 2)) { piece(4, 1, 1); piece(0, 1, 1); piece(1, 1, 1); }          // W0, A0, A1 of k-tile 1
      {
        const float* bs = reinterpret_cast<const float*>(smem + QT_BIAS_OFF);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(bs + (j >> 1) * 64 + (j & 1) * 32 + g * 8 + hi * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[j][g * 4 + 0] = b4.x; acc[j][g * 4 + 1] = b4.y; acc[j][g * 4 + 2] = b4.z; acc[j][g * 4 + 3] = b4.w;
          }
      }
      bf16x8 xf[4], wf[2][4];
      auto read_w = [&](const char* st, auto Cc) {
        constexpr int C = decltype(Cc)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) wf[jj][kk] = *reinterpret_cast<const bf16x8*>(st + QT_A_BYTES + (2 * C + jj) * 4096 + fo[kk]);
        if (MX) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) swv[jj][k2] = (int)((swd[2 * C + jj] >> (k2 * 16)) & 0xffu);
        }
      };
      auto mma = [&](auto Cc) {
        constexpr int C = decltype(Cc)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MX) {
          // a 128-byte LDS row = 128 fp8 k: the bf16 loop's fragments 2 k2 and 2 k2 + 1 are the 2 x 16 bytes a lane supplies to ONE 64-deep scaled MFMA
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            union { bf16x8 h[2]; qt_i32x8 v; } ux;
            ux.h[0] = xf[2 * k2]; ux.h[1] = xf[2 * k2 + 1];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              union { bf16x8 h[2]; qt_i32x8 v; } uw;
              uw.h[0] = wf[jj][2 * k2]; uw.h[1] = wf[jj][2 * k2 + 1];
              acc[2 * C + jj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(uw.v, ux.v, acc[2 * C + jj], 0 /* e4m3 */, 0 /* e4m3 */, 0, swv[jj][k2], 0, sxv[k2]);
            }
          }
          asm volatile("" : "+v"(acc[2 * C]), "+v"(acc[2 * C + 1]));   // pins the (pure) MFMAs inside their matrix segment
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            if (!(QT_ABL & 4)) acc[2 * C + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[jj][kk], xf[kk], acc[2 * C + jj], 0, 0, 0);
            else if (kk == 0 && jj == 0) asm volatile("" :: "v"(xf[0]), "v"(xf[3]), "v"(wf[0][0]), "v"(wf[1][3]));
          }
        }
        __builtin_amdgcn_s_setprio(0);
      };
      // one k-tile in stage S; more1 / more2: k-tiles kt+1 / kt+2 exist; first: k-tile 0 (its W thirds 1 and 2 have landed already)
      auto ktile = [&](auto Sc, int kt, bool more1, bool more2, bool first) {
        constexpr int S = decltype(Sc)::value;
        const char* st = smem + S * QT_STAGE;
        const bool ld1 = more1 && !(QT_ABL & 2), ld2 = more2 && !(QT_ABL & 2);
        // ---- phase 0: q third ----
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[kk] = *reinterpret_cast<const bf16x8*>(st + a_base + fo[kk]);
        if (MX) {                                                  // ALL scale dwords of the k-tile now: the (single) scale area is refilled in phase 1
          sxd = *reinterpret_cast<const uint32_t*>(smem + fsx) >> shi;
#pragma unroll
          for (int b = 0; b < 6; ++b) swd[b] = *reinterpret_cast<const uint32_t*>(smem + fsw + b * 16) >> shi;
          sxv[0] = (int)(sxd & 0xffu); sxv[1] = (int)((sxd >> 16) & 0xffu);
        }
        read_w(st, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        if (ld1) { piece(5, S ^ 1, kt + 1); piece(2, S ^ 1, kt + 1); piece(3, S ^ 1, kt + 1); }   // W1, A2, A3 of k-tile kt+1
        if (!first) { if (QT_ABL & 2) qt_wait_vmcnt<0>(); else if (more1) qt_wait_vmcnt<9>(); else qt_wait_vmcnt<3>(); }   // W third 1 of this k-tile has landed
        qt_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        qt_barrier();
        // ---- phase 1: k third ----
        read_w(st, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (MX && ld1) scale_piece(kt + 1);                                                        // (every wave has read k-tile kt's dwords in phase 0)
        if (ld1) piece(6, S ^ 1, kt + 1);                                                          // W2 of k-tile kt+1
        if (!first) { if (!more1 || (QT_ABL & 2)) qt_wait_vmcnt<0>(); else if (MX) qt_wait_vmcnt<8>(); else qt_wait_vmcnt<7>(); }   // W third 2 has landed
        qt_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        qt_barrier();
        // ---- phase 2: v third ----
        read_w(st, std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        if (ld2) { piece(4, S, kt + 2); piece(0, S, kt + 2); piece(1, S, kt + 2); }                // W0, A0, A1 of k-tile kt+2
        if (more1) { if (QT_ABL & 2) qt_wait_vmcnt<0>(); else if (more2) qt_wait_vmcnt<4>(); else qt_wait_vmcnt<1>(); }   // A | W third 0 of k-tile kt+1 have landed
        qt_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        qt_barrier();
      };
      if (wg == 1) qt_barrier();
      for (int kt = 0; kt < nk; kt += 2) {
        const bool more = kt + 2 < nk;
        ktile(std::integral_constant<int, 0>{}, kt, true, more, kt == 0);
        ktile(std::integral_constant<int, 1>{}, kt + 1, more, more, false);
      }
      if (wg == 0) qt_barrier();                                   // re-align; every wave is done with both stages
    } else {
    kstep(0, std::true_type{}, std::true_type{});
    for (int kt = 1; kt + 1 < nk; ++kt) kstep(kt, std::true_type{}, std::false_type{});
    kstep(nk - 1, std::false_type{}, std::false_type{});
    // every wave is done with both slots: slot 0 takes the next tile's first k-tile, the slabs (over slot 1) take this tile's q | k | v
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    }
    const uint32_t etm = tm; const int ehead = head;
    const uint32_t tnext = t + per_xcd_blocks;
    const bool more = tnext < t_end;
    if (more) {
      set_tile(tnext);
#pragma unroll
      for (int pc = 0; pc < 7; ++pc) piece(pc, 0, 0);
      bias_piece();
      if (MX) scale_piece(0);
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------------------
    // lane-derived values are re-derived from an opaque copy of the thread id: computed up front hipcc keeps them live across the k-loop
    int etid = threadIdx.x;
    asm volatile("" : "+v"(etid));
    const int elane = etid & 63, ewave = etid >> 6, el31 = elane & 31, ehi = elane >> 5;
    char* slab = smem + QT_STAGE + ewave * QT_SLAB_BYTES;
    // (1) accumulators -> bf16 -> slab row el31: [q 64 | k 64 | v 64], 4 consecutive features per 8-byte write
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        qt_u32x2 w;
        w.x = pack_bf2(acc[j][g * 4 + 0], acc[j][g * 4 + 1]);
        w.y = pack_bf2(acc[j][g * 4 + 2], acc[j][g * 4 + 3]);
        *reinterpret_cast<qt_u32x2*>(slab + el31 * QT_SLAB_LD + (j * 32 + g * 8 + ehi * 4) * 2) = w;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // The wave's four patches belong to ONE sequence (n_groups % 4 == 0); np of them exist (ragged last tile).
    const uint32_t g0 = etm * 32u + ewave * 4;
    const int np = g0 >= n_patches ? 0 : (n_patches - g0 < 4u ? (int)(n_patches - g0) : 4);
    const char* clsp = smem + (MX ? cls_slot_mx(etm, tpar) : (uint32_t)(QT_CLS_OFF + ewave * 384));
    uint4 qcA[4], kcA[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      qcA[c] = *reinterpret_cast<const uint4*>(clsp + (elane & 1) * 64 + c * 16);
      kcA[c] = *reinterpret_cast<const uint4*>(clsp + 128 + (elane & 1) * 64 + c * 16);
    }
    const uint4 vcB = *reinterpret_cast<const uint4*>(clsp + 256 + (elane & 7) * 16);
    if (PP && more) {                                              // this tile's CLS slices are in registers: the next tile's may land
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      cls_piece(tpar ^ 1);
    }
    if (np == 0 && PP) qt_wait_vmcnt<0>();                         // (waves with patches wait in front of their stores, below)
    if (np > 0) {                                                  // wave-uniform
      const int64_t seq = g0 / (uint32_t)p.n_groups;
      const int pp0 = (int)(g0 - (uint32_t)seq * (uint32_t)p.n_groups);
      const int pi = elane >> 4;                                    // patch of this lane in both phases
      const bool live = pi < np;
      // (2) SCORE phase: lane (patch pi, frame qi, half sub): 32 head dims of one query.  Scores against [CLS key; the patch's 8 frames], softmax,
      // the normalised probabilities go to the (now dead) q area of the query's own slab row; the CLS QUERY's score against the lane's own token
      // is reduced over the wave's 32 tokens (softmax state of sf_attention's cls_partial records, one record per wave).
      {
        const int qi = (elane >> 1) & 7, sub = elane & 1, tr = elane >> 1;
        // token masks (Synchformer.forward(vis_mask=): vit_helper.py:34-42, 107-141): bit 0 = the CLS key, bit j + 1 = frame j of the lane's patch
        uint32_t kb = 0x1ffu;
        if (p.key_keep) {                                           // wave-uniform
          const uint8_t* kk = p.key_keep + seq * p.seq_rows;
          const int pidx = pp0 + (pi < np ? pi : np - 1);
          kb = kk[0] ? 1u : 0u;
#pragma unroll
          for (int j = 0; j < 8; ++j) if (kk[1 + j * p.n_groups + pidx]) kb |= 2u << j;
        }
        const char* rowp = slab + tr * QT_SLAB_LD + sub * 64;
        uint4 q4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) q4[c] = *reinterpret_cast<const uint4*>(rowp + c * 16);
        float s[9];
        {
          float d = qt_dot8(q4[0], kcA[0]) + qt_dot8(q4[1], kcA[1]) + qt_dot8(q4[2], kcA[2]) + qt_dot8(q4[3], kcA[3]);
          s[0] = (kb & 1u) ? qt_sum2(d) * sc : -INFINITY;
        }
        float m = s[0];
        float cs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const char* kp = slab + (pi * 8 + j) * QT_SLAB_LD + 128 + sub * 64;
          uint4 k4[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) k4[c] = *reinterpret_cast<const uint4*>(kp + c * 16);
          const float d = qt_dot8(q4[0], k4[0]) + qt_dot8(q4[1], k4[1]) + qt_dot8(q4[2], k4[2]) + qt_dot8(q4[3], k4[3]);
          s[j + 1] = ((kb >> (j + 1)) & 1u) ? qt_sum2(d) * sc : -INFINITY;
          m = fmaxf(m, s[j + 1]);
        }
        {                                                           // the CLS query against the lane's OWN token
          const char* kp = slab + tr * QT_SLAB_LD + 128 + sub * 64;
          uint4 k4[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) k4[c] = *reinterpret_cast<const uint4*>(kp + c * 16);
          cs = qt_sum2(qt_dot8(qcA[0], k4[0]) + qt_dot8(qcA[1], k4[1]) + qt_dot8(qcA[2], k4[2]) + qt_dot8(qcA[3], k4[3])) * sc;
        }
        if (m == -INFINITY) m = 0.f;                                // every key of the query masked (the CLS key included): all-zero weights, not NaN
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) { s[j] = __builtin_amdgcn_exp2f(s[j] - m); l += s[j]; }
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        // CLS-query softmax over the wave's tokens: frame lanes come in pairs (sub) holding the same score
        float c0 = qt_sum2(qt_dot8(qcA[0], kcA[0]) + qt_dot8(qcA[1], kcA[1]) + qt_dot8(qcA[2], kcA[2]) + qt_dot8(qcA[3], kcA[3])) * sc;
        if (!(kb & 1u)) c0 = -INFINITY;
        if (!live || !((kb >> (qi + 1)) & 1u)) cs = -INFINITY;
        float M = qt_max_row16(cs);
        M = fmaxf(M, __shfl_xor(M, 16, 64)); M = fmaxf(M, __shfl_xor(M, 32, 64));
        if (pp0 == 0) M = fmaxf(M, c0);                             // the CLS key itself is counted by the wave that holds patch 0
        const float Ms = M == -INFINITY ? 0.f : M;                  // every token of the wave masked: an empty record (M = -inf, L = 0), not NaN
        const float ec = __builtin_amdgcn_exp2f(cs - Ms);
        float L = qt_sum_row16(sub == 0 ? ec : 0.f);
        L += __shfl_xor(L, 16, 64); L += __shfl_xor(L, 32, 64);
        const float e0 = pp0 == 0 ? __builtin_amdgcn_exp2f(c0 - Ms) : 0.f;
        L += e0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // every q read above is done
        if (sub == 0) {
          float* prow = reinterpret_cast<float*>(slab + tr * QT_SLAB_LD);
          *reinterpret_cast<float4*>(prow) = make_float4(s[0] * inv, s[1] * inv, s[2] * inv, s[3] * inv);
          *reinterpret_cast<float4*>(prow + 4) = make_float4(s[4] * inv, s[5] * inv, s[6] * inv, s[7] * inv);
          prow[8] = s[8] * inv;
          reinterpret_cast<float*>(slab + (pi * 8) * QT_SLAB_LD + 64)[qi] = ec;      // the patch's eight CLS-query weights, contiguous in its first row
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (3) P V phase: lane (patch pi, query half qh, slice ds): 8 head dims of 4 queries of the patch - every V element is unpacked twice per
        // patch instead of once per query lane, and a query row leaves as eight 16-byte stores (store instructions, not bytes, are what the
        // vector-memory path of a CU charges for)
        // PP: the next tile's k-tile 0, bias and CLS pieces (issued at the top of this epilogue) have landed by now; waiting for them HERE, in front of
        // this epilogue's stores, keeps the stores out of every counted wait of the next k-loop
        if (PP) qt_wait_vmcnt<0>();
        const int qh = (elane >> 3) & 1, ds = elane & 7;
        sf_f32x2_t v[9][4];
        {
          const uint32_t w[4] = {vcB.x, vcB.y, vcB.z, vcB.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) v[0][i] = sf_f32x2_t{__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint4 vv = *reinterpret_cast<const uint4*>(slab + (pi * 8 + j) * QT_SLAB_LD + 256 + ds * 16);
          const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) v[j + 1][i] = sf_f32x2_t{__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
        }
        const int64_t orow0 = seq * p.seq_rows + 1 + pp0 + pi;
        bf16_t* optr = p.out + orow0 * p.ldo + ehead * 64 + ds * 8;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float* prow = reinterpret_cast<const float*>(slab + (pi * 8 + qh * 4 + qq) * QT_SLAB_LD);
          const float4 pa = *reinterpret_cast<const float4*>(prow), pb = *reinterpret_cast<const float4*>(prow + 4);
          const float pc = prow[8];
          const float pj[9] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w, pc};
          sf_f32x2_t o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = sf_f32x2_t{0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 9; ++j) {
            const sf_f32x2_t pj2 = {pj[j], pj[j]};
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = pj2 * v[j][i] + o[i];
          }
          uint4 w;
          w.x = pack_bf2(o[0].x, o[0].y); w.y = pack_bf2(o[1].x, o[1].y); w.z = pack_bf2(o[2].x, o[2].y); w.w = pack_bf2(o[3].x, o[3].y);
          if (MX && p.out_q) {
            // MXFP8 output (the operand of the MX projection): a 32-dim scale block is the 8 values of each of the four lanes ds & ~3 .. + 3 (one DPP quad);
            // quantised from the bf16-rounded values, as sf_quantize_mxfp8 would from the bf16 output
            const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
            float f[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(wv[i] << 16); f[2 * i + 1] = __uint_as_float(wv[i] & 0xffff0000u); }
            float amax = fmaxf(fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3]))), fmaxf(fmaxf(fabsf(f[4]), fabsf(f[5])), fmaxf(fabsf(f[6]), fabsf(f[7]))));
            amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0xB1, 0xF, 0xF, true)));   // quad_perm [1, 0, 3, 2]
            amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0x4E, 0xF, 0xF, true)));   // quad_perm [2, 3, 0, 1]
            const int be = sf_mx_exp(amax);
            const float inv = sf_mx_inv(be);
            const int be_hi = __shfl_xor(be, 4, 64);                                   // the head's other block (lanes ds ^ 4)
            if (live) {
              const int64_t orow = orow0 + (int64_t)(qh * 4 + qq) * p.n_groups;
              qt_u32x2 q8; q8.x = sf_fp8x4(f, inv); q8.y = sf_fp8x4(f + 4, inv);
              *reinterpret_cast<qt_u32x2*>(p.out_q + orow * p.ldq + ehead * 64 + ds * 8) = q8;
              if (ds == 0) *reinterpret_cast<uint16_t*>(p.out_s + (int64_t)(ehead >> 1) * p.splane + orow * 4 + (ehead & 1) * 2) = (uint16_t)(be | (be_hi << 8));
            }
          } else
          if (live) {
            typedef __attribute__((ext_vector_type(4))) unsigned int qt_u32x4;
            qt_u32x4* dst = reinterpret_cast<qt_u32x4*>(optr + (int64_t)(qh * 4 + qq) * p.n_groups * p.ldo);
            const qt_u32x4 wv = {w.x, w.y, w.z, w.w};
            if (QT_OUT_NT) __builtin_nontemporal_store(wv, dst); else *dst = wv;
          }
        }
        // the CLS query's weighted values over this patch's 8 tokens, then over the wave's patches (lanes 16 and 32 apart), + the CLS key's own share
        {
          const float* ep = reinterpret_cast<const float*>(slab + (pi * 8) * QT_SLAB_LD + 64);
          const float4 ea = *reinterpret_cast<const float4*>(ep), eb = *reinterpret_cast<const float4*>(ep + 4);
          const float ej[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
          sf_f32x2_t c[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) c[i] = sf_f32x2_t{0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const sf_f32x2_t e2 = {ej[j], ej[j]};
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = e2 * v[j + 1][i] + c[i];
          }
          float co[8] = {c[0].x, c[0].y, c[1].x, c[1].y, c[2].x, c[2].y, c[3].x, c[3].y};
#pragma unroll
          for (int i = 0; i < 8; ++i) { co[i] += __shfl_xor(co[i], 16, 64); co[i] += __shfl_xor(co[i], 32, 64); }
#pragma unroll
          for (int i = 0; i < 4; ++i) { co[2 * i] += e0 * v[0][i].x; co[2 * i + 1] += e0 * v[0][i].y; }
          if (pi == 0 && qh == 0) {
            float* part = p.cls_part + ((seq * QT_HEADS + ehead) * (p.n_groups >> 2) + (pp0 >> 2)) * 66;
            if (ds == 0) *reinterpret_cast<float2*>(part) = make_float2(M, L);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float2*>(part + 2 + ds * 8 + 2 * i) = make_float2(co[2 * i], co[2 * i + 1]);
          }
        }
      }
    }
    if (!more) break;
    t = tnext;
    tpar ^= 1;
  }
}

// X (n_seq * seq_rows, 768) bf16: the LayerNorm'ed tokens, seq_rows = 1 + 8 * n_groups ([CLS; frame-major patches]); W (2304, 768) bf16 = [q; k; v]
// rows, bias 2304 fp32 or NULL; qkv_cls (n_seq, 2304) bf16 = the same projection of every sequence's CLS row (sf_gemm_bf16 on the strided CLS
// rows); out (rows as X, 768) bf16: the PATCH rows are written (row 0 of every sequence comes from sf_attention_cls_combine on cls_partial,
// [n_seq][12][n_groups / 4][66] fp32: one record per wave = 4 patches).  Reference: vit_helper.py:97-150 (time attention of DividedSpaceTimeBlock), heads = 12, head dim 64.
static thread_local int g_qt_force_sched = -1;   // test hook (per calling thread): -1 default, 0 round 2's loop, 1 quadrant-phased
extern "C" void sf_qkv_time_force_schedule(int sched) { g_qt_force_sched = sched; }

static int qkv_time_impl(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* qkv_cls,
                         int64_t ldc, uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_groups, float scale,
                         const uint8_t* key_keep, void* stream);

extern "C" int sf_qkv_time_attention(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* qkv_cls,
                                     int64_t ldc, uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_groups, float scale,
                                     void* stream) {
  return qkv_time_impl(X, ldx, W, ldw, bias, qkv_cls, ldc, out, ldo, cls_partial, n_seq, n_groups, scale, nullptr, stream);
}

// The same with token masks: key_keep[row] == 0 (one byte per row of X, CLS rows included) masks K/V row `row` in the patch queries' time attention and
// in the CLS query's partials, as sf_attention_masked / sf_attention_cls_masked do on the un-fused path (vit_helper.py:34-42, 107-141).
extern "C" int sf_qkv_time_attention_masked(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* qkv_cls,
                                            int64_t ldc, uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_groups, float scale,
                                            const uint8_t* key_keep, void* stream) {
  SF_CHECK_ARG(key_keep, "sf_qkv_time_attention_masked: null key_keep");
  return qkv_time_impl(X, ldx, W, ldw, bias, qkv_cls, ldc, out, ldo, cls_partial, n_seq, n_groups, scale, key_keep, stream);
}

static int qkv_time_impl(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* qkv_cls,
                         int64_t ldc, uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_groups, float scale,
                         const uint8_t* key_keep, void* stream) {
  SF_CHECK_ARG(X && W && qkv_cls && out && cls_partial, "sf_qkv_time_attention: null pointer");
  SF_CHECK_ARG((n_groups % 4) == 0, "sf_qkv_time_attention: n_groups must be a multiple of 4 (a wave's four patches share one sequence)");
  SF_CHECK_ARG(n_groups >= 1 && (ldx % 8) == 0 && (ldw % 8) == 0 && (ldc % 8) == 0 && (ldo % 8) == 0, "sf_qkv_time_attention: row strides must be multiples of 8 elements");
  SF_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)qkv_cls % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                   (!bias || ((uintptr_t)bias % 16) == 0) && ((uintptr_t)cls_partial % 8) == 0, "sf_qkv_time_attention: operands must be 16-byte aligned");
  if (n_seq <= 0) return 0;
  const int64_t seq_rows = 1 + 8 * (int64_t)n_groups;
  SF_CHECK_ARG(n_seq * seq_rows * ldx * 2 < ((int64_t)1 << 32) && (int64_t)3 * QT_D * ldw * 2 < ((int64_t)1 << 32),
               "sf_qkv_time_attention: X and W must stay below 4 GiB (32-bit lane offsets)");
  if (int rc = sf_prepare_kernel((const void*)qkv_time_attn_kernel<true>, QT_LDS, "sf_qkv_time_attention")) return rc;
  if (int rc = sf_prepare_kernel((const void*)qkv_time_attn_kernel<false>, QT_LDS, "sf_qkv_time_attention")) return rc;
  const int n_cu = sf_cu_count("sf_qkv_time_attention");
  if (n_cu <= 0) return -1;
  QtArgs a;
  a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.bias = bias; a.qkv_cls = qkv_cls; a.ldc = ldc; a.out = out; a.ldo = ldo; a.cls_part = cls_partial;
  a.n_seq = n_seq; a.seq_rows = seq_rows; a.n_groups = n_groups; a.scale = scale; a.key_keep = key_keep;
  a.sX = nullptr; a.ldsx = 0; a.sW = nullptr; a.ldsw = 0;
  const int64_t tiles_m = (n_seq * n_groups + 31) / 32;
  SF_CHECK_ARG(tiles_m * QT_HEADS < ((int64_t)1 << 31) && n_seq * n_groups < ((int64_t)1 << 31), "sf_qkv_time_attention: too many tiles");
  a.tiles_m = (uint32_t)tiles_m;
  static int env_hc = -1;
  if (env_hc < 0) { const char* e = getenv("SF_QT_HEAD_CHUNK"); env_hc = e ? atoi(e) : 12; if (env_hc < 1 || QT_HEADS % env_hc) env_hc = 12; }   // round 4: all 12 heads per sweep (1276-1283 us against 1288-1293 for chunks of 6 on the quadrant-phased loop; r02 measured the opposite on the old loop)
  a.head_chunk = (uint32_t)env_hc;
  int64_t blocks = (n_cu / 8) * 8;
  if (blocks < 8) blocks = 8;                                    // (a device / partition with fewer than 8 CUs: never an empty grid)
  const int64_t need = ((tiles_m * QT_HEADS + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  static int env_sched = -1;
  if (env_sched < 0) { const char* e = getenv("SF_QT_SCHED"); env_sched = e ? atoi(e) : QT_PP; }
  const bool pp = g_qt_force_sched >= 0 ? g_qt_force_sched != 0 : env_sched != 0;
  if (pp) hipLaunchKernelGGL(qkv_time_attn_kernel<true>, dim3((unsigned)blocks), dim3(512), QT_LDS, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(qkv_time_attn_kernel<false>, dim3((unsigned)blocks), dim3(512), QT_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}

// The same launch on MXFP8 operands (fp8 towers of the synchronizability fine-tune): X (rows, 768) e4m3 bytes with its stage-major scale planes sX (6 planes,
// ldsx bytes apart, one dword per row) - what sf_layernorm768_mxfp8 / sf_gemm_mx_res_ln768 write -, W (2304, 768) e4m3 + sW (6 planes of 2304 dwords).  qkv_cls,
// out and cls_partial as in sf_qkv_time_attention (bf16 / fp32): the attention itself runs on the bf16-rounded projection, exactly as on the un-fused MX path
// (sf_gemm_mxfp8 with a bf16 output, then sf_attention).  Replaces sf_gemm_mxfp8 (temporal qkv) + sf_attention (time groups) + sf_attention_cls.
static int qkv_time_mx_impl(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW,
                            int64_t ldsw, const float* bias, const uint16_t* qkv_cls, int64_t ldc, uint16_t* out, int64_t ldo, uint8_t* out_q, int64_t ldq, uint8_t* out_s,
                            int64_t splane, float* cls_partial, int64_t n_seq, int n_groups, float scale, void* stream);

extern "C" int sf_qkv_time_attention_mx(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW,
                                        int64_t ldsw, const float* bias, const uint16_t* qkv_cls, int64_t ldc, uint16_t* out, int64_t ldo, float* cls_partial,
                                        int64_t n_seq, int n_groups, float scale, void* stream) {
  SF_CHECK_ARG(out && ((uintptr_t)out % 16) == 0 && (ldo % 8) == 0, "sf_qkv_time_attention_mx: out must be a 16-byte aligned bf16 buffer, ldo a multiple of 8");
  return qkv_time_mx_impl(X, ldx, sX, ldsx, W, ldw, sW, ldsw, bias, qkv_cls, ldc, out, ldo, nullptr, 0, nullptr, 0, cls_partial, n_seq, n_groups, scale, stream);
}

// ... with the attention output written as MXFP8 (out_q e4m3 bytes (rows, 768), row stride ldq; out_s the scale planes [6][rows][4], splane bytes apart) - the A operand
// of the MX projection that follows; byte for byte sf_quantize_mxfp8 of sf_qkv_time_attention_mx's bf16 output.  out_q / out_s must not alias X / sX (other
// workgroups still read them).  The CLS rows come from sf_attention_cls_combine_mx.
extern "C" int sf_qkv_time_attention_mx_q(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW,
                                          int64_t ldsw, const float* bias, const uint16_t* qkv_cls, int64_t ldc, uint8_t* out_q, int64_t ldq, uint8_t* out_s, int64_t splane,
                                          float* cls_partial, int64_t n_seq, int n_groups, float scale, void* stream) {
  SF_CHECK_ARG(out_q && out_s && ((uintptr_t)out_q % 8) == 0 && ((uintptr_t)out_s % 2) == 0 && (ldq % 8) == 0 && ldq >= QT_D && out_q != X && out_s != sX,
               "sf_qkv_time_attention_mx_q: out_q (8-byte aligned, ldq %% 8 == 0) / out_s must be buffers of their own");
  SF_CHECK_ARG(splane >= n_seq * (1 + 8 * (int64_t)n_groups) * 4, "sf_qkv_time_attention_mx_q: a scale plane holds 4 bytes per row");
  return qkv_time_mx_impl(X, ldx, sX, ldsx, W, ldw, sW, ldsw, bias, qkv_cls, ldc, nullptr, 0, out_q, ldq, out_s, splane, cls_partial, n_seq, n_groups, scale, stream);
}

static int qkv_time_mx_impl(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW,
                            int64_t ldsw, const float* bias, const uint16_t* qkv_cls, int64_t ldc, uint16_t* out, int64_t ldo, uint8_t* out_q, int64_t ldq, uint8_t* out_s,
                            int64_t splane, float* cls_partial, int64_t n_seq, int n_groups, float scale, void* stream) {
  SF_CHECK_ARG(X && sX && W && sW && qkv_cls && (out || out_q) && cls_partial, "sf_qkv_time_attention_mx: null pointer");
  SF_CHECK_ARG((n_groups % 4) == 0 && n_groups >= 4, "sf_qkv_time_attention_mx: n_groups must be a multiple of 4 (a wave's four patches share one sequence)");
  SF_CHECK_ARG((ldx % 16) == 0 && (ldw % 16) == 0 && ldx >= QT_D && ldw >= QT_D && (ldc % 8) == 0, "sf_qkv_time_attention_mx: bad row strides");
  SF_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)sX % 4) == 0 && ((uintptr_t)sW % 4) == 0 && ((uintptr_t)qkv_cls % 16) == 0 &&
                   (!bias || ((uintptr_t)bias % 16) == 0) && ((uintptr_t)cls_partial % 8) == 0,
               "sf_qkv_time_attention_mx: operands must be 16-byte aligned (scale planes 4-byte)");
  if (n_seq <= 0) return 0;
  const int64_t seq_rows = 1 + 8 * (int64_t)n_groups;
  SF_CHECK_ARG(n_seq * seq_rows * ldx < ((int64_t)1 << 32) && (int64_t)3 * QT_D * ldw < ((int64_t)1 << 32), "sf_qkv_time_attention_mx: X and W must stay below 4 GiB");
  SF_CHECK_ARG((ldsx % 4) == 0 && (ldsw % 4) == 0 && ldsx >= n_seq * seq_rows * 4 && ldsw >= 3 * QT_D * 4 && ldsx < ((int64_t)1 << 32) && ldsw < ((int64_t)1 << 32),
               "sf_qkv_time_attention_mx: scale planes must hold one dword per row of X / W");
  if (int rc = sf_prepare_kernel((const void*)qkv_time_attn_kernel<true, true>, QT_LDS, "sf_qkv_time_attention_mx")) return rc;
  const int n_cu = sf_cu_count("sf_qkv_time_attention_mx");
  if (n_cu <= 0) return -1;
  QtArgs a;
  a.X = reinterpret_cast<const bf16_t*>(X); a.ldx = ldx; a.W = reinterpret_cast<const bf16_t*>(W); a.ldw = ldw; a.bias = bias; a.qkv_cls = qkv_cls; a.ldc = ldc;
  a.out = out; a.ldo = ldo; a.cls_part = cls_partial; a.n_seq = n_seq; a.seq_rows = seq_rows; a.n_groups = n_groups; a.scale = scale; a.key_keep = nullptr;
  a.sX = sX; a.ldsx = ldsx; a.sW = sW; a.ldsw = ldsw;
  a.out_q = out_q; a.ldq = ldq; a.out_s = out_s; a.splane = splane;
  const int64_t tiles_m = (n_seq * n_groups + 31) / 32;
  SF_CHECK_ARG(tiles_m * QT_HEADS < ((int64_t)1 << 31) && n_seq * n_groups < ((int64_t)1 << 31), "sf_qkv_time_attention_mx: too many tiles");
  a.tiles_m = (uint32_t)tiles_m;
  a.head_chunk = 6;
  int64_t blocks = (n_cu / 8) * 8;
  if (blocks < 8) blocks = 8;                                    // (a device / partition with fewer than 8 CUs: never an empty grid)
  const int64_t need = ((tiles_m * QT_HEADS + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL((qkv_time_attn_kernel<true, true>), dim3((unsigned)blocks), dim3(512), QT_LDS, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
