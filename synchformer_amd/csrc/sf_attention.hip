// Attention kernels for the Synchformer hot path on gfx950.  All read q/k/v as bf16 column slices of a packed
// projection output (row stride `ld`, head h at columns h*D .. h*D+D-1 from each base pointer), compute
// softmax(scale * q k^T) v in fp32 and write bf16 (row stride `ldo`).  No score matrix ever reaches HBM.
//
// Token groups.  A "sequence" is `seq_rows` consecutive rows.  Group g of a sequence owns tokens
//     row(i) = row0 + g*group_stride + i*tok_stride,  i < n_tok
// and (optionally) an extra first key/value = row `cls_row` of the same sequence.  This one description
// covers every attention in the model:
//   * Motionformer time  attention (vit_helper.py:341-344 '(b n) f d'): 196 groups, n_tok 8,  tok_stride 196, +CLS key
//   * Motionformer space attention (            '(b f) n d'): 8 groups,   n_tok 196, tok_stride 1,   +CLS key
//   * AST / sync-transformer / aggregator full self-attention: 1 group, n_tok = L, no extra key
// The CLS query itself (attends to ALL rows of its sequence, vit_helper.py:126) is `sf_attention_cls`.
#include "sf_common.h"
#include <stdlib.h>
#include "../../include/synchformer_hip.h"

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; int64_t ld;
  bf16_t* out; int64_t ldo;
  int64_t seq_rows;       // rows per sequence in q/k/v and out
  int n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads;
  float scale;
  // optional: partial softmax state (m, l, o[64], base-2 domain) of the CLS QUERY over this group's keys,
  // [seq][head][group][66] fp32 - merged by sf_attention_cls_combine into output row `cls_row` (vit_helper.py:126)
  float* cls_part;
  // optional key mask (vis_mask / aud_mask token masks, vit_helper.py:34-42): key_keep[row] == 0 -> that K/V row gets -inf
  const uint8_t* key_keep;
  // optional MXFP8 output (attn_mfma_kernel<64, 13, true>, sf_attention_cls_partial_mx): e4m3 bytes (rows, heads * 64) + one E8M0 byte per row and 32 columns in
  // the scale planes [heads * 64 / 128][rows][4] (plane stride `splane` bytes) - the A operand layout of sf_gemm_mxfp8 / sf_gemm_mx_res_ln768
  uint8_t* out_q = nullptr; int64_t ldq = 0; uint8_t* out_s = nullptr; int64_t splane = 0;
};

// ======================================================================================================
// (1) tiny groups (n_tok <= 8, D = 64): pure VALU, one wave per (seq, group, head).
// lane = (query qi = lane>>3, slice sub = lane&7 -> 8 of the 64 head dims).  Every q/k/v access is a 16-byte
// load; k/v rows are shared by the 8 query lanes (same address -> one fetch).  HBM-bound by construction.
// ======================================================================================================
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

typedef __attribute__((ext_vector_type(2))) __bf16 att_bf2;
// <q, k> over 8 bf16 pairs packed in two uint4: v_dot2c_f32_bf16 multiplies the bf16 pairs exactly and accumulates in fp32 -
// no unpacking of q or k at all (the kernel is VALU-issue bound, not HBM bound: ~400 -> ~250 instructions per wave).
__device__ __forceinline__ float dot8_bf16(const uint4& a, const uint4& b) {
  float d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(att_bf2, a.x), __builtin_bit_cast(att_bf2, b.x), 0.f, false);
  d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(att_bf2, a.y), __builtin_bit_cast(att_bf2, b.y), d, false);
  d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(att_bf2, a.z), __builtin_bit_cast(att_bf2, b.z), d, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(att_bf2, a.w), __builtin_bit_cast(att_bf2, b.w), d, false);
}
// o[0..7] += e * v (8 bf16 in a uint4), as four packed fp32 FMAs
__device__ __forceinline__ void axpy8_bf16(sf_f32x2_t (&o)[4], float e, const uint4& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  const sf_f32x2_t e2 = {e, e};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const sf_f32x2_t vf = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
    o[i] = e2 * vf + o[i];
  }
}

// K/V (and the CLS query) are fetched ONCE per wave - lane (r, sub) loads 16 bytes of token r - and shared through 2.4 KB of
// wave-private LDS; every query lane then reads key j's slice back with a broadcasting ds_read_b128.  Compared with each query
// lane loading all 9 keys itself (18 vector loads per wave that are 8x redundant across lanes and 72 live registers) this is 4
// vector loads and ~70 VGPRs, i.e. 7 waves per SIMD instead of 3: the kernel is bound by bytes in flight, not by VALU.
#define TINY_SLOTS 9
template <bool PART>
__global__ __launch_bounds__(256) void attn_tiny64_kernel(AttnArgs p, int64_t total_units) {
  __shared__ __attribute__((aligned(16))) uint4 lds_k[4][TINY_SLOTS][8], lds_v[4][TINY_SLOTS][8], lds_qc[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;                   // (seq, group, head), head fastest
  if (unit >= total_units) return;
  const int head = (int)(unit % p.heads);
  const int64_t sg = unit / p.heads;
  const int g = (int)(sg % p.n_groups);
  const int64_t seq = sg / p.n_groups;
  const int qi = lane >> 3, sub = lane & 7;
  const int64_t seq_base = seq * p.seq_rows;
  const int64_t first = seq_base + p.row0 + (int64_t)g * p.group_stride;
  const int col = head * 64 + sub * 8;
  const int has_cls = p.cls_row >= 0 ? 1 : 0;
  const int nk = p.n_tok + has_cls;
  const int qtok = qi < p.n_tok ? qi : p.n_tok - 1;                      // idle lanes shadow the last token
  const int64_t my_row = first + (int64_t)qtok * p.tok_stride;
  const uint4 qraw = *reinterpret_cast<const uint4*>(p.q + my_row * p.ld + col);
  const uint4 kraw = *reinterpret_cast<const uint4*>(p.k + my_row * p.ld + col);
  const uint4 vraw = *reinterpret_cast<const uint4*>(p.v + my_row * p.ld + col);
  if (has_cls && qi < (PART ? 3 : 2)) {                                  // lanes 0-7: CLS key, 8-15: CLS value, 16-23: CLS query
    const bf16_t* src = qi == 0 ? p.k : (qi == 1 ? p.v : p.q);
    const uint4 c = *reinterpret_cast<const uint4*>(src + (seq_base + p.cls_row) * p.ld + col);
    if (qi == 0) lds_k[wave][0][sub] = c; else if (qi == 1) lds_v[wave][0][sub] = c; else lds_qc[wave][sub] = c;
  }
  if (qi < p.n_tok) { lds_k[wave][has_cls + qi][sub] = kraw; lds_v[wave][has_cls + qi][sub] = vraw; }
  uint32_t keep_bits = 0x1ffu;                                   // token keep flags of the 9 key slots (masked entry point only)
  if (p.key_keep) {
#pragma unroll
    for (int j = 0; j < TINY_SLOTS; ++j) {
      const int jj = j < nk ? j : nk - 1;
      const int64_t row = (has_cls && jj == 0) ? seq_base + p.cls_row : first + (int64_t)(jj - has_cls) * p.tok_stride;
      if (p.key_keep[row] == 0) keep_bits &= ~(1u << j);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float sc = p.scale * 1.44269504088896f;                 // softmax in base 2: exp(x) = exp2(x * log2 e)
  float s[TINY_SLOTS];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < TINY_SLOTS; ++j) {
    const uint4 kk = lds_k[wave][j < nk ? j : nk - 1][sub];
    float d = dot8_bf16(qraw, kk);
    d = sum8_dpp(d);
    s[j] = j < nk ? d * sc : -INFINITY;
    m = fmaxf(m, s[j]);
  }
  if (keep_bits != 0x1ffu) {                                     // wave-uniform: only groups that actually contain a masked key
    m = -INFINITY;
#pragma unroll
    for (int j = 0; j < TINY_SLOTS; ++j) {
      if (!((keep_bits >> j) & 1u)) s[j] = -INFINITY;
      m = fmaxf(m, s[j]);
    }
  }
  float l = 0.f;
  sf_f32x2_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = sf_f32x2_t{0.f, 0.f};
#pragma unroll
  for (int j = 0; j < TINY_SLOTS; ++j) {
    const float e = __builtin_amdgcn_exp2f(s[j] - m);                             // 0 for the masked tail (s = -inf)
    l += e;
    axpy8_bf16(o, e, lds_v[wave][j < nk ? j : nk - 1][sub]);
  }
  if (qi < p.n_tok) {
    const float inv = 1.0f / l;
    uint4 w;
    w.x = pack_bf2(o[0].x * inv, o[0].y * inv); w.y = pack_bf2(o[1].x * inv, o[1].y * inv);
    w.z = pack_bf2(o[2].x * inv, o[2].y * inv); w.w = pack_bf2(o[3].x * inv, o[3].y * inv);
    *reinterpret_cast<uint4*>(p.out + (first + (int64_t)qi * p.tok_stride) * p.ldo + col) = w;
  }
  if (PART) {
    // the CLS query's share of this group: keys 1..n_tok (plus the CLS key itself in group 0 only).  Row group qi scores ITS token
    // (k/v are the registers it loaded), the softmax state and the weighted values are then reduced across the eight row groups.
    const uint4 qc = lds_qc[wave][sub];
    float d = dot8_bf16(qc, lds_k[wave][has_cls + qtok][sub]);          // this row group's own token, back from LDS (keeps registers low)
    d = sum8_dpp(d);
    const float cs = (qi < p.n_tok && ((keep_bits >> (has_cls + qi)) & 1u)) ? d * sc : -INFINITY;
    float c0 = -INFINITY;
    if (has_cls && g == 0 && (keep_bits & 1u)) {                              // wave-uniform
      float d0 = dot8_bf16(qc, lds_k[wave][0][sub]);
      d0 = sum8_dpp(d0);
      c0 = d0 * sc;
    }
    float cm = cs;
    cm = fmaxf(cm, __shfl_xor(cm, 8, 64)); cm = fmaxf(cm, __shfl_xor(cm, 16, 64)); cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
    cm = fmaxf(cm, c0);
    const float e = __builtin_amdgcn_exp2f(cs - cm);
    float cl = e;
    sf_f32x2_t co[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) co[i] = sf_f32x2_t{0.f, 0.f};
    axpy8_bf16(co, e, lds_v[wave][has_cls + qtok][sub]);
#pragma unroll
    for (int sh = 8; sh < 64; sh <<= 1) {
      cl += __shfl_xor(cl, sh, 64);
#pragma unroll
      for (int i = 0; i < 4; ++i) { co[i].x += __shfl_xor(co[i].x, sh, 64); co[i].y += __shfl_xor(co[i].y, sh, 64); }
    }
    if (c0 != -INFINITY) {                                                     // wave-uniform
      const float e0 = __builtin_amdgcn_exp2f(c0 - cm);
      cl += e0;
      axpy8_bf16(co, e0, lds_v[wave][0][sub]);
    }
    if (qi == 0) {
      float* part = p.cls_part + ((seq * p.heads + head) * p.n_groups + g) * 66;
      if (sub == 0) *reinterpret_cast<float2*>(part) = make_float2(cm, cl);            // 66-float records are 8-byte aligned
#pragma unroll
      for (int t = 0; t < 4; ++t) *reinterpret_cast<float2*>(part + 2 + sub * 8 + 2 * t) = make_float2(co[t].x, co[t].y);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Backward of the tiny-group attention above (Motionformer time attention in the Stage-1 train step): one wave per (seq, group, head),
// n_tok <= 8 queries against [CLS key; n_tok keys].  Phase 1 is query-major like the forward (lane = (query, 8-dim slice)): recompute
// p = softmax(q k^T * scale), dp = dO v^T, ds = p (dp - sum_j p dp) scale, and dq = ds k.  Phase 2 is key-major (lane = (key, slice)):
// dk_j = sum_i ds_ij q_i, dv_j = sum_i p_ij dO_i, with q, dO, ds, p exchanged through 2.6 KB of wave-private LDS.  Token keys write
// their dk | dv rows directly (a token belongs to exactly one group); the CLS key's share goes to cls_part[(seq, group)][k|v][head*64+d]
// and is summed over groups by sf_reduce_groups_bf16.
// ------------------------------------------------------------------------------------------------------
struct AttnBwdArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; int64_t ld;
  const bf16_t* dO; int64_t lddo;
  bf16_t* dq; bf16_t* dk; bf16_t* dv; int64_t ldg;
  bf16_t* cls_part;                       // (n_seq * n_groups, 2 * heads * 64)
  int64_t seq_rows;
  int n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads;
  float scale;
  // optional (attn_group_bwd_kernel, sf_attention_group_bwd_clsq): the CLS QUERY's backward (vit_helper.py:126: it attends every key of the sequence) rides along as
  // one more query row of every group - with the statistics of its softmax over ALL keys from the forward (cls_stats [seq][head][2] = m in the base-2 domain, l),
  // its output row `o` (delta = <dO, o>), and its dq as one partial row per group in dq_cls_part (n_seq * n_groups, heads * 64)
  const float* cls_stats = nullptr; const bf16_t* o = nullptr; int64_t ldo = 0; bf16_t* dq_cls_part = nullptr;
};

__global__ __launch_bounds__(256) void attn_tiny64_bwd_kernel(AttnBwdArgs p, int64_t total_units) {
  __shared__ __attribute__((aligned(16))) bf16_t q_l[4][8][64], do_l[4][8][64];
  __shared__ float ds_l[4][8][12], p_l[4][8][12];
  __shared__ __attribute__((aligned(16))) uint4 k_l[4][TINY_SLOTS][8], v_l[4][TINY_SLOTS][8];   // keys / values: fetched once per wave, like the forward
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wv;
  if (unit >= total_units) return;
  const int head = (int)(unit % p.heads);
  const int64_t sg = unit / p.heads;
  const int g = (int)(sg % p.n_groups);
  const int64_t seq = sg / p.n_groups;
  const int qi = lane >> 3, sub = lane & 7;
  const int64_t seq_base = seq * p.seq_rows;
  const int64_t first = seq_base + p.row0 + (int64_t)g * p.group_stride;
  const int col = head * 64 + sub * 8;
  const bool q_valid = qi < p.n_tok;
  const int qtok = q_valid ? qi : p.n_tok - 1;
  const int64_t qrow = first + (int64_t)qtok * p.tok_stride;
  const uint4 qraw = *reinterpret_cast<const uint4*>(p.q + qrow * p.ld + col);
  const uint4 doraw = *reinterpret_cast<const uint4*>(p.dO + qrow * p.lddo + col);
  const int has_cls = p.cls_row >= 0 ? 1 : 0;
  const int nk = p.n_tok + has_cls;
  // lane (r, sub) fetches 16 bytes of ITS token's key and value (lanes 0-7 / 8-15 the CLS key / value as well) and every query lane reads key j's slice back from
  // wave-private LDS: 4 vector loads per lane instead of 18 eight-fold redundant ones and 72 fewer live registers (the forward's layout, attn_tiny64_kernel)
  {
    const uint4 kmine = *reinterpret_cast<const uint4*>(p.k + qrow * p.ld + col);
    const uint4 vmine = *reinterpret_cast<const uint4*>(p.v + qrow * p.ld + col);
    if (has_cls && qi < 2) {
      const uint4 c = *reinterpret_cast<const uint4*>((qi == 0 ? p.k : p.v) + (seq_base + p.cls_row) * p.ld + col);
      if (qi == 0) k_l[wv][0][sub] = c; else v_l[wv][0][sub] = c;
    }
    if (q_valid) { k_l[wv][has_cls + qi][sub] = kmine; v_l[wv][has_cls + qi][sub] = vmine; }
  }
  *reinterpret_cast<uint4*>(&q_l[wv][qi][sub * 8]) = qraw;
  *reinterpret_cast<uint4*>(&do_l[wv][qi][sub * 8]) = doraw;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#define TINY_K(j) k_l[wv][(j) < nk ? (j) : nk - 1][sub]
#define TINY_V(j) v_l[wv][(j) < nk ? (j) : nk - 1][sub]
  // ---- the CLS query (optional, sf_attention_tiny_bwd_clsq): it attends every key of the sequence, so here it is a ninth query on this group's keys, normalised with
  // the forward's statistics; lane group qi handles key qi, key 8 is computed by every group (used by group 0's lanes in the second pass); the CLS KEY counts for
  // it in group 0 only.  pc / dsc: its probability and score gradient for this lane group's key, p8 / ds8 for key 8.
  const bool clsq = p.cls_stats != nullptr;
  uint4 qc_raw = make_uint4(0, 0, 0, 0), doc_raw = qc_raw;
  float pc = 0.f, dsc = 0.f, p8 = 0.f, ds8 = 0.f;
  if (clsq) {
    const int64_t cr = seq_base + p.cls_row;
    qc_raw = *reinterpret_cast<const uint4*>(p.q + cr * p.ld + col);
    doc_raw = *reinterpret_cast<const uint4*>(p.dO + cr * p.lddo + col);
    const uint4 oc = *reinterpret_cast<const uint4*>(p.o + cr * p.ldo + col);
    const float* gs = p.cls_stats + (seq * p.heads + head) * 2;
    const float Mg = gs[0], Linv = 1.0f / gs[1];
    const uint4 ksel = TINY_K(qi), vsel = TINY_V(qi), k8 = TINY_K(8), v8 = TINY_V(8);
    float dl = dot8_bf16(doc_raw, oc), sc_ = dot8_bf16(qc_raw, ksel), dpc = dot8_bf16(doc_raw, vsel), s8 = dot8_bf16(qc_raw, k8), dp8 = dot8_bf16(doc_raw, v8);
    dl = sum8_dpp(dl); sc_ = sum8_dpp(sc_); dpc = sum8_dpp(dpc); s8 = sum8_dpp(s8); dp8 = sum8_dpp(dp8);
    const float sc2c = p.scale * 1.44269504088896f;
    if (qi < nk && !(has_cls && qi == 0 && g != 0)) { pc = __builtin_amdgcn_exp2f(fmaf(sc_, sc2c, -Mg)) * Linv; dsc = pc * (dpc - dl) * p.scale; }
    if (8 < nk) { p8 = __builtin_amdgcn_exp2f(fmaf(s8, sc2c, -Mg)) * Linv; ds8 = p8 * (dp8 - dl) * p.scale; }
    // its dq over this group's keys: lane group qi holds ds_qi * k_qi (+ key 8 in group 0's lanes), summed over the lane groups
    sf_f32x2_t dqc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dqc[i] = sf_f32x2_t{0.f, 0.f};
    axpy8_bf16(dqc, dsc, ksel);
    if (qi == 0) axpy8_bf16(dqc, ds8, k8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dqc[i].x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dqc[i].x), 0x128, 0xF, 0xF, true));   // row_ror:8 = the lane 8 away in the 16-lane row
      dqc[i].y += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dqc[i].y), 0x128, 0xF, 0xF, true));
#pragma unroll
      for (int o_ = 16; o_ < 64; o_ <<= 1) { dqc[i].x += __shfl_xor(dqc[i].x, o_, 64); dqc[i].y += __shfl_xor(dqc[i].y, o_, 64); }
    }
    if (qi == 0) {
      uint4 w;
      w.x = pack_bf2(dqc[0].x, dqc[0].y); w.y = pack_bf2(dqc[1].x, dqc[1].y); w.z = pack_bf2(dqc[2].x, dqc[2].y); w.w = pack_bf2(dqc[3].x, dqc[3].y);
      *reinterpret_cast<uint4*>(p.dq_cls_part + (seq * p.n_groups + g) * (int64_t)(p.heads * 64) + col) = w;
    }
  }
  // ---- phase 1: probabilities, dp, ds, dq ---------------------------------------------------------------------------------
  const float sc2 = p.scale * 1.44269504088896f;
  float s[9], dp[9], m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    float d = dot8_bf16(qraw, TINY_K(j)), e = dot8_bf16(doraw, TINY_V(j));
    d = sum8_dpp(d);
    e = sum8_dpp(e);
    s[j] = j < nk ? d * sc2 : -INFINITY;
    dp[j] = e;
    m = fmaxf(m, s[j]);
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < 9; ++j) { s[j] = __builtin_amdgcn_exp2f(s[j] - m); l += s[j]; }
  const float linv = q_valid ? 1.0f / l : 0.f;                   // idle query lanes contribute nothing to dk / dv
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < 9; ++j) { s[j] *= linv; delta += s[j] * dp[j]; }
  sf_f32x2_t dqa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dqa[i] = sf_f32x2_t{0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const float ds = s[j] * (dp[j] - delta) * p.scale;
    axpy8_bf16(dqa, ds, TINY_K(j));
    if (sub == 0) { ds_l[wv][qi][j] = ds; p_l[wv][qi][j] = s[j]; }
  }
  if (q_valid) {
    uint4 w;
    w.x = pack_bf2(dqa[0].x, dqa[0].y); w.y = pack_bf2(dqa[1].x, dqa[1].y);
    w.z = pack_bf2(dqa[2].x, dqa[2].y); w.w = pack_bf2(dqa[3].x, dqa[3].y);
    *reinterpret_cast<uint4*>(p.dq + qrow * p.ldg + col) = w;
  }
  // ---- phase 2: key-major.  lane (kj = lane >> 3, sub) owns key kj; key 8 is done by the kj == 0 lanes in a second pass ---------
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int j = pass == 0 ? qi : 8;
    if (j >= nk || (pass == 1 && qi != 0)) continue;
    sf_f32x2_t dka[4], dva[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dka[i] = sf_f32x2_t{0.f, 0.f}; dva[i] = sf_f32x2_t{0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint4 qv = *reinterpret_cast<const uint4*>(&q_l[wv][i][sub * 8]);
      const uint4 dv_ = *reinterpret_cast<const uint4*>(&do_l[wv][i][sub * 8]);
      axpy8_bf16(dka, ds_l[wv][i][j], qv);
      axpy8_bf16(dva, p_l[wv][i][j], dv_);
    }
    if (clsq) {                                                    // the CLS query's share of this key
      axpy8_bf16(dka, pass == 0 ? dsc : ds8, qc_raw);
      axpy8_bf16(dva, pass == 0 ? pc : p8, doc_raw);
    }
    uint4 wk, wvv;
    wk.x = pack_bf2(dka[0].x, dka[0].y); wk.y = pack_bf2(dka[1].x, dka[1].y); wk.z = pack_bf2(dka[2].x, dka[2].y); wk.w = pack_bf2(dka[3].x, dka[3].y);
    wvv.x = pack_bf2(dva[0].x, dva[0].y); wvv.y = pack_bf2(dva[1].x, dva[1].y); wvv.z = pack_bf2(dva[2].x, dva[2].y); wvv.w = pack_bf2(dva[3].x, dva[3].y);
    if (has_cls && j == 0) {
      bf16_t* cp = p.cls_part + (seq * p.n_groups + g) * (int64_t)(2 * p.heads * 64);
      *reinterpret_cast<uint4*>(cp + col) = wk;
      *reinterpret_cast<uint4*>(cp + p.heads * 64 + col) = wvv;
    } else {
      const int64_t row = first + (int64_t)(j - has_cls) * p.tok_stride;
      *reinterpret_cast<uint4*>(p.dk + row * p.ldg + col) = wk;
      *reinterpret_cast<uint4*>(p.dv + row * p.ldg + col) = wvv;
    }
  }
}
#undef TINY_K
#undef TINY_V

static int attention_tiny_bwd_impl(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dO, int64_t lddo, bf16_t* dq, bf16_t* dk,
                                   bf16_t* dv, int64_t ldg, bf16_t* cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride,
                                   int n_tok, int cls_row, int heads, int head_dim, float scale, const float* cls_stats, const bf16_t* o, int64_t ldo, bf16_t* dq_cls_part,
                                   void* stream);

extern "C" int sf_attention_tiny_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dO, int64_t lddo, bf16_t* dq, bf16_t* dk,
                                     bf16_t* dv, int64_t ldg, bf16_t* cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0,
                                     int group_stride, int tok_stride, int n_tok, int cls_row, int heads, int head_dim, float scale, void* stream) {
  return attention_tiny_bwd_impl(q, k, v, ld, dO, lddo, dq, dk, dv, ldg, cls_part, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim,
                                 scale, nullptr, nullptr, 0, nullptr, stream);
}

// sf_attention_tiny_bwd with the CLS QUERY's backward in the same launch (arguments as sf_attention_group_bwd_clsq; statistics from sf_attention_cls_stats).
extern "C" int sf_attention_tiny_bwd_clsq(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dO, int64_t lddo, bf16_t* dq, bf16_t* dk,
                                          bf16_t* dv, int64_t ldg, bf16_t* cls_part, const float* cls_stats, const bf16_t* o, int64_t ldo, bf16_t* dq_cls_part,
                                          int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads,
                                          int head_dim, float scale, void* stream) {
  SF_CHECK_ARG(cls_stats && o && dq_cls_part && cls_row >= 0 && (ldo % 8) == 0, "sf_attention_tiny_bwd_clsq: needs statistics, the forward output, a partial buffer and a CLS row");
  return attention_tiny_bwd_impl(q, k, v, ld, dO, lddo, dq, dk, dv, ldg, cls_part, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim,
                                 scale, cls_stats, o, ldo, dq_cls_part, stream);
}

static int attention_tiny_bwd_impl(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dO, int64_t lddo, bf16_t* dq, bf16_t* dk,
                                   bf16_t* dv, int64_t ldg, bf16_t* cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride,
                                   int n_tok, int cls_row, int heads, int head_dim, float scale, const float* cls_stats, const bf16_t* o, int64_t ldo, bf16_t* dq_cls_part,
                                   void* stream) {
  SF_CHECK_ARG(q && k && v && dO && dq && dk && dv, "sf_attention_tiny_bwd: null pointer");
  SF_CHECK_ARG(head_dim == 64 && n_tok >= 1 && n_tok <= 8 && (cls_row < 0 || cls_part), "sf_attention_tiny_bwd: head_dim 64, n_tok <= 8, cls_part with cls_row");
  SF_CHECK_ARG((ld % 8) == 0 && (lddo % 8) == 0 && (ldg % 8) == 0 && n_groups >= 1 && heads >= 1, "sf_attention_tiny_bwd: bad strides / counts");
  if (n_seq <= 0) return 0;
  AttnBwdArgs a;
  a.q = q; a.k = k; a.v = v; a.ld = ld; a.dO = dO; a.lddo = lddo; a.dq = dq; a.dk = dk; a.dv = dv; a.ldg = ldg; a.cls_part = cls_part;
  a.seq_rows = seq_rows; a.n_groups = n_groups; a.row0 = row0; a.group_stride = group_stride; a.tok_stride = tok_stride; a.n_tok = n_tok;
  a.cls_row = cls_row; a.heads = heads; a.scale = scale;
  a.cls_stats = cls_stats; a.o = o; a.ldo = ldo; a.dq_cls_part = dq_cls_part;
  const int64_t units = n_seq * n_groups * heads;
  hipLaunchKernelGGL(attn_tiny64_bwd_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, units);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Fused backward of the grouped attention for up to 208 keys, head_dim 64 (Motionformer space attention: 196 queries x [CLS; 196 keys];
// AST: 74 x 74).  One 512-thread workgroup per (seq, group, head); K, V, Q, dO rows of the group are staged once in LDS (row-major,
// 144-byte rows).  Everything runs on v_mfma_f32_16x16x16_bf16, whose C layout (lane: column l & 15, rows (l >> 4)*4 + r) IS its A / B
// operand layout (lane: row l & 15, k = (l >> 4)*4 + e): a probability tile computed as S^T = K Q^T is directly the A operand
// "rows = query, k = key" of dQ = dS K, and the same tile computed as S = Q K^T is directly the A operand "rows = key, k = query" of
// dK = dS^T Q and dV = P^T dO - so P and dS never go through LDS; computing the score / dP tiles twice (once per layout) is cheaper
// than transposing them.  B operands that contract over keys or queries are column fragments of the row-major LDS copies, read
// with ds_read_b64_tr_b16.
//   pass 1 (query tiles over the waves): softmax statistics per query (max, 1/sum, delta = sum_j p dp) -> LDS, and dQ;
//   pass 2 (key tiles over the waves):   dK and dV, with p = exp2(s*c - m) / l rebuilt from the statistics.
// The CLS key's dk | dv share goes to cls_part like in the tiny-group kernel.
// ------------------------------------------------------------------------------------------------------
#ifndef GB_ABL
#define GB_ABL 0                 // measurement builds (tools/ab_pp.sh with SRC=sf_attention): 1 = no pass-1 tile loop, 2 = no pass-2 tile loop, 4 = no staging loads
#endif
#define GB_LD 80                 // bf16 per LDS row of the staged K | Q | dO: 160 B (64 + 16 pad), chunk bit 0 flipped in rows 8-15 of every tile (gb_off)
#define GB_LDV 72                // bf16 per LDS row of the staged V (read as row fragments only) and of the per-wave output staging tiles: 144 B
#define GB_ROWS 208
#define GB_MAT (GB_ROWS * GB_LD * 2)                       // 33,280 B per staged K / Q / dO
#define GB_MATV (GB_ROWS * GB_LDV * 2)                     // 29,952 B for V
#define GB_WAVES 16               // one wave per query tile (pass 1) / key tile (pass 2) of the 13; four waves per SIMD hide the dependent LDS round trips of the fragment reads
#define GB_ST 4                  // floats per query in the LDS statistics: m (base-2 domain), 1/l, delta, pad (one ds_read_b128)
#define GB_OUTW(WAVES) ((WAVES) < 13 ? (WAVES) : 13)      // waves that ever own a tile (208 rows = 13 tiles) and therefore an output staging tile
#define GB_LDS(WAVES) (3 * GB_MAT + GB_MATV + GB_ROWS * GB_ST * 4 + GB_OUTW(WAVES) * 16 * GB_LDV * 2 + 16)   // 163,088 B of 163,840 (+ 16: the CLS query's delta)

// Element offset of X[row][k] in a staged K / Q / dO.  The kernel reads these matrices two ways, and one row stride cannot serve both without bank conflicts:
//   * row fragments (ds_read_b128; 16 lanes = 16 consecutive rows, the same 16-byte chunk): 36-dword rows (round 3) put the 16 rows on 16 different 4-bank groups;
//   * ds_read_b64_tr_b16 column fragments (32 lanes = 8 consecutive rows x 32 B): 36-dword rows wrap row 7 onto row 0's banks - measured: 33 % of the kernel's
//     LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r05_round.md).  40-dword rows put the eight 8-dword blocks on the
//     eight 8-bank groups exactly, but rows r and r + 8 of a row-fragment read then share their banks.
// 40-dword rows AND 16-byte chunk c stored at slot c ^ ((row >> 3) & 1): rows 8-15 of a tile move by four banks into the groups rows 0-7 leave free (row
// fragments conflict-free), the column fragments of a 32-lane pass all carry the same flip (their 32-byte blocks only swap halves).  The flip touches chunk bit 0
// only - k-step (chunk bits 2) and column tile (chunk bits 1-2) stay immediate offsets of one lane address.
__device__ __forceinline__ int gb_off(int row, int k) { return row * GB_LD + ((((k >> 3) ^ ((row >> 3) & 1)) << 3) | (k & 7)); }
__device__ __forceinline__ bf16x4 gb_row_frag(const bf16_t* X, int row, int k0) { return *reinterpret_cast<const bf16x4*>(X + gb_off(row, k0)); }
__device__ __forceinline__ bf16x4 gb_row_frag_v(const bf16_t* V, int row, int k0) { return *reinterpret_cast<const bf16x4*>(V + row * GB_LDV + k0); }
// Column fragment X[row0 + 0..3][col0 + lr] of a row-major LDS matrix (lane = 16*lg + lr; row0 is the same for the 16 lanes of a group)
// with ONE ds_read_b64_tr_b16: the lane points at the 8 bytes X[row0 + (lr >> 2)][col0 + 4*(lr & 3) .. +3] and the hardware hands lane i
// element (i & 3) of lanes (i >> 2) + 4 j (was: four 2-byte reads + packing per fragment).
typedef short gb_s4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 gb_col_frag(const bf16_t* X, int row0, int col0, int lr) {
  const bf16_t* src = X + gb_off(row0 + (lr >> 2), col0 + (lr & 3) * 4);
  const gb_s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gb_s4*)src);
  return __builtin_bit_cast(bf16x4, r);
}
__device__ __forceinline__ bf16x4 gb_pack(const f32x4& v) {
  union { bf16x4 f; uint32_t u[2]; } r;
  r.u[0] = pack_bf2(v[0], v[1]); r.u[1] = pack_bf2(v[2], v[3]);
  return r.f;
}

// X32 (the product): the same passes on v_mfma_f32_16x16x32_bf16 - half the matrix instructions.  Its A / B operands hold 8 consecutive k per lane (k = (l >> 4)*8 + e);
// the contraction index of the second-stage products (keys for dQ, queries for dK / dV) is only summed over, so TWO probability tiles - key (query) tiles 2j and 2j + 1,
// each in the C layout "rows (l >> 4)*4 + r" - are laid side by side as one operand: slots e < 4 = tile 2j, e >= 4 = tile 2j + 1, and the other operand is the two
// ds_read_b64_tr_b16 column fragments of the same two tiles concatenated the same way.  The first-stage products (contraction over the 64 head dimensions) read 16-byte
// row fragments.  A missing odd tile (13 = 6 pairs + 1) is a zero half.
__device__ __forceinline__ bf16x8 gb_row_frag8(const bf16_t* X, int row, int k0) { return *reinterpret_cast<const bf16x8*>(X + gb_off(row, k0)); }
__device__ __forceinline__ bf16x8 gb_row_frag8_v(const bf16_t* V, int row, int k0) { return *reinterpret_cast<const bf16x8*>(V + row * GB_LDV + k0); }
__device__ __forceinline__ bf16x8 gb_cat(const bf16x4& lo, const bf16x4& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

template <int GBW, bool X32>
__global__ __launch_bounds__(GBW * 64) void attn_group_bwd_kernel(AttnBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Kr = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vr = Kr + GB_ROWS * GB_LD;                          // (144-byte rows: GB_LDV)
  bf16_t* Qr = Vr + GB_ROWS * GB_LDV;
  bf16_t* Dr = Qr + GB_ROWS * GB_LD;                          // dO
  float* stats = reinterpret_cast<float*>(smem + 3 * GB_MAT + GB_MATV);   // [208][GB_ST]: m (base-2 domain), 1/l, delta
  bf16_t* outl = reinterpret_cast<bf16_t*>(smem + 3 * GB_MAT + GB_MATV + GB_ROWS * GB_ST * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  // the lane's element offsets inside a 16-row tile (gb_off spelled out once, so that tile, k-step and column tile are immediate offsets of ONE address register each):
  // its row-fragment chunk (row lr, chunk lg), the same in V's 144-byte rows, and its ds_read_b64_tr_b16 piece (row lg*4 + lr/4, 8 bytes at (lr & 3) * 4)
  const int l_row = lr * GB_LD + ((lg ^ ((lr >> 3) & 1)) << 3), l_rowv = lr * GB_LDV + lg * 8;
  const int l_tr = (lg * 4 + (lr >> 2)) * GB_LD + (((((lr & 3) >> 1) ^ ((lg >> 1) & 1))) << 3) + (lr & 1) * 4;
  auto row8 = [&](const bf16_t* X, int tile, int ks) { return *reinterpret_cast<const bf16x8*>(X + tile * 16 * GB_LD + l_row + ks * 32); };
  auto row8v = [&](int tile, int ks) { return *reinterpret_cast<const bf16x8*>(Vr + tile * 16 * GB_LDV + l_rowv + ks * 32); };
  auto col4 = [&](const bf16_t* X, int tile, int dt) {
    const gb_s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gb_s4*)(X + tile * 16 * GB_LD + l_tr + dt * 16));
    return __builtin_bit_cast(bf16x4, r);
  };
  const int64_t unit = blockIdx.x;
  const int head = (int)(unit % p.heads);
  const int64_t sg = unit / p.heads;
  const int g = (int)(sg % p.n_groups);
  const int64_t seq = sg / p.n_groups;
  const int64_t seq_base = seq * p.seq_rows;
  const int64_t first = seq_base + p.row0 + (int64_t)g * p.group_stride;
  const int has_cls = p.cls_row >= 0 ? 1 : 0;
  const int nk = p.n_tok + has_cls, nq = p.n_tok;
  const int nkt = (nk + 15) >> 4, nqt = (nq + 15) >> 4;
  const int hcol = head * 64;
  auto key_row = [&](int j) -> int64_t { return (has_cls && j == 0) ? seq_base + p.cls_row : first + (int64_t)(j - has_cls) * p.tok_stride; };
  auto tok_row = [&](int i) -> int64_t { return first + (int64_t)i * p.tok_stride; };
  // The CLS query as query row nq (the host guarantees a free slot: nq % 16 != 0).  Its probabilities are normalised with the forward's statistics over ALL keys of the
  // sequence (this group holds 1/n_groups of them), its delta is <dO, o>; the CLS KEY (key 0 of every group) counts for it in group 0 only.
  const bool clsq = p.cls_stats != nullptr;
  const int nqe = nq + (clsq ? 1 : 0);
  float* cls_d = reinterpret_cast<float*>(outl + GB_OUTW(GBW) * 16 * GB_LDV);   // one float behind the staging tiles
  if (clsq && wave == 0) {
    const int64_t r = seq_base + p.cls_row;
    float d = bf2f(p.dO[r * p.lddo + hcol + lane]) * bf2f(p.o[r * p.ldo + hcol + lane]);
    d = wave_sum(d);
    if (lane == 0) *cls_d = d;
  }

  // ---- stage K, V (key rows) and Q, dO (query rows): 208 rows x 8 chunks of 16 B each, zero beyond the valid rows -------------
  for (int idx = tid; idx < GB_ROWS * 8; idx += GBW * 64) {
    const int row = idx >> 3, ch = idx & 7;
    uint4 kk = make_uint4(0, 0, 0, 0), vv = kk, qq = kk, dd = kk;
    if (!(GB_ABL & 4) && row < nk) {
      const int64_t r = key_row(row);
      kk = *reinterpret_cast<const uint4*>(p.k + r * p.ld + hcol + ch * 8);
      vv = *reinterpret_cast<const uint4*>(p.v + r * p.ld + hcol + ch * 8);
    }
    if (!(GB_ABL & 4) && (row < nq || (clsq && row == nq))) {  // slot nq: the CLS query and its dO
      const int64_t r = row < nq ? tok_row(row) : seq_base + p.cls_row;
      qq = *reinterpret_cast<const uint4*>(p.q + r * p.ld + hcol + ch * 8);
      dd = *reinterpret_cast<const uint4*>(p.dO + r * p.lddo + hcol + ch * 8);
    }
    *reinterpret_cast<uint4*>(Kr + gb_off(row, ch * 8)) = kk;
    *reinterpret_cast<uint4*>(Vr + row * GB_LDV + ch * 8) = vv;
    *reinterpret_cast<uint4*>(Qr + gb_off(row, ch * 8)) = qq;
    *reinterpret_cast<uint4*>(Dr + gb_off(row, ch * 8)) = dd;
  }
  __syncthreads();
  const float sc2 = p.scale * 1.44269504088896f;
  bf16_t* myout = outl + wave * 16 * GB_LDV;

  // ---- pass 1: per query tile - statistics and dQ.  Tiles are S^T: lane's query = qt*16 + lr, its keys = kt*16 + lg*4 + r ---------
  for (int qt = wave; qt < nqt; qt += GBW) {
    // Online softmax over the key tiles (running maximum, accumulators rescaled when it moves): no score or dP tile outlives its key tile, and
    // dQ = scale / l * (sum_k e dP K - delta sum_k e K) is accumulated as those two sums (e = un-normalised probability) - 4 more MFMAs per key tile than
    // forming dS first, but ~90 registers instead of 167, which is what lets 16 waves (every query / key tile its own wave) share the CU.
    float m = -INFINITY, l = 0.f, delta = 0.f;                    // m is kept equal in the four lanes (lg) of a query column; l, delta are per-lane partials
    f32x4 dq1[4], dq2[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dq1[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dq2[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if constexpr (!X32) {
      bf16x4 qf[4], df[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { qf[ks] = gb_row_frag(Qr, qt * 16 + lr, ks * 16 + lg * 4); df[ks] = gb_row_frag(Dr, qt * 16 + lr, ks * 16 + lg * 4); }
      for (int kt = 0; kt < nkt; ++kt) {
        f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          sc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gb_row_frag(Kr, kt * 16 + lr, ks * 16 + lg * 4), qf[ks], sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gb_row_frag_v(Vr, kt * 16 + lr, ks * 16 + lg * 4), df[ks], dp, 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kt * 16 + lg * 4 + r >= nk) sc[r] = -INFINITY;
          tmax = fmaxf(tmax, sc[r]);
        }
        if (clsq && g != 0 && kt == 0 && lg == 0 && qt * 16 + lr == nq) { sc[0] = -INFINITY; tmax = fmaxf(fmaxf(sc[1], sc[2]), sc[3]); }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);                          // finite: key 0 of tile 0 always exists
        if (__any(m_new != m)) {                                     // the running maximum of some query moved: rescale what has been summed under the old one
          const float alpha = __builtin_amdgcn_exp2f((m - m_new) * sc2);   // first tile: exp2(-inf) = 0
          l *= alpha; delta *= alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) {                              // the accumulators hold query ROWS lg*4 + r: their factor sits in the lanes of that query column
            const float ar = __shfl(alpha, lg * 4 + r, 64);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dq1[dt][r] *= ar; dq2[dt][r] *= ar; }
          }
          m = m_new;
        }
        const float msc_ = m * sc2;
        f32x4 e, edp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          e[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], sc2, -msc_));     // -inf -> 0
          edp[r] = e[r] * dp[r];
          l += e[r]; delta += edp[r];
        }
        const bf16x4 ef = gb_pack(e), edpf = gb_pack(edp);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x4 kc = col4(Kr, kt, dt);
          dq1[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(edpf, kc, dq1[dt], 0, 0, 0);
          dq2[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ef, kc, dq2[dt], 0, 0, 0);
        }
      }
    } else {
      bf16x8 qf[2], df[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { qf[ks] = row8(Qr, qt, ks); df[ks] = row8(Dr, qt, ks); }
      const bf16x4 zero4 = {0, 0, 0, 0};
      for (int kt = 0; kt < ((GB_ABL & 1) ? 0 : nkt); kt += 2) {
        const bool two = kt + 1 < nkt;                             // wave-uniform
        f32x4 sc[2], dp[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) { sc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          sc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row8(Kr, kt, ks), qf[ks], sc[0], 0, 0, 0);
          dp[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row8v(kt, ks), df[ks], dp[0], 0, 0, 0);
        }
        if (two) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            sc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row8(Kr, kt + 1, ks), qf[ks], sc[1], 0, 0, 0);
            dp[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row8v(kt + 1, ks), df[ks], dp[1], 0, 0, 0);
          }
        }
        float tmax = -INFINITY;
        if ((kt + 2) * 16 > nk || (clsq && g != 0 && kt == 0)) {      // wave-uniform: only the last key pair and the CLS key's tile hold anything to mask
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if ((h == 1 && !two) || (kt + h) * 16 + lg * 4 + r >= nk) sc[h][r] = -INFINITY;
          if (clsq && g != 0 && kt == 0 && lg == 0 && qt * 16 + lr == nq) sc[0][0] = -INFINITY;   // the CLS key counts for the CLS query in group 0 only
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sc[h][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);                        // finite: key 0 of tile 0 always exists (for the CLS query of a group > 0: key 1)
        if (__any(m_new != m)) {
          const float alpha = __builtin_amdgcn_exp2f((m - m_new) * sc2);
          l *= alpha; delta *= alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha, lg * 4 + r, 64);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dq1[dt][r] *= ar; dq2[dt][r] *= ar; }
          }
          m = m_new;
        }
        const float msc_ = m * sc2;
        f32x4 e[2], edp[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            e[h][r] = __builtin_amdgcn_exp2f(fmaf(sc[h][r], sc2, -msc_));
            edp[h][r] = e[h][r] * dp[h][r];
            l += e[h][r]; delta += edp[h][r];
          }
        const bf16x8 ef = gb_cat(gb_pack(e[0]), gb_pack(e[1])), edpf = gb_cat(gb_pack(edp[0]), gb_pack(edp[1]));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 kc = gb_cat(col4(Kr, kt, dt), two ? col4(Kr, kt + 1, dt) : zero4);
          dq1[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(edpf, kc, dq1[dt], 0, 0, 0);
          dq2[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, kc, dq2[dt], 0, 0, 0);
        }
      }
    }
    const float msc = m * sc2;
    l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
    delta += __shfl_xor(delta, 16, 64); delta += __shfl_xor(delta, 32, 64);
    float linv = 1.0f / l;
    delta *= linv;
    float st_m = msc, st_linv = linv;
    if (clsq && qt * 16 + lr == nq) {                            // the CLS query: global statistics; the sums above were taken under this group's running maximum
      const float* gs = p.cls_stats + (seq * p.heads + head) * 2;
      st_m = gs[0]; st_linv = 1.0f / gs[1];
      linv = __builtin_amdgcn_exp2f(msc - st_m) * st_linv;
      delta = *cls_d;
    }
    if (X32 && qt * 16 + lr >= nqe) { st_m = INFINITY; st_linv = 0.f; }   // an empty query slot: p = exp2(s - inf) * 0 = 0 in pass 2 without a per-element test (delta is 0: its dO row is)
    if (lg == 0) *reinterpret_cast<float4*>(stats + (qt * 16 + lr) * GB_ST) = make_float4(st_m, st_linv, (X32 && qt * 16 + lr >= nqe) ? 0.f : delta, 0.f);
    // l and delta belong to the lane's query COLUMN (qt*16 + lr); the dQ accumulators hold query ROWS lg*4 + r: fetch the row's values from its column lane
    f32x4 dq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float lrow = __shfl(linv, lg * 4 + r, 64), drow = __shfl(delta, lg * 4 + r, 64);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt][r] = (dq1[dt][r] - drow * dq2[dt][r]) * (lrow * p.scale);
    }
    // dq[dt][r] = dQ[query qt*16 + lg*4 + r][d = dt*16 + lr] -> staging -> 16-byte row stores
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) myout[(lg * 4 + r) * GB_LDV + dt * 16 + lr] = f2bf(dq[dt][r]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = lane + i * 64, row = idx >> 3, ch = idx & 7, qi = qt * 16 + row;
      if (qi < nqe) {
        const uint2* sp = reinterpret_cast<const uint2*>(myout + row * GB_LDV + ch * 8);
        const uint2 a = sp[0], b = sp[1];
        bf16_t* dst = qi < nq ? p.dq + tok_row(qi) * p.ldg + hcol + ch * 8
                              : p.dq_cls_part + (seq * p.n_groups + g) * (int64_t)(p.heads * 64) + hcol + ch * 8;   // this group's share of the CLS query's dq
        *reinterpret_cast<uint4*>(dst) = make_uint4(a.x, a.y, b.x, b.y);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();                                             // statistics of every query are in LDS

  // ---- pass 2: per key tile - dK, dV.  Tiles are S: lane's key = kt*16 + lr, its queries = qt*16 + lg*4 + r ----------------------
  for (int kt = wave; kt < nkt; kt += GBW) {
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const bool key_ok = kt * 16 + lr < nk;
    if constexpr (!X32) {
      bf16x4 kf[4], vf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { kf[ks] = gb_row_frag(Kr, kt * 16 + lr, ks * 16 + lg * 4); vf[ks] = gb_row_frag_v(Vr, kt * 16 + lr, ks * 16 + lg * 4); }
      for (int qt = 0; qt < nqt; ++qt) {
        f32x4 s2 = f32x4{0.f, 0.f, 0.f, 0.f}, dp2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gb_row_frag(Qr, qt * 16 + lr, ks * 16 + lg * 4), kf[ks], s2, 0, 0, 0);
          dp2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gb_row_frag(Dr, qt * 16 + lr, ks * 16 + lg * 4), vf[ks], dp2, 0, 0, 0);
        }
        f32x4 pp, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = qt * 16 + lg * 4 + r;
          const float* st = stats + qi * GB_ST;
          const bool live = key_ok && qi < nqe && !(clsq && g != 0 && qi == nq && kt * 16 + lr == 0);
          const float pr = live ? __builtin_amdgcn_exp2f(fmaf(s2[r], sc2, -st[0])) * st[1] : 0.f;
          pp[r] = pr;
          ds[r] = pr * (dp2[r] - st[2]) * p.scale;
        }
        const bf16x4 pf = gb_pack(pp), dsf = gb_pack(ds);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pf, col4(Dr, qt, dt), dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dsf, col4(Qr, qt, dt), dk[dt], 0, 0, 0);
        }
      }
    } else {
      bf16x8 kf[2], vf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { kf[ks] = row8(Kr, kt, ks); vf[ks] = row8v(kt, ks); }
      const bf16x4 zero4 = {0, 0, 0, 0};
      const bool fix_keys = kt * 16 + 16 > nk || (clsq && g != 0 && kt == 0);       // wave-uniform
      for (int qt = 0; qt < ((GB_ABL & 2) ? 0 : nqt); qt += 2) {
        const bool two = qt + 1 < nqt;                             // wave-uniform
        // p = exp2(s c - m) / l and ds = p (dp - delta) (the softmax scale is applied to dK once, below), one query tile after the other (a tile's scores are packed to
        // bf16 before the next tile's are formed: 8 accumulator registers live instead of 16).  Empty query slots carry (m = inf, 1/l = 0) and give p = 0 by themselves;
        // only the last key tile (keys beyond nk) and the CLS key's tile (CLS query x CLS key outside group 0) need a per-element test
        bf16x4 pk[2], dsk[2];
        pk[1] = zero4; dsk[1] = zero4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 1 && !two) break;
          const int qh = qt + h;
          f32x4 s2 = f32x4{0.f, 0.f, 0.f, 0.f}, dp2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row8(Qr, qh, ks), kf[ks], s2, 0, 0, 0);
            dp2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row8(Dr, qh, ks), vf[ks], dp2, 0, 0, 0);
          }
          f32x4 pp, ds;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qh * 16 + lg * 4 + r;
            const float4 st = *reinterpret_cast<const float4*>(stats + qi * GB_ST);
            float pr = __builtin_amdgcn_exp2f(fmaf(s2[r], sc2, -st.x)) * st.y;
            if (fix_keys && (!key_ok || (clsq && g != 0 && qi == nq && kt * 16 + lr == 0))) pr = 0.f;
            pp[r] = pr;
            ds[r] = pr * (dp2[r] - st.z);
          }
          pk[h] = gb_pack(pp); dsk[h] = gb_pack(ds);
        }
        const bf16x8 pf = gb_cat(pk[0], pk[1]), dsf = gb_cat(dsk[0], dsk[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 dcol = gb_cat(col4(Dr, qt, dt), two ? col4(Dr, qt + 1, dt) : zero4);
          const bf16x8 qcol = gb_cat(col4(Qr, qt, dt), two ? col4(Qr, qt + 1, dt) : zero4);
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, dcol, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, qcol, dk[dt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dk[dt] *= p.scale;
    }
    // dk[dt][r] = dK[key kt*16 + lg*4 + r][d = dt*16 + lr]: two staging rounds (dk, then dv)
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) myout[(lg * 4 + r) * GB_LDV + dt * 16 + lr] = f2bf(which == 0 ? dk[dt][r] : dv[dt][r]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = lane + i * 64, row = idx >> 3, ch = idx & 7, kj = kt * 16 + row;
        if (kj < nk) {
          const uint2* sp = reinterpret_cast<const uint2*>(myout + row * GB_LDV + ch * 8);
          const uint2 a = sp[0], b = sp[1];
          bf16_t* dst;
          if (has_cls && kj == 0) dst = p.cls_part + (seq * p.n_groups + g) * (int64_t)(2 * p.heads * 64) + which * p.heads * 64 + hcol + ch * 8;
          else dst = (which == 0 ? p.dk : p.dv) + key_row(kj) * p.ldg + hcol + ch * 8;
          *reinterpret_cast<uint4*>(dst) = make_uint4(a.x, a.y, b.x, b.y);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

static int attention_group_bwd_impl(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dO, int64_t lddo, bf16_t* dq,
                                    bf16_t* dk, bf16_t* dv, int64_t ldg, bf16_t* cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0,
                                    int group_stride, int tok_stride, int n_tok, int cls_row, int heads, int head_dim, float scale, const float* cls_stats,
                                    const bf16_t* o, int64_t ldo, bf16_t* dq_cls_part, void* stream);

extern "C" int sf_attention_group_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dO, int64_t lddo, bf16_t* dq,
                                      bf16_t* dk, bf16_t* dv, int64_t ldg, bf16_t* cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0,
                                      int group_stride, int tok_stride, int n_tok, int cls_row, int heads, int head_dim, float scale, void* stream) {
  return attention_group_bwd_impl(q, k, v, ld, dO, lddo, dq, dk, dv, ldg, cls_part, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads,
                                  head_dim, scale, nullptr, nullptr, 0, nullptr, stream);
}

// sf_attention_group_bwd + the backward of the CLS QUERY (the row sf_attention_cls_bwd handles: it attends all keys of the sequence) in the same launch: the CLS query
// is one more query row of every group, normalised with the forward's softmax statistics cls_stats [n_seq][heads][2] (m in the base-2 domain incl. the scale, l -
// sf_attention_cls_combine_stats) and delta = <dO, o> from the forward's output row `o` (row seq * seq_rows + cls_row, stride ldo).  dk / dv of every row then hold
// BOTH contributions (no read-modify-write pass over them), cls_part includes the CLS query's share of the CLS key, and the CLS query's dq comes out as one partial row
// per group: dq_cls_part (n_seq * n_groups, heads * 64) bf16, to be summed into row cls_row of dq by sf_reduce_groups_bf16.  Needs n_tok % 16 != 0.
extern "C" int sf_attention_group_bwd_clsq(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dO, int64_t lddo, bf16_t* dq,
                                           bf16_t* dk, bf16_t* dv, int64_t ldg, bf16_t* cls_part, const float* cls_stats, const bf16_t* o, int64_t ldo,
                                           bf16_t* dq_cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride,
                                           int n_tok, int cls_row, int heads, int head_dim, float scale, void* stream) {
  SF_CHECK_ARG(cls_stats && o && dq_cls_part && cls_row >= 0 && (n_tok % 16) != 0 && (ldo % 8) == 0,
               "sf_attention_group_bwd_clsq: needs statistics, the forward output, a partial buffer, a CLS row and a free query slot (n_tok %% 16 != 0)");
  return attention_group_bwd_impl(q, k, v, ld, dO, lddo, dq, dk, dv, ldg, cls_part, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads,
                                  head_dim, scale, cls_stats, o, ldo, dq_cls_part, stream);
}

static int attention_group_bwd_impl(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dO, int64_t lddo, bf16_t* dq,
                                    bf16_t* dk, bf16_t* dv, int64_t ldg, bf16_t* cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0,
                                    int group_stride, int tok_stride, int n_tok, int cls_row, int heads, int head_dim, float scale, const float* cls_stats,
                                    const bf16_t* o, int64_t ldo, bf16_t* dq_cls_part, void* stream) {
  SF_CHECK_ARG(q && k && v && dO && dq && dk && dv, "sf_attention_group_bwd: null pointer");
  SF_CHECK_ARG(head_dim == 64 && n_tok >= 1 && n_tok + (cls_row >= 0 ? 1 : 0) <= GB_ROWS && (cls_row < 0 || cls_part),
               "sf_attention_group_bwd: head_dim 64, n_tok (+1) <= 208, cls_part with cls_row");
  SF_CHECK_ARG((ld % 8) == 0 && (lddo % 8) == 0 && (ldg % 8) == 0 && n_groups >= 1 && heads >= 1, "sf_attention_group_bwd: bad strides / counts");
  if (n_seq <= 0) return 0;
  AttnBwdArgs a;
  a.q = q; a.k = k; a.v = v; a.ld = ld; a.dO = dO; a.lddo = lddo; a.dq = dq; a.dk = dk; a.dv = dv; a.ldg = ldg; a.cls_part = cls_part;
  a.seq_rows = seq_rows; a.n_groups = n_groups; a.row0 = row0; a.group_stride = group_stride; a.tok_stride = tok_stride; a.n_tok = n_tok;
  a.cls_row = cls_row; a.heads = heads; a.scale = scale;
  a.cls_stats = cls_stats; a.o = o; a.ldo = ldo; a.dq_cls_part = dq_cls_part;
  const int64_t units = n_seq * n_groups * heads;
  SF_CHECK_ARG(units < ((int64_t)1 << 31), "sf_attention_group_bwd: too many groups");
  static int waves = -1;
  if (waves < 0) { const char* e = getenv("SF_GB_WAVES"); waves = e ? atoi(e) : GB_WAVES; }      // measurement hook: 8 = round 2's workgroup (two tiles per wave)
  static int x32 = -1;
  if (x32 < 0) { const char* e = getenv("SF_GB_X32"); x32 = e ? atoi(e) : 1; }                     // measurement hook: 0 = round 3's v_mfma_f32_16x16x16_bf16 passes
  if (waves == 8) {
    if (int rc = sf_prepare_kernel((const void*)attn_group_bwd_kernel<8, false>, GB_LDS(8), "sf_attention_group_bwd")) return rc;
    hipLaunchKernelGGL((attn_group_bwd_kernel<8, false>), dim3((unsigned)units), dim3(8 * 64), GB_LDS(8), (hipStream_t)stream, a);
  } else if (!x32) {
    if (int rc = sf_prepare_kernel((const void*)attn_group_bwd_kernel<GB_WAVES, false>, GB_LDS(GB_WAVES), "sf_attention_group_bwd")) return rc;
    hipLaunchKernelGGL((attn_group_bwd_kernel<GB_WAVES, false>), dim3((unsigned)units), dim3(GB_WAVES * 64), GB_LDS(GB_WAVES), (hipStream_t)stream, a);
  } else {
    if (int rc = sf_prepare_kernel((const void*)attn_group_bwd_kernel<GB_WAVES, true>, GB_LDS(GB_WAVES), "sf_attention_group_bwd")) return rc;
    hipLaunchKernelGGL((attn_group_bwd_kernel<GB_WAVES, true>), dim3((unsigned)units), dim3(GB_WAVES * 64), GB_LDS(GB_WAVES), (hipStream_t)stream, a);
  }
  SF_LAUNCH_CHECK();
  return 0;
}

// ======================================================================================================
// (2) one query row per sequence against n_keys rows (D = 64): the Motionformer CLS query (1569 keys) and the
// aggregator layers, whose encoder output is only ever read at row 0 (motionformer.py:332, ast.py:274-277).
// One 256-thread workgroup per (seq, head); lane = (key slot = lane>>3, slice = lane&7); 32 keys in flight per
// step, online softmax per slot, slots merged by shuffles then through LDS.  K and V are each read once.
// ======================================================================================================
struct ClsArgs {
  const bf16_t* q; int64_t q_seq_rows;        // query row of sequence s = s * q_seq_rows + q_row
  int q_row;
  const bf16_t* k; const bf16_t* v; int64_t ld;
  int64_t kv_seq_rows; int kv_row0, n_keys;
  bf16_t* out; int64_t ldo; int64_t out_seq_rows; int out_row;
  int heads; float scale;
  const uint8_t* key_keep;   // optional, indexed by K/V row
  float* stats = nullptr;    // optional: [seq][head][2] = (softmax maximum in the base-2 domain incl. the scale, sum) - the backward's statistics (sf_attention_*_bwd_clsq)
};

__global__ __launch_bounds__(256) void attn_cls64_kernel(ClsArgs p) {
  __shared__ float red[4][8][10];   // [wave][slice][m, l, acc0..7]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = blockIdx.x % p.heads;
  const int64_t seq = blockIdx.x / p.heads;
  const int slot = lane >> 3, sub = lane & 7;
  const int col = head * 64 + sub * 8;
  float qf[8];
  unpack8(*reinterpret_cast<const uint4*>(p.q + (seq * p.q_seq_rows + p.q_row) * p.ld + col), qf);
  const int64_t kv0 = seq * p.kv_seq_rows + p.kv_row0;
  float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = wave * 8 + slot; j < p.n_keys; j += 32) {
    if (p.key_keep && p.key_keep[kv0 + j] == 0) continue;        // masked key: contributes nothing (the 8 lanes of a slot agree)
    float kf[8], vf[8];
    unpack8(*reinterpret_cast<const uint4*>(p.k + (kv0 + j) * p.ld + col), kf);
    unpack8(*reinterpret_cast<const uint4*>(p.v + (kv0 + j) * p.ld + col), vf);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += qf[e] * kf[e];
    d = sum8_dpp(d);
    const float s = d * p.scale;
    const float mn = fmaxf(m, s);
    const float corr = __expf(m - mn), e = __expf(s - mn);   // m = -inf first time: corr = 0
    l = l * corr + e;
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = acc[t] * corr + e * vf[t];
    m = mn;
  }
  // merge the 8 key slots of this wave (lanes differing in bits 3..5)
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    const float mo = __shfl_xor(m, off, 64), lo = __shfl_xor(l, off, 64);
    const float mn = fmaxf(m, mo);
    const float c0 = (m == -INFINITY) ? 0.f : __expf(m - mn), c1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
    l = l * c0 + lo * c1;
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = acc[t] * c0 + __shfl_xor(acc[t], off, 64) * c1;
    m = mn;
  }
  if (slot == 0) {
    red[wave][sub][0] = m; red[wave][sub][1] = l;
#pragma unroll
    for (int t = 0; t < 8; ++t) red[wave][sub][2 + t] = acc[t];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int sb = threadIdx.x;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, red[w][sb][0]);
    float L = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = red[w][sb][0];
      const float c = (mw == -INFINITY) ? 0.f : __expf(mw - M);
      L += red[w][sb][1] * c;
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] += red[w][sb][2 + t] * c;
    }
    const float inv = 1.0f / L;
    uint4 wv;
    wv.x = pack_bf2(o[0] * inv, o[1] * inv); wv.y = pack_bf2(o[2] * inv, o[3] * inv);
    wv.z = pack_bf2(o[4] * inv, o[5] * inv); wv.w = pack_bf2(o[6] * inv, o[7] * inv);
    *reinterpret_cast<uint4*>(p.out + (seq * p.out_seq_rows + p.out_row) * p.ldo + head * 64 + sb * 8) = wv;
    if (p.stats && sb == 0) { p.stats[(int64_t)blockIdx.x * 2] = M * 1.44269504088896f; p.stats[(int64_t)blockIdx.x * 2 + 1] = L; }
  }
}

// ======================================================================================================
// (3) MFMA grouped attention, n_tok (+1) <= 208 keys, D in {64, 96}.  One 256-thread workgroup per
// (seq, group, head).  K (row-major) and V^T (key-major pairs) are staged once in LDS; each wave owns 16-query
// tiles.  Scores are computed TRANSPOSED, S^T = K Q^T (A = K fragment from LDS, B = Q fragment from HBM), so
// that after v_mfma_f32_16x16x32_bf16 every lane holds, for ITS query column (lane & 15), the scores of keys
// kt*16 + (lane>>4)*4 + r: the row softmax is then in-lane + two xor-shuffles, and the exponentiated
// probabilities are already laid out as the A operand (query x key) of the P V product - no LDS round trip for
// P.  The P V contraction uses a permuted key order per 32-key step (slot g*8+i <-> key 16*(i>>2) + 4g + (i&3));
// the V^T fragments are read from LDS with the same permutation (two 8-byte reads per fragment).
// ======================================================================================================
#define ATT_MAX_KT 13            // 13 x 16 = 208 keys / queries max
#ifndef SF_ATT_ABL
#define SF_ATT_ABL 0             // tools/ab_pp.sh (SRC=sf_attention) + tools/ab_att_run.sh only: 1 skip stores, 2 skip P V, 4 skip softmax, 8 one query tile per wave, 16 / 32 one K / V fragment read for all key tiles
#endif

typedef short att_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) att_s4 att_lds_s4;

template <int D>
struct AttLds {
  static constexpr int K_LD = (D == 64) ? 128 : (D * 2 + 16);   // bytes per K row (D=64: XOR-swizzled 128 B rows)
  static constexpr int K_BYTES = 208 * K_LD;
  static constexpr int V_BYTES = K_BYTES;                         // V is staged row-major exactly like K; the P V operand is read with ds_read_b64_tr_b16
  static constexpr int MASK_BYTES = 208;                          // one keep-flag byte per key slot (used only with a key mask)
  static constexpr int TOTAL = K_BYTES + V_BYTES + MASK_BYTES;    // D = 64: 53,456 B -> three workgroups per CU
};

template <int D>
__device__ __forceinline__ int k_lds_off(int row, int chunk) {   // byte offset of 16-B chunk `chunk` of K row `row`
  if (D == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
  return row * AttLds<D>::K_LD + (chunk << 4);
}

#ifndef SF_ATT_LD_NT
#define SF_ATT_LD_NT 1   // K / V rows with the nt hint: each is read by exactly one workgroup (+0.25 % on the step, interleaved A/B)
#endif
__device__ __forceinline__ uint4 att_ld16(const bf16_t* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned int att_u32x4;
  if (SF_ATT_LD_NT) { const att_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const att_u32x4*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
  return *reinterpret_cast<const uint4*>(p);
}

template <int D, int NKT, bool MXO = false>
__global__ __launch_bounds__(256, 3) void attn_mfma_kernel(AttnArgs p) {
  static_assert(!MXO || D == 64, "the MXFP8 output is written per 64-wide head");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_lds = smem;
  char* v_lds = smem + AttLds<D>::K_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x % p.heads;
  const int64_t sg = blockIdx.x / p.heads;
  const int g = (int)(sg % p.n_groups);
  const int64_t seq = sg / p.n_groups;
  const int64_t seq_base = seq * p.seq_rows;
  const int64_t first = seq_base + p.row0 + (int64_t)g * p.group_stride;
  const int has_cls = p.cls_row >= 0 ? 1 : 0;
  const int nk = p.n_tok + has_cls, nq = p.n_tok;
  constexpr int nkt = NKT;                                       // key tiles (host guarantees ceil(nk/16) == NKT)
  const bool do_cls = p.cls_part != nullptr;                      // host guarantees nq % 16 != 0 (a free query slot) and cls_row >= 0
  const int nqt = (nq + 15) >> 4;
  const int hcol = head * D;
  constexpr int CH = D / 8;          // 16-byte chunks per row

  auto key_row = [&](int j) -> int64_t {
    return (has_cls && j == 0) ? seq_base + p.cls_row : first + (int64_t)(j - has_cls) * p.tok_stride;
  };
  // Addressing: everything this workgroup touches lies inside ONE sequence, so rows are addressed as 32-bit element offsets from the
  // (wave-uniform, scalar) sequence base - one 24-bit multiply per row instead of 64-bit multiply-adds (the prologue was ~30 % of the
  // VALU time of a wave, a quarter of it quarter-rate integer multiplies).  The host checks seq_rows * ld < 2^31 and ld < 2^24.
  const bf16_t* qb = p.q + seq_base * p.ld + hcol;
  const bf16_t* kb = p.k + seq_base * p.ld + hcol;
  const bf16_t* vb_ = p.v + seq_base * p.ld + hcol;
  const uint32_t ld32 = (uint32_t)p.ld;
  const int rel_first = p.row0 + g * p.group_stride;              // first token row of the group, relative to the sequence
  auto key_off = [&](int j) -> uint32_t {                         // element offset of key/value row j (callers clamp j to nk - 1)
    const int r = (has_cls && j == 0) ? p.cls_row : rel_first + __mul24(j - has_cls, p.tok_stride);
    return __umul24((uint32_t)r, ld32);
  };

  // ---- all global loads of this workgroup are issued up front (Q fragments of this wave's query tiles, then the
  // K rows and V row-pairs this thread stages), THEN written to LDS: one memory latency per workgroup instead of one per
  // loop iteration (the first version's runtime-trip-count staging loops serialised load -> wait -> ds_write).
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int MAXQ = (NKT + 3) / 4;                             // query tiles per wave (nq <= nk)
  // Query tile qt goes to wave (qt - rot) & 3.  With 13 tiles one wave gets four and the others three; wave w of every workgroup sits on
  // SIMD w, so without the rotation SIMD 0 of each CU would carry 4/3 of the work of the others.  rot differs between workgroups that
  // share a CU whichever way the dispatcher fills it (neighbouring ids of one XCD, or ids 32 apart).
  const int xw = blockIdx.x >> 3;
  const int wq = (wave + xw + (xw >> 5)) & 3;
  bf16x8 qf[MAXQ][D / 32];
#pragma unroll
  for (int t = 0; t < MAXQ; ++t) {
    const int qt = wq + 4 * t;
    int qi = qt * 16 + fr;
    const bool is_cls_q = do_cls && qi == nq;                      // the free slot right after the last query holds the CLS query
    if (qi > nq - 1) qi = nq - 1;                                  // clamp (also for tiles beyond nqt: harmless reload)
    const uint32_t qoff = __umul24((uint32_t)(is_cls_q ? p.cls_row : rel_first + __mul24(qi, p.tok_stride)), ld32) + fg * 8;
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) qf[t][ks] = *reinterpret_cast<const bf16x8*>(qb + qoff + ks * 32);
  }
  constexpr int K_IT = (NKT * 16 * CH + 255) / 256;               // 16-B K chunks per thread
  uint4 kreg[K_IT];
#pragma unroll
  for (int it = 0; it < K_IT; ++it) {
    const int idx = tid + it * 256, row = idx / CH, ch = idx - row * CH;
    kreg[it] = make_uint4(0, 0, 0, 0);
    if (row < nk) kreg[it] = att_ld16(kb + key_off(row) + ch * 8);
  }
  // V rows are fetched and staged exactly like K rows (whole 128-byte rows per 8 lanes; zero beyond nk, because P = 0 there must meet
  // finite values).  The first version transposed V in registers (each wave fetched a 32-byte slice of every row - four 32-byte
  // requests per cache line - and wrote V^T with 64 ds_write_b32 per thread).
  uint4 vreg[K_IT];
#pragma unroll
  for (int it = 0; it < K_IT; ++it) {
    const int idx = tid + it * 256, row = idx / CH, ch = idx - row * CH;
    vreg[it] = make_uint4(0, 0, 0, 0);
    if (row < nk) vreg[it] = att_ld16(vb_ + key_off(row) + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < K_IT; ++it) {
    const int idx = tid + it * 256, row = idx / CH, ch = idx - row * CH;
    if (row < nkt * 16) *reinterpret_cast<uint4*>(k_lds + k_lds_off<D>(row, ch)) = kreg[it];
  }
  uint8_t* m_lds = reinterpret_cast<uint8_t*>(smem + AttLds<D>::K_BYTES + AttLds<D>::V_BYTES);
  if (p.key_keep && tid < 208) m_lds[tid] = tid < nk ? p.key_keep[key_row(tid)] : (uint8_t)0;
#pragma unroll
  for (int it = 0; it < K_IT; ++it) {
    const int idx = tid + it * 256, row = idx / CH, ch = idx - row * CH;
    if (row < nkt * 16) *reinterpret_cast<uint4*>(v_lds + k_lds_off<D>(row, ch)) = vreg[it];
  }
  // P V operand: the A fragment of v_mfma_f32_16x16x32_bf16 is V^T (16 dims x 32 key slots), lane (fr, fg) holding dim dt*16 + fr of key
  // slots fg*8 .. +7 = keys 32 kk + 4 fg + {0..3} and 32 kk + 16 + 4 fg + {0..3} (the order P comes out of the S^T tiles in).
  // ds_read_b64_tr_b16 transposes a 4 x 16 block inside each 16-lane group: lane i receives element (i & 3) of lanes (i >> 2) + 4 j,
  // j = 0..3, so lane (fr, fg) points at the 8 bytes V[key 32 kk + 4 fg + (fr >> 2)][dt*16 + 4 (fr & 3) .. +3] and gets back
  // V[32 kk + 4 fg + 0..3][dt*16 + fr].  (key & 7) is lane-constant, so the swizzled chunk offsets are computed once per dim tile.
  int v_off[D / 16];
  {
    const int krow = fg * 4 + (fr >> 2), c1 = (fr & 3) >> 1, hb = (fr & 1) * 8;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) v_off[dt] = k_lds_off<D>(krow, dt * 2 + c1) + hb;
  }
  __syncthreads();

#pragma unroll
  for (int t = 0; t < MAXQ; ++t) {
    const int qt = wq + 4 * t;
    if (qt >= nqt || ((SF_ATT_ABL & 8) && t >= 1)) break;

    // ---- S^T tiles -------------------------------------------------------------------------------------
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + k_lds_off<D>(((SF_ATT_ABL & 16) ? 0 : kt * 16) + fr, ks * 4 + fg));   // (ABL 16: every key tile reads tile 0's fragment - the compiler keeps one read: what do the K fragment reads cost?)
          s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ks], s[kt], 0, 0, 0);
        }
      }
    }
    // ---- softmax (base 2, scale folded with log2 e) over keys for query column (lane & 15) ---------------------
    const float sc2 = p.scale * 1.44269504088896f;
    const bool cls_slot = do_cls && (qt * 16 + fr == nq);            // this lane's query column is the CLS query
    // Masking touches only what can be invalid: the last key tile (keys >= nk), key 0 for the CLS-query slot outside group 0, and -
    // on the masked entry points only - the token keep flags.  The scale is folded into the exponent: exp2(s*sc2 - m*sc2), so the
    // row maximum is taken on the raw scores (sc2 > 0) and each score costs max + fma + v_exp + add.
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if ((NKT - 1) * 16 + fg * 4 + r >= nk) s[NKT - 1][r] = -INFINITY;
    if (cls_slot && g != 0 && fg == 0) s[0][0] = -INFINITY;
    if (p.key_keep) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const uint32_t fl = *reinterpret_cast<const uint32_t*>(m_lds + kt * 16 + fg * 4);   // keep flags of keys kt*16 + fg*4 + 0..3
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (((fl >> (8 * r)) & 0xffu) == 0) s[kt][r] = -INFINITY;
      }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float msc = m * sc2;
    const float msafe = m == -INFINITY ? 0.f : msc;                  // every key of this query masked (the CLS query's slot of a fully masked group): zeros, not NaN
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (SF_ATT_ABL & 4) ? s[kt][r] : __builtin_amdgcn_exp2f(fmaf(s[kt][r], sc2, -msafe));   // -inf -> 0
        s[kt][r] = e; l += e;
      }
    }
    l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
    const float linv = 1.0f / l;

    // ---- O = P V ---------------------------------------------------------------------------------------
    f32x4 o[D / 16];
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < ((SF_ATT_ABL & 2) ? 1 : (NKT + 1) / 2); ++kk) {
      {
        // A operand: slots 0..3 <- tile 2kk regs, slots 4..7 <- tile 2kk+1 regs (zero beyond nkt)
        union { bf16x8 v; uint32_t u[4]; } pa;
        pa.u[0] = pack_bf2(s[2 * kk][0], s[2 * kk][1]);
        pa.u[1] = pack_bf2(s[2 * kk][2], s[2 * kk][3]);
        if (2 * kk + 1 < NKT) {
          const int t1 = (2 * kk + 1 < NKT) ? 2 * kk + 1 : 0;
          pa.u[2] = pack_bf2(s[t1][0], s[t1][1]);
          pa.u[3] = pack_bf2(s[t1][2], s[t1][3]);
        } else { pa.u[2] = 0; pa.u[3] = 0; }
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
          union { bf16x8 v; att_s4 h[2]; } vb;
          vb.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4*)(v_lds + v_off[dt] + ((SF_ATT_ABL & 32) ? 0 : kk * 32) * AttLds<D>::K_LD));
          vb.h[1] = att_s4{0, 0, 0, 0};
          if (2 * kk + 1 < NKT)                                                       // no 14th key tile: P is zero there, skip the read
            vb.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4*)(v_lds + v_off[dt] + (((SF_ATT_ABL & 32) ? 0 : kk * 32) + 16) * AttLds<D>::K_LD));
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb.v, pa.v, o[dt], 0, 0, 0);   // O^T tile: (d, query)
        }
      }
    }
    // ---- normalise + store.  The product was formed transposed (A = V^T fragment, B = P - the same registers serve
    // as either operand), so o[dt][r] is (dim dt*16 + fg*4 + r, query qt*16 + fr): each lane owns 4 consecutive dims of
    // ITS query row -> its own 1/l, and one 8-byte store per 16-dim tile.
    const int qo = qt * 16 + fr;
    if (cls_slot) {                                              // unnormalised partial of the CLS query over this group's keys
      float* part = p.cls_part + ((seq * p.heads + head) * p.n_groups + g) * (D + 2);
      if (fg == 0) { part[0] = msc; part[1] = l; }
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[2 + dt * 16 + fg * 4 + r] = o[dt][r];
    }
    if constexpr (MXO) {
      // MXFP8 output: the head's 64 dims are two scale blocks; block b = dim tiles 2b, 2b + 1, 8 values in this lane and 8 in each of the lanes fg' != fg of
      // the same query (16 and 32 lanes away).  Quantised from the bf16-rounded value, exactly as sf_quantize_mxfp8 would from the bf16 output.  Lane pairs
      // (fg, fg ^ 1) swap one dword so that every lane stores 8 contiguous bytes; lane fg = 0 stores the query's two scale bytes.
      const uint32_t row = (uint32_t)(rel_first + __mul24(qo < nq ? qo : nq - 1, p.tok_stride));
      uint8_t* qrow = p.out_q + (seq_base + row) * p.ldq + hcol + (fg & 1) * 16 + (fg >> 1) * 8;
      uint32_t be2 = 0;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float f[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t p01 = pack_bf2(o[2 * b + h][0] * linv, o[2 * b + h][1] * linv), p23 = pack_bf2(o[2 * b + h][2] * linv, o[2 * b + h][3] * linv);
          f[h * 4 + 0] = __uint_as_float(p01 << 16); f[h * 4 + 1] = __uint_as_float(p01 & 0xffff0000u);
          f[h * 4 + 2] = __uint_as_float(p23 << 16); f[h * 4 + 3] = __uint_as_float(p23 & 0xffff0000u);
        }
        float amax = fmaxf(fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3]))), fmaxf(fmaxf(fabsf(f[4]), fabsf(f[5])), fmaxf(fabsf(f[6]), fabsf(f[7]))));
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64)); amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const int be = sf_mx_exp(amax);
        const float inv = __uint_as_float((uint32_t)(254 - be) << 23);
        be2 |= (uint32_t)be << (8 * b);
        const uint32_t d_lo = sf_fp8x4(f, inv), d_hi = sf_fp8x4(f + 4, inv);      // dims 32 b + fg * 4 + 0..3 and 32 b + 16 + fg * 4 + 0..3
        const uint32_t got = (uint32_t)__shfl_xor((int)((fg & 1) ? d_lo : d_hi), 16, 64);
        uint2 w;
        if (fg & 1) { w.x = got; w.y = d_hi; } else { w.x = d_lo; w.y = got; }
        if (qo < nq) *reinterpret_cast<uint2*>(qrow + b * 32) = w;
      }
      if (qo < nq && fg == 0)
        *reinterpret_cast<uint16_t*>(p.out_s + (int64_t)(head >> 1) * p.splane + (seq_base + row) * 4 + (head & 1) * 2) = (uint16_t)be2;
    } else
    if ((SF_ATT_ABL & 1) ? (linv == 1.2345e30f) : (qo < nq)) {
      bf16_t* orow = p.out + seq_base * p.ldo + hcol + (__umul24((uint32_t)(rel_first + __mul24(qo, p.tok_stride)), (uint32_t)p.ldo) + fg * 4);
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) {
        uint2 w;
        w.x = pack_bf2(o[dt][0] * linv, o[dt][1] * linv);
        w.y = pack_bf2(o[dt][2] * linv, o[dt][3] * linv);
        *reinterpret_cast<uint2*>(orow + dt * 16) = w;
      }
    }
  }
}

template <int D, int NKT, bool MXO = false>
static int launch_attn_mfma(const AttnArgs& a, int64_t n_seq, hipStream_t s) {
  auto kern = attn_mfma_kernel<D, NKT, MXO>;
  if (int rc = sf_prepare_kernel((const void*)kern, AttLds<D>::TOTAL, "sf_attention")) return rc;
  const int64_t blocks = n_seq * a.n_groups * a.heads;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), AttLds<D>::TOTAL, s, a);
  SF_LAUNCH_CHECK();
  return 0;
}

template <int D>
static int dispatch_attn_mfma(const AttnArgs& a, int64_t n_seq, int nkt, hipStream_t s) {
  switch (nkt) {   // exact key-tile counts keep the kernel free of per-tile guards
    case 1: return launch_attn_mfma<D, 1>(a, n_seq, s);
    case 2: return launch_attn_mfma<D, 2>(a, n_seq, s);
    case 3: return launch_attn_mfma<D, 3>(a, n_seq, s);
    case 4: return launch_attn_mfma<D, 4>(a, n_seq, s);
    case 5: return launch_attn_mfma<D, 5>(a, n_seq, s);      // AST: 74 tokens
    case 6: return launch_attn_mfma<D, 6>(a, n_seq, s);
    case 7: return launch_attn_mfma<D, 7>(a, n_seq, s);
    case 8: return launch_attn_mfma<D, 8>(a, n_seq, s);
    case 9: return launch_attn_mfma<D, 9>(a, n_seq, s);
    case 10: return launch_attn_mfma<D, 10>(a, n_seq, s);
    case 11: return launch_attn_mfma<D, 11>(a, n_seq, s);
    case 12: return launch_attn_mfma<D, 12>(a, n_seq, s);    // syncability: 184 tokens
    default: return launch_attn_mfma<D, 13>(a, n_seq, s);    // space attention 197 keys, sync transformer 198 tokens
  }
}

// Merge the per-group partials of the CLS query: out[seq*out_seq_rows + out_row, head*64 + d] = sum_p o_p[d] 2^(m_p - M) / sum_p l_p 2^(m_p - M)
__global__ __launch_bounds__(64) void attn_cls_combine64_kernel(const float* __restrict__ part, int n_part, bf16_t* __restrict__ out, int64_t ldo,
                                                                 int64_t out_seq_rows, int out_row, int heads, float* __restrict__ stats = nullptr) {
  const int head = blockIdx.x % heads, d = threadIdx.x;
  const int64_t seq = blockIdx.x / heads;
  const float* pp = part + (int64_t)blockIdx.x * n_part * 66;
  float M = -INFINITY, L = 0.f, O = 0.f;
  if (n_part <= 64) {
    // lane i holds record i's (m, l): one load each, the maximum and the weights by wave reductions, then the 64-wide rows weighted by a lane broadcast - the loads of
    // the second loop do not wait for anything (the serial two-pass form took 41 us for the 49 records of a time block: dependent scalar-like loads)
    const float mi = d < n_part ? pp[d * 66] : -INFINITY, li = d < n_part ? pp[d * 66 + 1] : 0.f;
    M = wave_max(mi);
    const float wi = d < n_part ? __builtin_amdgcn_exp2f(mi - M) : 0.f;
    L = wave_sum(li * wi);
#pragma unroll 7
    for (int i = 0; i < n_part; ++i) O += pp[i * 66 + 2 + d] * __shfl(wi, i, 64);
  } else {
    for (int i = 0; i < n_part; ++i) M = fmaxf(M, pp[i * 66]);
    for (int i = 0; i < n_part; ++i) {
      const float w = __builtin_amdgcn_exp2f(pp[i * 66] - M);
      L += pp[i * 66 + 1] * w;
      O += pp[i * 66 + 2 + d] * w;
    }
  }
  out[(seq * out_seq_rows + out_row) * ldo + head * 64 + d] = f2bf(O / L);
  if (stats && d == 0) { stats[(int64_t)blockIdx.x * 2] = M; stats[(int64_t)blockIdx.x * 2 + 1] = L; }   // the merged softmax statistics (the backward's, sf_attention_group_bwd_clsq)
}

extern "C" int sf_attention_cls_combine_stats(const float* partials, int n_part, uint16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row,
                                              int64_t n_seq, int heads, float* stats, void* stream) {
  SF_CHECK_ARG(partials && out && stats && n_part >= 1 && heads >= 1, "sf_attention_cls_combine_stats: bad arguments");
  if (n_seq <= 0) return 0;
  hipLaunchKernelGGL(attn_cls_combine64_kernel, dim3((unsigned)(n_seq * heads)), dim3(64), 0, (hipStream_t)stream, partials, n_part, out, ldo,
                     out_seq_rows, out_row, heads, stats);
  SF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sf_attention_cls_combine(const float* partials, int n_part, uint16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row,
                                        int64_t n_seq, int heads, void* stream) {
  SF_CHECK_ARG(partials && out && n_part >= 1 && heads >= 1, "sf_attention_cls_combine: bad arguments");
  if (n_seq <= 0) return 0;
  hipLaunchKernelGGL(attn_cls_combine64_kernel, dim3((unsigned)(n_seq * heads)), dim3(64), 0, (hipStream_t)stream, partials, n_part, out, ldo,
                     out_seq_rows, out_row, heads, (float*)nullptr);
  SF_LAUNCH_CHECK();
  return 0;
}

// ... the same merge written as MXFP8 (the CLS row of sf_attention_cls_partial_mx's output): thread d holds dim d, a scale block is one half-wave
__global__ __launch_bounds__(64) void attn_cls_combine64_mx_kernel(const float* __restrict__ part, int n_part, uint8_t* __restrict__ out_q, int64_t ldq,
                                                                    uint8_t* __restrict__ out_s, int64_t splane, int64_t out_seq_rows, int out_row, int heads) {
  const int head = blockIdx.x % heads, d = threadIdx.x;
  const int64_t seq = blockIdx.x / heads;
  const float* pp = part + (int64_t)blockIdx.x * n_part * 66;
  float M = -INFINITY, L = 0.f, O = 0.f;
  if (n_part <= 64) {                                            // (the same arithmetic, in the same order, as attn_cls_combine64_kernel: the outputs must agree bit for bit)
    const float mi = d < n_part ? pp[d * 66] : -INFINITY, li = d < n_part ? pp[d * 66 + 1] : 0.f;
    M = wave_max(mi);
    const float wi = d < n_part ? __builtin_amdgcn_exp2f(mi - M) : 0.f;
    L = wave_sum(li * wi);
#pragma unroll 7
    for (int i = 0; i < n_part; ++i) O += pp[i * 66 + 2 + d] * __shfl(wi, i, 64);
  } else {
    for (int i = 0; i < n_part; ++i) M = fmaxf(M, pp[i * 66]);
    for (int i = 0; i < n_part; ++i) {
      const float w = __builtin_amdgcn_exp2f(pp[i * 66] - M);
      L += pp[i * 66 + 1] * w;
      O += pp[i * 66 + 2 + d] * w;
    }
  }
  const float x = __uint_as_float((uint32_t)f2bf(O / L) << 16);
  float amax = fabsf(x);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const int be = sf_mx_exp(amax);
  const float inv = __uint_as_float((uint32_t)(254 - be) << 23);
  const int64_t row = seq * out_seq_rows + out_row;
  out_q[row * ldq + head * 64 + d] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(x * inv, 448.f, -448.f), 0.f, 0, false) & 0xff);
  if ((d & 31) == 0) out_s[(int64_t)(head >> 1) * splane + row * 4 + (head & 1) * 2 + (d >> 5)] = (uint8_t)be;
}

extern "C" int sf_attention_cls_combine_mx(const float* partials, int n_part, uint8_t* out_q, int64_t ldq, uint8_t* out_s, int64_t splane, int64_t out_seq_rows,
                                           int out_row, int64_t n_seq, int heads, void* stream) {
  SF_CHECK_ARG(partials && out_q && out_s && n_part >= 1 && heads >= 1 && (heads % 2) == 0, "sf_attention_cls_combine_mx: bad arguments (heads must be even: 128-column scale planes)");
  if (n_seq <= 0) return 0;
  hipLaunchKernelGGL(attn_cls_combine64_mx_kernel, dim3((unsigned)(n_seq * heads)), dim3(64), 0, (hipStream_t)stream, partials, n_part, out_q, ldq, out_s, splane,
                     out_seq_rows, out_row, heads);
  SF_LAUNCH_CHECK();
  return 0;
}

static int attention_impl(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, bf16_t* out, int64_t ldo, int64_t n_seq,
                          int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads,
                          int head_dim, float scale, float* cls_partial, void* stream, const uint8_t* key_keep = nullptr);

// sf_attention_cls_partial writing MXFP8 (FT path: the attention output is only ever the A operand of the MX projection - was bf16 output + sf_quantize_mxfp8, a 0.75 GB
// round trip per block on the 13-segment batch).  head_dim 64, 193..208 keys (13 key tiles: the space attention), an even number of heads; the CLS row comes from
// sf_attention_cls_combine_mx.  Bytes and scales equal sf_quantize_mxfp8 of sf_attention_cls_partial's output.
extern "C" int sf_attention_cls_partial_mx(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, uint8_t* out_q, int64_t ldq, uint8_t* out_s, int64_t splane,
                                           int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row,
                                           int heads, float scale, float* cls_partial, void* stream) {
  SF_CHECK_ARG(q && k && v && out_q && out_s && cls_partial && cls_row >= 0, "sf_attention_cls_partial_mx: null pointer / no CLS row");
  SF_CHECK_ARG((n_tok + 1 + 15) / 16 == 13 && (n_tok % 16) != 0, "sf_attention_cls_partial_mx: n_tok %d outside 192..207 (13 key tiles with a free query slot)", n_tok);
  SF_CHECK_ARG((heads % 2) == 0 && (ld % 8) == 0 && (ldq % 8) == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out_q % 8) == 0 &&
               ((uintptr_t)out_s % 2) == 0, "sf_attention_cls_partial_mx: alignment / even head count");
  SF_CHECK_ARG(seq_rows >= 1 && seq_rows < (1 << 24) && ld < (1 << 24) && seq_rows * ld < ((int64_t)1 << 31), "sf_attention_cls_partial_mx: a sequence must span < 2^31 elements");
  if (n_seq <= 0) return 0;
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.ld = ld; a.out = nullptr; a.ldo = 0; a.seq_rows = seq_rows;
  a.n_groups = n_groups; a.row0 = row0; a.group_stride = group_stride; a.tok_stride = tok_stride; a.n_tok = n_tok;
  a.cls_row = cls_row; a.heads = heads; a.scale = scale; a.cls_part = cls_partial; a.key_keep = nullptr;
  a.out_q = out_q; a.ldq = ldq; a.out_s = out_s; a.splane = splane;
  return launch_attn_mfma<64, 13, true>(a, n_seq, (hipStream_t)stream);
}

extern "C" int sf_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, bf16_t* out, int64_t ldo, int64_t n_seq,
                            int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads,
                            int head_dim, float scale, void* stream) {
  return attention_impl(q, k, v, ld, out, ldo, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim, scale,
                        nullptr, stream);
}

// sf_attention + the CLS query's per-group partials ([n_seq][heads][n_groups][66] fp32), for sf_attention_cls_combine.
extern "C" int sf_attention_cls_partial(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, bf16_t* out, int64_t ldo, int64_t n_seq,
                                        int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row,
                                        int heads, int head_dim, float scale, float* cls_partial, void* stream) {
  SF_CHECK_ARG(cls_partial && cls_row >= 0 && head_dim == 64, "sf_attention_cls_partial: needs a partial buffer, cls_row >= 0 and head_dim 64");
  SF_CHECK_ARG(n_tok <= 8 || (n_tok % 16) != 0, "sf_attention_cls_partial: n_tok %% 16 == 0 leaves no free query slot");
  return attention_impl(q, k, v, ld, out, ldo, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim, scale,
                        cls_partial, stream);
}

// ... and with token masks (the masked forward on the fused-CLS schedule): key_keep as in sf_attention_masked
extern "C" int sf_attention_cls_partial_masked(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, bf16_t* out, int64_t ldo, int64_t n_seq,
                                               int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row,
                                               int heads, int head_dim, float scale, float* cls_partial, const uint8_t* key_keep, void* stream) {
  SF_CHECK_ARG(cls_partial && key_keep && cls_row >= 0 && head_dim == 64 && n_tok > 8, "sf_attention_cls_partial_masked: needs a partial buffer, key flags, cls_row >= 0, head_dim 64 and the MFMA path (n_tok > 8)");
  SF_CHECK_ARG((n_tok % 16) != 0, "sf_attention_cls_partial_masked: n_tok %% 16 == 0 leaves no free query slot");
  return attention_impl(q, k, v, ld, out, ldo, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim, scale,
                        cls_partial, stream, key_keep);
}

static int attention_impl(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, bf16_t* out, int64_t ldo,
                            int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride,
                            int n_tok, int cls_row, int heads, int head_dim, float scale, float* cls_partial, void* stream,
                            const uint8_t* key_keep) {
  SF_CHECK_ARG(q && k && v && out, "sf_attention: null pointer");
  SF_CHECK_ARG(head_dim == 64 || head_dim == 96, "sf_attention: head_dim %d not supported (64, 96)", head_dim);
  SF_CHECK_ARG((ld % 8) == 0 && (ldo % 8) == 0, "sf_attention: ld/ldo must be multiples of 8 elements");
  SF_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % 16) == 0,
               "sf_attention: pointers must be 16-byte aligned");
  SF_CHECK_ARG(n_tok >= 1 && n_tok + (cls_row >= 0 ? 1 : 0) <= ATT_MAX_KT * 16, "sf_attention: n_tok %d out of range", n_tok);
  SF_CHECK_ARG(n_groups >= 1 && heads >= 1, "sf_attention: bad group/head count");
  SF_CHECK_ARG(seq_rows >= 1 && seq_rows < (1 << 24) && ld < (1 << 24) && ldo < (1 << 24) && seq_rows * ld < ((int64_t)1 << 31) && seq_rows * ldo < ((int64_t)1 << 31),
               "sf_attention: a sequence must span < 2^31 elements (32-bit row offsets inside a sequence)");
  if (n_seq <= 0) return 0;
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.ld = ld; a.out = out; a.ldo = ldo; a.seq_rows = seq_rows;
  a.n_groups = n_groups; a.row0 = row0; a.group_stride = group_stride; a.tok_stride = tok_stride; a.n_tok = n_tok;
  a.cls_row = cls_row; a.heads = heads; a.scale = scale; a.cls_part = cls_partial; a.key_keep = key_keep;
  hipStream_t s = (hipStream_t)stream;
  if (head_dim == 64 && n_tok <= 8) {
    const int64_t units = n_seq * n_groups * heads;
    if (cls_partial) hipLaunchKernelGGL(attn_tiny64_kernel<true>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, a, units);
    else hipLaunchKernelGGL(attn_tiny64_kernel<false>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, a, units);
    SF_LAUNCH_CHECK();
    return 0;
  }
  const int nkt = (n_tok + (cls_row >= 0 ? 1 : 0) + 15) / 16;
  return head_dim == 64 ? dispatch_attn_mfma<64>(a, n_seq, nkt, s) : dispatch_attn_mfma<96>(a, n_seq, nkt, s);
}

extern "C" int sf_attention_masked(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, bf16_t* out, int64_t ldo, int64_t n_seq,
                                   int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row,
                                   int heads, int head_dim, float scale, const uint8_t* key_keep, void* stream) {
  return attention_impl(q, k, v, ld, out, ldo, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim, scale,
                        nullptr, stream, key_keep);
}

static int attention_cls_impl(const bf16_t* q, int64_t q_seq_rows, int q_row, const bf16_t* k, const bf16_t* v, int64_t ld, int64_t kv_seq_rows,
                              int kv_row0, int n_keys, bf16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row, int64_t n_seq, int heads,
                              int head_dim, float scale, const uint8_t* key_keep, void* stream, float* stats = nullptr);

// sf_attention_cls that also writes the query's softmax statistics stats[(seq * heads + head) * 2 + {0, 1}] = (maximum in the base-2 domain incl. the scale, sum)
extern "C" int sf_attention_cls_stats(const bf16_t* q, int64_t q_seq_rows, int q_row, const bf16_t* k, const bf16_t* v, int64_t ld, int64_t kv_seq_rows, int kv_row0,
                                      int n_keys, bf16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row, int64_t n_seq, int heads, int head_dim, float scale,
                                      float* stats, void* stream) {
  SF_CHECK_ARG(stats, "sf_attention_cls_stats: null statistics buffer");
  return attention_cls_impl(q, q_seq_rows, q_row, k, v, ld, kv_seq_rows, kv_row0, n_keys, out, ldo, out_seq_rows, out_row, n_seq, heads, head_dim,
                            scale, nullptr, stream, stats);
}

extern "C" int sf_attention_cls(const bf16_t* q, int64_t q_seq_rows, int q_row, const bf16_t* k, const bf16_t* v,
                                int64_t ld, int64_t kv_seq_rows, int kv_row0, int n_keys, bf16_t* out, int64_t ldo,
                                int64_t out_seq_rows, int out_row, int64_t n_seq, int heads, int head_dim, float scale,
                                void* stream) {
  return attention_cls_impl(q, q_seq_rows, q_row, k, v, ld, kv_seq_rows, kv_row0, n_keys, out, ldo, out_seq_rows, out_row, n_seq, heads, head_dim,
                            scale, nullptr, stream);
}

extern "C" int sf_attention_cls_masked(const bf16_t* q, int64_t q_seq_rows, int q_row, const bf16_t* k, const bf16_t* v, int64_t ld,
                                       int64_t kv_seq_rows, int kv_row0, int n_keys, bf16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row,
                                       int64_t n_seq, int heads, int head_dim, float scale, const uint8_t* key_keep, void* stream) {
  return attention_cls_impl(q, q_seq_rows, q_row, k, v, ld, kv_seq_rows, kv_row0, n_keys, out, ldo, out_seq_rows, out_row, n_seq, heads, head_dim,
                            scale, key_keep, stream);
}

static int attention_cls_impl(const bf16_t* q, int64_t q_seq_rows, int q_row, const bf16_t* k, const bf16_t* v, int64_t ld, int64_t kv_seq_rows,
                              int kv_row0, int n_keys, bf16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row, int64_t n_seq, int heads,
                              int head_dim, float scale, const uint8_t* key_keep, void* stream, float* stats) {
  SF_CHECK_ARG(q && k && v && out, "sf_attention_cls: null pointer");
  SF_CHECK_ARG(head_dim == 64, "sf_attention_cls: head_dim %d not supported (64)", head_dim);
  SF_CHECK_ARG((ld % 8) == 0 && (ldo % 8) == 0 && n_keys >= 1, "sf_attention_cls: bad shape");
  if (n_seq <= 0) return 0;
  ClsArgs a;
  a.q = q; a.q_seq_rows = q_seq_rows; a.q_row = q_row; a.k = k; a.v = v; a.ld = ld; a.kv_seq_rows = kv_seq_rows;
  a.kv_row0 = kv_row0; a.n_keys = n_keys; a.out = out; a.ldo = ldo; a.out_seq_rows = out_seq_rows; a.out_row = out_row;
  a.heads = heads; a.scale = scale; a.key_keep = key_keep; a.stats = stats;
  hipLaunchKernelGGL(attn_cls64_kernel, dim3((unsigned)(n_seq * heads)), dim3(256), 0, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
