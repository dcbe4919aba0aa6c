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
#include "../../include/synchformer_hip.h"

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; int64_t ld;
  bf16_t* out; int64_t ldo;
  int64_t seq_rows;       // rows per sequence in q/k/v and out
  int n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads;
  float scale;
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

__global__ __launch_bounds__(256) void attn_tiny64_kernel(AttnArgs p, int64_t total_units) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // (seq, group, head), head fastest
  if (unit >= total_units) return;
  const int head = (int)(unit % p.heads);
  const int64_t sg = unit / p.heads;
  const int g = (int)(sg % p.n_groups);
  const int64_t seq = sg / p.n_groups;
  const int qi = lane >> 3, sub = lane & 7;
  const int64_t seq_base = seq * p.seq_rows;
  const int64_t first = seq_base + p.row0 + (int64_t)g * p.group_stride;
  const int col = head * 64 + sub * 8;
  const int qtok = qi < p.n_tok ? qi : p.n_tok - 1;                      // idle lanes shadow the last query
  float qf[8];
  unpack8(*reinterpret_cast<const uint4*>(p.q + (first + (int64_t)qtok * p.tok_stride) * p.ld + col), qf);
  const int has_cls = p.cls_row >= 0 ? 1 : 0;
  const int nk = p.n_tok + has_cls;
  float s[9];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    s[j] = -INFINITY;
    if (j < nk) {
      const int64_t krow = (has_cls && j == 0) ? seq_base + p.cls_row : first + (int64_t)(j - has_cls) * p.tok_stride;
      float kf[8];
      unpack8(*reinterpret_cast<const uint4*>(p.k + krow * p.ld + col), kf);
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d += qf[e] * kf[e];
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
      s[j] = d * p.scale;
      m = fmaxf(m, s[j]);
    }
  }
  float l = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    if (j < nk) {
      const float e = __expf(s[j] - m);
      l += e;
      const int64_t vrow = (has_cls && j == 0) ? seq_base + p.cls_row : first + (int64_t)(j - has_cls) * p.tok_stride;
      float vf[8];
      unpack8(*reinterpret_cast<const uint4*>(p.v + vrow * p.ld + col), vf);
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] += e * vf[t];
    }
  }
  if (qi < p.n_tok) {
    const float inv = 1.0f / l;
    uint4 w;
    w.x = pack_bf2(o[0] * inv, o[1] * inv); w.y = pack_bf2(o[2] * inv, o[3] * inv);
    w.z = pack_bf2(o[4] * inv, o[5] * inv); w.w = pack_bf2(o[6] * inv, o[7] * inv);
    *reinterpret_cast<uint4*>(p.out + (first + (int64_t)qi * p.tok_stride) * p.ldo + col) = w;
  }
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
    float kf[8], vf[8];
    unpack8(*reinterpret_cast<const uint4*>(p.k + (kv0 + j) * p.ld + col), kf);
    unpack8(*reinterpret_cast<const uint4*>(p.v + (kv0 + j) * p.ld + col), vf);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += qf[e] * kf[e];
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
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
#define ATT_VT_LD 232            // V^T row stride in keys (bf16): 464 B, conflict-free ds_read_b64 / ds_write_b32

template <int D>
struct AttLds {
  static constexpr int K_LD = (D == 64) ? 128 : (D * 2 + 16);   // bytes per K row (D=64: XOR-swizzled 128 B rows)
  static constexpr int K_BYTES = 224 * K_LD;
  static constexpr int VT_BYTES = D * ATT_VT_LD * 2;
  static constexpr int TOTAL = K_BYTES + VT_BYTES;
};

template <int D>
__device__ __forceinline__ int k_lds_off(int row, int chunk) {   // byte offset of 16-B chunk `chunk` of K row `row`
  if (D == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
  return row * AttLds<D>::K_LD + (chunk << 4);
}

template <int D>
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_lds = smem;
  bf16_t* vt_lds = reinterpret_cast<bf16_t*>(smem + AttLds<D>::K_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x % p.heads;
  const int64_t sg = blockIdx.x / p.heads;
  const int g = (int)(sg % p.n_groups);
  const int64_t seq = sg / p.n_groups;
  const int64_t seq_base = seq * p.seq_rows;
  const int64_t first = seq_base + p.row0 + (int64_t)g * p.group_stride;
  const int has_cls = p.cls_row >= 0 ? 1 : 0;
  const int nk = p.n_tok + has_cls, nq = p.n_tok;
  const int nkt = (nk + 15) >> 4, nqt = (nq + 15) >> 4;
  const int hcol = head * D;
  constexpr int CH = D / 8;          // 16-byte chunks per row

  auto key_row = [&](int j) -> int64_t {
    return (has_cls && j == 0) ? seq_base + p.cls_row : first + (int64_t)(j - has_cls) * p.tok_stride;
  };

  // ---- stage K: rows [0, nkt*16) (rows >= nk zero-filled) ------------------------------------------------
  for (int idx = tid; idx < nkt * 16 * CH; idx += 256) {
    const int row = idx / CH, ch = idx - row * CH;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (row < nk) val = *reinterpret_cast<const uint4*>(p.k + key_row(row) * p.ld + hcol + ch * 8);
    *reinterpret_cast<uint4*>(k_lds + k_lds_off<D>(row, ch)) = val;
  }
  // ---- stage V^T: vt[d][key], written as dwords holding keys (2*pp, 2*pp+1); keys in [nk, 32*ceil(nkt/2)) = 0 --
  {
    const int npairs = ((nkt + 1) >> 1) * 16;            // key pairs to cover all 32-key PV steps
    const int pp_l = lane & 31, half = lane >> 5;        // 32 consecutive pairs per half-wave -> conflict-free b32 stores
    for (int pbase = 0; pbase < npairs; pbase += 32) {
      const int pp = pbase + pp_l;
      for (int ch = wave * 2 + half; ch < CH; ch += 8) {
        if (pp < npairs) {
          uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0);
          if (2 * pp < nk) v0 = *reinterpret_cast<const uint4*>(p.v + key_row(2 * pp) * p.ld + hcol + ch * 8);
          if (2 * pp + 1 < nk) v1 = *reinterpret_cast<const uint4*>(p.v + key_row(2 * pp + 1) * p.ld + hcol + ch * 8);
          const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, b[4] = {v1.x, v1.y, v1.z, v1.w};
          uint32_t* dst = reinterpret_cast<uint32_t*>(vt_lds) + pp;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dst[(ch * 8 + 2 * e) * (ATT_VT_LD / 2)] = (a[e] & 0xffffu) | (b[e] << 16);
            dst[(ch * 8 + 2 * e + 1) * (ATT_VT_LD / 2)] = (a[e] >> 16) | (b[e] & 0xffff0000u);
          }
        }
      }
    }
  }
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int qt = wave; qt < nqt; qt += 4) {
    // ---- Q fragments (B operand of S^T): query qt*16 + fr, dims ks*32 + fg*8 .. +7 ------------------------
    int qi = qt * 16 + fr; if (qi > nq - 1) qi = nq - 1;
    const bf16_t* qrow = p.q + (first + (int64_t)qi * p.tok_stride) * p.ld + hcol;
    bf16x8 qf[D / 32];
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 32 + fg * 8);

    // ---- S^T tiles -------------------------------------------------------------------------------------
    f32x4 s[ATT_MAX_KT];
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_KT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + k_lds_off<D>(kt * 16 + fr, ks * 4 + fg));
          s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
        }
      }
    }
    // ---- softmax over keys for query column (lane & 15) ---------------------------------------------------
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_KT; ++kt)
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + fg * 4 + r;
          const float v = (key < nk) ? s[kt][r] * p.scale : -INFINITY;
          s[kt][r] = v;
          m = fmaxf(m, v);
        }
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_KT; ++kt)
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __expf(s[kt][r] - m); s[kt][r] = e; l += e; }
      }
    l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
    const float linv = 1.0f / l;

    // ---- O = P V ---------------------------------------------------------------------------------------
    f32x4 o[D / 16];
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < (ATT_MAX_KT + 1) / 2; ++kk) {
      if (2 * kk < nkt) {
        // A operand: slots 0..3 <- tile 2kk regs, slots 4..7 <- tile 2kk+1 regs (zero beyond nkt)
        union { bf16x8 v; uint32_t u[4]; } pa;
        pa.u[0] = pack_bf2(s[2 * kk][0], s[2 * kk][1]);
        pa.u[1] = pack_bf2(s[2 * kk][2], s[2 * kk][3]);
        if (2 * kk + 1 < ATT_MAX_KT && 2 * kk + 1 < nkt) {
          const int t1 = (2 * kk + 1 < ATT_MAX_KT) ? 2 * kk + 1 : 0;
          pa.u[2] = pack_bf2(s[t1][0], s[t1][1]);
          pa.u[3] = pack_bf2(s[t1][2], s[t1][3]);
        } else { pa.u[2] = 0; pa.u[3] = 0; }
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
          const bf16_t* vrow = vt_lds + (dt * 16 + fr) * ATT_VT_LD + kk * 32 + fg * 4;
          union { bf16x8 v; uint2 h[2]; } vb;
          vb.h[0] = *reinterpret_cast<const uint2*>(vrow);
          vb.h[1] = *reinterpret_cast<const uint2*>(vrow + 16);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa.v, vb.v, o[dt], 0, 0, 0);
        }
      }
    }
    // ---- normalise + store: o[dt][r] is (query qt*16 + fg*4 + r, dim dt*16 + fr) ---------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qo = qt * 16 + fg * 4 + r;
      const float inv = __shfl(linv, fg * 4 + r, 64);        // lane (fg*4+r) holds query column fg*4+r of this tile
      if (qo < nq) {
        bf16_t* orow = p.out + (first + (int64_t)qo * p.tok_stride) * p.ldo + hcol + fr;
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) orow[dt * 16] = f2bf(o[dt][r] * inv);
      }
    }
  }
}

template <int D>
static int launch_attn_mfma(const AttnArgs& a, int64_t n_seq, hipStream_t s) {
  auto kern = attn_mfma_kernel<D>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, AttLds<D>::TOTAL);
    if (e != hipSuccess) { sf_set_error("sf_attention: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  const int64_t blocks = n_seq * a.n_groups * a.heads;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), AttLds<D>::TOTAL, s, a);
  SF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sf_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, bf16_t* out, int64_t ldo,
                            int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride,
                            int n_tok, int cls_row, int heads, int head_dim, float scale, void* stream) {
  SF_CHECK_ARG(q && k && v && out, "sf_attention: null pointer");
  SF_CHECK_ARG(head_dim == 64 || head_dim == 96, "sf_attention: head_dim %d not supported (64, 96)", head_dim);
  SF_CHECK_ARG((ld % 8) == 0 && (ldo % 8) == 0, "sf_attention: ld/ldo must be multiples of 8 elements");
  SF_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % 16) == 0,
               "sf_attention: pointers must be 16-byte aligned");
  SF_CHECK_ARG(n_tok >= 1 && n_tok + (cls_row >= 0 ? 1 : 0) <= ATT_MAX_KT * 16, "sf_attention: n_tok %d out of range", n_tok);
  SF_CHECK_ARG(n_groups >= 1 && heads >= 1, "sf_attention: bad group/head count");
  if (n_seq <= 0) return 0;
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.ld = ld; a.out = out; a.ldo = ldo; a.seq_rows = seq_rows;
  a.n_groups = n_groups; a.row0 = row0; a.group_stride = group_stride; a.tok_stride = tok_stride; a.n_tok = n_tok;
  a.cls_row = cls_row; a.heads = heads; a.scale = scale;
  hipStream_t s = (hipStream_t)stream;
  if (head_dim == 64 && n_tok <= 8) {
    const int64_t units = n_seq * n_groups * heads;
    hipLaunchKernelGGL(attn_tiny64_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, a, units);
    SF_LAUNCH_CHECK();
    return 0;
  }
  return head_dim == 64 ? launch_attn_mfma<64>(a, n_seq, s) : launch_attn_mfma<96>(a, n_seq, s);
}

extern "C" int sf_attention_cls(const bf16_t* q, int64_t q_seq_rows, int q_row, const bf16_t* k, const bf16_t* v,
                                int64_t ld, int64_t kv_seq_rows, int kv_row0, int n_keys, bf16_t* out, int64_t ldo,
                                int64_t out_seq_rows, int out_row, int64_t n_seq, int heads, int head_dim, float scale,
                                void* stream) {
  SF_CHECK_ARG(q && k && v && out, "sf_attention_cls: null pointer");
  SF_CHECK_ARG(head_dim == 64, "sf_attention_cls: head_dim %d not supported (64)", head_dim);
  SF_CHECK_ARG((ld % 8) == 0 && (ldo % 8) == 0 && n_keys >= 1, "sf_attention_cls: bad shape");
  if (n_seq <= 0) return 0;
  ClsArgs a;
  a.q = q; a.q_seq_rows = q_seq_rows; a.q_row = q_row; a.k = k; a.v = v; a.ld = ld; a.kv_seq_rows = kv_seq_rows;
  a.kv_row0 = kv_row0; a.n_keys = n_keys; a.out = out; a.ldo = ldo; a.out_seq_rows = out_seq_rows; a.out_row = out_row;
  a.heads = heads; a.scale = scale;
  hipLaunchKernelGGL(attn_cls64_kernel, dim3((unsigned)(n_seq * heads)), dim3(256), 0, (hipStream_t)stream, a);
  SF_LAUNCH_CHECK();
  return 0;
}
