// Shared device/host helpers for the gfx950 (MI355X, CDNA4) Synchformer kernels.
// wave = 64 lanes everywhere in this directory; nothing here is portable and nothing is meant to be.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef uint16_t bf16_t;   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SF_WAVE 64

// ---- error plumbing (host) --------------------------------------------------------------------------
void sf_set_error(const char* fmt, ...);
#define SF_CHECK_ARG(cond, ...)                                  \
  do {                                                           \
    if (!(cond)) { sf_set_error(__VA_ARGS__); return -1; }       \
  } while (0)
#define SF_LAUNCH_CHECK()                                                        \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) { sf_set_error("%s: %s", __func__, hipGetErrorString(e__)); return (int)e__; } \
  } while (0)

// ---- per-device launch set-up (sf_misc.hip): safe for a process that drives several GPUs and for concurrent host threads ---------------
// sf_prepare_kernel: hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, current device) - the attribute is per device, a process-wide
// `static bool` would leave the second GPU of a process without it; 0 or the HIP error (sf_last_error set).  sf_cu_count: multiprocessors of the
// current device, cached per device (0 on error, sf_last_error set).
int sf_prepare_kernel(const void* kernel, int lds_bytes, const char* who);
int sf_cu_count(const char* who);

// ---- row maps -----------------------------------------------------------------------------------------
// A logical row r (< 2^31) of an operand lives at physical row
//     (r / n12) * sA + ((r % n12) / n2) * s1 + (r % n2) * s2 + off
// Host passes `const int64_t[6] = {n12, n2, sA, s1, s2, off}` or NULL for identity (n12 == 0 marks identity).
struct RowMap {
  uint32_t n12, n2;
  int64_t sA, s1, s2, off;
};
static inline RowMap sf_rowmap(const int64_t* m) {
  RowMap r;
  if (m) { r.n12 = (uint32_t)m[0]; r.n2 = (uint32_t)m[1]; r.sA = m[2]; r.s1 = m[3]; r.s2 = m[4]; r.off = m[5]; }
  else   { r.n12 = 0; r.n2 = 1; r.sA = 0; r.s1 = 0; r.s2 = 1; r.off = 0; }
  return r;
}
__device__ __forceinline__ int64_t map_row(const RowMap& m, int64_t r) {
  if (m.n12 == 0) return r;                                // identity (wave-uniform branch)
  const uint32_t ur = (uint32_t)r;
  const uint32_t a = ur / m.n12, rem = ur - a * m.n12;
  const uint32_t i1 = rem / m.n2, i2 = rem - i1 * m.n2;
  return (int64_t)a * m.sA + (int64_t)i1 * m.s1 + (int64_t)i2 * m.s2 + m.off;
}

// ---- bf16 <-> f32 ---------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// f32 -> bf16, round-to-nearest-even with NaN preserved: gfx950 has the conversion in hardware
// (v_cvt_pk_bf16_f32); the vector convert below is what makes hipcc select it.
typedef __attribute__((ext_vector_type(2))) __bf16 sf_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float sf_f32x2_t;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const sf_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sf_bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return __half2float(__ushort_as_half(h)); }

// ---- MXFP8 quantisation pieces shared by the kernels that write an MX operand themselves (the rule of sf_quantize_mxfp8) ------------
// E8M0 byte of a 32-element block with absolute maximum `amax` (>= 0): its exponent - 8 (448 = 1.75 * 2^8), clamped to [1, 254]
// SF_MX_SCALE_RULE - how a block's shared exponent follows its absolute maximum:
//   0: the OCP Microscaling floor rule, e = floor(log2 amax) - 8: amax / 2^e lies in [256, 512), everything above 448 (the largest e4m3 normal) SATURATES - up to
//      12.5 % off on the largest element of every fifth block, always towards zero;
//   1: the no-saturation rule: one exponent up whenever amax / 2^e would exceed 448 (mantissa of amax above 1.75), so the block maximum is always representable; the
//      other elements of such a block lose one bit.  Which of the two is the product rule, and what it does to the logits: DESIGN 4 / profiles/r06_mxfp8_scale_rule.md.
#ifndef SF_MX_SCALE_RULE
#define SF_MX_SCALE_RULE 0
#endif
// biased E8M0 exponent of a block with absolute maximum `amax` (>= 0), before clamping to the byte's range
__device__ __forceinline__ int sf_mx_be(float amax) {
  const uint32_t u = __float_as_uint(amax);
  int be = (int)((u >> 23) & 0xff) - 8;
  if (SF_MX_SCALE_RULE == 1) be += (u & 0x7fffffu) > 0x600000u ? 1 : 0;
  return be;
}
__device__ __forceinline__ int sf_mx_exp(float amax) {
  const int be = sf_mx_be(amax);
  return be < 1 ? 1 : (be > 254 ? 254 : be);
}
// the reciprocal of that block scale, 2^(127 - be)
__device__ __forceinline__ float sf_mx_inv(int be) { return __uint_as_float((uint32_t)(254 - be) << 23); }
// e4m3 bytes of four values scaled by `inv`
__device__ __forceinline__ uint32_t sf_fp8x4(const float* f, float inv) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f[0] * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(f[1] * inv, 448.f, -448.f), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f[2] * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(f[3] * inv, 448.f, -448.f), w, true);
  return (uint32_t)w;
}

// Sum over the 8 lanes of an aligned lane group (bit-identical in all eight, same tree as xor-shuffles 1, 2, 4): three DPP adds - quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror (lane i <-> 7 - i, i.e. the other quad) - instead of three ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ float sum8_dpp(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));
  return x;
}

// ---- wave reductions ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU (nn.GELU default), 0.5 x (1 + erf(x / sqrt 2)) = 0.5 x + 0.5 |x| erf(|x| / sqrt 2), with erf from
// Abramowitz-Stegun 7.1.28:  erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16, |err| <= 3e-7 (measured in fp32 against float64 over
// [-12, 12]: |gelu err| <= 7.1e-7 in the folded form below, a fifth of a bf16 half-ulp at worst).  ONE transcendental (v_rcp) and no exp; the polynomial
// and the four squarings are written on float2 so hipcc emits v_pk_fma_f32 / v_pk_mul_f32 (two elements per instruction): ~9
// full-rate slots + 1 quarter-rate op per element, against ~17 + 2 for the 7.1.26 form (rcp AND exp) used before - the fc1
// epilogue is VALU-bound, this is worth ~4 % on that GEMM.
__device__ __forceinline__ sf_f32x2_t gelu_erf2(sf_f32x2_t x) {
  // gelu(x) = max(x, 0) - 0.5 |x| (1 - erf(|x| / sqrt 2)) = max(x, 0) - 0.5 |x| / t^16,  t = 1 + b1 |x| + ... + b6 |x|^6 with b_k = a_k / sqrt(2)^k folded
  // (15 scalar-equivalent operations per element instead of 17: no separate z = |x| / sqrt 2, no 1 - r).  Every finite x is handled (t^16 overflows to
  // inf beyond |x| ~ 25: r = 0); x = +-inf itself gives 0 * inf = NaN where nn.GELU gives +inf / 0.
  const sf_f32x2_t ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
  sf_f32x2_t p = ax * 5.3829750000e-06f + 4.8890635643e-05f;
  p = p * ax + 3.8003575000e-05f;
  p = p * ax + 3.2776263241e-03f;
  p = p * ax + 2.1141006150e-02f;
  p = p * ax + 4.9867346967e-02f;
  p = p * ax + 1.0f;
  p = p * p; p = p * p; p = p * p; p = p * p;                  // ^16 (inf for |x| > ~25: rcp(inf) = 0, erf = 1)
  const sf_f32x2_t r = {__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
  const sf_f32x2_t relu = {__builtin_fmaxf(x.x, 0.f), __builtin_fmaxf(x.y, 0.f)};
  return (ax * -0.5f) * r + relu;
}
// Two pairs in lockstep: dependent v_pk_* operations need a wait state between them (hipcc fills it with s_nop: 11 of the 29 issue slots of one
// pair's chain); written interleaved, the second pair's operation sits in that slot.
__device__ __forceinline__ void gelu_erf4(sf_f32x2_t& a, sf_f32x2_t& b) {
  const sf_f32x2_t aa = {__builtin_fabsf(a.x), __builtin_fabsf(a.y)}, ab = {__builtin_fabsf(b.x), __builtin_fabsf(b.y)};
  sf_f32x2_t pa = aa * 5.3829750000e-06f + 4.8890635643e-05f, pb = ab * 5.3829750000e-06f + 4.8890635643e-05f;
  pa = pa * aa + 3.8003575000e-05f; pb = pb * ab + 3.8003575000e-05f;
  pa = pa * aa + 3.2776263241e-03f; pb = pb * ab + 3.2776263241e-03f;
  pa = pa * aa + 2.1141006150e-02f; pb = pb * ab + 2.1141006150e-02f;
  pa = pa * aa + 4.9867346967e-02f; pb = pb * ab + 4.9867346967e-02f;
  pa = pa * aa + 1.0f; pb = pb * ab + 1.0f;
  pa = pa * pa; pb = pb * pb; pa = pa * pa; pb = pb * pb; pa = pa * pa; pb = pb * pb; pa = pa * pa; pb = pb * pb;
  const sf_f32x2_t ra = {__builtin_amdgcn_rcpf(pa.x), __builtin_amdgcn_rcpf(pa.y)}, rb = {__builtin_amdgcn_rcpf(pb.x), __builtin_amdgcn_rcpf(pb.y)};
  const sf_f32x2_t za = {__builtin_fmaxf(a.x, 0.f), __builtin_fmaxf(a.y, 0.f)}, zb = {__builtin_fmaxf(b.x, 0.f), __builtin_fmaxf(b.y, 0.f)};
  a = (aa * -0.5f) * ra + za;
  b = (ab * -0.5f) * rb + zb;
}
__device__ __forceinline__ float gelu_erf(float x) {
  const sf_f32x2_t v = {x, x};
  return gelu_erf2(v).x;
}
