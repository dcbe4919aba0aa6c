// The PyTorch DISPATCHER LIBRARY of the hot path ("registered as PyTorch-ROCm custom ops"): TORCH_LIBRARY(synchformer) defines the launches the default
// schedule of Synchformer.forward is made of as out-variant operators (they mutate their output arguments and return nothing) and TORCH_LIBRARY_IMPL(synchformer, CUDA) - the
// dispatch key HIP tensors carry on PyTorch-ROCm - implements each one as a call into the C ABI of libsynchformer_hip.so (include/synchformer_hip.h) on the CURRENT HIP stream.
// Loaded with torch.ops.load_library("synchformer_amd/lib/libsynchformer_torch.so") (synchformer_amd/ops.py::register_torch_ops); the FakeTensor / Meta implementations and the
// two functional ops with autograd (synchformer::linear, synchformer::layer_norm768) are registered from Python on the same namespace.  Host-only C++: torch is plumbing here
// (tensors -> pointers, strides, the stream); no device code, no kernels, no fallback - every operator needs HIP tensors and fails loudly otherwise.
//
// Each operator mirrors the argument marshalling of the ctypes wrapper of the same name in synchformer_amd/ops.py (the drop-in module and the tests compare the two routes bit
// for bit); shape / dtype contracts beyond what is needed to form the call are checked by the C ABI itself (sf_last_error()).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // PyTorch-ROCm files HIP devices under DeviceType cuda: the plain c10::hip::HIPGuard refuses them
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <optional>

#include "../../include/synchformer_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

void* dev(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "synchformer::", name, ": expected a HIP device tensor, got ", t.device(), " (no CPU fallback exists)");
  return t.data_ptr();
}
void* devo(const OptTensor& t, const char* name) { return t.has_value() ? dev(*t, name) : nullptr; }
int64_t ld(const Tensor& t) {
  TORCH_CHECK(t.dim() == 2 && t.stride(1) == 1, "synchformer: expected a row-major 2-D view, got sizes ", t.sizes(), " strides ", t.strides());
  return t.stride(0);
}
void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
int dtype_code(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return SF_F32;
    case at::kBFloat16: return SF_BF16;
    case at::kHalf: return SF_F16;
    case at::kByte: return SF_U8;
    default: TORCH_CHECK(false, "synchformer: unsupported dtype ", t.scalar_type()); return -1;
  }
}
void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", sf_last_error()); }
const uint16_t* bf(const void* p) { return static_cast<const uint16_t*>(p); }
uint16_t* bfm(void* p) { return static_cast<uint16_t*>(p); }
const uint8_t* u8(const void* p) { return static_cast<const uint8_t*>(p); }
uint8_t* u8m(void* p) { return static_cast<uint8_t*>(p); }
const float* f32(const void* p) { return static_cast<const float*>(p); }
float* f32m(void* p) { return static_cast<float*>(p); }

// ---- argument contracts: the C ABI only sees raw pointers, so everything a wrong tensor could turn into an out-of-bounds device access is checked here -
// the same asserts the ctypes wrappers of synchformer_amd/ops.py make
#define SF_ARG(cond, name, ...) TORCH_CHECK(cond, "synchformer::", name, ": ", __VA_ARGS__)
bool is_bf16(const Tensor& t) { return t.scalar_type() == at::kBFloat16; }
bool is_f32(const Tensor& t) { return t.scalar_type() == at::kFloat; }
bool is_u8(const Tensor& t) { return t.scalar_type() == at::kByte; }
void check_bias(const OptTensor& bias, int64_t n, const char* name) {
  if (bias.has_value()) SF_ARG(is_f32(*bias) && bias->is_contiguous() && bias->numel() >= n, name, "bias is contiguous fp32 with at least ", n, " elements");
}
void check_scale_planes(const Tensor& s_, int64_t planes, int64_t rows, const char* name, const char* what) {
  SF_ARG(is_u8(s_) && s_.dim() == 3 && s_.size(0) == planes && s_.size(1) >= rows && s_.size(2) == 4 && s_.is_contiguous(), name, what, " is uint8 (", planes, ", >= ", rows,
         ", 4) contiguous, got ", s_.sizes());
}
// the fused halves of DividedSpaceTimeBlock: x / out (>= n_seq * 1569, 768), w (2304, 768), side (>= n_seq * 33, 2304), partials fp32 with n_part records per sequence and head
void check_qkv_fused(const char* name, const Tensor& x, const Tensor& w, const OptTensor& bias, const Tensor& side, const Tensor& out, const Tensor& partials, int64_t n_seq,
                     int64_t n_part, bool mx_in, bool mx_out) {
  const int64_t rows = n_seq * 1569;
  SF_ARG(n_seq > 0, name, "n_seq > 0");
  SF_ARG((mx_in ? is_u8(x) && is_u8(w) : is_bf16(x) && is_bf16(w)) && is_bf16(side) && (mx_out ? is_u8(out) : is_bf16(out)) && is_f32(partials), name,
         mx_in ? "x / w uint8 (e4m3), side bf16, partials fp32" : "x / w / side / out bf16, partials fp32");
  SF_ARG(x.dim() == 2 && x.size(1) == 768 && x.size(0) >= rows && out.dim() == 2 && out.size(1) == 768 && out.size(0) >= rows, name, "x / out have 768 columns and at least n_seq * 1569 = ",
         rows, " rows, got ", x.sizes(), " / ", out.sizes());
  SF_ARG(w.dim() == 2 && w.size(0) == 2304 && w.size(1) == 768, name, "w is (2304, 768), got ", w.sizes());
  SF_ARG(side.dim() == 2 && side.size(1) == 2304 && side.size(0) >= n_seq * 33, name, "side is (>= n_seq * 33, 2304), got ", side.sizes());
  SF_ARG(partials.is_contiguous() && partials.numel() >= n_seq * 12 * n_part * 66, name, "partials holds n_seq * 12 * ", n_part, " * 66 floats, got ", partials.numel());
  SF_ARG(out.data_ptr() != x.data_ptr(), name, "out must not alias x (other work items still read x)");
  check_bias(bias, 2304, name);
}
void check_key_keep(const Tensor& key_keep, int64_t rows, const char* name) {
  SF_ARG(is_u8(key_keep) && key_keep.is_contiguous() && key_keep.numel() >= rows, name, "key_keep is contiguous uint8, one flag per row of x (", rows, ")");
}
void check_packed_qkv(const char* name, const Tensor& q, const Tensor& k, const Tensor& v) {
  SF_ARG(is_bf16(q) && is_bf16(k) && is_bf16(v) && ld(q) == ld(k) && ld(q) == ld(v), name, "q / k / v are bf16 column slices of one packed projection");
}
// round 3's temporal launch: x / out (>= n_seq * (1 + 8 n_groups), 768), qkv_cls (>= n_seq, 2304), one partial record per 4 groups
void check_qkv_time_r3(const char* name, const Tensor& x, const Tensor& w, const OptTensor& bias, const Tensor& qkv_cls, const Tensor& out, const Tensor& partials, int64_t n_seq,
                       int64_t n_groups, bool mx_in, bool mx_out) {
  const int64_t rows = n_seq * (1 + 8 * n_groups);
  SF_ARG(n_seq > 0 && n_groups > 0 && n_groups % 4 == 0, name, "n_seq > 0, n_groups a positive multiple of 4");
  SF_ARG((mx_in ? is_u8(x) && is_u8(w) : is_bf16(x) && is_bf16(w)) && is_bf16(qkv_cls) && (mx_out ? is_u8(out) : is_bf16(out)) && is_f32(partials), name, "operand / output dtypes");
  SF_ARG(x.dim() == 2 && x.size(1) == 768 && x.size(0) >= rows && out.dim() == 2 && out.size(1) == 768 && out.size(0) >= rows, name, "x / out have 768 columns and at least ", rows, " rows");
  SF_ARG(w.dim() == 2 && w.size(0) == 2304 && w.size(1) == 768 && qkv_cls.dim() == 2 && qkv_cls.size(0) >= n_seq && qkv_cls.size(1) == 2304, name, "w (2304, 768), qkv_cls (>= n_seq, 2304)");
  SF_ARG(partials.is_contiguous() && partials.numel() >= n_seq * 12 * (n_groups / 4) * 66, name, "partials holds n_seq * 12 * (n_groups / 4) * 66 floats");
  check_bias(bias, 2304, name);
}

// ---- GEMM / LayerNorm ------------------------------------------------------------------------------------------------------------------------
void gemm_bf16(const Tensor& a, const Tensor& w, const OptTensor& bias, Tensor& out, const OptTensor& residual, bool gelu) {
  TORCH_CHECK(a.scalar_type() == at::kBFloat16 && w.scalar_type() == at::kBFloat16, "synchformer::gemm_bf16: bf16 operands");
  int64_t N, K, ldw;
  if (w.dim() == 3) {                                             // k-tile-major weight (K / 64, N, 64), ops.ktile_major_weight()
    TORCH_CHECK(w.size(2) == 64 && w.is_contiguous(), "synchformer::gemm_bf16: a k-tile-major weight is (K / 64, N, 64) contiguous");
    K = w.size(0) * 64; N = w.size(1); ldw = 64;
  } else { N = w.size(0); K = w.size(1); ldw = ld(w); }
  TORCH_CHECK(a.size(1) == K, "synchformer::gemm_bf16: a (M, K) against w (N, K)");
  check_bias(bias, N, "gemm_bf16");
  SF_ARG(out.dim() == 2 && out.size(0) >= a.size(0) && out.size(1) >= N, "gemm_bf16", "out is (>= M, >= N), got ", out.sizes());
  if (residual.has_value()) SF_ARG(is_f32(*residual) && residual->dim() == 2 && residual->size(0) >= a.size(0) && residual->size(1) >= N, "gemm_bf16", "fp32 residual (>= M, >= N)");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard1_(a.device());   // the C ABI launches on the CURRENT device
  check(sf_gemm_bf16(bf(dev(a, "a")), ld(a), bf(dev(w, "w")), ldw, f32(devo(bias, "bias")), dev(out, "out"), dtype_code(out), ld(out), nullptr,
                     f32(devo(residual, "residual")), residual.has_value() ? ld(*residual) : 0, nullptr, gelu ? SF_EPI_GELU : SF_EPI_NONE, a.size(0), N, K, stream_of(a)),
        "sf_gemm_bf16");
}

void layernorm768(const Tensor& x, const Tensor& gamma, const Tensor& beta, Tensor& out, double eps) {
  SF_ARG(is_f32(gamma) && is_f32(beta) && gamma.numel() >= 768 && beta.numel() >= 768 && out.size(0) >= x.size(0), "layernorm768", "fp32 gamma / beta of 768, out rows >= x rows");
  TORCH_CHECK(x.scalar_type() == at::kFloat && x.size(1) == 768 && out.size(1) == 768, "synchformer::layernorm768: fp32 (rows, 768) in, 768 columns out");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard2_(x.device());   // the C ABI launches on the CURRENT device
  check(sf_layernorm768(f32(dev(x, "x")), ld(x), nullptr, f32(dev(gamma, "gamma")), f32(dev(beta, "beta")), dev(out, "out"), dtype_code(out), ld(out), nullptr, 0, x.size(0),
                        (float)eps, stream_of(x)),
        "sf_layernorm768");
}

void gemm_res_ln768(const Tensor& a, const Tensor& w, const OptTensor& bias, Tensor& x, const Tensor& gamma, const Tensor& beta, Tensor& y, double eps) {
  TORCH_CHECK(a.scalar_type() == at::kBFloat16 && w.scalar_type() == at::kBFloat16 && x.scalar_type() == at::kFloat && y.scalar_type() == at::kBFloat16,
              "synchformer::gemm_res_ln768: a / w / y bf16, x fp32");
  const int64_t K = a.size(1);
  int64_t ldw;
  if (w.dim() == 3) {                                             // k-step-major weight (K / 32, 768, 32), ops.kmajor_weight()
    TORCH_CHECK(w.size(0) == K / 32 && w.size(1) == 768 && w.size(2) == 32 && w.is_contiguous(), "synchformer::gemm_res_ln768: a k-step-major weight is (K / 32, 768, 32) contiguous");
    ldw = 32;
  } else { TORCH_CHECK(w.size(0) == 768 && w.size(1) == K, "synchformer::gemm_res_ln768: w (768, K)"); ldw = ld(w); }
  TORCH_CHECK(x.size(1) == 768 && y.size(1) == 768, "synchformer::gemm_res_ln768: 768 columns");
  SF_ARG(x.size(0) >= a.size(0) && y.size(0) >= a.size(0) && is_f32(gamma) && is_f32(beta) && gamma.numel() >= 768 && beta.numel() >= 768, "gemm_res_ln768", "x / y rows >= a rows, fp32 gamma / beta of 768");
  check_bias(bias, 768, "gemm_res_ln768");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard3_(a.device());   // the C ABI launches on the CURRENT device
  check(sf_gemm_res_ln768(bf(dev(a, "a")), ld(a), bf(dev(w, "w")), ldw, f32(devo(bias, "bias")), f32(dev(x, "x")), ld(x), f32m(dev(x, "x")), ld(x), f32(dev(gamma, "gamma")),
                          f32(dev(beta, "beta")), (float)eps, bfm(dev(y, "y")), ld(y), a.size(0), K, stream_of(a)),
        "sf_gemm_res_ln768");
}

// ---- attention ---------------------------------------------------------------------------------------------------------------------------------
void attention(const Tensor& q, const Tensor& k, const Tensor& v, Tensor& out, int64_t n_seq, int64_t seq_rows, int64_t n_groups, int64_t row0, int64_t group_stride,
               int64_t tok_stride, int64_t n_tok, int64_t cls_row, int64_t heads, int64_t head_dim, double scale) {
  check_packed_qkv("attention", q, k, v);
  SF_ARG(is_bf16(out) && out.size(0) >= n_seq * seq_rows && q.size(0) >= n_seq * seq_rows, "attention", "bf16 out, q / out rows >= n_seq * seq_rows");
  TORCH_CHECK(ld(q) == ld(k) && ld(q) == ld(v), "synchformer::attention: q / k / v are column slices of one packed projection");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard4_(q.device());   // the C ABI launches on the CURRENT device
  check(sf_attention(bf(dev(q, "q")), bf(dev(k, "k")), bf(dev(v, "v")), ld(q), bfm(dev(out, "out")), ld(out), n_seq, seq_rows, (int)n_groups, (int)row0, (int)group_stride,
                     (int)tok_stride, (int)n_tok, (int)cls_row, (int)heads, (int)head_dim, (float)scale, stream_of(q)),
        "sf_attention");
}

void attention_cls(const Tensor& q, const Tensor& k, const Tensor& v, Tensor& out, int64_t n_seq, int64_t q_seq_rows, int64_t q_row, int64_t kv_seq_rows, int64_t kv_row0,
                   int64_t n_keys, int64_t out_seq_rows, int64_t out_row, int64_t heads, int64_t head_dim, double scale) {
  check_packed_qkv("attention_cls", q, k, v);
  SF_ARG(is_bf16(out) && out.size(0) >= n_seq * out_seq_rows && q.size(0) >= n_seq * q_seq_rows && k.size(0) >= n_seq * kv_seq_rows, "attention_cls", "bf16 out; q / k / out row counts");
  TORCH_CHECK(ld(q) == ld(k) && ld(q) == ld(v), "synchformer::attention_cls: q / k / v are column slices of one packed projection");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard5_(q.device());   // the C ABI launches on the CURRENT device
  check(sf_attention_cls(bf(dev(q, "q")), q_seq_rows, (int)q_row, bf(dev(k, "k")), bf(dev(v, "v")), ld(q), kv_seq_rows, (int)kv_row0, (int)n_keys, bfm(dev(out, "out")), ld(out),
                         out_seq_rows, (int)out_row, n_seq, (int)heads, (int)head_dim, (float)scale, stream_of(q)),
        "sf_attention_cls");
}

void attention_cls_partial(const Tensor& q, const Tensor& k, const Tensor& v, Tensor& out, Tensor& partials, int64_t n_seq, int64_t seq_rows, int64_t n_groups, int64_t row0,
                           int64_t group_stride, int64_t tok_stride, int64_t n_tok, int64_t cls_row, int64_t heads, int64_t head_dim, double scale, const OptTensor& key_keep) {
  check_packed_qkv("attention_cls_partial", q, k, v);
  SF_ARG(is_bf16(out) && out.size(0) >= n_seq * seq_rows && q.size(0) >= n_seq * seq_rows && partials.is_contiguous(), "attention_cls_partial", "bf16 out, q / out rows >= n_seq * seq_rows, contiguous partials");
  TORCH_CHECK(ld(q) == ld(k) && ld(q) == ld(v) && partials.scalar_type() == at::kFloat && partials.numel() >= n_seq * heads * n_groups * 66,
              "synchformer::attention_cls_partial: packed q / k / v, fp32 partials of n_seq * heads * n_groups * 66 elements");
  if (key_keep.has_value()) {
    TORCH_CHECK(key_keep->scalar_type() == at::kByte && key_keep->numel() >= n_seq * seq_rows, "synchformer::attention_cls_partial: key_keep is uint8, one flag per row");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard6_(q.device());   // the C ABI launches on the CURRENT device
    check(sf_attention_cls_partial_masked(bf(dev(q, "q")), bf(dev(k, "k")), bf(dev(v, "v")), ld(q), bfm(dev(out, "out")), ld(out), n_seq, seq_rows, (int)n_groups, (int)row0,
                                          (int)group_stride, (int)tok_stride, (int)n_tok, (int)cls_row, (int)heads, (int)head_dim, (float)scale, f32m(dev(partials, "partials")),
                                          u8(dev(*key_keep, "key_keep")), stream_of(q)),
          "sf_attention_cls_partial_masked");
    return;
  }
  const c10::hip::HIPGuardMasqueradingAsCUDA guard7_(q.device());   // the C ABI launches on the CURRENT device
  check(sf_attention_cls_partial(bf(dev(q, "q")), bf(dev(k, "k")), bf(dev(v, "v")), ld(q), bfm(dev(out, "out")), ld(out), n_seq, seq_rows, (int)n_groups, (int)row0,
                                 (int)group_stride, (int)tok_stride, (int)n_tok, (int)cls_row, (int)heads, (int)head_dim, (float)scale, f32m(dev(partials, "partials")), stream_of(q)),
        "sf_attention_cls_partial");
}

void attention_cls_combine(const Tensor& partials, Tensor& out, int64_t n_part, int64_t n_seq, int64_t out_seq_rows, int64_t out_row, int64_t heads) {
  SF_ARG(is_f32(partials) && partials.is_contiguous() && partials.numel() >= n_seq * heads * n_part * 66 && is_bf16(out) && out.size(0) >= (n_seq - 1) * out_seq_rows + out_row + 1 && out.size(1) >= heads * 64,
         "attention_cls_combine", "fp32 partials of n_seq * heads * n_part * 66, bf16 out covering row (n_seq - 1) * out_seq_rows + out_row");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard8_(out.device());   // the C ABI launches on the CURRENT device
  check(sf_attention_cls_combine(f32(dev(partials, "partials")), (int)n_part, bfm(dev(out, "out")), ld(out), out_seq_rows, (int)out_row, n_seq, (int)heads, stream_of(out)),
        "sf_attention_cls_combine");
}

void im2col_video(const Tensor& vid, Tensor& out) {
  TORCH_CHECK(vid.is_contiguous() && vid.dim() == 5 && vid.size(1) == 16 && vid.size(2) == 3 && vid.size(3) == 224 && vid.size(4) == 224, "synchformer::im2col_video: (n, 16, 3, 224, 224) contiguous");
  TORCH_CHECK(out.scalar_type() == at::kBFloat16 && out.is_contiguous() && out.size(1) == 1536, "synchformer::im2col_video: out bf16 (n * 1568, 1536) contiguous");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard9_(vid.device());   // the C ABI launches on the CURRENT device
  check(sf_im2col_video(dev(vid, "vid"), dtype_code(vid), bfm(dev(out, "out")), vid.size(0), stream_of(vid)), "sf_im2col_video");
}

// ---- the fused temporal / spatial halves of DividedSpaceTimeBlock ---------------------------------------------------------------------------------
void qkv_time_attention(const Tensor& x, const Tensor& w, const OptTensor& bias, const Tensor& qkv_cls, Tensor& out, Tensor& partials, int64_t n_seq, int64_t n_groups,
                        double scale, const OptTensor& key_keep) {
  check_qkv_time_r3("qkv_time_attention", x, w, bias, qkv_cls, out, partials, n_seq, n_groups, false, false);
  if (key_keep.has_value()) check_key_keep(*key_keep, n_seq * (1 + 8 * n_groups), "qkv_time_attention");
  if (key_keep.has_value()) {
    const c10::hip::HIPGuardMasqueradingAsCUDA guard10_(x.device());   // the C ABI launches on the CURRENT device
    check(sf_qkv_time_attention_masked(bf(dev(x, "x")), ld(x), bf(dev(w, "w")), ld(w), f32(devo(bias, "bias")), bf(dev(qkv_cls, "qkv_cls")), ld(qkv_cls), bfm(dev(out, "out")),
                                       ld(out), f32m(dev(partials, "partials")), n_seq, (int)n_groups, (float)scale, u8(dev(*key_keep, "key_keep")), stream_of(x)),
          "sf_qkv_time_attention_masked");
    return;
  }
  const c10::hip::HIPGuardMasqueradingAsCUDA guard11_(x.device());   // the C ABI launches on the CURRENT device
  check(sf_qkv_time_attention(bf(dev(x, "x")), ld(x), bf(dev(w, "w")), ld(w), f32(devo(bias, "bias")), bf(dev(qkv_cls, "qkv_cls")), ld(qkv_cls), bfm(dev(out, "out")), ld(out),
                              f32m(dev(partials, "partials")), n_seq, (int)n_groups, (float)scale, stream_of(x)),
        "sf_qkv_time_attention");
}

void qkv_time_attention2(const Tensor& x, const Tensor& w, const OptTensor& bias, const Tensor& side, Tensor& out, Tensor& partials, int64_t n_seq, double scale) {
  check_qkv_fused("qkv_time_attention2", x, w, bias, side, out, partials, n_seq, 33, false, false);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard12_(x.device());   // the C ABI launches on the CURRENT device
  check(sf_qkv_time_attention2(bf(dev(x, "x")), ld(x), bf(dev(w, "w")), ld(w), f32(devo(bias, "bias")), bf(dev(side, "side")), ld(side), bfm(dev(out, "out")), ld(out),
                               f32m(dev(partials, "partials")), n_seq, 196, (float)scale, stream_of(x)),
        "sf_qkv_time_attention2");
}

void qkv_space_attention(const Tensor& x, const Tensor& w, const OptTensor& bias, const Tensor& side, Tensor& out, Tensor& partials, int64_t n_seq, double scale) {
  check_qkv_fused("qkv_space_attention", x, w, bias, side, out, partials, n_seq, 8, false, false);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard13_(x.device());   // the C ABI launches on the CURRENT device
  check(sf_qkv_space_attention(bf(dev(x, "x")), ld(x), bf(dev(w, "w")), ld(w), f32(devo(bias, "bias")), bf(dev(side, "side")), ld(side), bfm(dev(out, "out")), ld(out),
                               f32m(dev(partials, "partials")), n_seq, 196, (float)scale, stream_of(x)),
        "sf_qkv_space_attention");
}

void qkv_time_attention2_masked(const Tensor& x, const Tensor& w, const OptTensor& bias, const Tensor& side, Tensor& out, Tensor& partials, int64_t n_seq, double scale,
                                const Tensor& key_keep) {
  check_qkv_fused("qkv_time_attention2_masked", x, w, bias, side, out, partials, n_seq, 33, false, false);
  check_key_keep(key_keep, n_seq * 1569, "qkv_time_attention2_masked");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard14_(x.device());   // the C ABI launches on the CURRENT device
  check(sf_qkv_time_attention2_masked(bf(dev(x, "x")), ld(x), bf(dev(w, "w")), ld(w), f32(devo(bias, "bias")), bf(dev(side, "side")), ld(side), bfm(dev(out, "out")), ld(out),
                                      f32m(dev(partials, "partials")), n_seq, 196, (float)scale, u8(dev(key_keep, "key_keep")), stream_of(x)),
        "sf_qkv_time_attention2_masked");
}

void qkv_space_attention_masked(const Tensor& x, const Tensor& w, const OptTensor& bias, const Tensor& side, Tensor& out, Tensor& partials, int64_t n_seq, double scale,
                                const Tensor& key_keep) {
  check_qkv_fused("qkv_space_attention_masked", x, w, bias, side, out, partials, n_seq, 8, false, false);
  check_key_keep(key_keep, n_seq * 1569, "qkv_space_attention_masked");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard15_(x.device());   // the C ABI launches on the CURRENT device
  check(sf_qkv_space_attention_masked(bf(dev(x, "x")), ld(x), bf(dev(w, "w")), ld(w), f32(devo(bias, "bias")), bf(dev(side, "side")), ld(side), bfm(dev(out, "out")), ld(out),
                                      f32m(dev(partials, "partials")), n_seq, 196, (float)scale, u8(dev(key_keep, "key_keep")), stream_of(x)),
        "sf_qkv_space_attention_masked");
}

void space_side_rows(const Tensor& x, Tensor& out, int64_t n_seq) {
  TORCH_CHECK(x.scalar_type() == at::kBFloat16 && out.scalar_type() == at::kBFloat16 && x.size(1) == 768 && out.size(1) == 768, "synchformer::space_side_rows: bf16 (rows, 768)");
  SF_ARG(n_seq > 0 && x.size(0) >= n_seq * 1569 && out.size(0) >= n_seq * 33, "space_side_rows", "x rows >= n_seq * 1569, out rows >= n_seq * 33");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard16_(x.device());   // the C ABI launches on the CURRENT device
  check(sf_side_rows(dev(x, "x"), ld(x) * 2, dev(out, "out"), ld(out) * 2, 1536, nullptr, 0, nullptr, 0, 0, n_seq, 196, stream_of(x)), "sf_side_rows");
}

void space_side_rows_mx(const Tensor& x_q, const Tensor& x_s, Tensor& side_q, Tensor& side_s, int64_t n_seq) {
  SF_ARG(is_u8(x_q) && is_u8(side_q) && x_q.size(1) == 768 && side_q.size(1) == 768 && n_seq > 0 && x_q.size(0) >= n_seq * 1569 && side_q.size(0) >= n_seq * 33, "space_side_rows_mx", "uint8 (rows, 768) operands; row counts");
  check_scale_planes(x_s, 6, n_seq * 1569, "space_side_rows_mx", "x_s");
  check_scale_planes(side_s, 6, n_seq * 33, "space_side_rows_mx", "side_s");
  TORCH_CHECK(x_s.dim() == 3 && side_s.dim() == 3 && x_s.size(0) == 6 && side_s.size(0) == 6 && x_s.is_contiguous() && side_s.is_contiguous(),
              "synchformer::space_side_rows_mx: scale planes (6, rows, 4) contiguous");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard17_(x_q.device());   // the C ABI launches on the CURRENT device
  check(sf_side_rows(dev(x_q, "x_q"), ld(x_q), dev(side_q, "side_q"), ld(side_q), 768, u8(dev(x_s, "x_s")), x_s.stride(0), u8m(dev(side_s, "side_s")), side_s.stride(0), 6, n_seq,
                     196, stream_of(x_q)),
        "sf_side_rows");
}

// ---- MXFP8 (fp8 towers) -------------------------------------------------------------------------------------------------------------------------
void quantize_mxfp8(const Tensor& x, Tensor& q, Tensor& scales) {
  SF_ARG(q.size(0) >= x.size(0) && q.size(1) == x.size(1) && x.size(1) % 128 == 0, "quantize_mxfp8", "q (>= rows, K), K a multiple of 128");
  check_scale_planes(scales, x.size(1) / 128, x.size(0), "quantize_mxfp8", "scales");
  TORCH_CHECK(x.scalar_type() == at::kBFloat16 && q.scalar_type() == at::kByte && scales.scalar_type() == at::kByte && scales.dim() == 3 && scales.is_contiguous(),
              "synchformer::quantize_mxfp8: x bf16, q uint8, scales uint8 (K / 128, rows, 4) contiguous");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard18_(x.device());   // the C ABI launches on the CURRENT device
  check(sf_quantize_mxfp8(bf(dev(x, "x")), ld(x), u8m(dev(q, "q")), ld(q), u8m(dev(scales, "scales")), scales.stride(0), x.size(0), x.size(1), stream_of(x)), "sf_quantize_mxfp8");
}

void layernorm768_mxfp8(const Tensor& x, const Tensor& gamma, const Tensor& beta, Tensor& q, Tensor& scales, double eps) {
  SF_ARG(is_f32(gamma) && is_f32(beta) && gamma.numel() >= 768 && beta.numel() >= 768 && q.size(0) >= x.size(0), "layernorm768_mxfp8", "fp32 gamma / beta of 768, q rows >= x rows");
  check_scale_planes(scales, 6, x.size(0), "layernorm768_mxfp8", "scales");
  TORCH_CHECK(x.scalar_type() == at::kFloat && x.size(1) == 768 && q.scalar_type() == at::kByte && q.size(1) == 768 && scales.dim() == 3 && scales.is_contiguous(),
              "synchformer::layernorm768_mxfp8: x fp32 (rows, 768), q uint8 (rows, 768), scales (6, rows, 4) contiguous");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard19_(x.device());   // the C ABI launches on the CURRENT device
  check(sf_layernorm768_mxfp8(f32(dev(x, "x")), ld(x), f32(dev(gamma, "gamma")), f32(dev(beta, "beta")), u8m(dev(q, "q")), ld(q), u8m(dev(scales, "scales")), scales.stride(0),
                              x.size(0), (float)eps, stream_of(x)),
        "sf_layernorm768_mxfp8");
}

void gemm_mxfp8(const Tensor& a_q, const Tensor& a_s, const Tensor& w_q, const Tensor& w_s, const OptTensor& bias, Tensor& out, const OptTensor& out_scales,
                const OptTensor& residual, bool gelu) {
  TORCH_CHECK((out.scalar_type() == at::kByte) == out_scales.has_value(), "synchformer::gemm_mxfp8: a uint8 out comes with out_scales");
  const int64_t N = w_q.size(0), K = w_q.size(1);
  TORCH_CHECK(a_q.size(1) == K && a_s.dim() == 3 && w_s.dim() == 3 && a_s.size(0) == K / 128 && w_s.size(0) == K / 128 && a_s.is_contiguous() && w_s.is_contiguous(),
              "synchformer::gemm_mxfp8: operands (rows, K) uint8 with stage-major scale planes (K / 128, rows, 4)");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard20_(a_q.device());   // the C ABI launches on the CURRENT device
  SF_ARG(is_u8(a_q) && is_u8(w_q) && K % 128 == 0 && out.dim() == 2 && out.size(0) >= a_q.size(0) && out.size(1) >= N, "gemm_mxfp8", "uint8 operands, K a multiple of 128, out (>= M, >= N)");
  check_scale_planes(a_s, K / 128, a_q.size(0), "gemm_mxfp8", "a_s");
  check_scale_planes(w_s, K / 128, N, "gemm_mxfp8", "w_s");
  if (out_scales.has_value()) check_scale_planes(*out_scales, N / 128, a_q.size(0), "gemm_mxfp8", "out_scales");
  if (residual.has_value()) SF_ARG(is_f32(*residual) && residual->dim() == 2 && residual->size(0) >= a_q.size(0) && residual->size(1) >= N, "gemm_mxfp8", "fp32 residual (>= M, >= N)");
  check_bias(bias, N, "gemm_mxfp8");
  check(sf_gemm_mxfp8(u8(dev(a_q, "a_q")), ld(a_q), u8(dev(a_s, "a_s")), a_s.stride(0), u8(dev(w_q, "w_q")), ld(w_q), u8(dev(w_s, "w_s")), w_s.stride(0), f32(devo(bias, "bias")),
                      dev(out, "out"), dtype_code(out), ld(out), u8m(devo(out_scales, "out_scales")), out_scales.has_value() ? out_scales->stride(0) : 0,
                      f32(devo(residual, "residual")), residual.has_value() ? ld(*residual) : 0, gelu ? SF_EPI_GELU : SF_EPI_NONE, a_q.size(0), N, K, stream_of(a_q)),
        "sf_gemm_mxfp8");
}

void gemm_mx_res_ln768(const Tensor& a_q, const Tensor& a_s, const Tensor& w_q, const Tensor& w_s, const OptTensor& bias, Tensor& x, const Tensor& gamma, const Tensor& beta,
                       Tensor& y_q, Tensor& y_s, double eps) {
  const int64_t K = a_q.size(1);
  TORCH_CHECK(x.scalar_type() == at::kFloat && x.size(1) == 768 && y_q.size(1) == 768 && w_q.size(0) == 768 && w_q.size(1) == K && a_s.dim() == 3 && w_s.dim() == 3 && y_s.dim() == 3 &&
                  a_s.is_contiguous() && w_s.is_contiguous() && y_s.is_contiguous(),
              "synchformer::gemm_mx_res_ln768: x fp32 (rows, 768), w (768, K), contiguous scale planes");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard21_(a_q.device());   // the C ABI launches on the CURRENT device
  SF_ARG(is_u8(a_q) && is_u8(w_q) && is_u8(y_q) && K % 128 == 0 && x.size(0) >= a_q.size(0) && y_q.size(0) >= a_q.size(0) && is_f32(gamma) && is_f32(beta) && gamma.numel() >= 768 && beta.numel() >= 768,
         "gemm_mx_res_ln768", "uint8 operands / output, x / y_q rows >= a_q rows, fp32 gamma / beta of 768");
  check_scale_planes(a_s, K / 128, a_q.size(0), "gemm_mx_res_ln768", "a_s");
  check_scale_planes(w_s, K / 128, 768, "gemm_mx_res_ln768", "w_s");
  check_scale_planes(y_s, 6, a_q.size(0), "gemm_mx_res_ln768", "y_s");
  check_bias(bias, 768, "gemm_mx_res_ln768");
  check(sf_gemm_mx_res_ln768(u8(dev(a_q, "a_q")), ld(a_q), u8(dev(a_s, "a_s")), a_s.stride(0), u8(dev(w_q, "w_q")), ld(w_q), u8(dev(w_s, "w_s")), w_s.stride(0),
                             f32(devo(bias, "bias")), f32(dev(x, "x")), ld(x), f32m(dev(x, "x")), ld(x), f32(dev(gamma, "gamma")), f32(dev(beta, "beta")), (float)eps,
                             u8m(dev(y_q, "y_q")), ld(y_q), u8m(dev(y_s, "y_s")), y_s.stride(0), a_q.size(0), K, stream_of(a_q)),
        "sf_gemm_mx_res_ln768");
}

void qkv_time_attention_mx(const Tensor& x_q, const Tensor& x_s, const Tensor& w_q, const Tensor& w_s, const OptTensor& bias, const Tensor& qkv_cls, Tensor& out, Tensor& partials,
                           int64_t n_seq, int64_t n_groups, double scale) {
  check_qkv_time_r3("qkv_time_attention_mx", x_q, w_q, bias, qkv_cls, out, partials, n_seq, n_groups, true, false);
  check_scale_planes(x_s, 6, n_seq * (1 + 8 * n_groups), "qkv_time_attention_mx", "x_s");
  check_scale_planes(w_s, 6, 2304, "qkv_time_attention_mx", "w_s");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard22_(x_q.device());   // the C ABI launches on the CURRENT device
  check(sf_qkv_time_attention_mx(u8(dev(x_q, "x_q")), ld(x_q), u8(dev(x_s, "x_s")), x_s.stride(0), u8(dev(w_q, "w_q")), ld(w_q), u8(dev(w_s, "w_s")), w_s.stride(0),
                                 f32(devo(bias, "bias")), bf(dev(qkv_cls, "qkv_cls")), ld(qkv_cls), bfm(dev(out, "out")), ld(out), f32m(dev(partials, "partials")), n_seq, (int)n_groups,
                                 (float)scale, stream_of(x_q)),
        "sf_qkv_time_attention_mx");
}

void qkv_time_attention_mx_q(const Tensor& x_q, const Tensor& x_s, const Tensor& w_q, const Tensor& w_s, const OptTensor& bias, const Tensor& qkv_cls, Tensor& out_q, Tensor& out_s,
                             Tensor& partials, int64_t n_seq, int64_t n_groups, double scale) {
  check_qkv_time_r3("qkv_time_attention_mx_q", x_q, w_q, bias, qkv_cls, out_q, partials, n_seq, n_groups, true, true);
  check_scale_planes(x_s, 6, n_seq * (1 + 8 * n_groups), "qkv_time_attention_mx_q", "x_s");
  check_scale_planes(w_s, 6, 2304, "qkv_time_attention_mx_q", "w_s");
  check_scale_planes(out_s, 6, n_seq * (1 + 8 * n_groups), "qkv_time_attention_mx_q", "out_s");
  TORCH_CHECK(out_q.data_ptr() != x_q.data_ptr() && out_s.data_ptr() != x_s.data_ptr() && out_s.dim() == 3 && out_s.is_contiguous(), "synchformer::qkv_time_attention_mx_q: out_q / out_s are buffers of their own");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard23_(x_q.device());   // the C ABI launches on the CURRENT device
  check(sf_qkv_time_attention_mx_q(u8(dev(x_q, "x_q")), ld(x_q), u8(dev(x_s, "x_s")), x_s.stride(0), u8(dev(w_q, "w_q")), ld(w_q), u8(dev(w_s, "w_s")), w_s.stride(0),
                                   f32(devo(bias, "bias")), bf(dev(qkv_cls, "qkv_cls")), ld(qkv_cls), u8m(dev(out_q, "out_q")), ld(out_q), u8m(dev(out_s, "out_s")), out_s.stride(0),
                                   f32m(dev(partials, "partials")), n_seq, (int)n_groups, (float)scale, stream_of(x_q)),
        "sf_qkv_time_attention_mx_q");
}

void attention_cls_partial_mx(const Tensor& q, const Tensor& k, const Tensor& v, Tensor& out_q, Tensor& out_s, Tensor& partials, int64_t n_seq, int64_t seq_rows, int64_t n_groups,
                              int64_t row0, int64_t group_stride, int64_t tok_stride, int64_t n_tok, int64_t cls_row, int64_t heads, double scale) {
  check_packed_qkv("attention_cls_partial_mx", q, k, v);
  SF_ARG(is_u8(out_q) && is_u8(out_s) && is_f32(partials) && partials.is_contiguous() && partials.numel() >= n_seq * heads * n_groups * 66 && out_q.size(0) >= n_seq * seq_rows && out_s.size(1) >= n_seq * seq_rows,
         "attention_cls_partial_mx", "uint8 out_q / out_s covering n_seq * seq_rows rows, fp32 partials of n_seq * heads * n_groups * 66");
  TORCH_CHECK(ld(q) == ld(k) && ld(q) == ld(v) && out_s.dim() == 3 && out_s.size(0) * 2 == heads, "synchformer::attention_cls_partial_mx: packed q / k / v, scale planes (heads / 2, rows, 4)");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard24_(q.device());   // the C ABI launches on the CURRENT device
  check(sf_attention_cls_partial_mx(bf(dev(q, "q")), bf(dev(k, "k")), bf(dev(v, "v")), ld(q), u8m(dev(out_q, "out_q")), ld(out_q), u8m(dev(out_s, "out_s")), out_s.stride(0), n_seq,
                                    seq_rows, (int)n_groups, (int)row0, (int)group_stride, (int)tok_stride, (int)n_tok, (int)cls_row, (int)heads, (float)scale,
                                    f32m(dev(partials, "partials")), stream_of(q)),
        "sf_attention_cls_partial_mx");
}

void attention_cls_combine_mx(const Tensor& partials, Tensor& out_q, Tensor& out_s, int64_t n_part, int64_t n_seq, int64_t out_seq_rows, int64_t out_row, int64_t heads) {
  SF_ARG(is_f32(partials) && partials.is_contiguous() && partials.numel() >= n_seq * heads * n_part * 66 && is_u8(out_q) && is_u8(out_s) && out_q.size(0) >= (n_seq - 1) * out_seq_rows + out_row + 1,
         "attention_cls_combine_mx", "fp32 partials of n_seq * heads * n_part * 66, uint8 outputs covering row (n_seq - 1) * out_seq_rows + out_row");
  TORCH_CHECK(out_s.dim() == 3 && out_s.size(0) * 2 == heads, "synchformer::attention_cls_combine_mx: scale planes (heads / 2, rows, 4)");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard25_(out_q.device());   // the C ABI launches on the CURRENT device
  check(sf_attention_cls_combine_mx(f32(dev(partials, "partials")), (int)n_part, u8m(dev(out_q, "out_q")), ld(out_q), u8m(dev(out_s, "out_s")), out_s.stride(0), out_seq_rows,
                                    (int)out_row, n_seq, (int)heads, stream_of(out_q)),
        "sf_attention_cls_combine_mx");
}

// one body for the four MXFP8 forms of the fused attention halves: bf16 output (out_s absent) or MXFP8 output
template <class F>
void qkv_mx_launch(F fn, const char* what, const Tensor& x_q, const Tensor& x_s, const Tensor& w_q, const Tensor& w_s, const OptTensor& bias, const Tensor& side, Tensor& out,
                   Tensor* out_s, Tensor& partials, int64_t n_seq, double scale, int64_t n_part) {
  check_qkv_fused(what, x_q, w_q, bias, side, out, partials, n_seq, n_part, true, out_s != nullptr);
  check_scale_planes(x_s, 6, n_seq * 1569, what, "x_s");
  check_scale_planes(w_s, 6, 2304, what, "w_s");
  if (out_s != nullptr) check_scale_planes(*out_s, 6, n_seq * 1569, what, "out_s");
  const bool q = out_s != nullptr;
  if (q) {
    TORCH_CHECK(out.scalar_type() == at::kByte && out_s->dim() == 3 && out_s->is_contiguous() && out.data_ptr() != x_q.data_ptr() && out_s->data_ptr() != x_s.data_ptr(),
                what, ": the MXFP8 output is a uint8 buffer + contiguous scale planes of its own");
  } else {
    TORCH_CHECK(out.scalar_type() == at::kBFloat16, what, ": the bf16 output form");
  }
  const c10::hip::HIPGuardMasqueradingAsCUDA guard26_(x_q.device());   // the C ABI launches on the CURRENT device
  check(fn(u8(dev(x_q, "x_q")), ld(x_q), u8(dev(x_s, "x_s")), x_s.stride(0), u8(dev(w_q, "w_q")), ld(w_q), u8(dev(w_s, "w_s")), w_s.stride(0), f32(devo(bias, "bias")),
           bf(dev(side, "side")), ld(side), q ? nullptr : bfm(dev(out, "out")), q ? 0 : ld(out), q ? u8m(dev(out, "out")) : nullptr, q ? ld(out) : 0,
           q ? u8m(dev(*out_s, "out_s")) : nullptr, q ? out_s->stride(0) : 0, f32m(dev(partials, "partials")), n_seq, 196, (float)scale, stream_of(x_q)),
        what);
}
void qkv_space_attention_mx(const Tensor& x_q, const Tensor& x_s, const Tensor& w_q, const Tensor& w_s, const OptTensor& bias, const Tensor& side, Tensor& out, Tensor& partials,
                            int64_t n_seq, double scale) {
  qkv_mx_launch(sf_qkv_space_attention_mx, "sf_qkv_space_attention_mx", x_q, x_s, w_q, w_s, bias, side, out, nullptr, partials, n_seq, scale, 8);
}
void qkv_space_attention_mx_q(const Tensor& x_q, const Tensor& x_s, const Tensor& w_q, const Tensor& w_s, const OptTensor& bias, const Tensor& side, Tensor& out_q, Tensor& out_s,
                              Tensor& partials, int64_t n_seq, double scale) {
  qkv_mx_launch(sf_qkv_space_attention_mx, "sf_qkv_space_attention_mx", x_q, x_s, w_q, w_s, bias, side, out_q, &out_s, partials, n_seq, scale, 8);
}
void qkv_time_attention2_mx(const Tensor& x_q, const Tensor& x_s, const Tensor& w_q, const Tensor& w_s, const OptTensor& bias, const Tensor& side, Tensor& out, Tensor& partials,
                            int64_t n_seq, double scale) {
  qkv_mx_launch(sf_qkv_time_attention2_mx, "sf_qkv_time_attention2_mx", x_q, x_s, w_q, w_s, bias, side, out, nullptr, partials, n_seq, scale, 33);
}
void qkv_time_attention2_mx_q(const Tensor& x_q, const Tensor& x_s, const Tensor& w_q, const Tensor& w_s, const OptTensor& bias, const Tensor& side, Tensor& out_q, Tensor& out_s,
                              Tensor& partials, int64_t n_seq, double scale) {
  qkv_mx_launch(sf_qkv_time_attention2_mx, "sf_qkv_time_attention2_mx", x_q, x_s, w_q, w_s, bias, side, out_q, &out_s, partials, n_seq, scale, 33);
}

}  // namespace

// Schemas: out-variant operators; (a!) .. mark the tensors a launch writes.  `Tensor?` = optional (None from Python).
TORCH_LIBRARY(synchformer, m) {
  m.def("gemm_bf16(Tensor a, Tensor w, Tensor? bias, Tensor(a!) out, Tensor? residual, bool gelu) -> ()");
  m.def("layernorm768(Tensor x, Tensor gamma, Tensor beta, Tensor(a!) out, float eps) -> ()");
  m.def("gemm_res_ln768(Tensor a, Tensor w, Tensor? bias, Tensor(a!) x, Tensor gamma, Tensor beta, Tensor(b!) y, float eps) -> ()");
  m.def("attention(Tensor q, Tensor k, Tensor v, Tensor(a!) out, int n_seq, int seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads, int head_dim, float scale) -> ()");
  m.def("attention_cls(Tensor q, Tensor k, Tensor v, Tensor(a!) out, int n_seq, int q_seq_rows, int q_row, int kv_seq_rows, int kv_row0, int n_keys, int out_seq_rows, int out_row, int heads, int head_dim, float scale) -> ()");
  m.def("attention_cls_partial(Tensor q, Tensor k, Tensor v, Tensor(a!) out, Tensor(b!) partials, int n_seq, int seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads, int head_dim, float scale, Tensor? key_keep) -> ()");
  m.def("attention_cls_combine(Tensor partials, Tensor(a!) out, int n_part, int n_seq, int out_seq_rows, int out_row, int heads) -> ()");
  m.def("im2col_video(Tensor vid, Tensor(a!) out) -> ()");
  m.def("qkv_time_attention(Tensor x, Tensor w, Tensor? bias, Tensor qkv_cls, Tensor(a!) out, Tensor(b!) partials, int n_seq, int n_groups, float scale, Tensor? key_keep) -> ()");
  m.def("qkv_time_attention2(Tensor x, Tensor w, Tensor? bias, Tensor side, Tensor(a!) out, Tensor(b!) partials, int n_seq, float scale) -> ()");
  m.def("qkv_space_attention(Tensor x, Tensor w, Tensor? bias, Tensor side, Tensor(a!) out, Tensor(b!) partials, int n_seq, float scale) -> ()");
  m.def("qkv_time_attention2_masked(Tensor x, Tensor w, Tensor? bias, Tensor side, Tensor(a!) out, Tensor(b!) partials, int n_seq, float scale, Tensor key_keep) -> ()");
  m.def("qkv_space_attention_masked(Tensor x, Tensor w, Tensor? bias, Tensor side, Tensor(a!) out, Tensor(b!) partials, int n_seq, float scale, Tensor key_keep) -> ()");
  m.def("space_side_rows(Tensor x, Tensor(a!) out, int n_seq) -> ()");
  m.def("space_side_rows_mx(Tensor x_q, Tensor x_s, Tensor(a!) side_q, Tensor(b!) side_s, int n_seq) -> ()");
  m.def("quantize_mxfp8(Tensor x, Tensor(a!) q, Tensor(b!) scales) -> ()");
  m.def("layernorm768_mxfp8(Tensor x, Tensor gamma, Tensor beta, Tensor(a!) q, Tensor(b!) scales, float eps) -> ()");
  m.def("gemm_mxfp8(Tensor a_q, Tensor a_s, Tensor w_q, Tensor w_s, Tensor? bias, Tensor(a!) out, Tensor(b!)? out_scales, Tensor? residual, bool gelu) -> ()");
  m.def("gemm_mx_res_ln768(Tensor a_q, Tensor a_s, Tensor w_q, Tensor w_s, Tensor? bias, Tensor(a!) x, Tensor gamma, Tensor beta, Tensor(b!) y_q, Tensor(c!) y_s, float eps) -> ()");
  m.def("qkv_time_attention_mx(Tensor x_q, Tensor x_s, Tensor w_q, Tensor w_s, Tensor? bias, Tensor qkv_cls, Tensor(a!) out, Tensor(b!) partials, int n_seq, int n_groups, float scale) -> ()");
  m.def("qkv_time_attention_mx_q(Tensor x_q, Tensor x_s, Tensor w_q, Tensor w_s, Tensor? bias, Tensor qkv_cls, Tensor(a!) out_q, Tensor(b!) out_s, Tensor(c!) partials, int n_seq, int n_groups, float scale) -> ()");
  m.def("attention_cls_partial_mx(Tensor q, Tensor k, Tensor v, Tensor(a!) out_q, Tensor(b!) out_s, Tensor(c!) partials, int n_seq, int seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads, float scale) -> ()");
  m.def("attention_cls_combine_mx(Tensor partials, Tensor(a!) out_q, Tensor(b!) out_s, int n_part, int n_seq, int out_seq_rows, int out_row, int heads) -> ()");
  m.def("qkv_space_attention_mx(Tensor x_q, Tensor x_s, Tensor w_q, Tensor w_s, Tensor? bias, Tensor side, Tensor(a!) out, Tensor(b!) partials, int n_seq, float scale) -> ()");
  m.def("qkv_space_attention_mx_q(Tensor x_q, Tensor x_s, Tensor w_q, Tensor w_s, Tensor? bias, Tensor side, Tensor(a!) out_q, Tensor(b!) out_s, Tensor(c!) partials, int n_seq, float scale) -> ()");
  m.def("qkv_time_attention2_mx(Tensor x_q, Tensor x_s, Tensor w_q, Tensor w_s, Tensor? bias, Tensor side, Tensor(a!) out, Tensor(b!) partials, int n_seq, float scale) -> ()");
  m.def("qkv_time_attention2_mx_q(Tensor x_q, Tensor x_s, Tensor w_q, Tensor w_s, Tensor? bias, Tensor side, Tensor(a!) out_q, Tensor(b!) out_s, Tensor(c!) partials, int n_seq, float scale) -> ()");
}

// PyTorch-ROCm files HIP tensors under the CUDA dispatch key: these are the HIP implementations (there is no other backend)
TORCH_LIBRARY_IMPL(synchformer, CUDA, m) {
  m.impl("gemm_bf16", gemm_bf16);
  m.impl("layernorm768", layernorm768);
  m.impl("gemm_res_ln768", gemm_res_ln768);
  m.impl("attention", attention);
  m.impl("attention_cls", attention_cls);
  m.impl("attention_cls_partial", attention_cls_partial);
  m.impl("attention_cls_combine", attention_cls_combine);
  m.impl("im2col_video", im2col_video);
  m.impl("qkv_time_attention", qkv_time_attention);
  m.impl("qkv_time_attention2", qkv_time_attention2);
  m.impl("qkv_space_attention", qkv_space_attention);
  m.impl("qkv_time_attention2_masked", qkv_time_attention2_masked);
  m.impl("qkv_space_attention_masked", qkv_space_attention_masked);
  m.impl("space_side_rows", space_side_rows);
  m.impl("space_side_rows_mx", space_side_rows_mx);
  m.impl("quantize_mxfp8", quantize_mxfp8);
  m.impl("layernorm768_mxfp8", layernorm768_mxfp8);
  m.impl("gemm_mxfp8", gemm_mxfp8);
  m.impl("gemm_mx_res_ln768", gemm_mx_res_ln768);
  m.impl("qkv_time_attention_mx", qkv_time_attention_mx);
  m.impl("qkv_time_attention_mx_q", qkv_time_attention_mx_q);
  m.impl("attention_cls_partial_mx", attention_cls_partial_mx);
  m.impl("attention_cls_combine_mx", attention_cls_combine_mx);
  m.impl("qkv_space_attention_mx", qkv_space_attention_mx);
  m.impl("qkv_space_attention_mx_q", qkv_space_attention_mx_q);
  m.impl("qkv_time_attention2_mx", qkv_time_attention2_mx);
  m.impl("qkv_time_attention2_mx_q", qkv_time_attention2_mx_q);
}
