// Probe (gfx950): what does v_cvt_scalef32_pk_fp8_f32 compute?  For inputs a, scale it prints the e4m3 byte of the instruction next to the bytes of
// cvt_pk_fp8(med3(a / scale, +-448)) and cvt_pk_fp8(med3(a * scale, +-448)) (the sequence sf_quantize_mxfp8 uses today).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(const float* a, const float* sc, int n, int* o) {
  const int i = threadIdx.x;
  if (i >= n) return;
  v2s old = {0, 0};
  const v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a[i], a[i], sc[i], false);
  const float d = a[i] / sc[i], m = a[i] * sc[i];
  const int rd = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(d, 448.f, -448.f), 0.f, 0, false);
  const int rm = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(m, 448.f, -448.f), 0.f, 0, false);
  o[i * 3] = r[0] & 0xff; o[i * 3 + 1] = rd & 0xff; o[i * 3 + 2] = rm & 0xff;
}
int main() {
  const int n = 12;
  float ha[n] = {1.f, 1.f, 1.f, 3.3f, 448.f, 1000.f, -1000.f, 0.017f, 5.5f, 5.5f, 1e-6f, 232.f};
  float hs[n] = {1.f, 2.f, 0.5f, 4.f, 1.f, 1.f, 2.f, 0.0078125f, 3.f, 6.f, 1.f, 0.5f};   // (3, 6: not powers of two - exponent only?)
  float *a, *s; int* o; int ho[3 * n];
  hipMalloc(&a, sizeof(ha)); hipMalloc(&s, sizeof(hs)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(s, hs, sizeof(hs), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, s, n, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("a %10g scale %10g : scalef32 %02x | a/scale %02x | a*scale %02x\n", ha[i], hs[i], ho[i * 3], ho[i * 3 + 1], ho[i * 3 + 2]);
  return 0;
}
