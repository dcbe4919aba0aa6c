// Probe (gfx950): does `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS for lanes whose offset is beyond the buffer's num_records?
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/buf_lds_dma.hip -o gpurun_out/buf_lds_dma && ./gpurun_out/buf_lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint32_t* src, int nbytes, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
  const int lane = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), (short)0, nbytes, 0x00020000);
  const uint32_t voff = lane * 16;
  const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds;
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(voff), "s"(r), "s"(l0) : "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
  uint32_t *src, *out;
  std::vector<uint32_t> h(256), o(256);
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  hipMalloc(&src, 1024); hipMalloc(&out, 1024);
  hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 40 * 16, out);     // lanes 40..63 are out of range
  hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
  hipError_t e = hipDeviceSynchronize();
  printf("err %d\n", (int)e);
  for (int l : {0, 1, 39, 40, 41, 63}) printf("lane %2d: %08x %08x %08x %08x\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  return 0;
}
