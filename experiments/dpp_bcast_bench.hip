// Micro-benchmark: cost of v_fmac_f32_dpp row_newbcast (operand broadcast from a lane of the 16-lane row, no LDS)
// against v_pk_fma_f32 with the operand in registers, at 1 and 2 waves per SIMD.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 experiments/dpp_bcast_bench.hip -o /tmp/dppb && /tmp/dppb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define FM(n) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(v[n]) : "v"(pb0), "v"(nt));
#define FM2(n) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(v[16 + n]) : "v"(pb1), "v"(nt));

__global__ void __launch_bounds__(256) k_dpp(float* out, const float* in, int reps, long long* cyc) {
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = in[threadIdx.x + i];
  float pb0 = in[threadIdx.x + 40], pb1 = in[threadIdx.x + 41], nt = in[threadIdx.x + 42];
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    FM(0) FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(8) FM(9) FM(10) FM(11) FM(12) FM(13) FM(14) FM(15)
    FM2(0) FM2(1) FM2(2) FM2(3) FM2(4) FM2(5) FM2(6) FM2(7) FM2(8) FM2(9) FM2(10) FM2(11) FM2(12) FM2(13)
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_pk(float* out, const float* in, int reps, long long* cyc) {
  f2 v[16], pb[15];
  for (int i = 0; i < 16; ++i) v[i] = f2{in[threadIdx.x + 2 * i], in[threadIdx.x + 2 * i + 1]};
  for (int i = 0; i < 15; ++i) pb[i] = f2{in[threadIdx.x + 40 + i], in[threadIdx.x + 60 + i]};
  const float nt = in[threadIdx.x + 42];
  const f2 t2 = {nt, nt};
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 15; ++i) v[i] = __builtin_elementwise_fma(t2, pb[i], v[i]);
    asm volatile("" ::: "memory");
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float *in, *out; long long* cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8192 * 8);
  hipMemset(in, 0, 4096 * 4);
  const int reps = 2000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int threads = 256 * wps;            // 4 or 8 waves per workgroup, one workgroup per CU
    for (int which = 0; which < 2; ++which) {
      long long h[256];
      for (int it = 0; it < 2; ++it) {
        if (which == 0) hipLaunchKernelGGL(k_dpp, dim3(256), dim3(threads), 0, 0, out, in, reps, cyc);
        else hipLaunchKernelGGL(k_pk, dim3(256), dim3(threads), 0, 0, out, in, reps, cyc);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
      printf("%s  %d wave(s)/SIMD: %.1f cycles per group of %s\n", which == 0 ? "30 x v_fmac_f32_dpp row_newbcast" : "15 x v_pk_fma_f32              ",
             wps, m / reps, which == 0 ? "30 FMAs" : "15 packed FMAs");
    }
  }
  return 0;
}
