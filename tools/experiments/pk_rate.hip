// VALU issue rate on gfx950: v_fma_f32 against v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32, wave64, with 1, 2 and 4 waves
// per SIMD.  Prints wave-instructions per cycle and SIMD, and the flops per cycle and CU that follow.
// hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int kIter = 4096;
template <int MODE>
__global__ void __launch_bounds__(1024) rate(float *out, float x, float y) {
  // eight independent chains per thread
  v2 a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = v2{x + k, y - k};
  const v2 b = v2{y, x}, c = v2{x * 0.5f, y * 0.25f};
  for (int i = 0; i < kIter; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (MODE == 0) { a[k].x = __builtin_fmaf(a[k].x, b.x, c.x); }                  // v_fma_f32
      if constexpr (MODE == 1) { a[k] = __builtin_elementwise_fma(a[k], b, c); }                // v_pk_fma_f32
      if constexpr (MODE == 2) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b)); }
      if constexpr (MODE == 3) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b)); }
      if constexpr (MODE == 4) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k].x) : "v"(b.x)); }
    }
  }
  float r = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) r += a[k].x + a[k].y;
  if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int MODE>
void run(const char *name, int waves_per_simd, float *d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int threads = 64 * 4 * waves_per_simd;     // one workgroup per CU
  hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0001f, 0.9999f);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0001f, 0.9999f);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)kIter * 8 * waves_per_simd;
  printf("%-14s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction and SIMD (%.2f cycles at 2.1 GHz)\n", name,
         waves_per_simd, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.1);
}
int main() {
  float *d; (void)hipMalloc(&d, 4096);
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", w, d); run<4>("v_add_f32", w, d); run<1>("v_pk_fma_f32", w, d); run<2>("v_pk_add_f32", w, d); run<3>("v_pk_mul_f32", w, d);
  }
  return 0;
}
