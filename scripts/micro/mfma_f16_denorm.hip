// does v_mfma_f32_32x32x16_f16 on gfx950 honour subnormal fp16 inputs?  (scripts/micro: experiment, not product code)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)a_val;
    b[i] = (_Float16)b_val;
  }
  float16v acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  float* d;
  hipMalloc(&d, 4);
  const float cases[][2] = {{1.f, 1.f}, {9.5367431640625e-07f /*2^-20*/, 1.f}, {1.f, 9.5367431640625e-07f},
                            {5.9604644775390625e-08f /*2^-24*/, 1.f}, {3.0517578125e-05f /*2^-15*/, 1.f}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
    float h;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a=%g b=%g  acc=%g  expected=%g\n", c[0], c[1], h, 16.0 * c[0] * c[1]);
  }
  return 0;
}
