// Transposed reduction of 16 per-lane values over the 64 lanes of a wave: instead of 16 full butterflies (6 steps each),
// every step halves the number of live values while it halves the lane group a value is spread over — the upper /
// lower 32 lanes trade halves (v_permlane32_swap, gfx950), then odd / even rows of 16 (v_permlane16_swap), then lanes 8
// and 4 apart (DPP), and the last value is finished inside its quad.  35 VALU instructions for 16 totals; lane l ends
// with the total of value gigl_wave_reduce16_index(l) (the four lanes of a quad hold the same one).  The summation
// order is fixed: the same inputs give the same bits on every wave.
#pragma once
#include <hip/hip_runtime.h>

__host__ __device__ __forceinline__ int gigl_wave_reduce16_index(int lane) {
  return ((lane >> 2) & 1) | (((lane >> 3) & 1) << 1) | (((lane >> 4) & 1) << 2) | (((lane >> 5) & 1) << 3);
}

__device__ __forceinline__ float gigl_wave_reduce16(const float (&p)[16]) {
  typedef unsigned int gigl_u2 __attribute__((ext_vector_type(2)));
  auto f2u = [](float x) { return __builtin_bit_cast(unsigned, x); };
  auto u2f = [](unsigned x) { return __builtin_bit_cast(float, x); };
  const int lane = (int)(threadIdx.x & 63);
  float v8[8], v4[4], v2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // lanes < 32 keep value k, lanes >= 32 value k + 8
    const gigl_u2 r = __builtin_amdgcn_permlane32_swap(f2u(p[k]), f2u(p[k + 8]), false, false);
    v8[k] = u2f(r.x) + u2f(r.y);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // even rows of 16 keep value k (+8), odd rows value k + 4 (+8)
    const gigl_u2 r = __builtin_amdgcn_permlane16_swap(f2u(v8[k]), f2u(v8[k + 4]), false, false);
    v4[k] = u2f(r.x) + u2f(r.y);
  }
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {  // lanes 8 apart (row_ror:8): bit 3 clear keeps value k, set keeps k + 2
    const float keep = b3 ? v4[k + 2] : v4[k], give = b3 ? v4[k] : v4[k + 2];
    v2[k] = keep + u2f((unsigned)__builtin_amdgcn_update_dpp(0, (int)f2u(give), 0x128, 0xF, 0xF, true));
  }
  float v1;
  {  // lanes l and 7 - l of an 8-lane group (row_half_mirror: bit 2 differs): bit 2 clear keeps value 0, set keeps 1
    const float keep = b2 ? v2[1] : v2[0], give = b2 ? v2[0] : v2[1];
    v1 = keep + u2f((unsigned)__builtin_amdgcn_update_dpp(0, (int)f2u(give), 0x141, 0xF, 0xF, true));
  }
  v1 += u2f((unsigned)__builtin_amdgcn_update_dpp(0, (int)f2u(v1), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
  v1 += u2f((unsigned)__builtin_amdgcn_update_dpp(0, (int)f2u(v1), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
  return v1;
}
