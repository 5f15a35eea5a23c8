// Measurement only (scripts/micro_rowcopy.py): what the hardware does with the record encoder's payload pattern in
// isolation — random 4*d-byte rows of a feature table copied to byte-misaligned, nearly contiguous destinations
// (one row every 4*d + gap bytes), as dense (row, 16-byte chunk) items.  variant 0: unaligned 16-byte stores;
// variant 1: aligned dword-funnel stores (what the encoder did before); variant 2: loads only; variant 3: stores only.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_u;
template <int UNR>
__global__ __launch_bounds__(256) void rowcopy(const uint32_t* feat, int d, const uint32_t* ids, const int64_t* dst_off,
                                               int64_t n_rows, uint8_t* out, int variant) {
  const uint32_t nch = (uint32_t)d / 4;
  const int64_t total = n_rows * nch;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += stride * UNR) {
    uint4 v[UNR];
    uint8_t* dst[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      const int64_t i = i0 + j * stride;
      dst[j] = nullptr;
      v[j] = uint4{0, 0, 0, 0};
      if (i < total) {
        const int64_t row = i / nch;
        const uint32_t c = (uint32_t)(i - row * nch);
        dst[j] = out + dst_off[row] + 16 * c;
        if (variant != 3) v[j] = *(const uint4*)(feat + (int64_t)ids[row] * d + 4 * c);
      }
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      if (!dst[j]) continue;
      if (variant == 2) {
        if ((v[j].x ^ v[j].y ^ v[j].z ^ v[j].w) == 0x12345679u) *dst[j] = 1;
      } else {
        *(u32x4_u*)dst[j] = u32x4_u{v[j].x, v[j].y, v[j].z, v[j].w};
      }
    }
  }
}
extern "C" int rowcopy_launch(const void* feat, int d, const void* ids, const void* dst_off, long long n_rows, void* out,
                              int variant, int blocks, int unr, void* stream) {
  if (unr == 8)
    hipLaunchKernelGGL(rowcopy<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)feat, d,
                       (const uint32_t*)ids, (const int64_t*)dst_off, (int64_t)n_rows, (uint8_t*)out, variant);
  else
    hipLaunchKernelGGL(rowcopy<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)feat, d,
                       (const uint32_t*)ids, (const int64_t*)dst_off, (int64_t)n_rows, (uint8_t*)out, variant);
  return (int)hipGetLastError();
}
