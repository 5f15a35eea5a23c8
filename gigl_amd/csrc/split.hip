// split.hip — hash slots of the split generator's assigners, in bulk on the device.
//
// Replaces the per-object hashing of HashingAssigner.assign (scala/split_generator/src/main/scala/lib/assigners/
// AbstractAssigners.scala:30-111): slot = floorMod(MurmurHash3.bytesHash(coder(obj)), 10000) with
//   coder(node) = "<nodeId>-<condensedNodeType>"                 (NodeToDatasetSplitHashingAssigner.scala,
//                                                                 GraphPbWrappers.scala:37-39)
//   coder(edge) = "<src>-<condensedEdgeType>-<dst>", the endpoints ordered (min, max) first when the assigner splits
//                 edges symmetrically (TransductiveEdgeToLinkSplitHashingAssigner.scala:66-78)
// scala.util.hashing.MurmurHash3.bytesHash is MurmurHash3_x86_32 with seed 0x3c074a61 ("arraySeed").  One thread per
// object: the decimal key (<= 32 bytes) is formatted into registers / scratch and hashed there; 4 B in, 4 B out per
// object — the kernel exists to take a ~2 us-per-key Python loop off the host, not because it is heavy.
// The bucket a slot falls in (cumulative float32 weights) stays with the caller.
#include "common.h"

namespace {

constexpr uint32_t SCALA_ARRAY_SEED = 0x3C074A61u;
constexpr int32_t HASH_SPACE = 10000;

__device__ __forceinline__ int put_decimal(uint32_t v, uint8_t* out) {  // -> number of digits written
  uint8_t tmp[10];
  int n = 0;
  do {
    tmp[n++] = (uint8_t)('0' + v % 10u);
    v /= 10u;
  } while (v);
  for (int i = 0; i < n; ++i) out[i] = tmp[n - 1 - i];
  return n;
}

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ int32_t murmur3_x86_32(const uint8_t* data, int n, uint32_t seed) {
  const uint32_t c1 = 0xCC9E2D51u, c2 = 0x1B873593u;
  uint32_t h = seed;
  const int nb = n / 4;
  for (int i = 0; i < nb; ++i) {
    uint32_t k = (uint32_t)data[4 * i] | ((uint32_t)data[4 * i + 1] << 8) | ((uint32_t)data[4 * i + 2] << 16) |
                 ((uint32_t)data[4 * i + 3] << 24);
    k *= c1;
    k = rotl32(k, 15);
    k *= c2;
    h ^= k;
    h = rotl32(h, 13);
    h = h * 5u + 0xE6546B64u;
  }
  uint32_t k = 0;
  const int t = n & 3, base = nb * 4;
  if (t == 3) k ^= (uint32_t)data[base + 2] << 16;
  if (t >= 2) k ^= (uint32_t)data[base + 1] << 8;
  if (t >= 1) {
    k ^= (uint32_t)data[base];
    k *= c1;
    k = rotl32(k, 15);
    k *= c2;
    h ^= k;
  }
  h ^= (uint32_t)n;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return (int32_t)h;
}

__global__ __launch_bounds__(256) void split_slots_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                          int64_t n, uint32_t type, int symmetric,
                                                          int32_t* __restrict__ slots) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t key[36];
  int len = 0;
  uint32_t x = a[i];
  if (b) {
    uint32_t y = b[i];
    if (symmetric && x > y) {
      const uint32_t t = x;
      x = y;
      y = t;
    }
    len += put_decimal(x, key + len);
    key[len++] = '-';
    len += put_decimal(type, key + len);
    key[len++] = '-';
    len += put_decimal(y, key + len);
  } else {
    len += put_decimal(x, key + len);
    key[len++] = '-';
    len += put_decimal(type, key + len);
  }
  const int32_t h = murmur3_x86_32(key, len, SCALA_ARRAY_SEED);
  int32_t m = h % HASH_SPACE;  // Math.floorMod: result takes the sign of the (positive) modulus
  if (m < 0) m += HASH_SPACE;
  slots[i] = m;
}

}  // namespace

extern "C" int32_t gigl_split_hash_slots(gigl_ctx* ctx, const uint32_t* a, const uint32_t* b, int64_t n,
                                         int32_t condensed_type, int32_t symmetric, int32_t* slots) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, n >= 0 && (n == 0 || (a && slots)) && condensed_type >= 0, "bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n == 0) return GIGL_OK;
  hipLaunchKernelGGL(split_slots_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, b, n,
                     (uint32_t)condensed_type, symmetric, slots);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}
