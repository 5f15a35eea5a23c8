// Ceiling of the step's dominant access pattern: feature rows of R bytes fetched at random from a table of T bytes (what
// gather_mean / gat_input_online_kernel do per aggregated edge), nothing else — one wave per row, 16-byte loads per lane,
// U rows in flight per wave, a xor checksum so the loads are not dead.  Prints TB/s per (row bytes, table size).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/rowgather.hip -o scripts/micro/bin/rowgather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void gather_rows(const uint4* __restrict__ table, int64_t n_rows, int units_per_row,
                                                   const uint32_t* __restrict__ ids, int64_t n_ids, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  uint32_t acc = 0;
  for (int64_t i = wave * U; i < n_ids; i += waves * U) {
    uint4 v[U][2];
#pragma unroll
    for (int t = 0; t < U; ++t) {
      const uint4* row = table + (int64_t)(i + t < n_ids ? ids[i + t] : 0) * units_per_row;
#pragma unroll
      for (int q = 0; q < 2; ++q) v[t][q] = lane + 64 * q < units_per_row ? row[lane + 64 * q] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < U; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) acc ^= v[t][q].x ^ v[t][q].y ^ v[t][q].z ^ v[t][q].w;
  }
  if (acc == 0x12345678u) out[0] = acc;  // (never: keeps the loads alive)
}

int main() {
  const int64_t n_ids = 1 << 22;
  uint32_t* ids_h = (uint32_t*)malloc(n_ids * 4);
  uint32_t *ids, *out;
  CK(hipMalloc(&ids, n_ids * 4));
  CK(hipMalloc(&out, 64));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int row_bytes[] = {400, 512, 1024, 1536, 2048};
  const double table_gb[] = {1.0, 8.0, 47.0};
  for (double gb : table_gb) {
    const size_t bytes = (size_t)(gb * (1ull << 30));
    uint4* table;
    CK(hipMalloc(&table, bytes));
    CK(hipMemset(table, 1, bytes));
    for (int rb : row_bytes) {
      const int upr = rb / 16;
      const int64_t n_rows = (int64_t)(bytes / ((size_t)upr * 16));
      uint64_t s = 88172645463325252ull;
      for (int64_t i = 0; i < n_ids; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        ids_h[i] = (uint32_t)(s % (uint64_t)n_rows);
      }
      CK(hipMemcpy(ids, ids_h, n_ids * 4, hipMemcpyHostToDevice));
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        gather_rows<8><<<256 * 16, 256>>>(table, n_rows, upr, ids, n_ids, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
      }
      printf("table %5.1f GB  row %4d B  %lld rows fetched in %.3f ms  = %.2f TB/s\n", gb, upr * 16, (long long)n_ids, best,
             (double)n_ids * upr * 16 / (best * 1e-3) / 1e12);
    }
    CK(hipFree(table));
  }
  return 0;
}
