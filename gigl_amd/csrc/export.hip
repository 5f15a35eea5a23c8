// export.hip — (node id, embedding row) batches -> Avro object-container data blocks, on the device.
//
// Replaces the per-record Python loop of the reference's inference output writer (paths relative to the
// reference root):
//   AVRO_SCHEMA                      python/gigl/common/data/export.py:34-43
//                                    record Embedding {node_id: long, node_type: string, emb: array<float>}
//   EmbeddingExporter.add_embedding  python/gigl/common/data/export.py:103-135 (int(node_id), embedding.tolist(),
//                                    fastavro.writer appending data blocks to the shard's buffer)
// fastavro (the reference's third-party writer, not vendored) follows the Apache Avro 1.x specification, restated
// here: long = zig-zag LEB128 varint; string = long byte count + UTF-8; float = 4 bytes little-endian IEEE;
// array = long item count + items + long 0 (just the 0 for an empty array); a data block = long record count,
// long byte size of the serialized records, the records, the file's 16-byte sync marker.
//
// Byte work, HBM-bound: a record is 4*D + ~6 + len(type) bytes, every embedding word is read once and written
// once.  One 256-thread workgroup per DATA BLOCK (a fixed number of consecutive records chosen by the host so a
// block is ~16 kB, the writer's sync interval), two passes:
//   size  : per block, sum of the records' sizes (only the id varint varies) -> framed block size
//   (one single-workgroup scan over the block sizes gives every block's byte offset in the output)
//   write : block scan of the record sizes -> offset of each record inside the block (kept in LDS, also
//           returned), framing by one thread, then one wave per record: header bytes by byte stores, the float
//           payload as dword stores after a funnel shift by the destination's byte misalignment (record
//           offsets are arbitrary byte offsets), coalesced loads of the row.
#include "common.h"

#include <charconv>
#include <cstring>
#include <thread>

namespace {

constexpr int MAX_BLOCK_RECORDS = 2048;
constexpr int MAX_TYPE_BYTES = 240;

struct AvroText {
  uint8_t b[16 + MAX_TYPE_BYTES];
};

__global__ void avro_text_kernel(AvroText t, uint8_t* __restrict__ out) {
  if (threadIdx.x < sizeof(t.b)) out[threadIdx.x] = t.b[threadIdx.x];
}

struct AvroArgs {
  const int64_t* ids;
  const float* emb;
  int64_t stride;  // floats between rows
  int64_t n;
  int32_t d;
  int32_t per_block;  // records per data block
  int32_t type_len;
  int32_t fixed;  // bytes of a record besides the id varint
  const uint8_t* text;  // device: 16 sync-marker bytes, then the node type's UTF-8 bytes
};

__device__ __host__ __forceinline__ uint64_t zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
__device__ __host__ __forceinline__ int vlen64(uint64_t v) {
  int n = 1;
  while (v >= 0x80) {
    v >>= 7;
    ++n;
  }
  return n;
}
__device__ __forceinline__ uint8_t* put_varint(uint8_t* p, uint64_t v) {
  while (v >= 0x80) {
    *p++ = (uint8_t)(v | 0x80);
    v >>= 7;
  }
  *p++ = (uint8_t)v;
  return p;
}

__host__ __device__ __forceinline__ int fixed_bytes(int32_t type_len, int32_t d) {
  return vlen64(zigzag(type_len)) + type_len + (d > 0 ? vlen64(zigzag(d)) + 4 * d : 0) + 1;
}

// serialized bytes of the records of block b, by block reduction; every thread returns the total
__device__ __forceinline__ int64_t block_body_bytes(const AvroArgs& a, int64_t first, int cnt, int64_t* s_red) {
  int64_t mine = 0;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) mine += vlen64(zigzag(a.ids[first + i])) + a.fixed;
#pragma unroll
  for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mine;
  __syncthreads();
  int64_t t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void avro_size_kernel(AvroArgs a, int64_t* blk_size) {
  __shared__ int64_t s_red[4];
  const int64_t first = (int64_t)blockIdx.x * a.per_block;
  const int cnt = (int)min((int64_t)a.per_block, a.n - first);
  const int64_t body = block_body_bytes(a, first, cnt, s_red);
  if (threadIdx.x == 0) blk_size[blockIdx.x] = vlen64(zigzag(cnt)) + vlen64(zigzag(body)) + body + 16;
}

__global__ __launch_bounds__(256) void avro_write_kernel(AvroArgs a, const int64_t* blk_off, const int32_t* status,
                                                         uint8_t* out, int64_t* rec_off) {
  __shared__ uint32_t s_off[MAX_BLOCK_RECORDS + 1];
  __shared__ uint32_t s_wsum[4];
  if (*status) return;
  const int64_t first = (int64_t)blockIdx.x * a.per_block;
  const int cnt = (int)min((int64_t)a.per_block, a.n - first);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // offsets of the records inside the block: thread t owns the run [t*per, (t+1)*per)
  const int per = (cnt + 255) >> 8;
  const int lo = min((int)threadIdx.x * per, cnt), hi = min(lo + per, cnt);
  uint32_t run = 0;
  for (int i = lo; i < hi; ++i) run += (uint32_t)(vlen64(zigzag(a.ids[first + i])) + a.fixed);
  uint32_t inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) s_wsum[w] = inc;
  __syncthreads();
  uint32_t pre = inc - run;
  for (int j = 0; j < w; ++j) pre += s_wsum[j];
  for (int i = lo; i < hi; ++i) {
    s_off[i] = pre;
    pre += (uint32_t)(vlen64(zigzag(a.ids[first + i])) + a.fixed);
  }
  if (threadIdx.x == 255) s_off[cnt] = pre;  // the last thread's running sum is the body size (empty runs add 0)
  __syncthreads();
  const uint32_t body = s_off[cnt];
  uint8_t* const blk = out + blk_off[blockIdx.x];
  const int head = vlen64(zigzag(cnt)) + vlen64(zigzag((int64_t)body));
  if (threadIdx.x == 0) {
    uint8_t* p = put_varint(blk, zigzag(cnt));
    put_varint(p, zigzag((int64_t)body));
  }
  if (threadIdx.x < 16) blk[head + body + threadIdx.x] = a.text[threadIdx.x];
  const int64_t body_pos = blk_off[blockIdx.x] + head;
  for (int i = threadIdx.x; i < cnt; i += 256) rec_off[first + i] = body_pos + s_off[i];
  // one wave per record
  const int tl = vlen64(zigzag(a.type_len));
  const int dl = a.d > 0 ? vlen64(zigzag(a.d)) : 0;
  for (int i = w; i < cnt; i += 4) {
    uint8_t* const rec = blk + head + s_off[i];
    const int64_t id = a.ids[first + i];
    const uint64_t zz = zigzag(id);
    const int il = vlen64(zz);
    const int hlen = il + tl + a.type_len + dl;
    for (int h = lane; h < hlen; h += 64) {
      uint8_t v;
      if (h < il) {
        v = (uint8_t)((zz >> (7 * h)) & 0x7F) | (h + 1 < il ? 0x80 : 0);
      } else if (h < il + tl) {
        const int k = h - il;
        v = (uint8_t)((zigzag(a.type_len) >> (7 * k)) & 0x7F) | (k + 1 < tl ? 0x80 : 0);
      } else if (h < il + tl + a.type_len) {
        v = a.text[16 + h - il - tl];
      } else {
        const int k = h - il - tl - a.type_len;
        v = (uint8_t)((zigzag(a.d) >> (7 * k)) & 0x7F) | (k + 1 < dl ? 0x80 : 0);
      }
      rec[h] = v;
    }
    uint8_t* const pay = rec + hlen;
    const int len = 4 * a.d;
    if (lane == 0) pay[len] = 0;  // array terminator
    if (a.d > 0) {
      const uint32_t* src = (const uint32_t*)(a.emb + (first + i) * a.stride);
      const int lead = (int)((4 - ((uintptr_t)pay & 3)) & 3);  // bytes before the first aligned dword
      const int words = (len - lead) >> 2;                      // aligned dwords fully inside the payload
      uint32_t* const dst = (uint32_t*)(pay + lead);
      if (lead == 0) {
        for (int j = lane; j < words; j += 64) dst[j] = src[j];
      } else {
        const int sh = 8 * lead;
        for (int j = lane; j < words; j += 64) dst[j] = __funnelshift_r(src[j], src[j + 1], sh);
        // edges: `lead` leading bytes of word 0, 4 - lead trailing bytes of the last word
        if (lane < lead) pay[lane] = (uint8_t)(src[0] >> (8 * lane));
        const int tail0 = lead + 4 * words;
        if (lane < len - tail0) pay[tail0 + lane] = (uint8_t)(src[a.d - 1] >> (8 * (lane + lead)));
      }
    }
  }
}

}  // namespace

extern "C" {

int32_t gigl_avro_embeddings_layout(int64_t n, int32_t dim, int32_t type_len, int32_t* records_per_block,
                                    int64_t* n_blocks, int64_t* bytes) {
  if (n < 0 || dim < 0 || type_len < 0 || type_len > MAX_TYPE_BYTES || !records_per_block || !n_blocks || !bytes)
    return GIGL_E_INVALID_ARG;
  const int64_t rec_max = 10 + fixed_bytes(type_len, dim);
  int64_t per = (16000 + rec_max - 1) / rec_max;  // the writer's sync interval: 1000 * 16 bytes
  if (per < 1) per = 1;
  if (per > MAX_BLOCK_RECORDS) per = MAX_BLOCK_RECORDS;
  *records_per_block = (int32_t)per;
  *n_blocks = (n + per - 1) / per;
  *bytes = n * rec_max + *n_blocks * (10 + 10 + 16);
  return GIGL_OK;
}

int32_t gigl_avro_embeddings_encode(gigl_ctx* ctx, const int64_t* ids, const float* emb, int64_t emb_stride,
                                    int64_t n, int32_t dim, const uint8_t* type_utf8, int32_t type_len,
                                    const uint8_t* sync_marker, uint8_t* out, int64_t out_cap, int64_t* rec_off,
                                    int64_t* total_bytes, int32_t* status) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, sync_marker && total_bytes && status && (type_utf8 || type_len == 0), "null argument");
  GIGL_REQUIRE(ctx, n >= 0 && dim >= 0 && out_cap >= 0 && emb_stride >= dim, "bad sizes");
  GIGL_REQUIRE(ctx, type_len >= 0 && type_len <= MAX_TYPE_BYTES, "node type of %d bytes (at most %d)", type_len,
               MAX_TYPE_BYTES);
  GIGL_REQUIRE(ctx, n == 0 || (ids && rec_off && out && (emb || dim == 0)), "null argument");
  int32_t per;
  int64_t n_blocks, bound;
  int32_t rc = gigl_avro_embeddings_layout(n, dim, type_len, &per, &n_blocks, &bound);
  if (rc != GIGL_OK) return gigl_fail(ctx, rc, "bad layout arguments");
  GIGL_REQUIRE(ctx, n_blocks < ((int64_t)1 << 31), "too many data blocks");
  AvroArgs a{};
  a.ids = ids;
  a.emb = emb;
  a.stride = emb_stride;
  a.n = n;
  a.d = dim;
  a.per_block = per;
  a.type_len = type_len;
  a.fixed = fixed_bytes(type_len, dim);
  uint8_t text[16 + MAX_TYPE_BYTES];
  for (int i = 0; i < 16; ++i) text[i] = sync_marker[i];
  for (int i = 0; i < type_len; ++i) text[16 + i] = type_utf8[i];
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  rc = gigl_arena_reset(ctx, (n_blocks + 1) * 16 + 1024);
  if (rc != GIGL_OK) return rc;
  uint8_t* text_dev = (uint8_t*)gigl_arena_alloc(ctx, sizeof(text));
  if (!text_dev) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  // the marker / type bytes travel as a kernel argument (copied at launch): no upload to wait for, the call never
  // synchronises with the host
  AvroText tx;
  for (size_t i = 0; i < sizeof(text); ++i) tx.b[i] = i < (size_t)(16 + type_len) ? text[i] : 0;
  hipLaunchKernelGGL(avro_text_kernel, dim3(1), dim3(256), 0, ctx->stream, tx, text_dev);
  a.text = text_dev;
  int64_t* blk_size = (int64_t*)gigl_arena_alloc(ctx, (n_blocks + 1) * 8);
  int64_t* blk_off = (int64_t*)gigl_arena_alloc(ctx, (n_blocks + 1) * 8);
  if (!blk_size || !blk_off) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  if (n_blocks > 0) hipLaunchKernelGGL(avro_size_kernel, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, a, blk_size);
  gigl_scan_i64(ctx, blk_size, n_blocks, out_cap, blk_off, status);
  if (n_blocks > 0)
    hipLaunchKernelGGL(avro_write_kernel, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, a, blk_off, status, out,
                       rec_off);
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(total_bytes, blk_off + n_blocks, 8, hipMemcpyDeviceToDevice, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}


// ---- line-per-root JSON rows (the local stand-in for the reference's BigQuery rows: one {"node_id", "emb"} /
// {"node_id", "pred"} object per root, python/gigl/src/inference/v1/lib/base_inference_blueprint.py:76-103).  Host
// code: the rows are already on the host when they are written to a file; replaces a per-root Python loop with one
// call that formats row ranges on worker threads.  Floats are printed with the shortest decimal that parses back to
// the same fp32 (std::to_chars), non-finite values as Python's json module prints them (NaN / Infinity / -Infinity).
int64_t gigl_json_rows_capacity(int64_t n, int32_t dim) {
  if (n < 0 || dim < 0) return -1;
  return n * (48 + (int64_t)dim * 18) + 16;
}

int32_t gigl_json_rows_format(const int64_t* ids, const float* emb, int64_t emb_stride, const int32_t* pred, int64_t n,
                              int32_t dim, char* out, int64_t out_cap, int64_t* bytes) {
  if (!ids || !out || !bytes || n < 0 || dim < 0 || (!emb && !pred) || (emb && emb_stride < dim))
    return GIGL_E_INVALID_ARG;
  if (out_cap < gigl_json_rows_capacity(n, dim)) return GIGL_E_INVALID_ARG;
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)(hw ? (hw > 32 ? 32 : hw) : 4);
  if (n < 4096) nt = 1;
  std::vector<std::string> part((size_t)nt);
  auto work = [&](int t) {
    const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
    std::string& o = part[(size_t)t];
    o.reserve((size_t)((hi - lo) * (emb ? 32 + (int64_t)dim * 12 : 40)));
    char buf[48];
    for (int64_t i = lo; i < hi; ++i) {
      o += "{\"node_id\": ";
      auto r = std::to_chars(buf, buf + sizeof(buf), (long long)ids[i]);
      o.append(buf, (size_t)(r.ptr - buf));
      if (emb) {
        o += ", \"emb\": [";
        const float* row = emb + i * emb_stride;
        for (int c = 0; c < dim; ++c) {
          if (c) o += ", ";
          const float v = row[c];
          if (v != v) o += "NaN";
          else if (v > 3.402823466e38f) o += "Infinity";
          else if (v < -3.402823466e38f) o += "-Infinity";
          else {
            r = std::to_chars(buf, buf + sizeof(buf), v);
            o.append(buf, (size_t)(r.ptr - buf));
          }
        }
        o += "]";
      }
      if (pred) {
        o += ", \"pred\": ";
        r = std::to_chars(buf, buf + sizeof(buf), (int)pred[i]);
        o.append(buf, (size_t)(r.ptr - buf));
      }
      o += "}\n";
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  int64_t at = 0;
  for (auto& o : part) {
    if (at + (int64_t)o.size() > out_cap) return GIGL_E_INVALID_ARG;
    memcpy(out + at, o.data(), o.size());
    at += (int64_t)o.size();
  }
  *bytes = at;
  return GIGL_OK;
}

}  // extern "C"
