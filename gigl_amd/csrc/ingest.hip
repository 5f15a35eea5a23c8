// ingest.hip — host-side native reader of the preprocessed node / edge tables: TFRecord framing + tf.Example
// decoding into dense columns.  (No device code: the records live in host files; what follows them — CSC build,
// bidirectionalisation, feature upload — runs on the device, graph_build.hip.)
//
// Replaces (paths relative to the reference root):
//   spark-tfrecord `format("tfrecord").option("recordType", "Example")` reads +
//   loadNodeDataframeIntoSparkSql / loadEdgeDataframeIntoSparkSql column selection and casts
//       scala/subgraph_sampler/src/main/scala/libs/task/pureSpark/SGSPureSparkV1Task.scala:52-118 (features = the
//       featureKeys columns concatenated in order :90-104), :120-216 (ids cast to int32 :164-168)
//   scala/common/src/main/scala/utils/TFRecordIO.scala:20-51 (readDataframeFromTfrecord)
// Formats: TFRecord = u64 length | masked crc32c(length) | payload | masked crc32c(payload);
//   tf.Example{features=1: Features{feature=1: map<string, Feature{bytes_list=1 | float_list=2 | int64_list=3}>}},
//   each list {value=1, packed or one element per tag}.
#include "common.h"

#include <atomic>
#include <cstring>
#include <thread>

namespace {

uint32_t g_crc_tbl[8][256];
std::atomic<int> g_crc_ready{0};

void crc_init() {
  if (g_crc_ready.load(std::memory_order_acquire)) return;
  static std::atomic<int> building{0};
  int expect = 0;
  if (!building.compare_exchange_strong(expect, 1)) {
    while (!g_crc_ready.load(std::memory_order_acquire)) std::this_thread::yield();
    return;
  }
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_crc_tbl[0][i] = c;
  }
  for (int t = 1; t < 8; ++t)
    for (uint32_t i = 0; i < 256; ++i) g_crc_tbl[t][i] = (g_crc_tbl[t - 1][i] >> 8) ^ g_crc_tbl[0][g_crc_tbl[t - 1][i] & 0xFF];
  g_crc_ready.store(1, std::memory_order_release);
}

uint32_t crc32c(const uint8_t* p, size_t n) {  // slicing-by-8
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = g_crc_tbl[7][w & 0xFF] ^ g_crc_tbl[6][(w >> 8) & 0xFF] ^ g_crc_tbl[5][(w >> 16) & 0xFF] ^
        g_crc_tbl[4][(w >> 24) & 0xFF] ^ g_crc_tbl[3][(w >> 32) & 0xFF] ^ g_crc_tbl[2][(w >> 40) & 0xFF] ^
        g_crc_tbl[1][(w >> 48) & 0xFF] ^ g_crc_tbl[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = g_crc_tbl[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
uint32_t masked(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }

struct Span {
  const uint8_t* p;
  const uint8_t* e;
};
bool varint(Span& s, uint64_t& v) {
  v = 0;
  for (int shift = 0; shift < 70 && s.p < s.e; shift += 7) {
    const uint8_t b = *s.p++;
    v |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) return true;
  }
  return false;
}
// next field of a message: fno, wire type; for len-delimited `sub` = its bytes, for varint `val`
bool field(Span& s, uint32_t& fno, uint32_t& wt, Span& sub, uint64_t& val) {
  uint64_t key;
  if (!varint(s, key)) return false;
  fno = (uint32_t)(key >> 3);
  wt = (uint32_t)(key & 7);
  if (wt == 0) return varint(s, val);
  if (wt == 2) {
    uint64_t ln;
    if (!varint(s, ln) || ln > (uint64_t)(s.e - s.p)) return false;
    sub = Span{s.p, s.p + ln};
    s.p += ln;
    return true;
  }
  const size_t fix = wt == 1 ? 8 : wt == 5 ? 4 : 0;
  if (!fix || (size_t)(s.e - s.p) < fix) return false;
  sub = Span{s.p, s.p + fix};
  s.p += fix;
  return true;
}

// values of one Feature into column c, row i.  Returns number of values seen, or -1 on malformed / wrong kind.
int64_t put_feature(Span feat, const gigl_column& c, int64_t i) {
  int64_t n = 0;
  uint32_t kind, wt;
  Span lst{nullptr, nullptr};
  uint64_t v;
  while (feat.p < feat.e) {
    if (!field(feat, kind, wt, lst, v)) return -1;
    if (wt != 2) continue;
    if (kind == 2) {  // FloatList
      if (c.kind != GIGL_COL_F32) return -1;
      float* out = (float*)c.out + i * c.width;
      Span l = lst, x{nullptr, nullptr};
      uint32_t f4, w4;
      while (l.p < l.e) {
        if (!field(l, f4, w4, x, v)) return -1;
        if (f4 != 1) continue;
        if (w4 == 2) {  // packed
          const int64_t m = (x.e - x.p) / 4;
          for (int64_t k = 0; k < m; ++k, ++n)
            if (n < c.width) memcpy(out + n, x.p + 4 * k, 4);
        } else if (w4 == 5) {
          if (n < c.width) memcpy(out + n, x.p, 4);
          ++n;
        } else {
          return -1;
        }
      }
    } else if (kind == 3) {  // Int64List
      Span l = lst, x{nullptr, nullptr};
      uint32_t f4, w4;
      auto put = [&](uint64_t u) {
        if (n < c.width) {
          if (c.kind == GIGL_COL_I64) ((int64_t*)c.out)[i * c.width + n] = (int64_t)u;
          else ((float*)c.out)[i * c.width + n] = (float)(int64_t)u;  // integer feature column cast to float
        }
        ++n;
      };
      while (l.p < l.e) {
        if (!field(l, f4, w4, x, v)) return -1;
        if (f4 != 1) continue;
        if (w4 == 0) {
          put(v);
        } else if (w4 == 2) {
          while (x.p < x.e) {
            uint64_t u;
            if (!varint(x, u)) return -1;
            put(u);
          }
        } else {
          return -1;
        }
      }
    } else if (kind == 1) {
      return -1;  // bytes features are not numeric columns
    }
  }
  return n;
}

}  // namespace

extern "C" {

int32_t gigl_tfrecord_index(const uint8_t* buf, int64_t len, int32_t verify_crc, int64_t cap, int64_t* payload_off,
                            int64_t* payload_len, int64_t* n_records) {
  if (!buf || len < 0 || !n_records) return GIGL_E_INVALID_ARG;
  if (verify_crc) crc_init();
  int64_t pos = 0, n = 0;
  while (pos < len) {
    if (pos + 12 > len) return GIGL_E_INVALID_ARG;  // truncated header
    uint64_t ln;
    uint32_t hc;
    memcpy(&ln, buf + pos, 8);
    memcpy(&hc, buf + pos + 8, 4);
    if (verify_crc && hc != masked(crc32c(buf + pos, 8))) return GIGL_E_INVALID_ARG;
    if (ln > (uint64_t)(len - pos - 16)) return GIGL_E_INVALID_ARG;  // truncated payload
    if (verify_crc) {
      uint32_t pc;
      memcpy(&pc, buf + pos + 12 + ln, 4);
      if (pc != masked(crc32c(buf + pos + 12, (size_t)ln))) return GIGL_E_INVALID_ARG;
    }
    if (payload_off && n < cap) {
      payload_off[n] = pos + 12;
      if (payload_len) payload_len[n] = (int64_t)ln;
    }
    ++n;
    pos += 16 + (int64_t)ln;
  }
  *n_records = n;
  return GIGL_OK;
}

int32_t gigl_tfexample_decode(const uint8_t* buf, const int64_t* payload_off, const int64_t* payload_len, int64_t n,
                              const gigl_column* cols, int32_t n_cols, int32_t n_threads, int64_t* bad_record) {
  if (!buf || !payload_off || !payload_len || n < 0 || !cols || n_cols < 1) return GIGL_E_INVALID_ARG;
  for (int32_t c = 0; c < n_cols; ++c)
    if (!cols[c].name || !cols[c].out || cols[c].width < 1 ||
        (cols[c].kind != GIGL_COL_I64 && cols[c].kind != GIGL_COL_F32))
      return GIGL_E_INVALID_ARG;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  std::atomic<int64_t> bad{-1};
  auto work = [&](int64_t lo, int64_t hi) {
    std::vector<size_t> name_len(n_cols);
    for (int32_t c = 0; c < n_cols; ++c) name_len[c] = strlen(cols[c].name);
    for (int64_t i = lo; i < hi && bad.load(std::memory_order_relaxed) < 0; ++i) {
      for (int32_t c = 0; c < n_cols; ++c) {
        if (cols[c].kind == GIGL_COL_I64) memset((int64_t*)cols[c].out + i * cols[c].width, 0, 8 * (size_t)cols[c].width);
        else memset((float*)cols[c].out + i * cols[c].width, 0, 4 * (size_t)cols[c].width);
        if (cols[c].counts) cols[c].counts[i] = 0;
      }
      Span ex{buf + payload_off[i], buf + payload_off[i] + payload_len[i]};
      bool ok = true;
      uint32_t fno, wt;
      Span feats{nullptr, nullptr}, entry{nullptr, nullptr}, x{nullptr, nullptr};
      uint64_t v;
      while (ok && ex.p < ex.e) {
        if (!field(ex, fno, wt, feats, v)) { ok = false; break; }
        if (fno != 1 || wt != 2) continue;
        while (ok && feats.p < feats.e) {
          if (!field(feats, fno, wt, entry, v)) { ok = false; break; }
          if (fno != 1 || wt != 2) continue;
          Span key{nullptr, nullptr}, feat{nullptr, nullptr};
          Span en = entry;
          while (en.p < en.e) {
            if (!field(en, fno, wt, x, v)) { ok = false; break; }
            if (wt != 2) continue;
            if (fno == 1) key = x;
            else if (fno == 2) feat = x;
          }
          if (!ok || !key.p) continue;
          for (int32_t c = 0; c < n_cols; ++c) {
            if ((size_t)(key.e - key.p) != name_len[c] || memcmp(key.p, cols[c].name, name_len[c]) != 0) continue;
            const int64_t got = feat.p ? put_feature(feat, cols[c], i) : 0;
            if (got < 0) ok = false;
            else if (cols[c].counts) cols[c].counts[i] = (int32_t)got;
          }
        }
      }
      if (!ok) {
        int64_t expect = -1;
        bad.compare_exchange_strong(expect, i);
      }
    }
  };
  if (n_threads == 1 || n < 4096) {
    work(0, n);
  } else {
    std::vector<std::thread> th;
    const int64_t per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
      const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
      if (lo < hi) th.emplace_back(work, lo, hi);
    }
    for (auto& t : th) t.join();
  }
  if (bad_record) *bad_record = bad.load();
  return bad.load() < 0 ? GIGL_OK : GIGL_E_INVALID_ARG;
}

}  // extern "C"
