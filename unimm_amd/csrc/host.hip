// Host half of the input path (no device code): what has to happen to the caller's CPU tensors on the HOST side of the
// PCIe copy.  The reference's callers hand CPU tensors to forward() (train.py:113-129 -- the .to(device) lines are
// commented out; the DataParallel wrapper scatters them, utils/data_parallel.py:123-124), among them int64 [B, 256, 256]
// attention masks: 512 KiB per sequence, 126 MB per 240 sequences, of which the kernels need one BIT per element.
#include <stdint.h>
#include <string.h>
#include <thread>
#include <vector>

#include "../../include/unimm_hip.h"

namespace {

template <typename T>
void pack_rows(const T* m, uint32_t* out, int64_t r0, int64_t r1, int t) {
  const int nw = (t + 31) / 32;
  for (int64_t r = r0; r < r1; ++r) {
    const T* row = m + (size_t)r * t;
    uint32_t* o = out + (size_t)r * nw;
    int c = 0;
    for (int w = 0; w < nw; ++w) {
      const int n = t - c < 32 ? t - c : 32;
      uint32_t bits = 0;
      if (n == 32) {
#pragma unroll
        for (int j = 0; j < 32; ++j) bits |= (uint32_t)(row[c + j] != (T)0) << j;
      } else {
        for (int j = 0; j < n; ++j) bits |= (uint32_t)(row[c + j] != (T)0) << j;
      }
      o[w] = bits;
      c += 32;
    }
  }
}

template <typename T>
int pack_threads(const void* mask, uint32_t* out, int64_t rows, int t, int threads) {
  const T* m = (const T*)mask;
  int hw = (int)std::thread::hardware_concurrency();
  if (hw <= 0) hw = 1;
  int nt = threads > 0 ? threads : (hw > 16 ? 16 : hw);
  const int64_t min_rows = (int64_t)(1 << 20) / (t > 0 ? t : 1) + 1;      // >= ~1 M elements per thread
  if ((int64_t)nt > rows / min_rows) nt = (int)(rows / min_rows);
  if (nt <= 1) { pack_rows<T>(m, out, 0, rows, t); return UNIMM_OK; }
  std::vector<std::thread> th;
  const int64_t per = (rows + nt - 1) / nt;
  for (int i = 0; i < nt; ++i) {
    const int64_t r0 = i * per, r1 = r0 + per < rows ? r0 + per : rows;
    if (r0 >= r1) break;
    th.emplace_back(pack_rows<T>, m, out, r0, r1, t);
  }
  for (auto& x : th) x.join();
  return UNIMM_OK;
}

}  // namespace

extern "C" int unimm_host_mask_pack(const void* mask, int dtype, uint32_t* out, int64_t rows, int32_t t, int32_t threads) {
  if (!mask || !out || rows <= 0 || t <= 0) return UNIMM_E_ARG;
  switch (dtype) {
    case UNIMM_DT_U8: return pack_threads<uint8_t>(mask, out, rows, t, threads);
    case UNIMM_DT_I32: return pack_threads<int32_t>(mask, out, rows, t, threads);
    case UNIMM_DT_I64: return pack_threads<int64_t>(mask, out, rows, t, threads);
    case UNIMM_DT_F32: return pack_threads<float>(mask, out, rows, t, threads);
    default: return UNIMM_E_ARG;
  }
}

// Parallel memcpy between host buffers (pageable tensor -> pinned staging buffer): one thread copies at ~5-12 GB/s, which for
// the 130 MB of region features and targets of a 240-sequence batch is most of a training step; `threads` (0 = up to 8) chunks
// run side by side.
extern "C" int unimm_host_memcpy(void* dst, const void* src, int64_t bytes, int32_t threads) {
  if (!dst || !src || bytes < 0) return UNIMM_E_ARG;
  int hw = (int)std::thread::hardware_concurrency();
  if (hw <= 0) hw = 1;
  int nt = threads > 0 ? threads : (hw > 8 ? 8 : hw);
  const int64_t min_chunk = 4 << 20;
  if ((int64_t)nt > bytes / min_chunk) nt = (int)(bytes / min_chunk);
  if (nt <= 1) { memcpy(dst, src, (size_t)bytes); return UNIMM_OK; }
  std::vector<std::thread> th;
  const int64_t per = ((bytes + nt - 1) / nt + 4095) & ~(int64_t)4095;
  for (int i = 0; i < nt; ++i) {
    const int64_t o = i * per, n = o + per < bytes ? per : bytes - o;
    if (n <= 0) break;
    th.emplace_back([=]() { memcpy((char*)dst + o, (const char*)src + o, (size_t)n); });
  }
  for (auto& x : th) x.join();
  return UNIMM_OK;
}
