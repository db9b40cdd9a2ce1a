// api.cpp -- library-wide C-ABI helpers: version, last-error string.  (Kernels live in the *.hip files.)
#include <stdarg.h>
#include <stdio.h>

#include "pxr_common.h"
#include "../../include/pxr.h"   // PXR_ABI_VERSION; also type-checks the entries defined in this file against their declarations

static thread_local char g_err[512] = "";

void pxr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Device status word (caller-owned int32 in device memory, one process per GPU): the gather kernels OR bit 0 into it
// when they meet an id outside [0, N) -- the device-side counterpart of the IndexError / device assert that
// nn.Embedding raises in the reference (sasrec.py:68) -- and then clamp the id so the access itself stays in bounds.
static int32_t* g_status_word = nullptr;
int32_t* pxr_status_word(void) { return g_status_word; }
extern "C" int pxr_set_status_word(int32_t* dev_word) {
  g_status_word = dev_word;
  return PXR_OK;
}

// CUs of the current device (cached per device; CU masks and partition modes are reflected in what the runtime reports): kernels
// whose workgroups wait for each other check their grid against it
int pxr_cu_count(void) {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cached[dev] = n;
  }
  return cached[dev];
}

extern "C" int pxr_version(void) { return PXR_ABI_VERSION; }  // include/pxr.h
extern "C" const char* pxr_last_error(void) { return g_err; }
extern "C" const char* pxr_target_arch(void) { return "gfx950"; }

// Host evaluation of the kernels' dropout keep-mask (the SAME inline hash as the device code, pxr_common.h): lets
// the CPU test suite pin oracle/dropout_rng.py against the exact function the kernels use, without a GPU.
extern "C" int pxr_dropout_keep_host(uint64_t seed, uint32_t stream_id, uint64_t first_index, int64_t n, float p,
                                     uint8_t* keep_out) {
  if (!keep_out || n < 0) return PXR_ERR_BAD_ARG;
  const uint32_t thr = pxr_drop_threshold(p);
  for (int64_t i = 0; i < n; ++i) keep_out[i] = pxr_keep(seed, stream_id, first_index + (uint64_t)i, thr) ? 1 : 0;
  return PXR_OK;
}
