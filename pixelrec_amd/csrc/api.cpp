// api.cpp -- library-wide C-ABI helpers: version, last-error string.  (Kernels live in the *.hip files.)
#include <stdarg.h>
#include <stdio.h>

#include "pxr_common.h"

static thread_local char g_err[512] = "";

void pxr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int pxr_version(void) { return 100; }  // major*10000 + minor*100 + patch
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
