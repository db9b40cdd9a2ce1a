// api.cpp -- library-wide C-ABI helpers: version, last-error string.  (Kernels live in the *.hip files.)
#include <stdarg.h>
#include <stdio.h>

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
