// Error reporting and build identification of libfk.
#include <stdarg.h>
#include <stdio.h>

#include "fk_common.h"

static thread_local char g_err[512] = "";

void fk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fk_last_error(void) { return g_err; }

extern "C" const char* fk_version(void) { return "fk 0.1 gfx950"; }
