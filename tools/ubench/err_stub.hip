// set_error / grit_last_error for the private GEMM libraries of the A/B harness (the product library defines them in elementwise.hip)
#include <stdarg.h>
#include "common.h"
namespace grit { static char buf[512]; void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); } }
extern "C" const char* grit_last_error(void) { return grit::buf; }
