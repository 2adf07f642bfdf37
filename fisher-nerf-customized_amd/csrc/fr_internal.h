// fr_internal.h -- shared between the translation units of libfisher_rast.so (not part of the C ABI).
#ifndef FR_INTERNAL_H_INCLUDED
#define FR_INTERNAL_H_INCLUDED
#include "../../include/fisher_rast.h"

// records the message for fr_last_error() (thread-local) and returns `code`
__attribute__((visibility("hidden"))) int fr_fail(int code, const char* msg);
// FR_OK, or FR_ELAUNCH with hipGetErrorString of the pending launch error
__attribute__((visibility("hidden"))) int fr_check_launch(const char* what);
#endif
