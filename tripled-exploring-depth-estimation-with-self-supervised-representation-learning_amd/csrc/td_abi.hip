// ABI bookkeeping: version, error strings, per-thread record of the last HIP launch error.
#include <string.h>

#include "td_common.h"

namespace td {

static thread_local char g_last_err[256] = "";

int record_launch_error(hipError_t e, const char* what) {
  if (e == hipSuccess) return TD_OK;
  snprintf(g_last_err, sizeof(g_last_err), "%s: %s", what, hipGetErrorString(e));
  return TD_ERR_LAUNCH;
}

}  // namespace td

extern "C" int td_abi_version(void) { return TD_ABI_VERSION; }

extern "C" const char* td_error_string(int code) {
  switch (code) {
    case TD_OK: return "ok";
    case TD_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size, or n_src out of range)";
    case TD_ERR_UNSUPPORTED: return "unsupported shape";
    case TD_ERR_LAUNCH: return "HIP kernel launch failed";
    case TD_ERR_WORKSPACE: return "workspace too small";
  }
  return "unknown error code";
}

extern "C" const char* td_last_hip_error(void) { return td::g_last_err; }
