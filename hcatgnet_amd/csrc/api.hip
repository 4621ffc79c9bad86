// Version / error-string entry points of libhcatgnet_hip.so.
#include "common.h"

extern "C" int hcg_version(void) { return HCG_ABI_VERSION; }

extern "C" const char* hcg_error_string(int code) {
  switch (code) {
    case HCG_OK: return "ok";
    case HCG_ERR_INVALID_ARG: return "invalid argument";
    case HCG_ERR_WORKSPACE: return "workspace too small";
    case HCG_ERR_UNSUPPORTED: return "shape not supported by this entry point";
    default: break;
  }
  if (code <= HCG_ERR_HIP_BASE) return hipGetErrorString((hipError_t)(HCG_ERR_HIP_BASE - code));
  return "unknown error";
}
