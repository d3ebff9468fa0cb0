// Non-kernel entry points of the C ABI.
#include "flashmd_hip.h"

extern "C" const char* fmd_strerror(int code) {
  switch (code) {
    case FMD_OK: return "ok";
    case FMD_EINVAL: return "invalid argument";
    case FMD_ELAUNCH: return "HIP launch/runtime error";
    case FMD_ECAPACITY: return "fixed-capacity buffer too small";
    case FMD_EUNSUPPORTED: return "unsupported configuration";
    case FMD_ENONFINITE: return "non-finite or overflowing force contribution";
    default: return "unknown error";
  }
}

extern "C" int fmd_abi_version(void) { return 6; }
