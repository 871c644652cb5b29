// saf_host.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/saf.h"

namespace saf {

// Thread-local message behind saf_last_error(); defined in saf_fuse.hip.
char* err_buf();
constexpr size_t kErrLen = 512;

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), kErrLen, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SAF_E_HIP, "%s: %s", what, hipGetErrorString(e));
  return SAF_OK;
}

inline int64_t n_voxels(const saf_volume* v) { return (int64_t)v->nx * v->ny * v->nz; }

inline int device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
  return cus;
}

}  // namespace saf
