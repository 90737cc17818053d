// Error reporting and device queries for librlpyt_hip.so (see include/rlpyt_hip.h).
#include "common.h"
#include <stdarg.h>
#include <string.h>

namespace rlpyt {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace rlpyt

extern "C" const char* rlpyt_hip_last_error(void) { return rlpyt::g_err; }

extern "C" int rlpyt_hip_abi_version(void) { return RLPYT_ABI_VERSION; }

extern "C" int rlpyt_hip_device_info(char* name, int cap) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    rlpyt::set_error("rlpyt_hip_device_info: %s", hipGetErrorString(e));
    return RLPYT_EHIP;
  }
  if (name && cap > 0) {
    strncpy(name, prop.gcnArchName, cap - 1);
    name[cap - 1] = 0;
  }
  return prop.multiProcessorCount;
}

extern "C" int rlpyt_host_register(void* host_ptr, int64_t bytes) {
  RL_CHECK_ARG(host_ptr != nullptr && bytes > 0, RLPYT_EINVAL, "rlpyt_host_register: bad range");
  RL_HIP(hipHostRegister(host_ptr, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped));
  return RLPYT_OK;
}

extern "C" int rlpyt_host_device_pointer(void* host_ptr, void** dev_ptr) {
  RL_CHECK_ARG(host_ptr != nullptr && dev_ptr != nullptr, RLPYT_EINVAL,
               "rlpyt_host_device_pointer: null pointer");
  RL_HIP(hipHostGetDevicePointer(dev_ptr, host_ptr, 0));
  return RLPYT_OK;
}

extern "C" int rlpyt_host_unregister(void* host_ptr) {
  RL_CHECK_ARG(host_ptr != nullptr, RLPYT_EINVAL, "rlpyt_host_unregister: null pointer");
  RL_HIP(hipHostUnregister(host_ptr));
  return RLPYT_OK;
}
