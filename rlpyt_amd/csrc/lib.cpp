// Error reporting and device queries for librlpyt_hip.so (see include/rlpyt_hip.h).
#include "common.h"
#include <cxxabi.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <string>

namespace rlpyt {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- kernel-variant registry --------------------------------------------------------------
namespace {
constexpr int kMaxVariants = 512;
VariantSlot g_slots[kMaxVariants];
std::atomic<int> g_n_slots{0};
std::mutex g_slot_mutex;
thread_local VariantSlot* g_last_slot = nullptr;
thread_local char g_name_buf[1024];

// "void rlpyt::(anonymous namespace)::k<rlpyt::(anonymous namespace)::B16>(B16 const*, ...)"
// -> "k<B16>": drop the return type, the namespaces and the argument list.
void normalise_kernel_name(const char* mangled, char* out, size_t cap) {
  int status = 0;
  char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  std::string s = (status == 0 && dem) ? dem : mangled;
  free(dem);
  for (const char* drop : {"rlpyt::", "(anonymous namespace)::"}) {
    size_t pos;
    while ((pos = s.find(drop)) != std::string::npos) s.erase(pos, strlen(drop));
  }
  if (s.rfind("void ", 0) == 0) s.erase(0, 5);
  int depth = 0;  // cut at the '(' that opens the argument list (outside any <...>)
  for (size_t i = 0; i < s.size(); ++i) {
    if (s[i] == '<') ++depth;
    else if (s[i] == '>') --depth;
    else if (s[i] == '(' && depth == 0) { s.erase(i); break; }
  }
  snprintf(out, cap, "%s", s.c_str());
}

const char* slot_name(const VariantSlot* slot) {
  const char* dev = hipKernelNameRefByPtr(slot->host_fn, nullptr);
  if (dev && *dev) normalise_kernel_name(dev, g_name_buf, sizeof(g_name_buf));
  else snprintf(g_name_buf, sizeof(g_name_buf), "%s", slot->site);
  return g_name_buf;
}
}  // namespace

VariantSlot* variant_slot(const void* host_fn, const char* site) {
  std::lock_guard<std::mutex> lock(g_slot_mutex);
  const int n = g_n_slots.load();
  for (int i = 0; i < n; ++i)
    if (g_slots[i].host_fn == host_fn) return &g_slots[i];
  if (n >= kMaxVariants) return &g_slots[kMaxVariants - 1];
  g_slots[n] = VariantSlot{host_fn, site, 0ull};
  g_n_slots.store(n + 1);
  return &g_slots[n];
}

void variant_hit(VariantSlot* slot) {
  __atomic_fetch_add(&slot->count, 1ull, __ATOMIC_RELAXED);
  g_last_slot = slot;
}
}  // namespace rlpyt

extern "C" const char* rlpyt_hip_last_variant(void) {
  return rlpyt::g_last_slot ? rlpyt::slot_name(rlpyt::g_last_slot) : "";
}

extern "C" void rlpyt_hip_variant_reset(void) {
  const int n = rlpyt::g_n_slots.load();
  for (int i = 0; i < n; ++i) __atomic_store_n(&rlpyt::g_slots[i].count, 0ull, __ATOMIC_RELAXED);
  rlpyt::g_last_slot = nullptr;
}

extern "C" int64_t rlpyt_hip_variant_dump(char* buf, int64_t cap) {
  // "name\tcount\n" per kernel instantiation launched at least once since the last reset.
  int64_t used = 0;
  const int n = rlpyt::g_n_slots.load();
  for (int i = 0; i < n; ++i) {
    const unsigned long long c = __atomic_load_n(&rlpyt::g_slots[i].count, __ATOMIC_RELAXED);
    if (c == 0) continue;
    const char* name = rlpyt::slot_name(&rlpyt::g_slots[i]);
    const int need = snprintf(nullptr, 0, "%s\t%llu\n", name, c);
    if (buf && used + need < cap) snprintf(buf + used, (size_t)(cap - used), "%s\t%llu\n", name, c);
    used += need;
  }
  return used;
}

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
