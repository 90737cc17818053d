// Host-side step synchronisation between the sampler master and its forked env workers
// (role of the per-worker semaphore pairs in rlpyt/samplers/parallel/gpu/action_server.py:44-58
// and collectors.py:29-50).  The reference posts / acquires 2 x n_workers semaphores per time
// step from Python; here every hand-off is ONE 32-bit sequence word in fork-shared memory:
//   * master -> workers: store the step number, one FUTEX_WAKE(all);
//   * workers -> master: atomic increment of an arrival counter, the last arriver wakes
//     the master.
// Waiters poll for `spin_iters` iterations, then sleep in the kernel.  Measured on the
// 256-thread bench host with 64 workers: waking 64 sleepers costs the posting thread ~35 us per
// post; a tree fan-out (each woken sleeper waking 4 more) moved that cost into the workers'
// wake-up latency (+80 us per step) and lost overall; workers polling through the whole device
// phase instead of sleeping lost 2-3x (0.28 -> 0.65 ms per step).  So: short poll, wake all.
// No HIP call in this file: it is safe in forked children.
#include <errno.h>
#include <limits.h>
#include <linux/futex.h>
#include <stdint.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include "../../include/rlpyt_hip.h"

namespace {
inline long futex(uint32_t* addr, int op, uint32_t val, const struct timespec* ts) {
  return syscall(SYS_futex, addr, op, val, ts, nullptr, 0);
}
inline bool reached(uint32_t cur, uint32_t target) { return (int32_t)(cur - target) >= 0; }
}  // namespace

extern "C" int rlpyt_seq_wait(uint32_t* word, uint32_t target, int spin_iters, int timeout_ms) {
  if (!word) return RLPYT_EINVAL;
  for (int i = 0; i < spin_iters; ++i) {
    if (reached(__atomic_load_n(word, __ATOMIC_ACQUIRE), target)) return RLPYT_OK;
    __builtin_ia32_pause();
  }
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (;;) {
    const uint32_t cur = __atomic_load_n(word, __ATOMIC_ACQUIRE);
    if (reached(cur, target)) return RLPYT_OK;
    struct timespec ts = {0, 50 * 1000 * 1000};  // re-check at least every 50 ms
    futex(word, FUTEX_WAIT, cur, &ts);
    if (timeout_ms > 0) {
      struct timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const long ms = (t1.tv_sec - t0.tv_sec) * 1000L + (t1.tv_nsec - t0.tv_nsec) / 1000000L;
      if (ms > timeout_ms) return RLPYT_ETIMEOUT;
    }
  }
}

extern "C" int rlpyt_seq_post(uint32_t* word, uint32_t value) {
  if (!word) return RLPYT_EINVAL;
  __atomic_store_n(word, value, __ATOMIC_RELEASE);
  futex(word, FUTEX_WAKE, INT_MAX, nullptr);
  return RLPYT_OK;
}

extern "C" int rlpyt_seq_arrive(uint32_t* word, uint32_t wake_at) {
  if (!word) return RLPYT_EINVAL;
  const uint32_t now = __atomic_add_fetch(word, 1u, __ATOMIC_ACQ_REL);
  if (reached(now, wake_at)) futex(word, FUTEX_WAKE, INT_MAX, nullptr);
  return RLPYT_OK;
}
