// The sampler master's steady-state time-step loop in native code: the role of
// ActionServer.serve_actions (rlpyt/samplers/parallel/gpu/action_server.py:17-74) once the
// per-step device work is a captured hipGraph.  Per step and pipeline group it waits for the
// env workers (futex sequence word), enqueues the H2D copies of the page-locked step buffer,
// launches the group's graph, enqueues the D2H copy of the actions (none when the head kernel
// writes them in place) and records an event; when the event has fired it publishes the actions to
// the workers.
// Running this loop in C removes ~20 interpreter-level calls per group-step from the critical
// path (the device work of a step is ~100 us, so they were of the same order).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <linux/futex.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <thread>

#include "common.h"

namespace {
inline double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
}  // namespace

namespace {
inline bool reached(uint32_t cur, uint32_t target) { return (int32_t)(cur - target) >= 0; }

// Host-dependent uploads + graph launch + action download of one group-step.
int issue_group_step(rlpyt_step_group& g, int t, bool tail) {
  hipStream_t s = (hipStream_t)g.stream;
  *g.t_host = t;
  if (g.dedup) {
    if (t == 0) {
      for (int b = 0; b < g.Bg; ++b) g.slot_host[b] = b;
      RL_HIP(hipMemcpyAsync(g.full_rows_dev, g.obs_host, (size_t)g.Bg * g.row_bytes,
                            hipMemcpyHostToDevice, s));
    } else {
      int k = 0;
      for (int b = 0; b < g.Bg; ++b) {
        if (g.reset_flags[b]) {
          g.slot_host[b] = k;
          RL_HIP(hipMemcpyAsync(g.full_rows_dev + (size_t)k * g.row_bytes,
                                g.obs_host + (size_t)b * g.row_bytes, (size_t)g.row_bytes,
                                hipMemcpyHostToDevice, s));
          ++k;
        } else {
          g.slot_host[b] = -1;
        }
      }
    }
  }
  for (int i = 0; i < g.n_h2d; ++i)
    RL_HIP(hipMemcpyAsync(g.h2d[i].dst, g.h2d[i].src, (size_t)g.h2d[i].nbytes,
                          hipMemcpyHostToDevice, s));
  // tail: the pass on the observation AFTER the last step of the batch (bootstrap value, reward /
  // done rows T; action_server.py:60-62) -- same uploads, its own graph, no actions come back
  RL_HIP(hipGraphLaunch((hipGraphExec_t)(tail ? g.tail_graph_exec : g.graph_exec), s));
  for (int i = 0; i < (tail ? 0 : g.n_d2h); ++i)
    RL_HIP(hipMemcpyAsync(g.d2h[i].dst, g.d2h[i].src, (size_t)g.d2h[i].nbytes,
                          hipMemcpyDeviceToHost, s));
  // (completion through hipStreamWriteValue32 of a page-locked word + a memory poll instead of
  // event record / query was built and measured in round 5: the chain's device leg grew from 67-71
  // to 87-90 us per group-step, 843-870 K -> 769-781 K SPS, profiles/r5_ab_completion_marker.jsonl)
  RL_HIP(hipEventRecord((hipEvent_t)g.event, s));
  return RLPYT_OK;
}
}  // namespace

// Event-driven: every pipeline group runs its own cycle
//   workers stepped (obs word) -> issue -> device done (event) -> publish actions -> ...
// and the hand-offs are serviced as they become ready, so a group never waits for the other
// groups' device work (the first version of this loop -- issue all groups, then wait for all
// events in order -- exposed one full device chain per time step: ~200 us per step at B=256
// whatever the number of groups).  Two threads share the work: the caller issues (4-5 HIP calls,
// ~25 us per group-step), a helper thread retires (event query + the futex wake of the group's
// workers, ~15 us) -- together that is more host work per time step than one thread has time
// for once the device side of a step is down to ~100 us.  Groups may be at different time
// steps; each carries its own `t`.  Ownership of a group's counters moves with its state word
// (release / acquire).
namespace {
enum : int { WAIT_OBS = 0, WAIT_DEV = 1, DONE = 2, FAILED = 3 };

// Idle policy of both threads: poll for `spin_iters` consecutive empty passes (hand-offs are
// ~50 us apart when the pipeline is full), then sleep ~20 us per pass until something moves --
// with slow environments, or when the process tree runs under a tight CPU quota (spin_iters = 0),
// polling threads would only take cores away from the env workers.
inline void idle_pause(int& idle, int spin_iters) {
  if (++idle > spin_iters) {
    struct timespec ts = {0, 20 * 1000};
    nanosleep(&ts, nullptr);
  } else {
    __builtin_ia32_pause();
  }
}

struct ServeShared {
  rlpyt_step_group* groups;
  int n_groups, t_end, device, spin_iters;
  std::atomic<int> state[16];
  int tcur[16];
  std::atomic<int> remaining;
  std::atomic<int> error;      // first error code (0 = none)
  double t_wait_dev;           // retire thread: idle time with work in flight on the device
  // hand-off chain of a group-step, summed over all of them (seconds): workers all arrived ->
  // issue calls returned -> completion seen by the retire thread -> actions published
  double t_ready[16], t_issued[16];
  double lat_issue, lat_device, lat_post;
  long n_steps;
};

void retire_loop(ServeShared* sh) {
  (void)hipSetDevice(sh->device);
  double idle = 0., t_prev = now_s();
  int idle_passes = 0;
  while (sh->remaining.load(std::memory_order_acquire) > 0 &&
         sh->error.load(std::memory_order_relaxed) == 0) {
    bool progressed = false, in_flight = false;
    for (int gi = 0; gi < sh->n_groups; ++gi) {
      if (sh->state[gi].load(std::memory_order_acquire) != WAIT_DEV) continue;
      rlpyt_step_group& g = sh->groups[gi];
      const hipError_t q = hipEventQuery((hipEvent_t)g.event);
      if (q == hipErrorNotReady) {
        in_flight = true;
        continue;
      }
      if (q != hipSuccess) {
        rlpyt::set_error("rlpyt_sampler_serve: hipEventQuery: %s", hipGetErrorString(q));
        sh->error.store(RLPYT_EHIP);
        return;
      }
      const double t_seen = now_s();
      if (sh->tcur[gi] == sh->t_end) {      // that was the tail pass: nothing to publish
        sh->state[gi].store(DONE, std::memory_order_release);
        sh->remaining.fetch_sub(1, std::memory_order_acq_rel);
        progressed = true;
        continue;
      }
      g.acts += 1;
      // CLOCK_MONOTONIC ns of this post, next to the sequence word: a worker that had to wait
      // measures its wake-up latency against it (csrc/envloop.c)
      *reinterpret_cast<volatile uint64_t*>(reinterpret_cast<char*>(g.act_word) + 8) =
          (uint64_t)(t_seen * 1e9);
      rlpyt_seq_post(g.act_word, g.acts);
      sh->lat_device += t_seen - sh->t_issued[gi];
      sh->lat_post += now_s() - t_seen;
      sh->n_steps += 1;
      if (++sh->tcur[gi] == sh->t_end && g.tail_graph_exec == nullptr) {
        sh->state[gi].store(DONE, std::memory_order_release);
        sh->remaining.fetch_sub(1, std::memory_order_acq_rel);
      } else {               // next step -- or, at t_end, the tail pass -- once the workers arrived
        g.rounds += 1;
        sh->state[gi].store(WAIT_OBS, std::memory_order_release);
      }
      progressed = true;
    }
    const double t_now = now_s();
    if (!progressed) {
      if (in_flight) idle += t_now - t_prev;
      idle_pause(idle_passes, sh->spin_iters);
    } else {
      idle_passes = 0;
    }
    t_prev = t_now;
  }
  sh->t_wait_dev = idle;
}
}  // namespace

extern "C" int rlpyt_sampler_serve(rlpyt_step_group* groups, int n_groups, int t_begin, int t_end,
                                   int spin_iters, int timeout_ms, double* timing) {
  RL_CHECK_ARG(groups != nullptr && n_groups > 0 && n_groups <= 16 && t_begin >= 0 &&
                   t_end >= t_begin,
               RLPYT_EINVAL, "rlpyt_sampler_serve: bad arguments");
  if (t_begin == t_end) return RLPYT_OK;
  ServeShared sh;
  sh.spin_iters = spin_iters;
  sh.groups = groups;
  sh.n_groups = n_groups;
  sh.t_end = t_end;
  sh.device = 0;
  (void)hipGetDevice(&sh.device);
  sh.remaining.store(n_groups);
  sh.error.store(0);
  sh.t_wait_dev = 0.;
  sh.lat_issue = sh.lat_device = sh.lat_post = 0.;
  sh.n_steps = 0;
  double lat_issue = 0.;
  for (int gi = 0; gi < n_groups; ++gi) {
    sh.tcur[gi] = t_begin;
    groups[gi].rounds += 1;
    sh.state[gi].store(WAIT_OBS);
  }
  std::thread retire(retire_loop, &sh);
  double t_wait_env = 0., t_issue = 0.;
  double t_prev = now_s(), t_progress = t_prev;
  int rc_out = RLPYT_OK, idle_passes = 0;
  while (sh.remaining.load(std::memory_order_acquire) > 0) {
    if (sh.error.load(std::memory_order_relaxed) != 0) break;
    bool progressed = false, waiting_env = false;
    for (int gi = 0; gi < n_groups; ++gi) {
      if (sh.state[gi].load(std::memory_order_acquire) != WAIT_OBS) continue;
      rlpyt_step_group& g = groups[gi];
      if (!reached(__atomic_load_n(g.obs_word, __ATOMIC_ACQUIRE),
                   g.rounds * (uint32_t)g.n_workers)) {
        waiting_env = true;
        continue;
      }
      const double t0 = now_s();
      const int rc = issue_group_step(g, sh.tcur[gi], sh.tcur[gi] == t_end);
      if (rc != RLPYT_OK) {
        sh.error.store(rc);
        break;
      }
      const double t1 = now_s();
      t_issue += t1 - t0;
      sh.t_issued[gi] = t1;
      lat_issue += t1 - t0;
      sh.state[gi].store(WAIT_DEV, std::memory_order_release);
      progressed = true;
    }
    const double t_now = now_s();
    if (progressed) {
      t_progress = t_now;
      idle_passes = 0;
    } else {
      if (waiting_env) t_wait_env += t_now - t_prev;
      if (timeout_ms > 0 && (t_now - t_progress) * 1e3 > (double)timeout_ms) {
        // progress of the retire thread counts too: only give up when nothing moves
        bool any_dev = false;
        for (int gi = 0; gi < n_groups; ++gi)
          any_dev = any_dev || sh.state[gi].load(std::memory_order_relaxed) == WAIT_DEV;
        if (!any_dev || (t_now - t_progress) * 1e3 > 2.0 * timeout_ms) {
          rlpyt::set_error("rlpyt_sampler_serve: no progress for %d ms (env worker or device hung)",
                           timeout_ms);
          sh.error.store(RLPYT_ETIMEOUT);
          break;
        }
      }
      idle_pause(idle_passes, spin_iters);
    }
    t_prev = t_now;
  }
  retire.join();
  rc_out = sh.error.load();
  if (timing != nullptr) {
    timing[0] += t_wait_env;
    timing[1] += t_issue;
    timing[2] += sh.t_wait_dev;
    // [3..6]: per-group-step chain sums (issue calls, issue-returned -> completion seen, post call)
    // and the number of group-steps they cover
    timing[3] += lat_issue;
    timing[4] += sh.lat_device;
    timing[5] += sh.lat_post;
    timing[6] += (double)sh.n_steps;
  }
  return rc_out;
}
