// The sampler master's steady-state time-step loop in native code: the role of
// ActionServer.serve_actions (rlpyt/samplers/parallel/gpu/action_server.py:17-74) once the
// per-step device work is a captured hipGraph.  Per step and pipeline group it waits for the
// env workers (futex sequence word), enqueues the H2D copies of the page-locked step buffer,
// launches the group's graph, enqueues the D2H copy of the actions and records an event;
// then, group by group, it waits for the event and publishes the actions to the workers.
// Running this loop in C removes ~20 interpreter-level calls per group-step from the critical
// path (the device work of a step is ~100 us, so they were of the same order).
#include <hip/hip_runtime.h>
#include <string.h>
#include <time.h>

#include "common.h"

namespace {
inline double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
}  // namespace

extern "C" int rlpyt_sampler_serve(rlpyt_step_group* groups, int n_groups, int t_begin, int t_end,
                                   int spin_iters, int timeout_ms, double* timing) {
  RL_CHECK_ARG(groups != nullptr && n_groups > 0 && n_groups <= 16 && t_begin >= 0 &&
                   t_end >= t_begin,
               RLPYT_EINVAL, "rlpyt_sampler_serve: bad arguments");
  double t_wait_env = 0., t_issue = 0., t_wait_dev = 0.;
  for (int t = t_begin; t < t_end; ++t) {
    for (int gi = 0; gi < n_groups; ++gi) {
      rlpyt_step_group& g = groups[gi];
      hipStream_t s = (hipStream_t)g.stream;
      double t0 = now_s();
      g.rounds += 1;
      int rc = rlpyt_seq_wait(g.obs_word, g.rounds * (uint32_t)g.n_workers, spin_iters, timeout_ms);
      if (rc != RLPYT_OK) {
        rlpyt::set_error("rlpyt_sampler_serve: env workers of group %d did not report (step %d)", gi, t);
        return rc;
      }
      double t1 = now_s();
      t_wait_env += t1 - t0;
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
      RL_HIP(hipGraphLaunch((hipGraphExec_t)g.graph_exec, s));
      for (int i = 0; i < g.n_d2h; ++i)
        RL_HIP(hipMemcpyAsync(g.d2h[i].dst, g.d2h[i].src, (size_t)g.d2h[i].nbytes,
                              hipMemcpyDeviceToHost, s));
      RL_HIP(hipEventRecord((hipEvent_t)g.event, s));
      t_issue += now_s() - t1;
    }
    for (int gi = 0; gi < n_groups; ++gi) {
      rlpyt_step_group& g = groups[gi];
      double t0 = now_s();
      RL_HIP(hipEventSynchronize((hipEvent_t)g.event));
      t_wait_dev += now_s() - t0;
      g.acts += 1;
      rlpyt_seq_post(g.act_word, g.acts);
    }
  }
  if (timing != nullptr) {
    timing[0] += t_wait_env;
    timing[1] += t_issue;
    timing[2] += t_wait_dev;
  }
  return RLPYT_OK;
}
