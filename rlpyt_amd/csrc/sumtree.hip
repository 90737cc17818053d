// HBM-resident float64 sum tree for prioritized replay on gfx950.
//
// Reference: rlpyt/replays/sum_tree.py:8-222 (numpy on the host).  Geometry and index
// rules follow SURVEY.md App. B.3: node i has children 2i+1, 2i+2; levels =
// ceil(log2(T*B+1))+1; leaf(t,b) = low_idx + t*B + b.
//
// Bit-exactness: parents are maintained by ADDING DIFFS in the order np.add.at applies
// them (sum_tree.py:206-209): per level, the diffs of the touched leaves are accumulated
// into their ancestor sequentially in batch order.  propagate_kernel reproduces that order
// (one workgroup per level; the head lane of each run of leaves sharing an ancestor sums
// the run left-to-right in f64), so the whole tree -- not just the sampled indices -- is
// bit-identical to the reference.  Compiled with -ffp-contract=off.
//
// sample(): the descent is latency-bound pointer chasing (levels-1 dependent f64 loads).
// find_kernel descends three levels per round trip by fetching the 7 candidate "left"
// nodes of the next three levels at once (1+2+4 independent loads), cutting the dependent
// chain from 20 to 7 for a 1M-leaf tree; comparisons/subtractions are the reference's.
#include "common.h"
#include <algorithm>

struct rlpyt_sumtree {
  int T, B;
  int64_t size;
  int off_backward, off_forward;
  double default_value;
  int input_priority_shift;
  int levels;
  int64_t low_idx, high_idx, n_nodes;
  int t;                      // cursor
  bool initial_wrap_guard;
  double* tree;               // device, n_nodes
  double* input_priorities;   // device [T,B] or null
  int64_t* prev_idx;          // device: tree idxs of the last sample()
  int64_t* uniq_idx;          // device scratch (sorted unique idxs)
  double* diffs;              // device scratch
  int* d_count;               // device: number of unique idxs
  int64_t prev_cap, diff_cap;
  int n_prev;
  int device;                 // HIP device the tree lives on (hipGetDevice at create time)
};

namespace rlpyt {
namespace {

// The handle owns HBM on ONE device; calls that allocate or launch for it make that device
// current for their duration (the caller's current device may be another rank's GPU).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev && dev >= 0)
      switched = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

struct Run {          // contiguous leaves [leaf0, leaf0+count) with diffs at diffs[off...]
  int64_t leaf0, count, off;
};
struct Runs {
  Run r[4];
  int n;
};

__device__ __forceinline__ int64_t ancestor_at(int64_t node, int up) {
  // parent(i) = (i-1)/2  <=>  (i+1)>>1 - 1 ; applied `up` times.
  return ((node + 1) >> up) - 1;
}

// One workgroup per tree level k in [0, levels-2]; `up` = levels-1-k.
// idx == nullptr: runs of contiguous leaves; else one run over idx[0..count) (ascending).
__global__ __launch_bounds__(256) void propagate_kernel(double* __restrict__ tree, int levels,
                                                        Runs runs,
                                                        const int64_t* __restrict__ idx,
                                                        const double* __restrict__ diffs,
                                                        const int* __restrict__ d_count) {
  const int up = levels - 1 - (int)blockIdx.x;
  for (int r = 0; r < runs.n; ++r) {
    const Run run = runs.r[r];
    const int64_t count = (idx != nullptr && d_count != nullptr) ? (int64_t)*d_count : run.count;
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) {
      const int64_t leaf = idx ? idx[i] : run.leaf0 + i;
      const int64_t a = ancestor_at(leaf, up);
      bool head = (i == 0);
      if (!head) {
        const int64_t prev = idx ? idx[i - 1] : leaf - 1;
        head = ancestor_at(prev, up) != a;
      }
      if (head) {
        double acc = tree[a];
        for (int64_t j = i; j < count; ++j) {
          const int64_t lj = idx ? idx[j] : run.leaf0 + j;
          if (ancestor_at(lj, up) != a) break;
          acc = acc + diffs[run.off + j];
        }
        tree[a] = acc;
      }
    }
    __syncthreads();  // next run may touch the same ancestors (np.add.at is sequential)
  }
}

// ON rows: leaf <- input priority (or default), diff = new - old  (sum_tree.py:162-189)
// OFF rows: leaf <- 0, diff = -old                                 (sum_tree.py:190-200)
__global__ __launch_bounds__(256) void advance_apply_kernel(
    double* __restrict__ tree, int64_t low_idx, const double* __restrict__ input_pri,
    double default_value, Run run, int is_on, double* __restrict__ diffs) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < run.count;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t leaf = run.leaf0 + i;
    const double old = tree[leaf];
    double nv, d;
    if (is_on) {
      nv = input_pri ? input_pri[leaf - low_idx] : default_value;
      d = nv - old;
    } else {
      nv = 0.0;
      d = -old;
    }
    tree[leaf] = nv;
    diffs[run.off + i] = d;
  }
}

// input_priorities[(input_t + i) mod T, :] = priorities (broadcast)  (sum_tree.py:90-95)
__global__ __launch_bounds__(256) void write_input_pri_kernel(
    double* __restrict__ input_pri, const double* __restrict__ pri, int kind, int input_t,
    int T_new, int T, int B) {
  const int64_t total = (int64_t)T_new * B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / B, b = i - r * B;
    int64_t row = (input_t + r) % T;
    if (row < 0) row += T;
    const double v = kind == 1 ? pri[0] : (kind == 2 ? pri[b] : pri[i]);
    input_pri[row * B + b] = v;
  }
}

__global__ __launch_bounds__(256) void fill_f64_kernel(double* __restrict__ p, int64_t n,
                                                       double v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

// find(): sum_tree.py:211-222.  v = tree[0]*u ; repeat levels-1: idx=2idx+1; left=tree[idx];
// if v > left: idx += 1; v -= left.
__global__ __launch_bounds__(64) void find_kernel(const double* __restrict__ tree, int levels,
                                                  int64_t low_idx, int B,
                                                  const double* __restrict__ uniforms, int n,
                                                  int64_t* __restrict__ tree_idx,
                                                  int64_t* __restrict__ T_idxs,
                                                  int64_t* __restrict__ B_idxs,
                                                  double* __restrict__ pri) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = tree[0] * uniforms[i];
  int64_t idx = 0;
  int rem = levels - 1;
  while (rem >= 3) {
    // speculative 3-level fetch: left children along every path below idx
    const int64_t c1 = 2 * idx + 1;                       // level +1 left
    const int64_t c2a = 2 * c1 + 1, c2b = 2 * (c1 + 1) + 1;  // level +2 lefts
    const double l1 = tree[c1];
    const double l2a = tree[c2a], l2b = tree[c2b];
    const double l3a = tree[2 * c2a + 1], l3b = tree[2 * (c2a + 1) + 1];
    const double l3c = tree[2 * c2b + 1], l3d = tree[2 * (c2b + 1) + 1];
    // level +1
    idx = c1;
    bool right1 = v > l1;
    if (right1) { idx += 1; v -= l1; }
    // level +2
    const double l2 = right1 ? l2b : l2a;
    idx = 2 * idx + 1;
    bool right2 = v > l2;
    if (right2) { idx += 1; v -= l2; }
    // level +3
    const double l3 = right1 ? (right2 ? l3d : l3c) : (right2 ? l3b : l3a);
    idx = 2 * idx + 1;
    if (v > l3) { idx += 1; v -= l3; }
    rem -= 3;
  }
  while (rem > 0) {
    idx = 2 * idx + 1;
    const double left = tree[idx];
    if (v > left) { idx += 1; v -= left; }
    --rem;
  }
  tree_idx[i] = idx;
  const int64_t p = idx - low_idx;
  T_idxs[i] = p / B;     // np.divmod(tree_idxs - low_idx, B)  (p >= 0)
  B_idxs[i] = p - (p / B) * B;
  if (pri) pri[i] = tree[idx];
}

// np.unique(prev_idx, return_index=True): sorted unique values + FIRST occurrence index.
// Single workgroup bitonic sort on (idx, pos) keys in LDS; n <= kUniqMax.
constexpr int kUniqMax = 4096;
__global__ __launch_bounds__(1024) void unique_first_kernel(
    const int64_t* __restrict__ prev_idx, int n, int64_t* __restrict__ uniq_idx,
    int* __restrict__ first_pos, int* __restrict__ d_count) {
  __shared__ int64_t key[kUniqMax];
  __shared__ int pos[kUniqMax];
  __shared__ int flags[kUniqMax];
  int P = 1;
  while (P < n) P <<= 1;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    key[i] = i < n ? prev_idx[i] : INT64_MAX;
    pos[i] = i;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool up = ((i & k) == 0);
          const bool gt = (key[i] > key[ixj]) || (key[i] == key[ixj] && pos[i] > pos[ixj]);
          if (gt == up) {
            const int64_t tk = key[i]; key[i] = key[ixj]; key[ixj] = tk;
            const int tp = pos[i]; pos[i] = pos[ixj]; pos[ixj] = tp;
          }
        }
      }
      __syncthreads();
    }
  }
  // head flags + exclusive scan (simple Hillis-Steele over P in LDS)
  for (int i = threadIdx.x; i < P; i += blockDim.x)
    flags[i] = (i < n && (i == 0 || key[i] != key[i - 1])) ? 1 : 0;
  __syncthreads();
  for (int off = 1; off < P; off <<= 1) {
    int tmp[kUniqMax / 1024];
    int c = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x, ++c)
      tmp[c] = flags[i] + (i >= off ? flags[i - off] : 0);
    __syncthreads();
    c = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x, ++c) flags[i] = tmp[c];
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const bool head = (i == 0 || key[i] != key[i - 1]);
    if (head) {
      const int o = flags[i] - 1;
      uniq_idx[o] = key[i];
      first_pos[o] = pos[i];
    }
  }
  if (threadIdx.x == 0) *d_count = n > 0 ? flags[n - 1] : 0;
}

// reconstruct(): diffs = values - tree[idx]; tree[idx] = values  (sum_tree.py:150-153)
__global__ __launch_bounds__(256) void update_leaves_kernel(
    double* __restrict__ tree, const int64_t* __restrict__ uniq_idx,
    const int* __restrict__ first_pos, const int* __restrict__ d_count,
    const double* __restrict__ new_pri, double* __restrict__ diffs) {
  const int n = *d_count;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int64_t leaf = uniq_idx[i];
    const double v = new_pri[first_pos[i]];
    diffs[i] = v - tree[leaf];
    tree[leaf] = v;
  }
}

int ensure_diffs(rlpyt_sumtree* t, int64_t need) {
  if (need <= t->diff_cap) return RLPYT_OK;
  int64_t cap = std::max<int64_t>(need, t->diff_cap * 2);
  if (t->diffs) RL_HIP(hipFree(t->diffs));
  t->diffs = nullptr;
  t->diff_cap = 0;
  RL_HIP(hipMalloc(&t->diffs, cap * sizeof(double)));
  t->diff_cap = cap;
  return RLPYT_OK;
}

int ensure_prev(rlpyt_sumtree* t, int64_t need) {
  if (need <= t->prev_cap) return RLPYT_OK;
  int64_t cap = std::max<int64_t>(need, 1024);
  if (t->prev_idx) RL_HIP(hipFree(t->prev_idx));
  if (t->uniq_idx) RL_HIP(hipFree(t->uniq_idx));
  t->prev_idx = t->uniq_idx = nullptr;
  t->prev_cap = 0;
  RL_HIP(hipMalloc(&t->prev_idx, cap * sizeof(int64_t)));
  // uniq_idx holds [cap] idxs followed by [cap] int first positions
  RL_HIP(hipMalloc(&t->uniq_idx, cap * (sizeof(int64_t) + sizeof(int))));
  t->prev_cap = cap;
  return RLPYT_OK;
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

extern "C" int rlpyt_sumtree_create(rlpyt_sumtree** out, int T, int B, int off_backward,
                                    int off_forward, double default_value,
                                    int enable_input_priorities, int input_priority_shift) {
  RL_CHECK_ARG(out != nullptr, RLPYT_EINVAL, "rlpyt_sumtree_create: null out");
  RL_CHECK_ARG(T > 0 && B > 0 && off_backward >= 0 && off_forward >= 0, RLPYT_EINVAL,
               "rlpyt_sumtree_create: bad geometry T=%d B=%d", T, B);
  rlpyt_sumtree* t = new rlpyt_sumtree();
  *t = rlpyt_sumtree{};
  t->T = T; t->B = B; t->size = (int64_t)T * B;
  t->off_backward = off_backward; t->off_forward = off_forward;
  t->default_value = default_value;
  t->input_priority_shift = input_priority_shift;
  t->device = -1;
  (void)hipGetDevice(&t->device);
  // tree_levels = int(np.ceil(np.log2(size + 1)) + 1)   (sum_tree.py:39)
  int lv = 0;
  while (((int64_t)1 << lv) < t->size + 1) ++lv;  // ceil(log2(size+1))
  t->levels = lv + 1;
  t->low_idx = ((int64_t)1 << (t->levels - 1)) - 1;
  t->high_idx = t->size + t->low_idx;
  t->n_nodes = ((int64_t)1 << t->levels) - 1;
  hipError_t e = hipMalloc(&t->tree, t->n_nodes * sizeof(double));
  if (e == hipSuccess && enable_input_priorities)
    e = hipMalloc(&t->input_priorities, t->size * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&t->d_count, sizeof(int));
  if (e != hipSuccess) {
    set_error("rlpyt_sumtree_create: hipMalloc failed: %s", hipGetErrorString(e));
    rlpyt_sumtree_destroy(t);
    return RLPYT_EHIP;
  }
  int rc = rlpyt_sumtree_reset(t, nullptr);
  if (rc != RLPYT_OK) { rlpyt_sumtree_destroy(t); return rc; }
  RL_HIP(hipStreamSynchronize(nullptr));
  *out = t;
  return RLPYT_OK;
}

extern "C" void rlpyt_sumtree_destroy(rlpyt_sumtree* t) {
  if (!t) return;
  DeviceGuard dev_guard(t->device);
  if (t->tree) (void)hipFree(t->tree);
  if (t->input_priorities) (void)hipFree(t->input_priorities);
  if (t->prev_idx) (void)hipFree(t->prev_idx);
  if (t->uniq_idx) (void)hipFree(t->uniq_idx);
  if (t->diffs) (void)hipFree(t->diffs);
  if (t->d_count) (void)hipFree(t->d_count);
  delete t;
}

extern "C" int rlpyt_sumtree_reset(rlpyt_sumtree* t, rlpyt_stream_t stream) {
  RL_CHECK_ARG(t != nullptr, RLPYT_EINVAL, "rlpyt_sumtree_reset: null handle");
  DeviceGuard dev_guard(t->device);
  hipStream_t s = (hipStream_t)stream;
  RL_HIP(hipMemsetAsync(t->tree, 0, t->n_nodes * sizeof(double), s));
  t->t = 0;
  t->initial_wrap_guard = true;
  t->n_prev = -1;
  if (t->input_priorities) {
    RL_LAUNCH(fill_f64_kernel, dim3(1024), dim3(256), 0, s, t->input_priorities,
                       t->size, t->default_value);
    RL_LAUNCH_CHECK();
  }
  return RLPYT_OK;
}

extern "C" int rlpyt_sumtree_levels(const rlpyt_sumtree* t) { return t ? t->levels : -1; }
extern "C" int64_t rlpyt_sumtree_low_idx(const rlpyt_sumtree* t) { return t ? t->low_idx : -1; }
extern "C" int rlpyt_sumtree_cursor(const rlpyt_sumtree* t) { return t ? t->t : -1; }
extern "C" double* rlpyt_sumtree_data(rlpyt_sumtree* t) { return t ? t->tree : nullptr; }

extern "C" int rlpyt_sumtree_copy_tree(rlpyt_sumtree* t, double* dst, rlpyt_stream_t stream) {
  RL_CHECK_ARG(t && dst, RLPYT_EINVAL, "rlpyt_sumtree_copy_tree: null pointer");
  RL_HIP(hipMemcpyAsync(dst, t->tree, t->n_nodes * sizeof(double), hipMemcpyDeviceToDevice,
                        (hipStream_t)stream));
  return RLPYT_OK;
}

extern "C" int rlpyt_sumtree_advance(rlpyt_sumtree* tr, int T_new, const double* priorities,
                                     int kind, rlpyt_stream_t stream) {
  RL_CHECK_ARG(tr != nullptr, RLPYT_EINVAL, "rlpyt_sumtree_advance: null handle");
  RL_CHECK_ARG(T_new >= 0, RLPYT_EINVAL, "rlpyt_sumtree_advance: negative T");
  RL_CHECK_ARG(kind >= 0 && kind <= 3, RLPYT_EINVAL, "rlpyt_sumtree_advance: bad kind %d", kind);
  RL_CHECK_ARG((kind == 0) == (priorities == nullptr), RLPYT_EINVAL,
               "rlpyt_sumtree_advance: kind/priorities mismatch");
  if (T_new == 0) return RLPYT_OK;
  hipStream_t s = (hipStream_t)stream;
  const int T = tr->T, B = tr->B;
  const int t = tr->t, b = tr->off_backward, f = tr->off_forward;
  auto pymod = [](int64_t a, int64_t m) { int64_t r = a % m; return r < 0 ? r + m : r; };
  // sum_tree.py:75-83
  int64_t low_on_t = pymod(t - b, T);
  int64_t high_on_t = pymod((int64_t)t + T_new - b - 1, T) + 1;
  int64_t low_off_t = pymod((int64_t)t + T_new - b, T);
  int64_t high_off_t = pymod((int64_t)t + T_new + f - 1, T) + 1;
  const bool guard_was_up = tr->initial_wrap_guard;
  bool guard_after = guard_was_up;
  if (guard_was_up) {
    low_on_t = std::max<int64_t>(f, t - b);
    high_on_t = low_off_t = std::max<int64_t>(low_on_t, (int64_t)t + T_new - b);
    if (t + T_new - b >= f) guard_after = false;
  }
  // validate first: an error return leaves guard, cursor and tree untouched (a retry takes the
  // same path); the guard flag is committed with the cursor once the launches are issued
  RL_CHECK_ARG(high_on_t <= T && low_off_t <= T, RLPYT_ESHAPE,
               "rlpyt_sumtree_advance: advance of %d rows overruns the ring during start-up",
               T_new);
  DeviceGuard dev_guard(tr->device);
  if (priorities != nullptr) {
    RL_CHECK_ARG(tr->input_priorities != nullptr, RLPYT_ESTATE,
                 "rlpyt_sumtree_advance: Must enable input priorities.");
    const int input_t = t - tr->input_priority_shift;   // sum_tree.py:90-97
    const int64_t total = (int64_t)T_new * B;
    RL_LAUNCH(write_input_pri_kernel,
                       dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 2048)), dim3(256),
                       0, s, tr->input_priorities, priorities, kind, input_t, T_new, T, B);
    RL_LAUNCH_CHECK();
    if (guard_was_up && input_t < 0) {
      const int64_t rows = -input_t;  // input_priorities[input_t:] = default
      RL_LAUNCH(fill_f64_kernel, dim3(64), dim3(256), 0, s,
                         tr->input_priorities + ((int64_t)T - rows) * B, rows * B,
                         tr->default_value);
      RL_LAUNCH_CHECK();
    }
  }
  // reconstruct_advance: sum_tree.py:155-204
  Runs on{}, off{};
  int64_t doff = 0;
  auto add = [&](Runs& R, int64_t lo_t, int64_t hi_t) {
    if (hi_t <= lo_t) return;
    Run r; r.leaf0 = tr->low_idx + lo_t * B; r.count = (hi_t - lo_t) * B; r.off = doff;
    doff += r.count;
    R.r[R.n++] = r;
  };
  if (high_on_t > low_on_t) add(on, low_on_t, high_on_t);
  else if (high_on_t < low_on_t) { add(on, low_on_t, T); add(on, 0, high_on_t); }
  if (high_off_t > low_off_t) add(off, low_off_t, high_off_t);
  else { add(off, low_off_t, T); add(off, 0, high_off_t); }
  if (doff > 0) {
    int rc = ensure_diffs(tr, doff);
    if (rc != RLPYT_OK) return rc;
    Runs all{};
    for (int i = 0; i < on.n; ++i) {
      const Run r = on.r[i];
      RL_LAUNCH(advance_apply_kernel,
                         dim3((unsigned)std::min<int64_t>(ceil_div(r.count, 256), 2048)),
                         dim3(256), 0, s, tr->tree, tr->low_idx, tr->input_priorities,
                         tr->default_value, r, 1, tr->diffs);
      all.r[all.n++] = r;
    }
    for (int i = 0; i < off.n; ++i) {
      const Run r = off.r[i];
      RL_LAUNCH(advance_apply_kernel,
                         dim3((unsigned)std::min<int64_t>(ceil_div(r.count, 256), 2048)),
                         dim3(256), 0, s, tr->tree, tr->low_idx, (const double*)nullptr, 0.0, r,
                         0, tr->diffs);
      all.r[all.n++] = r;
    }
    RL_LAUNCH_CHECK();
    RL_LAUNCH(propagate_kernel, dim3(tr->levels - 1), dim3(256), 0, s, tr->tree,
                       tr->levels, all, (const int64_t*)nullptr, tr->diffs, (const int*)nullptr);
    RL_LAUNCH_CHECK();
  }
  tr->initial_wrap_guard = guard_after;
  tr->t = (int)pymod((int64_t)t + T_new, T);
  return RLPYT_OK;
}

extern "C" int rlpyt_sumtree_sample(rlpyt_sumtree* t, const double* uniforms, int n,
                                    int64_t* T_idxs, int64_t* B_idxs, double* priorities,
                                    rlpyt_stream_t stream) {
  RL_CHECK_ARG(t && uniforms && T_idxs && B_idxs, RLPYT_EINVAL,
               "rlpyt_sumtree_sample: null pointer");
  RL_CHECK_ARG(n >= 0, RLPYT_EINVAL, "rlpyt_sumtree_sample: negative n");
  DeviceGuard dev_guard(t->device);
  int rc = ensure_prev(t, std::max(n, 1));
  if (rc != RLPYT_OK) return rc;
  t->n_prev = n;
  if (n == 0) return RLPYT_OK;
  RL_LAUNCH(find_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0,
                     (hipStream_t)stream, t->tree, t->levels, t->low_idx, t->B, uniforms, n,
                     t->prev_idx, T_idxs, B_idxs, priorities);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

namespace rlpyt {
namespace {
__global__ __launch_bounds__(256) void set_sampled_kernel(const double* __restrict__ tree,
                                                          int64_t low_idx, int64_t n_leaves,
                                                          const int64_t* __restrict__ leaves, int n,
                                                          int64_t* __restrict__ prev_idx,
                                                          double* __restrict__ priorities) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t leaf = min(max(leaves[i], (int64_t)0), n_leaves - 1);
  prev_idx[i] = low_idx + leaf;
  if (priorities != nullptr) priorities[i] = tree[low_idx + leaf];
}
}  // namespace
}  // namespace rlpyt

// Declares `leaves` (leaf = T_idx * B + B_idx, device int64 [n]) the set the next
// rlpyt_sumtree_update applies to, and returns their priorities: the last step of the reference's
// `sample(n, unique=True)` (rlpyt/replays/sum_tree.py:109-128), whose de-duplication / re-draw loop
// runs on the host RNG stream in the binding.
extern "C" int rlpyt_sumtree_set_sampled(rlpyt_sumtree* t, const int64_t* leaves, int n,
                                         double* priorities, rlpyt_stream_t stream) {
  RL_CHECK_ARG(t && (leaves || n == 0), RLPYT_EINVAL, "rlpyt_sumtree_set_sampled: null pointer");
  RL_CHECK_ARG(n >= 0, RLPYT_EINVAL, "rlpyt_sumtree_set_sampled: negative n");
  DeviceGuard dev_guard(t->device);
  int rc = ensure_prev(t, std::max(n, 1));
  if (rc != RLPYT_OK) return rc;
  t->n_prev = n;
  if (n == 0) return RLPYT_OK;
  RL_LAUNCH(rlpyt::set_sampled_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
            (hipStream_t)stream, t->tree, t->low_idx, (int64_t)t->T * t->B, leaves, n, t->prev_idx,
            priorities);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_sumtree_update(rlpyt_sumtree* t, const double* new_priorities, int n,
                                    rlpyt_stream_t stream) {
  RL_CHECK_ARG(t && new_priorities, RLPYT_EINVAL, "rlpyt_sumtree_update: null pointer");
  RL_CHECK_ARG(t->n_prev >= 0, RLPYT_ESTATE,
               "rlpyt_sumtree_update: no preceding sample() to update");
  RL_CHECK_ARG(n == t->n_prev, RLPYT_ESHAPE,
               "rlpyt_sumtree_update: %d priorities for %d sampled indices", n, t->n_prev);
  RL_CHECK_ARG(n <= kUniqMax, RLPYT_ESHAPE, "rlpyt_sumtree_update: batch %d > %d unsupported", n,
               kUniqMax);
  if (n == 0) return RLPYT_OK;
  DeviceGuard dev_guard(t->device);
  hipStream_t s = (hipStream_t)stream;
  int rc = ensure_diffs(t, n);
  if (rc != RLPYT_OK) return rc;
  int* first_pos = reinterpret_cast<int*>(t->uniq_idx + t->prev_cap);
  RL_LAUNCH(unique_first_kernel, dim3(1), dim3(1024), 0, s, t->prev_idx, n, t->uniq_idx,
                     first_pos, t->d_count);
  RL_LAUNCH_CHECK();
  RL_LAUNCH(update_leaves_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s,
                     t->tree, t->uniq_idx, first_pos, t->d_count, new_priorities, t->diffs);
  RL_LAUNCH_CHECK();
  Runs one{};
  one.n = 1;
  one.r[0].leaf0 = 0; one.r[0].count = n; one.r[0].off = 0;
  RL_LAUNCH(propagate_kernel, dim3(t->levels - 1), dim3(256), 0, s, t->tree, t->levels,
                     one, t->uniq_idx, t->diffs, t->d_count);
  RL_LAUNCH_CHECK();
  // (The reference also replaces prev_tree_idxs by the unique set, sum_tree.py:135; a
  // repeated update recomputes the same unique set here, so nothing to mirror.)
  return RLPYT_OK;
}

// Importance-sampling weights of a prioritized batch (rlpyt/replays/non_sequence/prioritized.py:52-56,
// sequence/prioritized.py:96-100): w = (1 / (p + eps)) ** beta in float64, divided by its maximum, as
// float32 -- ONE launch instead of the seven elementwise / reduction launches of the tensor expression
// (add, reciprocal, pow, max, divide, cast + the exponent's dtype copy).  beta from device memory (captured
// update graphs) or by value.  One workgroup; n <= 65536.
namespace rlpyt {
namespace {
__global__ __launch_bounds__(1024) void is_weights_kernel(const double* __restrict__ pri, int n, double eps,
                                                          const double* __restrict__ beta_dev, double beta,
                                                          float* __restrict__ out) {
  __shared__ double red[16];
  const double b = beta_dev != nullptr ? beta_dev[0] : beta;
  double mx = 0.0;                                      // weights are positive
  for (int i = threadIdx.x; i < n; i += 1024) mx = fmax(mx, pow(1.0 / (pri[i] + eps), b));
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, kWave));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < 16; ++w) mx = fmax(mx, red[w]);
  for (int i = threadIdx.x; i < n; i += 1024) out[i] = (float)(pow(1.0 / (pri[i] + eps), b) / mx);
}
}  // namespace
}  // namespace rlpyt

extern "C" int rlpyt_is_weights_f64(const double* priorities, int64_t n, double eps, const double* beta_dev,
                                    double beta, float* out, rlpyt_stream_t stream) {
  RL_CHECK_ARG(n >= 0 && n <= 65536, RLPYT_EINVAL, "rlpyt_is_weights_f64: need 0 <= n <= 65536");
  if (n == 0) return RLPYT_OK;
  RL_CHECK_ARG(priorities && out, RLPYT_EINVAL, "rlpyt_is_weights_f64: null pointer");
  RL_LAUNCH(rlpyt::is_weights_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, priorities, (int)n, eps,
            beta_dev, beta, out);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
