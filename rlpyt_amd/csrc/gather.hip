// Row / frame gathers for gfx950: minibatch extraction out of [T,B,...] rollout batches
// and observation re-assembly out of the unique-frame replay ring, all HBM -> HBM.
//
// Reference routines replaced (Python fancy indexing / list comprehensions on the host):
//   rlpyt/algos/pg/ppo.py:94-100            minibatch slicing  idx -> (idx % T, idx // T)
//   rlpyt/replays/non_sequence/n_step.py:16-43   extract_batch
//   rlpyt/replays/non_sequence/frame.py:14-30    4-frame stack + post-reset blanking
//   rlpyt/replays/sequence/frame.py:17-50        sequence frame stack (wrap + blanking)
//   rlpyt/utils/misc.py:38-56                    extract_sequences
//
// All kernels are pure byte movers (HBM-bound): 16 B per lane where alignment allows,
// one workgroup per destination row for wide rows so a row is one contiguous burst.
#include "common.h"

namespace rlpyt {
namespace {

struct alignas(16) B16 { uint32_t x, y, z, w; };

// ---- index maps -------------------------------------------------------------------------
struct MapTB {            // ppo.py:94-95
  const int64_t* flat; int T; int64_t B;
  __device__ __forceinline__ int64_t row(int64_t m) const {
    const int64_t idx = flat[m];
    return (idx % T) * B + (idx / T);
  }
};
struct MapPair {          // numpy [t_idx, b_idx]; negative t wraps once
  const int64_t* t; const int64_t* b; int T; int64_t B;
  __device__ __forceinline__ int64_t row(int64_t m) const {
    int64_t tt = t[m];
    if (tt < 0) tt += T;
    return tt * B + b[m];
  }
};

// One workgroup per destination row (gridDim.y strides rows), VT = bytes per lane access.
template <typename V, typename Map>
__global__ __launch_bounds__(256) void gather_wide_kernel(const V* __restrict__ src,
                                                          V* __restrict__ dst, Map map,
                                                          int64_t M, int64_t nvec) {
  for (int64_t m = blockIdx.y; m < M; m += gridDim.y) {
    const V* __restrict__ s = src + map.row(m) * nvec;
    V* __restrict__ d = dst + m * nvec;
    for (int64_t v0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; v0 < nvec;
         v0 += (int64_t)gridDim.x * 1024) {
      V tmp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (v0 + k * 256 < nvec) tmp[k] = s[v0 + k * 256];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (v0 + k * 256 < nvec) d[v0 + k * 256] = tmp[k];
    }
  }
}

// Narrow rows: one lane per vector element of the flattened [M, nvec] destination.
template <typename V, typename Map>
__global__ __launch_bounds__(256) void gather_flat_kernel(const V* __restrict__ src,
                                                          V* __restrict__ dst, Map map,
                                                          int64_t M, int64_t nvec) {
  const int64_t total = M * nvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / nvec, e = i - m * nvec;
    dst[i] = src[map.row(m) * nvec + e];
  }
}

template <typename V, typename Map>
int launch_gather_v(const void* src, void* dst, Map map, int64_t M, int64_t elem_bytes,
                    hipStream_t s) {
  const int64_t nvec = elem_bytes / (int64_t)sizeof(V);
  if (nvec >= 256) {
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(nvec, 1024), 64);
    const unsigned gy = (unsigned)std::min<int64_t>(M, 65535);
    RL_LAUNCH((gather_wide_kernel<V, Map>), dim3(gx, gy), dim3(256), 0, s,
                       (const V*)src, (V*)dst, map, M, nvec);
  } else {
    const int64_t total = M * nvec;
    const unsigned g = (unsigned)std::min<int64_t>(ceil_div(total, 256), 256 * 16);
    RL_LAUNCH((gather_flat_kernel<V, Map>), dim3(g), dim3(256), 0, s, (const V*)src,
                       (V*)dst, map, M, nvec);
  }
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

template <typename Map>
int launch_gather(const void* src, void* dst, Map map, int64_t M, int64_t elem_bytes,
                  hipStream_t s) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) |
                      (uintptr_t)elem_bytes;
  if ((a & 15) == 0) return launch_gather_v<B16, Map>(src, dst, map, M, elem_bytes, s);
  if ((a & 7) == 0) return launch_gather_v<uint64_t, Map>(src, dst, map, M, elem_bytes, s);
  if ((a & 3) == 0) return launch_gather_v<uint32_t, Map>(src, dst, map, M, elem_bytes, s);
  return launch_gather_v<uint8_t, Map>(src, dst, map, M, elem_bytes, s);
}

// ---- frame re-assembly ------------------------------------------------------------------
// Channel c of the observation at ring time tt is blank iff done was set at any of the
// C-c-1 steps before tt (frame.py:27-29 / sequence/frame.py:41-48).
__device__ __forceinline__ bool channel_blank(const uint8_t* __restrict__ done, int64_t tt,
                                              int64_t b, int c, int C, int T, int64_t B) {
  bool blank = false;
  for (int f = 1; f <= C - c - 1; ++f) {
    int64_t td = (tt - f) % T;
    if (td < 0) td += T;
    blank = blank || (done[td * B + b] != 0);
  }
  return blank;
}

// grid = (chunks of the frame, C, n);  V = 16 B vectors of one frame (nvec per frame).
template <typename V>
__global__ __launch_bounds__(256) void frames_gather_kernel(
    const V* __restrict__ frames, const uint8_t* __restrict__ done,
    const int64_t* __restrict__ t_idx, const int64_t* __restrict__ b_idx, V* __restrict__ obs,
    int64_t n, int seq_T, int s_stride, int T, int64_t B, int C, int64_t nvec) {
  const int c = blockIdx.y;
  const int64_t items = n * (int64_t)seq_T;
  for (int64_t item = blockIdx.z; item < items; item += gridDim.z) {
    // obs layout [seq_T, n, C, HW]: item = s * n + i; step s is ring time t_idx[i] + s * s_stride
    // (stride 1: a sequence; seq_T = 2, stride n: the agent and the n-step target observation)
    const int64_t s = item / n, i = item - s * n;
    const int64_t b = b_idx[i];
    int64_t tt = (t_idx[i] + s * s_stride) % T;
    if (tt < 0) tt += T;
    const bool blank = channel_blank(done, tt, b, c, C, T, B);  // workgroup-uniform
    const V* __restrict__ src = frames + ((tt + c) * B + b) * nvec;
    V* __restrict__ dst = obs + (item * C + c) * nvec;
    const V zero = V{};         // (memset of a local put it in scratch memory)
    // (as `blank ? zero : src[v]` hipcc selected between the two ADDRESSES, with `zero` in scratch)
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
         v += (int64_t)gridDim.x * blockDim.x) {
      if (blank) dst[v] = zero;
      else dst[v] = src[v];
    }
  }
}

// Wide variant (16-byte vectors): one workgroup per (step, sample) item moves all C frames of
// the stack, 4 independent 16-byte loads in flight per lane (the one-vector-per-lane kernel above
// reached 28 % of HBM peak on the R2D1 batch; this is the gather_wide recipe applied to frames).
__global__ __launch_bounds__(256) void frames_gather_wide_kernel(
    const B16* __restrict__ frames, const uint8_t* __restrict__ done,
    const int64_t* __restrict__ t_idx, const int64_t* __restrict__ b_idx, B16* __restrict__ obs,
    int64_t n, int seq_T, int s_stride, int T, int64_t B, int C, int64_t nvec) {
  const int64_t items = n * (int64_t)seq_T;
  const int64_t total = (int64_t)C * nvec;
  const B16 zero = B16{};
  for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
    const int64_t s = item / n, i = item - s * n;
    const int64_t b = b_idx[i];
    int64_t tt = (t_idx[i] + s * s_stride) % T;
    if (tt < 0) tt += T;
    // blank flags of the C channels (bit c), workgroup-uniform
    unsigned blank_mask = 0;
    for (int c = 0; c < C; ++c)
      blank_mask |= channel_blank(done, tt, b, c, C, T, B) ? (1u << c) : 0u;
    const B16* __restrict__ src0 = frames + (tt * B + b) * nvec;
    B16* __restrict__ dst = obs + item * total;
    const int64_t cstride = B * nvec;
    for (int64_t v0 = threadIdx.x; v0 < total; v0 += 4 * 256) {
      B16 val[4];
      int64_t vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        vv[u] = v0 + u * 256;
        if (vv[u] < total) {
          const int64_t c = vv[u] / nvec, e = vv[u] - c * nvec;
          if ((blank_mask >> c) & 1u) val[u] = zero;
          else val[u] = src0[c * cstride + e];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (vv[u] < total) dst[vv[u]] = val[u];
    }
  }
}

template <typename V>
int launch_frames_v(const uint8_t* frames, const uint8_t* done, const int64_t* t_idx,
                    const int64_t* b_idx, uint8_t* obs, int64_t n, int seq_T, int s_stride, int T,
                    int64_t B, int C, int64_t HW, hipStream_t s) {
  const int64_t nvec = HW / (int64_t)sizeof(V);
  const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(nvec, 256), 16);
  const unsigned gz = (unsigned)std::min<int64_t>(n * seq_T, 65535);
  RL_LAUNCH((frames_gather_kernel<V>), dim3(gx, C, gz), dim3(256), 0, s,
                     (const V*)frames, done, t_idx, b_idx, (V*)obs, n, seq_T, s_stride, T, B, C, nvec);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

int launch_frames(const uint8_t* frames, const uint8_t* done, const int64_t* t_idx,
                  const int64_t* b_idx, uint8_t* obs, int64_t n, int seq_T, int s_stride, int T,
                  int64_t B, int C, int64_t HW, hipStream_t s) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(frames) | reinterpret_cast<uintptr_t>(obs) |
                      (uintptr_t)HW;
  // (a frame-reuse variant -- SB consecutive steps of a sample per workgroup, each lane loading
  //  its column of the SB+C-1 distinct frames once -- measured slower, 153 vs 112 us on the R2D1
  //  batch: the re-reads of the wide kernel are served by L2/MALL anyway)
  if ((a & 15) == 0 && C <= 32 && n * seq_T >= 2048) {
    const int64_t nvec = HW / 16;
    const unsigned g = (unsigned)std::min<int64_t>(n * seq_T, 256 * 32);
    RL_LAUNCH(frames_gather_wide_kernel, dim3(g), dim3(256), 0, s, (const B16*)frames, done,
                       t_idx, b_idx, (B16*)obs, n, seq_T, s_stride, T, B, C, nvec);
    RL_LAUNCH_CHECK();
    return RLPYT_OK;
  }
  if ((a & 15) == 0)
    return launch_frames_v<B16>(frames, done, t_idx, b_idx, obs, n, seq_T, s_stride, T, B, C, HW, s);
  if ((a & 3) == 0)
    return launch_frames_v<uint32_t>(frames, done, t_idx, b_idx, obs, n, seq_T, s_stride, T, B, C, HW, s);
  return launch_frames_v<uint8_t>(frames, done, t_idx, b_idx, obs, n, seq_T, s_stride, T, B, C, HW, s);
}

// ---- extract_sequences (utils/misc.py:38-56) --------------------------------------------
// Non-negative start: dst[s,i] = src[(t+s) mod T, b].  Negative start t<0 ("wrap
// beginning", misc.py:49-51) is reproduced literally: the reference writes
//   sequences[t:, i] = array[t:, b]   (the LAST |t| output rows get the LAST |t| ring rows)
//   sequences[:t, i] = array[:t+T, b] (the first seq_T-|t| output rows get ring rows 0..)
template <typename V>
__global__ __launch_bounds__(256) void gather_seq_kernel(
    const V* __restrict__ src, const int64_t* __restrict__ t_idx,
    const int64_t* __restrict__ b_idx, V* __restrict__ dst, int64_t n, int seq_T, int T,
    int64_t B, int64_t nvec) {
  const int64_t total = (int64_t)seq_T * n * nvec;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx % nvec;
    const int64_t si = idx / nvec;
    const int64_t s = si / n, i = si - s * n;
    const int64_t t = t_idx[i], b = b_idx[i];
    int64_t row;
    if (t < 0) {
      const int64_t head = seq_T + t;            // rows taken from the ring start
      row = (s < head) ? s : (T + (s - seq_T));  // tail rows: array[t:] = ring rows T+t..T-1
    } else {
      row = (t + s) % T;
    }
    dst[idx] = src[(row * B + b) * nvec + e];
  }
}

}  // namespace
}  // namespace rlpyt

using namespace rlpyt;

extern "C" int rlpyt_gather_tb(const void* src, const int64_t* flat_idx, void* dst, int T,
                               int64_t B, int64_t elem_bytes, int64_t M,
                               rlpyt_stream_t stream) {
  RL_CHECK_ARG(src && flat_idx && dst, RLPYT_EINVAL, "rlpyt_gather_tb: null pointer");
  RL_CHECK_ARG(T > 0 && B > 0 && elem_bytes > 0 && M >= 0, RLPYT_EINVAL,
               "rlpyt_gather_tb: bad sizes T=%d B=%ld E=%ld M=%ld", T, (long)B, (long)elem_bytes,
               (long)M);
  if (M == 0) return RLPYT_OK;
  MapTB map{flat_idx, T, B};
  return launch_gather(src, dst, map, M, elem_bytes, (hipStream_t)stream);
}

extern "C" int rlpyt_gather_rows(const void* src, const int64_t* t_idx, const int64_t* b_idx,
                                 void* dst, int T, int64_t B, int64_t elem_bytes, int64_t M,
                                 rlpyt_stream_t stream) {
  RL_CHECK_ARG(src && t_idx && b_idx && dst, RLPYT_EINVAL, "rlpyt_gather_rows: null pointer");
  RL_CHECK_ARG(T > 0 && B > 0 && elem_bytes > 0 && M >= 0, RLPYT_EINVAL,
               "rlpyt_gather_rows: bad sizes");
  if (M == 0) return RLPYT_OK;
  MapPair map{t_idx, b_idx, T, B};
  return launch_gather(src, dst, map, M, elem_bytes, (hipStream_t)stream);
}

// The small fields of one single-step replay batch in ONE launch -- NStepReturnBuffer.extract_batch,
// rlpyt/replays/non_sequence/n_step.py:16-43 minus the observations (rlpyt_frames_gather_pair):
//   prev_action / prev_reward = action / reward of ring row t - 1 (row -1 = the last row), nulled
//     where done[t - 1] (n_step.py:30-33);   action, return_, done, done_n of row t;
//   target_prev_action / target_prev_reward = action / reward of row (t + n_step) % T - 1, as stored.
// One thread per sample; pure index arithmetic and moves: bit-exact.  Replaces eleven row gathers,
// two selects and the index arithmetic between them (~18 launches of a few microseconds).
namespace rlpyt {
namespace {
__global__ __launch_bounds__(256) void replay_step_fields_kernel(
    const int64_t* __restrict__ action, const float* __restrict__ reward,
    const uint8_t* __restrict__ done, const float* __restrict__ return_,
    const uint8_t* __restrict__ done_n, const int64_t* __restrict__ t_idx,
    const int64_t* __restrict__ b_idx, int64_t n, int T, int64_t B, int n_step,
    int64_t* __restrict__ prev_action, float* __restrict__ prev_reward,
    int64_t* __restrict__ out_action, float* __restrict__ out_return, uint8_t* __restrict__ out_done,
    uint8_t* __restrict__ out_done_n, int64_t* __restrict__ tgt_prev_action,
    float* __restrict__ tgt_prev_reward) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t t = t_idx[i];
  const int64_t b = b_idx[i];
  if (t < 0) t += T;                                   // numpy negative indexing (wraps once)
  const int64_t tm1 = t == 0 ? T - 1 : t - 1;
  const int64_t nxt = (t + n_step) % T;
  const int64_t nm1 = nxt == 0 ? T - 1 : nxt - 1;
  const bool was_done = done[tm1 * B + b] != 0;
  prev_action[i] = was_done ? 0 : action[tm1 * B + b];
  prev_reward[i] = was_done ? 0.f : reward[tm1 * B + b];
  out_action[i] = action[t * B + b];
  out_return[i] = return_[t * B + b];
  out_done[i] = done[t * B + b];
  out_done_n[i] = done_n[t * B + b];
  tgt_prev_action[i] = action[nm1 * B + b];
  tgt_prev_reward[i] = reward[nm1 * B + b];
}
}  // namespace
}  // namespace rlpyt

extern "C" int rlpyt_replay_step_fields(const int64_t* action, const float* reward,
                                        const uint8_t* done, const float* return_,
                                        const uint8_t* done_n, const int64_t* t_idx,
                                        const int64_t* b_idx, int64_t n, int T, int64_t B,
                                        int n_step, int64_t* prev_action, float* prev_reward,
                                        int64_t* out_action, float* out_return, uint8_t* out_done,
                                        uint8_t* out_done_n, int64_t* tgt_prev_action,
                                        float* tgt_prev_reward, rlpyt_stream_t stream) {
  RL_CHECK_ARG(action && reward && done && return_ && done_n && t_idx && b_idx && prev_action &&
                   prev_reward && out_action && out_return && out_done && out_done_n &&
                   tgt_prev_action && tgt_prev_reward,
               RLPYT_EINVAL, "rlpyt_replay_step_fields: null pointer");
  RL_CHECK_ARG(n >= 0 && T > 0 && B > 0 && n_step > 0, RLPYT_EINVAL,
               "rlpyt_replay_step_fields: bad sizes");
  if (n == 0) return RLPYT_OK;
  RL_LAUNCH(rlpyt::replay_step_fields_kernel, dim3((unsigned)rlpyt::ceil_div(n, 256)), dim3(256), 0,
            (hipStream_t)stream, action, reward, done, return_, done_n, t_idx, b_idx, n, T, B, n_step,
            prev_action, prev_reward, out_action, out_return, out_done, out_done_n, tgt_prev_action,
            tgt_prev_reward);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

extern "C" int rlpyt_frames_gather(const uint8_t* frames, const uint8_t* done,
                                   const int64_t* t_idx, const int64_t* b_idx, uint8_t* obs,
                                   int64_t n, int T, int64_t B, int C, int64_t HW,
                                   rlpyt_stream_t stream) {
  RL_CHECK_ARG(frames && done && t_idx && b_idx && obs, RLPYT_EINVAL,
               "rlpyt_frames_gather: null pointer");
  RL_CHECK_ARG(n >= 0 && T > 0 && B > 0 && C > 0 && C <= 65535 && HW > 0, RLPYT_EINVAL,
               "rlpyt_frames_gather: bad sizes");
  if (n == 0) return RLPYT_OK;
  return launch_frames(frames, done, t_idx, b_idx, obs, n, 1, 1, T, B, C, HW, (hipStream_t)stream);
}

extern "C" int rlpyt_frames_gather_seq(const uint8_t* frames, const uint8_t* done,
                                       const int64_t* t_idx, const int64_t* b_idx,
                                       uint8_t* obs, int64_t n, int seq_T, int T, int64_t B,
                                       int C, int64_t HW, rlpyt_stream_t stream) {
  RL_CHECK_ARG(frames && done && t_idx && b_idx && obs, RLPYT_EINVAL,
               "rlpyt_frames_gather_seq: null pointer");
  RL_CHECK_ARG(n >= 0 && seq_T > 0 && T > 0 && B > 0 && C > 0 && C <= 65535 && HW > 0,
               RLPYT_EINVAL, "rlpyt_frames_gather_seq: bad sizes");
  RL_CHECK_ARG(seq_T <= T, RLPYT_ESHAPE, "rlpyt_frames_gather_seq: seq_T=%d > ring T=%d", seq_T,
               T);
  if (n == 0) return RLPYT_OK;
  return launch_frames(frames, done, t_idx, b_idx, obs, n, seq_T, 1, T, B, C, HW,
                       (hipStream_t)stream);
}

extern "C" int rlpyt_frames_gather_pair(const uint8_t* frames, const uint8_t* done,
                                        const int64_t* t_idx, const int64_t* b_idx,
                                        uint8_t* obs, int64_t n, int n_step, int T, int64_t B,
                                        int C, int64_t HW, rlpyt_stream_t stream) {
  RL_CHECK_ARG(frames && done && t_idx && b_idx && obs, RLPYT_EINVAL,
               "rlpyt_frames_gather_pair: null pointer");
  RL_CHECK_ARG(n >= 0 && n_step > 0 && T > 0 && B > 0 && C > 0 && C <= 65535 && HW > 0,
               RLPYT_EINVAL, "rlpyt_frames_gather_pair: bad sizes");
  if (n == 0) return RLPYT_OK;
  return launch_frames(frames, done, t_idx, b_idx, obs, n, 2, n_step, T, B, C, HW,
                       (hipStream_t)stream);
}

extern "C" int rlpyt_gather_sequences(const void* src, const int64_t* t_idx,
                                      const int64_t* b_idx, void* dst, int64_t n, int seq_T,
                                      int T, int64_t B, int64_t elem_bytes,
                                      rlpyt_stream_t stream) {
  RL_CHECK_ARG(src && t_idx && b_idx && dst, RLPYT_EINVAL,
               "rlpyt_gather_sequences: null pointer");
  RL_CHECK_ARG(n >= 0 && seq_T > 0 && T > 0 && B > 0 && elem_bytes > 0, RLPYT_EINVAL,
               "rlpyt_gather_sequences: bad sizes");
  if (n == 0) return RLPYT_OK;
  hipStream_t s = (hipStream_t)stream;
  const uintptr_t a = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) |
                      (uintptr_t)elem_bytes;
  const int64_t total_bytes = (int64_t)seq_T * n * elem_bytes;
#define RL_SEQ(V)                                                                          \
  do {                                                                                     \
    const int64_t nvec = elem_bytes / (int64_t)sizeof(V);                                  \
    const unsigned g =                                                                     \
        (unsigned)std::min<int64_t>(ceil_div(total_bytes / (int64_t)sizeof(V), 256), 4096); \
    RL_LAUNCH((gather_seq_kernel<V>), dim3(g), dim3(256), 0, s, (const V*)src, t_idx, \
                       b_idx, (V*)dst, n, seq_T, T, B, nvec);                              \
  } while (0)
  if ((a & 15) == 0) RL_SEQ(B16);
  else if ((a & 3) == 0) RL_SEQ(uint32_t);
  else RL_SEQ(uint8_t);
#undef RL_SEQ
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}

// ---------------------------------------------------------------------------------------
// Fused minibatch gather + uint8 -> float32 (x scale) + CHW -> HWC for the conv stack input
// (replaces: fancy-index gather, .type(float), .mul_(1/255) of
// rlpyt/models/pg/atari_ff_model.py:50-51 and the NCHW->NHWC transpose MIOpen's fp32
// implicit-GEMM kernels need).  src u8 [R, C, HW]; dst f32 [M, HW, C] (channels-last
// storage of a logical [M, C, H, W] tensor).  One workgroup per destination image; a lane
// converts 4 consecutive pixels: C 4-byte plane reads -> 4*C contiguous floats.
// ---------------------------------------------------------------------------------------
namespace rlpyt {
namespace {

template <int C>
__global__ __launch_bounds__(256) void obs_to_nhwc_f32_kernel(
    const uint8_t* __restrict__ src, const int64_t* __restrict__ flat_idx, float* __restrict__ dst,
    int T, int64_t B, int64_t HW, int64_t M, float scale) {
  const int64_t quads = HW / 4;
  for (int64_t m = blockIdx.y; m < M; m += gridDim.y) {
    int64_t row = m;
    if (flat_idx != nullptr) {
      const int64_t idx = flat_idx[m];
      row = (idx % T) * B + (idx / T);
    }
    const uint8_t* __restrict__ s = src + row * C * HW;
    float* __restrict__ d = dst + m * HW * C;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads;
         q += (int64_t)gridDim.x * blockDim.x) {
      uint32_t px[C];
#pragma unroll
      for (int c = 0; c < C; ++c) px[c] = *reinterpret_cast<const uint32_t*>(s + c * HW + q * 4);
      float o[4][C];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < C; ++c) o[p][c] = (float)((px[c] >> (8 * p)) & 0xffu) * scale;
      float* dp = d + q * 4 * C;
      if constexpr (C == 4) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          *reinterpret_cast<float4*>(dp + p * 4) = make_float4(o[p][0], o[p][1], o[p][2], o[p][3]);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int c = 0; c < C; ++c) dp[p * C + c] = o[p][c];
      }
    }
  }
}

__global__ __launch_bounds__(256) void obs_to_nhwc_f32_generic_kernel(
    const uint8_t* __restrict__ src, const int64_t* __restrict__ flat_idx, float* __restrict__ dst,
    int T, int64_t B, int C, int64_t HW, int64_t M, float scale) {
  const int64_t per = HW * C;
  for (int64_t m = blockIdx.y; m < M; m += gridDim.y) {
    int64_t row = m;
    if (flat_idx != nullptr) {
      const int64_t idx = flat_idx[m];
      row = (idx % T) * B + (idx / T);
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < per;
         e += (int64_t)gridDim.x * blockDim.x) {
      const int64_t hw = e / C, c = e - hw * C;
      dst[m * per + e] = (float)src[row * per + c * HW + hw] * scale;
    }
  }
}

}  // namespace
}  // namespace rlpyt

extern "C" int rlpyt_obs_to_nhwc_f32(const uint8_t* src, const int64_t* flat_idx, float* dst,
                                     int T, int64_t B, int C, int64_t HW, int64_t M, float scale,
                                     rlpyt_stream_t stream) {
  RL_CHECK_ARG(T > 0 && B > 0 && C > 0 && HW > 0 && M >= 0, RLPYT_EINVAL,
               "rlpyt_obs_to_nhwc_f32: bad sizes");
  if (M == 0) return RLPYT_OK;
  RL_CHECK_ARG(src && dst, RLPYT_EINVAL, "rlpyt_obs_to_nhwc_f32: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const unsigned gy = (unsigned)std::min<int64_t>(M, 65535);
  const bool fast = (C == 4) && (HW % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(src) & 3) == 0) &&
                    ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  if (fast) {
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(HW / 4, 256), 16);
    RL_LAUNCH((obs_to_nhwc_f32_kernel<4>), dim3(gx, gy), dim3(256), 0, s, src, flat_idx,
                       dst, T, B, HW, M, scale);
  } else {
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(HW * C, 256), 64);
    RL_LAUNCH(obs_to_nhwc_f32_generic_kernel, dim3(gx, gy), dim3(256), 0, s, src,
                       flat_idx, dst, T, B, C, HW, M, scale);
  }
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
