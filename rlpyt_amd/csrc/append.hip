// Replay append for gfx950: the [T, B] rows of a sampler batch into the replay ring in ONE launch.
//
// Reference routines replaced (numpy slice assignments on the host):
//   rlpyt/replays/n_step.py:60-83     BaseNStepReturnBuffer.append_samples: samples[idxs] = ...
//   rlpyt/replays/frame.py:39-59      FrameBufferMixin.append_samples: newest frame of every step,
//                                     the history of row 0, the first C - 1 rows mirrored after a lap
//
// Everything is a SEGMENT of equally long rows copied from a strided source into ring rows that wrap
// at ring_T: a small field is T rows of B * item bytes, the newest frames are T * B rows of one frame
// out of a [T, B, C, frame] stack (source stride C frames), the history / mirror rows are B rows
// each.  The entry point lays the segments out; the kernel is a pure byte mover (HBM-bound, 16 B per
// lane where alignment allows) and every destination byte has exactly one writer, so the result
// does not depend on the order of the workgroups.
#include <algorithm>
#include "common.h"

namespace rlpyt {
namespace {

struct alignas(16) Q16 { uint32_t x, y, z, w; };

constexpr int kMaxSegments = 20;
constexpr int kLanesPerBlock = 256 * 4;       // vector elements one workgroup moves

struct Segment {
  uint8_t* dst;            // ring base of this segment
  const uint8_t* src;
  int64_t rows;            // source rows
  int64_t row_bytes;
  int64_t src_stride;      // bytes between source rows
  int64_t group;           // rows per time step (1: a [T, B * item] field, B: frames)
  int64_t start;           // ring row of source time step 0
  int64_t ring_T;          // wrap (0: no wrap, rows land at start + r)
  int32_t vec;             // bytes per lane access: 16 / 8 / 4 / 1
  uint32_t block0;         // first workgroup of this segment
};
struct Segments {
  int n;
  uint32_t blocks;
  Segment s[kMaxSegments];
};

__global__ __launch_bounds__(256) void replay_append_kernel(Segments segs) {
  int k = 0;
#pragma unroll 1
  while (k + 1 < segs.n && blockIdx.x >= segs.s[k + 1].block0) ++k;
  const Segment& g = segs.s[k];
  const int64_t nvec = g.row_bytes / g.vec;
  const int64_t total = g.rows * nvec;
  const int64_t base = (int64_t)(blockIdx.x - g.block0) * kLanesPerBlock + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = base + u * 256;
    if (i >= total) break;
    const int64_t r = i / nvec, e = i - r * nvec;
    const int64_t step = r / g.group, col = r - step * g.group;
    int64_t ring_row = g.start + step;
    if (g.ring_T > 0) ring_row %= g.ring_T;
    const uint8_t* s = g.src + r * g.src_stride + e * g.vec;
    uint8_t* d = g.dst + (ring_row * g.group + col) * g.row_bytes + e * g.vec;
    switch (g.vec) {
      case 16: *reinterpret_cast<Q16*>(d) = *reinterpret_cast<const Q16*>(s); break;
      case 8: *reinterpret_cast<uint64_t*>(d) = *reinterpret_cast<const uint64_t*>(s); break;
      case 4: *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<const uint32_t*>(s); break;
      default: *d = *s;
    }
  }
}

int widest_vec(const void* dst, const void* src, int64_t row_bytes, int64_t src_stride) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) |
                      (uintptr_t)row_bytes | (uintptr_t)src_stride;
  return (a & 15) == 0 ? 16 : (a & 7) == 0 ? 8 : (a & 3) == 0 ? 4 : 1;
}

bool add_segment(Segments& segs, void* dst, const void* src, int64_t rows, int64_t row_bytes,
                 int64_t src_stride, int64_t group, int64_t start, int64_t ring_T) {
  if (rows <= 0 || row_bytes <= 0) return true;
  if (segs.n == kMaxSegments) return false;
  Segment& g = segs.s[segs.n++];
  g.dst = static_cast<uint8_t*>(dst);
  g.src = static_cast<const uint8_t*>(src);
  g.rows = rows;
  g.row_bytes = row_bytes;
  g.src_stride = src_stride;
  g.group = group;
  g.start = start;
  g.ring_T = ring_T;
  g.vec = widest_vec(dst, src, row_bytes, src_stride);
  g.block0 = segs.blocks;
  segs.blocks += (uint32_t)ceil_div(rows * (row_bytes / g.vec), kLanesPerBlock);
  return true;
}

}  // namespace
}  // namespace rlpyt

extern "C" int rlpyt_replay_append(const rlpyt_append_field* fields, int n_fields,
                                   const uint8_t* obs, uint8_t* frames, int64_t frame_bytes, int C,
                                   int64_t T, int64_t B, int64_t start, int64_t ring_T,
                                   rlpyt_stream_t stream) {
  using namespace rlpyt;
  RL_CHECK_ARG(n_fields >= 0 && (n_fields == 0 || fields), RLPYT_EINVAL,
               "rlpyt_replay_append: null field table");
  RL_CHECK_ARG(T >= 0 && B > 0 && ring_T > 0 && T <= ring_T && start >= 0 && start < ring_T,
               RLPYT_EINVAL, "rlpyt_replay_append: bad sizes (an append is at most one lap)");
  RL_CHECK_ARG((obs == nullptr) == (frames == nullptr), RLPYT_EINVAL,
               "rlpyt_replay_append: obs and frames go together");
  RL_CHECK_ARG(!frames || (frame_bytes > 0 && C >= 1), RLPYT_EINVAL,
               "rlpyt_replay_append: bad frame sizes");
  if (T == 0) return RLPYT_OK;
  Segments segs;
  segs.n = 0;
  segs.blocks = 0;
  bool ok = true;
  for (int f = 0; f < n_fields; ++f) {
    RL_CHECK_ARG(fields[f].ring && fields[f].src && fields[f].row_bytes > 0, RLPYT_EINVAL,
                 "rlpyt_replay_append: field %d: null pointer or empty row", f);
    ok = ok && add_segment(segs, fields[f].ring, fields[f].src, T, fields[f].row_bytes,
                           fields[f].row_bytes, 1, start, ring_T);
  }
  if (frames) {
    const int64_t older = C - 1, stack = (int64_t)C * frame_bytes, row = B * frame_bytes;
    // newest frame of every step: frames[C - 1 + (start + t) % ring_T, b] = obs[t, b, C - 1]
    ok = ok && add_segment(segs, frames + older * row, obs + older * frame_bytes, T * B,
                           frame_bytes, stack, B, start, ring_T);
    const int64_t after = (start + T) % ring_T;
    if (start == 0) {
      // the history of row 0 (frame.py:47-50): frames[f, b] = obs[0, b, f]
      for (int64_t f = 0; f < older; ++f)
        ok = ok && add_segment(segs, frames + f * row, obs + f * frame_bytes, B, frame_bytes,
                               stack, B, 0, 0);
    } else if (older > 0 && after <= start) {
      // the lap closed: frames[j] = frames[ring_T + j] (frame.py:57-59) -- the value AFTER this
      // append, i.e. straight from obs when the source row is one of the rows being written
      for (int64_t j = 0; j < older; ++j) {
        const int64_t newest_row = ring_T - older + j;          // ring time of frames row ring_T + j
        if (newest_row < 0) continue;                           // (ring shorter than the stack)
        const int64_t t = (newest_row - start + ring_T) % ring_T;
        if (t < T)
          ok = ok && add_segment(segs, frames + j * row, obs + (t * B * C + older) * frame_bytes, B,
                                 frame_bytes, stack, B, 0, 0);
        else
          ok = ok && add_segment(segs, frames + j * row, frames + (ring_T + j) * row, B,
                                 frame_bytes, frame_bytes, B, 0, 0);
      }
    }
  }
  RL_CHECK_ARG(ok, RLPYT_EINVAL, "rlpyt_replay_append: more than %d segments", kMaxSegments);
  if (segs.blocks == 0) return RLPYT_OK;
  RL_LAUNCH(replay_append_kernel, dim3(segs.blocks), dim3(256), 0, (hipStream_t)stream, segs);
  RL_LAUNCH_CHECK();
  return RLPYT_OK;
}
