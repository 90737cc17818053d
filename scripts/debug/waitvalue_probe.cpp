// Probe: can the GPU's command processor take the sampler's hand-off itself?
//   stream: hipStreamWaitValue32(arrive >= k) -> [H2D copy] -> kernel -> hipStreamWriteValue32(act = k)
// enqueued AHEAD of time; the host then only flips `arrive` (as the env workers would) and polls `act`.
// Prints the round-trip latency host-write -> host-sees-act for: no payload, a tiny kernel, a
// 533 KB H2D copy + tiny kernel.  Build: hipcc --offload-arch=gfx950 -O2 waitvalue_probe.cpp -o waitvalue_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now_us() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

__global__ void touch(int* p, const unsigned char* src, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(p, (int)src[i * 64 % n] & 1);
}

int main() {
  int dev = 0, can = -1;
  CK(hipSetDevice(dev));
  hipError_t e = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev);
  printf("hipDeviceAttributeCanUseStreamWaitValue: %d (%s)\n", can, hipGetErrorString(e));
  // fork-shared style memory, registered afterwards (what the sampler has)
  const size_t words_bytes = 4096, blk = 64 * 8320 + 1040;
  uint32_t* words = (uint32_t*)mmap(NULL, words_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  unsigned char* hblk = (unsigned char*)mmap(NULL, blk, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset(words, 0, words_bytes);
  memset(hblk, 3, blk);
  CK(hipHostRegister(words, words_bytes, hipHostRegisterMapped));
  CK(hipHostRegister(hblk, blk, hipHostRegisterMapped));
  void* dwords = nullptr;
  CK(hipHostGetDevicePointer(&dwords, words, 0));
  volatile uint32_t* arrive = words;          // host writes
  volatile uint32_t* act = words + 32;        // GPU writes
  uint32_t* d_arrive = (uint32_t*)dwords;
  uint32_t* d_act = (uint32_t*)dwords + 32;
  unsigned char* dblk = nullptr;
  int* dcnt = nullptr;
  CK(hipMalloc(&dblk, blk));
  CK(hipMalloc(&dcnt, 4));
  CK(hipMemset(dcnt, 0, 4));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int mode = 0; mode < 3; ++mode) {
    const int N = 200;
    const uint32_t base = *act;
    double t_enq0 = now_us();
    for (int k = 1; k <= N; ++k) {
      CK(hipStreamWaitValue32(s, d_arrive, base + k, hipStreamWaitValueGte, 0xffffffffu));
      if (mode == 2) CK(hipMemcpyAsync(dblk, hblk, blk, hipMemcpyHostToDevice, s));
      if (mode >= 1) hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, s, dcnt, dblk, 16384);
      CK(hipStreamWriteValue32(s, d_act, base + k, 0));
    }
    double t_enq1 = now_us();
    double sum = 0, mx = 0, mn = 1e9;
    for (int k = 1; k <= N; ++k) {
      usleep(50);                       // the "env stepping"
      const double t0 = now_us();
      __atomic_store_n((uint32_t*)arrive, base + k, __ATOMIC_RELEASE);
      double t1;
      for (;;) {
        if (__atomic_load_n((uint32_t*)act, __ATOMIC_ACQUIRE) == base + k) { t1 = now_us(); break; }
        if (now_us() - t0 > 2e6) { printf("TIMEOUT mode %d k %d act=%u\n", mode, k, *act); return 2; }
      }
      const double d = t1 - t0;
      sum += d; if (d > mx) mx = d; if (d < mn) mn = d;
    }
    CK(hipStreamSynchronize(s));
    printf("mode %d (%s): enqueue %.1f us per step (%d calls); round trip mean %.1f min %.1f max %.1f us\n", mode,
           mode == 0 ? "wait+write" : mode == 1 ? "wait+kernel+write" : "wait+H2D 533KB+kernel+write",
           (t_enq1 - t_enq0) / N, mode == 0 ? 2 : mode == 1 ? 3 : 4, sum / N, mn, mx);
  }
  // reference: the same payloads issued AFTER the flag flips, completion through an event (today's path)
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int mode = 1; mode < 3; ++mode) {
    double sum = 0;
    const int N = 200;
    for (int k = 0; k < N; ++k) {
      usleep(50);
      const double t0 = now_us();
      if (mode == 2) CK(hipMemcpyAsync(dblk, hblk, blk, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, s, dcnt, dblk, 16384);
      CK(hipEventRecord(ev, s));
      while (hipEventQuery(ev) == hipErrorNotReady) {}
      sum += now_us() - t0;
    }
    printf("issue-after-flip reference, mode %d: mean %.1f us\n", mode, sum / N);
  }
  printf("OK\n");
  return 0;
}
