// Measurement aid (no reference counterpart): a stand-in for one rank's share of a ring all-reduce, for sizing the
// data-parallel step on ONE GPU (tools/dp_emulate.py, profiles/r06_dp_emulation.txt).
//
// What an RCCL ring all-reduce of S bytes over N ranks looks like from one GPU: a kernel of `channels` workgroups (RCCL: one
// workgroup of 256 threads per channel, 8...64 of them) that stay resident for the whole collective, each streaming its slice
// of the buffer -- 2 (N - 1) / N x S bytes read and as many written over the 2 (N - 1) steps -- at whatever rate the xGMI
// links deliver, spinning on flags in between.  The stand-in reproduces exactly that footprint: `wgs` workgroups x 256 threads,
// each read-modify-writing (x * 1.0f: the values are unchanged, so a training step that runs it stays correct) its slice
// `traffic` times, optionally paced to `gbps` (aggregate read rate; 0 = as fast as HBM allows) by sleeping on the wall clock --
// a workgroup that waits for a link occupies its CU slot the same way.
#include "common.h"

__global__ __launch_bounds__(256) void comm_standin_kernel(float4* __restrict__ buf, int64_t n4, double traffic, float one,
                                                           double ticks_per_float4) {
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  int64_t hi = lo + per;
  if (hi > n4) hi = n4;
  const int64_t len = hi - lo;
  if (len <= 0) return;
  const int64_t ops = (int64_t)((double)len * traffic);  // float4 read + write pairs of this workgroup
  const unsigned long long t0 = wall_clock64();          // constant-rate counter (100 MHz)
  constexpr int CHUNK = 256 * 16;                        // pace check every 64 KiB read per workgroup
  for (int64_t base = 0; base < ops; base += CHUNK) {
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      const int64_t i = base + k * 256 + threadIdx.x;
      if (i < ops) {
        float4* p = buf + lo + (i % len);
        float4 v = *p;
        v.x *= one;
        v.y *= one;
        v.z *= one;
        v.w *= one;
        *p = v;
      }
    }
    if (ticks_per_float4 > 0.0) {
      const unsigned long long due = t0 + (unsigned long long)((double)(base + CHUNK) * ticks_per_float4);
      while (wall_clock64() < due) __builtin_amdgcn_s_sleep(32);
    }
  }
}

extern "C" int scan_comm_standin(float* buf, int64_t n_floats, double traffic, int32_t wgs, double gbps, void* stream) {
  SCAN_CHECK_ARG(buf && n_floats >= 0 && traffic >= 0.0 && wgs >= 1 && wgs <= 1024 && gbps >= 0.0,
                 "comm_standin: bad arguments (n=%lld traffic=%g wgs=%d gbps=%g)", (long long)n_floats, traffic, wgs, gbps);
  SCAN_CHECK_ARG((reinterpret_cast<uintptr_t>(buf) & 15) == 0, "comm_standin: buffer must be 16-byte aligned");
  const int64_t n4 = n_floats / 4;
  if (n4 == 0 || traffic == 0.0) return 0;
  // gbps = aggregate bytes READ per second over all workgroups; each workgroup moves 16 B per float4 at gbps / wgs;
  // wall_clock64 ticks at 100 MHz
  const double ticks = gbps > 0.0 ? 16.0 * (double)wgs / (gbps * 1e9) * 1e8 : 0.0;
  hipLaunchKernelGGL(comm_standin_kernel, dim3(wgs), dim3(256), 0, as_stream(stream), reinterpret_cast<float4*>(buf), n4, traffic,
                     1.0f, ticks);
  SCAN_LAUNCH_CHECK("comm_standin");
  return 0;
}
