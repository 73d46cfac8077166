// Ping-pong latency between two CUs of the same XCD through L2, for several store / load flavours.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/pingpong tools/micro/pingpong.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }

template <int ST>
__device__ __forceinline__ void put(unsigned* p, unsigned v) {
  if (ST == 0) __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // plain store
  else if (ST == 1) __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // sc1 store
  else if (ST == 2) __hip_atomic_exchange((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // RMW at L2
  else if (ST == 3) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                      __builtin_amdgcn_s_waitcnt(0x0070); }                                          // store + vmcnt(0)
  else if (ST == 4) __hip_atomic_store((gu32*)p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // release store
}
template <int LD>
__device__ __forceinline__ unsigned get(unsigned* p) {
  if (LD == 0) return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (LD == 1) return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else return __hip_atomic_fetch_add((gu32*)p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ctrl[0..7] tickets; flags at f[0], f[64]; out[0] = cycles; SAMEX: partner on the same XCD or on another one
template <int ST, int LD>
__global__ __launch_bounds__(64) void pingpong(unsigned* ctrl, unsigned* f, u64* out, int n, int samex, int nw) {
  extern __shared__ float pad[];
  __shared__ int role;
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id() & 7u;
    const unsigned tk = atomicAdd(ctrl + x, 1u);
    role = -1;
    if (x == 0 && tk == 0) role = 0;
    if (samex ? (x == 0 && tk == 1) : (x == 1 && tk == 0)) role = 1;
  }
  __syncthreads();
  if (role < 0) return;
  unsigned* mine = f + (role == 0 ? 0 : 64);
  unsigned* other = f + (role == 0 ? 64 : 0);
  const int lane = threadIdx.x;
  u64 t0 = 0;
  for (int i = 1; i <= n; ++i) {
    if (i == 101 && lane == 0) t0 = clock64();
    if (role == 0) {
      if (lane < nw) put<ST>(other + lane, (unsigned)i);
      for (unsigned sp = 0; sp < 200000u; ++sp) { unsigned v = lane < nw ? get<LD>(mine + lane) : (unsigned)i; if (__all(v == (unsigned)i)) break; __builtin_amdgcn_s_sleep(1); }
    } else {
      for (unsigned sp = 0; sp < 200000u; ++sp) { unsigned v = lane < nw ? get<LD>(mine + lane) : (unsigned)i; if (__all(v == (unsigned)i)) break; __builtin_amdgcn_s_sleep(1); }
      if (lane < nw) put<ST>(other + lane, (unsigned)i);
    }
  }
  if (role == 0 && lane == 0) out[0] = clock64() - t0;
}

template <int ST, int LD>
void run(const char* name, unsigned* ctrl, unsigned* f, u64* out, int samex, int nw) {
  const int n = 1100;
  hipMemset(ctrl, 0, 64); hipMemset(f, 0, 1024); hipMemset(out, 0, 8);
  hipFuncSetAttribute((const void*)pingpong<ST, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024);
  hipLaunchKernelGGL((pingpong<ST, LD>), dim3(256), dim3(64), 90 * 1024, 0, ctrl, f, out, n, samex, nw);
  hipDeviceSynchronize();
  u64 h = 0; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("%-44s %s nw=%2d : %7.0f cycles per one-way hand-off\n", name, samex ? "same XCD " : "other XCD", nw, (double)h / (n - 100) / 2.0);
  fflush(stdout);
}

int main() {
  unsigned *ctrl, *f; u64* out;
  hipMalloc(&ctrl, 64); hipMalloc(&f, 1024); hipMalloc(&out, 8);
  for (int samex = 1; samex >= 1; --samex) {
    run<0, 0>("plain store / agent(sc1) load", ctrl, f, out, samex, 1);
    run<1, 0>("agent(sc1) store / agent load", ctrl, f, out, samex, 1);
    run<2, 0>("atomic exchange / agent load", ctrl, f, out, samex, 1);
    run<3, 0>("plain store + vmcnt(0) / agent load", ctrl, f, out, samex, 1);
    run<4, 0>("release store / agent load", ctrl, f, out, samex, 1);
    run<0, 1>("plain store / system load", ctrl, f, out, samex, 1);
    run<0, 2>("plain store / fetch_add(0) load", ctrl, f, out, samex, 1);
    run<2, 2>("atomic exchange / fetch_add(0) load", ctrl, f, out, samex, 1);
    run<0, 0>("plain store / agent load", ctrl, f, out, samex, 32);
  }
  // across XCDs only flavours that reach the memory side (a plain store stays in the producer's L2: never visible)
  run<1, 1>("agent(sc1) store / system load", ctrl, f, out, 0, 1);
  run<2, 2>("atomic exchange / fetch_add(0) load", ctrl, f, out, 0, 1);
  return 0;
}
