// What does the hand-off of the persistent LSTM kernels cost without any arithmetic?  32 CUs of one XCD, 8 waves each; per
// step every CU publishes its 1/32 of a tile (plain stores, the step number as the value) and every wave polls its 1/8 of
// the WHOLE tile with L1-bypassing 16-byte loads until all words carry the step, then one workgroup barrier.  Cycles per
// step by tile size (8 rows x 512 units x 4 B = 16 KB is the forward kernel's tile, 32 KB its 16-row form).
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/allgather tools/micro/allgather.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }

template <int QPL, int SLEEP, int LDV>
__global__ __launch_bounds__(512) void allgather(unsigned* ctrl, unsigned* buf, u64* out, int steps, int ncu) {
  extern __shared__ float pad[];
  __shared__ int role;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    const unsigned x = xcc_id() & 7u;
    const unsigned tk = atomicAdd(ctrl + x, 1u);
    role = (x == 0 && tk < (unsigned)ncu) ? (int)tk : -1;
  }
  __syncthreads();
  const int slice = role;
  if (slice < 0) return;
  constexpr int WPW = QPL * 64 * 4;          // words per wave
  constexpr int TOT = WPW * 8;               // words of the tile
  const int wslice = TOT / ncu;              // words this CU publishes
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7ffffff0, 0x00020000);
  u64 t0 = 0;
  unsigned bad_total = 0;
  u64 rounds = 0;
  for (int s = 1; s <= steps; ++s) {
    if (s == 101 && tid == 0) t0 = clock64();
    unsigned* b = buf + (s & 1) * TOT;
    for (int i = tid; i < wslice; i += 512) __hip_atomic_store((gu32*)(b + slice * wslice + i), (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned spins = 0;
    while (true) {
      u32x4 q[QPL];
      if (LDV == 3) asm volatile("buffer_inv sc1" ::: "memory");
#pragma unroll
      for (int k = 0; k < QPL; ++k)
      {
        const unsigned off = (unsigned)((s & 1) * TOT + wave * WPW + (k * 64 + lane) * 4) * 4u;
        if (LDV == 0) q[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);        // sc1: agent scope, the kernels' flavour
        else if (LDV == 1) q[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 17);   // sc0 sc1: system scope
        else if (LDV == 2) q[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 18);   // sc1 nt
        else if (LDV == 3) q[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);    // wave scope behind an L1 invalidate
        else q[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 1);                  // sc0: workgroup scope
      }
      unsigned bad = 0;
#pragma unroll
      for (int k = 0; k < QPL; ++k) bad |= (q[k].x ^ (unsigned)s) | (q[k].y ^ (unsigned)s) | (q[k].z ^ (unsigned)s) | (q[k].w ^ (unsigned)s);
      if (__all(bad == 0u)) break;
      ++rounds;
      if (++spins > 200000u) { bad_total = 1; break; }
      if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    }
    __syncthreads();
  }
  if (slice == 0 && tid == 0) { out[0] = clock64() - t0; out[1] = bad_total; out[2] = rounds; }
}

template <int QPL, int SLEEP, int LDV = 0>
void run(unsigned* ctrl, unsigned* buf, u64* out, int ncu) {
  const int n = 1100;
  hipMemset(ctrl, 0, 64); hipMemset(buf, 0, 2 * QPL * 64 * 4 * 8 * 4); hipMemset(out, 0, 24);
  hipFuncSetAttribute((const void*)allgather<QPL, SLEEP, LDV>, hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024);
  hipLaunchKernelGGL((allgather<QPL, SLEEP, LDV>), dim3(256), dim3(512), 90 * 1024, 0, ctrl, buf, out, n, ncu);
  hipDeviceSynchronize();
  u64 h[3] = {0, 0, 0}; hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
  static const char* ldn[] = {"sc1", "sc0 sc1", "sc1 nt", "inv + plain", "sc0"};
  printf("tile %3d KB, %2d CUs, s_sleep %d, loads %-11s: %7.0f cycles per step, %.2f failed poll rounds per step (wave 0 of slice 0)%s\n",
         QPL * 64 * 16 * 8 / 1024, ncu, SLEEP, ldn[LDV], (double)h[0] / (n - 100), (double)h[2] / n, h[1] ? "  (TIMED OUT)" : "");
  fflush(stdout);
}

int main() {
  unsigned *ctrl, *buf; u64* out;
  hipMalloc(&ctrl, 64); hipMalloc(&buf, 1 << 20); hipMalloc(&out, 24);
  for (int ncu : {32, 2}) {
    run<1, 1>(ctrl, buf, out, ncu); run<2, 1>(ctrl, buf, out, ncu); run<4, 1>(ctrl, buf, out, ncu); run<8, 1>(ctrl, buf, out, ncu);
  }
  run<2, 0>(ctrl, buf, out, 32); run<2, 4>(ctrl, buf, out, 32); run<2, 8>(ctrl, buf, out, 32);
  run<2, 1, 1>(ctrl, buf, out, 32); run<2, 1, 2>(ctrl, buf, out, 32); run<2, 1, 3>(ctrl, buf, out, 32); run<2, 1, 4>(ctrl, buf, out, 32);
  run<4, 1, 2>(ctrl, buf, out, 32); run<4, 1, 3>(ctrl, buf, out, 32);
  return 0;
}
