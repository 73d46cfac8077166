// persist.h — shared pieces of the persistent, XCD-local sequence kernels (lstm_persist.hip, dec_persist.hip):
// role assignment from HW_REG_XCC_ID, the L2-resident exchange primitives and the bounded poll.
//
// Exchange protocol.  Producer and consumers of a group sit on the SAME XCD by construction (the group id is the
// XCC id), so a producer's store may stay in that XCD's L2 (plain workgroup-scope store; the L1 is write-through)
// and consumers read with agent-scope (sc1) loads that bypass their stale L1.  The data is the flag: either an
// 8-byte {tag, value} granule, or a bare fp32 word whose mantissa LSB is a validity bit that flips each time the
// slot is rewritten (buffers start zeroed = invalid).  No fences, no counters.  Every spin is bounded; a timeout
// raises the abort word ctrl[8] (reason in ctrl[9]) and every other poll loop drains.
#pragma once
#include "common.h"

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

// ~1 us per attempt (round trip + s_sleep): a wait gives up after ~1-2 s.  Long enough for a workgroup of the same launch
// that is still waiting for its CU (another kernel - e.g. a collective that waits for a late rank - holding resources
// there), short enough that a placement that can never complete ends in an abort, not in a hang.
constexpr unsigned SPIN_LIMIT = 2000000u;

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }

__device__ __forceinline__ u64 granule_load(const u64* p) {
  return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(u64* p, unsigned tag, float v) {
  __hip_atomic_store((gu64*)p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned flag_load(const unsigned* p) {
  return __hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(unsigned* p, unsigned v) {
  __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LSB-tagged fp32 words
__device__ __forceinline__ float tag_word(float v, unsigned bit) {
  return __uint_as_float((__float_as_uint(v) & ~1u) | bit);
}
__device__ __forceinline__ unsigned tag_bit_of_step(int s) { return (((unsigned)s >> 1) & 1u) ^ 1u; }
__device__ __forceinline__ void word_store(float* p, float v, unsigned bit) {
  __hip_atomic_store((gu32*)p, (__float_as_uint(v) & ~1u) | bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Control block handed over by the caller: 32 words = [16 latch words | 16 per-launch words].
//   per-launch words (what the kernels receive as `ctrl`; zeroed before every launch): [0..7] tickets per XCC, [8] abort,
//     [9] abort code;
//   latch words (ctrl - 16 inside a kernel; NEVER written by the host side of the library): [0] abort latch, [1] code of the
//     abort that set it.  A sequence operator is several launches (one per LSTM layer, per row block, the decoder); the
//     per-launch abort word of all but the last is gone when the host looks, the latch is not.  The caller clears it.
static inline unsigned* persist_launch_words(void* ctrl) { return (unsigned*)ctrl + 16; }
// FAULT instantiations (template parameter of lstm_persist_fwd_bf3_kernel and dec_persist_fwd_kernel; the launch argument
// ASR_DEBUG_FAULT / the entry asr_dec_seq_fwd_persist_fault select them; tests only): slice 1 of group 0 stops publishing
// after its first step and every wait gives up after DEBUG_SPIN_LIMIT attempts instead of SPIN_LIMIT - the kernels' own abort
// path (bounded spin expires -> raise_abort -> NaN poison -> every other workgroup drains) runs in a millisecond.  A template
// parameter, not a run-time flag: with the flag in the shipped instantiation the decoder forward ran 5 % slower (10.2 ->
// 10.7 us per step; same lesson as the row map of lstm_persist.hip).
constexpr unsigned DEBUG_SPIN_LIMIT = 4096u;
__device__ __forceinline__ void raise_abort(unsigned* ctrl, unsigned code) {
  flag_store(ctrl + 9, code);
  flag_store(ctrl + 8, 1u);
  flag_store(ctrl - 15, code);
  flag_store(ctrl - 16, 1u);
}

// Host side: zero the per-launch control words and `bytes` of exchange before a launch.  When the caller placed the
// 128-byte control block directly in front of the exchange buffer (hip_backend.persist_scratch does) it is ONE fill.
static inline hipError_t persist_reset(void* xch, void* ctrl, size_t bytes, hipStream_t stream) {
  char* lw = (char*)persist_launch_words(ctrl);
  if (lw + 64 == (char*)xch) return hipMemsetAsync(lw, 0, 64 + bytes, stream);
  hipError_t e = hipMemsetAsync(lw, 0, 16 * sizeof(unsigned), stream);
  if (e != hipSuccess) return e;
  return hipMemsetAsync(xch, 0, bytes, stream);
}

// role of this workgroup: (group g in 0..7 = XCC id, slice in 0..31 = arrival ticket); slice < 0 = no role
__device__ __forceinline__ void take_role(unsigned* ctrl, int* lds_role, int& g, int& slice) {
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id() & 7u;
    const unsigned tk = atomicAdd(ctrl + x, 1u);
    lds_role[0] = (int)x;
    lds_role[1] = tk < 32u ? (int)tk : -1;
    if (tk >= 32u) raise_abort(ctrl, 2u);   // unexpected placement
  }
  __syncthreads();
  // workgroup-uniform: in SGPRs, so that everything derived from the role (direction, rows, slices of the weights,
  // exchange addresses) is scalar arithmetic instead of per-lane 64-bit VALU math
  g = __builtin_amdgcn_readfirstlane(lds_role[0]);
  slice = __builtin_amdgcn_readfirstlane(lds_role[1]);
}

// Poll N 8-byte slots, each holding two LSB-tagged fp32 words, until every word carries tag bit `want`.  The last
// slot is the sentinel: the other N-1 are only fetched once it is valid (a failed poll of everything would
// saturate the XCD's L2 and delay the producers themselves).  Lanes without work point at any slot that becomes
// valid in the same hand-off.  FULL = fetch all N slots on every attempt (one L2 round trip less when the data is
// already there; only for small N).  Returns with `aborted` set (values undefined) on a timeout or a raised abort word.
template <int N, bool FULL = false>
__device__ __forceinline__ void poll_pairs(const u64* const (&p)[N], unsigned want, u64 (&v)[N], unsigned* ctrl,
                                           bool& aborted, unsigned code, unsigned limit = SPIN_LIMIT) {
  const u64 m = 0x0000000100000001ull;
  const u64 expect = want ? m : 0ull;
  if (aborted) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = 0ull;
    return;
  }
  unsigned spins = 0;
  while (true) {
    if (FULL) {
      bool ok = true;
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = granule_load(p[i]);
#pragma unroll
      for (int i = 0; i < N; ++i) ok = ok && ((v[i] & m) == expect);
      if (__all(ok)) return;
    } else {
      v[N - 1] = granule_load(p[N - 1]);
      if (__all((v[N - 1] & m) == expect)) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N - 1; ++i) v[i] = granule_load(p[i]);
#pragma unroll
        for (int i = 0; i < N - 1; ++i) ok = ok && ((v[i] & m) == expect);
        if (__all(ok)) return;
      }
    }
#ifdef ASR_NO_POLL   /* measurement only: never wait (results are garbage) */
    return;
#endif
    if (++spins > limit || ((spins & 63u) == 0u && flag_load(ctrl + 8) != 0u)) {
      if ((threadIdx.x & 63) == 0) raise_abort(ctrl, code);
      aborted = true;
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
// The same poll over 16-byte slots (four LSB-tagged words each) read with ONE sc1 buffer load per slot: agent-scope
// atomic loads stop at 8 bytes, and in these latency-bound exchanges the number of load instructions per lane is what
// a gather costs (LSTM hand-offs: 16 -> 8 loads per lane took 5-12 % off the step).  `off` = byte offsets into the
// exchange buffer described by `rs` (make_xch_rsrc).  FULL as in poll_pairs.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_xch_rsrc(void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);      // raw dwords, no swizzle
}
__device__ __forceinline__ bool quad_ok(const u4v& q, unsigned tb) {
  return ((((q.x ^ tb) | (q.y ^ tb) | (q.z ^ tb) | (q.w ^ tb)) & 1u) == 0u);
}
template <int N, bool FULL = false>
__device__ __forceinline__ void poll_quads(__amdgpu_buffer_rsrc_t rs, const unsigned (&off)[N], unsigned want, u4v (&v)[N],
                                           unsigned* ctrl, bool& aborted, unsigned code, unsigned limit = SPIN_LIMIT) {
  const unsigned tb = want ? 1u : 0u;
  if (aborted) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (u4v){0u, 0u, 0u, 0u};
    return;
  }
  unsigned spins = 0;
  while (true) {
    bool ok = true;
    if (FULL) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[i], 0, 16);
#pragma unroll
      for (int i = 0; i < N; ++i) ok = ok && quad_ok(v[i], tb);
      if (__all(ok)) return;
    } else {
      v[N - 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[N - 1], 0, 16);
      if (__all(quad_ok(v[N - 1], tb))) {
#pragma unroll
        for (int i = 0; i < N - 1; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[i], 0, 16);
#pragma unroll
        for (int i = 0; i < N - 1; ++i) ok = ok && quad_ok(v[i], tb);
        if (__all(ok)) return;
      }
    }
#ifdef ASR_NO_POLL
    return;
#endif
    if (++spins > limit || ((spins & 63u) == 0u && flag_load(ctrl + 8) != 0u)) {
      if ((threadIdx.x & 63) == 0) raise_abort(ctrl, code);
      aborted = true;
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ float pair_lo(u64 v) { return __uint_as_float((unsigned)v); }
__device__ __forceinline__ float pair_hi(u64 v) { return __uint_as_float((unsigned)(v >> 32)); }

}  // namespace
