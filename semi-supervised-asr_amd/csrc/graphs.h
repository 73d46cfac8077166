// graphs.h — memo of instantiated HIP graphs for the per-time-step launch chains.
// A chain (e.g. the 800 step kernels of one LSTM layer) is pure device work whose kernel arguments are
// determined by the call's argument tuple.  The first time a tuple is seen the chain runs eagerly; the second
// time it is stream-captured into a hipGraph and instantiated; from then on it is replayed with one
// hipGraphLaunch.  This removes the host launch cost (~8 us per kernel from Python+HIP, which makes the chains
// host-bound) without changing any kernel.  Steady-state training re-uses the same buffer addresses (caching
// allocator), so the memo hits; a miss simply runs eagerly.  The cache is owned by the caller (opaque handle),
// one per launching host thread — no global state, no locking.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>

struct AsrGraphEntry {
  std::vector<unsigned char> key;
  int seen = 0;
  bool failed = false;
  hipGraphExec_t exec = nullptr;
  uint64_t stamp = 0;
};

struct AsrGraphCache {
  std::vector<AsrGraphEntry> entries;
  size_t max_entries = 64;
  uint64_t clock = 0;
  uint64_t hits = 0, captures = 0, eager = 0;
};

// Runs `launch(stream)` (which must only enqueue kernels on `stream`) eagerly, captured, or as a replay.
template <class F>
int asr_graph_run(AsrGraphCache* gc, const void* key, size_t key_bytes, hipStream_t stream, F&& launch) {
  if (!gc) return launch(stream);
  AsrGraphEntry* e = nullptr;
  for (auto& it : gc->entries)
    if (it.key.size() == key_bytes && memcmp(it.key.data(), key, key_bytes) == 0) { e = &it; break; }
  gc->clock++;
  if (!e) {
    if (gc->entries.size() >= gc->max_entries) {   // evict least recently used
      size_t victim = 0;
      for (size_t i = 1; i < gc->entries.size(); ++i)
        if (gc->entries[i].stamp < gc->entries[victim].stamp) victim = i;
      if (gc->entries[victim].exec) (void)hipGraphExecDestroy(gc->entries[victim].exec);
      gc->entries.erase(gc->entries.begin() + victim);
    }
    AsrGraphEntry ne;
    ne.key.assign((const unsigned char*)key, (const unsigned char*)key + key_bytes);
    ne.seen = 1;
    ne.stamp = gc->clock;
    gc->entries.push_back(ne);
    gc->eager++;
    return launch(stream);
  }
  e->stamp = gc->clock;
  if (e->exec) {
    gc->hits++;
    hipError_t err = hipGraphLaunch(e->exec, stream);
    return err == hipSuccess ? 0 : (int)err;
  }
  if (e->failed) { gc->eager++; return launch(stream); }
  // second sighting: capture
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    e->failed = true;
    (void)hipGetLastError();
    return launch(stream);
  }
  int rc = launch(stream);
  hipError_t err = hipStreamEndCapture(stream, &graph);
  if (rc != 0 || err != hipSuccess || !graph) {
    e->failed = true;
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    return launch(stream);   // nothing ran during the failed capture: do the work now
  }
  hipGraphExec_t exec = nullptr;
  err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (err != hipSuccess || !exec) {
    e->failed = true;
    (void)hipGetLastError();
    return launch(stream);
  }
  e->exec = exec;
  gc->captures++;
  err = hipGraphLaunch(exec, stream);
  return err == hipSuccess ? 0 : (int)err;
}
