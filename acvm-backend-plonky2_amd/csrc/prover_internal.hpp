// prover_internal.hpp -- what prover.hip (the proof) and handle.hip (the circuit handle, the stage-level operators) share.
#pragma once
#include "circuit.hpp"
#include "transport.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace p2 {
// P2GPU_TRACE=1: synchronise after every phase and report progress on stderr (debugging aid)
bool trace_on();
double now_ms();
#define TRACE(c, label)                                                                     \
  do {                                                                                      \
    if (p2::trace_on()) {                                                                   \
      hipError_t e_ = hipStreamSynchronize((c)->stream);                                    \
      fprintf(stderr, "[p2gpu] %s: %s\n", label, e_ == hipSuccess ? "ok" : hipGetErrorString(e_)); \
      fflush(stderr);                                                                       \
    }                                                                                       \
  } while (0)

// ---- prover.hip ----
void flush_kstats(p2gpu_circuit *c);                 // read the pending per-launch event pairs into c->kstats
int pin_exhausted();
void tree_layout(Batch &b, uint32_t cosets, size_t m0, size_t cap_per);
int tree_alloc(Batch &b, uint32_t C, size_t m0, size_t cap_per);
int wait_stream(p2gpu_circuit *c);
int tree_build(p2gpu_circuit *c, Batch &b, size_t m0, uint32_t levels_done = 0);
int batch_alloc(p2gpu_circuit *c, Batch &b, uint32_t cols);
int batch_commit_from_values(p2gpu_circuit *c, Batch &b, const gl_t *vals_dev);
int batch_commit_from_coeffs(p2gpu_circuit *c, Batch &b);
const gl_t *hprc(const p2gpu_circuit *c);
// ---- handle.hip ----
int ensure_device();
extern int g_device;
extern std::vector<int> g_devices, g_peer_access;
void circuit_release(p2gpu_circuit *c);

// run f(rank handle, rank) on one host thread per rank of a device group (rank 0 on the caller's thread); a rank
// that fails releases the others from their rendezvous.  Returns the first failing rank's code with its message.
template <class F>
inline int group_run(p2gpu_circuit *c, F f) {
  const int n = 1 + (int)c->group.size();
  std::vector<int> rcs(n, 0);
  std::vector<std::string> errs(n);
  c->peer->reset();
  auto work = [&](int q) {
    p2gpu_circuit *m = q ? c->group[q - 1] : c;
    int rc;
    try {
      rc = f(m, q);
    } catch (const std::exception &e) {
      set_err("internal error: %s", e.what());
      rc = P2GPU_E_DEVICE;
    }
    if (rc) {
      errs[q] = p2gpu_last_error();
      c->peer->abort();
    }
    rcs[q] = rc;
  };
  std::vector<std::thread> th;
  for (int q = 1; q < n; q++) th.emplace_back(work, q);
  work(0);
  for (auto &t : th) t.join();
  (void)hipSetDevice(c->device);
  // prefer the code of a rank that failed on its own over "another rank failed"
  int first = -1;
  for (int q = 0; q < n; q++)
    if (rcs[q] && (first < 0 || (errs[first].find("another rank") != std::string::npos && errs[q].find("another rank") == std::string::npos))) first = q;
  if (first < 0) return P2GPU_OK;
  set_err("%s%s", errs[first].c_str(), n > 1 ? (" (device group rank " + std::to_string(first) + ")").c_str() : "");
  return rcs[first];
}

}  // namespace p2
