// transport.hpp -- the exchanges of a coset-sharded proof: three transports behind two calls.
//   shard_allgather   equal blocks, recv = [world][bytes] (caps, quotient interpolants, opening sums, PoW minima, query rows,
//                     the witness column blocks of the host-witness entry)
//   shard_allgatherv  unequal blocks, IN PLACE (the coefficient blocks of the column-sharded inverse transforms, knob shard_intt)
// Transports: a device group of one process (hipMemcpyPeerAsync between the ranks' streams: PeerGroup), RCCL called by the
// library on the circuit's own stream (librccl.so.1 resolved with dlopen, never linked), a host callback (the gloo tests).
#pragma once
#include "circuit.hpp"
#include <rccl/rccl.h>  // types and prototypes only: resolved with dlopen, never linked
#include <condition_variable>
#include <mutex>
#include <vector>

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      p2::set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
      return P2GPU_E_DEVICE;                                                                   \
    }                                                                                          \
  } while (0)

namespace p2 {
// ---- RCCL, bound at run time -------------------------------------------------------------------
// The library has no link-time dependency on RCCL: the collectives of a sharded proof resolve
// librccl.so.1 when sharding is switched on -- the copy that is already mapped into the process when
// the host side runs torch.distributed (its bundled RCCL has the same SONAME), /opt/rocm's otherwise.
// Calls are stream-ordered on the circuit's own stream: no host synchronisation around a collective.
struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};
const RcclApi &rccl();
#define RCCL_TRY(expr)                                                                            \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess) {                                                                      \
      p2::set_err("%s failed: %s (%s:%d)", #expr, p2::rccl().GetErrorString(r_), __FILE__, __LINE__); \
      return P2GPU_E_DEVICE;                                                                      \
    }                                                                                             \
  } while (0)


// The ranks of ONE process (p2gpu_init with several device ids): a host rendezvous for the rank threads and two
// events per rank.  An all-gather is `world` peer copies per rank, each enqueued on the SENDING rank's own stream
// straight into the receiver's buffer (hipMemcpyPeerAsync: xGMI on a multi-GPU node, and every rank drives all its
// links at once -- SURVEY 8(e) step 2's "not a ring"); readiness travels as events, never as a host wait:
//   recv_free[p]  recorded by p before the exchange: everything p enqueued that still reads its receive buffer
//   sent[q]       recorded by q behind its copies: p's consumers wait for all of them
// The two host barriers only order the event RECORDS before the cross-stream WAITS that name them.
struct PeerGroup {
  int n = 0;
  std::vector<p2gpu_circuit *> cs;
  std::vector<hipEvent_t> recv_free, sent;
  std::vector<const void *> send_ptr;
  std::vector<void *> recv_ptr;
  std::vector<std::vector<uint8_t>> proof_scratch;  // ranks > 0 write their (identical) proof bytes here: kept across proofs
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  uint64_t gen = 0;
  bool aborted = false;
  bool barrier() {
    std::unique_lock<std::mutex> l(m);
    if (aborted) return false;
    const uint64_t g = gen;
    if (++waiting == n) {
      waiting = 0;
      gen++;
      cv.notify_all();
      return true;
    }
    cv.wait(l, [&] { return gen != g || aborted; });
    return gen != g;
  }
  void abort() {  // a rank left the proof with an error: nobody may wait for it
    std::lock_guard<std::mutex> l(m);
    aborted = true;
    cv.notify_all();
  }
  void reset() {
    std::lock_guard<std::mutex> l(m);
    aborted = false;
    waiting = 0;
  }
};

int shard_allgather(p2gpu_circuit *c, const void *send_dev, void *recv_dev, size_t bytes);
int shard_allgatherv(p2gpu_circuit *c, uint8_t *base, const size_t *off, const size_t *sz);
void intt_blocks(const uint32_t *dense, uint32_t nd, uint32_t G, uint32_t *lo, uint32_t *hi);
bool sharded(const p2gpu_circuit *c);
}  // namespace p2
