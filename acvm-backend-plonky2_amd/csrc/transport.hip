// transport.hip -- see transport.hpp.  Replaces nothing of the reference (it is single-process, CPU): this is the multi-GPU half
// of the drop-in for `circuit_data.prove` (plonky2-backend/src/actions/prove_action.rs:96), SURVEY.md 8(e).
#include "transport.hpp"
#include <dlfcn.h>
#include <algorithm>
#include <cstdlib>

namespace p2 {
const RcclApi &rccl() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the one the process already uses, if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
      api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
      api.Send = (decltype(api.Send))dlsym(h, "ncclSend");
      api.Recv = (decltype(api.Recv))dlsym(h, "ncclRecv");
      api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
      api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
      api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
    }
  }
  return api;
}

namespace {

int peer_allgather(p2gpu_circuit *c, const void *send_dev, void *recv_dev, size_t bytes) {
  PeerGroup &g = *c->peer;
  const int q = c->shard_rank;
  auto failed = [] {
    set_err("another rank of the device group failed");
    return P2GPU_E_DEVICE;
  };
  g.send_ptr[q] = send_dev;
  g.recv_ptr[q] = recv_dev;
  HIP_TRY(hipEventRecord(g.recv_free[q], c->stream));
  if (!g.barrier()) return failed();
  for (int p = 0; p < g.n; p++) {
    if (p != q) HIP_TRY(hipStreamWaitEvent(c->stream, g.recv_free[p], 0));
    uint8_t *dst = (uint8_t *)g.recv_ptr[p] + (size_t)q * bytes;
    if ((const void *)dst != send_dev)  // (an in-place all-gather already has the rank's own block where it belongs)
      HIP_TRY(hipMemcpyPeerAsync(dst, g.cs[p]->device, send_dev, c->device, bytes, c->stream));
  }
  HIP_TRY(hipEventRecord(g.sent[q], c->stream));
  if (!g.barrier()) return failed();
  for (int p = 0; p < g.n; p++)
    if (p != q) HIP_TRY(hipStreamWaitEvent(c->stream, g.sent[p], 0));
  return 0;
}

// all-gather over the ranks of a sharded proof, device buffers, recv = [world][bytes].  With an RCCL
// communicator (p2gpu_circuit_set_shard_rccl) it is an ncclAllGather enqueued on the circuit's stream;
// with a host callback (p2gpu_circuit_set_shard: gloo in the CPU-side tests) the stream is drained and
// the callback returns when the data is in place.
// (`profile` = 2 brackets every exchange with HIP events on the rank's stream like a kernel launch: pseudo-kernels
// "exchange[...]" in p2gpu_kernel_stats, by payload class -- the per-exchange microseconds of a sharded proof)
static const char *exchange_name(size_t bytes) {
  return bytes <= 4096 ? "exchange[<=4KB: caps, PoW minima]" : bytes < ((size_t)1 << 20) ? "exchange[<1MB: opening sums, query rows]"
                                                                                        : "exchange[>=1MB: witness blocks, quotient interpolants]";
}
static int shard_allgather_impl(p2gpu_circuit *c, const void *send_dev, void *recv_dev, size_t bytes) {
  if (c->peer) return peer_allgather(c, send_dev, recv_dev, bytes);
  if (c->rccl_comm) {
    const RcclApi &r = rccl();
    // Large payloads (the witness column blocks: 31 MB per rank at 2^20 rows, the quotient interpolants) go as one
    // grouped send / receive per peer -- on xGMI's point-to-point links all seven transfers of a rank run at once,
    // where a ring all-gather is bound by one link; the small ones (caps, PoW minima, query rows)
    // stay with ncclAllGather's latency-optimised protocols.  P2GPU_RCCL_P2P_BYTES moves the threshold (0: never).
    static const size_t p2p_min = [] {
      const char *e = getenv("P2GPU_RCCL_P2P_BYTES");
      return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)1 << 20);
    }();
    if (p2p_min && bytes >= p2p_min && r.Send && r.Recv && r.GroupStart && r.GroupEnd) {
      const int world = c->shard_world, q = c->shard_rank;
      uint8_t *mine = (uint8_t *)recv_dev + (size_t)q * bytes;
      if ((const void *)mine != send_dev) HIP_TRY(hipMemcpyAsync(mine, send_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
      RCCL_TRY(r.GroupStart());
      // a failure inside the group still closes it: an open NCCL group would swallow every later call on this thread
      ncclResult_t bad = ncclSuccess;
      const char *what = "";
      for (int p = 0; p < world && bad == ncclSuccess; p++) {
        if (p == q) continue;
        bad = r.Send(send_dev, bytes, ncclUint8, p, (ncclComm_t)c->rccl_comm, c->stream);
        what = "ncclSend";
        if (bad != ncclSuccess) break;
        bad = r.Recv((uint8_t *)recv_dev + (size_t)p * bytes, bytes, ncclUint8, p, (ncclComm_t)c->rccl_comm, c->stream);
        what = "ncclRecv";
      }
      const ncclResult_t ge = r.GroupEnd();
      if (bad != ncclSuccess) {
        set_err("%s failed inside the grouped exchange: %s", what, r.GetErrorString(bad));
        return P2GPU_E_DEVICE;
      }
      RCCL_TRY(ge);
      return 0;
    }
    RCCL_TRY(r.AllGather(send_dev, recv_dev, bytes, ncclUint8, (ncclComm_t)c->rccl_comm, c->stream));
    return 0;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  int rc = c->shard_fn(c->shard_ctx, (uint64_t)(uintptr_t)send_dev, (uint64_t)(uintptr_t)recv_dev, (uint64_t)bytes);
  if (rc) {
    set_err("all-gather callback failed (%d)", rc);
    return P2GPU_E_DEVICE;
  }
  return 0;
}
}  // namespace
int shard_allgather(p2gpu_circuit *c, const void *send_dev, void *recv_dev, size_t bytes) {
  ProfScope ps(exchange_name(bytes), (double)bytes * (double)(c->shard_world > 0 ? c->shard_world : 1));
  return shard_allgather_impl(c, send_dev, recv_dev, bytes);
}
// All-gather of blocks of DIFFERENT sizes, in place: on every rank the block of rank p lives at base + off[p] (sz[p] bytes, the
// same off / sz on all ranks; a rank sends its own block and receives the others where they belong).  The exchange of the
// column-sharded inverse transforms: each rank's block of coefficient columns goes straight into the
// peers' coefficient buffers -- no staging, no unpack pass.
//   device group: one hipMemcpyPeerAsync per peer on the sender's stream (all links of a rank at once);
//   RCCL: one grouped ncclSend / ncclRecv pair per peer, sizes per peer;
//   host callback (fixed-size all-gather, the gloo tests): pieces staged through xchg_recv.
static int shard_allgatherv_impl(p2gpu_circuit *c, uint8_t *base, const size_t *off, const size_t *sz) {
  const int world = c->shard_world, q = c->shard_rank;
  if (c->peer) {
    PeerGroup &g = *c->peer;
    auto failed = [] {
      set_err("another rank of the device group failed");
      return P2GPU_E_DEVICE;
    };
    g.recv_ptr[q] = base;
    HIP_TRY(hipEventRecord(g.recv_free[q], c->stream));
    if (!g.barrier()) return failed();
    for (int p = 0; p < g.n; p++) {
      if (p == q) continue;
      HIP_TRY(hipStreamWaitEvent(c->stream, g.recv_free[p], 0));
      if (sz[q]) HIP_TRY(hipMemcpyPeerAsync((uint8_t *)g.recv_ptr[p] + off[q], g.cs[p]->device, base + off[q], c->device, sz[q], c->stream));
    }
    HIP_TRY(hipEventRecord(g.sent[q], c->stream));
    if (!g.barrier()) return failed();
    for (int p = 0; p < g.n; p++)
      if (p != q) HIP_TRY(hipStreamWaitEvent(c->stream, g.sent[p], 0));
    return 0;
  }
  if (c->rccl_comm) {
    const RcclApi &r = rccl();
    if (world == 1) {
      // one rank: its block is in place.  With "shard_exercise" the grouped send / receive pair itself still runs once -- to this
      // rank, into the exchange scratch -- so that the symbols resolve and the grouped form executes on a one-GPU box
      if (c->shard_exercise && sz[0] && r.Send && r.Recv && r.GroupStart && r.GroupEnd) {
        const size_t nb = std::min(sz[0], std::min((size_t)1 << 20, c->xchg_recv.count * sizeof(gl_t)));
        RCCL_TRY(r.GroupStart());
        ncclResult_t a = r.Send(base + off[0], nb, ncclUint8, 0, (ncclComm_t)c->rccl_comm, c->stream);
        ncclResult_t b = a == ncclSuccess ? r.Recv(c->xchg_recv.p, nb, ncclUint8, 0, (ncclComm_t)c->rccl_comm, c->stream) : a;
        const ncclResult_t ge = r.GroupEnd();
        if (a != ncclSuccess || b != ncclSuccess) {
          set_err("grouped self send / receive failed: %s", r.GetErrorString(a != ncclSuccess ? a : b));
          return P2GPU_E_DEVICE;
        }
        RCCL_TRY(ge);
      }
      return 0;
    }
    if (!(r.Send && r.Recv && r.GroupStart && r.GroupEnd)) {
      set_err("librccl.so.1 has no ncclSend / ncclRecv: shard_intt needs them");
      return P2GPU_E_DEVICE;
    }
    RCCL_TRY(r.GroupStart());
    ncclResult_t bad = ncclSuccess;
    const char *what = "";
    for (int p = 0; p < world && bad == ncclSuccess; p++) {
      if (p == q) continue;
      if (sz[q]) {
        bad = r.Send(base + off[q], sz[q], ncclUint8, p, (ncclComm_t)c->rccl_comm, c->stream);
        what = "ncclSend";
        if (bad != ncclSuccess) break;
      }
      if (sz[p]) {
        bad = r.Recv(base + off[p], sz[p], ncclUint8, p, (ncclComm_t)c->rccl_comm, c->stream);
        what = "ncclRecv";
      }
    }
    const ncclResult_t ge = r.GroupEnd();  // (a failure inside the group still closes it)
    if (bad != ncclSuccess) {
      set_err("%s failed inside the grouped exchange: %s", what, r.GetErrorString(bad));
      return P2GPU_E_DEVICE;
    }
    RCCL_TRY(ge);
    return 0;
  }
  // host callback: equal-sized pieces, [send piece][world x piece] in xchg_recv
  size_t mx = 0;
  for (int p = 0; p < world; p++) mx = std::max(mx, sz[p]);
  const size_t cap = (c->xchg_recv.count * sizeof(gl_t) / (size_t)(world + 1)) & ~(size_t)63;
  if (!cap) {
    set_err("internal: exchange staging buffer too small");
    return P2GPU_E_DEVICE;
  }
  uint8_t *stage = (uint8_t *)c->xchg_recv.p, *recv = stage + cap;
  for (size_t done = 0; done < mx; done += cap) {
    const size_t piece = std::min(cap, mx - done);
    if (sz[q] > done) HIP_TRY(hipMemcpyAsync(stage, base + off[q] + done, std::min(piece, sz[q] - done), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = c->shard_fn(c->shard_ctx, (uint64_t)(uintptr_t)stage, (uint64_t)(uintptr_t)recv, (uint64_t)piece);
    if (rc) {
      set_err("all-gather callback failed (%d)", rc);
      return P2GPU_E_DEVICE;
    }
    for (int p = 0; p < world; p++)
      if (p != q && sz[p] > done)
        HIP_TRY(hipMemcpyAsync(base + off[p] + done, recv + (size_t)p * piece, std::min(piece, sz[p] - done), hipMemcpyDeviceToDevice, c->stream));
  }
  return 0;
}
int shard_allgatherv(p2gpu_circuit *c, uint8_t *base, const size_t *off, const size_t *sz) {
  size_t total = 0, mx = 0;
  for (int p = 0; p < c->shard_world; p++) {
    total += sz[p];
    mx = std::max(mx, sz[p]);
  }
  ProfScope ps(exchange_name(mx), (double)total);
  return shard_allgatherv_impl(c, base, off, sz);
}
// Blocks of the column-sharded inverse transform: the dense columns (sorted list `dense`, nd of them) are dealt out in G
// contiguous runs of the list, sizes differing by at most one; rank p's BLOCK is the column range from the first to the last
// column of its run (structured columns in between travel with it: they hold what the receiver's own fill writes anyway).
// parallel.intt_blocks restates this for the exchange plan.
void intt_blocks(const uint32_t *dense, uint32_t nd, uint32_t G, uint32_t *lo, uint32_t *hi) {
  for (uint32_t p = 0; p < G; p++) {
    const uint32_t s = (uint32_t)((uint64_t)p * nd / G), e = (uint32_t)((uint64_t)(p + 1) * nd / G);
    lo[p] = e > s ? dense[s] : 0;
    hi[p] = e > s ? dense[e - 1] + 1 : 0;
  }
}
// does this proof go through the exchange steps?  (world 1 + "shard_exercise": the same code with one
// rank, which is how the RCCL plumbing is exercised on a single-GPU box)
bool sharded(const p2gpu_circuit *c) { return c->shard_world > 1 || (c->shard_exercise && (c->rccl_comm || c->shard_fn || c->peer)); }

}  // namespace p2
