// Multi-GPU exchange step of the DP path behind the C ABI: an RCCL all-gatherv of the per-task result records and CIGAR
// words (the reference has no counterpart -- its processes meet in files, sedef.sh:187-190,218-221).
//
// DP tasks are independent, so a batch is sharded over the GPUs of a node without any data-path collective; after the DP
// every GPU holds the records and CIGAR words of its shard, and one exchange gives every GPU every shard's results: an
// all-gather of the two counts and the two capacities per rank (ncclAllGather, 32 bytes a rank), a host read of that small table through pinned
// memory behind an event, and ONE group of point-to-point transfers on the exact sizes (ncclGroupStart ... ncclSend /
// ncclRecv ... ncclGroupEnd: RCCL has no native gatherv; on the fully connected xGMI mesh the world - 1 transfers of a rank
// are one hop each and run side by side).  Nothing padded travels.
//
// RCCL is loaded when the first communicator is made (dlopen: a process that uses one GPU never maps it); a copy of the
// library that the process has already loaded -- PyTorch brings its own -- is used rather than a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "sdf_ctx.h"

struct sdf_comm {
  int device = 0, world = 1, rank = 0;
  ncclComm_t comm = nullptr;
  uint64_t *h_counts = nullptr;  // pinned: [0..3] this rank's (tasks, CIGAR words, record capacity, CIGAR capacity); [4 ..] the table of all ranks
  uint64_t *d_counts = nullptr;  // device: the same layout
  hipEvent_t ev = nullptr;
  hipStream_t own = nullptr;
  bool selftest = false;  // SDF_COMM_SELFTEST=1 (one rank): its own ranges travel through ncclSend / ncclRecv to itself
  std::string err;
};

namespace {

std::string g_comm_err;

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so"}) {  // a copy the process already has (PyTorch's), first
      r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (r.lib) break;
    }
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (r.lib) break;
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.lib) {
      const char *e = dlerror();  // (one call: it returns the message AND clears it)
      r.err = std::string("cannot load librccl.so: ") + (e ? e : "?");
      return;
    }
#define SDF_SYM(field, sym)                                      \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym)); \
  if (!r.field) r.err = "librccl.so lacks " #sym;
    SDF_SYM(GetUniqueId, ncclGetUniqueId)
    SDF_SYM(CommInitRank, ncclCommInitRank)
    SDF_SYM(CommInitAll, ncclCommInitAll)
    SDF_SYM(CommDestroy, ncclCommDestroy)
    SDF_SYM(AllGather, ncclAllGather)
    SDF_SYM(Send, ncclSend)
    SDF_SYM(Recv, ncclRecv)
    SDF_SYM(GroupStart, ncclGroupStart)
    SDF_SYM(GroupEnd, ncclGroupEnd)
    SDF_SYM(GetErrorString, ncclGetErrorString)
#undef SDF_SYM
  });
  return r;
}

bool finish_comm(sdf_comm *c) {  // the small buffers of a communicator whose ncclComm_t exists
  if (hipSetDevice(c->device) != hipSuccess) return false;
  const size_t bytes = (size_t)(4 + 4 * c->world) * sizeof(uint64_t);
  if (hipHostMalloc((void **)&c->h_counts, bytes, hipHostMallocDefault) != hipSuccess) return false;
  if (hipMalloc((void **)&c->d_counts, bytes) != hipSuccess) return false;
  if (hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) return false;
  if (hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking) != hipSuccess) return false;
  const char *st = getenv("SDF_COMM_SELFTEST");
  c->selftest = st && st[0] == '1';
  return true;
}

}  // namespace

#define SDF_NCCL(call)                                                                             \
  do {                                                                                             \
    ncclResult_t r_ = (call);                                                                      \
    if (r_ != ncclSuccess) {                                                                       \
      c->err = std::string(#call) + ": " + (R.GetErrorString ? R.GetErrorString(r_) : "RCCL error"); \
      return SDF_ERR_HIP;                                                                          \
    }                                                                                              \
  } while (0)
#define SDF_CHIP(call)                                            \
  do {                                                            \
    hipError_t e_ = (call);                                       \
    if (e_ != hipSuccess) {                                       \
      c->err = std::string(#call) + ": " + hipGetErrorString(e_); \
      return SDF_ERR_HIP;                                         \
    }                                                             \
  } while (0)

extern "C" const char *sdf_comm_last_error(const sdf_comm *c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

extern "C" int sdf_comm_unique_id(void *id, size_t bytes) {
  Rccl &R = rccl();
  if (!R.err.empty()) {
    g_comm_err = R.err;
    return SDF_ERR_UNSUPPORTED;
  }
  if (!id || bytes < sizeof(ncclUniqueId)) {
    g_comm_err = "the id buffer holds 128 bytes";
    return SDF_ERR_INVALID;
  }
  ncclUniqueId u;
  const ncclResult_t r = R.GetUniqueId(&u);
  if (r != ncclSuccess) {
    g_comm_err = std::string("ncclGetUniqueId: ") + R.GetErrorString(r);
    return SDF_ERR_HIP;
  }
  memcpy(id, &u, sizeof(u));
  return SDF_OK;
}

extern "C" sdf_comm *sdf_comm_create(int device, int world, int rank, const void *id) {
  Rccl &R = rccl();
  if (!R.err.empty()) {
    g_comm_err = R.err;
    return nullptr;
  }
  if (!id || world < 1 || rank < 0 || rank >= world) {
    g_comm_err = "invalid arguments";
    return nullptr;
  }
  sdf_comm *c = new sdf_comm;
  c->device = device, c->world = world, c->rank = rank;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclResult_t r = ncclSuccess;
  if (hipSetDevice(device) != hipSuccess || (r = R.CommInitRank(&c->comm, world, u, rank)) != ncclSuccess || !finish_comm(c)) {
    g_comm_err = r != ncclSuccess ? std::string("ncclCommInitRank: ") + R.GetErrorString(r) : "cannot set up the communicator's buffers";
    (void)hipGetLastError();
    if (c->comm) R.CommDestroy(c->comm);
    delete c;
    return nullptr;
  }
  return c;
}

// One process, several devices (ncclCommInitAll): out[i] is the communicator of devices[i], rank i of n.  The collective
// calls of the ranks are then made by one thread per device (RCCL matches them across threads).
extern "C" int sdf_comm_create_all(const int *devices, int n, sdf_comm **out) {
  Rccl &R = rccl();
  if (!R.err.empty()) {
    g_comm_err = R.err;
    return SDF_ERR_UNSUPPORTED;
  }
  if (!devices || !out || n < 1) {
    g_comm_err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  std::vector<ncclComm_t> comms(n);
  const ncclResult_t r = R.CommInitAll(comms.data(), n, devices);
  if (r != ncclSuccess) {
    g_comm_err = std::string("ncclCommInitAll: ") + R.GetErrorString(r);
    return SDF_ERR_HIP;
  }
  for (int i = 0; i < n; ++i) {
    sdf_comm *c = new sdf_comm;
    c->device = devices[i], c->world = n, c->rank = i, c->comm = comms[i];
    out[i] = c;
    if (!finish_comm(c)) {
      g_comm_err = "cannot set up the communicator's buffers";
      return SDF_ERR_HIP;
    }
  }
  return SDF_OK;
}

extern "C" void sdf_comm_destroy(sdf_comm *c) {
  if (!c) return;
  Rccl &R = rccl();
  (void)hipSetDevice(c->device);
  if (c->own) {
    (void)hipStreamSynchronize(c->own);
    (void)hipStreamDestroy(c->own);
  }
  if (c->comm && R.CommDestroy) R.CommDestroy(c->comm);
  if (c->ev) (void)hipEventDestroy(c->ev);
  if (c->h_counts) (void)hipHostFree(c->h_counts);
  if (c->d_counts) (void)hipFree(c->d_counts);
  delete c;
}

extern "C" int sdf_comm_world(const sdf_comm *c) { return c ? c->world : 0; }
extern "C" int sdf_comm_rank(const sdf_comm *c) { return c ? c->rank : -1; }

// d_out[n_tasks], d_cig[cig_used]: this rank's results (HBM, the communicator's device).  d_all_out / d_all_cig receive
// every rank's, back to back in rank order; counts[2 r] = tasks, counts[2 r + 1] = CIGAR words of rank r (host, 2 * world
// entries).  Work is enqueued on `stream` (NULL: a stream of the communicator); the call returns when the counts are known
// and the transfers are enqueued -- the caller synchronises the stream before it reads the gathered buffers.
// SDF_ERR_CIGAR_OVERFLOW: a capacity is too small (counts[] is filled in: the sizes needed).
extern "C" int sdf_allgatherv_results(sdf_comm *c, const sdf_result *d_out, size_t n_tasks, const uint32_t *d_cig,
                                      size_t cig_used, sdf_result *d_all_out, size_t all_out_cap, uint32_t *d_all_cig,
                                      size_t all_cig_cap, uint64_t *counts, void *stream) {
  if (!c) return SDF_ERR_INVALID;
  Rccl &R = rccl();
  c->err.clear();
  if (!counts || (n_tasks && !d_out) || (cig_used && !d_cig)) {
    c->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  SDF_CHIP(hipSetDevice(c->device));
  hipStream_t st = stream ? (hipStream_t)stream : c->own;
  const int W = c->world, me = c->rank;
  // ---- the counts AND the capacities: 32 bytes a rank.  Whether the gathered buffers are large enough is decided from
  // this shared table, so that every rank takes the same branch: a rank that returned on a local check would leave its
  // peers waiting in ncclRecv for ever. ----
  c->h_counts[0] = n_tasks;
  c->h_counts[1] = cig_used;
  c->h_counts[2] = d_all_out ? all_out_cap : 0;
  c->h_counts[3] = d_all_cig ? all_cig_cap : 0;
  SDF_CHIP(hipMemcpyAsync(c->d_counts, c->h_counts, 4 * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  SDF_NCCL(R.AllGather(c->d_counts, c->d_counts + 4, 4, ncclUint64, c->comm, st));
  SDF_CHIP(hipMemcpyAsync(c->h_counts + 4, c->d_counts + 4, (size_t)4 * W * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  SDF_CHIP(hipEventRecord(c->ev, st));
  SDF_CHIP(hipEventSynchronize(c->ev));  // (this copy alone, not the device)
  std::vector<uint64_t> rec_off(W + 1, 0), cig_off(W + 1, 0);
  uint64_t min_out_cap = ~0ull, min_cig_cap = ~0ull;
  for (int r = 0; r < W; ++r) {
    counts[2 * r] = c->h_counts[4 + 4 * r];
    counts[2 * r + 1] = c->h_counts[5 + 4 * r];
    rec_off[r + 1] = rec_off[r] + counts[2 * r];
    cig_off[r + 1] = cig_off[r] + counts[2 * r + 1];
    min_out_cap = std::min<uint64_t>(min_out_cap, c->h_counts[6 + 4 * r]);
    min_cig_cap = std::min<uint64_t>(min_cig_cap, c->h_counts[7 + 4 * r]);
  }
  if (rec_off[W] > min_out_cap || cig_off[W] > min_cig_cap) {  // the same answer on every rank
    c->err = (rec_off[W] > (d_all_out ? all_out_cap : 0) || cig_off[W] > (d_all_cig ? all_cig_cap : 0))
                 ? "the gathered buffers are too small"
                 : "the gathered buffers of another rank are too small";
    return SDF_ERR_CIGAR_OVERFLOW;
  }
  // ---- this rank's part: a local copy (or, in the one-rank self test, a send to itself) ----
  const bool self = c->selftest && W == 1;
  if (!self) {
    if (n_tasks)
      SDF_CHIP(hipMemcpyAsync(d_all_out + rec_off[me], d_out, n_tasks * sizeof(sdf_result), hipMemcpyDeviceToDevice, st));
    if (cig_used)
      SDF_CHIP(hipMemcpyAsync(d_all_cig + cig_off[me], d_cig, cig_used * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  }
  // ---- ONE group of point-to-point transfers: this rank's two ranges to every peer, every peer's into their places.
  // Every rank derives the same list from the same table; empty ranges are skipped on both sides. ----
  if (W > 1 || self) {
    SDF_NCCL(R.GroupStart());
    ncclResult_t bad = ncclSuccess;  // an error inside the group: close the group before returning, or the thread's
    const char *what = "";           // next RCCL call would still be inside it
    auto in_group = [&](ncclResult_t r_, const char *call) {
      if (r_ != ncclSuccess && bad == ncclSuccess) bad = r_, what = call;
      return bad == ncclSuccess;
    };
    for (int r = 0; r < W && bad == ncclSuccess; ++r) {
      if (r == me && !self) continue;
      if (n_tasks && !in_group(R.Send(d_out, n_tasks * sizeof(sdf_result), ncclUint8, r, c->comm, st), "ncclSend(records)")) break;
      if (cig_used && !in_group(R.Send(d_cig, cig_used, ncclUint32, r, c->comm, st), "ncclSend(cigar)")) break;
      if (counts[2 * r] &&
          !in_group(R.Recv(d_all_out + rec_off[r], counts[2 * r] * sizeof(sdf_result), ncclUint8, r, c->comm, st), "ncclRecv(records)"))
        break;
      if (counts[2 * r + 1] &&
          !in_group(R.Recv(d_all_cig + cig_off[r], counts[2 * r + 1], ncclUint32, r, c->comm, st), "ncclRecv(cigar)"))
        break;
    }
    const ncclResult_t end = R.GroupEnd();
    if (bad != ncclSuccess || end != ncclSuccess) {
      c->err = std::string(bad != ncclSuccess ? what : "ncclGroupEnd") + ": " +
               (R.GetErrorString ? R.GetErrorString(bad != ncclSuccess ? bad : end) : "RCCL error");
      return SDF_ERR_HIP;
    }
  }
  if (!stream) SDF_CHIP(hipStreamSynchronize(st));
  return SDF_OK;
}

#undef SDF_NCCL
#undef SDF_CHIP
