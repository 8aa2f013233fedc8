// shard.cpp -- particles sharded over the GPUs of one node: the collective of the GMapping step on RCCL
// (xGMI) behind the C-ABI, so that a C++ host can shard without Python.
//
// What is distributed (paths relative to the reference root): the particle loop of
// ParticleFilter / GmappingParticleFilter::handle_sensor_data (src/core/particle_filter.h:108-112,
// src/slams/gmapping/gmapping_particle_filter.h:45-77); what every rank needs back is the full vector of
// raw weights in particle order, because normalize_weights and UniformResamling::resample
// (particle_filter.h:34-66) add them up in that order and the resampling indices have to stay bit-exact.
// Hence ONE all-gather per step (n_total doubles); an all-reduce of two sums would be smaller and would
// change the order of the additions.  When a resampling happens the particle records are all-gathered too
// (5 KB each, dominated by the particle's mt19937).
//
// librccl.so is opened with dlopen at the first use: libslamhip.so itself links only the HIP runtime.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "slamhip_internal.h"

namespace slamhip {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // (optional)
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl *rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
      r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
      r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
      r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
      r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
      r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
      r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
      if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString || !r.Send ||
          !r.Recv || !r.GroupStart || !r.GroupEnd) {
        dlclose(r.lib);
        r.lib = nullptr;
      }
    }
  }
  return r.lib ? &r : nullptr;
}

struct ShardState {
  ncclComm_t comm = nullptr;             // the built-in transport: an RCCL communicator ...
  slamhip_shard_transport ext{};         // ... or the caller's own (slamhip_shard_attach)
  bool attached = false;
  int rank = 0, world = 1;
  char *d_send = nullptr, *d_recv = nullptr;  // device staging of the padded blocks
  char *h_send = nullptr, *h_recv = nullptr;  // pinned mirrors
  size_t cap = 0;                             // bytes per rank the buffers hold
  long long collectives = 0, bytes = 0;
  long long exchanges = 0, p2p_bytes = 0;
  int timeout_ms = 30000;  // deadline of every wait on a collective (slamhip_shard_set_timeout)
  bool broken = false;     // a collective timed out: the communicator is being aborted
  std::thread aborter;     // ncclCommAbort waits for the communicator's work on the stream: off the caller's thread
  // the attached transport's staging (padded blocks), kept between calls
  std::vector<char> ext_snd, ext_rcv;
};

static int rccl_fail(ncclResult_t e, const char *what) {
  Rccl *r = rccl();
  set_error(std::string(what) + ": " + (r ? r->GetErrorString(e) : "RCCL not loaded"));
  return SLAMHIP_ERR_HIP;
}

static int no_rccl() {
  set_error("librccl.so could not be loaded: sharding needs RCCL");
  return SLAMHIP_ERR_UNSUPPORTED;
}

static int invalid_arg(const char *msg) {
  set_error(msg);
  return SLAMHIP_ERR_INVALID;
}

static int broken_group() {
  set_error("the shard group is broken (a collective timed out earlier): slamhip_shard_destroy, then join a new group");
  return SLAMHIP_ERR_STATE;
}

// Waits for what has been queued on the stream -- the collective and the copies around it -- but not for ever: a
// peer that dies inside a collective never completes it.  On the deadline the communicator is aborted (so that the
// kernel RCCL left on the stream ends) and the group is marked broken.
static int bounded_wait(slamhip_ctx *ctx, ShardState *s, const char *what) {
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    const hipError_t q = hipStreamQuery(ctx->stream);
    if (q == hipSuccess) return SLAMHIP_OK;
    if (q != hipErrorNotReady) return hip_fail(q, what);
    if ((++spins & 63u) == 0u) {
      const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (ms > s->timeout_ms) break;
      if (ms > 2) std::this_thread::sleep_for(std::chrono::microseconds(50));  // (a healthy collective is through long before)
    }
  }
  s->broken = true;
  Rccl *r = rccl();
  if (s->comm && r && r->CommAbort) {
    // (ncclCommAbort itself waits for what the communicator has on the stream: the caller gets its answer NOW, the
    // abort runs beside it and is joined by slamhip_shard_destroy)
    ncclComm_t comm = s->comm;
    s->comm = nullptr;
    const int device = ctx->device;
    s->aborter = std::thread([r, comm, device] {
      (void)hipSetDevice(device);
      r->CommAbort(comm);
    });
  }
  set_error(std::string(what) + ": not complete after " + std::to_string(s->timeout_ms) +
            " ms -- a peer of the shard group stopped responding; the communicator was aborted");
  return SLAMHIP_ERR_TIMEOUT;
}

static void free_buffers(ShardState *s) {
  if (s->d_send) hipFree(s->d_send);
  if (s->d_recv) hipFree(s->d_recv);
  if (s->h_send) hipHostFree(s->h_send);
  if (s->h_recv) hipHostFree(s->h_recv);
  s->d_send = s->d_recv = s->h_send = s->h_recv = nullptr;
  s->cap = 0;
}

void shard_release(slamhip_ctx *ctx) {
  auto *s = static_cast<ShardState *>(ctx->shard);
  if (!s) return;
  if (s->comm) {
    Rccl *r = rccl();
    if (r) r->CommDestroy(s->comm);
  }
  if (s->aborter.joinable()) s->aborter.join();  // (a broken group's communicator goes with ncclCommAbort)
  if (s->attached && s->ext.destroy) s->ext.destroy(s->ext.user);
  free_buffers(s);
  delete s;
  ctx->shard = nullptr;
}

}  // namespace slamhip

using namespace slamhip;

extern "C" {

int slamhip_shard_unique_id(void *id_out) {
  if (!id_out) return invalid_arg("null id");
  Rccl *r = rccl();
  if (!r) return no_rccl();
  static_assert(sizeof(ncclUniqueId) == SLAMHIP_SHARD_ID_BYTES, "unique id size");
  ncclUniqueId id;
  ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) return rccl_fail(e, "ncclGetUniqueId");
  std::memcpy(id_out, &id, sizeof(id));
  return SLAMHIP_OK;
}

int slamhip_shard_init(slamhip_ctx *ctx, int rank, int world, const void *id) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return invalid_arg("bad shard geometry");
  if (ctx->shard) return invalid_arg("the context already belongs to a shard group");
  Rccl *r = rccl();
  if (!r) return no_rccl();
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  auto *s = new ShardState;
  s->rank = rank;
  s->world = world;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclResult_t e = r->CommInitRank(&s->comm, world, uid, rank);
  if (e != ncclSuccess) {
    delete s;
    return rccl_fail(e, "ncclCommInitRank");
  }
  ctx->shard = s;
  return SLAMHIP_OK;
}

int slamhip_shard_attach(slamhip_ctx *ctx, int rank, int world, const slamhip_shard_transport *t) {
  if (!ctx || !t || world < 1 || rank < 0 || rank >= world) return invalid_arg("bad shard geometry");
  if (!t->allgather || !t->exchange) return invalid_arg("a shard transport needs allgather and exchange");
  if (ctx->shard) return invalid_arg("the context already belongs to a shard group");
  auto *s = new ShardState;
  s->rank = rank;
  s->world = world;
  s->ext = *t;
  s->attached = true;
  ctx->shard = s;
  return SLAMHIP_OK;
}

int slamhip_shard_exchange(slamhip_ctx *ctx, int n_send, const slamhip_shard_msg *send, int n_recv,
                           const slamhip_shard_msg *recv) {
  if (!ctx || n_send < 0 || n_recv < 0 || (n_send > 0 && !send) || (n_recv > 0 && !recv))
    return invalid_arg("bad exchange arguments");
  auto *s = static_cast<ShardState *>(ctx->shard);
  if (!s) {
    set_error("slamhip_shard_init has not been called on this context");
    return SLAMHIP_ERR_STATE;
  }
  for (int k = 0; k < n_send; ++k)
    if (send[k].peer < 0 || send[k].peer >= s->world || (send[k].bytes && !send[k].buf)) return invalid_arg("bad send message");
  for (int k = 0; k < n_recv; ++k)
    if (recv[k].peer < 0 || recv[k].peer >= s->world || (recv[k].bytes && !recv[k].buf)) return invalid_arg("bad receive message");
  if (s->broken) return broken_group();
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  long long moved = 0;
  for (int k = 0; k < n_send; ++k) moved += (long long)send[k].bytes;
  if (s->attached) {
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));  // what filled the send buffers has run
    const int trc = s->ext.exchange(s->ext.user, n_send, send, n_recv, recv);
    if (trc) {
      set_error("the attached shard transport failed in its exchange");
      return trc < 0 ? trc : SLAMHIP_ERR_STATE;
    }
  } else {
    Rccl *r = rccl();
    if (!r) return no_rccl();
    // one group: every send and receive of this rank is posted before any of them blocks (point-to-point xGMI
    // links: a pair's messages travel on the link between the two GPUs)
    ncclResult_t e = r->GroupStart();
    if (e != ncclSuccess) return rccl_fail(e, "ncclGroupStart");
    for (int k = 0; k < n_send && e == ncclSuccess; ++k)
      if (send[k].bytes) e = r->Send(send[k].buf, send[k].bytes, ncclUint8, send[k].peer, s->comm, ctx->stream);
    for (int k = 0; k < n_recv && e == ncclSuccess; ++k)
      if (recv[k].bytes) e = r->Recv(recv[k].buf, recv[k].bytes, ncclUint8, recv[k].peer, s->comm, ctx->stream);
    const ncclResult_t e2 = r->GroupEnd();
    if (e != ncclSuccess) return rccl_fail(e, "ncclSend / ncclRecv");
    if (e2 != ncclSuccess) return rccl_fail(e2, "ncclGroupEnd");
    const int wrc = bounded_wait(ctx, s, "slamhip_shard_exchange");
    if (wrc) return wrc;
  }
  s->exchanges += 1;
  s->p2p_bytes += moved;
  return SLAMHIP_OK;
}

#ifdef SLAMHIP_TESTING
// testing aid, not part of include/slamhip.h: a kernel that keeps the context's stream busy for `ms` (at most 5 s)
int slamhip_debug_stall(slamhip_ctx *ctx, int ms) {
  if (!ctx || ms < 0 || ms > 5000) return invalid_arg("bad stall");
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  SLAMHIP_CHECK(launch_stall(ms, ctx->stream));
  return SLAMHIP_OK;
}
#endif  // SLAMHIP_TESTING

int slamhip_shard_set_timeout(slamhip_ctx *ctx, int ms) {
  if (!ctx) return invalid_arg("null ctx");
  auto *s = static_cast<ShardState *>(ctx->shard);
  if (!s) {
    set_error("slamhip_shard_init has not been called on this context");
    return SLAMHIP_ERR_STATE;
  }
  s->timeout_ms = ms > 0 ? ms : 30000;
  return SLAMHIP_OK;
}

int slamhip_shard_p2p_stats(slamhip_ctx *ctx, long long *exchanges, long long *bytes_sent) {
  if (!ctx) return invalid_arg("null ctx");
  auto *s = static_cast<ShardState *>(ctx->shard);
  if (exchanges) *exchanges = s ? s->exchanges : 0;
  if (bytes_sent) *bytes_sent = s ? s->p2p_bytes : 0;
  return SLAMHIP_OK;
}

int slamhip_shard_destroy(slamhip_ctx *ctx) {
  if (!ctx) return invalid_arg("null ctx");
  if (ctx->shard) {
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    shard_release(ctx);
  }
  return SLAMHIP_OK;
}

int slamhip_shard_info(slamhip_ctx *ctx, int *rank, int *world) {
  if (!ctx) return invalid_arg("null ctx");
  auto *s = static_cast<ShardState *>(ctx->shard);
  if (rank) *rank = s ? s->rank : 0;
  if (world) *world = s ? s->world : 1;
  return SLAMHIP_OK;
}

int slamhip_shard_stats(slamhip_ctx *ctx, long long *collectives, long long *bytes) {
  if (!ctx) return invalid_arg("null ctx");
  auto *s = static_cast<ShardState *>(ctx->shard);
  if (collectives) *collectives = s ? s->collectives : 0;
  if (bytes) *bytes = s ? s->bytes : 0;
  return SLAMHIP_OK;
}

int slamhip_shard_allgather(slamhip_ctx *ctx, const void *local, const int *counts, int elem_bytes,
                            void *all_out) {
  if (!ctx || !counts || !all_out || elem_bytes <= 0) return invalid_arg("bad all-gather arguments");
  auto *s = static_cast<ShardState *>(ctx->shard);
  if (!s) {
    set_error("slamhip_shard_init has not been called on this context");
    return SLAMHIP_ERR_STATE;
  }
  int max_count = 0;
  for (int q = 0; q < s->world; ++q) {
    if (counts[q] < 0) return invalid_arg("negative block size");
    max_count = std::max(max_count, counts[q]);
  }
  if (counts[s->rank] > 0 && !local) return invalid_arg("null local block");
  if (s->broken) return broken_group();
  const size_t block = (size_t)max_count * elem_bytes;
  if (block == 0) return SLAMHIP_OK;
  if (s->attached) {
    // the caller's transport moves equal host blocks; padding and unpadding happen here (staging kept between calls)
    std::vector<char> &snd = s->ext_snd, &rcv = s->ext_rcv;
    if (snd.size() < block) snd.resize(block);
    if (rcv.size() < block * (size_t)s->world) rcv.resize(block * (size_t)s->world);
    const size_t mine_b = (size_t)counts[s->rank] * elem_bytes;
    if (mine_b > 0) std::memcpy(snd.data(), local, mine_b);
    std::memset(snd.data() + mine_b, 0, block - mine_b);
    const int trc = s->ext.allgather(s->ext.user, snd.data(), block, rcv.data());
    if (trc) {
      set_error("the attached shard transport failed in its all-gather");
      return trc < 0 ? trc : SLAMHIP_ERR_STATE;
    }
    char *out = static_cast<char *>(all_out);
    for (int q = 0; q < s->world; ++q) {
      const size_t nb = (size_t)counts[q] * elem_bytes;
      std::memcpy(out, rcv.data() + (size_t)q * block, nb);
      out += nb;
    }
    s->collectives += 1;
    s->bytes += (long long)(block * s->world);
    return SLAMHIP_OK;
  }
  Rccl *r = rccl();
  if (!r) return no_rccl();
  SLAMHIP_CHECK(hipSetDevice(ctx->device));
  if (block > s->cap) {
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
    free_buffers(s);
    size_t cap = 4096;
    while (cap < block) cap *= 2;
    SLAMHIP_CHECK(hipMalloc(&s->d_send, cap));
    SLAMHIP_CHECK(hipMalloc(&s->d_recv, cap * s->world));
    SLAMHIP_CHECK(hipHostMalloc(&s->h_send, cap, hipHostMallocDefault));
    SLAMHIP_CHECK(hipHostMalloc(&s->h_recv, cap * s->world, hipHostMallocDefault));
    s->cap = cap;
  }
  const size_t mine = (size_t)counts[s->rank] * elem_bytes;
  if (mine > 0) std::memcpy(s->h_send, local, mine);  // (local may be null when this rank contributes nothing)
  std::memset(s->h_send + mine, 0, block - mine);
  SLAMHIP_CHECK(hipMemcpyAsync(s->d_send, s->h_send, block, hipMemcpyHostToDevice, ctx->stream));
  ncclResult_t e = r->AllGather(s->d_send, s->d_recv, block, ncclUint8, s->comm, ctx->stream);
  if (e != ncclSuccess) return rccl_fail(e, "ncclAllGather");
  SLAMHIP_CHECK(hipMemcpyAsync(s->h_recv, s->d_recv, block * s->world, hipMemcpyDeviceToHost, ctx->stream));
  {
    const int wrc = bounded_wait(ctx, s, "slamhip_shard_allgather");
    if (wrc) return wrc;
  }
  char *out = static_cast<char *>(all_out);
  for (int q = 0; q < s->world; ++q) {
    const size_t nb = (size_t)counts[q] * elem_bytes;
    std::memcpy(out, s->h_recv + (size_t)q * block, nb);
    out += nb;
  }
  s->collectives += 1;
  s->bytes += (long long)(block * s->world);
  return SLAMHIP_OK;
}

}  // extern "C"
