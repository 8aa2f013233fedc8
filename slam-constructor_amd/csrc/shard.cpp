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

#include <cstring>
#include <string>
#include <vector>

#include "slamhip_internal.h"

namespace slamhip {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
    SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  s->exchanges += 1;
  s->p2p_bytes += moved;
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
  const size_t block = (size_t)max_count * elem_bytes;
  if (block == 0) return SLAMHIP_OK;
  if (s->attached) {
    // the caller's transport moves equal host blocks; padding and unpadding happen here
    std::vector<char> snd(block, 0), rcv(block * (size_t)s->world);
    if (counts[s->rank] > 0) std::memcpy(snd.data(), local, (size_t)counts[s->rank] * elem_bytes);
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
  std::memcpy(s->h_send, local, mine);
  std::memset(s->h_send + mine, 0, block - mine);
  SLAMHIP_CHECK(hipMemcpyAsync(s->d_send, s->h_send, block, hipMemcpyHostToDevice, ctx->stream));
  ncclResult_t e = r->AllGather(s->d_send, s->d_recv, block, ncclUint8, s->comm, ctx->stream);
  if (e != ncclSuccess) return rccl_fail(e, "ncclAllGather");
  SLAMHIP_CHECK(hipMemcpyAsync(s->h_recv, s->d_recv, block * s->world, hipMemcpyDeviceToHost, ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(ctx->stream));
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
