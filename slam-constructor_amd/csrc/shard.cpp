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

#include <condition_variable>
#include <cstring>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
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
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
      if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        dlclose(r.lib);
        r.lib = nullptr;
      }
    }
  }
  return r.lib ? &r : nullptr;
}

// An in-process group: the "ranks" are threads of ONE process (each with its own context, all on whatever GPUs the
// process sees), the all-gather goes through host memory behind a mutex.  It exists so that
// slamhip_gmapping_step_sharded -- the protocol above the collective -- can be run with world > 1 where only one
// GPU is available (tests/test_gpu_shard.py); RCCL itself admits one rank per device.
struct LoopbackBoard {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, arrived = 0, left = 0;
  long long generation = 0;
  std::vector<std::vector<char>> blocks;
};
static std::mutex g_boards_mu;
static std::map<std::string, std::shared_ptr<LoopbackBoard>> g_boards;
static constexpr char kLoopbackTag[] = "SLAMHIP-LOOPBACK:";

struct ShardState {
  ncclComm_t comm = nullptr;
  std::shared_ptr<LoopbackBoard> board;  // non-null: an in-process group instead of an RCCL communicator
  int rank = 0, world = 1;
  char *d_send = nullptr, *d_recv = nullptr;  // device staging of the padded blocks
  char *h_send = nullptr, *h_recv = nullptr;  // pinned mirrors
  size_t cap = 0;                             // bytes per rank the buffers hold
  long long collectives = 0, bytes = 0;
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
  if (s->board) {
    s->board.reset();
    std::lock_guard<std::mutex> lk(g_boards_mu);  // a group nobody belongs to any more is forgotten
    for (auto it = g_boards.begin(); it != g_boards.end();) it = it->second.use_count() == 1 ? g_boards.erase(it) : std::next(it);
  }
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
  if (std::memcmp(id, kLoopbackTag, sizeof(kLoopbackTag) - 1) == 0) {
    // in-process group: the rest of the id names it
    const std::string name(static_cast<const char *>(id), strnlen(static_cast<const char *>(id), SLAMHIP_SHARD_ID_BYTES));
    std::shared_ptr<LoopbackBoard> b;
    {
      std::lock_guard<std::mutex> lk(g_boards_mu);
      auto &slot = g_boards[name];
      if (!slot) {
        slot = std::make_shared<LoopbackBoard>();
        slot->world = world;
        slot->blocks.resize(world);
      }
      b = slot;
    }
    if (b->world != world) return invalid_arg("the in-process group was created with another size");
    auto *s = new ShardState;
    s->rank = rank;
    s->world = world;
    s->board = b;
    ctx->shard = s;
    return SLAMHIP_OK;
  }
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
  if (s->board) {
    // every rank posts its block, waits for the others, copies all of them out, and the last one to leave
    // opens the board for the next collective
    LoopbackBoard &b = *s->board;
    std::unique_lock<std::mutex> lk(b.mu);
    b.cv.wait(lk, [&] { return b.left == 0; });  // the previous collective has been read by everybody
    const long long gen = b.generation;
    b.blocks[s->rank].assign(static_cast<const char *>(local), static_cast<const char *>(local) + (size_t)counts[s->rank] * elem_bytes);
    if (++b.arrived == b.world) {
      b.arrived = 0;
      b.left = b.world;
      ++b.generation;
      b.cv.notify_all();
    } else {
      b.cv.wait(lk, [&] { return b.generation != gen; });
    }
    char *out = static_cast<char *>(all_out);
    for (int q = 0; q < s->world; ++q) {
      const size_t nb = (size_t)counts[q] * elem_bytes;
      if (b.blocks[q].size() != nb) {
        --b.left;
        b.cv.notify_all();
        return invalid_arg("the ranks of an in-process group disagree about the block sizes");
      }
      std::memcpy(out, b.blocks[q].data(), nb);
      out += nb;
    }
    if (--b.left == 0) b.cv.notify_all();
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
