// loopback_transport.cpp -- TEST INFRASTRUCTURE, not part of the product library.
//
// A slamhip_shard_transport (include/slamhip.h) whose "ranks" are threads of ONE process, each with its own
// slamhip context on whatever GPU the process sees: the all-gather goes through a board in host memory, the
// point-to-point exchange is a device-to-device copy between the ranks' buffers.  RCCL admits one rank per device,
// so this is how slamhip_gmapping_step_sharded -- the protocol above the transport: one all-gather per step,
// status words, map migration on resampling -- runs with world > 1 on the single GPU of a test box
// (tests/test_gpu_shard.py), and on no GPU at all for the host-only parts (tests/test_sharded_filter.py).
//
// Build (tests/loopback.py does it):  g++ -std=c++17 -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
//                                     -I<repo>/include loopback_transport.cpp -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "slamhip.h"

namespace {

struct Posted {
  int from, to;
  const void *buf;
  size_t bytes;
};

struct Board {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0;
  // a reusable barrier
  int waiting = 0;
  long long generation = 0;
  std::vector<std::vector<char>> blocks;  // all-gather
  std::vector<Posted> posted;             // exchange
  int failed = 0;
  // every wait of this transport is bounded (include/slamhip.h: "an attached transport bounds its own waits"): a rank
  // that does not arrive within the deadline kills the group -- everybody waiting, and everybody who comes later,
  // gets SLAMHIP_ERR_TIMEOUT
  int timeout_ms = 600000;
  bool dead = false;
  bool barrier(std::unique_lock<std::mutex> &lk) {
    if (dead) return false;
    const long long gen = generation;
    if (++waiting == world) {
      waiting = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    if (!cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return generation != gen || dead; }) || dead) {
      dead = true;
      cv.notify_all();
      return false;
    }
    return true;
  }
};

std::mutex g_mu;
std::map<std::string, std::shared_ptr<Board>> g_boards;

struct Rank {
  std::shared_ptr<Board> board;
  int rank = 0;
};

int lb_allgather(void *user, const void *send, size_t block, void *recv) {
  Rank *r = static_cast<Rank *>(user);
  Board &b = *r->board;
  std::unique_lock<std::mutex> lk(b.mu);
  b.blocks[r->rank].assign(static_cast<const char *>(send), static_cast<const char *>(send) + block);
  if (!b.barrier(lk)) return SLAMHIP_ERR_TIMEOUT;  // every block is posted
  int rc = 0;
  char *out = static_cast<char *>(recv);
  for (int q = 0; q < b.world; ++q) {
    if (b.blocks[q].size() != block) rc = -1;  // the ranks disagree about the block size
    else std::memcpy(out + (size_t)q * block, b.blocks[q].data(), block);
  }
  if (!b.barrier(lk)) return SLAMHIP_ERR_TIMEOUT;  // every block is read: the board may be rewritten
  return rc;
}

int lb_exchange(void *user, int n_send, const slamhip_shard_msg *send, int n_recv, const slamhip_shard_msg *recv) {
  Rank *r = static_cast<Rank *>(user);
  Board &b = *r->board;
  std::unique_lock<std::mutex> lk(b.mu);
  for (int k = 0; k < n_send; ++k) b.posted.push_back(Posted{r->rank, send[k].peer, send[k].buf, send[k].bytes});
  if (!b.barrier(lk)) return SLAMHIP_ERR_TIMEOUT;  // everything that will be sent is on the board (in each sender's order)
  int rc = 0;
  std::vector<size_t> taken(b.world, 0);  // per sender: how many of its messages to me I have matched
  for (int k = 0; k < n_recv; ++k) {
    const int q = recv[k].peer;
    size_t seen = 0;
    const Posted *hit = nullptr;
    for (const Posted &p : b.posted) {
      if (p.from != q || p.to != r->rank) continue;
      if (seen++ == taken[q]) {
        hit = &p;
        break;
      }
    }
    if (!hit || hit->bytes != recv[k].bytes) {
      rc = -1;  // no matching send, or the two sides disagree about the size
      continue;
    }
    ++taken[q];
    if (hit->bytes) {
      lk.unlock();  // (the copy is synchronous: do not hold the board over it)
      const hipError_t e = hipMemcpy(recv[k].buf, hit->buf, hit->bytes, hipMemcpyDefault);
      lk.lock();
      if (e != hipSuccess) rc = -2;
    }
  }
  if (rc) b.failed = 1;
  if (!b.barrier(lk)) return SLAMHIP_ERR_TIMEOUT;  // every receive has copied: the senders' buffers are free again
  if (r->rank == 0) b.posted.clear();
  const int failed = b.failed;
  if (!b.barrier(lk)) return SLAMHIP_ERR_TIMEOUT;
  if (r->rank == 0) b.failed = 0;
  return rc ? rc : (failed ? -3 : 0);
}

void lb_destroy(void *user) {
  Rank *r = static_cast<Rank *>(user);
  r->board.reset();
  delete r;
  std::lock_guard<std::mutex> lk(g_mu);  // a group nobody belongs to any more is forgotten
  for (auto it = g_boards.begin(); it != g_boards.end();) it = it->second.use_count() == 1 ? g_boards.erase(it) : std::next(it);
}

}  // namespace

extern "C" int loopback_transport_create(const char *name, int rank, int world, slamhip_shard_transport *out) {
  if (!name || !out || world < 1 || rank < 0 || rank >= world) return -1;
  std::shared_ptr<Board> b;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto &slot = g_boards[name];
    if (!slot) {
      slot = std::make_shared<Board>();
      slot->world = world;
      slot->blocks.resize(world);
    }
    b = slot;
  }
  if (b->world != world) return -2;
  Rank *r = new Rank;
  r->board = b;
  r->rank = rank;
  out->user = r;
  out->allgather = lb_allgather;
  out->exchange = lb_exchange;
  out->destroy = lb_destroy;
  return 0;
}

// deadline of every wait of group `name` (default: ten minutes)
extern "C" int loopback_transport_set_timeout(const char *name, int ms) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_boards.find(name ? name : "");
  if (it == g_boards.end() || ms <= 0) return -1;
  std::lock_guard<std::mutex> lk2(it->second->mu);
  it->second->timeout_ms = ms;
  return 0;
}
