// tile_pool.hip -- see tile_pool.h.  Kernels: tile copy (copy-on-write), table patch / assign,
// dense window <-> tiles.

#include <algorithm>
#include <cstring>

#include "tile_pool.h"

namespace slamhip {

namespace {

constexpr int kChunksPerTile = 64;  // workgroups per tile copy: 256 cells each

// one (src, dst) pair per 64 workgroups: payload 512 KB + counters 256 KB, 16-byte accesses
__global__ __launch_bounds__(256) void k_tile_copy(double *pool, double *aux, unsigned *state, unsigned *pend, const int *pairs,
                                                   int n_pairs) {
  const int pair = blockIdx.x / kChunksPerTile, chunk = blockIdx.x % kChunksPerTile;
  if (pair >= n_pairs) return;
  if (chunk < kTileStateWords / 256)
    state[(size_t)pairs[2 * pair + 1] * kTileStateWords + chunk * 256 + threadIdx.x] =
        state[(size_t)pairs[2 * pair] * kTileStateWords + chunk * 256 + threadIdx.x];
  const size_t src = (size_t)pairs[2 * pair] * kTileCells, dst = (size_t)pairs[2 * pair + 1] * kTileCells;
  const size_t cell = (size_t)chunk * 256 + threadIdx.x;
  const double4 *ps = reinterpret_cast<const double4 *>(pool) + src;
  double4 *pd = reinterpret_cast<double4 *>(pool) + dst;
  pd[cell] = ps[cell];
  const double2 *as = reinterpret_cast<const double2 *>(aux) + src;
  double2 *ad = reinterpret_cast<double2 *>(aux) + dst;
  // (the clone starts with its pending observations folded into its counters: the source stays as it is)
  const double2 c = as[cell];
  ad[cell] = make_double2(c.x, c.y + (double)pend[src + cell]);
  pend[dst + cell] = 0u;
}

// pending observations of one tile into its counters (before the tile's counters are read raw: exports)
__global__ __launch_bounds__(256) void k_tile_fold(double *aux, unsigned *pend, int tile) {
  const size_t cell = (size_t)tile * kTileCells + (size_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned p = pend[cell];
  if (p) {
    aux[2 * cell + 1] += (double)p;
    pend[cell] = 0u;
  }
}

__global__ void k_table_patch(int *tables, int stride, const int *patches, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  tables[(size_t)patches[3 * i] * stride + patches[3 * i + 1]] = patches[3 * i + 2];
}

__global__ void k_table_assign(int *dst, const int *src, const int *src_of_new, int stride) {
  const int slot = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < stride) dst[(size_t)slot * stride + i] = src[(size_t)src_of_new[slot] * stride + i];
}

__global__ void k_table_fill(int *tables, size_t n, int v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tables[i] = v;
}

__global__ void k_tile_fill_unknown(double *pool, double *aux, int tile, double u0, double u1, double u2, double u3) {
  const size_t cell = (size_t)tile * kTileCells + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  reinterpret_cast<double4 *>(pool)[cell] = make_double4(u0, u1, u2, u3);
  reinterpret_cast<double2 *>(aux)[cell] = make_double2(0.0, 0.0);
}

// tiles (tx0.., ty0..) x (ntx, nty) of the pool <- dense window placed at virtual (vx0, vy0)
__global__ __launch_bounds__(256) void k_tiles_from_dense(double *pool, double *aux, unsigned *pend, const int *tile_ids, int tx0,
                                                          int ty0, int ntx, const double *dense,
                                                          const double *dense_aux, int pitch, int w, int h, int vx0,
                                                          int vy0, double u0, double u1, double u2, double u3) {
  const int t = blockIdx.x / kChunksPerTile, chunk = blockIdx.x % kChunksPerTile;
  const int tx = tx0 + t % ntx, ty = ty0 + t / ntx;
  const int in_tile = chunk * 256 + threadIdx.x;
  const int vx = tx * kTileSide + (in_tile & kTileMask), vy = ty * kTileSide + (in_tile >> kTileShift);
  const int dx = vx - vx0, dy = vy - vy0;
  double4 v = make_double4(u0, u1, u2, u3);
  double2 c = make_double2(0.0, 0.0);
  if ((unsigned)dx < (unsigned)w && (unsigned)dy < (unsigned)h) {
    v = reinterpret_cast<const double4 *>(dense)[(size_t)dy * pitch + dx];
    if (dense_aux) c = reinterpret_cast<const double2 *>(dense_aux)[(size_t)dy * pitch + dx];
  }
  const size_t at = (size_t)tile_ids[t] * kTileCells + in_tile;
  reinterpret_cast<double4 *>(pool)[at] = v;
  reinterpret_cast<double2 *>(aux)[at] = c;
  pend[at] = 0u;
}

// the settle states (tile_pool.h) of tiles first .. first + n - 1 from their payloads: a thread per word, 4 workgroups per tile
__global__ __launch_bounds__(256) void k_tile_state_build(const double *pool, unsigned *state, int first, long long unknown_bits,
                                                          int fresh_ok) {
  const int tile = first + blockIdx.x / (kTileStateWords / 256);
  const int word = (blockIdx.x % (kTileStateWords / 256)) * 256 + threadIdx.x;
  const double *base = pool + ((size_t)tile * kTileCells + (size_t)word * 16) * 4;
  unsigned w = 0u;
#pragma unroll
  for (int c = 0; c < 16; ++c) w |= mu_settle_class(base[4 * c], unknown_bits, fresh_ok) << (2 * c);
  state[(size_t)tile * kTileStateWords + word] = w;
}
__global__ __launch_bounds__(256) void k_tile_state_check(const double *pool, const unsigned *state, int first,
                                                          long long unknown_bits, int fresh_ok, unsigned long long *count) {
  const int tile = first + blockIdx.x / (kTileStateWords / 256);
  const int word = (blockIdx.x % (kTileStateWords / 256)) * 256 + threadIdx.x;
  const double *base = pool + ((size_t)tile * kTileCells + (size_t)word * 16) * 4;
  unsigned w = 0u;
#pragma unroll
  for (int c = 0; c < 16; ++c) w |= mu_settle_class(base[4 * c], unknown_bits, fresh_ok) << (2 * c);
  if (state[(size_t)tile * kTileStateWords + word] != w) atomicAdd(count, 1ull);
}

// in-tile neighbourhood masks of tiles first .. first + n - 1 (tile_pool.h): 64 workgroups per tile
__device__ __forceinline__ unsigned tile_nbr_mask_of(const double *tile, double th, int lx, int ly) {
  unsigned m9 = 0u;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int x = lx + i / 3 - 1, y = ly + i % 3 - 1;
    if ((unsigned)x < (unsigned)kTileSide && (unsigned)y < (unsigned)kTileSide && !(tile[4 * ((size_t)y * kTileSide + x)] < th))
      m9 |= 1u << i;
  }
  return m9;
}
__global__ __launch_bounds__(256) void k_tile_nbr_build(double *pool, int first, double th) {
  const int tile = first + blockIdx.x / kChunksPerTile, chunk = blockIdx.x % kChunksPerTile;
  const int in_tile = chunk * 256 + threadIdx.x;
  double *base = pool + (size_t)tile * kTileCells * 4;
  reinterpret_cast<unsigned *>(base + 4 * (size_t)in_tile + 3)[0] =
      tile_nbr_mask_of(base, th, in_tile & kTileMask, in_tile >> kTileShift);
}
__global__ __launch_bounds__(256) void k_tile_nbr_check(const double *pool, int first, double th, unsigned long long *count) {
  const int tile = first + blockIdx.x / kChunksPerTile, chunk = blockIdx.x % kChunksPerTile;
  const int in_tile = chunk * 256 + threadIdx.x;
  const double *base = pool + (size_t)tile * kTileCells * 4;
  const unsigned have = reinterpret_cast<const unsigned *>(base + 4 * (size_t)in_tile + 3)[0];
  if (have != tile_nbr_mask_of(base, th, in_tile & kTileMask, in_tile >> kTileShift)) atomicAdd(count, 1ull);
}

__global__ void k_tiles_to_dense(const double *pool, const double *aux, const unsigned *pend, const int *table, int tiles_x, int tiles_y,
                                 int vx0, int vy0, int w, int h, double *payload3, double *aux2, double u0,
                                 double u1, double u2) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int vx = vx0 + x, vy = vy0 + y;
  double p0 = u0, p1 = u1, p2 = u2, a0 = 0, a1 = 0;
  if ((unsigned)vx < (unsigned)(tiles_x * kTileSide) && (unsigned)vy < (unsigned)(tiles_y * kTileSide)) {
    const int tile = table[(vy >> kTileShift) * tiles_x + (vx >> kTileShift)];
    const size_t at = (size_t)tile * kTileCells + ((size_t)(vy & kTileMask) << kTileShift) + (vx & kTileMask);
    const double4 v = reinterpret_cast<const double4 *>(pool)[at];
    const double2 c = reinterpret_cast<const double2 *>(aux)[at];
    p0 = v.x; p1 = v.y; p2 = v.z; a0 = c.x; a1 = c.y + (double)pend[at];
  }
  const size_t o = (size_t)y * w + x;
  if (payload3) {
    payload3[3 * o] = p0;
    payload3[3 * o + 1] = p1;
    payload3[3 * o + 2] = p2;
  }
  if (aux2) {
    aux2[2 * o] = a0;
    aux2[2 * o + 1] = a1;
  }
}

int tp_fail(const char *msg, int code = SLAMHIP_ERR_INVALID) {
  set_error(msg);
  return code;
}

// settle states of tiles first .. first + n - 1 from their payloads, queued on the context's stream
int tile_state_build(TilePool *tp, int first, int n) {
  if (n <= 0) return SLAMHIP_OK;
  long long ub;
  std::memcpy(&ub, &tp->unknown[0], 8);
  hipLaunchKernelGGL(k_tile_state_build, dim3(n * (kTileStateWords / 256)), dim3(256), 0, tp->ctx->stream, tp->d_pool, tp->d_state,
                     first, ub, tp->unknown[0] < 0.0 ? 1 : 0);
  SLAMHIP_CHECK(hipGetLastError());
  return SLAMHIP_OK;
}

int alloc_tile(TilePool *tp, int *out) {
  if (!tp->free_list.empty()) {
    *out = tp->free_list.back();
    tp->free_list.pop_back();
    return SLAMHIP_OK;
  }
  if (tp->next_unused >= tp->capacity)
    return tp_fail("tile pool exhausted: create the particle maps with a larger capacity", SLAMHIP_ERR_STATE);
  *out = tp->next_unused++;
  return SLAMHIP_OK;
}

int ensure_staging(TilePool *tp, int pairs, int patches) {
  constexpr unsigned kPinned = hipHostMallocMapped | hipHostMallocCoherent;
  if (pairs > tp->cap_pairs) {
    int cap = std::max(1024, tp->cap_pairs);
    while (cap < pairs) cap *= 2;
    int *n = nullptr;
    SLAMHIP_CHECK(hipHostMalloc(&n, sizeof(int) * 2 * cap, kPinned));
    if (tp->h_pairs) {
      std::memcpy(n, tp->h_pairs, sizeof(int) * 2 * tp->n_pairs);
      SLAMHIP_CHECK(hipStreamSynchronize(tp->ctx->stream));
      hipHostFree(tp->h_pairs);
    }
    tp->h_pairs = n;
    tp->cap_pairs = cap;
  }
  if (patches > tp->cap_patches) {
    int cap = std::max(1024, tp->cap_patches);
    while (cap < patches) cap *= 2;
    int *n = nullptr;
    SLAMHIP_CHECK(hipHostMalloc(&n, sizeof(int) * 3 * cap, kPinned));
    if (tp->h_patches) {
      std::memcpy(n, tp->h_patches, sizeof(int) * 3 * tp->n_patches);
      SLAMHIP_CHECK(hipStreamSynchronize(tp->ctx->stream));
      hipHostFree(tp->h_patches);
    }
    tp->h_patches = n;
    tp->cap_patches = cap;
  }
  return SLAMHIP_OK;
}

}  // namespace

int tile_pool_create(slamhip_ctx *ctx, int n_slots, int tiles_x, int tiles_y, double scale, const double unknown[4],
                     int capacity, TilePool **out) {
  if (!ctx || !out || n_slots <= 0 || tiles_x <= 0 || tiles_y <= 0 || capacity < 2) return tp_fail("bad tile pool shape");
  if ((long long)tiles_x * tiles_y > (1 << 20)) return tp_fail("tile table too large");
  auto *tp = new TilePool;
  tp->ctx = ctx;
  tp->n_slots = n_slots;
  tp->tiles_x = tiles_x;
  tp->tiles_y = tiles_y;
  tp->origin_x = tiles_x * kTileSide / 2;
  tp->origin_y = tiles_y * kTileSide / 2;
  tp->scale = scale;
  for (int k = 0; k < 4; ++k) tp->unknown[k] = unknown[k];
  tp->capacity = capacity;
  const size_t cells = (size_t)capacity * kTileCells;
  const size_t tab = (size_t)n_slots * tiles_x * tiles_y;
  hipError_t e = hipMalloc(&tp->d_pool, cells * 4 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&tp->d_aux, cells * 2 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&tp->d_state, (size_t)capacity * kTileStateWords * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&tp->d_pend, cells * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemsetAsync(tp->d_pend, 0, cells * sizeof(unsigned), ctx->stream);
  if (e == hipSuccess) e = hipMalloc(&tp->d_tables[0], tab * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&tp->d_tables[1], tab * sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc(&tp->h_assign, sizeof(int) * n_slots, hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) {
    tile_pool_destroy(tp);
    return hip_fail(e, "tile pool allocation");
  }
  tp->h_tables.assign(tab, 0);
  tp->refcnt.assign(capacity, 0);
  tp->refcnt[0] = 1 << 30;
  tp->ancestor_of.assign(capacity, -1);
  hipLaunchKernelGGL(k_table_fill, dim3((unsigned)((tab + 255) / 256)), dim3(256), 0, ctx->stream, tp->d_tables[0], tab, 0);
  hipLaunchKernelGGL(k_tile_fill_unknown, dim3(kTileCells / 256), dim3(256), 0, ctx->stream, tp->d_pool, tp->d_aux, 0,
                     unknown[0], unknown[1], unknown[2], unknown[3]);
  SLAMHIP_CHECK(hipGetLastError());
  {
    const int rcs = tile_state_build(tp, 0, 1);
    if (rcs) {
      tile_pool_destroy(tp);
      return rcs;
    }
  }
  *out = tp;
  return SLAMHIP_OK;
}

void tile_pool_destroy(TilePool *tp) {
  if (!tp) return;
  if (tp->ctx) hipStreamSynchronize(tp->ctx->stream);
  if (tp->d_pool) hipFree(tp->d_pool);
  if (tp->d_aux) hipFree(tp->d_aux);
  if (tp->d_state) hipFree(tp->d_state);
  if (tp->d_pend) hipFree(tp->d_pend);
  if (tp->d_tables[0]) hipFree(tp->d_tables[0]);
  if (tp->d_tables[1]) hipFree(tp->d_tables[1]);
  if (tp->h_pairs) hipHostFree(tp->h_pairs);
  if (tp->h_patches) hipHostFree(tp->h_patches);
  if (tp->h_assign) hipHostFree(tp->h_assign);
  delete tp;
}

int tile_pool_nbr_masks(TilePool *tp, double th) {
  if (tp->nbr_ok && tp->nbr_th == th) return SLAMHIP_OK;
  tp->nbr_ok = false;
  // the writers keep the masks by flipping bits when a cell changes sides; two writers do not look: a fresh tile is
  // filled with the unknown cell (mask 0: nothing full) and a never-observed cell's first free observation takes its
  // mean from the unknown one's to +0 (k_mu_classify) -- neither may be full
  if (!(th > 0.0) || !(tp->unknown[0] < th)) return SLAMHIP_OK;
  if (tp->next_unused > 0) {
    hipLaunchKernelGGL(k_tile_nbr_build, dim3(tp->next_unused * kChunksPerTile), dim3(256), 0, tp->ctx->stream, tp->d_pool, 0, th);
    SLAMHIP_CHECK(hipGetLastError());
    SLAMHIP_CHECK(hipStreamSynchronize(tp->ctx->stream));
  }
  tp->nbr_th = th;
  tp->nbr_ok = true;
  return SLAMHIP_OK;
}

int tile_pool_state_check(TilePool *tp, long long *mismatches) {
  *mismatches = 0;
  if (tp->next_unused <= 0) return SLAMHIP_OK;
  unsigned long long *d_count = nullptr, h_count = 0;
  long long ub;
  std::memcpy(&ub, &tp->unknown[0], 8);
  SLAMHIP_CHECK(hipMalloc(&d_count, sizeof(unsigned long long)));
  SLAMHIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), tp->ctx->stream));
  hipLaunchKernelGGL(k_tile_state_check, dim3(tp->next_unused * (kTileStateWords / 256)), dim3(256), 0, tp->ctx->stream, tp->d_pool,
                     tp->d_state, 0, ub, tp->unknown[0] < 0.0 ? 1 : 0, d_count);
  SLAMHIP_CHECK(hipMemcpyAsync(&h_count, d_count, sizeof(h_count), hipMemcpyDeviceToHost, tp->ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(tp->ctx->stream));
  hipFree(d_count);
  *mismatches = (long long)h_count;
  return SLAMHIP_OK;
}

int tile_pool_nbr_check(TilePool *tp, long long *mismatches) {
  *mismatches = 0;
  if (!tp->nbr_ok || tp->next_unused <= 0) return SLAMHIP_OK;
  unsigned long long *d_count = nullptr, h_count = 0;
  SLAMHIP_CHECK(hipMalloc(&d_count, sizeof(unsigned long long)));
  SLAMHIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), tp->ctx->stream));
  hipLaunchKernelGGL(k_tile_nbr_check, dim3(tp->next_unused * kChunksPerTile), dim3(256), 0, tp->ctx->stream, tp->d_pool, 0,
                     tp->nbr_th, d_count);
  SLAMHIP_CHECK(hipMemcpyAsync(&h_count, d_count, sizeof(h_count), hipMemcpyDeviceToHost, tp->ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(tp->ctx->stream));
  hipFree(d_count);
  *mismatches = (long long)h_count;
  return SLAMHIP_OK;
}

int tile_pool_init_from_dense(TilePool *tp, const DeviceMap &m) {
  tp->nbr_ok = false;  // (the dense window's pads hold ITS masks or none: the next scorer call derives the tiles')
  if (m.cell_model != SLAMHIP_CELL_GMAPPING) return tp_fail("particle maps need a SLAMHIP_CELL_GMAPPING window");
  if (m.aux_stride != 0 && m.aux_stride != 2) return tp_fail("unexpected counter layout");
  // virtual position of the dense window: same external coordinates
  const int vx0 = tp->origin_x - m.origin_x, vy0 = tp->origin_y - m.origin_y;
  if (vx0 < 0 || vy0 < 0 || vx0 + m.width > tp->width() || vy0 + m.height > tp->height())
    return tp_fail("the dense window does not fit the tile extent");
  const int tx0 = vx0 >> kTileShift, ty0 = vy0 >> kTileShift;
  const int tx1 = (vx0 + m.width - 1) >> kTileShift, ty1 = (vy0 + m.height - 1) >> kTileShift;
  const int ntx = tx1 - tx0 + 1, nty = ty1 - ty0 + 1, nt = ntx * nty;
  std::vector<int> ids(nt);
  for (int k = 0; k < nt; ++k) {
    int rc = alloc_tile(tp, &ids[k]);
    if (rc) return rc;
    tp->refcnt[ids[k]] = tp->n_slots;
    tp->ancestor.push_back(ids[k]);
    tp->ancestor_of[ids[k]] = (int)tp->ancestor.size() - 1;
  }
  int *d_ids = nullptr;
  SLAMHIP_CHECK(hipMalloc(&d_ids, sizeof(int) * nt));
  SLAMHIP_CHECK(hipMemcpyAsync(d_ids, ids.data(), sizeof(int) * nt, hipMemcpyHostToDevice, tp->ctx->stream));
  hipLaunchKernelGGL(k_tiles_from_dense, dim3(nt * kChunksPerTile), dim3(256), 0, tp->ctx->stream, tp->d_pool, tp->d_aux,
                     tp->d_pend, d_ids, tx0, ty0, ntx, m.d_payload, m.aux_stride == 2 ? m.d_aux : nullptr, m.pitch, m.width,
                     m.height, vx0, vy0, tp->unknown[0], tp->unknown[1], tp->unknown[2], tp->unknown[3]);
  {  // the settle states of the new tiles (runs of consecutive ids in one launch each)
    int k0 = 0;
    for (int k = 1; k <= nt; ++k)
      if (k == nt || ids[k] != ids[k - 1] + 1) {
        const int rcs = tile_state_build(tp, ids[k0], k - k0);
        if (rcs) return rcs;
        k0 = k;
      }
  }
  const int stride = tp->table_stride();
  for (int s = 0; s < tp->n_slots; ++s)
    for (int k = 0; k < nt; ++k)
      tp->h_tables[(size_t)s * stride + (ty0 + k / ntx) * tp->tiles_x + tx0 + k % ntx] = ids[k];
  SLAMHIP_CHECK(hipMemcpyAsync(tp->d_tables[tp->cur], tp->h_tables.data(), sizeof(int) * tp->h_tables.size(),
                               hipMemcpyHostToDevice, tp->ctx->stream));
  SLAMHIP_CHECK(hipStreamSynchronize(tp->ctx->stream));
  hipFree(d_ids);
  return SLAMHIP_OK;
}

// UnboundedLazyTiledGridMap::ensure_inside (lazy_tiled_grid_map.h:128-187): the extent grows by whole tiles on
// the sides that a write reaches beyond; new area is the shared unknown tile.  Every slot's table is
// re-laid out (host mirror, one upload); tile ids, refcounts and the pool itself do not move.  Growth is
// generous -- at least a quarter of the current extent on a side that grows -- so that a robot driving
// off does not re-lay the tables every scan.
int tile_pool_grow(TilePool *tp, int x0, int y0, int x1, int y1) {
  auto tiles_for = [](int cells) { return (cells + kTileSide - 1) >> kTileShift; };
  int add_l = x0 < 0 ? tiles_for(-x0) : 0, add_r = x1 >= tp->width() ? tiles_for(x1 - tp->width() + 1) : 0;
  int add_t = y0 < 0 ? tiles_for(-y0) : 0, add_b = y1 >= tp->height() ? tiles_for(y1 - tp->height() + 1) : 0;
  if (!(add_l | add_r | add_t | add_b)) return SLAMHIP_OK;
  if (tp->n_pairs || tp->n_patches) return tp_fail("tile pool growth with queued copies", SLAMHIP_ERR_STATE);
  if (add_l) add_l = std::max(add_l, (tp->tiles_x + 3) / 4);
  if (add_r) add_r = std::max(add_r, (tp->tiles_x + 3) / 4);
  if (add_t) add_t = std::max(add_t, (tp->tiles_y + 3) / 4);
  if (add_b) add_b = std::max(add_b, (tp->tiles_y + 3) / 4);
  const int ntx = tp->tiles_x + add_l + add_r, nty = tp->tiles_y + add_t + add_b;
  if ((long long)ntx * nty > (1 << 20)) return tp_fail("tile table too large: the particle maps cannot grow that far");
  const size_t tab = (size_t)tp->n_slots * ntx * nty;
  std::vector<int> nt(tab, 0);
  for (int s = 0; s < tp->n_slots; ++s) {
    const int *from = tp->h_tables.data() + (size_t)s * tp->table_stride();
    int *to = nt.data() + (size_t)s * ntx * nty;
    for (int ty = 0; ty < tp->tiles_y; ++ty)
      std::memcpy(to + (size_t)(ty + add_t) * ntx + add_l, from + (size_t)ty * tp->tiles_x, sizeof(int) * tp->tiles_x);
  }
  hipStream_t st = tp->ctx->stream;
  SLAMHIP_CHECK(hipStreamSynchronize(st));
  if (tp->ctx->stream_b) SLAMHIP_CHECK(hipStreamSynchronize(tp->ctx->stream_b));
  int *fresh[2] = {nullptr, nullptr};
  hipError_t e = hipMalloc(&fresh[0], tab * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&fresh[1], tab * sizeof(int));
  if (e == hipSuccess) e = hipMemcpyAsync(fresh[tp->cur], nt.data(), tab * sizeof(int), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    if (fresh[0]) hipFree(fresh[0]);
    if (fresh[1]) hipFree(fresh[1]);
    return hip_fail(e, "tile table growth");
  }
  hipFree(tp->d_tables[0]);
  hipFree(tp->d_tables[1]);
  tp->d_tables[0] = fresh[0];
  tp->d_tables[1] = fresh[1];
  tp->h_tables.swap(nt);
  tp->tiles_x = ntx;
  tp->tiles_y = nty;
  tp->origin_x += add_l * kTileSide;
  tp->origin_y += add_t * kTileSide;
  tp->growths += 1;
  return SLAMHIP_OK;
}

int tile_pool_make_private(TilePool *tp, int slot, int x0, int y0, int x1, int y1) {
  if (slot < 0 || slot >= tp->n_slots) return tp_fail("bad slot");
  x0 = std::max(x0, 0);
  y0 = std::max(y0, 0);
  x1 = std::min(x1, tp->width() - 1);
  y1 = std::min(y1, tp->height() - 1);
  if (x1 < x0 || y1 < y0) return SLAMHIP_OK;
  const int tx0 = x0 >> kTileShift, ty0 = y0 >> kTileShift, tx1 = x1 >> kTileShift, ty1 = y1 >> kTileShift;
  int *row = tp->h_tables.data() + (size_t)slot * tp->table_stride();
  int rc = ensure_staging(tp, tp->n_pairs + (tx1 - tx0 + 1) * (ty1 - ty0 + 1),
                          tp->n_patches + (tx1 - tx0 + 1) * (ty1 - ty0 + 1));
  if (rc) return rc;
  for (int ty = ty0; ty <= ty1; ++ty) {
    for (int tx = tx0; tx <= tx1; ++tx) {
      const int idx = ty * tp->tiles_x + tx;
      const int old = row[idx];
      // already private -- ancestor tiles never are: they stay intact for maps that arrive by ordinal
      if (old != 0 && tp->refcnt[old] == 1 && tp->ancestor_of[old] < 0) continue;
      int fresh = 0;
      rc = alloc_tile(tp, &fresh);
      if (rc) return rc;
      tp->refcnt[fresh] = 1;
      if (old != 0) tp->refcnt[old] -= 1;
      row[idx] = fresh;
      tp->h_pairs[2 * tp->n_pairs] = old;
      tp->h_pairs[2 * tp->n_pairs + 1] = fresh;
      tp->n_pairs += 1;
      tp->h_patches[3 * tp->n_patches] = slot;
      tp->h_patches[3 * tp->n_patches + 1] = idx;
      tp->h_patches[3 * tp->n_patches + 2] = fresh;
      tp->n_patches += 1;
    }
  }
  return SLAMHIP_OK;
}

int tile_pool_flush(TilePool *tp) {
  hipStream_t st = tp->ctx->stream;
  if (tp->n_pairs) {
    hipLaunchKernelGGL(k_tile_copy, dim3(tp->n_pairs * kChunksPerTile), dim3(256), 0, st, tp->d_pool, tp->d_aux, tp->d_state, tp->d_pend,
                       tp->h_pairs, tp->n_pairs);
    tp->cow_copies += tp->n_pairs;
  }
  if (tp->n_patches)
    hipLaunchKernelGGL(k_table_patch, dim3((tp->n_patches + 255) / 256), dim3(256), 0, st, tp->d_tables[tp->cur],
                       tp->table_stride(), tp->h_patches, tp->n_patches);
  SLAMHIP_CHECK(hipGetLastError());
  if (tp->n_pairs || tp->n_patches) {
    // the kernels read the pinned lists: they must be done before the host reuses them
    SLAMHIP_CHECK(hipStreamSynchronize(st));
  }
  tp->n_pairs = tp->n_patches = 0;
  return SLAMHIP_OK;
}

int tile_pool_assign(TilePool *tp, const int *src_of_new) {
  const int stride = tp->table_stride();
  std::vector<int> nt((size_t)tp->n_slots * stride);
  for (int s = 0; s < tp->n_slots; ++s) {
    if (src_of_new[s] < 0 || src_of_new[s] >= tp->n_slots) return tp_fail("bad source slot");
    std::memcpy(nt.data() + (size_t)s * stride, tp->h_tables.data() + (size_t)src_of_new[s] * stride,
                sizeof(int) * stride);
    tp->h_assign[s] = src_of_new[s];
  }
  // refcounts of the new generation; tiles nobody references any more return to the free list
  std::vector<int> rc(tp->refcnt.size(), 0);
  for (int t : nt) rc[t] += 1;
  for (int t = 1; t < tp->next_unused; ++t)
    if (tp->refcnt[t] > 0 && rc[t] == 0 && tp->ancestor_of[t] < 0) tp->free_list.push_back(t);
  rc[0] = 1 << 30;
  tp->refcnt.swap(rc);
  tp->h_tables.swap(nt);
  hipLaunchKernelGGL(k_table_assign, dim3((stride + 255) / 256, tp->n_slots), dim3(256), 0, tp->ctx->stream,
                     tp->d_tables[tp->cur ^ 1], tp->d_tables[tp->cur], tp->h_assign, stride);
  SLAMHIP_CHECK(hipGetLastError());
  SLAMHIP_CHECK(hipStreamSynchronize(tp->ctx->stream));
  tp->cur ^= 1;
  return SLAMHIP_OK;
}

namespace {
constexpr size_t kTilePayloadBytes = (size_t)kTileCells * 4 * sizeof(double);
constexpr size_t kTileAuxBytes = (size_t)kTileCells * 2 * sizeof(double);
// header: int64 n_entries, then n_entries x (int32 x, int32 y: EXTERNAL cell of the tile's first cell -- pools
// grown differently still agree on those --, int32 ancestor ordinal or -1, int32 0)
size_t export_header_bytes(long long n_entries) { return 8 + (size_t)n_entries * 16; }
}  // namespace

void tile_pool_export_sizes(const TilePool *tp, int slot, size_t *header_bytes, size_t *body_bytes) {
  *header_bytes = *body_bytes = 0;
  if (slot < 0 || slot >= tp->n_slots) return;
  const int *row = tp->h_tables.data() + (size_t)slot * tp->table_stride();
  long long n = 0, with_content = 0;
  for (int i = 0; i < tp->table_stride(); ++i) {
    if (row[i] == 0) continue;
    ++n;
    with_content += tp->ancestor_of[row[i]] < 0;
  }
  *header_bytes = export_header_bytes(n);
  *body_bytes = (size_t)with_content * (kTilePayloadBytes + kTileAuxBytes);
}

size_t tile_pool_export_size(const TilePool *tp, int slot) {
  size_t h = 0, b = 0;
  tile_pool_export_sizes(tp, slot, &h, &b);
  return (slot < 0 || slot >= tp->n_slots) ? 0 : h + b;
}

int tile_pool_export_split(TilePool *tp, int slot, void *header_host, void *body, bool wait) {
  if (slot < 0 || slot >= tp->n_slots || !header_host) return tp_fail("bad export arguments");
  const int *row = tp->h_tables.data() + (size_t)slot * tp->table_stride();
  std::vector<int> idx;
  for (int i = 0; i < tp->table_stride(); ++i)
    if (row[i] != 0) idx.push_back(i);
  char *out = static_cast<char *>(header_host);
  const long long n = (long long)idx.size();
  std::memcpy(out, &n, 8);
  int *ent = reinterpret_cast<int *>(out + 8);
  bool any = false;
  for (size_t k = 0; k < idx.size(); ++k) {
    ent[4 * k] = (idx[k] % tp->tiles_x) * kTileSide - tp->origin_x;
    ent[4 * k + 1] = (idx[k] / tp->tiles_x) * kTileSide - tp->origin_y;
    ent[4 * k + 2] = tp->ancestor_of[row[idx[k]]];  // an untouched ancestor tile travels as its ordinal
    ent[4 * k + 3] = 0;
    any |= ent[4 * k + 2] < 0;
  }
  if (any && !body) return tp_fail("null export body");
  char *p = static_cast<char *>(body);
  for (int i : idx) {
    const size_t tile = (size_t)row[i];
    if (tp->ancestor_of[tile] >= 0) continue;
    // (hipMemcpyDefault: the body may be host memory -- a buffer for the caller's own transport -- or device memory,
    // from where RCCL sends it over xGMI)
    SLAMHIP_CHECK(hipMemcpyAsync(p, tp->d_pool + tile * kTileCells * 4, kTilePayloadBytes, hipMemcpyDefault, tp->ctx->stream));
    p += kTilePayloadBytes;
    hipLaunchKernelGGL(k_tile_fold, dim3(kTileCells / 256), dim3(256), 0, tp->ctx->stream, tp->d_aux, tp->d_pend, (int)tile);
    SLAMHIP_CHECK(hipGetLastError());
    SLAMHIP_CHECK(hipMemcpyAsync(p, tp->d_aux + tile * kTileCells * 2, kTileAuxBytes, hipMemcpyDefault, tp->ctx->stream));
    p += kTileAuxBytes;
  }
  if (wait) SLAMHIP_CHECK(hipStreamSynchronize(tp->ctx->stream));
  return SLAMHIP_OK;
}

int tile_pool_export(TilePool *tp, int slot, void *host_buf, size_t cap) {
  if (slot < 0 || slot >= tp->n_slots || !host_buf) return tp_fail("bad export arguments");
  size_t hb = 0, bb = 0;
  tile_pool_export_sizes(tp, slot, &hb, &bb);
  if (cap < hb + bb) return tp_fail("export buffer too small");
  return tile_pool_export_split(tp, slot, host_buf, static_cast<char *>(host_buf) + hb, true);
}

int tile_pool_assign_mixed(TilePool *tp, const int *src, int n_remote, const void *const *remote_bufs) {
  // one buffer per map: the header, then the body
  std::vector<const void *> bodies(n_remote > 0 ? n_remote : 0, nullptr);
  for (int r = 0; r < n_remote; ++r) {
    const char *in = static_cast<const char *>(remote_bufs[r]);
    if (!in) continue;
    long long n = 0;
    std::memcpy(&n, in, 8);
    if (n < 0 || n > (1 << 20)) return tp_fail("corrupt exported map");
    bodies[r] = in + export_header_bytes(n);
  }
  return tile_pool_assign_mixed_split(tp, src, n_remote, remote_bufs, bodies.data());
}

int tile_pool_assign_mixed_split(TilePool *tp, const int *src, int n_remote, const void *const *remote_bufs,
                                 const void *const *remote_bodies) {
  // a map that arrives from a pool grown further than this one: grow first (tables are re-laid out)
  for (int r = 0; r < n_remote; ++r) {
    bool used = false;
    for (int s = 0; s < tp->n_slots; ++s) used |= (src[s] == -r - 1);
    if (!used) continue;
    const char *in = static_cast<const char *>(remote_bufs[r]);
    if (!in) return tp_fail("missing exported map");
    long long n = 0;
    std::memcpy(&n, in, 8);
    if (n < 0 || n > (1 << 20)) return tp_fail("corrupt exported map");
    const int *ent = reinterpret_cast<const int *>(in + 8);
    for (long long k = 0; k < n; ++k) {
      const int ix = ent[4 * k] + tp->origin_x, iy = ent[4 * k + 1] + tp->origin_y;
      if ((ix & kTileMask) || (iy & kTileMask)) return tp_fail("exported map from a pool with another tile grid");
      const int rc = tile_pool_grow(tp, ix, iy, ix + kTileSide - 1, iy + kTileSide - 1);
      if (rc) return rc;
    }
  }
  const int stride = tp->table_stride();
  std::vector<int> nt((size_t)tp->n_slots * stride, 0);
  // tiles nobody will reference after this generation change can be reused for the imports: recount
  // the surviving local references first
  std::vector<int> rc(tp->refcnt.size(), 0);
  for (int s = 0; s < tp->n_slots; ++s) {
    if (src[s] >= tp->n_slots || src[s] < -n_remote) return tp_fail("bad source slot");
    if (src[s] < 0) continue;
    const int *from = tp->h_tables.data() + (size_t)src[s] * stride;
    std::memcpy(nt.data() + (size_t)s * stride, from, sizeof(int) * stride);
    for (int i = 0; i < stride; ++i) rc[from[i]] += 1;
  }
  // Everything that can fail is checked BEFORE the pool's bookkeeping is touched (an import that failed
  // half way -- pool exhausted, corrupt header -- used to leave tables pointing at tiles already on the free list):
  // the headers of every map that will be imported, and that the fresh tiles they need exist
  {
    long long fresh_needed = 0;
    for (int r = 0; r < n_remote; ++r) {
      bool used = false;
      for (int s = 0; s < tp->n_slots; ++s) used |= (src[s] == -r - 1);
      if (!used) continue;
      const char *in = static_cast<const char *>(remote_bufs[r]);
      long long n = 0;
      std::memcpy(&n, in, 8);
      if (n < 0 || n > stride) return tp_fail("corrupt exported map");
      const int *ent = reinterpret_cast<const int *>(in + 8);
      bool body_needed = false;
      for (long long k = 0; k < n; ++k) {
        const int tx = (ent[4 * k] + tp->origin_x) >> kTileShift, ty = (ent[4 * k + 1] + tp->origin_y) >> kTileShift;
        const int ord = ent[4 * k + 2];
        if (tx < 0 || tx >= tp->tiles_x || ty < 0 || ty >= tp->tiles_y) return tp_fail("corrupt exported map (tile position)");
        if (ord >= 0) {
          if (ord >= (int)tp->ancestor.size()) return tp_fail("exported map names an ancestor tile this pool lacks");
        } else {
          ++fresh_needed;
          body_needed = true;
        }
      }
      if (body_needed && !remote_bodies[r]) return tp_fail("missing body of an exported map");
    }
    long long available = (long long)tp->free_list.size() + (long long)(tp->capacity - tp->next_unused);
    for (int t = 1; t < tp->next_unused; ++t)
      if (tp->refcnt[t] > 0 && rc[t] == 0 && tp->ancestor_of[t] < 0) ++available;  // released by this generation change
    if (fresh_needed > available)
      return tp_fail("tile pool exhausted: create the particle maps with a larger capacity", SLAMHIP_ERR_STATE);
  }
  for (int t = 1; t < tp->next_unused; ++t)
    if (tp->refcnt[t] > 0 && rc[t] == 0 && tp->ancestor_of[t] < 0) tp->free_list.push_back(t);
  rc[0] = 1 << 30;
  tp->refcnt.swap(rc);
  // import every remote map once (none of the checks below can fire any more: they are the ones made above)
  std::vector<std::vector<int>> imported(n_remote);  // table row of the imported map
  hipStream_t st = tp->ctx->stream;
  for (int r = 0; r < n_remote; ++r) {
    bool used = false;
    for (int s = 0; s < tp->n_slots; ++s) used |= (src[s] == -r - 1);
    if (!used) continue;
    const char *in = static_cast<const char *>(remote_bufs[r]);
    if (!in) return tp_fail("missing exported map");
    long long n = 0;
    std::memcpy(&n, in, 8);
    if (n < 0 || n > stride) return tp_fail("corrupt exported map");
    const int *ent = reinterpret_cast<const int *>(in + 8);
    const char *p = static_cast<const char *>(remote_bodies[r]);  // host or device memory
    imported[r].assign(stride, 0);
    for (long long k = 0; k < n; ++k) {
      const int tx = (ent[4 * k] + tp->origin_x) >> kTileShift, ty = (ent[4 * k + 1] + tp->origin_y) >> kTileShift;
      const int ord = ent[4 * k + 2];
      if (tx < 0 || tx >= tp->tiles_x || ty < 0 || ty >= tp->tiles_y) return tp_fail("corrupt exported map (tile position)");
      const int ti = ty * tp->tiles_x + tx;
      if (ord >= 0) {  // the sender's untouched ancestor tile = this pool's ancestor tile of the same ordinal
        if (ord >= (int)tp->ancestor.size()) return tp_fail("exported map names an ancestor tile this pool lacks");
        imported[r][ti] = tp->ancestor[ord];
        continue;
      }
      int fresh = 0;
      int rcode = alloc_tile(tp, &fresh);
      if (rcode) return rcode;
      tp->refcnt[fresh] = 0;  // counted below, once per new slot that takes this map
      if (!p) return tp_fail("missing body of an exported map");
      SLAMHIP_CHECK(hipMemcpyAsync(tp->d_pool + (size_t)fresh * kTileCells * 4, p, kTilePayloadBytes,
                                   hipMemcpyDefault, st));
      p += kTilePayloadBytes;
      SLAMHIP_CHECK(hipMemcpyAsync(tp->d_aux + (size_t)fresh * kTileCells * 2, p, kTileAuxBytes, hipMemcpyDefault, st));
      p += kTileAuxBytes;
      SLAMHIP_CHECK(hipMemsetAsync(tp->d_pend + (size_t)fresh * kTileCells, 0, kTileCells * sizeof(unsigned), st));
      {
        const int rcs = tile_state_build(tp, fresh, 1);
        if (rcs) return rcs;
      }
      if (tp->nbr_ok) {  // (whatever state the sender's masks were in)
        hipLaunchKernelGGL(k_tile_nbr_build, dim3(kChunksPerTile), dim3(256), 0, st, tp->d_pool, fresh, tp->nbr_th);
        SLAMHIP_CHECK(hipGetLastError());
      }
      imported[r][ti] = fresh;
    }
  }
  for (int s = 0; s < tp->n_slots; ++s) {
    if (src[s] >= 0) continue;
    const std::vector<int> &row = imported[-src[s] - 1];
    std::memcpy(nt.data() + (size_t)s * stride, row.data(), sizeof(int) * stride);
    for (int i = 0; i < stride; ++i)
      if (row[i] != 0) tp->refcnt[row[i]] += 1;
  }
  tp->h_tables.swap(nt);
  SLAMHIP_CHECK(hipMemcpyAsync(tp->d_tables[tp->cur ^ 1], tp->h_tables.data(), sizeof(int) * tp->h_tables.size(),
                               hipMemcpyHostToDevice, st));
  SLAMHIP_CHECK(hipStreamSynchronize(st));  // the caller's host buffers may go away now
  tp->cur ^= 1;
  return SLAMHIP_OK;
}

int tile_pool_download(TilePool *tp, int slot, int x0, int y0, int w, int h, double *payload3, double *aux2) {
  if (slot < 0 || slot >= tp->n_slots || w <= 0 || h <= 0) return tp_fail("bad window");
  double *d_p = nullptr, *d_a = nullptr;
  const size_t cells = (size_t)w * h;
  if (payload3) SLAMHIP_CHECK(hipMalloc(&d_p, cells * 3 * sizeof(double)));
  if (aux2) SLAMHIP_CHECK(hipMalloc(&d_a, cells * 2 * sizeof(double)));
  hipLaunchKernelGGL(k_tiles_to_dense, dim3((w + 255) / 256, h), dim3(256), 0, tp->ctx->stream, tp->d_pool, tp->d_aux, tp->d_pend,
                     tp->d_table() + (size_t)slot * tp->table_stride(), tp->tiles_x, tp->tiles_y, x0 + tp->origin_x,
                     y0 + tp->origin_y, w, h, d_p, d_a, tp->unknown[0], tp->unknown[1], tp->unknown[2]);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && payload3)
    e = hipMemcpyAsync(payload3, d_p, cells * 3 * sizeof(double), hipMemcpyDeviceToHost, tp->ctx->stream);
  if (e == hipSuccess && aux2)
    e = hipMemcpyAsync(aux2, d_a, cells * 2 * sizeof(double), hipMemcpyDeviceToHost, tp->ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(tp->ctx->stream);
  if (d_p) hipFree(d_p);
  if (d_a) hipFree(d_a);
  if (e != hipSuccess) return hip_fail(e, "tile download");
  return SLAMHIP_OK;
}

void tile_pool_stats(const TilePool *tp, long long *tiles_in_use, long long *tiles_shared, long long *bytes,
                     long long *cow_copies) {
  long long used = 0, shared = 0;
  for (int t = 1; t < tp->next_unused; ++t) {
    if (tp->refcnt[t] > 0) ++used;
    if (tp->refcnt[t] > 1) ++shared;
  }
  if (tiles_in_use) *tiles_in_use = used;
  if (tiles_shared) *tiles_shared = shared;
  if (bytes) *bytes = used * (long long)kTileCells * 6 * (long long)sizeof(double);
  if (cow_copies) *cow_copies = tp->cow_copies;
}

}  // namespace slamhip
