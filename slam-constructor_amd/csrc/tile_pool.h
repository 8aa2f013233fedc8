// tile_pool.h -- device tile pool for per-particle copy-on-write maps.
//
// Mirrors the sharing semantics of the reference's LazyTiledGridMap / UnboundedLazyTiledGridMap
// (src/core/maps/lazy_tiled_grid_map.h:18-187): a map is a table of tile references; copying a map
// copies the table (tiles shared, :40-45); all never-touched area is ONE shared unknown tile (:28-34);
// a write first makes the tile private (Tile::update -> clone when shared, :57-71,88-104).  Here a
// "map" is a SLOT of the pool (one per particle), tiles are 128 x 128 cells of 4 doubles (GMapping
// payload: prob_occ, obstacle x, obstacle y, pad) plus 2 doubles of update counters (hits, tries) in
// a parallel array, refcounts and the free list live on the host, and the copy itself is one kernel
// over (src, dst) tile pairs.  The extent starts as tiles_x x tiles_y tiles around the origin and grows by
// whole tiles when a scan reaches beyond it (tile_pool_grow); cells outside it read as unknown.
#pragma once

#include <vector>

#include "slamhip_internal.h"

namespace slamhip {

struct TilePool {
  slamhip_ctx *ctx = nullptr;
  int n_slots = 0, tiles_x = 0, tiles_y = 0;
  int origin_x = 0, origin_y = 0;  // internal (virtual) cell = external cell + origin
  double scale = 1.0;
  double unknown[4] = {0, 0, 0, 0};
  int capacity = 0;                // tiles
  double *d_pool = nullptr;        // [capacity][kTileCells][4]
  double *d_aux = nullptr;         // [capacity][kTileCells][2]
  // two bits per cell, sixteen cells per word: what the batched map update's free-space fast path (k_mu_classify)
  // has to know of a cell before it settles an observation with one atomic -- 1: the mean is +0, 3: the cell holds
  // the never-observed value (unknown[0] < 0; or -0), 0: anything else (the sorted chains).  4 KB per tile that stay in the
  // L2, read INSTEAD of the cell's payload (32 bytes of HBM per record: a third of that kernel's traffic).  Bits are only
  // ever cleared (3 -> 1 by the first settled observation, -> 0 by a cell store that leaves another mean): no races.
  unsigned *d_state = nullptr;     // [capacity][kTileCells / 16]
  // free observations settled by that fast path and not yet added to the cells' `tries` counters: one 4-byte word per
  // cell, so that the fast path's 195 M atomics per cfg5 step land sixteen cells to a 64-byte line instead of four
  // (the counters' read-modify-write was 1.8 of the kernel's 3.6 ms).  A cell's tries = aux tries + pending: whoever
  // reads the counters adds it (mu_cell_load -- which also folds it in --, clones, downloads, exports).
  unsigned *d_pend = nullptr;      // [capacity][kTileCells]
  int *d_tables[2] = {nullptr, nullptr};  // [n_slots][tiles_x * tiles_y]; double-buffered for resampling
  int cur = 0;
  std::vector<int> h_tables;       // host mirror of d_tables[cur]
  std::vector<int> refcnt;         // per tile; tile 0 = the shared unknown tile (never written, never freed)
  std::vector<int> free_list;
  int next_unused = 1;
  // pending work of the current batch (pinned, read by the kernels over PCIe)
  int *h_pairs = nullptr;          // (src, dst) per copy
  int *h_patches = nullptr;        // (slot, index, tile) per table patch
  int *h_assign = nullptr;         // source slot per new slot
  int cap_pairs = 0, cap_patches = 0, n_pairs = 0, n_patches = 0;
  long long cow_copies = 0;        // tiles copied so far
  long long growths = 0;           // times the extent grew
  // tiles of the common ancestor map (tile_pool_init_from_dense): never freed, so that a migrating map
  // can name them by ordinal instead of carrying their 768 KiB -- every rank builds the same ancestor
  std::vector<int> ancestor;       // ordinal -> tile id
  std::vector<int> ancestor_of;    // tile id -> ordinal or -1
  // neighbourhood masks in the cells' pads (slamhip_internal.h MapView), IN-TILE neighbours only: a tile is shared by
  // maps whose neighbouring tiles differ, so a cell on a tile's rim cannot know the cells across it -- the scorer asks
  // the cell across the rim for its own mask instead (gm_score_device.h).  Valid for threshold nbr_th.
  bool nbr_ok = false;
  double nbr_th = 0.0;

  int table_stride() const { return tiles_x * tiles_y; }
  int width() const { return tiles_x * kTileSide; }
  int height() const { return tiles_y * kTileSide; }
  const int *d_table() const { return d_tables[cur]; }
};

int tile_pool_create(slamhip_ctx *ctx, int n_slots, int tiles_x, int tiles_y, double scale, const double unknown[4],
                     int capacity, TilePool **out);
void tile_pool_destroy(TilePool *tp);
// every slot starts as a copy of the bound dense GMAPPING window `m` (payload and, if present, counters)
int tile_pool_init_from_dense(TilePool *tp, const DeviceMap &m);
// make internal cells [x0, x1] x [y0, y1] (possibly negative / beyond the extent) part of the extent.
// Nothing may be queued (tile_pool_flush first); TiledTarget views must be rebuilt afterwards.
int tile_pool_grow(TilePool *tp, int x0, int y0, int x1, int y1);
// make the tiles of `slot` that intersect internal cells [x0, x1] x [y0, y1] private (queued)
int tile_pool_make_private(TilePool *tp, int slot, int x0, int y0, int x1, int y1);
// run the queued copies and table patches on the context's stream
int tile_pool_flush(TilePool *tp);
// resampling: new slot i becomes a copy of old slot src[i] (tables only; tiles get shared)
int tile_pool_assign(TilePool *tp, const int *src_of_new);
// ---- migration between pools (particles that move to another GPU on resampling) --------------------
// A slot's map leaves as one host buffer: int64 n_tiles, per tile (int32 x, y of its first cell in
// EXTERNAL coordinates, int32 ancestor ordinal or -1, int32 0), then per tile that is not an ancestor
// tile 16384 x 4 payload doubles and 16384 x 2 counter doubles.  Every tile the slot
// references except the unknown tile is included (the receiving pool cannot know which of them equal
// its own ancestor tiles).
size_t tile_pool_export_size(const TilePool *tp, int slot);
int tile_pool_export(TilePool *tp, int slot, void *host_buf, size_t cap);
// the same in two parts: the header (tile positions, ancestor ordinals: host memory, header_bytes) and the body (tile
// contents: host OR device memory, body_bytes) -- the sharded filter step all-gathers the small headers and sends
// the bodies device to device (RCCL send / recv over xGMI).  wait = false leaves the copies queued on the stream.
void tile_pool_export_sizes(const TilePool *tp, int slot, size_t *header_bytes, size_t *body_bytes);
int tile_pool_export_split(TilePool *tp, int slot, void *header_host, void *body, bool wait);
// new generation with remote sources: new slot s becomes a copy of old local slot src[s] when
// src[s] >= 0, or of the exported map remote_bufs[-src[s] - 1] otherwise (imported once, shared by
// every new slot that names it -- the tiles are then shared like after a local copy)
int tile_pool_assign_mixed(TilePool *tp, const int *src, int n_remote, const void *const *remote_bufs);
// ... with the bodies apart from the headers (remote_bufs[k] = header in host memory, remote_bodies[k] = tile contents
// in host or device memory)
int tile_pool_assign_mixed_split(TilePool *tp, const int *src, int n_remote, const void *const *remote_bufs,
                                 const void *const *remote_bodies);
// external window [x0, x0+w) x [y0, y0+h) of a slot: payload (3 doubles per cell: prob, obst.x, obst.y)
// and counters (2 per cell: hits, tries); either may be null
int tile_pool_download(TilePool *tp, int slot, int x0, int y0, int w, int h, double *payload3, double *aux2);
// derives the masks of every tile in use for threshold th, unless they are there (waited for).  Leaves nbr_ok false
// where the masks cannot be kept by the writers: th <= 0 or an unknown cell that counts as full.
int tile_pool_nbr_masks(TilePool *tp, double th);
// (testing) cells whose stored mask differs from the one their in-tile neighbours give
int tile_pool_nbr_check(TilePool *tp, long long *mismatches);
// (testing) words of the settle-state plane that differ from what the cells' payloads give (tiles in use)
int tile_pool_state_check(TilePool *tp, long long *mismatches);
void tile_pool_stats(const TilePool *tp, long long *tiles_in_use, long long *tiles_shared, long long *bytes,
                     long long *cow_copies);

// map_update.hip: GridMapScanAdder::append_scan of one scan from n_jobs poses, job k into slot slots[k]
int mu_append_batch(slamhip_ctx *ctx, TilePool *tp, const slamhip_scan_adder_cfg *cfg, int n_jobs,
                    const double *poses, const int *slots, int n, const double *range, const double *cos_a,
                    const double *sin_a, const int *is_occ, long long *n_updates_out);

}  // namespace slamhip
