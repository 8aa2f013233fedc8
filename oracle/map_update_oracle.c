/* oracle/map_update_oracle.c -- TEST INFRASTRUCTURE ONLY (part of liboracle.so).
 *
 * CPU restatement of the map update (const occupancy estimator):
 *   GridMapScanAdder::append_scan                 src/core/maps/grid_map_scan_adders.h:54-75
 *   WallDistanceBlurringScanAdder::handle_scan_point  :138-172, blur_cell_dist :176-189
 *   ConstOccupancyEstimator                       src/core/maps/const_occupancy_estimator.h:6-17
 *   cell updates: GridCell::operator+= (grid_cell.h:27-30), AffineQualityMergeCell (naive_grid_cells.h:14-20),
 *     MeanProbabilityCell (:33-40), TbmBaseCell (tbm_grid_cells.h:12-19, aoo2tbm :57-66,
 *     transferable_belief_model.h:102-143), GmappingBaseCell (src/slams/gmapping/gmapping_grid_cell.h:20-33)
 * The ray walk is orc_world_to_cells (slam_oracle.c).  Pinned by tests/golden/map_update.npz
 * (payloads and update counters exported from the compiled reference after every scan).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "slam_oracle.h"
#include "area_estimator.h"

/* AreaOccupancyEstimator::estimate_occupancy on its own (area_occupancy_estimator.h:27-64), for the
 * known answers of test/core/maps/area_occupancy_estimator_test.cpp: beam (x0, y0, x1, y1), cell
 * (bot, top, left, right), base4 = occupied (prob, qual), empty (prob, qual); out = (prob, qual) */
void orc_area_estimate(const double *beam4, const double *cell4, int is_occ, const double *base4,
                       double low_qual, double unknown_qual, double *out2) {
  ae_pt b = {beam4[0], beam4[1]}, e = {beam4[2], beam4[3]};
  ae_rect c = {cell4[0], cell4[1], cell4[2], cell4[3]};
  /* Shift_Amount = low_qual * cell.side() of the first cell seen (Q27): every test fixture has one cell */
  ae_occ o = ae_estimate_ex(b, e, c, is_occ, base4, low_qual * (c.top - c.bot), unknown_qual);
  out2[0] = o.prob;
  out2[1] = o.qual;
}

static void tbm_conj(const double *lhs, const double *rhs, double *out) {
  double tmp[4] = {0.0, 0.0, 0.0, 0.0};
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) tmp[a | b] += lhs[a] * rhs[b];
  double tot = tmp[0] + tmp[1] + tmp[2] + tmp[3];
  if (tot == 0.0) {
    out[0] = 1.0;
    out[1] = out[2] = out[3] = 0.0;
  } else {
    for (int i = 0; i < 4; ++i) out[i] = tmp[i] / tot;
  }
}

/* cell += AreaOccupancyObservation{is_occupied, {prob, est_qual}, obstacle, quality} */
static void cell_update(int rule, double *cell, double *aux, double prob, double est_qual,
                        double quality, double obx, double oby) {
  if (isnan(prob) || isnan(est_qual)) {
    if (rule != ORC_RULE_LAST) return; /* every model but the base cell skips invalid occupancy */
  }
  switch (rule) {
    case ORC_RULE_LAST:
      cell[0] = prob;
      break;
    case ORC_RULE_AFFINE:
      cell[0] = (1.0 - quality) * cell[0] + quality * prob;
      break;
    case ORC_RULE_MEAN: {
      aux[0] += 1;
      double that_p = 0.5 + (prob - 0.5) * quality;
      cell[0] = (cell[0] * (aux[0] - 1) + that_p) / aux[0];
      break;
    }
    case ORC_RULE_TBM: {
      double eq = est_qual * quality;
      double occupied = prob * eq, empty = (1 - prob) * eq;
      double that[4] = {1.0 - occupied - empty, empty, occupied, 0.0};
      double b[4];
      tbm_conj(cell, that, b);
      double weight = b[0] + b[1] + b[2];
      if (weight == 0.0) {
        cell[0] = 1.0;
        cell[1] = cell[2] = cell[3] = 0.0;
      } else {
        cell[0] = b[0] / weight;
        cell[1] = b[1] / weight;
        cell[2] = b[2] / weight;
        cell[3] = 0.0;
      }
      break;
    }
    case ORC_RULE_GMAPPING: {
      int hits = (int)aux[0], tries = (int)aux[1];
      ++tries;
      int is_free = prob <= 0.5;
      double aoo_p = is_free ? 0.0 : prob;
      cell[0] = (cell[0] * (tries - 1) + aoo_p) / tries;
      if (!is_free) {
        ++hits;
        cell[1] = (cell[1] * (hits - 1) + obx) / hits;
        cell[2] = (cell[2] * (hits - 1) + oby) / hits;
      }
      aux[0] = hits;
      aux[1] = tries;
      break;
    }
  }
}

/* returns the number of cell updates, or -1 when a touched cell lies outside the window */
long long orc_append_scan(const orc_map *map, double *payload, double *aux, int rule, const double *pose,
                          int n, const double *range, const double *angle, const int *is_occ,
                          const orc_scan *trig, double scan_quality, const double *base4, double blur,
                          double max_range) {
  return orc_append_scan_ex(map, payload, aux, rule, pose, n, range, angle, is_occ, trig, scan_quality,
                            base4, blur, max_range, 0, 0.0);
}

/* est_kind 0: ConstOccupancyEstimator, 1: AreaOccupancyEstimator (area_estimator.h) */
long long orc_append_scan_ex(const orc_map *map, double *payload, double *aux, int rule, const double *pose,
                             int n, const double *range, const double *angle, const int *is_occ,
                             const orc_scan *trig, double scan_quality, const double *base4, double blur,
                             double max_range, int est_kind, double shift_amount) {
  return orc_append_scan_q(map, payload, aux, rule, pose, n, range, angle, is_occ, trig, scan_quality, base4, blur,
                           max_range, est_kind, shift_amount, NULL);
}

/* ObservationMappingQualityEstimator::quality per point (grid_map_scan_adders.h:17-43): kind 0 IdleOMQE (1.0),
 * kind 1 AngleHistogramResiprocalOMQE = 1 / AngleHistogram::value (src/core/features/angle_histogram.h:17-43,
 * 71-96): twenty bins over [0, pi) of the direction of the segment from point i-1 to point i (points in the
 * sensor frame), a point's value = the count of its segment's bin, point 0 = the number of points */
void orc_omqe_quality(int kind, int n, const double *range, const double *angle, double *out) {
  if (kind == 0) {
    for (int i = 0; i < n; ++i) out[i] = 1.0;
    return;
  }
  unsigned hist[20];
  memset(hist, 0, sizeof(hist));
  double *dirs = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
  const double step = (180 * M_PI / 180) / 20;
  for (int i = 1; i < n; ++i) {
    double s1, c1, s0, c0;
    sincos(angle[i], &s1, &c1);
    sincos(angle[i - 1], &s0, &c0);
    const double d_x = range[i] * c1 - range[i - 1] * c0;
    const double d_y = range[i] * s1 - range[i - 1] * s0;
    double a = 0;
    if (d_y != 0) {
      a = acos(d_x / sqrt(d_x * d_x + d_y * d_y));
      if (d_y < 0 && d_x != 0) a = M_PI - a;
    }
    hist[(size_t)floor(a / step)]++;
    dirs[i] = a;
  }
  for (int i = 0; i < n; ++i) {
    const unsigned v = i == 0 ? (unsigned)n : hist[(size_t)floor(dirs[i] / step)];
    out[i] = 1.0 / v;
  }
  free(dirs);
}

/* beam_quality (n values, or NULL = IdleOMQE): what _omqe->quality(points, pt_i) returns for every point */
long long orc_append_scan_q(const orc_map *map, double *payload, double *aux, int rule, const double *pose,
                            int n, const double *range, const double *angle, const int *is_occ,
                            const orc_scan *trig, double scan_quality, const double *base4, double blur,
                            double max_range, int est_kind, double shift_amount, const double *beam_quality) {
  if (n <= 0) return 0;
  orc_scan s = *trig;
  s.range = range;
  s.angle = angle;
  double sin_b, cos_b;
  sincos(pose[2], &sin_b, &cos_b);
  const int st = map->cell_model == ORC_CELL_TBM ? 4 : (map->cell_model == ORC_CELL_GMAPPING ? 3 : 1);
  const int ast = rule == ORC_RULE_MEAN ? 1 : (rule == ORC_RULE_GMAPPING ? 2 : 0);
  const double scale = map->scale;
  const double max_range_sq = max_range * max_range; /* std::pow(max_usable_range, 2) */
  int cap = 4 * (map->width + map->height) + 16;
  int *cells = (int *)malloc(sizeof(int) * 2 * (size_t)cap);
  long long updates = 0;
  for (int i = 0; i < n; ++i) {
    double wx, wy;
    orc_endpoint(&s, i, pose, sin_b, cos_b, &wx, &wy);
    const int occ = is_occ ? is_occ[i] : 1;
    const double quality = scan_quality * (beam_quality ? beam_quality[i] : 1.0); /* x _omqe->quality(points, i) */
    const double ddx = wx - pose[0], ddy = wy - pose[1];
    if (max_range_sq < ddx * ddx + ddy * ddy) continue; /* Segment2D::length_sq */
    const int rcx = (int)floor(pose[0] / scale), rcy = (int)floor(pose[1] / scale);
    const int ocx = (int)floor(wx / scale), ocy = (int)floor(wy / scale);
    const double odx = rcx - ocx, ody = rcy - ocy;
    const double obst_dist_sq = odx * odx + ody * ody;
    double blur_dist = 0;
    if (occ) {
      blur_dist = blur / scale;
      if (blur_dist < 0) blur_dist *= -(ddx * ddx + ddy * ddy);
    }
    const double hole_dist_sq = blur_dist * blur_dist;
    int nc = orc_world_to_cells(scale, pose[0], pose[1], wx, wy, cap, cells);
    if (nc > cap) {
      free(cells);
      return -1;
    }
    /* obstacle cell first (grid_map_scan_adders.h:159-162), then the rest in walk order */
    double base_prob = occ ? base4[0] : base4[2], base_qual = occ ? base4[1] : base4[3];
    const ae_pt bbeg = {pose[0], pose[1]}, bend = {wx, wy};
    if (est_kind == 1) {
      const int lx = cells[2 * (nc - 1)], ly = cells[2 * (nc - 1) + 1];
      const ae_rect cb = {scale * ly, scale * (ly + 1), scale * lx, scale * (lx + 1)};
      const ae_occ o = ae_estimate(bbeg, bend, cb, occ, base4, shift_amount);
      base_prob = o.prob;
      base_qual = o.qual;
    }
    for (int pass = 0; pass < 2; ++pass) {
      const int lo = pass == 0 ? nc - 1 : 0, hi = pass == 0 ? nc : nc - 1;
      for (int k = lo; k < hi; ++k) {
        const int cx = cells[2 * k], cy = cells[2 * k + 1];
        const int ix = cx + map->origin_x, iy = cy + map->origin_y;
        if (ix < 0 || ix >= map->width || iy < 0 || iy >= map->height) {
          free(cells);
          return -1;
        }
        double prob, qual;
        if (pass == 0) {
          prob = base_prob;
          qual = base_qual;
        } else {
          prob = base4[2];
          qual = base4[3];
          if (est_kind == 1) {
            const ae_rect cb = {scale * cy, scale * (cy + 1), scale * cx, scale * (cx + 1)};
            const ae_occ o = ae_estimate(bbeg, bend, cb, 0, base4, shift_amount);
            prob = o.prob;
            qual = o.qual;
          }
          const double cdx = cx - ocx, cdy = cy - ocy;
          const double dist_sq = cdx * cdx + cdy * cdy;
          if (dist_sq < hole_dist_sq && hole_dist_sq < obst_dist_sq) {
            const double prob_scale = 1.0 - dist_sq / hole_dist_sq;
            prob = base_prob * prob_scale;
          }
        }
        const size_t ci = (size_t)iy * map->width + ix;
        cell_update(rule, payload + ci * st, aux ? aux + ci * ast : NULL, prob, qual, quality, wx, wy);
        ++updates;
      }
    }
  }
  free(cells);
  return updates;
}
