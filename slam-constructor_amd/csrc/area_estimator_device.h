// area_estimator_device.h -- device restatement of AreaOccupancyEstimator::estimate_occupancy for K6.
//
//   src/core/maps/area_occupancy_estimator.h:27-64 (estimate), :68-86 (edge shift with its
//     function-local static, Q27), :88-137 (segment classification), :139-160 (intersections),
//     :162-216 (chunk area: triangle / trapezoid), :218-240 (area rate -> Occupancy)
//   src/core/geometry_primitives.h: Segment2D :36-92, Ray :115-163, Rectangle :322-407,
//     LightWeightRectangle::contains :246-250;  src/core/math_utils.h:15-25,37-51
// Evaluated per (beam, cell) inside k_mu_emit; all comparisons are the reference's fuzzy ones, all
// arithmetic FP64 in its operation order (no FMA contraction), so the result is bit-identical to
// the CPU path.  Invalid occupancy = (NaN, NaN): the cell update skips it.
#pragma once

namespace slamhip {
namespace ae {
struct ae_pt { double x, y; };
struct ae_seg { ae_pt beg, end; int is_horiz, is_vert, valid; };
struct ae_rect { double bot, top, left, right; };
/* edges are named 0 Bot, 1 Left, 2 Top, 3 Right (`loc`) */
struct ae_occ { double prob, qual; };

__device__ static __forceinline__ int ae_equal(double a, double b) {
  double m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
  double s = 1.0 > m ? 1.0 : m;
  return fabs(a - b) <= 1e-7 * s;
}
__device__ static __forceinline__ int ae_less(double a, double b) { return a < b + 2.220446049250313e-16; }
__device__ static __forceinline__ int ae_le(double a, double b) { return ae_equal(a, b) || ae_less(a, b); }
__device__ static __forceinline__ int ae_ordered(double a, double b, double c) { return ae_le(a, b) && ae_le(b, c); }
__device__ static __forceinline__ int ae_pt_equal(ae_pt a, ae_pt b) { return ae_equal(a.x, b.x) && ae_equal(a.y, b.y); }

__device__ static __forceinline__ ae_seg ae_make_seg(ae_pt b, ae_pt e) {
  ae_seg s;
  s.beg = b;
  s.end = e;
  s.is_horiz = ae_equal(b.y, e.y);
  s.is_vert = ae_equal(b.x, e.x);
  s.valid = 1;
  return s;
}
/* Segment2D::contains (axis-aligned segments only) */
__device__ static __forceinline__ int ae_seg_contains(ae_seg s, ae_pt p) {
  if (s.is_horiz) return ae_equal(p.y, s.beg.y) && ae_ordered(s.beg.x, p.x, s.end.x);
  if (s.is_vert) return ae_equal(p.x, s.beg.x) && ae_ordered(s.beg.y, p.y, s.end.y);
  return 0;
}
__device__ static __forceinline__ int ae_seg_contains_intersection(ae_seg s, ae_pt p) {
  int xin = ae_ordered(s.beg.x, p.x, s.end.x) || ae_ordered(s.end.x, p.x, s.beg.x);
  int yin = ae_ordered(s.beg.y, p.y, s.end.y) || ae_ordered(s.end.y, p.y, s.beg.y);
  return xin && yin;
}
__device__ static __forceinline__ int ae_rect_contains(ae_rect r, ae_pt p) {
  return ae_ordered(r.left, p.x, r.right) && ae_ordered(r.bot, p.y, r.top);
}
/* edges in Rectangle order: 0 bot, 1 top, 2 left, 3 right */
__device__ static __forceinline__ ae_seg ae_edge(ae_rect r, int i) {
  ae_pt lb = {r.left, r.bot}, rb = {r.right, r.bot}, lt = {r.left, r.top}, rt = {r.right, r.top};
  switch (i) {
    case 0: return ae_make_seg(lb, rb);
    case 1: return ae_make_seg(lt, rt);
    case 2: return ae_make_seg(lb, lt);
    default: return ae_make_seg(rb, rt);
  }
}
/* Rectangle::has_on_edge_line */
__device__ static __forceinline__ int ae_on_edge_line(ae_rect r, ae_seg s) {
  if (s.is_vert) return ae_equal(s.beg.x, r.left) || ae_equal(s.beg.x, r.right);
  if (s.is_horiz) return ae_equal(s.beg.y, r.bot) || ae_equal(s.beg.y, r.top);
  return 0;
}
/* Rectangle::find_containing_edge -> 1 if some edge contains p */
__device__ static __forceinline__ int ae_on_some_edge(ae_rect r, ae_pt p) {
  for (int i = 0; i < 4; ++i)
    if (ae_seg_contains(ae_edge(r, i), p)) return 1;
  return 0;
}
/* The intersections of a ray with the cell's four edges, in the order Rectangle::find_intersections(Ray) collects
 * them (geometry_primitives.h:360-384): slot 0 top, 1 left, 2 bot, 3 right.  The reference pushes them into a vector,
 * drops the last one if it equals the first, runs std::unique, and later filters again; here every edge keeps its
 * SLOT and carries a flag instead -- all indices are compile-time constants, so the four points live in registers
 * (the vector form, `out[m++] = ...`, cost 208-304 bytes of scratch per lane in every area-estimator kernel). */
struct ae_hits {
  ae_pt p[4];
  int ok[4];
};
__device__ static __forceinline__ int ae_slot_loc(int slot) { return slot == 0 ? 2 : (slot == 1 ? 1 : (slot == 2 ? 0 : 3)); }

/* Ray::intersect with one edge; ray = beg + alpha * delta */
__device__ static __forceinline__ void ae_ray_edge(ae_pt rb, ae_pt rd, ae_seg e, ae_pt *p, int *ok) {
  *ok = 0;
  p->x = p->y = 0.0;
  if (e.is_horiz) {
    if (ae_equal(rd.y, 0)) return;
    double alpha = (e.beg.y - rb.y) / rd.y;
    double ix = rb.x + alpha * rd.x;
    if (ix < e.beg.x || e.end.x < ix) return;
    p->x = ix;
    p->y = e.beg.y;
    *ok = 1;
    return;
  }
  if (e.is_vert) {
    if (ae_equal(rd.x, 0)) return;
    double alpha = (e.beg.x - rb.x) / rd.x;
    double iy = rb.y + alpha * rd.y;
    if (iy < e.beg.y || e.end.y < iy) return;
    p->x = e.beg.x;
    p->y = iy;
    *ok = 1;
  }
}
__device__ static __forceinline__ int ae_count(const ae_hits &h) { return h.ok[0] + h.ok[1] + h.ok[2] + h.ok[3]; }
/* Rectangle::find_intersections(Ray): order top, left, bot, right; vertex duplicates removed */
__device__ static __forceinline__ void ae_rect_ray(ae_rect r, ae_pt rb, ae_pt rd, ae_hits &h) {
  ae_ray_edge(rb, rd, ae_edge(r, 1), &h.p[0], &h.ok[0]);
  ae_ray_edge(rb, rd, ae_edge(r, 2), &h.p[1], &h.ok[1]);
  ae_ray_edge(rb, rd, ae_edge(r, 0), &h.p[2], &h.ok[2]);
  ae_ray_edge(rb, rd, ae_edge(r, 3), &h.p[3], &h.ok[3]);
  /* `if (1 < n && first == last) --n`: the last collected point goes when it repeats the first */
  const int n = ae_count(h);
  if (1 < n) {
    const ae_pt first = h.ok[0] ? h.p[0] : (h.ok[1] ? h.p[1] : (h.ok[2] ? h.p[2] : h.p[3]));
    const ae_pt last = h.ok[3] ? h.p[3] : (h.ok[2] ? h.p[2] : (h.ok[1] ? h.p[1] : h.p[0]));
    if (ae_pt_equal(first, last)) {
      if (h.ok[3]) h.ok[3] = 0;
      else if (h.ok[2]) h.ok[2] = 0;
      else if (h.ok[1]) h.ok[1] = 0;
      else h.ok[0] = 0;
    }
  }
  /* std::unique: a point equal to the one kept before it goes */
  ae_pt prev = {0.0, 0.0};
  int has_prev = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!h.ok[i]) continue;
    if (has_prev && ae_pt_equal(prev, h.p[i])) {
      h.ok[i] = 0;
    } else {
      prev = h.p[i];
      has_prev = 1;
    }
  }
}
/* Rectangle::find_intersections(Segment2D) */
__device__ static __forceinline__ void ae_rect_seg(ae_rect r, ae_seg s, ae_hits &h) {
  ae_pt d = {s.end.x - s.beg.x, s.end.y - s.beg.y};
  ae_rect_ray(r, s.beg, d, h);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (h.ok[i] && !ae_seg_contains_intersection(s, h.p[i])) h.ok[i] = 0;
}
/* the first two points a vector of the kept ones would hold, with their edges (values, not stores through selected
 * pointers: those were the last 48 bytes of scratch) */
struct ae_two {
  ae_pt p0, p1;
  int loc0, loc1;
};
__device__ static __forceinline__ ae_two ae_first_two(const ae_hits &h) {
  ae_two t;
  t.p0 = t.p1 = ae_pt{0.0, 0.0};
  t.loc0 = t.loc1 = 0;
  int seen = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool ok = h.ok[i] != 0;
    const bool is0 = ok && seen == 0, is1 = ok && seen == 1;
    t.p0.x = is0 ? h.p[i].x : t.p0.x;
    t.p0.y = is0 ? h.p[i].y : t.p0.y;
    t.loc0 = is0 ? ae_slot_loc(i) : t.loc0;
    t.p1.x = is1 ? h.p[i].x : t.p1.x;
    t.p1.y = is1 ? h.p[i].y : t.p1.y;
    t.loc1 = is1 ? ae_slot_loc(i) : t.loc1;
    seen += ok ? 1 : 0;
  }
  return t;
}
__device__ static __forceinline__ int ae_loc_is_horiz(int loc) { return loc == 0 || loc == 2; }

__device__ static __forceinline__ ae_occ ae_area_rate(double chunk, double total, int is_occ, const double *base4) {
  double rate = chunk / total;
  ae_occ o;
  if (is_occ) {
    o.prob = (rate < base4[2]) ? base4[2] : rate; /* std::max(area_rate, base_empty.prob) */
    o.qual = base4[1];
  } else {
    if (0.5 < rate) rate = 1 - rate;
    o.prob = base4[2];
    o.qual = base4[3] * rate;
  }
  return o;
}

/* AreaOccupancyEstimator::estimate_occupancy.  shift_amount = the function-local static of
 * ensure_segment_not_on_edge (low_qual 0.01 x side of the FIRST cell ever estimated, Q27);
 * unknown_qual = 0.5.  Invalid occupancy = (NaN, NaN). */
__device__ static __forceinline__ ae_occ ae_estimate(ae_pt beg, ae_pt end, ae_rect cell, int is_occ, const double *base4,
                         double shift_amount) {
  const ae_occ invalid = {__builtin_nan(""), __builtin_nan("")};
  const double unknown_qual = 0.5;
  ae_seg s = ae_make_seg(beg, end);
  if (ae_on_edge_line(cell, s)) {
    ae_pt sh = {0, 0};
    if (s.is_horiz) sh.y = (ae_equal(s.beg.y, cell.top) ? -1 : 1) * shift_amount;
    else if (s.is_vert) sh.x = (ae_equal(s.beg.x, cell.right) ? -1 : 1) * shift_amount;
    ae_pt nb = {s.beg.x + sh.x, s.beg.y + sh.y}, ne = {s.end.x + sh.x, s.end.y + sh.y};
    s = ae_make_seg(nb, ne);
  }
  /* classify_segment */
  int beg_in, end_in;
  {
    int beg_edge = ae_on_some_edge(cell, s.beg), end_edge = ae_on_some_edge(cell, s.end);
    if (beg_edge && end_edge) {
      beg_in = end_in = 0;
    } else {
      int bc = ae_rect_contains(cell, s.beg), ec = ae_rect_contains(cell, s.end);
      if (!beg_edge && !end_edge) {
        beg_in = bc;
        end_in = ec;
      } else if (beg_edge) {
        beg_in = 0;
        end_in = ec;
      } else {
        beg_in = bc;
        end_in = !bc;
      }
    }
  }
  enum { UNRELATED, LIES_INSIDE, STOPS_INSIDE, STARTS_INSIDE, PIERCES, TOUCHES } cls;
  ae_hits tmp;
  if (beg_in ^ end_in) {
    cls = beg_in ? STARTS_INSIDE : STOPS_INSIDE;
  } else if (beg_in) {
    cls = LIES_INSIDE;
  } else {
    ae_rect_seg(cell, s, tmp);
    const int k = ae_count(tmp);
    cls = k == 0 ? UNRELATED : (k == 1 ? TOUCHES : PIERCES);
  }
  switch (cls) {
    case UNRELATED:
    case TOUCHES: return invalid;
    case PIERCES:
    case STARTS_INSIDE:
      if (is_occ) return invalid;
      break;
    case LIES_INSIDE: {
      if (is_occ) return invalid;
      ae_occ o = {base4[2], unknown_qual};
      return o;
    }
    default: break;
  }
  /* find_intersections(beam, cell, is_occ): the occupied case intersects a ray through the beam's
   * end, perpendicular to the beam */
  ae_hits hits;
  if (is_occ) {
    ae_pt rb = {s.end.x, s.end.y}, rd = {s.beg.y - s.end.y, s.end.x - s.beg.x};
    ae_rect_ray(cell, rb, rd, hits);
  } else {
    ae_rect_seg(cell, s, hits); /* the ray from the beam's start along it, kept where the segment holds the point */
  }
  int ni = ae_count(hits);
  const double area = (cell.top - cell.bot) * (cell.right - cell.left);
  if (ni == 1) {
    if (!is_occ) {
      ae_occ o = {base4[2], unknown_qual};
      return o;
    }
    ae_rect_seg(cell, s, hits);
    if (ae_count(hits) <= 1) return ae_area_rate(area, area, is_occ, base4); /* stops at the front vertex */
    ni = 2; /* stops at the rear vertex: treat the cell as empty (the first two points of the segment's own list) */
    is_occ = 0;
  }
  const ae_two two = ae_first_two(hits);
  const ae_pt ip0 = two.p0, ip1 = two.p1;
  const int il0 = two.loc0, il1 = two.loc1;
  /* compute_chunk_area */
  double chunk;
  if (ni == 0) {
    chunk = area / 2;
  } else {
    double corner_x = 0, corner_y = 0;
    const int h0 = ae_loc_is_horiz(il0), h1 = ae_loc_is_horiz(il1);
    int is_triangle = h0 ^ h1;
    if (is_triangle) {
      /* the corner both edges share: each intersection's edge fixes one coordinate, in the order [0], [1] */
      if (il0 == 0) corner_y = cell.bot;
      else if (il0 == 2) corner_y = cell.top;
      else if (il0 == 1) corner_x = cell.left;
      else corner_x = cell.right;
      if (il1 == 0) corner_y = cell.bot;
      else if (il1 == 2) corner_y = cell.top;
      else if (il1 == 1) corner_x = cell.left;
      else corner_x = cell.right;
      chunk = 0.5;
      chunk *= h0 ? fabs(ip0.x - corner_x) : fabs(ip0.y - corner_y);
      chunk *= h1 ? fabs(ip1.x - corner_x) : fabs(ip1.y - corner_y);
    } else {
      corner_x = cell.left;
      corner_y = cell.bot;
      double base_sum = 0;
      base_sum += h0 ? fabs(ip0.x - corner_x) : fabs(ip0.y - corner_y);
      base_sum += h1 ? fabs(ip1.x - corner_x) : fabs(ip1.y - corner_y);
      chunk = 0.5 * (cell.top - cell.bot) * base_sum;
    }
    if (is_occ) {
      /* are_on_the_same_side(inters[0], inters[1], beam.beg(), corner) */
      double dx = ip1.x - ip0.x, dy = ip1.y - ip0.y;
      double a = dy * s.beg.y - dx * s.beg.x + dy * s.beg.x - dx * s.beg.y;
      double b = dy * corner_y - dx * corner_x + dy * corner_x - dx * corner_y;
      if (0 < a * b) chunk = area - chunk;
    }
  }
  return ae_area_rate(chunk, area, is_occ, base4);
}
}  // namespace ae
}  // namespace slamhip
