"""On-disk fixture formats of slam-constructor, so that states dumped on a machine
that has ROS can be replayed through the HIP matcher (tools/sm_runner_hip.py).

Formats restated (paths relative to the reference root):
  .pose2D      "x y theta" text                      src/utils/sm_runner.cpp:23-34
  .scan2D      count, then "range angle is_occ" rows src/core/states/sensor_data.h:178-201
  .map         UnboundedPlainGridMap::save_state     src/core/maps/plain_grid_map.h:79-129 over
               Serializer (src/core/serialization.h:13-67): int h, int w, double scale, int origin.x,
               int origin.y, then h*w cells row by row; a cell is GridCell::serialize
               (grid_cell.h:37-41: prob_occ f64, estimation_quality f64, is_unknown 1 byte) and, for TBM
               cells, u, e, o, c f64 behind it (tbm_grid_cells.h:37-43).  Little endian, unpadded.
  .properties  key=value, '#' comments, <relative/include>   src/utils/properties_providers.h:127-191
  .pgm         map_dumpers.h-style grey map (255 * (1 - prob_occ))
"""
import os
import struct

import numpy as np

_HDR = struct.Struct("<iidii")  # h, w, scale, origin.x, origin.y


class MapFile:
    """Dense window decoded from a .map file; `payload` is what slamhip_map_upload_window takes."""

    def __init__(self, cell_model, payload, origin, scale, quality=None, is_unknown=None):
        self.cell_model = cell_model            # 0 OCC, 1 TBM
        self.payload = np.ascontiguousarray(payload, dtype=np.float64)
        self.height, self.width = self.payload.shape[:2]
        self.origin = (int(origin[0]), int(origin[1]))
        self.scale = float(scale)
        self.quality = quality                  # estimation_quality per cell (not used by the scorer)
        self.is_unknown = is_unknown
        self.unknown = np.array([0.5, 0, 0, 0]) if cell_model == 0 else np.array([1.0, 0.0, 0.0, 0.0])
        self.bounded = False


def read_map(path, cell="base"):
    """cell: 'base' (GridCell / MeanProbabilityCell / AffineQualityMergeCell: 17 bytes) or 'tbm' (49)."""
    raw = open(path, "rb").read()
    h, w, scale, ox, oy = _HDR.unpack_from(raw, 0)
    cell_bytes = 17 if cell == "base" else 49
    need = _HDR.size + h * w * cell_bytes
    if len(raw) != need:
        raise ValueError("%s: %d bytes, expected %d for %dx%d %s cells" % (path, len(raw), need, w, h, cell))
    dt = [("prob", "<f8"), ("qual", "<f8"), ("unk", "u1")]
    if cell != "base":
        dt += [("u", "<f8"), ("e", "<f8"), ("o", "<f8"), ("c", "<f8")]
    cells = np.frombuffer(raw, dtype=np.dtype(dt), count=h * w, offset=_HDR.size).reshape(h, w)
    if cell == "base":
        payload = cells["prob"][:, :, None]
        model = 0
    else:
        payload = np.stack([cells["u"], cells["e"], cells["o"], cells["c"]], axis=2)
        model = 1
    mf = MapFile(model, payload, (ox, oy), scale, cells["qual"].copy(), cells["unk"].astype(bool))
    mf.prob = cells["prob"].copy()  # Occupancy::prob_occ plane (what the PGM dump shows)
    return mf


def write_map(path, m, quality=None, is_unknown=None):
    """Inverse of read_map for OCC / TBM windows (m: cell_model, payload, origin, scale)."""
    h, w = m.payload.shape[:2]
    base = m.cell_model == 0
    dt = [("prob", "<f8"), ("qual", "<f8"), ("unk", "u1")]
    if not base:
        dt += [("u", "<f8"), ("e", "<f8"), ("o", "<f8"), ("c", "<f8")]
    cells = np.zeros((h, w), dtype=np.dtype(dt))
    cells["qual"] = 1.0 if quality is None else quality
    cells["unk"] = 0 if is_unknown is None else is_unknown
    if base:
        cells["prob"] = m.payload[..., 0]
    else:
        for k, name in enumerate("ueoc"):
            cells[name] = m.payload[..., k]
        qual = m.payload[..., 2] + m.payload[..., 1]
        with np.errstate(invalid="ignore", divide="ignore"):
            cells["prob"] = np.where(qual > 0, m.payload[..., 2] / qual, 0.5)
    with open(path, "wb") as f:
        f.write(_HDR.pack(h, w, m.scale, m.origin[0], m.origin[1]))
        f.write(cells.tobytes())


def read_pose2d(path):
    return np.array([float(v) for v in open(path).read().split()[:3]])


def write_pose2d(path, pose):
    with open(path, "w") as f:
        f.write("%.17g %.17g %.17g\n" % tuple(pose))


def read_scan2d(path):
    tok = open(path).read().split()
    n = int(tok[0])
    vals = np.array(tok[1:1 + 3 * n], dtype=np.float64).reshape(n, 3)
    return vals[:, 0].copy(), vals[:, 1].copy(), vals[:, 2].astype(np.int32)


def write_scan2d(path, rng, ang, is_occ=None):
    occ = np.ones(len(rng), np.int32) if is_occ is None else is_occ
    with open(path, "w") as f:
        f.write("%d\n" % len(rng))
        for r, a, o in zip(rng, ang, occ):
            f.write("%.17g %.17g %d\n" % (r, a, int(o)))


def read_properties(path, _glob=None, _depth=0):
    """FilePropertiesProvider::append_file_content (properties_providers.h:93-96,126-190).
    Inside one file a later key resets an earlier one; a file's keys are merged into the provider
    only when its parse ends and the merge keeps what is already there (unordered_map::insert), so
    an '<included/file>' (path relative to the including file) is merged BEFORE its includer and its
    keys win over the includer's.  No trimming; lines without '=' and missing files are skipped."""
    glob = {} if _glob is None else _glob
    if _depth > 64:
        raise RecursionError("circular include at " + path)
    if not os.path.isfile(path):
        return glob
    local = {}
    for line in open(path).read().split("\n"):
        if not line or line[0] == "#":
            continue
        if line[0] == "<":
            if line[-1] == ">":
                read_properties((os.path.dirname(path) or ".") + "/" + line[1:-1], glob, _depth + 1)
            continue
        i = line.find("=")
        if i < 0:
            continue
        local[line[:i]] = line[i + 1:]
    for k, v in local.items():
        glob.setdefault(k, v)
    return glob


def pgm_bytes(prob):
    """GridMapToPgmDumber::dump_map (src/utils/map_dumpers.h:65-90): header 'P5\\nW\\nH\\n255\\n',
    rows from the top (largest y) down, intensity = (unsigned char)(255 * (1 - clamp(prob, 0, 1)))."""
    prob = np.asarray(prob, dtype=np.float64)
    val = 1.0 - np.clip(prob, 0.0, 1.0)
    img = (255 * val).astype(np.uint8)[::-1]  # float -> integer conversion truncates, like the cast
    return b"P5\n%d\n%d\n255\n" % (img.shape[1], img.shape[0]) + img.tobytes()


def write_pgm(path, m):
    """PGM of a map window (occupancy probability plane)."""
    if m.cell_model == 0:
        prob = m.payload[..., 0]
    else:
        prob = getattr(m, "prob", None)
        if prob is None:
            raise ValueError("TBM window without its prob_occ plane")
    with open(path, "wb") as f:
        f.write(pgm_bytes(prob))


class UnboundedWindow:
    """Geometry of an UnboundedPlainGridMap as updates grow it: ensure_inside
    (src/core/maps/plain_grid_map.h:133-176) with its Expansion_Rate = 1.2 rule, unsigned
    arithmetic and the integer quotient `prep / (new - dim)` (1 when cells are prepended, 0 when
    appended).  Host-side mirror for tools that must reproduce the reference's map geometry
    (PGM size, `.map` header); the device window itself can be any superset."""
    EXPANSION_RATE = 1.2

    def __init__(self, width, height, origin=None):
        self.width, self.height = int(width), int(height)
        # RegularSquaresGrid starts with the origin in the middle (regular_squares_grid.h:120-122)
        self.origin = (self.width // 2, self.height // 2) if origin is None else (int(origin[0]), int(origin[1]))

    @staticmethod
    def _cells_nm(val, hi):
        if val < 0:
            return -val, 0
        if hi <= val:
            return 0, val - hi + 1
        return 0, 0

    def _grow_dim(self, dim, prep, app):
        new = prep + dim + app
        if dim < new and new < self.EXPANSION_RATE * dim:
            scale = float(prep // (new - dim))
            prep = int(prep + (self.EXPANSION_RATE * dim - new) * scale)
            new = int(self.EXPANSION_RATE * dim)
            app = new - (prep + dim)
        return new, prep, app

    def ensure_inside(self, cx, cy):
        """True when the window had to grow to hold external cell (cx, cy)."""
        ix, iy = cx + self.origin[0], cy + self.origin[1]
        if 0 <= ix < self.width and 0 <= iy < self.height:
            return False
        px, ax = self._cells_nm(ix, self.width)
        py, ay = self._cells_nm(iy, self.height)
        new_w, px, ax = self._grow_dim(self.width, px, ax)
        new_h, py, ay = self._grow_dim(self.height, py, ay)
        self.width, self.height = new_w, new_h
        self.origin = (self.origin[0] + px, self.origin[1] + py)
        return True
