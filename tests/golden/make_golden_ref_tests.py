#!/usr/bin/env python3
"""Known answers held by the reference's OWN unit tests, extracted as data.

gtest is not in the image, so those tests cannot run; but their inputs and expected values are plain
literals.  This script (run where /root/reference exists) parses them out of the test files and writes
tests/golden/reference_test_vectors.json -- inputs and expected outputs only, no source text:

  world_to_cells      test/core/maps/regular_squares_grid_test.cpp      RSGSegmentRasterizationTest
  discrete_segment    test/core/geometry_discrete_primitives_test.cpp   DiscreteSegment2DRasterizationTest
  area_estimator      test/core/maps/area_occupancy_estimator_test.cpp  AreaOccupancyEstimatorTest
  angle_histogram     test/core/features/angle_histogram_test.cpp       AHAngleEstimationTest
  map_growth          test/core/maps/unbounded_plain_grid_map_test.cpp  UnboundedPlainGridMapTest expand*
  trig_cache          test/core/trigonometry_utils_test.cpp             the two sector / step / rotation cases

tests/test_reference_test_vectors.py checks the oracle (and the host mirrors of the product) against
them.  Cases a parser cannot take as literals (loops, helper-built geometry) are listed under `skipped`.
"""
import json
import math
import os
import re

REF = "/root/reference/test/core"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_test_vectors.json")

NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def bodies(text, fixture):
    """(name, body) of every TEST / TEST_F of a fixture."""
    for m in re.finditer(r"TEST(?:_F)?\(\s*%s\s*,\s*(\w+)\s*\)\s*\{" % fixture, text):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        yield m.group(1), text[m.end():i - 1]


def int_pairs(s):
    return [[int(a), int(b)] for a, b in re.findall(r"\{\s*(-?\d+)\s*,\s*(-?\d+)\s*\}", s)]


def balanced(s, start):
    """text inside the parenthesis/brace that opens at s[start]"""
    op = s[start]
    cl = {"(": ")", "{": "}"}[op]
    depth, i = 1, start + 1
    while depth:
        depth += 1 if s[i] == op else (-1 if s[i] == cl else 0)
        i += 1
    return s[start + 1:i - 1]


def ev(expr, consts=None):
    ns = {"tan": math.tan, "deg2rad": math.radians, "M_PI": math.pi}
    ns.update(consts or {})
    return float(eval(expr.replace("std::", ""), {"__builtins__": {}}, ns))  # noqa: S307 (literals of a test file)


def world_to_cells():
    text = open(os.path.join(REF, "maps", "regular_squares_grid_test.cpp")).read()
    cases, skipped = [], []
    scale = 0.1
    for name, body in bodies(text, "RSGSegmentRasterizationTest"):
        ds = body.find("DSegment(")
        wc = body.find("world_to_cells(")
        if ds < 0 or wc < 0:
            skipped.append(name)
            continue
        cells = int_pairs(balanced(body, ds + len("DSegment")))
        arg = balanced(body, wc + len("world_to_cells"))
        pts = []
        for m in re.finditer(r"cell_middle\(\{\s*(-?\d+)\s*,\s*(-?\d+)\s*\}\)|\{\s*(%s)\s*,\s*(%s)\s*\}" % (NUM, NUM), arg):
            if m.group(1) is not None:  # RegularSquaresGrid::cell_to_world: scale * (c + 0.5)
                pts.append([scale * (int(m.group(1)) + 0.5), scale * (int(m.group(2)) + 0.5)])
            else:
                pts.append([float(m.group(3)), float(m.group(4))])
        if len(pts) != 2 or not cells:
            skipped.append(name)
            continue
        cases.append({"name": name, "segment": pts[0] + pts[1], "cells": cells})
    return {"scale": scale, "cases": cases, "skipped": skipped}


def discrete_segment():
    text = open(os.path.join(REF, "geometry_discrete_primitives_test.cpp")).read()
    cases, skipped = [], []
    for name, body in bodies(text, "DiscreteSegment2DRasterizationTest"):
        sg = body.find("DiscreteSegment2D{")
        dp = body.find("DPoints(")
        if sg < 0 or dp < 0:
            skipped.append(name)
            continue
        ends = int_pairs(balanced(body, sg + len("DiscreteSegment2D")))
        pts = int_pairs(balanced(body, dp + len("DPoints")))
        if len(ends) != 2 or not pts:
            skipped.append(name)
            continue
        cases.append({"name": name, "ends": ends[0] + ends[1], "points": pts})
    return {"cases": cases, "skipped": skipped}


def area_estimator():
    text = open(os.path.join(REF, "maps", "area_occupancy_estimator_test.cpp")).read()
    consts = {k: float(v) for k, v in re.findall(r"static constexpr double (\w+) = (%s);" % NUM, text)}
    cell = [float(v) for v in re.search(r"cell\{\s*(%s)\s*,\s*(%s)\s*,\s*(%s)\s*,\s*(%s)\s*\}" % ((NUM,) * 4), text).groups()]
    cases, skipped = [], []
    for name, body in bodies(text, "AreaOccupancyEstimatorTest"):
        bm = re.search(r"Segment2D\{\s*\{(.+?),(.+?)\}\s*,\s*\{(.+?),(.+?)\}\s*\}", body)
        calls = re.findall(r"estimate_occupancy\(\s*beam\s*,\s*cell\s*,\s*(true|false)\s*\)", body)
        oc = body.find("Occupancy(")
        invalid = "Occupancy::invalid()" in body
        if not bm or len(calls) != 1 or (not invalid and (oc < 0 or body.count("Occupancy(") != 1)):
            skipped.append(name)
            continue
        try:
            beam = [ev(g, consts) for g in bm.groups()]
            # Occupancy::invalid() = (NaN, NaN) (state_data.h): stored as null
            exp = None if invalid else [ev(v, consts) for v in balanced(body, oc + len("Occupancy")).split(",")]
        except Exception:  # noqa: BLE001
            skipped.append(name)
            continue
        cases.append({"name": name, "beam": beam, "is_occ": calls[0] == "true", "expected": exp})
    return {"base_occupied": [consts["Base_Occup_Prob"], 1.0], "base_empty": [consts["Base_Empty_Prob"], 1.0],
            "low_qual": consts["Low_Est_Qual"], "unknown_qual": consts["Unknown_Est_Qual"], "cell_btlr": cell,
            "cases": cases, "skipped": skipped}


def angle_histogram():
    text = open(os.path.join(REF, "features", "angle_histogram_test.cpp")).read()
    cases = []
    for name, body in bodies(text, "AHAngleEstimationTest"):
        m = re.search(r"test_angle_estimation\(\{(.+?),(.+?)\}\s*,\s*\{(.+?),(.+)\}\s*,\s*(%s)\s*\)" % NUM, body)
        p = [ev(g) for g in m.groups()[:4]]
        cases.append({"name": name, "p1": p[:2], "p2": p[2:], "expected_deg": float(m.group(5))})
    return {"tolerance_deg": 0.1, "cases": cases}


def map_growth():
    text = open(os.path.join(REF, "maps", "unbounded_plain_grid_map_test.cpp")).read()
    cases, skipped = [], []
    for name, body in bodies(text, "UnboundedPlainGridMapTest"):
        up = re.findall(r"map\.update\(\{\s*(-?\d+)\s*,\s*(-?\d+)\s*\}", body)
        mi = re.search(r"MapInfo\(\s*(\d+)\s*,\s*(\d+)\s*,\s*(-?\d+)\s*,\s*(-?\d+)\s*\)", body)
        if len(up) != 1 or not mi:
            skipped.append(name)
            continue
        cases.append({"name": name, "update": [int(v) for v in up[0]], "expected_whoxoy": [int(v) for v in mi.groups()]})
    return {"start_wh": [1, 1], "cases": cases, "skipped": skipped}


def trig_cache():
    text = open(os.path.join(REF, "trigonometry_utils_test.cpp")).read()
    cases = []
    for name, body in bodies(text, "CachedTrigonometryProviderTest"):
        c = dict(re.findall(r"(\w+) = (deg2rad\(-?\d+\))", body))
        cases.append({"name": name, "min": ev(c["Min"]), "max": ev(c["Max"]), "step": ev(c["Step"]),
                      "rotation": ev(c["D_Theta"]) if "D_Theta" in c else 0.0})
    return {"tolerance": 2.220446049250313e-16, "cases": cases}


def main():
    out = {"world_to_cells": world_to_cells(), "discrete_segment": discrete_segment(),
           "area_estimator": area_estimator(), "angle_histogram": angle_histogram(), "map_growth": map_growth(),
           "trig_cache": trig_cache()}
    json.dump(out, open(OUT, "w"), indent=0, separators=(",", ":"))
    for k, v in out.items():
        print("%-18s %3d cases, skipped %s" % (k, len(v["cases"]), v.get("skipped", [])))
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
