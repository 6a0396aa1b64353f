#!/usr/bin/env python3
"""Dumps a flat-scene BVH the way the reference's debugging aids do: the flat float arrays as text
(Program/Utils.cpp:13-31) and the node boxes of a depth window as an OBJ wireframe (Program/BVHUI.hpp:59-103).
Host-only, no GPU.  Example: tools/dump_bvh.py --obj /path/model.obj --out /tmp/dragon --min-depth 1 --max-depth 6"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracing_amd as R
from raytracing_amd import bvhdebug

ap = argparse.ArgumentParser()
ap.add_argument("--obj", required=True)
ap.add_argument("--out", required=True, help="prefix: <out>_nodes.txt, <out>_tris.txt, <out>_boxes.obj")
ap.add_argument("--min-depth", type=int, default=0)
ap.add_argument("--max-depth", type=int, default=40)
ap.add_argument("--leaves", action="store_true")
a = ap.parse_args()
sc = R.SdScene().load_obj(a.obj).finalize()
fn, ft = sc.flat()
open(a.out + "_nodes.txt", "w").write(bvhdebug.dump_flat_floats(fn))
open(a.out + "_tris.txt", "w").write(bvhdebug.dump_flat_floats(ft))
n = bvhdebug.write_wireframe_obj(a.out + "_boxes.obj", bvhdebug.wireframe_boxes(sc, a.min_depth, a.max_depth, a.leaves))
print("nodes %d  triangles %d  root %d  boxes written %d" % (len(fn), len(ft), sc.root, n))
