"""Minimal Wavefront OBJ reader used by the tests as an independent check of the product's C++ reader.

TEST INFRASTRUCTURE ONLY.  Produces what sd::ModelLoader::ProcessMesh (Include/ModelLoader.hpp:197-232)
hands to the mesh constructor after assimp's Triangulate step: one (pos3, nrm3, uv2) record per
triangle corner, faces in file order, uv = 0 when the file has none, polygons split as a fan around
their first corner (assimp's own n-gon order is un-pinned; the fan is this build's documented choice).
One mesh per `o`/`g` group that holds faces.
"""
import numpy as np


def load_obj(path):
    v, vn, vt = [], [], []
    meshes = [[]]
    with open(path, "r") as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            k = p[0]
            if k == "v":
                v.append((float(p[1]), float(p[2]), float(p[3])))
            elif k == "vn":
                vn.append((float(p[1]), float(p[2]), float(p[3])))
            elif k == "vt":
                vt.append((float(p[1]), float(p[2]) if len(p) > 2 else 0.0))
            elif k in ("o", "g"):
                if meshes[-1]:
                    meshes.append([])
            elif k == "f":
                corners = []
                for tok in p[1:]:
                    parts = tok.split("/")
                    vi = int(parts[0])
                    ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                    ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                    pos = v[vi - 1] if vi > 0 else v[vi]
                    nrm = (vn[ni - 1] if ni > 0 else vn[ni]) if ni != 0 else (0.0, 0.0, 0.0)
                    uv = (vt[ti - 1] if ti > 0 else vt[ti]) if ti != 0 else (0.0, 0.0)
                    corners.append(pos + nrm + uv)
                for i in range(1, len(corners) - 1):
                    meshes[-1].append(corners[0] + corners[i] + corners[i + 1])
    meshes = [m for m in meshes if m]
    sizes = [len(m) for m in meshes]
    tris = np.array([t for m in meshes for t in m], dtype=np.float32).reshape(-1, 24)
    return tris, sizes
