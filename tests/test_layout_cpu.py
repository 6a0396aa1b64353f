"""Host-only checks of what rt_upload_scene lays out for the device (rt_layout_scene; no GPU needed).

The fast walk's exactness argument rests on three properties of the fast tree, checked here on real scenes:
every leaf of the reference tree appears exactly once with its box kept bit for bit; every interior box is
the exact union of the boxes below it (so it contains them); and the pre-order links (skip, right child) are
consistent, so that "descend = next record, prune = skip" visits every leaf whose ancestors' boxes pass.
The reference-order array must be the reference's own tree in its own visiting order.
"""
import numpy as np

import os

from helpers import R, ROOT, WL, objparse

KIND_INTERNAL, KIND_TRI, KIND_MESH, KIND_SPHERE = 0, 1, 2, 4


def u32(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def b1(x):
    """uint32 bit pattern of one float32 as a Python int."""
    return int(np.asarray(x, dtype=np.float32).reshape(1).view(np.uint32)[0])


def reference_order(scene):
    """Positions of the scene's nodes in the reference's visiting order (right child first for the flat scene,
    SimplifiedData.cpp:458-459; left first for the pointer scene, BVH.hpp:143-144)."""
    nodes = scene.nodes()
    order, stack = [], [scene.root]
    while stack:
        ni = stack.pop()
        order.append(ni)
        f = nodes["flags"][ni]
        if f == KIND_INTERNAL:
            kids = (nodes["left"][ni], nodes["right"][ni])
            stack.extend(kids if scene.order == 0 else kids[::-1])
        elif f == KIND_MESH:
            stack.append(nodes["left"][ni])
    return order


def check_fast_tree(fast, lo, hi, leaves_seen):
    """Recursively checks the pre-order subtree whose root is record `lo`; returns (bmin, bmax, end)."""
    skip = b1(fast[lo, 6]) // 32
    word = b1(fast[lo, 7])
    bmin, bmax = fast[lo, 0:3], fast[lo, 3:6]
    assert lo < skip <= hi
    kind = word & 7
    if kind == KIND_INTERNAL:
        right = word // 32
        assert word % 32 == 0 and lo + 1 < right < skip
        lmin, lmax, lend = check_fast_tree(fast, lo + 1, right, leaves_seen)
        assert lend == right
        rmin, rmax, rend = check_fast_tree(fast, right, skip, leaves_seen)
        assert rend == skip
        # exact union: bit-equal to the float min / max of the children
        assert np.array_equal(u32(bmin), u32(np.minimum(lmin, rmin)))
        assert np.array_equal(u32(bmax), u32(np.maximum(lmax, rmax)))
    elif kind == KIND_MESH:
        leaves_seen.append(("mesh", word >> 3, bmin.copy(), bmax.copy()))
        if skip > lo + 1:  # the mesh's own tree follows inline; its boxes need not lie inside the mesh's scene-level box
            _, _, iend = check_fast_tree(fast, lo + 1, skip, leaves_seen)
            assert iend == skip
    else:
        assert skip == lo + 1
        leaves_seen.append(("tri" if kind == KIND_TRI else "sph", word >> 3, bmin.copy(), bmax.copy()))
    return bmin, bmax, skip


def check_scene(scene):
    lay = R.device_layout(scene)
    nodes = scene.nodes()
    order = reference_order(scene)
    ref, fast = lay["ref"], lay["fast"]
    # ---- reference-order array: the reference's own boxes in its own visiting order, terminator at the end
    assert len(ref) == len(order) + 1
    for p in (0, 1, len(order) // 2, len(order) - 1):
        nd = nodes[order[p]]
        assert np.array_equal(u32(ref[p, 0:3]), u32(nd["pmin"])) and np.array_equal(u32(ref[p, 3:6]), u32(nd["pmax"]))
    assert np.array_equal(u32(ref[:-1, 0:3]), u32(nodes["pmin"][order])) and np.array_equal(u32(ref[:-1, 3:6]), u32(nodes["pmax"][order]))
    assert b1(ref[-1, 7]) == 0xFFFFFFFF
    ref_kind = u32(ref[:-1, 7]) & 7
    assert np.array_equal(ref_kind, nodes["flags"][order].astype(np.uint32))
    skips = u32(ref[:-1, 6]) // 32
    assert skips[0] == len(order) and np.all(skips > np.arange(len(order))) and np.all(skips <= len(order))
    # ---- fast tree
    seen = []
    _, _, end = check_fast_tree(fast, 0, len(fast), seen)
    assert end == len(fast)
    pos_of = {ni: p for p, ni in enumerate(order)}
    leaf_nodes = [ni for ni in order if nodes["flags"][ni] != KIND_INTERNAL]
    assert len(seen) == len(leaf_nodes)
    tri_pos, sph = lay["tri_pos"], lay["spheres"]
    got_positions = []
    for kind, ref_or_slot, bmin, bmax in seen:
        if kind == "mesh":
            p = ref_or_slot
        elif kind == "tri":
            p = b1(tri_pos[ref_or_slot, 11])
            ni = order[p]
            assert b1(tri_pos[ref_or_slot, 9]) == nodes["left"][ni]  # the triangle this leaf holds
        else:
            p = b1(sph[ref_or_slot, 7])
        got_positions.append(p)
        nd = nodes[order[p]]
        # the leaf keeps the reference's own leaf box, bit for bit
        assert np.array_equal(u32(bmin), u32(nd["pmin"])) and np.array_equal(u32(bmax), u32(nd["pmax"]))
    assert sorted(got_positions) == sorted(pos_of[ni] for ni in leaf_nodes)  # every reference leaf exactly once
    # triangle records: every slot once -- in fast-tree leaf order, or, where the scene has a wide tree, in that tree's order
    # (check_wide) --, edges = the test's own subtractions
    tris = scene.tris()
    tri_slots = [s for k, s, _, _ in seen if k == "tri"]
    if len(lay["wide"]):
        assert sorted(tri_slots) == list(range(len(tri_slots)))
    else:
        assert tri_slots == list(range(len(tri_slots)))
    for s in (0, len(tri_slots) // 3, len(tri_slots) - 1) if tri_slots else ():
        t = tris[b1(tri_pos[s, 9])]
        assert np.array_equal(u32(tri_pos[s, 0:3]), u32(t["pos"][0]))
        assert np.array_equal(u32(tri_pos[s, 3:6]), u32(t["pos"][1] - t["pos"][0]))
        assert np.array_equal(u32(tri_pos[s, 6:9]), u32(t["pos"][2] - t["pos"][0]))
    return lay, seen


def check_grid(lay, flat):
    """The grid records of a flat scene against its fast tree: same order and links, leaves with their slots, every box
    enclosed by its six planes with at least half a grid step to spare and by no more than three (the planes are the next
    grid lines but one), leaf records = triangle + the leaf's own box bit for bit."""
    fast, grid = lay["fast"], lay["grid"]
    if not flat:
        assert len(grid) == 0 and len(lay["leaf_records"]) == 0
        return
    assert len(grid) == len(fast) and len(lay["leaf_records"]) == len(lay["tri_pos"])
    origin, step = lay["grid_origin"].astype(np.float64), lay["grid_step"].astype(np.float64)
    assert np.all(step > 0)
    q_lo = np.stack([grid[:, k] & 0xFFFF for k in range(3)], 1).astype(np.float64)
    q_hi = np.stack([grid[:, k] >> 16 for k in range(3)], 1).astype(np.float64)
    lo, hi = origin + q_lo * step, origin + q_hi * step
    bmin, bmax = fast[:, 0:3].astype(np.float64), fast[:, 3:6].astype(np.float64)
    assert np.all(lo <= bmin - 0.5 * step) and np.all(hi >= bmax + 0.5 * step)
    assert np.all(lo >= bmin - 3.0 * step) and np.all(hi <= bmax + 3.0 * step)
    skip, word = u32(fast[:, 6]), u32(fast[:, 7])
    leaf = (word & 7) == KIND_TRI
    assert np.array_equal(grid[leaf, 3], ((word[leaf] >> 3) << 4) | 1)
    assert np.array_equal(grid[~leaf, 3], (skip[~leaf] // 32) * 16)
    rec, tp = lay["leaf_records"], lay["tri_pos"]
    slots = word[leaf] >> 3
    assert np.array_equal(u32(rec[slots, 0:9]), u32(tp[slots, 0:9])) and np.array_equal(u32(rec[slots, 11]), u32(tp[slots, 11]))
    box = np.concatenate([rec[slots, 9:11], rec[slots, 12:16]], 1)
    assert np.array_equal(u32(box), u32(fast[leaf, 0:6]))
    # the grid covers the scene: plane 0 and plane 65535 lie outside the root box on every axis
    assert np.all(origin < bmin[0]) and np.all(origin + 65535.0 * step > bmax[0])


def decode_wide(wide, origin, step):
    """Per node: (n children, n interior, first interior child, first leaf slot, axis, lo[8,3], hi[8,3]) with the planes as
    world coordinates, as the walk places them: origin + (org + (p << shift)) * step."""
    out = []
    for rec in wide:
        planes = rec[:12].astype(np.uint32)
        org = np.array([rec[12] & 0xFFFF, rec[12] >> 16, rec[13] & 0xFFFF], dtype=np.int64)
        sh = np.array([(rec[13] >> 16) & 15, (rec[13] >> 20) & 15, (rec[13] >> 24) & 15], dtype=np.int64)
        axis = int((rec[13] >> 28) & 3)
        n_int, n_all = int((rec[15] >> 24) & 15), int(rec[15] >> 28)
        lo = np.zeros((8, 3)); hi = np.zeros((8, 3))
        for k in range(3):
            for i in range(8):
                l8 = (int(planes[4 * k + (i >> 2)]) >> (8 * (i & 3))) & 255
                h8 = (int(planes[4 * k + 2 + (i >> 2)]) >> (8 * (i & 3))) & 255
                lo[i, k] = float(origin[k]) + float(org[k] + (l8 << sh[k])) * float(step[k])
                hi[i, k] = float(origin[k]) + float(org[k] + (h8 << sh[k])) * float(step[k])
        out.append((n_all, n_int, int(rec[14]), int(rec[15] & 0xFFFFFF), axis, lo, hi))
    return out


def check_wide(lay):
    """The wide form of the fast tree: every triangle slot is the leaf of exactly one node, a node's leaves are consecutive
    slots, every node but the root is the interior child of exactly one node, the tree is at most 16 levels deep, and the
    box of every child -- as the walk decodes it -- encloses every leaf box below it with at least a unit and a half of the
    grid to spare."""
    wide, rec = lay["wide"], lay["leaf_records"]
    assert len(wide) > 0
    step = lay["grid_step"].astype(np.float64)
    nodes = decode_wide(wide, lay["grid_origin"].astype(np.float64), step)
    leaf_lo = np.concatenate([rec[:, 9:11], rec[:, 12:13]], 1).astype(np.float64)
    leaf_hi = rec[:, 13:16].astype(np.float64)
    slot_seen = np.zeros(len(rec), dtype=np.int32)
    node_seen = np.zeros(len(wide), dtype=np.int32)
    node_seen[0] = 1

    def visit(ni, depth):
        """exact union of the leaf boxes below node ni"""
        n_all, n_int, child_base, leaf_base, axis, lo, hi = nodes[ni]
        assert 1 <= n_all <= 8 and n_int <= n_all and axis < 3 and depth <= 16
        ulo, uhi = np.full(3, np.inf), np.full(3, -np.inf)
        for i in range(n_all):
            if i < n_int:
                node_seen[child_base + i] += 1
                clo, chi = visit(child_base + i, depth + 1)
            else:
                s = leaf_base + (i - n_int)
                slot_seen[s] += 1
                clo, chi = leaf_lo[s], leaf_hi[s]
            assert np.all(lo[i] <= clo - 1.5 * step) and np.all(hi[i] >= chi + 1.5 * step), (ni, i)
            ulo, uhi = np.minimum(ulo, clo), np.maximum(uhi, chi)
        return ulo, uhi

    import sys
    sys.setrecursionlimit(10000)
    visit(0, 0)
    assert np.all(slot_seen == 1) and np.all(node_seen == 1)



def test_grid_records_of_flat_scenes():
    for scene in (R.SdScene().load_obj(WL.DRAGON_2426).finalize(), R.SdScene().load_obj(WL.DRAGON_18520).finalize()):
        lay = R.device_layout(scene)
        check_grid(lay, True)
        check_wide(lay)
    tris, _ = objparse.load_obj(WL.CYLINDER)
    # a mesh on a ground quad a hundred times its size; a scene of no thickness; a scene far from the origin
    lay = R.device_layout(R.SdScene().add_model(tris).add_model(WL.ground_quad(y=-1.0, x0=-200.0, x1=200.0, z0=-200.0, z1=200.0)).finalize())
    check_grid(lay, True)
    check_wide(lay)
    flat = R.device_layout(R.SdScene().add_model(WL.ground_quad()).finalize())
    check_grid(flat, True)
    check_wide(flat)
    assert flat["grid_step"][1] * 65530.0 >= (flat["fast"][0, 3:6] - flat["fast"][0, 0:3]).max() / 8.0 * 0.999  # thin axis: an eighth of the largest extent
    far = tris.copy().reshape(-1, 3, 8)
    far[:, :, 0:3] += np.float32(300.0)
    lay = R.device_layout(R.SdScene().add_model(far.reshape(-1, 24)).finalize())
    check_grid(lay, True)
    check_wide(lay)
    one = R.device_layout(R.SdScene().add_model(WL.ground_quad()[:1]).finalize())   # a scene of one triangle: a root with one leaf
    check_wide(one)
    assert len(one["wide"]) == 1 and one["wide"][0, 15] >> 28 == 1


def _slab_pass(lo, hi, o, inv):
    """The walk's slab test with closest = +inf in fp32 (fmaxf / fminf drop a NaN operand), boxes (n, 3), one ray."""
    f = np.float32
    with np.errstate(invalid="ignore", over="ignore"):
        t0, t1 = ((lo - o).astype(f) * inv).astype(f), ((hi - o).astype(f) * inv).astype(f)
    neg = inv < 0
    near, far = np.where(neg, t1, t0), np.where(neg, t0, t1)
    tmin = np.fmax(near[:, 2], np.fmax(near[:, 1], np.fmax(near[:, 0], f(1e-6))))
    tmax = np.fmin(far[:, 2], np.fmin(far[:, 1], np.fmin(far[:, 0], f(np.inf))))
    return ~(tmax <= tmin)


def _grid_pass(q_lo, q_hi, origin, step, o, inv):
    """The grid walk's test as the kernel evaluates it (grid_axis / grid_box_missed in rt_walks.hpp): per axis
    t = fma(q, A, B) with A = step * inv, B = (origin - o) * inv, or, for |inv| > 1e25, A = +-1e30, B = -((o - origin) / step) * A."""
    f = np.float32
    A, B = np.zeros(3, f), np.zeros(3, f)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        for k in range(3):
            if abs(inv[k]) > 1e25:
                A[k] = f(np.copysign(f(1e30), inv[k]))
                B[k] = f(-f(f(o[k] - origin[k]) / step[k]) * A[k])
            else:
                A[k] = f(step[k] * inv[k])
                B[k] = f(f(origin[k] - o[k]) * inv[k])
        fma = lambda q: (q.astype(np.float64) * A.astype(np.float64) + B.astype(np.float64)).astype(f)  # products of floats are exact in double
        t0, t1 = fma(q_lo), fma(q_hi)
    neg = inv < 0
    near, far = np.where(neg, t1, t0), np.where(neg, t0, t1)
    tmin = np.fmax(near[:, 2], np.fmax(near[:, 1], np.fmax(near[:, 0], f(1e-6))))
    tmax = np.fmin(far[:, 2], np.fmin(far[:, 1], np.fmin(far[:, 0], f(np.inf))))
    return ~(tmax <= tmin)


def test_grid_walk_never_prunes_what_the_exact_walk_keeps():
    """Conservativeness of the grid records under the kernel's own fp32 arithmetic: whenever a ray meets a record's exact box
    it meets the record's grid box -- for rays from inside the scene, from a camera up to 32 grid extents away, along the
    coordinate axes (infinite 1 / d) and nearly along them."""
    rng = np.random.default_rng(5)
    tris, _ = objparse.load_obj(WL.CYLINDER)
    scenes = [R.SdScene().load_obj(WL.DRAGON_2426).finalize(),
              R.SdScene().add_model(tris).add_model(WL.ground_quad(y=-1.0, x0=-200.0, x1=200.0, z0=-200.0, z1=200.0)).finalize()]
    f = np.float32
    for scene in scenes:
        lay = R.device_layout(scene)
        fast, grid = lay["fast"], lay["grid"]
        origin, step = lay["grid_origin"], lay["grid_step"]
        q_lo = np.stack([grid[:, k] & 0xFFFF for k in range(3)], 1).astype(f)
        q_hi = np.stack([grid[:, k] >> 16 for k in range(3)], 1).astype(f)
        lo, hi = fast[:, 0:3], fast[:, 3:6]
        root_lo, root_hi = lo[0].astype(np.float64), hi[0].astype(np.float64)
        ext = (step.astype(np.float64) * 65536.0)
        kept = pruned_by_grid = 0
        for i in range(300):
            if i % 3 == 0:      # from a surface point towards anywhere
                j = rng.integers(len(fast))
                o = (lo[j] + (hi[j] - lo[j]) * rng.random(3)).astype(f)
            elif i % 3 == 1:    # from a camera far outside, towards the scene
                o = (root_lo + ext * rng.uniform(-30.0, 31.0, 3)).astype(f)
            else:               # from just outside the root box
                o = (root_lo + (root_hi - root_lo) * rng.uniform(-0.2, 1.2, 3)).astype(f)
            target = root_lo + (root_hi - root_lo) * rng.random(3)
            d = (target - o.astype(np.float64)) if i % 3 else rng.normal(size=3)
            d = (d / np.linalg.norm(d)).astype(f)
            if i % 10 == 7:
                d[rng.integers(3)] = f(0.0)                      # along a coordinate plane
            if i % 10 == 8:
                d = np.zeros(3, f); d[rng.integers(3)] = f(rng.choice([-1.0, 1.0]))  # along an axis
            if i % 10 == 9:
                d[rng.integers(3)] = f(rng.choice([-1.0, 1.0]) * 1e-30)   # 1 / d overflows the products
            with np.errstate(divide="ignore"):
                inv = (f(1.0) / d).astype(f)
            exact = _slab_pass(lo, hi, o, inv)
            gridp = _grid_pass(q_lo, q_hi, origin, step, o, inv)
            assert not np.any(exact & ~gridp), (i, o, d)
            kept += int(exact.sum())
            pruned_by_grid += int((~gridp).sum())
        assert kept > 1000 and pruned_by_grid > 100 * 300    # the rays do meet the scene, and the grid does prune


def _wide_leaves_met(wide, origin, step, o, inv):
    """The wide walk without a distance limit, as the kernel evaluates it (wide_walk_from in rt_walks.hpp): per node the
    frame a = ldexp(A, shift), b = fma(org, A, B) on the ray's grid slopes (grid_axis), per child t = fma(plane byte, a, b),
    slab test against (1e-6, +inf).  Returns the set of triangle slots whose leaves the walk notes."""
    f = np.float32
    A, B = np.zeros(3, f), np.zeros(3, f)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        for k in range(3):
            if abs(inv[k]) > 1e25:
                A[k] = f(np.copysign(f(1e30), inv[k]))
                B[k] = f(-f(f(o[k] - origin[k]) / step[k]) * A[k])
            else:
                A[k] = f(step[k] * inv[k])
                B[k] = f(f(origin[k] - o[k]) * inv[k])
    fma = lambda x, a, b: (np.asarray(x, np.float64) * np.float64(a) + np.float64(b)).astype(f)
    met, stack = set(), [0]
    with np.errstate(invalid="ignore", over="ignore"):
        while stack:
            rec = wide[stack.pop()]
            planes = rec[:12].astype(np.uint32)
            org = (float(rec[12] & 0xFFFF), float(rec[12] >> 16), float(rec[13] & 0xFFFF))
            sh = (int((rec[13] >> 16) & 15), int((rec[13] >> 20) & 15), int((rec[13] >> 24) & 15))
            n_int, n_all = int((rec[15] >> 24) & 15), int(rec[15] >> 28)
            tmin, tmax = np.full(8, f(1e-6)), np.full(8, f(np.inf))
            for k in range(3):
                a, b = f(np.ldexp(A[k], sh[k])), fma(org[k], A[k], B[k])
                lo8 = np.array([(int(planes[4 * k + (i >> 2)]) >> (8 * (i & 3))) & 255 for i in range(8)], f)
                hi8 = np.array([(int(planes[4 * k + 2 + (i >> 2)]) >> (8 * (i & 3))) & 255 for i in range(8)], f)
                t0, t1 = fma(lo8, a, b), fma(hi8, a, b)
                near, far = (t1, t0) if A[k] < 0 else (t0, t1)
                tmin, tmax = np.fmax(tmin, near), np.fmin(tmax, far)
            hit = tmax > tmin
            for i in range(n_all):
                if hit[i]:
                    if i < n_int:
                        stack.append(int(rec[14]) + i)
                    else:
                        met.add(int(rec[15] & 0xFFFFFF) + i - n_int)
    return met


def test_wide_walk_never_prunes_a_leaf_the_ray_meets():
    """Conservativeness of the wide records under the kernel's own fp32 arithmetic (two roundings per plane: the node's
    frame, then the plane): every leaf whose EXACT box a ray meets is among the leaves the wide walk notes -- rays from the
    surface, from a camera up to 30 grid extents away, along the coordinate axes and nearly along them."""
    rng = np.random.default_rng(11)
    tris, _ = objparse.load_obj(WL.CYLINDER)
    scenes = [R.SdScene().load_obj(WL.DRAGON_2426).finalize(),
              R.SdScene().add_model(tris).add_model(WL.ground_quad(y=-1.0, x0=-200.0, x1=200.0, z0=-200.0, z1=200.0)).finalize()]
    f = np.float32
    for scene in scenes:
        lay = R.device_layout(scene)
        wide, rec = lay["wide"], lay["leaf_records"]
        assert len(wide) > 0
        origin, step = lay["grid_origin"], lay["grid_step"]
        leaf_lo = np.concatenate([rec[:, 9:11], rec[:, 12:13]], 1).astype(f)
        leaf_hi = rec[:, 13:16].astype(f)
        root_lo, root_hi = leaf_lo.min(0).astype(np.float64), leaf_hi.max(0).astype(np.float64)
        ext = step.astype(np.float64) * 65536.0
        met_exact = met_wide = 0
        for i in range(200):
            if i % 3 == 0:      # from a point of a leaf box towards anywhere
                j = rng.integers(len(rec))
                o = (leaf_lo[j] + (leaf_hi[j] - leaf_lo[j]) * rng.random(3)).astype(f)
            elif i % 3 == 1:    # from a camera far outside, towards the scene
                o = (root_lo + ext * rng.uniform(-30.0, 31.0, 3)).astype(f)
            else:               # from just outside the root box
                o = (root_lo + (root_hi - root_lo) * rng.uniform(-0.2, 1.2, 3)).astype(f)
            target = root_lo + (root_hi - root_lo) * rng.random(3)
            d = (target - o.astype(np.float64)) if i % 3 else rng.normal(size=3)
            d = (d / np.linalg.norm(d)).astype(f)
            if i % 10 == 7:
                d[rng.integers(3)] = f(0.0)
            if i % 10 == 8:
                d = np.zeros(3, f); d[rng.integers(3)] = f(rng.choice([-1.0, 1.0]))
            if i % 10 == 9:
                d[rng.integers(3)] = f(rng.choice([-1.0, 1.0]) * 1e-30)
            with np.errstate(divide="ignore"):
                inv = (f(1.0) / d).astype(f)
            exact = set(np.nonzero(_slab_pass(leaf_lo, leaf_hi, o, inv))[0].tolist())
            noted = _wide_leaves_met(wide, origin, step, o, inv)
            assert exact <= noted, (i, o, d, sorted(exact - noted)[:5])
            met_exact += len(exact)
            met_wide += len(noted)
        assert met_exact > 300 and met_wide < 40 * met_exact + 200 * 64  # the rays meet the scene, and the wide boxes prune


def test_pointer_scenes_have_no_grid_records():
    check_grid(R.device_layout(WL.build_old_scene(WL.rtiow_spheres()).build()), False)
    tris, _ = objparse.load_obj(WL.CYLINDER)
    sc = WL.build_old_scene(WL.small_sphere_scene())
    sc.add_mesh(tris, R.MAT_LAMBERTIAN, (0.8, 0.3, 0.3, 1.0), 0.0)
    check_grid(R.device_layout(sc.build()), False)


def test_flat_scene_layout_dragon2426():
    lay, seen = check_scene(R.SdScene().load_obj(WL.DRAGON_2426).finalize())
    assert len(seen) == 2426 and len(lay["fast"]) == 2 * 2426 - 1


def test_flat_scene_layout_quirk_tree_dragon18520():
    """The reference tree of this mesh has every big box stretched to z = 0 (FLT_MIN as initial pMax,
    SimplifiedData.hpp:64-69); the fast tree above the same leaves does not."""
    sc = R.SdScene().load_obj(WL.DRAGON_18520).finalize()
    lay, seen = check_scene(sc)
    nodes = sc.nodes()
    assert nodes["pmax"][sc.root][2] > 0.0 and nodes["pmax"][sc.root][2] < 1e-30   # the quirk itself
    assert lay["fast"][0, 5] < -0.8                                               # exact union: the dragon's own extent
    # surface-area measure of the expected slab tests per ray: two orders of magnitude apart
    def cost(rec):
        e = np.maximum(rec[:, 3:6] - rec[:, 0:3], 0)
        a = e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]
        return a.sum() / a.max()
    assert cost(lay["ref"][:-1]) > 50 * cost(lay["fast"])


def test_multi_mesh_flat_scene_and_field():
    tris, _ = objparse.load_obj(WL.CYLINDER)
    grid = WL.dragon_grid_tris(tris, 3)
    check_scene(R.SdScene().add_model(grid).finalize())
    t2 = tris.copy().reshape(-1, 3, 8)
    t2[:, :, 0] += 3.0
    check_scene(R.SdScene().add_model(tris).add_model(t2.reshape(-1, 24)).finalize())


def test_pointer_scene_layout_spheres_and_meshes():
    tris, _ = objparse.load_obj(WL.CYLINDER)
    sc = WL.build_old_scene(WL.small_sphere_scene())
    sc.add_mesh(tris, R.MAT_LAMBERTIAN, (0.8, 0.3, 0.3, 1.0), 0.0)
    t2 = tris.copy().reshape(-1, 3, 8)
    t2[:, :, 2] -= 4.0
    sc.add_mesh(t2.reshape(-1, 24), R.MAT_METAL, (0.8, 0.8, 0.9, 1.0), 0.1)
    sc.build()
    lay, seen = check_scene(sc)
    kinds = [k for k, _, _, _ in seen]
    assert kinds.count("sph") == 5 and kinds.count("mesh") == 2 and kinds.count("tri") == 2 * len(tris)
    check_scene(WL.build_old_scene(WL.rtiow_spheres()).build())


def test_layout_rejects_bad_trees():
    sc = R.SdScene().load_obj(WL.CYLINDER).finalize()
    nodes = sc.nodes().copy()
    nodes["left"][sc.root] = sc.root  # a cycle
    v = R._native.LayoutViewC()
    import ctypes as C
    rc = R._native.hip().rt_layout_scene(nodes.ctypes.data, len(nodes), sc.root, sc.tris().ctypes.data, len(sc.tris()),
                                         None, 0, None, 0, 0, C.byref(v))
    assert rc == -5


def test_fast_tree_is_the_same_whether_one_thread_or_many_build_it(tmp_path):
    """Large scenes share the top splits of the fast-tree build between threads; the nodes must not depend on it."""
    import subprocess
    exe = str(tmp_path / "fast_bvh_check")
    src = os.path.join(ROOT, "tests", "native", "fast_bvh_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", src, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok ") == 8
