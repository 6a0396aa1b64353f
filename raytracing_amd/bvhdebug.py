"""BVH debugging aids of the reference, headless: its text dump of the flat float arrays and its wireframe view of
the node boxes.

* `dump_flat_floats` reproduces SimplifiedData::DumpFlatFloatDataString (Program/Utils.cpp:33-50) character for
  character: every float through the stream's default formatting (`%g`, 6 significant digits), an `f` suffix, a comma
  after all but the last, `std::setw(12)` applied to whatever is written next (the next number, or the line break
  after every 40th), which is how GLSL/GPURayTrace/testScene.glsl:4-16 was produced.
* `wireframe_boxes` lists the boxes BVHSettings::RenderVisualization would draw (Program/BVHUI.hpp:30-103) for a
  depth window; `write_wireframe_obj` writes them as an OBJ of line segments, one group per depth, for viewing or
  diffing two trees.  The reference's two variants differ and both quirks are kept: the pointer tree counts depth
  from 0 at the root and always draws leaf boxes (red, :40-46); the flat tree's "depth" is the height of its traversal
  stack when the node is popped (:70, so the root has depth 1 and a right child that is popped while its sibling is
  still stacked is one deeper than that sibling) and leaves are drawn only with `show_leaf` (:81-89).
"""
import numpy as np

NODE_INTERNAL, NODE_LEAF, NODE_MESH, NODE_SPHERE = 0, 1, 2, 4


def _stream_float(x):
    s = "%g" % float(x)
    return s


def dump_flat_floats(data):
    data = np.asarray(data, dtype=np.float32).ravel()
    out = []
    width = 0  # pending std::setw for the next insertion
    for i, x in enumerate(data):
        out.append(_stream_float(x).rjust(width))
        width = 0
        out.append("f")
        if i < len(data) - 1:
            out.append(",")
        width = 12
        if (i + 1) % 40 == 0:
            out.append("\n".rjust(width))
            width = 0
    return "".join(out)


def wireframe_boxes(scene, min_depth=0, max_depth=40, show_leaf=False):
    """[(pmin, pmax, depth, is_leaf)] in the reference's drawing order.  `scene.order` tells the variant: 0 = flat
    tree (stack walk, right child popped first), 1 = pointer tree (recursion, left child first)."""
    nodes = scene.nodes()
    out = []
    if len(nodes) == 0:
        return out
    if scene.order == 0:
        stack = [scene.root]
        while stack:
            depth = len(stack)
            n = nodes[stack.pop()]
            if int(n["flags"]) == NODE_LEAF:
                if show_leaf:
                    out.append((n["pmin"].copy(), n["pmax"].copy(), depth, True))
                continue
            if min_depth <= depth <= max_depth:
                out.append((n["pmin"].copy(), n["pmax"].copy(), depth, False))
            stack += [int(n["left"]), int(n["right"])]
        return out
    todo = [(scene.root, 0)]
    while todo:
        i, depth = todo.pop()
        n = nodes[i]
        kind = int(n["flags"])
        if kind == NODE_INTERNAL:
            if min_depth <= depth <= max_depth:
                out.append((n["pmin"].copy(), n["pmax"].copy(), depth, False))
            todo += [(int(n["right"]), depth + 1), (int(n["left"]), depth + 1)]
        else:  # an object: its own box, always (BVHUI.hpp:40-46); a mesh's inner tree is not descended into
            out.append((n["pmin"].copy(), n["pmax"].copy(), depth, True))
    return out


_EDGES = ((0, 1), (1, 3), (3, 2), (2, 0), (4, 5), (5, 7), (7, 6), (6, 4), (0, 4), (1, 5), (2, 6), (3, 7))


def write_wireframe_obj(path, boxes):
    """OBJ with 8 vertices and 12 `l` segments per box; interior boxes grouped by depth, leaf boxes in group `leaves`."""
    order = sorted(range(len(boxes)), key=lambda k: (boxes[k][3], boxes[k][2]))
    with open(path, "w") as fh:
        fh.write("# node boxes: interior = (0, 1, depth/8), leaves = red in the reference's view (Program/BVHUI.hpp:44,54)\n")
        last = None
        base = 1
        for k in order:
            lo, hi, depth, leaf = boxes[k]
            group = "leaves" if leaf else "depth_%d" % depth
            if group != last:
                fh.write("g %s\n" % group)
                last = group
            for c in range(8):
                p = [hi[a] if (c >> a) & 1 else lo[a] for a in range(3)]
                fh.write("v %.9g %.9g %.9g\n" % tuple(float(v) for v in p))
            for a, b in _EDGES:
                fh.write("l %d %d\n" % (base + a, base + b))
            base += 8
    return len(boxes)
