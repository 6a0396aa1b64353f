// scene_layout.hpp: host-side construction of what rt_upload_scene puts into HBM (reference-order records, fast tree, primitive arrays) -- part of rt_hip.hip (included there, inside its unnamed namespace); see that file for the overall design.
#pragma once

// What rt_upload_scene puts into HBM, as host arrays (built without touching a device).
struct SceneLayout {
    std::vector<float4> dn, fn, tp, tn, sp, mt; // reference-order nodes, fast-tree nodes, triangle positions / normals, spheres, materials
    uint32_t n_ref = 0, n_fast = 0, n_tri_slots = 0, n_sph_slots = 0;
    double expected_visits = 0.0;
    bool nested = true; // every interior box of the caller's tree contains its children's boxes
    // Compact form of the fast tree, flat scenes only: one 16-byte record per node -- six 16-bit planes of a box that
    // CONTAINS the node's box with one grid unit to spare on every side, and a word: byte offset (in this array) of the
    // first record after the subtree for an interior node, (triangle slot << 4) | 1 for a leaf -- and the leaves' exact
    // boxes by triangle slot.  The walk over it only has to be conservative; the exact leaf test runs with the triangle test.
    std::vector<uint4> cn;
    std::vector<float4> lb; // 4 x float4 per triangle slot, all a leaf's tests need: p0.xyz, e1.x | e1.yz, e2.xy | e2.z, pMin.xy, pos | pMin.z, pMax.xyz
    float q_origin[3] = {0, 0, 0}, q_step[3] = {1, 1, 1};
    bool compact = false;
    double fast_visits = 0.0; // surface area of all fast-tree boxes / the root's: records a random ray through the root box meets
    double compact_inflation = 1.0; // surface area of all grid boxes / surface area of all exact boxes (visits of a random ray)
    // Wide form of the fast tree, flat scenes only: one 64-byte record per node with up to eight children, see build_wide.
    std::vector<uint4> wn;   // 4 x uint4 per node
    uint32_t n_wide = 0, wide_depth = 0;
    bool wide = false;
    double wide_visits = 0.0; // surface area of all wide nodes' boxes / the root's
    // Distance limit of the wide walk (see LeafShape): constants c1, 2 c2, c2 W + c3 of the three classes of regular leaves
    // (each class includes the ones below it), the class of every triangle slot (3: not regular, never skipped), and whether
    // the scene offers a limit at all.
    float prune_c1[3] = {0.f, 0.f, 0.f}, prune_c2x2[3] = {0.f, 0.f, 0.f}, prune_c3[3] = {0.f, 0.f, 0.f};
    std::vector<uint8_t> leaf_class;
    bool prune = false;
    uint32_t n_irregular = 0;
    // A cut through the fast tree: the boxes (pMin.xyz, pMax.xyz) of at most kCutBoxes subtrees that together hold every leaf.
    // A ray that misses all of them meets no leaf box; the camera stage projects them to find the tiles worth testing.
    std::vector<float> cut_boxes;
};
constexpr size_t kCutBoxes = 2048;

// ------------------------------------------------------------------------------------------------
// How far in front of its own leaf box can a triangle test report a hit?
//
// The fast walk may skip a subtree whose box the ray enters at distance e only if no leaf below it can turn out to be the
// reference's answer.  With tau the smallest hit distance found so far, a leaf can matter only as a candidate whose
// reported hit distance t is <= tau; its own box is entered at m >= e.  So it is enough to bound m - t over ALL rays for
// which the leaf's triangle test (sd::IntersectTriangle, SimplifiedData.cpp:269-307, evaluated in fp32 without fused
// multiply-add) reports a hit.  With u = 2^-24, a = |e1|_inf, b = |e2|_inf, W the largest extent of the leaf box,
// D = |d|_inf, S >= |o - p0|_inf, and the test's own guard |det| >= 1e-7 as the only lower bound on the determinant,
// forward error analysis of the test gives (first order in u, constants rounded up)
//     |t - t*| <= 82.7 S a b + 36.1 a b (a + b),     |u - u*| a + |v - v*| b <= (86.4 S a b + 36 a b (a + b)) D + 2.5e-7 (a + b),
// t*, u*, v* the exact values for the stored operands, provided a b D <= 0.0277.  The exact plane point
// p0 + u* e1 + v* e2 lies within |u - u*| a + |v - v*| b of the triangle, hence of the box, and along the axis on which the
// ray moves fastest the box's far plane is at most (W + that distance) / D beyond t*; the slab test passed, so m is
// below that far plane's computed distance.  Finally S <= m D + W, because the ray is inside the box at distance m and p0
// is a corner of the triangle.  Together:
//     m (1 - c2 D) <= t + c1 / D + c2 W + c3,    c1 = W + 3e-7 (a + b),  c2 = 170 a b,  c3 = 73 a b (a + b).
// The walk uses twice these constants (c1, c2, c3 below are the doubled ones, maxima over the scene's REGULAR leaves) and
// 1 / (1 - x) <= 1 + 2 x for x <= 1/2:  a subtree is skipped when  e > (tau * A + B) * (1 + 1e-5),  A = 1 + 2 c2 D,
// B = (c1 / D + c2 W + c3) * A,  for rays with 0.25 <= D <= 1.25.  A leaf is regular when its numbers are finite,
// a b <= 1e-3 (so c2 D <= 0.425), p0 + e1 and p0 + e2 lie inside its box, and its size is not far above the scene's
// typical leaf (a ground quad under a mesh is not regular).  Because the constants grow with the leaves' size, regular
// leaves come in three classes by size -- up to twice the median width (four times the median a b), up to twice the
// 90th percentile, up to four times the 99th (sixteen times for a b) -- each with the maxima over its own and the smaller
// classes; a node of the wide tree carries the largest class among the leaves below it and its children are skipped by
// that class's limit; class 3 (an irregular leaf below) is never skipped.
struct LeafShape {
    float a, b, w; // |e1|_inf, |e2|_inf, largest extent of the leaf box
    bool sane;     // finite, a b <= 1e-3, the triangle as stored lies inside its box
};
inline LeafShape leaf_shape(const rt_tri &t, const rt_node &leaf) {
    LeafShape s{0.f, 0.f, 0.f, true};
    for (int k = 0; k < 3; k++) {
        const float e1 = t.pos[1][k] - t.pos[0][k], e2 = t.pos[2][k] - t.pos[0][k];
        const float w = leaf.pmax[k] - leaf.pmin[k];
        s.a = std::fabs(e1) > s.a ? std::fabs(e1) : s.a;
        s.b = std::fabs(e2) > s.b ? std::fabs(e2) : s.b;
        s.w = w > s.w ? w : s.w;
        const double c1 = (double)t.pos[0][k] + (double)e1, c2 = (double)t.pos[0][k] + (double)e2;
        if (!(std::fabs((double)t.pos[0][k]) < 1e15) || !(std::fabs((double)e1) < 1e15) || !(std::fabs((double)e2) < 1e15) || !(w >= 0.f) ||
            !(w < 1e15f) || !(c1 >= leaf.pmin[k] && c1 <= leaf.pmax[k]) || !(c2 >= leaf.pmin[k] && c2 <= leaf.pmax[k]) ||
            !((double)t.pos[0][k] >= leaf.pmin[k] && (double)t.pos[0][k] <= leaf.pmax[k]))
            s.sane = false;
    }
    if (!((double)s.a * (double)s.b <= 1e-3))
        s.sane = false;
    return s;
}


// ------------------------------------------------------------------------------------------------
// The wide form of the fast tree (flat scenes).
//
// The binary fast tree is collapsed top-down into nodes of up to eight children.  Which binary nodes become the children
// of a wide node is chosen to make the sum of the wide nodes' surface areas -- the number of records a random ray is
// expected to fetch -- as small as this binary tree allows: C(q, i), the smallest such sum for the subtree of binary node q
// represented by at most i roots, follows from its children's by C(q, 1) = area(q) + min_k [C(l, k) + C(r, 8 - k)] and
// C(q, i) = min(C(q, i - 1), min_k [C(l, k) + C(r, i - k)]) (a leaf costs nothing: every leaf whose box a ray meets is tested
// whatever the grouping).  Against the greedy rule -- keep replacing the child of largest area by its own two children --
// a third fewer records and 6 % fewer fetched per ray on the 18 520-triangle dragon.  (`greedy`: that rule, kept for trees
// whose optimal grouping is deeper than the walk's stack.)
// One node is one 64-byte record, so a step of the walk is one trip to memory for eight slab tests instead of one:
//   bytes  0..47  the children's boxes, 8 bits per plane, as six arrays of eight: lo x, hi x, lo y, hi y, lo z, hi z;
//                 plane value p of axis k lies at grid unit  org_k + (p << shift_k)  of the 16-bit grid the compact tree
//                 uses (SceneLayout::q_origin / q_step), and the planes enclose the child's box with TWO grid units to
//                 spare (the compact tree's one, and one more for the second rounding of the two-step evaluation);
//   dword 12      org_x | org_y << 16
//   dword 13      org_z | shift_x << 16 | shift_y << 20 | shift_z << 24 | axis << 28 | class << 30: the largest class, under
//                 the distance limit, of the leaves below the node (LeafShape; 3 = no limit at this node)
//   dword 14      index of the node's first interior child; interior children are consecutive records
//   dword 15      first leaf slot (24 bits) | interior children << 24 (4 bits) | children << 28 (4 bits)
// Children are ordered along `axis` (the axis over which their centres spread most), interior ones first, leaves after
// them, each group by ascending centre: a ray that runs up that axis takes them from the front, one that runs down from
// the back.  Leaf slots are numbered in the order the nodes are made, a node's leaves consecutively, and every
// per-triangle array of the layout (tri_pos, tri_nrm, leaf_boxes) follows that numbering.
constexpr uint32_t kWideMaxDepth = 16;        // interior levels below the root the walk's stack can hold
constexpr uint32_t kWideMaxNodes = 1u << 23;  // a stack entry keeps a node index in 23 bits
constexpr uint32_t kWideMaxSlots = 1u << 24;

struct WideBuild {
    std::vector<uint4> wn;
    std::vector<uint32_t> slot_of; // binary node index of a leaf -> its slot
    uint32_t n_slots = 0, depth = 0;
    double area_sum = 0.0;
    bool ok = false;
};

// bn: the binary fast tree in pre-order (fastbvh::Node); q_origin / q_step: the grid.
// leaf_class: per binary node, for a leaf the class of its triangle under the distance limit (0..2, 3 = never skipped); may be empty.
inline void build_wide(const std::vector<fastbvh::Node> &bn, const float *q_origin, const float *q_step, const std::vector<uint8_t> &leaf_class,
                       WideBuild &W, bool greedy = false) {
    const uint32_t nb = (uint32_t)bn.size();
    W.ok = false;
    W.slot_of.assign(nb, kNone);
    W.wn.clear();
    if (nb == 0)
        return;
    // the largest leaf class in every binary subtree (children follow their parent in the array: one pass from the back)
    std::vector<uint8_t> tainted(nb, 0);
    if (!leaf_class.empty())
        for (uint32_t q = nb; q-- > 0;) {
            if (bn[q].item != 0xFFFFFFFFu)
                tainted[q] = leaf_class[q];
            else
                tainted[q] = std::max(tainted[q + 1u], tainted[q + 1u + bn[q + 1u].size]);
        }
    auto half_area = [&](uint32_t c) {
        const double x = (double)bn[c].bmax[0] - bn[c].bmin[0], y = (double)bn[c].bmax[1] - bn[c].bmin[1], z = (double)bn[c].bmax[2] - bn[c].bmin[2];
        if (!(x >= 0.0) || !(y >= 0.0) || !(z >= 0.0))
            return 0.0;
        return x * y + y * z + z * x;
    };
    auto is_leaf = [&](uint32_t c) { return bn[c].item != 0xFFFFFFFFu; };
    // 16-bit grid planes that enclose a coordinate with two units to spare
    auto qlo = [&](double p, int k) {
        const double q = std::floor((p - (double)q_origin[k]) / (double)q_step[k]) - 2.0;
        return (int64_t)(q < 0.0 ? 0.0 : q > 65535.0 ? 65535.0 : q);
    };
    auto qhi = [&](double p, int k) {
        const double q = std::ceil((p - (double)q_origin[k]) / (double)q_step[k]) + 2.0;
        return (int64_t)(q < 0.0 ? 0.0 : q > 65535.0 ? 65535.0 : q);
    };
    // cost[8 q + i], i = 1 .. 7: C(q, i); split[8 q + j - 1], j = 2 .. 8: the k that minimises C(l, k) + C(r, j - k)
    std::vector<float> cost;
    std::vector<uint8_t> split;
    if (!greedy) {
        cost.assign(8 * (size_t)nb, 0.0f);
        split.assign(8 * (size_t)nb, 0);
        for (uint32_t q = nb; q-- > 0;) {
            if (is_leaf(q))
                continue; // zeros
            const uint32_t l = q + 1u, r = l + bn[l].size;
            const float *cl = &cost[8 * (size_t)l], *cr = &cost[8 * (size_t)r];
            float dist[9];
            for (int j = 2; j <= 8; j++) {
                float best = 3.0e38f;
                int bk = 1;
                for (int k = 1; k < j; k++) {
                    const float v = cl[k < 7 ? k : 7] + cr[j - k < 7 ? j - k : 7];
                    if (v < best) {
                        best = v;
                        bk = k;
                    }
                }
                dist[j] = best;
                split[8 * (size_t)q + (size_t)(j - 1)] = (uint8_t)bk;
            }
            float *c = &cost[8 * (size_t)q];
            c[1] = (float)half_area(q) + dist[8];
            for (int i = 2; i <= 7; i++)
                c[i] = dist[i] < c[i - 1] ? dist[i] : c[i - 1];
        }
    }
    // the at most i binary nodes that represent the subtree of q at cost C(q, i), appended to ch
    std::function<void(uint32_t, int, uint32_t *, int &)> roots = [&](uint32_t q, int i, uint32_t *ch, int &n) {
        while (true) {
            if (is_leaf(q) || i == 1) {
                ch[n++] = q;
                return;
            }
            const uint32_t l = q + 1u, r = l + bn[l].size;
            const int k = split[8 * (size_t)q + (size_t)(i - 1)];
            const float d = cost[8 * (size_t)l + (size_t)(k < 7 ? k : 7)] + cost[8 * (size_t)r + (size_t)(i - k < 7 ? i - k : 7)];
            if (d < cost[8 * (size_t)q + (size_t)(i - 1)]) { // its two children's roots are cheaper than q with one root fewer
                roots(l, k, ch, n);
                roots(r, i - k, ch, n);
                return;
            }
            i--;
        }
    };
    struct Todo {
        uint32_t q, me, depth;
    };
    std::vector<Todo> todo;
    W.wn.assign(4, make_uint4(0u, 0u, 0u, 0u));
    todo.push_back(Todo{0u, 0u, 0u});
    uint32_t next_slot = 0, max_depth = 0;
    const double root_area = half_area(0);
    double area_sum = 0.0;
    while (!todo.empty()) {
        const Todo t = todo.back();
        todo.pop_back();
        if (t.depth > max_depth)
            max_depth = t.depth;
        if (t.depth > kWideMaxDepth)
            return; // deeper than the walk's stack: the scene keeps its compact tree
        area_sum += half_area(t.q);
        uint32_t ch[8];
        int n = 0;
        if (is_leaf(t.q)) {
            ch[n++] = t.q; // a scene of one triangle
        } else if (!greedy) {
            const uint32_t l = t.q + 1u, r = l + bn[l].size;
            const int k = split[8 * (size_t)t.q + 7];
            roots(l, k < 7 ? k : 7, ch, n);
            roots(r, 8 - k < 7 ? 8 - k : 7, ch, n);
        } else {
            ch[n++] = t.q + 1u;
            ch[n++] = t.q + 1u + bn[t.q + 1u].size;
            while (n < 8) {
                int best = -1;
                double ba = -1.0;
                for (int i = 0; i < n; i++)
                    if (!is_leaf(ch[i]) && half_area(ch[i]) > ba) {
                        ba = half_area(ch[i]);
                        best = i;
                    }
                if (best < 0)
                    break;
                const uint32_t c = ch[best];
                ch[best] = c + 1u;
                ch[n++] = c + 1u + bn[c + 1u].size;
            }
        }
        // the axis over which the children's centres spread most; interior children first, both groups by ascending centre
        double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                const double c = (double)bn[ch[i]].bmin[k] + (double)bn[ch[i]].bmax[k];
                clo[k] = c < clo[k] ? c : clo[k];
                chi[k] = c > chi[k] ? c : chi[k];
            }
        int axis = 0;
        for (int k = 1; k < 3; k++)
            if (chi[k] - clo[k] > chi[axis] - clo[axis])
                axis = k;
        std::stable_sort(ch, ch + n, [&](uint32_t a, uint32_t b) {
            if (is_leaf(a) != is_leaf(b))
                return !is_leaf(a);
            return (double)bn[a].bmin[axis] + (double)bn[a].bmax[axis] < (double)bn[b].bmin[axis] + (double)bn[b].bmax[axis];
        });
        int ni = 0;
        while (ni < n && !is_leaf(ch[ni]))
            ni++;
        const uint32_t child_base = (uint32_t)(W.wn.size() / 4), leaf_base = next_slot;
        if ((size_t)child_base + (size_t)ni > kWideMaxNodes || (size_t)leaf_base + (size_t)(n - ni) > kWideMaxSlots)
            return;
        if (ni > 0)
            W.wn.resize(W.wn.size() + 4 * (size_t)ni, make_uint4(0u, 0u, 0u, 0u));
        for (int i = ni; i < n; i++)
            W.slot_of[ch[i]] = next_slot++;
        // planes
        int64_t lo[8][3], hi[8][3], org[3], sh[3];
        for (int k = 0; k < 3; k++) {
            int64_t mn = 65535, mx = 0;
            for (int i = 0; i < n; i++) {
                lo[i][k] = qlo((double)bn[ch[i]].bmin[k], k);
                hi[i][k] = qhi((double)bn[ch[i]].bmax[k], k);
                mn = lo[i][k] < mn ? lo[i][k] : mn;
                mx = hi[i][k] > mx ? hi[i][k] : mx;
            }
            org[k] = mn;
            int64_t s = 0;
            while (((mx - mn) + ((int64_t)1 << s) - 1) >> s > 255)
                s++;
            sh[k] = s;
        }
        uint32_t planes[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // lo x (2 words), hi x, lo y, hi y, lo z, hi z
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                const int64_t l8 = (lo[i][k] - org[k]) >> sh[k];
                const int64_t h8 = ((hi[i][k] - org[k]) + ((int64_t)1 << sh[k]) - 1) >> sh[k];
                planes[4 * k + (i >> 2)] |= (uint32_t)l8 << (8 * (i & 3));
                planes[4 * k + 2 + (i >> 2)] |= (uint32_t)h8 << (8 * (i & 3));
                // the planes as the walk will place them must enclose the box with at least a unit and a half to spare
                // (they do unless a coordinate is not finite or lies off the grid)
                const double pl = (double)q_origin[k] + (double)(org[k] + (l8 << sh[k])) * (double)q_step[k];
                const double ph = (double)q_origin[k] + (double)(org[k] + (h8 << sh[k])) * (double)q_step[k];
                if (l8 < 0 || l8 > 255 || h8 < 0 || h8 > 255 || !(pl <= (double)bn[ch[i]].bmin[k] - 1.5 * (double)q_step[k]) ||
                    !(ph >= (double)bn[ch[i]].bmax[k] + 1.5 * (double)q_step[k]))
                    return;
            }
        uint4 *rec = &W.wn[4 * (size_t)t.me];
        rec[0] = make_uint4(planes[0], planes[1], planes[2], planes[3]);
        rec[1] = make_uint4(planes[4], planes[5], planes[6], planes[7]);
        rec[2] = make_uint4(planes[8], planes[9], planes[10], planes[11]);
        rec[3] = make_uint4((uint32_t)org[0] | ((uint32_t)org[1] << 16),
                            (uint32_t)org[2] | ((uint32_t)sh[0] << 16) | ((uint32_t)sh[1] << 20) | ((uint32_t)sh[2] << 24) | ((uint32_t)axis << 28) |
                                ((uint32_t)tainted[t.q] << 30),
                            child_base, leaf_base | ((uint32_t)ni << 24) | ((uint32_t)n << 28));
        for (int i = ni; i-- > 0;) // first child on top: nodes are made in depth-first order
            todo.push_back(Todo{ch[i], child_base + (uint32_t)i, t.depth + 1u});
    }
    W.n_slots = next_slot;
    W.depth = max_depth;
    W.area_sum = root_area > 0.0 ? area_sum / root_area : (double)(W.wn.size() / 4);
    W.ok = true;
}


int lfail(std::string &err, int code, const std::string &msg) {
    err = msg;
    return code;
}

int build_layout(const rt_node *nodes, uint32_t n_nodes, uint32_t root, const rt_tri *tris, uint32_t n_tris,
                 const rt_sphere *spheres, uint32_t n_spheres, const rt_material *mats, uint32_t n_mats, int order,
                 SceneLayout &L, std::string &err) {
    if (!nodes || n_nodes == 0 || root >= n_nodes)
        return lfail(err, RT_ERR_BAD_SCENE, "Build Failed. end - start <= 0 (empty node array or bad root)");
    if ((n_tris && !tris) || (n_spheres && !spheres) || (n_mats && !mats))
        return lfail(err, RT_ERR_INVALID_ARGUMENT, "null primitive array");
    if (order != RT_ORDER_RIGHT_FIRST && order != RT_ORDER_LEFT_FIRST)
        return lfail(err, RT_ERR_INVALID_ARGUMENT, "bad traversal order");
    if (n_tris >= (1u << 28) || n_spheres >= (1u << 28) || n_nodes >= (1u << 27))
        return lfail(err, RT_ERR_BAD_SCENE, "too many nodes or primitives");

    // 1. Walk the tree once in the reference's visiting order and record that order.
    std::vector<uint32_t> order_of; // emission position -> node index
    std::vector<uint32_t> pos_of(n_nodes, kNone);
    order_of.reserve(n_nodes);
    std::vector<uint32_t> stack;
    stack.push_back(root);
    while (!stack.empty()) {
        uint32_t ni = stack.back();
        stack.pop_back();
        if (ni >= n_nodes)
            return lfail(err, RT_ERR_BAD_SCENE, "node index out of range");
        if (pos_of[ni] != kNone)
            return lfail(err, RT_ERR_BAD_SCENE, "node referenced twice (not a tree)");
        pos_of[ni] = (uint32_t)order_of.size();
        order_of.push_back(ni);
        const rt_node &nd = nodes[ni];
        switch (nd.flags) {
        case RT_NODE_INTERNAL:
            if (order == RT_ORDER_RIGHT_FIRST) {
                stack.push_back(nd.left);
                stack.push_back(nd.right);
            } else {
                stack.push_back(nd.right);
                stack.push_back(nd.left);
            }
            break;
        case RT_NODE_LEAF:
            if (nd.left >= n_tris)
                return lfail(err, RT_ERR_BAD_SCENE, "triangle index out of range");
            break;
        case RT_NODE_SPHERE:
            if (nd.left >= n_spheres)
                return lfail(err, RT_ERR_BAD_SCENE, "sphere index out of range");
            break;
        case RT_NODE_MESH:
            stack.push_back(nd.left);
            break;
        default:
            return lfail(err, RT_ERR_BAD_SCENE, "unknown node kind");
        }
    }
    const uint32_t m = (uint32_t)order_of.size();
    // 2. Subtree sizes, children before parents (children are emitted after their parent).
    std::vector<uint32_t> size(m, 1);
    for (uint32_t p = m; p-- > 0;) {
        const rt_node &nd = nodes[order_of[p]];
        if (nd.flags == RT_NODE_INTERNAL)
            size[p] = 1 + size[pos_of[nd.left]] + size[pos_of[nd.right]];
        else if (nd.flags == RT_NODE_MESH)
            size[p] = 1 + size[pos_of[nd.left]];
    }
    // The fast walk reproduces the reference's result only if every interior box contains the boxes of its children
    // (then a leaf that passes its own slab test has ancestors that passed theirs).  The reference's builders
    // guarantee it; a caller's hand-made tree may not, and is then walked in reference order only.
    L.nested = true;
    for (uint32_t p = 0; p < m && L.nested; p++) {
        const rt_node &nd = nodes[order_of[p]];
        if (nd.flags != RT_NODE_INTERNAL)
            continue;
        for (const uint32_t child : {nd.left, nd.right}) {
            const rt_node &c = nodes[child];
            for (int k = 0; k < 3; k++)
                if (!(nd.pmin[k] <= c.pmin[k]) || !(nd.pmax[k] >= c.pmax[k]))
                    L.nested = false;
        }
    }
    // 3. The fast tree: a surface-area-heuristic hierarchy over the reference's leaf boxes (see fast_bvh.hpp).
    //    Scene-level leaves (triangles of the flat scene, spheres, mesh objects) form the top tree; the
    //    triangles of a mesh object form that mesh's own tree, emitted inline right after the mesh's leaf
    //    record, exactly as the reference nests Mesh::intersect inside the scene traversal.
    struct MeshTree {
        std::vector<fastbvh::Item> items;
        std::vector<fastbvh::Node> nodes;
    };
    std::vector<fastbvh::Item> top_items;
    std::vector<MeshTree> mesh_trees;
    std::vector<uint32_t> mesh_of(m, kNone); // reference position of a mesh leaf -> index into mesh_trees
    {
        uint32_t mesh_end = 0;
        for (uint32_t p = 0; p < m; p++) {
            const rt_node &nd = nodes[order_of[p]];
            if (nd.flags == RT_NODE_INTERNAL)
                continue;
            const bool inside = p < mesh_end;
            fastbvh::Item it;
            for (int k = 0; k < 3; k++) {
                it.bmin[k] = nd.pmin[k];
                it.bmax[k] = nd.pmax[k];
            }
            it.id = p;
            if (nd.flags == RT_NODE_LEAF) {
                (inside ? mesh_trees.back().items : top_items).push_back(it);
            } else {
                if (inside)
                    return lfail(err, RT_ERR_BAD_SCENE, "a mesh's inner tree may hold triangles only");
                top_items.push_back(it);
                if (nd.flags == RT_NODE_MESH) {
                    mesh_of[p] = (uint32_t)mesh_trees.size();
                    mesh_trees.emplace_back();
                    mesh_end = p + size[p];
                }
            }
        }
    }
    std::vector<fastbvh::Node> top_nodes;
    fastbvh::build(top_items, top_nodes);
    for (auto &tree : mesh_trees)
        fastbvh::build(tree.items, tree.nodes);

    // The cut for the camera stage: the root's subtree is replaced by its two children, largest box first, until kCutBoxes
    // boxes (or only leaves) are left.
    if (!top_nodes.empty()) {
        auto area = [&](uint32_t q) {
            const double x = (double)top_nodes[q].bmax[0] - top_nodes[q].bmin[0], y = (double)top_nodes[q].bmax[1] - top_nodes[q].bmin[1],
                         z = (double)top_nodes[q].bmax[2] - top_nodes[q].bmin[2];
            return x * y + y * z + z * x;
        };
        std::vector<std::pair<double, uint32_t>> heap; // (area, node): interior nodes still to split
        std::vector<uint32_t> cut;
        auto put = [&](uint32_t q) {
            if (top_nodes[q].item != 0xFFFFFFFFu) {
                cut.push_back(q);
            } else {
                heap.emplace_back(area(q), q);
                std::push_heap(heap.begin(), heap.end());
            }
        };
        put(0u);
        while (!heap.empty() && cut.size() + heap.size() < kCutBoxes) {
            std::pop_heap(heap.begin(), heap.end());
            const uint32_t q = heap.back().second;
            heap.pop_back();
            put(q + 1u);
            put(q + 1u + top_nodes[q + 1u].size);
        }
        for (const auto &h : heap)
            cut.push_back(h.second);
        L.cut_boxes.reserve(6 * cut.size());
        for (uint32_t q : cut) {
            for (int k = 0; k < 3; k++)
                L.cut_boxes.push_back(top_nodes[q].bmin[k]);
            for (int k = 0; k < 3; k++)
                L.cut_boxes.push_back(top_nodes[q].bmax[k]);
        }
    }

    // 3b. Flat scenes: the 16-bit grid over the root box that the compact and the wide form of the fast tree place their
    //     planes on, and the wide form's topology -- it decides the numbering of the triangle slots (see build_wide).
    const bool flat = n_spheres == 0 && mesh_trees.empty() && !top_nodes.empty();
    bool grid_ok = false;
    WideBuild wide;
    if (flat) {
        const double lo[3] = {top_nodes[0].bmin[0], top_nodes[0].bmin[1], top_nodes[0].bmin[2]};
        const double hi[3] = {top_nodes[0].bmax[0], top_nodes[0].bmax[1], top_nodes[0].bmax[2]};
        grid_ok = true;
        double ext_max = 0.0;
        for (int k = 0; k < 3; k++)
            ext_max = hi[k] - lo[k] > ext_max ? hi[k] - lo[k] : ext_max;
        for (int k = 0; k < 3; k++) {
            double ext = hi[k] - lo[k];
            if (!(ext >= 0.0) || !(ext < 1e30))
                grid_ok = false;
            // A thin scene keeps a grid of at least an eighth of its largest extent on every axis: the grid arithmetic's
            // error grows with the distance of a ray's origin in grid extents, and a camera stands off a flat scene by the
            // scene's size, not its thickness.
            if (ext < ext_max / 8.0)
                ext = ext_max / 8.0;
            float step = (float)(ext / 65530.0);
            if (!(step > 1e-30f))
                step = 1e-30f;
            L.q_step[k] = step;
            L.q_origin[k] = (float)(lo[k] - 2.0 * (double)step);
        }
        if (grid_ok) {
            // the leaves' shapes: which are regular, and the distance limit's constants over those (see above LeafShape)
            const uint32_t nbn = (uint32_t)top_nodes.size();
            std::vector<uint8_t> irregular(nbn, 0); // per binary node: a leaf's class (3 = irregular)
            std::vector<LeafShape> shape(nbn);
            std::vector<float> widths, products;
            for (uint32_t q = 0; q < nbn; q++)
                if (top_nodes[q].item != 0xFFFFFFFFu) {
                    const rt_node &leaf = nodes[order_of[top_items[top_nodes[q].item].id]];
                    shape[q] = leaf_shape(tris[leaf.left], leaf);
                    if (shape[q].sane) {
                        widths.push_back(shape[q].w);
                        products.push_back(shape[q].a * shape[q].b);
                    }
                }
            float w_cap[3] = {0.f, 0.f, 0.f}, ab_cap[3] = {0.f, 0.f, 0.f};
            if (!widths.empty()) {
                const size_t n = widths.size(), at[3] = {(n - 1) / 2, (n - 1) * 90 / 100, (n - 1) * 99 / 100};
                const float wf[3] = {2.f, 2.f, 4.f}, pf[3] = {4.f, 4.f, 16.f};
                for (int k = 0; k < 3; k++) {
                    std::nth_element(widths.begin(), widths.begin() + at[k], widths.end());
                    std::nth_element(products.begin(), products.begin() + at[k], products.end());
                    w_cap[k] = wf[k] * widths[at[k]];
                    ab_cap[k] = pf[k] * products[at[k]];
                    if (k > 0) { // (the caps never shrink from one class to the next)
                        w_cap[k] = std::max(w_cap[k], w_cap[k - 1]);
                        ab_cap[k] = std::max(ab_cap[k], ab_cap[k - 1]);
                    }
                }
            }
            double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0}, c3[3] = {0, 0, 0}, wmax[3] = {0, 0, 0};
            uint32_t n_irr = 0, n_leaf = 0;
            for (uint32_t q = 0; q < nbn; q++)
                if (top_nodes[q].item != 0xFFFFFFFFu) {
                    const LeafShape &sh = shape[q];
                    n_leaf++;
                    int cls = 3;
                    for (int k = 2; k >= 0; k--)
                        if (sh.sane && sh.w <= w_cap[k] && sh.a * sh.b <= ab_cap[k])
                            cls = k;
                    irregular[q] = (uint8_t)cls;
                    if (cls == 3) {
                        n_irr++;
                        continue;
                    }
                    const double a = sh.a, b = sh.b;
                    for (int k = cls; k < 3; k++) {
                        c1[k] = std::max(c1[k], 2.0 * ((double)sh.w + 3e-7 * (a + b)));
                        c2[k] = std::max(c2[k], 340.0 * a * b);
                        c3[k] = std::max(c3[k], 146.0 * a * b * (a + b));
                        wmax[k] = std::max(wmax[k], (double)sh.w);
                    }
                }
            L.n_irregular = n_irr;
            L.prune = n_leaf > 0 && (double)n_irr <= 0.02 * (double)n_leaf + 16.0 && c2[2] * 1.25 <= 0.5;
            for (int k = 0; k < 3; k++) {
                L.prune_c1[k] = (float)(c1[k] * 1.000001);
                L.prune_c2x2[k] = (float)(2.0 * c2[k] * 1.000001);
                L.prune_c3[k] = (float)((c2[k] * wmax[k] + c3[k]) * 1.000001); // c2 W + c3
            }
            build_wide(top_nodes, L.q_origin, L.q_step, irregular, wide);
            if (!wide.ok) // (the grouping of least area may be deeper than the greedy one)
                build_wide(top_nodes, L.q_origin, L.q_step, irregular, wide, true);
            if (wide.ok) { // the classes by triangle slot, for whoever wants to check the limit (rt_layout_scene)
                L.leaf_class.assign(wide.n_slots, 3);
                for (uint32_t q = 0; q < nbn; q++)
                    if (top_nodes[q].item != 0xFFFFFFFFu)
                        L.leaf_class[wide.slot_of[q]] = irregular[q];
            }
        }
    }

    // 4. Emit the fast tree, and with it the primitive arrays by triangle slot: slots follow the fast tree's leaf order,
    //    or, when the scene has a wide tree, that tree's (a wide node's leaves are consecutive slots).
    std::vector<float4> &fn = L.fn, &tp = L.tp, &tn = L.tn, &sp = L.sp, &mt = L.mt;
    mt.assign(2 * (size_t)n_mats, make_float4(0.f, 0.f, 0.f, 0.f));
    size_t n_tri_leaves = 0;
    for (uint32_t p = 0; p < m; p++)
        n_tri_leaves += nodes[order_of[p]].flags == RT_NODE_LEAF ? 1u : 0u;
    tp.assign(3 * n_tri_leaves, make_float4(0.f, 0.f, 0.f, 0.f));
    tn.assign(3 * n_tri_leaves, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<uint32_t> slot_of(m, 0); // reference position of a leaf -> its triangle / sphere slot
    uint32_t tri_slots = 0, sph_slots = 0;
    auto push_node = [&](const float *bmin, const float *bmax, uint32_t skip_index, uint32_t word) {
        fn.push_back(make_float4(bmin[0], bmin[1], bmin[2], bmax[0]));
        fn.push_back(make_float4(bmax[1], bmax[2], u2f(skip_index * 32u), u2f(word)));
    };
    auto emit_tri = [&](uint32_t p, uint32_t slot) -> int { // p = reference position of the leaf
        const rt_node &nd = nodes[order_of[p]];
        const rt_tri &t = tris[nd.left];
        if (n_mats && t.mat >= n_mats)
            return RT_ERR_BAD_SCENE;
        if ((size_t)slot >= n_tri_leaves)
            return RT_ERR_BAD_SCENE;
        slot_of[p] = slot;
        tri_slots++;
        // edges are the same fp32 subtractions the triangle test starts with (SimplifiedData.cpp:272-273)
        const float e1x = t.pos[1][0] - t.pos[0][0], e1y = t.pos[1][1] - t.pos[0][1], e1z = t.pos[1][2] - t.pos[0][2];
        const float e2x = t.pos[2][0] - t.pos[0][0], e2y = t.pos[2][1] - t.pos[0][1], e2z = t.pos[2][2] - t.pos[0][2];
        tp[3 * (size_t)slot] = make_float4(t.pos[0][0], t.pos[0][1], t.pos[0][2], e1x);
        tp[3 * (size_t)slot + 1] = make_float4(e1y, e1z, e2x, e2y);
        tp[3 * (size_t)slot + 2] = make_float4(e2z, u2f(nd.left), u2f((uint32_t)t.mat), u2f(p));
        tn[3 * (size_t)slot] = make_float4(t.nrm[0][0], t.nrm[0][1], t.nrm[0][2], t.nrm[1][0]);
        tn[3 * (size_t)slot + 1] = make_float4(t.nrm[1][1], t.nrm[1][2], t.nrm[2][0], t.nrm[2][1]);
        tn[3 * (size_t)slot + 2] = make_float4(t.nrm[2][2], 0.f, 0.f, 0.f);
        return RT_OK;
    };
    const uint32_t n_top = (uint32_t)top_nodes.size();
    std::vector<uint32_t> epos(n_top + 1, 0);
    for (uint32_t q = 0; q < n_top; q++) {
        const fastbvh::Node &fnode = top_nodes[q];
        epos[q] = (uint32_t)(fn.size() / 2);
        if (fnode.item == 0xFFFFFFFFu) {
            push_node(fnode.bmin, fnode.bmax, 0u, KIND_INTERNAL); // skip and the right-child offset are patched below
            continue;
        }
        const uint32_t p = top_items[fnode.item].id;
        const rt_node &nd = nodes[order_of[p]];
        if (nd.flags == RT_NODE_LEAF) {
            if (emit_tri(p, wide.ok ? wide.slot_of[q] : tri_slots) != RT_OK)
                return lfail(err, RT_ERR_BAD_SCENE, "triangle material out of range");
            push_node(fnode.bmin, fnode.bmax, 0u, (slot_of[p] << 3) | KIND_TRI);
        } else if (nd.flags == RT_NODE_SPHERE) {
            const rt_sphere &sph = spheres[nd.left];
            if (sph.mat >= n_mats)
                return lfail(err, RT_ERR_BAD_SCENE, "sphere material out of range");
            const uint32_t dielectric = mats[sph.mat].kind == RT_MAT_DIELECTRIC ? 1u : 0u;
            slot_of[p] = sph_slots++;
            sp.push_back(make_float4(sph.center[0], sph.center[1], sph.center[2], sph.radius));
            sp.push_back(make_float4(u2f(sph.mat), u2f(sph.prim_id), u2f(dielectric), u2f(p)));
            push_node(fnode.bmin, fnode.bmax, 0u, (slot_of[p] << 3) | KIND_SPHERE);
        } else { // mesh object: its leaf record carries the leaf's reference position, its own tree follows
            push_node(fnode.bmin, fnode.bmax, 0u, (p << 3) | KIND_MESH);
            const MeshTree &tree = mesh_trees[mesh_of[p]];
            const uint32_t base = (uint32_t)(fn.size() / 2);
            for (uint32_t j = 0; j < (uint32_t)tree.nodes.size(); j++) {
                const fastbvh::Node &in = tree.nodes[j];
                uint32_t word = KIND_INTERNAL;
                if (in.item == 0xFFFFFFFFu) { // interior: byte offset of the right child (low bits 0 = KIND_INTERNAL)
                    word = (base + j + 1u + tree.nodes[j + 1].size) * 32u;
                } else {
                    const uint32_t pj = tree.items[in.item].id;
                    if (emit_tri(pj, tri_slots) != RT_OK)
                        return lfail(err, RT_ERR_BAD_SCENE, "triangle material out of range");
                    word = (slot_of[pj] << 3) | KIND_TRI;
                }
                push_node(in.bmin, in.bmax, base + j + in.size, word);
            }
        }
    }
    const uint32_t n_fast = (uint32_t)(fn.size() / 2);
    epos[n_top] = n_fast;
    for (uint32_t q = 0; q < n_top; q++) { // a top node's subtree ends where the next top node outside it starts
        fn[2 * (size_t)epos[q] + 1].z = u2f(epos[q + top_nodes[q].size] * 32u);
        if (top_nodes[q].item == 0xFFFFFFFFu) // interior: byte offset of the right child (low bits 0 = KIND_INTERNAL)
            fn[2 * (size_t)epos[q] + 1].w = u2f(epos[q + 1u + top_nodes[q + 1].size] * 32u);
    }

    // 5. The reference's own tree in its visiting order (the exact-order walks), sharing the primitive arrays.
    std::vector<float4> &dn = L.dn;
    dn.assign(2 * (size_t)m + 2, make_float4(0.f, 0.f, 0.f, 0.f)); // + the terminator record
    for (uint32_t p = 0; p < m; p++) {
        const rt_node &nd = nodes[order_of[p]];
        const uint32_t kind = nd.flags;
        const uint32_t ref = (nd.flags == RT_NODE_LEAF || nd.flags == RT_NODE_SPHERE) ? slot_of[p] : 0u;
        dn[2 * (size_t)p] = make_float4(nd.pmin[0], nd.pmin[1], nd.pmin[2], nd.pmax[0]);
        dn[2 * (size_t)p + 1] = make_float4(nd.pmax[1], nd.pmax[2], u2f((p + size[p]) * 32u), u2f((ref << 3) | kind));
    }
    // terminator: kind = all ones, never entered
    dn[2 * (size_t)m] = make_float4(1.f, 1.f, 1.f, -1.f);
    dn[2 * (size_t)m + 1] = make_float4(-1.f, -1.f, u2f(m * 32u), u2f(kNone));
    for (uint32_t i = 0; i < n_mats; i++) {
        mt[2 * (size_t)i] = make_float4(mats[i].albedo[0], mats[i].albedo[1], mats[i].albedo[2], mats[i].albedo[3]);
        mt[2 * (size_t)i + 1] = make_float4(u2f(mats[i].kind), mats[i].param, 0.f, 0.f);
    }

    // Surface-area estimate of the nodes a ray through the root box visits.  Thousands (a tree whose boxes
    // barely prune, like the reference's stretched boxes on the 18520-triangle dragon) favour the
    // wave-synchronous walk; tens to hundreds favour the per-lane walk.
    {
        auto area = [](const rt_node &nd) {
            double e[3];
            for (int k = 0; k < 3; k++) {
                e[k] = (double)nd.pmax[k] - (double)nd.pmin[k];
                if (!(e[k] > 0.0))
                    e[k] = 0.0;
            }
            return 2.0 * (e[0] * e[1] + e[1] * e[2] + e[0] * e[2]);
        };
        double sum = 0.0;
        for (uint32_t p = 0; p < m; p++)
            sum += area(nodes[order_of[p]]);
        const double ra = area(nodes[root]);
        L.expected_visits = ra > 0.0 ? sum / ra : (double)m;
    }
    // 6. The compact form of the fast tree (see SceneLayout::cn), on the grid of step 3b.
    L.compact = false;
    if (flat && n_fast > 0 && tri_slots < (1u << 28)) {
        const float4 ra = fn[0], rb = fn[1];
        const double lo[3] = {ra.x, ra.y, ra.z}, hi[3] = {ra.w, rb.x, rb.y};
        bool ok = grid_ok;
        if (ok) {
            L.cn.assign((size_t)n_fast + 1, make_uint4(0u, 0u, 0u, 0u)); // one unused record at the end: the walk loads records in twos
            L.lb.assign(4 * (size_t)tri_slots, make_float4(0.f, 0.f, 0.f, 0.f));
            auto qlo = [&](double p, int k) {
                double q = std::floor((p - (double)L.q_origin[k]) / (double)L.q_step[k]) - 1.0;
                return (uint32_t)(q < 0.0 ? 0.0 : q > 65535.0 ? 65535.0 : q);
            };
            auto qhi = [&](double p, int k) {
                double q = std::ceil((p - (double)L.q_origin[k]) / (double)L.q_step[k]) + 1.0;
                return (uint32_t)(q < 0.0 ? 0.0 : q > 65535.0 ? 65535.0 : q);
            };
            double area_q = 0.0, area_x = 0.0;
            for (uint32_t i = 0; i < n_fast && ok; i++) {
                const float4 a = fn[2 * (size_t)i], b = fn[2 * (size_t)i + 1];
                const double bl[3] = {a.x, a.y, a.z}, bh[3] = {a.w, b.x, b.y};
                uint32_t q[6];
                for (int k = 0; k < 3; k++) {
                    q[k] = qlo(bl[k], k);
                    q[3 + k] = qhi(bh[k], k);
                    // the grid planes must enclose the box with a unit to spare (they do unless a coordinate is not finite)
                    if (!((double)L.q_origin[k] + q[k] * (double)L.q_step[k] <= bl[k] - 0.5 * (double)L.q_step[k]) ||
                        !((double)L.q_origin[k] + q[3 + k] * (double)L.q_step[k] >= bh[k] + 0.5 * (double)L.q_step[k]))
                        ok = false;
                }
                uint32_t skip_w, word_w;
                std::memcpy(&skip_w, &b.z, 4);
                std::memcpy(&word_w, &b.w, 4);
                uint32_t w;
                if ((word_w & 7u) == KIND_TRI) {
                    const uint32_t slot = word_w >> 3;
                    w = (slot << 4) | 1u;
                    const float4 t0 = tp[3 * (size_t)slot], t1 = tp[3 * (size_t)slot + 1], t2 = tp[3 * (size_t)slot + 2];
                    L.lb[4 * (size_t)slot] = t0;
                    L.lb[4 * (size_t)slot + 1] = t1;
                    L.lb[4 * (size_t)slot + 2] = make_float4(t2.x, a.x, a.y, t2.w);
                    L.lb[4 * (size_t)slot + 3] = make_float4(a.z, a.w, b.x, b.y);
                } else {
                    w = (skip_w / 32u) * 16u;
                }
                L.cn[i] = make_uint4(q[0] | (q[3] << 16), q[1] | (q[4] << 16), q[2] | (q[5] << 16), w);
                const double ex[3] = {bh[0] - bl[0], bh[1] - bl[1], bh[2] - bl[2]};
                const double eq[3] = {(double)(q[3] - q[0]) * L.q_step[0], (double)(q[4] - q[1]) * L.q_step[1], (double)(q[5] - q[2]) * L.q_step[2]};
                area_x += ex[0] * ex[1] + ex[1] * ex[2] + ex[2] * ex[0];
                area_q += eq[0] * eq[1] + eq[1] * eq[2] + eq[2] * eq[0];
            }
            {
                const double rx[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
                const double root_area = rx[0] * rx[1] + rx[1] * rx[2] + rx[2] * rx[0];
                L.fast_visits = root_area > 0.0 ? area_x / root_area : (double)n_fast;
            }
            L.compact = ok;
            L.compact_inflation = area_x > 0.0 ? area_q / area_x : 1e30;
            if (!ok) {
                L.cn.clear();
                L.lb.clear();
            }
        }
    }
    // 7. The wide form: its records were made in step 3b; it needs the compact form's leaf records (by slot).
    L.wide = false;
    if (L.compact && wide.ok && wide.n_slots == tri_slots) {
        L.wn.swap(wide.wn);
        L.n_wide = (uint32_t)(L.wn.size() / 4);
        L.wide_depth = wide.depth;
        L.wide_visits = wide.area_sum;
        L.wide = true;
    }
    L.n_ref = m;
    L.n_fast = n_fast;
    L.n_tri_slots = tri_slots;
    L.n_sph_slots = sph_slots;
    return RT_OK;
}

