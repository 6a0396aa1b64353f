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
};

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

    // 4. Emit the fast tree, and with it the primitive arrays in fast-tree leaf order.
    std::vector<float4> &fn = L.fn, &tp = L.tp, &tn = L.tn, &sp = L.sp, &mt = L.mt;
    mt.assign(2 * (size_t)n_mats, make_float4(0.f, 0.f, 0.f, 0.f));
    tp.reserve(3 * (size_t)n_tris);
    tn.reserve(3 * (size_t)n_tris);
    std::vector<uint32_t> slot_of(m, 0); // reference position of a leaf -> its triangle / sphere slot
    uint32_t tri_slots = 0, sph_slots = 0;
    auto push_node = [&](const float *bmin, const float *bmax, uint32_t skip_index, uint32_t word) {
        fn.push_back(make_float4(bmin[0], bmin[1], bmin[2], bmax[0]));
        fn.push_back(make_float4(bmax[1], bmax[2], u2f(skip_index * 32u), u2f(word)));
    };
    auto emit_tri = [&](uint32_t p) -> int { // p = reference position of the leaf
        const rt_node &nd = nodes[order_of[p]];
        const rt_tri &t = tris[nd.left];
        if (n_mats && t.mat >= n_mats)
            return RT_ERR_BAD_SCENE;
        slot_of[p] = tri_slots++;
        // edges are the same fp32 subtractions the triangle test starts with (SimplifiedData.cpp:272-273)
        const float e1x = t.pos[1][0] - t.pos[0][0], e1y = t.pos[1][1] - t.pos[0][1], e1z = t.pos[1][2] - t.pos[0][2];
        const float e2x = t.pos[2][0] - t.pos[0][0], e2y = t.pos[2][1] - t.pos[0][1], e2z = t.pos[2][2] - t.pos[0][2];
        tp.push_back(make_float4(t.pos[0][0], t.pos[0][1], t.pos[0][2], e1x));
        tp.push_back(make_float4(e1y, e1z, e2x, e2y));
        tp.push_back(make_float4(e2z, u2f(nd.left), u2f((uint32_t)t.mat), u2f(p)));
        tn.push_back(make_float4(t.nrm[0][0], t.nrm[0][1], t.nrm[0][2], t.nrm[1][0]));
        tn.push_back(make_float4(t.nrm[1][1], t.nrm[1][2], t.nrm[2][0], t.nrm[2][1]));
        tn.push_back(make_float4(t.nrm[2][2], 0.f, 0.f, 0.f));
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
            if (emit_tri(p) != RT_OK)
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
                    if (emit_tri(pj) != RT_OK)
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
    // 6. The compact form of the fast tree (see SceneLayout::cn).
    L.compact = false;
    if (n_spheres == 0 && mesh_trees.empty() && n_fast > 0 && tri_slots < (1u << 28)) {
        const float4 ra = fn[0], rb = fn[1];
        const double lo[3] = {ra.x, ra.y, ra.z}, hi[3] = {ra.w, rb.x, rb.y};
        bool ok = true;
        double ext_max = 0.0;
        for (int k = 0; k < 3; k++)
            ext_max = hi[k] - lo[k] > ext_max ? hi[k] - lo[k] : ext_max;
        for (int k = 0; k < 3; k++) {
            double ext = hi[k] - lo[k];
            if (!(ext >= 0.0) || !(ext < 1e30))
                ok = false;
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
    L.n_ref = m;
    L.n_fast = n_fast;
    L.n_tri_slots = tri_slots;
    L.n_sph_slots = sph_slots;
    return RT_OK;
}

