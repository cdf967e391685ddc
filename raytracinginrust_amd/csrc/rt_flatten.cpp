// csrc/rt_flatten.cpp — host side: turn the builder tree (one node per reference constructor) into the
// flat device scene of rt_ir.h; BVH build (src/bvh.rs:18-73), Camera::new (src/camera.rs:19-49).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <limits>
#include <functional>
#include <map>
#include <vector>
#include "rt_scene.h"

namespace rt {

namespace {

const double F64_MAX = std::numeric_limits<double>::max();

struct Box { double mn[3], mx[3]; };

struct Flattener {
    Scene& s;
    HostFlat& f;
    std::map<int, uint32_t> prim_of_node;   // node id -> first primitive index in its pool (emitted once, shared)
    std::string err;
    // the wrappers an object stands in, outermost first; n_outer of them lie outside the BVH whose leaf the object is (0 at the top
    // level), med_at of them outside its ConstantMedium (-1: no medium met yet)
    struct Chain { DOp<double> ops[RT_MAX_OPS]; int n = 0; int n_outer = 0; int med_at = -1; };
    std::vector<DObject> subs;              // sub-objects of G_OBJ leaves, appended to f.objects behind the world's own once the world is done
    std::vector<DObject>* target = nullptr; // where emit_object puts an object: f.objects (the world list) or subs
    explicit Flattener(Scene& sc) : s(sc), f(sc.flat) {}

    bool fail(const std::string& m) { if (err.empty()) err = m; return false; }

    // ---- primitive pools
    uint32_t rect_of(int n) {
        auto it = prim_of_node.find(n);
        if (it != prim_of_node.end()) return it->second;
        const HNode& h = s.nodes[n];
        DRect<double> r{h.v[0], h.v[1], h.v[2], h.v[3], h.v[4], (uint32_t)h.plane_or_axis, (uint32_t)h.mat};
        f.rects.push_back(r);
        return prim_of_node[n] = (uint32_t)f.rects.size() - 1;
    }
    uint32_t cube_of(int n) {   // Cube::new, src/cube.rs:14-31: six AARects in this order
        auto it = prim_of_node.find(n);
        if (it != prim_of_node.end()) return it->second;
        const HNode& h = s.nodes[n];
        const double* mn = h.v; const double* mx = h.v + 3;
        uint32_t first = (uint32_t)f.rects.size();
        uint32_t m = (uint32_t)h.mat;
        f.rects.push_back({mn[0], mx[0], mn[1], mx[1], mx[2], 0u /*XY*/, m});
        f.rects.push_back({mn[0], mx[0], mn[1], mx[1], mn[2], 0u, m});
        f.rects.push_back({mn[0], mx[0], mn[2], mx[2], mx[1], 1u /*XZ*/, m});
        f.rects.push_back({mn[0], mx[0], mn[2], mx[2], mn[1], 1u, m});
        f.rects.push_back({mn[1], mx[1], mn[2], mx[2], mx[0], 2u /*YZ*/, m});
        f.rects.push_back({mn[1], mx[1], mn[2], mx[2], mn[0], 2u, m});
        return prim_of_node[n] = first;
    }
    uint32_t sphere_of(int n) {
        auto it = prim_of_node.find(n);
        if (it != prim_of_node.end()) return it->second;
        const HNode& h = s.nodes[n];
        f.spheres.push_back({{h.v[0], h.v[1], h.v[2]}, h.v[3], (uint32_t)h.mat, 0u});
        f.feats |= F_SPHERES;
        return prim_of_node[n] = (uint32_t)f.spheres.size() - 1;
    }
    uint32_t msphere_of(int n) {
        auto it = prim_of_node.find(n);
        if (it != prim_of_node.end()) return it->second;
        const HNode& h = s.nodes[n];
        f.mspheres.push_back({{h.v[0], h.v[1], h.v[2]}, {h.v[3], h.v[4], h.v[5]}, h.v[6], h.v[7], h.v[8], (uint32_t)h.mat, 0u});
        f.feats |= F_SPHERES;
        return prim_of_node[n] = (uint32_t)f.mspheres.size() - 1;
    }
    uint32_t tri_of(int n) {
        auto it = prim_of_node.find(n);
        if (it != prim_of_node.end()) return it->second;
        const HNode& h = s.nodes[n];
        DTri<double> t;
        for (int k = 0; k < 3; k++) {
            t.v0[k] = h.v[k];
            t.e1[k] = h.v[3 + k] - h.v[k];     // tri.rs:27
            t.e2[k] = h.v[6 + k] - h.v[k];     // tri.rs:28
        }
        t.mat = (uint32_t)h.mat; t.pad = 0;
        f.tris.push_back(t);
        f.feats |= F_TRIS;
        return prim_of_node[n] = (uint32_t)f.tris.size() - 1;
    }

    static bool is_bare_prim(HNode::Kind k) { return k == HNode::SPHERE || k == HNode::MSPHERE || k == HNode::RECT || k == HNode::TRI; }

    // geometry of a bare primitive / cube node as (kind, first, count)
    bool simple_geom(int n, uint32_t& kind, uint32_t& first, uint32_t& count) {
        const HNode& h = s.nodes[n];
        switch (h.kind) {
        case HNode::RECT: kind = G_RECT; first = rect_of(n); count = 1; return true;
        case HNode::CUBE: kind = G_RECT; first = cube_of(n); count = 6; return true;
        case HNode::SPHERE: kind = G_SPHERE; first = sphere_of(n); count = 1; return true;
        case HNode::MSPHERE: kind = G_MSPHERE; first = msphere_of(n); count = 1; return true;
        case HNode::TRI: kind = G_TRI; first = tri_of(n); count = 1; return true;
        default: return false;
        }
    }

    // ---- bounding boxes (only what BVH::new needs: src/sphere.rs:97-102,191-201, src/rect.rs:83-89,
    //      src/cube.rs:39-46, src/tri.rs:59-70)
    bool bbox(int n, Box& b) {
        const HNode& h = s.nodes[n];
        switch (h.kind) {
        case HNode::SPHERE:
            for (int k = 0; k < 3; k++) { b.mn[k] = h.v[k] - h.v[3]; b.mx[k] = h.v[k] + h.v[3]; }
            return true;
        case HNode::MSPHERE:
            for (int k = 0; k < 3; k++) {
                double r = h.v[8];
                b.mn[k] = std::fmin(h.v[k] - r, h.v[3 + k] - r);
                b.mx[k] = std::fmax(h.v[k] + r, h.v[3 + k] + r);
            }
            return true;
        case HNode::RECT:     // ignores `plane` (reference quirk)
            b.mn[0] = h.v[0]; b.mn[1] = h.v[2]; b.mn[2] = h.v[4] - 0.0001;
            b.mx[0] = h.v[1]; b.mx[1] = h.v[3]; b.mx[2] = h.v[4] + 0.0001;
            return true;
        case HNode::CUBE:
            for (int k = 0; k < 3; k++) { b.mn[k] = h.v[k]; b.mx[k] = h.v[3 + k]; }
            return true;
        case HNode::TRI:
            for (int k = 0; k < 3; k++) {
                b.mn[k] = std::fmin(h.v[k], std::fmin(h.v[3 + k], h.v[6 + k]));
                b.mx[k] = std::fmax(h.v[k], std::fmax(h.v[3 + k], h.v[6 + k]));
            }
            return true;
        case HNode::LIST: {       // hit.rs:73-88: the first item's box, then surrounding_box over the rest; None for an empty list or a box-less item
            if (h.items.empty() || !bbox(h.items[0], b)) return false;
            for (size_t i = 1; i < h.items.size(); i++) { Box o; if (!bbox(h.items[i], o)) return false; surround(b, o); }
            return true;
        }
        case HNode::FLIP: case HNode::MEDIUM:      // hit.rs:122-124, medium.rs:63-65
            return bbox(h.child, b);
        case HNode::TRANSLATE:                     // translate.rs:32-40
            if (!bbox(h.child, b)) return false;
            for (int k = 0; k < 3; k++) { b.mn[k] += h.v[k]; b.mx[k] += h.v[k]; }
            return true;
        case HNode::ROTATE: {                      // rotate.rs:37-66 (Rotate::new): min starts at f64::MIN, max at f64::MAX, with `<` / `>` updates that
            Box c;                                 // fire only for an infinite corner — the box is all of space (reference quirk B3), reproduced as computed
            if (!bbox(h.child, c)) return false;
            const double radiants = (3.14159265358979323846264338327950288 / 180.0) * h.v[0];
            double sn, cs; ::sincos(radiants, &sn, &cs);
            const int ax = h.plane_or_axis;
            const int r_axis = ax == 0 ? 0 : (ax == 1 ? 1 : 2), a_axis = ax == 0 ? 1 : 0, b_axis = ax == 2 ? 1 : 2;      // rotate.rs:14-20
            for (int k = 0; k < 3; k++) { b.mn[k] = -F64_MAX; b.mx[k] = F64_MAX; }
            for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
                const double r = (double)k * c.mx[r_axis] + (double)(1 - k) * c.mn[r_axis];
                const double a = (double)i * c.mx[a_axis] + (double)(1 - i) * c.mn[a_axis];
                const double bb = (double)j * c.mx[b_axis] + (double)(1 - j) * c.mn[b_axis];
                const double new_a = cs * a + sn * bb;
                const double new_b = -sn * a + cs * bb;
                if (new_a < b.mn[a_axis]) b.mn[a_axis] = new_a;
                if (new_b < b.mn[b_axis]) b.mn[b_axis] = new_b;
                if (r < b.mn[r_axis]) b.mn[r_axis] = r;
                if (new_a > b.mx[a_axis]) b.mx[a_axis] = new_a;
                if (new_b > b.mx[b_axis]) b.mx[b_axis] = new_b;
                if (r > b.mx[r_axis]) b.mx[r_axis] = r;
            }
            return true;
        }
        case HNode::BVH: {        // bvh.rs:93-95: the root's box = surrounding_box folded over the tree (min / max: the same for every tree shape)
            if (h.items.empty() || !bbox(h.items[0], b)) return false;
            for (size_t i = 1; i < h.items.size(); i++) { Box o; if (!bbox(h.items[i], o)) return false; surround(b, o); }
            return true;
        }
        }
        return false;
    }
    static void surround(Box& b, const Box& o) { for (int k = 0; k < 3; k++) { b.mn[k] = std::fmin(b.mn[k], o.mn[k]); b.mx[k] = std::fmax(b.mx[k], o.mx[k]); } }   // aabb.rs:40-51

    // ---- BVH::new, src/bvh.rs:18-73.  Emits nodes in DFS preorder (left child = parent + 1); relayout_bvh_by_depth() below then
    // renumbers them by depth and stores both children.
    // `sort_unstable_by` leaves ties unspecified; a stable sort is used (tie order changes cost, never results).
    // A child that is not a bare primitive / Cube (bvh.rs:18 takes any Box<dyn Hittable>) becomes a leaf of kind G_OBJ: the run of sub-objects
    // it flattens to under the chain of wrappers the BVH itself stands in (`chain`: each sub-object carries the whole chain from the world
    // down, the first chain.n ops of which lie outside this BVH).  `nest`: how many BVHs this one is inside of.
    bool build_bvh(std::vector<int> items, uint32_t depth, uint32_t& out_index, Box& out_box, const Chain& chain, int nest) {
        if (items.empty()) return fail("no object in the scene");                    // bvh.rs:55
        if (depth > (uint32_t)RT_MAX_BVH_DEPTH) return fail("BVH deeper than RT_MAX_BVH_DEPTH");
        f.bvh_depth = std::max(f.bvh_depth, depth);
        std::vector<Box> boxes(items.size());
        for (size_t i = 0; i < items.size(); i++)
            if (!bbox(items[i], boxes[i])) return fail("no bounding box in bvh node");                 // bvh.rs:28,61 panic (an empty list, or a wrapper of one)
        int axis = 0; double best = 0;
        for (int a = 0; a < 3; a++) {                                                 // bvh.rs:33-48
            double mn = F64_MAX, mx = -F64_MAX;
            for (const Box& b : boxes) { mn = std::fmin(mn, b.mn[a]); mx = std::fmax(mx, b.mx[a]); }
            double range = mx - mn;
            if (a == 0 || range > best) { best = range; axis = a; }
        }
        std::vector<size_t> order(items.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {       // bvh.rs:19-31,51
            return (boxes[x].mn[axis] + boxes[x].mx[axis]) < (boxes[y].mn[axis] + boxes[y].mx[axis]);
        });
        size_t n_lower = items.size() / 2;                                           // bvh.rs:65: the object median
        // Opt-in builder (rt_scene_set_bvh_builder(RT_BVH_SAH)), not the reference's: binned surface-area heuristic.  Same
        // node / leaf forms (one object per leaf), so the kernels do not care; only the tree's shape — and with it the order
        // in which equal-t hits are met — differs.  Falls back to the median split when no SAH split separates the objects
        // and near the depth limit (an SAH tree can be lopsided).
        if (s.bvh_builder == 1 && items.size() > 2 && depth + 8 < (uint32_t)RT_MAX_BVH_DEPTH) {
            auto area = [](const Box& b) {
                double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
                return 2.0 * (dx * dy + dy * dz + dz * dx);
            };
            auto grow = [](Box& b, const Box& o) { for (int k = 0; k < 3; k++) { b.mn[k] = std::fmin(b.mn[k], o.mn[k]); b.mx[k] = std::fmax(b.mx[k], o.mx[k]); } };
            const int NB = 16;
            double best_cost = F64_MAX; int best_axis = -1, best_split = -1; double best_lo = 0, best_scale = 0;
            for (int a = 0; a < 3; a++) {
                double cmin = F64_MAX, cmax = -F64_MAX;
                for (const Box& b : boxes) { double c = 0.5 * (b.mn[a] + b.mx[a]); cmin = std::fmin(cmin, c); cmax = std::fmax(cmax, c); }
                if (!(cmax > cmin)) continue;
                const double scale = NB / (cmax - cmin);
                Box bb[NB]; size_t cnt[NB];
                for (int i = 0; i < NB; i++) { cnt[i] = 0; for (int k = 0; k < 3; k++) { bb[i].mn[k] = F64_MAX; bb[i].mx[k] = -F64_MAX; } }
                for (const Box& b : boxes) {
                    int bi = (int)((0.5 * (b.mn[a] + b.mx[a]) - cmin) * scale); if (bi >= NB) bi = NB - 1; if (bi < 0) bi = 0;
                    cnt[bi]++; grow(bb[bi], b);
                }
                double right_area[NB]; size_t right_cnt[NB];
                Box acc; for (int k = 0; k < 3; k++) { acc.mn[k] = F64_MAX; acc.mx[k] = -F64_MAX; }
                size_t c = 0;
                for (int i = NB - 1; i >= 1; i--) { if (cnt[i]) grow(acc, bb[i]); c += cnt[i]; right_area[i] = c ? area(acc) : 0.0; right_cnt[i] = c; }
                for (int k = 0; k < 3; k++) { acc.mn[k] = F64_MAX; acc.mx[k] = -F64_MAX; }
                c = 0;
                for (int i = 0; i < NB - 1; i++) {                  // split between bin i and i + 1
                    if (cnt[i]) grow(acc, bb[i]);
                    c += cnt[i];
                    if (c == 0 || right_cnt[i + 1] == 0) continue;
                    double cost = area(acc) * (double)c + right_area[i + 1] * (double)right_cnt[i + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = i; best_lo = cmin; best_scale = scale; }
                }
            }
            if (best_axis >= 0) {
                axis = best_axis;
                std::vector<size_t> lo, hi;
                for (size_t i = 0; i < boxes.size(); i++) {
                    int bi = (int)((0.5 * (boxes[i].mn[axis] + boxes[i].mx[axis]) - best_lo) * best_scale); if (bi >= NB) bi = NB - 1; if (bi < 0) bi = 0;
                    (bi <= best_split ? lo : hi).push_back(i);
                }
                n_lower = lo.size();
                order = lo; order.insert(order.end(), hi.begin(), hi.end());
            }
        }
        uint32_t me = (uint32_t)f.bvh.size();
        out_index = me;
        f.bvh.push_back(DBvhNode<double>{});
        size_t length = items.size();
        if (length == 1) {
            uint32_t kind, first, count;
            if (!simple_geom(items[0], kind, first, count)) {
                // any other Hittable: the sub-objects it flattens to (a list may give several: HittableList::hit over them, hit.rs:59-71)
                // (collected apart and appended as one run: a nested BVH among them puts ITS leaves' sub-objects into `subs` on the way)
                std::vector<DObject>* const saved = target;
                std::vector<DObject> mine;
                target = &mine;
                Chain c2 = chain; c2.n_outer = chain.n; c2.med_at = -1;
                const bool ok = emit(items[0], c2, -1, nest + 1);
                target = saved;
                if (!ok) return false;
                const size_t s0 = subs.size();
                subs.insert(subs.end(), mine.begin(), mine.end());
                kind = G_OBJ; first = (uint32_t)s0; count = (uint32_t)mine.size();
                f.feats |= F_NESTED;
                // (count == 0 cannot be reached: a child without sub-objects is an empty list, which has no bounding box)
            }
            if (first >= (1u << 28)) return fail("too many primitives");
            out_box = boxes[0];
            DBvhNode<double>& nd = f.bvh[me];
            for (int k = 0; k < 3; k++) { nd.mn[k] = out_box.mn[k]; nd.mx[k] = out_box.mx[k]; }
            nd.a = BVH_LEAF | (kind << 28) | first;
            nd.b = count;
            return true;
        }
        std::vector<int> lower, upper;
        for (size_t i = 0; i < n_lower; i++) lower.push_back(items[order[i]]);
        for (size_t i = n_lower; i < length; i++) upper.push_back(items[order[i]]);      // bvh.rs:65: drain(length/2..) -> right
        uint32_t li, ri; Box lb, rb;
        if (!build_bvh(lower, depth + 1, li, lb, chain, nest)) return false;
        if (!build_bvh(upper, depth + 1, ri, rb, chain, nest)) return false;
        for (int k = 0; k < 3; k++) { out_box.mn[k] = std::fmin(lb.mn[k], rb.mn[k]); out_box.mx[k] = std::fmax(lb.mx[k], rb.mx[k]); }   // aabb.rs:40-51
        DBvhNode<double>& nd = f.bvh[me];
        for (int k = 0; k < 3; k++) { nd.mn[k] = out_box.mn[k]; nd.mx[k] = out_box.mx[k]; }
        (void)li;                  // the left child is always me + 1 (preorder), so `a` carries the split axis instead:
        nd.a = (uint32_t)axis;     // children were ordered by centroid along it (used only by the opt-in near-first traversal)
        nd.b = ri;
        return true;
    }

    // ---- world flattening
    bool emit_object(uint32_t kind, uint32_t first, uint32_t count, const Chain& chain, int medium, bool is_cube = false) {
        DObject o{};
        o.geom_kind = kind; o.geom_first = first; o.geom_count = count; o.is_cube = is_cube ? 1u : 0u;
        o.first_op = (uint32_t)f.ops.size(); o.n_ops = (uint32_t)chain.n; o.medium = medium;
        const int med_at = medium >= 0 ? chain.med_at : 0;
        o.nest = (uint32_t)chain.n_outer | ((uint32_t)med_at << 8);
        bool flips_only = chain.n > 0 && medium < 0;
        for (int i = 0; i < chain.n; i++) flips_only = flips_only && chain.ops[i].kind == OP_FLIP;
        if (flips_only) o.nest |= 0x10000u;            // every wrapper is a FlipNormal: the hit test may use the incoming ray as it is
        if (med_at != 0) f.feats |= F_NESTED;          // a ConstantMedium under a wrapper: the all-features kernel's object_hit serves it
        for (int i = 0; i < chain.n; i++) f.ops.push_back(chain.ops[i]);
        target->push_back(o);
        return true;
    }

    bool emit(int n, const Chain& chain, int medium, int nest = 0) {
        if (n < 0 || n >= (int)s.nodes.size()) return fail("bad hittable handle");
        const HNode& h = s.nodes[n];
        uint32_t kind, first, count;
        switch (h.kind) {
        case HNode::SPHERE: case HNode::MSPHERE: case HNode::RECT: case HNode::TRI: case HNode::CUBE:
            simple_geom(n, kind, first, count);
            return emit_object(kind, first, count, chain, medium, h.kind == HNode::CUBE);
        case HNode::LIST: {
            // HittableList::hit (hit.rs:59-71) keeps the closest hit, later items winning ties, and wrappers
            // act per hit, so Wrapper(List[a,b]) == List[Wrapper(a), Wrapper(b)].  A ConstantMedium boundary
            // is different (two boundary queries, medium.rs:29-30) and must stay one object.
            // (As a medium boundary a list must flatten to exactly one object: one wrapped item, or a homogeneous run of
            // bare primitives such as a Mesh's triangles, which becomes one typed range.)
            size_t i = 0;
            bool emitted_any = false;
            while (i < h.items.size()) {
                const HNode& c = s.nodes[h.items[i]];
                // run of consecutive bare primitives of one kind whose records are contiguous -> one range object
                if (is_bare_prim(c.kind) && prim_of_node.find(h.items[i]) == prim_of_node.end()) {
                    // (each item is emitted as the run grows, so a handle pushed twice — `[a, a]`, legal through clone() in the
                    // reference — ends the run at its second occurrence, which then becomes an object of its own over the one shared
                    // record; a newly created record always lands at the end of its pool, i.e. at f0 + c0)
                    size_t j = i;
                    uint32_t k0 = 0, f0 = 0, c0 = 0;
                    while (j < h.items.size() && s.nodes[h.items[j]].kind == c.kind && prim_of_node.find(h.items[j]) == prim_of_node.end()) {
                        uint32_t kk, ff, cc; simple_geom(h.items[j], kk, ff, cc);
                        if (c0 == 0) { k0 = kk; f0 = ff; }
                        else if (ff != f0 + c0) return fail("internal: primitive run is not contiguous");
                        c0++; j++;
                    }
                    if (medium >= 0 && (emitted_any || j < h.items.size())) return fail("ConstantMedium boundary must flatten to one object");
                    if (!emit_object(k0, f0, c0, chain, medium)) return false;
                    emitted_any = true;
                    i = j;
                } else {
                    if (medium >= 0 && (emitted_any || h.items.size() != 1)) return fail("ConstantMedium boundary must flatten to one object");
                    if (!emit(h.items[i], chain, medium, nest)) return false;
                    emitted_any = true;
                    i++;
                }
            }
            // an empty boundary list never reports a hit (hit.rs:59-71), so such a medium never scatters and draws nothing
            // (medium.rs:29): emitting no object at all is the same thing
            return true;
        }
        case HNode::FLIP: case HNode::TRANSLATE: case HNode::ROTATE: {
            if (chain.n >= RT_MAX_OPS) return fail("wrapper chain longer than RT_MAX_OPS");
            Chain c2 = chain;
            DOp<double>& op = c2.ops[c2.n++];
            op = DOp<double>{};
            if (h.kind == HNode::FLIP) op.kind = OP_FLIP;
            else if (h.kind == HNode::TRANSLATE) { op.kind = OP_TRANSLATE; op.x = h.v[0]; op.y = h.v[1]; op.z = h.v[2]; }
            else {
                op.kind = OP_ROTATE; op.axis = (uint32_t)h.plane_or_axis;
                double radiants = (3.14159265358979323846264338327950288 / 180.0) * h.v[0];    // rotate.rs:34-36
                // `radians.sin()` and `radians.cos()` of one operand in one block: LLVM's legaliser turns the pair into ONE sincos libcall
                // where the C library has it (x86-64 linux-gnu), and glibc's sincos differs from its sin() / cos() in the last ulp for
                // 0.13 % of arguments (*measured*, glibc 2.35: 2594 of 2e6) — enough to flip which of two coincident triangles behind such
                // a Rotate wins an exact tie (found by the round-3 fuzz sweep).  So: sincos, explicitly, here and in the oracle.
                double sn, cs; ::sincos(radiants, &sn, &cs);
                op.x = sn; op.y = cs;
            }
            return emit(h.child, c2, medium, nest);
        }
        case HNode::MEDIUM: {
            if (medium >= 0) return fail("nested ConstantMedium is not supported");
            // Isotropic::new(texture), medium.rs:21
            DMaterial<double> iso{}; iso.kind = M_ISOTROPIC; iso.tex = (uint32_t)h.mat;
            f.materials.push_back(iso);
            DMedium<double> m{}; m.neg_inv_density = -(1.0 / h.v[0]); m.mat = (uint32_t)f.materials.size() - 1;
            f.media.push_back(m);
            f.feats |= F_MEDIUM;
            Chain c2 = chain; c2.med_at = chain.n;      // the wrappers met so far lie outside the medium
            return emit(h.child, c2, (int)f.media.size() - 1, nest);
        }
        case HNode::BVH: {
            if (nest > RT_MAX_NEST) return fail("BVHs nested more than RT_MAX_NEST deep inside BVH leaves");
            uint32_t root; Box b;
            if (!build_bvh(h.items, 1, root, b, chain, nest)) return false;
            f.feats |= F_BVH;
            return emit_object(G_BVH, root, 1, chain, medium);
        }
        }
        return fail("unknown node kind");
    }

    // ---- a ROOM: bare AARects of the world list that are exact faces of ONE axis-aligned box become one object (list scenes only)
    // HittableList::hit over n such walls is n exact rect tests — n f64 divisions for every lane of a wave; as faces of a box they are what
    // Cube::hit's fast path decides with ONE exact test (rt_kernel.hip: cube_fast, ROOM form; the Cornell room's five walls: *measured*
    // tools/room_as_cube_probe.py, six walls as one Cube +6.7 %).  What makes it exact:
    //   * a wall joins only if its record IS a face of the box: its plane coordinate equals the box's min or max on that axis and its two
    //     ranges equal the box's, bit for bit; no wrapper, no medium, at most one wall per face, every axis with min < max; the walls keep
    //     their list order inside the room's run of records (copies: the originals stay where lights and other objects refer to them);
    //     a run of bare rects that gives only some of its rects keeps the others, each stretch of them as an object where it stood;
    //   * HittableList::hit returns the hit with the smallest t, the later item on an exact tie (hit.rs:59-71) — a function of the items'
    //     own hits and their ORDER, not of the sequence in which they are tried, as long as no item's test depends on the closest hit
    //     beyond `t <= closest` (true of rects, Cubes and wrapped ones; a ConstantMedium draws from the RNG by it, a BVH has none here:
    //     rooms are formed only when the scene has neither — feats == 0, the list-scene kernels).  The room is tried where its LAST wall
    //     stood, so every object that stood between two walls is tried before them; a tie between a wall and such an object that came
    //     AFTER the wall must still go to the object: per wall, the new index of the first object that came after it (five bits each in
    //     first_op, which a wrapper-less object does not use) — the kernel rejects a wall's hit on an exact tie with an object at or
    //     beyond that index (rt_kernel.hip: object_hit).
    // At least four walls (three exact tests cost what the fast path's approximate phase does); RT_NO_ROOM (A/B runs, tests) turns it off.
    void form_room() {
        // (list scenes — no feature bit —; and, in the SIMPLE form only — the walls next to each other in the list, nothing between them: the
        // room then stands exactly where they stood, no order changes, no tie rule, no second list —, scenes whose only other feature is a
        // BVH of triangles: the mesh kernels carry the fast path at the world list's site, rt_kernel.hip RoomSite; the teapot room, C4)
        const bool mesh_scene = f.feats != 0u && (f.feats & ~(uint32_t)(F_BVH | F_TRIS)) == 0u;
        if ((f.feats != 0u && !mesh_scene) || !subs.empty() || std::getenv("RT_NO_ROOM")) return;
        const uint32_t n = (uint32_t)f.objects.size();
        // the list, item by item: a run of bare rects is the rects it holds (HittableList[a, b] is a then b), anything else is itself
        struct Item { uint32_t obj; int rect; };                // rect < 0: the whole object, not a candidate
        std::vector<Item> items;
        for (uint32_t i = 0; i < n; i++) {
            const DObject& o = f.objects[i];
            const bool bare_rects = o.geom_kind == G_RECT && o.n_ops == 0u && o.medium < 0 && o.is_cube == 0u && o.nest == 0u;
            if (bare_rects) for (uint32_t r = 0; r < o.geom_count; r++) items.push_back({i, (int)(o.geom_first + r)});
            else items.push_back({i, -1});
        }
        auto axes = [](uint32_t plane, int& k, int& a, int& b) { k = 2 - (int)plane; a = plane == 2u ? 1 : 0; b = plane == 0u ? 1 : 2; };     // rect.rs:26-32
        std::vector<size_t> cand;                               // items that are bare rects on a known plane, in list order
        for (size_t x = 0; x < items.size(); x++) if (items[x].rect >= 0 && f.rects[(size_t)items[x].rect].plane <= 2u) cand.push_back(x);
        std::vector<size_t> best; double bmn[3] = {0, 0, 0}, bmx[3] = {0, 0, 0}; uint32_t best_slot[6] = {7u, 7u, 7u, 7u, 7u, 7u};
        const size_t n_seed = std::min<size_t>(cand.size(), 48u);       // (the pairs that may define the box: among the first 48 bare rects of the list)
        for (size_t xi = 0; xi < n_seed; xi++) for (size_t yi = 0; yi < n_seed; yi++) {
            const size_t x = cand[xi], y = cand[yi];
            // two walls on different planes fix a box: x gives two ranges and a plane coordinate, y the range of x's plane axis
            const DRect<double>& rx = f.rects[(size_t)items[x].rect]; const DRect<double>& ry = f.rects[(size_t)items[y].rect];
            if (rx.plane == ry.plane) continue;
            int kx, ax, bx, ky, ay, by; axes(rx.plane, kx, ax, bx); axes(ry.plane, ky, ay, by);
            double mn[3], mx[3];
            mn[ax] = rx.a0; mx[ax] = rx.a1; mn[bx] = rx.b0; mx[bx] = rx.b1;
            if (ay == kx) { mn[kx] = ry.a0; mx[kx] = ry.a1; } else if (by == kx) { mn[kx] = ry.b0; mx[kx] = ry.b1; } else continue;
            bool ok = true;
            for (int q = 0; q < 3; q++) ok = ok && std::isfinite(mn[q]) && std::isfinite(mx[q]) && mn[q] < mx[q];
            if (!ok) continue;
            std::vector<size_t> got; uint32_t slot_of[6] = {7u, 7u, 7u, 7u, 7u, 7u};
            for (size_t w : cand) {
                const DRect<double>& r = f.rects[(size_t)items[w].rect];
                int k, a, b; axes(r.plane, k, a, b);
                if (!(r.a0 == mn[a] && r.a1 == mx[a] && r.b0 == mn[b] && r.b1 == mx[b])) continue;
                int face;                                       // cube.rs:17-24: (XY, XZ, YZ) x (max, min)
                if (r.k == mx[k]) face = 2 * (int)r.plane; else if (r.k == mn[k]) face = 2 * (int)r.plane + 1; else continue;
                if (slot_of[face] != 7u) continue;              // a second wall on the same face stays an ordinary rect
                slot_of[face] = (uint32_t)got.size(); got.push_back(w);
            }
            if (got.size() > best.size()) {
                best = got;
                for (int q = 0; q < 3; q++) { bmn[q] = mn[q]; bmx[q] = mx[q]; }
                for (int q = 0; q < 6; q++) best_slot[q] = slot_of[q];
            }
        }
        if (best.size() < 4u) return;
        std::vector<unsigned char> is_wall(items.size(), 0);
        for (size_t w : best) is_wall[w] = 1;
        const size_t last = best.back();                        // (`best` is in list order)
        // the new list: the walls leave, the room stands where the last of them stood, what is left of a run of rects stays where it was
        // (as one object per stretch of consecutive records)
        std::vector<DObject> out; std::vector<uint32_t> new_of(items.size(), 0u);
        uint32_t room_at = 0;
        for (size_t x = 0; x < items.size(); x++) {
            if (is_wall[x]) { if (x == last) { room_at = (uint32_t)out.size(); out.push_back(DObject{}); } continue; }
            const Item& it = items[x];
            if (it.rect < 0) { new_of[x] = (uint32_t)out.size(); out.push_back(f.objects[it.obj]); continue; }
            const bool joins = x > 0 && !is_wall[x - 1] && items[x - 1].obj == it.obj && items[x - 1].rect + 1 == it.rect && !out.empty() && x - 1 != last;
            if (joins) { out.back().geom_count++; new_of[x] = (uint32_t)out.size() - 1u; }
            else { DObject o = f.objects[it.obj]; o.geom_first = (uint32_t)it.rect; o.geom_count = 1u; new_of[x] = (uint32_t)out.size(); out.push_back(o); }
        }
        if (room_at > 31u) return;
        if (mesh_scene) for (size_t x = best.front() + 1u; x < last; x++) if (!is_wall[x]) return;
        // What stands between the first and the last wall is searched BEFORE walls it stood behind.  That is the same search as long as
        // every plane distance is a number (the argument above); a NaN one — 0 / 0: a ray with a zero direction component that starts ON
        // a plane — is accepted by `t < t_min || t > t_max` and makes every later item pass `t <= closest`, i.e. the result then depends
        // on the ORDER.  The kernel sends every wave that holds a ray with a zero or non-finite direction component through the list as
        // the reference has it (n_alt, below; rt_kernel.hip world_hit_list); for that test to cover the objects in between they must
        // see the path's own ray: no Translate / Rotate among them (FlipNormals change the record only).
        for (size_t x = best.front() + 1u; x < last; x++) {
            if (is_wall[x]) continue;
            const DObject& o = f.objects[items[x].obj];
            if (o.n_ops != 0u && !(o.nest & 0x10000u)) return;
        }
        DObject room{};
        room.geom_kind = G_RECT; room.geom_first = (uint32_t)f.rects.size(); room.geom_count = (uint32_t)best.size();
        room.first_op = 0u; room.n_ops = 0u; room.medium = -1; room.nest = 0u;
        uint32_t map = 0u;
        for (int q = 0; q < 6; q++) map |= best_slot[q] << (3 * q);
        room.is_cube = 2u | (map << 8);
        for (size_t j = 0; j < best.size(); j++) {
            // the first object that stood AFTER this wall and is tried BEFORE the room (none: the room's own index, which no earlier hit carries)
            uint32_t thr = room_at;
            for (size_t x = best[j] + 1u; x < last; x++) if (!is_wall[x]) { thr = new_of[x]; break; }
            room.first_op |= std::min(thr, 31u) << (5u * (uint32_t)j);
        }
        const uint32_t mat0 = f.rects[(size_t)items[best[0]].rect].mat;
        for (size_t w : best) f.rects.push_back(DRect<double>(f.rects[(size_t)items[w].rect]));
        f.rects.push_back({bmn[0], bmx[0], bmn[1], bmx[1], bmx[2], 0u, mat0});       // the box, laid out like a Cube's first two faces (never tested, never hit)
        f.rects.push_back({bmn[0], bmx[0], bmn[1], bmx[1], bmn[2], 0u, mat0});
        out[room_at] = room;
        // the list as the reference has it stays behind the new one: what a wave searches when a ray of it could produce a NaN plane
        // distance (rt_kernel.hip: world_hit)
        room_n_top = (uint32_t)out.size();
        if (!mesh_scene) { f.n_alt = n; out.insert(out.end(), f.objects.begin(), f.objects.end()); }
        f.objects.swap(out);
    }
    uint32_t room_n_top = 0;               // != 0: form_room made a room; the world list is objects[0, room_n_top)

    bool run() {
        f = HostFlat{};
        f.materials = s.materials;
        f.textures = s.textures;
        for (const auto& m : f.materials) { if (m.kind == M_DIELECTRIC) f.feats |= F_DIELECTRIC; if (m.kind == M_PBR) f.feats |= F_PBR; if (m.kind == M_ISOTROPIC) f.feats |= F_MEDIUM; }
        for (const auto& t : f.textures) if (t.kind != T_CONSTANT) f.feats |= F_TEXTURES;
        // a constant texture's colour rides in the material record too (albedo is otherwise unused by these kinds): kernels
        // without texture arms then take it from the record they hold instead of gathering the texture record behind it
        for (auto& m : f.materials)
            if ((m.kind == M_LAMBERTIAN || m.kind == M_DIFFUSE_LIGHT) && m.tex < f.textures.size() && f.textures[m.tex].kind == T_CONSTANT)
                for (int k = 0; k < 3; k++) m.albedo[k] = f.textures[m.tex].color[k];
        if (s.world < 0) return fail("world not set");
        Chain c;
        target = &f.objects;
        if (!emit(s.world, c, -1)) return false;
        form_room();
        // the sub-objects follow the world's own objects in the one table: G_OBJ leaves learn their final indices
        f.n_top = room_n_top ? room_n_top : (uint32_t)f.objects.size();
        if (!subs.empty()) {
            for (DBvhNode<double>& nd : f.bvh)
                if ((nd.a & BVH_LEAF) && ((nd.a >> 28) & 7u) == G_OBJ) {
                    const uint32_t first = (nd.a & 0x0FFFFFFFu) + f.n_top;
                    if (first >= (1u << 28)) return fail("too many objects");
                    nd.a = BVH_LEAF | (G_OBJ << 28) | first;
                }
            f.objects.insert(f.objects.end(), subs.begin(), subs.end());
        }
        // which materials' textures read (u, v): ImageTexture, possibly under CheckTextures (texture.rs:45-54)
        {
            std::vector<int> uv(f.textures.size(), -1);          // -1 unknown, 0 / 1 known
            std::function<bool(uint32_t, int)> reads_uv = [&](uint32_t t, int depth) -> bool {
                if (t >= f.textures.size() || depth > 64) return false;
                if (uv[t] >= 0) return uv[t] != 0;
                const auto& tx = f.textures[t];
                bool r = tx.kind == T_IMAGE || (tx.kind == T_CHECK && (reads_uv(tx.a, depth + 1) || reads_uv(tx.b, depth + 1)));
                uv[t] = r ? 1 : 0;
                return r;
            };
            for (auto& m : f.materials)
                if ((m.kind == M_LAMBERTIAN || m.kind == M_DIFFUSE_LIGHT || m.kind == M_ISOTROPIC || m.kind == M_PBR) && reads_uv(m.tex, 0)) m.kind |= MAT_NEEDS_UV;
        }
        // lights: HittableList of FlipNormal(AARect) / AARect / Sphere (hit.rs:90-96, 125-132; rect.rs:91-111; sphere.rs:104-119);
        // anything else has the trait defaults pdf_value = 0, random = (1,0,0) (hit.rs:29-30)
        for (int l : s.lights) {
            int n = l;
            while (s.nodes[n].kind == HNode::FLIP) n = s.nodes[n].child;
            const HNode& h = s.nodes[n];
            if (h.kind == HNode::RECT) f.lights.push_back({L_RECT, rect_of(n)});
            else if (h.kind == HNode::SPHERE) f.lights.push_back({L_SPHERE, sphere_of(n)});
            else if (h.kind == HNode::LIST) return fail("a HittableList nested inside `lights` is not supported");
            else f.lights.push_back({L_OTHER, 0});
        }
        return true;
    }
};

// The kernels stage the first n_cached node ids in LDS, so ids are handed out by depth: every tree's root first, then all the
// depth-1 nodes, ... (ties in build order).  Traversal order does not depend on ids: an inner node names both its children
// (c = left, b = right), and a leaf keeps its rank in DFS preorder (c), which is what "later in the reference's visiting order" means
// for the near-first tie rule.
void relayout_bvh_by_depth(HostFlat& f) {
    const size_t n = f.bvh.size();
    if (n == 0) return;
    const uint32_t DONE = 0xFFFFFFFFu;
    std::vector<uint32_t> depth(n, DONE), skip(n, DONE);      // skip: where BVH::hit's recursion goes once this subtree is finished or culled
    struct Todo { uint32_t i, d, skip; };
    std::vector<Todo> todo;                                   // (preorder id, depth, skip link)
    for (const DObject& ob : f.objects) if (ob.geom_kind == G_BVH) todo.push_back({ob.geom_first, 0u, DONE});
    while (!todo.empty()) {
        const Todo t = todo.back(); todo.pop_back();
        depth[t.i] = t.d; skip[t.i] = t.skip;
        // left child (preorder successor): when it is done the right child is next; right child: whatever follows the parent
        if (!(f.bvh[t.i].a & BVH_LEAF)) { todo.push_back({f.bvh[t.i].b, t.d + 1u, t.skip}); todo.push_back({t.i + 1u, t.d + 1u, f.bvh[t.i].b}); }
    }
    std::vector<uint32_t> order(n), new_id(n);
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return depth[x] < depth[y]; });
    for (size_t k = 0; k < n; k++) new_id[order[k]] = (uint32_t)k;
    std::vector<DBvhNode<double>> out(n);
    for (size_t i = 0; i < n; i++) {
        DBvhNode<double> nd = f.bvh[i];
        nd.skip = skip[i] == DONE ? DONE : new_id[skip[i]];
        if (nd.a & BVH_LEAF) nd.c = (uint32_t)i;
        else { nd.c = new_id[i + 1]; nd.b = new_id[nd.b]; }
        out[new_id[i]] = nd;
    }
    f.bvh.swap(out);
    for (DObject& ob : f.objects) if (ob.geom_kind == G_BVH) ob.geom_first = new_id[ob.geom_first];
}

// The filtered walk's nodes (rt_kernel.hip: bvh_hit_filt): each f64 box rounded OUTWARD to f32, the same skip link, and the one word a
// passing box step moves to — the left child, or for a leaf its own id with FNODE_LEAF set.
void make_filter_nodes(HostFlat& f, const std::vector<unsigned char>* take_out = nullptr) {
    const size_t n = f.bvh.size();
    f.bvh_f.assign(n, DFNode{});
    f.filter_m = 0.0f;
    if (n == 0 || n >= FNODE_LEAF) return;
    const float neg_inf = -std::numeric_limits<float>::infinity(), pos_inf = std::numeric_limits<float>::infinity();
    auto down = [&](double x) { float v = (float)x; if ((double)v > x) v = std::nextafterf(v, neg_inf); return v; };
    auto up = [&](double x) { float v = (float)x; if ((double)v < x) v = std::nextafterf(v, pos_inf); return v; };
    float m = 1.0f; bool ok = true;
    for (size_t i = 0; i < n; i++) {
        const DBvhNode<double>& nd = f.bvh[i];
        DFNode& o = f.bvh_f[i];
        for (int k = 0; k < 3; k++) {
            o.b[2 * k] = down(nd.mn[k]); o.b[2 * k + 1] = up(nd.mx[k]);
            ok = ok && std::isfinite(o.b[2 * k]) && std::isfinite(o.b[2 * k + 1]) && nd.mn[k] <= nd.mx[k];
            m = std::fmax(m, std::fmax(std::fabs(o.b[2 * k]), std::fabs(o.b[2 * k + 1])));
        }
        o.skip = nd.skip;
        o.info = (nd.a & BVH_LEAF) ? ((uint32_t)i | FNODE_LEAF) : nd.c;
    }
    if (ok && m <= 0x1p40f) f.filter_m = m;
    // Contraction.  By the ordered-scan form of BVH::hit (rt_kernel.hip: the filtered walk) ANY conservative hierarchy over the leaves in
    // the reference's order gives the reference's samples.  An inner node whose box is nearly its parent's almost always passes when the
    // parent does — its test buys nothing — so it leaves the filter tree: every link to it goes to its left child instead (its right
    // child is already where the left subtree's skip links lead, and what follows the right subtree is what followed the node).  The
    // f64 nodes keep the reference's tree (exact walk of untamed waves).  Area ratio above which a node goes: *measured* (round 5,
    // profiles/r05_collapse_sweep.log) 0.5 — the break-even if a box were hit in proportion to its area — is far too eager (final scene
    // 2.5x slower: rays are culled by the closest hit, not by area), 0.7 - 0.8 is best on all three BVH scenes (+2 ... +4 %).
    double tau = 0.75;
    if (const char* v = std::getenv("RT_COLLAPSE_TAU")) {       // A/B runs only (>= 1: no contraction).  An empty or garbled value is ignored — read as 0 it
        char* end = nullptr;                                    // would contract every inner node away and leave a near-linear scan of the leaves —
        const double t = std::strtod(v, &end);                  // and anything below the surface-area break-even is clamped to it
        if (end != v && *end == '\0' && std::isfinite(t)) tau = std::min(2.0, std::max(0.5, t));
    }
    auto area = [&](const DBvhNode<double>& b) { const double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2]; return dx * dy + dy * dz + dz * dx; };
    std::vector<uint32_t> parent(n, 0xFFFFFFFFu), redirect(n);
    for (size_t i = 0; i < n; i++) if (!(f.bvh[i].a & BVH_LEAF)) { parent[f.bvh[i].c] = (uint32_t)i; parent[f.bvh[i].b] = (uint32_t)i; }
    for (size_t i = 0; i < n; i++) {
        redirect[i] = (uint32_t)i;
        if ((f.bvh[i].a & BVH_LEAF) || parent[i] == 0xFFFFFFFFu) continue;       // leaves and roots stay
        if (take_out) { if ((*take_out)[i]) redirect[i] = f.bvh[i].c; continue; } // (a view's estimated pass rates decide instead: tune_filter_tree)
        const double ap = area(f.bvh[parent[i]]);
        if (ap > 0.0 && area(f.bvh[i]) > tau * ap) redirect[i] = f.bvh[i].c;
    }
    auto resolve = [&](uint32_t x) { while (x != 0xFFFFFFFFu && redirect[x] != x) x = redirect[x]; return x; };
    for (size_t i = 0; i < n; i++) {
        DFNode& o = f.bvh_f[i];
        o.skip = resolve(o.skip);
        if (!(o.info & FNODE_LEAF)) o.info = resolve(o.info);
    }
}

} // namespace

// Worlds that are ONE bare BVH (every ray walks the tree: the random-spheres scene): which inner nodes leave the filter tree is decided by
// their estimated PASS RATE for a view instead of by box areas.  Any conservative hierarchy over the leaves in the reference's order gives
// the reference's samples (rt_kernel.hip: the ordered-scan form of BVH::hit), so this moves kernel time only, never a bit.  *Measured*
// (round 6, profiles/r06_passrate_contraction.log): taking out the nodes whose MEASURED pass rate exceeds 0.6 ... 0.9 instead of the area
// rule's set makes random spheres 5.5 ... 7.1 % faster; the estimate below — rays through the f64 tree on the host, the leaves' boxes
// standing in for their primitives — reproduces that (+7.3 %) when half of its rays are the view's primary rays (correlation with the
// measured rates 0.79; 0.57 and +1 % without the view), and it loses 2 ... 5 % on scenes whose BVHs stand beside other objects (the final
// scene, the teapot room: their measured rates give nothing either) — hence the one-BVH condition.  4096 rays, ~40 box tests each: < 1 ms.
bool tune_filter_tree(Scene& s, const DCamera<double>& cam) {
    HostFlat& f = s.flat;
    if (std::getenv("RT_NO_FILTER_TUNING")) return false;                 // A/B runs
    if (f.n_top != 1 || f.objects[0].geom_kind != G_BVH || f.objects[0].n_ops != 0 || f.objects[0].medium >= 0 || (f.feats & F_NESTED)) return false;
    if ((f.feats & F_PBR) || (f.feats & ~(uint32_t)(F_BVH | F_TRIS)) == 0u) return false;      // (the scenes the walk-ahead kernel serves: what was measured; a bare mesh was not)
    const size_t n = f.bvh.size();
    if (n < 64 || f.filter_m == 0.0f) return false;
    const uint32_t DONE = 0xFFFFFFFFu, root = f.objects[0].geom_first;
    std::vector<uint32_t> leaves;
    for (size_t i = 0; i < n; i++) if (f.bvh[i].a & BVH_LEAF) leaves.push_back((uint32_t)i);
    std::vector<uint32_t> visits(n, 0u), passes(n, 0u);
    Rng g = rng_for_stream(0x5EEDull, 7u);
    const int N = 4096;
    for (int k = 0; k < N; k++) {
        double o[3], d[3];
        if (k & 1) {                                                      // a primary ray of the view (Camera::get_ray without the lens, camera.rs:51-59)
            const double u = rng_u01(g, 0.0), v = rng_u01(g, 0.0);
            for (int a = 0; a < 3; a++) { o[a] = cam.origin[a]; d[a] = cam.lower_left_corner[a] + u * cam.horizontal[a] + v * cam.vertical[a] - cam.origin[a]; }
        } else {                                                          // a ray that leaves a random leaf's box in a uniform direction (a bounce)
            const DBvhNode<double>& lf = f.bvh[leaves[rng_index(g, (uint32_t)leaves.size())]];
            double len2;
            do { len2 = 0.0; for (int a = 0; a < 3; a++) { d[a] = rng_range(g, -1.0, 1.0); len2 += d[a] * d[a]; } } while (len2 > 1.0 || len2 < 1e-6);
            const double len = std::sqrt(len2);
            double ext[3], diag2 = 0.0;
            for (int a = 0; a < 3; a++) { ext[a] = std::fmin(lf.mx[a] - lf.mn[a], 1e4); diag2 += ext[a] * ext[a]; }      // (a giant ground sphere: stay near the scene)
            for (int a = 0; a < 3; a++) { d[a] /= len; o[a] = 0.5 * (lf.mn[a] + lf.mx[a]) + (rng_u01(g, 0.0) - 0.5) * ext[a] + d[a] * 0.51 * std::sqrt(diag2); }
        }
        double inv[3]; for (int a = 0; a < 3; a++) inv[a] = 1.0 / d[a];
        double closest = std::numeric_limits<double>::infinity();
        for (uint32_t i = root; i != DONE;) {                             // the ordered walk of bvh.rs:77-91 through the skip links
            const DBvhNode<double>& nd = f.bvh[i];
            double t_in = 1e-5, t_out = closest;
            for (int a = 0; a < 3; a++) {
                const double t0 = (nd.mn[a] - o[a]) * inv[a], t1 = (nd.mx[a] - o[a]) * inv[a];
                t_in = std::fmax(t_in, std::fmin(t0, t1)); t_out = std::fmin(t_out, std::fmax(t0, t1));
            }
            const bool ok = t_out > t_in;
            visits[i]++; passes[i] += ok ? 1u : 0u;
            if (nd.a & BVH_LEAF) { if (ok) closest = std::fmin(closest, t_in); i = nd.skip; }
            else i = ok ? nd.c : nd.skip;
        }
    }
    std::vector<unsigned char> take_out(n, 0);
    for (size_t i = 0; i < n; i++)
        take_out[i] = !(f.bvh[i].a & BVH_LEAF) && visits[i] >= 8u && (double)passes[i] > 0.7 * (double)visits[i];
    make_filter_nodes(f, &take_out);                                      // (leaves and roots stay whatever the table says)
    return true;
}

bool flatten_scene(Scene& s) {
    if (s.flat_valid) return true;
    Flattener fl(s);
    if (!fl.run()) { s.error = fl.err; return false; }
    relayout_bvh_by_depth(s.flat);
    // the NaN-free form of AABB::hit (rt_kernel.hip: box_inside_tame) needs finite boxes with min <= max
    s.flat.bvh_tame = true;
    for (const DBvhNode<double>& nd : s.flat.bvh)
        for (int k = 0; k < 3; k++)
            if (!(std::fabs(nd.mn[k]) < 1e300 && std::fabs(nd.mx[k]) < 1e300 && nd.mn[k] <= nd.mx[k])) s.flat.bvh_tame = false;
    make_filter_nodes(s.flat);
    {   // the Cube fast path's scene-wide bound; off when a Cube is inverted on some axis (its six rects then never report a hit
        // through the bounds test, which the fast path's slab logic does not model) or a rect is not finite
        float m = 1.0f; bool ok = true;
        for (const DRect<double>& r : s.flat.rects)
            for (double v : {r.a0, r.a1, r.b0, r.b1, r.k}) { ok = ok && std::isfinite(v) && std::fabs(v) <= 0x1p40; m = std::fmax(m, (float)std::fabs(v) * 1.0000002f); }
        for (const HNode& h : s.nodes) if (h.kind == HNode::CUBE) for (int k = 0; k < 3; k++) ok = ok && h.v[k] <= h.v[3 + k];
        s.flat.rect_m = ok ? m : 0.0f;
    }
    s.flat_valid = true;
    return true;
}

// Camera::new, src/camera.rs:19-49
void camera_new(const rt_camera_args& a, DCamera<double>& out) {
    const double PI = 3.14159265358979323846264338327950288;
    auto sub = [](const double* x, const double* y, double* r) { for (int k = 0; k < 3; k++) r[k] = x[k] - y[k]; };
    auto cross = [](const double* x, const double* y, double* r) {
        r[0] = x[1] * y[2] - x[2] * y[1]; r[1] = x[2] * y[0] - x[0] * y[2]; r[2] = x[0] * y[1] - x[1] * y[0];
    };
    auto normalize = [](double* x) { double l = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); for (int k = 0; k < 3; k++) x[k] = x[k] / l; };
    double theta = PI / 180.0 * a.vfov;
    double viewport_height = 2.0 * std::tan(theta / 2.0);
    double viewport_width = viewport_height * a.aspect;
    double cw[3], cu[3], cv[3];
    sub(a.lookfrom, a.lookat, cw); normalize(cw);
    cross(a.vup, cw, cu); normalize(cu);
    cross(cw, cu, cv);
    for (int k = 0; k < 3; k++) {
        double h = a.focus_dist * viewport_width * cu[k];     // (focus_dist * viewport_width) * cu
        double v = a.focus_dist * viewport_height * cv[k];
        out.horizontal[k] = h; out.vertical[k] = v;
        out.lower_left_corner[k] = a.lookfrom[k] - h / 2.0 - v / 2.0 - a.focus_dist * cw[k];
        out.origin[k] = a.lookfrom[k]; out.cu[k] = cu[k]; out.cv[k] = cv[k];
    }
    out.lens_radius = a.aperture / 2.0; out.time0 = a.time0; out.time1 = a.time1;
}

} // namespace rt
