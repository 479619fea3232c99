// Deterministic binned-SAH BVH builder.
//
// Produces the same tree, node numbering and element permutation as the
// reference builder (BVH::ConstructBVH / SplitOrFinish, renderer/bvh.cu:7-149):
// top-down, 3 axes x 31 candidate planes at pmin + bin * (extent / 32), cost
// 1 + SA(l)/SA(node) * nl + SA(r)/SA(node) * nr in float, first strictly lower
// cost wins, a node becomes a leaf when it holds one element or when the best
// cost exceeds its element count and it holds at most 8, elements with
// centre <= plane go left through the same in-place two-pointer partition, and
// nodes are numbered in pre-order.
//
// The reference re-scans every element for every one of the 93 planes.  Here
// each element is binned once per axis (the planes of one axis are monotone, so
// "first plane the centre is <= to" decides every one of the 31 comparisons) and
// the per-plane boxes and counts come from prefix / suffix unions of 32 bin
// boxes.  min/max are exact and order independent, so the chosen split, the
// child boxes and every cost are the same floats; only the sign of a zero box
// coordinate can differ.
#include <cstring>

#include "host.h"

namespace crt {

namespace {

struct Box6 {
    float lo[3], hi[3];
    void init() { for (int k = 0; k < 3; k++) { lo[k] = CRT_BIGFLOAT; hi[k] = -CRT_BIGFLOAT; } }
    void zero() { for (int k = 0; k < 3; k++) { lo[k] = 0; hi[k] = 0; } }
    void add_point(const float* p) {
        for (int k = 0; k < 3; k++) { if (lo[k] > p[k]) lo[k] = p[k]; if (hi[k] < p[k]) hi[k] = p[k]; }
    }
    // Box::operator+=(Box) adds the other box's two corners as points (scene/rays.cuh:132-135)
    void add_box(const Box6& b) { add_point(b.lo); add_point(b.hi); }
    float area() const {
        const float l = hi[0] - lo[0], w = hi[1] - lo[1], h = hi[2] - lo[2];
        return 2 * l * w + 2 * l * h + 2 * w * h;
    }
};

struct Builder {
    const float* v;
    const uint32_t* f;
    uint32_t nf;
    HostBvh& out;
    std::vector<float> centre[3];     // per element id
    std::vector<Box6> bounds;         // per element id
    bool too_deep = false;

    Builder(const float* v_, const uint32_t* f_, uint32_t nf_, HostBvh& o) : v(v_), f(f_), nf(nf_), out(o) {}

    void precompute() {
        for (int a = 0; a < 3; a++) centre[a].resize(nf);
        bounds.resize(nf);
        for (uint32_t e = 0; e < nf; e++) {
            bounds[e].init();
            for (int a = 0; a < 3; a++) {
                float sum = 0.0f;
                for (int i = 0; i < 3; i++) sum += v[3 * f[3 * e + i] + a];
                centre[a][e] = sum * 0.3333333333333333f;
            }
            for (int i = 0; i < 3; i++) bounds[e].add_point(&v[3 * f[3 * e + i]]);
        }
    }

    uint32_t new_node(uint32_t start, uint32_t count, const Box6& box) {
        const uint32_t id = out.node_count();
        out.leaf.push_back(0);
        out.elem_start.push_back(start);
        out.elem_count.push_back(count);
        out.left.push_back(0); out.right.push_back(0); out.next.push_back(0);
        out.box.insert(out.box.end(), box.lo, box.lo + 3);
        out.box.insert(out.box.end(), box.hi, box.hi + 3);
        return id;
    }

    // returns the depth of the subtree rooted at id
    uint32_t split_or_finish(uint32_t id, uint32_t level) {
        const uint32_t start = out.elem_start[id], count = out.elem_count[id];
        if (count == 1 || level > 128) {
            if (level > 128) too_deep = true;
            out.leaf[id] = 1; out.next[id] = out.node_count();
            return 1;
        }
        Box6 box;
        memcpy(box.lo, &out.box[6 * id], 12); memcpy(box.hi, &out.box[6 * id + 3], 12);
        const float outer = 1.0f / box.area();
        float bin_dim[3];
        for (int a = 0; a < 3; a++) bin_dim[a] = (box.hi[a] - box.lo[a]) * (1.0f / 32.0f);

        unsigned best_axis = 0, best_bin = 0;
        float best_cost = CRT_BIGFLOAT;
        Box6 best_l, best_r;
        best_l.zero(); best_r.zero();     // an unset split result carries zero boxes (bvh.cu:43-52)

        for (unsigned axis = 0; axis < 3; axis++) {
            float plane[32];
            for (int b = 1; b < 32; b++) plane[b] = box.lo[axis] + (float)b * bin_dim[axis];
            Box6 bin_box[33];
            uint32_t bin_cnt[33];
            for (int b = 0; b < 33; b++) { bin_box[b].init(); bin_cnt[b] = 0; }
            const float* cen = centre[axis].data();
            for (uint32_t i = 0; i < count; i++) {
                const uint32_t e = out.elements[start + i];
                const float c = cen[e];
                // first plane index b in [1,31] with c <= plane[b]; 32 when none
                int lo = 1, hi = 32;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (c <= plane[mid]) hi = mid; else lo = mid + 1;
                }
                bin_box[lo].add_box(bounds[e]);
                bin_cnt[lo]++;
            }
            // suffix unions: right side of plane b holds bins b+1 .. 32
            Box6 rbox[33];
            uint32_t rcnt[33];
            rbox[32].init(); rcnt[32] = 0;
            for (int b = 31; b >= 1; b--) {
                rbox[b] = rbox[b + 1];
                if (bin_cnt[b + 1]) rbox[b].add_box(bin_box[b + 1]);
                rcnt[b] = rcnt[b + 1] + bin_cnt[b + 1];
            }
            Box6 lbox; lbox.init();
            uint32_t lcnt = 0;
            for (int b = 1; b < 32; b++) {
                if (bin_cnt[b]) lbox.add_box(bin_box[b]);
                lcnt += bin_cnt[b];
                const float lcost = (lbox.area() * outer) * (float)(size_t)lcnt;
                const float rcost = (rbox[b].area() * outer) * (float)(size_t)rcnt[b];
                const float cost = 1 + lcost + rcost;
                if (cost < best_cost) {
                    best_cost = cost; best_axis = axis; best_bin = (unsigned)b;
                    best_l = lbox; best_r = rbox[b];
                }
            }
        }

        if (best_cost > (float)(size_t)count && count <= 8) {
            out.leaf[id] = 1; out.next[id] = out.node_count();
            return 1;
        }

        out.leaf[id] = 0;
        const float split = box.lo[best_axis] + (float)best_bin * bin_dim[best_axis];
        const float* cen = centre[best_axis].data();
        uint32_t i = 0, j = count;
        while (i < j) {
            if (cen[out.elements[start + i]] <= split) i++;
            else { j--; const uint32_t t = out.elements[start + i]; out.elements[start + i] = out.elements[start + j]; out.elements[start + j] = t; }
        }
        const uint32_t l = new_node(start, i, best_l);
        out.left[id] = l;
        const uint32_t dl = split_or_finish(l, level + 1);
        const uint32_t r = new_node(start + i, count - i, best_r);
        out.right[id] = r;
        const uint32_t dr = split_or_finish(r, level + 1);
        out.next[id] = out.node_count();
        return 1 + (dl > dr ? dl : dr);
    }
};

}  // namespace

bool bvh_build(const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
               const float* box_min, const float* box_max, HostBvh& out, std::string& err) {
    out = HostBvh();
    for (uint32_t i = 0; i < 3 * nf; i++)
        if (f[i] >= nv) { err = "face index out of range"; return false; }
    out.elements.resize(nf);
    for (uint32_t i = 0; i < nf; i++) out.elements[i] = i;
    Builder b(v, f, nf, out);
    b.precompute();
    Box6 root;
    memcpy(root.lo, box_min, 12); memcpy(root.hi, box_max, 12);
    b.new_node(0, nf, root);
    out.depth = b.split_or_finish(0, 1);
    if (b.too_deep || out.depth > 64) {
        err = "BVH depth " + std::to_string(out.depth) + " exceeds the maximum of 64";
        return false;
    }
    return true;
}

}  // namespace crt
