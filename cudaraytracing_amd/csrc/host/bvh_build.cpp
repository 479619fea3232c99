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
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

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

// Node arrays of one (sub)tree, indices relative to its own root.
struct Tree {
    std::vector<uint8_t> leaf;
    std::vector<uint32_t> elem_start, elem_count, left, right, next;
    std::vector<float> box;
    uint32_t depth = 0;
    uint32_t size() const { return (uint32_t)leaf.size(); }
    uint32_t new_node(uint32_t start, uint32_t count, const Box6& b) {
        const uint32_t id = size();
        leaf.push_back(0);
        elem_start.push_back(start);
        elem_count.push_back(count);
        left.push_back(0); right.push_back(0); next.push_back(0);
        box.insert(box.end(), b.lo, b.lo + 3);
        box.insert(box.end(), b.hi, b.hi + 3);
        return id;
    }
};

struct Split {
    bool leaf = false;
    unsigned axis = 0, bin = 0;
    Box6 l, r;
    float bin_dim[3];
};

struct Bins {
    Box6 box[33];
    uint32_t cnt[33];
    void init() { for (int b = 0; b < 33; b++) { box[b].init(); cnt[b] = 0; } }
};

struct Builder {
    const float* v;
    const uint32_t* f;
    uint32_t nf;
    uint32_t* elements;               // the permutation under construction; subtrees own disjoint ranges
    std::vector<float> centre[3];     // per element id
    std::vector<Box6> bounds;         // per element id
    std::atomic<bool> too_deep{false};
    unsigned threads = 1;

    Builder(const float* v_, const uint32_t* f_, uint32_t nf_, uint32_t* el) : v(v_), f(f_), nf(nf_), elements(el) {}

    template <class Fn>
    void parallel_ranges(uint32_t count, Fn fn) const {
        const unsigned t = (threads > 1 && count >= 65536) ? threads : 1;
        if (t == 1) { fn(0u, 0u, count); return; }
        std::vector<std::thread> pool;
        for (unsigned k = 0; k < t; k++) {
            const uint32_t a = (uint32_t)((uint64_t)count * k / t), b = (uint32_t)((uint64_t)count * (k + 1) / t);
            pool.emplace_back(fn, k, a, b);
        }
        for (auto& th : pool) th.join();
    }

    void precompute() {
        for (int a = 0; a < 3; a++) centre[a].resize(nf);
        bounds.resize(nf);
        parallel_ranges(nf, [this](unsigned, uint32_t e0, uint32_t e1) {
            for (uint32_t e = e0; e < e1; e++) {
                bounds[e].init();
                for (int a = 0; a < 3; a++) {
                    float sum = 0.0f;
                    for (int i = 0; i < 3; i++) sum += v[3 * f[3 * e + i] + a];
                    centre[a][e] = sum * 0.3333333333333333f;
                }
                for (int i = 0; i < 3; i++) bounds[e].add_point(&v[3 * f[3 * e + i]]);
            }
        });
    }

    void bin_range(unsigned axis, const float* plane, uint32_t begin, uint32_t end, Bins& bins) const {
        const float* cen = centre[axis].data();
        for (uint32_t i = begin; i < end; i++) {
            const uint32_t e = elements[i];
            const float c = cen[e];
            // first plane index b in [1,31] with c <= plane[b]; 32 when none
            int lo = 1, hi = 32;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (c <= plane[mid]) hi = mid; else lo = mid + 1;
            }
            bins.box[lo].add_box(bounds[e]);
            bins.cnt[lo]++;
        }
    }

    // The split decision of one node (SplitOrFinish up to the partition, renderer/bvh.cu:37-118).  Large nodes bin
    // their elements on several threads; bin boxes are min / max unions and bin counts are sums, so the result does
    // not depend on how the range is cut.
    Split choose_split(uint32_t start, uint32_t count, const Box6& box) const {
        Split sp;
        const float outer = 1.0f / box.area();
        for (int a = 0; a < 3; a++) sp.bin_dim[a] = (box.hi[a] - box.lo[a]) * (1.0f / 32.0f);
        float best_cost = CRT_BIGFLOAT;
        sp.l.zero(); sp.r.zero();         // an unset split result carries zero boxes (bvh.cu:43-52)

        for (unsigned axis = 0; axis < 3; axis++) {
            float plane[32];
            for (int b = 1; b < 32; b++) plane[b] = box.lo[axis] + (float)b * sp.bin_dim[axis];
            Bins bins;
            bins.init();
            if (threads > 1 && count >= 65536) {
                std::vector<Bins> part(threads);
                for (auto& p : part) p.init();
                parallel_ranges(count, [&](unsigned k, uint32_t a, uint32_t b) { bin_range(axis, plane, start + a, start + b, part[k]); });
                for (const auto& p : part)
                    for (int b = 1; b <= 32; b++)
                        if (p.cnt[b]) { bins.box[b].add_box(p.box[b]); bins.cnt[b] += p.cnt[b]; }
            } else {
                bin_range(axis, plane, start, start + count, bins);
            }
            // suffix unions: right side of plane b holds bins b+1 .. 32
            Box6 rbox[33];
            uint32_t rcnt[33];
            rbox[32].init(); rcnt[32] = 0;
            for (int b = 31; b >= 1; b--) {
                rbox[b] = rbox[b + 1];
                if (bins.cnt[b + 1]) rbox[b].add_box(bins.box[b + 1]);
                rcnt[b] = rcnt[b + 1] + bins.cnt[b + 1];
            }
            Box6 lbox; lbox.init();
            uint32_t lcnt = 0;
            for (int b = 1; b < 32; b++) {
                if (bins.cnt[b]) lbox.add_box(bins.box[b]);
                lcnt += bins.cnt[b];
                const float lcost = (lbox.area() * outer) * (float)(size_t)lcnt;
                const float rcost = (rbox[b].area() * outer) * (float)(size_t)rcnt[b];
                const float cost = 1 + lcost + rcost;
                if (cost < best_cost) {
                    best_cost = cost; sp.axis = axis; sp.bin = (unsigned)b;
                    sp.l = lbox; sp.r = rbox[b];
                }
            }
        }
        sp.leaf = best_cost > (float)(size_t)count && count <= 8;
        return sp;
    }

    // in-place two-pointer partition of the node's element range (bvh.cu:120-134); returns the left count
    uint32_t partition(uint32_t start, uint32_t count, const Box6& box, const Split& sp) {
        const float split = box.lo[sp.axis] + (float)sp.bin * sp.bin_dim[sp.axis];
        const float* cen = centre[sp.axis].data();
        uint32_t i = 0, j = count;
        while (i < j) {
            if (cen[elements[start + i]] <= split) i++;
            else { j--; const uint32_t t = elements[start + i]; elements[start + i] = elements[start + j]; elements[start + j] = t; }
        }
        return i;
    }

    // Sequential pre-order construction of the subtree under node id of t; returns its depth.
    uint32_t split_or_finish(Tree& t, uint32_t id, uint32_t level) {
        const uint32_t start = t.elem_start[id], count = t.elem_count[id];
        if (count == 1 || level > 128) {
            if (level > 128) too_deep = true;
            t.leaf[id] = 1; t.next[id] = t.size();
            return 1;
        }
        Box6 box;
        memcpy(box.lo, &t.box[6 * id], 12); memcpy(box.hi, &t.box[6 * id + 3], 12);
        const Split sp = choose_split(start, count, box);
        if (sp.leaf) {
            t.leaf[id] = 1; t.next[id] = t.size();
            return 1;
        }
        t.leaf[id] = 0;
        const uint32_t i = partition(start, count, box, sp);
        const uint32_t l = t.new_node(start, i, sp.l);
        t.left[id] = l;
        const uint32_t dl = split_or_finish(t, l, level + 1);
        const uint32_t r = t.new_node(start + i, count - i, sp.r);
        t.right[id] = r;
        const uint32_t dr = split_or_finish(t, r, level + 1);
        t.next[id] = t.size();
        return 1 + (dl > dr ? dl : dr);
    }
};

// The top of a large tree: nodes above the size at which subtrees are handed to worker threads.
struct TopNode {
    uint32_t start = 0, count = 0, level = 1;
    Box6 box;
    bool leaf = false;
    int left = -1, right = -1;        // top-node indices
    int task = -1;                    // or: index of the subtree built by a worker
    uint32_t size = 0, depth = 0;     // filled bottom-up
};

struct ParallelBuild {
    Builder& b;
    uint32_t task_limit;              // nodes with at most this many elements become worker tasks
    std::vector<TopNode> top;
    std::deque<Tree> trees;           // one per task; a deque keeps references valid while it grows
    std::vector<int> task_root;       // top-node index of each task
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> ready;
    bool closed = false;

    ParallelBuild(Builder& b_, uint32_t limit) : b(b_), task_limit(limit) {}

    int add_top(uint32_t start, uint32_t count, uint32_t level, const Box6& box) {
        TopNode n; n.start = start; n.count = count; n.level = level; n.box = box;
        top.push_back(n);
        return (int)top.size() - 1;
    }

    void worker() {
        for (;;) {
            int t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !ready.empty() || closed; });
                if (ready.empty()) return;
                t = ready.front(); ready.pop_front();
            }
            TopNode n;
            Tree* tree;
            {
                std::lock_guard<std::mutex> lk(mu);   // top / trees may be growing on the main thread
                n = top[task_root[t]];
                tree = &trees[t];
            }
            tree->new_node(n.start, n.count, n.box);
            tree->depth = b.split_or_finish(*tree, 0, n.level);
        }
    }

    // main thread: split the big nodes (their binning runs on all threads), queue the rest
    void expand(int idx) {
        TopNode n;
        { std::lock_guard<std::mutex> lk(mu); n = top[idx]; }
        if (n.count <= task_limit || n.level > 48) {
            std::lock_guard<std::mutex> lk(mu);
            trees.emplace_back();
            task_root.push_back(idx);
            top[idx].task = (int)trees.size() - 1;
            ready.push_back(top[idx].task);
            cv.notify_one();
            return;
        }
        const Split sp = b.choose_split(n.start, n.count, n.box);
        if (sp.leaf || n.count == 1) {            // cannot happen above the task limit, kept for completeness
            std::lock_guard<std::mutex> lk(mu);
            top[idx].leaf = true;
            return;
        }
        const uint32_t i = b.partition(n.start, n.count, n.box, sp);
        int l, r;
        {
            std::lock_guard<std::mutex> lk(mu);
            l = add_top(n.start, i, n.level + 1, sp.l);
            r = add_top(n.start + i, n.count - i, n.level + 1, sp.r);
            top[idx].left = l; top[idx].right = r;
        }
        // larger side first so that its tasks reach the workers early
        if (i >= n.count - i) { expand(l); expand(r); } else { expand(r); expand(l); }
    }

    void measure(int idx) {
        TopNode& n = top[idx];
        if (n.task >= 0) { n.size = trees[n.task].size(); n.depth = trees[n.task].depth; return; }
        if (n.leaf) { n.size = 1; n.depth = 1; return; }
        measure(n.left); measure(n.right);
        n.size = 1 + top[n.left].size + top[n.right].size;
        const uint32_t dl = top[n.left].depth, dr = top[n.right].depth;
        n.depth = 1 + (dl > dr ? dl : dr);
    }

    // pre-order placement: node, left subtree, right subtree (the numbering of the sequential build)
    void place(int idx, uint32_t base, HostBvh& out) {
        const TopNode& n = top[idx];
        if (n.task >= 0) {
            const Tree& t = trees[n.task];
            const uint32_t m = t.size();
            memcpy(&out.leaf[base], t.leaf.data(), m);
            memcpy(&out.elem_start[base], t.elem_start.data(), 4ull * m);
            memcpy(&out.elem_count[base], t.elem_count.data(), 4ull * m);
            memcpy(&out.box[6ull * base], t.box.data(), 24ull * m);
            for (uint32_t k = 0; k < m; k++) {
                out.left[base + k] = t.leaf[k] ? 0 : t.left[k] + base;
                out.right[base + k] = t.leaf[k] ? 0 : t.right[k] + base;
                out.next[base + k] = t.next[k] + base;
            }
            return;
        }
        out.leaf[base] = n.leaf ? 1 : 0;
        out.elem_start[base] = n.start; out.elem_count[base] = n.count;
        memcpy(&out.box[6ull * base], n.box.lo, 12); memcpy(&out.box[6ull * base + 3], n.box.hi, 12);
        out.next[base] = base + n.size;
        if (n.leaf) { out.left[base] = 0; out.right[base] = 0; return; }
        out.left[base] = base + 1;
        out.right[base] = base + 1 + top[n.left].size;
        place(n.left, base + 1, out);
        place(n.right, base + 1 + top[n.left].size, out);
    }
};

unsigned build_threads() {
    for (const char* name : {"CRT_BUILD_THREADS", "OMP_NUM_THREADS"})
        if (const char* e = getenv(name)) { const int n = atoi(e); if (n > 0) return (unsigned)(n > 64 ? 64 : n); }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw == 0 ? 1 : (hw > 16 ? 16 : hw);
}

}  // namespace

bool bvh_build(const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
               const float* box_min, const float* box_max, HostBvh& out, std::string& err) {
    out = HostBvh();
    for (uint32_t i = 0; i < 3 * nf; i++)
        if (f[i] >= nv) { err = "face index out of range"; return false; }
    out.elements.resize(nf);
    for (uint32_t i = 0; i < nf; i++) out.elements[i] = i;
    Builder b(v, f, nf, out.elements.data());
    Box6 root;
    memcpy(root.lo, box_min, 12); memcpy(root.hi, box_max, 12);

    // Small meshes: one sequential pass.  Large ones: the top of the tree is split on the calling thread (binning
    // spread over all threads) and every subtree below the task limit is built by a worker into arrays of its own;
    // a final pass lays the pieces out in pre-order, which is the numbering the sequential pass produces.
    b.threads = nf >= 262144 ? build_threads() : 1;
    b.precompute();
    if (b.threads <= 1) {
        Tree t;
        t.new_node(0, nf, root);
        t.depth = b.split_or_finish(t, 0, 1);
        out.leaf = std::move(t.leaf); out.elem_start = std::move(t.elem_start); out.elem_count = std::move(t.elem_count);
        out.left = std::move(t.left); out.right = std::move(t.right); out.next = std::move(t.next); out.box = std::move(t.box);
        out.depth = t.depth;
    } else {
        ParallelBuild pb(b, nf / (8 * b.threads) + 1);
        std::vector<std::thread> workers;
        for (unsigned k = 0; k < b.threads; k++) workers.emplace_back([&pb] { pb.worker(); });
        const int root_idx = pb.add_top(0, nf, 1, root);
        pb.expand(root_idx);
        { std::lock_guard<std::mutex> lk(pb.mu); pb.closed = true; }
        pb.cv.notify_all();
        for (auto& w : workers) w.join();
        pb.measure(root_idx);
        const uint32_t total = pb.top[root_idx].size;
        out.leaf.resize(total); out.elem_start.resize(total); out.elem_count.resize(total);
        out.left.resize(total); out.right.resize(total); out.next.resize(total); out.box.resize(6ull * total);
        pb.place(root_idx, 0, out);
        out.depth = pb.top[root_idx].depth;
    }
    if (b.too_deep || out.depth > 64) {
        err = "BVH depth " + std::to_string(out.depth) + " exceeds the maximum of 64";
        return false;
    }
    return true;
}

}  // namespace crt
