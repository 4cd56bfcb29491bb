// pcr_octree.h -- linear octree over a Morton-sorted cloud and its 8-lane ("octet") traversal.
//
// Nodes of level l are the runs of points that share the Morton prefix key >> 3l; the children of a node are the
// <= 8 runs of level l-1 inside it, stored contiguously, each with the TIGHT axis-aligned box of its points.  Cells
// of one level are disjoint, so a query ball only meets the few cells it geometrically touches -- unlike fixed
// groups of consecutive Morton points, whose boxes straddle the curve's jumps (measured: p90 leaf diagonal 3.5 m at
// 0.1 m voxels, ~500 box tests per 1-NN query; DESIGN.md "spatial index").
//
// Layout (device), level index li = Morton level - l0, node j of level li lives at slot off[li] + j:
//   nodes[2 slot]     = (lo.x, lo.y, lo.z, int first)   first POINT (li = 0) / first CHILD in level li-1 (li > 0)
//   nodes[2 slot + 1] = (hi.x, hi.y, hi.z, int count)   number of points / children (<= 8 children)
//   up[slot]          = (parent node of level li+1, first sibling, number of siblings, 0): one read per climb
//   leaf_of[i]        = leaf (li = 0 node) holding point i ;  keys[i] = Morton key of point i
//   meta              = OctMeta: levels, counts, offsets, key geometry (origin + unit of the integer lattice)
//
// Search = bottom-up from a start leaf (the query's own leaf, or the leaf of the previous match): scan the leaf,
// then climb; at every ancestor test its other children and walk down those the ball meets; stop as soon as the
// ball lies inside the ancestor's Morton cell (then nothing outside the subtree can be closer).  With a warm bound
// this touches 2-4 nodes instead of a root-to-leaf spine per query.
//
// 8 consecutive lanes serve one query.  Lane c tests child c of a node (one coalesced 256-B read), the verdicts
// return as one byte of a wave ballot; pending siblings live in a per-octet LDS stack (<= 16 levels).
#pragma once
#include "pcr_device.h"

#define OCT 8
#define OCT_MAXL 16          // stored levels (leaf level .. root)
#define OCT_KEY_LEVELS 22    // Morton levels of a 63-bit key (+1)

struct OctMeta {
    int n;                   // points
    int l0;                  // Morton level of the leaves
    int nl;                  // stored levels; root = level index nl-1 (exactly one node)
    int cnt[OCT_MAXL];
    int off[OCT_MAXL];
    int total;               // node slots used
    float org[3];            // lattice origin:   integer coordinate i  <->  [org + i*unit, org + (i+1)*unit)
    float unit[3];
};

struct OctView {             // what a kernel needs to walk a tree
    const float4 *pts;
    const float4 *nodes;
    const int4 *up;
    const OctMeta *meta;
    const int *leaf_of;
    const uint64_t *keys;
    const int4 *pinfo;       // point -> (node, its first point, its point count, level): the level-1 node holding the point
                             // (the leaf when the tree has one level) -- the "fat leaf" a warm-started search scans first
    const int2 *l1rng;       // level-1 node -> (first point, point count)
};

__host__ __device__ static inline uint32_t pcr_compact21(uint64_t x) {
    x &= 0x1249249249249249ull;
    x = (x | (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x | (x >> 4)) & 0x100f00f00f00f00full;
    x = (x | (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x | (x >> 16)) & 0x1f00000000ffffull;
    x = (x | (x >> 32)) & 0x1fffffull;
    return (uint32_t)x;
}

// per-octet traversal stack (LDS): [level][octet in block]
template <int OPB>
struct OctStack {
    int cs[OCT_MAXL][OPB];
    unsigned char mask[OCT_MAXL][OPB];
};

// Test the <= 8 nodes [cs, cs+cnt) of level li against the ball; returns the octet's byte of the wave ballot and,
// through (first, count), this lane's node payload.  MUST be called by all 64 lanes.
__device__ static inline uint32_t oct_test_nodes(const OctView &t, const OctMeta &m, bool want, int li, int cs, int cnt,
                                                 float qx, float qy, float qz, float bound, int ol, int oct,
                                                 int &first, int &count) {
    bool pass = false;
    first = 0; count = 0;
    if (want && ol < cnt) {
        const size_t j = (size_t)(m.off[li] + cs + ol);
        const float4 lo = t.nodes[2 * j], hi = t.nodes[2 * j + 1];
        pass = pcr_box_d2(lo, hi, qx, qy, qz) < bound;
        first = __float_as_int(lo.w); count = __float_as_int(hi.w);
    }
    const unsigned long long bal = __ballot(pass);
    return (uint32_t)(bal >> (oct * 8)) & 0xffu;
}

// Is the ball (q, sqrt(bound)) strictly inside the Morton cell of level `lvl` that contains lattice point (ix,iy,iz)?
__device__ static inline bool oct_ball_in_cell(const OctMeta &m, uint32_t ix, uint32_t iy, uint32_t iz, int lvl, float qx, float qy,
                                               float qz, float bound) {
    if (!(bound < 3.0e38f)) return false;
    const float r = sqrtf(bound) * 1.00001f;
    const float w = (float)(1u << lvl);
    const float q[3] = {qx, qy, qz};
    const uint32_t ic[3] = {ix, iy, iz};
    bool in = true;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float side = w * m.unit[d];
        const float cmin = m.org[d] + (float)((ic[d] >> lvl) << lvl) * m.unit[d];
        const float eps = 1e-4f * side + 1e-6f * fabsf(cmin);
        in = in && (q[d] - r > cmin + eps) && (q[d] + r < cmin + side - eps);
    }
    return in;
}

// ---------------------------------------------------------------------------------------------------------------
// Bottom-up exact search driver for one query per octet.  All 64 lanes call it; `live`, `start_leaf` are
// octet-uniform.  bound() -> current squared radius (octet-uniform, may shrink);  visit(first, count) is a WAVE-WIDE
// operation that tests `count` points starting at `first` (count == 0: idle octet);  skip(first, count) -> true if a
// leaf need not be scanned at all (e.g. already covered by the seed range).
// s_*: what the walk needs about its start leaf (points, key of the first point, up-link).  Callers that know the leaf
// one hop earlier (k_icp_nn: from the previous match) issue these loads together with their own, so that the record
// arrives with the leaf's points instead of after them: the search is a chain of dependent loads and every hop costs
// ~1 us of wavefront life.
template <int OPB, class BoundFn, class VisitFn, class SkipFn>
__device__ static inline void oct_search(const OctView &t, const OctMeta &m, OctStack<OPB> &stk, bool live, int start_node, int start_li,
                                         int s_first, int s_count, uint64_t s_key, int s_parent, int s_sib, int s_nsib,
                                         float qx, float qy, float qz, BoundFn bound, VisitFn visit, SkipFn skip, int ol, int oct, int ob,
                                         int *visits) {
    bool done = !live || m.nl < 1;
    // ---- the start node's own points (a leaf, or a level-1 node taken as one fat leaf)
    visit(done ? 0 : s_first, done ? 0 : s_count);
    uint32_t ix = 0, iy = 0, iz = 0;
    if (!done) { ix = pcr_compact21(s_key); iy = pcr_compact21(s_key >> 1); iz = pcr_compact21(s_key >> 2); }
    int anc = start_node, anc_li = start_li;              // subtree already covered
    int li = start_li, cs = 0, base_li = start_li;        // current sibling list (nodes of level li starting at cs), pending mask
    uint32_t mask = 0;
    bool first_ascent = true;
    int pay_f = 0, pay_c = 0, pay_li = -1;                // lane c: (first, count) of child c of the list tested last (level pay_li)
    for (;;) {
        bool ascend = false;
        while (!done && mask == 0) {
            if (li < base_li) { li++; cs = stk.cs[li][ob]; mask = stk.mask[li][ob]; }
            else {
                if (anc_li >= m.nl - 1 || oct_ball_in_cell(m, ix, iy, iz, m.l0 + anc_li, qx, qy, qz, bound())) done = true;
                else ascend = true;
                break;
            }
        }
        if (__ballot(!done) == 0ull) break;
        // ---- climb one level: the parent's other children become the sibling list
        if (__ballot(ascend) != 0ull) {
            int p = 0, pf = 0, pc = 0;
            if (ascend) {
                if (first_ascent) { p = s_parent; pf = s_sib; pc = s_nsib; }
                else { const int4 u = t.up[m.off[anc_li] + anc]; p = u.x; pf = u.y; pc = u.z; }
            }
            int f, c;
            const uint32_t nm = oct_test_nodes(t, m, ascend, anc_li, pf, pc, qx, qy, qz, bound(), ol, oct, f, c);
            if (ascend) {
                if (visits) *visits += 1 << 10;
                li = anc_li; base_li = anc_li; cs = pf;
                mask = nm & ~(1u << (anc - pf));
                anc = p; anc_li++;
                first_ascent = false;
                pay_f = f; pay_c = c; pay_li = li;
            }
        }
        // ---- pop one pending sibling: a leaf is scanned, an inner node is opened.  Its (first, count) sit in lane c of
        // the octet when the list was tested last (a shuffle instead of a dependent load), else they are re-read
        int vf = 0, vc = 0; bool descend = false; int dcs = 0, dcnt = 0;
        const bool popping = !done && mask != 0;
        const int pc_ = popping ? __builtin_ctz(mask) : 0;
        const int sh_f = __shfl(pay_f, pc_, OCT), sh_c = __shfl(pay_c, pc_, OCT);
        if (popping) {
            mask &= mask - 1;
            int nf = sh_f, nc = sh_c;
            if (pay_li != li) {
                const size_t j = (size_t)(m.off[li] + cs + pc_);
                nf = __float_as_int(t.nodes[2 * j].w); nc = __float_as_int(t.nodes[2 * j + 1].w);
            }
            if (visits) *visits += 1;
            if (li == 0) { if (!skip(nf, nc)) { vf = nf; vc = nc; if (visits) *visits += 1 << 20; } }
            else {
                if (ol == 0) { stk.cs[li][ob] = cs; stk.mask[li][ob] = (unsigned char)mask; }
                li--; cs = nf; dcs = nf; dcnt = nc; descend = true;
            }
        }
        if (__ballot(vc > 0) != 0ull) visit(vf, vc);
        if (__ballot(descend) != 0ull) {
            int f, c;
            const uint32_t nm = oct_test_nodes(t, m, descend, li, dcs, dcnt, qx, qy, qz, bound(), ol, oct, f, c);
            if (descend) { mask = nm; pay_f = f; pay_c = c; pay_li = li; }
        }
    }
}

// greedy nearest-box descent from the root (cold start of a 1-NN query): returns a leaf id (octet-uniform)
__device__ static inline int oct_greedy_leaf(const OctView &t, const OctMeta &m, bool want, float qx, float qy, float qz, int ol) {
    int li = m.nl - 1, node = 0;
    while (__ballot(want && li > 0) != 0ull) {
        float d = 3.4e38f; int c = 0, cs = 0;
        if (want && li > 0) {
            const size_t jn = (size_t)(m.off[li] + node);
            cs = __float_as_int(t.nodes[2 * jn].w);
            const int cnt = __float_as_int(t.nodes[2 * jn + 1].w);
            if (ol < cnt) {
                const size_t j = (size_t)(m.off[li - 1] + cs + ol);
                d = pcr_box_d2(t.nodes[2 * j], t.nodes[2 * j + 1], qx, qy, qz); c = ol;
            }
        }
#pragma unroll
        for (int o = 1; o < OCT; o <<= 1) {
            const float od = __shfl_xor(d, o, OCT); const int oc = __shfl_xor(c, o, OCT);
            if (od < d || (od == d && oc < c)) { d = od; c = oc; }
        }
        if (want && li > 0) { node = cs + c; li--; }
    }
    return node;
}


// ---------------------------------------------------------------------------------------------------------------
// GROUP search: the 8 queries of a wavefront (8 Morton-consecutive points) share ONE walk.  All state is
// wave-uniform (plain control flow, no ballots to agree on phases); a node is opened when ANY of the 8 queries can
// still improve inside it, and all 64 lanes work on every step: node test = 8 queries x 8 children, leaf scan =
// 8 queries x 8 candidates.  Used by the k-NN kernels, where the 8 balls overlap almost completely.
struct OctGroupStack {            // per wavefront (LDS)
    int cs[OCT_MAXL];
    int mask[OCT_MAXL];
    int first[OCT_MAXL][OCT];
    int count[OCT_MAXL][OCT];
    unsigned gd2[OCT_MAXL][OCT];   // per child: min over the group's live queries of the squared box distance (float bits)
};

__device__ static inline float wave_or_octets_max(float v) {     // max over the 8 octets (v is octet-uniform)
#pragma unroll
    for (int o = OCT; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// worst() -> this lane's (octet-uniform) squared bound; visit(first, count) scans points for all 8 queries;
// skip(first, count) -> leaf already covered.  q*: this lane's query; live: octet-uniform.
template <class WorstFn, class VisitFn, class SkipFn>
__device__ static inline void oct_search_group(const OctView &t, const OctMeta &m, OctGroupStack &stk, bool live, int start_leaf,
                                               float qx, float qy, float qz, WorstFn worst, VisitFn visit, SkipFn skip, int ol, int *visits) {
    if (m.nl < 1 || __ballot(live) == 0ull) return;
    // group box (live queries only)
    float glo[3] = {live ? qx : 3.4e38f, live ? qy : 3.4e38f, live ? qz : 3.4e38f};
    float ghi[3] = {live ? qx : -3.4e38f, live ? qy : -3.4e38f, live ? qz : -3.4e38f};
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
        for (int o = OCT; o < 64; o <<= 1) { glo[d] = fminf(glo[d], __shfl_xor(glo[d], o, 64)); ghi[d] = fmaxf(ghi[d], __shfl_xor(ghi[d], o, 64)); }
    start_leaf = __builtin_amdgcn_readfirstlane(start_leaf);

    // children [cs, cs+cnt) of level li: bit c set when any live query can still improve inside child c; gd2[li][c] <- the
    // smallest box distance over the group's queries (the pop order: nearest child first, so the bounds tighten before
    // the farther children are looked at -- and are mostly gone by then)
    auto test = [&](int li, int cs, int cnt) -> int {
        bool pass = false; int f = 0, c = 0; float d2 = __builtin_inff();
        if ((threadIdx.x & 63) < OCT) stk.gd2[li][ol] = 0x7f800000u;
        if (ol < cnt) {
            const size_t j = (size_t)(m.off[li] + cs + ol);
            const float4 lo = t.nodes[2 * j], hi = t.nodes[2 * j + 1];
            if (live) d2 = pcr_box_d2(lo, hi, qx, qy, qz);
            pass = d2 < worst();
            f = __float_as_int(lo.w); c = __float_as_int(hi.w);
        }
        if (pass) atomicMin(&stk.gd2[li][ol], __float_as_uint(d2));
        unsigned long long bal = __ballot(pass);
        bal |= bal >> 32; bal |= bal >> 16; bal |= bal >> 8;
        if ((threadIdx.x & 63) < OCT) { stk.first[li][ol] = f; stk.count[li][ol] = c; }
        return (int)(bal & 0xffull);
    };
    // nearest pending child of level li (mask != 0); returns -1 when no query can improve in ANY pending child any more
    auto pop = [&](int li, int mask) -> int {
        const unsigned g = ((mask >> ol) & 1) ? stk.gd2[li][ol] : 0xffffffffu;
        unsigned k = g;
        k = min(k, (unsigned)pcr_dpp_i<PCR_DPP_XOR1>((int)k));
        k = min(k, (unsigned)pcr_dpp_i<PCR_DPP_XOR2>((int)k));
        k = min(k, (unsigned)pcr_dpp_i<PCR_DPP_HMIRROR>((int)k));
        if (__ballot(live && __uint_as_float(k) < worst()) == 0ull) return -1;
        return __builtin_ctz((unsigned)(__ballot(g == k) & 0xffull));
    };
    auto contained = [&](int lvl, uint32_t ix, uint32_t iy, uint32_t iz) -> bool {
        const float wmax = wave_or_octets_max(live ? worst() : 0.0f);
        if (!(wmax < 3.0e38f)) return false;
        const float r = sqrtf(wmax) * 1.00001f;
        const float w = (float)(1u << lvl);
        const uint32_t ic[3] = {ix, iy, iz};
        bool in = true;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float side = w * m.unit[d];
            const float cmin = m.org[d] + (float)((ic[d] >> lvl) << lvl) * m.unit[d];
            const float eps = 1e-4f * side + 1e-6f * fabsf(cmin);
            in = in && (glo[d] - r > cmin + eps) && (ghi[d] + r < cmin + side - eps);
        }
        return in;
    };

    // ---- the start leaf
    int lf, lc;
    {
        const size_t j = (size_t)(m.off[0] + start_leaf);
        lf = __float_as_int(t.nodes[2 * j].w); lc = __float_as_int(t.nodes[2 * j + 1].w);
    }
    if (!skip(lf, lc)) visit(lf, lc);
    const uint64_t key = t.keys[lf];
    const uint32_t ix = pcr_compact21(key), iy = pcr_compact21(key >> 1), iz = pcr_compact21(key >> 2);
    int anc = start_leaf, anc_li = 0, li = 0, cs = 0, base_li = 0, mask = 0;
    for (;;) {
        while (mask == 0) {
            if (li < base_li) { li++; cs = stk.cs[li]; mask = stk.mask[li]; }
            else {
                if (anc_li >= m.nl - 1 || contained(m.l0 + anc_li, ix, iy, iz)) return;
                const int4 u = t.up[m.off[anc_li] + anc];
                const int nm = test(anc_li, u.y, u.z);
                li = anc_li; base_li = anc_li; cs = u.y;
                mask = nm & ~(1 << (anc - u.y));
                anc = u.x; anc_li++;
            }
        }
        const int c = pop(li, mask);
        if (c < 0) { mask = 0; continue; }
        mask &= ~(1 << c);
        const int nf = stk.first[li][c], nc = stk.count[li][c];
        if (visits) *visits += 1;
        if (li == 0) { if (!skip(nf, nc)) visit(nf, nc); }
        else {
            stk.cs[li] = cs; stk.mask[li] = mask;
            li--; cs = nf;
            mask = test(li, nf, nc);
        }
    }
}

