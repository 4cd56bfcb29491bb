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

// ---------------------------------------------------------------------------------------------------------------
// CELL HASH over a voxel-lattice cloud: the points of a Morton cell of level L (edge c = unit * 2^L) are a contiguous range of the
// sorted cloud; `tab` is an open-addressing hash from the cell's Morton code (key >> 3L) to that range.  A ball of radius r <= 0.49 c
// around q meets only the 2 x 2 x 2 block of cells on q's side, so the nearest point inside the ball comes out of 8 INDEPENDENT lookups
// (one per lane of the octet) and then coalesced point loads -- two or three round trips where the bottom-up tree walk chains 6-20
// dependent loads (the GICP iteration kernel spent 26 of its 38 us in the slowest workgroup's walks).  (A table sized by the
// cells -- a tenth to a fortieth of the points at the levels used, counted on the device -- instead of by the points was measured: no
// gain at 200k or 2M points, the lookups are not what the searches wait for.)
struct GridEntry { unsigned long long code; int first, count; };     // 16 B: one load per probe; code 0xffff... = empty slot
struct GridView {
    const GridEntry *tab;
    const unsigned *dmask; int L;             // slots - 1 of the table, ON THE DEVICE: the table is sized by the cloud's point COUNT (round 5), which only the device knows
                                              //   (sized by the capacity, a 1.6M-point scale of a 2M-point cloud cleared 64 MB per call: 4.5 % of config 5's kernel time)
    float org[3], inv_unit[3], cell[3];       // lattice origin, 1 / voxel, cell edge (voxel * 2^L)
};
#define PCR_GRID_EMPTY 0xffffffffffffffffull
// slots of the table of a cloud of n points: the power of two from 2 n (load <= 0.5), at least 64; the same rule in k_grid_clear, k_grid_build and the host's allocation
__host__ __device__ static inline unsigned pcr_grid_slots(unsigned n) { unsigned s = 64; while (s < n * 2u && s < (1u << 30)) s <<= 1; return s; }
__host__ __device__ static inline unsigned pcr_grid_hash(unsigned long long code, unsigned mask) {
    return (unsigned)((code * 0x9E3779B97F4A7C15ull) >> 40) & mask;
}
__device__ static inline int2 pcr_grid_lookup(const GridView &g, unsigned mask, int x, int y, int z) {
    const int lim = (1 << 21) >> g.L;
    if (x < 0 || y < 0 || z < 0 || x >= lim || y >= lim || z >= lim) return make_int2(0, 0);
    const unsigned long long code = pcr_morton3((uint32_t)x, (uint32_t)y, (uint32_t)z);
    unsigned h = pcr_grid_hash(code, mask);
    for (unsigned probe = 0; probe <= mask; probe++) {                             // until the cell or an empty slot, as k_grid_build inserts (load factor <= 0.5: 1-2 probes)
        const int4 e = *(const int4 *)&g.tab[h];                                   // one 16-B load
        const unsigned long long c = ((unsigned long long)(unsigned)e.y << 32) | (unsigned)e.x;
        if (c == code) return make_int2(e.z, e.w);
        if (c == PCR_GRID_EMPTY) break;
        h = (h + 1) & mask;
    }
    return make_int2(0, 0);
}

// ONE QUERY PER OCTET over cells of edge c >= 2.04 r (pcr_grid_level_for): the ball around q then meets, per axis, only q's own
// cell and the neighbour on the NEARER side, i.e. the 2 x 2 x 2 block of cells around q: lane b of the octet looks up the cell whose
// offset bits are b (bit set = the neighbour along that axis; lane 0 = q's own cell) -- one lookup per lane, one round trip for the
// block -- and then the octet scans the cells TOGETHER, own cell first, face / edge / corner neighbours after it, 8 PCR_GRID8_PPS consecutive
// points per step (that many coalesced loads per lane in flight); a cell is skipped when the ball that must still be searched does not
// reach its cube.  Against the first form of this search (27 cells of edge >= r: 4 lookups and up to 8 scattered point loads per lane
// and step, most lanes without work): a third of the load instructions and of the VALU work per query.
//
// Round 5: the search returns the K NEAREST points (ordered by distance; the first with ties -> lower index, exactly the answer of the
// tree walk) and the squared distance of the (K+1)-th (r2cap when fewer lie inside the cap).  That is the query's LIST CERTIFICATE
// (pcr_gicp.hip, icp_list_eval): while the query has moved by delta and its nearest LISTED point is closer than D_{K+1} - delta, that
// point is its exact nearest neighbour -- decided from K gathered points, without a search.  The (d2 - d1) / 2 margin of the rounds
// before lost ~8 % of the queries per launch to a search again (the same chronic near-bisector queries every launch); (d5 - d1) / 2 is three
// times wider and has no mass near zero.  Every lane keeps its own K+1 smallest, sorted (an insertion is a compare-select chain run only
// for points under the lane's (K+1)-th); the octet merges by K pops of the smallest head.
#ifndef PCR_GRID8_PPS
#define PCR_GRID8_PPS 8      // points per lane and step of a cell scan (loads in flight)
#endif
#define PCR_NN_K 4           // listed neighbours of a certificate (int4 record)
template <int K> __device__ static inline int pcr_octet_bcast(int v, int ol) {       // value of octet lane K in all 8 lanes
    const int q = pcr_dpp_i<(K & 3) * 0x55>(v);                                     // quad_perm [K&3 x 4]
    const int h = pcr_dpp_i<PCR_DPP_HMIRROR>(q);
    return ((ol >> 2) == (K >> 2)) ? q : h;
}
__device__ static inline int pcr_octet_min_i(int v) {
    v = min(v, pcr_dpp_i<PCR_DPP_XOR1>(v)); v = min(v, pcr_dpp_i<PCR_DPP_XOR2>(v)); v = min(v, pcr_dpp_i<PCR_DPP_HMIRROR>(v));
    return v;
}
// sorted insertion of (du, idx) into the lane's K+1 smallest: strict order by distance, the head also by index among equal distances
template <int K> __device__ static inline void pcr_nn_insert(float (&d)[K + 1], int (&id)[K], float du, int idx) {
    float cd = du; int ci = idx;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const bool lt = cd < d[k] || (k == 0 && cd == d[k] && ci < id[k]);
        const float td = lt ? d[k] : cd; const int ti = lt ? id[k] : ci;
        d[k] = lt ? cd : d[k]; id[k] = lt ? ci : id[k];
        cd = td; ci = ti;
    }
    d[K] = fminf(d[K], cd);
}
template <int K>
__device__ static inline void grid_nn_query8(const GridView &g, const unsigned mask, const float4 *__restrict__ pts, bool live, float qx, float qy, float qz, float r2cap, int ol,
                                             int (&out_id)[K], float *d1_out, float *dnext_out) {
    float d[K + 1]; int id[K];
#pragma unroll
    for (int k = 0; k < K; k++) { d[k] = r2cap; id[k] = -1; }
    d[K] = r2cap;
    int rng = 0;                                    // the lane's cell: first point | count << 22 (0: absent, empty or out of reach)
    float gx2 = 0.0f, gy2 = 0.0f, gz2 = 0.0f;       // squared gap between q and the neighbour cell, per axis (the same in all lanes of the octet)
    if (live) {
        const float fx = (qx - g.org[0]) * g.inv_unit[0], fy = (qy - g.org[1]) * g.inv_unit[1], fz = (qz - g.org[2]) * g.inv_unit[2];
        const int cx = (int)floorf(fx) >> g.L, cy = (int)floorf(fy) >> g.L, cz = (int)floorf(fz) >> g.L;
        const float lox = qx - (g.org[0] + (float)cx * g.cell[0]), loy = qy - (g.org[1] + (float)cy * g.cell[1]), loz = qz - (g.org[2] + (float)cz * g.cell[2]);
        const float slack = 2e-3f * g.cell[0];
        const bool upx = lox + lox > g.cell[0], upy = loy + loy > g.cell[1], upz = loz + loz > g.cell[2];
        const float ax = fmaxf((upx ? g.cell[0] - lox : lox) - slack, 0.0f), ay = fmaxf((upy ? g.cell[1] - loy : loy) - slack, 0.0f),
                    az = fmaxf((upz ? g.cell[2] - loz : loz) - slack, 0.0f);
        gx2 = ax * ax; gy2 = ay * ay; gz2 = az * az;
        const int x = cx + ((ol & 1) ? (upx ? 1 : -1) : 0), y = cy + ((ol & 2) ? (upy ? 1 : -1) : 0), z = cz + ((ol & 4) ? (upz ? 1 : -1) : 0);
        const float g2 = ((ol & 1) ? gx2 : 0.0f) + ((ol & 2) ? gy2 : 0.0f) + ((ol & 4) ? gz2 : 0.0f);
        const int lim = (1 << 21) >> g.L;
        if (g2 < r2cap && x >= 0 && y >= 0 && z >= 0 && x < lim && y < lim && z < lim) {
            const int2 r = pcr_grid_lookup(g, mask, x, y, z);
            rng = r.y > 0 ? (r.x | (r.y << 22)) : 0;
        }
    }
    float bound = r2cap;                            // upper bound of the octet's (K+1)-th smallest squared distance: what a cell must be nearer than
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float g2k = ((k & 1) ? gx2 : 0.0f) + ((k & 2) ? gy2 : 0.0f) + ((k & 4) ? gz2 : 0.0f);
        int r;
        switch (k) {        // (template argument)
            case 0: r = pcr_octet_bcast<0>(rng, ol); break; case 1: r = pcr_octet_bcast<1>(rng, ol); break;
            case 2: r = pcr_octet_bcast<2>(rng, ol); break; case 3: r = pcr_octet_bcast<3>(rng, ol); break;
            case 4: r = pcr_octet_bcast<4>(rng, ol); break; case 5: r = pcr_octet_bcast<5>(rng, ol); break;
            case 6: r = pcr_octet_bcast<6>(rng, ol); break; default: r = pcr_octet_bcast<7>(rng, ol); break;
        }
        if (!(g2k < bound)) r = 0;
        const int cnt = (int)((unsigned)r >> 22), first = r & 0x3fffff;
        if (__ballot(cnt > 0) == 0ull) continue;
        for (int j0 = 0; __ballot(j0 < cnt) != 0ull; j0 += PCR_GRID8_PPS * OCT) {
            float4 pp[PCR_GRID8_PPS];
#pragma unroll
            for (int u = 0; u < PCR_GRID8_PPS; u++) { const int j = j0 + u * OCT + ol; pp[u] = pts[first + (j < cnt ? j : 0)]; }
#pragma unroll
            for (int u = 0; u < PCR_GRID8_PPS; u++) {
                const int j = j0 + u * OCT + ol;
                const float du = pcr_d2(pp[u].x - qx, pp[u].y - qy, pp[u].z - qz);
                if (j < cnt && (du < d[K] || (du == d[0] && first + j < id[0]))) pcr_nn_insert<K>(d, id, du, first + j);
            }
        }
        // octet-wide bound: K rounds of "the lanes holding the smallest head drop it" on a copy of the distances, then the smallest head left
        // is the (K+1)-th smallest or later (lanes tying for a head drop together): an upper bound, which is all the pruning needs
        {
            float h[K + 1];
#pragma unroll
            for (int t = 0; t <= K; t++) h[t] = d[t];
#pragma unroll
            for (int rd = 0; rd < K; rd++) {
                const float m = pcr_octet_min(h[0]);
                const bool pop = h[0] == m && m < r2cap;
#pragma unroll
                for (int t = 0; t < K; t++) h[t] = pop ? h[t + 1] : h[t];
                h[K] = pop ? r2cap : h[K];
            }
            bound = pcr_octet_min(h[0]);
        }
    }
    // merge: K pops of the octet's smallest head (ties -> lower index), then the smallest head left is the (K+1)-th squared distance
    float d1 = r2cap;
#pragma unroll
    for (int rd = 0; rd < K; rd++) {
        const float m = pcr_octet_min(d[0]);
        const int w = pcr_octet_min_i((d[0] == m && id[0] >= 0) ? id[0] : 0x7fffffff);
        const bool pop = d[0] == m && id[0] == w && w != 0x7fffffff;
        out_id[rd] = w != 0x7fffffff ? w : -1;
        if (rd == 0) d1 = m;
#pragma unroll
        for (int t = 0; t + 1 < K; t++) { d[t] = pop ? d[t + 1] : d[t]; id[t] = pop ? id[t + 1] : id[t]; }
        d[K - 1] = pop ? d[K] : d[K - 1]; id[K - 1] = pop ? -1 : id[K - 1];
        d[K] = pop ? r2cap : d[K];
    }
    *d1_out = d1;
    *dnext_out = pcr_octet_min(d[0]);
}

// The same search with ONE QUERY PER WAVEFRONT (wave-uniform query): the eight 8-lane groups of the wavefront take the eight cells of the
// block at once -- group b looks cell b up and scans it, 64 points per step -- so a search is one lookup and (cells of up to 64 points) one
// scan step, where the octet form walks the cells one after the other with the bound between them.  For workgroups with a handful of
// pending queries (the usual case since the list certificates: 0.3 % of the queries are searched again per launch): a wavefront per query
// has the answer after two round trips, an octet per query after four or five.  Same list (the head exactly: ties -> lower index) and
// the same (K+1)-th distance as grid_nn_query8.
__device__ static inline int pcr_wave_min_i(int v) {          // over all 64 lanes, result in every lane (DPP inside the rows, permlane swaps across)
    v = min(v, pcr_dpp_i<PCR_DPP_XOR1>(v)); v = min(v, pcr_dpp_i<PCR_DPP_XOR2>(v)); v = min(v, pcr_dpp_i<PCR_DPP_HMIRROR>(v)); v = min(v, pcr_dpp_i<PCR_DPP_MIRROR>(v));
    unsigned o; unsigned a = pcr_swap16((unsigned)v, &o);
    v = min((int)a, (int)o);
    a = pcr_swap32((unsigned)v, &o);
    return min((int)a, (int)o);
}
template <int K>
__device__ static inline void grid_nn_query64(const GridView &g, const unsigned mask, const float4 *__restrict__ pts, float qx, float qy, float qz, float r2cap, int lane,
                                              int (&out_id)[K], float *d1_out, float *dnext_out) {
    const int b = lane >> 3, ol = lane & 7;
    float d[K + 1]; int id[K];
#pragma unroll
    for (int k = 0; k < K; k++) { d[k] = r2cap; id[k] = -1; }
    d[K] = r2cap;
    int first = 0, cnt = 0;
    {
        const float fx = (qx - g.org[0]) * g.inv_unit[0], fy = (qy - g.org[1]) * g.inv_unit[1], fz = (qz - g.org[2]) * g.inv_unit[2];
        const int cx = (int)floorf(fx) >> g.L, cy = (int)floorf(fy) >> g.L, cz = (int)floorf(fz) >> g.L;
        const float lox = qx - (g.org[0] + (float)cx * g.cell[0]), loy = qy - (g.org[1] + (float)cy * g.cell[1]), loz = qz - (g.org[2] + (float)cz * g.cell[2]);
        const float slack = 2e-3f * g.cell[0];
        const bool upx = lox + lox > g.cell[0], upy = loy + loy > g.cell[1], upz = loz + loz > g.cell[2];
        const float ax = fmaxf((upx ? g.cell[0] - lox : lox) - slack, 0.0f), ay = fmaxf((upy ? g.cell[1] - loy : loy) - slack, 0.0f),
                    az = fmaxf((upz ? g.cell[2] - loz : loz) - slack, 0.0f);
        const int x = cx + ((b & 1) ? (upx ? 1 : -1) : 0), y = cy + ((b & 2) ? (upy ? 1 : -1) : 0), z = cz + ((b & 4) ? (upz ? 1 : -1) : 0);
        const float g2 = ((b & 1) ? ax * ax : 0.0f) + ((b & 2) ? ay * ay : 0.0f) + ((b & 4) ? az * az : 0.0f);
        const int lim = (1 << 21) >> g.L;
        if (g2 < r2cap && x >= 0 && y >= 0 && z >= 0 && x < lim && y < lim && z < lim) {
            const int2 r = pcr_grid_lookup(g, mask, x, y, z);
            first = r.x; cnt = r.y > 0 ? r.y : 0;
        }
    }
    for (int j0 = 0; __ballot(j0 < cnt) != 0ull; j0 += PCR_GRID8_PPS * OCT) {
        float4 pp[PCR_GRID8_PPS];
#pragma unroll
        for (int u = 0; u < PCR_GRID8_PPS; u++) { const int j = j0 + u * OCT + ol; pp[u] = pts[first + (j < cnt ? j : 0)]; }
#pragma unroll
        for (int u = 0; u < PCR_GRID8_PPS; u++) {
            const int j = j0 + u * OCT + ol;
            const float du = pcr_d2(pp[u].x - qx, pp[u].y - qy, pp[u].z - qz);
            if (j < cnt && (du < d[K] || (du == d[0] && first + j < id[0]))) pcr_nn_insert<K>(d, id, du, first + j);
        }
    }
    // merge over the 64 lanes: K pops of the smallest head (ties -> lower index); distances are >= 0, so they order like their bit patterns
    float d1 = r2cap;
#pragma unroll
    for (int rd = 0; rd < K; rd++) {
        const float m = __int_as_float(pcr_wave_min_i(__float_as_int(d[0])));
        const int w = pcr_wave_min_i((d[0] == m && id[0] >= 0) ? id[0] : 0x7fffffff);
        const bool pop = d[0] == m && id[0] == w && w != 0x7fffffff;
        out_id[rd] = w != 0x7fffffff ? w : -1;
        if (rd == 0) d1 = m;
#pragma unroll
        for (int t = 0; t + 1 < K; t++) { d[t] = pop ? d[t + 1] : d[t]; id[t] = pop ? id[t + 1] : id[t]; }
        d[K - 1] = pop ? d[K] : d[K - 1]; id[K - 1] = pop ? -1 : id[K - 1];
        d[K] = pop ? r2cap : d[K];
    }
    *d1_out = d1;
    *dnext_out = __int_as_float(pcr_wave_min_i(__float_as_int(d[0])));
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
        // arg-min over the octet by DPP (xor 1, xor 2, then the half mirror: after two steps a quad is uniform, so the mirror acts as xor 4)
        { const float od = pcr_dpp_f<PCR_DPP_XOR1>(d); const int oc = pcr_dpp_i<PCR_DPP_XOR1>(c); if (od < d || (od == d && oc < c)) { d = od; c = oc; } }
        { const float od = pcr_dpp_f<PCR_DPP_XOR2>(d); const int oc = pcr_dpp_i<PCR_DPP_XOR2>(c); if (od < d || (od == d && oc < c)) { d = od; c = oc; } }
        { const float od = pcr_dpp_f<PCR_DPP_HMIRROR>(d); const int oc = pcr_dpp_i<PCR_DPP_HMIRROR>(c); if (od < d || (od == d && oc < c)) { d = od; c = oc; } }
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

__device__ static inline float wave_or_octets_max(float v) { return pcr_xoct_max(v); }     // max over the 8 octets (v is octet-uniform), inside the VALU

// worst() -> this lane's (octet-uniform) squared bound; visit(first, count) scans points for all 8 queries;
// skip(first, count) -> leaf already covered.  q*: this lane's query; live: octet-uniform.
template <class WorstFn, class VisitFn, class SkipFn>
__device__ static inline void oct_search_group(const OctView &t, const OctMeta &m, OctGroupStack &stk, bool live, int start_leaf,
                                               float qx, float qy, float qz, WorstFn worst, VisitFn visit, SkipFn skip, int ol, int *visits) {
    if (m.nl < 1 || __ballot(live) == 0ull) return;
    // group box (live queries only)
    float glo[3] = {live ? qx : 3.4e38f, live ? qy : 3.4e38f, live ? qz : 3.4e38f};
    float ghi[3] = {live ? qx : -3.4e38f, live ? qy : -3.4e38f, live ? qz : -3.4e38f};
    // (the queries of a wavefront's 8 octets: one query per octet, so the group box is the min / max across the octets)
#pragma unroll
    for (int d = 0; d < 3; d++) { glo[d] = pcr_xoct_min(glo[d]); ghi[d] = pcr_xoct_max(ghi[d]); }
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

