// pcr_knn_wave.h -- exact k-NN with ONE QUERY PER LANE (round 3).  Included by pcr_cloud.hip after KnnArgs / d_fast_eigen3x3.
//
// Replaces the octet kernel (8 lanes per query, one surviving candidate per octet and round, 36 VALU instructions per insertion that
// all 8 lanes execute: 680 wave-instructions per query at k = 30) where every point of a cloud is a query: Open3D's
// RemoveStatisticalOutliers / EstimateNormals(KNN) (ALL_FUNCTIONS.py:297-302).
//
//   * a wavefront serves 64 Morton-consecutive queries, lane l = query g0 + l.  Their neighbourhoods overlap: the union of the true
//     30-NN sets of such a group is ~210 points (68 for a group of 8), the leaves holding them ~350-450 points;
//   * candidates are staged 64 at a time (one coalesced gather, xyz + index) in LDS and read back with a wave-uniform address -- a
//     broadcast -- so one candidate costs 7 VALU instructions for 64 (query, candidate) pairs;
//   * PASS 1 keeps only DISTANCES: every lane holds its k smallest squared distances sorted in registers; an insertion is the v_med3
//     chain sorted{x, s1..s_{K-1}} (K instructions, no indices, no cross-lane traffic) and runs only when the candidate beats the bound of
//     at least one lane (wave ballot).  At the end sd[0] is the exact k-th distance of the lane's query;
//   * PASS 2 goes over the candidates again with the FINAL bounds and appends the index of every candidate with d2 < bound (plus as
//     many ties of the bound as the sorted list holds) to the query's row in global memory (`rows`: the k-best list the caller asked
//     for, or scratch): no ordering work at all, and no list in LDS (7.8 KB per wavefront at k = 30 halved the occupancy);
//   * the epilogue works lane-parallel on the rows (float64 mean distance / raw moments + analytic eigen solver with all 64 lanes
//     live); a row is 128 B of int32 (the octet kernel wrote 256-B rows of index + distance).
//
// The walk is the bottom-up group walk of pcr_octree.h with wave-wide state, and its unit is the FAT LEAF (level-1 node, a Morton cell
// of ~50-130 points with its point range in OctView::l1rng): a list of siblings is tested conservatively (child box against the box
// of each 8-lane subgroup of queries, bound = the subgroup's largest k-th distance: 64 (child, subgroup) tests in one step), inner
// nodes and fat leaves that pass are tested against every lane's own ball, and a fat leaf that passes is staged whole.  (Walking down
// to the leaves of 6-16 points saved a third of the candidates but cost ~45 dependent node loads per pass: the kernel sat in load
// latency.)  Pass 2 does not walk at all when the candidates that beat some bound in pass 1 fit a log in LDS (they are a superset of
// every lane's k-best): it replays the log.  Results are exact (ties aside), as before: tests/test_gpu_stages.py::
// test_knn_index_is_exact, test_knn_wave_equals_octet, test_sor_mask_is_exact.
#pragma once
#include <type_traits>

#define KW_BS 128                       // 2 wavefronts per workgroup (LDS: ~6 KB per wavefront at K <= 32, ~9 KB above)

// candidate batch in LDS, structure of arrays: a step of 4 candidates is three wave-uniform 16-B reads (and their register pairs could
// feed packed float32 math, see PK below)
struct KwStage { float x[68], y[68], z[68]; int i[68]; };          // 64 + one step of read-ahead
typedef float kw_f2 __attribute__((ext_vector_type(2)));
// squared distances of two staged candidates to the lane's query, same operation order as pcr_d2 (bit-identical values)
__device__ static inline kw_f2 kw_d2x2(kw_f2 px, kw_f2 py, kw_f2 pz, kw_f2 qx, kw_f2 qy, kw_f2 qz) {
    const kw_f2 dx = px - qx, dy = py - qy, dz = pz - qz;
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

struct KwStack {                        // per wavefront (LDS)
    int cs[OCT_MAXL];
    int mask[OCT_MAXL];
    int first[OCT_MAXL][OCT];
    int count[OCT_MAXL][OCT];
    unsigned cd2[OCT_MAXL][OCT];        // per child: smallest conservative box distance over the subgroups (float bits): pop order
    float4 leafbox[OCT][2];             // boxes of the fat-leaf (level-1) list tested last (such a list is consumed before the next one is tested)
};
// BLOCK mode (below): the fat leaves around the wavefront's own one or two, faces first, then edges, then corners
#define KW_NBR 64                       // listed neighbour leaves (more: the wavefront is handed over; 4 adjacent cells rarely have 40 occupied neighbours)
struct KwBlock {
    int sc[6];                          // low corners (level-1 cell coordinates) of the one or two level-2 cells holding the seed leaves
    int first[KW_NBR], count[KW_NBR];
    float4 box[KW_NBR][2];
};
template <int K, bool BLOCK = false>
struct KwShared {
    static constexpr int LOG = K <= 32 ? 768 : 1536;   // pass-1 log capacity (events per wavefront at k = 30: mean 260, p99 580)
    KwStage stage;                      // candidate batch
    int log[LOG];                       // indices of the candidates that beat the bound of some lane in pass 1
    typename std::conditional<BLOCK, KwBlock, KwStack>::type stk;
};

__device__ static inline float kw_octet_fmin(float v) {
    v = fminf(v, pcr_dpp_f<PCR_DPP_XOR1>(v)); v = fminf(v, pcr_dpp_f<PCR_DPP_XOR2>(v)); v = fminf(v, pcr_dpp_f<PCR_DPP_HMIRROR>(v));
    return v;
}
__device__ static inline float kw_octet_fmax(float v) {
    v = fmaxf(v, pcr_dpp_f<PCR_DPP_XOR1>(v)); v = fmaxf(v, pcr_dpp_f<PCR_DPP_XOR2>(v)); v = fmaxf(v, pcr_dpp_f<PCR_DPP_HMIRROR>(v));
    return v;
}
__device__ static inline float kw_wave_fmin(float v) {
    v = kw_octet_fmin(v);
#pragma unroll
    for (int o = OCT; o < 64; o <<= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ static inline float kw_wave_fmax(float v) {
    v = kw_octet_fmax(v);
#pragma unroll
    for (int o = OCT; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// squared distance between two boxes, same operation order as pcr_box_d2 (monotone: never above the box distance of a point inside g)
__device__ static inline float kw_boxbox_d2(const float4 lo, const float4 hi, const float *glo, const float *ghi) {
    const float dx = fmaxf(fmaxf(lo.x - ghi[0], glo[0] - hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(lo.y - ghi[1], glo[1] - hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(lo.z - ghi[2], glo[2] - hi.z), 0.0f);
    return pcr_d2(dx, dy, dz);
}

// One pass over the candidates of the wavefront's 64 queries: the seed range [plo, phi] first (the fat leaves of the queries
// themselves), then the fat leaves a bottom-up walk from the level-1 node `start_node` finds (pcr_octree.h: climb, test the siblings,
// open the nearest first, stop once every ball lies inside the ancestor's cell).  The pass is a LOOP AROUND ONE scan() CALL: the walk advances until ~a batch of candidates is staged
// (lane l of the batch <- point index bidx, nb of them), scan() consumes it, the walk resumes -- so the scan body (the v_med3 chains)
// is instantiated once per pass, not once per place the walk can emit points from.  All control flow is wave-uniform.
//   worst()  -> this lane's squared bound (-1: lane without a query);   INCL: bounds are inclusive (pass 2 must also reach the points
//   AT the k-th distance).  scan(nb) consumes the nb candidates staged in LDS (stage[j] = xyz + index bits, slots beyond nb at infinity).
// The gather of a batch is issued one batch AHEAD whenever the pending point range continues (seed ranges and fat leaves are longer than
// a batch more often than not): the scan of the current batch then hides the load latency of the next.
struct KwStats { int tests, exact_inner, pops, leaf_hits, climbs, batches, cands, events; unsigned long long t_walk, t_stage, t_scan; };
// Returns false when the pass was given up after `budget` batches (a wavefront whose queries lie far apart: the caller hands them over).
template <bool INCL, bool STATS, class WorstFn, class ScanFn>
__device__ static inline bool kw_pass(const OctView &t, const OctMeta &m, KwStack &stk, bool live, int start_node, int start_point, float qx, float qy, float qz,
                                      int plo, int phi, const uint8_t *keep, KwStage &stage, int budget, WorstFn worst, ScanFn scan, KwStats &st) {
    const int lane = threadIdx.x & 63, c8 = lane & 7;
    // boxes of the wavefront's queries and of each 8-lane subgroup
    float sglo[3] = {live ? qx : 3.4e38f, live ? qy : 3.4e38f, live ? qz : 3.4e38f};
    float sghi[3] = {live ? qx : -3.4e38f, live ? qy : -3.4e38f, live ? qz : -3.4e38f};
    float glo[3], ghi[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { sglo[d] = kw_octet_fmin(sglo[d]); sghi[d] = kw_octet_fmax(sghi[d]); }
#pragma unroll
    for (int d = 0; d < 3; d++) {
        glo[d] = sglo[d]; ghi[d] = sghi[d];
#pragma unroll
        for (int o = OCT; o < 64; o <<= 1) { glo[d] = fminf(glo[d], __shfl_xor(glo[d], o, 64)); ghi[d] = fmaxf(ghi[d], __shfl_xor(ghi[d], o, 64)); }
    }
    auto beats = [&](float d2, float bound) -> bool { return INCL ? d2 <= bound : d2 < bound; };
    // children [cs, cs + cnt) of level li: bit c set when the box of child c comes within the bound of some subgroup (conservative: the
    // child's box against the subgroup's box and largest bound -- 64 (child, subgroup) tests in one step)
    auto test = [&](int li, int cs, int cnt) -> int {
        bool pass = false; int f = 0, c = 0; float d2 = __builtin_inff();
        float4 lo = make_float4(0, 0, 0, 0), hi = make_float4(0, 0, 0, 0);
        if (lane < OCT) stk.cd2[li][lane] = 0x7f800000u;
        if (STATS) st.tests++;
        const float wsub = pcr_octet_max(worst());
        if (c8 < cnt) {
            const size_t j = (size_t)(m.off[li] + cs + c8);
            lo = t.nodes[2 * j]; hi = t.nodes[2 * j + 1];
            d2 = kw_boxbox_d2(lo, hi, sglo, sghi);
            pass = beats(d2, wsub);
            f = __float_as_int(lo.w); c = __float_as_int(hi.w);
            if (li == 1) { const int2 r = t.l1rng[cs + c8]; f = r.x; c = r.y; }      // fat leaf: its POINT range
        }
        if (pass) atomicMin(&stk.cd2[li][c8], __float_as_uint(d2));
        unsigned long long bal = __ballot(pass);
        bal |= bal >> 32; bal |= bal >> 16; bal |= bal >> 8;
        if (lane < OCT) {
            stk.first[li][lane] = f; stk.count[li][lane] = c;
            if (li == 1) { stk.leafbox[lane][0] = lo; stk.leafbox[lane][1] = hi; }
        }
        int cmask = (int)(bal & 0xffull);
        if (li > 1) {
            // inner nodes that pass the subgroup test are re-tested against every lane's OWN ball (the box of child c sits in lane c):
            // a subgroup that straddles a jump of the Morton curve has a box as large as the gap, and every node inside the gap
            // would pass for it (measured: launches of 6-50 ms instead of 0.2).  Fat leaves get this test when they are popped.
            int keep = 0;
            for (int mm = cmask; mm != 0; mm &= mm - 1) {
                const int cc = __builtin_ctz((unsigned)mm);
                if (STATS) st.exact_inner++;
                const float4 blo = make_float4(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo.x), cc)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo.y), cc)),
                                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo.z), cc)), 0.0f);
                const float4 bhi = make_float4(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(hi.x), cc)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hi.y), cc)),
                                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hi.z), cc)), 0.0f);
                if (__ballot(beats(pcr_box_d2(blo, bhi, qx, qy, qz), worst())) != 0ull) keep |= 1 << cc;
            }
            cmask = keep;
        }
        return cmask;
    };
    // nearest pending child of level li; -1 when no subgroup can still improve in any of them
    auto pop = [&](int li, int mask) -> int {
        const unsigned g = ((mask >> c8) & 1) ? stk.cd2[li][c8] : 0xffffffffu;
        unsigned k = g;
        k = min(k, (unsigned)pcr_dpp_i<PCR_DPP_XOR1>((int)k));
        k = min(k, (unsigned)pcr_dpp_i<PCR_DPP_XOR2>((int)k));
        k = min(k, (unsigned)pcr_dpp_i<PCR_DPP_HMIRROR>((int)k));
        if (__ballot(beats(__uint_as_float(k), worst())) == 0ull) return -1;
        return __builtin_ctz((unsigned)(__ballot(g == k) & 0xffull));
    };
    auto contained = [&](int lvl, uint32_t ix, uint32_t iy, uint32_t iz) -> bool {
        const float wmax = kw_wave_fmax(worst());
        if (!(wmax < 3.0e38f)) return false;
        const float r = sqrtf(fmaxf(wmax, 0.0f)) * 1.00001f;
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
        return __builtin_amdgcn_readfirstlane((int)in) != 0;
    };

    // pending point ranges (at most two: a leaf minus the seed range), drained into the batch before the walk advances
    int r0f = plo, r0c = phi - plo + 1, r1f = 0, r1c = 0;
    auto add_clipped = [&](int f, int c) {                     // [f, f + c) minus [plo, phi]; r0c == r1c == 0 here
        const int e = f + c;
        if (f < plo) { r0f = f; r0c = (e < plo ? e : plo) - f; }
        if (e > phi + 1) {
            const int s = f > phi + 1 ? f : phi + 1;
            if (r0c > 0) { r1f = s; r1c = e - s; } else { r0f = s; r0c = e - s; }
        }
    };
    int phase = 0;                                              // 0: walk not yet started, 1: walking, 2: finished
    int anc = 0, anc_li = 1, li = 1, cs = 0, base_li = 1, mask = 0;
    uint32_t ix = 0, iy = 0, iz = 0;
    start_node = __builtin_amdgcn_readfirstlane(start_node);
    int nb = 0, bidx = 0;                                       // the batch being assembled: lane l <- point bidx, nb lanes filled
    int pre_n = 0; float4 pre = make_float4(0, 0, 0, 0);        // its first pre_n lanes were loaded a batch ago
    int4 u_pre = make_int4(0, 0, 1, 0);                         // up-link of `anc`, loaded when anc is set (one round trip less per climb)
    for (;;) {
        bool done = false;
        const unsigned long long tw0 = STATS ? __builtin_readcyclecounter() : 0ull;
        for (;;) {
            while (r0c > 0 && nb < 64) {
                const int take = r0c < 64 - nb ? r0c : 64 - nb;
                if (lane >= nb && lane < nb + take) bidx = r0f + (lane - nb);
                nb += take; r0f += take; r0c -= take;
                if (r0c == 0) { r0f = r1f; r0c = r1c; r1c = 0; }
            }
            if (r0c > 0 || nb >= 48) break;                     // a batch is ready
            if (phase == 2) { done = true; break; }
            if (phase == 0) {
                if (m.nl < 2) { phase = 2; continue; }          // a tree of one level: the seed range is the whole cloud
                const uint64_t key = t.keys[start_point];
                ix = __builtin_amdgcn_readfirstlane(pcr_compact21(key)); iy = __builtin_amdgcn_readfirstlane(pcr_compact21(key >> 1)); iz = __builtin_amdgcn_readfirstlane(pcr_compact21(key >> 2));
                anc = start_node; phase = 1;
                u_pre = t.up[m.off[1] + anc];
                continue;
            }
            if (mask == 0) {
                if (li < base_li) { li++; cs = __builtin_amdgcn_readfirstlane(stk.cs[li]); mask = __builtin_amdgcn_readfirstlane(stk.mask[li]); continue; }
                if (nb > 0) break;                              // the bounds are consulted for a climb: consume what is staged first
                if (anc_li >= m.nl - 1 || contained(m.l0 + anc_li, ix, iy, iz)) { phase = 2; done = true; break; }
                const int4 u = u_pre;
                const int ux = __builtin_amdgcn_readfirstlane(u.x), uy = __builtin_amdgcn_readfirstlane(u.y), uz = __builtin_amdgcn_readfirstlane(u.z);
                const int nm = test(anc_li, uy, uz);
                if (STATS) st.climbs++;
                li = anc_li; base_li = anc_li; cs = uy;
                mask = nm & ~(1 << (anc - uy));
                anc = ux; anc_li++;
                if (anc_li < m.nl - 1) u_pre = t.up[m.off[anc_li] + anc];
                continue;
            }
            const int c = pop(li, mask);
            if (c < 0) { mask = 0; continue; }
            mask &= ~(1 << c);
            if (STATS) st.pops++;
            const int nf = __builtin_amdgcn_readfirstlane(stk.first[li][c]), nc = __builtin_amdgcn_readfirstlane(stk.count[li][c]);
            if (li == 1) {
                // a fat leaf is staged only when its box meets the ball of some lane (exact test; the boxes of the current list are in LDS)
                const float4 lo = stk.leafbox[c][0], hi = stk.leafbox[c][1];
                const float bd2 = pcr_box_d2(lo, hi, qx, qy, qz);
                if (__ballot(beats(bd2, worst())) != 0ull) { add_clipped(nf, nc); if (STATS) st.leaf_hits++; }
            } else {
                if (lane == 0) { stk.cs[li] = cs; stk.mask[li] = mask; }
                li--; cs = nf;
                mask = test(li, nf, nc);
            }
        }
        const unsigned long long tw1 = STATS ? __builtin_readcyclecounter() : 0ull;
        if (STATS) st.t_walk += tw1 - tw0;
        if (nb > 0) {
            if (STATS) { st.batches++; st.cands += nb; }
            if (--budget < 0) return false;
            const float inf = __builtin_inff();
            float4 p = make_float4(inf, inf, inf, 0.0f);        // empty slots and points outside `keep` are staged at infinity
            if (lane < nb && (!keep || keep[bidx])) { p = lane < pre_n ? pre : t.pts[bidx]; p.w = __int_as_float(bidx); }
            // the pending range goes on: issue the gather of the next batch's first lanes now, it lands while this batch is scanned
            int nx = 0, nidx = 0;
            if (r0c > 0) {
                nx = r0c < 64 ? r0c : 64;
                if (lane < nx) { nidx = r0f + lane; pre = t.pts[nidx]; }
                r0f += nx; r0c -= nx;
                if (r0c == 0) { r0f = r1f; r0c = r1c; r1c = 0; }
            }
            __builtin_amdgcn_wave_barrier();
            stage.x[lane] = p.x; stage.y[lane] = p.y; stage.z[lane] = p.z; stage.i[lane] = __float_as_int(p.w);
            __builtin_amdgcn_wave_barrier();
            unsigned long long ts0 = 0;
            if (STATS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts0 = __builtin_readcyclecounter(); st.t_stage += ts0 - tw1; }
            scan(nb);
            if (STATS) st.t_scan += __builtin_readcyclecounter() - ts0;
            nb = nx; pre_n = nx; bidx = nidx;
        }
        if (done && nb == 0) return true;
    }
}

// BLOCK mode of pass 1: no tree walk.  The level-1 cells (fat leaves, ~25 points on a surface) are a uniform grid of edge E = unit 2^(l0+1),
// and the ball of a query's k-th distance almost never leaves the 3 x 3 x 3 block of cells around the query's own (measured: 0.2 % of
// the queries of the bench clouds).  The wavefront's 64 Morton-consecutive queries lie in 2-5 consecutive fat leaves (the seed range),
// which are children of ONE level-2 cell or of two consecutive ones; the blocks of all of them lie inside the 4 x 4 x 4 cells made of
// that parent's 2 x 2 x 2 children and a rim of one cell -- 64 cells, one per LANE.  So, after the seed range, lane l looks cell l of the
// first parent's 4 x 4 x 4 up in a hash of the level-1 nodes (`tab`: cell code -> node; ONE round trip for all of them, one more for
// their point ranges and tight boxes; a second round for the second parent's cells outside the first's), the found leaves are listed
// in LDS, and every listed leaf whose box meets the ball of some lane is staged whole, as the walk did.  Afterwards a lane whose ball
// left its block (sparse regions), or a wavefront spanning more than two parents, flags its queries `hard` for the octet kernel
// (d_knn_wave).  The walk cost ~9 000 of the ~37 000 VALU instructions of a wavefront at k = 30 (18 sibling tests of ~190, 38 pops of
// ~100); this costs ~1 000.
template <bool STATS, class WorstFn, class ScanFn>
__device__ static inline bool kw_pass_block(const OctView &t, const OctMeta &m, KwBlock &blk, const GridEntry *__restrict__ tab, unsigned tmask, float qx, float qy, float qz,
                                            int plo, int phi, int u /* unit level of the block: 1 or 2 */, int ja, int jb /* seed nodes of level u */,
                                            int parents /* 1, 2; 0: give up (blk.sc: their low corners in level-u cells) */,
                                            const uint8_t *keep, KwStage &stage, int budget, WorstFn worst, ScanFn scan, KwStats &st) {
    const int lane = threadIdx.x & 63;
    int r0f = plo, r0c = phi - plo + 1;
    int nlist = -1, nk = 0;
    int nb = 0, bidx = 0;
    int pre_n = 0; float4 pre = make_float4(0, 0, 0, 0);
    for (;;) {
        bool done = false;
        for (;;) {
            if (r0c > 0 && nb < 64) {
                const int take = r0c < 64 - nb ? r0c : 64 - nb;
                if (lane >= nb && lane < nb + take) bidx = r0f + (lane - nb);
                nb += take; r0f += take; r0c -= take;
            }
            if (r0c > 0 || nb >= 48) break;                     // a batch is ready
            if (nlist < 0) {
                if (nb > 0) break;                              // the seeds are scanned before the first neighbour is tested (tight bounds)
                if (m.nl < 2 || parents == 0) { nlist = 0; continue; }      // one level: the seed range was the whole cloud; too long a seed range: the caller gives up
                const int sh1 = m.l0 + u;
                const int lim = sh1 < 21 ? (1 << (21 - sh1)) : 1;
                const int ox = blk.sc[0] - 1, oy = blk.sc[1] - 1, oz = blk.sc[2] - 1;
                int base = 0;
                if (u == 2) {
                    // the seed nodes of level 2 hold more than the seed range (the fat leaves of the queries themselves): the rest of the first
                    // and of the last one comes first in the list (boxes: the nodes' own, conservative)
                    const size_t sa = (size_t)(m.off[2] + ja), sb = (size_t)(m.off[2] + jb);
                    const float4 alo = t.nodes[2 * sa], ahi = t.nodes[2 * sa + 1], blo = t.nodes[2 * sb], bhi = t.nodes[2 * sb + 1];
                    const int fa = t.l1rng[__float_as_int(alo.w)].x;
                    const int2 rb = t.l1rng[__float_as_int(blo.w) + __float_as_int(bhi.w) - 1];
                    const int eb = rb.x + rb.y;
                    if (lane == 0) {
                        blk.first[0] = fa; blk.count[0] = plo - fa; blk.box[0][0] = alo; blk.box[0][1] = ahi;
                        blk.first[1] = phi + 1; blk.count[1] = eb - (phi + 1); blk.box[1][0] = blo; blk.box[1][1] = bhi;
                    }
                    base = 2;
                }
                for (int r = 0; r < parents; r++) {
                    const int cx = blk.sc[3 * r] - 1 + (lane & 3), cy = blk.sc[3 * r + 1] - 1 + ((lane >> 2) & 3), cz = blk.sc[3 * r + 2] - 1 + (lane >> 4);
                    bool want = cx >= 0 && cy >= 0 && cz >= 0 && cx < lim && cy < lim && cz < lim;
                    if (r > 0) want = want && !((unsigned)(cx - ox) < 4u && (unsigned)(cy - oy) < 4u && (unsigned)(cz - oz) < 4u);      // the first round had it
                    int node = -1;
                    if (want) {
                        const unsigned long long code = pcr_morton3((uint32_t)cx, (uint32_t)cy, (uint32_t)cz);
                        unsigned h = pcr_grid_hash(code, tmask);
                        for (int probe = 0; probe < 64; probe++) {
                            const int4 e = *(const int4 *)&tab[h];
                            const unsigned long long c = ((unsigned long long)(unsigned)e.y << 32) | (unsigned)e.x;
                            if (c == code) { node = e.z; break; }
                            if (c == PCR_GRID_EMPTY) break;
                            h = (h + 1) & tmask;
                        }
                    }
                    if (node >= ja && node <= jb) node = -1;    // a seed node: scanned already / listed above
                    int f = 0, c = 0; float4 lo = make_float4(0, 0, 0, 0), hi = lo;
                    if (node >= 0) {
                        const size_t j = (size_t)(m.off[u] + node);
                        lo = t.nodes[2 * j]; hi = t.nodes[2 * j + 1];
                        if (u == 1) { const int2 rg = t.l1rng[node]; f = rg.x; c = rg.y; }
                        else { const int c0 = __float_as_int(lo.w), cn = __float_as_int(hi.w); f = t.l1rng[c0].x; const int2 rl = t.l1rng[c0 + cn - 1]; c = rl.x + rl.y - f; }
                    }
                    const unsigned long long b = __ballot(node >= 0);
                    const int pos = base + __builtin_popcountll(b & ((1ull << lane) - 1ull));
                    if (node >= 0 && pos < KW_NBR) { blk.first[pos] = f; blk.count[pos] = c; blk.box[pos][0] = lo; blk.box[pos][1] = hi; }
                    base += __builtin_popcountll(b);
                }
                __builtin_amdgcn_wave_barrier();
                nlist = __builtin_amdgcn_readfirstlane(base);
                if (nlist > KW_NBR) return false;               // (the list kept its first KW_NBR entries only)
                continue;
            }
            if (nk >= nlist) { done = true; break; }
            const float4 lo = blk.box[nk][0], hi = blk.box[nk][1];
            const int nf = __builtin_amdgcn_readfirstlane(blk.first[nk]), nc = __builtin_amdgcn_readfirstlane(blk.count[nk]);
            nk++;
            if (STATS) st.pops++;
            if (__ballot(pcr_box_d2(lo, hi, qx, qy, qz) < worst()) != 0ull) { r0f = nf; r0c = nc; if (STATS) st.leaf_hits++; }
        }
        if (nb > 0) {
            if (STATS) { st.batches++; st.cands += nb; }
            if (--budget < 0) return false;
            const float inf = __builtin_inff();
            float4 p = make_float4(inf, inf, inf, 0.0f);        // empty slots and points outside `keep` are staged at infinity
            if (lane < nb && (!keep || keep[bidx])) { p = lane < pre_n ? pre : t.pts[bidx]; p.w = __int_as_float(bidx); }
            int nx = 0, nidx = 0;                               // the pending range goes on: its next batch is gathered while this one is scanned
            if (r0c > 0) {
                nx = r0c < 64 ? r0c : 64;
                if (lane < nx) { nidx = r0f + lane; pre = t.pts[nidx]; }
                r0f += nx; r0c -= nx;
            }
            __builtin_amdgcn_wave_barrier();
            stage.x[lane] = p.x; stage.y[lane] = p.y; stage.z[lane] = p.z; stage.i[lane] = __float_as_int(p.w);
            __builtin_amdgcn_wave_barrier();
            scan(nb);
            nb = nx; pre_n = nx; bidx = nidx;
        }
        if (done && nb == 0) return true;
    }
}

// PK: packed float32 distance math (v_pk_add / v_pk_mul / v_pk_fma, two candidates per instruction): 10 % fewer VALU instructions, but 7 % MORE
// wave-cycles and 1-3 % fewer pairs/s (measured, interleaved A/B) -- packed float32 issues at half rate here.  Off.
template <int MODE, int K, bool STATS = false, bool PK = false, bool BLOCK = false>
__device__ static inline void d_knn_wave(const KnnArgs &a) {
    constexpr int WPB = KW_BS / 64;
    __shared__ OctMeta m;
    __shared__ KwShared<K, BLOCK> shw[WPB];
    if (threadIdx.x == 0) m = *a.t.meta;
    if (blockIdx.x == 0 && threadIdx.x == 0) { if (a.zero_a) *a.zero_a = 0; if (a.zero_b) *a.zero_b = 0; }
    __syncthreads();
    const int n = __builtin_amdgcn_readfirstlane(m.n);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    KwShared<K, BLOCK> &sh = shw[wv];
    const int g0 = (blockIdx.x * WPB + wv) * 64;
    if (g0 >= n) return;
    const int qi = g0 + lane;
    const int oq = (a.keep && qi < n) ? a.pos[qi] : qi;                 // output / todo index of this query
    const bool live = qi < n && (!a.keep || a.keep[qi]) && (!a.todo || a.todo[oq]);
    const unsigned long long live_mask = __ballot(live);
    if (live_mask == 0ull) return;
    const float4 q = a.t.pts[qi < n ? qi : g0];
    const int k = a.k < K ? a.k : K;
    // the query's k-best row: MODE_SOR / MODE_NORMALS -> a.list_idx with a.list_pitch entries per row (indexed like the outputs), MODE_DEBUG -> dbg_idx
    int32_t *const row = MODE == KNN_MODE_DEBUG ? a.dbg_idx + (size_t)qi * a.k : a.list_idx + (size_t)(MODE == KNN_MODE_SOR ? qi : oq) * a.list_pitch;

    KwStats st1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long c_begin = STATS ? __builtin_readcyclecounter() : 0ull;
    // ---- the lane's k smallest squared distances, DESCENDING (sd[0] = the bound); slots beyond k and lanes without a query hold -1
    float sd[K];
#pragma unroll
    for (int j = 0; j < K; j++) sd[j] = (live && j < k) ? a.r2cap_f : -1.0f;

    // ---- candidates come staged 64 at a time in LDS (kw_pass); the scans read them back with a wave-uniform address, in steps of 4
    const int glast = g0 + 63 < n - 1 ? g0 + 63 : n - 1;
    // seed range: the fat leaves of the first and the last query and what lies between them in Morton order (one or two leaves, rarely
    // three); scanned first, skipped by the walk.  A tree of one level is scanned whole.
    int plo = 0, phi = n - 1, start_node = 0, last_node = 0;
    if (m.nl >= 2) {
        const int4 pa = a.t.pinfo[g0], pb = a.t.pinfo[glast];
        plo = __builtin_amdgcn_readfirstlane(pa.y); phi = __builtin_amdgcn_readfirstlane(pb.y + pb.z - 1); start_node = __builtin_amdgcn_readfirstlane(pa.x);
        last_node = __builtin_amdgcn_readfirstlane(pb.x);
    }
    // BLOCK: level-1 cell of this lane's query, and the level-2 parents of the first and the last seed leaf (consecutive seeds have
    // consecutive parents: when those two are the same or neighbours in the node order, every seed is a child of one of them)
    int cq[3] = {0, 0, 0};
    int parents = 0, unit_lv = 1, seed_a = start_node, seed_b = last_node;
    if (BLOCK && m.nl >= 2) {
        const uint64_t code_q = a.t.keys[qi < n ? qi : g0] >> (3 * (m.l0 + 1));
        cq[0] = (int)pcr_compact21(code_q); cq[1] = (int)pcr_compact21(code_q >> 1); cq[2] = (int)pcr_compact21(code_q >> 2);
        if constexpr (BLOCK) {
            if (m.nl >= 3) {
                const int pa = __builtin_amdgcn_readfirstlane(a.t.up[m.off[1] + start_node].x), pb = __builtin_amdgcn_readfirstlane(a.t.up[m.off[1] + last_node].x);
                parents = pb == pa ? 1 : (pb - pa == 1 ? 2 : 0);
                if (parents == 0 && a.l2tab) {
                    // a sparse stretch of the curve (the 64 queries under three level-2 cells or more): the same one level up -- level-2 cells
                    // as the unit, their level-3 parents' 4 x 4 x 4 as the block
                    unit_lv = 2; seed_a = pa; seed_b = pb;
                    if (m.nl >= 4) {
                        const int ga = __builtin_amdgcn_readfirstlane(a.t.up[m.off[2] + pa].x), gb = __builtin_amdgcn_readfirstlane(a.t.up[m.off[2] + pb].x);
                        parents = gb == ga ? 1 : (gb - ga == 1 ? 2 : 0);
                    } else parents = 1;
                }
            } else parents = 1;                                  // two levels: the root is the one parent
            const int us = unit_lv - 1;                          // level-1 cell -> level-u cell
            if (lane == 0) { sh.stk.sc[0] = (cq[0] >> us) & ~1; sh.stk.sc[1] = (cq[1] >> us) & ~1; sh.stk.sc[2] = (cq[2] >> us) & ~1; }
            if (lane == glast - g0) { sh.stk.sc[3] = (cq[0] >> us) & ~1; sh.stk.sc[4] = (cq[1] >> us) & ~1; sh.stk.sc[5] = (cq[2] >> us) & ~1; }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // pass 1: distances only; every candidate that beats the bound of some lane is logged for pass 2
    int nlog = 0;
    const kw_f2 qx2 = {q.x, q.x}, qy2 = {q.y, q.y}, qz2 = {q.z, q.z};
    auto scan1 = [&](int nb) {
        float4 nX = *(const float4 *)&sh.stage.x[0], nY = *(const float4 *)&sh.stage.y[0], nZ = *(const float4 *)&sh.stage.z[0];
        int4 nI = *(const int4 *)&sh.stage.i[0];
        for (int j4 = 0; j4 < nb; j4 += 4) {
            const float4 X = nX, Y = nY, Z = nZ;           // the LDS reads of the next step are in flight while this one is worked on
            const int i4[4] = {nI.x, nI.y, nI.z, nI.w};    // (indices included: fetching one at the moment it is needed drained the read-ahead)
            nX = *(const float4 *)&sh.stage.x[j4 + 4]; nY = *(const float4 *)&sh.stage.y[j4 + 4]; nZ = *(const float4 *)&sh.stage.z[j4 + 4];
            nI = *(const int4 *)&sh.stage.i[j4 + 4];
            float d4[4];
            if (PK) {
                const kw_f2 da = kw_d2x2(kw_f2{X.x, X.y}, kw_f2{Y.x, Y.y}, kw_f2{Z.x, Z.y}, qx2, qy2, qz2);
                const kw_f2 db = kw_d2x2(kw_f2{X.z, X.w}, kw_f2{Y.z, Y.w}, kw_f2{Z.z, Z.w}, qx2, qy2, qz2);
                d4[0] = da.x; d4[1] = da.y; d4[2] = db.x; d4[3] = db.y;
            } else {
                d4[0] = pcr_d2(X.x - q.x, Y.x - q.y, Z.x - q.z); d4[1] = pcr_d2(X.y - q.x, Y.y - q.y, Z.y - q.z);
                d4[2] = pcr_d2(X.z - q.x, Y.z - q.y, Z.z - q.z); d4[3] = pcr_d2(X.w - q.x, Y.w - q.y, Z.w - q.z);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
            const float d2 = d4[u];
            if (__ballot(d2 < sd[0]) != 0ull) {
                if (STATS) st1.events++;
                sh.log[nlog < KwShared<K, BLOCK>::LOG ? nlog : KwShared<K, BLOCK>::LOG - 1] = i4[u];      // all lanes, one address, one value
                nlog = __builtin_amdgcn_readfirstlane(nlog + 1);
                // x replaces the largest entry when it is smaller, else the chain leaves the list as it is; entries are >= 0 or -1, so
                // they order like their bit patterns (integer min / max: no NaN canonicalisation)
                const float x = __int_as_float(min(__float_as_int(d2), __float_as_int(sd[0])));
                if (K == 1) sd[0] = x;
                else {
                    float nd[K];
                    nd[0] = __int_as_float(max(__float_as_int(x), __float_as_int(sd[1])));
#pragma unroll
                    for (int s = 1; s + 1 < K; s++) nd[s] = __builtin_amdgcn_fmed3f(x, sd[s], sd[s + 1]);
                    nd[K - 1] = __int_as_float(min(__float_as_int(x), __float_as_int(sd[K - 1])));
#pragma unroll
                    for (int s = 0; s < K; s++) sd[s] = nd[s];
                }
            }
            }
        }
    };
    // A wavefront whose 64 queries lie far apart (sparse regions, a jump of the Morton curve inside the group) would drag all its lanes
    // through the neighbourhoods of each: ~1 % of the wavefronts took 3-6x the mean and set the length of the launch.  They stop after
    // a.wave_budget batches (or when the log is full) and flag their queries in a.hard for the octet kernel, which serves 8 per wavefront.
    bool finished;
    if constexpr (BLOCK) finished = kw_pass_block<STATS>(a.t, m, sh.stk, unit_lv == 2 ? a.l2tab : a.l1tab, unit_lv == 2 ? a.l2mask : a.l1mask, q.x, q.y, q.z, plo, phi, unit_lv, seed_a, seed_b, parents,
                                                         a.keep, sh.stage, a.wave_budget, [&]() { return sd[0]; }, scan1, st1);
    else finished = kw_pass<false, STATS>(a.t, m, sh.stk, live, start_node, g0, q.x, q.y, q.z, plo, phi, a.keep, sh.stage, a.hard ? a.wave_budget : 0x7fffffff, [&]() { return sd[0]; }, scan1, st1);
    bool served = live;                 // this lane's query is answered here (not handed to the octet kernel)
    if (a.hard) {
        bool give_up = !finished || nlog > KwShared<K, BLOCK>::LOG;
        if (BLOCK && m.nl >= 2) {
            // the block around the lane's own cell holds every point within the distance from the query to the block's surface: the k-best
            // of the lane are final if its bound lies inside that (a little spare for the rounding of org + i * unit)
            give_up = give_up || parents == 0;
            const float qq[3] = {q.x, q.y, q.z};
            float dmin = 3.0e38f;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const float E = m.unit[d] * (float)(1 << (m.l0 + unit_lv));
                const float lo = m.org[d] + (float)((cq[d] >> (unit_lv - 1)) - 1) * E;
                dmin = fminf(dmin, fminf(qq[d] - lo, lo + 3.0f * E - qq[d]) - 2e-3f * E);
            }
            const bool inside = sd[0] < 3.0e38f && dmin > 0.0f && sd[0] * 1.00002f < dmin * dmin;
            served = live && inside;
        }
        // (the value is the reason, for PCR_KNNW_HARDSTAT: 1 budget / list overflow, 2 log overflow, 3 seed leaves under more than two level-2 cells, 4 ball outside the block)
        const int why = !finished ? 1 : (nlog > KwShared<K, BLOCK>::LOG ? 2 : ((BLOCK && parents == 0) ? 3 : 4));
        if (give_up) served = false;
        if (live) a.hard[oq] = served ? 0 : (uint8_t)why;
        if (__ballot(served) == 0ull) return;
    }
    const unsigned long long c_mid = STATS ? __builtin_readcyclecounter() : 0ull;

    // ---- pass 2: indices under the final bound.  `ties` = how many list entries equal the bound: that many candidates AT the bound belong
    // to the k-best (a radius-capped list that is not full takes none: its bound is the cap itself)
    const float bound = sd[0];
    int ties = 0;
#pragma unroll
    for (int j = 0; j < K; j++) ties += (sd[j] == bound) ? 1 : 0;
    if (!live || bound == a.r2cap_f) ties = 0;
    const int k_out = served ? k : 0;                      // lanes handed over take nothing
    int cnt = 0, tcnt = 0;
    auto scan2 = [&](int nb) {
        float4 nX = *(const float4 *)&sh.stage.x[0], nY = *(const float4 *)&sh.stage.y[0], nZ = *(const float4 *)&sh.stage.z[0];
        int4 nI = *(const int4 *)&sh.stage.i[0];
        for (int j4 = 0; j4 < nb; j4 += 4) {
            const float4 X = nX, Y = nY, Z = nZ;           // the LDS reads of the next step are in flight while this one is worked on
            const int i4[4] = {nI.x, nI.y, nI.z, nI.w};    // (indices included: fetching one at the moment it is needed drained the read-ahead)
            nX = *(const float4 *)&sh.stage.x[j4 + 4]; nY = *(const float4 *)&sh.stage.y[j4 + 4]; nZ = *(const float4 *)&sh.stage.z[j4 + 4];
            nI = *(const int4 *)&sh.stage.i[j4 + 4];
            float d4[4];
            if (PK) {
                const kw_f2 da = kw_d2x2(kw_f2{X.x, X.y}, kw_f2{Y.x, Y.y}, kw_f2{Z.x, Z.y}, qx2, qy2, qz2);
                const kw_f2 db = kw_d2x2(kw_f2{X.z, X.w}, kw_f2{Y.z, Y.w}, kw_f2{Z.z, Z.w}, qx2, qy2, qz2);
                d4[0] = da.x; d4[1] = da.y; d4[2] = db.x; d4[3] = db.y;
            } else {
                d4[0] = pcr_d2(X.x - q.x, Y.x - q.y, Z.x - q.z); d4[1] = pcr_d2(X.y - q.x, Y.y - q.y, Z.y - q.z);
                d4[2] = pcr_d2(X.z - q.x, Y.z - q.y, Z.z - q.z); d4[3] = pcr_d2(X.w - q.x, Y.w - q.y, Z.w - q.z);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
            const float d2 = d4[u];
            const bool le = d2 <= bound;
            if (__ballot(le) != 0ull) {
                if (STATS) st2.events++;
                const bool eq = d2 == bound;
                const bool take = le && (!eq || tcnt < ties) && cnt < k_out;
                if (take) row[cnt] = i4[u];
                cnt += take ? 1 : 0; tcnt += (take && eq) ? 1 : 0;
            }
            }
        }
    };
    if (nlog <= KwShared<K, BLOCK>::LOG) {
        // every member of a lane's k-best beat that lane's bound when pass 1 saw it, so the log holds them all: replay it, the gather of
        // the next 64 entries in flight while the current ones are scanned
        const float inf = __builtin_inff();
        float4 pn = make_float4(inf, inf, inf, 0.0f);
        if (lane < nlog) { const int id = sh.log[lane]; pn = a.t.pts[id]; pn.w = __int_as_float(id); }
        for (int b = 0; b < nlog; b += 64) {
            const int nb = nlog - b < 64 ? nlog - b : 64;
            const float4 p = pn;
            pn = make_float4(inf, inf, inf, 0.0f);
            if (b + 64 + lane < nlog) { const int id = sh.log[b + 64 + lane]; pn = a.t.pts[id]; pn.w = __int_as_float(id); }
            __builtin_amdgcn_wave_barrier();
            sh.stage.x[lane] = p.x; sh.stage.y[lane] = p.y; sh.stage.z[lane] = p.z; sh.stage.i[lane] = __float_as_int(p.w);
            __builtin_amdgcn_wave_barrier();
            if (STATS) { st2.batches++; st2.cands += nb; }
            scan2(nb);
        }
    } else if constexpr (!BLOCK) kw_pass<true, STATS>(a.t, m, sh.stk, live, start_node, g0, q.x, q.y, q.z, plo, phi, a.keep, sh.stage, 0x7fffffff, [&]() { return bound; }, scan2, st2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the rows this wavefront has just stored are read back below
    if (STATS && a.stamps && lane == 0) {                       // diagnostics (PCR_KNNW_STATS): 24 words per wavefront
        unsigned long long *w = a.stamps + 24 * (size_t)(g0 / 64);
        const unsigned long long c_end = __builtin_readcyclecounter();
        w[0] = c_mid - c_begin; w[1] = c_end - c_mid; w[2] = wall_clock64();
        const KwStats *ss[2] = {&st1, &st2};
        for (int h = 0; h < 2; h++) {
            w[4 + 8 * h] = ss[h]->tests; w[5 + 8 * h] = ss[h]->exact_inner; w[6 + 8 * h] = ss[h]->pops; w[7 + 8 * h] = ss[h]->leaf_hits;
            w[8 + 8 * h] = ss[h]->climbs; w[9 + 8 * h] = ss[h]->batches; w[10 + 8 * h] = ss[h]->cands; w[11 + 8 * h] = ss[h]->events;
        }
        w[20] = st1.t_walk; w[21] = st1.t_stage; w[22] = st1.t_scan; w[23] = c_end - c_mid;
    }

    // ---- epilogue in float64 on the selected neighbours (inputs are exact float32 -> same values as the oracle), one query per lane
    const double qx = q.x, qy = q.y, qz = q.z;
    const int cmax = __builtin_amdgcn_readfirstlane((int)kw_wave_fmax((float)cnt));
    if (MODE == KNN_MODE_SOR) {
        double s = 0, c = 0;
        for (int j = 0; j < cmax; j++) {
            if (j < cnt) {
                const float4 p = a.t.pts[row[j]];
                const double dx = (double)p.x - qx, dy = (double)p.y - qy, dz = (double)p.z - qz;
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (d2 < a.r2cap) { s += sqrt(d2); c += 1.0; }
            }
        }
        if (served) {
            a.avg[qi] = c > 0 ? s / c : -1.0;
            for (int j = cnt; j < a.list_pitch; j++) row[j] = -1;              // empty slots of the caller's list
        }
    } else if (MODE == KNN_MODE_NORMALS) {
        double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, c = 0;
        for (int j = 0; j < cmax; j++) {
            if (j < cnt) {
                const float4 p = a.t.pts[row[j]];
                const double x = p.x, y = p.y, z = p.z;
                const double dx = x - qx, dy = y - qy, dz = z - qz;
                if (dx * dx + dy * dy + dz * dz < a.r2cap) {
                    cu[0] += x; cu[1] += y; cu[2] += z;
                    cu[3] += x * x; cu[4] += x * y; cu[5] += x * z; cu[6] += y * y; cu[7] += y * z; cu[8] += z * z;
                    c += 1.0;
                }
            }
        }
        if (served) {
            double C6[6];
            if (c >= 3.0) {
                for (int t = 0; t < 9; t++) cu[t] = cu[t] / c;       // cumulants /= n, as Open3D
                C6[0] = cu[3] - cu[0] * cu[0]; C6[1] = cu[4] - cu[0] * cu[1]; C6[2] = cu[5] - cu[0] * cu[2];
                C6[3] = cu[6] - cu[1] * cu[1]; C6[4] = cu[7] - cu[1] * cu[2]; C6[5] = cu[8] - cu[2] * cu[2];
            } else { C6[0] = 1; C6[1] = 0; C6[2] = 0; C6[3] = 1; C6[4] = 0; C6[5] = 1; }
            if (a.cov6) { for (int t = 0; t < 6; t++) a.cov6[(size_t)oq * 6 + t] = (float)C6[t]; }
            if (a.normals) {
                double nv[3];
                d_fast_eigen3x3(C6, nv);
                const double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                double px = 0, py = 0, pz = 0;
                if (a.prior) { const float4 pr = a.prior[oq]; px = pr.x; py = pr.y; pz = pr.z; }
                if (nn == 0.0 || !(nn == nn)) { if (a.prior) { nv[0] = px; nv[1] = py; nv[2] = pz; } else { nv[0] = 0; nv[1] = 0; nv[2] = 1; } }
                if (a.prior && nv[0] * px + nv[1] * py + nv[2] * pz < 0.0) { nv[0] = -nv[0]; nv[1] = -nv[1]; nv[2] = -nv[2]; }
                a.normals[oq] = make_float4((float)nv[0], (float)nv[1], (float)nv[2], 0.0f);
            }
        }
    } else {
        if (served) {
            for (int j = 0; j < a.k; j++) {
                float d2 = __builtin_inff();
                if (j < cnt) { const float4 p = a.t.pts[row[j]]; d2 = pcr_d2(p.x - q.x, p.y - q.y, p.z - q.z); }
                else row[j] = -1;
                a.dbg_d2[(size_t)qi * a.k + j] = d2;
            }
            if (a.dbg_cnt) a.dbg_cnt[qi] = cnt;
        }
    }
}

// 4 wavefronts per SIMD up to K = 32 (128 VGPRs, no spill; the compiler left alone takes 131 and gets 3; 5 would spill 45)
#define KW_OCC(K) __attribute__((amdgpu_waves_per_eu(K <= 32 ? 4 : 2, K <= 32 ? 4 : 3)))
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC(K) k_knn_wave(KnnArgs a) { d_knn_wave<MODE, K>(a); }
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC(K) k_knn_wave_stats(KnnArgs a) { d_knn_wave<MODE, K, true>(a); }
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC(K) k_knn_wave_batch(KnnBatch b) { d_knn_wave<MODE, K>(b.a[blockIdx.y]); }
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC(K) k_knn_wave_batchp(const KnnArgs *a) { d_knn_wave<MODE, K>(a[blockIdx.y]); }
// BLOCK mode (kw_pass_block): needs a.l1tab and a.hard
// (93 VGPRs without the walk's state: 5 wavefronts per SIMD)
#define KW_OCC_BLK __attribute__((amdgpu_waves_per_eu(5, 5)))
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC_BLK k_knn_wave_blk_batch(KnnBatch b) { d_knn_wave<MODE, K, false, false, true>(b.a[blockIdx.y]); }
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC_BLK k_knn_wave_blk_batchp(const KnnArgs *a) { d_knn_wave<MODE, K, false, false, true>(a[blockIdx.y]); }
