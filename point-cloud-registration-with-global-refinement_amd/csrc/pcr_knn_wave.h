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

#define KW_BS 128                       // 2 wavefronts per workgroup (LDS: ~6 KB per wavefront at K <= 32, ~9 KB above)

// candidate batch in LDS, structure of arrays: a step of 4 candidates is three wave-uniform 16-B reads (and their register pairs could
// feed packed float32 math, see PK below)
#ifndef KW_STEP
#define KW_STEP 4          // candidates per scan step (read back from LDS with a wave-uniform address); 2: diagnostic variant
#endif
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
template <int K>
struct KwShared {
    static constexpr int LOG = K <= 32 ? 768 : 1536;   // pass-1 log capacity (events per wavefront at k = 30: mean 260, p99 580)
    KwStage stage;                      // candidate batch
    int log[LOG];                       // indices of the candidates that beat the bound of some lane in pass 1
    KwStack stk;
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
    return pcr_xoct_min(kw_octet_fmin(v));             // (across the octets inside the VALU: row_ror:8 + the permlane swaps of gfx950, no ds_bpermute)
}
__device__ static inline float kw_wave_fmax(float v) {
    return pcr_xoct_max(kw_octet_fmax(v));
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
struct KwStats { int tests, exact_inner, pops, leaf_hits, climbs, batches, cands, events; unsigned long long t_walk, t_stage, t_scan; int culled; };
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
        glo[d] = pcr_xoct_min(sglo[d]); ghi[d] = pcr_xoct_max(sghi[d]);
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
            // CULL at staging time, one candidate per lane: a point farther from the BOX of the wavefront's queries than the largest bound of
            // any lane beats nobody's bound (fat leaves are staged whole and straddle the balls: half of their points go here, at ~25
            // instructions per batch instead of 8 per candidate); the survivors are compacted to the front, infinity behind them
            const float wmax = kw_wave_fmax(worst());
            const bool in = p.x < inf && beats(pcr_box_d2(make_float4(glo[0], glo[1], glo[2], 0.0f), make_float4(ghi[0], ghi[1], ghi[2], 0.0f), p.x, p.y, p.z), wmax);
            const unsigned long long km = __ballot(in);
            const int nkeep = __builtin_popcountll(km), below = __builtin_popcountll(km & ((1ull << lane) - 1ull));
            const int slot = in ? below : nkeep + (lane - below);
            __builtin_amdgcn_wave_barrier();
            stage.x[slot] = in ? p.x : inf; stage.y[slot] = in ? p.y : inf; stage.z[slot] = in ? p.z : inf; stage.i[slot] = __float_as_int(p.w);
            __builtin_amdgcn_wave_barrier();
            unsigned long long ts0 = 0;
            if (STATS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts0 = __builtin_readcyclecounter(); st.t_stage += ts0 - tw1; st.culled += nb - nkeep; }
            if (nkeep > 0) scan(nkeep);
            if (STATS) st.t_scan += __builtin_readcyclecounter() - ts0;
            nb = nx; pre_n = nx; bidx = nidx;
        }
        if (done && nb == 0) return true;
    }
}

// PK: packed float32 distance math (v_pk_add / v_pk_mul / v_pk_fma, two candidates per instruction): 10 % fewer VALU instructions, but 7 % MORE
// wave-cycles and 1-3 % fewer pairs/s (measured, interleaved A/B) -- packed float32 issues at half rate here.  Off.
template <int MODE, int K, bool STATS = false, bool PK = false>
__device__ static inline void d_knn_wave(const KnnArgs &a) {
    constexpr int WPB = KW_BS / 64;
    __shared__ OctMeta m;
    __shared__ KwShared<K> shw[WPB];
    if (threadIdx.x == 0) m = *a.t.meta;
    if (blockIdx.x == 0 && threadIdx.x == 0) { if (a.zero_a) *a.zero_a = 0; if (a.zero_b) *a.zero_b = 0; }
    __syncthreads();
    const int n = __builtin_amdgcn_readfirstlane(m.n);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    KwShared<K> &sh = shw[wv];
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

    KwStats st1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long c_begin = STATS ? __builtin_readcyclecounter() : 0ull;
    // ---- the lane's k smallest squared distances, DESCENDING (sd[0] = the bound); slots beyond k and lanes without a query hold -1
    float sd[K];
#pragma unroll
    for (int j = 0; j < K; j++) sd[j] = (live && j < k) ? a.r2cap_f : -1.0f;

    // ---- candidates come staged 64 at a time in LDS (kw_pass); the scans read them back with a wave-uniform address, in steps of 4
    const int glast = g0 + 63 < n - 1 ? g0 + 63 : n - 1;
    // seed range: the fat leaves of the first and the last query and what lies between them in Morton order (one or two leaves, rarely
    // three); scanned first, skipped by the walk.  A tree of one level is scanned whole.
    int plo = 0, phi = n - 1, start_node = 0;
    if (m.nl >= 2) {
        const int4 pa = a.t.pinfo[g0], pb = a.t.pinfo[glast];
        plo = __builtin_amdgcn_readfirstlane(pa.y); phi = __builtin_amdgcn_readfirstlane(pb.y + pb.z - 1); start_node = __builtin_amdgcn_readfirstlane(pa.x);
    }
    // pass 1: distances only; every candidate that beats the bound of some lane is logged for pass 2
    int nlog = 0;
    const kw_f2 qx2 = {q.x, q.x}, qy2 = {q.y, q.y}, qz2 = {q.z, q.z};
    // x replaces the largest entry when it is smaller, else the chain leaves the list as it is; entries are >= 0 or -1, so they order like
    // their bit patterns (integer min / max: no NaN canonicalisation).  (A lambda of its own since round 5: written out inside the four-way
    // unrolled step the same chain cost the 30-NN instance 12 spilled VGPRs, so 3.)
    auto insert = [&](const float d2) {
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
    };
    auto scan1 = [&](int nb) {
#if KW_STEP == 2         // two candidates per step: 16 VGPRs of read-ahead instead of 32
        float2 nX = *(const float2 *)&sh.stage.x[0], nY = *(const float2 *)&sh.stage.y[0], nZ = *(const float2 *)&sh.stage.z[0];
        int2 nI = *(const int2 *)&sh.stage.i[0];
        for (int j4 = 0; j4 < nb; j4 += 2) {
            const float2 X = nX, Y = nY, Z = nZ;
            const int i4[2] = {nI.x, nI.y};
            nX = *(const float2 *)&sh.stage.x[j4 + 2]; nY = *(const float2 *)&sh.stage.y[j4 + 2]; nZ = *(const float2 *)&sh.stage.z[j4 + 2];
            nI = *(const int2 *)&sh.stage.i[j4 + 2];
            float d4[2];
            d4[0] = pcr_d2(X.x - q.x, Y.x - q.y, Z.x - q.z); d4[1] = pcr_d2(X.y - q.x, Y.y - q.y, Z.y - q.z);
#else
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
#endif
#pragma unroll
            for (int u = 0; u < KW_STEP; u++) {
            const float d2 = d4[u];
            if (__ballot(d2 < sd[0]) != 0ull) {
                if (STATS) st1.events++;
                sh.log[nlog < KwShared<K>::LOG ? nlog : KwShared<K>::LOG - 1] = i4[u];      // all lanes, one address, one value
                nlog = __builtin_amdgcn_readfirstlane(nlog + 1);
                insert(d2);
            }
            }
        }
    };
    // A wavefront whose 64 queries lie far apart (sparse regions, a jump of the Morton curve inside the group) would drag all its lanes
    // through the neighbourhoods of each: ~1 % of the wavefronts took 3-6x the mean and set the length of the launch.  They stop after
    // a.wave_budget batches (or when the log is full) and flag their queries in a.hard for the octet kernel, which serves 8 per wavefront.
    const bool handing = a.hard || a.hard_list;
    const bool finished = kw_pass<false, STATS>(a.t, m, sh.stk, live, start_node, g0, q.x, q.y, q.z, plo, phi, a.keep, sh.stage, handing ? a.wave_budget : 0x7fffffff, [&]() { return sd[0]; }, scan1, st1);
    if (handing) {
        const bool give_up = !finished || nlog > KwShared<K>::LOG;      // (wave-uniform)
        if (a.hard) { if (qi < n) a.hard[oq] = (give_up && live) ? 1 : 0; }
        else if (give_up && lane == 0) a.hard_list[atomicAdd(a.hard_count, 1)] = g0;      // one atomic per wavefront that gives up (~1 %)
        if (give_up) return;
    }
    const unsigned long long c_mid = STATS ? __builtin_readcyclecounter() : 0ull;

    // ---- pass 2: indices under the final bound.  `ties` = how many list entries equal the bound: that many candidates AT the bound belong
    // to the k-best (a radius-capped list that is not full takes none: its bound is the cap itself)
    const float bound = sd[0];
    int ties = 0;
#pragma unroll
    for (int j = 0; j < K; j++) ties += (sd[j] == bound) ? 1 : 0;
    if (!live || bound == a.r2cap_f) ties = 0;
    int cnt = 0, tcnt = 0;
    auto scan2 = [&](int nb) {
#if KW_STEP == 2         // two candidates per step: 16 VGPRs of read-ahead instead of 32
        float2 nX = *(const float2 *)&sh.stage.x[0], nY = *(const float2 *)&sh.stage.y[0], nZ = *(const float2 *)&sh.stage.z[0];
        int2 nI = *(const int2 *)&sh.stage.i[0];
        for (int j4 = 0; j4 < nb; j4 += 2) {
            const float2 X = nX, Y = nY, Z = nZ;
            const int i4[2] = {nI.x, nI.y};
            nX = *(const float2 *)&sh.stage.x[j4 + 2]; nY = *(const float2 *)&sh.stage.y[j4 + 2]; nZ = *(const float2 *)&sh.stage.z[j4 + 2];
            nI = *(const int2 *)&sh.stage.i[j4 + 2];
            float d4[2];
            d4[0] = pcr_d2(X.x - q.x, Y.x - q.y, Z.x - q.z); d4[1] = pcr_d2(X.y - q.x, Y.y - q.y, Z.y - q.z);
#else
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
#endif
#pragma unroll
            for (int u = 0; u < KW_STEP; u++) {
            const float d2 = d4[u];
            const bool le = d2 <= bound;
            if (__ballot(le) != 0ull) {
                if (STATS) st2.events++;
                const bool eq = d2 == bound;
                const bool take = le && (!eq || tcnt < ties) && cnt < k;
                if (take) row[cnt] = i4[u];
                cnt += take ? 1 : 0; tcnt += (take && eq) ? 1 : 0;
            }
            }
        }
    };
    if (nlog <= KwShared<K>::LOG) {
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
    } else kw_pass<true, STATS>(a.t, m, sh.stk, live, start_node, g0, q.x, q.y, q.z, plo, phi, a.keep, sh.stage, 0x7fffffff, [&]() { return bound; }, scan2, st2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the rows this wavefront has just stored are read back below
    if (STATS && a.stamps && lane == 0) {                       // diagnostics (PCR_KNNW_STATS): 24 words per wavefront
        unsigned long long *w = a.stamps + 24 * (size_t)(g0 / 64);
        const unsigned long long c_end = __builtin_readcyclecounter();
        w[0] = c_mid - c_begin; w[1] = c_end - c_mid; w[2] = wall_clock64(); w[3] = (unsigned long long)st1.culled;
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
        if (live) {
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
        if (live) {
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
        if (live) {
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

// 4 wavefronts per SIMD up to K = 32 (128 VGPRs, 18 spilled; in the default bench, groups of 6 x 4 in flight: 3 wavefronts, 144 VGPRs and no
// spill 617 pairs/s, 4 wavefronts 643, 5 wavefronts, 96 VGPRs and 64 spilled 588)
#define KW_OCC(K) __attribute__((amdgpu_waves_per_eu(K <= 32 ? 4 : 2, K <= 32 ? 4 : 3)))
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC(K) k_knn_wave(KnnArgs a) { d_knn_wave<MODE, K>(a); }
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC(K) k_knn_wave_stats(KnnArgs a) { d_knn_wave<MODE, K, true>(a); }
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC(K) k_knn_wave_batch(KnnBatch b) { d_knn_wave<MODE, K>(b.a[blockIdx.y]); }
template <int MODE, int K> __global__ void __launch_bounds__(KW_BS) KW_OCC(K) k_knn_wave_batchp(const KnnArgs *a) { d_knn_wave<MODE, K>(a[blockIdx.y]); }
