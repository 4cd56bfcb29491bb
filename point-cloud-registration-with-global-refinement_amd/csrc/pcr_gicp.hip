// pcr_gicp.hip -- the GICP inner loop on gfx950 (K10 + K11 + K12 fused into ONE kernel per iteration):
//   transform -> exact 1-NN within max_dist over the target BVH (warm-started from the previous match)
//   -> plane-to-plane whitening W = (Cs+Ct)^(-1/2) in closed form -> 3 robustly weighted rows
//   -> wavefront/LDS reduction of the 21+6 normal-equation sums -> last-arriving workgroup sums the
//   per-workgroup partials in fixed order, tests convergence, solves the 6x6 system (pivoted LDL^T) and
//   left-multiplies the pose, all on the device.  The host only polls a 'done' word every few launches.
// Reference behaviour: Open3D RegistrationGeneralizedICP / RegistrationICP /
// TransformationEstimationForGeneralizedICP::ComputeTransformation as called at ALL_FUNCTIONS.py:304-311 and
// 2_MGICP_refinement_in_NCLT_dataset.py:155-162.
// Floating-point contraction per SOURCE EXPRESSION only (the command line's default for HIP, "fast", lets the back end fuse a multiply with an add
// of another statement, and whether it does depends on the code around it: the float64 linearisation inlined into the fused kernel with its
// inputs already in registers came out with other fused pairs than the same function in k_icp_lin -- poses 1e-7 apart under L1, where the
// kernel forms are promised to give the same bits).  Before the includes: the helpers inlined into these kernels follow the same rule.
#pragma clang fp contract(on)
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "pcr_octree.h"

#define ICP_BS 256
#define NV 30            // 21 (upper JTJ) + 6 (JTr) + sum r^2 + sum d^2 + count
#define NVP 32           // padded row of a partial
#define LIN_BS 512           // linearise kernel: few fat workgroups => few partial rows for the last one to gather
#define LIN_MAX_BLOCKS 128

enum { ICP_MODE_GICP = 0, ICP_MODE_EVAL = 1, ICP_MODE_GICP_COV = 2 };   // _COV: per-point covariances given (GICP_robusto path)

struct IcpState {
    double T[16];
    double fitness, rmse;       // of the most recent search
    double sums[NVP];           // reduced sums of the most recent launch
    long long count;
    int iter;                   // pose updates applied
    int launches;               // searches done
    int done, converged;
    unsigned int ticket;
    int ns;                     // source points of this problem (for the byte model)
    unsigned long long t_start; // s_memrealtime (100 MHz) stamp of workgroup 0 at kernel entry
    unsigned long long t_live;  // sum over live launches of (last workgroup's exit stamp - t_start)
    unsigned long long t_dbg[4]; // diagnostic stamps (sum): wg0 after search, wg0 after reduce, last wg entering tail, last wg after partial sums
    unsigned long long searched; // queries whose certificate did not hold, summed over the launches after the cold one (bench: fraction searched)
    double dfit, drmse;         // |change| of fitness / rmse at the most recent launch (what the criteria test): the host sizes the next chunk of launches by them
};

struct IcpArgs {
    const float4 *src_pts, *src_nrm; const int *ns_ptr;
    const float *src_cov6, *tgt_cov6;            // optional raw covariances (xx,xy,xz,yy,yz,zz), Morton order
    const float4 *tgt_pts, *tgt_nrm; OctView tgt; const int *nt_ptr;
    GridView grid;                               // cell hash of the target (grid.tab == nullptr: search the octree)
    int32_t *match; int src_cap;
    int tile_rt;                                 // fused kernel: source points per workgroup of THIS problem when smaller than the kernel's tile (0: the kernel's): a launch of a lockstep
                                                 //   group has one kernel form, its pairs their own tile by their own size (same bits however the batch is cut)
    float4 *ref; int32_t *rbest;                 // per source point: position and margin / nearest point of its last search (skip certificate, tree-walk form)
    int4 *clist;                                 // cell-hash form: the PCR_NN_K nearest target points of the last search (list certificate); ref.w = distance of the next one
    float r2s, rs_minus_r;                       // search cap (r + g)^2 of the certificate mode and g = the unmatched margin
    int verify;                                  // diagnostics (PCR_ICP_VERIFY): search certified queries too and report disagreements
    IcpState *state;
    double *partials;
    double max_dist2; float r2f;
    int loss; double loss_k; double a;          // a = 1 - epsilon
    double rel_fit, rel_rmse; int max_it;
    int single;                                  // 1: linearise once, never update (debug / evaluate)
    int dbg_visits;                              // diagnostics: store node/leaf visit counts instead of matches
    int dbg_phase;                               // diagnostics (PCR_ICP_PHASE = 1 / 2): t_dbg[1] <- slowest workgroup's end of phase A / B of the fused kernel
    unsigned long long *stamps_nn, *stamps_it;   // diagnostics (PCR_ICP_STAMPS): per-wavefront clocks of the first 16 launches
    // all scales of a pair behind ONE argument slot (pcr_dev_gicp_group_scales; ms_scales == 0: a single scale): when the criteria of scale ms_index
    // hold, the last workgroup stores the state in ms_hist[ms_index] and -- unless it was the last scale -- starts the next one itself: launches,
    // iterations and flags back to their start values, the pose kept, and ms_args[ms_index + 1] copied over *ms_self, the slot the next launch reads.
    // A launch that finds launches == 0 is the scale's first: no certificate is valid, every query is searched (what k_icp_nn + k_icp_lin do).
    const IcpArgs *ms_args; IcpArgs *ms_self; IcpState *ms_hist; int ms_scales, ms_index;
};
#define ICP_STAMP_LAUNCHES 16

struct IcpInit { double T[16]; };
__device__ static inline void d_icp_init(IcpState *st, const IcpInit &in) {
    if (threadIdx.x == 0) {
        for (int k = 0; k < 16; k++) st->T[k] = in.T[k];
        st->fitness = 0; st->rmse = 0; st->count = 0; st->iter = 0; st->launches = 0; st->done = 0; st->converged = 0; st->ticket = 0; st->ns = 0; st->t_start = 0; st->t_live = 0; st->searched = 0; st->dfit = 1e300; st->drmse = 1e300; for (int k = 0; k < 4; k++) st->t_dbg[k] = 0;
        for (int k = 0; k < NVP; k++) st->sums[k] = 0;
    }
}
__global__ void k_icp_init(IcpState *st, IcpInit in) { d_icp_init(st, in); }
// group form: state g of a lockstep group starts from pose g (poses in device memory)
__global__ void k_icp_init_g(IcpState *st, const IcpInit *in) { d_icp_init(st + blockIdx.x, in[blockIdx.x]); }

// ---- exact 1-NN of one query per octet over the target's linear octree.  Called by all 64 lanes of a wavefront;
// `live` is octet-uniform.  hint >= 0: a target point near the answer (previous match, or the previous start point of
// an unmatched query) -- the search starts bottom-up from its leaf; hint < 0: greedy nearest-box descent from the root.
// Returns the best index within r2cap (or -1) and, through start_pt, a point of the start leaf (next launch's hint).
template <int OPB>
__device__ static inline int oct_nn_query(const OctView &t, const OctMeta &m, OctStack<OPB> &stk, bool live, float qx, float qy, float qz,
                                          float r2cap, int hint, int ol, int oct, int ob, int *start_pt, int *visits, float *d1_out, float *d2_out) {
    // The walk keeps the TWO smallest squared distances (both capped at r2cap) and prunes with the second: the gap between
    // them is the certificate that lets later launches skip this query while it has moved by less than half the gap.
    int best = -1; float bestd = r2cap, secd = r2cap;
    const bool active = live && m.nl >= 1;
    auto visit = [&](int first, int count) {                 // wave-wide; count == 0: octet idle
        int base = first; const int end = first + count;
        float d = 3.4e38f, dd = 3.4e38f; int id = -1;         // lane-local best / second over all steps (4 loads in flight per step)
        while (__ballot(base < end) != 0ull) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int idx = base + OCT * u + ol; p[u] = t.pts[idx < end ? idx : (end > first ? end - 1 : 0)]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = base + OCT * u + ol;
                const float du = pcr_d2(p[u].x - qx, p[u].y - qy, p[u].z - qz);
                if (idx < end) {
                    if (du < d) { dd = d; d = du; id = idx; }        // increasing idx: ties keep the lower index
                    else dd = fminf(dd, du);
                }
            }
            base += 4 * OCT;
        }
        // octet arg-min (ties -> lower index) without the LDS crossbar: min distance, then min index among its holders;
        // the runner-up is the smallest of the winner lane's second and the other lanes' best
        const float dmin = pcr_octet_min(d);
        int cand = (d == dmin && id >= 0) ? id : 0x7fffffff;
        cand = min(cand, pcr_dpp_i<PCR_DPP_XOR1>(cand)); cand = min(cand, pcr_dpp_i<PCR_DPP_XOR2>(cand)); cand = min(cand, pcr_dpp_i<PCR_DPP_HMIRROR>(cand));
        const float sec = pcr_octet_min(id == cand ? dd : d);
        if (cand != 0x7fffffff) {
            if (dmin < bestd || (dmin == bestd && (unsigned)cand < (unsigned)best)) { secd = fminf(fminf(secd, bestd), sec); bestd = dmin; best = cand; }
            else secd = fminf(secd, dmin);
        }
    };
    int node = 0, node_li = 0;
    const bool warm = active && hint >= 0;
    int s_first = 0, s_count = 0, s_parent = 0, s_sib = 0, s_nsib = 1; uint64_t s_key = 0;
    if (warm) {
        // one hop: the hint point and the record of the level-1 node around it ("fat leaf"); next hop (issued here,
        // consumed after the scan of the node's points): the node's up-link and key
        const int4 pi = t.pinfo[hint];
        node = pi.x; s_first = pi.y; s_count = pi.z; node_li = pi.w;
        const int4 u = t.up[m.off[node_li] + node]; s_key = t.keys[pi.y];
        s_parent = u.x; s_sib = u.y; s_nsib = u.z;
    }
    const bool cold = active && !warm;
    if (__ballot(cold) != 0ull) {
        const int g = oct_greedy_leaf(t, m, cold, qx, qy, qz, ol);
        if (cold) {
            node = g; node_li = 0;
            const size_t j = (size_t)(m.off[0] + g);
            s_first = __float_as_int(t.nodes[2 * j].w); s_count = __float_as_int(t.nodes[2 * j + 1].w);
            const int4 u = t.up[j]; s_key = t.keys[s_first];
            s_parent = u.x; s_sib = u.y; s_nsib = u.z;
        }
    }
    if (active) *start_pt = s_first;
    oct_search<OPB>(t, m, stk, active, node, node_li, s_first, s_count, s_key, s_parent, s_sib, s_nsib, qx, qy, qz, [&]() { return secd; }, visit,
                    [](int, int) { return false; }, ol, oct, ob, visits);
    *d1_out = bestd; *d2_out = secd;
    return best;
}

// ---- LIST CERTIFICATE of the cell-hash searches (grid_nn_query8).  The last search of the query, at position ref.xyz, found the listed K
// nearest target points and D = ref.w, the distance of the (K+1)-th (or the search cap when fewer lie inside it): every point NOT listed
// is at least D - delta away from the query now, delta = |q - ref| (triangle inequality).  So while the nearest LISTED point is closer
// than that, it is the exact nearest neighbour (ties inside the list -> lower index, as in a search; a tie with an unlisted point is
// excluded by the strict test and the slack); when even D - delta lies beyond the float32 radius and no listed point is nearer, the
// query is unmatched; otherwise it must be searched again.  Returns 1 (decided: *cand = the neighbour, or -1) or 0 (search).
// The test needs K gathered points per query and launch, one dependent load after the record -- against a search it is one round trip
// instead of four or five, and (d5 - d1) / 2 of room instead of (d2 - d1) / 2: 0.3 instead of 0.1 voxels on a planar patch.
__device__ static inline void icp_list_load(const float4 *__restrict__ tgt_pts, const int4 lst, float4 (&tp)[PCR_NN_K]) {
    const int ids[PCR_NN_K] = {lst.x, lst.y, lst.z, lst.w};
#pragma unroll
    for (int k = 0; k < PCR_NN_K; k++) tp[k] = tgt_pts[ids[k] >= 0 ? ids[k] : 0];
}
__device__ static inline int icp_list_decide(float qx, float qy, float qz, const float4 refv, const int4 lst, const float4 (&tp)[PCR_NN_K], float r2f, int *cand, float4 *cand_pt) {
    const int ids[PCR_NN_K] = {lst.x, lst.y, lst.z, lst.w};
    const float delta = sqrtf(pcr_d2(qx - refv.x, qy - refv.y, qz - refv.z));
    const float slack = 2e-5f + 1e-6f * (fabsf(qx) + fabsf(qy) + fabsf(qz));
    const float room = refv.w - delta - slack;
    float best = 3.4e38f; int bi = -1; float4 bp = tp[0];
#pragma unroll
    for (int k = 0; k < PCR_NN_K; k++) {
        const float c = pcr_d2(tp[k].x - qx, tp[k].y - qy, tp[k].z - qz);
        if (ids[k] >= 0 && (c < best || (c == best && ids[k] < bi))) { best = c; bi = ids[k]; bp = tp[k]; }
    }
    *cand = -1; *cand_pt = bp;
    if (!(room > 0.0f)) return 0;
    const float room2 = room * room;
    if (bi >= 0 && best < room2) { *cand = bi; return 1; }
    return room2 > r2f ? 1 : 0;
}
__device__ static inline int icp_list_eval(const float4 *__restrict__ tgt_pts, float qx, float qy, float qz, const float4 refv, const int4 lst, float r2f, int *cand) {
    float4 tp[PCR_NN_K], bp;
    icp_list_load(tgt_pts, lst, tp);
    return icp_list_decide(qx, qy, qz, refv, lst, tp, r2f, cand, &bp);
}

__device__ static inline double icp_weight(int loss, double k, double r) {
    if (loss == PCR_LOSS_L1) return 1.0 / fmax(fabs(r), 1e-300);    // Open3D: 1/|r| (unguarded); guard only against r == 0
    if (loss == PCR_LOSS_GM) { const double d = k + r * r; return k / (d * d); }
    return 1.0;
}

// W = (M^-1)^(1/2), symmetric principal root of an SPD 3x3 (general covariances: no closed form): cyclic Jacobi in
// float64, fully unrolled (static indexing only -> registers)
__device__ static inline void icp_sym3_inv_sqrt(const double *M6 /*xx,xy,xz,yy,yz,zz*/, double *W) {
    double a[3][3] = {{M6[0], M6[1], M6[2]}, {M6[1], M6[3], M6[4]}, {M6[2], M6[4], M6[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
    for (int sweep = 0; sweep < 6; sweep++) {
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = a[p][q];
            if (apq != 0.0) {
                const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
                for (int k = 0; k < 3; k++) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - sn * akq; a[k][q] = sn * akp + c * akq; }
#pragma unroll
                for (int k = 0; k < 3; k++) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - sn * aqk; a[q][k] = sn * apk + c * aqk; }
#pragma unroll
                for (int k = 0; k < 3; k++) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - sn * vkq; v[k][q] = sn * vkp + c * vkq; }
            }
        }
    }
    const double l0 = 1.0 / sqrt(a[0][0]), l1 = 1.0 / sqrt(a[1][1]), l2 = 1.0 / sqrt(a[2][2]);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) W[i * 3 + j] = v[i][0] * v[j][0] * l0 + v[i][1] * v[j][1] * l1 + v[i][2] * v[j][2] * l2;
}

// 6x6 symmetric solve on ONE lane.  Fast path: LDL^T without pivoting, fully unrolled so that every array lives in
// registers (no scratch traffic on the critical path of the iteration).  If a pivot is not strictly positive/finite
// (semi-definite or indefinite system) fall back to the diagonally pivoted LDL^T of Eigen::LDLT on LDS arrays.
__device__ static bool icp_ldlt6_fast(const double *S /*21 upper-triangular sums*/, const double *b /*6*/, double *x) {
    double A[6][6];
    {
        int t = 0;
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = p; q < 6; q++) { A[q][p] = S[t]; t++; }      // lower triangle
    }
    double D[6], y[6];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double d = A[k][k];
        ok = ok && (d > 0.0) && isfinite(d);
        D[k] = d;
        const double inv = 1.0 / d;
        double col[6];
#pragma unroll
        for (int i = k + 1; i < 6; i++) col[i] = A[i][k];            // original column k below the diagonal
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
            const double l = col[i] * inv;
#pragma unroll
            for (int j = k + 1; j <= i; j++) A[i][j] -= l * col[j];
            A[i][k] = l;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = b[i];
#pragma unroll
        for (int j = 0; j < i; j++) s -= A[i][j] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] /= D[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; j++) s -= A[j][i] * x[j];
        x[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) ok = ok && isfinite(x[i]);
    return ok;
}

// pivoted fallback; w = LDS workspace of >= 36+36+6+6+6 doubles and 6 ints
__device__ static bool icp_ldlt6_pivoted(const double *S, const double *b6, double *x6, double *w, int *perm) {
    double *A = w, *L = w + 36, *D = w + 72, *y = w + 78, *z = w + 84;
    {
        int t = 0;
        for (int p = 0; p < 6; p++) for (int q = p; q < 6; q++) { A[p * 6 + q] = S[t]; A[q * 6 + p] = S[t]; t++; }
    }
    for (int i = 0; i < 36; i++) L[i] = 0;
    for (int i = 0; i < 6; i++) perm[i] = i;
    for (int k = 0; k < 6; k++) {
        int piv = k; double best = fabs(A[k * 6 + k]);
        for (int i = k + 1; i < 6; i++) if (fabs(A[i * 6 + i]) > best) { best = fabs(A[i * 6 + i]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < 6; j++) { double t = A[k * 6 + j]; A[k * 6 + j] = A[piv * 6 + j]; A[piv * 6 + j] = t; }
            for (int j = 0; j < 6; j++) { double t = A[j * 6 + k]; A[j * 6 + k] = A[j * 6 + piv]; A[j * 6 + piv] = t; }
            for (int j = 0; j < k; j++) { double t = L[k * 6 + j]; L[k * 6 + j] = L[piv * 6 + j]; L[piv * 6 + j] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        const double d = A[k * 6 + k];
        D[k] = d; L[k * 6 + k] = 1.0;
        if (d == 0.0 || !isfinite(d)) return false;
        for (int i = k + 1; i < 6; i++) L[i * 6 + k] = A[i * 6 + k] / d;
        for (int i = k + 1; i < 6; i++)
            for (int j = k + 1; j < 6; j++) A[i * 6 + j] -= L[i * 6 + k] * d * L[j * 6 + k];
    }
    for (int i = 0; i < 6; i++) { double s = b6[perm[i]]; for (int j = 0; j < i; j++) s -= L[i * 6 + j] * y[j]; y[i] = s; }
    for (int i = 0; i < 6; i++) y[i] /= D[i];
    for (int i = 5; i >= 0; i--) { double s = y[i]; for (int j = i + 1; j < 6; j++) s -= L[j * 6 + i] * z[j]; z[i] = s; }
    bool ok = true;
    for (int i = 0; i < 6; i++) { const double v = z[i]; ok = ok && isfinite(v); w[90 + perm[i]] = v; }
    for (int i = 0; i < 6; i++) x6[i] = w[90 + i];
    return ok;
}

// what a search leaves for the launches after it: the certificate of the query (list form for the cell-hash searches, margin form for
// the tree walks) and its match / start hint
static_assert(PCR_NN_K == 4, "the list certificate is an int4 record");
template <bool GRID>
__device__ static inline void icp_store_cert(const IcpArgs &a, int qi, float qx, float qy, float qz, int best, const int (&nnk)[PCR_NN_K], float d1, float d2, int start_pt) {
    if (GRID) {
        a.ref[qi] = make_float4(qx, qy, qz, sqrtf(d2));          // d2 = squared distance of the first point NOT listed (or the search cap)
        a.clist[qi] = make_int4(nnk[0], nnk[1], nnk[2], nnk[3]);
    } else {
        const float slack = 2e-5f + 1e-6f * (fabsf(qx) + fabsf(qy) + fabsf(qz));
        // nothing within the (enlarged) search cap: still nothing within max_dist while the query moves by less than the gap
        const float margin = best >= 0 ? 0.5f * (sqrtf(d2) - sqrtf(d1)) - slack : a.rs_minus_r - slack;
        a.ref[qi] = make_float4(qx, qy, qz, margin > 0.0f ? margin : 0.0f);
        a.rbest[qi] = best;
    }
    a.match[qi] = best >= 0 ? best : -(start_pt + 2);
}

// ---- kernel 1 of an iteration: exact 1-NN of every transformed source point, ONE query per octet (32 per
// workgroup).  Latency-bound pointer chasing, so it runs at full occupancy (few registers, many wavefronts).
template <bool GRID>
__device__ static inline void d_icp_nn(const IcpArgs &a) {
    IcpState *st = a.state;
    constexpr int OPB = ICP_BS / OCT;
    __shared__ OctMeta m;
    __shared__ OctStack<OPB> stk;
    // everything the prologue needs is requested at once (each dependent hop costs ~1 us of wavefront life)
    const int done = st->done, launches = st->launches;
    const int ns = *a.ns_ptr, nt = *a.nt_ptr;
    const int lane = threadIdx.x & 63, oct = lane >> 3, ol = lane & 7, ob = threadIdx.x >> 3;
    const int i = blockIdx.x * OPB + ob;
    const int ic = i < a.src_cap ? i : 0;
    const float4 pf = a.src_pts[ic];
    const int mraw = a.match[ic];
    const float4 refv = a.ref ? a.ref[ic] : make_float4(0, 0, 0, 0);
    int rb = (!GRID && a.rbest) ? a.rbest[ic] : -1;
    const int4 lst = (GRID && a.ref) ? a.clist[ic] : make_int4(-1, -1, -1, -1);
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = st->T[k];
    int mword = 0;
    if (threadIdx.x < (int)(sizeof(OctMeta) / 4)) mword = ((const int *)a.tgt.meta)[threadIdx.x];
    const unsigned gmask = GRID ? *a.grid.dmask : 0u;          // (the cell hash's size lives on the device: requested with everything else)
    if (done) return;
    if ((int)blockIdx.x * OPB >= ns) return;
    const unsigned long long t_wave0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) st->t_start = t_wave0;
    __shared__ float4 rec_q[OPB];          // compacted work list of the workgroup: queries that really need a search
    __shared__ int rec_i[OPB];
    __shared__ int rec_c[OPB];             // diagnostics: certified claim of the record (-2: none)
    __shared__ int n_rec;
    if (threadIdx.x < (int)(sizeof(OctMeta) / 4)) ((int *)&m)[threadIdx.x] = mword;
    if (threadIdx.x == 0) n_rec = 0;
    __syncthreads();
    float qx = 0, qy = 0, qz = 0; int hint = -1;
    bool need = false, cert_flag = false;
    if (i < ns) {
        const double px = pf.x, py = pf.y, pz = pf.z;
        qx = (float)(T[0] * px + T[1] * py + T[2] * pz + T[3]);
        qy = (float)(T[4] * px + T[5] * py + T[6] * pz + T[7]);
        qz = (float)(T[8] * px + T[9] * py + T[10] * pz + T[11]);
        // match[] carries the warm-start hint across launches: >= 0 matched target point, <= -2 -> start point -(v+2)
        const int mv = launches > 0 ? mraw : -1;
        hint = mv >= 0 ? mv : (mv <= -2 ? -(mv + 2) : -1);
        // Certificate of the last search of this query: nearest target point rb at distance d1, every other one at >= d2,
        // margin = (d2 - d1)/2 - slack.  While the query has moved by less than the margin since then, rb is still its
        // unique nearest point (triangle inequality) and no search is needed; k_icp_iter re-tests the radius in float64.
        const float ex = qx - refv.x, ey = qy - refv.y, ez = qz - refv.z;
        bool certified = launches > 0 && a.ref && !a.dbg_visits && refv.w > 0.0f;
        if (GRID) { if (certified) certified = icp_list_eval(a.tgt_pts, qx, qy, qz, refv, lst, a.r2f, &rb) != 0; }   // list certificate: rb = the decided neighbour (or -1)
        else certified = certified && pcr_d2(ex, ey, ez) < refv.w * refv.w;
        need = nt > 0 && (!certified || a.verify);
        cert_flag = certified;
        if (certified && ol == 0 && rb >= 0 && mraw != rb) a.match[i] = rb;     // k_icp_iter may have turned it into a hint (beyond max_dist)
        if (GRID && certified && ol == 0 && rb < 0 && mraw >= 0) a.match[i] = -1;   // decided unmatched after having been matched
    }
    // ---- compact the workgroup's pending queries to the front: wavefronts left without work retire at once
    {
        const unsigned long long nb = __ballot(need && ol == 0);
        int base = 0;
        if (lane == 0 && nb != 0ull) base = atomicAdd(&n_rec, __builtin_popcountll(nb));
        base = __builtin_amdgcn_readfirstlane(base);            // (lane 0 holds it; no ds_bpermute round trip)
        if (need && ol == 0) {
            const int slot = base + __builtin_popcountll(nb & ((1ull << lane) - 1ull));
            rec_q[slot] = make_float4(qx, qy, qz, __int_as_float(hint)); rec_i[slot] = i; rec_c[slot] = cert_flag ? (rb >= 0 ? rb : -1) : -2;
        }
    }
    __syncthreads();
    const bool live = ob < n_rec;
    if (__ballot(live) == 0ull) return;
    int qi = 0;
    if (live) { const float4 r = rec_q[ob]; qx = r.x; qy = r.y; qz = r.z; hint = __float_as_int(r.w); qi = rec_i[ob]; }
    int visits = 0, start_pt = 0; float d1 = 0, d2 = 0;
    int best; int nnk[PCR_NN_K];
    if (GRID) { grid_nn_query8<PCR_NN_K>(a.grid, gmask, a.tgt_pts, live, qx, qy, qz, a.ref ? a.r2s : a.r2f, ol, nnk, &d1, &d2); best = nnk[0]; start_pt = hint >= 0 ? hint : 0; }
    else best = oct_nn_query<OPB>(a.tgt, m, stk, live, qx, qy, qz, a.ref ? a.r2s : a.r2f, hint, ol, oct, ob, &start_pt, a.dbg_visits ? &visits : nullptr, &d1, &d2);
    if (a.verify && ol == 0 && live && rec_c[ob] != -2) {
        const int claim = rec_c[ob];
        // (a decided neighbour may have drifted beyond the search cap r + g -- it is beyond r then, and the search finds nothing: no contradiction)
        bool beyond = false;
        if (claim >= 0 && best < 0) { const float4 cp = a.tgt_pts[claim]; beyond = pcr_d2(cp.x - qx, cp.y - qy, cp.z - qz) >= (a.ref ? a.r2s : a.r2f); }
        const bool bad = claim >= 0 ? (best != claim && !beyond) : (best >= 0 && d1 < a.r2f);
        if (bad) printf("certificate violated: launch %d query %d claim %d found %d d1 %.6f d2 %.6f ref margin %.6f moved %.6f\n", launches, qi, claim, best, sqrtf(d1), sqrtf(d2),
                        a.ref[qi].w, sqrtf(pcr_d2(qx - a.ref[qi].x, qy - a.ref[qi].y, qz - a.ref[qi].z)));
    }
    if (ol == 0 && live) {
        a.match[qi] = a.dbg_visits ? visits : (best >= 0 ? best : -(start_pt + 2));
        if (a.ref && GRID) {
            a.ref[qi] = make_float4(qx, qy, qz, sqrtf(d2));          // d2 = squared distance of the first point NOT listed (or the cap)
            a.clist[qi] = make_int4(nnk[0], nnk[1], nnk[2], nnk[3]);
        } else if (a.ref) {
            const float slack = 2e-5f + 1e-6f * (fabsf(qx) + fabsf(qy) + fabsf(qz));
            // nothing within the (enlarged) search cap: still nothing within max_dist while the query moves by less than the gap
            const float margin = best >= 0 ? 0.5f * (sqrtf(d2) - sqrtf(d1)) - slack : a.rs_minus_r - slack;
            a.ref[qi] = make_float4(qx, qy, qz, margin > 0.0f ? margin : 0.0f);
            a.rbest[qi] = best;
        }
    }
    if (a.stamps_nn && lane == 0 && launches < ICP_STAMP_LAUNCHES) {
        unsigned long long *w = a.stamps_nn + 2 * ((size_t)launches * gridDim.x * (ICP_BS / 64) + (size_t)blockIdx.x * (ICP_BS / 64) + (threadIdx.x >> 6));
        w[0] = t_wave0; w[1] = wall_clock64();
    }
}

// GRID: the correspondence search goes through the target's cell hash (a.grid) instead of its octree -- a compile-time choice, so that
// neither form carries the other's registers (both in one kernel: 52 VGPRs spilled here, 150 VGPRs = one workgroup per CU in the fused kernel)
template <bool GRID> __global__ void __launch_bounds__(ICP_BS) __attribute__((amdgpu_waves_per_eu(8, 8))) k_icp_nn(IcpArgs a) { d_icp_nn<GRID>(a); }
// group form (lockstep group of pairs, blockIdx.y = pair; arguments in device memory)
template <bool GRID> __global__ void __launch_bounds__(ICP_BS) __attribute__((amdgpu_waves_per_eu(8, 8))) k_icp_nn_g(const IcpArgs *__restrict__ a) { d_icp_nn<GRID>(a[blockIdx.y]); }

// ---- one correspondence in float64: accumulates its 30 sums into acc[]; `cand` = candidate target point (or < 0)
struct IcpPre { float4 pf, sn, tf, tn; };    // the inputs of a correspondence already in registers (fused kernel: gathered under its certificate phase)
template <int MODE>
__device__ static inline void icp_point(const IcpArgs &a, const double *T, int i, int ns, int cand_in, double *acc, const IcpPre *pre = nullptr) {
    double qx = 0, qy = 0, qz = 0;
    int cand = -1;
    {
        const float4 pf = pre ? pre->pf : a.src_pts[i];
        const double px = pf.x, py = pf.y, pz = pf.z;
        qx = T[0] * px + T[1] * py + T[2] * pz + T[3];
        qy = T[4] * px + T[5] * py + T[6] * pz + T[7];
        qz = T[8] * px + T[9] * py + T[10] * pz + T[11];
        cand = cand_in;
    }
    if (a.dbg_visits) cand = -1;

    // ---- one correspondence per lane, float64
    if (i < ns && cand >= 0) {
        int best = cand;
        {
            const float4 tf = pre ? pre->tf : a.tgt_pts[best];
            const double dx = qx - (double)tf.x, dy = qy - (double)tf.y, dz = qz - (double)tf.z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < a.max_dist2) {
                acc[28] += d2; acc[29] += 1.0;
                if (MODE == ICP_MODE_GICP || MODE == ICP_MODE_GICP_COV) {
                    double W[9];
                    if (MODE == ICP_MODE_GICP_COV) {
                        // given covariances are used untouched (Open3D: HasCovariances): M = R Cs R^T + Ct
                        const float *cs = a.src_cov6 + (size_t)i * 6, *ct = a.tgt_cov6 + (size_t)best * 6;
                        const double C[3][3] = {{cs[0], cs[1], cs[2]}, {cs[1], cs[3], cs[4]}, {cs[2], cs[4], cs[5]}};
                        const double R[3][3] = {{T[0], T[1], T[2]}, {T[4], T[5], T[6]}, {T[8], T[9], T[10]}};
                        double RC[3][3], M[3][3];
#pragma unroll
                        for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
                            for (int c_ = 0; c_ < 3; c_++) RC[r_][c_] = R[r_][0] * C[0][c_] + R[r_][1] * C[1][c_] + R[r_][2] * C[2][c_];
#pragma unroll
                        for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
                            for (int c_ = 0; c_ < 3; c_++) M[r_][c_] = RC[r_][0] * R[c_][0] + RC[r_][1] * R[c_][1] + RC[r_][2] * R[c_][2];
                        const double M6[6] = {M[0][0] + ct[0], 0.5 * (M[0][1] + M[1][0]) + ct[1], 0.5 * (M[0][2] + M[2][0]) + ct[2],
                                              M[1][1] + ct[3], 0.5 * (M[1][2] + M[2][1]) + ct[4], M[2][2] + ct[5]};
                        icp_sym3_inv_sqrt(M6, W);
                    } else {
                    // effective covariance normals: C = I - a m m^T, m = e1 when n.x < -0.99 (Open3D GetRotationFromE1ToX)
                    const float4 sn = pre ? pre->sn : a.src_nrm[i], tn = pre ? pre->tn : a.tgt_nrm[best];
                    double sx = sn.x, sy = sn.y, sz = sn.z, tx = tn.x, ty = tn.y, tz = tn.z;
                    if (sx < -0.99) { sx = 1; sy = 0; sz = 0; }
                    if (tx < -0.99) { tx = 1; ty = 0; tz = 0; }
                    double inv = 1.0 / sqrt(sx * sx + sy * sy + sz * sz);
                    sx *= inv; sy *= inv; sz *= inv;
                    inv = 1.0 / sqrt(tx * tx + ty * ty + tz * tz);
                    tx *= inv; ty *= inv; tz *= inv;
                    // rotate the source normal into the target frame (covariance R C R^T)
                    const double ux = T[0] * sx + T[1] * sy + T[2] * sz, uy = T[4] * sx + T[5] * sy + T[6] * sz, uz = T[8] * sx + T[9] * sy + T[10] * sz;
                    const double c = ux * tx + uy * ty + uz * tz;
                    // M = 2I - a(uu^T + vv^T): eigenpairs (2 - a(1+c), u+v), (2 - a(1-c), u-v), (2, u x v)
                    const double SQ2 = 1.4142135623730951;
                    const double lp = 2.0 - a.a * (1.0 + c), lm = 2.0 - a.a * (1.0 - c);
                    const double slp = sqrt(lp), slm = sqrt(lm);
                    const double gp = a.a / (2.0 * SQ2 * slp * (SQ2 + slp)), gm = a.a / (2.0 * SQ2 * slm * (SQ2 + slm));
                    const double ex = ux + tx, ey = uy + ty, ez = uz + tz, fx = ux - tx, fy = uy - ty, fz = uz - tz;
                    const double w0 = 1.0 / SQ2;
                    W[0] = w0 + gp * ex * ex + gm * fx * fx; W[1] = gp * ex * ey + gm * fx * fy; W[2] = gp * ex * ez + gm * fx * fz;
                    W[4] = w0 + gp * ey * ey + gm * fy * fy; W[5] = gp * ey * ez + gm * fy * fz; W[8] = w0 + gp * ez * ez + gm * fz * fz;
                    W[3] = W[1]; W[6] = W[2]; W[7] = W[5];
                    }
                    double r2 = 0;
#pragma unroll
                    for (int row = 0; row < 3; row++) {
                        const double wx = W[row * 3], wy = W[row * 3 + 1], wz = W[row * 3 + 2];
                        double J[6];
                        J[0] = qy * wz - qz * wy; J[1] = qz * wx - qx * wz; J[2] = qx * wy - qy * wx;   // q x W_row
                        J[3] = wx; J[4] = wy; J[5] = wz;
                        const double r = wx * dx + wy * dy + wz * dz;
                        const double w = icp_weight(a.loss, a.loss_k, r);
                        int t = 0;
#pragma unroll
                        for (int p = 0; p < 6; p++) {
                            const double wj = w * J[p];
#pragma unroll
                            for (int q = p; q < 6; q++) acc[t++] += wj * J[q];
                            acc[21 + p] += wj * r;
                        }
                        r2 += r * r;
                    }
                    acc[27] += r2;
                } else {
                    // GetInformationMatrixFromPointClouds: sum G^T G over matched TARGET points
                    const double x = tf.x, y = tf.y, z = tf.z;
                    const double G[3][6] = {{0, z, -y, 1, 0, 0}, {-z, 0, x, 0, 1, 0}, {y, -x, 0, 0, 0, 1}};
#pragma unroll
                    for (int row = 0; row < 3; row++) {
                        int t = 0;
#pragma unroll
                        for (int p = 0; p < 6; p++)
#pragma unroll
                            for (int q = p; q < 6; q++) acc[t++] += G[row][p] * G[row][q];
                    }
                }
            } else best = -1;
        }
        if (best < 0) a.match[i] = -(cand + 2);      // beyond max_dist in float64: keep the candidate as next start hint
    }
}

// ---- workgroup reduction of acc[], write-through partial row + ticket, and -- in the last-arriving workgroup -- the end of
// the iteration: gather the rows, fixed-order sums, convergence test, 6x6 solve, pose update.  BS = workgroup size.
template <int MODE, int BS>
__device__ static inline void icp_finish(const IcpArgs &a, IcpState *st, const double *T, double *acc, int nb, int ns, int launches, unsigned long long t_entry, int row,
                                         unsigned long long t_p = 0, unsigned long long t_a = 0, unsigned long long t_b = 0) {
    __shared__ double red[BS / 16][NVP];           // one row per 16-lane DPP row
    __shared__ double fin[16][NVP];
    __shared__ int is_last;
    __shared__ double ldl_w[96];
    __shared__ int ldl_perm[6];
    const unsigned long long t_search = wall_clock64() - t_entry;
    // ---- workgroup reduction: DPP butterfly inside every 16-lane row (no LDS crossbar), then the 64 rows in order
    const int lane = threadIdx.x & 63;
#ifdef ICP_ROWSUM_FULL
#pragma unroll
    for (int k = 0; k < NV; k++) { const double s = pcr_row16_sum(acc[k]); if ((lane & 15) == 0) red[threadIdx.x >> 4][k] = s; }
#else
    {   // halving butterfly (pcr_device.h): lane r of a row ends with the sums of values r and 16 + r -- the same bits as pcr_row16_sum, 218 instead of 384 instructions
        double lo, hi;
        pcr_row16_sum_halving<NV>(acc, lane, &lo, &hi);
        const int r = lane & 15;
        red[threadIdx.x >> 4][r] = lo;
        if (16 + r < NV) red[threadIdx.x >> 4][16 + r] = hi;
    }
#endif
    const unsigned long long t_ws = wall_clock64();
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = red[0][threadIdx.x];
#pragma unroll 8
        for (int w = 1; w < BS / 16; w++) s += red[w][threadIdx.x];
        // publish write-through (sc1): no per-workgroup release fence (a release = whole-L2 write-back; ~700 of
        // them per launch serialised to >100 us).  cdna_hip_programming.md Guideline 16, recipe R1.
        __hip_atomic_store(&a.partials[(size_t)row * NVP + threadIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 63) {   // diagnostics in the two padding columns: ticks to end-of-search / end-of-reduction
        // PCR_ICP_PHASE = phase + 16 * mean: column 31 <- this workgroup's ticks from its entry to the end of phase 1 prologue / 2 certificates /
        // 3 searches / 4 linearisation / 5 row sums (0: to here); gathered as the maximum over the workgroups, or their mean with + 16
        const int ph = a.dbg_phase & 15;
        const unsigned long long t_sel = ph == 1 ? t_p : ph == 2 ? t_a : ph == 3 ? t_b : ph == 4 ? t_search : ph == 5 ? t_ws - t_entry : wall_clock64() - t_entry;
        __hip_atomic_store(&a.partials[(size_t)row * NVP + 30], (double)t_search, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.partials[(size_t)row * NVP + 31], (double)t_sel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its stores ...
    const unsigned long long t_drain = wall_clock64();
    __syncthreads();                                       // ... before ONE lane signals for the workgroup
    if (threadIdx.x == 0) {
        const unsigned int t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned int)(nb - 1));
        is_last = last;
    }
    __syncthreads();
    if (a.stamps_it && (threadIdx.x & 63) == 0 && launches < ICP_STAMP_LAUNCHES) {
        unsigned long long *w = a.stamps_it + 12 * ((size_t)launches * gridDim.x * (BS / 64) + (size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6));
        w[0] = t_entry; w[1] = t_entry + t_search; w[2] = wall_clock64(); w[3] = 0; w[4] = t_ws; w[5] = t_drain;
    }
    unsigned long long *wl = (a.stamps_it && launches < ICP_STAMP_LAUNCHES) ? a.stamps_it + 12 * ((size_t)launches * gridDim.x * (BS / 64) + (size_t)blockIdx.x * (BS / 64)) : nullptr;
    if (!is_last) return;

    // ---- last workgroup: gather the partial rows with sc0 sc1 loads (coherent at agent scope without invalidating this XCD's
    // L2: the acquire fence that plain loads would need took ~10 us here).  Buffer loads through a raw descriptor are visible to the
    // compiler, which keeps all 20 of a lane in flight behind counted waits: 320 rows (160k source points) in ONE round trip where
    // hand-written 8-load groups took three.
    {
        typedef unsigned int icp_u2 __attribute__((ext_vector_type(2)));
        constexpr int NCH = BS / 32;                                       // chunks of 32 columns; chunk c <- rows c, c + NCH, ...
        const int vcol = threadIdx.x & 31, chunk = threadIdx.x >> 5;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.partials, 0, nb * NVP * 8, 0x00020000);
        double s = 0.0;
        for (int b0 = 0; b0 < nb; b0 += NCH * 20) {
            double v[20];
#pragma unroll
            for (int r = 0; r < 20; r++) {
                const int row = b0 + chunk + NCH * r;
                union { icp_u2 u; double d; } x;
                x.u = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (row < nb ? row : 0) * (NVP * 8) + vcol * 8, 0, 17 /* sc0 | sc1 */);
                v[r] = x.d;
            }
#pragma unroll
            for (int r = 0; r < 20; r++) { const double x = b0 + chunk + NCH * r < nb ? v[r] : 0.0; s = (vcol < NV || (vcol == 31 && a.dbg_phase >= 16)) ? s + x : fmax(s, x); }
        }
        fin[chunk][vcol] = s;
    }
    if (wl && threadIdx.x == 0) wl[6] = wall_clock64();
    __syncthreads();
    if (threadIdx.x < NVP) {
        double s = 0;
#pragma unroll
        for (int c = 0; c < BS / 32; c++) s = (threadIdx.x < NV || (threadIdx.x == 31 && a.dbg_phase >= 16)) ? s + fin[c][threadIdx.x] : fmax(s, fin[c][threadIdx.x]);
        fin[0][threadIdx.x] = s;
        st->sums[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // The end of the iteration on the first WAVEFRONT: every lane runs the same scalar work on the same sums (no extra time), so that the
        // three sine / cosine pairs of the pose update -- two thirds of this tail's dependent instructions when one lane computed them one
        // after the other -- are taken by three lanes at once; lane 0 alone writes the state.
        const bool lead = threadIdx.x == 0;
        const unsigned long long t0k = st->t_start;    // stamped by k_icp_nn (previous kernel)
        if (lead) {
            st->t_dbg[3] += wall_clock64() - t0k;
            st->t_dbg[0] += (unsigned long long)fin[0][30]; st->t_dbg[1] += (unsigned long long)(a.dbg_phase >= 16 ? fin[0][31] / nb : fin[0][31]);
            if (wl) wl[7] = wall_clock64();
        }
        const double *S = fin[0];
        const long long count = (long long)(S[29] + 0.5);
        const double fit = ns > 0 ? (double)count / (double)ns : 0.0;
        const double rmse = count > 0 ? sqrt(S[28] / (double)count) : 0.0;
        const double fit_prev = st->fitness, rmse_prev = st->rmse; const int iter_prev = st->iter;
        bool stop = false, conv = false;
        if (a.single) stop = true;
        else if (launches > 0 && fabs(fit_prev - fit) < a.rel_fit && fabs(rmse_prev - rmse) < a.rel_rmse) { stop = true; conv = true; }
        else if (iter_prev >= a.max_it) stop = true;
        if (lead) {
            st->dfit = launches > 0 ? fabs(fit_prev - fit) : 1e300; st->drmse = launches > 0 ? fabs(rmse_prev - rmse) : 1e300;
            st->fitness = fit; st->rmse = rmse; st->count = count;
        }
        if (!stop && (MODE == ICP_MODE_GICP || MODE == ICP_MODE_GICP_COV)) {
            double U[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
            if (count > 0) {
                double nb6[6], x[6];
#pragma unroll
                for (int p = 0; p < 6; p++) nb6[p] = -S[21 + p];
                bool solved = icp_ldlt6_fast(S, nb6, x);
                if (!solved) {                         // rare: indefinite / semi-definite system (wave-uniform: every lane saw the same sums)
                    double *rhs = &fin[1][0];          // LDS copies (dynamic indexing must not touch scratch)
                    double *sol = &fin[2][0];
                    if (lead) {
                        for (int p = 0; p < 6; p++) rhs[p] = -S[21 + p];
                        sol[6] = icp_ldlt6_pivoted(S, rhs, sol, ldl_w, ldl_perm) ? 1.0 : 0.0;
                    }
                    __builtin_amdgcn_wave_barrier();
                    solved = sol[6] != 0.0;
                    for (int p = 0; p < 6; p++) x[p] = sol[p];
                }
                if (solved) {
                    // lane j < 3 takes angle j; the six values come back by readlane
                    const int lj = threadIdx.x;
                    const double ang = lj == 0 ? x[0] : (lj == 1 ? x[1] : x[2]);
                    union { double d; int i[2]; } sv, cv, t;
                    sv.d = sin(ang); cv.d = cos(ang);
                    auto from = [&](const decltype(sv) &v, int l) { t.i[0] = __builtin_amdgcn_readlane(v.i[0], l); t.i[1] = __builtin_amdgcn_readlane(v.i[1], l); return t.d; };
                    const double sa = from(sv, 0), ca = from(cv, 0), sb = from(sv, 1), cb = from(cv, 1), sg = from(sv, 2), cg = from(cv, 2);
                    U[0] = cg * cb; U[1] = cg * sb * sa - sg * ca; U[2] = cg * sb * ca + sg * sa; U[3] = x[3];
                    U[4] = sg * cb; U[5] = sg * sb * sa + cg * ca; U[6] = sg * sb * ca - cg * sa; U[7] = x[4];
                    U[8] = -sb;     U[9] = cb * sa;                U[10] = cb * ca;               U[11] = x[5];
                }
            }
            if (wl && lead) wl[8] = wall_clock64();
            if (lead) {
                double Tn[16];
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c < 4; c++) {
                        double s = 0;
                        for (int k = 0; k < 4; k++) s += U[r * 4 + k] * (k < 3 ? T[k * 4 + c] : (c == 3 ? 1.0 : 0.0));
                        Tn[r * 4 + c] = s;
                    }
                for (int k = 0; k < 16; k++) st->T[k] = Tn[k];
                st->iter = iter_prev + 1;
            }
        }
        if (lead) {
            st->launches = launches + 1;
            st->converged = conv ? 1 : 0;
            __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st->ns = ns;
            st->t_live += wall_clock64() - t0k;
            if (a.stamps_it && launches < ICP_STAMP_LAUNCHES)
                a.stamps_it[12 * ((size_t)launches * gridDim.x * (BS / 64) + (size_t)blockIdx.x * (BS / 64)) + 3] = wall_clock64();
            st->done = stop ? 1 : 0;          // visible to the next launch through the kernel boundary
        }
        if (a.ms_scales > 0 && stop) {        // (wave-uniform) the scale is over: keep its result; not the last one: the pair goes on by itself
            __builtin_amdgcn_wave_barrier();
            const bool more = a.ms_index + 1 < a.ms_scales;
            const IcpArgs *next = a.ms_args + (a.ms_index + 1); IcpArgs *self = a.ms_self; IcpState *hist = a.ms_hist + a.ms_index;
            if (lead) {
                __threadfence();
                *hist = *st;
                if (more) {
                    st->launches = 0; st->iter = 0; st->converged = 0; st->done = 0; st->fitness = 0; st->rmse = 0; st->count = 0;
                    st->dfit = 1e300; st->drmse = 1e300;
                }
            }
            if (more) {                        // the argument slot, a dword per lane and round (nobody reads it any more in this launch: the other workgroups have left)
                const unsigned *srcw = reinterpret_cast<const unsigned *>(next); unsigned *dstw = reinterpret_cast<unsigned *>(self);
                for (int w = threadIdx.x; w < (int)(sizeof(IcpArgs) / 4); w += 64) dstw[w] = srcw[w];
            }
        }
    }
}

// ---- kernel 2 of an iteration: one correspondence per lane in float64 -> wave/LDS reduction -> last workgroup
// finishes the iteration (sum partials, convergence test, 6x6 solve, pose update).
template <int MODE>
__device__ static inline void d_icp_iter(const IcpArgs &a) {
    IcpState *st = a.state;
    if (st->done) return;
    const int ns = *a.ns_ptr;
    int nb = (ns + LIN_BS - 1) / LIN_BS;
    if (nb < 1) nb = 1;
    if (nb > (int)gridDim.x) nb = gridDim.x;
    if ((int)blockIdx.x >= nb) return;
    const unsigned long long t_entry = wall_clock64();
    const int launches = st->launches;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = st->T[k];
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0.0;
    for (int i = blockIdx.x * LIN_BS + threadIdx.x; i < ns; i += nb * LIN_BS) icp_point<MODE>(a, T, i, ns, a.match[i], acc);
    icp_finish<MODE, LIN_BS>(a, st, T, acc, nb, ns, launches, t_entry, (int)blockIdx.x);
}
template <int MODE> __global__ void __launch_bounds__(LIN_BS) k_icp_iter(IcpArgs a) { d_icp_iter<MODE>(a); }
// group form: blockIdx.y = pair.  A pair uses min(its own tile count, the grid's width) partial rows, which is what the one-pair launch
// uses (its grid is min(tiles, LIN_MAX_BLOCKS) wide and the group's grid is the widest of those): same sums, same order.
template <int MODE> __global__ void __launch_bounds__(LIN_BS) k_icp_iter_g(const IcpArgs *__restrict__ a) { d_icp_iter<MODE>(a[blockIdx.y]); }
// tile form (every GICP linearisation that is not fused into k_icp_fused): workgroup b owns the 512-point tile b, so the partial rows and
// their order are those of k_icp_fused<512>; held at 128 VGPRs = two workgroups per CU (the strided loop of k_icp_iter takes 218 and is
// capped at 128 workgroups: a cold launch over 1.6M points took 0.8-3 ms)
__device__ static inline void d_icp_lin(const IcpArgs &a) {
    IcpState *st = a.state;
    const int done = st->done, launches = st->launches;
    const int ns = *a.ns_ptr;
    const int i = blockIdx.x * LIN_BS + threadIdx.x;
    const int mi = a.match[i < a.src_cap ? i : 0];
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = st->T[k];
    if (done) return;
    int nb = (ns + LIN_BS - 1) / LIN_BS;
    if (nb < 1) nb = 1;
    if ((int)blockIdx.x >= nb) return;
    const unsigned long long t_entry = wall_clock64();
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0.0;
    if (i < ns) icp_point<ICP_MODE_GICP>(a, T, i, ns, mi, acc);
    icp_finish<ICP_MODE_GICP, LIN_BS>(a, st, T, acc, nb, ns, launches, t_entry, (int)blockIdx.x);
}
__global__ void __launch_bounds__(LIN_BS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_icp_lin(IcpArgs a) { d_icp_lin(a); }
__global__ void __launch_bounds__(LIN_BS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_icp_lin_g(const IcpArgs *__restrict__ a) { d_icp_lin(a[blockIdx.y]); }

// ---- ONE kernel per iteration (launches after the first of a scale): workgroup b owns source points [512 b, 512 b + 512):
// certificates per lane -> its pending queries compacted into LDS -> the workgroup's 64 octets search them -> barrier ->
// one correspondence per lane in float64 -> reduction / ticket / last-workgroup finish as in k_icp_iter.  Halves the
// launches of the loop (the in-flight throughput is bound by the dispatch rate of small dependent kernels, ~90k/s
// system-wide) and removes the match[] round trip between the two kernels.
#ifndef FUSED_BS
#define FUSED_BS 512
#endif
#ifdef FUSED_WAVES
#define FUSED_OCC __attribute__((amdgpu_waves_per_eu(FUSED_WAVES, FUSED_WAVES)))
#else
#define FUSED_OCC
#endif
// a wave-uniform double held in SGPRs (the pose: twelve doubles every lane loaded into 24 VGPRs of its own)
__device__ static inline double icp_uniform(double v) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]); u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
}
// PPL = source points per lane (PCR_ICP_PPL, default 1).  The kernel is latency-bound (dependent loads, publish -> ticket -> gather), so
// its wavefronts mostly wait, and at one point per lane a 160k-point launch is 312 workgroups of 8 wavefronts at 122 VGPRs -- 61 % of
// the chip's wavefront slots for ONE pair's iteration.  Two / four points per lane (half / a quarter of the wavefronts, partial rows
// and tickets; a lane's certificate tests and float64 linearisations back to back) were measured: 4 pairs in flight 327 / 318 / 293
// pairs/s for PPL 1 / 2 / 4, 8 in flight 314 / 334 / 313, one pair alone 178 / 158 / 125 -- the wavefront slots are not what bounds the
// pairs in flight ("what bounds the throughput"), and alone the longer lanes cost what they cost.  Kept as a switch.
template <int TILE_PTS, bool GRID>
__device__ static inline void d_icp_fused(const IcpArgs &a) {
    IcpState *st = a.state;
    constexpr int OPB = FUSED_BS / OCT;
    constexpr int PPL = TILE_PTS >= FUSED_BS ? TILE_PTS / FUSED_BS : 1;      // source points per lane; tiles under 512 points leave lanes without one
    __shared__ OctMeta m;
    __shared__ OctStack<OPB> stk;
    __shared__ float4 rec_q[TILE_PTS];         // pending queries of the workgroup: position + hint
    __shared__ short rec_l[TILE_PTS];          //   and their local point index
    __shared__ int cand_l[TILE_PTS];           // candidate target point per local point after the search phase
    __shared__ int n_rec;
    const int done = st->done, launches = st->launches;
    const int tid = threadIdx.x, lane = tid & 63, oct = lane >> 3, ol = lane & 7, ob = tid >> 3;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (round-robin dispatch), so XCD x takes the CONTIGUOUS tiles [x per, (x + 1) per) of
    // the nb live ones: source tiles are in Morton order, an eighth of them is a compact region, and every XCD's L2 then holds its own
    // part of the target tree instead of all of it (the grid is a multiple of 8 and covers 8 per; rows of the partial sums stay in tile
    // order: same arithmetic)
    const int ns = *a.ns_ptr, nt = *a.nt_ptr;
    const int tile = (a.tile_rt > 0 && a.tile_rt < TILE_PTS) ? a.tile_rt : TILE_PTS;      // (kernel argument: uniform)
    const int nb = (ns + tile - 1) / tile > 0 ? (ns + tile - 1) / tile : 1;
    const int per = (nb + 7) >> 3;
    const int bid = (int)(blockIdx.x >> 3) < per ? (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3) : nb;
    const int tile0 = (bid < nb ? bid : 0) * tile;
    float4 pf[PPL], refv[PPL]; int mraw[PPL], rb[PPL]; int4 lst[PPL];
    IcpPre pre[PPL];                            // phase C's inputs, requested as early as they are known
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const int i = tile0 + p * FUSED_BS + tid;
        const int ic = (i < a.src_cap && p * FUSED_BS + tid < tile) ? i : 0;
        pf[p] = a.src_pts[ic]; mraw[p] = a.match[ic]; refv[p] = a.ref[ic];
        if (GRID) { lst[p] = a.clist[ic]; rb[p] = -1; } else { rb[p] = a.rbest[ic]; lst[p] = make_int4(-1, -1, -1, -1); }
        pre[p].pf = pf[p]; pre[p].sn = a.src_nrm[ic];
    }
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = st->T[k];
    int mword = 0;
    if (tid < (int)(sizeof(OctMeta) / 4)) mword = ((const int *)a.tgt.meta)[tid];
    const unsigned gmask = GRID ? *a.grid.dmask : 0u;
    if (done) return;
    if (bid >= nb) return;
    if (a.ms_scales > 0 && launches == 0) {       // first launch of a scale in the all-scales loop: what match / ref / clist hold belongs to the scale before
#pragma unroll
        for (int p = 0; p < PPL; p++) { mraw[p] = -1; refv[p].w = 0.0f; lst[p] = make_int4(-1, -1, -1, -1); }
    }
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = icp_uniform(T[k]);
    const unsigned long long t_entry = wall_clock64();
    if (bid == 0 && tid == 0) st->t_start = t_entry;
    if (tid < (int)(sizeof(OctMeta) / 4)) ((int *)&m)[tid] = mword;
    if (tid == 0) n_rec = 0;
    __syncthreads();
    const unsigned long long t_p = wall_clock64() - t_entry;
    // ---- phase A: per point -- certificate or a place in the pending list.  The listed points of ALL the lane's queries are requested first
    // (one round trip for the lane), and the moment a neighbour is decided its normal is requested for phase C.
    float4 ltp[PPL][PCR_NN_K];
    if (GRID) {
#pragma unroll
        for (int p = 0; p < PPL; p++) {
            // (records beyond the cloud's count are never written: their indices are whatever the arena held)
            if (!(p * FUSED_BS + tid < tile && tile0 + p * FUSED_BS + tid < ns)) lst[p] = make_int4(-1, -1, -1, -1);
            icp_list_load(a.tgt_pts, lst[p], ltp[p]);
        }
    }
    int pre_c[PPL];
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const int i = tile0 + p * FUSED_BS + tid;
        const bool mine = p * FUSED_BS + tid < tile;
        bool need = false;
        int cand = -1;
        float qx = 0, qy = 0, qz = 0; int hint = -1;
        pre_c[p] = -1;
        if (mine && i < ns) {
            const double px = pf[p].x, py = pf[p].y, pz = pf[p].z;
            qx = (float)(T[0] * px + T[1] * py + T[2] * pz + T[3]);
            qy = (float)(T[4] * px + T[5] * py + T[6] * pz + T[7]);
            qz = (float)(T[8] * px + T[9] * py + T[10] * pz + T[11]);
            hint = mraw[p] >= 0 ? mraw[p] : (mraw[p] <= -2 ? -(mraw[p] + 2) : -1);
            const float ex = qx - refv[p].x, ey = qy - refv[p].y, ez = qz - refv[p].z;
            bool certified = refv[p].w > 0.0f;
            if (GRID) { if (certified) certified = icp_list_decide(qx, qy, qz, refv[p], lst[p], ltp[p], a.r2f, &rb[p], &pre[p].tf) != 0; }
            else certified = certified && pcr_d2(ex, ey, ez) < refv[p].w * refv[p].w;
            need = nt > 0 && !certified;
            cand = certified ? rb[p] : -1;                 // rb < 0: certified unmatched
            if (GRID && certified && rb[p] < 0 && mraw[p] >= 0) a.match[i] = -1;      // decided unmatched after having been matched
            if (cand >= 0) {
                if (!GRID) pre[p].tf = a.tgt_pts[cand];
                pre[p].tn = a.tgt_nrm[cand]; pre_c[p] = cand;
            }
        }
        const unsigned long long nbm = __ballot(need);
        int base = 0;
        if (lane == 0 && nbm != 0ull) base = atomicAdd(&n_rec, __builtin_popcountll(nbm));
        base = __builtin_amdgcn_readfirstlane(base);            // (lane 0 holds it; no ds_bpermute round trip)
        if (need) {
            const int slot = base + __builtin_popcountll(nbm & ((1ull << lane) - 1ull));
            rec_q[slot] = make_float4(qx, qy, qz, __int_as_float(hint)); rec_l[slot] = (short)(p * FUSED_BS + tid);
        }
        if (mine) cand_l[p * FUSED_BS + tid] = cand;
    }
    __syncthreads();
    const unsigned long long t_a = wall_clock64() - t_entry;
    // ---- phase B: the pending list, one query per octet, 64 per round (one query per LANE through the cell hash was measured: divergent
    // per-lane scans, 47 us against 27 us per launch at the coarsest scale)
    const int npend = n_rec;
    if (tid == 0 && npend > 0) atomicAdd(&st->searched, (unsigned long long)npend);
    // a handful of pending queries (the usual case): a WAVEFRONT per query -- all eight cells of its block at once, two round trips
    const bool few = GRID && npend <= 2 * (FUSED_BS / 64);
    if (GRID && few) {
        for (int e = tid >> 6; e < npend; e += FUSED_BS / 64) {
            const float4 r = rec_q[e]; const int l = rec_l[e];
            float d1 = 0, d2 = 0; int nnk[PCR_NN_K];
            grid_nn_query64<PCR_NN_K>(a.grid, gmask, a.tgt_pts, r.x, r.y, r.z, a.r2s, lane, nnk, &d1, &d2);
            if (lane == 0) {
                const int qi = tile0 + l;
                a.ref[qi] = make_float4(r.x, r.y, r.z, sqrtf(d2));
                a.clist[qi] = make_int4(nnk[0], nnk[1], nnk[2], nnk[3]);
                if (nnk[0] < 0) { const int hint = __float_as_int(r.w); a.match[qi] = -((hint >= 0 ? hint : 0) + 2); }
                cand_l[l] = nnk[0];
            }
        }
    }
    for (int e0 = 0; e0 < (few ? 0 : npend); e0 += OPB) {
        const int e = e0 + ob;
        const bool live = e < npend;
        if (__ballot(live) == 0ull) continue;
        float qx = 0, qy = 0, qz = 0; int hint = -1, l = 0;
        if (live) { const float4 r = rec_q[e]; qx = r.x; qy = r.y; qz = r.z; hint = __float_as_int(r.w); l = rec_l[e]; }
        int start_pt = 0; float d1 = 0, d2 = 0;
        int best; int nnk[PCR_NN_K];
        if (GRID) { grid_nn_query8<PCR_NN_K>(a.grid, gmask, a.tgt_pts, live, qx, qy, qz, a.r2s, ol, nnk, &d1, &d2); best = nnk[0]; start_pt = hint >= 0 ? hint : 0; }
        else best = oct_nn_query<OPB>(a.tgt, m, stk, live, qx, qy, qz, a.r2s, hint, ol, oct, ob, &start_pt, nullptr, &d1, &d2);
        if (ol == 0 && live) {
            const int qi = tile0 + l;
            if (GRID) {
                a.ref[qi] = make_float4(qx, qy, qz, sqrtf(d2));
                a.clist[qi] = make_int4(nnk[0], nnk[1], nnk[2], nnk[3]);
            } else {
                const float slack = 2e-5f + 1e-6f * (fabsf(qx) + fabsf(qy) + fabsf(qz));
                const float margin = best >= 0 ? 0.5f * (sqrtf(d2) - sqrtf(d1)) - slack : a.rs_minus_r - slack;
                a.ref[qi] = make_float4(qx, qy, qz, margin > 0.0f ? margin : 0.0f);
                a.rbest[qi] = best;
            }
            if (best < 0) a.match[qi] = -(start_pt + 2);       // next launch's start hint
            cand_l[l] = best;
        }
    }
    __syncthreads();
    const unsigned long long t_b = wall_clock64() - t_entry;
    // ---- phase C: the lane's correspondences one after the other (icp_point stores the match, or the hint when the radius test fails)
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0.0;
#pragma unroll
    for (int p = 0; p < PPL; p++) {
        const int i = tile0 + p * FUSED_BS + tid;
        if (p * FUSED_BS + tid < tile && i < ns) {
            const int c = cand_l[p * FUSED_BS + tid];
            if (c >= 0 && c != mraw[p]) a.match[i] = c;
            if (c >= 0 && c != pre_c[p]) { pre[p].tf = a.tgt_pts[c]; pre[p].tn = a.tgt_nrm[c]; }      // searched in phase B
            icp_point<ICP_MODE_GICP>(a, T, i, ns, c, acc, &pre[p]);
        }
    }
    icp_finish<ICP_MODE_GICP, FUSED_BS>(a, st, T, acc, nb, ns, launches, t_entry, bid, t_p, t_a, t_b);
}
template <int TILE_PTS, bool GRID> __global__ void __launch_bounds__(FUSED_BS) FUSED_OCC k_icp_fused(IcpArgs a) { d_icp_fused<TILE_PTS, GRID>(a); }
// (the problem's arguments are copied out of the device buffer ONCE, by scalar loads, like by-value kernel arguments: read through the
// pointer wherever they are used they sat in VGPRs -- 146-162 instead of 122, one workgroup per CU instead of two)
template <int TILE_PTS, bool GRID> __global__ void __launch_bounds__(FUSED_BS) FUSED_OCC k_icp_fused_g(const IcpArgs *__restrict__ a) { d_icp_fused<TILE_PTS, GRID>(a[blockIdx.y]); }
// groups of up to ICP_BYVAL problems: the argument structs travel BY VALUE in the kernel arguments (scalar loads on demand, 123 VGPRs = two
// workgroups per CU; read through a device pointer the fields sit in VGPRs: 147-150, one workgroup per CU)
#define ICP_BYVAL 8
struct IcpArgsB { IcpArgs a[ICP_BYVAL]; };
template <int TILE_PTS, bool GRID> __global__ void __launch_bounds__(FUSED_BS) FUSED_OCC k_icp_fused_b(IcpArgsB b) { d_icp_fused<TILE_PTS, GRID>(b.a[blockIdx.y]); }

// Source points per workgroup of the fused kernel (PCR_ICP_TILE = 128 ... 2048; PCR_ICP_PPL = 1 / 2 / 4 is the older spelling of
// 512 / 1024 / 2048).  The workgroup's 64 octets serve its pending queries 64 at a time, so a 512-point tile whose certificates do not
// hold walks the tree up to 8 times in a row -- but that is NOT what the launch waits for: one pair alone, 25 iterations per scale,
// 43.6 / 40.7 / 46.0 us per launch for tiles of 512 / 256 / 128 points (the slowest single walk is); lockstep groups of 2, four in
// flight: 333 / 304 / 258 pairs/s for 1024 / 512 / 256.  One pair: 256 from 40k points (below: 512, the tile of k_icp_iter, so that
// PCR_ICP_FUSED=0 stays the same arithmetic on NCLT-size clouds); groups: 1024.
static int fused_tile_points(const pcr_context *ctx, int cap, int G) {
    static const int fixed = getenv("PCR_ICP_TILE") ? atoi(getenv("PCR_ICP_TILE")) : (getenv("PCR_ICP_PPL") ? FUSED_BS * atoi(getenv("PCR_ICP_PPL")) : 0);
    int t = fixed;
    if (t <= 0) {
        // group_forms: a unit of a lockstep-group plan, whatever its size.  One pair: 256 from 40k points (the slowest workgroup's search sets the
        // launch), 1024 from 400k (round 5, config 5's 0.5-1.6M-point scales: the launch is several rounds of workgroups and fewer, fatter ones win:
        // 568 against 617 us per launch in flight, 26.0 against 24.9 pairs/s; 2048: 603 / 25.4)
        t = (G > 1 || ctx->group_forms) ? 1024 : (cap >= 400000 ? 1024 : (cap >= 40000 ? 256 : 512));
    }
    t = t >= 2048 ? 2048 : (t >= 1024 ? 1024 : (t >= 512 ? 512 : 256));
    // the last workgroup of the fused kernel gathers one partial row per tile: at most 4096 of them (2M-point clouds of config 5: tiles of 512)
    while (t < 2048 && (cap + t - 1) / t > 4088) t *= 2;
    return t;
}
// Tile of a pair in the GROUP forms (round 5): 512 points up to 40k source points -- the NCLT scans: five scales of 5-20k points, where a launch is one
// round of a few workgroups and halving their work shortens it (script-2 stage on the shipped scans, groups of 8 x 4: 1590-1780 -> 1900-2020 pairs/s) --
// else the kernel's 1024 (50k points: 1892 against 1841 with 512; 100k: 1197 against 1164; 200k: 698 against 672).  By the PAIR alone.
static int icp_group_tile(int cap, int kernel_tile) {
    static const bool fixed = getenv("PCR_ICP_TILE") || getenv("PCR_ICP_PPL");
    if (fixed || kernel_tile != 1024) return 0;
    return cap < 40000 ? 512 : 0;
}
#define PCR_FUSED_LAUNCH_(ctx, KERNEL, GRID, tile, grid, arg)                                                          \
    do {                                                                                                               \
        switch (tile) {                                                                                                \
            case 256: PCR_LAUNCH(ctx, (KERNEL<256, GRID>), grid, dim3(FUSED_BS), 0, (ctx)->stream, arg); break;        \
            case 1024: PCR_LAUNCH(ctx, (KERNEL<1024, GRID>), grid, dim3(FUSED_BS), 0, (ctx)->stream, arg); break;      \
            case 2048: PCR_LAUNCH(ctx, (KERNEL<2048, GRID>), grid, dim3(FUSED_BS), 0, (ctx)->stream, arg); break;      \
            default: PCR_LAUNCH(ctx, (KERNEL<512, GRID>), grid, dim3(FUSED_BS), 0, (ctx)->stream, arg); break;         \
        }                                                                                                              \
    } while (0)
#define PCR_FUSED_LAUNCH(ctx, KERNEL, use_grid, tile, grid, arg)                                                       \
    do { if (use_grid) PCR_FUSED_LAUNCH_(ctx, KERNEL, true, tile, grid, arg); else PCR_FUSED_LAUNCH_(ctx, KERNEL, false, tile, grid, arg); } while (0)

// PCR_ICP_GRID=0: every correspondence search over the octree (the independent check of tests/test_gpu_gicp.py::test_switches_do_not_change_the_result)
static bool icp_use_grid() { static const bool on = !(getenv("PCR_ICP_GRID") && atoi(getenv("PCR_ICP_GRID")) == 0); return on; }
// The search cap of the certificate mode is r + g.  With the cell hash the cap costs nothing up to what the level's cells cover (edge >= 2.04 cap,
// pcr_grid_level_for), and the room of a certificate is bounded by it: a query with fewer than K + 1 target points inside the cap -- every
// unmatched one: 45 % of the finest scale -- carries D = cap, i.e. g of room, and with g = 2.5 cm it was searched again every few launches.
// So once the level is known, g grows to what its cells allow (at most r itself).
static double icp_gap_for_level(const DevCloud *tgt, int L, double max_dist, double g) {
    if (L < 0) return g;
    const double cap_max = 0.49 * (double)tgt->key_unit[0] * (double)(1 << L) * (1.0 - 2e-3);
    double gl = cap_max - max_dist;
    if (gl > max_dist) gl = max_dist;
    return gl > g ? gl : g;
}
static void fill_args(IcpArgs &a, const DevCloud *src, const DevCloud *tgt, double max_dist, const pcr_gicp_params *p,
                      int32_t *match, IcpState *st, double *partials, int single) {
    a.src_pts = src->pts; a.src_nrm = src->nrm; a.ns_ptr = src->n;
    a.src_cov6 = src->cov6; a.tgt_cov6 = tgt->cov6;
    a.tgt_pts = tgt->pts; a.tgt_nrm = tgt->nrm; a.nt_ptr = tgt->n;
    a.tgt.pts = tgt->pts; a.tgt.nodes = tgt->oct_nodes; a.tgt.up = tgt->oct_up; a.tgt.meta = tgt->oct_meta; a.tgt.leaf_of = tgt->leaf_of; a.tgt.keys = tgt->keys; a.tgt.pinfo = tgt->pinfo; a.tgt.l1rng = tgt->oct_l1;
    a.match = match; a.src_cap = src->cap > 0 ? src->cap : 1; a.state = st; a.partials = partials;
    a.max_dist2 = max_dist * max_dist;
    // The float32 search works on float32-rounded query positions (ulp 8e-6 m at 100 m), the radius itself is tested in
    // float64 by k_icp_iter on the unrounded position: the search cap is widened by more than that rounding so that the
    // float64 test alone decides (a cap of exactly r dropped ~1 in 1e5 borderline matches the oracle keeps).
    const double rw = max_dist * (1.0 + 1e-4) + 3e-4;
    const double r2w = max_dist < 1e18 ? rw * rw : 1e300;
    a.r2f = r2w < 3.0e38 ? (float)r2w : 3.0e38f;
    a.loss = p ? p->loss : 0; a.loss_k = p ? p->loss_k : 1.0; a.a = 1.0 - (p ? p->epsilon : 1e-3);
    a.rel_fit = p ? p->relative_fitness : 1e-6; a.rel_rmse = p ? p->relative_rmse : 1e-6; a.max_it = p ? p->max_iteration : 30;
    a.single = single;
    a.dbg_visits = (single && pcr_options().debug_visits.load(std::memory_order_relaxed)) ? 1 : 0;      // (diagnostic switches: pcr_set_option, latched from the environment once)
    a.dbg_phase = pcr_options().icp_phase.load(std::memory_order_relaxed);
    a.stamps_nn = nullptr; a.stamps_it = nullptr;
    a.ref = nullptr; a.rbest = nullptr; a.clist = nullptr;
    memset(&a.grid, 0, sizeof a.grid); a.grid.L = -1;
}

static int read_state(pcr_context *ctx, const IcpState *st_dev, IcpState *host) {
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(ctx->pinned, st_dev, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host, ctx->pinned, sizeof(IcpState));
    return PCR_OK;
}

static void state_to_result(const IcpState &s, pcr_result *out) {
    for (int k = 0; k < 16; k++) out->transformation[k] = s.T[k];
    out->fitness = s.fitness; out->inlier_rmse = s.rmse; out->n_correspondences = s.count;
    out->iterations = s.iter; out->converged = s.converged;
}

// Length of the next chunk of launches.  A chunk is queued before the state of the one before it has come back, so after convergence the
// rest of the current chunk and all of the next are launches that return at once (4.4 us each: with chunks of 8 a third of all launches of a
// pair run alone).  Once the last changes of fitness and RMSE are within a small factor of the criteria the loop is about to stop: short chunks.
// Measured (48 pairs of 200k points): lockstep groups, four in flight: 26 % -> 13 % of the launches return at once, 692 -> 695 pairs/s (nothing:
// the other groups' kernels fill those slots either way); ONE pair at a time: 34 % -> 26 %, but 244 -> 236 pairs/s -- every chunk boundary is a
// ~10 us gap on the one stream there is, and short chunks have more of them than they save in 4.4-us launches.  So: on for groups, off
// for the one-pair loop (PCR_ICP_CHUNK_ADAPT = 0 / 1 forces either).
static int icp_next_chunk(const IcpState &s, double rel_fit, double rel_rmse, int chunk, bool group) {
    static const int forced = getenv("PCR_ICP_CHUNK_ADAPT") ? atoi(getenv("PCR_ICP_CHUNK_ADAPT")) : -1;
    const bool adaptive = forced < 0 ? group : forced != 0;
    if (!adaptive || s.launches < 2 || !(rel_rmse > 0.0)) return chunk;
    const double quantum = s.ns > 0 ? 1.0 / (double)s.ns : 0.0;           // the fitness moves in steps of one correspondence
    const double fit_tol = rel_fit > quantum ? rel_fit : quantum;
    if (s.dfit <= 2.5 * fit_tol && s.drmse <= 30.0 * rel_rmse) return chunk < 2 ? chunk : 2;
    if (s.dfit <= 8.5 * fit_tol && s.drmse <= 300.0 * rel_rmse) return chunk < 4 ? chunk : 4;
    return chunk;
}

int pcr_dev_gicp(pcr_context *ctx, const DevCloud *src, const DevCloud *tgt, double max_dist, const double *T0,
                 const pcr_gicp_params *p, pcr_result *out, int32_t *match_dev) {
    if (!(max_dist > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    const bool use_cov = src->cov6 && tgt->cov6;
    if (!use_cov && (!src->nrm || !tgt->nrm)) { ctx->err = "GICP needs normals (or covariances) on both clouds"; return PCR_EINVAL; }
    ArenaMark mark(ctx);
    const int cap = src->cap > 0 ? src->cap : 1;
    static const bool use_skip = !(getenv("PCR_ICP_SKIP") && atoi(getenv("PCR_ICP_SKIP")) == 0);
    const int nblin = (cap + LIN_BS - 1) / LIN_BS;                  // k_icp_lin: one tile per workgroup
    const int nbmax = nblin < LIN_MAX_BLOCKS ? nblin : LIN_MAX_BLOCKS;
    const int nbnn = (cap + ICP_BS / OCT - 1) / (ICP_BS / OCT);
    const int tile_pts = fused_tile_points(ctx, cap, 1);
    const int tile_eff = (ctx->group_forms && icp_group_tile(cap, tile_pts)) ? icp_group_tile(cap, tile_pts) : tile_pts;
    const int nbf = ((cap + tile_eff - 1) / tile_eff + 7) & ~7;      // workgroups of the fused kernel: one per tile of source points, a multiple of 8 (XCD order)
    IcpState *st = arena<IcpState>(ctx, 1);
    const int rows = nblin > nbf ? nblin : nbf;
    double *partials = arena<double>(ctx, (size_t)rows * NVP);
    int32_t *match = match_dev ? match_dev : arena<int32_t>(ctx, cap);
    if (!st || !partials || !match) return PCR_ENOMEM;
    IcpArgs a; memset(&a, 0, sizeof a); fill_args(a, src, tgt, max_dist, p, match, st, partials, 0);
    a.tile_rt = ctx->group_forms ? icp_group_tile(cap, tile_pts) : 0;
    if (use_skip && max_dist < 1e15) {
        a.ref = arena<float4>(ctx, cap); a.rbest = arena<int32_t>(ctx, cap); a.clist = arena<int4>(ctx, cap);
        if (!a.ref || !a.rbest || !a.clist) return PCR_ENOMEM;
        // certificate mode searches a slightly larger ball (r + g): a query with nothing inside it stays unmatched, without
        // a search, until it has moved by g; candidates between r and r + g are rejected by k_icp_iter's float64 radius test
        static const double gfrac = getenv("PCR_ICP_GAP") ? atof(getenv("PCR_ICP_GAP")) : 0.25;
        double g = gfrac * max_dist; g = g < 0.01 ? 0.01 : (g > 0.05 ? 0.05 : g);
        const double rs = max_dist + g;
        a.r2s = (float)(rs * rs * (1.0 + 1e-6)); a.rs_minus_r = (float)g;
        a.verify = pcr_options().icp_verify.load(std::memory_order_relaxed) ? 1 : 0;
    }
    if (icp_use_grid() && !use_cov) {       // cell hash of the target for radii of a few voxels (pcr_octree.h GridView); the octree serves the others
        const int L = pcr_grid_level_for(tgt, std::sqrt((double)(a.ref ? a.r2s : a.r2f)));
        PCR_TRY(pcr_dev_build_grid_batch(ctx, &tgt, &L, 1, &a.grid));
        if (a.ref && a.grid.tab) {
            const double rs = max_dist + icp_gap_for_level(tgt, L, max_dist, (double)a.rs_minus_r);
            a.r2s = (float)(rs * rs * (1.0 + 1e-6)); a.rs_minus_r = (float)(rs - max_dist);
        }
    }
    const bool grid = a.grid.tab != nullptr;
    IcpInit in; memcpy(in.T, T0, sizeof in.T);
    PCR_LAUNCH(ctx, k_icp_init, dim3(1), dim3(64), 0, ctx->stream, st, in);
    static const char *const stamp_path = getenv("PCR_ICP_STAMPS");          // diagnostics only (latched once)
    const size_t sw_nn = (size_t)ICP_STAMP_LAUNCHES * nbnn * (ICP_BS / 64) * 2, sw_it = (size_t)ICP_STAMP_LAUNCHES * nbmax * (LIN_BS / 64) * 12;
    if (stamp_path) {
        if (hipMalloc(&a.stamps_nn, sw_nn * 8) != hipSuccess || hipMalloc(&a.stamps_it, sw_it * 8) != hipSuccess) return PCR_ENOMEM;
        PCR_HIP_CHECK(ctx, hipMemsetAsync(a.stamps_nn, 0, sw_nn * 8, ctx->stream)); PCR_HIP_CHECK(ctx, hipMemsetAsync(a.stamps_it, 0, sw_it * 8, ctx->stream));
    }

    // Launch in chunks; the state of chunk c is copied back while chunk c+1 is already queued, so the GPU never
    // idles on the host.  Launches after 'done' return at their first instruction.
    static const int chunk_env = getenv("PCR_ICP_CHUNK") ? atoi(getenv("PCR_ICP_CHUNK")) : 8;
    const int total = a.max_it + 1, CHUNK = chunk_env < 1 ? 1 : (chunk_env > 32 ? 32 : chunk_env);
    // One chunk = CHUNK x (k_icp_nn, k_icp_iter) replayed as ONE hipGraph launch: the loop is launch-bound (a 3000-point
    // pair still takes 5 ms), and a graph costs one runtime call instead of 16.  The arena hands out the same addresses
    // for the same problem sizes, so the instantiated graph is cached in the context under its argument bytes.
    static const bool use_graph = !(getenv("PCR_ICP_GRAPH") && atoi(getenv("PCR_ICP_GRAPH")) == 0);
    static const bool use_fused = !(getenv("PCR_ICP_FUSED") && atoi(getenv("PCR_ICP_FUSED")) == 0);
    // launch 0 of a scale searches every query (cold): two kernels at full occupancy; later launches: the fused kernel
    // (round 5: also for the 0.5-1.6M-point scales of config 5, which ran three streaming kernels per iteration -- certificates, one pending
    // list for the cloud, tile linearisation -- while 17 % of their queries were searched again per launch; with list certificates it is 3 %
    // and the one kernel is the faster form there too: 167 against 184 us per launch, 26.7 against 26.4 pairs/s.  The streaming kernels are gone.)
    const bool fused = use_fused && a.ref && !use_cov && !stamp_path && nbf <= 4096;
    auto enqueue = [&](int launch_index) {
        if (fused && launch_index > 0) {
            PCR_FUSED_LAUNCH(ctx, k_icp_fused, grid, tile_pts, dim3(nbf), a);
            return;
        }
        if (grid) PCR_LAUNCH(ctx, k_icp_nn<true>, dim3(nbnn), dim3(ICP_BS), 0, ctx->stream, a);
        else PCR_LAUNCH(ctx, k_icp_nn<false>, dim3(nbnn), dim3(ICP_BS), 0, ctx->stream, a);
        if (use_cov) PCR_LAUNCH(ctx, k_icp_iter<ICP_MODE_GICP_COV>, dim3(nbmax), dim3(LIN_BS), 0, ctx->stream, a);
        else if (stamp_path) PCR_LAUNCH(ctx, k_icp_iter<ICP_MODE_GICP>, dim3(nbmax), dim3(LIN_BS), 0, ctx->stream, a);
        else PCR_LAUNCH(ctx, k_icp_lin, dim3(nblin), dim3(LIN_BS), 0, ctx->stream, a);
    };
    // graph of a chunk of `len` launches: which = 0 starts with launch 0 (cold search + linearisation), which = 1 holds later launches only
    auto graph_for = [&](int which, int len, hipGraphExec_t *out) -> int {
        *out = nullptr;
        if (!use_graph || stamp_path) return PCR_OK;
        std::string key((const char *)&a, sizeof a);
        const int extra[7] = {nbnn, nbmax, use_cov ? 1 : 0, len, fused ? nbf + (tile_pts << 16) : 0, which, grid ? 1 : 0};
        key.append((const char *)extra, sizeof extra);
        for (auto &g : ctx->icp_graphs) if (g.key == key) { *out = g.exec; return PCR_OK; }
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        PCR_HIP_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < len; k++) enqueue(which == 0 ? k : len + k);
        PCR_HIP_CHECK(ctx, hipStreamEndCapture(ctx->stream, &graph));
        PCR_HIP_CHECK(ctx, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        if (ctx->icp_graphs.size() >= 48) {           // evict the oldest entry (the stream is drained first: a replay of it may still be queued)
            PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipGraphExecDestroy(ctx->icp_graphs[0].exec);
            if (ctx->icp_graphs[0].graph) (void)hipGraphDestroy(ctx->icp_graphs[0].graph);
            ctx->icp_graphs.erase(ctx->icp_graphs.begin());
        }
        { IcpGraph e; e.key = std::move(key); e.exec = exec; ctx->icp_graphs.push_back(std::move(e)); }
        *out = exec;
        return PCR_OK;
    };
    IcpState *slots = (IcpState *)ctx->pinned;      // two read-back slots
    int launched = 0, cur = 0, prev = -1, n_chunks = 0, next_len = CHUNK;
    std::vector<int> chunk_first, chunk_last;
    IcpState fin; bool have = false;
    for (;;) {
        const bool enq = launched < total;
        if (enq) {
            const int want = launched == 0 ? CHUNK : next_len;
            const int c = total - launched < want ? total - launched : want;
            if (ctx->profiling) {
                while ((int)ctx->prof_events.size() < 2 * (n_chunks + 1)) { hipEvent_t e; PCR_HIP_CHECK(ctx, hipEventCreate(&e)); ctx->prof_events.push_back(e); }
                PCR_HIP_CHECK(ctx, hipEventRecord(ctx->prof_events[2 * n_chunks], ctx->stream));
            }
            hipGraphExec_t ge = nullptr;
            if (c == want) PCR_TRY(graph_for(launched == 0 ? 0 : 1, c, &ge));      // (a ragged last chunk before max_iteration is launched directly)
            if (ge) PCR_HIP_CHECK(ctx, hipGraphLaunch(ge, ctx->stream));
            else for (int k = 0; k < c; k++) enqueue(launched + k);
            if (ctx->profiling) PCR_HIP_CHECK(ctx, hipEventRecord(ctx->prof_events[2 * n_chunks + 1], ctx->stream));
            chunk_first.push_back(launched); chunk_last.push_back(launched + c);
            n_chunks++;
            launched += c;
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(&slots[cur], st, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[cur], ctx->stream));
        }
        if (prev >= 0) {
            PCR_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev[prev]));
            if (slots[prev].done) { fin = slots[prev]; have = true; break; }
            next_len = icp_next_chunk(slots[prev], a.rel_fit, a.rel_rmse, CHUNK, false);
        }
        if (!enq) break;
        prev = cur; cur ^= 1;
    }
    if (!have) { ctx->err = "GICP loop ended without a final state"; return PCR_EHIP; }
    // drain the (no-op) tail so the pinned slots and the arena can be reused safely
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->profiling) {
        // chunks whose launches were all live (before 'done'): HIP-event time / launches = launch-to-launch period
        for (int c = 0; c < n_chunks; c++) {
            const int first = chunk_first[c], last = chunk_last[c];
            if (last <= fin.launches) {
                float ms = 0;
                PCR_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->prof_events[2 * c], ctx->prof_events[2 * c + 1]));
                ctx->prof[0] += ms; ctx->prof[1] += last - first;
            }
        }
        ctx->prof[2] += (double)fin.t_live * 0.01;            // 100 MHz ticks -> microseconds
        ctx->prof[3] += fin.launches;
        ctx->prof[4] += 48.0 * (double)fin.ns * (double)fin.launches;   // 48 B per source point per launch
        ctx->prof[5] += launched; ctx->prof[13] += fin.launches;          // issued / live launches of the loop: the rest returned at once
        ctx->prof[6] += (double)fin.t_dbg[0] * 0.01; ctx->prof[7] += (double)fin.t_dbg[3] * 0.01; ctx->prof[14] += (double)fin.t_dbg[1] * 0.01;
        ctx->prof[11] += (double)fin.searched;
        if (pcr_options().debug_stamps.load(std::memory_order_relaxed)) fprintf(stderr, "icp stamps (us/launch): slowest-wg search %.1f slowest-wg reduce %.1f (unused %.1f) sums-done %.1f end %.1f (launches %d ns %d)\n", fin.t_dbg[0] * 0.01 / fin.launches, fin.t_dbg[1] * 0.01 / fin.launches, fin.t_dbg[2] * 0.01 / fin.launches, fin.t_dbg[3] * 0.01 / fin.launches, fin.t_live * 0.01 / fin.launches, fin.launches, fin.ns);
    }
    if (stamp_path) {
        unsigned long long *h = (unsigned long long *)malloc((sw_nn + sw_it) * 8);
        (void)hipMemcpy(h, a.stamps_nn, sw_nn * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(h + sw_nn, a.stamps_it, sw_it * 8, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(stamp_path, "ab")) {
            const unsigned long long hdr[6] = {0x49435053ull, (unsigned long long)nbnn * (ICP_BS / 64), (unsigned long long)nbmax * (LIN_BS / 64), ICP_STAMP_LAUNCHES, (unsigned long long)fin.launches, (unsigned long long)fin.ns};
            fwrite(hdr, 8, 6, f); fwrite(h, 8, sw_nn + sw_it, f); fclose(f);
        }
        free(h); (void)hipFree(a.stamps_nn); (void)hipFree(a.stamps_it);
    }
    state_to_result(fin, out);
    for (int k = 0; k < 16; k++) if (!std::isfinite(fin.T[k])) { ctx->err = "non-finite pose"; return PCR_ENUMERIC; }
    return PCR_OK;
}

// ---- the GICP loops of a GROUP of pairs in lockstep: every launch serves all G problems (blockIdx.y = pair; argument structs and
// states in device memory), a problem that has converged returns at its first instruction, the group ends when all have.  One chain of
// dependent kernels per GROUP instead of one per pair: the kernels are G times fatter, the device retires small dependent kernels at
// a fixed rate whatever feeds it, and the replayed graph reads its arguments from a fixed device buffer, so it
// is captured once per (G, grid) and never again.  Per-problem arithmetic, tiles and summation order are those of pcr_dev_gicp:
// the results are bit-identical to registering the pairs one by one WITH THE SAME KERNEL FORMS (ctx->group_forms: 1024-point tiles, the
// wavefront k-NN kernel in the preprocessing) -- which is how pcr_register_pairs_plan runs every unit of a plan with group > 1.
int pcr_dev_gicp_group(pcr_context *ctx, int G, const DevCloud *const *src, const DevCloud *const *tgt, const double *max_dists, const double *T0,
                       const pcr_gicp_params *p, pcr_result *out, int32_t *const *match_dev) {
    for (int g = 0; g < G; g++) if (!(max_dists[g] > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    if (G < 1 || G > 32) { ctx->err = "GICP group size must be in 1..32"; return PCR_EINVAL; }
    for (int g = 0; g < G; g++) if (!src[g]->nrm || !tgt[g]->nrm) { ctx->err = "GICP needs normals on both clouds"; return PCR_EINVAL; }
    ArenaMark mark(ctx);
    // two source points per lane in lockstep groups (G x the workgroups per launch: half of them leave more of the chip to the other
    // groups in flight: 200k points, groups of 2, 4 groups in flight 367 -> 408 pairs/s); PCR_ICP_PPL overrides
    int max_cap = 1;
    for (int g = 0; g < G; g++) max_cap = src[g]->cap > max_cap ? src[g]->cap : max_cap;
    const int tile_pts = fused_tile_points(ctx, max_cap, G);
    int nbmax = 1, nbnn = 1, nbf = 1;
    std::vector<IcpArgs> args((size_t)G); std::vector<IcpInit> inits((size_t)G);
    IcpState *st = arena<IcpState>(ctx, G);
    if (!st) return PCR_ENOMEM;
    static const double gfrac = getenv("PCR_ICP_GAP") ? atof(getenv("PCR_ICP_GAP")) : 0.25;
    for (int g = 0; g < G; g++) {
        const int cap = src[g]->cap > 0 ? src[g]->cap : 1;
        const int m_ = (cap + LIN_BS - 1) / LIN_BS;                      // k_icp_lin_g: one 512-point tile per workgroup
        const int tile_g = icp_group_tile(cap, tile_pts) ? icp_group_tile(cap, tile_pts) : tile_pts;
        const int n_ = (cap + ICP_BS / OCT - 1) / (ICP_BS / OCT), f_ = ((cap + tile_g - 1) / tile_g + 7) & ~7;
        nbmax = m_ > nbmax ? m_ : nbmax; nbnn = n_ > nbnn ? n_ : nbnn; nbf = f_ > nbf ? f_ : nbf;
        double *partials = arena<double>(ctx, (size_t)(m_ > f_ ? m_ : f_) * NVP);
        int32_t *match = (match_dev && match_dev[g]) ? match_dev[g] : arena<int32_t>(ctx, cap);
        if (!partials || !match) return PCR_ENOMEM;
        IcpArgs &a = args[g]; memset(&a, 0, sizeof a);
        const double max_dist = max_dists[g];
        fill_args(a, src[g], tgt[g], max_dist, p, match, st + g, partials, 0);
        a.tile_rt = icp_group_tile(cap, tile_pts);
        a.ref = arena<float4>(ctx, cap); a.rbest = arena<int32_t>(ctx, cap); a.clist = arena<int4>(ctx, cap);
        if (!a.ref || !a.rbest || !a.clist) return PCR_ENOMEM;
        double gg = gfrac * max_dist; gg = gg < 0.01 ? 0.01 : (gg > 0.05 ? 0.05 : gg);
        const double rs = max_dist + gg;
        a.r2s = (float)(rs * rs * (1.0 + 1e-6)); a.rs_minus_r = (float)gg;
        memcpy(inits[g].T, T0 + 16 * g, sizeof inits[g].T);
    }
    bool grid = false, grid_ok = false; (void)grid_ok;
    if (icp_use_grid()) {
        std::vector<int> levels((size_t)G); std::vector<GridView> views((size_t)G);
        for (int g = 0; g < G; g++) levels[g] = pcr_grid_level_for(tgt[g], std::sqrt((double)args[g].r2s));
        bool all = true;
        for (int g = 0; g < G; g++) all = all && levels[g] >= 0 && tgt[g]->cap > 0;
        if (all) {                          // one kernel form per launch: the grid form only when every target of the group has a cell hash
            PCR_TRY(pcr_dev_build_grid_batch(ctx, tgt, levels.data(), G, views.data()));
            for (int g = 0; g < G; g++) { args[g].grid = views[g]; grid = grid_ok = true; }
            for (int g = 0; g < G; g++) grid = grid && views[g].tab != nullptr;
            if (grid)
                for (int g = 0; g < G; g++) {
                    const double rs = max_dists[g] + icp_gap_for_level(tgt[g], levels[g], max_dists[g], (double)args[g].rs_minus_r);
                    args[g].r2s = (float)(rs * rs * (1.0 + 1e-6)); args[g].rs_minus_r = (float)(rs - max_dists[g]);
                }
        }
    }
    if (nbf > 4096) { ctx->err = "GICP group: cloud too large for the fused iteration kernel"; return PCR_EINVAL; }
    // arguments and start poses live in a per-context device buffer at a FIXED address (the captured graph reads them there)
    const size_t args_bytes = sizeof(IcpArgs) * 32 + sizeof(IcpInit) * 32;
    if (!ctx->icp_group_dev) {
        if (hipMalloc((void **)&ctx->icp_group_dev, args_bytes) != hipSuccess || hipHostMalloc((void **)&ctx->icp_group_host, args_bytes, hipHostMallocDefault) != hipSuccess) {
            ctx->err = "GICP group: argument buffers"; return PCR_ENOMEM;
        }
    }
    memcpy(ctx->icp_group_host, args.data(), sizeof(IcpArgs) * (size_t)G);
    memcpy(ctx->icp_group_host + sizeof(IcpArgs) * 32, inits.data(), sizeof(IcpInit) * (size_t)G);
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(ctx->icp_group_dev, ctx->icp_group_host, args_bytes, hipMemcpyHostToDevice, ctx->stream));
    const IcpArgs *dargs = (const IcpArgs *)ctx->icp_group_dev;
    const IcpInit *dinit = (const IcpInit *)(ctx->icp_group_dev + sizeof(IcpArgs) * 32);
    PCR_LAUNCH(ctx, k_icp_init_g, dim3(G), dim3(64), 0, ctx->stream, st, dinit);
    static const int chunk_env = getenv("PCR_ICP_CHUNK") ? atoi(getenv("PCR_ICP_CHUNK")) : 8;
    const int max_it = p ? p->max_iteration : 30;
    const int total = max_it + 1, CHUNK = chunk_env < 1 ? 1 : (chunk_env > 32 ? 32 : chunk_env);
    static const bool use_graph = !(getenv("PCR_ICP_GRAPH") && atoi(getenv("PCR_ICP_GRAPH")) == 0);
    static const bool use_fused_g = !(getenv("PCR_ICP_FUSED") && atoi(getenv("PCR_ICP_FUSED")) == 0);
    static const bool byval_env = !(getenv("PCR_ICP_BYVAL") && atoi(getenv("PCR_ICP_BYVAL")) == 0);
    const bool byval = byval_env && G <= ICP_BYVAL;
    IcpArgsB hb; memset(&hb, 0, sizeof hb);
    for (int g = 0; g < G && g < ICP_BYVAL; g++) hb.a[g] = args[g];
    auto enqueue = [&](int launch_index) {
        if (launch_index > 0 && use_fused_g) {
            if (byval) PCR_FUSED_LAUNCH(ctx, k_icp_fused_b, grid, tile_pts, dim3(nbf, G), hb);
            else PCR_FUSED_LAUNCH(ctx, k_icp_fused_g, grid, tile_pts, dim3(nbf, G), dargs);
            return;
        }
        if (grid) PCR_LAUNCH(ctx, k_icp_nn_g<true>, dim3(nbnn, G), dim3(ICP_BS), 0, ctx->stream, dargs);
        else PCR_LAUNCH(ctx, k_icp_nn_g<false>, dim3(nbnn, G), dim3(ICP_BS), 0, ctx->stream, dargs);
        PCR_LAUNCH(ctx, k_icp_lin_g, dim3(nbmax, G), dim3(LIN_BS), 0, ctx->stream, dargs);
    };
    // One captured chunk per launch FORM (group size, kernel forms, tile, chunk length): what differs from call to call -- the
    // grid widths and, for the by-value kernels, the argument batch -- is patched into the instantiated graph's kernel nodes
    // (hipGraphExecKernelNodeSetParams), so circuits whose clouds differ in size do not pay a capture + instantiate per call.
    const void *dargs_v = dargs;
    std::string now;
    { const int w[3] = {nbnn, nbmax, nbf}; now.assign((const char *)w, sizeof w); if (byval) now.append((const char *)&hb, sizeof hb); }
    auto graph_for = [&](int which, int len, hipGraphExec_t *out) -> int {
        *out = nullptr;
        if (!use_graph) return PCR_OK;
        const long long kv[6] = {0x47525550ll /* "GRUP" */, G + (grid ? 1000 : 0) + (use_fused_g ? 0 : 2000) + (byval ? 4000 : 0), tile_pts, len, which, (long long)(uintptr_t)dargs_v};
        std::string key((const char *)kv, sizeof kv);
        IcpGraph *hit = nullptr;
        for (auto &gr : ctx->icp_graphs) if (gr.key == key) { hit = &gr; break; }
        if (!hit) {
            IcpGraph e; e.key = key;
            // (the graph objects belong to `e` until it is stored: an early error return below must not leak them)
            struct Owner { IcpGraph *g; ~Owner() { if (g) { if (g->exec) (void)hipGraphExecDestroy(g->exec); if (g->graph) (void)hipGraphDestroy(g->graph); } } } owner{&e};
            PCR_HIP_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < len; k++) enqueue(which == 0 ? k : len + k);
            PCR_HIP_CHECK(ctx, hipStreamEndCapture(ctx->stream, &e.graph));
            PCR_HIP_CHECK(ctx, hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0));
            // the captured chunk is a chain: walk it from its root so that nodes[] is in launch order
            size_t nr = 1; hipGraphNode_t node = nullptr;
            PCR_HIP_CHECK(ctx, hipGraphGetRootNodes(e.graph, &node, &nr));
            if (nr != 1) { ctx->err = "GICP group: captured chunk has more than one root"; return PCR_EHIP; }
            while (node) {
                hipGraphNodeType nt;
                PCR_HIP_CHECK(ctx, hipGraphNodeGetType(node, &nt));
                if (nt != hipGraphNodeTypeKernel) { ctx->err = "GICP group: captured chunk holds a node that is not a kernel"; return PCR_EHIP; }
                e.nodes.push_back(node);
                size_t nd = 0;
                PCR_HIP_CHECK(ctx, hipGraphNodeGetDependentNodes(node, nullptr, &nd));
                if (nd == 0) break;
                if (nd != 1) { ctx->err = "GICP group: captured chunk is not a chain"; return PCR_EHIP; }
                hipGraphNode_t next = nullptr;
                PCR_HIP_CHECK(ctx, hipGraphNodeGetDependentNodes(node, &next, &nd));
                node = next;
            }
            {   // launch order of a chunk: launch 0 is search + linearisation (two kernels), every later launch one fused kernel (or the same two)
                size_t expect = 0;
                for (int k = 0; k < len; k++) expect += ((which == 0 ? k : len + k) > 0 && use_fused_g) ? 1 : 2;
                if (e.nodes.size() != expect) { ctx->err = "GICP group: captured chunk does not match its launch list"; return PCR_EHIP; }
            }
            e.baked = now;
            if (ctx->icp_graphs.size() >= 48) {           // evict the oldest entry (the stream is drained first: a replay of it may still be queued)
                PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                (void)hipGraphExecDestroy(ctx->icp_graphs[0].exec);
                if (ctx->icp_graphs[0].graph) (void)hipGraphDestroy(ctx->icp_graphs[0].graph);
                ctx->icp_graphs.erase(ctx->icp_graphs.begin());
            }
            owner.g = nullptr;                            // stored: the context owns the graph from here on
            ctx->icp_graphs.push_back(std::move(e));
            hit = &ctx->icp_graphs.back();
        } else if (hit->baked != now) {
            // One captured chunk per launch FORM (group size, kernel forms, tile, chunk length): what differs from call to call -- the
            // grid widths and, for the by-value kernels, the argument batch -- is patched into the instantiated graph's kernel nodes
            // (hipGraphExecKernelNodeSetParams), so circuits whose clouds differ in size do not pay a capture + instantiate per call.
            size_t i = 0;
            for (int k = 0; k < len; k++) {
                const int launch_index = which == 0 ? k : len + k;
                const bool one = launch_index > 0 && use_fused_g;
                for (int part = 0; part < (one ? 1 : 2); part++, i++) {
                    if (i >= hit->nodes.size()) { ctx->err = "GICP group: captured chunk shorter than its launch list"; return PCR_EHIP; }
                    hipKernelNodeParams np;
                    PCR_HIP_CHECK(ctx, hipGraphKernelNodeGetParams(hit->nodes[i], &np));
                    void *kp[1];
                    if (one) { np.gridDim = dim3(nbf, G); kp[0] = byval ? (void *)&hb : (void *)&dargs_v; }
                    else { np.gridDim = part == 0 ? dim3(nbnn, G) : dim3(nbmax, G); kp[0] = (void *)&dargs_v; }
                    np.kernelParams = kp; np.extra = nullptr;
                    PCR_HIP_CHECK(ctx, hipGraphExecKernelNodeSetParams(hit->exec, hit->nodes[i], &np));
                }
            }
            hit->baked = now;
        }
        *out = hit->exec;
        return PCR_OK;
    };
    // two read-back slots of G states each in the pinned window
    const size_t slot_bytes = sizeof(IcpState) * (size_t)G;
    if (2 * slot_bytes > ctx->pinned_cap) { ctx->err = "GICP group: pinned window too small"; return PCR_ENOMEM; }
    IcpState *slots[2] = {(IcpState *)ctx->pinned, (IcpState *)(ctx->pinned + slot_bytes)};
    int launched = 0, cur = 0, prev = -1, n_chunks = 0, next_len = CHUNK;
    std::vector<int> chunk_first, chunk_last;
    std::vector<IcpState> fin((size_t)G); bool have = false;
    for (;;) {
        const bool enq = launched < total;
        if (enq) {
            const int want = launched == 0 ? CHUNK : next_len;
            const int c = total - launched < want ? total - launched : want;
            if (ctx->profiling) {
                while ((int)ctx->prof_events.size() < 2 * (n_chunks + 1)) { hipEvent_t e; PCR_HIP_CHECK(ctx, hipEventCreate(&e)); ctx->prof_events.push_back(e); }
                PCR_HIP_CHECK(ctx, hipEventRecord(ctx->prof_events[2 * n_chunks], ctx->stream));
            }
            hipGraphExec_t ge = nullptr;
            if (c == want) PCR_TRY(graph_for(launched == 0 ? 0 : 1, c, &ge));
            if (ge) PCR_HIP_CHECK(ctx, hipGraphLaunch(ge, ctx->stream));
            else for (int k = 0; k < c; k++) enqueue(launched + k);
            if (ctx->profiling) PCR_HIP_CHECK(ctx, hipEventRecord(ctx->prof_events[2 * n_chunks + 1], ctx->stream));
            chunk_first.push_back(launched); chunk_last.push_back(launched + c);
            n_chunks++;
            launched += c;
            PCR_HIP_CHECK(ctx, hipMemcpyAsync(slots[cur], st, slot_bytes, hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[cur], ctx->stream));
        }
        if (prev >= 0) {
            PCR_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev[prev]));
            bool all = true;
            for (int g = 0; g < G; g++) all = all && slots[prev][g].done;
            if (all) { memcpy(fin.data(), slots[prev], slot_bytes); have = true; break; }
            next_len = 1;                               // the group goes on until its last pair has stopped: the chunk that pair asks for
            for (int g = 0; g < G; g++)
                if (!slots[prev][g].done) { const int l = icp_next_chunk(slots[prev][g], args[g].rel_fit, args[g].rel_rmse, CHUNK, true); next_len = l > next_len ? l : next_len; }
        }
        if (!enq) break;
        prev = cur; cur ^= 1;
    }
    if (!have) { ctx->err = "GICP group loop ended without a final state"; return PCR_EHIP; }
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // drain the no-op tail: pinned slots, argument buffer and arena are reused
    if (ctx->profiling) {
        int longest = 0;
        for (int g = 0; g < G; g++) longest = fin[g].launches > longest ? fin[g].launches : longest;
        for (int c = 0; c < n_chunks; c++) {
            const int first = chunk_first[c], last = chunk_last[c];
            if (last <= longest) {
                float ms = 0;
                PCR_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->prof_events[2 * c], ctx->prof_events[2 * c + 1]));
                ctx->prof[0] += ms; ctx->prof[1] += last - first;
            }
        }
        for (int g = 0; g < G; g++) {
            ctx->prof[2] += (double)fin[g].t_live * 0.01; ctx->prof[3] += fin[g].launches;
            ctx->prof[4] += 48.0 * (double)fin[g].ns * (double)fin[g].launches;
            ctx->prof[11] += (double)fin[g].searched;
            ctx->prof[6] += (double)fin[g].t_dbg[0] * 0.01; ctx->prof[7] += (double)fin[g].t_dbg[3] * 0.01; ctx->prof[14] += (double)fin[g].t_dbg[1] * 0.01;
        }
        ctx->prof[5] += launched; ctx->prof[13] += longest;                // issued / live launches of the GROUP's loop (live: some pair still iterating)
        if (pcr_options().debug_stamps.load(std::memory_order_relaxed)) {
            double l = 0, t0 = 0, t1 = 0, t3 = 0, tl = 0;
            for (int g = 0; g < G; g++) { l += fin[g].launches; t0 += fin[g].t_dbg[0] * 0.01; t1 += fin[g].t_dbg[1] * 0.01; t3 += fin[g].t_dbg[3] * 0.01; tl += fin[g].t_live * 0.01; }
            fprintf(stderr, "icp group stamps (us/launch, mean over %d pairs): slowest-wg A+B+C %.1f phase %.1f sums-done %.1f end %.1f (pair-launches %.0f ns %d)\n", G, t0 / l, t1 / l, t3 / l, tl / l, l, fin[0].ns);
        }
    }
    for (int g = 0; g < G; g++) {
        state_to_result(fin[g], &out[g]);
        for (int k = 0; k < 16; k++) if (!std::isfinite(fin[g].T[k])) { ctx->err = "non-finite pose"; return PCR_ENUMERIC; }
    }
    return PCR_OK;
}

// ---- ALL scales of a group in one loop (round 5).  pcr_dev_gicp_group per scale makes the whole group wait for the slowest pair of every scale:
// on the shipped NCLT scans a pair needs 222 launches over its five scales, a lockstep group of 24 pairs 384 (the sum over the scales of the group's
// maximum), and if every pair went on to its next scale by itself the group would need 242 (the largest sum of a pair).  Here it does, on the
// DEVICE: the kernels read a pair's arguments from a fixed slot of a device table, the arguments of all its scales wait next to it, and the last
// workgroup of the launch in which the criteria of a scale hold stores the result and copies the next scale's arguments into the slot
// (icp_finish, IcpArgs.ms_*).  The first launch of a scale is then k_icp_fused itself with no certificate valid: every query searched, the
// linearisation in the same 512-point rows as k_icp_lin -- which is why only groups whose tiles are 512 points take this path (NCLT-size clouds):
// the sums, hence the bits, are those of the scale-by-scale loop.  (A host-driven form -- read-back, argument update and cold-start kernels per
// pair that moved on -- was built first: its 40-60 extra stream operations per group ate the gain.)
// src / tgt / max_dists are indexed [g * S + s]; out likewise; match_dev[g] (optional) receives the matches of g's last scale.  Returns 1
// (nothing done) for groups it does not serve: the caller then runs the scales one after the other.
int pcr_dev_gicp_group_scales(pcr_context *ctx, int G, int S, const DevCloud *const *src, const DevCloud *const *tgt, const double *max_dists, const double *T0,
                              const pcr_gicp_params *p, pcr_result *out, int32_t *const *match_dev) {
    if (G < 1 || G > 32 || S < 1 || S > 8) return 1;
    static const bool use_fused_g = !(getenv("PCR_ICP_FUSED") && atoi(getenv("PCR_ICP_FUSED")) == 0);
    if (!use_fused_g || !icp_use_grid()) return 1;
    if (pcr_options().icp_verify.load(std::memory_order_relaxed) || pcr_options().debug_visits.load(std::memory_order_relaxed)) return 1;
    for (int k = 0; k < G * S; k++) {
        if (!(max_dists[k] > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
        if (!src[k]->nrm || !tgt[k]->nrm) { ctx->err = "GICP needs normals on both clouds"; return PCR_EINVAL; }
        if (tgt[k]->cap <= 0) return 1;
    }
    int max_cap = 1;
    for (int k = 0; k < G * S; k++) max_cap = src[k]->cap > max_cap ? src[k]->cap : max_cap;
    const int tile_pts = fused_tile_points(ctx, max_cap, G);
    for (int k = 0; k < G * S; k++) {          // rows of 512 points, as k_icp_lin's
        const int cap = src[k]->cap > 0 ? src[k]->cap : 1;
        const int tile_k = icp_group_tile(cap, tile_pts) ? icp_group_tile(cap, tile_pts) : tile_pts;
        if (tile_k != LIN_BS) return 1;
    }
    ArenaMark mark(ctx);
    static const double gfrac = getenv("PCR_ICP_GAP") ? atof(getenv("PCR_ICP_GAP")) : 0.25;
    // cell hashes of all G x S targets in one batch
    std::vector<int> levels((size_t)G * S); std::vector<GridView> views((size_t)G * S);
    for (int k = 0; k < G * S; k++) {
        double gg = gfrac * max_dists[k]; gg = gg < 0.01 ? 0.01 : (gg > 0.05 ? 0.05 : gg);
        const double rs = max_dists[k] + gg;
        levels[k] = pcr_grid_level_for(tgt[k], std::sqrt((double)(float)(rs * rs * (1.0 + 1e-6))));
        if (levels[k] < 0) return 1;
    }
    PCR_TRY(pcr_dev_build_grid_batch(ctx, tgt, levels.data(), G * S, views.data()));
    for (int k = 0; k < G * S; k++) if (!views[k].tab) return 1;
    // per pair: one state and one set of certificate / partial-sum / match buffers (a pair is at one scale at a time), sized by its largest scale
    const size_t args_bytes = sizeof(IcpArgs) * 32 + sizeof(IcpInit) * 32;
    if (!ctx->icp_group_dev) {
        if (hipMalloc((void **)&ctx->icp_group_dev, args_bytes) != hipSuccess || hipHostMalloc((void **)&ctx->icp_group_host, args_bytes, hipHostMallocDefault) != hipSuccess) {
            ctx->err = "GICP group: argument buffers"; return PCR_ENOMEM;
        }
    }
    IcpArgs *dargs = (IcpArgs *)ctx->icp_group_dev;
    IcpInit *dinit = (IcpInit *)(ctx->icp_group_dev + sizeof(IcpArgs) * 32);
    const void *dargs_v = dargs;
    IcpState *st = arena<IcpState>(ctx, G), *hist = arena<IcpState>(ctx, (size_t)G * S);
    IcpArgs *all_dev = arena<IcpArgs>(ctx, (size_t)G * S);
    if (!st || !hist || !all_dev) return PCR_ENOMEM;
    std::vector<IcpArgs> all((size_t)G * S);
    int nbf = 1;
    for (int g = 0; g < G; g++) {
        int cap_g = 1, rows_g = 1;
        for (int s_ = 0; s_ < S; s_++) {
            const int cap = src[g * S + s_]->cap > 0 ? src[g * S + s_]->cap : 1;
            const int f_ = ((cap + LIN_BS - 1) / LIN_BS + 7) & ~7;
            nbf = f_ > nbf ? f_ : nbf; cap_g = cap > cap_g ? cap : cap_g; rows_g = f_ > rows_g ? f_ : rows_g;
        }
        double *partials = arena<double>(ctx, (size_t)rows_g * NVP);
        int32_t *match = (match_dev && match_dev[g]) ? match_dev[g] : arena<int32_t>(ctx, cap_g);
        float4 *ref = arena<float4>(ctx, cap_g); int32_t *rbest = arena<int32_t>(ctx, cap_g); int4 *clist = arena<int4>(ctx, cap_g);
        if (!partials || !match || !ref || !rbest || !clist) return PCR_ENOMEM;
        for (int s_ = 0; s_ < S; s_++) {
            const int k = g * S + s_;
            IcpArgs &a = all[k]; memset(&a, 0, sizeof a);
            fill_args(a, src[k], tgt[k], max_dists[k], p, match, st + g, partials, 0);
            a.tile_rt = icp_group_tile(src[k]->cap > 0 ? src[k]->cap : 1, tile_pts);
            a.ref = ref; a.rbest = rbest; a.clist = clist;
            double gg = gfrac * max_dists[k]; gg = gg < 0.01 ? 0.01 : (gg > 0.05 ? 0.05 : gg);
            const double rs = max_dists[k] + icp_gap_for_level(tgt[k], levels[k], max_dists[k], gg);      // (as pcr_dev_gicp_group: the cap grows to what the level's cells cover)
            a.r2s = (float)(rs * rs * (1.0 + 1e-6)); a.rs_minus_r = (float)(rs - max_dists[k]);
            a.grid = views[k];
            a.ms_args = all_dev + (size_t)g * S; a.ms_self = dargs + g; a.ms_hist = hist + (size_t)g * S; a.ms_scales = S; a.ms_index = s_;
        }
    }
    if (nbf > 4096) return 1;
    std::vector<IcpArgs> first((size_t)G); std::vector<IcpInit> inits((size_t)G);
    for (int g = 0; g < G; g++) { first[g] = all[(size_t)g * S]; memcpy(inits[g].T, T0 + 16 * g, sizeof inits[g].T); }
    // (pageable sources: staged before the calls return)
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(all_dev, all.data(), sizeof(IcpArgs) * all.size(), hipMemcpyHostToDevice, ctx->stream));
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(dargs, first.data(), sizeof(IcpArgs) * (size_t)G, hipMemcpyHostToDevice, ctx->stream));
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(dinit, inits.data(), sizeof(IcpInit) * (size_t)G, hipMemcpyHostToDevice, ctx->stream));
    PCR_LAUNCH(ctx, k_icp_init_g, dim3(G), dim3(64), 0, ctx->stream, st, (const IcpInit *)dinit);
    static const int chunk_env = getenv("PCR_ICP_CHUNK") ? atoi(getenv("PCR_ICP_CHUNK")) : 8;
    const int CHUNK = chunk_env < 1 ? 1 : (chunk_env > 32 ? 32 : chunk_env);
    static const bool use_graph = !(getenv("PCR_ICP_GRAPH") && atoi(getenv("PCR_ICP_GRAPH")) == 0);
    // (every tile of the group is 512 points: the <512> form of the kernel, one point per lane, instead of the <1024> form with its second point masked off)
    static const bool form512 = !(getenv("PCR_ICP_FORM512") && atoi(getenv("PCR_ICP_FORM512")) == 0);
    const int launch_tile = form512 ? LIN_BS : tile_pts;
    auto enqueue_fused = [&]() { PCR_FUSED_LAUNCH(ctx, k_icp_fused_g, true, launch_tile, dim3(nbf, G), (const IcpArgs *)dargs); };
    auto graph_for = [&](int len, hipGraphExec_t *outg) -> int {     // a chunk of `len` fused launches, captured once per (G, tile, grid width, length)
        *outg = nullptr;
        if (!use_graph) return PCR_OK;
        const long long kv[6] = {0x47525053ll /* "GRPS" */, G, launch_tile, len, nbf, (long long)(uintptr_t)dargs_v};
        std::string key((const char *)kv, sizeof kv);
        for (auto &gr : ctx->icp_graphs) if (gr.key == key) { *outg = gr.exec; return PCR_OK; }
        IcpGraph e; e.key = key;
        struct Owner { IcpGraph *g; ~Owner() { if (g) { if (g->exec) (void)hipGraphExecDestroy(g->exec); if (g->graph) (void)hipGraphDestroy(g->graph); } } } owner{&e};
        PCR_HIP_CHECK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < len; k++) enqueue_fused();
        PCR_HIP_CHECK(ctx, hipStreamEndCapture(ctx->stream, &e.graph));
        PCR_HIP_CHECK(ctx, hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0));
        if (ctx->icp_graphs.size() >= 48) {
            PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipGraphExecDestroy(ctx->icp_graphs[0].exec);
            if (ctx->icp_graphs[0].graph) (void)hipGraphDestroy(ctx->icp_graphs[0].graph);
            ctx->icp_graphs.erase(ctx->icp_graphs.begin());
        }
        owner.g = nullptr;
        ctx->icp_graphs.push_back(std::move(e));
        *outg = ctx->icp_graphs.back().exec;
        return PCR_OK;
    };
    const size_t slot_bytes = sizeof(IcpState) * (size_t)G;
    if (2 * slot_bytes > ctx->pinned_cap) { ctx->err = "GICP group: pinned window too small"; return PCR_ENOMEM; }
    IcpState *slots[2] = {(IcpState *)ctx->pinned, (IcpState *)(ctx->pinned + slot_bytes)};
    int cur = 0, prev = -1, next_len = CHUNK, launched = 0, live_launches = 0, n_chunks = 0;
    std::vector<int> chunk_first, chunk_last;
    const int max_it = p ? p->max_iteration : 30;
    const long long launch_limit = (long long)S * ((long long)max_it + 2) + 4 * CHUNK + 64;
    bool finished = false;
    for (;;) {
        const int c = next_len;
        if (ctx->profiling) {
            while ((int)ctx->prof_events.size() < 2 * (n_chunks + 1)) { hipEvent_t e; PCR_HIP_CHECK(ctx, hipEventCreate(&e)); ctx->prof_events.push_back(e); }
            PCR_HIP_CHECK(ctx, hipEventRecord(ctx->prof_events[2 * n_chunks], ctx->stream));
        }
        hipGraphExec_t ge = nullptr;
        PCR_TRY(graph_for(c, &ge));
        if (ge) PCR_HIP_CHECK(ctx, hipGraphLaunch(ge, ctx->stream));
        else for (int k = 0; k < c; k++) enqueue_fused();
        if (ctx->profiling) PCR_HIP_CHECK(ctx, hipEventRecord(ctx->prof_events[2 * n_chunks + 1], ctx->stream));
        chunk_first.push_back(launched); chunk_last.push_back(launched + c); n_chunks++;
        launched += c;
        PCR_HIP_CHECK(ctx, hipMemcpyAsync(slots[cur], st, slot_bytes, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[cur], ctx->stream));
        if (prev >= 0) {
            PCR_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev[prev]));
            bool all_done = true;
            next_len = 1;
            for (int g = 0; g < G; g++) {
                const IcpState &sg = slots[prev][g];
                if (sg.done) continue;                    // (done is set by the LAST scale only)
                all_done = false;
                const int l = icp_next_chunk(sg, all[(size_t)g * S].rel_fit, all[(size_t)g * S].rel_rmse, CHUNK, true);
                next_len = l > next_len ? l : next_len;
            }
            if (all_done) { finished = true; break; }
            live_launches = launched - c;                 // (the chunk just queued may still turn out to be needed: counted when the next read-back says so)
        }
        if (launched > launch_limit) { ctx->err = "GICP group loop (all scales) did not end"; return PCR_EHIP; }
        prev = cur; cur ^= 1;
    }
    if (!finished) { ctx->err = "GICP group loop ended without a final state"; return PCR_EHIP; }
    std::vector<IcpState> hh((size_t)G * S);
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(hh.data(), hist, sizeof(IcpState) * hh.size(), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // (also drains the no-op tail: pinned slots, argument table and arena are reused)
    for (int k = 0; k < G * S; k++) {
        const IcpState &sg = hh[k];
        state_to_result(sg, &out[k]);
        for (int q = 0; q < 16; q++) if (!std::isfinite(sg.T[q])) { ctx->err = "non-finite pose"; return PCR_ENUMERIC; }
        if (ctx->profiling) {
            ctx->prof[2] += (double)sg.t_live * 0.01; ctx->prof[3] += sg.launches; ctx->prof[4] += 48.0 * (double)sg.ns * (double)sg.launches; ctx->prof[11] += (double)sg.searched;
            ctx->prof[6] += (double)sg.t_dbg[0] * 0.01; ctx->prof[7] += (double)sg.t_dbg[3] * 0.01; ctx->prof[14] += (double)sg.t_dbg[1] * 0.01;
        }
    }
    if (ctx->profiling) {
        for (int c = 0; c < n_chunks; c++)
            if (chunk_last[c] <= live_launches) {          // chunks during which some pair was still iterating: HIP-event time / launches = launch period of the group
                float ms = 0;
                PCR_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->prof_events[2 * c], ctx->prof_events[2 * c + 1]));
                ctx->prof[0] += ms; ctx->prof[1] += chunk_last[c] - chunk_first[c];
            }
        ctx->prof[5] += launched; ctx->prof[13] += live_launches;
    }
    if (pcr_options().debug_stamps.load(std::memory_order_relaxed)) {
        long long sum = 0, mx = 0;
        for (int g = 0; g < G; g++) { long long t = 0; for (int s_ = 0; s_ < S; s_++) t += hh[(size_t)g * S + s_].launches; sum += t; mx = t > mx ? t : mx; }
        fprintf(stderr, "icp group, all scales in one loop: %d pairs x %d scales, %d launches queued in %d chunks (%d before the last pair stopped); launches of a pair: mean %.1f, largest %lld\n",
                G, S, launched, n_chunks, live_launches, (double)sum / G, mx);
    }
    return PCR_OK;
}

int pcr_dev_linearize_once(pcr_context *ctx, const DevCloud *src, const DevCloud *tgt, double max_dist, const double *T,
                           const pcr_gicp_params *p, double *JTJ36, double *JTr6, double *stats3, int32_t *match_dev) {
    if (!(max_dist > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    ArenaMark mark(ctx);
    const int cap = src->cap > 0 ? src->cap : 1;
    const int nbmax = (cap + LIN_BS - 1) / LIN_BS < LIN_MAX_BLOCKS ? (cap + LIN_BS - 1) / LIN_BS : LIN_MAX_BLOCKS;
    const int nbnn = (cap + ICP_BS / OCT - 1) / (ICP_BS / OCT);
    IcpState *st = arena<IcpState>(ctx, 1);
    double *partials = arena<double>(ctx, (size_t)nbmax * NVP);
    int32_t *match = match_dev ? match_dev : arena<int32_t>(ctx, cap);
    if (!st || !partials || !match) return PCR_ENOMEM;
    IcpArgs a; memset(&a, 0, sizeof a); fill_args(a, src, tgt, max_dist, p, match, st, partials, 1);
    IcpInit in; memcpy(in.T, T, sizeof in.T);
    PCR_LAUNCH(ctx, k_icp_init, dim3(1), dim3(64), 0, ctx->stream, st, in);
    PCR_LAUNCH(ctx, k_icp_nn<false>, dim3(nbnn), dim3(ICP_BS), 0, ctx->stream, a);
    PCR_LAUNCH(ctx, k_icp_iter<ICP_MODE_GICP>, dim3(nbmax), dim3(LIN_BS), 0, ctx->stream, a);
    IcpState h;
    PCR_TRY(read_state(ctx, st, &h));
    int t = 0;
    for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) { JTJ36[r * 6 + c] = h.sums[t]; JTJ36[c * 6 + r] = h.sums[t]; t++; }
    for (int r = 0; r < 6; r++) JTr6[r] = h.sums[21 + r];
    stats3[0] = (double)h.count; stats3[1] = h.sums[28]; stats3[2] = h.sums[27];
    return PCR_OK;
}

int pcr_dev_evaluate(pcr_context *ctx, const DevCloud *src, const DevCloud *tgt, double max_dist, const double *T,
                     pcr_result *out, int32_t *match_dev, double *info36) {
    if (!(max_dist > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    ArenaMark mark(ctx);
    const int cap = src->cap > 0 ? src->cap : 1;
    const int nbmax = (cap + LIN_BS - 1) / LIN_BS < LIN_MAX_BLOCKS ? (cap + LIN_BS - 1) / LIN_BS : LIN_MAX_BLOCKS;
    const int nbnn = (cap + ICP_BS / OCT - 1) / (ICP_BS / OCT);
    IcpState *st = arena<IcpState>(ctx, 1);
    double *partials = arena<double>(ctx, (size_t)nbmax * NVP);
    int32_t *match = match_dev ? match_dev : arena<int32_t>(ctx, cap);
    if (!st || !partials || !match) return PCR_ENOMEM;
    IcpArgs a; memset(&a, 0, sizeof a); fill_args(a, src, tgt, max_dist, nullptr, match, st, partials, 1);
    IcpInit in; memcpy(in.T, T, sizeof in.T);
    PCR_LAUNCH(ctx, k_icp_init, dim3(1), dim3(64), 0, ctx->stream, st, in);
    PCR_LAUNCH(ctx, k_icp_nn<false>, dim3(nbnn), dim3(ICP_BS), 0, ctx->stream, a);
    PCR_LAUNCH(ctx, k_icp_iter<ICP_MODE_EVAL>, dim3(nbmax), dim3(LIN_BS), 0, ctx->stream, a);
    IcpState h;
    PCR_TRY(read_state(ctx, st, &h));
    if (out) { state_to_result(h, out); out->iterations = 0; out->converged = 0; }
    if (info36) {
        int t = 0;
        for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) { info36[r * 6 + c] = h.sums[t]; info36[c * 6 + r] = h.sums[t]; t++; }
    }
    return PCR_OK;
}

int pcr_dev_evaluate_group(pcr_context *ctx, int G, const DevCloud *const *src, const DevCloud *const *tgt, double max_dist, const double *T,
                           pcr_result *out, int32_t *const *match_dev) {
    if (!(max_dist > 0.0)) { ctx->err = "max_correspondence_distance <= 0"; return PCR_EINVAL; }
    if (G < 1) return PCR_OK;
    ArenaMark mark(ctx);
    std::vector<IcpArgs> args((size_t)G); std::vector<IcpInit> inits((size_t)G);
    IcpState *st = arena<IcpState>(ctx, G);
    if (!st) return PCR_ENOMEM;
    int gmax = 1, gnn = 1;
    for (int g = 0; g < G; g++) {
        const int cap = src[g]->cap > 0 ? src[g]->cap : 1;
        const int nbmax = (cap + LIN_BS - 1) / LIN_BS < LIN_MAX_BLOCKS ? (cap + LIN_BS - 1) / LIN_BS : LIN_MAX_BLOCKS;
        const int nbnn = (cap + ICP_BS / OCT - 1) / (ICP_BS / OCT);
        double *partials = arena<double>(ctx, (size_t)nbmax * NVP);
        if (!partials || !match_dev[g]) return PCR_ENOMEM;
        memset(&args[g], 0, sizeof(IcpArgs));
        fill_args(args[g], src[g], tgt[g], max_dist, nullptr, match_dev[g], st + g, partials, 1);
        memcpy(inits[g].T, T + 16 * g, sizeof inits[g].T);
        gmax = nbmax > gmax ? nbmax : gmax; gnn = nbnn > gnn ? nbnn : gnn;
    }
    const IcpArgs *da = pcr_desc_upload(ctx, args.data(), G);
    const IcpInit *di = pcr_desc_upload(ctx, inits.data(), G);
    if (!da || !di) return PCR_ENOMEM;
    PCR_LAUNCH(ctx, k_icp_init_g, dim3(G), dim3(64), 0, ctx->stream, st, di);
    PCR_LAUNCH(ctx, k_icp_nn_g<false>, dim3(gnn, G), dim3(ICP_BS), 0, ctx->stream, da);
    PCR_LAUNCH(ctx, k_icp_iter_g<ICP_MODE_EVAL>, dim3(gmax, G), dim3(LIN_BS), 0, ctx->stream, da);
    std::vector<IcpState> h((size_t)G);
    PCR_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), st, sizeof(IcpState) * (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int g = 0; g < G; g++) { state_to_result(h[g], &out[g]); out[g].iterations = 0; out[g].converged = 0; }
    return PCR_OK;
}

// ---- correspondence_set: (source index, target index) rows, mapped back to the caller's point order
__global__ void __launch_bounds__(256) k_match_flags(const int32_t *__restrict__ match, const int *__restrict__ n_ptr, uint8_t *__restrict__ flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < *n_ptr) flags[i] = match[i] >= 0 ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_match_emit(const int32_t *__restrict__ match, const int *__restrict__ n_ptr, const uint8_t *__restrict__ flags, const int *__restrict__ pos,
                                                    const uint32_t *__restrict__ sperm, const uint32_t *__restrict__ tperm, int32_t *__restrict__ corr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= *n_ptr || !flags[i]) return;
    const int o = pos[i], t = match[i];
    corr[2 * (size_t)o] = sperm ? (int32_t)sperm[i] : i;
    corr[2 * (size_t)o + 1] = tperm ? (int32_t)tperm[t] : t;
}

// the same for `count` clouds in three batched launches (blockIdx.y = cloud; indices in the clouds' own order), counts not read back
struct MatchBatchDesc { const int32_t *match; const int *n; uint8_t *flags; const int *pos; int32_t *corr; const uint32_t *sperm, *tperm; };      // (perms: back to the caller's point order, or null)
__global__ void __launch_bounds__(256) k_match_flags_g(const MatchBatchDesc *d) { const MatchBatchDesc a = d[blockIdx.y]; const int i = blockIdx.x * 256 + threadIdx.x; if (i < *a.n) a.flags[i] = a.match[i] >= 0 ? 1 : 0; }
__global__ void __launch_bounds__(256) k_match_emit_g(const MatchBatchDesc *d) {
    const MatchBatchDesc a = d[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= *a.n || !a.flags[i]) return;
    const int o = a.pos[i];
    const int t = a.match[i];
    a.corr[2 * (size_t)o] = a.sperm ? (int32_t)a.sperm[i] : i; a.corr[2 * (size_t)o + 1] = a.tperm ? (int32_t)a.tperm[t] : t;
}
int pcr_dev_compact_matches_batch(pcr_context *ctx, int count, const int32_t *const *match, const int *const *n, const int *cap, int32_t *const *corr_out,
                                  const uint32_t *const *src_perm, const uint32_t *const *tgt_perm) {
    ArenaMark mark(ctx);
    std::vector<MatchBatchDesc> d((size_t)count); std::vector<uint8_t *> flags((size_t)count); std::vector<int *> pos((size_t)count), total((size_t)count);
    int mc = 1;
    int *totals = arena<int>(ctx, count);
    if (!totals) return PCR_ENOMEM;
    for (int k = 0; k < count; k++) {
        const int c = cap[k] > 0 ? cap[k] : 1;
        flags[k] = arena<uint8_t>(ctx, c); pos[k] = arena<int>(ctx, c); total[k] = totals + k;
        if (!flags[k] || !pos[k]) return PCR_ENOMEM;
        d[k] = MatchBatchDesc{match[k], n[k], flags[k], pos[k], corr_out[k], src_perm ? src_perm[k] : nullptr, tgt_perm ? tgt_perm[k] : nullptr};
        mc = c > mc ? c : mc;
    }
    const MatchBatchDesc *dd = pcr_desc_upload(ctx, d.data(), count);
    if (!dd) return PCR_ENOMEM;
    const dim3 grid((unsigned)((mc + 255) / 256), (unsigned)count);
    PCR_LAUNCH(ctx, k_match_flags_g, grid, dim3(256), 0, ctx->stream, dd);
    PCR_TRY(pcr_dev_flag_scan_batch(ctx, count, flags.data(), n, cap, pos.data(), total.data()));
    PCR_LAUNCH(ctx, k_match_emit_g, grid, dim3(256), 0, ctx->stream, dd);
    return PCR_OK;
}

int pcr_dev_compact_matches(pcr_context *ctx, const int32_t *match, const int *n, int cap, const uint32_t *src_perm,
                            const uint32_t *tgt_perm, int32_t *corr_out, int64_t *n_corr) {
    if (cap <= 0) { if (n_corr) *n_corr = 0; return PCR_OK; }
    ArenaMark mark(ctx);
    uint8_t *flags = arena<uint8_t>(ctx, cap);
    int *pos = arena<int>(ctx, cap);
    int *total = arena<int>(ctx, 1);
    if (!flags || !pos || !total) return PCR_ENOMEM;
    const int nb = (cap + 255) / 256;
    PCR_LAUNCH(ctx, k_match_flags, dim3(nb), dim3(256), 0, ctx->stream, match, n, flags);
    PCR_TRY(pcr_dev_flag_scan(ctx, flags, n, cap, pos, total));
    PCR_LAUNCH(ctx, k_match_emit, dim3(nb), dim3(256), 0, ctx->stream, match, n, flags, pos, src_perm, tgt_perm, corr_out);
    if (n_corr) PCR_TRY(pcr_read_count(ctx, total, n_corr));
    return PCR_OK;
}
