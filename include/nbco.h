/* nbco.h -- C ABI of libnbco_hip.so, the MI355X (gfx950) force / integrate engine that drops in
 * behind the reference's evaluator / step / integrator function-pointer interface
 * (locuoco/coulomb_oscillators @ 2024_08_07, Simulation/).
 *
 * Conventions (they mirror the reference's GPU path):
 *   - every `p`, `a`, `b`, `buf`, `param` argument is a DEVICE pointer (main3.cu:699-704);
 *   - positions / velocities / accelerations are fp32 xyz triplets with a 12-byte stride, and a
 *     state buffer is [pos n | vel n | acc n] (kernel.cuh:67, integrator.cuh:24, main3.cu:655-657);
 *   - `param` is the 6-float array {xi/N, 0, 0, kx, ky, kz} of main3.cu:685-692; evaluators use
 *     param[0], the elastic term uses param+3;
 *   - evaluators overwrite `a`; tree evaluators may permute `p` (and then `p + 3n`, the
 *     velocities) unless opts.unsort is set (fmm_cart3_kdtree.cuh:1746-1760);
 *   - the reference's mutable globals (constants.cuh:36-52) become the explicit nbco_opts;
 *   - instead of gpuErrchk's exit() (kernel.cuh:52-65) every entry point returns an int status
 *     (0 = success) and nbco_last_error() gives the message;
 *   - work is enqueued on opts.stream; with opts.sync != 0 an evaluator returns after the stream
 *     has drained, as the reference's evaluators do (direct.cuh:243-244,
 *     fmm_cart3_kdtree.cuh:1762-1763).
 * There is no CPU fallback: every entry point fails with NBCO_ERR_HIP when no HIP device is usable.
 */
#ifndef NBCO_H
#define NBCO_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbco_ctx nbco_ctx;

enum {
	NBCO_OK = 0,
	NBCO_ERR_HIP = 1,        /* a HIP runtime call failed (message in nbco_last_error) */
	NBCO_ERR_ARG = 2,        /* invalid argument */
	NBCO_ERR_CAPACITY = 3,   /* interaction-list / traversal-frontier capacity exceeded
	                            (reference: printf + clamp, fmm_cart3_kdtree.cuh:552-566) */
	NBCO_ERR_UNSUPPORTED = 4
};

/* force evaluators: which f(p, a, n, param) an integrator or nbco_force drives */
enum {
	NBCO_EVAL_DIRECT = 0,         /* direct.cuh:104 `direct` / :171 `direct2` (LDS-tiled all pairs) */
	NBCO_EVAL_DIRECT_KAHAN = 1,   /* direct.cuh:233 `direct3` (compensated accumulation) */
	NBCO_EVAL_FMM_KDTREE = 2,     /* fmm_cart3_kdtree.cuh:1478 `fmm_cart3_kdtree` */
	NBCO_EVAL_FMM_TRACELESS = 3,  /* fmm_cart3_traceless.cuh:282 `fmm_cart3_traceless` */
	NBCO_EVAL_FMM_SYMMETRIC = 4   /* fmm_cart3_symmetric.cuh:413 `fmm_cart3` (uniform octree, symmetric multipoles) */
};

/* integrators of integrator.cuh:32-167 */
enum {
	NBCO_INTEG_EULER = 0,       /* symplectic_euler      :32 */
	NBCO_INTEG_PRE_EULER = 1,   /* pre_symplectic_euler  :50 */
	NBCO_INTEG_LEAPFROG = 2,    /* leapfrog              :68 */
	NBCO_INTEG_FORESTRUTH = 3,  /* forestruth            :100 */
	NBCO_INTEG_PEFRL = 4        /* pefrl                 :134 */
};

typedef struct nbco_opts {
	int   fmm_order;    /* constants.cuh:42 fmm_order, 1..10 */
	float tree_radius;  /* constants.cuh:43 tree_radius (used as given; the reference's CPU driver
	                       truncates it to int, fmm_cart3_kdtree.cuh:1775 -- callers that want that
	                       behaviour pass an integer value) */
	float eps2;         /* constants.cuh:39 EPS2 (softening squared) */
	int   coll;         /* constants.cuh:48 coll: 0 skips the P2P pass */
	int   unsort;       /* constants.cuh:48 b_unsort: restore the caller's particle order */
	float dens_inhom;   /* constants.cuh:50 dens_inhom */
	int   tree_L;       /* constants.cuh:44 tree_L (0 = derive from n and order) */
	int   tree_steps;   /* constants.cuh:45 tree_steps: rebuild the kd-tree every this many
	                       evaluations when unsort == 0.  1 = every evaluation, which is what the
	                       reference's CPU driver does (fmm_cart3_kdtree.cuh:1773-1929).  Setting a different
	                       value (nbco_set_opts) starts a new schedule: the next evaluation rebuilds -- which
	                       is also how a caller that puts a NEW state into an old context gets rid of the
	                       tree of the state before. */
	int   m2l_first;    /* 0: leaf-leaf pairs go to P2P before the admissibility test (reference CPU
	                       traversal, fmm_cart3_kdtree.cuh:586-598); 1: admissibility first
	                       (reference GPU traversal <true>, :520-534) */
	int   sync;         /* != 0: evaluators return after the stream has drained */
	int   list_factor;  /* capacity of the P2P / M2L lists and of the traversal frontier, in units
	                       of the node count (reference: 1000, fmm_cart3_kdtree.cuh:1584-1586).  The
	                       lists are kept in 16 regions of list_factor * nodes / 8 pairs each; a region
	                       that runs full makes the evaluation grow the lists (list_grow) or return
	                       NBCO_ERR_CAPACITY with the caller's arrays untouched (default 48: ~10x the lists of
	                       the BASELINE runs at their start) */
	int   list_grow;    /* != 0 (default): a traversal that overflows its lists doubles the capacity (up to 64 x
	                       list_factor, and at most 2^31 pairs) and repeats the evaluation instead of failing:
	                       long runs change shape -- a ball that starts with 2.7e5 leaf pairs at N = 1M holds 6e6
	                       after 1200 steps, when a few ejected particles have stretched the outer leaves */
	int   far_fp64;     /* != 0: the FMM evaluators (nbco_fmm_kdtree, nbco_fmm_traceless, nbco_fmm_symmetric and the sharded
	                       nbco_dist_* forms) keep multipole / local expansions in double and do P2M, M2M, M2L, L2L and L2P
	                       in fp64 (BASELINE config 5: fp64 far field, fp32 P2P).  Positions, the tree geometry (centres,
	                       boxes, lists -- unchanged, bit for bit) and the near field stay fp32.  The fp32 far field overflows
	                       beyond N ~ 1e5 at orders 9-10 (r^-11 19!! ~ 1e40); this mode does not.
	                       No reference counterpart other than the -DSCAL=double build.  Sharded runs: the exchanged
	                       multipole blocks / LET node records are doubles (nbco_dist_layout sizes follow the option). */
	int   p2p_mutual;   /* != 0: nbco_fmm_kdtree evaluates every leaf pair of its near field once and applies the force to both
	                       leaves (Newton III), as the reference's GPU pair kernel does (fmm_cart3_kdtree.cuh:874-959) -- here
	                       without atomics: the second leaf's sums go to fixed-order "reaction" records, bit-reproducible.
	                       Leaves are taken as 1, 2 or 4 halves of up to 32 particles (nbco_kd_info.p2p_halves; sizes that fill
	                       the 16-lane rows too badly fall back to the one-directional kernel).  The pair kernel itself is 20 %
	                       faster (0.43 against 0.36 of the fp32 peak at N = 1M, p = 6), but writing, linking and summing the
	                       reaction records costs more than that: a step is 8 % slower.  Default 0: the one-directional kernel,
	                       whose sharded results also equal the single-GPU ones bit for bit. */
	int   track_order;  /* != 0: with unsort = 0 the context composes the permutations of all rebuilds since tracking started, so that
	                       NBCO_KD_ORDER maps a position of the (tree-ordered) state to the particle's number in the state the
	                       first tracked evaluation received: snapshots can be written in input order (the
	                       reference writes them in tree order, main3.cu:855-858) */
	void *stream;       /* hipStream_t; NULL = the null stream */
} nbco_opts;

/* reference defaults (constants.cuh:36-52), tree_steps = 1, m2l_first = 0, sync = 1 */
int nbco_opts_default(nbco_opts *o);

int nbco_create(nbco_ctx **out, const nbco_opts *o);   /* owns all scratch, like the reference's
                                                           function-local statics */
int nbco_destroy(nbco_ctx *c);
int nbco_set_opts(nbco_ctx *c, const nbco_opts *o);
int nbco_get_opts(const nbco_ctx *c, nbco_opts *o);
const char *nbco_last_error(const nbco_ctx *c);
int nbco_sync(nbco_ctx *c);

/* ---- basic kernels (kernel.cuh, appel.cuh) -------------------------------------------------- */
int nbco_step(nbco_ctx *c, float *b, const float *a, float ds, long long n);              /* kernel.cuh:100 step: b += a*ds */
int nbco_add_elastic(nbco_ctx *c, const float *p, float *a, long long n, const float *k);  /* kernel.cuh:145 add_elastic: a -= k o p (k == NULL: a -= p) */
int nbco_elastic(nbco_ctx *c, const float *p, float *a, long long n, const float *k);      /* kernel.cuh:198 elastic: a = -k o p */
int nbco_rescale(nbco_ctx *c, float *a, long long n, const float *param);                  /* appel.cuh:514 rescale: a *= param[0] */
int nbco_gather(nbco_ctx *c, float *dst, const float *src, const int *map, long long n);          /* kernel.cuh:229 dst[i] = src[map[i]] (xyz triplets) */
int nbco_gather_inverse(nbco_ctx *c, float *dst, const float *src, const int *map, long long n);  /* kernel.cuh:255 dst[map[i]] = src[i] */
int nbco_copy(nbco_ctx *c, float *dst, const float *src, long long n);                            /* kernel.cuh:281 */

/* ---- force evaluators: void f(VEC *p, VEC *a, int n, const SCAL *param) --------------------- */
int nbco_direct(nbco_ctx *c, const float *p, float *a, long long n, const float *param);   /* direct.cuh:104,171 */
int nbco_direct3(nbco_ctx *c, const float *p, float *a, long long n, const float *param);  /* direct.cuh:233 */
int nbco_fmm_kdtree(nbco_ctx *c, float *p, float *a, long long n, const float *param);     /* fmm_cart3_kdtree.cuh:1478 */
int nbco_fmm_traceless(nbco_ctx *c, float *p, float *a, long long n, const float *param);  /* fmm_cart3_traceless.cuh:282 */
int nbco_fmm_symmetric(nbco_ctx *c, float *p, float *a, long long n, const float *param);  /* fmm_cart3_symmetric.cuh:413 `fmm_cart3`, orders 1..9 */

/* evaluator `kind` on buf = [pos|vel|acc], followed by add_elastic(param+3) when elastic != 0:
 * compute_force (integrator.cuh:22) over the coulombOscillator* wrappers of main3.cu:47-69 */
int nbco_force(nbco_ctx *c, int kind, float *buf, long long n, const float *param, int elastic);

/* one step of integrator `scheme` with evaluator `kind` (+ elastic term), integrator.cuh:32-167.
 * dt and scale are the reference's long double arguments narrowed to double. */
int nbco_integrate(nbco_ctx *c, int scheme, int kind, float *buf, long long n, const float *param,
                   double dt, double scale, int elastic);
/* `steps` steps in one call: the loop body of main3.cu:840-870 between two snapshots.  The final state is bit-identical to
 * `steps` calls of nbco_integrate; leapfrog over the kd-tree evaluator (opts.unsort = 0, no order tracking) runs what lies
 * between two force evaluations -- tree order of x and v, elastic term, both half kicks, the drift, the next build's
 * prologue -- as ONE pass over the state instead of four.  Intermediate states are not observable. */
int nbco_integrate_steps(nbco_ctx *c, int scheme, int kind, float *buf, long long n, const float *param,
                         double dt, double scale, int elastic, int steps);

/* ---- reductions (reductions.cuh) -------------------------------------------------------------- */
int nbco_minmax(nbco_ctx *c, const float *p, long long n, float *minmax6_dev);           /* reductions.cuh:67 minmaxReduce2 */
int nbco_mean_relerr(nbco_ctx *c, const float *x, const float *ref, long long n, float *out_host);  /* reductions.cuh:99 relerrReduce2 (intent) */
int nbco_pow_sum(nbco_ctx *c, const float *x, int expo, long long n, double *out3_host);  /* reductions.cuh:590 powReduce */
/* total energy {kinetic, elastic, coulomb}; no reference counterpart */
int nbco_energy(nbco_ctx *c, const float *buf, long long n, const float *param, double *out3_host);
/* the same with the Coulomb part from the interaction lists and multipoles of the LAST nbco_fmm_kdtree evaluation (which must
 * have been made at the positions in buf): near field pair by pair over the P2P list, far field by evaluating the multipole
 * expansions of the M2L sources at the particles (the reference's m2p_pot3, fmm_cart_base3.cuh:1474-1490), fp64, O(N log N)
 * instead of O(N^2).  Sharded runs: n = the domain's particles, every rank gets its share of the three sums. */
int nbco_energy_fmm(nbco_ctx *c, const float *buf, long long n, const float *param, double *out3_host);

/* ---- introspection of the last kd-tree evaluation (parity tests, benchmarks) ----------------- */
typedef struct nbco_kd_info {
	int L, ntot, order, mlt_max;
	long long n;
	long long p2p_pairs;        /* unordered leaf pairs in the P2P list */
	long long m2l_pairs;        /* unordered node pairs in the M2L list */
	long long directed_p2p;     /* sum 2*m1*m2 over the list + sum m^2 over leaves */
	int rebuilt;                /* the last evaluation rebuilt the tree */
	int build_mode;             /* how this context builds trees: 0 = median selection with two radix passes + exact
	                               resolution of the pivot bucket (default), 1 = three radix passes (after a bucket held
	                               more candidates than the resolver takes), 2 = stable-sort chain (after a pivot had
	                               more exact ties than that).  The trees are identical; only the speed differs. */
	int p2p_halves;             /* near-field kernel of the last evaluation: 0 = one-directional; 1, 2, 4 = mutual (Newton III) with
	                               leaves taken as that many halves of up to 32 particles (opts.p2p_mutual) */
	long long warm_builds;      /* builds whose median selection ran ONE histogram pass per level around the previous build's pivots */
	long long warm_misses;      /* .. of which a window missed the median: the evaluation was repeated with the two-pass select */
	int real_bytes;             /* element size of NBCO_KD_MPOLE / NBCO_KD_LOCAL: 4, or 8 when the tree was built with opts.far_fp64 */
	int long_lists;             /* 1: the last evaluation sorted the long ranges of its near-field list (more than 512 entries of one
	                               target: leaves stretched by ejected particles, late in a run) with the kernel made for them */
} nbco_kd_info;
int nbco_kd_get_info(nbco_ctx *c, nbco_kd_info *info);

enum {
	NBCO_KD_MULT = 0, NBCO_KD_INDEX = 1, NBCO_KD_SPLITDIM = 2,   /* int[ntot] */
	NBCO_KD_CENTER = 3, NBCO_KD_LBOUND = 4, NBCO_KD_RBOUND = 5,  /* float[ntot][3] */
	NBCO_KD_MPOLE = 6,   /* float[ntot][offM], offM = p(p+1)(p+2)/6 */
	NBCO_KD_LOCAL = 7,   /* float[ntot][offL], offL = (p+1)^2 */
	NBCO_KD_P2P_LIST = 8, NBCO_KD_M2L_LIST = 9,                  /* int[pairs][2], unordered */
	NBCO_KD_UNSORT = 10, /* int[n]: tree position -> caller's index (of the last rebuild) */
	NBCO_KD_ORDER = 11   /* int[n]: position in the state -> particle number since opts.track_order was set (cumulative) */
};
int nbco_kd_copy(nbco_ctx *c, int which, void *host_dst, long long host_bytes);

/* ---- octree-traceless evaluator introspection (fmmTree, fmm_cart3_symmetric.cuh:24-30) ---------------------
 * Valid after nbco_fmm_traceless.  Cells are level-major (level l starts at (8^l - 1) / 7), row-major inside a
 * level; tuples are traceless, (order + 1)^2 reals per cell (orders 0..order); a real is a float, or a double
 * when the evaluation ran with opts.far_fp64 (nbco_oct_info.real_bytes). */
typedef struct nbco_oct_info {
	int L, ntot, order;
	int tpl;                /* target-group width of the near-field work units */
	long long n;
	long long m2l_entries;  /* directed (target, source) stencil entries with a non-empty source */
	long long p2p_groups, p2p_desc, p2p_chunks;
	int real_bytes;         /* 4 or 8: element size of NBCO_OCT_MPOLE / NBCO_OCT_LOCAL */
	int mpole_reals;        /* reals per tuple of NBCO_OCT_MPOLE: (order + 1)^2 after nbco_fmm_traceless (traceless, orders 0..order),
	                           (order + 1)(order + 2)(order + 3) / 6 after nbco_fmm_symmetric (symmetric layout, orders 0..order) */
} nbco_oct_info;
int nbco_oct_get_info(nbco_ctx *c, nbco_oct_info *out);
enum {
	NBCO_OCT_MULT = 0, NBCO_OCT_INDEX = 1,   /* int[ntot] (index: leaves only) */
	NBCO_OCT_CENTER4 = 2,                    /* float[ntot][4]: centre xyz, 0 */
	NBCO_OCT_MPOLE = 3, NBCO_OCT_LOCAL = 4,  /* real[ntot][(order+1)^2] */
	NBCO_OCT_KEYS = 5,                       /* uint32[n]: sorted cell keys (appel.cuh:44-55) */
	NBCO_OCT_PERM = 6                        /* uint32[n]: cell-order position -> caller's index */
};
int nbco_oct_copy(nbco_ctx *c, int which, void *host_dst, long long host_bytes);

/* ---- multi-GPU, uniform-octree evaluators: slabs of the sorted cell keys (fmm_cart3_traceless.cuh:196-254 and
 *      appel.cuh:320-381 are the single-GPU passes) ----
 * Every rank holds the whole state buf = [pos | vel | ..] and calls this with the same arguments but its own rank: the tree
 * and the upward pass are built redundantly (O(N)), M2L / L2L / P2P / L2P only for the rank's slab -- the leaf cells
 * [c_r, c_r+1) of the key order (x-layers), c_r = the first cell whose first particle is >= n r / world.  p and the
 * velocities behind it are re-ordered into cell order exactly as by nbco_fmm_traceless (identically on every rank);
 * a[3 i], i in [bounds_host[rank], bounds_host[rank + 1]) are written, the other accelerations are left alone --
 * bounds_host[0 .. world] (host) are the particle boundaries of all slabs, so the caller can all-gather the pieces.
 * Each written acceleration is bit-identical to the single-GPU nbco_fmm_traceless / nbco_fmm_symmetric. */
int nbco_fmm_oct_shard(nbco_ctx *c, float *p, float *a, long long n, const float *param, int symmetric, int world, int rank,
                       long long *bounds_host);

/* ---- multi-GPU: kd-domain sharding (the reference is single-GPU, the sharded tree is the
 *      balanced kd-tree of fmm_cart3_kdtree.cuh:109-137 whose level-log2(G) nodes hold N/G particles each) ----
 * One process per GPU, each with its own nbco_ctx.  This library never communicates: the caller moves
 * the two send buffers with an all-gather (RCCL) between nbco_dist_local and nbco_dist_finish.
 *
 *   nbco_dist_layout_query   sizes of the exchange buffers for (n_global, world, rank) under the current opts
 *   nbco_dist_partition      every `rebalance` steps: state_all = [pos N x 3 | vel N x 3] gathered from all
 *                            ranks (identical everywhere); runs the top log2(world) median splits and
 *                            writes the rank's domain [pos n_local x 3 | vel n_local x 3] to state_local
 *   nbco_dist_local          builds the domain's subtree over buf_local = [pos | vel | ..] (n_local
 *                            particles), upward pass; fills nodes_send (nodes_bytes) and pos_send (pos_bytes)
 *   nbco_dist_finish         nodes_all / pos_all = the world x gathered buffers in rank order; far + near
 *                            field for the own particles -> a_local (tree order); after a rebuild the
 *                            positions and velocities in buf_local are permuted into tree order, exactly
 *                            like nbco_fmm_kdtree with opts.unsort = 0
 * The accelerations equal those of a single-GPU nbco_fmm_kdtree over the n_global particles bit for bit
 * (rank r's particles are positions [r n_local, (r+1) n_local) of the single-GPU tree order).
 * Requirements: world a power of two, n_global % world == 0, n_local >= 4096, opts.unsort = 0. */
typedef struct nbco_dist_layout {
	int world, rank;
	int d;                 /* log2(world): global level of the domain roots */
	int L, L_local;        /* leaf level of the global tree, of the domain's subtree (L - d) */
	int ntot_local;        /* nodes of the domain's subtree */
	int order;
	long long n_global, n_local;
	long long nodes_bytes; /* per-rank node block: float4 csz[ntot_local], float mpole[ntot_local][offM] */
	long long pos_bytes;   /* per-rank position block: float4[n_local], tree order */
	long long csz_bytes;   /* the two parts of the node block on their own: traversal records (centre + squared box */
	long long mpole_bytes; /* diagonal) of the subtree's nodes, and their multipoles; csz_bytes + mpole_bytes = nodes_bytes */
	long long let_node_bytes; /* LET exchange: bytes of one node record {node id, multipole}, padded to 16 */
	int let_counts;           /* LET exchange: 64-bit values in one rank's count block (2 world + 2) */
} nbco_dist_layout;
int nbco_dist_layout_query(nbco_ctx *c, long long n_global, int world, int rank, nbco_dist_layout *out);
int nbco_dist_partition(nbco_ctx *c, const float *state_all, long long n_global, int world, int rank, float *state_local);
int nbco_dist_local(nbco_ctx *c, float *buf_local, long long n_local, void *nodes_send, void *pos_send);
/* The same re-partition WITHOUT gathering the state (no rank ever holds more than its own n_local particles): per level the
 * exact medians by a radix select whose histograms are summed across the ranks, pivot ties ordered by the stable-sort chain's
 * remaining keys, evalBox for the children, local partition; then one all-to-all of [pos | vel] by destination.  Top boxes,
 * split axes, the particles of every domain and their ORDER equal nbco_dist_partition's: both deliver a domain in the order of the
 * gathered state (source rank, then index in the source's state) -- the local build takes the local index as the last key of its
 * stable-sort chain, as the single-GPU build takes the index in its input.
 * The library never communicates: _begin / _next run the local stages and describe, in *next, the collective the caller runs
 * on its workspace (device memory, _workspace bytes) before calling _next again, until op == NBCO_COLL_DONE:
 *   ALLREDUCE_MIN_I32 / ALLREDUCE_SUM_I32   in place, `count` int32 at work + send_off
 *   ALLGATHER                               `count` bytes at work + send_off -> world x count bytes at work + recv_off (rank order)
 *   ALLTOALL                                records of row_bytes: rows_send[r] rows for rank r, consecutive from work + send_off;
 *                                           rows_recv[s] rows from rank s, consecutive at work + recv_off
 * state_local = [pos n_local x 3 | vel n_local x 3] is replaced by the rank's new domain.  world <= 32.  More than 64 particles
 * of one rank tying with a pivot: NBCO_ERR_UNSUPPORTED (nbco_dist_partition handles any input). */
enum {
	NBCO_COLL_DONE = 0,
	NBCO_COLL_ALLREDUCE_MIN_I32 = 1,
	NBCO_COLL_ALLREDUCE_SUM_I32 = 2,
	NBCO_COLL_ALLGATHER = 3,
	NBCO_COLL_ALLTOALL = 4
};
typedef struct nbco_dist_step {
	int op, row_bytes;
	long long send_off, recv_off, count;
	long long rows_send[64], rows_recv[64];
} nbco_dist_step;
int nbco_dist_repartition_workspace(nbco_ctx *c, long long n_global, int world, long long *bytes);
int nbco_dist_repartition_begin(nbco_ctx *c, float *state_local, long long n_global, int world, int rank, void *work,
                                long long work_bytes, nbco_dist_step *next);
int nbco_dist_repartition_next(nbco_ctx *c, nbco_dist_step *next);
/* nbco_dist_local in two halves, so that the all-gather of the positions can run beside the upward pass:
 * _build fills pos_send (subtree build), _upward fills nodes_send (multipoles). */
int nbco_dist_local_build(nbco_ctx *c, float *buf_local, long long n_local, void *pos_send);
int nbco_dist_local_upward(nbco_ctx *c, float *buf_local, long long n_local, void *nodes_send);
int nbco_dist_finish(nbco_ctx *c, const void *nodes_all, const void *pos_all, float *buf_local, float *a_local,
                     const float *param);
/* The same evaluation with the node block travelling in two all-gathers, so that the multipoles (224 of the 240 bytes per
 * node at order 6) are on the wire while the GPUs traverse:
 *   _local_geom        subtree build; fills pos_send (pos_bytes) and csz_send (csz_bytes)     -> all-gather both
 *   _local_mpole       upward pass; fills mpole_send (mpole_bytes)                            -> all-gather
 *   _finish_traverse   needs csz_all / pos_all (world x the blocks, rank order): global tree geometry + dual traversal;
 *                      returns with the work enqueued, pos_all must stay valid until _finish_rest has run
 *   _finish_rest       needs mpole_all: lists, near and far field, L2P -> a_local, buf_local as nbco_dist_finish
 * Results are identical to nbco_dist_local + nbco_dist_finish. */
int nbco_dist_local_geom(nbco_ctx *c, float *buf_local, long long n_local, void *pos_send, void *csz_send);
int nbco_dist_local_mpole(nbco_ctx *c, float *buf_local, long long n_local, void *mpole_send);
int nbco_dist_finish_traverse(nbco_ctx *c, const void *csz_all, const void *pos_all);
int nbco_dist_finish_rest(nbco_ctx *c, const void *mpole_all, float *buf_local, float *a_local, const float *param);
/* ---- the same evaluation with a locally-essential-tree (LET) exchange: a rank receives only the multipoles and positions
 *      its own interaction lists name, instead of every domain's whole block (north_star) ----
 * The dual traversal is symmetric and every rank runs it over the same global geometry, so the pairs a rank emits are also the
 * list of what the other ranks read of it: nothing is estimated.  Per evaluation:
 *   _let_local_geom   subtree build; fills csz_send (csz_bytes)                                  -> all-gather (16 B per node)
 *   _let_local_mpole  upward pass on the second stream (the multipoles stay on the device)
 *   _let_select       csz_all = world x csz_bytes: global geometry, traversal, selection; writes this rank's count block to
 *                     counts_send (DEVICE, let_counts 64-bit values: [2 r] node records / [2 r + 1] position records it
 *                     will send to rank r, [2 world] = 1 when its traversal ran out of list room, [2 world + 1] = 1 when its
 *                     tree build was flagged -- pivot ties beyond the resolver, a missed warm-select window)
 *                                                                                                  -> all-gather, copy to host
 *                     If any rank's block reports list overflow, every rank calls _let_select again (it repeats the traversal
 *                     with more room where needed, is a no-op elsewhere) and the blocks are gathered again.  If any reports a
 *                     flagged build, every rank starts the evaluation again from _let_local_geom (the flagged rank builds more
 *                     conservatively, the others reproduce their trees): the LET form needs no host round trip behind the build.
 *   _let_pack         counts_all = the gathered blocks on the HOST, [sender][let_counts]; fills pos_send (16 B records
 *                     {x, y, z, global particle index}) and mpole_send (let_node_bytes records {global node id, multipole}),
 *                     segments in receiver order                                                  -> two all-to-alls with
 *                     the splits counts_all[me][2 r + 1] / counts_all[me][2 r]
 *   _let_finish       pos_recv / mpole_recv = the received records in sender order; lists, near and far field, L2P ->
 *                     a_local, buf_local as nbco_dist_finish.  A guard checks every source of the sorted M2L and P2P lists
 *                     against what arrived; a miss is reported by the next _let_pack and by _let_check (which synchronises).
 * Accelerations are bit-identical to nbco_dist_finish and to the single-GPU nbco_fmm_kdtree. */
int nbco_dist_let_local_geom(nbco_ctx *c, float *buf_local, long long n_local, void *csz_send);
int nbco_dist_let_local_mpole(nbco_ctx *c, float *buf_local, long long n_local);
int nbco_dist_let_select(nbco_ctx *c, const void *csz_all, long long *counts_send);
int nbco_dist_let_pack(nbco_ctx *c, const long long *counts_all, void *pos_send, void *mpole_send);
int nbco_dist_let_finish(nbco_ctx *c, const long long *counts_all, const void *pos_recv, const void *mpole_recv, float *buf_local,
                         float *a_local, const float *param);
int nbco_dist_let_check(nbco_ctx *c);
/* The same exchange without the host round trip in the middle of the evaluation (nbco_dist_let_pack needs the gathered counts
 * on the host before anything behind it can be queued).  The caller sizes the segments from the counts of the evaluation BEFORE,
 * with head room -- caps_out[2 r] node records and caps_out[2 r + 1] position records for receiver r (0, 0 for itself), the
 * same numbers receiver r passes as caps_in[2 me], caps_in[2 me + 1] -- and exchanges whole segments:
 *   nbco_dist_let_select -> all-gather of the counts, copied to the host asynchronously
 *   nbco_dist_let_pack_capped(caps_out) -> the two all-to-alls with splits from the caps -> nbco_dist_let_finish_capped(caps_in)
 *   [everything queued; now wait for the counts]  ok = every counts_all[s][2 r], [2 r + 1] within the segment rank s sized for
 *   rank r, and no counts_all[s][2 world], [2 world + 1] flag  (the same matrix, hence the same verdict, on every rank)
 *   nbco_dist_let_settle(ok)
 * Records that do not fit a segment are dropped and free records are marked, so an attempt that is not ok has read nothing
 * out of bounds: it is void -- the accelerations are not to be used, buf_local holds the same particles (possibly reordered) --
 * and every rank repeats the evaluation from nbco_dist_let_local_geom, with nbco_dist_let_pack and exact sizes.  _settle also
 * reports the receiver guard of the attempt before (the guard of the last attempt: the next _settle / _pack, or _check). */
int nbco_dist_let_pack_capped(nbco_ctx *c, const long long *caps_out, void *pos_send, void *mpole_send);
int nbco_dist_let_finish_capped(nbco_ctx *c, const long long *caps_in, const void *pos_recv, const void *mpole_recv, float *buf_local,
                                float *a_local, const float *param);
int nbco_dist_let_settle(nbco_ctx *c, int ok);
/* Between the force evaluation of one leapfrog step of a sharded run and that of the next (any of the three forms above; the
 * accelerations a_local = buf_local + 6 n_local are those nbco_dist_*finish* wrote, WITHOUT the elastic term): one pass over the
 * domain's state that does  a -= k o x (elastic != 0), v += a dt scale / 2  [end of this step]  and  v += a dt scale / 2,
 * x += v dt  [start of the next]  -- the same operations with the same roundings as nbco_add_elastic + three nbco_step calls --
 * and the next local build's prologue.  The next call on this context must be the next evaluation's nbco_dist_local* /
 * nbco_dist_let_local_geom (a re-partition in between is fine: it discards the prologue). */
int nbco_dist_turnaround(nbco_ctx *c, float *buf_local, long long n_local, const float *param, double dt, double scale, int elastic);
/* The context's second stream (hipStream_t), on which the far-field chain runs and which alone reads mpole_all in
 * nbco_dist_finish_rest: a caller whose collective runs on its own stream can make THIS stream wait for the multipole
 * all-gather (hipStreamWaitEvent before calling _finish_rest) instead of the compute stream, so that the near-field
 * lists and the pair kernel do not wait for the multipoles either. */
int nbco_aux_stream(nbco_ctx *c, void **stream_out);

/* ---- per-phase device timing (HIP events on the context's stream) ---------------------------- */
enum {
	NBCO_PH_BUILD = 0, NBCO_PH_P2M_M2M = 1, NBCO_PH_TRAVERSE = 2, NBCO_PH_LISTS = 3, NBCO_PH_P2P = 4,
	NBCO_PH_M2L = 5, NBCO_PH_L2L = 6, NBCO_PH_L2P = 7, NBCO_PH_FINISH = 8, NBCO_PH_DIRECT = 9,
	NBCO_PH_AXPY = 10, NBCO_PH_COUNT = 11
};
int nbco_profile_enable(nbco_ctx *c, int mask);     /* bit i: record a pair of events around phase i; -1 = all, 0 = off */
int nbco_profile_reset(nbco_ctx *c);
int nbco_profile_get(nbco_ctx *c, int phase, double *total_ms, long long *launches);  /* syncs */

/* ---- checked build (libnbco_hip_checked.so, compiled with -DNBCO_CHECKED) -----------------------------------------
 * Same ABI; every index a kernel reads from a list (traversal frontier, pair lists, sorted entries, source descriptors,
 * work units) is range-checked on the device, counted per site and replaced by a harmless one instead of being used as
 * an address.  out8[site] = violations since the library was loaded (sites: 0 frontier, 1 list fill, 2 list sort,
 * 3 work unit, 4 source descriptor, 5 L2P); the production library returns NBCO_ERR_UNSUPPORTED.
 * NBCO_POISON=1 in the environment (either build) fills every new scratch allocation with 0x7f bytes. */
int nbco_debug_violations(nbco_ctx *c, long long *out8);

/* ---- initial condition (host only, no GPU needed) ----------------------------------------------
 * The reference's synthetic state: initGA (main3.cu:113-137) over std::mt19937_64(seed) after discard(discard)
 * (main3.cu:662-664: seed 5351550349027530206, discard 1248) -- 3n position deviates then 3n velocity deviates
 * from std::normal_distribution<float>, scaled by sigma_x / sigma_u per axis, centred and rescaled to exactly those
 * RMS values; with uniform_positions != 0 the positions are then redrawn uniformly in [-1, 1)^3 and centred, as
 * the -test mode does (initU, main3.cu:94-111,666).  host_state = [pos n x 3 | vel n x 3] floats in HOST memory. */
#define NBCO_REF_SEED 5351550349027530206ULL
#define NBCO_REF_DISCARD 1248ULL
int nbco_init_gaussian(float *host_state, long long n, const float *sigma_x3, const float *sigma_u3,
                       unsigned long long seed, unsigned long long discard, int uniform_positions);
/* Rows [first, first + count) of that state -- host_slice = [pos count x 3 | vel count x 3] -- bit for bit, without holding more
 * than the slice: the generator runs through the whole stream twice (the centring and the RMS rescaling of main3.cu:71-92 are
 * sums over all n particles).  What a rank of a multi-GPU run calls to draw its share of a system that fits no single host buffer
 * comfortably (N = 64M: 1.5 GB for the full state). */
int nbco_init_gaussian_slice(float *host_slice, long long n, long long first, long long count, const float *sigma_x3,
                             const float *sigma_u3, unsigned long long seed, unsigned long long discard, int uniform_positions);

#ifdef __cplusplus
}
#endif
#endif /* NBCO_H */
