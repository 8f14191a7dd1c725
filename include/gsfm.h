/* gsfm.h — C ABI of libgsfm: MI355X (gfx950) implementation of GLOMAP's estimator hot path.
 *
 * Each entry point replaces one C++ estimator call of the reference (colmap/glomap v1.1.0):
 *
 *   gsfm_ra_solve   <->  RotationEstimator::EstimateRotations   glomap/estimators/global_rotation_averaging.h:84-87
 *                        (called from SolveRotationAveraging, glomap/controllers/rotation_averager.cc:58,165,180,193)
 *   gsfm_gp_solve   <->  GlobalPositioner::Solve                glomap/estimators/global_positioning.h:63-68
 *                        (called from GlobalMapper::Solve, glomap/controllers/global_mapper.cc:160)
 *   gsfm_ba_solve   <->  BundleAdjuster::Solve                  glomap/estimators/bundle_adjustment.h:45-49
 *                        (called from glomap/controllers/global_mapper.cc:209,221,302,315)
 *
 * The reference walks pointer-linked std::unordered_map containers; at this boundary they are
 * flattened to structure-of-arrays with dense int32 indices.  The header-only adapter
 * include/gsfm_glomap_adapter.hpp re-creates the three reference classes on top of these calls.
 *
 * Conventions
 *   - plain C, no exceptions, no retained pointers after return; every function returns a
 *     gsfm_status (0 = ok, < 0 = error) and fills an optional gsfm_report.
 *   - `mem` says where the problem arrays (and the in/out state arrays) live:
 *     GSFM_MEM_HOST (the drop-in case: the library does H2D at entry, D2H at exit) or
 *     GSFM_MEM_DEVICE (arrays already resident in HBM of the ctx's device; used by bench.py).
 *   - all floating point is IEEE double, as in the reference.
 *   - one ctx drives one GPU from one host thread (thread-compatible, not re-entrant per ctx).
 *     Multi-GPU = one process (ctx) per GPU; the track / edge shards are combined with RCCL
 *     all-reduce on the reduced-system vectors (gsfm_comm_*).
 */
#ifndef GSFM_H_
#define GSFM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSFM_VERSION 100

typedef enum {
  GSFM_OK = 0,
  GSFM_ERR_INVALID_ARGUMENT = -1,
  GSFM_ERR_HIP = -2,           /* a HIP runtime call failed; report.hip_error holds hipError_t */
  GSFM_ERR_NO_DEVICE = -3,
  GSFM_ERR_NUMERICAL = -4,     /* NaN step / NaN weight: reference returns false (gra.cc:508-512,590-593) */
  GSFM_ERR_EMPTY_PROBLEM = -5, /* reference returns false (gp.cc:37-50, ba.cc:17-24) */
  GSFM_ERR_NOT_USABLE = -6,    /* !summary.IsSolutionUsable() (gp.cc:92, ba.cc:105) */
  GSFM_ERR_UNSUPPORTED = -7,
  GSFM_ERR_COMM = -8           /* RCCL failure */
} gsfm_status;

typedef enum { GSFM_MEM_HOST = 0, GSFM_MEM_DEVICE = 1 } gsfm_mem;

/* Termination reasons (mirrors ceres::TerminationType where a Ceres solve is replaced). */
typedef enum {
  GSFM_TERM_CONVERGENCE = 0,
  GSFM_TERM_NO_CONVERGENCE = 1, /* iteration cap reached; solution still usable */
  GSFM_TERM_FAILURE = 2
} gsfm_termination;

typedef struct gsfm_report {
  int32_t iterations;        /* RA: L1 outer + IRLS iterations; GP/BA: LM iterations (accepted + rejected) */
  int32_t iterations_l1;     /* RA only */
  int32_t iterations_irls;   /* RA only */
  int32_t successful_steps;  /* GP/BA: accepted LM steps */
  int64_t linear_iterations; /* total PCG iterations */
  double initial_cost;
  double final_cost;
  int32_t termination;       /* gsfm_termination */
  int32_t hip_error;
  double seconds_total;      /* wall time inside the call */
  double seconds_solve;      /* device time of the solve proper (H2D/D2H excluded) */
  double last_step_norm;     /* RA: mean |delta| of the last iteration (gra.cc:758-772) */
  int32_t line_search_trials; /* GP: evaluations of the projected line search beyond the LM step itself (Ceres: Summary::num_line_search_steps) */
  int32_t line_search_shrunk; /* GP: LM iterations whose step the line search shortened */
} gsfm_report;

typedef struct gsfm_ctx gsfm_ctx;

/* ---- context ------------------------------------------------------------------------------ */
/* device_id = -1 -> current device.  Replaces colmap::SetBestCudaDevice(gpu_indices[0])
 * (gp.cc:536-541, ba.cc:79-84). */
int gsfm_ctx_create(int device_id, gsfm_ctx** out);
void gsfm_ctx_destroy(gsfm_ctx* ctx);
int gsfm_version(void);
const char* gsfm_status_string(int status);
/* HIP stream (hipStream_t) all work of this ctx is enqueued on; callers that time the solve with
 * HIP events must record on this stream. */
void* gsfm_ctx_stream(gsfm_ctx* ctx);
/* Device name + gcnArchName, for reports. Returns GSFM_OK and writes a NUL-terminated string. */
int gsfm_ctx_device_name(gsfm_ctx* ctx, char* buf, size_t buflen);

/* ---- device memory ---------------------------------------------------------------------------
 * libgsfm owns exactly one HIP runtime per process (the ROCm one it is linked against).  Callers
 * that want problem arrays resident in HBM (GSFM_MEM_DEVICE) allocate and fill them through these
 * calls, so no second HIP runtime (e.g. the copy bundled inside a PyTorch wheel) ever has to share
 * pointers with this one. */
int gsfm_device_alloc(gsfm_ctx* ctx, size_t bytes, void** out);
int gsfm_device_free(gsfm_ctx* ctx, void* ptr);
int gsfm_memcpy_h2d(gsfm_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int gsfm_memcpy_d2h(gsfm_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int gsfm_memcpy_d2d(gsfm_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
/* Blocks until all work enqueued on the ctx stream has finished (hipStreamSynchronize). */
int gsfm_ctx_synchronize(gsfm_ctx* ctx);

/* ---- per-kernel timing (HIP events on the ctx stream) -------------------------------------
 * When enabled, every launch of the dominant kernel of each solver is bracketed by a HIP event
 * pair on the ctx stream; bench.py uses this for the roofline line.  Off by default (the event
 * records cost host time), never needed for correctness. */
enum {
  GSFM_KERNEL_RA_LAPLACIAN = 0, /* RA: weighted-Laplacian SpMV (3 RHS) inside the PCG */
  GSFM_KERNEL_GP_SCHUR = 1,     /* k_gp_phaseA: implicit Schur product, track-major half (BATA blocks) */
  GSFM_KERNEL_BA_SCHUR = 2,     /* k_ba_phaseA: implicit Schur product, track-major half (stored Jacobian planes) */
  GSFM_KERNEL_GP_SCHUR_B = 3,   /* k_gp_phaseB: camera-major half */
  GSFM_KERNEL_BA_SCHUR_B = 4,   /* k_ba_phaseB: camera-major half */
  GSFM_KERNEL_RA_GJ = 5,        /* k_dense_gj_step: one block Gauss-Jordan step of the dense RA inverse (f64 MFMA) */
  GSFM_KERNEL_FILTER_OBS = 6,   /* k_filter_obs: per-observation reprojection / angle test of the track filters */
  GSFM_KERNEL_TRACK_HOOK = 7,   /* k_uf_hook: union-find hooking sweep over the inlier matches (track establishment) */
  GSFM_KERNEL_GP_WSUM = 8,      /* k_gp_wsum: per-camera fixed-order sum of the chunked sweep's piece partials (second half of the camera side) */
  GSFM_KERNEL_COUNT = 9
};
int gsfm_ctx_profile_enable(gsfm_ctx* ctx, int enable);
/* Reads and resets the accumulated launch count / total milliseconds of one kernel id. */
int gsfm_ctx_profile_read(gsfm_ctx* ctx, int kernel_id, int64_t* launches, double* total_ms);
/* Which solver paths ran on this ctx (counters since creation or the last reset): lets a test assert that a solve took
 * the path it was meant to cover (deflated multi-block PCG, closed-form gauge products, joint pose / intrinsics blocks,
 * second-level preconditioner, collectives). */
enum gsfm_stat {
  GSFM_STAT_PCG_SOLVES = 0,          /* reduced-system solves (cg_solve calls) */
  GSFM_STAT_PCG_DEFLATED = 1,        /* ... with the gauge modes deflated */
  GSFM_STAT_PCG_CLOSED_FORM_AW = 2,  /* ... whose A W products came from k_gp_aw_modes / k_ba_aw_modes (no operator applications) */
  GSFM_STAT_PCG_SINGLE_WORKGROUP = 3,/* ... run by the single-workgroup vector kernels (small GP problems) */
  GSFM_STAT_PCG_JOINT_BLOCKS = 4,    /* ... with joint pose + intrinsics blocks (BA, one intrinsics block per camera) */
  GSFM_STAT_PCG_SECOND_LEVEL = 5,    /* ... with the second-level (cluster) preconditioner */
  GSFM_STAT_ALLREDUCES = 6,          /* collectives issued (any transport) */
  GSFM_STAT_PCG_ITERATIONS = 7,      /* PCG iterations (operator applications of the iterations proper) */
  GSFM_STAT_PCG_CHUNKED_SWEEPS = 8,  /* reduced-system solves whose camera-side sweep ran in the chunked (XCD-partitioned) order */
  GSFM_STAT_PCG_RECYCLED = 9,        /* ... preconditioned with recycled Ritz vectors of earlier solves (cg.hpp CgRecycle) */
  GSFM_STAT_RITZ_HARVESTED = 10,     /* Ritz vectors harvested from reduced-system solves */
  GSFM_STAT_DENSE_SOLVES = 11,       /* GP / BA reduced systems assembled densely and solved directly (small, chain-like problems) */
  GSFM_STAT_COUNT = 12
};
/* Copies min(n, GSFM_STAT_COUNT) counters to out; reset != 0 zeroes them afterwards. */
int gsfm_ctx_stats(gsfm_ctx* ctx, int64_t* out, int n, int reset);
/* The Levenberg-Marquardt iterations of the last gsfm_gp_solve / gsfm_ba_solve on this ctx, what Ceres prints with
 * minimizer_progress_to_stdout (optimization_base.h:21): one row of GSFM_LM_TRACE_COLS doubles per iteration —
 *   cost before the step | trust-region radius | model cost change | cost at the candidate | line-search step size
 *   (1 when no search ran or it kept the step, < 1 when it shortened it, -1 when it failed) | 1 accepted, 0 rejected, -1 invalid
 *   step | linear (PCG) iterations of the step.
 * Copies min(rows, max_rows) rows to out (may be NULL) and returns the number of rows recorded.  Tests compare trajectories
 * with it, iteration by iteration. */
#define GSFM_LM_TRACE_COLS 7
int gsfm_ctx_lm_trace(gsfm_ctx* ctx, double* out, int32_t max_rows);
/* Diagnostic / A-B knobs of a context, for tests and measurements (0 = what the library ships).  The library itself reads no
 * environment variable for any of them: a caller that wants to compare two variants says so through this call. */
enum gsfm_knob {
  GSFM_KNOB_BA_AW_BY_APPLICATION = 0, /* BA: form the gauge products A W by operator applications, not in closed form */
  GSFM_KNOB_BA_AW_CHECK = 1,          /* BA: form them both ways and print the difference (stderr) */
  GSFM_KNOB_BA_SEPARATE_BLOCKS = 2,   /* BA: 6x6 + 8x8 block-Jacobi instead of the joint pose + intrinsics blocks */
  GSFM_KNOB_BA_NO_NONTEMPORAL = 3,    /* BA phase A: plain instead of non-temporal plane loads */
  GSFM_KNOB_RA_NO_BLOCKDENSE = 4,     /* RA, 2048 < N <= 32768: Jacobi-PCG instead of the dense block preconditioners */
  GSFM_KNOB_RA_NO_SUBSTRUCTURE = 5,   /* RA: plain block-diagonal instead of the substructured preconditioner */
  GSFM_KNOB_RA_DENSE_REFACTOR = 6,    /* RA, N <= 2048: re-invert at every IRLS iteration */
  GSFM_KNOB_GP_COARSE_CLUSTER = 7,    /* GP second level: cameras per cluster (0: 32) */
  GSFM_KNOB_SEG_LEN = 8,              /* camera-major order: observations per camera segment (0: 1024) */
  GSFM_KNOB_CHUNKED_SWEEPS = 9,       /* camera-side sweeps of the PCG in the chunked order: 0 = when the point records exceed the
                                         L2 (default), 1 = always, 2 = never (plain camera-major order),
                                         >= 8 = always, with that many point chunks (A/B runs) */
  GSFM_KNOB_EXPERIMENT = 10,          /* bit mask of kernel variants kept for A/B measurements (tools/ab_gp_sweeps.py); 0 = shipped.
                                         2: k_gp_phaseA reads its tile streams with plain instead of non-temporal loads */
  GSFM_KNOB_GP_NO_RECYCLE = 11,       /* GP: no recycled Ritz vectors in the reduced-system preconditioner */
  GSFM_KNOB_GP_RECYCLE_MIN_ITERS = 12, /* GP: harvest Ritz vectors from solves of at least this many iterations (0: 25) — tests
                                          lower it so that small problems, whose solves are short, exercise the path */
  GSFM_KNOB_GP_RECYCLE_CUT_PERCENT = 13, /* GP: harvest Ritz values below this many hundredths (0: 30) — tests raise it for the same reason */
  GSFM_KNOB_GP_DENSE = 14,            /* GP (at most 1024 cameras) and BA (at most 6144 reduced unknowns, 16 intrinsics blocks): 0 = assemble and invert the reduced system densely once a PCG solve of the LM
                                       * problem ran past 100 iterations; 1 = never; 2 = always (every solve, A/B and tests) */
  GSFM_KNOB_COUNT = 15
};
int gsfm_ctx_set_knob(gsfm_ctx* ctx, int knob, int value);
/* Text of the last failure on this ctx (what the HIP / RCCL call or the argument check said); "" when there was none.  The
 * pointer stays valid until the next failing call on the ctx. */
const char* gsfm_ctx_last_error(gsfm_ctx* ctx);

/* Flat on-disk problem format.  With a directory set — here, or through GSFM_DUMP_DIR in the
 * environment when the ctx is created — every gsfm_{ra,gp,ba}_solve on this ctx writes <directory>/<kind>_<seq>.gsfm:
 * the flat problem as it crossed this ABI, the options, the results and the report (layout: glomap_amd/csrc/dump.hpp,
 * reader: glomap_amd/flatio.py, replay: tools/replay.py).  NULL or "" disables. */
int gsfm_ctx_set_dump_dir(gsfm_ctx* ctx, const char* directory);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) ------------------------------------- */
#define GSFM_COMM_ID_BYTES 128
/* Rank 0 calls gsfm_comm_unique_id and broadcasts the bytes out of band (e.g. through
 * torch.distributed); every rank then calls gsfm_comm_init.  After that every gsfm_*_solve on
 * this ctx treats its problem as ONE SHARD (a subset of edges / tracks over the full, replicated
 * node / camera arrays) and all-reduces the reduced-system vectors. */
int gsfm_comm_unique_id(char id[GSFM_COMM_ID_BYTES]);
int gsfm_comm_init(gsfm_ctx* ctx, const char id[GSFM_COMM_ID_BYTES], int rank, int world_size);
int gsfm_comm_destroy(gsfm_ctx* ctx);
/* One small RCCL all-reduce (sum of 1 + i over the ranks) on the ctx stream, checked: a cheap probe that communicator,
 * stream and device memory work together in this process (world_size 1 included).  *sum_out = world_size. */
int gsfm_comm_selftest(gsfm_ctx* ctx, double* sum_out);
/* Host-only self test of the generator behind global positioning's random start (global_positioning.cc:135,261,449:
 * std::mt19937 + std::uniform_real_distribution<double>(-1, 1)): after skipping `skip` 32-bit outputs, `count` values
 * scale * U(-1, 1) from the library's block generator into out_fast and from the C++ standard library into out_std.  The
 * two must be bit-identical.  Needs no GPU. */
int gsfm_selftest_mt19937(uint32_t seed, uint64_t skip, int64_t count, double scale, double* out_fast, double* out_std);
/* Host-staged transport for validation only (several ranks sharing ONE device, or a box without
 * xGMI): every collective becomes D2H, fn(buf, n, op, user) — which must all-reduce buf in place
 * across the ranks (op 0 = sum, 1 = max) and return 0 — and H2D.  Same sharding semantics as the
 * RCCL transport; never the measured path. */
typedef int (*gsfm_host_allreduce_fn)(double* buf, int64_t n, int op, void* user);
int gsfm_comm_init_host(gsfm_ctx* ctx, gsfm_host_allreduce_fn fn, void* user, int rank, int world_size);

/* Peer-mailbox transport: a one-shot all-reduce for the small per-iteration vectors of the sharded solves (csrc/peer.hpp).
 * Every rank owns a mailbox in its HBM with one slot per rank; the ranks map each other's mailboxes (hipIpc*: peer access
 * over xGMI between the GPUs of a node, plain device memory between processes that share one GPU), push their vector into
 * every mailbox, and add the slots of their own mailbox in rank order — two launches, no ring, the same bits on every rank.
 *   every rank:  gsfm_comm_peer_open(ctx, rank, world, capacity, handle)      (at most 8 ranks; one node)
 *   out of band: all-gather the GSFM_PEER_HANDLE_BYTES of every rank, in rank order
 *   every rank:  gsfm_comm_peer_connect(ctx, all_handles)
 * From then on every collective of this ctx goes through the mailboxes (vectors longer than `capacity_doubles` in pieces);
 * an RCCL communicator or host transport attached before is released.  A rank that does not arrive within
 * GSFM_PEER_TIMEOUT_S seconds (environment, default 60) makes the next collective fail with GSFM_ERR_COMM.
 * gsfm_comm_destroy releases it; all ranks must have finished their solves before any of them does. */
#define GSFM_PEER_HANDLE_BYTES 64
int gsfm_comm_peer_open(gsfm_ctx* ctx, int rank, int world_size, int64_t capacity_doubles, char handle_out[GSFM_PEER_HANDLE_BYTES]);
int gsfm_comm_peer_connect(gsfm_ctx* ctx, const char* all_handles /* world_size x GSFM_PEER_HANDLE_BYTES */);
/* Checked sum / max all-reduces through the mailboxes (both slot sets); *world_out = world_size.  Collective. */
int gsfm_comm_peer_selftest(gsfm_ctx* ctx, double* world_out);
/* `repeats` back-to-back all-reduces of n doubles through the attached transport (RCCL, peer mailboxes or host-staged), HIP
 * events around the train: *avg_us_out = microseconds per collective as a PCG loop pays it.  Collective. */
int gsfm_comm_allreduce_bench(gsfm_ctx* ctx, int64_t n, int repeats, double* avg_us_out);

/* ---- rotation averaging ------------------------------------------------------------------- */
/* Options: mirror of RotationEstimatorOptions, global_rotation_averaging.h:39-75. */
typedef struct gsfm_ra_options {
  int32_t max_num_l1_iterations;           /* 5 */
  double l1_step_convergence_threshold;    /* 1e-3 */
  int32_t max_num_irls_iterations;         /* 100 */
  double irls_step_convergence_threshold;  /* 1e-3 */
  double irls_loss_parameter_sigma;        /* 5.0 degrees */
  int32_t weight_type;                     /* 0 = GEMAN_MCCLURE, 1 = HALF_NORM */
  int32_t skip_initialization;             /* 0: maximum-spanning-tree init (gra.cc:87-138) */
  int32_t use_weight;                      /* 0 */
  int32_t use_gravity;                     /* 0; 1: frames flagged in gsfm_ra_problem.node_gravity have a 1-DoF rotation */
  /* colmap::LeastAbsoluteDeviationSolver::Options as set at gra.cc:483-486 */
  int32_t l1_admm_max_num_iterations;      /* 10 */
  double l1_admm_rho;                      /* 1.0 */
  double l1_admm_alpha;                    /* 1.0 */
  double l1_admm_absolute_tolerance;       /* 1e-4 */
  double l1_admm_relative_tolerance;       /* 1e-2 */
  /* linear solver replacing CHOLMOD (gra.cc:547-611): dense tiled Gauss-Jordan inverse on the matrix
   * cores for num_nodes <= 2048 (a direct solve, like the reference); PCG preconditioned by dense diagonal
   * blocks of the BFS-relabelled graph up to 32768 nodes; Jacobi-PCG on the weighted Laplacian above
   * that, when sharded over ranks, or when force_iterative is set */
  double pcg_relative_tolerance;           /* 1e-10: |r|_2 <= tol * |b|_2 per right-hand side */
  int32_t pcg_max_iterations;              /* 2000 */
  int32_t force_iterative;                 /* 0 */
  double pcg_relative_tolerance_admm;      /* 1e-10 on the residual of the warm start: x-updates INSIDE the ADMM loop of the L1 stage
                                              (warm-started corrections).  Looser values are a parity hazard that grows with the
                                              condition number: 1e-3 moved the final rotations by 1.3e-3 rad at 2 500 nodes, 1e-6
                                              is exact to 4e-8 rad there but off by radians on a 20 000-node ring with the Jacobi
                                              preconditioner (tools/exp_ra_bd_check.py, exp_ra_save.py); the tight value costs < 3 %
                                              more PCG iterations */
} gsfm_ra_options;

void gsfm_ra_options_default(gsfm_ra_options* opt);

/* View graph, valid edges only (ImagePair::is_valid, image_pair.h:32), largest connected
 * component already kept (rotation_averager.cc:13).  Node = registered frame with trivial rig. */
typedef struct gsfm_ra_problem {
  int32_t mem;              /* gsfm_mem for every pointer below and for rot_aa_inout */
  int32_t num_nodes;        /* N */
  int64_t num_edges;        /* E (this rank's shard when a comm is attached) */
  const int32_t* edge_i;    /* [E] node of image_id1 */
  const int32_t* edge_j;    /* [E] node of image_id2 */
  const double* edge_q;     /* [E][4] (w,x,y,z) cam2_from_cam1.rotation: x_2 = R x_1 */
  const double* edge_weight;/* [E] ImagePair::weight, may be NULL unless use_weight */
  const int32_t* edge_ninl; /* [E] inliers.size() (MST weight, tree.cc:99,124); may be NULL if skip_initialization */
  int32_t fixed_node;       /* gauge node (reference: first registered frame in map order, gra.cc:248-257) */
  /* Rigs with cam_from_rig ROTATIONS among the unknowns (global_rotation_averaging.cc:173-191, 396-446, 646-693, 718-739).
   * num_images = 0 (and NULL pointers): every node is a frame, as above.  num_images = I > 0: edge_i / edge_j index
   * IMAGES, num_nodes / rot_aa_inout / fixed_node are the N FRAMES, and every image carries
   *   image_frame[i]  its frame,
   *   image_cam[i]    the block (0..num_cams-1) of its sensor's cam_from_rig rotation, or -1 for the reference sensor and
   *                   for calibrated sensors, whose cam_from_rig the caller folds into edge_q (gra.cc:306-309).
   * cam_from_world(i) = Exp(cam) Exp(frame); an edge's rows carry -1 / +1 at the frame AND cam columns of its two
   * images; the cam blocks are updated by the quaternion average of gra.cc:676-690.  cam_rot_aa [C][3] (host memory) is
   * in/out like rot_aa_inout.  Needs skip_initialization (the spanning-tree start and ConvertRotationsFromImageToRig,
   * rotation_initializer.cc:7-125, run on the image-level graph first: see the adapter), one rank, use_gravity = 0. */
  int32_t num_images;
  const int32_t* image_frame; /* [I] */
  const int32_t* image_cam;   /* [I] */
  int32_t num_cams;           /* C */
  double* cam_rot_aa;         /* [C][3] host, in/out */
  /* Gravity-aligned frames (use_gravity, global_rotation_averaging.cc:19-36, 207-217, 312-341, 376-418, 455-460,
   * 639-645, 709-713, 746-749).  node_gravity [N] (NULL = no frame has gravity): 1 = frame.HasGravity(); such a frame
   * has ONE unknown, the angle about the aligned vertical, carried in rot_aa_inout as (0, angle, 0)
   * [AngleToRotUp(angle) = Exp((0, angle, 0)), math/gravity.cc:30-33; in: RotUpToAngle(R_align^T R_rig_from_world),
   * gra.cc:207-211; out: the caller writes R_align * AngleToRotUp(angle) back, gra.cc:786-793].  edge_q must already be
   * aligned, R_align2^T R_rel R_align1 for the images that have gravity (gra.cc:315-327).  A pair of two gravity
   * frames is one row (RelAngleError of the y components; its x / z components only enter the IRLS weight), the gauge
   * is one row when the fixed frame has gravity.  No spanning-tree start in this mode (gra.cc:60-62).  The rand()
   * jitter RelAngleError adds within 0.01 rad of +-pi is not reproduced. */
  const uint8_t* node_gravity;
} gsfm_ra_problem;

/* rot_aa_inout: [N][3] angle-axis of rig_from_world; in = initial estimate, out = result
 * (reference writes frame.RigFromWorld = (quat(Exp(r)), t = 0), gra.cc:774-816). */
int gsfm_ra_solve(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt,
                  double* rot_aa_inout, gsfm_report* report);

/* Building blocks exposed for parity tests and roofline micro-benchmarks (same kernels the
 * solver uses).  residual_out [E][3] = -Log(R_j^T R_rel R_i) (gra.cc:715-742), weight_out [E] =
 * IRLS weight (gra.cc:583-588).  Either output may be NULL. */
int gsfm_ra_residuals(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt,
                      const double* rot_aa, double* residual_out, double* weight_out);
/* `repeat` launches of the per-edge residual / IRLS-weight sweep (the kernel behind gsfm_ra_residuals) for timing;
 * reports the average kernel milliseconds (HIP events on the ctx stream). */
int gsfm_ra_residuals_timed(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt,
                            const double* rot_aa, int repeat, double* avg_kernel_ms);
/* y = (L_w (x) I3 + gauge) x for the IRLS-weighted Laplacian; x,y [N][3]; w [E] edge weights.
 * `repeat` launches of the SpMV kernel (for timing); reports average kernel milliseconds. */
int gsfm_ra_laplacian_apply(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const double* w,
                            const double* x, double* y, int repeat, double* avg_kernel_ms);

/* ---- global positioning ------------------------------------------------------------------ */
/* Options: mirror of GlobalPositionerOptions (global_positioning.h:9-54) + the Ceres options the
 * reference sets (optimization_base.h:18-23) + the Ceres defaults it relies on. */
typedef struct gsfm_lm_options {
  int32_t max_num_iterations;        /* GP 100, BA 200 */
  double function_tolerance;         /* 1e-5 */
  double gradient_tolerance;         /* 1e-10 */
  double parameter_tolerance;        /* 1e-8 */
  double initial_trust_region_radius;/* 1e4 */
  double max_trust_region_radius;    /* 1e16 */
  double min_trust_region_radius;    /* 1e-32 */
  double min_relative_decrease;      /* 1e-3 */
  double min_lm_diagonal;            /* 1e-6 */
  double max_lm_diagonal;            /* 1e32 */
  int32_t jacobi_scaling;            /* 1 */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  /* linear solver replacing SPARSE_SCHUR + sparse Cholesky: implicit-Schur block-Jacobi PCG */
  double pcg_relative_tolerance;     /* |r|_2 <= tol * |b|_2 on the reduced camera system: 1e-10 (gsfm_gp_options_default), 1e-6 (gsfm_ba_options_default) */
  int32_t pcg_max_iterations;        /* 1000 */
  int32_t max_num_line_search_step_size_iterations; /* 20 (Ceres' Solver::Options default).  Used on bounds-constrained problems only —
                                        global positioning, whose scales carry a lower bound (gp.cc:204,373): Ceres then runs a projected
                                        Armijo line search on every valid LM step (trust_region_minimizer.cc DoLineSearch).  0 switches it
                                        off, as in Ceres.  Occupies what was tail padding until round 6: a zero-initialised struct that
                                        never went through gsfm_*_options_default runs WITHOUT the search. */
} gsfm_lm_options;

typedef struct gsfm_gp_options {
  gsfm_lm_options lm;
  double thres_loss_function;     /* Huber 0.1 (global_positioning.h:47-49) */
  int32_t generate_random_positions; /* 1 */
  int32_t generate_random_points;    /* 1 */
  int32_t generate_scales;           /* 1: scales start at 1 (gp.cc:298-305) */
  int32_t optimize_positions;        /* 1 */
  int32_t optimize_points;           /* 1 */
  int32_t optimize_scales;           /* 1 */
  int32_t min_num_view_per_track;    /* 3 */
  uint32_t seed;                     /* 1 */
  int32_t constraint_type;           /* GlobalPositionerOptions::ConstraintType (global_positioning.h:11-20): 0 = ONLY_POINTS
                                        (the only mode `mapper` accepts, gm.cc:145-149), 1 = ONLY_CAMERAS,
                                        2 = POINTS_AND_CAMERAS_BALANCED, 3 = POINTS_AND_CAMERAS; 1-3 need gsfm_gp_problem.num_pairs > 0,
                                        trivial frames and one rank */
  double constraint_reweight_scale;  /* 1.0; POINTS_AND_CAMERAS_BALANCED only: the point-to-camera losses are scaled by
                                        constraint_reweight_scale * num_pairs / num_pts (gp.cc:223-255) */
  int32_t rand_vector_order;         /* Which coordinate of a random start vector takes the FIRST of its three draws.  The reference
                                        writes Eigen::Vector3d(dist(gen), dist(gen), dist(gen)) (gp.cc:12-19): C++ leaves the evaluation
                                        order of the three arguments unspecified, and compilers differ — clang evaluates left to right
                                        (first draw -> x), g++ right to left (first draw -> z).  0 (default) = x, y, z; 1 = z, y, x.
                                        Found in round 5 by compiling the reference's own builder with g++ (oracle/_ref,
                                        tests/test_oracle_ref.py); the C++ adapter picks the order of the compiler it is built with. */
} gsfm_gp_options;

void gsfm_gp_options_default(gsfm_gp_options* opt);

/* Tracks. Observations are track-major: track p owns observations
 * [pt_offset[p], pt_offset[p+1]).  Only observations of registered images with finite rays are
 * passed (gp.cc:279-292); tracks shorter than min_num_view_per_track are skipped by the library
 * (gp.cc:258) and their xyz left untouched. */
typedef struct gsfm_gp_problem {
  int32_t mem;
  int32_t num_cams;             /* N frames */
  int64_t num_pts;              /* P tracks (this rank's shard when a comm is attached) */
  int64_t num_obs;              /* M */
  const int64_t* pt_offset;     /* [P+1] */
  const int32_t* obs_cam;       /* [M] frame index */
  const double* obs_dir;        /* [M][3] R_cw^T * features_undist (gp.cc:294-296) */
  const uint8_t* obs_calibrated;/* [M] cameras[..].has_prior_focal_length (gp.cc:313-316); NULL = all 1 */
  /* Calibrated (known) multi-camera rigs — RigBATAPairwiseDirectionError, cost_function.h:49-82, added at
   * global_positioning.cc:318-350 with the rig scale held constant at 1 (:470-478).  num_images = 0 (and NULL pointers):
   * trivial rigs, obs_cam indexes frames.  num_images = I > 0: obs_cam indexes IMAGES, num_cams / cam_center_inout are
   * the FRAMES (rig_from_world per time step), and for every image
   *   image_frame[i]   its frame,
   *   image_offset[i]  = R_cam_from_world^T * t_cam_from_rig  (translation_rig at gp.cc:329-333; zero for reference sensors),
   * so that the residual is  v - s (X - c_frame + image_offset).
   * Sensors whose cam_from_rig translation is unknown (NaN after rotation averaging) — RigUnknownBATAPairwiseDirectionError,
   * cost_function.h:90-136, added at global_positioning.cc:354-368: residual v - s (X - c_frame - R_rig^T c_s) with the
   * camera centre in rig coordinates c_s = -R_cam_from_rig^T t_cam_from_rig one 3-vector block per sensor.  num_sensors =
   * S > 0: image i of such a sensor has image_sensor[i] >= 0 (else -1), image_offset[i] = 0 and image_sensor_rot[i] =
   * R_rig_from_world of its frame (row-major).  sensor_center [S][3] (host memory, in/out) is re-drawn in [-1,1]^3 from
   * the solver's random stream after all other draws (gp.cc:442-456) and holds the estimate on return; the caller
   * converts back t = -R c (gp.cc:576-582).  Needs optimize_positions. */
  int32_t num_images;
  const int32_t* image_frame;   /* [I] */
  const double* image_offset;   /* [I][3] */
  int32_t num_sensors;
  const int32_t* image_sensor;     /* [I] */
  const double* image_sensor_rot;  /* [I][9] */
  double* sensor_center;           /* [S][3] host, in/out */
  /* Camera-to-camera constraints — BATAPairwiseDirectionError per valid image pair, AddCameraToCameraConstraints,
   * global_positioning.cc:167-210 (read when constraint_type != 0): residual pair_dir - s (c_j - c_i) with a scale of its own
   * per pair (start 1, lower bound 1e-5; the first pair's is the constant one, gp.cc:484-489) and a plain Huber loss.
   * pair_i / pair_j: frame indices of image 1 / image 2 of the pair, pair_dir = -R_cam2_from_world^T t_cam2_from_cam1
   * (gp.cc:195-197).  With ONLY_CAMERAS the tracks only mark which frames are re-drawn at random; pt_xyz is left untouched. */
  int64_t num_pairs;               /* E */
  const int32_t* pair_i;           /* [E] */
  const int32_t* pair_j;           /* [E] */
  const double* pair_dir;          /* [E][3] */
  /* Order of the random draws (HOST memory, both optional; NULL = index order).  The reference draws the random start while
   * it iterates its own containers (frames: gp.cc:128-162, tracks: gp.cc:258-264) — hash maps, whose iteration order has
   * nothing to do with capture order.  A caller that numbers frames and tracks for memory locality (sorted ids: co-visible
   * cameras next to each other, which is also what the second-level preconditioner's index clusters want) passes the
   * container order here and still gets the reference's start bit for bit: camera draws visit cam_draw_order[0],
   * cam_draw_order[1], ... (a permutation of 0 .. N-1), then the kept tracks in the order pt_draw_order[0], ... (a
   * permutation of 0 .. P-1); the constant scale (gp.cc:484-489: the first one the reference adds) is then the first
   * observation of the first kept, non-empty track of that walk.  Single rank only (a sharded problem draws in index order). */
  const int32_t* cam_draw_order;   /* [N] host */
  const int32_t* pt_draw_order;    /* [P] host */
} gsfm_gp_problem;

/* cam_center_inout [N][3]: camera centres c = -R^T t (in: used when !generate_random_positions;
 * out: result; the adapter converts back t = -R c, gp.cc:562-572).
 * pt_xyz_inout [P][3]: in: used when !generate_random_points; out: result. */
int gsfm_gp_solve(gsfm_ctx* ctx, const gsfm_gp_problem* prob, const gsfm_gp_options* opt,
                  double* cam_center_inout, double* pt_xyz_inout, gsfm_report* report);

/* ---- bundle adjustment --------------------------------------------------------------------- */
/* COLMAP camera model ids (colmap/sensor/models.h) of the supported models: every model the reference's
 * colmap::CreateCameraCostFunction dispatches on at bundle_adjustment.cc:136-139,149-152,167-170 with a perspective or
 * fisheye projection.  The nine models with at most 8 parameters live in 8-wide intrinsics blocks
 * (GSFM_CAMERA_MAX_PARAMS); FULL_OPENCV, THIN_PRISM_FISHEYE and RAD_TAN_THIN_PRISM_FISHEYE need the 16-wide blocks
 * (GSFM_CAMERA_MAX_PARAMS_WIDE): a problem that contains one of them passes intr_stride = 16 and [K][16] parameter arrays
 * (unused tail entries are ignored and returned unchanged), and is solved by the library's 16-wide instances. */
enum {
  GSFM_CAMERA_SIMPLE_PINHOLE = 0, /* f, cx, cy */
  GSFM_CAMERA_PINHOLE = 1,        /* fx, fy, cx, cy */
  GSFM_CAMERA_SIMPLE_RADIAL = 2,  /* f, cx, cy, k */
  GSFM_CAMERA_RADIAL = 3,         /* f, cx, cy, k1, k2 */
  GSFM_CAMERA_OPENCV = 4,         /* fx, fy, cx, cy, k1, k2, p1, p2 */
  GSFM_CAMERA_OPENCV_FISHEYE = 5, /* fx, fy, cx, cy, k1, k2, k3, k4 */
  GSFM_CAMERA_FULL_OPENCV = 6,    /* fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6 (intr_stride 16) */
  GSFM_CAMERA_FOV = 7,            /* fx, fy, cx, cy, omega */
  GSFM_CAMERA_SIMPLE_RADIAL_FISHEYE = 8, /* f, cx, cy, k */
  GSFM_CAMERA_RADIAL_FISHEYE = 9, /* f, cx, cy, k1, k2 */
  GSFM_CAMERA_THIN_PRISM_FISHEYE = 10,         /* fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, sx1, sy1 (intr_stride 16) */
  GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE = 11  /* fx, fy, cx, cy, k0 .. k5, p0, p1, s0 .. s3 (intr_stride 16) */
};
#define GSFM_CAMERA_MAX_PARAMS 8
#define GSFM_CAMERA_MAX_PARAMS_WIDE 16

/* Options: mirror of BundleAdjusterOptions (bundle_adjustment.h:12-37). */
typedef struct gsfm_ba_options {
  gsfm_lm_options lm;
  double thres_loss_function;        /* Huber 1.0 px */
  int32_t optimize_rotations;        /* 1 */
  int32_t optimize_translation;      /* 1 */
  int32_t optimize_intrinsics;       /* 1 */
  int32_t optimize_principal_point;  /* 0 */
  int32_t optimize_points;           /* 1 */
  int32_t min_num_view_per_track;    /* 3 */
  int32_t optimize_rig_poses;        /* 0: bundle_adjustment.h:15; needs the sensor tables of gsfm_ba_problem */
} gsfm_ba_options;

void gsfm_ba_options_default(gsfm_ba_options* opt);

typedef struct gsfm_ba_problem {
  int32_t mem;
  int32_t num_cams;           /* N frames (trivial rigs: one image per frame) */
  int32_t num_intr;           /* K intrinsics blocks (colmap cameras) */
  int32_t fixed_cam;          /* frame with constant q and t (ba.cc:261-266); -1 = none */
  int64_t num_pts;            /* P */
  int64_t num_obs;            /* M */
  const int64_t* pt_offset;   /* [P+1] track-major */
  const int32_t* obs_cam;     /* [M] */
  const double* obs_xy;       /* [M][2] image.features[feat] (distorted pixels, ba.cc:139) */
  const int32_t* cam_intr;    /* [N] intrinsics block of each frame's image (image.camera_id) */
  const int32_t* intr_model;  /* [K] GSFM_CAMERA_* */
  /* Calibrated (known) multi-camera rigs — colmap::RigReprojErrorConstantRigCostFunctor as added at
   * bundle_adjustment.cc:147-160 (optimize_rig_poses = false, the default): x_c = cam_from_rig * (rig_from_world * X).
   * num_images = 0: trivial rigs.  num_images = I > 0: obs_cam indexes IMAGES, cam_q / cam_t are the FRAMES'
   * rig_from_world, cam_intr is ignored and every image carries
   *   image_frame[i], image_cam_from_rig[i] = (qw,qx,qy,qz,tx,ty,tz) — identity for reference sensors —, image_intr[i].
   * Sensor blocks — colmap::RigReprojErrorCostFunctor as added at bundle_adjustment.cc:161-179 (optimize_rig_poses):
   * with num_sensors = S > 0, image i of a non-reference sensor (image_sensor[i] >= 0; -1 = HasTrivialFrame or a
   * constant entry) takes its cam_from_rig from sensor_cam_from_rig[image_sensor[i]] instead of image_cam_from_rig[i].
   * The S blocks are parameter blocks (quaternion manifold + translation, never constant: ba.cc:296-309) when
   * gsfm_ba_options.optimize_rig_poses is set and are then UPDATED IN PLACE like the other in/out arrays; otherwise they
   * are read as constants.  Always host memory (S is the number of sensors of the rigs, a handful). */
  int32_t num_images;
  const int32_t* image_frame;        /* [I] */
  const double* image_cam_from_rig;  /* [I][7] */
  const int32_t* image_intr;         /* [I] */
  int32_t num_sensors;
  const int32_t* image_sensor;       /* [I] */
  double* sensor_cam_from_rig;       /* [S][7] host, in/out */
  /* doubles per row of intr_params_inout: 0 or GSFM_CAMERA_MAX_PARAMS (8), or GSFM_CAMERA_MAX_PARAMS_WIDE (16) — required
   * when intr_model holds a model with more than 8 parameters (GSFM_ERR_UNSUPPORTED otherwise) */
  int32_t intr_stride;
} gsfm_ba_problem;

/* cam_q_inout [N][4] (w,x,y,z), cam_t_inout [N][3], pt_xyz_inout [P][3],
 * intr_params_inout [K][intr_stride] (GSFM_CAMERA_MAX_PARAMS unless the problem says 16): updated in place like the
 * reference's parameter blocks (ba.cc:143-146). */
int gsfm_ba_solve(gsfm_ctx* ctx, const gsfm_ba_problem* prob, const gsfm_ba_options* opt,
                  double* cam_q_inout, double* cam_t_inout, double* pt_xyz_inout,
                  double* intr_params_inout, gsfm_report* report);


/* ---- processors that run between the estimator calls ("next" rows 1-2) -------------
 * gsfm_filter_*   <->  TrackFilter::FilterTracksByReprojection / FilterTracksByAngle / FilterTrackTriangulationAngle
 *                      glomap/processors/track_filter.cc:7-52, 54-90, 92-127   (called global_mapper.cc:164-183,
 *                      244-271, 309-333) and RelPoseFilter::FilterRotations relpose_filter.cc:7-33 (gm.cc:94,106)
 * gsfm_normalize_reconstruction  <->  NormalizeReconstruction  glomap/processors/reconstruction_normalizer.cc:5-85
 *                      (global_mapper.cc:186,231,322)
 * The reference rewrites Track::observations / ImagePair::is_valid in place; at the flat boundary the result is
 * a keep mask per observation / track / edge (1 = stays) plus the counter the reference logs and returns. */
typedef struct gsfm_scene_view {
  int32_t mem;
  int32_t num_cams;               /* N images (trivial rigs: cam_from_world = rig_from_world) */
  int64_t num_pts;                /* P tracks */
  int64_t num_obs;                /* M */
  const int64_t* pt_offset;       /* [P+1] track-major */
  const int32_t* obs_cam;         /* [M] */
  const double* obs_undist;       /* [M][3] image.features_undist[feat]; may be NULL for pixel-space reprojection */
  const double* obs_xy;           /* [M][2] image.features[feat]; only read when in_normalized_image == 0 */
  const double* cam_q;            /* [N][4] (w,x,y,z) cam_from_world */
  const double* cam_t;            /* [N][3] */
  const double* pt_xyz;           /* [P][3] */
  const uint8_t* cam_calibrated;  /* [N] cameras[...].has_prior_focal_length; NULL = all 1 */
  int32_t num_intr;               /* pixel-space reprojection only */
  const int32_t* cam_intr;        /* [N] */
  const int32_t* intr_model;      /* [K] GSFM_CAMERA_* */
  const double* intr_params;      /* [K][GSFM_CAMERA_MAX_PARAMS], or [K][intr_stride] */
  int32_t intr_stride;            /* 0 / 8, or 16 when intr_model holds a model with more than 8 parameters */
} gsfm_scene_view;

/* obs_keep_out [M]; *tracks_changed = number of tracks that lost at least one observation. */
int gsfm_filter_tracks_by_reprojection(gsfm_ctx* ctx, const gsfm_scene_view* view, double max_reprojection_error,
                                       int in_normalized_image, uint8_t* obs_keep_out, int64_t* tracks_changed);
int gsfm_filter_tracks_by_angle(gsfm_ctx* ctx, const gsfm_scene_view* view, double max_angle_error_deg,
                                uint8_t* obs_keep_out, int64_t* tracks_changed);
/* track_keep_out [P]: 0 = the reference clears the track's observations; *tracks_removed counts them. */
int gsfm_filter_tracks_triangulation_angle(gsfm_ctx* ctx, const gsfm_scene_view* view, double min_angle_deg,
                                           uint8_t* track_keep_out, int64_t* tracks_removed);
/* UndistortImages (glomap/processors/image_undistorter.cc:7-46; called at global_mapper.cc:62,155,237,263,307,325 — before GP
 * and again after every bundle-adjustment round that moved the intrinsics): rays_out[i] [F][3] =
 * camera.CamFromImg(feat_xy[i]).value_or(Zero).homogeneous().normalized(), the unit bearing of pixel feat_xy[i] [F][2] seen
 * through intrinsics row feat_intr[i] [F] of intr_model [K] / intr_params [K][intr_stride] (0 / 8, or 16 for the models with
 * more than eight parameters).  A pixel whose iteration does not converge to finite coordinates gets (0, 0, 1), as
 * value_or(Zero) gives.  mem == GSFM_MEM_DEVICE: all five arrays in HBM, nothing crosses PCIe — the bearings for the filters
 * of the BA outer loop (global_mapper.cc:229-263) are refreshed where the intrinsics live. */
int gsfm_undistort_features(gsfm_ctx* ctx, int32_t mem, int64_t num_feat, const double* feat_xy, const int32_t* feat_intr,
                            int32_t num_intr, const int32_t* intr_model, const double* intr_params, int32_t intr_stride,
                            double* rays_out);
/* Compaction after a filter: the reference erases the dropped observations from Track::observations
 * (track_filter.cc:36-44, 75-83) or clears a whole track's list (:120-123); in the flat layout the survivors move up, in
 * order.  Observation k of track p survives when obs_keep[k] != 0 (NULL: all) and track_keep[p] != 0 (NULL: all).
 * pt_offset_inout [P+1] is rewritten; each of the num_arrays (<= 16) per-observation arrays — elem_bytes[a] bytes per
 * observation, a multiple of 4: obs_cam 4, obs_xy 16, obs_undist 24, ... — is compacted in place (entries past the new
 * count are unspecified).  mem == GSFM_MEM_DEVICE: everything stays in HBM, only *num_obs_out crosses PCIe — with the
 * filters, the normaliser and gsfm_ba_solve on device arrays this keeps the BA outer loop of global_mapper.cc:201-275
 * device-resident (tests/test_pipeline_gpu.py::test_ba_outer_loop_device_resident). */
int gsfm_tracks_compact(gsfm_ctx* ctx, int32_t mem, int64_t num_pts, int64_t num_obs, int64_t* pt_offset_inout,
                        const uint8_t* obs_keep, const uint8_t* track_keep, int32_t num_arrays, void* const* arrays_inout,
                        const int32_t* elem_bytes, int64_t* num_obs_out);
/* Robust-bounding-box similarity of the registered camera centres (extent 10, 10-90 percentiles by default);
 * cam_t and pt_xyz are transformed in place, sim3_out = {scale, tx, ty, tz} with X' = scale * X + t. */
int gsfm_normalize_reconstruction(gsfm_ctx* ctx, int32_t mem, int32_t num_cams, const uint8_t* cam_registered,
                                  const double* cam_q, double* cam_t_inout, int64_t num_pts, double* pt_xyz_inout,
                                  int32_t fixed_scale, double extent, double p0, double p1, double sim3_out[4]);
/* edge_keep_out [E]: 0 where angularDistance(R_j R_i^T, R_ij) > max_angle_deg (the reference sets is_valid = false). */
int gsfm_filter_rotations(gsfm_ctx* ctx, int32_t mem, int32_t num_nodes, const double* node_q, int64_t num_edges,
                          const int32_t* edge_i, const int32_t* edge_j, const double* edge_q, double max_angle_deg,
                          uint8_t* edge_keep_out, int64_t* num_invalid);

/* ---------------------------------------------------------------------------------------------------------
 * Producers of the GP / BA inputs.
 *
 *   TrackEngine::EstablishFullTracks       glomap/controllers/track_establishment.cc:5-152
 *   TrackEngine::FindTracksForProblem      glomap/controllers/track_establishment.cc:154-227
 *   ViewGraph::KeepLargestConnectedComponents   glomap/scene/view_graph.cc:56-97
 *
 * Images are dense indices 0..num_images-1 in ASCENDING image_id order, so that the reference's global feature id
 * (image_id << 32 | feature) and (index << 32 | feature) order identically.  What the reference leaves to
 * hash-table iteration order is fixed canonically: a track's id is its smallest global
 * feature id, its observations are ascending (image, feature), the full set is ascending by id, the selected set
 * is in selection order (descending (length, id)).  All results are integers: bit-exact against oracle/tracks.py.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct gsfm_match_graph {
  int32_t mem;
  int32_t num_images;
  const int64_t* feat_offset;   /* [num_images+1] prefix sums of image.features.size() */
  const double* feat_xy;        /* [F][2] image.features (pixels), F = feat_offset[num_images] < 2^31 */
  int64_t num_pairs;
  const int32_t* pair_image1;   /* [num_pairs] ImagePair::image_id1 as dense index */
  const int32_t* pair_image2;   /* [num_pairs] */
  const uint8_t* pair_valid;    /* [num_pairs] ImagePair::is_valid; NULL = all valid */
  const int64_t* pair_offset;   /* [num_pairs+1] into match_feat*: the rows of `matches` listed in `inliers` */
  const uint32_t* match_feat1;  /* [num_matches] matches(inliers[i], 0) */
  const uint32_t* match_feat2;  /* [num_matches] matches(inliers[i], 1) */
} gsfm_match_graph;

/* Mirror of TrackEstablishmentOptions (track_establishment.h:9-24).  The int options are compared against
 * size_t / uint64 quantities as C++ does: a negative value acts as 2^64 - |x| (min_num_tracks_per_view = -1,
 * the default, means "no per-view cap"). */
typedef struct gsfm_track_options {
  double thres_inconsistency;        /* 10. */
  int32_t min_num_tracks_per_view;   /* -1 */
  int32_t min_num_view_per_track;    /* 3 */
  int32_t max_num_view_per_track;    /* 100 */
  int32_t max_num_tracks;            /* 10000000 */
} gsfm_track_options;
void gsfm_track_options_default(gsfm_track_options* o);

/* CSR set of tracks.  As an input every pointer is read; as a gsfm_tracks_fetch output the caller allocates
 * track_id[num_tracks], track_offset[num_tracks+1], obs_image[num_obs], obs_feature[num_obs]. */
typedef struct gsfm_track_set {
  int32_t mem;
  int64_t num_tracks;
  int64_t num_obs;
  int64_t* track_id;       /* image << 32 | feature of the track's smallest member */
  int64_t* track_offset;
  int32_t* obs_image;
  uint32_t* obs_feature;
} gsfm_track_set;

enum { GSFM_TRACKS_FULL = 0, GSFM_TRACKS_SELECTED = 1 };

/* EstablishFullTracks.  The result stays in HBM inside the ctx (GSFM_TRACKS_FULL); *num_tracks counts every
 * track including the discarded ones, which keep a zero-length slot like the reference's empty Track objects. */
int gsfm_tracks_establish(gsfm_ctx* ctx, const gsfm_match_graph* graph, const gsfm_track_options* opt,
                          int64_t* num_tracks, int64_t* num_obs, int64_t* num_discarded);
/* FindTracksForProblem on `full` (NULL = the ctx's GSFM_TRACKS_FULL).  image_registered: [num_images] bytes in
 * space `mem` (Image::IsRegistered()).  Result: GSFM_TRACKS_SELECTED inside the ctx. */
int gsfm_tracks_select(gsfm_ctx* ctx, const gsfm_track_set* full, int32_t num_images, const uint8_t* image_registered,
                       int32_t mem, const gsfm_track_options* opt, int64_t* num_tracks, int64_t* num_obs);
/* Copies one of the two sets held by the ctx into caller arrays (host or device per out->mem). */
int gsfm_tracks_fetch(gsfm_ctx* ctx, int32_t which, gsfm_track_set* out);
/* KeepLargestConnectedComponents over dense node (frame) indices.  edge_valid_inout [E] is cleared for edges
 * that leave the component; node_registered_out [N]; node_num_images [N] or NULL (= 1 image per frame);
 * *num_images_out = the reference's return value (0: no valid edge, nothing is written). */
int gsfm_keep_largest_connected_component(gsfm_ctx* ctx, int32_t mem, int32_t num_nodes, int64_t num_edges,
                                          const int32_t* edge_i, const int32_t* edge_j, uint8_t* edge_valid_inout,
                                          const int32_t* node_num_images, uint8_t* node_registered_out,
                                          int64_t* num_images_out);

#ifdef __cplusplus
}
#endif
#endif /* GSFM_H_ */
