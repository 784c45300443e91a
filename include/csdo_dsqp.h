/*
 * csdo_dsqp.h — C ABI of the MI355X decentralized-optimization (DO) backend for CSDO.
 *
 * Drop-in boundary for the reference's `sqp/` subsystem.  Every entry point names the reference interface it
 * replaces (paths relative to the reference tree):
 *
 *   csdo_dsqp_solve        <->  SolverDSQP::SolverDSQP(...)            sqp/dsqp_solver.h:26-34, .cc:1133-1249
 *                               + getters getSolverStatus / getMaxOfRuntimes / get_initial_static_legal and the
 *                               public members num_iterations / corridors   sqp/dsqp_solver.h:41-47
 *   csdo_dsqp_solve_batch  <->  (extension) several independent worlds in one launch; each world is exactly one
 *                               SolverDSQP construction
 *   csdo_preprocess        <->  InterpolateInitalGuess + findNeighborPairsByTrustRegion + calcEqualInterPlanes
 *                               sqp/inter_agent_cons.h:11-13,40-45,69-73; call sites csdo.cc:116-129
 *   csdo_preprocess_device <->  the same three calls with the pair search and plane generation on the device
 *   csdo_preprocess_device_batch  the same for several worlds in one call (the loop over instances of the authors' sweep,
 *                               scripts/test_through_benchmark.sh:47,91-97, is the batch axis of this backend)
 *   csdo_dsqp_estimate_work <-> (extension) the relative per-agent work estimate the launcher orders agents by; shards
 *                               a batch over GPUs by work instead of by agent count (dsqp_solver.cc:1198-1220 is the loop
 *                               that shards)
 *   csdo_front_end_plan    <->  PBS::solve over the spatiotemporal hybrid A* (the "CS" half of CSDO)
 *                               pbs/PBS.cc:28-66,665-719, hybrid_a_star/hybrid_astar.h:91-207, environment.h:128-521;
 *                               call site csdo.cc:93-110.  Host code: the search is pointer-chasing, branchy and serial
 *   csdo_validate          <->  collision_rect_and_rect / collision_circle_and_rect over a result
 *                               scripts/collision_detection.py:20-96 (the authors' post-hoc check, scripts/visualize.py:219-247)
 *   csdo_validate_frames   <->  the same per animation frame: Animation.getState + the prints of animate_func
 *                               scripts/visualize.py:181-249,256-281
 *   csdo_generate_boxes    <->  generateBox                            sqp/corridor.h:84-88, .cc:124-159
 *   csdo_vehicle_default / csdo_qp_parm_default
 *                          <->  readAgentConfig / readQpSolverConfig   common/motion_planning.cc:54-93,
 *                               sqp/utils.cc:34-59 evaluated on the shipped config.yaml
 *
 * Plain C: pointers + sizes, no C++ or torch types.  All floating point is IEEE binary64 unless noted.
 * Thread-safety: one call at a time per handle; different handles are independent.  No global mutable state.
 * Errors: functions return 0 on success or a negative CSDO_E* code; they never abort or throw.
 */
#ifndef CSDO_DSQP_H
#define CSDO_DSQP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSDO_OK 0
#define CSDO_EINVAL (-1)   /* bad argument (null pointer, Nt < 2, sizes inconsistent) */
#define CSDO_ENODEV (-2)   /* no HIP device / device error */
#define CSDO_ENOMEM (-3)   /* device or host allocation failed */
#define CSDO_ELIMIT (-4)   /* problem exceeds a compiled limit: Nt > CSDO_MAX_NT, or a world whose obstacle list does not fit the LDS beside its horizon (csdo_dsqp_last_limit) */
#define CSDO_EDEVICE (-5)  /* kernel launch / execution failed */

#define CSDO_MAX_NT 512    /* longest supported horizon (one lane per timestep, <= 512 lanes per agent) */

/* OSQP-compatible per-agent status codes reported in last_status[] (osqp/constants.h of OSQP 0.6.3). */
#define CSDO_STATUS_SOLVED 1
#define CSDO_STATUS_SOLVED_INACCURATE 2
#define CSDO_STATUS_PRIMAL_INFEASIBLE_INACCURATE 3
#define CSDO_STATUS_MAX_ITER_REACHED (-2)
#define CSDO_STATUS_PRIMAL_INFEASIBLE (-3)
#define CSDO_STATUS_NON_CVX (-7)

/* Vehicle constants: the reference keeps these as `float` statics (Constants::*, common/motion_planning.h:12-49).
 * Fields hold the float-rounded values widened to double (e.g. deltat = (double)0.706f). */
typedef struct csdo_vehicle {
  double r, deltat;
  double LF, LB, car_width, WB;
  double f2x, r2x, rv;   /* derived: motion_planning.cc:82-85 */
  double obs_radius;
} csdo_vehicle;

/* QpParm mirror (sqp/common.h:39-52) + the one restatement parameter. */
typedef struct csdo_qp_parm {
  double r_trust, max_omega, max_v, max_iter, delta_solution_threshold, max_violation;
  int32_t osqp_max_iter;
  int32_t num_interpolation;
  double dt;
  int32_t fixed_corridor;
  /* OSQP's default adaptive_rho_interval = 0 chooses the rho-update period from wall-clock timing; this backend
   * pins it (0 here means the documented default, 25). */
  int32_t adaptive_rho_interval;
  /* Round 6, no reference counterpart: refinement of every ADMM iteration's linear solve on the residual of OSQP's KKT system (formed
   * through the constraint rows, not through the reduced matrix).  The backend solves the REDUCED system (P + sigma I + A' R A) x = b by
   * block cyclic reduction, whose error is cond(H) eps |x| - about fifty times that of OSQP's LDL' of the quasi-definite KKT matrix;
   * over an SQP chain that leaves 1.3 - 1.5 times as many agents beyond 1e-4 of the exact-arithmetic iterate path as a
   * double-precision OSQP leaves (scripts/chain_parity.py against the binary128 arbiter).
   *   1: a second solve on that residual in every iteration.  The product then is CLOSER to the exact path than a double-precision
   *      OSQP (map100: 13 agents beyond 1e-4 against OSQP's 20 and the default's 30), at about 1.8 x the kernel time.
   *   2: LAGGED - the residual is formed but not solved for; it joins the next iteration's right-hand side, so that every x~ carries
   *      the correction its predecessor missed (one solve and one pass over the rows per iteration): the distance of a
   *      double-precision OSQP (map100: 19), at 1.3 - 1.45 x the time.
   * Separate kernel instantiations: the default ones are unchanged.  0 (default) or any other value: off. */
  int32_t solve_refinement;
  int32_t _reserved;
} csdo_qp_parm;

/* InterPlane (sqp/inter_agent_cons.h:47-63): c = {a_f2f,b_f2f,c_f2f, a_f2r,b_f2r,c_f2r, a_r2f,.., a_r2r,b_r2r,c_r2r} */
typedef struct csdo_plane {
  int32_t t;
  int32_t _pad;
  double c[12];
} csdo_plane;

/* One world = one SolverDSQP construction. */
typedef struct csdo_problem {
  int32_t Na, Nt;
  const double* x0_bar;        /* [Na][Nt][6]: x, y, yaw, steer, v, d_steer (OptimizeResult minus `a`) */
  const int32_t* plane_off;    /* [Na+1] CSR offsets into planes */
  const csdo_plane* planes;    /* per agent, in pair order (t ascending) */
  double dimx, dimy;
  int32_t n_obs;
  int32_t _pad;
  const double* obstacles;     /* [n_obs][3]: x, y, r in input order */
  csdo_vehicle veh;
  csdo_qp_parm parm;
  int32_t logger_level;        /* accepted for signature parity; the library prints nothing (the C++ mirror host/solver_dsqp.hpp prints a
                                  per-agent summary at level >= 2: counts, status, device time - the reference's stage timers of
                                  sqp/dsqp_solver.cc:116-120,199-202,504-508 have no counterpart in a one-launch solve but the phase-timer build) */
  int32_t _pad2;
} csdo_problem;

typedef struct csdo_result {
  double* solutions;           /* [Na][Nt][6]  x,y,yaw,steer,v,d_steer; v,d_steer at t=Nt-1 are 0 */
  double* corridors;           /* [Na][Nt][8]  xf_min,xf_max,yf_min,yf_max,xr_min,xr_max,yr_min,yr_max */
  int32_t* sqp_iters;          /* [Na]  SolverDSQP::num_iterations */
  int32_t* admm_iters;         /* [Na]  sum of ADMM iterations over the agent's QPs (new: needed for the metric) */
  int32_t* last_status;        /* [Na]  status of the agent's last QP */
  int32_t solver_status;       /* getSolverStatus(): 1 or the "worst" status by dsqp_solver.cc:1224-1237 */
  int32_t initial_static_legal;/* get_initial_static_legal() */
  double t_total;              /* seconds, host wall clock of the call (H2D + kernels + D2H) */
  double t_device;             /* seconds, device time of the solve kernels (HIP events) */
  double t_max_individual;     /* getMaxOfRuntimes(): slowest agent's device time */
  double* agent_seconds;       /* [Na] or NULL: per-agent solve time (the reference's SolutionStatistics runtimes,
                                  sqp/dsqp_solver.cc:1206-1219); device wall clock of the agent's workgroup */
} csdo_result;

typedef struct csdo_handle_s* csdo_handle;

/* Create / destroy a solver bound to one HIP device.  Device buffers persist across calls and grow on demand. */
int csdo_dsqp_create(csdo_handle* out, int device_ordinal);
void csdo_dsqp_destroy(csdo_handle h);

/* The same solver on several GPUs of one node behind ONE handle - the "GPU-count selector" of the boundary.  The reference's
 * loop over agents (sqp/dsqp_solver.cc:1198-1220) is what shards: an upload cuts the batch's agents (worlds concatenated in
 * order) into n_devices contiguous blocks of equal estimated work (csdo_dsqp_estimate_work; csdo_dsqp_shard_bounds is the
 * rule), a block may end inside a world; every device is driven by a host thread of its own through a child handle, and a
 * download scatters each block straight into the caller's result arrays - no collective: a host caller has no use for one,
 * and the agents exchange nothing once the planes are fixed.  Results are the bits of the single-device solve.
 * Work on a multi-device handle: csdo_dsqp_upload / run / run_async / wait / download / solve / solve_batch,
 * csdo_dsqp_launch_groups and csdo_dsqp_agent_groups (the children's groups one after the other), csdo_dsqp_last_*,
 * csdo_dsqp_set_min_residency_mode; the single-device entries (bridge, boxes, validator) run on the first device.
 * csdo_dsqp_run_async(h, stream): every device starts at once on its own streams, `stream` (if given) only joins them.
 * csdo_dsqp_device_solutions returns NULL for it: a collective wants one buffer per device - csdo_dsqp_multi_child(h, k)
 * is device k's handle (owned by h; do not destroy it), csdo_dsqp_multi_count(h) their number (0 for a single-device handle).
 * The same ordinal may be listed twice (two independent batches in flight on one GPU; how the tests run it on one GPU). */
int csdo_dsqp_create_multi(csdo_handle* out, const int32_t* device_ordinals, int32_t n_devices);
int32_t csdo_dsqp_multi_count(csdo_handle h);
csdo_handle csdo_dsqp_multi_child(csdo_handle h, int32_t k);
/* The sharding rule by itself (host code, no GPU): cuts[0..n_blocks] with block r = items [cuts[r], cuts[r + 1]) of near-equal
 * total weight - block r ends where the running sum first reaches (r + 1) / n_blocks of the total (the closer of the two candidate
 * cuts), every block keeps at least one item while there are enough; non-positive total: equal counts. */
int csdo_dsqp_shard_bounds(const double* weights, int32_t n_items, int32_t n_blocks, int32_t* cuts /* [n_blocks + 1] */);

/* Host-buffer entry: upload, solve on the handle's stream, download.  Replaces the SolverDSQP constructor. */
int csdo_dsqp_solve(csdo_handle h, const csdo_problem* in, csdo_result* out);

/* Several worlds in one launch (agents of all worlds become workgroups of one grid).  results[w] per world.
 * One launch reads ONE parameter block: every world of a batch must carry the same `veh` and `parm` as worlds[0]
 * (CSDO_EINVAL otherwise; solve differently-parameterised worlds in separate calls).  CSDO_ELIMIT if a world's
 * obstacle list does not fit the 160 KB of LDS beside the exchange vectors. */
int csdo_dsqp_solve_batch(csdo_handle h, const csdo_problem* worlds, int32_t n_worlds, csdo_result* results);
/* After a CSDO_ELIMIT from an upload / solve: the first world (index in the batch) and agent (index in that world) whose
 * working set fits no residency mode, and the dynamic LDS it would need (limit: 160 KB - 64 B).  One such world rejects the
 * whole batch - nothing is launched -; drop it and call again.  A world whose horizon exceeds CSDO_MAX_NT is named the same way
 * (agent 0, 0 bytes).  -1 / -1 if the last upload did not hit the limit. */
int csdo_dsqp_last_limit(csdo_handle h, int32_t* world, int32_t* agent, int64_t* lds_bytes_needed);

/* Split-phase form used by bench.py so the timed region starts with inputs resident in HBM:
 *   upload (H2D, builds the device problem) -> run (kernels only, repeatable) -> download (D2H). */
int csdo_dsqp_upload(csdo_handle h, const csdo_problem* worlds, int32_t n_worlds);
int csdo_dsqp_run(csdo_handle h, void* hip_stream /* hipStream_t or NULL for the handle's stream */);
/* The same in two halves: csdo_dsqp_run_async enqueues the solve (every launch group's kernels, forked from and joined back
 * into the stream) and returns; csdo_dsqp_wait blocks until it is done and collects the timings.  Between the two the
 * handle's batch must not be uploaded again or downloaded (CSDO_EINVAL); work the caller enqueues on the same stream in
 * between - a copy of csdo_dsqp_device_solutions, a collective - is ordered behind the solve.  No reference counterpart
 * (the reference's constructor blocks, sqp/dsqp_solver.cc:1133-1249). */
int csdo_dsqp_run_async(csdo_handle h, void* hip_stream);
int csdo_dsqp_wait(csdo_handle h);
/* A further batch in flight on the same GPU: a handle with device buffers of its own whose launches go to `parent`'s
 * streams, starting at stream `lane` (0..3; HIP serves streams from four hardware queues, and kernels that share a queue
 * run one after the other - give batches in flight different lanes).  Its kernels queue up behind the workgroups of the
 * batches launched before and take the CUs those release, so the preparation (bridge, packing, H2D) of one part of a job
 * can run under the solve of the part before it: DO phase of csdo.cc:111-148 streamed in world chunks.  The parent must
 * outlive the handle; destroy it with csdo_dsqp_destroy. */
int csdo_dsqp_create_shared(csdo_handle* out, csdo_handle parent, int32_t lane);
/* Re-points a shared handle at stream `lane` of its parent; its launch groups take lane, lane + 1, ... (modulo four).  A batch
 * with G launch groups (csdo_dsqp_launch_groups, known once it is uploaded) occupies G streams: deal the lanes out by those
 * counts.  CSDO_EINVAL for a handle that owns its streams or while a run is pending. */
int csdo_dsqp_set_lane(csdo_handle h, int32_t lane);
int csdo_dsqp_download(csdo_handle h, csdo_result* results, int32_t n_worlds);
/* Device time of the last csdo_dsqp_run in seconds (HIP events on the launch stream) and the kernel's own name. */
double csdo_dsqp_last_kernel_seconds(csdo_handle h);
/* Host seconds of the last upload / download: out[0] packing the worlds, out[1] staging into page-locked memory,
 * out[2] H2D copies (+ first-call device allocation), out[3] D2H copies, out[4] scattering into the caller's buffers. */
int csdo_dsqp_last_transfer_seconds(csdo_handle h, double out[5]);
/* How the uploaded batch is launched: agents are grouped by kernel class - workgroup size by horizon (256 threads and two
 * workgroups per CU for Nt <= 128 when the working set fits 80 KB, 512 up to Nt = 256, 768 up to 384, 1024 beyond) and LDS residency by
 * working set: 0 = exchange vectors, bounds and the third of the factor that is not in registers in LDS (per agent also
 * the inter-vehicle rows' duals / slacks, where they fit); 1 = that part of the factor read from the workspace instead
 * (512-thread class only; an agent whose obstacle list does not fit beside that layout either runs in the 768-thread class); 2 = the 768-thread class with F_r in LDS (horizons to about 350
 * beside the room set's obstacle count; chosen per agent); 3 = the 768- and 1024-thread classes: exchange vectors only.  Every group is a set of persistent
 * workgroups that take its agents off a queue ordered heaviest first, and the groups run concurrently.  `lds_bytes` may
 * be the full 80 / 160 KB of the class: what the agents do not need caches the planes' read-only coefficients.
 * Fills up to `cap` entries, returns the number of groups (or a negative error code).  `seconds` is
 * the duration of the group's kernel in the last csdo_dsqp_run.  No reference counterpart (the reference loops over
 * agents serially, sqp/dsqp_solver.cc:1198-1205). */
typedef struct csdo_launch_group {
  int32_t n_agents;
  int32_t threads;             /* per workgroup: 256, 512, 768 or 1024 */
  int32_t residency_mode;      /* 0, 1, 2 or 3, see above */
  int32_t max_nt;
  int64_t lds_bytes;
  double seconds;
} csdo_launch_group;
int32_t csdo_dsqp_launch_groups(csdo_handle h, csdo_launch_group* out, int32_t cap);
/* Which launch group every agent of the uploaded batch belongs to, in upload order (world 0's agents, world 1's, ...). */
int csdo_dsqp_agent_groups(csdo_handle h, int32_t* group_of_agent, int32_t n_agents);
/* Testing / tuning knob, from the next upload on (0 restores the automatic choice): 1 keeps the inter-vehicle rows' duals
 * and slacks in the workspace for every agent (the mode stays 0); >= 2 additionally puts the 512-thread class into mode 1.
 * >= 3 additionally keeps the 768-thread class in mode 3.  The 256- and 1024-thread classes have one mode each.  Levels 1 and 2
 * change the speed only (modes 0, 1 and 2 run the same pair-split solve: same bits); level 3 moves agents of the 768-thread class
 * from the pair-split solve to the one-lane form of mode 3, which sums a node's partials in another order: last bits differ. */
int csdo_dsqp_set_min_residency_mode(csdo_handle h, int32_t mode);
/* Where the kernels put their results, from the next upload on.  0 (default): device memory; csdo_dsqp_download copies them out
 * (57 MB for 3000 vehicles: 1 ms of PCIe behind the last kernel).  1: page-locked host memory mapped into the device's address space
 * - every workgroup writes its agent's trajectory, safe boxes and counters across PCIe when the agent is done, under the other
 * agents' iterations; csdo_dsqp_download then has nothing to copy, only to scatter into the caller's arrays.  Same values either way.
 * csdo_dsqp_device_solutions returns that memory's device address (a collective reading it reads across PCIe: keep 0 there). */
int csdo_dsqp_set_host_results(csdo_handle h, int32_t on);
/* The launcher's relative work estimate per agent (the quantity the launch order and the CU shares of the groups come
 * from: horizon, plane count and how much of the initial guess sits in tight spots), in upload order; host code, no GPU
 * needed.  For sharding a batch over GPUs by work instead of by agent count. */
int csdo_dsqp_estimate_work(const csdo_problem* worlds, int32_t n_worlds, double* est /* [sum Na] */);
/* The kernel class one agent runs in, from its horizon, the obstacle count of its world and its number of inter-vehicle planes (host
 * code, no GPU needed): out[0] threads per workgroup (256 / 512 / 768 / 1024), out[1] LDS residency mode, out[2] whether the
 * inter-vehicle rows' state fits LDS, out[3] capacity of the dense tail of the block cyclic reduction in 6x6 nodes (6, or 8 for
 * horizons of the 512-thread class that lose a level by it), out[4] LDS bytes of that working set (above the per-workgroup limit of
 * 163776: such a world is turned away with CSDO_ELIMIT).  Two items decide last bits, the rest only the speed: the tail's capacity
 * (elimination order of the last nodes) and whether the mode is 3 (one-lane form of the solve) or not (pair-split).  Both are
 * functions of the agent alone - the launcher never changes an agent's mode because of other agents in the batch -, so results do
 * not depend on how agents are batched, chunked or sharded (tests/test_gpu_multi.py: a batch that mixes modes 2 and 3). */
int csdo_dsqp_agent_class(int32_t Nt, int32_t n_obstacles, int32_t n_planes, int64_t out[5]);
/* Device pointer to the packed solutions of the last run ([sum Na][Nt_stride][6] doubles) for collectives. */
void* csdo_dsqp_device_solutions(csdo_handle h, int64_t* n_doubles);

/* Bridge: coarse front-end paths -> fixed-length initial guess + separating planes.
 *   states: concatenated [sum L_a][3] (x,y,yaw); actions: concatenated [sum (L_a-1)]; path_off[Na+1] in states.
 *   goals [Na][3].  Outputs are malloc'ed by the library and released with csdo_free. */
typedef struct csdo_bridge_out {
  int32_t Na, Nt;
  double* x0_bar;              /* [Na][Nt][6] */
  int32_t* plane_off;          /* [Na+1] */
  csdo_plane* planes;
  int32_t n_pairs;
  int32_t initial_inter_legal; /* findNeighborPairsByTrustRegion's return value */
  int32_t* pairs;              /* [n_pairs][3] = t, i, j */
} csdo_bridge_out;
int csdo_preprocess(const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                    const double* goals, const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* out);
void csdo_bridge_free(csdo_bridge_out* out);
/* The same bridge with its two O(Nt Na^2) stages - findNeighborPairsByTrustRegion and calcEqualInterPlanes,
 * sqp/inter_agent_cons.cc:12-49,54-140 - on the device (one 64-lane workgroup per (t, i), pairs emitted in the reference's
 * (t, i, j) order); interpolation and CSR assembly stay on the host.  Outputs are bit-identical to csdo_preprocess. */
int csdo_preprocess_device(csdo_handle h, const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                           const double* goals, const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* out);
/* The device bridge for n_worlds worlds in one call (every argument of csdo_preprocess_device as an array over the worlds;
 * veh and parm are shared): one copy up, the kernels of all worlds back to back, two host-device synchronisations in all
 * instead of four per world.  outs[w] equals what csdo_preprocess returns for world w, bit for bit; on an error every
 * outs[w] is released and zeroed. */
int csdo_preprocess_device_batch(csdo_handle h, int32_t n_worlds, const double* const* states, const int32_t* const* actions,
                                 const int32_t* const* path_off, const int32_t* Na, const double* const* goals,
                                 const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* outs /* [n_worlds] */);
/* The library's host threads: ONE persistent pool (started at the first call that has something to share; at most 63 threads + the
 * caller) serves the bridge, the packing of an upload and the scatter of a download, each cut into blocks.  The environment variable
 * CSDO_HOST_THREADS caps it (1: every loop runs on its caller).  A forked child starts with an empty pool. */
/* csdo_preprocess for a batch of worlds on a pool of host threads, no device work (it runs beside a solve that occupies every
 * CU: the streamed DO phase prepares its next chunk of worlds with it).  outs[w] as csdo_preprocess fills it; on error every
 * outs[w] is released and zeroed. */
int csdo_preprocess_batch(int32_t n_worlds, const double* const* states, const int32_t* const* actions,
                          const int32_t* const* path_off, const int32_t* Na, const double* const* goals,
                          const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* outs);

/* The whole DO phase of a batch of worlds in one call - what csdo.cc:111-148 does for one world: InterpolateInitalGuess,
 * findNeighborPairsByTrustRegion, calcEqualInterPlanes, SolverDSQP - coarse front-end paths in, trajectories out.
 * A job whose worlds all run in the 512-thread kernel class (horizons 129 .. 234, known from the coarse paths) is STREAMED in three
 * growing chunks of worlds on csdo_dsqp_create_shared handles that `h` keeps: chunk k + 1 is bridged on the library's host threads,
 * packed and copied under chunk k's solve, every workgroup writes its agent's results into page-locked host memory when the agent
 * is done (csdo_dsqp_set_host_results), each chunk is scattered into `results` under the later chunks' solves.  Any other job is
 * bridged at once and solved by one launch on `h` (chunks of several launch groups in flight at once fragment the CUs).  Either way
 * results[w] holds the bits csdo_preprocess + csdo_dsqp_solve return for world w alone.  The batch `h` held is replaced.
 *   results[w]: caller-allocated as for csdo_dsqp_solve_batch, with Nt = csdo_do_phase_horizon(path_off, Na, parm) of world w;
 *   initial_inter_legal [n_worlds] (may be null): findNeighborPairsByTrustRegion's return value per world;
 *   timing (may be null): host seconds, per chunk and overall.
 * One vehicle / parameter block for the whole batch.  On a csdo_dsqp_create_multi handle the WORLDS are dealt out - contiguous runs of
 * equal weight (agents x horizon), one per device - and every device runs its own DO phase on a host thread of its own (no collective:
 * worlds are independent); `timing` then describes the first device's chunks and the batch's overall times. */
typedef struct csdo_coarse_world {
  const double* states;        /* [path_off[Na]][3]: x, y, yaw of every path state, agent after agent */
  const int32_t* actions;      /* [path_off[Na] - Na]: the move between consecutive states (0..5; csdo_front_end_plan's layout) */
  const int32_t* path_off;     /* [Na + 1] */
  const double* goals;         /* [Na][3] */
  const double* obstacles;     /* [n_obs][3]: x, y, r */
  int32_t Na, n_obs;
  double dimx, dimy;
} csdo_coarse_world;
typedef struct csdo_do_phase_timing {
  double first_launch, kernels_done, total;   /* seconds since the call began */
  int32_t n_chunks, streamed;                 /* streamed = 0: one launch (chunk 0 = everything) */
  int32_t chunk_worlds[4];
  double chunk_bridge[4], chunk_upload[4], chunk_kernel[4];
} csdo_do_phase_timing;
int32_t csdo_do_phase_horizon(const int32_t* path_off, int32_t Na, const csdo_qp_parm* parm);   /* Nt of the world's fixed-length guess (sqp/inter_agent_cons.cc:320-325), or a negative error code */
int csdo_do_phase(csdo_handle h, const csdo_coarse_world* worlds, int32_t n_worlds, const csdo_vehicle* veh, const csdo_qp_parm* parm,
                  csdo_result* results /* [n_worlds] */, int32_t* initial_inter_legal, csdo_do_phase_timing* timing);
/* The chunk boundaries csdo_do_phase would use for worlds of these sizes if they are all streamable: cuts[0 .. n] = 0, c1, ..., n_worlds
 * (unused entries -1); returns the number of chunks (host code). */
int csdo_do_phase_cuts(const int32_t* agents_per_world, int32_t n_worlds, int32_t min_first_agents, int32_t* cuts /* [5] */);

/* Front end (host): priority-based search, each agent planned by a spatiotemporal hybrid A* that yields to the agents
 * ranked above it.  starts / goals [Na][3] = x, y, yaw.  On success (status 1) the paths come back in exactly the layout
 * csdo_preprocess takes: states [sum L_a][3], actions [sum (L_a - 1)] (0..5 = the six arc primitives of
 * common/motion_planning.cc:94-109, 6 = wait), path_off [Na+1].  status 0 = no solution within the limits.
 * Paths satisfy what PBS itself checks; they are not claimed to equal the reference's paths step for step (heap
 * tie-breaking and OMPL's Reeds-Shepp internals differ). */
typedef struct csdo_front_end_parm {
  double penalty_turning, penalty_reversing, penalty_cod;   /* config.yaml: 1.5, 2.0, 2.0 */
  double map_resolution;                                    /* 2.0 */
  double max_closed_set_size;                               /* per low-level search; 1e5 (environment.h:455-458) */
  double time_limit_s;                                      /* csdo.cc:100: 20 */
  int32_t node_limit;                                       /* high-level nodes; 0 = unlimited */
  uint32_t rand_seed;                                       /* csdo.cc:93: srand(0); seeds a generator owned by the call */
  /* Two rules of this front end that the reference does NOT have (environment.h:350-392 tests the map, the obstacles, the
   * higher agents at t and t +- 1 and the parked higher agents only; target conflicts are left to hasConflicts and the
   * priorities).  1 (default): a vehicle never drives through the goal rectangle of an agent that has to yield to it, and an
   * agent that is boxed in by such goals in the root node is planned alone and left to the priorities.  0: the reference's
   * rule set.  The default solves 58 / 49 of the 60 / 60 map100 / map50 obstacle instances; profiles/r03_front_end_rules.json
   * holds the counts for both settings. */
  int32_t keep_off_lower_goals;
  /* 1: the gate of the analytic shot draws from glibc's rand() sequence after srand(rand_seed) (environment.h:163, csdo.cc:93),
   * reproduced inside the call; 0 (default): from a 64-bit LCG - what the stored benchmark paths were planned with. */
  int32_t rand_glibc;
} csdo_front_end_parm;
typedef struct csdo_paths {
  int32_t Na, status;
  int32_t* path_off;
  double* states;
  int32_t* actions;
  double seconds;
  int32_t hl_expanded, hl_generated;
  int64_t ll_expanded;
} csdo_paths;
void csdo_front_end_parm_default(csdo_front_end_parm* p);
int csdo_front_end_plan(const double* starts, const double* goals, int32_t Na, double dimx, double dimy,
                        const double* obstacles /* [n_obs][3] = x, y, r */, int32_t n_obs, const csdo_vehicle* veh,
                        const csdo_front_end_parm* parm, csdo_paths* out);
void csdo_paths_free(csdo_paths* p);
/* Shortest Reeds-Shepp curve between two poses for turning radius rho (the reference takes it from OMPL's
 * ReedsSheppStateSpace, an external dependency: environment.h:165-200; this library computes it from the 1990 paper's
 * word families).  types[5]: 0 none, 1 left, 2 straight, 3 right; lengths[5] in units of rho, negative = reverse.
 * Returns the path length (rho * sum |lengths|). */
/* The first n draws of the generator behind the analytic shot's gate (csdo_front_end_parm::rand_seed, rand_glibc): with
 * rand_glibc = 1 the values rand() returns after srand(rand_seed) (hybrid_a_star/environment.h:163, csdo.cc:93). */
int csdo_front_end_gate_draws(uint32_t rand_seed, int32_t rand_glibc, int32_t n, uint32_t* out);
double csdo_reeds_shepp(const double from[3], const double to[3], double rho, int32_t types[5], double lengths[5]);

/* Independent trajectory validator on the device (the reference checks results the same way after the fact:
 * scripts/collision_detection.py:20-96 through scripts/visualize.py:40-52,219-247): vehicle rectangles (rear-axle
 * reference, LF / LB / car_width of `veh`, inflated by `margin`) against each other per timestep - separating axes,
 * touching counts - and against the obstacle discs; with dimx, dimy > 0 also rectangle corners against the map. */
typedef struct csdo_validation {
  int64_t vehicle_collisions;    /* (t, i, j) triples with overlapping rectangles */
  int64_t obstacle_collisions;   /* (t, agent, obstacle) triples */
  int64_t out_of_map;            /* (t, agent) pairs with a corner outside the map */
  int32_t first_vehicle[3];      /* smallest (t, i, j), or -1 */
  int32_t first_obstacle[3];     /* smallest (t, agent, obstacle), or -1 */
  double min_obstacle_clearance; /* smallest signed distance rectangle - disc (negative: overlap; +inf without obstacles) */
} csdo_validation;
int csdo_validate(csdo_handle h, const double* solutions /* [Na][Nt][6] */, int32_t Na, int32_t Nt,
                  const double* obstacles /* [n_obs][3] */, int32_t n_obs, double dimx, double dimy,
                  const csdo_vehicle* veh, double margin, csdo_validation* out);
/* The same check between the states: the authors' animation prints its collisions per FRAME, frames_per_move >= 1 frames per
 * move (framesPerMove, scripts/visualize.py:27,186), looking at poses interpolated linearly in x, y and yaw between two states
 * (getState, scripts/visualize.py:256-281, with its 2 pi yaw unwrapping; whole times go through the same formula).  The
 * trajectory becomes (Nt - 1) * frames_per_move + 1 frames; every index in `out` is a frame index. */
int csdo_validate_frames(csdo_handle h, const double* solutions /* [Na][Nt][6] */, int32_t Na, int32_t Nt,
                         int32_t frames_per_move, const double* obstacles /* [n_obs][3] */, int32_t n_obs, double dimx,
                         double dimy, const csdo_vehicle* veh, double margin, csdo_validation* out);

/* Safe boxes for arbitrary points on the device (one lane per point). boxes: [n][4] x_min,y_min,x_max,y_max;
 * status: [n] bit0 = success, bits 1-2 = initial status (0 legal, 1 out of map, 2 collision). */
int csdo_generate_boxes(csdo_handle h, const double* points_xy, int32_t n, const double* obstacles, int32_t n_obs,
                        double dimx, double dimy, const csdo_vehicle* veh, double* boxes, int32_t* status);

/* Diagnostic: the device program's own sin (fn 0), cos (1), tan (2) of a[i] and atan2 (3) of (a[i], b[i]) - csrc/csdo_math.h, the
 * ONE implementation every build of the program uses where the reference calls the C library's (sqp/dsqp_solver.cc:646-744,
 * 828-831, 336-339, sqp/corridor.cc:84-122) - evaluated on the device.  The same header compiled by g++ returns the same bits
 * (tests/test_shared_math.py); that is what makes the HIP build and its lane-serial host build agree to the bit over a whole
 * SQP chain.  b may be null unless fn == 3. */
int csdo_math_eval(csdo_handle h, int32_t fn, const double* a, const double* b, double* out, int32_t n);

/* The shipped config.yaml evaluated the way readAgentConfig / readQpSolverConfig do. */
void csdo_vehicle_default(csdo_vehicle* v);
void csdo_qp_parm_default(const csdo_vehicle* v, csdo_qp_parm* p);

/* Library identification: returns "hip-gfx950". */
const char* csdo_backend_name(void);
/* The first 16 hex digits of the SHA-256 over the device sources (the headers and .hip files of csrc/, this header) the library was built from
 * (csrc/Makefile: CSDO_SOURCE_HASH).  bench.py prints it, scripts/summarize_profiles.py stores it with every counter summary under
 * profiles/, and bench.py only quotes a summary's counter figures when the two agree: counters of another kernel are not evidence. */
const char* csdo_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* CSDO_DSQP_H */
