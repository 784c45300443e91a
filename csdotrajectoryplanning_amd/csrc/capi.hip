// capi.hip — the C ABI of include/csdo_dsqp.h on top of the gfx950 kernels.
// Host side only: packs the caller's buffers (batch_pack.h), owns device memory behind the handle, launches
// dsqp_agent_kernel on the handle's (or the caller's) HIP stream and measures it with HIP events.
// There is no CPU fallback: without a usable HIP device every entry point returns CSDO_ENODEV.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/csdo_dsqp.h"
#include "batch_pack.h"
#include "bridge_host.h"
#include "aux_kernels.h"
#include "dsqp_launch.h"

using namespace csdo;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap && p) return CSDO_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes < 256 ? 256 : bytes;
    if (hipMalloc(&p, want) != hipSuccess) return CSDO_ENOMEM;
    cap = want;
    return CSDO_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Page-locked host staging: H2D / D2H copies from pageable memory run at a fraction of the PCIe rate and serialise
// with the host; the packed batch is staged here once and every array goes out as one asynchronous copy.
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap && p) return CSDO_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes < 4096 ? 4096 : bytes + bytes / 8;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return CSDO_ENOMEM;
    cap = want;
    return CSDO_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// Several GPUs behind one handle (csdo_dsqp_create_multi): one child handle per device, the batch's agents cut into contiguous
// blocks of equal estimated work (the reference's loop over agents, sqp/dsqp_solver.cc:1198-1220, is what shards), every child
// driven by a host thread of its own, results scattered straight into the caller's arrays.
struct MultiPart { int world, lo, hi; };   // agents [lo, hi) of the caller's world `world`
struct MultiState {
  std::vector<csdo_handle> kids;
  std::vector<std::vector<MultiPart>> parts;               // per child: its block, world by world
  std::vector<std::vector<csdo_problem>> probs;            // ... as (sub-)problems: views into the caller's arrays
  std::vector<std::vector<std::vector<int32_t>>> offs;     // ... with the plane offsets of a cut world re-based
  std::vector<std::vector<csdo_result>> res;               // per child: views into the caller's results (download)
  std::vector<int> world_na;                               // agents per world of the uploaded batch
  int n_worlds = 0;
};

struct csdo_handle_s {
  MultiState* multi = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // launch groups (dsqp_launch.h): group 0 runs on the caller's stream, the others concurrently on side streams
  std::vector<LaunchGroup> groups;
  int min_mode = 0;            // csdo_dsqp_set_min_residency_mode
  bool host_results = false;   // csdo_dsqp_set_host_results: the kernels write results and counters into stage_down (page-locked host
                               // memory, mapped into the device's address space) instead of sol / corr / sqp / ...: nothing is left to copy
  bool results_in_stage = false;   // ... what the uploaded batch was set up with
  int n_cu = 256;              // compute units of the device: persistent workgroups per launch group
  std::vector<int32_t> order;
  std::vector<hipStream_t> side;
  std::vector<hipEvent_t> g_begin, g_end, g_zeroed, g_end2;
  HostBatch hb;
  bool uploaded = false;
  int n_worlds = 0;
  double last_kernel_s = 0.0;
  // inputs: ONE device arena filled by ONE copy (small copies are done by blit kernels, which wait for a CU while another batch's
  // persistent workgroups hold all of them: 8 ms for the upload of a streamed chunk); in_off: agents, worlds, x0, planes, tstart,
  // obstacles, order
  DevBuf in_arena;
  size_t in_off[7] = {0, 0, 0, 0, 0, 0, 0};
  DevBuf rows_ws, fac_ws, sol, corr, sqp, admm, stat, legal, ticks, queues;
  DevBuf box_pts, box_obs, box_out, box_status;
  DevBuf k0_centres, k0_counts, k0_offsets, k0_pairs, k0_coef, k0_flag, val_sol, val_obs, val_out, val_frames;
  DevBuf prof;
  PinnedBuf stage_up, stage_down;   // page-locked staging of the packed inputs / outputs
  PinnedBuf bridge_up, bridge_down; // ... of the batched device bridge (csdo_preprocess_device_batch)
  int limit_world = -1, limit_agent = -1;   // what the last CSDO_ELIMIT of an upload was about (csdo_dsqp_last_limit)
  int64_t limit_bytes = 0;
  double t_pack = 0, t_stage = 0, t_h2d = 0, t_d2h = 0, t_unpack = 0;   // host seconds of the last upload / download
  DeviceBatch dev{};
  hipStream_t copy = nullptr;  // the stream of upload's H2D and download's D2H: the handle's own stream, or - a shared handle - a private
                               // one, so that a copy never queues up behind another batch's persistent kernels on a lent stream
  bool borrowed = false;       // csdo_dsqp_create_shared: the streams belong to another handle (never destroyed here)
  hipStream_t four[4] = {nullptr, nullptr, nullptr, nullptr};   // ... the owner's four streams
  bool run_pending = false;    // csdo_dsqp_run_async has been called and csdo_dsqp_wait has not
  csdo_handle stream_kids[4] = {nullptr, nullptr, nullptr, nullptr};   // csdo_do_phase: the chunks' shared handles, made once, gone with this one
  std::vector<char> pending_second;   // which groups of the pending run have a second launch
};

// where the results of a batch lie in stage_down: sol [steps][6], corr [steps][8], four int32 per agent, ticks int64 per agent
struct ResultLayout {
  size_t b_sol, b_corr, o_corr, o_sqp, o_admm, o_stat, o_legal, o_ticks, total;
  explicit ResultLayout(const HostBatch& hb) {
    const size_t Na = hb.agents.size();
    b_sol = (size_t)hb.steps_total * 6 * sizeof(double);
    b_corr = (size_t)hb.steps_total * 8 * sizeof(double);
    const size_t b_i32 = (Na * sizeof(int32_t) + 255) & ~(size_t)255;
    o_corr = (b_sol + 255) & ~(size_t)255;
    o_sqp = o_corr + ((b_corr + 255) & ~(size_t)255);
    o_admm = o_sqp + b_i32;
    o_stat = o_admm + b_i32;
    o_legal = o_stat + b_i32;
    o_ticks = o_legal + b_i32;
    total = o_ticks + Na * sizeof(int64_t);
  }
};

#define HIP_OK(expr, code)                 \
  do {                                     \
    if ((expr) != hipSuccess) return code; \
  } while (0)


// Groups the agents of the uploaded batch by kernel class and orders each group by batch_pack.h's launch rank, likely
// long runners first (workgroups are dispatched in index order, so the agents expected to run longest start first:
// longest-processing-time scheduling over the 256 CUs).
static int build_groups(csdo_handle h) {
  const HostBatch& hb = h->hb;
  const int Na = (int)hb.agents.size();
  struct Key { int block; int mode; };
  std::vector<Key> key(Na);
  std::vector<size_t> need(Na);
  HostBatch& hbm = h->hb;
  for (int a = 0; a < Na; ++a) {
    AgentDesc& ad = hbm.agents[a];
    const int n_obs = hb.worlds[ad.world].n_obs;
    int rows = 0, tail = TAIL_NODES;
    key[a].block = dsqp_agent_class(ad.Nt, n_obs, ad.n_planes, &key[a].mode, &rows, &tail);
    ad.tail_nodes = tail;   // (a function of the agent alone: the one item of the class that the results' last bits depend on)
    // testing knob (results never depend on it): 1 keeps the inter-vehicle rows' state in the workspace, 2 also reads the
    // LDS part of the factor from the workspace (512-thread class; the 1024-thread class is always mode 3)
    if (h->min_mode >= 1) rows = 0;
    if (h->min_mode >= 2 && key[a].block == 512) key[a].mode = 1;
    ad.rows_lds = rows;
    need[a] = dsqp_lds_bytes(ad.Nt, n_obs, ad.n_planes, key[a].mode, rows != 0, tail);
  }
  // The 768-thread class has two residency modes whose solves differ in form (mode 2: pair-split; mode 3: one lane per node) and
  // therefore in the last bits, so an agent's mode must be a function of THAT AGENT alone (dsqp_agent_class) - results do not
  // depend on what else is in the batch, how it is chunked or sharded (include/csdo_dsqp.h promises that; round 5 put the whole
  // class of a batch into mode 3 as soon as one of its agents needed it, which made a long-horizon agent's bits depend on its
  // neighbours: ADVICE r5).  A batch that mixes the two modes has a launch group more; beyond four groups a group's first launch
  // shares a hardware queue with an earlier group's (slower, same results).  min_mode >= 3: the testing knob, every such agent lean.
  if (h->min_mode >= 3)
    for (int a = 0; a < Na; ++a)
      if (key[a].block == 768 && key[a].mode != 3) {
        key[a].mode = 3;
        need[a] = dsqp_lds_bytes(hbm.agents[a].Nt, hb.worlds[hbm.agents[a].world].n_obs, hbm.agents[a].n_planes, 3, false);
      }
  h->order.resize(Na);
  for (int a = 0; a < Na; ++a) h->order[a] = a;
  std::stable_sort(h->order.begin(), h->order.end(), [&](int p, int q) {
    // groups are launched in this order and the first launch gets the CUs first (each group launches one persistent
    // workgroup per CU): the classes whose agents run longest - workspace-resident modes, long horizons - go first, so
    // that their longest agents start at once and the faster classes fill the CUs they release
    if (key[p].mode != key[q].mode) return key[p].mode > key[q].mode;
    if (key[p].block != key[q].block) return key[p].block > key[q].block;
    if (hb.launch_rank[p] != hb.launch_rank[q]) return hb.launch_rank[p] > hb.launch_rank[q];
    return hb.est_work[p] > hb.est_work[q];
  });
  h->groups.clear();
  for (int i = 0; i < Na; ++i) {
    const int a = h->order[i];
    if (h->groups.empty() || h->groups.back().block != key[a].block || h->groups.back().mode != key[a].mode) {
      LaunchGroup g;
      g.first = i;
      g.block = key[a].block;
      g.mode = key[a].mode;
      h->groups.push_back(g);
    }
    LaunchGroup& g = h->groups.back();
    g.count++;
    g.max_nt = std::max(g.max_nt, (int)hb.agents[a].Nt);
    g.lds_bytes = std::max(g.lds_bytes, need[a]);
  }
  // The planes' read-only coefficients take whatever LDS a launch has left (Shm::pco): a group with inter-vehicle rows in LDS
  // asks for everything its class may use (80 KB for the class that runs two workgroups per CU, 160 KB otherwise).
  for (LaunchGroup& g : h->groups) {
    bool any = false;
    for (int i = 0; i < g.count && !any; ++i) {
      const AgentDesc& ad = hb.agents[h->order[g.first + i]];
      any = ad.rows_lds != 0 && ad.n_planes > 0;
    }
    if (any && g.mode == 0 && g.lds_bytes <= dsqp_lds_capacity())
      g.lds_bytes = dsqp_workgroups_per_cu(g.block, g.lds_bytes) == 2 ? dsqp_lds_capacity_two_per_cu() : dsqp_lds_capacity();
  }
  // Streams.  HIP maps streams onto four hardware queues (GPU_MAX_HW_QUEUES), round robin in the order of their creation, and
  // kernels that share a queue run one after the other: with three launch groups and a stream per launch - six - a group's
  // FIRST launch was seen to wait tens of milliseconds behind another group's kernel (room set: 77 ms instead of 61 with
  // every group faster than before).  So the handle owns four streams, created back to back, and rations them: a stream per
  // group's first launch (groups beyond the fourth queue up behind the first ones), the streams that are left for the second
  // ("elastic") launches - one each with two groups, one for all with three, none beyond.
  while (h->side.size() < 3) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return CSDO_EDEVICE;
    h->side.push_back(s);
  }
  while (h->g_begin.size() < h->groups.size()) {
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess ||
        hipEventCreate(&e3) != hipSuccess)
      return CSDO_EDEVICE;
    h->g_begin.push_back(e0);
    h->g_end.push_back(e1);
    h->g_zeroed.push_back(e2);
    h->g_end2.push_back(e3);
  }
  return CSDO_OK;
}

// Packing and scattering use std::thread pools and growing vectors: a std::system_error (thread limit of the caller's cgroup) or
// std::bad_alloc must not cross the C ABI.
template <class F>
static int guarded(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return CSDO_ENOMEM;
  } catch (...) {
    return CSDO_EDEVICE;
  }
}

static int upload_impl(csdo_handle h, const csdo_problem* worlds, int32_t n_worlds);
static int download_impl(csdo_handle h, csdo_result* results, int32_t n_worlds);

// ---------------------------------------------------------------------------------------------------------------------
// Several GPUs behind one handle (csdo_dsqp_create_multi)
// ---------------------------------------------------------------------------------------------------------------------
// CSDO_ELIMIT from the packing: a horizon beyond CSDO_MAX_NT (one lane per timestep; the reference has no cap,
// sqp/inter_agent_cons.cc:320-325).  csdo_dsqp_last_limit then names that world (agent 0; no LDS figure: 0 bytes).
static void note_horizon_limit(csdo_handle h, const csdo_problem* worlds, int32_t n_worlds) {
  for (int w = 0; worlds && w < n_worlds; ++w)
    if (worlds[w].Nt > CSDO_MAX_NT) {
      h->limit_world = w;
      h->limit_agent = 0;
      h->limit_bytes = 0;
      return;
    }
}

// Contiguous blocks [cuts[r], cuts[r + 1]) of near-equal total weight: block r ends where the running sum first reaches
// (r + 1) / n_blocks of the total - the closer of the two candidate cuts -, every block keeps at least one item while there are
// enough (the rule of sharding.py: shard_bounds_weighted, which the N-process path uses; tests/test_multi_host.py holds them equal).
static std::vector<int> weighted_cuts(const double* w, int n, int n_blocks) {
  std::vector<int> cuts{0};
  std::vector<double> cum((size_t)std::max(n, 0));
  double total = 0.0;
  for (int i = 0; i < n; ++i) {
    total += w[i] > 0.0 ? w[i] : 0.0;
    cum[(size_t)i] = total;
  }
  for (int r = 1; r < n_blocks; ++r) {
    int i;
    if (!(total > 0.0)) {   // no information: equal counts
      const int base = n / n_blocks, rem = n % n_blocks;
      i = r * base + std::min(r, rem);
    } else {
      const double target = total * r / n_blocks;
      i = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin()) + 1;
      if (i - 1 > cuts.back() && i >= 2 && std::fabs(cum[(size_t)i - 2] - target) <= std::fabs(cum[(size_t)std::min(i, n) - 1] - target)) i -= 1;
      const int left = n - cuts.back(), behind = n_blocks - r;
      i = std::max(i, cuts.back() + (left > behind ? 1 : 0));
      i = std::min(i, left > behind ? n - std::min(behind, left) : n);
    }
    cuts.push_back(std::min(std::max(i, cuts.back()), n));
  }
  cuts.push_back(std::max(n, 0));
  return cuts;
}

// The multi-device entry points visit every child's device on the CALLER's thread (hipSetDevice is per thread); a caller with
// work of its own on another device - torch's current device - must find that device current again when the entry returns.
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

// f(k) for every child, each on a host thread of its own (child 0 on the caller's); a thread that cannot be had runs inline, and
// no exception leaves a joinable thread behind.  Returns the first error.
template <class F>
static int for_each_kid(MultiState& M, F&& f) {
  const int n = (int)M.kids.size();
  std::vector<int> rc((size_t)n, CSDO_OK);
  struct Joiner {
    std::vector<std::thread> t;
    ~Joiner() {
      for (auto& x : t)
        if (x.joinable()) x.join();
    }
  } pool;
  auto run = [&](int k) {
    try {
      rc[(size_t)k] = f(k);
    } catch (const std::bad_alloc&) {
      rc[(size_t)k] = CSDO_ENOMEM;
    } catch (...) {
      rc[(size_t)k] = CSDO_EDEVICE;
    }
  };
  try {
    pool.t.reserve((size_t)n);
  } catch (...) {
  }
  for (int k = 1; k < n; ++k) {
    try {
      pool.t.emplace_back(run, k);
    } catch (...) {
      run(k);
    }
  }
  run(0);
  for (auto& x : pool.t) x.join();
  pool.t.clear();
  for (int k = 0; k < n; ++k)
    if (rc[(size_t)k] != CSDO_OK) return rc[(size_t)k];
  return CSDO_OK;
}

static int multi_upload(csdo_handle h, const csdo_problem* worlds, int32_t n_worlds) {
  MultiState& M = *h->multi;
  const DeviceGuard keep_callers_device;
  if (h->run_pending) return CSDO_EINVAL;
  for (csdo_handle kid : M.kids)
    if (kid->run_pending) return CSDO_EINVAL;
  h->uploaded = false;
  h->limit_world = h->limit_agent = -1;
  h->limit_bytes = 0;
  const double t0 = now_s();
  HostBatch all;   // validates the batch and yields the per-agent work estimate (csdo_dsqp_estimate_work's)
  const PackPlacer estimates_only = pack_nothing;
  int rc = pack_worlds(worlds, n_worlds, all, &estimates_only);
  if (rc == CSDO_ELIMIT) note_horizon_limit(h, worlds, n_worlds);
  if (rc != CSDO_OK) return rc;
  const int nk = (int)M.kids.size(), Na = (int)all.agents.size();
  std::vector<double> est((size_t)Na);
  for (int a = 0; a < Na; ++a) est[(size_t)a] = (double)all.est_work[(size_t)a];
  const std::vector<int> cuts = weighted_cuts(est.data(), Na, nk);
  M.n_worlds = n_worlds;
  M.world_na.assign((size_t)n_worlds, 0);
  for (int w = 0; w < n_worlds; ++w) M.world_na[(size_t)w] = worlds[w].Na;
  M.parts.assign((size_t)nk, {});
  M.probs.assign((size_t)nk, {});
  M.offs.assign((size_t)nk, {});
  for (int k = 0; k < nk; ++k) {
    const int lo = cuts[(size_t)k], hi = cuts[(size_t)k + 1];
    for (int w = 0; w < n_worlds && lo < hi; ++w) {
      const int first = all.world_first_agent[(size_t)w], last = all.world_first_agent[(size_t)w + 1];
      const int a = std::max(lo, first), b = std::min(hi, last);
      if (b <= a) continue;
      M.parts[(size_t)k].push_back(MultiPart{w, a - first, b - first});
    }
    M.offs[(size_t)k].resize(M.parts[(size_t)k].size());
    for (size_t i = 0; i < M.parts[(size_t)k].size(); ++i) {
      const MultiPart& pt = M.parts[(size_t)k][i];
      csdo_problem P = worlds[pt.world];
      if (pt.lo > 0 || pt.hi < P.Na) {   // a cut world: agents [lo, hi) of it, plane offsets re-based
        std::vector<int32_t>& off = M.offs[(size_t)k][i];
        off.resize((size_t)(pt.hi - pt.lo) + 1);
        const int32_t base = P.plane_off[pt.lo];
        for (int j = 0; j <= pt.hi - pt.lo; ++j) off[(size_t)j] = P.plane_off[pt.lo + j] - base;
        P.x0_bar += (size_t)pt.lo * (size_t)P.Nt * 6;
        P.planes += base;
        P.plane_off = off.data();
        P.Na = pt.hi - pt.lo;
      }
      M.probs[(size_t)k].push_back(P);
    }
  }
  const double t_plan = now_s() - t0;
  rc = for_each_kid(M, [&](int k) {
    csdo_handle kid = M.kids[(size_t)k];
    kid->uploaded = false;
    if (M.probs[(size_t)k].empty()) return (int)CSDO_OK;   // more devices than agents
    return upload_impl(kid, M.probs[(size_t)k].data(), (int32_t)M.probs[(size_t)k].size());
  });
  if (rc == CSDO_ELIMIT)
    for (int k = 0; k < nk && h->limit_world < 0; ++k) {
      const csdo_handle kid = M.kids[(size_t)k];
      if (kid->limit_world < 0 || kid->limit_world >= (int)M.parts[(size_t)k].size()) continue;
      const MultiPart& pt = M.parts[(size_t)k][(size_t)kid->limit_world];
      h->limit_world = pt.world;
      h->limit_agent = pt.lo + kid->limit_agent;
      h->limit_bytes = kid->limit_bytes;
    }
  if (rc != CSDO_OK) return rc;
  h->t_pack = h->t_stage = h->t_h2d = 0.0;
  for (csdo_handle kid : M.kids) {   // the children work side by side: the slowest one's times, plus the plan
    h->t_pack = std::max(h->t_pack, kid->t_pack);
    h->t_stage = std::max(h->t_stage, kid->t_stage);
    h->t_h2d = std::max(h->t_h2d, kid->t_h2d);
  }
  h->t_pack += t_plan;
  h->n_worlds = n_worlds;
  h->uploaded = true;
  return CSDO_OK;
}

static int multi_run_async(csdo_handle h, void* hip_stream) {
  MultiState& M = *h->multi;
  const DeviceGuard keep_callers_device;
  if (!h->uploaded || h->run_pending) return CSDO_EINVAL;
  int rc = CSDO_OK;
  size_t started = 0;
  for (; started < M.kids.size() && rc == CSDO_OK; ++started) {
    csdo_handle kid = M.kids[started];
    if (!kid->uploaded) continue;
    rc = csdo_dsqp_run_async(kid, nullptr);   // every device starts at once, on its own streams
    // the caller's stream, if there is one, joins: what it enqueues next is ordered behind every device's solve
    if (rc == CSDO_OK && hip_stream && hipStreamWaitEvent((hipStream_t)hip_stream, kid->ev1, 0) != hipSuccess) rc = CSDO_EDEVICE;
  }
  if (rc != CSDO_OK) {   // nothing keeps running behind the caller's back
    for (size_t k = 0; k < started; ++k)
      if (M.kids[k]->run_pending) (void)csdo_dsqp_wait(M.kids[k]);
    return rc;
  }
  h->run_pending = true;
  return CSDO_OK;
}

static int multi_wait(csdo_handle h) {
  MultiState& M = *h->multi;
  const DeviceGuard keep_callers_device;
  if (!h->run_pending) return CSDO_EINVAL;
  h->run_pending = false;
  int rc = CSDO_OK;
  h->last_kernel_s = 0.0;
  for (csdo_handle kid : M.kids) {
    if (!kid->run_pending) continue;
    const int r = csdo_dsqp_wait(kid);
    if (r != CSDO_OK && rc == CSDO_OK) rc = r;
    h->last_kernel_s = std::max(h->last_kernel_s, kid->last_kernel_s);
  }
  return rc;
}

static int multi_download(csdo_handle h, csdo_result* results, int32_t n_worlds) {
  MultiState& M = *h->multi;
  const DeviceGuard keep_callers_device;
  if (!h->uploaded || h->run_pending || !results || n_worlds != M.n_worlds) return CSDO_EINVAL;
  const int nk = (int)M.kids.size();
  M.res.assign((size_t)nk, {});
  for (int k = 0; k < nk; ++k)
    for (const MultiPart& pt : M.parts[(size_t)k]) {
      const csdo_result& R = results[pt.world];
      if (!R.solutions || !R.corridors || !R.sqp_iters || !R.admm_iters || !R.last_status) return CSDO_EINVAL;
      const size_t Nt = (size_t)M.probs[(size_t)k][M.res[(size_t)k].size()].Nt;
      csdo_result V{};
      V.solutions = R.solutions + (size_t)pt.lo * Nt * 6;
      V.corridors = R.corridors + (size_t)pt.lo * Nt * 8;
      V.sqp_iters = R.sqp_iters + pt.lo;
      V.admm_iters = R.admm_iters + pt.lo;
      V.last_status = R.last_status + pt.lo;
      V.agent_seconds = R.agent_seconds ? R.agent_seconds + pt.lo : nullptr;
      M.res[(size_t)k].push_back(V);
    }
  const int rc = for_each_kid(M, [&](int k) {
    if (M.res[(size_t)k].empty()) return (int)CSDO_OK;
    return download_impl(M.kids[(size_t)k], M.res[(size_t)k].data(), (int32_t)M.res[(size_t)k].size());
  });
  if (rc != CSDO_OK) return rc;
  // a world's scalars from all of its parts: the status rule of sqp/dsqp_solver.cc:1224-1243 over every agent of the world
  for (int w = 0; w < n_worlds; ++w) {
    results[w].initial_static_legal = 1;
    results[w].t_max_individual = 0.0;
    results[w].t_device = h->last_kernel_s;
    bool any_bad = false;
    int worst = 2;
    for (int a = 0; a < M.world_na[(size_t)w]; ++a) {
      const int st = results[w].last_status[a];
      if (std::abs(st) > 1) {
        any_bad = true;
        if (std::abs(st) > worst) worst = st;
      }
    }
    results[w].solver_status = any_bad ? worst : 1;
  }
  h->t_d2h = h->t_unpack = 0.0;
  for (int k = 0; k < nk; ++k) {
    for (size_t i = 0; i < M.parts[(size_t)k].size(); ++i) {
      csdo_result& R = results[M.parts[(size_t)k][i].world];
      const csdo_result& V = M.res[(size_t)k][i];
      if (!V.initial_static_legal) R.initial_static_legal = 0;
      R.t_max_individual = std::max(R.t_max_individual, V.t_max_individual);
    }
    h->t_d2h = std::max(h->t_d2h, M.kids[(size_t)k]->t_d2h);
    h->t_unpack = std::max(h->t_unpack, M.kids[(size_t)k]->t_unpack);
  }
  return CSDO_OK;
}

extern "C" {

const char* csdo_backend_name(void) { return "hip-gfx950"; }

#if !defined(CSDO_SOURCE_HASH)
#define CSDO_SOURCE_HASH "unknown"
#endif
const char* csdo_source_hash(void) { return CSDO_SOURCE_HASH; }

int csdo_dsqp_create(csdo_handle* out, int device_ordinal) {
  if (!out) return CSDO_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return CSDO_ENODEV;
  HIP_OK(hipSetDevice(device_ordinal), CSDO_ENODEV);
  csdo_handle h = new (std::nothrow) csdo_handle_s();
  if (!h) return CSDO_ENOMEM;
  h->device = device_ordinal;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0) h->n_cu = cus;
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    csdo_dsqp_destroy(h);
    return CSDO_ENODEV;
  }
  h->copy = h->stream;
  // the handle's four streams, created back to back (build_groups explains why four and why in a row); created here so that
  // csdo_dsqp_create_shared never has to touch its parent
  try {
    for (int k = 0; k < 3; ++k) {
      hipStream_t s = nullptr;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        csdo_dsqp_destroy(h);
        return CSDO_ENODEV;
      }
      h->side.push_back(s);
    }
  } catch (...) {
    csdo_dsqp_destroy(h);
    return CSDO_ENOMEM;
  }
  *out = h;
  return CSDO_OK;
}

// A second (third, ...) batch in flight on the same GPU: a handle with device buffers and events of its own whose launches go to
// the PARENT's four streams, starting at stream `lane` (HIP maps streams onto four hardware queues in the order of their
// creation and kernels that share a queue run one after the other: handles that each created four streams would collide on
// them).  Its groups launch all their persistent workgroups at once (no second launch): they queue up behind the workgroups
// of the batches in front and take the CUs those release.  The parent must outlive it.
int csdo_dsqp_create_shared(csdo_handle* out, csdo_handle parent, int32_t lane) {
  if (!out || !parent || lane < 0 || parent->multi) return CSDO_EINVAL;   // (a multi-device handle has no streams to lend: use a child)
  *out = nullptr;
  HIP_OK(hipSetDevice(parent->device), CSDO_ENODEV);
  if (parent->side.size() < 3) return CSDO_EINVAL;
  csdo_handle h = new (std::nothrow) csdo_handle_s();
  if (!h) return CSDO_ENOMEM;
  h->device = parent->device;
  h->n_cu = parent->n_cu;
  h->min_mode = parent->min_mode;
  h->borrowed = true;
  const hipStream_t four[4] = {parent->stream, parent->side[0], parent->side[1], parent->side[2]};
  for (int k = 0; k < 4; ++k) h->four[k] = four[k];
  h->stream = four[lane & 3];
  try {
    for (int k = 1; k < 4; ++k) h->side.push_back(four[(lane + k) & 3]);
  } catch (...) {
    delete h;
    return CSDO_ENOMEM;
  }
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
      hipStreamCreateWithFlags(&h->copy, hipStreamNonBlocking) != hipSuccess) {
    csdo_dsqp_destroy(h);
    return CSDO_ENODEV;
  }
  *out = h;
  return CSDO_OK;
}

// Re-points a shared handle at stream `lane` of its parent (its launch groups take lane, lane + 1, ... modulo four): a batch
// with several launch groups needs as many streams, so the caller that keeps batches in flight deals the lanes out by the
// group counts (csdo_dsqp_launch_groups) once the batches are uploaded.  Not while a run is pending.
int csdo_dsqp_set_lane(csdo_handle h, int32_t lane) {
  if (!h || h->multi || !h->borrowed || h->run_pending || lane < 0) return CSDO_EINVAL;
  h->stream = h->four[lane & 3];
  for (int k = 1; k < 4; ++k) h->side[(size_t)k - 1] = h->four[(lane + k) & 3];
  return CSDO_OK;
}

int csdo_dsqp_create_multi(csdo_handle* out, const int32_t* devices, int32_t n_devices) {
  if (!out || !devices || n_devices < 1) return CSDO_EINVAL;
  *out = nullptr;
  csdo_handle h = new (std::nothrow) csdo_handle_s();
  MultiState* M = new (std::nothrow) MultiState();
  if (!h || !M) {
    delete h;
    delete M;
    return CSDO_ENOMEM;
  }
  h->multi = M;
  h->device = devices[0];
  const DeviceGuard keep_callers_device;
  int rc = CSDO_OK;
  try {
    for (int k = 0; k < n_devices && rc == CSDO_OK; ++k) {
      csdo_handle kid = nullptr;
      rc = csdo_dsqp_create(&kid, devices[k]);   // the same ordinal twice: two independent handles on one GPU
      if (rc == CSDO_OK) M->kids.push_back(kid);
    }
  } catch (...) {
    rc = CSDO_ENOMEM;
  }
  if (rc != CSDO_OK) {
    csdo_dsqp_destroy(h);
    return rc;
  }
  *out = h;
  return CSDO_OK;
}

int csdo_dsqp_shard_bounds(const double* weights, int32_t n_items, int32_t n_blocks, int32_t* cuts) {
  if (!cuts || n_items < 0 || n_blocks < 1 || (n_items > 0 && !weights)) return CSDO_EINVAL;
  try {
    const std::vector<int> c = weighted_cuts(weights, n_items, n_blocks);
    for (int r = 0; r <= n_blocks; ++r) cuts[r] = c[(size_t)r];
    return CSDO_OK;
  } catch (...) {
    return CSDO_ENOMEM;
  }
}

int32_t csdo_dsqp_multi_count(csdo_handle h) { return !h ? CSDO_EINVAL : (h->multi ? (int32_t)h->multi->kids.size() : 0); }

csdo_handle csdo_dsqp_multi_child(csdo_handle h, int32_t k) {
  return (h && h->multi && k >= 0 && k < (int32_t)h->multi->kids.size()) ? h->multi->kids[(size_t)k] : nullptr;
}

void csdo_dsqp_destroy(csdo_handle h) {
  if (!h) return;
  if (h->multi) {
    const DeviceGuard keep_callers_device;
    for (csdo_handle kid : h->multi->kids) csdo_dsqp_destroy(kid);
    delete h->multi;
    delete h;
    return;
  }
  for (csdo_handle& kid : h->stream_kids) {
    if (kid) csdo_dsqp_destroy(kid);
    kid = nullptr;
  }
  (void)hipSetDevice(h->device);
  if (h->run_pending && h->ev1) (void)hipEventSynchronize(h->ev1);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->borrowed && h->copy) {
    (void)hipStreamSynchronize(h->copy);
    (void)hipStreamDestroy(h->copy);
  }
  for (DevBuf* b : {&h->in_arena, &h->rows_ws, &h->fac_ws,
                    &h->sol, &h->corr, &h->sqp, &h->admm, &h->stat, &h->legal, &h->ticks, &h->queues, &h->box_pts,
                    &h->box_obs, &h->box_out, &h->box_status, &h->prof, &h->k0_centres, &h->k0_counts, &h->k0_offsets,
                    &h->k0_pairs, &h->k0_coef, &h->k0_flag, &h->val_sol, &h->val_obs, &h->val_out, &h->val_frames})
    b->release();
  h->stage_up.release();
  h->stage_down.release();
  h->bridge_up.release();
  h->bridge_down.release();
  if (!h->borrowed)
    for (hipStream_t s : h->side) (void)hipStreamDestroy(s);
  for (hipEvent_t e : h->g_begin) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->g_end) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->g_zeroed) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->g_end2) (void)hipEventDestroy(e);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream && !h->borrowed) (void)hipStreamDestroy(h->stream);
  delete h;
}

int csdo_dsqp_last_limit(csdo_handle h, int32_t* world, int32_t* agent, int64_t* lds_bytes_needed) {
  if (!h) return CSDO_EINVAL;
  if (world) *world = h->limit_world;
  if (agent) *agent = h->limit_agent;
  if (lds_bytes_needed) *lds_bytes_needed = h->limit_bytes;
  return CSDO_OK;
}

int csdo_dsqp_estimate_work(const csdo_problem* worlds, int32_t n_worlds, double* est) {
  if (!worlds || n_worlds < 1 || !est) return CSDO_EINVAL;
  try {
    HostBatch hb;
    const PackPlacer estimates_only = pack_nothing;
    const int rc = pack_worlds(worlds, n_worlds, hb, &estimates_only);
    if (rc != CSDO_OK) return rc;
    for (size_t a = 0; a < hb.est_work.size(); ++a) est[a] = (double)hb.est_work[a];
    return CSDO_OK;
  } catch (const std::bad_alloc&) {
    return CSDO_ENOMEM;
  } catch (...) {
    return CSDO_EINVAL;
  }
}

int csdo_dsqp_agent_class(int32_t Nt, int32_t n_obstacles, int32_t n_planes, int64_t out[5]) {
  if (!out || Nt < 2 || Nt > CSDO_MAX_NT || n_obstacles < 0 || n_planes < 0) return CSDO_EINVAL;
  int mode = 0, rows = 0, tail = TAIL_NODES;
  out[0] = dsqp_agent_class(Nt, n_obstacles, n_planes, &mode, &rows, &tail);
  out[1] = mode;
  out[2] = rows;
  out[3] = tail;
  out[4] = (int64_t)dsqp_lds_bytes(Nt, n_obstacles, n_planes, mode, rows != 0, tail);
  return CSDO_OK;
}

int csdo_dsqp_upload(csdo_handle h, const csdo_problem* worlds, int32_t n_worlds) {
  if (h && h->multi) return guarded([&]() { return multi_upload(h, worlds, n_worlds); });
  return guarded([&]() { return upload_impl(h, worlds, n_worlds); });
}
static int upload_impl(csdo_handle h, const csdo_problem* worlds, int32_t n_worlds) {
  if (!h || h->run_pending) return CSDO_EINVAL;
  HIP_OK(hipSetDevice(h->device), CSDO_ENODEV);
  h->uploaded = false;
  h->limit_world = h->limit_agent = -1;
  h->limit_bytes = 0;
  const double t0 = now_s();
  // Everything the kernels read goes into ONE page-locked arena (256-byte aligned slots: agents, worlds, x0, planes, tstart, obstacles,
  // order) and from there to the device in one copy.  The four big arrays are packed straight into their slots (pack_worlds' placer).
  size_t total = 0;
  const PackPlacer place = [&](const PackSizes& n, PackPlace& at) -> int {
    const size_t bytes[7] = {n.agents * sizeof(AgentDesc), n.worlds * sizeof(WorldDesc), n.x0 * sizeof(double), n.planes * sizeof(PlaneDev),
                             n.tstart * sizeof(int32_t), n.obstacles * sizeof(double), n.agents * sizeof(int32_t)};
    total = 0;
    for (int k = 0; k < 7; ++k) {
      h->in_off[k] = total;
      total += (bytes[k] + 255) & ~(size_t)255;
    }
    const int prc = h->stage_up.ensure(total);
    if (prc != CSDO_OK) return prc;
    char* st = (char*)h->stage_up.p;
    at = PackPlace{(double*)(st + h->in_off[2]), (PlaneDev*)(st + h->in_off[3]), (int32_t*)(st + h->in_off[4]), (double*)(st + h->in_off[5])};
    return CSDO_OK;
  };
  int rc = pack_worlds(worlds, n_worlds, h->hb, &place);
  if (rc == CSDO_ELIMIT) note_horizon_limit(h, worlds, n_worlds);
  if (rc != CSDO_OK) return rc;
  HostBatch& hb = h->hb;
  const size_t Na = hb.agents.size();
  if ((rc = build_groups(h)) != CSDO_OK) return rc;
  for (const LaunchGroup& g : h->groups)
    if (g.lds_bytes > dsqp_lds_capacity()) {   // e.g. more obstacles than fit beside the exchange vectors
      // which world it is: csdo_dsqp_last_limit (one oversized world rejects the whole batch; the caller can drop it and retry)
      h->limit_world = h->limit_agent = -1;
      for (int i = 0; i < g.count && h->limit_world < 0; ++i) {
        const int a = h->order[g.first + i];
        const AgentDesc& ad = hb.agents[a];
        int mode = 0, rows = 0;
        int tail_ = 0;
        dsqp_agent_class(ad.Nt, hb.worlds[ad.world].n_obs, ad.n_planes, &mode, &rows, &tail_);
        if (dsqp_lds_bytes(ad.Nt, hb.worlds[ad.world].n_obs, ad.n_planes, mode, false) > dsqp_lds_capacity()) {
          h->limit_world = ad.world;
          h->limit_agent = a - hb.world_first_agent[ad.world];
          h->limit_bytes = (int64_t)dsqp_lds_bytes(ad.Nt, hb.worlds[ad.world].n_obs, ad.n_planes, mode, false);
        }
      }
      return CSDO_ELIMIT;
    }
  const double t1 = now_s();
  {   // the small arrays: descriptors (build_groups has set the agents' classes) and the launch order
    char* st = (char*)h->stage_up.p;
    std::memcpy(st + h->in_off[0], hb.agents.data(), hb.agents.size() * sizeof(AgentDesc));
    std::memcpy(st + h->in_off[1], hb.worlds.data(), hb.worlds.size() * sizeof(WorldDesc));
    std::memcpy(st + h->in_off[6], h->order.data(), h->order.size() * sizeof(int32_t));
  }
  const double t2 = now_s();
  if ((rc = h->in_arena.ensure(total)) != CSDO_OK) return rc;
  if (total) HIP_OK(hipMemcpyAsync(h->in_arena.p, h->stage_up.p, total, hipMemcpyHostToDevice, h->copy), CSDO_EDEVICE);
  h->t_pack = t1 - t0;
  h->t_stage = t2 - t1;
  if ((rc = h->queues.ensure(h->groups.size() * 64)) != CSDO_OK) return rc;   // one counter per group, a cache line apart
  for (size_t g = 0; g < h->groups.size(); ++g) {
    h->groups[g].queue = (int*)((char*)h->queues.p + g * 64);
  }
  // CU shares: every group first launches its share of the CUs (by estimated work; workspace-resident modes iterate
  // slower), then - after all first launches - the rest up to one workgroup per CU.  The second launch only gets CUs
  // that other groups release, so a group that finishes early hands its CUs over and nobody idles on a bad estimate.
  {
    const double mode_cost[4] = {1.0, 1.2, 1.6, 5.5};   // CU time per agent and timestep relative to mode 0 (measured on the room set; mode 3: 87 k cycles per iteration against 16 k)
    std::vector<double> work(h->groups.size(), 0.0);
    double total = 0.0;
    for (size_t g = 0; g < h->groups.size(); ++g) {
      for (int i = 0; i < h->groups[g].count; ++i) work[g] += hb.est_work[h->order[h->groups[g].first + i]];
      work[g] *= mode_cost[h->groups[g].mode];
      total += work[g];
    }
    int left = h->n_cu;
    for (size_t g = 0; g < h->groups.size(); ++g) {
      LaunchGroup& G = h->groups[g];
      const int per_cu = dsqp_workgroups_per_cu(G.block, G.lds_bytes);   // 2 for the 256-thread class when its LDS allows
      const int cap_cu = std::min((G.count + per_cu - 1) / per_cu, h->n_cu);
      int n = (g + 1 == h->groups.size()) ? left : (int)std::lround(h->n_cu * work[g] / std::max(total, 1e-30));
      n = std::max(1, std::min(n, std::min(cap_cu, std::max(left, 1))));
      G.primary = std::min(n * per_cu, G.count);
      // (the first group is the one whose agents run longest: its share is sized for them and it hands its CUs over as
      // its queue drains; a second launch of it was observed to take CUs ahead of the later groups' first launches)
      G.elastic = std::max(0, std::min(cap_cu * per_cu, G.count) - G.primary);
      // no stream is left for second launches with more than three groups, and none is wanted behind another batch's
      // workgroups: the one launch then asks for every workgroup the group can use and the hardware starts them as CUs free up
      if (h->groups.size() > 3 || h->borrowed) {
        G.primary = std::min(cap_cu * per_cu, G.count);
        G.elastic = 0;
      }
      left -= n;
    }
  }
  if ((rc = h->rows_ws.ensure((size_t)hb.rows_total * ROWS_WS_STRIDE * sizeof(double))) != CSDO_OK) return rc;
  if ((rc = h->fac_ws.ensure((size_t)hb.fac_total * sizeof(double))) != CSDO_OK) return rc;
  const ResultLayout RL(hb);
  char* res_host = nullptr;   // device address of stage_down when the kernels write there
  if (h->host_results) {
    if ((rc = h->stage_down.ensure(RL.total)) != CSDO_OK) return rc;
    void* dp = nullptr;
    HIP_OK(hipHostGetDevicePointer(&dp, h->stage_down.p, 0), CSDO_EDEVICE);
    res_host = (char*)dp;
  } else {
    if ((rc = h->sol.ensure(RL.b_sol)) != CSDO_OK) return rc;
    if ((rc = h->corr.ensure(RL.b_corr)) != CSDO_OK) return rc;
    if ((rc = h->sqp.ensure(Na * sizeof(int32_t))) != CSDO_OK) return rc;
    if ((rc = h->admm.ensure(Na * sizeof(int32_t))) != CSDO_OK) return rc;
    if ((rc = h->stat.ensure(Na * sizeof(int32_t))) != CSDO_OK) return rc;
    if ((rc = h->legal.ensure(Na * sizeof(int32_t))) != CSDO_OK) return rc;
    if ((rc = h->ticks.ensure(Na * sizeof(int64_t))) != CSDO_OK) return rc;
  }
  h->results_in_stage = h->host_results;
  DeviceBatch& B = h->dev;
  const char* const in = (const char*)h->in_arena.p;
  B.agents = (const AgentDesc*)(in + h->in_off[0]);
  B.worlds = (const WorldDesc*)(in + h->in_off[1]);
  B.x0 = (const double*)(in + h->in_off[2]);
  B.planes = (const PlaneDev*)(in + h->in_off[3]);
  B.tstart = (const int32_t*)(in + h->in_off[4]);
  B.obstacles = (const double*)(in + h->in_off[5]);
  B.rows_ws = (double*)h->rows_ws.p;
  B.fac_ws = (double*)h->fac_ws.p;
  B.sol = res_host ? (double*)res_host : (double*)h->sol.p;
  B.corr = res_host ? (double*)(res_host + RL.o_corr) : (double*)h->corr.p;
  B.sqp_iters = res_host ? (int32_t*)(res_host + RL.o_sqp) : (int32_t*)h->sqp.p;
  B.admm_iters = res_host ? (int32_t*)(res_host + RL.o_admm) : (int32_t*)h->admm.p;
  B.last_status = res_host ? (int32_t*)(res_host + RL.o_stat) : (int32_t*)h->stat.p;
  B.static_legal = res_host ? (int32_t*)(res_host + RL.o_legal) : (int32_t*)h->legal.p;
  B.agent_ticks = res_host ? (int64_t*)(res_host + RL.o_ticks) : (int64_t*)h->ticks.p;
  B.order = (const int32_t*)(in + h->in_off[6]);
  B.n_agents = (int32_t)Na;
  B.prof = nullptr;
#if defined(CSDO_PROFILE_PHASES)
  if ((rc = h->prof.ensure(Na * 48 * sizeof(int64_t))) != CSDO_OK) return rc;
  B.prof = (int64_t*)h->prof.p;
  (void)hipMemsetAsync(h->prof.p, 0, Na * 48 * sizeof(int64_t), h->copy);
#endif
  B.prm = hb.prm;
  HIP_OK(hipStreamSynchronize(h->copy), CSDO_EDEVICE);
  h->t_h2d = now_s() - t2;
  h->n_worlds = n_worlds;
  h->uploaded = true;
  return CSDO_OK;
}

// Enqueues the solve of the uploaded batch and returns: every group's launches, forked from and joined back into the
// caller's stream (or the handle's).  csdo_dsqp_wait blocks until it is done and collects the timings; until then the batch
// must not be uploaded again or downloaded.  Work enqueued on the same stream afterwards (a copy of the device results,
// a collective) is ordered behind the solve.
int csdo_dsqp_run_async(csdo_handle h, void* hip_stream) {
  if (h && h->multi) return guarded([&]() { return multi_run_async(h, hip_stream); });
  if (!h || !h->uploaded || h->run_pending) return CSDO_EINVAL;
  HIP_OK(hipSetDevice(h->device), CSDO_ENODEV);
  // after the fork an error must not leave side streams running behind the caller's back: drain the device first
#define RUN_OK(expr)                      \
  do {                                    \
    if ((expr) != hipSuccess) {           \
      (void)hipDeviceSynchronize();       \
      return CSDO_EDEVICE;                \
    }                                     \
  } while (0)
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  RUN_OK(hipEventRecord(h->ev0, s));
  const int ng = (int)h->groups.size();
  auto primary_stream = [&](int g) { return (g & 3) == 0 ? s : h->side[(g & 3) - 1]; };
  for (int g = 0; g < ng; ++g) {   // fork: every group starts when the caller's stream reaches this point
    hipStream_t gs = primary_stream(g);
    if (g > 0) RUN_OK(hipStreamWaitEvent(gs, h->ev0, 0));
    RUN_OK(hipEventRecord(h->g_begin[g], gs));
    RUN_OK(hipMemsetAsync(h->groups[g].queue, 0, sizeof(int), gs));
    RUN_OK(hipEventRecord(h->g_zeroed[g], gs));
    RUN_OK(launch_dsqp(h->dev, h->groups[g], h->groups[g].primary, gs));
    RUN_OK(hipEventRecord(h->g_end[g], gs));
  }
  // second launches: same queues, workgroups that start on CUs other groups release; the first group's comes last so
  // that it does not take CUs ahead of the later groups' first launches
  h->pending_second.assign(ng, 0);
  for (int g = ng - 1; g >= 0; --g) {
    if (h->groups[g].elastic <= 0 || ng > 3 || h->borrowed) continue;
    hipStream_t es = ng == 1 ? h->side[0] : (ng == 2 ? h->side[1 + g] : h->side[2]);   // (what the first launches leave)
    RUN_OK(hipStreamWaitEvent(es, h->g_zeroed[g], 0));
    RUN_OK(launch_dsqp(h->dev, h->groups[g], h->groups[g].elastic, es));
    RUN_OK(hipEventRecord(h->g_end2[g], es));
    RUN_OK(hipStreamWaitEvent(s, h->g_end2[g], 0));
    h->pending_second[g] = 1;
  }
  for (int g = 1; g < ng; ++g) RUN_OK(hipStreamWaitEvent(s, h->g_end[g], 0));   // join
  RUN_OK(hipEventRecord(h->ev1, s));
  h->run_pending = true;
  return CSDO_OK;
}

int csdo_dsqp_wait(csdo_handle h) {
  if (h && h->multi) return guarded([&]() { return multi_wait(h); });
  if (!h || !h->run_pending) return CSDO_EINVAL;
  HIP_OK(hipSetDevice(h->device), CSDO_ENODEV);
  h->run_pending = false;
  RUN_OK(hipEventSynchronize(h->ev1));
  const int ng = (int)h->groups.size();
  float ms = 0.f;
  for (int g = 0; g < ng; ++g) {
    RUN_OK(hipEventElapsedTime(&ms, h->g_begin[g], h->g_end[g]));
    h->groups[g].seconds = (double)ms * 1e-3;
    if (h->pending_second[g]) {   // the group is done when both of its launches are
      RUN_OK(hipEventElapsedTime(&ms, h->g_begin[g], h->g_end2[g]));
      h->groups[g].seconds = std::max(h->groups[g].seconds, (double)ms * 1e-3);
    }
  }
  RUN_OK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_kernel_s = (double)ms * 1e-3;
#undef RUN_OK
  return CSDO_OK;
}

int csdo_dsqp_run(csdo_handle h, void* hip_stream) {
  const int rc = csdo_dsqp_run_async(h, hip_stream);
  return rc != CSDO_OK ? rc : csdo_dsqp_wait(h);
}

double csdo_dsqp_last_kernel_seconds(csdo_handle h) { return h ? h->last_kernel_s : 0.0; }

int csdo_dsqp_last_transfer_seconds(csdo_handle h, double out[5]) {
  if (!h || !out) return CSDO_EINVAL;
  out[0] = h->t_pack;
  out[1] = h->t_stage;
  out[2] = h->t_h2d;
  out[3] = h->t_d2h;
  out[4] = h->t_unpack;
  return CSDO_OK;
}

int csdo_dsqp_agent_groups(csdo_handle h, int32_t* group_of_agent, int32_t n_agents) {
  if (h && h->multi) {   // upload order = the children's blocks one after the other; group indices continue from child to child
    if (!h->uploaded || !group_of_agent) return CSDO_EINVAL;
    int32_t done = 0, base = 0;
    for (csdo_handle kid : h->multi->kids) {
      if (!kid->uploaded) continue;
      const int32_t n = (int32_t)kid->order.size();
      if (done + n > n_agents) return CSDO_EINVAL;
      const int rc = csdo_dsqp_agent_groups(kid, group_of_agent + done, n);
      if (rc != CSDO_OK) return rc;
      for (int32_t a = 0; a < n; ++a) group_of_agent[done + a] += base;
      done += n;
      base += (int32_t)kid->groups.size();
    }
    return done == n_agents ? CSDO_OK : CSDO_EINVAL;
  }
  if (!h || !h->uploaded || !group_of_agent || n_agents != (int32_t)h->order.size()) return CSDO_EINVAL;
  for (int g = 0; g < (int)h->groups.size(); ++g)
    for (int i = 0; i < h->groups[g].count; ++i) group_of_agent[h->order[h->groups[g].first + i]] = g;
  return CSDO_OK;
}

int csdo_dsqp_set_host_results(csdo_handle h, int32_t on) {
  if (!h) return CSDO_EINVAL;
  if (h->multi)
    for (csdo_handle kid : h->multi->kids) kid->host_results = on != 0;
  h->host_results = on != 0;
  return CSDO_OK;
}

int csdo_dsqp_set_min_residency_mode(csdo_handle h, int32_t mode) {
  if (!h || mode < 0 || mode > 3) return CSDO_EINVAL;
  if (h->multi)
    for (csdo_handle kid : h->multi->kids) kid->min_mode = mode;
  h->min_mode = mode;
  return CSDO_OK;
}

int32_t csdo_dsqp_launch_groups(csdo_handle h, csdo_launch_group* out, int32_t cap) {
  if (!h || !h->uploaded || (cap > 0 && !out)) return CSDO_EINVAL;
  if (h->multi) {   // every child's groups, child by child
    int32_t total = 0;
    for (csdo_handle kid : h->multi->kids) {
      if (!kid->uploaded) continue;
      const int32_t n = csdo_dsqp_launch_groups(kid, cap > total ? out + total : nullptr, cap > total ? cap - total : 0);
      if (n < 0) return n;
      total += n;
    }
    return total;
  }
  const int ng = (int)h->groups.size();
  for (int g = 0; g < ng && g < cap; ++g) {
    const LaunchGroup& G = h->groups[g];
    out[g].n_agents = G.count;
    out[g].threads = G.block;
    out[g].residency_mode = G.mode;
    out[g].max_nt = G.max_nt;
    out[g].lds_bytes = (int64_t)G.lds_bytes;
    out[g].seconds = G.seconds;
  }
  return ng;
}

void* csdo_dsqp_device_solutions(csdo_handle h, int64_t* n_doubles) {
  if (!h || !h->uploaded || h->multi) return nullptr;   // (several devices: ask the children, csdo_dsqp_multi_child)
  if (n_doubles) *n_doubles = h->hb.steps_total * 6;
  return h->dev.sol;   // (csdo_dsqp_set_host_results: the device address of page-locked host memory)
}

int csdo_dsqp_download(csdo_handle h, csdo_result* results, int32_t n_worlds) {
  if (h && h->multi) return guarded([&]() { return multi_download(h, results, n_worlds); });
  return guarded([&]() { return download_impl(h, results, n_worlds); });
}
static int download_impl(csdo_handle h, csdo_result* results, int32_t n_worlds) {
  if (!h || !h->uploaded || h->run_pending || !results || n_worlds != h->n_worlds) return CSDO_EINVAL;
  HIP_OK(hipSetDevice(h->device), CSDO_ENODEV);
  const HostBatch& hb = h->hb;
  const size_t Na = hb.agents.size();
  const double t0 = now_s();
  const ResultLayout RL(hb);
  const size_t o_corr = RL.o_corr, o_sqp = RL.o_sqp, o_admm = RL.o_admm, o_stat = RL.o_stat, o_legal = RL.o_legal, o_ticks = RL.o_ticks;
  int rc;
  if ((rc = h->stage_down.ensure(RL.total)) != CSDO_OK) return rc;   // (host results: no-op, the upload sized it)
  char* st = (char*)h->stage_down.p;
  if (!h->results_in_stage) {
    hipStream_t s = h->copy;   // (the solve is over - csdo_dsqp_wait has returned -, so nothing orders this copy but itself)
    HIP_OK(hipMemcpyAsync(st, h->sol.p, RL.b_sol, hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
    HIP_OK(hipMemcpyAsync(st + o_corr, h->corr.p, RL.b_corr, hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
    HIP_OK(hipMemcpyAsync(st + o_sqp, h->sqp.p, Na * sizeof(int32_t), hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
    HIP_OK(hipMemcpyAsync(st + o_admm, h->admm.p, Na * sizeof(int32_t), hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
    HIP_OK(hipMemcpyAsync(st + o_stat, h->stat.p, Na * sizeof(int32_t), hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
    HIP_OK(hipMemcpyAsync(st + o_legal, h->legal.p, Na * sizeof(int32_t), hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
    HIP_OK(hipMemcpyAsync(st + o_ticks, h->ticks.p, Na * sizeof(int64_t), hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
    HIP_OK(hipStreamSynchronize(s), CSDO_EDEVICE);
  }
  const double t1 = now_s();
  const int64_t* h_ticks = (const int64_t*)(st + o_ticks);
  unpack_results(hb, nullptr, n_worlds, (const double*)st, (const double*)(st + o_corr), (const int32_t*)(st + o_sqp),
                 (const int32_t*)(st + o_admm), (const int32_t*)(st + o_stat), (const int32_t*)(st + o_legal), results);
  h->t_d2h = t1 - t0;
  h->t_unpack = now_s() - t1;
  for (int w = 0; w < n_worlds; ++w) {
    int64_t mx = 0;
    for (int a = hb.world_first_agent[w]; a < hb.world_first_agent[w + 1]; ++a) {
      mx = std::max(mx, h_ticks[a]);
      if (results[w].agent_seconds) results[w].agent_seconds[a - hb.world_first_agent[w]] = (double)h_ticks[a] * 1e-8;
    }
    results[w].t_max_individual = (double)mx * 1e-8;  // wall_clock64 ticks at 100 MHz
    results[w].t_device = h->last_kernel_s;
  }
  return CSDO_OK;
}

int csdo_dsqp_solve_batch(csdo_handle h, const csdo_problem* worlds, int32_t n_worlds, csdo_result* results) {
  const auto t0 = std::chrono::steady_clock::now();
  int rc = csdo_dsqp_upload(h, worlds, n_worlds);
  if (rc != CSDO_OK) return rc;
  if ((rc = csdo_dsqp_run(h, nullptr)) != CSDO_OK) return rc;
  if ((rc = csdo_dsqp_download(h, results, n_worlds)) != CSDO_OK) return rc;
  const double tt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int w = 0; w < n_worlds; ++w) results[w].t_total = tt;
  return CSDO_OK;
}

int csdo_dsqp_solve(csdo_handle h, const csdo_problem* in, csdo_result* out) {
  if (!in || !out) return CSDO_EINVAL;
  return csdo_dsqp_solve_batch(h, in, 1, out);
}

int csdo_generate_boxes(csdo_handle h, const double* points_xy, int32_t n, const double* obstacles, int32_t n_obs,
                        double dimx, double dimy, const csdo_vehicle* veh, double* boxes, int32_t* status) {
  if (h && h->multi) h = h->multi->kids.empty() ? nullptr : h->multi->kids[0];   // single-device work: the first device
  if (!h || !points_xy || !veh || !boxes || !status || n < 0 || n_obs < 0 || (n_obs > 0 && !obstacles))
    return CSDO_EINVAL;
  if (n == 0) return CSDO_OK;
  if ((size_t)3 * n_obs * sizeof(double) > 60 * 1024) return CSDO_ELIMIT;
  HIP_OK(hipSetDevice(h->device), CSDO_ENODEV);
  int rc;
  if ((rc = h->box_pts.ensure((size_t)n * 2 * sizeof(double))) != CSDO_OK) return rc;
  if ((rc = h->box_obs.ensure((size_t)n_obs * 3 * sizeof(double))) != CSDO_OK) return rc;
  if ((rc = h->box_out.ensure((size_t)n * 4 * sizeof(double))) != CSDO_OK) return rc;
  if ((rc = h->box_status.ensure((size_t)n * sizeof(int32_t))) != CSDO_OK) return rc;
  hipStream_t s = h->stream;
  HIP_OK(hipMemcpyAsync(h->box_pts.p, points_xy, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, s), CSDO_EDEVICE);
  if (n_obs)
    HIP_OK(hipMemcpyAsync(h->box_obs.p, obstacles, (size_t)n_obs * 3 * sizeof(double), hipMemcpyHostToDevice, s), CSDO_EDEVICE);
  if (launch_boxes((const double*)h->box_pts.p, n, (const double*)h->box_obs.p, n_obs, dimx, dimy, veh->rv,
                   (double*)h->box_out.p, (int*)h->box_status.p, s) != hipSuccess)
    return CSDO_EDEVICE;
  HIP_OK(hipMemcpyAsync(boxes, h->box_out.p, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
  HIP_OK(hipMemcpyAsync(status, h->box_status.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
  HIP_OK(hipStreamSynchronize(s), CSDO_EDEVICE);
  return CSDO_OK;
}

int csdo_math_eval(csdo_handle h, int32_t fn, const double* a, const double* b, double* out, int32_t n) {
  if (h && h->multi) h = h->multi->kids.empty() ? nullptr : h->multi->kids[0];   // single-device work: the first device
  if (!h || !a || !out || n < 0 || fn < 0 || fn > 3 || (fn == 3 && !b)) return CSDO_EINVAL;
  if (n == 0) return CSDO_OK;
  HIP_OK(hipSetDevice(h->device), CSDO_ENODEV);
  int rc;
  const size_t bytes = (size_t)n * sizeof(double);
  if ((rc = h->box_pts.ensure(2 * bytes)) != CSDO_OK) return rc;
  if ((rc = h->box_out.ensure(bytes)) != CSDO_OK) return rc;
  hipStream_t s = h->stream;
  double* da = (double*)h->box_pts.p;
  double* db = da + n;
  HIP_OK(hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, s), CSDO_EDEVICE);
  HIP_OK(hipMemcpyAsync(db, fn == 3 ? b : a, bytes, hipMemcpyHostToDevice, s), CSDO_EDEVICE);
  if (launch_math_probe(fn, da, db, (double*)h->box_out.p, n, s) != hipSuccess) return CSDO_EDEVICE;
  HIP_OK(hipMemcpyAsync(out, h->box_out.p, bytes, hipMemcpyDeviceToHost, s), CSDO_EDEVICE);
  HIP_OK(hipStreamSynchronize(s), CSDO_EDEVICE);
  return CSDO_OK;
}

int csdo_preprocess(const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                    const double* goals, const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* out) {
  return bridge_preprocess(states, actions, path_off, Na, goals, veh, parm, out);
}

void csdo_bridge_free(csdo_bridge_out* out) { bridge_free(out); }

// The bridge with its two O(Nt Na^2) stages on the device (aux_kernels.hip): interpolation and float disc centres on the
// host (O(Na Nt), glibc trigonometry exactly as the host bridge), neighbour search + plane coefficients on the device in
// the reference's (t, i, j) order, per-agent CSR assembly on the host.  Same outputs as csdo_preprocess, bit for bit.
int csdo_preprocess_device(csdo_handle h, const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                           const double* goals, const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* out) {
  if (h && h->multi) h = h->multi->kids.empty() ? nullptr : h->multi->kids[0];   // single-device work: the first device
  if (!h) return CSDO_EINVAL;
  HIP_OK(hipSetDevice(h->device), CSDO_ENODEV);
  BridgeCentres C;
  int rc = bridge_interpolate(states, actions, path_off, Na, goals, veh, parm, out, C);
  if (rc != CSDO_OK) return rc;
  const int Nt = C.Nt;
  const size_t NN = (size_t)Na * Nt;
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(h->stream);   // copies enqueued from C's (pageable, local) vectors may still be reading them
    bridge_free(out);
    return code;
  };
  if ((rc = h->k0_centres.ensure(8 * NN * sizeof(float))) != CSDO_OK) return fail(rc);
  if ((rc = h->k0_counts.ensure(NN * sizeof(int))) != CSDO_OK) return fail(rc);
  if ((rc = h->k0_offsets.ensure((NN + 1) * sizeof(long long))) != CSDO_OK) return fail(rc);
  if ((rc = h->k0_flag.ensure(sizeof(int))) != CSDO_OK) return fail(rc);
  hipStream_t s = h->stream;
  float* base = (float*)h->k0_centres.p;
  const std::vector<float>* src[8] = {&C.xf, &C.yf, &C.xr, &C.yr, &C.xc, &C.yc, &C.cs, &C.sn};
  for (int k = 0; k < 8; ++k)
    if (hipMemcpyAsync(base + k * NN, src[k]->data(), NN * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess)
      return fail(CSDO_EDEVICE);
  K0Centres kc{base, base + NN, base + 2 * NN, base + 3 * NN, base + 4 * NN, base + 5 * NN, base + 6 * NN, base + 7 * NN};
  const double reach = 2 * std::sqrt(2) * parm->r_trust;
  const float length = (float)veh->LF + (float)veh->LB, width = (float)veh->car_width;
  if (hipMemsetAsync(h->k0_flag.p, 0, sizeof(int), s) != hipSuccess) return fail(CSDO_EDEVICE);
  if (k0_count(kc, Na, Nt, reach, length, width, (int*)h->k0_counts.p, (int*)h->k0_flag.p, (long long*)h->k0_offsets.p, s) !=
      hipSuccess)
    return fail(CSDO_EDEVICE);
  long long n_pairs = 0;
  int collide = 0;
  if (hipMemcpyAsync(&n_pairs, (long long*)h->k0_offsets.p + NN, sizeof(long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipMemcpyAsync(&collide, h->k0_flag.p, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return fail(CSDO_EDEVICE);
  if (n_pairs > (long long)0x7fffffff / 3) return fail(CSDO_ELIMIT);
  std::vector<int32_t> pairs((size_t)3 * n_pairs);
  std::vector<double> coef((size_t)24 * n_pairs);
  if (n_pairs > 0) {
    if ((rc = h->k0_pairs.ensure(pairs.size() * sizeof(int32_t))) != CSDO_OK) return fail(rc);
    if ((rc = h->k0_coef.ensure(coef.size() * sizeof(double))) != CSDO_OK) return fail(rc);
    if (k0_emit(kc, Na, Nt, reach, length, width, (double)(float)veh->rv, (const long long*)h->k0_offsets.p,
                (int32_t*)h->k0_pairs.p, (double*)h->k0_coef.p, s) != hipSuccess)
      return fail(CSDO_EDEVICE);
    if (hipMemcpyAsync(pairs.data(), h->k0_pairs.p, pairs.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(coef.data(), h->k0_coef.p, coef.size() * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return fail(CSDO_EDEVICE);
  }
  if ((rc = bridge_planes(C, pairs, veh, coef.data(), out)) != CSDO_OK) return fail(rc);   // (releases x0_bar as well)
  out->initial_inter_legal = collide ? 0 : 1;
  return CSDO_OK;
}

// The device bridge for a whole batch of worlds in ONE call: the per-world call above costs four host-device round trips
// (centres up, pair count down, pairs and coefficients down), which is what its time is (31.7 ms for the 60 worlds of the
// map100 set against 17 ms for the host bridge on 16 cores).  Here every world is interpolated by a pool of host threads,
// all centres go up in one copy from page-locked memory, the count / scan kernels of all worlds are enqueued back to back,
// ONE synchronisation fetches the 60 pair counts, the emit kernels follow, ONE copy brings all pairs and coefficients
// back, and the per-agent CSR assembly runs on the pool again.  Outputs equal csdo_preprocess's, bit for bit, per world.
int csdo_preprocess_device_batch(csdo_handle h, int32_t n_worlds, const double* const* states, const int32_t* const* actions,
                                 const int32_t* const* path_off, const int32_t* Na, const double* const* goals,
                                 const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* outs) {
  if (h && h->multi) h = h->multi->kids.empty() ? nullptr : h->multi->kids[0];   // single-device work: the first device
  if (!h || n_worlds < 1 || !states || !actions || !path_off || !Na || !goals || !veh || !parm || !outs) return CSDO_EINVAL;
  for (int w = 0; w < n_worlds; ++w) std::memset(&outs[w], 0, sizeof(csdo_bridge_out));   // before anything can throw: the catch below frees them
  HIP_OK(hipSetDevice(h->device), CSDO_ENODEV);
  try {
    const bool timing = std::getenv("CSDO_BRIDGE_TIMING") != nullptr;   // diagnostic: stage times to stderr
    double t_mark = now_s();
    auto lap = [&](const char* what) {
      if (!timing) return;
      const double t = now_s();
      std::fprintf(stderr, "[bridge batch] %-28s %.3f ms\n", what, (t - t_mark) * 1e3);
      t_mark = t;
    };
    std::vector<BridgeCentres> C((size_t)n_worlds);
    std::vector<int> rcs((size_t)n_worlds, CSDO_OK);
    // (parallel_for, batch_pack.h: a body that throws or a thread that cannot be created never unwinds through joinable threads)
    auto pool = [&](auto&& body) {
      if (parallel_for(n_worlds, 64, body) != CSDO_OK)
        for (int w = 0; w < n_worlds; ++w)
          if (rcs[w] == CSDO_OK) rcs[w] = CSDO_ENOMEM;
    };
    auto fail = [&](int code) {
      (void)hipStreamSynchronize(h->stream);   // pending copies may still read the staging buffers / write the results
      for (int w = 0; w < n_worlds; ++w) bridge_free(&outs[w]);
      return code;
    };
    pool([&](int w) { rcs[w] = bridge_interpolate(states[w], actions[w], path_off[w], Na[w], goals[w], veh, parm, &outs[w], C[w]); });
    for (int w = 0; w < n_worlds; ++w)
      if (rcs[w] != CSDO_OK) return fail(rcs[w]);
    lap("interpolate (host pool)");
    // ---- layout: world w's eight centre arrays at cen_off[w] + k * NN[w]; its counts at cnt_off[w]; its offsets (NN + 1) at off_off[w]
    std::vector<size_t> NN((size_t)n_worlds), cen_off((size_t)n_worlds + 1, 0), off_off((size_t)n_worlds + 1, 0);
    for (int w = 0; w < n_worlds; ++w) {
      NN[w] = (size_t)C[w].Na * (size_t)C[w].Nt;
      cen_off[w + 1] = cen_off[w] + 8 * NN[w];
      off_off[w + 1] = off_off[w] + NN[w] + 1;
    }
    int rc;
    if ((rc = h->k0_centres.ensure(cen_off[n_worlds] * sizeof(float))) != CSDO_OK) return fail(rc);
    if ((rc = h->k0_counts.ensure(off_off[n_worlds] * sizeof(int))) != CSDO_OK) return fail(rc);
    if ((rc = h->k0_offsets.ensure(off_off[n_worlds] * sizeof(long long))) != CSDO_OK) return fail(rc);
    if ((rc = h->k0_flag.ensure((size_t)n_worlds * sizeof(int))) != CSDO_OK) return fail(rc);
    if ((rc = h->bridge_up.ensure(cen_off[n_worlds] * sizeof(float))) != CSDO_OK) return fail(rc);
    if ((rc = h->bridge_down.ensure((size_t)n_worlds * 16)) != CSDO_OK) return fail(rc);
    float* up = (float*)h->bridge_up.p;
    pool([&](int w) {
      const std::vector<float>* src[8] = {&C[w].xf, &C[w].yf, &C[w].xr, &C[w].yr, &C[w].xc, &C[w].yc, &C[w].cs, &C[w].sn};
      for (int k = 0; k < 8; ++k) std::memcpy(up + cen_off[w] + k * NN[w], src[k]->data(), NN[w] * sizeof(float));
    });
    for (int w = 0; w < n_worlds; ++w)
      if (rcs[w] != CSDO_OK) return fail(rcs[w]);
    lap("stage centres");
    hipStream_t s = h->stream;
    HIP_OK(hipMemcpyAsync(h->k0_centres.p, up, cen_off[n_worlds] * sizeof(float), hipMemcpyHostToDevice, s), fail(CSDO_EDEVICE));
    HIP_OK(hipMemsetAsync(h->k0_flag.p, 0, (size_t)n_worlds * sizeof(int), s), fail(CSDO_EDEVICE));
    const double reach = 2 * std::sqrt(2) * parm->r_trust;
    const float length = (float)veh->LF + (float)veh->LB, width = (float)veh->car_width;
    auto centres_of = [&](int w) {
      float* base = (float*)h->k0_centres.p + cen_off[w];
      const size_t n = NN[w];
      return K0Centres{base, base + n, base + 2 * n, base + 3 * n, base + 4 * n, base + 5 * n, base + 6 * n, base + 7 * n};
    };
    long long* totals = (long long*)h->bridge_down.p;          // [n_worlds] pair counts, then [n_worlds] collision flags (int)
    int* flags = (int*)(totals + n_worlds);
    for (int w = 0; w < n_worlds; ++w) {
      if (k0_count(centres_of(w), C[w].Na, C[w].Nt, reach, length, width, (int*)h->k0_counts.p + off_off[w],
                   (int*)h->k0_flag.p + w, (long long*)h->k0_offsets.p + off_off[w], s) != hipSuccess)
        return fail(CSDO_EDEVICE);
      HIP_OK(hipMemcpyAsync(&totals[w], (long long*)h->k0_offsets.p + off_off[w] + NN[w], sizeof(long long), hipMemcpyDeviceToHost, s),
             fail(CSDO_EDEVICE));
    }
    HIP_OK(hipMemcpyAsync(flags, h->k0_flag.p, (size_t)n_worlds * sizeof(int), hipMemcpyDeviceToHost, s), fail(CSDO_EDEVICE));
    HIP_OK(hipStreamSynchronize(s), fail(CSDO_EDEVICE));
    lap("H2D + count + scan + sync");
    std::vector<size_t> pair_off((size_t)n_worlds + 1, 0);
    std::vector<int> collide((size_t)n_worlds);
    for (int w = 0; w < n_worlds; ++w) {
      if (totals[w] > (long long)0x7fffffff / 3) return fail(CSDO_ELIMIT);
      pair_off[w + 1] = pair_off[w] + (size_t)totals[w];
      collide[w] = flags[w];
    }
    const size_t n_all = pair_off[n_worlds];
    const size_t b_pairs = n_all * 3 * sizeof(int32_t), b_coef = n_all * 24 * sizeof(double);
    const size_t o_coef = (b_pairs + 255) & ~(size_t)255;
    if (n_all > 0) {
      if ((rc = h->k0_pairs.ensure(b_pairs)) != CSDO_OK) return fail(rc);
      if ((rc = h->k0_coef.ensure(b_coef)) != CSDO_OK) return fail(rc);
      if ((rc = h->bridge_down.ensure(o_coef + b_coef)) != CSDO_OK) return fail(rc);
      for (int w = 0; w < n_worlds; ++w) {
        if (pair_off[w + 1] == pair_off[w]) continue;
        if (k0_emit(centres_of(w), C[w].Na, C[w].Nt, reach, length, width, (double)(float)veh->rv,
                    (const long long*)h->k0_offsets.p + off_off[w], (int32_t*)h->k0_pairs.p + 3 * pair_off[w],
                    (double*)h->k0_coef.p + 24 * pair_off[w], s) != hipSuccess)
          return fail(CSDO_EDEVICE);
      }
      HIP_OK(hipMemcpyAsync(h->bridge_down.p, h->k0_pairs.p, b_pairs, hipMemcpyDeviceToHost, s), fail(CSDO_EDEVICE));
      HIP_OK(hipMemcpyAsync((char*)h->bridge_down.p + o_coef, h->k0_coef.p, b_coef, hipMemcpyDeviceToHost, s), fail(CSDO_EDEVICE));
      HIP_OK(hipStreamSynchronize(s), fail(CSDO_EDEVICE));
    }
    lap("emit + D2H + sync");
    const int32_t* pairs_all = (const int32_t*)h->bridge_down.p;
    const double* coef_all = (const double*)((const char*)h->bridge_down.p + o_coef);
    pool([&](int w) {
      const size_t n = pair_off[w + 1] - pair_off[w];
      rcs[w] = bridge_planes(C[w], pairs_all + 3 * pair_off[w], n, veh, coef_all + 24 * pair_off[w], &outs[w]);
      outs[w].initial_inter_legal = collide[w] ? 0 : 1;
    });
    for (int w = 0; w < n_worlds; ++w)
      if (rcs[w] != CSDO_OK) return fail(rcs[w]);
    lap("planes (host pool)");
    return CSDO_OK;
  } catch (...) {   // (allocation of the bookkeeping vectors: every outs[w] released and zeroed, as on the other error paths)
    (void)hipStreamSynchronize(h->stream);
    for (int w = 0; w < n_worlds; ++w) bridge_free(&outs[w]);
    return CSDO_ENOMEM;
  }
}

// The host bridge for a batch of worlds: csdo_preprocess per world on a pool of host threads (no device work: it runs beside a
// solve that occupies every CU - the streamed DO phase prepares its next chunk of worlds with it).  Outputs as csdo_preprocess.
int csdo_preprocess_batch(int32_t n_worlds, const double* const* states, const int32_t* const* actions,
                          const int32_t* const* path_off, const int32_t* Na, const double* const* goals,
                          const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* outs) {
  if (n_worlds < 1 || !states || !actions || !path_off || !Na || !goals || !veh || !parm || !outs) return CSDO_EINVAL;
  for (int w = 0; w < n_worlds; ++w) std::memset(&outs[w], 0, sizeof(csdo_bridge_out));   // before anything can throw
  try {
    std::vector<int> rcs((size_t)n_worlds, CSDO_OK);
    int rc = parallel_for(n_worlds, 64, [&](int w) {
      rcs[w] = bridge_preprocess(states[w], actions[w], path_off[w], Na[w], goals[w], veh, parm, &outs[w]);
    });
    for (int w = 0; w < n_worlds && rc == CSDO_OK; ++w) rc = rcs[w];
    if (rc != CSDO_OK)
      for (int w = 0; w < n_worlds; ++w) bridge_free(&outs[w]);
    return rc;
  } catch (...) {
    for (int w = 0; w < n_worlds; ++w) bridge_free(&outs[w]);
    return CSDO_ENOMEM;
  }
}

// Independent geometric check of final trajectories on the device: vehicle rectangles against each other per timestep
// (separating axes, touching counts) and against the obstacle discs, optional map bounds (dimx <= 0: skipped).
static int validate_impl(csdo_handle h, const double* solutions, int32_t Na, int32_t Nt, int32_t frames_per_move,
                         const double* obstacles, int32_t n_obs, double dimx, double dimy, const csdo_vehicle* veh, double margin,
                         csdo_validation* out) {
  if (h && h->multi) h = h->multi->kids.empty() ? nullptr : h->multi->kids[0];   // single-device work: the first device
  if (!h || !solutions || !veh || !out || Na < 1 || Nt < 1 || n_obs < 0 || (n_obs > 0 && !obstacles) || frames_per_move < 0)
    return CSDO_EINVAL;
  // frames_per_move == 0: the Nt states as they are; S >= 1: the (Nt - 1) * S + 1 frames of the authors' animation
  const long long n_frames = frames_per_move ? ((long long)(Nt - 1) * frames_per_move + 1) : Nt;
  if (Na >= (1 << 20) || n_frames >= (1 << 20) || n_obs >= (1 << 20)) return CSDO_ELIMIT;
  HIP_OK(hipSetDevice(h->device), CSDO_ENODEV);
  int rc;
  const size_t b_sol = (size_t)Na * Nt * 6 * sizeof(double), b_obs = (size_t)n_obs * 3 * sizeof(double);
  if ((rc = h->val_sol.ensure(b_sol)) != CSDO_OK) return rc;
  if ((rc = h->val_obs.ensure(b_obs)) != CSDO_OK) return rc;
  if ((rc = h->val_out.ensure(6 * sizeof(unsigned long long))) != CSDO_OK) return rc;
  if (frames_per_move && (rc = h->val_frames.ensure((size_t)Na * n_frames * 6 * sizeof(double))) != CSDO_OK) return rc;
  hipStream_t s = h->stream;
  // (from here on copies out of the caller's pageable arrays and out of `init` below are in flight: no return without draining the stream)
  auto drained = [&](int code) {
    (void)hipStreamSynchronize(s);
    return code;
  };
  const unsigned long long none = ~0ull;
  unsigned long long init[6] = {0ull, none, 0ull, none, 0ull, none}, res[6];
  HIP_OK(hipMemcpyAsync(h->val_sol.p, solutions, b_sol, hipMemcpyHostToDevice, s), drained(CSDO_EDEVICE));
  if (n_obs) HIP_OK(hipMemcpyAsync(h->val_obs.p, obstacles, b_obs, hipMemcpyHostToDevice, s), drained(CSDO_EDEVICE));
  HIP_OK(hipMemcpyAsync(h->val_out.p, init, sizeof(init), hipMemcpyHostToDevice, s), drained(CSDO_EDEVICE));
  const double* poses = (const double*)h->val_sol.p;
  if (frames_per_move) {
    if (expand_frames_launch((const double*)h->val_sol.p, Na, Nt, frames_per_move, (double*)h->val_frames.p, s) != hipSuccess)
      return drained(CSDO_EDEVICE);
    poses = (const double*)h->val_frames.p;
  }
  const double half_shift = 0.5 * (veh->LF - veh->LB), hl = 0.5 * (veh->LF + veh->LB) + margin, hw = 0.5 * veh->car_width + margin;
  if (validate_launch(poses, Na, (int)n_frames, (const double*)h->val_obs.p, n_obs, half_shift, hl, hw, dimx, dimy,
                      dimx > 0 && dimy > 0, (unsigned long long*)h->val_out.p, s) != hipSuccess)
    return drained(CSDO_EDEVICE);
  HIP_OK(hipMemcpyAsync(res, h->val_out.p, sizeof(res), hipMemcpyDeviceToHost, s), drained(CSDO_EDEVICE));
  HIP_OK(hipStreamSynchronize(s), CSDO_EDEVICE);
  out->vehicle_collisions = (int64_t)res[0];
  out->obstacle_collisions = (int64_t)res[2];
  out->out_of_map = (int64_t)res[4];
  for (int k = 0; k < 3; ++k) out->first_vehicle[k] = out->first_obstacle[k] = -1;
  if (res[0]) {
    out->first_vehicle[0] = (int32_t)(res[1] >> 40);
    out->first_vehicle[1] = (int32_t)((res[1] >> 20) & 0xfffff);
    out->first_vehicle[2] = (int32_t)(res[1] & 0xfffff);
  }
  if (res[2]) {
    out->first_obstacle[0] = (int32_t)(res[3] >> 40);
    out->first_obstacle[1] = (int32_t)((res[3] >> 20) & 0xfffff);
    out->first_obstacle[2] = (int32_t)(res[3] & 0xfffff);
  }
  out->min_obstacle_clearance = INFINITY;
  if (n_obs > 0 && res[5] != none) {
    const unsigned long long key = res[5];
    const unsigned long long bits = (key & 0x8000000000000000ull) ? (key & 0x7fffffffffffffffull) : ~key;
    double d;
    std::memcpy(&d, &bits, sizeof(d));
    out->min_obstacle_clearance = d;
  }
  return CSDO_OK;
}

int csdo_validate(csdo_handle h, const double* solutions, int32_t Na, int32_t Nt, const double* obstacles, int32_t n_obs,
                  double dimx, double dimy, const csdo_vehicle* veh, double margin, csdo_validation* out) {
  return validate_impl(h, solutions, Na, Nt, 0, obstacles, n_obs, dimx, dimy, veh, margin, out);
}

// The same check on the frames of the authors' animation: frames_per_move frames per move (framesPerMove of
// scripts/visualize.py:27,186), poses between two states interpolated as getState does (:256-281).  Indices in `out` are frames.
int csdo_validate_frames(csdo_handle h, const double* solutions, int32_t Na, int32_t Nt, int32_t frames_per_move,
                         const double* obstacles, int32_t n_obs, double dimx, double dimy, const csdo_vehicle* veh,
                         double margin, csdo_validation* out) {
  if (frames_per_move < 1) return CSDO_EINVAL;
  return validate_impl(h, solutions, Na, Nt, frames_per_move, obstacles, n_obs, dimx, dimy, veh, margin, out);
}

void csdo_vehicle_default(csdo_vehicle* v) {
  if (!v) return;
  // config.yaml:4-27 through readAgentConfig (common/motion_planning.cc:64-85): YAML doubles stored in floats
  const float r = 3.0f, deltat = 0.706f, W = 2.0f, LF = 2.0f, LB = 1.0f, WB = 1.0f, obsR = 0.8f;
  const float f2x = (float)(1 / 4.0 * (3.0 * LF - LB));
  const float r2x = (float)(1 / 4.0 * (LF - 3.0 * LB));
  const float rv = (float)(1.0 / 2.0 * std::pow(std::pow(LF + LB, 2) / 4 + W * W, 0.5));
  v->r = r;
  v->deltat = deltat;
  v->LF = LF;
  v->LB = LB;
  v->car_width = W;
  v->WB = WB;
  v->f2x = f2x;
  v->r2x = r2x;
  v->rv = rv;
  v->obs_radius = obsR;
}

void csdo_qp_parm_default(const csdo_vehicle* v, csdo_qp_parm* p) {
  if (!v || !p) return;
  std::memset(p, 0, sizeof(*p));
  p->r_trust = 2.0;
  p->max_omega = 0.07;
  p->max_v = 1.0;
  p->max_iter = 10;
  p->delta_solution_threshold = 1.0;
  p->max_violation = 0.001;
  p->osqp_max_iter = 400;
  p->num_interpolation = 2;
  const float step = (float)v->r * (float)v->deltat;  // sqp/utils.cc:55-56: float product, double divisions
  p->dt = step / p->max_v / (p->num_interpolation + 1) / 0.8;
  p->fixed_corridor = 0;
  p->adaptive_rho_interval = 25;
  p->solve_refinement = 0;
  p->_reserved = 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The DO phase of csdo.cc:111-148 for a batch of worlds in ONE call: coarse paths in, trajectories out.
int32_t csdo_do_phase_horizon(const int32_t* path_off, int32_t Na, const csdo_qp_parm* parm) {
  if (!path_off || !parm || Na < 1) return CSDO_EINVAL;
  int longest = 0;
  for (int a = 0; a < Na; ++a) longest = std::max(longest, path_off[a + 1] - path_off[a]);
  if (longest < 1) return CSDO_EINVAL;
  // sqp/inter_agent_cons.cc:320-325: the longest path, num_interpolation points inserted per move
  return (int32_t)((longest - 1) * (parm->num_interpolation + 1) + 1);
}

extern "C++" {
// Chunk boundaries of a streamed DO phase (solver.py: stream_cuts is the same rule, tests/test_do_phase.py holds them together):
// growing chunks 8 % / 27 % / 65 % of the worlds, the first enlarged until it holds ~230 agents - it has to fill the GPU by itself -,
// and a job that such a first chunk takes a fifth of is cut in two.
static std::vector<int> do_phase_cuts(const std::vector<int>& agents_per_world, const int min_first_agents) {
  const int n = (int)agents_per_world.size();
  const double fr_all[3] = {0.08, 0.27, 0.65};
  const int nf = std::min(3, n);
  const double* fr = fr_all + (3 - nf);
  double tot = 0.0, acc = 0.0;
  for (int c = 0; c < nf; ++c) tot += fr[c];
  std::vector<int> cuts{0};
  for (int c = 0; c < nf; ++c) {
    acc += fr[c];
    const int hi = c == nf - 1 ? n : std::max(cuts.back() + 1, std::min(n - (nf - 1 - c), (int)std::nearbyint(n * acc / tot)));
    cuts.push_back(hi);
  }
  long long run = 0;
  int fill = n + 1;   // worlds until min_first_agents agents are reached, + 1 (numpy.searchsorted(cumsum, m) + 1)
  for (int w = 0; w < n; ++w) {
    run += agents_per_world[w];
    if (run >= min_first_agents) {
      fill = w + 1;
      break;
    }
  }
  if (cuts.size() > 2 && cuts[1] < fill) {
    if (fill >= n) cuts = {0, n};
    else if (fill >= 0.2 * n) cuts = {0, fill, n};
    else {
      std::vector<int> c2{0, fill};
      for (size_t k = 2; k + 1 < cuts.size(); ++k) c2.push_back(std::max(cuts[k], fill + (int)(k - 2) + 1));
      c2.push_back(n);
      cuts = c2;
    }
  }
  return cuts;
}
}  // extern "C++"

int csdo_do_phase_cuts(const int32_t* agents_per_world, int32_t n_worlds, int32_t min_first_agents, int32_t* cuts /* [5] */) {
  if (!agents_per_world || n_worlds < 1 || !cuts) return CSDO_EINVAL;
  return guarded([&]() {
    const std::vector<int> c = do_phase_cuts(std::vector<int>(agents_per_world, agents_per_world + n_worlds), min_first_agents);
    for (size_t k = 0; k < 5; ++k) cuts[k] = k < c.size() ? c[k] : -1;
    return (int)c.size() - 1;
  });
}

int csdo_do_phase(csdo_handle h, const csdo_coarse_world* worlds, int32_t n_worlds, const csdo_vehicle* veh, const csdo_qp_parm* parm,
                  csdo_result* results, int32_t* initial_inter_legal, csdo_do_phase_timing* timing) {
  if (!h || h->borrowed || h->run_pending || !worlds || n_worlds < 1 || !veh || !parm || !results) return CSDO_EINVAL;
  if (h->multi) {
    // Several GPUs: the WORLDS are dealt out - contiguous runs of equal weight (agents x horizon), one per device -, every device runs
    // its own DO phase on its own host thread, results land in the caller's arrays directly.  No collective: worlds are independent.
    return guarded([&]() -> int {
      const double t0 = now_s();
      const std::vector<csdo_handle>& kids = h->multi->kids;
      const int K = (int)kids.size();
      std::vector<double> weight((size_t)n_worlds);
      double total = 0.0;
      for (int w = 0; w < n_worlds; ++w) {
        if (!worlds[w].path_off || worlds[w].Na < 1) return CSDO_EINVAL;
        const int nt = csdo_do_phase_horizon(worlds[w].path_off, worlds[w].Na, parm);
        if (nt < 2) return CSDO_EINVAL;
        weight[w] = (double)worlds[w].Na * nt;
        total += weight[w];
      }
      std::vector<int> cut((size_t)K + 1, n_worlds);
      cut[0] = 0;
      double run = 0.0;
      for (int w = 0, k = 1; w < n_worlds && k < K; ++w) {
        run += weight[w];
        while (k < K && run >= total * k / K) cut[k++] = w + 1;
      }
      std::vector<int> rcs((size_t)K, CSDO_OK);
      std::vector<csdo_do_phase_timing> tms((size_t)K);
      {
        struct Joiner {
          std::vector<std::thread> t;
          ~Joiner() {
            for (auto& x : t)
              if (x.joinable()) x.join();
          }
        } pool;
        auto part = [&](const int k) {
          const int n_k = cut[k + 1] - cut[k];
          if (n_k > 0)
            rcs[k] = csdo_do_phase(kids[k], worlds + cut[k], n_k, veh, parm, results + cut[k],
                                   initial_inter_legal ? initial_inter_legal + cut[k] : nullptr, &tms[k]);
        };
        for (int k = 1; k < K; ++k) {
          try {
            pool.t.emplace_back(part, k);
          } catch (...) {
            part(k);
          }
        }
        const DeviceGuard keep_callers_device;
        part(0);
      }
      for (int k = 0; k < K; ++k)
        if (rcs[k] != CSDO_OK) return rcs[k];
      csdo_do_phase_timing T = tms[0];   // the first device's chunks; the batch is done when the last device is
      for (int k = 1; k < K; ++k) {
        T.kernels_done = std::max(T.kernels_done, tms[k].kernels_done);
        T.first_launch = std::max(T.first_launch, tms[k].first_launch);
      }
      T.total = now_s() - t0;
      for (int w = 0; w < n_worlds; ++w) results[w].t_total = T.total;
      if (timing) *timing = T;
      return CSDO_OK;
    });
  }
  return guarded([&]() -> int {
    const double t0 = now_s();
    csdo_do_phase_timing T{};
    std::vector<int> agents((size_t)n_worlds), horizon((size_t)n_worlds);
    bool one_class = true;   // horizons 129 .. 234 run in ONE kernel class (512 threads, everything in LDS): only such a job is streamed
    for (int w = 0; w < n_worlds; ++w) {
      const csdo_coarse_world& W = worlds[w];
      if (!W.states || !W.actions || !W.path_off || !W.goals || W.Na < 1 || W.n_obs < 0 || (W.n_obs > 0 && !W.obstacles)) return CSDO_EINVAL;
      agents[w] = W.Na;
      horizon[w] = csdo_do_phase_horizon(W.path_off, W.Na, parm);
      if (horizon[w] < 2) return CSDO_EINVAL;
      one_class = one_class && horizon[w] > 128 && horizon[w] <= 234;
    }
    std::vector<int> cuts = one_class ? do_phase_cuts(agents, 230) : std::vector<int>{0, n_worlds};
    // bridge of worlds [w0, w1) on the host threads, then the csdo_problem views of its outputs
    struct Bridged {
      std::vector<csdo_bridge_out> outs;
      std::vector<csdo_problem> probs;
      ~Bridged() {
        for (auto& o : outs) bridge_free(&o);
      }
    };
    auto bridge = [&](const int w0, const int w1, Bridged& B) -> int {
      const int n = w1 - w0;
      B.outs.assign((size_t)n, csdo_bridge_out{});
      std::vector<const double*> st((size_t)n), go((size_t)n);
      std::vector<const int32_t*> ac((size_t)n), po((size_t)n);
      std::vector<int32_t> na((size_t)n);
      for (int k = 0; k < n; ++k) {
        const csdo_coarse_world& W = worlds[w0 + k];
        st[k] = W.states; ac[k] = W.actions; po[k] = W.path_off; go[k] = W.goals; na[k] = W.Na;
      }
      const int rc = csdo_preprocess_batch(n, st.data(), ac.data(), po.data(), na.data(), go.data(), veh, parm, B.outs.data());
      if (rc != CSDO_OK) {
        B.outs.clear();   // (csdo_preprocess_batch has released what it had made)
        return rc;
      }
      B.probs.assign((size_t)n, csdo_problem{});
      for (int k = 0; k < n; ++k) {
        const csdo_coarse_world& W = worlds[w0 + k];
        const csdo_bridge_out& o = B.outs[(size_t)k];
        if (o.Nt != horizon[w0 + k]) return CSDO_EINVAL;   // (the caller sized its result arrays by csdo_do_phase_horizon)
        csdo_problem& P = B.probs[(size_t)k];
        P.Na = o.Na; P.Nt = o.Nt; P.x0_bar = o.x0_bar; P.plane_off = o.plane_off; P.planes = o.planes;
        P.dimx = W.dimx; P.dimy = W.dimy; P.n_obs = W.n_obs; P.obstacles = W.obstacles; P.veh = *veh; P.parm = *parm;
        if (initial_inter_legal) initial_inter_legal[w0 + k] = o.initial_inter_legal;
      }
      return CSDO_OK;
    };
    auto one_launch = [&](Bridged& first, const int w_first_end) -> int {   // everything on this handle; `first` = worlds [0, w_first_end) bridged
      Bridged rest;
      int rc;
      const double tb = now_s();
      if (w_first_end < n_worlds && (rc = bridge(w_first_end, n_worlds, rest)) != CSDO_OK) return rc;
      std::vector<csdo_problem> all(first.probs);
      all.insert(all.end(), rest.probs.begin(), rest.probs.end());
      const double tu = now_s();
      const bool was = h->host_results;
      h->host_results = true;
      rc = upload_impl(h, all.data(), n_worlds);
      h->host_results = was;
      if (rc != CSDO_OK) return rc;
      const double tr = now_s();
      if ((rc = csdo_dsqp_run(h, nullptr)) != CSDO_OK) return rc;
      T.kernels_done = now_s() - t0;
      if ((rc = download_impl(h, results, n_worlds)) != CSDO_OK) return rc;
      T.first_launch = tr - t0;
      T.n_chunks = 1;
      T.streamed = 0;
      T.chunk_worlds[0] = n_worlds;
      T.chunk_bridge[0] += tu - tb;
      T.chunk_upload[0] = tr - tu;
      T.chunk_kernel[0] = h->last_kernel_s;
      return CSDO_OK;
    };
    int rc = CSDO_OK;
    if (cuts.size() == 2) {
      Bridged all;
      const double tb = now_s();
      if ((rc = bridge(0, n_worlds, all)) != CSDO_OK) return rc;
      T.chunk_bridge[0] = now_s() - tb;
      rc = one_launch(all, n_worlds);
    } else {
      const int n_chunks = (int)cuts.size() - 1;
      std::vector<csdo_handle> flying;
      // a chunk that fails (an obstacle-heavy world hits CSDO_ELIMIT at its upload) leaves the earlier ones running: collect them
      auto drain = [&](int code) {
        for (csdo_handle k : flying) (void)csdo_dsqp_wait(k);
        return code;
      };
      int next_lane = 0;
      bool done = false;
      for (int c = 0; c < n_chunks && !done; ++c) {
        if (!h->stream_kids[c] && (rc = csdo_dsqp_create_shared(&h->stream_kids[c], h, c)) != CSDO_OK) return drain(rc);
        csdo_handle kid = h->stream_kids[c];
        if (kid->run_pending) return drain(CSDO_EINVAL);
        kid->host_results = true;   // the last chunk's D2H copy is the one nothing would hide
        kid->min_mode = h->min_mode;
        Bridged B;
        const double tb = now_s();
        if ((rc = bridge(cuts[c], cuts[c + 1], B)) != CSDO_OK) return drain(rc);
        const double tu = now_s();
        if ((rc = upload_impl(kid, B.probs.data(), cuts[c + 1] - cuts[c])) != CSDO_OK) return drain(rc);
        const double tr = now_s();
        const int ng = (int)kid->groups.size();
        if (c == 0 && ng > 1) {
          // several kernel classes after all (plane counts decide too): chunks of such launches in flight at once fragment the CUs
          // (measured: map50 87 ms streamed against 59, room50 115 against 68) - one launch of everything, on this handle
          T.chunk_bridge[0] = tu - tb;
          rc = one_launch(B, cuts[1]);
          done = true;
          break;
        }
        if ((rc = csdo_dsqp_set_lane(kid, next_lane)) != CSDO_OK) return drain(rc);
        next_lane = (next_lane + std::max(ng, 1)) % 4;
        if ((rc = csdo_dsqp_run_async(kid, nullptr)) != CSDO_OK) return drain(rc);
        flying.push_back(kid);
        if (c == 0) T.first_launch = now_s() - t0;
        T.chunk_worlds[c] = cuts[c + 1] - cuts[c];
        T.chunk_bridge[c] = tu - tb;
        T.chunk_upload[c] = tr - tu;
      }
      if (!done) {
        T.n_chunks = n_chunks;
        T.streamed = 1;
        for (int c = 0; c < n_chunks; ++c) {
          csdo_handle kid = flying[(size_t)c];
          int rc_c = csdo_dsqp_wait(kid);
          T.kernels_done = now_s() - t0;
          T.chunk_kernel[c] = kid->last_kernel_s;
          if (rc_c == CSDO_OK) rc_c = download_impl(kid, results + cuts[c], cuts[c + 1] - cuts[c]);
          if (rc_c != CSDO_OK && rc == CSDO_OK) rc = rc_c;   // (the later chunks are still collected)
        }
      }
    }
    if (rc != CSDO_OK) return rc;
    T.total = now_s() - t0;
    for (int w = 0; w < n_worlds; ++w) results[w].t_total = T.total;
    if (timing) *timing = T;
    return CSDO_OK;
  });
}

#if defined(CSDO_PROFILE_PHASES)
// diagnostic build only (libcsdo_hip_prof.so): per-agent shader-clock ticks per phase and per-agent wall ticks
int csdo_debug_phase_ticks(csdo_handle h, int64_t* phases16, int64_t* agent_ticks) {
  if (!h || h->multi || !h->uploaded) return CSDO_EINVAL;
  const size_t Na = h->hb.agents.size();
  if (hipMemcpy(phases16, h->prof.p, Na * 48 * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return CSDO_EDEVICE;
  if (hipMemcpy(agent_ticks, h->dev.agent_ticks, Na * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return CSDO_EDEVICE;
  return CSDO_OK;
}
#endif

}  // extern "C"
