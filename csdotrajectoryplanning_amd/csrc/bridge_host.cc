// bridge_host.cc — see bridge_host.h.  Flat-array implementation; every float cast mirrors a `float` member or a
// float-returning helper of the reference (State::xf.., Constants::*, normalizeAngleAbsInPi, atan(float)).
#include "bridge_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "host_pool.h"

namespace csdo {

namespace {

// One world's bridge is cut into pieces - agents, blocks of timesteps, blocks of pairs - for up to this many of the library's host
// threads: a single 50-vehicle instance is 1.3 ms of bridge on one thread, a streamed DO phase's first chunk is five worlds on a
// machine with a few dozen cores.  The pieces write disjoint outputs; the pair list is concatenated in (t, i, j) order.
constexpr int HOST_THREADS_PER_WORLD_MAX = 16;
// ... but no more than the pool's threads divided by the worlds being bridged right now (a batch of sixty worlds on sixty-four threads
// is one thread per world: cutting those worlds into pieces as well only made the threads queue up at the pool - 60 worlds of the
// map50 set 1.7 ms against 8 ms of serial work / 64).  The stages of a world's bridge read the share through a thread-local.
std::atomic<int> worlds_in_flight{0};
thread_local int HOST_THREADS_PER_WORLD = HOST_THREADS_PER_WORLD_MAX;
struct WorldInFlight {
  int before;
  WorldInFlight() : before(HOST_THREADS_PER_WORLD) {
    const int n = worlds_in_flight.fetch_add(1) + 1;
    HOST_THREADS_PER_WORLD = std::max(1, std::min(HOST_THREADS_PER_WORLD_MAX, (HostPool::get().helpers() + 1) / n));
  }
  ~WorldInFlight() {
    worlds_in_flight.fetch_sub(1);
    HOST_THREADS_PER_WORLD = before;
  }
};

struct Veh {
  float r, LF, LB, W, f2x, r2x, rv;
};

inline float wrap_pi_float(double a) {  // common/motion_planning.h:70-75
  a = std::fmod(a + M_PI, 2 * M_PI);
  if (a < 0) a += 2 * M_PI;
  return (float)(a - M_PI);
}

// One motion segment s0 -> s1 with `n` interior samples (action_sample, inter_agent_cons.cc:194-271).
// Appends n+1 poses (the last one is the head of the next segment with a continuous yaw) and n+1 action copies.
void refine_segment(int action, const double s0[3], const double s1[3], int n, const Veh& v,
                    std::vector<double>& px, std::vector<double>& py, std::vector<double>& pyaw,
                    std::vector<int>& acts) {
  acts.insert(acts.end(), n + 1, action);
  if (action == 6) {
    for (int i = 0; i < n + 1; ++i) {
      px.push_back(s0[0]);
      py.push_back(s0[1]);
      pyaw.push_back(s0[2]);
    }
    return;
  }
  const double ex = s1[0] - s0[0], ey = s1[1] - s0[1];
  const double chord = std::sqrt(std::pow(ex, 2) + std::pow(ey, 2));
  const bool straight = (action == 0 || action == 3);
  double radius = v.r, sweep;
  if (straight) {
    sweep = chord / v.r;
  } else {
    sweep = wrap_pi_float(s1[2] - s0[2]);
    radius = chord / (2.0 * std::sin(std::fabs(sweep) / 2.0));  // arc radius re-fitted to the chord
  }
  const double da = std::fabs(sweep) / (double)(n + 1);
  // body-frame increment of one sub-step (calcActionD, :169-190)
  const bool reverse = action >= 3;
  const int turn = action % 3;  // 0 straight, 1 right, 2 left (forward); mirrored yaw sign when reversing
  double dx, dy, dyaw;
  if (turn == 0) {
    dx = radius * da;
    dy = 0;
    dyaw = 0;
  } else {
    dx = radius * std::sin(da);
    dy = radius * (1 - std::cos(da));
    if (turn == 1) dy = -dy;
    dyaw = (turn == 1) ? -da : da;
  }
  if (reverse) {
    dx = -dx;
    dyaw = -dyaw;
  }
  double cx = s0[0], cy = s0[1], cyaw = s0[2];
  for (int i = 0; i < n; ++i) {
    const double nx = cx + dx * std::cos(cyaw) - dy * std::sin(cyaw);
    const double ny = cy + dx * std::sin(cyaw) + dy * std::cos(cyaw);
    const double nyaw = cyaw + dyaw;
    px.push_back(nx);
    py.push_back(ny);
    pyaw.push_back(nyaw);
    cx = nx;
    cy = ny;
    cyaw = nyaw;
  }
  px.push_back(s1[0]);
  py.push_back(s1[1]);
  pyaw.push_back((straight ? 0.0 : sweep) + s0[2]);
}

inline double sq(float a, float b) {  // pow(float - float, 2): float difference, double square
  const float d = a - b;
  return std::pow((double)d, 2);
}

}  // namespace

int bridge_interpolate(const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                       const double* goals, const csdo_vehicle* vehp, const csdo_qp_parm* parm, csdo_bridge_out* out,
                       BridgeCentres& C) {
  if (!states || !actions || !path_off || !goals || !vehp || !parm || !out || Na < 1) return CSDO_EINVAL;
  std::memset(out, 0, sizeof(*out));
  const Veh v{(float)vehp->r, (float)vehp->LF, (float)vehp->LB, (float)vehp->car_width,
              (float)vehp->f2x, (float)vehp->r2x, (float)vehp->rv};
  const int n = parm->num_interpolation;
  const double dt = parm->dt;

  // ---- interpolation (interpolateXYYaw :277-310), the agents side by side on the library's host threads ----
  std::vector<std::vector<double>> X(Na), Y(Na), YAW(Na);
  std::vector<std::vector<int>> ACT(Na);
  for (int a = 0; a < Na; ++a)
    if (path_off[a + 1] - path_off[a] < 1) return CSDO_EINVAL;
  std::atomic<int> bad{0};
  int prc = parallel_for(Na, HOST_THREADS_PER_WORLD, [&](const int a) {
    const int L = path_off[a + 1] - path_off[a];
    const int act_base = (path_off[a] - path_off[0]) - a;   // L - 1 actions per earlier agent
    const double* S = states + (size_t)path_off[a] * 3;
    double cur[3] = {S[0], S[1], S[2]};
    if (L == 1) {  // single-state path: the goal overwrite applies to it
      cur[0] = goals[3 * a];
      cur[1] = goals[3 * a + 1];
      cur[2] = goals[3 * a + 2];
    }
    const size_t cap = (size_t)(L - 1) * (size_t)(n + 1) + 1;   // n + 1 poses per move: one allocation per array instead of a dozen
    X[a].reserve(cap);
    Y[a].reserve(cap);
    YAW[a].reserve(cap);
    ACT[a].reserve(cap);
    X[a].push_back(cur[0]);
    Y[a].push_back(cur[1]);
    YAW[a].push_back(cur[2]);
    for (int i = 0; i + 1 < L; ++i) {
      double nxt[3] = {S[3 * (i + 1)], S[3 * (i + 1) + 1], S[3 * (i + 1) + 2]};
      if (i + 1 == L - 1) {  // last state := goal (:149-151)
        nxt[0] = goals[3 * a];
        nxt[1] = goals[3 * a + 1];
        nxt[2] = goals[3 * a + 2];
      }
      const int act = actions[act_base + i];
      if (act < 0 || act > 6) {
        bad.store(1);
        return;
      }
      refine_segment(act, cur, nxt, n, v, X[a], Y[a], YAW[a], ACT[a]);
      cur[0] = X[a].back();
      cur[1] = Y[a].back();
      cur[2] = YAW[a].back();
    }
  });
  if (bad.load()) return CSDO_EINVAL;
  if (prc != CSDO_OK) return prc;
  size_t Nt = 0;
  for (int a = 0; a < Na; ++a) Nt = std::max(Nt, X[a].size());
  if (Nt < 2) return CSDO_EINVAL;

  // ---- fixed-length guess with steer / v / d_steer (calcVSteerW :315-411) ----
  out->Na = Na;
  out->Nt = (int32_t)Nt;
  out->x0_bar = (double*)std::calloc((size_t)Na * Nt * 6, sizeof(double));
  if (!out->x0_bar) return CSDO_ENOMEM;
  const double phi = (double)std::atan((v.LF - v.LB) / v.r);  // float atan (:352-353)
  // ---- ... and the float disc centres and rectangle centres of every (agent, t) (State ctor, motion_planning.h:115-132) ----
  const size_t NN = (size_t)Na * Nt;
  C.Na = Na;
  C.Nt = (int)Nt;
  for (std::vector<float>* a_ : {&C.xf, &C.yf, &C.xr, &C.yr, &C.xc, &C.yc, &C.cs, &C.sn}) a_->resize(NN);
  std::vector<float>&xf = C.xf, &yf = C.yf, &xr = C.xr, &yr = C.yr, &xc = C.xc, &yc = C.yc, &cs = C.cs, &sn = C.sn;
  const float c2r = (v.LF + v.LB) / 2 - v.LB;
  prc = parallel_for(Na, HOST_THREADS_PER_WORLD, [&](const int a) {
    double* g = out->x0_bar + (size_t)a * Nt * 6;
    const size_t xs = X[a].size();
    for (size_t i = 0; i < Nt; ++i) {
      const size_t k = std::min(i, xs - 1);
      g[i * 6 + 0] = X[a][k];
      g[i * 6 + 1] = Y[a][k];
      g[i * 6 + 2] = YAW[a][k];
    }
    for (size_t i = 1; i < xs; ++i) {
      const int act = ACT[a][i - 1];
      g[i * 6 + 3] = (act == 1 || act == 4) ? -phi : ((act == 2 || act == 5) ? phi : 0.0);
    }
    for (size_t i = 0; i + 1 < xs; ++i) {
      const double yaw = g[i * 6 + 2];
      g[i * 6 + 4] = ((g[(i + 1) * 6] - g[i * 6]) / dt) * std::cos(yaw) + ((g[(i + 1) * 6 + 1] - g[i * 6 + 1]) / dt) * std::sin(yaw);
      g[i * 6 + 5] = (g[(i + 1) * 6 + 3] - g[i * 6 + 3]) / dt;
    }
    for (size_t k = (size_t)a * Nt; k < (size_t)(a + 1) * Nt; ++k) {
      const double x = out->x0_bar[k * 6], y = out->x0_bar[k * 6 + 1], yaw = out->x0_bar[k * 6 + 2];
      const double c = std::cos(yaw), s = std::sin(yaw);
      xf[k] = (float)(x + v.f2x * c);
      xr[k] = (float)(x + v.r2x * c);
      yf[k] = (float)(y + v.f2x * s);
      yr[k] = (float)(y + v.r2x * s);
      xc[k] = (float)(x + c2r * c);
      yc[k] = (float)(y + c2r * s);
      cs[k] = (float)c;
      sn[k] = (float)s;
    }
  });
  if (prc != CSDO_OK) return prc;

  return CSDO_OK;
}

// findNeighborPairsByTrustRegion (:12-49) on the float centres: pairs (t, i, j) in that order; false if two rectangles overlap
bool bridge_pairs(const BridgeCentres& C, double r_trust, const csdo_vehicle* vehp, std::vector<int32_t>& pairs) {
  const Veh v{(float)vehp->r, (float)vehp->LF, (float)vehp->LB, (float)vehp->car_width,
              (float)vehp->f2x, (float)vehp->r2x, (float)vehp->rv};
  const int Na = C.Na;
  const size_t Nt = (size_t)C.Nt;
  const std::vector<float>&xf = C.xf, &yf = C.yf, &xr = C.xr, &yr = C.yr, &xc = C.xc, &yc = C.yc, &cs = C.cs, &sn = C.sn;
  csdo_qp_parm parm_{};
  parm_.r_trust = r_trust;
  const csdo_qp_parm* parm = &parm_;
  pairs.clear();
  // ---- neighbour pairs (findNeighborPairsByTrustRegion :12-49), order (t, i, j): blocks of timesteps side by side ----
  const double reach = 2 * std::sqrt(2) * parm->r_trust;
  const float length = v.LF + v.LB, width = v.W;
  // A pair whose rectangle centres are further apart than the reach plus what the four disc centres can lie off the rectangle centres
  // (c2r to the front disc, c2r to the rear one; 1e-3 for the float roundings of the centres) cannot have a disc pair within reach: one
  // distance instead of four for the nine pairs in ten that are nowhere near each other (same pair list; round 6: the pair search is
  // the larger half of the 1.8 ms a single world's host bridge takes, of the 11 ms of a 50-agent instance's DO phase).
  const double c2r = (double)((v.LF + v.LB) / 2 - v.LB);
  const double off = std::max(std::fabs((double)v.f2x - c2r), std::fabs((double)v.r2x - c2r));
  const double far2 = (reach + 2.0 * off + 1e-3) * (reach + 2.0 * off + 1e-3);
  constexpr size_t T_BLOCK = 16;
  const int n_blocks = (int)((Nt + T_BLOCK - 1) / T_BLOCK);
  std::vector<std::vector<int32_t>> found((size_t)n_blocks);
  std::atomic<int> illegal{0};
  const int prc = parallel_for(n_blocks, HOST_THREADS_PER_WORLD, [&](const int blk) {
   std::vector<int32_t>& mine = found[(size_t)blk];
   for (size_t t = (size_t)blk * T_BLOCK; t < std::min(Nt, (size_t)(blk + 1) * T_BLOCK); ++t)
    for (int i = 0; i < Na - 1; ++i) {
      const size_t ki = (size_t)i * Nt + t;
      for (int j = i + 1; j < Na; ++j) {
        const size_t kj = (size_t)j * Nt + t;
        {
          const double ddx = (double)xc[ki] - (double)xc[kj], ddy = (double)yc[ki] - (double)yc[kj];
          if (ddx * ddx + ddy * ddy > far2) continue;
        }
        double d2 = sq(xf[ki], xf[kj]) + sq(yf[ki], yf[kj]);
        d2 = std::min(d2, sq(xf[ki], xr[kj]) + sq(yf[ki], yr[kj]));
        d2 = std::min(d2, sq(xr[ki], xf[kj]) + sq(yr[ki], yf[kj]));
        d2 = std::min(d2, sq(xr[ki], xr[kj]) + sq(yr[ki], yr[kj]));
        if (!(std::sqrt(d2) < reach)) continue;
        mine.push_back((int32_t)t);
        mine.push_back(i);
        mine.push_back(j);
        // rectangle SAT in float (State::agentCollision, motion_planning.h:140-183)
        const float sx = xc[kj] - xc[ki], sy = yc[kj] - yc[ki];
        const float cv = cs[ki], sv = sn[ki], co = cs[kj], so = sn[kj];
        const float hl = length / 2, hw = width / 2;
        const float dx1 = cv * length / 2, dy1 = sv * length / 2, dx2 = sv * width / 2, dy2 = -cv * width / 2;
        const float dx3 = co * length / 2, dy3 = so * length / 2, dx4 = so * width / 2, dy4 = -co * width / 2;
        const bool hit = (std::fabs(sx * cv + sy * sv) <= std::fabs(dx3 * cv + dy3 * sv) + std::fabs(dx4 * cv + dy4 * sv) + hl) &&
                         (std::fabs(sx * sv - sy * cv) <= std::fabs(dx3 * sv - dy3 * cv) + std::fabs(dx4 * sv - dy4 * cv) + hw) &&
                         (std::fabs(sx * co + sy * so) <= std::fabs(dx1 * co + dy1 * so) + std::fabs(dx2 * co + dy2 * so) + hl) &&
                         (std::fabs(sx * so - sy * co) <= std::fabs(dx1 * so - dy1 * co) + std::fabs(dx2 * so - dy2 * co) + hw);
        if (hit) illegal.store(1, std::memory_order_relaxed);
      }
    }
  });
  if (prc != CSDO_OK) throw std::bad_alloc();   // a block's list could not grow: the callers (capi.hip) return CSDO_ENOMEM
  size_t total = 0;
  for (const auto& f : found) total += f.size();
  pairs.reserve(total);
  for (const auto& f : found) pairs.insert(pairs.end(), f.begin(), f.end());
  const bool legal = illegal.load() == 0;

  return legal;
}

// calcEqualInterPlanes (:71-140): per-agent CSR of planes in pair order.  `coef` = precomputed coefficients
// [n_pairs][24] (agent i's 12 then agent j's 12, e.g. from the device kernel) or null to compute them here.
int bridge_planes(const BridgeCentres& C, const std::vector<int32_t>& pairs, const csdo_vehicle* vehp, const double* coef,
                  csdo_bridge_out* out) {
  return bridge_planes(C, pairs.data(), pairs.size() / 3, vehp, coef, out);
}

int bridge_planes(const BridgeCentres& C, const int32_t* pairs, size_t n_pairs_in, const csdo_vehicle* vehp, const double* coef,
                  csdo_bridge_out* out) {
  const Veh v{(float)vehp->r, (float)vehp->LF, (float)vehp->LB, (float)vehp->car_width,
              (float)vehp->f2x, (float)vehp->r2x, (float)vehp->rv};
  const int Na = C.Na;
  const size_t Nt = (size_t)C.Nt;
  const std::vector<float>&xf = C.xf, &yf = C.yf, &xr = C.xr, &yr = C.yr;
  const int n_pairs = (int)n_pairs_in;
  // ---- separating planes (calcEqualInterPlanes :71-140, calcPerpendicular :54-69) ----
  std::vector<int32_t> cnt(Na + 1, 0);
  for (int p = 0; p < n_pairs; ++p) {
    cnt[pairs[3 * p + 1] + 1]++;
    cnt[pairs[3 * p + 2] + 1]++;
  }
  for (int a = 0; a < Na; ++a) cnt[a + 1] += cnt[a];
  const int total = cnt[Na];
  out->plane_off = (int32_t*)std::malloc(sizeof(int32_t) * (Na + 1));
  out->planes = (csdo_plane*)std::calloc((size_t)std::max(total, 1), sizeof(csdo_plane));
  out->pairs = (int32_t*)std::malloc(sizeof(int32_t) * 3 * (size_t)std::max(n_pairs, 1));
  if (!out->plane_off || !out->planes || !out->pairs) {
    bridge_free(out);
    return CSDO_ENOMEM;
  }
  std::memcpy(out->plane_off, cnt.data(), sizeof(int32_t) * (Na + 1));
  if (n_pairs) std::memcpy(out->pairs, pairs, sizeof(int32_t) * 3 * (size_t)n_pairs);
  // the slot of every pair in its two agents' lists (pair order), then the coefficients in blocks of pairs side by side
  std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1), slot((size_t)2 * std::max(n_pairs, 1));
  for (int p = 0; p < n_pairs; ++p) {
    slot[2 * (size_t)p] = fill[pairs[3 * p + 1]]++;
    slot[2 * (size_t)p + 1] = fill[pairs[3 * p + 2]]++;
  }
  const double rv = v.rv;
  constexpr int P_BLOCK = 256;
  (void)parallel_for((n_pairs + P_BLOCK - 1) / P_BLOCK, HOST_THREADS_PER_WORLD, [&](const int blk) {
   for (int p = blk * P_BLOCK; p < std::min(n_pairs, (blk + 1) * P_BLOCK); ++p) {
    const int t = pairs[3 * p], i = pairs[3 * p + 1], j = pairs[3 * p + 2];
    const size_t ki = (size_t)i * Nt + t, kj = (size_t)j * Nt + t;
    const double Pi[2][2] = {{xf[ki], yf[ki]}, {xr[ki], yr[ki]}};  // own front, own rear
    const double Pj[2][2] = {{xf[kj], yf[kj]}, {xr[kj], yr[kj]}};
    csdo_plane& pi = out->planes[slot[2 * (size_t)p]];
    csdo_plane& pj = out->planes[slot[2 * (size_t)p + 1]];
    pi.t = pj.t = t;
    if (coef) {
      std::memcpy(pi.c, coef + (size_t)p * 24, sizeof(pi.c));
      std::memcpy(pj.c, coef + (size_t)p * 24 + 12, sizeof(pj.c));
      continue;
    }
    for (int own = 0; own < 2; ++own)
      for (int oth = 0; oth < 2; ++oth) {
        const double x1 = Pi[own][0], y1 = Pi[own][1], x2 = Pj[oth][0], y2 = Pj[oth][1];
        const double a = x2 - x1, b = y2 - y1;
        const double c = (x1 * x1 + y1 * y1 - x2 * x2 - y2 * y2) / 2;
        const double d = std::sqrt(std::pow(x1 - x2, 2) + std::pow(y1 - y2, 2));
        const double c_i = c + rv * d, c_j = c - rv * d;
        const int slot_i = 2 * own + oth;  // f2f, f2r, r2f, r2r
        const int slot_j = 2 * oth + own;  // agent j sees the pair from the other side (:131-135)
        pi.c[3 * slot_i] = a;
        pi.c[3 * slot_i + 1] = b;
        pi.c[3 * slot_i + 2] = c_i;
        pj.c[3 * slot_j] = -a;
        pj.c[3 * slot_j + 1] = -b;
        pj.c[3 * slot_j + 2] = -c_j;
      }
   }
  });
  out->n_pairs = n_pairs;
  return CSDO_OK;
}

int bridge_preprocess(const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                      const double* goals, const csdo_vehicle* vehp, const csdo_qp_parm* parm, csdo_bridge_out* out) {
  const WorldInFlight share_of_the_pool;
  BridgeCentres C;
  int rc = bridge_interpolate(states, actions, path_off, Na, goals, vehp, parm, out, C);
  if (rc != CSDO_OK) return rc;
  std::vector<int32_t> pairs;
  const bool legal = bridge_pairs(C, parm->r_trust, vehp, pairs);
  if ((rc = bridge_planes(C, pairs, vehp, nullptr, out)) != CSDO_OK) return rc;
  out->initial_inter_legal = legal ? 1 : 0;
  return CSDO_OK;
}

void bridge_free(csdo_bridge_out* out) {
  if (!out) return;
  std::free(out->x0_bar);
  std::free(out->plane_off);
  std::free(out->planes);
  std::free(out->pairs);
  std::memset(out, 0, sizeof(*out));
}

}  // namespace csdo
