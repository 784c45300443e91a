// ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes-friendly entry points over the CPU restatement.
// Uses the struct layouts of include/csdo_dsqp.h so tests can hand the same buffers to the oracle and to the HIP
// library.  Nothing in the product links against this file.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/csdo_dsqp.h"
#include "sqp_restate.h"

using namespace csdo_oracle;

static Vehicle to_vehicle(const csdo_vehicle& v) {
  Vehicle o;
  o.r = (float)v.r;
  o.deltat = (float)v.deltat;
  o.LF = (float)v.LF;
  o.LB = (float)v.LB;
  o.carWidth = (float)v.car_width;
  o.WB = (float)v.WB;
  o.f2x = (float)v.f2x;
  o.r2x = (float)v.r2x;
  o.rv = (float)v.rv;
  o.obsRadius = (float)v.obs_radius;
  return o;
}
static QpParm to_parm(const csdo_qp_parm& p) {
  QpParm o;
  o.r_trust = p.r_trust;
  o.max_omega = p.max_omega;
  o.max_v = p.max_v;
  o.max_iter = p.max_iter;
  o.delta_solution_threshold = p.delta_solution_threshold;
  o.max_violation = p.max_violation;
  o.osqp_max_iter = p.osqp_max_iter;
  o.dt = p.dt;
  o.num_interpolation = p.num_interpolation;
  o.fixed_corridor = p.fixed_corridor != 0;
  o.adaptive_rho_interval = p.adaptive_rho_interval > 0 ? p.adaptive_rho_interval : 25;
  return o;
}
static void to_problem(const csdo_problem* in, DsqpProblem& P) {
  P.dimx = in->dimx;
  P.dimy = in->dimy;
  P.veh = to_vehicle(in->veh);
  P.parm = to_parm(in->parm);
  P.obstacles.clear();
  for (int k = 0; k < in->n_obs; ++k)
    P.obstacles.push_back(Obstacle{in->obstacles[3 * k], in->obstacles[3 * k + 1], in->obstacles[3 * k + 2]});
  P.x0_bar.assign(in->Na, std::vector<OptRes>(in->Nt));
  P.planes.assign(in->Na, {});
  for (int a = 0; a < in->Na; ++a) {
    for (int t = 0; t < in->Nt; ++t) {
      const double* g = in->x0_bar + ((size_t)a * in->Nt + t) * 6;
      OptRes& r = P.x0_bar[a][t];
      r.x = g[0];
      r.y = g[1];
      r.yaw = g[2];
      r.steer = g[3];
      r.v = g[4];
      r.d_steer = g[5];
    }
    for (int k = in->plane_off[a]; k < in->plane_off[a + 1]; ++k) {
      InterPlane pl;
      pl.t = in->planes[k].t;
      std::memcpy(pl.c, in->planes[k].c, sizeof(pl.c));
      P.planes[a].push_back(pl);
    }
  }
}

extern "C" {

static void copy_out(const csdo_problem* in, const DsqpResult& R, csdo_result* out) {
  for (int a = 0; a < in->Na; ++a) {
    for (int t = 0; t < in->Nt; ++t) {
      const OptRes& r = R.solutions[a][t];
      double* s = out->solutions + ((size_t)a * in->Nt + t) * 6;
      s[0] = r.x; s[1] = r.y; s[2] = r.yaw; s[3] = r.steer; s[4] = r.v; s[5] = r.d_steer;
      const Corridor& c = R.corridors[a][t];
      double* cc = out->corridors + ((size_t)a * in->Nt + t) * 8;
      cc[0] = c.xf_min; cc[1] = c.xf_max; cc[2] = c.yf_min; cc[3] = c.yf_max;
      cc[4] = c.xr_min; cc[5] = c.xr_max; cc[6] = c.yr_min; cc[7] = c.yr_max;
    }
    out->sqp_iters[a] = R.sqp_iters[a];
    out->admm_iters[a] = R.admm_iters[a];
    out->last_status[a] = R.last_status[a];
  }
  out->solver_status = R.solver_status;
  out->initial_static_legal = R.initial_static_legal ? 1 : 0;
  out->t_total = R.t_total;
  out->t_device = 0.0;
  out->t_max_individual = R.t_max_individual;
  if (out->agent_seconds)
    for (size_t a = 0; a < R.agent_seconds.size(); ++a) out->agent_seconds[a] = R.agent_seconds[a];
}

int csdo_oracle_solve(const csdo_problem* in, csdo_result* out, int n_threads) {
  if (!in || !out || in->Nt < 2 || in->Na < 1) return CSDO_EINVAL;
  DsqpProblem P;
  to_problem(in, P);
  DsqpResult R;
  dsqp_solve(P, R, n_threads, nullptr);
  copy_out(in, R, out);
  return CSDO_OK;
}

// Several worlds, one thread pool over all their agents (the all-core CPU baseline of a batch).
int csdo_oracle_solve_batch(const csdo_problem* worlds, int32_t n_worlds, csdo_result* results, int n_threads) {
  if (!worlds || !results || n_worlds < 1) return CSDO_EINVAL;
  std::vector<DsqpProblem> P(n_worlds);
  for (int w = 0; w < n_worlds; ++w) {
    if (worlds[w].Nt < 2 || worlds[w].Na < 1) return CSDO_EINVAL;
    to_problem(&worlds[w], P[w]);
  }
  std::vector<DsqpResult> R;
  dsqp_solve_batch(P, R, n_threads);
  for (int w = 0; w < n_worlds; ++w) copy_out(&worlds[w], R[w], &results[w]);
  return CSDO_OK;
}

// Per-SQP-iteration trace for golden fixtures: returns number of entries written (<= cap); each entry is
// {agent, sqp_iter, status, admm_iter} in meta[4*e..] and delta in deltas[e]; sols[e*(6Nt-2)..].
int csdo_oracle_trace(const csdo_problem* in, int cap, int32_t* meta, double* deltas, double* sols) {
  DsqpProblem P;
  to_problem(in, P);
  DsqpResult R;
  std::vector<SqpTraceEntry> tr;
  dsqp_solve(P, R, 1, &tr);
  const int n = 6 * in->Nt - 2;
  int e = 0;
  for (const auto& t : tr) {
    if (e >= cap) break;
    meta[4 * e] = t.agent;
    meta[4 * e + 1] = t.sqp_iter;
    meta[4 * e + 2] = t.status;
    meta[4 * e + 3] = t.admm_iter;
    deltas[e] = t.delta;
    std::memcpy(sols + (size_t)e * n, t.sol.data(), sizeof(double) * n);
    ++e;
  }
  return e;
}

int csdo_oracle_preprocess(const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                           const double* goals, const csdo_vehicle* veh, const csdo_qp_parm* parm,
                           csdo_bridge_out* out) {
  const Vehicle V = to_vehicle(*veh);
  const QpParm Q = to_parm(*parm);
  std::vector<CoarsePath> paths(Na);
  std::vector<std::array<double, 3>> G(Na);
  int act_off = 0;
  for (int a = 0; a < Na; ++a) {
    const int L = path_off[a + 1] - path_off[a];
    for (int i = 0; i < L; ++i) {
      const double* s = states + (size_t)(path_off[a] + i) * 3;
      paths[a].states.push_back({s[0], s[1], s[2]});
    }
    for (int i = 0; i + 1 < L; ++i) paths[a].actions.push_back(actions[act_off + i]);
    act_off += L - 1;
    G[a] = {goals[3 * a], goals[3 * a + 1], goals[3 * a + 2]};
  }
  std::vector<std::vector<OptRes>> x0;
  interpolate_initial_guess(paths, G, Q, V, x0);
  std::vector<std::array<int, 3>> pairs;
  const bool legal = find_neighbor_pairs(x0, Q.r_trust, V, pairs);
  std::vector<std::vector<InterPlane>> planes;
  calc_inter_planes(x0, pairs, V, planes);
  const int Nt = (int)x0[0].size();
  out->Na = Na;
  out->Nt = Nt;
  out->x0_bar = (double*)std::malloc(sizeof(double) * (size_t)Na * Nt * 6);
  for (int a = 0; a < Na; ++a)
    for (int t = 0; t < Nt; ++t) {
      double* g = out->x0_bar + ((size_t)a * Nt + t) * 6;
      const OptRes& r = x0[a][t];
      g[0] = r.x; g[1] = r.y; g[2] = r.yaw; g[3] = r.steer; g[4] = r.v; g[5] = r.d_steer;
    }
  out->plane_off = (int32_t*)std::malloc(sizeof(int32_t) * (Na + 1));
  int tot = 0;
  for (int a = 0; a < Na; ++a) {
    out->plane_off[a] = tot;
    tot += (int)planes[a].size();
  }
  out->plane_off[Na] = tot;
  out->planes = (csdo_plane*)std::malloc(sizeof(csdo_plane) * (size_t)(tot > 0 ? tot : 1));
  for (int a = 0; a < Na; ++a)
    for (size_t k = 0; k < planes[a].size(); ++k) {
      csdo_plane& p = out->planes[out->plane_off[a] + k];
      p.t = planes[a][k].t;
      p._pad = 0;
      std::memcpy(p.c, planes[a][k].c, sizeof(p.c));
    }
  out->n_pairs = (int)pairs.size();
  out->initial_inter_legal = legal ? 1 : 0;
  out->pairs = (int32_t*)std::malloc(sizeof(int32_t) * 3 * (pairs.empty() ? 1 : pairs.size()));
  for (size_t k = 0; k < pairs.size(); ++k) {
    out->pairs[3 * k] = pairs[k][0];
    out->pairs[3 * k + 1] = pairs[k][1];
    out->pairs[3 * k + 2] = pairs[k][2];
  }
  return CSDO_OK;
}

void csdo_oracle_bridge_free(csdo_bridge_out* out) {
  std::free(out->x0_bar);
  std::free(out->plane_off);
  std::free(out->planes);
  std::free(out->pairs);
  std::memset(out, 0, sizeof(*out));
}

int csdo_oracle_generate_boxes(const double* pts, int32_t n, const double* obstacles, int32_t n_obs, double dimx,
                               double dimy, const csdo_vehicle* veh, double* boxes, int32_t* status) {
  const Vehicle V = to_vehicle(*veh);
  std::vector<Obstacle> obs;
  for (int k = 0; k < n_obs; ++k) obs.push_back(Obstacle{obstacles[3 * k], obstacles[3 * k + 1], obstacles[3 * k + 2]});
  for (int i = 0; i < n; ++i) {
    Box b{0, 0, 0, 0};
    const BoxStatus st = generate_box(dimx, dimy, pts[2 * i], pts[2 * i + 1], obs, V, b);
    boxes[4 * i] = b.x_min;
    boxes[4 * i + 1] = b.y_min;
    boxes[4 * i + 2] = b.x_max;
    boxes[4 * i + 3] = b.y_max;
    status[i] = (st.success ? 1 : 0) | (st.initial_status << 1);
  }
  return CSDO_OK;
}

// Assemble one agent QP.  Two-call pattern: with A_x == NULL only sizes are returned.
int csdo_oracle_assemble_qp(int32_t Nt, const double* sol0, const double* corr_lb, const double* corr_ub,
                            const double* x_trust, const double* y_trust, const double* cfg,
                            const csdo_plane* planes, int32_t n_planes, const csdo_vehicle* veh,
                            const csdo_qp_parm* parm, int32_t* m_out, int32_t* nnzA_out, int32_t* nnzP_out,
                            int32_t* A_p, int32_t* A_i, double* A_x, int32_t* P_p, int32_t* P_i, double* P_x,
                            double* l, double* u) {
  const int n = 6 * Nt - 2;
  std::vector<InterPlane> pl(n_planes);
  for (int k = 0; k < n_planes; ++k) {
    pl[k].t = planes[k].t;
    std::memcpy(pl[k].c, planes[k].c, sizeof(pl[k].c));
  }
  AgentQp qp;
  assemble_qp(Nt, std::vector<double>(sol0, sol0 + n), std::vector<double>(corr_lb, corr_lb + 4 * Nt),
              std::vector<double>(corr_ub, corr_ub + 4 * Nt), std::vector<double>(x_trust, x_trust + Nt),
              std::vector<double>(y_trust, y_trust + Nt), cfg, pl, to_parm(*parm), to_vehicle(*veh), qp);
  *m_out = qp.A.m;
  *nnzA_out = qp.A.nnz();
  *nnzP_out = qp.P_triu.nnz();
  if (!A_x) return CSDO_OK;
  std::copy(qp.A.p.begin(), qp.A.p.end(), A_p);
  std::copy(qp.A.i.begin(), qp.A.i.end(), A_i);
  std::copy(qp.A.x.begin(), qp.A.x.end(), A_x);
  std::copy(qp.P_triu.p.begin(), qp.P_triu.p.end(), P_p);
  std::copy(qp.P_triu.i.begin(), qp.P_triu.i.end(), P_i);
  std::copy(qp.P_triu.x.begin(), qp.P_triu.x.end(), P_x);
  std::copy(qp.l.begin(), qp.l.end(), l);
  std::copy(qp.u.begin(), qp.u.end(), u);
  return CSDO_OK;
}

// Restated OSQP on an arbitrary CSC QP (q given).  info[0..3] = status, iter, rho_updates, factor nnz.
int csdo_oracle_osqp(int32_t n, int32_t m, const int32_t* P_p, const int32_t* P_i, const double* P_x,
                     const double* q, const int32_t* A_p, const int32_t* A_i, const double* A_x, const double* l,
                     const double* u, const double* x_warm, int32_t max_iter, int32_t adaptive_rho_interval,
                     double eps_abs, double eps_rel, double* x_out, double* y_out, int32_t* info) {
  Csc P, A;
  P.m = P.n = n;
  P.p.assign(P_p, P_p + n + 1);
  P.i.assign(P_i, P_i + P.p[n]);
  P.x.assign(P_x, P_x + P.p[n]);
  A.m = m;
  A.n = n;
  A.p.assign(A_p, A_p + n + 1);
  A.i.assign(A_i, A_i + A.p[n]);
  A.x.assign(A_x, A_x + A.p[n]);
  Settings st;
  st.max_iter = max_iter;
  st.adaptive_rho_interval = adaptive_rho_interval;
  st.eps_abs = eps_abs;
  st.eps_rel = eps_rel;
  std::vector<double> xs, ys;
  const Info inf = osqp_solve_restated(P, std::vector<double>(q, q + n), A, std::vector<double>(l, l + m),
                                       std::vector<double>(u, u + m), std::vector<double>(x_warm, x_warm + n), st,
                                       xs, &ys);
  std::copy(xs.begin(), xs.end(), x_out);
  if (y_out) std::copy(ys.begin(), ys.end(), y_out);
  info[0] = inf.status;
  info[1] = inf.iter;
  info[2] = inf.rho_updates;
  info[3] = (int32_t)last_factor_nnz();
  return CSDO_OK;
}

// The same solve with the history the termination checks saw: hist[3 * k .. 3 * k + 2] = (rho, pri_res, dua_res) at check k
// (every check_termination iterations); n_hist receives the number of checks (at most cap are written).  For the independent
// cross-check of the iterate PATH (tests/test_oracle_admm_path.py): iteration count, status and rho history.
int csdo_oracle_osqp_hist(int32_t n, int32_t m, const int32_t* P_p, const int32_t* P_i, const double* P_x,
                          const double* q, const int32_t* A_p, const int32_t* A_i, const double* A_x, const double* l,
                          const double* u, const double* x_warm, int32_t max_iter, int32_t adaptive_rho_interval,
                          double eps_abs, double eps_rel, double* x_out, double* y_out, int32_t* info, int32_t cap,
                          double* hist, int32_t* n_hist) {
  Csc P, A;
  P.m = P.n = n;
  P.p.assign(P_p, P_p + n + 1);
  P.i.assign(P_i, P_i + P.p[n]);
  P.x.assign(P_x, P_x + P.p[n]);
  A.m = m;
  A.n = n;
  A.p.assign(A_p, A_p + n + 1);
  A.i.assign(A_i, A_i + A.p[n]);
  A.x.assign(A_x, A_x + A.p[n]);
  Settings st;
  st.max_iter = max_iter;
  st.adaptive_rho_interval = adaptive_rho_interval;
  st.eps_abs = eps_abs;
  st.eps_rel = eps_rel;
  std::vector<double> xs, ys;
  Trace tr;
  const Info inf = osqp_solve_restated(P, std::vector<double>(q, q + n), A, std::vector<double>(l, l + m),
                                       std::vector<double>(u, u + m), std::vector<double>(x_warm, x_warm + n), st,
                                       xs, &ys, &tr);
  std::copy(xs.begin(), xs.end(), x_out);
  if (y_out) std::copy(ys.begin(), ys.end(), y_out);
  info[0] = inf.status;
  info[1] = inf.iter;
  info[2] = inf.rho_updates;
  info[3] = (int32_t)last_factor_nnz();
  const int k = (int)tr.rho_hist.size();
  *n_hist = k;
  for (int e = 0; e < k && e < cap; ++e) {
    hist[3 * e] = tr.rho_hist[e];
    hist[3 * e + 1] = tr.pri_res_hist[e];
    hist[3 * e + 2] = tr.dua_res_hist[e];
  }
  return CSDO_OK;
}

}  // extern "C"
