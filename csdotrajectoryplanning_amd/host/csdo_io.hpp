// csdo_io.hpp — the wire formats either side of the DO path for a C++ caller that has no yaml-cpp.
// Header only, C++14, no dependency beyond include/csdo_dsqp.h.
//
// What it replaces in the reference (behaviour restated, nothing copied):
//   csdo::io::load_config     readAgentConfig + readQpSolverConfig      common/motion_planning.cc:54-93, sqp/utils.cc:34-59
//                             (YAML doubles through `float` statics, derived f2x / r2x / rv, dt from the float product r*deltat)
//   csdo::io::load_instance   Instance::loadMap                         hybrid_a_star/Instance.cc:25-63
//                             (dimensions as int, 2-element obstacles take obsRadius, `obstacles:` may be empty / null)
//   csdo::io::dump_solutions  dumpSolutions                             sqp/inter_agent_cons.cc:413-455 (header order is positional
//                             for scripts/analysis_result.py:53-101; %.3f; steer / omega with the reference's 180/3.14)
//   csdo::io::dump_corridors  dumpCorridors                             sqp/utils.cc:62-89 (default ostream formatting)
//   csdo::io::output_paths    csdo.cc:76,139,164                        <out>.yaml, <out>_guesses.yaml, <out>_corridors.yaml
//
// The readers take the YAML SUBSET these two file families use (what the reference's own benchmark/ and config.yaml hold):
// block mappings and block sequences by indentation, flow sequences `[a, b, c]` of scalars or of flow sequences, `#`
// comments, plain scalars.  Anything else is reported as a parse error (return false + message), never guessed at.
// tests/cpp/io_main.cc + tests/test_cpp_io.py hold this header to the Python loaders / writers value for value and byte for byte.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/csdo_dsqp.h"

namespace csdo {
namespace io {

struct Instance {
  double dimx = 0, dimy = 0;
  std::vector<double> obstacles;   // [n_obs][3] x, y, r in file order
  std::vector<double> starts;      // [Na][3]
  std::vector<double> goals;       // [Na][3]
  int num_agents() const { return (int)(starts.size() / 3); }
  int num_obstacles() const { return (int)(obstacles.size() / 3); }
};

struct SolutionStatistics {        // sqp/common.h:25-36 defaults: the floats -1, search_status 2, solver_status 0
  double cost = -1, makespan = -1, flowtime = -1, runtime = -1, rt_search = -1, rt_preprocess = -1, rt_optimization = -1,
         rt_max_optimization = -1;
  int search_status = 2, solver_status = 0;
};

// The reference's statistics for a DO-phase result, its semantics (csdo.cc:98-161, sqp/dsqp_solver.cc:1194-1248) on this
// backend's clocks: rt_optimization = wall clock of the call; rt_max_optimization ("ideal parallel") = slowest agent's device time
// (its corridors included) + everything outside the kernels; runtime = search + preprocess + that; search_status 1 when the initial
// guess was not statically legal (csdo.cc:152-154).
inline SolutionStatistics statistics_from(const csdo_result& r, double rt_search, double rt_preprocess, int search_status = 2) {
  SolutionStatistics s;
  const double other = r.t_total > r.t_device ? r.t_total - r.t_device : 0.0;
  s.rt_search = rt_search;
  s.rt_preprocess = rt_preprocess;
  s.rt_optimization = r.t_total;
  s.rt_max_optimization = r.t_max_individual + other;
  s.runtime = (rt_search >= 0 && rt_preprocess >= 0) ? rt_search + rt_preprocess + s.rt_max_optimization : -1.0;
  s.search_status = r.initial_static_legal ? search_status : 1;
  s.solver_status = r.solver_status;
  return s;
}

namespace detail {
inline std::string strip_comment(const std::string& s) {   // `#` at the start of the line or after white space
  for (size_t i = 0; i < s.size(); ++i)
    if (s[i] == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
  return s;
}
inline std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
  return s.substr(a, b - a);
}
inline bool to_double(const std::string& s, double& v) {
  const std::string t = trim(s);
  if (t.empty()) return false;
  char* end = nullptr;
  v = std::strtod(t.c_str(), &end);
  return end && *end == '\0';
}
// `[a, b, c]` -> numbers; nested lists are not accepted here
inline bool flow_numbers(const std::string& s, std::vector<double>& out) {
  const std::string t = trim(s);
  if (t.size() < 2 || t.front() != '[' || t.back() != ']') return false;
  std::stringstream ss(t.substr(1, t.size() - 2));
  std::string item;
  out.clear();
  while (std::getline(ss, item, ',')) {
    double v;
    if (!to_double(item, v)) return false;
    out.push_back(v);
  }
  return true;
}
struct Line {
  int indent;
  std::string text;   // without indentation and comment
};
inline bool read_lines(const std::string& path, std::vector<Line>& lines, std::string* err) {
  std::ifstream f(path);
  if (!f) {
    if (err) *err = "cannot open " + path;
    return false;
  }
  std::string raw;
  while (std::getline(f, raw)) {
    const std::string body = strip_comment(raw);
    if (trim(body).empty()) continue;
    int ind = 0;
    while ((size_t)ind < body.size() && body[ind] == ' ') ++ind;
    lines.push_back(Line{ind, trim(body)});
  }
  return true;
}
}  // namespace detail

// ---- config.yaml: flat `key: value` lines -----------------------------------------------------------------------------
inline bool load_config_map(const std::string& path, std::map<std::string, std::string>& kv, std::string* err = nullptr) {
  std::vector<detail::Line> lines;
  if (!detail::read_lines(path, lines, err)) return false;
  for (const auto& l : lines) {
    const size_t c = l.text.find(':');
    if (c == std::string::npos) {
      if (err) *err = "not a `key: value` line: " + l.text;
      return false;
    }
    kv[detail::trim(l.text.substr(0, c))] = detail::trim(l.text.substr(c + 1));
  }
  return true;
}

// The shipped config.yaml's values for the keys a file leaves out (config.yaml:4-57 of the reference).
inline bool load_config(const std::string& path, csdo_vehicle* veh, csdo_qp_parm* parm, csdo_front_end_parm* front = nullptr,
                        std::string* err = nullptr) {
  std::map<std::string, std::string> kv;
  if (!path.empty() && !load_config_map(path, kv, err)) return false;
  bool ok = true;
  auto num = [&](const char* key, double dflt) {
    auto it = kv.find(key);
    if (it == kv.end()) return dflt;
    double v;
    if (it->second == "true") return 1.0;
    if (it->second == "false") return 0.0;
    if (!detail::to_double(it->second, v)) {
      ok = false;
      if (err) *err = std::string("not a number: ") + key + ": " + it->second;
      return dflt;
    }
    return v;
  };
  // readAgentConfig: YAML doubles stored into float statics (motion_planning.cc:64-85)
  const float r = (float)num("r", 3), deltat = (float)num("deltat", 0.706), W = (float)num("carWidth", 2.0),
              LF = (float)num("LF", 2.0), LB = (float)num("LB", 1.0), WB = (float)num("WB", 1.0),
              obsR = (float)num("obsRadius", 0.8);
  const float f2x = (float)(1 / 4.0 * (3.0 * LF - LB));
  const float r2x = (float)(1 / 4.0 * (LF - 3.0 * LB));
  const float rv = (float)(1.0 / 2.0 * std::pow(std::pow(LF + LB, 2) / 4 + W * W, 0.5));
  if (veh) {
    veh->r = r;
    veh->deltat = deltat;
    veh->LF = LF;
    veh->LB = LB;
    veh->car_width = W;
    veh->WB = WB;
    veh->f2x = f2x;
    veh->r2x = r2x;
    veh->rv = rv;
    veh->obs_radius = obsR;
  }
  if (parm) {   // readQpSolverConfig, sqp/utils.cc:44-58
    parm->r_trust = num("r_trust", 2.0);
    parm->max_omega = num("max_omega", 0.07);
    parm->max_v = num("max_v", 1);
    parm->max_iter = num("max_iter", 10);
    parm->delta_solution_threshold = num("delta_solution_threshold", 1);
    parm->max_violation = num("max_violation", 0.001);
    parm->osqp_max_iter = (int32_t)num("osqp_max_iter", 400);
    parm->num_interpolation = (int32_t)num("num_interpolation", 2);
    const double step = (double)(float)(r * deltat);       // Constants::r * Constants::deltat is a float product
    parm->dt = step / parm->max_v / (parm->num_interpolation + 1) / num("decelerate_factor", 0.8);
    parm->fixed_corridor = num("fixed_corridor", 0) != 0 ? 1 : 0;
    parm->adaptive_rho_interval = 25;
    parm->solve_refinement = 0;
    parm->_reserved = 0;
  }
  if (front) {
    csdo_front_end_parm_default(front);
    front->penalty_turning = num("penaltyTurning", 1.5);
    front->penalty_reversing = num("penaltyReversing", 2.0);
    front->penalty_cod = num("penaltyCOD", 2.0);
    front->map_resolution = num("mapResolution", 2.0);
    front->max_closed_set_size = num("maxClosedSetSize", 1e5);
  }
  return ok;
}

// ---- instance files ---------------------------------------------------------------------------------------------------
inline bool load_instance(const std::string& path, double obs_radius, Instance& out, std::string* err = nullptr) {
  std::vector<detail::Line> lines;
  if (!detail::read_lines(path, lines, err)) return false;
  out = Instance();
  auto fail = [&](const std::string& m) {
    if (err) *err = path + ": " + m;
    return false;
  };
  enum { NONE, AGENTS, MAP } section = NONE;
  bool in_obstacles = false;
  std::vector<double> pending;          // a block-style obstacle `- - x` / `  - y` (/ `  - r`) being collected
  bool have_start = false, have_goal = false;
  auto flush_pending = [&]() {
    if (pending.empty()) return true;
    if (pending.size() != 2 && pending.size() != 3) return false;
    out.obstacles.insert(out.obstacles.end(), {pending[0], pending[1], pending.size() == 3 ? pending[2] : obs_radius});
    pending.clear();
    return true;
  };
  for (size_t i = 0; i < lines.size(); ++i) {
    const auto& l = lines[i];
    std::string t = l.text;
    if (l.indent == 0 && !(section == AGENTS && t.compare(0, 2, "- ") == 0)) {   // (a list may sit at its key's indentation)
      if (!flush_pending()) return fail("obstacle with " + std::to_string(pending.size()) + " elements");
      in_obstacles = false;
      if (t == "agents:") section = AGENTS;
      else if (t == "map:") section = MAP;
      else return fail("unexpected top-level key: " + t);
      continue;
    }
    if (section == AGENTS) {
      if (t.compare(0, 2, "- ") == 0) {   // a new agent
        if ((have_start != have_goal)) return fail("agent without start or goal");
        have_start = have_goal = false;
        t = detail::trim(t.substr(2));
      }
      const size_t c = t.find(':');
      if (c == std::string::npos) return fail("not a `key: value` line: " + t);
      const std::string key = detail::trim(t.substr(0, c)), val = detail::trim(t.substr(c + 1));
      if (key == "start" || key == "goal") {
        std::vector<double> v;
        if (!detail::flow_numbers(val, v) || v.size() != 3) return fail(key + " is not [x, y, yaw]: " + val);
        auto& dst = key == "start" ? out.starts : out.goals;
        dst.insert(dst.end(), v.begin(), v.end());
        (key == "start" ? have_start : have_goal) = true;
      }   // (`name` and anything else: ignored, as Instance::loadMap does)
      continue;
    }
    if (section == MAP) {
      if (in_obstacles && t.compare(0, 1, "-") == 0) {
        std::string item = detail::trim(t.substr(1));
        if (!item.empty() && item[0] == '[') {      // `- [x, y]` or `- [x, y, r]`
          if (!flush_pending()) return fail("obstacle with " + std::to_string(pending.size()) + " elements");
          std::vector<double> v;
          if (!detail::flow_numbers(item, v) || (v.size() != 2 && v.size() != 3)) return fail("obstacle is not [x, y(, r)]: " + item);
          out.obstacles.insert(out.obstacles.end(), {v[0], v[1], v.size() == 3 ? v[2] : obs_radius});
        } else {                                     // block style: `- - x` opens an obstacle, `- y` continues it
          if (item.compare(0, 1, "-") == 0) {
            if (!flush_pending()) return fail("obstacle with " + std::to_string(pending.size()) + " elements");
            item = detail::trim(item.substr(1));
          }
          double v;
          if (!detail::to_double(item, v)) return fail("obstacle element is not a number: " + item);
          pending.push_back(v);
        }
        continue;
      }
      if (!flush_pending()) return fail("obstacle with " + std::to_string(pending.size()) + " elements");
      in_obstacles = false;
      const size_t c = t.find(':');
      if (c == std::string::npos) return fail("not a `key: value` line: " + t);
      const std::string key = detail::trim(t.substr(0, c)), val = detail::trim(t.substr(c + 1));
      if (key == "dimensions") {
        std::vector<double> v;
        if (!detail::flow_numbers(val, v) || v.size() != 2) return fail("dimensions is not [X, Y]: " + val);
        out.dimx = (double)(int)v[0];               // dim[0].as<int>()
        out.dimy = (double)(int)v[1];
      } else if (key == "obstacles") {
        if (val.empty() || val == "null" || val == "~" || val == "[]") in_obstacles = val.empty();
        else return fail("obstacles: expected a block list, null or nothing: " + val);
      }
      continue;
    }
    return fail("line outside a section: " + t);
  }
  if (!flush_pending()) return fail("obstacle with " + std::to_string(pending.size()) + " elements");
  if (out.starts.size() != out.goals.size() || out.starts.empty()) return fail("agents without matching start / goal");
  if (!(out.dimx > 0 && out.dimy > 0)) return fail("map dimensions missing");
  return true;
}

// ---- writers ----------------------------------------------------------------------------------------------------------
inline bool output_paths(const std::string& output_file, std::string& guesses, std::string& corridors) {
  if (output_file.size() < 5 || output_file.compare(output_file.size() - 5, 5, ".yaml") != 0) return false;
  const std::string prefix = output_file.substr(0, output_file.size() - 5);   // csdo.cc:76
  guesses = prefix + "_guesses.yaml";
  corridors = prefix + "_corridors.yaml";
  return true;
}

// solutions [Na][Nt][6] = x, y, yaw, steer, v, d_steer (csdo_result.solutions, or x0_bar for the --initial_guess dump)
inline bool dump_solutions(const std::string& path, const double* solutions, int Na, int Nt, const SolutionStatistics& st) {
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) return false;
  std::fprintf(f, "statistics:\n  cost: %.3f\n  makespan: %.3f\n  flowtime: %.3f\n  runtime: %.3f\n  runtime_search: %.3f\n"
                  "  runtime_preprocess: %.3f\n  runtime_optimization: %.3f\n  runtime_decentralized_optimization: %.3f\n"
                  "  search_status: %d\n  solver_status: %d\nschedule:\n",
               st.cost, st.makespan, st.flowtime, st.runtime, st.rt_search, st.rt_preprocess, st.rt_optimization,
               st.rt_max_optimization, st.search_status, st.solver_status);
  const double deg = 180 / 3.14;   // the reference's constant, not pi
  for (int a = 0; a < Na; ++a) {
    std::fprintf(f, "  agent%d:\n", a);
    for (int t = 0; t < Nt; ++t) {
      const double* s = solutions + ((size_t)a * Nt + t) * 6;
      std::fprintf(f, "    - x: %.3f\n      y: %.3f\n      yaw: %.3f\n      steer: %.3f\n      t: %d\n", s[0], s[1], s[2], s[3] * deg, t);
      if (t == Nt - 1) continue;
      std::fprintf(f, "      v: %.3f\n      omega: %.3f\n", s[4], s[5] * deg);
    }
  }
  return std::fclose(f) == 0;
}

// corridors [Na][Nt][8] (csdo_result.corridors), x0_bar [Na][Nt][6]; the disc centres are State's float members
inline bool dump_corridors(const std::string& path, const double* corridors, const double* x0_bar, int Na, int Nt,
                           const csdo_vehicle& veh) {
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) return false;
  for (int a = 0; a < Na; ++a) {
    std::fprintf(f, "agent%d:\n", a);
    for (int t = 0; t < Nt; ++t) {
      const double* g = x0_bar + ((size_t)a * Nt + t) * 6;
      const double* c = corridors + ((size_t)a * Nt + t) * 8;
      const double xf = (double)(float)(g[0] + veh.f2x * std::cos(g[2])), yf = (double)(float)(g[1] + veh.f2x * std::sin(g[2]));
      const double xr = (double)(float)(g[0] + veh.r2x * std::cos(g[2])), yr = (double)(float)(g[1] + veh.r2x * std::sin(g[2]));
      std::fprintf(f, "  - [%g, %g, %g, %g, %g, %g]\n", xf, yf, c[0], c[1], c[2], c[3]);
      std::fprintf(f, "  - [%g, %g, %g, %g, %g, %g]\n", xr, yr, c[4], c[5], c[6], c[7]);
    }
  }
  return std::fclose(f) == 0;
}

}  // namespace io
}  // namespace csdo
