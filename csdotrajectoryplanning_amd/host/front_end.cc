// front_end.cc — host front end of CSDO: priority-based search over spatiotemporal hybrid A* (SURVEY 8f rank 2).
//
// What it reproduces (behaviour, not code; the reference's version needs Boost heaps and OMPL, absent here):
//   pbs/PBS.cc:28-66,68-214,665-719   PBS high level: root = agents planned one after another, each avoiding the earlier
//                                     ones; depth-first expansion; a node resolves its LAST conflict by trying both
//                                     priority orders and re-planning, in topological order, every agent that now collides
//                                     with a higher one
//   hybrid_a_star/hybrid_astar.h:91-207, environment.h:128-392,455-521
//                                     low level: A* over (x, y, yaw, t) with six arc primitives + wait, turning /
//                                     reversing / change-of-direction penalties, closed list on the (t, yaw, y, x) grid,
//                                     heuristic max(Reeds-Shepp, Euclid, 2-D obstacle-aware map) truncated to int as the
//                                     reference's `int admissibleHeuristic` does, and the Reeds-Shepp "shot" towards the goal
//                                     tried on every pop with the reference's random range test (rand() % 10 + 1 after
//                                     srand(0), csdo.cc:93; here a generator owned by the call, so that the entry is re-entrant)
//   common/motion_planning.h:140-199  rectangle SAT between vehicles (float), inflated-obstacle test in the vehicle frame
// Deliberate differences from the reference's rule set: (1) a Reeds-Shepp shot whose parked end pose a higher agent would
// hit later is refused; (2) csdo_front_end_parm::keep_off_lower_goals (default on): poses at t >= 1 that overlap the goal
// rectangle of an agent NOT ranked above the planning one are invalid - in the root node agent 0 therefore treats every other
// goal as an obstacle - and (3) with that rule an agent whose start is boxed in by such goals is planned alone in the root
// (make_root) and left to the priorities.  The reference has none of the three (environment.h:350-392, PBS.cc:665-719); with
// (2) off, (3) never triggers.
// Exact path equality with the reference is not attainable (heap tie-breaking, OMPL internals, unordered_set order) and
// not claimed; the tests check what PBS itself validates: every step is a motion primitive, no two rectangles overlap at
// equal times, no obstacle is touched, every path ends at its goal.
#include "front_end.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <list>
#include <new>
#include <map>
#include <queue>
#include <set>
#include <unordered_map>
#include <unordered_set>

#include "reeds_shepp.h"

namespace csdo {
namespace {

using clk = std::chrono::steady_clock;

struct Consts {   // Constants:: of the reference, float members as there (common/motion_planning.cc:54-109)
  bool keepOffLowerGoals;   // csdo_front_end_parm::keep_off_lower_goals (not in the reference)
  float r, deltat, penaltyTurning, penaltyReversing, penaltyCOD, mapResolution, xyResolution, yawResolution, maxClosed;
  float carWidth, LF, LB, f2x, r2x, rv;
  double dx[6], dy[6], dyaw[6];
};

Consts make_consts(const csdo_vehicle& v, const csdo_front_end_parm& p) {
  Consts c{};
  c.r = (float)v.r;
  c.deltat = (float)v.deltat;
  c.penaltyTurning = (float)p.penalty_turning;
  c.penaltyReversing = (float)p.penalty_reversing;
  c.penaltyCOD = (float)p.penalty_cod;
  c.mapResolution = (float)p.map_resolution;
  c.xyResolution = c.r * c.deltat;
  c.yawResolution = c.deltat;
  c.maxClosed = (float)p.max_closed_set_size;
  c.keepOffLowerGoals = p.keep_off_lower_goals != 0;
  c.carWidth = (float)v.car_width;
  c.LF = (float)v.LF;
  c.LB = (float)v.LB;
  c.f2x = (float)v.f2x;
  c.r2x = (float)v.r2x;
  c.rv = (float)v.rv;
  const double arc = c.r * std::sin(c.deltat), lat = c.r * (1 - std::cos(c.deltat)), st = c.r * c.deltat;
  const double dx[6] = {st, arc, arc, -st, -arc, -arc}, dy[6] = {0, -lat, lat, 0, -lat, lat};
  const double dyaw[6] = {0, -(double)c.deltat, c.deltat, 0, c.deltat, -(double)c.deltat};
  std::memcpy(c.dx, dx, sizeof dx);
  std::memcpy(c.dy, dy, sizeof dy);
  std::memcpy(c.dyaw, dyaw, sizeof dyaw);
  return c;
}

inline float heading_0_2pi(float t) {   // Constants::normalizeHeadingRad: float in, float out
  if (t < 0) {
    t = t - 2.f * (float)M_PI * (int)(t / (2.f * (float)M_PI));
    return 2.f * (float)M_PI + t;
  }
  return t - 2.f * (float)M_PI * (int)(t / (2.f * (float)M_PI));
}

struct Pose {   // State of the reference: pose at a timestep plus the float centres its constructor derives
  double x = 0, y = 0, yaw = 0;
  int t = 0;
  float xf = 0, yf = 0, xr = 0, yr = 0, xc = 0, yc = 0;
  Pose() = default;
  Pose(double x_, double y_, double yaw_, int t_, const Consts& c) : x(x_), y(y_), yaw(yaw_), t(t_) {
    const double cs = std::cos(yaw), sn = std::sin(yaw);
    xf = (float)(x + c.f2x * cs);
    xr = (float)(x + c.r2x * cs);
    yf = (float)(y + c.f2x * sn);
    yr = (float)(y + c.r2x * sn);
    const float c2r = (c.LF + c.LB) / 2 - c.LB;
    xc = (float)(x + c2r * cs);
    yc = (float)(y + c2r * sn);
  }
};

bool rect_overlap(const Pose& a, const Pose& b, const Consts& c) {   // State::agentCollision, float SAT
  const float length = c.LF + c.LB, width = c.carWidth;
  const float sx = b.xc - a.xc, sy = b.yc - a.yc;
  const float cv = (float)std::cos(a.yaw), sv = (float)std::sin(a.yaw), co = (float)std::cos(b.yaw), so = (float)std::sin(b.yaw);
  const float hl = length / 2, hw = width / 2;
  const float dx1 = cv * length / 2, dy1 = sv * length / 2, dx2 = sv * width / 2, dy2 = -cv * width / 2;
  const float dx3 = co * length / 2, dy3 = so * length / 2, dx4 = so * width / 2, dy4 = -co * width / 2;
  return (std::fabs(sx * cv + sy * sv) <= std::fabs(dx3 * cv + dy3 * sv) + std::fabs(dx4 * cv + dy4 * sv) + hl) &&
         (std::fabs(sx * sv - sy * cv) <= std::fabs(dx3 * sv - dy3 * cv) + std::fabs(dx4 * sv - dy4 * cv) + hw) &&
         (std::fabs(sx * co + sy * so) <= std::fabs(dx1 * co + dy1 * so) + std::fabs(dx2 * co + dy2 * so) + hl) &&
         (std::fabs(sx * so - sy * co) <= std::fabs(dx1 * so - dy1 * co) + std::fabs(dx2 * so - dy2 * co) + hw);
}

struct PlannedPath {
  std::vector<Pose> states;
  std::vector<int> actions;   // states.size() - 1
  bool empty() const { return states.empty(); }
  size_t size() const { return states.size(); }
};

// ---------------------------------------------------------------------------------------------------------------------
// low level: one agent, given the paths of the agents it must yield to
// ---------------------------------------------------------------------------------------------------------------------
// The analytic-shot gate's random numbers (environment.h:163: rand() % 10 + 1 after csdo.cc:93's srand(0)).  Owned by the call,
// so that the entry is re-entrant.  Two sources: a 64-bit LCG (Knuth's MMIX constants, high bits; the default, what the stored
// paths of tests/golden/front_end_paths were planned with), or - csdo_front_end_parm::rand_glibc - glibc's rand() itself:
// random_r's TYPE_3 additive feedback generator r[i] = r[i-3] + r[i-31] (mod 2^32), output r[i] >> 1, seeded as srandom_r does
// (seed 0 is taken as 1; r[i] = 16807 r[i-1] mod 2^31-1 for i < 31; the first 310 outputs discarded): the reference's sequence.
struct Rng {
  bool glibc = false;
  uint64_t lcg = 0;
  uint32_t r[34] = {0};
  int k = 0;                      // next position in the glibc generator's ring of 34
  void seed(uint32_t seed_, bool glibc_) {
    glibc = glibc_;
    lcg = seed_;
    if (!glibc) return;
    int32_t word = seed_ == 0 ? 1 : (int32_t)seed_;
    r[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
      const long hi = word / 127773, lo = word % 127773;      // Schrage: 16807 * word mod (2^31 - 1) without overflow
      word = (int32_t)(16807 * lo - 2836 * hi);
      if (word < 0) word += 2147483647;
      r[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
    k = 0;
    for (int i = 0; i < 310; ++i) (void)step();
  }
  uint32_t step() {               // position 34 + n of the sequence, kept in a ring of 34
    const uint32_t v = r[(k + 34 - 31) % 34] + r[(k + 34 - 3) % 34];
    r[k % 34] = v;
    k = (k + 1) % 34;
    return v >> 1;
  }
  uint32_t next() {
    if (glibc) return step();
    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(lcg >> 33);
  }
};

class LowLevel {
 public:
  LowLevel(const Consts& c, double maxx, double maxy, const std::vector<double>& obstacles, const std::vector<Pose>& goals,
           int agent)
      : C(c), maxx_(maxx), maxy_(maxy), obs_(obstacles), goals_(goals), agent_(agent), goal_(goals[agent]) {
    dimx_ = (int)((int)maxx / C.mapResolution);
    dimy_ = (int)((int)maxy / C.mapResolution);
    build_cost_map();
  }

  // higher: agents this one yields to; paths: current path of every agent (may be empty for unplanned ones)
  PlannedPath plan(const Pose& start, const std::set<int>& higher, const std::vector<const PlannedPath*>& paths,
                   clk::time_point deadline) {
    moving_.clear();
    parked_.clear();
    for (int a : higher) {
      const PlannedPath* p = paths[a];
      if (!p || p->empty()) continue;
      for (const Pose& s : p->states) moving_.emplace(s.t, s);
      parked_.emplace(p->states.back().t, p->states.back());   // stays at its last pose from then on
    }
    lower_goals_.clear();
    if (C.keepOffLowerGoals)   // (not a rule of the reference: environment.h:350-392) never drive through the goal of an agent that has to yield to us
      for (int a = 0; a < (int)goals_.size(); ++a)
        if (a != agent_ && !higher.count(a)) lower_goals_.push_back(goals_[a]);
    return search(start, deadline);
  }

  long expanded = 0;
  struct Rng* rng_state = nullptr;   // one generator per csdo_front_end_plan call, shared by its low-level searches

 private:
  uint32_t next_random() { return rng_state->next(); }
  struct Node {
    Pose s;
    int action;
    double f, g;
    int parent;     // index into nodes_, -1 for the start
    double step_cost;
    bool stale;
  };

  bool valid(const Pose& s) const {   // Environment::stateValid
    const double xi = s.x / C.mapResolution, yi = s.y / C.mapResolution;
    if (xi < 0 || xi >= dimx_ || yi < 0 || yi >= dimy_) return false;
    const double rv = C.rv;
    if (s.xf < rv || s.xr < rv || s.xf > maxx_ - rv || s.xr > maxx_ - rv || s.yf < rv || s.yr < rv || s.yf > maxy_ - rv ||
        s.yr > maxy_ - rv)
      return false;
    const double cs = std::cos(s.yaw), sn = std::sin(s.yaw);
    for (size_t k = 0; k + 2 < obs_.size(); k += 3) {   // State::obsCollision: obstacle in the vehicle frame
      const double ox = obs_[k] - s.x, oy = obs_[k + 1] - s.y, orad = obs_[k + 2];
      const double lx = ox * cs + oy * sn, ly = -ox * sn + oy * cs;
      if (lx > -C.LB - orad * 1.2 && lx < C.LF + orad * 1.2 && ly > -C.carWidth / 2.0 - orad * 1.2 &&
          ly < C.carWidth / 2.0 + orad * 1.2)
        return false;
    }
    for (int dt = -1; dt <= 1; ++dt) {
      const auto range = moving_.equal_range(s.t + dt);
      for (auto it = range.first; it != range.second; ++it)
        if (rect_overlap(s, it->second, C)) return false;
    }
    for (auto it = parked_.begin(); it != parked_.end() && it->first <= s.t; ++it)
      if (rect_overlap(s, it->second, C)) return false;
    if (s.t >= 1)
      for (const Pose& g : lower_goals_)
        if (rect_overlap(s, g, C)) return false;
    return true;
  }

  uint64_t index_of(const Pose& s) const {   // Environment::calcIndex
    const double cells_x = dimx_ / C.xyResolution, cells_y = dimy_ / C.xyResolution;
    return (uint64_t)((uint64_t)s.t * (2 * M_PI / C.deltat) * cells_x * cells_y) +
           (uint64_t)((uint64_t)(heading_0_2pi((float)s.yaw) / C.yawResolution) * cells_x * cells_y) +
           (uint64_t)((uint64_t)(s.y / C.xyResolution) * cells_x) + (uint64_t)(s.x / C.xyResolution);
  }

  int heuristic(const Pose& s) const {   // `int admissibleHeuristic`: the maximum of three bounds, truncated
    const rs::Path p = rs::shortest(s.x, s.y, s.yaw, goal_.x, goal_.y, goal_.yaw, C.r);
    const double rs_cost = C.r * p.total;
    const double eu = std::sqrt(std::pow(goal_.x - s.x, 2) + std::pow(goal_.y - s.y, 2));
    const double off = std::sqrt(std::pow((s.x - (int)s.x) - (goal_.x - (int)goal_.x), 2) +
                                 std::pow((s.y - (int)s.y) - (goal_.y - (int)goal_.y), 2));
    const size_t cx = (size_t)((int)s.x / C.mapResolution), cy = (size_t)((int)s.y / C.mapResolution);
    double two_d = 0;
    if (cx < (size_t)dimx_ && cy < (size_t)dimy_) two_d = cost_map_[cx * dimy_ + cy] - off;
    return (int)std::max({rs_cost, eu, two_d});
  }

  void build_cost_map() {   // Environment::updateCostmap: first-visit 8-connected wavefront from the goal cell
    cost_map_.assign((size_t)dimx_ * dimy_, 0.0);
    if (goal_.x < 0 || goal_.x > maxx_ || goal_.y < 0 || goal_.y > maxy_) return;
    std::set<std::pair<int, int>> blocked;
    for (size_t k = 0; k < obs_.size(); k += 3)
      blocked.insert({(int)((int)obs_[k] / C.mapResolution), (int)((int)obs_[k + 1] / C.mapResolution)});
    using Item = std::pair<double, std::pair<int, int>>;
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
    const int gx = (int)((int)goal_.x / C.mapResolution), gy = (int)((int)goal_.y / C.mapResolution);
    heap.push({0.0, {gx, gy}});
    while (!heap.empty()) {
      const auto node = heap.top();
      heap.pop();
      const int x = node.second.first, y = node.second.second;
      if (x < 0 || x >= dimx_ || y < 0 || y >= dimy_) continue;
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy) {
          if (!dx && !dy) continue;
          const int nx = x + dx, ny = y + dy;
          if (nx == gx && ny == gy) continue;
          if (nx >= 0 && nx < dimx_ && ny >= 0 && ny < dimy_ && cost_map_[(size_t)nx * dimy_ + ny] == 0 &&
              !blocked.count({nx, ny})) {
            cost_map_[(size_t)nx * dimy_ + ny] =
                cost_map_[(size_t)x * dimy_ + y] + std::sqrt(std::pow(dx * C.mapResolution, 2) + std::pow(dy * C.mapResolution, 2));
            heap.push({cost_map_[(size_t)nx * dimy_ + ny], {nx, ny}});
          }
        }
    }
  }

  Pose successor(const Pose& s, int act, double dx, double dy, double dyaw) const {
    (void)act;
    const double x = s.x + dx * std::cos(s.yaw) - dy * std::sin(s.yaw);
    const double y = s.y + dx * std::sin(s.yaw) + dy * std::cos(s.yaw);
    return Pose(x, y, heading_0_2pi((float)(s.yaw + dyaw)), s.t + 1, C);
  }

  // Environment::generatePath: whole primitives along one Reeds-Shepp segment, then the fractional remainder
  bool follow_segment(const Pose& from, int act, double turn, double length, std::vector<std::pair<Pose, double>>& out) const {
    out.clear();
    out.emplace_back(from, 0.0);
    double ratio;
    if (act == 0 || act == 3) {
      const size_t n = (size_t)(length / C.dx[act]);
      for (size_t i = 0; i < n; ++i) {
        const Pose nx = successor(out.back().first, act, C.dx[act], C.dy[act], C.dyaw[act]);
        if (!valid(nx)) return false;
        out.emplace_back(nx, C.dx[0]);
      }
      ratio = (length - (int)(length / C.dx[act]) * C.dx[act]) / C.dx[act];
      const Pose nx = successor(out.back().first, act, ratio * C.dx[act], 0.0, 0.0);
      if (!valid(nx)) return false;
      out.emplace_back(nx, ratio * C.dx[0]);
    } else {
      const size_t n = (size_t)(turn / C.dyaw[act]);
      for (size_t i = 0; i < n; ++i) {
        const Pose nx = successor(out.back().first, act, C.dx[act], C.dy[act], C.dyaw[act]);
        if (!valid(nx)) return false;
        out.emplace_back(nx, C.dx[0] * C.penaltyTurning);
      }
      ratio = (turn - (int)(turn / C.dyaw[act]) * C.dyaw[act]) / C.dyaw[act];
      const Pose nx = successor(out.back().first, act, ratio * C.dx[act], ratio * C.dy[act], ratio * C.dyaw[act]);
      if (!valid(nx)) return false;
      out.emplace_back(nx, ratio * C.dx[0]);
    }
    return true;
  }

  // Environment::isSolution: with probability growing towards the goal, try to finish with the Reeds-Shepp curve
  bool try_shot(int node_id, std::vector<std::pair<Pose, int>>& tail) {
    const Pose& s = nodes_[node_id].s;
    const int random = (int)(next_random() % 10u) + 1;
    const float dx = (float)(std::fabs(s.x - goal_.x) / random), dy = (float)(std::fabs(s.y - goal_.y) / random);
    if (!((dx * dx) + (dy * dy) < 100.0)) return false;
    const rs::Path p = rs::shortest(s.x, s.y, s.yaw, goal_.x, goal_.y, goal_.yaw, C.r);
    tail.clear();
    Pose cur = s;
    std::vector<std::pair<Pose, double>> seg;
    for (int k = 0; k < 5; ++k) {
      if (std::fabs(p.len[k]) < 1e-6 || p.type[k] == rs::NOP) continue;
      double turn = 0, length = 0, cost = 0;
      int act = 0;
      if (p.type[k] == rs::LEFT) {
        turn = p.len[k];
        length = C.r * p.len[k];
        act = 2;
        cost = p.len[k] * C.r * C.penaltyTurning;
      } else if (p.type[k] == rs::STRAIGHT) {
        length = p.len[k] * C.r;
        act = 0;
        cost = length;
      } else {
        turn = -p.len[k];
        length = C.r * p.len[k];
        act = 1;
        cost = p.len[k] * C.r * C.penaltyTurning;
      }
      if (cost < 0) act += 3;
      if (!follow_segment(cur, act, turn, length, seg)) return false;
      for (size_t i = 1; i < seg.size(); ++i) tail.emplace_back(seg[i].first, act);
      cur = seg.back().first;
    }
    // Deliberately stricter than the reference: the curve stops a fraction of a step short of the goal (follow_segment), and
    // an agent ranked above us only kept clear of our exact goal pose.  The reference never looks at what those agents do
    // after we arrive (stateValid tests a pose at its own time, PBS::generateChild skips conflicts with higher agents,
    // PBS.cc:189-192), so it can return a vehicle parked under a later passer-by.  Here such an ending is refused.
    for (auto it = moving_.upper_bound(cur.t + 1); it != moving_.end(); ++it)
      if (rect_overlap(cur, it->second, C)) return false;
    return true;
  }

  PlannedPath search(const Pose& start, clk::time_point deadline) {
    nodes_.clear();
    std::unordered_map<uint64_t, int> open_best;   // grid cell -> live node in the open list
    std::unordered_set<uint64_t> closed;
    struct Key {
      double f, g;
      int id;
      bool operator<(const Key& o) const { return f != o.f ? f > o.f : g < o.g; }   // lowest f, then highest g
    };
    std::priority_queue<Key> open;
    nodes_.push_back({start, 0, (double)heuristic(start), 0.0, -1, 0.0, false});
    open.push({nodes_[0].f, 0.0, 0});
    open_best[index_of(start)] = 0;
    std::vector<std::pair<Pose, int>> tail;
    long pops = 0;
    while (!open.empty()) {
      if ((double)closed.size() > C.maxClosed) {
        if (std::getenv("CSDO_FE_DEBUG")) std::fprintf(stderr, "agent %d: closed list full\n", agent_);
        return {};
      }
      if ((++pops & 255) == 0 && clk::now() > deadline) return {};
      const Key top = open.top();
      if (nodes_[top.id].stale) {
        open.pop();
        continue;
      }
      if (try_shot(top.id, tail)) {
        PlannedPath out;
        std::vector<int> chain;
        for (int n = top.id; n >= 0; n = nodes_[n].parent) chain.push_back(n);
        std::reverse(chain.begin(), chain.end());
        for (size_t i = 0; i < chain.size(); ++i) {
          out.states.push_back(nodes_[chain[i]].s);
          if (i) out.actions.push_back(nodes_[chain[i]].action);
        }
        for (const auto& st : tail) {
          out.states.push_back(st.first);
          out.actions.push_back(st.second);
        }
        return out;
      }
      open.pop();
      const int cur = top.id;
      const uint64_t cur_idx = index_of(nodes_[cur].s);
      open_best.erase(cur_idx);
      closed.insert(cur_idx);
      ++expanded;
      const Pose s = nodes_[cur].s;
      const int last_act = nodes_[cur].action;
      for (int act = 0; act <= 6; ++act) {
        Pose nx;
        double g = C.dx[0];
        if (act < 6) {
          nx = successor(s, act, C.dx[act], C.dy[act], C.dyaw[act]);
          if (act % 3 != 0) g = g * C.penaltyTurning;
          if ((act < 3 && last_act >= 3) || (last_act < 3 && act >= 3)) g = g * C.penaltyCOD;
          if (act >= 3) g = g * C.penaltyReversing;
        } else {
          nx = Pose(s.x, s.y, s.yaw, s.t + 1, C);   // wait
        }
        if (!valid(nx)) continue;
        const uint64_t idx = index_of(nx);
        if (closed.count(idx)) continue;
        const double tentative = nodes_[cur].g + g;
        const auto it = open_best.find(idx);
        double f;
        if (it == open_best.end()) {
          f = tentative + heuristic(nx);
        } else {
          Node& old = nodes_[it->second];
          if (tentative >= old.g) continue;
          f = old.f - (old.g - tentative);   // same cell: keep its heuristic value, as the reference's decrease-key does
          old.stale = true;
        }
        nodes_.push_back({nx, act, f, tentative, cur, g, false});
        const int id = (int)nodes_.size() - 1;
        open_best[idx] = id;
        open.push({f, tentative, id});
      }
    }
    return {};
  }

  const Consts& C;
  double maxx_, maxy_;
  int dimx_ = 0, dimy_ = 0;
  const std::vector<double>& obs_;
  const std::vector<Pose>& goals_;
  int agent_;
  Pose goal_;
  std::vector<double> cost_map_;
  std::multimap<int, Pose> moving_;   // timestep -> poses of the agents ahead of us
  std::multimap<int, Pose> parked_;   // arrival time -> final pose
  std::vector<Pose> lower_goals_;
  std::vector<Node> nodes_;
};

// ---------------------------------------------------------------------------------------------------------------------
// high level
// ---------------------------------------------------------------------------------------------------------------------
struct HlNode {
  std::deque<std::pair<int, PlannedPath>> paths;   // re-planned in this node (deque: growing keeps references valid)
  std::list<std::pair<int, int>> conflicts;
  HlNode* parent = nullptr;
  int low = -1, high = -1;   // the priority this node adds: `low` yields to `high`
  long cost = 0;
};

class Pbs {
 public:
  Pbs(const Consts& c, double dimx, double dimy, const std::vector<double>& obstacles, const std::vector<Pose>& starts,
      const std::vector<Pose>& goals)
      : C(c), starts_(starts), goals_(goals), n_((int)starts.size()) {
    for (int a = 0; a < n_; ++a) {
      engines_.emplace_back(new LowLevel(C, dimx, dimy, obstacles, goals_, a));
      engines_.back()->rng_state = &rng_;
    }
  }
  void seed(uint32_t s, bool glibc) { rng_.seed(s, glibc); }
  ~Pbs() {
    for (HlNode* n : all_) delete n;
    for (LowLevel* e : engines_) delete e;
  }

  bool solve(double time_limit, long node_limit) {
    deadline_ = clk::now() + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(time_limit));
    if (!make_root()) return false;
    while (!stack_.empty()) {
      HlNode* cur = stack_.back();
      stack_.pop_back();
      adopt(cur);
      ++hl_expanded;
      if (cur->conflicts.empty()) {
        goal_ = cur;
        return true;
      }
      if (clk::now() > deadline_ || hl_expanded > node_limit) return false;
      const auto conflict = cur->conflicts.back();   // PBS::chooseConflict: the most recent one
      const std::vector<const PlannedPath*> saved(paths_);
      HlNode* c0 = make_child(cur, conflict.first, conflict.second);
      paths_ = saved;
      HlNode* c1 = make_child(cur, conflict.second, conflict.first);
      // depth first, the cheaper child on top (PBS::pushNodes)
      if (c0 && c1) {
        if (c0->cost < c1->cost) {
          stack_.push_back(c1);
          stack_.push_back(c0);
        } else {
          stack_.push_back(c0);
          stack_.push_back(c1);
        }
      } else if (c0) {
        stack_.push_back(c0);
      } else if (c1) {
        stack_.push_back(c1);
      }
    }
    return false;
  }

  const std::vector<const PlannedPath*>& paths() const { return paths_; }
  long hl_expanded = 0, hl_generated = 0;
  long ll_expanded() const {
    long s = 0;
    for (const LowLevel* e : engines_) s += e->expanded;
    return s;
  }

 private:
  bool collide(int a1, int a2) const {   // PBS::hasConflicts: same timestep, then the shorter one parked at its goal
    const PlannedPath &p = *paths_[a1], &q = *paths_[a2];
    const size_t m = std::min(p.size(), q.size());
    for (size_t t = 0; t < m; ++t)
      if (rect_overlap(p.states[t], q.states[t], C)) return true;
    const PlannedPath& shorter = p.size() < q.size() ? p : q;
    const PlannedPath& longer = p.size() < q.size() ? q : p;
    for (size_t t = m; t < longer.size(); ++t)
      if (rect_overlap(shorter.states.back(), longer.states[t], C)) return true;
    return false;
  }

  bool make_root() {
    HlNode* root = new HlNode();
    all_.push_back(root);
    paths_.assign(n_, nullptr);
    std::set<int> higher;
    for (int a = 0; a < n_; ++a) {
      PlannedPath p = engines_[a]->plan(starts_[a], higher, paths_, deadline_);
      if (p.empty()) p = engines_[a]->plan(starts_[a], {}, paths_, deadline_);   // boxed in: plan alone, let PBS sort it out
      if (p.empty()) {
        if (std::getenv("CSDO_FE_DEBUG")) std::fprintf(stderr, "root: agent %d has no path\n", a);
        return false;
      }
      root->paths.emplace_back(a, std::move(p));
      paths_[a] = &root->paths.back().second;
      root->cost += (long)paths_[a]->size() - 1;
      higher.insert(a);
    }
    for (int a1 = 0; a1 < n_; ++a1)
      for (int a2 = a1 + 1; a2 < n_; ++a2)
        if (collide(a1, a2)) root->conflicts.emplace_back(a1, a2);
    ++hl_generated;
    stack_.push_back(root);
    return true;
  }

  // PBS::update: the node's view of the world = newest path of every agent along the branch + the branch's priorities
  void adopt(HlNode* node) {
    paths_.assign(n_, nullptr);
    prio_.assign(n_, std::vector<char>(n_, 0));
    for (HlNode* c = node; c; c = c->parent) {
      for (auto& p : c->paths)
        if (!paths_[p.first]) paths_[p.first] = &p.second;
      if (c->parent) prio_[c->low][c->high] = 1;
    }
  }

  void topo_visit(int v, std::vector<char>& seen, std::vector<int>& order) const {
    seen[v] = 1;
    for (int i = 0; i < n_; ++i)
      if (prio_[v][i] && !seen[i]) topo_visit(i, seen, order);
    order.push_back(v);   // every agent v yields to comes before v: `order` lists agents from first planned to last
  }
  void higher_of(int a, std::set<int>& out) const {
    for (int i = 0; i < n_; ++i)
      if (prio_[a][i] && out.insert(i).second) higher_of(i, out);
  }
  void lower_of(int a, std::set<int>& out) const {
    for (int i = 0; i < n_; ++i)
      if (prio_[i][a] && out.insert(i).second) lower_of(i, out);
  }

  HlNode* make_child(HlNode* parent, int low, int high) {
    prio_[high][low] = 0;   // the sibling's order, if it was tried first
    {   // `high` already yields to `low` along this branch: the opposite order would close a cycle
      std::set<int> below;
      lower_of(low, below);
      if (below.count(high)) {
        if (std::getenv("CSDO_FE_DEBUG")) std::fprintf(stderr, "child %d<%d: cycle\n", low, high);
        return nullptr;
      }
    }
    HlNode* node = new HlNode();
    node->parent = parent;
    node->low = low;
    node->high = high;
    node->cost = parent->cost;
    node->conflicts = parent->conflicts;
    prio_[high][low] = 0;
    prio_[low][high] = 1;
    std::vector<char> seen(n_, 0);
    std::vector<int> order;
    for (int i = 0; i < n_; ++i)
      if (!seen[i]) topo_visit(i, seen, order);
    std::vector<int> rank(n_);   // larger = planned earlier = higher priority
    for (int i = 0; i < n_; ++i) rank[order[i]] = n_ - 1 - i;
    std::priority_queue<std::pair<int, int>> todo;   // most important agent first
    std::vector<char> queued(n_, 0);
    todo.emplace(rank[low], low);
    queued[low] = 1;
    {   // known conflicts between something above `high` and something below `low` must be re-planned too
      std::set<int> above, below;
      higher_of(high, above);
      above.insert(high);
      lower_of(low, below);
      for (const auto& c : node->conflicts) {
        int a1 = c.first, a2 = c.second;
        if (a1 == low || a2 == low) continue;
        if (rank[a1] > rank[a2]) std::swap(a1, a2);
        if (!queued[a1] && below.count(a1) && above.count(a2)) {
          todo.emplace(rank[a1], a1);
          queued[a1] = 1;
        }
      }
    }
    while (!todo.empty()) {
      const int a = todo.top().second;
      todo.pop();
      queued[a] = 0;
      std::set<int> above;
      higher_of(a, above);
      PlannedPath p = engines_[a]->plan(starts_[a], above, paths_, deadline_);
      if (p.empty()) {
        if (std::getenv("CSDO_FE_DEBUG")) std::fprintf(stderr, "child %d<%d: agent %d has no path (above %zu)\n", low, high, a, above.size());
        prio_[low][high] = 0;
        delete node;
        return nullptr;
      }
      node->cost += (long)p.size() - (long)paths_[a]->size();
      node->paths.emplace_back(a, std::move(p));
      paths_[a] = &node->paths.back().second;
      for (auto it = node->conflicts.begin(); it != node->conflicts.end();)
        it = (it->first == a || it->second == a) ? node->conflicts.erase(it) : std::next(it);
      std::set<int> below;
      lower_of(a, below);
      for (int b = 0; b < n_; ++b) {
        if (b == a || queued[b] || above.count(b)) continue;
        if (collide(a, b)) {
          node->conflicts.emplace_back(a, b);
          if (below.count(b)) {   // the new path runs into an agent that has to yield to it
            todo.emplace(rank[b], b);
            queued[b] = 1;
          }
        }
      }
    }
    ++hl_generated;
    all_.push_back(node);
    return node;
  }

  const Consts& C;
  const std::vector<Pose>& starts_;
  const std::vector<Pose>& goals_;
  int n_;
  std::vector<LowLevel*> engines_;
  std::vector<const PlannedPath*> paths_;
  std::vector<std::vector<char>> prio_;   // prio_[a][b]: a yields to b
  std::vector<HlNode*> stack_, all_;
  HlNode* goal_ = nullptr;
  clk::time_point deadline_;
  Rng rng_;
};

}  // namespace

int front_end_plan(const double* starts, const double* goals, int Na, double dimx, double dimy, const double* obstacles,
                   int n_obs, const csdo_vehicle* veh, const csdo_front_end_parm* parm, csdo_paths* out) {
  if (!starts || !goals || !veh || !parm || !out || Na < 1 || n_obs < 0 || (n_obs > 0 && !obstacles)) return CSDO_EINVAL;
  std::memset(out, 0, sizeof(*out));
  const Consts C = make_consts(*veh, *parm);
  std::vector<double> obs(obstacles, obstacles + (size_t)3 * n_obs);
  std::vector<Pose> S, G;
  for (int a = 0; a < Na; ++a) {
    S.emplace_back(starts[3 * a], starts[3 * a + 1], starts[3 * a + 2], 0, C);
    G.emplace_back(goals[3 * a], goals[3 * a + 1], goals[3 * a + 2], 0, C);
  }
  const auto t0 = clk::now();
  Pbs pbs(C, dimx, dimy, obs, S, G);
  pbs.seed(parm->rand_seed, parm->rand_glibc != 0);   // csdo.cc:93 seeds the C library's generator with 0 before the search
  const bool ok = pbs.solve(parm->time_limit_s > 0 ? parm->time_limit_s : 20.0, parm->node_limit > 0 ? parm->node_limit : 1000000);
  out->seconds = std::chrono::duration<double>(clk::now() - t0).count();
  out->hl_expanded = (int32_t)pbs.hl_expanded;
  out->hl_generated = (int32_t)pbs.hl_generated;
  out->ll_expanded = (int64_t)pbs.ll_expanded();
  out->Na = Na;
  out->status = ok ? 1 : 0;
  if (!ok) return CSDO_OK;
  size_t total = 0;
  for (int a = 0; a < Na; ++a) total += pbs.paths()[a]->size();
  out->path_off = (int32_t*)std::malloc(sizeof(int32_t) * (Na + 1));
  out->states = (double*)std::malloc(sizeof(double) * 3 * total);
  out->actions = (int32_t*)std::malloc(sizeof(int32_t) * std::max<size_t>(total - Na, 1));
  if (!out->path_off || !out->states || !out->actions) {
    front_end_free(out);
    return CSDO_ENOMEM;
  }
  size_t so = 0, ao = 0;
  for (int a = 0; a < Na; ++a) {
    const PlannedPath& p = *pbs.paths()[a];
    out->path_off[a] = (int32_t)so;
    for (size_t i = 0; i < p.size(); ++i) {
      out->states[3 * (so + i)] = p.states[i].x;
      out->states[3 * (so + i) + 1] = p.states[i].y;
      out->states[3 * (so + i) + 2] = p.states[i].yaw;
    }
    for (size_t i = 0; i + 1 < p.size(); ++i) out->actions[ao + i] = p.actions[i];
    so += p.size();
    ao += p.size() - 1;
  }
  out->path_off[Na] = (int32_t)so;
  return CSDO_OK;
}

void front_end_free(csdo_paths* p) {
  if (!p) return;
  std::free(p->path_off);
  std::free(p->states);
  std::free(p->actions);
  p->path_off = nullptr;
  p->states = nullptr;
  p->actions = nullptr;
}

}  // namespace csdo

extern "C" {
void csdo_front_end_parm_default(csdo_front_end_parm* p) {   // config.yaml of the reference + csdo.cc:100
  if (!p) return;
  p->penalty_turning = 1.5;
  p->penalty_reversing = 2.0;
  p->penalty_cod = 2.0;
  p->map_resolution = 2.0;
  p->max_closed_set_size = 1e5;
  p->time_limit_s = 20.0;
  p->node_limit = 0;
  p->rand_seed = 0;
  p->keep_off_lower_goals = 1;
  p->rand_glibc = 0;
}

int csdo_front_end_gate_draws(uint32_t rand_seed, int32_t rand_glibc, int32_t n, uint32_t* out) {
  if (n < 0 || (n > 0 && !out)) return CSDO_EINVAL;
  csdo::Rng g;
  g.seed(rand_seed, rand_glibc != 0);
  for (int i = 0; i < n; ++i) out[i] = g.next();
  return CSDO_OK;
}
int csdo_front_end_plan(const double* starts, const double* goals, int32_t Na, double dimx, double dimy,
                        const double* obstacles, int32_t n_obs, const csdo_vehicle* veh, const csdo_front_end_parm* parm,
                        csdo_paths* out) {
  try {
    return csdo::front_end_plan(starts, goals, Na, dimx, dimy, obstacles, n_obs, veh, parm, out);
  } catch (const std::bad_alloc&) {
    return CSDO_ENOMEM;
  } catch (...) {
    return CSDO_EINVAL;
  }
}
void csdo_paths_free(csdo_paths* p) { csdo::front_end_free(p); }
double csdo_reeds_shepp(const double from[3], const double to[3], double rho, int32_t types[5], double lengths[5]) {
  if (!from || !to || !(rho > 0)) return -1.0;
  const csdo::rs::Path p = csdo::rs::shortest(from[0], from[1], from[2], to[0], to[1], to[2], rho);
  for (int k = 0; k < 5; ++k) {
    if (types) types[k] = (int32_t)p.type[k];
    if (lengths) lengths[k] = p.len[k];
  }
  return rho * p.total;
}
}
