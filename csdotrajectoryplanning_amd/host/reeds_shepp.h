// reeds_shepp.h — shortest Reeds-Shepp curve between two poses (unit turning radius), own implementation of the word
// families of Reeds & Shepp, "Optimal paths for a car that goes both forwards and backwards" (Pacific J. Math. 145, 1990):
// CSC, CCC, CCCC, CCSC, CCSCC with the time-flip / reflection / backward symmetries.  The reference's front end gets the same
// curves from OMPL's ReedsSheppStateSpace (hybrid_a_star/environment.h:128-137,182-190): length for the heuristic, segment list
// for the analytic expansion.  Header-only, no dependencies.
#pragma once
#include <cmath>
#include <limits>

namespace csdo {
namespace rs {

enum Seg : int { NOP = 0, LEFT = 1, STRAIGHT = 2, RIGHT = 3 };   // numbering of the reference's switch (environment.h:202-228)

struct Path {
  Seg type[5] = {NOP, NOP, NOP, NOP, NOP};
  double len[5] = {0, 0, 0, 0, 0};   // signed, in units of the turning radius (negative: driven in reverse)
  double total = std::numeric_limits<double>::infinity();
};

namespace detail {
constexpr double PI = 3.14159265358979323846, EPS = 1e-6, ZERO = 10 * 2.220446049250313e-16;

inline double wrap(double a) {   // (-pi, pi]
  double v = std::fmod(a, 2 * PI);
  if (v < -PI) v += 2 * PI;
  else if (v > PI) v -= 2 * PI;
  return v;
}
inline void polar(double x, double y, double& r, double& th) {
  r = std::sqrt(x * x + y * y);
  th = std::atan2(y, x);
}
inline void tau_omega(double u, double v, double xi, double eta, double phi, double& tau, double& omega) {
  const double delta = wrap(u - v), A = std::sin(u) - std::sin(delta), B = std::cos(u) - std::cos(delta) - 1.0;
  const double t1 = std::atan2(eta * A - xi * B, xi * A + eta * B);
  const double t2 = 2.0 * (std::cos(delta) - std::cos(v) - std::cos(u)) + 3.0;
  tau = (t2 < 0) ? wrap(t1 + PI) : wrap(t1);
  omega = wrap(tau - u + v - phi);
}

struct Best {
  Path p;
  void offer(const Seg (&ty)[5], double a, double b, double c, double d = 0.0, double e = 0.0) {
    const double L = std::fabs(a) + std::fabs(b) + std::fabs(c) + std::fabs(d) + std::fabs(e);
    if (L < p.total) {
      const double l[5] = {a, b, c, d, e};
      for (int k = 0; k < 5; ++k) {
        p.type[k] = ty[k];
        p.len[k] = l[k];
      }
      p.total = L;
    }
  }
};

// the five base formulas (positive-first words); each returns true and the parameters when the word exists
inline bool LpSpLp(double x, double y, double phi, double& t, double& u, double& v) {
  polar(x - std::sin(phi), y - 1.0 + std::cos(phi), u, t);
  if (t >= -ZERO) {
    v = wrap(phi - t);
    if (v >= -ZERO) return true;
  }
  return false;
}
inline bool LpSpRp(double x, double y, double phi, double& t, double& u, double& v) {
  double t1, u1;
  polar(x + std::sin(phi), y - 1.0 - std::cos(phi), u1, t1);
  u1 = u1 * u1;
  if (u1 >= 4.0) {
    u = std::sqrt(u1 - 4.0);
    const double theta = std::atan2(2.0, u);
    t = wrap(t1 + theta);
    v = wrap(t - phi);
    return t >= -ZERO && v >= -ZERO;
  }
  return false;
}
inline bool LpRmL(double x, double y, double phi, double& t, double& u, double& v) {
  const double xi = x - std::sin(phi), eta = y - 1.0 + std::cos(phi);
  double u1, theta;
  polar(xi, eta, u1, theta);
  if (u1 <= 4.0) {
    u = -2.0 * std::asin(0.25 * u1);
    t = wrap(theta + 0.5 * u + PI);
    v = wrap(phi - t + u);
    return t >= -ZERO && u <= ZERO;
  }
  return false;
}
inline bool LpRupLumRm(double x, double y, double phi, double& t, double& u, double& v) {
  const double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi), rho = 0.25 * (2.0 + std::sqrt(xi * xi + eta * eta));
  if (rho <= 1.0) {
    u = std::acos(rho);
    tau_omega(u, -u, xi, eta, phi, t, v);
    return t >= -ZERO && v <= ZERO;
  }
  return false;
}
inline bool LpRumLumRp(double x, double y, double phi, double& t, double& u, double& v) {
  const double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi), rho = (20.0 - xi * xi - eta * eta) / 16.0;
  if (rho >= 0 && rho <= 1) {
    u = -std::acos(rho);
    if (u >= -0.5 * PI) {
      tau_omega(u, u, xi, eta, phi, t, v);
      return t >= -ZERO && v >= -ZERO;
    }
  }
  return false;
}
inline bool LpRmSmLm(double x, double y, double phi, double& t, double& u, double& v) {
  const double xi = x - std::sin(phi), eta = y - 1.0 + std::cos(phi);
  double rho, theta;
  polar(xi, eta, rho, theta);
  if (rho >= 2.0) {
    const double r = std::sqrt(rho * rho - 4.0);
    u = 2.0 - r;
    t = wrap(theta + std::atan2(r, -2.0));
    v = wrap(phi - 0.5 * PI - t);
    return t >= -ZERO && u <= ZERO && v <= ZERO;
  }
  return false;
}
inline bool LpRmSmRm(double x, double y, double phi, double& t, double& u, double& v) {
  const double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi);
  double rho, theta;
  polar(-eta, xi, rho, theta);
  if (rho >= 2.0) {
    t = theta;
    u = 2.0 - rho;
    v = wrap(t + 0.5 * PI - phi);
    return t >= -ZERO && u <= ZERO && v <= ZERO;
  }
  return false;
}
inline bool LpRmSLmRp(double x, double y, double phi, double& t, double& u, double& v) {
  const double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi);
  double rho, theta;
  polar(xi, eta, rho, theta);
  if (rho >= 2.0) {
    u = 4.0 - std::sqrt(rho * rho - 4.0);
    if (u <= ZERO) {
      t = wrap(std::atan2((4.0 - u) * xi - 2.0 * eta, -2.0 * xi + (u - 4.0) * eta));
      v = wrap(t - phi);
      return t >= -ZERO && v >= -ZERO;
    }
  }
  return false;
}

// a word and its three mirror images: time flip (drive it backwards), reflection (swap left and right), both
template <class F>
inline void four_ways(F&& formula, double x, double y, double phi, const Seg (&ty)[5], const Seg (&ty_reflected)[5],
                      int n, const int (&slot)[3], const double (&fixed)[5], Best& best) {
  // slot[k]: which of the n segments gets t, u, v; fixed[]: preset lengths (the quarter turns of the CCSC / CCSCC words)
  double t, u, v;
  auto emit = [&](const Seg (&types)[5], double sign) {
    double l[5] = {fixed[0] * sign, fixed[1] * sign, fixed[2] * sign, fixed[3] * sign, fixed[4] * sign};
    l[slot[0]] = sign * t;
    l[slot[1]] = sign * u;
    l[slot[2]] = sign * v;
    (void)n;
    best.offer(types, l[0], l[1], l[2], l[3], l[4]);
  };
  if (formula(x, y, phi, t, u, v)) emit(ty, 1.0);
  if (formula(-x, y, -phi, t, u, v)) emit(ty, -1.0);
  if (formula(x, -y, -phi, t, u, v)) emit(ty_reflected, 1.0);
  if (formula(-x, -y, phi, t, u, v)) emit(ty_reflected, -1.0);
}
}  // namespace detail

// shortest curve from the origin pose (0, 0, 0) to (x, y, phi), everything in units of the turning radius
inline Path shortest(double x, double y, double phi) {
  using namespace detail;
  Best best;
  const Seg N = NOP, L = LEFT, S = STRAIGHT, R = RIGHT;
  const double Q = -0.5 * PI;
  // CSC
  four_ways(LpSpLp, x, y, phi, {L, S, L, N, N}, {R, S, R, N, N}, 3, {0, 1, 2}, {0, 0, 0, 0, 0}, best);
  four_ways(LpSpRp, x, y, phi, {L, S, R, N, N}, {R, S, L, N, N}, 3, {0, 1, 2}, {0, 0, 0, 0, 0}, best);
  // CCC, and the same word entered from the far end (the "backwards" image)
  const double xb = x * std::cos(phi) + y * std::sin(phi), yb = x * std::sin(phi) - y * std::cos(phi);
  four_ways(LpRmL, x, y, phi, {L, R, L, N, N}, {R, L, R, N, N}, 3, {0, 1, 2}, {0, 0, 0, 0, 0}, best);
  four_ways(LpRmL, xb, yb, phi, {L, R, L, N, N}, {R, L, R, N, N}, 3, {2, 1, 0}, {0, 0, 0, 0, 0}, best);
  // CCCC: the two inner arcs have equal length u (second one -u for the first word)
  {
    double t, u, v;
    auto cccc = [&](bool (*f)(double, double, double, double&, double&, double&), double sgn_u2) {
      const double xs[4] = {x, -x, x, -x}, ys[4] = {y, y, -y, -y}, ps[4] = {phi, -phi, -phi, phi};
      for (int k = 0; k < 4; ++k)
        if (f(xs[k], ys[k], ps[k], t, u, v)) {
          const double s = (k & 1) ? -1.0 : 1.0;
          const Seg a[5] = {L, R, L, R, N}, b[5] = {R, L, R, L, N};
          best.offer(k < 2 ? a : b, s * t, s * u, s * sgn_u2 * u, s * v);
        }
    };
    cccc(LpRupLumRm, -1.0);
    cccc(LpRumLumRp, 1.0);
  }
  // CCSC and its far-end images
  four_ways(LpRmSmLm, x, y, phi, {L, R, S, L, N}, {R, L, S, R, N}, 4, {0, 2, 3}, {0, Q, 0, 0, 0}, best);
  four_ways(LpRmSmRm, x, y, phi, {L, R, S, R, N}, {R, L, S, L, N}, 4, {0, 2, 3}, {0, Q, 0, 0, 0}, best);
  four_ways(LpRmSmLm, xb, yb, phi, {L, S, R, L, N}, {R, S, L, R, N}, 4, {3, 1, 0}, {0, 0, Q, 0, 0}, best);
  four_ways(LpRmSmRm, xb, yb, phi, {R, S, R, L, N}, {L, S, L, R, N}, 4, {3, 1, 0}, {0, 0, Q, 0, 0}, best);
  // CCSCC
  four_ways(LpRmSLmRp, x, y, phi, {L, R, S, L, R}, {R, L, S, R, L}, 5, {0, 2, 4}, {0, Q, 0, Q, 0}, best);
  return best.p;
}

// between two poses with turning radius rho; the curve's length is rho * path.total
inline Path shortest(double x0, double y0, double yaw0, double x1, double y1, double yaw1, double rho) {
  const double dx = x1 - x0, dy = y1 - y0, c = std::cos(yaw0), s = std::sin(yaw0);
  return shortest((c * dx + s * dy) / rho, (-s * dx + c * dy) / rho, yaw1 - yaw0);
}

// end pose of a curve driven from (x0, y0, yaw0): used by the tests to check that a curve arrives where it should
inline void end_pose(const Path& p, double x0, double y0, double yaw0, double rho, double& x, double& y, double& yaw) {
  x = x0;
  y = y0;
  yaw = yaw0;
  for (int k = 0; k < 5; ++k) {
    const double l = p.len[k];
    if (p.type[k] == STRAIGHT) {
      x += rho * l * std::cos(yaw);
      y += rho * l * std::sin(yaw);
    } else if (p.type[k] == LEFT) {
      x += rho * (std::sin(yaw + l) - std::sin(yaw));
      y += rho * (-std::cos(yaw + l) + std::cos(yaw));
      yaw += l;
    } else if (p.type[k] == RIGHT) {
      x += rho * (-std::sin(yaw - l) + std::sin(yaw));
      y += rho * (std::cos(yaw - l) - std::cos(yaw));
      yaw -= l;
    }
  }
}

}  // namespace rs
}  // namespace csdo
