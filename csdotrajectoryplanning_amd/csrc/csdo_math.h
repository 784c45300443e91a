// csdo_math.h — sin, cos, tan and atan2 of the per-agent program, ONE source for every build of it.
//
// Why: the device program calls sin / cos / tan / atan2 where the reference calls them (calcKineConstraint
// sqp/dsqp_solver.cc:646-744, updateCorridor :828-831, isFeasible :336-339, generateLegalPoint sqp/corridor.cc:84-122).  With the
// device's libm (ocml) in the HIP build and glibc in the lane-serial host build of the same source, the two builds differed by
// ulps in those values, and the SQP chain amplifies an ulp: the strongest witness there is - same program,
// run on the CPU - could not be held to bit equality.  These functions use nothing but +, -, *, /, fma and rint, all of them
// correctly rounded IEEE operations on gfx950 and on x86-64, in an order the compilers may not change (-ffp-contract=off, no
// fast-math: fused multiply-adds appear exactly where fma() is written), so hipcc and g++ produce the same bits.
//
// Included by dsqp_program.h (device build and tests/emu), and by the oracle's third build (oracle/Makefile:
// libcsdo_oracle_xm.so, -DCSDO_ORACLE_SHARED_TRIG), which separates "another libm" from "another formulation" in the
// chain-parity report (scripts/chain_parity.py).
//
// Accuracy (tests/test_shared_math.py against mpmath): sin, cos, tan < 0.85 ulp, atan2 < 0.6 ulp for |x| <= 2^20 pi/2; they
// agree with glibc 2.35's results in > 95 % of random arguments.  Not claimed: arguments beyond 2^20 pi/2 lose accuracy in
// the reduction (a yaw angle along a path does not get there), beyond 2^52 the result is that of 0; sin(-0.0) is +0.0.
//
// The polynomials are plain Taylor sums (coefficients 1 / n!, 1 / n rounded to double), two terms longer than a minimax fit
// would need; the argument reduction is the classical three-constant one (pi/2 = P1 + P2 + P3 with 33 + 33 + 53 bits, so
// that k P1 and k P2 are exact products for |k| < 2^20) carried in double-double.
#pragma once
#include <cmath>

#ifndef CSDO_FN
#define CSDO_FN inline
#endif

namespace csdo {
namespace xm {

// s + e = a + b exactly
CSDO_FN void two_sum(const double a, const double b, double& s, double& e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}

// x = k pi/2 + (r + rl), |r| <= pi/4 (+ an ulp); returns k mod 4
CSDO_FN int reduce_pio2(const double x, double& r, double& rl) {
  constexpr double PIO4 = 0.7853981633974483, TWO_OVER_PI = 0.6366197723675814;
  constexpr double P1 = 0x1.921fb54400000p+0, P2 = 0x1.0b4611a600000p-34, P3 = 0x1.3198a2e037073p-69;
  const double ax = fabs(x);
  if (ax <= PIO4) {
    r = x;
    rl = 0.0;
    return 0;
  }
  if (!(ax < 0x1p52)) {   // non-finite: NaN; huge: the result of 0 (see the header)
    r = x - x;
    rl = 0.0;
    return 0;
  }
  const double k = rint(x * TWO_OVER_PI);
  const double r0 = fma(-k, P1, x);   // exact for |k| < 2^20
  const double t = k * P2;            // exact
  double rh, e1;
  two_sum(r0, -t, rh, e1);
  const double w = k * P3, wl = fma(k, P3, -w);
  double rh2, e2;
  two_sum(rh, -w, rh2, e2);
  const double tail = (e1 + e2) - wl;
  r = rh2 + tail;
  rl = (rh2 - r) + tail;
  const double q = fma(-4.0, rint(k * 0.25), k);   // k mod 4 in [-2, 2], without an integer conversion of k itself
  return (int)q & 3;
}

// sin(r + rl) = s + sl, cos(r + rl) = c + cl for |r| <= pi/4.  The two leading terms are carried exactly - r and r^3 / 6 (a
// double-double product: it is up to a ninth of the result), 1 and r^2 / 2 (with what the square lost) -, the rest of the
// series is a correction of less than a hundredth of the result, and every sum ends in ONE rounding of an error-free pair.
CSDO_FN void sincos_kernel(const double r, const double rl, double& s, double& sl, double& c, double& cl) {
  constexpr double S1H = -0.16666666666666666, S1L = -9.25185853854297e-18;   // -1/6
  constexpr double S2 = 0.008333333333333333, S3 = -0.0001984126984126984, S4 = 2.7557319223985893e-06,
                   S5 = -2.505210838544172e-08, S6 = 1.6059043836821613e-10, S7 = -7.647163731819816e-13,
                   S8 = 2.8114572543455206e-15;
  constexpr double C2 = 0.041666666666666664, C3 = -0.001388888888888889, C4 = 2.48015873015873e-05,
                   C5 = -2.755731922398589e-07, C6 = 2.08767569878681e-09, C7 = -1.1470745597729725e-11,
                   C8 = 4.779477332387385e-14, C9 = -1.5619206968586225e-16;
  const double z = r * r, zl = fma(r, r, -z);
  double ps = fma(z, S8, S7);
  ps = fma(z, ps, S6);
  ps = fma(z, ps, S5);
  ps = fma(z, ps, S4);
  ps = fma(z, ps, S3);
  ps = fma(z, ps, S2);
  const double v = z * r, vl = fma(z, r, -v) + zl * r;            // r^3 = v + vl
  const double t = v * S1H, tl = fma(v, S1H, -t) + fma(v, S1L, vl * S1H);   // -r^3 / 6 = t + tl
  double h, hl;
  two_sum(r, t, h, hl);
  const double cs = (hl + tl) + fma(v * z, ps, rl * fma(-0.5, z, 1.0));   // + r^5 (S2 + ...) + rl cos(r)
  s = h + cs;
  sl = (h - s) + cs;
  double pc = fma(z, C9, C8);
  pc = fma(z, pc, C7);
  pc = fma(z, pc, C6);
  pc = fma(z, pc, C5);
  pc = fma(z, pc, C4);
  pc = fma(z, pc, C3);
  pc = fma(z, pc, C2);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cc = (((1.0 - w) - hz) - 0.5 * zl) + fma(z * z, pc, -(r * rl));   // r^4 (C2 + ...) - rl sin(r), and what 1 - hz lost
  c = w + cc;
  cl = (w - c) + cc;
}

CSDO_FN void sincos(const double x, double& sn, double& cs) {
  double r, rl, s, sl, c, cl;
  const int q = reduce_pio2(x, r, rl);
  sincos_kernel(r, rl, s, sl, c, cl);
  const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;
  sn = (q & 2) ? -a : a;
  cs = ((q + 1) & 2) ? -b : b;
}
CSDO_FN double sin(const double x) {
  double s, c;
  sincos(x, s, c);
  return s;
}
CSDO_FN double cos(const double x) {
  double s, c;
  sincos(x, s, c);
  return c;
}
// tan = (s + sl) / (c + cl), or -(c + cl) / (s + sl) in the odd quadrants: one division, then one step on the exact residual
CSDO_FN double tan(const double x) {
  double r, rl, s, sl, c, cl;
  const int q = reduce_pio2(x, r, rl);
  sincos_kernel(r, rl, s, sl, c, cl);
  const bool odd = (q & 1) != 0;
  const double n = odd ? -c : s, nl = odd ? -cl : sl, d = odd ? s : c, dl = odd ? sl : cl;
  const double inv = 1.0 / d;
  const double t = n * inv;
  const double rem = fma(-t, d, n);
  return fma((rem + nl) - t * dl, inv, t);
}

// atan(a / b) = ah + v + sm for 0 <= a <= b, b > 0: a / b = tan(atan(c) + atan(v)) with c the nearest of 0, 1/4, 1/2, 3/4, 1 and
// v = (a - c b) / (b + c a), |v| <= 1/8.  ah and v are the two leading parts, sm collects everything small: the low part
// of atan(c), what the quotient and its two operands lost to rounding, and the series behind its first term.
CSDO_FN void atan_ratio(const double a, const double b, double& ah, double& v, double& sm) {
  constexpr double A1 = -0.3333333333333333, A2 = 0.2, A3 = -0.14285714285714285, A4 = 0.1111111111111111,
                   A5 = -0.09090909090909091, A6 = 0.07692307692307693, A7 = -0.06666666666666667,
                   A8 = 0.058823529411764705, A9 = -0.05263157894736842;
  const double a8 = 8.0 * a;
  double c = 1.0, al = 3.061616997868383e-17;
  ah = 0.7853981633974483;
  if (a8 < b) {
    c = 0.0; ah = 0.0; al = 0.0;
  } else if (a8 < 3.0 * b) {
    c = 0.25; ah = 0.24497866312686414; al = 1.0698755618734451e-17;
  } else if (a8 < 5.0 * b) {
    c = 0.5; ah = 0.4636476090008061; al = 2.2698777452961687e-17;
  } else if (a8 < 7.0 * b) {
    c = 0.75; ah = 0.6435011087932844; al = 1.5834785051444286e-17;
  }
  const double pb = c * b, pbl = fma(c, b, -pb), pa = c * a, pal = fma(c, a, -pa);
  double num, nl, den, dl;
  two_sum(a, -pb, num, nl);
  nl -= pbl;
  two_sum(b, pa, den, dl);
  dl += pal;
  const double inv = 1.0 / den;
  v = num * inv;
  const double vl = ((fma(-v, den, num) + nl) - v * dl) * inv;
  const double z = v * v;
  double p = fma(z, A9, A8);
  p = fma(z, p, A7);
  p = fma(z, p, A6);
  p = fma(z, p, A5);
  p = fma(z, p, A4);
  p = fma(z, p, A3);
  p = fma(z, p, A2);
  p = fma(z, p, A1);
  sm = fma(v * z, p, al + vl);
}
// the octant's constant K (0, pi/2, pi) and sign go into ONE sum with the three parts: K_hi +- ah and +- v by error-free
// additions, everything small added to their errors, a single rounding at the end
CSDO_FN double atan2(const double y, const double x) {
  constexpr double PI_HI = 3.141592653589793, PI_LO = 1.2246467991473532e-16;
  constexpr double PIO2_HI = 1.5707963267948966, PIO2_LO = 6.123233995736766e-17;
  const double ay = fabs(y), ax = fabs(x);
  if (!(ay == ay && ax == ax)) return x + y;
  const bool swap = ay > ax, neg = x < 0.0 || (x == 0.0 && copysign(1.0, x) < 0.0);
  double ah = 0.0, v = 0.0, sm = 0.0;
  if (swap) atan_ratio(ax, ay, ah, v, sm);
  else if (ax != 0.0) atan_ratio(ay, ax, ah, v, sm);
  const double kh = swap ? PIO2_HI : (neg ? PI_HI : 0.0), kl = swap ? PIO2_LO : (neg ? PI_LO : 0.0);
  const double sg = (swap != neg) ? -1.0 : 1.0;
  double h1, l1, h2, l2;
  two_sum(kh, sg * ah, h1, l1);
  two_sum(h1, sg * v, h2, l2);
  const double r = h2 + ((l1 + l2) + fma(sg, sm, kl));
  return copysign(r, y);
}

}  // namespace xm
}  // namespace csdo
