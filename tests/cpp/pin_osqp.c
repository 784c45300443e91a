/* pin_osqp.c - the OSQP pin kit against the real library, in C: exactly the call sequence of sqp/dsqp_solver.cc:457-502 (osqp_setup,
 * osqp_warm_start_x, osqp_solve with default settings, max_iter 400, verbose off) on the QPs of tests/golden/osqp_pin, true -inf lower
 * bounds on the inter-vehicle rows included (sqp/dsqp_solver.cc:1121-1123).  NOT built by this repository (no osqp.h in its
 * containers; the oracle is "parity unpinned" until someone runs this or scripts/pin_against_osqp.py):
 *
 *   python scripts/pin_against_osqp.py --export /tmp/osqp_pin            # qp_XX.bin: the npz files as flat little-endian binaries
 *   gcc -O2 -I<osqp>/include tests/cpp/pin_osqp.c -L<osqp>/build/out -losqp -lm -o pin_osqp     # OSQP 0.6.3, as the reference's CMakeLists.txt:9,82
 *   ./pin_osqp /tmp/osqp_pin/qp_*.bin                                    # exit code 0: every QP agrees with the oracle
 *
 * File layout: int32 n, m, nnzP, nnzA, max_iter, adaptive_rho_interval, oracle_iter, oracle_status; then P_indptr[n + 1],
 * P_indices[nnzP] (int32), P_data[nnzP] (double), A_indptr[n + 1], A_indices[nnzA], A_data[nnzA], q[n], l[m], u[m], x_warm[n],
 * oracle_x[n].  adaptive_rho_interval = 25: upstream's default 0 derives the interval from wall-clock timing. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "osqp.h"

static int read_i32(FILE* f, int32_t* p, size_t k) { return fread(p, sizeof(int32_t), k, f) == k; }
static int read_f64(FILE* f, double* p, size_t k) { return fread(p, sizeof(double), k, f) == k; }
static c_int* widen(const int32_t* a, size_t k) {
  c_int* o = (c_int*)malloc(sizeof(c_int) * (k ? k : 1));
  for (size_t i = 0; i < k; ++i) o[i] = (c_int)a[i];
  return o;
}

static int one(const char* path, double tol) {
  FILE* f = fopen(path, "rb");
  int32_t h[8];
  if (!f || !read_i32(f, h, 8)) { fprintf(stderr, "%s: cannot read\n", path); return 1; }
  const size_t n = (size_t)h[0], m = (size_t)h[1], nzp = (size_t)h[2], nza = (size_t)h[3];
  int32_t *Pp = malloc(4 * (n + 1)), *Pi = malloc(4 * (nzp + 1)), *Ap = malloc(4 * (n + 1)), *Ai = malloc(4 * (nza + 1));
  double *Px = malloc(8 * (nzp + 1)), *Ax = malloc(8 * (nza + 1)), *q = malloc(8 * n), *l = malloc(8 * m), *u = malloc(8 * m),
         *xw = malloc(8 * n), *xo = malloc(8 * n);
  int ok = read_i32(f, Pp, n + 1) && read_i32(f, Pi, nzp) && read_f64(f, Px, nzp) && read_i32(f, Ap, n + 1) && read_i32(f, Ai, nza) &&
           read_f64(f, Ax, nza) && read_f64(f, q, n) && read_f64(f, l, m) && read_f64(f, u, m) && read_f64(f, xw, n) && read_f64(f, xo, n);
  fclose(f);
  if (!ok) { fprintf(stderr, "%s: truncated\n", path); return 1; }
  OSQPData* data = (OSQPData*)c_malloc(sizeof(OSQPData));
  OSQPSettings* settings = (OSQPSettings*)c_malloc(sizeof(OSQPSettings));
  OSQPWorkspace* work = OSQP_NULL;
  data->n = (c_int)n;
  data->m = (c_int)m;
  data->P = csc_matrix(data->n, data->n, (c_int)nzp, Px, widen(Pi, nzp), widen(Pp, n + 1));
  data->q = q;
  data->A = csc_matrix(data->m, data->n, (c_int)nza, Ax, widen(Ai, nza), widen(Ap, n + 1));
  data->l = l;
  data->u = u;
  osqp_set_default_settings(settings);             /* sqp/dsqp_solver.cc:480-487 */
  settings->max_iter = h[4];
  settings->verbose = 0;
  settings->adaptive_rho_interval = h[5];          /* the one setting the reference leaves at 0 (timing-derived) */
  if (osqp_setup(&work, data, settings) != 0) { fprintf(stderr, "%s: osqp_setup failed\n", path); return 1; }
  osqp_warm_start_x(work, xw);
  osqp_solve(work);
  double dx = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double d = fabs(work->solution->x[i] - xo[i]);
    if (!(d <= dx)) dx = d;                        /* (NaN sticks) */
  }
  const int same = (int)work->info->iter == h[6] && (int)work->info->status_val == h[7] && dx <= tol;
  printf("%-40s iter %4d (oracle %4d)  status %2d (%2d)  max|dx| %.2e  rho %.6g  %s\n", path, (int)work->info->iter, h[6],
         (int)work->info->status_val, h[7], dx, (double)work->settings->rho, same ? "ok" : "DIFFERENT");
  osqp_cleanup(work);
  return same ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: pin_osqp qp_XX.bin ...\n"); return 2; }
  int bad = 0;
  for (int k = 1; k < argc; ++k) bad += one(argv[k], 1e-6);
  printf("%d of %d QPs agree with the oracle\n", argc - 1 - bad, argc - 1);
  return bad ? 1 : 0;
}
