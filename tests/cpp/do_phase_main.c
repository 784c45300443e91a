/* A plain C99 consumer of include/csdo_dsqp.h: what a maintainer writes in place of csdo.cc:93-159 - front end, then the whole DO phase
 * in ONE call (csdo_do_phase), for a small scenario solved twice in one batch.  tests/test_do_phase.py builds it with gcc -std=c99
 * -pedantic (the header is C) and runs it: exit 0 = solved, both copies identical; exit 3 = no HIP device (the CPU test's case). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/csdo_dsqp.h"

int main(void) {
  csdo_vehicle veh;
  csdo_qp_parm parm;
  csdo_front_end_parm fp;
  csdo_vehicle_default(&veh);
  csdo_qp_parm_default(&veh, &parm);
  csdo_front_end_parm_default(&fp);
  /* three vehicles crossing a 40 x 40 map with two obstacles */
  const double starts[9] = {5, 5, 0, 35, 6, 3.14159, 20, 34, -1.5708};
  const double goals[9] = {34, 30, 0.5, 6, 28, 2.8, 20, 6, -1.5708};
  const double obstacles[6] = {20, 18, 0.8, 12, 25, 0.8};
  csdo_paths paths;
  int rc = csdo_front_end_plan(starts, goals, 3, 40.0, 40.0, obstacles, 2, &veh, &fp, &paths);
  if (rc != CSDO_OK || paths.status != 1) {
    printf("front end: rc %d status %d\n", rc, paths.status);
    return 2;
  }
  const int32_t nt = csdo_do_phase_horizon(paths.path_off, 3, &parm);
  printf("front end ok: %d states, horizon %d\n", (int)paths.path_off[3], (int)nt);
  if (nt < 2) return 2;
  csdo_handle h = NULL;
  rc = csdo_dsqp_create(&h, 0);
  if (rc != CSDO_OK) {
    printf("no device: error %d\n", rc);
    csdo_paths_free(&paths);
    return 3;
  }
  csdo_coarse_world cw[2];
  csdo_result res[2];
  int32_t inter_legal[2] = {-1, -1};
  csdo_do_phase_timing tm;
  int w;
  for (w = 0; w < 2; ++w) {
    memset(&cw[w], 0, sizeof(cw[w]));
    cw[w].states = paths.states;
    cw[w].actions = paths.actions;
    cw[w].path_off = paths.path_off;
    cw[w].goals = goals;
    cw[w].obstacles = obstacles;
    cw[w].Na = 3;
    cw[w].n_obs = 2;
    cw[w].dimx = cw[w].dimy = 40.0;
    memset(&res[w], 0, sizeof(res[w]));
    res[w].solutions = (double*)calloc((size_t)3 * nt * 6, sizeof(double));
    res[w].corridors = (double*)calloc((size_t)3 * nt * 8, sizeof(double));
    res[w].sqp_iters = (int32_t*)calloc(3, sizeof(int32_t));
    res[w].admm_iters = (int32_t*)calloc(3, sizeof(int32_t));
    res[w].last_status = (int32_t*)calloc(3, sizeof(int32_t));
  }
  rc = csdo_do_phase(h, cw, 2, &veh, &parm, res, inter_legal, &tm);
  if (rc != CSDO_OK) {
    printf("csdo_do_phase: error %d\n", rc);
    return 1;
  }
  const int same = memcmp(res[0].solutions, res[1].solutions, sizeof(double) * 3 * nt * 6) == 0 &&
                   memcmp(res[0].admm_iters, res[1].admm_iters, sizeof(int32_t) * 3) == 0;
  const double* last = res[0].solutions + ((size_t)0 * nt + (nt - 1)) * 6;
  printf("do phase: status %d, inter-legal %d, admm iterations %d %d %d, vehicle 0 ends at (%.3f, %.3f), %.2f ms, copies identical %d\n",
         (int)res[0].solver_status, (int)inter_legal[0], (int)res[0].admm_iters[0], (int)res[0].admm_iters[1], (int)res[0].admm_iters[2],
         last[0], last[1], tm.total * 1e3, same);
  for (w = 0; w < 2; ++w) {
    free(res[w].solutions); free(res[w].corridors); free(res[w].sqp_iters); free(res[w].admm_iters); free(res[w].last_status);
  }
  csdo_dsqp_destroy(h);
  csdo_paths_free(&paths);
  /* the goal is pinned by the configuration rows: the last state is the goal */
  return (same && res[0].solver_status != 0 && last[0] > 33.9 && last[0] < 34.1) ? 0 : 1;
}
