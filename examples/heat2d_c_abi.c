/* heat2d_c_abi.c -- the reference's examples/2D/Diffusion/Heat.jl shape (config 1: 80x80, circle r=1 at (2.01,2.01),
 * Dirichlet(1) interface, Dirichlet(0) on the four borders, T0 = [zeros; ones], BE, dt = 0.25 h^2) driven through
 * include/penguin_hip.h from plain C: no Python, no torch.  This is what the Julia ccall wrapper does.
 *
 *   gcc -std=c11 -O2 -Iinclude examples/heat2d_c_abi.c -Lpenguin/jl_amd/lib -lpenguin_hip \
 *       -Wl,-rpath,$PWD/penguin/jl_amd/lib -o heat2d_c_abi && ./heat2d_c_abi [n] [steps]
 *
 * prints: n M steps extremum sum(V*T_omega) T_omega(centre cell) total_iters
 */
#include <stdio.h>
#include <stdlib.h>

#include "penguin_hip.h"

#define CHECK(call)                                                     \
  do {                                                                  \
    if ((call) != 0) {                                                  \
      char msg[1024];                                                   \
      pg_last_error(msg, sizeof msg);                                   \
      fprintf(stderr, "%s failed: %s\n", #call, msg);                   \
      return 1;                                                         \
    }                                                                   \
  } while (0)

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 80;
  const int64_t steps = argc > 2 ? atoll(argv[2]) : 10;
  const int64_t nn[2] = {n, n};
  const double L[2] = {4.0, 4.0}, x0[2] = {0.0, 0.0};
  const int64_t M = (n + 1) * (n + 1);
  CHECK(pg_init(0));

  pg_mesh* mesh;
  CHECK(pg_mesh_create(2, nn, L, x0, &mesh));
  const double ball[3] = {2.01, 2.01, 1.0};
  pg_capacity* cap;
  CHECK(pg_capacity_create_levelset(mesh, PG_BODY_BALL, ball, 3, 0, &cap));
  pg_diffops* op;
  CHECK(pg_diffops_create(cap, &op));

  pg_bc_desc bc = {PG_BC_DIRICHLET, 0.0, 0.0, 1.0, NULL};
  pg_border_desc borders[4] = {{PG_KEY_LEFT, PG_BC_DIRICHLET, 0.0}, {PG_KEY_RIGHT, PG_BC_DIRICHLET, 0.0},
                               {PG_KEY_TOP, PG_BC_DIRICHLET, 0.0}, {PG_KEY_BOTTOM, PG_BC_DIRICHLET, 0.0}};
  double* T0 = calloc((size_t)(2 * M), sizeof(double));
  for (int64_t i = M; i < 2 * M; ++i) T0[i] = 1.0;
  const double h = 4.0 / (double)n, dt = 0.25 * h * h;
  pg_solver* s;
  CHECK(pg_solver_create_unsteady_mono(cap, op, &bc, borders, 4, NULL, NULL, dt, T0, PG_SCHEME_BE, &s));

  pg_krylov_opts opts = {PG_METHOD_BICGSTAB, 1e-13, 0.0, 0, 4, 1};
  pg_run_info run;
  /* solve_DiffusionUnsteadyMono!: first solve with the constructor's system, then `steps` loop iterations */
  CHECK(pg_solver_run(s, 1e30, PG_SCHEME_BE, &opts, 1, steps, 0, &run));

  double* x = malloc((size_t)(2 * M) * sizeof(double));
  double* V = malloc((size_t)M * sizeof(double));
  CHECK(pg_solver_get_state(s, -1, x, 2 * M));
  CHECK(pg_capacity_get(cap, PG_CAP_V, 0, V, M));
  double heat = 0.0;
  for (int64_t i = 0; i < M; ++i) heat += V[i] * x[i];
  const int64_t centre = (n / 2) * (n + 1) + n / 2;
  printf("%lld %lld %lld %.17g %.17g %.17g %lld\n", (long long)n, (long long)M, (long long)run.steps, run.extremum, heat,
         x[centre], (long long)run.total_iters);

  free(x); free(V); free(T0);
  CHECK(pg_solver_destroy(s));
  CHECK(pg_diffops_destroy(op));
  CHECK(pg_capacity_destroy(cap));
  CHECK(pg_mesh_destroy(mesh));
  CHECK(pg_finalize());
  return 0;
}
