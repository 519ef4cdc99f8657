/*
 * A row-partitioned solve driven from plain C (C99) with NO Python anywhere: what a C / C++ StormRuler driver does to
 * reach storm_hip_op_set_halo (SURVEY.md 8e "Partitioning").
 *
 *   abi_two_ranks <tetgen prefix> <dim> [n_ranks = 2]
 *
 * The process forks its rank processes BEFORE anything touches the GPU; the ranks are joined by pipes.  Every rank:
 *   storm_hip_mesh_read_tetgen        the mesh files -> face graph              (Mallard/IoTetgen.hpp:44-235)
 *   storm_hip_order_cells, storm_hip_mesh_permute_cells   the permute hook: cells renumbered from their centres
 *   storm_hip_partition_rcb           cell -> rank map (every rank computes the same one)
 *   storm_hip_mesh_partition          its owned + halo cells, its halo plan
 *   storm_hip_ctx_comm_init_host      the host-staged transport over the pipes (all-reduce, halo exchange)
 *   storm_hip_op_create_from_mesh_object, storm_hip_solve_cg   -L x = 1, the reference's default knobs (Solver.hpp:66-72)
 * Rank 0 then gathers the solution by global cell id, solves the same problem on one rank (a second context without a
 * communicator) and prints one JSON line with both iteration counts and the difference of the two solutions.
 * Exit status 0 iff both solves converged and agree (iterations +-2 %, solution to 1e-8).
 *
 *   gcc -std=c99 -Iinclude tests/c/abi_two_ranks.c -Lstormruler_amd -lstorm_hip -lm -o abi_two_ranks
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <storm_hip.h>

#define CHECK(call)                                                                         \
  do {                                                                                      \
    int st_ = (call);                                                                       \
    if (st_ != STORM_HIP_OK) {                                                              \
      fprintf(stderr, "[rank %d] %s -> %d: %s\n", g_rank, #call, st_, storm_hip_last_error()); \
      return 2;                                                                             \
    }                                                                                       \
  } while (0)

enum { MAX_RANKS = 6 };
static int g_rank = 0, g_ranks = 2;
static int g_rd[MAX_RANKS], g_wr[MAX_RANKS]; /* pipe ends towards every other rank */

static int put(int fd, const void *p, size_t n) {
  const char *c = (const char *)p;
  while (n > 0) {
    ssize_t k = write(fd, c, n);
    if (k <= 0) return 1;
    c += k, n -= (size_t)k;
  }
  return 0;
}
static int get(int fd, void *p, size_t n) {
  char *c = (char *)p;
  while (n > 0) {
    ssize_t k = read(fd, c, n);
    if (k <= 0) return 1;
    c += k, n -= (size_t)k;
  }
  return 0;
}
/* both directions of one pair without a deadlock: the lower rank writes first */
static int swap_with(int peer, const void *out, size_t n_out, void *in, size_t n_in) {
  if (g_rank < peer) return put(g_wr[peer], out, n_out) || get(g_rd[peer], in, n_in);
  return get(g_rd[peer], in, n_in) || put(g_wr[peer], out, n_out);
}

/* storm_hip_allreduce_fn: in-place sum over all ranks, summed in rank order on every rank (the same bits everywhere) */
static int allreduce(void *user, double *buf, int count) {
  double all[MAX_RANKS][64];
  (void)user;
  if (count > 64) return 1;
  memcpy(all[g_rank], buf, sizeof(double) * (size_t)count);
  for (int r = 0; r < g_ranks; ++r)
    if (r != g_rank && swap_with(r, buf, sizeof(double) * (size_t)count, all[r], sizeof(double) * (size_t)count)) return 1;
  for (int i = 0; i < count; ++i) {
    double s = 0.0;
    for (int r = 0; r < g_ranks; ++r) s += all[r][i];
    buf[i] = s;
  }
  return 0;
}
/* storm_hip_exchange_fn */
static int exchange(void *user, int n_nbrs, const int32_t *nbr_rank, const int64_t *send_ptr, const double *send,
                    const int64_t *recv_ptr, double *recv) {
  (void)user;
  for (int q = 0; q < n_nbrs; ++q)
    if (swap_with(nbr_rank[q], send + send_ptr[q], sizeof(double) * (size_t)(send_ptr[q + 1] - send_ptr[q]), recv + recv_ptr[q],
                  sizeof(double) * (size_t)(recv_ptr[q + 1] - recv_ptr[q])))
      return 1;
  return 0;
}

static int solve(storm_hip_ctx *ctx, const storm_hip_mesh *mesh, double *x_owned, storm_hip_solver_result *res) {
  storm_hip_mesh_view v;
  storm_hip_op *op = NULL;
  storm_hip_vec *b = NULL, *x = NULL;
  storm_hip_solver_params p;
  CHECK(storm_hip_mesh_get_view(mesh, &v));
  CHECK(storm_hip_op_create_from_mesh_object(ctx, mesh, &op)); /* sets the halo plan too */
  CHECK(storm_hip_vec_create(ctx, v.n_cells, v.n_halo, &b));
  CHECK(storm_hip_vec_create(ctx, v.n_cells, v.n_halo, &x));
  CHECK(storm_hip_fill(b, 1.0));
  storm_hip_solver_params_default(&p);
  CHECK(storm_hip_solve_cg(op, -1.0, 0.0, b, x, &p, res, NULL));
  CHECK(storm_hip_vec_download(x, x_owned, v.n_cells));
  CHECK(storm_hip_vec_destroy(b));
  CHECK(storm_hip_vec_destroy(x));
  CHECK(storm_hip_op_destroy(op));
  return 0;
}

static int rank_main(const char *prefix, int dim) {
  storm_hip_mesh *glob = NULL, *loc = NULL;
  storm_hip_mesh_view gv, lv;
  storm_hip_ctx *ctx = NULL;
  storm_hip_solver_result res, res1;
  CHECK(storm_hip_mesh_read_tetgen(prefix, dim, &glob));
  CHECK(storm_hip_mesh_get_view(glob, &gv));
  { /* the permute hook: the library's ordering from the cell centres (a lattice's own order, else the Z-order curve) */
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(gv.n_cells > 0 ? gv.n_cells : 1));
    CHECK(storm_hip_order_cells(gv.dim, gv.n_cells, gv.center, 0, order, NULL));
    CHECK(storm_hip_mesh_permute_cells(glob, order));
    CHECK(storm_hip_mesh_get_view(glob, &gv)); /* (the arrays moved; global_id now maps a cell to its number in the files) */
    free(order);
  }
  int32_t *part = (int32_t *)malloc(sizeof(int32_t) * (size_t)gv.n_cells);
  CHECK(storm_hip_partition_rcb(gv.dim, gv.n_cells, gv.center, g_ranks, part));
  CHECK(storm_hip_mesh_partition(glob, part, g_ranks, g_rank, &loc));
  CHECK(storm_hip_mesh_get_view(loc, &lv));
  CHECK(storm_hip_ctx_create(0, &ctx)); /* (every rank on device 0: a one-GPU box) */
  CHECK(storm_hip_ctx_comm_init_host(ctx, g_ranks, g_rank, allreduce, exchange, NULL));
  double *xl = (double *)malloc(sizeof(double) * (size_t)(lv.n_cells > 0 ? lv.n_cells : 1));
  if (solve(ctx, loc, xl, &res)) return 2;
  CHECK(storm_hip_ctx_sync(ctx));
  if (g_rank != 0) { /* owned rows and their global ids to rank 0 */
    int64_t n = lv.n_cells;
    if (put(g_wr[0], &n, sizeof n) || put(g_wr[0], lv.global_id, sizeof(int64_t) * (size_t)n) ||
        put(g_wr[0], xl, sizeof(double) * (size_t)n))
      return 3;
    int ok = 0;
    if (get(g_rd[0], &ok, sizeof ok)) return 3; /* nobody tears its context down while others still exchange */
    CHECK(storm_hip_ctx_destroy(ctx));
    return ok ? 0 : 1;
  }
  double *xg = (double *)calloc((size_t)gv.n_cells, sizeof(double)), *x1 = (double *)malloc(sizeof(double) * (size_t)gv.n_cells);
  for (int64_t i = 0; i < lv.n_cells; ++i) xg[lv.global_id[i]] = xl[i];
  for (int r = 1; r < g_ranks; ++r) {
    int64_t n = 0;
    if (get(g_rd[r], &n, sizeof n)) return 3;
    int64_t *gid = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    double *xr = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (get(g_rd[r], gid, sizeof(int64_t) * (size_t)n) || get(g_rd[r], xr, sizeof(double) * (size_t)n)) return 3;
    for (int64_t i = 0; i < n; ++i) xg[gid[i]] = xr[i];
    free(gid), free(xr);
  }
  /* the same problem on one rank */
  storm_hip_ctx *ctx1 = NULL;
  CHECK(storm_hip_ctx_create(0, &ctx1));
  if (solve(ctx1, glob, x1, &res1)) return 2;
  CHECK(storm_hip_ctx_destroy(ctx1));
  double d2 = 0.0, n2 = 0.0;
  /* (both in the files' numbering: the ranks' through their global ids, the one-rank solve through the renumbered mesh's) */
  for (int64_t i = 0; i < gv.n_cells; ++i) {
    const double want = x1[i], got = xg[gv.global_id ? gv.global_id[i] : i];
    d2 += (got - want) * (got - want), n2 += want * want;
  }
  const double rel = sqrt(d2 / n2);
  long tol_it = (long)(0.02 * (double)res1.iterations);
  if (tol_it < 2) tol_it = 2;
  const int ok = res.converged && res1.converged && labs((long)(res.iterations - res1.iterations)) <= tol_it && rel <= 1e-8;
  printf("{\"ranks\": %d, \"cells\": %lld, \"owned_rank0\": %lld, \"halo_rank0\": %lld, \"nbrs_rank0\": %d, "
         "\"iterations\": %lld, \"iterations_one_rank\": %lld, \"converged\": %d, \"relative_error\": %.6e, "
         "\"solution_rel_diff\": %.6e, \"x_norm2\": %.17g}\n",
         g_ranks, (long long)gv.n_cells, (long long)lv.n_cells, (long long)lv.n_halo, (int)lv.n_nbrs, (long long)res.iterations,
         (long long)res1.iterations, (int)(res.converged && res1.converged), res.relative_error, rel, sqrt(n2));
  fflush(stdout);
  for (int r = 1; r < g_ranks; ++r)
    if (put(g_wr[r], &ok, sizeof ok)) return 3;
  CHECK(storm_hip_ctx_destroy(ctx));
  return ok ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <tetgen prefix> <dim> [n_ranks]\n", argv[0]);
    return 2;
  }
  g_ranks = argc > 3 ? atoi(argv[3]) : 2;
  if (g_ranks < 1 || g_ranks > MAX_RANKS) return 2;
  int fd[MAX_RANKS][MAX_RANKS][2]; /* fd[a][b]: a writes, b reads */
  for (int a = 0; a < g_ranks; ++a)
    for (int b = 0; b < g_ranks; ++b)
      if (a != b && pipe(fd[a][b]) != 0) return 2;
  pid_t kids[MAX_RANKS];
  for (int r = 1; r < g_ranks; ++r) {
    kids[r] = fork(); /* before any HIP call */
    if (kids[r] < 0) return 2;
    if (kids[r] == 0) {
      g_rank = r;
      break;
    }
  }
  for (int o = 0; o < g_ranks; ++o)
    if (o != g_rank) g_wr[o] = fd[g_rank][o][1], g_rd[o] = fd[o][g_rank][0];
  const int st = rank_main(argv[1], atoi(argv[2]));
  if (g_rank != 0) _exit(st);
  int worst = st;
  for (int r = 1; r < g_ranks; ++r) {
    int ws = 0;
    waitpid(kids[r], &ws, 0);
    if (!WIFEXITED(ws) || WEXITSTATUS(ws) != 0) worst = worst ? worst : 1;
  }
  return worst;
}
