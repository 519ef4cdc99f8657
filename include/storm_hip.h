/*
 * storm_hip.h -- C ABI of the MI355X-native Krylov backend for StormRuler.
 *
 * This is the drop-in boundary: plain pointers and sizes, opaque
 * handles, `int` status returns, no exceptions, no torch / C++ types.  Every
 * entry point names the reference interface it stands in for (paths relative
 * to the reference root).  The C++ header `storm_hip/Storm.hpp` maps the
 * reference's `Storm::Operator / Vector / Solver` template interface onto
 * these calls; INTEGRATION.md shows the binding a StormRuler maintainer adds.
 *
 * Conventions
 *  - Status: 0 = ok; negative = error (STORM_HIP_E_*); text of the last error
 *    of the calling thread via storm_hip_last_error().
 *  - Ownership: the caller owns every handle it creates and destroys it; the
 *    library copies host arrays passed to *_create_* (the caller may free them
 *    on return) and never frees caller memory.
 *  - Threading: as the reference (single-threaded, stateful solver objects,
 *    Solvers/Solver.hpp:66-76) -- one host thread per context; a context owns
 *    its HIP streams; calls on one context are not re-entrant.
 *  - Multi-GPU: one process (context) per GPU.  Vectors hold `n_owned` rows
 *    followed by `n_halo` ghost rows; reductions are summed over all ranks.
 *  - All arithmetic is fp64 (`real_t = double`, Crow/Base/Types.hpp:38);
 *    indices cross the ABI as int64 (the reference's Index wraps size_t,
 *    Utils/Index.hpp:41) and are narrowed to int32 on the device after a
 *    range check.
 */
#ifndef STORM_HIP_H_
#define STORM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* every declaration below is an exported symbol of libstorm_hip.so */
#pragma GCC visibility push(default)

#define STORM_HIP_ABI_VERSION 6

enum {
  STORM_HIP_OK = 0,
  STORM_HIP_E_INVALID = -1,   /* bad argument (null handle, size mismatch, index out of range) */
  STORM_HIP_E_HIP = -2,       /* a HIP runtime call failed */
  STORM_HIP_E_NO_DEVICE = -3, /* no usable gfx950 device */
  STORM_HIP_E_COMM = -4,      /* RCCL failure / communicator misuse */
  STORM_HIP_E_ALLOC = -5,
  STORM_HIP_E_UNSUPPORTED = -6
};

typedef struct storm_hip_ctx storm_hip_ctx;
typedef struct storm_hip_vec storm_hip_vec;
typedef struct storm_hip_op storm_hip_op;

int storm_hip_abi_version(void);
const char *storm_hip_last_error(void);

/* ---- context -------------------------------------------------------------
 * Nothing in the reference (it has no device, SURVEY.md headline fact 1). */
int storm_hip_ctx_create(int device_id, storm_hip_ctx **out);
int storm_hip_ctx_destroy(storm_hip_ctx *ctx);
int storm_hip_ctx_sync(storm_hip_ctx *ctx);
/* name: >= 64 bytes. */
int storm_hip_ctx_info(storm_hip_ctx *ctx, char *name, int name_len, int *num_cus,
                       int64_t *total_mem_bytes);

/* HIP-event stopwatch on the context's compute stream (the stream every
 * kernel of this library is launched on). */
int storm_hip_timer_start(storm_hip_ctx *ctx);
int storm_hip_timer_stop(storm_hip_ctx *ctx, float *elapsed_ms);

/* With option "profile_spmv" = 1 every launch of the dominant kernel (the sliced-ELL SpMV)
 * is bracketed by a HIP-event pair on the compute stream; this returns and resets the
 * accumulated launch count and durations (synchronises the stream). */
int storm_hip_ctx_get_spmv_profile(storm_hip_ctx *ctx, int64_t *launches, double *total_ms,
                                   double *min_ms);
/* ... the same launches one by one (milliseconds each, in launch order; at most `capacity` are written, *count is
 * how many there were); resets the profile like the call above. */
int storm_hip_ctx_get_spmv_profile_samples(storm_hip_ctx *ctx, double *ms_out, int64_t capacity, int64_t *count);

/* ---- communicator (RCCL over xGMI) -------------------------
 * rank 0 fills a 128-byte id, the host distributes it (torch.distributed /
 * MPI / a file), every rank calls comm_init.  n_ranks == 1 needs neither. */
int storm_hip_comm_unique_id(void *id128);
int storm_hip_ctx_comm_init(storm_hip_ctx *ctx, const void *id128, int n_ranks, int rank);
int storm_hip_ctx_comm_size(storm_hip_ctx *ctx, int *n_ranks, int *rank);
/* What RCCL ITSELF says about this context's communicators (not what the host passed to comm_init): ncclCommCount /
 * ncclCommUserRank / ncclCommCuDevice of the halo communicator, ncclCommCount of the reduction communicator, the HIP
 * device of the context and its PCI bus id (>= 16 bytes).  All counts are 0 on a context without an RCCL communicator
 * (one rank, or the host-staged / peer-window transports).  Fails with STORM_HIP_E_COMM when RCCL reports an asynchronous
 * error.  For a run's record: a reader can check that RCCL saw N ranks on N distinct devices.  Nothing in the reference
 * corresponds to it (single process). */
int storm_hip_ctx_comm_rccl_view(storm_hip_ctx *ctx, int *halo_count, int *halo_user_rank, int *halo_device, int *red_count,
                                 int *hip_device, char *pci_bus_id, int pci_bus_id_len);
/* Host-staged transport: the same multi-rank protocol (halo planes before the boundary rows of an SpMV,
 * global sums behind every reduction) with the bytes moved by the host program's own messaging layer
 * instead of RCCL -- for hosts that already run MPI / gloo, and for putting several ranks on ONE device.
 *   allreduce(user, buf, count): in-place sum of `count` doubles over all ranks (host memory);
 *   exchange(user, n_nbrs, nbr_rank, send_ptr, send, recv_ptr, recv): deliver send[send_ptr[q]..send_ptr[q+1])
 *     to rank nbr_rank[q] and fill recv[recv_ptr[q]..recv_ptr[q+1]) with what that rank sent here.
 * Both return 0 on success.  Synchronous (no overlap with the interior rows).  Nothing in the reference
 * corresponds to it (single process); SURVEY.md 8e. */
typedef int (*storm_hip_allreduce_fn)(void *user, double *buf, int count);
typedef int (*storm_hip_exchange_fn)(void *user, int n_nbrs, const int32_t *nbr_rank, const int64_t *send_ptr,
                                     const double *send, const int64_t *recv_ptr, double *recv);
int storm_hip_ctx_comm_init_host(storm_hip_ctx *ctx, int n_ranks, int rank, storm_hip_allreduce_fn allreduce,
                                 storm_hip_exchange_fn exchange, void *user);

/* Peer-window transport: every rank owns a window of device memory that all ranks map through HIP IPC; halo planes
 * and reduction scalars are written straight into the RECEIVER's window by the sender's kernels (over xGMI between
 * GPUs) and picked up by polling -- a one-shot all-reduce of up to 64 doubles summed in rank order (the same bits
 * on every rank; SURVEY.md 8e "Determinism") and direct halo writes with flag + acknowledgement, in place of
 * RCCL's latency-bound small collectives.  Also works between processes that share ONE device.
 *   1. every rank: storm_hip_ctx_comm_ipc_export(ctx, n_ranks, rank, window_bytes (0 = 16 MiB), handle64)
 *   2. the host program all-gathers the 64-byte handles (rank order)
 *   3. every rank: storm_hip_ctx_comm_init_ipc(ctx, handles)
 * A (sender, receiver) pair may move up to (window_bytes / (2 n_ranks)) bytes per exchange.  Waits are bounded (5 s);
 * a timeout surfaces as STORM_HIP_E_COMM from the next call.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts whose
 * driver only supports dmabuf IPC.  Nothing in the reference corresponds to it (single process). */
int storm_hip_ctx_comm_ipc_export(storm_hip_ctx *ctx, int n_ranks, int rank, int64_t window_bytes, void *handle64);
int storm_hip_ctx_comm_init_ipc(storm_hip_ctx *ctx, const void *handles);

/* ---- vectors -------------------------------------------------------------
 * `Feathers::Field` as the solver `Vector` (Feathers/Field.hpp:60-114):
 * contiguous doubles, one per cell.  create == `assign(other, false)`, which
 * value-initialises (zero-fills) a new field (Field.hpp:82-84). */
int storm_hip_vec_create(storm_hip_ctx *ctx, int64_t n_owned, int64_t n_halo, storm_hip_vec **out);
int storm_hip_vec_create_like(const storm_hip_vec *other, storm_hip_vec **out);
int storm_hip_vec_destroy(storm_hip_vec *v);
int storm_hip_vec_size(const storm_hip_vec *v, int64_t *n_owned, int64_t *n_halo);
int storm_hip_vec_upload(storm_hip_vec *v, const double *host, int64_t n);
int storm_hip_vec_download(const storm_hip_vec *v, double *host, int64_t n);
/* Raw device pointer (n_owned + n_halo doubles), for zero-copy interop.  The address stays valid until the vector is
 * destroyed (a vector that has given its address out is never one whose storage option lazy_statements = 2 exchanges). */
int storm_hip_vec_device_ptr(storm_hip_vec *v, void **dev_ptr);
/* The context a vector lives on. */
int storm_hip_vec_context(const storm_hip_vec *v, storm_hip_ctx **ctx);
/* One element (waits for the stream; a debugging / concept-check accessor, never used by the solvers):
 * `Field::operator()(row, col)`, Feathers/Field.hpp:104-111. */
int storm_hip_vec_get(const storm_hip_vec *v, int64_t row, double *value);

/* ---- BLAS-1 --------------------------------------------------------------
 * One call per Bittern expression statement the solver bodies execute
 * (overload census); each is one kernel over the owned rows.
 * Reference loops: Bittern/MatrixAlgorithms.hpp:58-81 (matrix_for_each),
 * :162-205 (reduce).  Reductions return the sum over all ranks. */
int storm_hip_fill(storm_hip_vec *y, double value);                        /* fill_with(y, v)   Solver.hpp:281 */
int storm_hip_copy(storm_hip_vec *y, const storm_hip_vec *x);              /* y <<= x           MatrixAlgorithms.hpp:120-124 */
int storm_hip_scale(storm_hip_vec *y, double s);                           /* y *= s            MatrixTarget.hpp:96-99 */
int storm_hip_div_scalar(storm_hip_vec *y, double s);                      /* y /= s            MatrixTarget.hpp:101-105, SolverGmres.hpp:88 */
int storm_hip_axpy(storm_hip_vec *y, double a, const storm_hip_vec *x);    /* y += a*x (a<0: y -= |a|*x)  SolverCg.hpp:98-99 */
int storm_hip_xpay(storm_hip_vec *y, const storm_hip_vec *x, double b);    /* y <<= x + b*y     SolverCg.hpp:123 */
/* y <<= a*x + b*z  (covers r <<= b - r, Operator.hpp:98; y may alias x or z) */
int storm_hip_axpbz(storm_hip_vec *y, double a, const storm_hip_vec *x, double b, const storm_hip_vec *z);
/* p <<= r + beta*(p - omega*v)   SolverBiCgStab.hpp:119 */
int storm_hip_bicgstab_p(storm_hip_vec *p, const storm_hip_vec *r, double beta, double omega,
                         const storm_hip_vec *v);
/* y <<= r + s*(a*x + b*z): the nested three-term updates -- BiCGStab's p (above) and CGS's
 * `p <<= u + beta*(q + beta*p)`, SolverCgs.hpp:122.  y may alias any operand. */
int storm_hip_lin3(storm_hip_vec *y, const storm_hip_vec *r, double s, double a, const storm_hip_vec *x,
                   double b, const storm_hip_vec *z);
/* y += s * (a .* b), elementwise.  Not in the solver census: the nonlinear term a time-step
 * driver forms between solves (cf. `f <<= map(dF_dc, c)`, Playground.cpp:148). */
int storm_hip_vmul_add(storm_hip_vec *y, double s, const storm_hip_vec *a, const storm_hip_vec *b);
/* `out <<= map(func, mats...)`  Bittern/MatrixMath.hpp:44-105 (MapMatrixView, map) evaluated by the element loop of
 * MatrixAlgorithms.hpp:75-79, 120-124 -- the playground's `f <<= map(dF_dc, c)`, Playground.cpp:142-148.  `func` arrives as
 * a program in reverse Polish notation over the elements x0[i], x1[i], y[i] (the old value of the target) and up to 16
 * constants: program[k] = opcode | (constant index << 8), at most 48 operations, at most 8 operands at once; it must leave
 * exactly one value.  Only exactly-rounded operations (+ - * / sqrt abs neg min max), each rounded on its own, no
 * contraction: y[i] is what the host's scalar evaluation of the same expression in the same order gives, BIT FOR BIT.
 * x0 / x1 may be NULL when the program does not read them and may alias y.  include/storm_hip/Storm.hpp builds the
 * program by tracing a generic lambda: `f <<= map([](auto c) { return 2.0 * c * (c - 1.0) * (2.0 * c - 1.0); }, c)`. */
enum {
  STORM_HIP_MAP_X0 = 0, STORM_HIP_MAP_X1 = 1, STORM_HIP_MAP_Y = 2, STORM_HIP_MAP_CONST = 3, /* push an operand */
  STORM_HIP_MAP_NEG = 8, STORM_HIP_MAP_ABS = 9, STORM_HIP_MAP_SQRT = 10,                    /* top = op(top) */
  STORM_HIP_MAP_ADD = 16, STORM_HIP_MAP_SUB = 17, STORM_HIP_MAP_MUL = 18, STORM_HIP_MAP_DIV = 19, /* second OP top */
  STORM_HIP_MAP_MIN = 20, STORM_HIP_MAP_MAX = 21
};
int storm_hip_map(storm_hip_vec *y, const storm_hip_vec *x0, const storm_hip_vec *x1, const int32_t *program, int n_ops,
                  const double *constants, int n_constants);
/* y = a .* b, elementwise: `Preconditioner::mul` of a diagonal (Jacobi) preconditioner behind the
 * pre_op hook of Solvers/Solver.hpp:74-75 (the reference ships only IdentityPreconditioner,
 * Preconditioner.hpp:84-97).  y may alias a or b. */
int storm_hip_vmul(storm_hip_vec *y, const storm_hip_vec *a, const storm_hip_vec *b);
/* y = (s * a) ./ b, elementwise; a == NULL: y = s ./ b.  The quotient nodes of the reference's expression
 * templates, `scalar / mat` and `mat1 / mat2` (Bittern/MatrixMath.hpp:261-265, :298-302; known answers
 * tests/unit/BitternMath.cpp:160-171).  Not in the solver census.  y may alias a or b. */
int storm_hip_vdiv(storm_hip_vec *y, double s, const storm_hip_vec *a, const storm_hip_vec *b);
/* fill_randomly(y)  Bittern/MatrixAlgorithms.hpp:140-153: uniform [0, 1) numbers from a
 * function-static std::mt19937_64{} (default seed, state persists across calls), drawn
 * sequentially on the host in row order and uploaded -- the same engine, distribution and
 * standard library the reference uses, hence the same numbers.  Single rank only.
 * storm_hip_rng_reset() restarts the engine (what a fresh process is to the reference). */
int storm_hip_fill_randomly(storm_hip_vec *y);
void storm_hip_rng_reset(void);
/* dot_product(a, b)  MatrixAlgorithms.hpp:310-317;  norm_2(a)  :262-270 */
int storm_hip_dot(const storm_hip_vec *a, const storm_hip_vec *b, double *result);
int storm_hip_norm2(const storm_hip_vec *a, double *result);
/* out[i] = <a, bs[i]>, i < k: the Arnoldi multi-dot (SolverGmres.hpp:157-160 batched). */
int storm_hip_multi_dot(const storm_hip_vec *a, const storm_hip_vec *const *bs, int k, double *out);
/* The same reduction in two halves: _begin enqueues it and returns a request, _end waits for the
 * sums (in pinned host memory, written by the kernel's last block) -- up to 8 requests in flight,
 * ended in any order; the caller's host work, or further launches, overlap the ~9 us a synchronous
 * dot spends between the end of its kernel and the start of the next one.  storm_hip_multi_dot is
 * _begin + _end.  (With a communicator, or k > 8, _begin computes the sums and _end returns them.) */
int storm_hip_multi_dot_begin(const storm_hip_vec *a, const storm_hip_vec *const *bs, int k, int *request);
int storm_hip_multi_dot_end(storm_hip_ctx *ctx, int request, double *out);
/* y += sum_i coefs[i] * xs[i]  (SolverGmres.hpp:233-236 batched) */
int storm_hip_multi_axpy(storm_hip_vec *y, const double *coefs, const storm_hip_vec *const *xs, int k);

/* ---- operators -----------------------------------------------------------
 * The matrix-free FVM stencil `stormDivGrad` (source_apps/playground/
 * Playground.cpp:115-131) precomputed into gather form.  Applying an operator
 * computes   y = beta * x + alpha * M(x)   on the owned rows, where for the
 * face-graph constructors
 *     M(x)_i = sum_{faces f of i} w_if * (x_other(f) - x_i) + diag_extra_i * x_i .
 * `alpha = -1, beta = 0` is the Poisson operator -L; `alpha = -kappa, beta = 1`
 * the Helmholtz operator of the playground lambda (Playground.cpp:153-167).
 *
 * Row i's entries are summed in face order, the order in which the
 * reference's face loop accumulates into u[i].
 */

/* Diffusion operator from the mesh quantities the reference reads:
 *   inner/outer[F]   face -> cells (FaceView::inner_cell/outer_cell, Mallard/Mesh.hpp:269-280),
 *                    local ids in [0, n_owned + n_halo);
 *   coef[F]          A_f / |x_outer - x_inner|   (Playground.cpp:126-128);
 *   b_cell/b_coef[B] Dirichlet boundary faces: owning cell and A_b / |x_face - x_cell|
 *                    (ghost value 0 at the face; loop shape of Feathers/ConvectionScheme.hpp:95-106);
 *   volume[n_owned + n_halo]  cell volumes (CellView::volume, Mallard/Mesh.hpp:304).
 * w_if = coef_f / volume_i;  diag_extra_i = -sum_b b_coef_b / volume_i. */
int storm_hip_op_create_from_faces(storm_hip_ctx *ctx, int64_t n_owned, int64_t n_halo,
                                   int64_t n_faces, const int64_t *inner, const int64_t *outer,
                                   const double *coef, int64_t n_bfaces, const int64_t *b_cell,
                                   const double *b_coef, const double *volume, storm_hip_op **out);

/* The same operator straight from the mesh arrays the reference's face loop reads (Playground.cpp:119-129): the
 * library forms A_f / |x_outer - x_inner| itself -- squares summed left to right like `length`
 * (MatrixAlgorithms.hpp:262-270, 303-305), threaded -- instead of taking it as `coef`.
 *   area[F], center[(n_owned + n_halo) * dim] (row-major), b_area[B], b_center[B * dim]; dim in 1..3.
 * Bit-identical to storm_hip_op_create_from_faces with coefficients computed that way on the host. */
int storm_hip_op_create_from_mesh(storm_hip_ctx *ctx, int64_t n_owned, int64_t n_halo, int32_t dim, int64_t n_faces,
                                  const int64_t *inner, const int64_t *outer, const double *area, const double *center,
                                  int64_t n_bfaces, const int64_t *b_cell, const double *b_area, const double *b_center,
                                  const double *volume, storm_hip_op **out);

/* General (non-symmetric) face-graph operator: row inner[f] gets weight
 * w_inner[f] on (x_outer - x_inner), row outer[f] gets w_outer[f] on
 * (x_inner - x_outer); diag_extra[n_owned] may be NULL (zeros).  This is the
 * form the upwind convection face loop (Feathers/ConvectionScheme.hpp:80-107)
 * lowers to. */
int storm_hip_op_create_from_face_weights(storm_hip_ctx *ctx, int64_t n_owned, int64_t n_halo,
                                          int64_t n_faces, const int64_t *inner,
                                          const int64_t *outer, const double *w_inner,
                                          const double *w_outer, const double *diag_extra,
                                          storm_hip_op **out);

/* Assembled CSR rows (n_rows owned rows, columns in [0, n_rows + n_halo)). */
int storm_hip_op_create_csr(storm_hip_ctx *ctx, int64_t n_rows, int64_t n_halo,
                            const int64_t *row_ptr, const int64_t *col, const double *val,
                            storm_hip_op **out);

/* Options: 40 keys (csrc/context.hip), all with measured defaults.  Refinements that were measured and dropped are not
 * switches any more: their patch is profiles/experiments/r08_pruned_experiments.patch.
 *
 * Build knobs, read when an operator / vector is created:
 *   ell_cap (0 = none): rows longer than this spill their remaining entries to the CSR tail;
 *   spmv_dict (4): the most compact LOSSLESS record format an operator may take -- 0 fp64 weights + int32 columns (what any
 *              mesh gets); 1 byte-indexed weights (<= 256 distinct fp64 bit patterns, <= 7 neighbours per row); 2 + byte-indexed
 *              column offsets; 3 + two consecutive rows share one merged neighbour list and their 16-byte gathers; 4 + one
 *              common offset order for the whole operator (a structured box in natural ordering), +-1 neighbours from the
 *              adjacent lanes.  Every format gives the same bits (tests/test_gpu_formats.py);
 *   spmv_mixed (1): a partitioned operator keeps format 4 for the row groups that read no halo column, format 3 elsewhere;
 *   spmv_spw (0 = automatic): slices per wavefront of the byte-indexed kernel (1, 2, 4);
 *   spmv_canon_tile (2), spmv_canon_tile_min_rows (2^20): format 4 on a lattice -- tiles of 1024 rows x 2 (4) planes with the
 *              in-plane neighbours from LDS and the out-of-plane ones from registers, from this many rows on (0 = plain kernel);
 *   latency_rows (2^19): operators up to this size keep a compact fp64 copy for the one-kernel CG / BiCGStab;
 *   pool_bytes (16 GiB): vector storage released by vec_destroy is kept for the next vec_create of the same size;
 *   vec_arena (1): the vectors of one size are slots of ONE physically contiguous allocation a fixed distance apart.
 * Paths (which loop a solve takes; every path gives the reference's iteration to rounding):
 *   latency_path (1), latency_cache (1): CG / BiCGStab of a small halo-free operator as ONE cooperative kernel per solve
 *              (0: neither this nor the resident path; 2: this path only); operator records held in registers where they fit;
 *   resident_path (1), resident_max_rows (2^22), resident_planes (0 = automatic), resident_early (1): CG / BiCGStab of a
 *              halo-free LATTICE operator as one persistent kernel in which every block owns a box of the lattice
 *              (csrc/resident.hip); surfaces travelling under the all-reduces;
 *   coop_mgs (1), coop_mgs_lds (1), coop_mgs_quad (1): GMRES's Gram-Schmidt chain as one cooperative kernel per Arnoldi step;
 *              its LDS-ring and four-steps-per-synchronisation forms (2 = forced, for tests; 0 = off);
 *   mgs_steps (4): modified-Gram-Schmidt steps per pass over w on the kernel-per-statement path (2, 3, 4);
 *   generic_solvers (0): 1 sends storm_hip_krylov_solve through the engine even where a fused loop exists;
 *   cg_fuse (1), cg_march (8), cg_march_fill (2048): the SpMV launch of a tiled format-4 operator ends the previous CG
 *              iteration (x += alpha p, p = r + beta p) itself, as blocks marching through this many planes (0: tiles), fewer
 *              planes per block on small lattices so that the grid holds about cg_march_fill blocks (0: as given);
 *   fused_reduce (1), lin_fuse (1), ticket_reduce (1): engine reductions finished by the partials kernel's last block; two
 *              consecutive vector statements as one pass; fused-loop reductions finished inside the producing kernels;
 *   nontemporal (1), blas1_nt (1): non-temporal record / y traffic of the SpMV; of the BLAS-1 and solver kernels (0 never,
 *              2 always, 1 for vectors of at least 6 * 2^20 rows: shorter ones survive in the Infinity Cache between kernels);
 *   lazy_statements (0): HOST loops -- storm_hip_copy / _scale / _axpy / _xpay / _axpbz and storm_hip_op_apply are not
 *              launched when called but wait, in program order, for the call that needs their result; two consecutive linear
 *              statements leave as ONE pass, and a storm_hip_dot / _norm2 over a vector the last waiting statement writes
 *              rides in that statement's kernel (`x += alpha p; r -= alpha z; <r, r>`: one kernel; `z = A p; <p, z>`: the
 *              apply with its fused-dot epilogue).  Every other entry point launches what waits first, so nothing is
 *              observed out of order; nothing waits inside a solver's operator / preconditioner callback.  The
 *              linear statements and their reductions give the eager kernels' values bit for bit; the apply's fused dot
 *              sums in the SpMV kernel's order (equal to rounding) (csrc/lazy.hip).  A statement neither the last one
 *              nor the sum depends on keeps waiting (the x-update above, when `<r, r>` is asked for).  Value 2 adds the
 *              library's fused CG step for a host loop: `x += alpha p; ...; p <<= r + beta p; z = A p; <p, z>` on a lattice
 *              operator is ONE launch (x, the new p, z and the sum: 56 B/row, the device loop's kernel, its FMA roundings);
 *              the new p is written to a spare vector whose storage p's handle then takes over -- never for a vector whose
 *              address storm_hip_vec_device_ptr has handed out.  The host loops of Storm.hpp / api.py switch level 2 on
 *              for their duration (IterativeSolver::lazy_statements).
 * RCCL transport:
 *   rccl_fused (1), rccl_ticket (1): the fused CG step on a partitioned lattice operator (the boundary planes of the new
 *              direction packed by a small kernel and sent under the marching launch); local sums finished in the kernels
 *              (CG's <r, r>; BiCGStab's |r|^2 and <rt, r>, whose alpha and omega the update kernels form themselves from
 *              the all-reduced sums -- no scalar-step launch behind those all-reduces);
 *   rccl_early_halo (1): BiCGStab -- the boundary planes of s and of the new direction are formed by a small kernel and sent
 *              before the update kernel that forms the vector runs.  The same bits;
 *   rccl_flag_wait (1): the boundary rows of an apply are released by a flag in device memory (set by a one-thread kernel
 *              behind the send / recv group on the comm stream, polled by a one-thread kernel in front of the boundary
 *              launch; bounded: 10 s, then STORM_HIP_E_COMM from the next call) instead of a cross-stream event, which costs
 *              ~15 us between "exchange done" and "boundary rows start" on this platform (5 us with the flag).  The same bits.
 * Instrumentation: profile_spmv (HIP-event pair around every SpMV launch), profile_comm (RCCL path: device timestamps
 *   around every step of an exchange and every all-reduce), resident_profile (the resident kernels time their phases),
 *   ticket_verify (k > 0: every k-th iteration the fused loops recompute their in-kernel reductions by the two-launch path
 *   and compare on the device).
 * Test hooks (not options: they exist so that tests can force a path or compare a kernel with its plainer form):
 *   coop_force_fail (1: cooperative launches "fail"; 2: cooperative kernels "gave up"), ticket_verify_inject,
 *   test_disable (a bit mask; every bit switches ONE refinement off, the plain form must give the same bits: 1 the chain
 *   kernel's own operator apply, 2 its prefetch under the all-reduce, 4 the resident path's coefficient cache, 8 its halo
 *   interleave, 16 awaited publishing exchanges of the latency path, 32 ordinary launch of the one-kernel paths, 64 downward
 *   marching of odd z-chunks). */
int storm_hip_ctx_set_option(storm_hip_ctx *ctx, const char *key, int64_t value);

/* Which path the solves of this context took so far (no reference counterpart: a diagnostic of this library; the
 * reference logs one line per solve, Solver.hpp:144-145).  Keys: "resident_solves" (csrc/resident.hip),
 * "latency_solves" (csrc/latency.hip: one cooperative kernel per solve), "throughput_solves" (a kernel per statement,
 * fused loops of csrc/solvers.hip), "engine_solves" (csrc/krylov.hip), "cg_fused_steps" (solves whose CG step rode in
 * the SpMV launch), "lazy_fused_dots" / "lazy_fused_pairs" / "lazy_apply_dots" / "lazy_cg_steps" / "lazy_waiting" (option
 * lazy_statements: reductions that rode in a statement's kernel, pairs of statements that left as one pass, applies that
 * left with a fused dot, fused CG steps, statements waiting now).  On the peer-window transport, where the time of the exchanges went (ticks of 10 ns of the device's
 * real-time counter, and counts): "ipc_allreduce_wait_ticks" / "ipc_allreduces" (from a rank's own contribution being
 * stored to every rank's being read), "ipc_ack_wait_ticks" / "ipc_ack_waits" (a send waiting for the receivers to have
 * consumed the plane two exchanges back), "ipc_halo_slow_poll_ticks" / "ipc_halo_slow_polls" (halo values that had not
 * arrived when the boundary rows asked for them; thread-ticks).  On the RCCL transport with option "profile_comm" = 1
 * (one-thread stamp kernels around every step, both streams; setting the option restarts the sums): "rccl_prof_exchanges",
 * "rccl_prof_event_to_comm_ticks" (the compute stream has the vector ready -> the comm stream starts),
 * "rccl_prof_pack_ticks", "rccl_prof_sendrecv_ticks", "rccl_prof_unhidden_wait_ticks" (the exchange still running when the
 * interior rows had ended), "rccl_prof_resume_ticks" (exchange and interior rows done -> the boundary rows start),
 * "rccl_prof_allreduces", "rccl_prof_allreduce_ticks".  With option "resident_profile" = 1:
 * "resident_phase_mean_<k>" / "resident_phase_max_<k>" (csrc/resident.hip). */
int storm_hip_ctx_get_counter(storm_hip_ctx *ctx, const char *key, int64_t *value);

/* A cell ordering from geometry: host preprocessing in the role METIS plays in the north star; the reference's hook
 * is UnstructuredMesh::permute (Mallard/MeshUnstructured.hpp:443-459, 557-612: entities renumbered, adjacency rows keep
 * their order).  centers: [n_cells][dim] cell centres (CellView::center, Mallard/Mesh.hpp:304-311).  order_out[i] = the
 * cell that becomes cell i.  mode 0: the lexicographic order of a LATTICE where the centres form a tensor-product
 * grid (a renumbered structured mesh gets its natural order -- and the lattice record formats -- back), else the
 * Z-order (Morton) curve of the centres; 1: Morton always; 2: lattice or an error.  *kind_out (nullable): 1 lattice,
 * 2 Morton.  No device is touched. */
int storm_hip_order_cells(int32_t dim, int64_t n_cells, const double *centers, int32_t mode, int64_t *order_out,
                          int32_t *kind_out);

/* ---- host-side meshes: the data either side of the path (SURVEY.md 8f row 4, 8e "Partitioning") -----------------
 * A storm_hip_mesh is a face graph in host memory -- exactly the arrays stormDivGrad's face loop reads
 * (Playground.cpp:119-129: face -> inner / outer cell, face area, cell centre, cell volume) plus the boundary faces,
 * and, for a rank's part of a partitioned mesh, its halo cells and halo plan.  No device is touched by any of these
 * calls; all are threaded (STORM_HIP_BUILD_THREADS).  stormruler_amd/io_tetgen.py and partition.py are the numpy
 * restatements they are checked against array for array. */
typedef struct storm_hip_mesh storm_hip_mesh;
typedef struct storm_hip_mesh_view {
  int32_t dim, n_nbrs;
  int64_t n_cells, n_halo, n_faces, n_bfaces;
  const int64_t *inner, *outer;   /* [n_faces] local cell ids in [0, n_cells + n_halo)             Mesh.hpp:269-280 */
  const double *area;             /* [n_faces]                                                      Mesh.hpp:254 */
  const double *center, *volume;  /* [(n_cells + n_halo) * dim], [n_cells + n_halo]                 Mesh.hpp:304-311 */
  const int64_t *b_cell;          /* [n_bfaces] owning cell of a labelled (boundary) face */
  const double *b_area, *b_center;/* [n_bfaces], [n_bfaces * dim] */
  const int64_t *global_id;       /* [n_cells + n_halo] or NULL (an unpermuted single-rank mesh: the identity) */
  const int32_t *halo_owner;      /* [n_halo] or NULL */
  const int32_t *nbr_rank;        /* halo plan, as storm_hip_op_set_halo takes it */
  const int64_t *send_ptr, *send_idx, *recv_ptr;
} storm_hip_mesh_view;

/* `read_mesh_from_tetgen(mesh, path)`  Mallard/IoTetgen.hpp:44-235, both branches: <prefix>.node / .edge / [.face] /
 * .ele (prefix ends with ".1." or ".1"; '#' comments; zero-based ids used as written), then the face graph the
 * reference's insert() calls build (MeshUnstructured.hpp:350-425, 509-554: listed sides keep file order and their
 * marker as label, unlisted ones are created by the first cell that owns them in the order of Triangle::edges() /
 * Tetrahedron::faces() with label 0; inner = first owner, outer = second, which must see the side reversed;
 * interior = label 0).  dim = mesh_dim_v<Mesh> the caller expects (2, 3; 0 = what the node file says): a mismatch is the
 * reference's I/O error.  Errors (missing file, short file, bad header, bad topology) return STORM_HIP_E_INVALID with
 * the reference's message (STORM_THROW_IO -> std::runtime_error, Crow/Base/Exception.hpp:35-44).  Geometry:
 * Shape.hpp:155-167, 242-247, 309-321, 598-607; the tetrahedron's volume |det| / 6 is this build's own (the reference
 * has no volume(Tetrahedron): SURVEY.md headline fact 4). */
int storm_hip_mesh_read_tetgen(const char *prefix, int32_t dim, storm_hip_mesh **out);
/* The same face graph from arrays in memory: pos[n_nodes * dim], listed[n_listed * dim] (+ listed_label[n_listed] or
 * NULL = all 0), cells[n_cells * (dim + 1)]. */
int storm_hip_mesh_from_simplices(int32_t dim, int64_t n_nodes, const double *pos, int64_t n_listed, const int64_t *listed,
                                  const int64_t *listed_label, int64_t n_cells, const int64_t *cells, storm_hip_mesh **out);
/* Writes the four files in the format above (doubles in the shortest form that round-trips; 3-D: an .edge file
 * with no entries).  Test / bench infrastructure: the reference only reads. */
int storm_hip_mesh_write_tetgen(const char *prefix, int32_t dim, int64_t n_nodes, const double *pos, int64_t n_listed,
                                const int64_t *listed, const int64_t *listed_label, int64_t n_cells, const int64_t *cells);
/* A mesh from face-graph arrays the caller already has (copied); global_id / halo_owner may be NULL when n_halo == 0. */
int storm_hip_mesh_create(int32_t dim, int64_t n_cells, int64_t n_halo, int64_t n_faces, const int64_t *inner,
                          const int64_t *outer, const double *area, const double *center, const double *volume,
                          int64_t n_bfaces, const int64_t *b_cell, const double *b_area, const double *b_center,
                          const int64_t *global_id, const int32_t *halo_owner, storm_hip_mesh **out);
/* Pointers into the mesh's own arrays: valid until the mesh is permuted or destroyed. */
int storm_hip_mesh_get_view(const storm_hip_mesh *mesh, storm_hip_mesh_view *view);
/* `UnstructuredMesh::permute`  MeshUnstructured.hpp:443-459 for cells: new cell i is old cell order[i]; faces keep
 * their order and their inner / outer roles; halo cells keep their slot; global_id follows (and is created for a
 * single-rank mesh, so a later partition still knows the original ids). */
int storm_hip_mesh_permute_cells(storm_hip_mesh *mesh, const int64_t *order);
/* Cell -> rank maps (the reference has no partitioner, METIS is not in the image): recursive coordinate
 * bisection -- the longest axis of a part's bounding box cut at the k-th smallest (coordinate, cell id), k
 * proportional to the ranks on either side, any n_parts -- and slabs along one axis (contiguous ranges of the cells
 * ordered by (coordinate, cell id): k whole planes per rank for a structured box of k * n_parts planes). */
int storm_hip_partition_rcb(int32_t dim, int64_t n_cells, const double *centers, int32_t n_parts, int32_t *part_out);
int storm_hip_partition_slabs(int32_t dim, int64_t n_cells, const double *centers, int32_t axis, int32_t n_parts,
                              int32_t *part_out);
/* A rank's part of a single-rank mesh: its owned cells in ascending id, then the halo cells grouped by owner rank,
 * each group in ascending global id; every face that touches an owned cell, in mesh order; the boundary faces of the
 * owned cells; and the halo plan -- towards neighbour q the owned cells that share a face with one of q's cells, in
 * ascending global id (= q's halo group for this rank: no index lists are ever exchanged). */
int storm_hip_mesh_partition(const storm_hip_mesh *global_mesh, const int32_t *part, int32_t n_parts, int32_t rank,
                             storm_hip_mesh **out);
/* (Re)computes the halo plan of a local mesh built with storm_hip_mesh_create from its global ids and halo owners. */
int storm_hip_mesh_halo_plan(storm_hip_mesh *mesh, int32_t rank);
/* storm_hip_op_create_from_mesh on the mesh's arrays + storm_hip_op_set_halo with its halo plan. */
int storm_hip_op_create_from_mesh_object(storm_hip_ctx *ctx, const storm_hip_mesh *mesh, storm_hip_op **out);
int storm_hip_mesh_destroy(storm_hip_mesh *mesh);

/* Halo plan of a row-partitioned operator.  For neighbour q
 * (rank nbr_rank[q]) the owned rows send_idx[send_ptr[q] .. send_ptr[q+1]) are
 * sent, and the halo rows n_owned + [recv_ptr[q] .. recv_ptr[q+1]) received;
 * both sides order a segment by global cell id. */
int storm_hip_op_set_halo(storm_hip_op *op, int n_nbrs, const int32_t *nbr_rank,
                          const int64_t *send_ptr, const int64_t *send_idx,
                          const int64_t *recv_ptr);

/* `Operator::mul(y, x)`  Solvers/Operator.hpp:74:  y = beta*x + alpha*M(x).
 * Exchanges x's halo first when a halo plan is set.  x and y must not alias.
 * PRECONDITION: x is finite.  An Inf / NaN in x reaches the rows the reference's face loop lets it reach, and --
 * in the paired record format (spmv_dict = 3: two rows share their gathers; a row without a neighbour in a
 * shared slot multiplies the gathered value by weight 0) -- additionally rows whose index is one stencil
 * offset away from it without being its neighbour; rows stored in the CSR tail form a_ii x_i as
 * rowsum x_i - sum_j a_ij x_i, which is NaN instead of Inf for x_i = Inf.  Nothing else differs
 * (tests/test_gpu_edge_cases.py pins this). */
int storm_hip_op_apply(const storm_hip_op *op, double alpha, double beta, const storm_hip_vec *x,
                       storm_hip_vec *y);

/* d_i = diagonal entry i of beta*I + alpha*M; with invert != 0 its safe inverse (0 -> 0,
 * Crow/MathUtils.hpp:54-58).  What `Preconditioner::build(x, b, op)` (Preconditioner.hpp:70-72) of a
 * Jacobi preconditioner needs from the operator. */
int storm_hip_op_get_diagonal(const storm_hip_op *op, double alpha, double beta, int invert, storm_hip_vec *d);

/* `stormDivGrad(mesh, u, dt, c)`  source_apps/playground/Playground.cpp:115-131 in its own form:
 * u += dt * M(c)  (y += alpha*M(x)); what the playground's operator lambda calls twice per apply
 * (:153-167).  Exchanges x's halo first when a halo plan is set.  x and y must not alias. */
int storm_hip_op_apply_add(const storm_hip_op *op, double alpha, const storm_hip_vec *x, storm_hip_vec *y);

typedef struct storm_hip_op_stats {
  int64_t n_rows, n_cols, nnz_offdiag;  /* off-diagonal entries (2F for a face graph) */
  int64_t ell_slots;                    /* stored ELL slots incl. padding */
  int64_t tail_nnz, tail_rows;          /* CSR tail */
  int64_t n_slices, max_row_len;
  int64_t n_interior_slices;            /* slices whose rows reference no halo column */
  int64_t device_bytes;
  int64_t record_bytes;                 /* bytes of slice records one apply streams */
  int64_t value_dictionary_size;        /* > 0: weights stored as byte indices into this many distinct values */
  int64_t offset_dictionary_size;       /* > 0: columns stored as byte indices into this many distinct col - row */
  int64_t paired_rows;                  /* 1: two consecutive rows per lane share their 16-byte gathers; n_slices then counts 128-row groups; 2: the same with one common offset order (format 4); 3: format 5 (one byte per row) */
  int64_t tiled_planes;                 /* > 0: an unsplit apply runs the tiled format-4 kernel (lattice offsets -b,-a,-1,+1,+a,+b): tiles of 1024 rows x this many planes */
  int64_t spmv_blocks;                  /* workgroups of an unsplit apply */
  int64_t xcd_run_blocks;               /* (ABI 6) fp64-record kernel: runs of this many workgroups (256 rows each) go to one XCD */
} storm_hip_op_stats;
int storm_hip_op_get_stats(const storm_hip_op *op, storm_hip_op_stats *stats);
int storm_hip_op_destroy(storm_hip_op *op);

/* ---- whole-solver entry points --------------------------------------------
 * `solve<XSolver>(x, b, op)`  Solvers/Solver.hpp:261-265 for the operator
 * A = beta*I + alpha*M, with the reference's public knobs (Solver.hpp:66-72,
 * 158-159) and exactly its convergence rule (Solver.hpp:116-147): early exit
 * iff abs_tol > 0 && |r0| < abs_tol; then per iteration converged iff
 * (abs_tol > 0 && abs < abs_tol) || (rel_tol > 0 && abs/|r0| < rel_tol);
 * `iterations` = number of iterate() calls.  The loops run device-resident:
 * scalars (alpha, beta, rho, omega, Givens) never visit the host, and the
 * host polls the device's `done` flag `check_lag` iterations behind. */
typedef struct storm_hip_solver_params {
  int64_t num_iterations;           /* default 2000  (Solver.hpp:67) */
  double absolute_error_tolerance;  /* default 1e-6  (Solver.hpp:71) */
  double relative_error_tolerance;  /* default 1e-6  (Solver.hpp:72) */
  int64_t num_inner_iterations;     /* GMRES restart m, default 50 (Solver.hpp:159) */
  int32_t check_lag;                /* 0 = default (4) */
  int32_t gram_schmidt;             /* GMRES: 0 = modified (reference, SolverGmres.hpp:157-160), 1 = classical x2 */
} storm_hip_solver_params;

typedef struct storm_hip_solver_result {
  int64_t iterations;
  double absolute_error, relative_error, initial_error;
  int32_t converged;
  int32_t path_fallback;  /* 0: the path the library chose ran; 1: a cooperative (one-kernel) path could not be launched and
                             the solve ran on the kernel-per-statement path instead; 2: a cooperative kernel's bounded wait
                             gave up (the device shared with another cooperative kernel) -- x was restored and the solve
                             re-run on the kernel-per-statement path.  Same results either way (to rounding). */
  int64_t num_applies;
} storm_hip_solver_result;

void storm_hip_solver_params_default(storm_hip_solver_params *p);

/* history: NULL or room for num_iterations + 1 residual norms (entry 0 = initial). */
int storm_hip_solve_cg(const storm_hip_op *op, double alpha, double beta, const storm_hip_vec *b,
                       storm_hip_vec *x, const storm_hip_solver_params *params,
                       storm_hip_solver_result *result, double *history);       /* SolverCg.hpp:47-128 */
int storm_hip_solve_bicgstab(const storm_hip_op *op, double alpha, double beta, const storm_hip_vec *b,
                             storm_hip_vec *x, const storm_hip_solver_params *params,
                             storm_hip_solver_result *result, double *history); /* SolverBiCgStab.hpp:52-167 */
int storm_hip_solve_gmres(const storm_hip_op *op, double alpha, double beta, const storm_hip_vec *b,
                          storm_hip_vec *x, const storm_hip_solver_params *params,
                          storm_hip_solver_result *result, double *history);    /* SolverGmres.hpp:41-255 */

/* ---- the general Krylov engine ---------------------------------------------
 * Every solver of Solvers/  for ANY operator, device-resident:
 *   storm_hip_krylov_* objects stand in for the reference's solver objects (`CgSolver<Vector> s;`,
 *   Solvers/Solver.hpp:66-76): they hold the operator, the optional preconditioner (`pre_op`, `pre_side`,
 *   Solver.hpp:74-75) and the work vectors.
 * Operator and preconditioner are either native (a storm_hip_op / a diagonal held in a vector) or a
 * CALLBACK -- the reference's only call site passes a lambda through make_operator
 * (Playground.cpp:151-167, Operator.hpp:125-200).  A callback computes y = A(x) by calling this
 * library (storm_hip_op_apply, storm_hip_op_apply_add, BLAS-1 ...): those calls only ENQUEUE work on the
 * context's stream, so the solver loop around them never waits for the device -- every scalar of the
 * recurrences (alpha, beta, rho, omega, the IDR/BiCGStab(l) small systems, Hessenberg + Givens) stays in
 * HBM, the convergence rule of Solver.hpp:132-140 is evaluated there, and the host polls a pinned flag
 * `check_lag` iterations behind.  A callback that itself waits (storm_hip_dot ...) is still correct.
 * While a callback runs, the library calls it makes are predicated on the solve's `done` flag, so work
 * enqueued past convergence costs nothing.  Return 0 from a callback; anything else aborts the solve with
 * STORM_HIP_E_INVALID. */
typedef struct storm_hip_krylov storm_hip_krylov;
typedef int (*storm_hip_apply_fn)(void *user, storm_hip_vec *y, const storm_hip_vec *x);

enum storm_hip_method {
  STORM_HIP_CG = 0,          /* SolverCg.hpp:47-128 */
  STORM_HIP_BICGSTAB = 1,    /* SolverBiCgStab.hpp:52-167 */
  STORM_HIP_GMRES = 2,       /* SolverGmres.hpp:281-283 */
  STORM_HIP_FGMRES = 3,      /* SolverGmres.hpp:306-308 */
  STORM_HIP_CGS = 4,         /* SolverCgs.hpp:50-176 */
  STORM_HIP_TFQMR = 5,       /* SolverTfqmr.hpp:227-240 */
  STORM_HIP_TFQMR1 = 6,      /* SolverTfqmr.hpp:252-265 */
  STORM_HIP_BICGSTAB_L = 7,  /* SolverBiCgStab.hpp:184-383; num_inner_iterations = l (default 2) */
  STORM_HIP_IDRS = 8,        /* SolverIdrs.hpp:52-291;      num_inner_iterations = s (default 4) */
  STORM_HIP_RICHARDSON = 9   /* SolverRichardson.hpp:41-98 */
};
enum storm_hip_side { STORM_HIP_LEFT = 0, STORM_HIP_RIGHT = 1, STORM_HIP_SYMMETRIC = 2 }; /* Preconditioner.hpp:39-60 */

int storm_hip_krylov_create(storm_hip_ctx *ctx, int method, storm_hip_krylov **out);
int storm_hip_krylov_destroy(storm_hip_krylov *k);
/* A = beta*I + alpha*M of a stencil operator (CG / BiCGStab / GMRES without preconditioner then run the fused
 * kernels of storm_hip_solve_*), or a callback. */
int storm_hip_krylov_set_operator(storm_hip_krylov *k, const storm_hip_op *op, double alpha, double beta);
int storm_hip_krylov_set_operator_fn(storm_hip_krylov *k, storm_hip_apply_fn apply, void *user);
/* pre_op / pre_side.  fn == NULL removes it.  The diagonal form is y = d .* x (e.g. d from
 * storm_hip_op_get_diagonal(..., invert = 1): Jacobi); the vector must outlive the solves. */
int storm_hip_krylov_set_preconditioner_fn(storm_hip_krylov *k, storm_hip_apply_fn apply, void *user, int side);
int storm_hip_krylov_set_preconditioner_diag(storm_hip_krylov *k, const storm_hip_vec *d, int side);
/* Extra knobs: "relaxation_factor" (Richardson, default 1e-4, SolverRichardson.hpp:45). */
int storm_hip_krylov_set_real(storm_hip_krylov *k, const char *key, double value);
/* `Solver::solve(x, b, op)`: the whole solve, no host wait inside the loop.  num_applies counts operator
 * applications; *pre_applies (nullable) preconditioner applications. */
int storm_hip_krylov_solve(storm_hip_krylov *k, const storm_hip_vec *b, storm_hip_vec *x,
                           const storm_hip_solver_params *params, storm_hip_solver_result *result,
                           double *history, int64_t *pre_applies);
/* The reference's protected stepping interface (Solver.hpp:78-111): init returns |b - A x|, each iterate the
 * new residual norm (ONE host wait per call -- not per reduction); the caller owns the convergence decision,
 * as IterativeSolver::solve does; finalize after the last iterate.  `params` supplies num_inner_iterations /
 * gram_schmidt only. */
int storm_hip_krylov_init(storm_hip_krylov *k, const storm_hip_vec *b, storm_hip_vec *x,
                          const storm_hip_solver_params *params, double *initial_error);
int storm_hip_krylov_iterate(storm_hip_krylov *k, double *error);
int storm_hip_krylov_finalize(storm_hip_krylov *k);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* STORM_HIP_H_ */
