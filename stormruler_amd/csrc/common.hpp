// Internal declarations shared by the translation units of libstorm_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/storm_hip.h"

namespace storm {

// ---- error plumbing -------------------------------------------------------
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define STORM_FAIL(code, ...)        \
  do {                               \
    ::storm::set_error(__VA_ARGS__); \
    return (code);                   \
  } while (0)

#define STORM_REQUIRE(cond, ...) \
  do {                           \
    if (!(cond)) STORM_FAIL(STORM_HIP_E_INVALID, __VA_ARGS__); \
  } while (0)

#define HIP_TRY(expr)                                                               \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess)                                                           \
      STORM_FAIL(STORM_HIP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                 __FILE__, __LINE__);                                               \
  } while (0)

#define STORM_TRY(expr)        \
  do {                         \
    int s_ = (expr);           \
    if (s_ != 0) return s_;    \
  } while (0)

// ---- launch geometry --------------------------------------------------------
constexpr int kWave = 64;            // CDNA wavefront
constexpr int kBlock = 256;          // 4 waves, one per SIMD
constexpr int kNumXcd = 8;           // MI355X: 8 XCDs, blocks dealt round-robin
constexpr int kMaxReduceBlocks = 2048;  // 256 CUs x 8 resident 256-thread blocks
constexpr int kMaxMulti = 64;        // widest multi-dot / multi-axpy in one launch
constexpr int kResultRing = 8;       // host-returning reductions in flight (storm_hip_multi_dot_begin / _end)
constexpr int kSlab = 256;           // doubles in the device scalar slab
constexpr int kStateRing = 64;       // iterations the host may run ahead of the device's verdict
constexpr int kStage2 = 128;         // blocks of the first pass of a two-pass final reduction
// Up to this many per-block partials one block folds them in a single launch (32 loads per thread at the limit);
// beyond, a first pass of kStage2 blocks (raising the limit to 32 768 gave no measurable gain at 128^3: the longer
// single-block fold costs what the second launch did).
constexpr int kSinglePassPartials = 8192;

// Device-resident solver state: every scalar a Krylov loop carries, so no
// alpha/beta/omega/Givens value ever visits the host (SURVEY.md section 7
// "Reduction latency").  One instance lives in each context.
struct SolverState {
  double s[kSlab];          // named slots, see solvers.hip
  double initial_error;
  double absolute_error;
  double relative_error;
  double abs_tol, rel_tol;
  long long iteration;      // IterativeSolver::iteration (Solver.hpp:66)
  long long num_iterations;
  int done;                 // set by the device when converged or out of iterations
  int converged;
  double *history;          // device buffer [num_iterations + 1] or null
  unsigned long long *done_ring;  // device alias of a pinned host ring: after iteration i (1-based) the word ring_word(gen, i, done) at [(i - 1) % kStateRing]; ring_wait()
  int verify_failed;        // option ticket_verify: an in-kernel (ticketed) reduction disagreed with its two-launch recomputation (sticky)
  unsigned long long ring_gen;  // this solve's generation (20 bits): every word posted into the ring carries it, so that a kernel of an
                                // ABORTED solve that posts late cannot be taken for this solve's iteration of the same number (ring_word)
};
// The word iteration i (1-based) of generation g posts: g << 44 | i << 1 | done; i = all ones: "done at once" (begin()).
constexpr unsigned long long kRingIterMask = (1ull << 43) - 1;
__host__ __device__ inline unsigned long long ring_word(unsigned long long gen, unsigned long long iteration, bool done) {
  return ((gen & 0xfffffull) << 44) | ((iteration & kRingIterMask) << 1) | (done ? 1ull : 0ull);
}

struct Comm;  // comm.hip

// lazy.hip -- a vector statement / an operator apply whose launch is held back (option lazy_statements)
struct LazyStmt {
  int kind = 0;              // 0: y = c0 v0 [+ c1 v1] over n rows; 1: y = beta x + alpha M(x)
  double *y = nullptr;
  const double *v[2] = {nullptr, nullptr};
  double c[2] = {0.0, 0.0};
  int nt = 0;
  int64_t n = 0;
  const storm_hip_op *op = nullptr;
  double alpha = 0.0, beta = 0.0;
  const double *x = nullptr;
  storm_hip_vec *yvec = nullptr;  // the vector y is the storage of (kind 0)
};

// What a storm_hip_krylov object owns on the device and in pinned host memory; a destroyed engine leaves it with its
// context (storm_hip_ctx::krylov_free) for the next one: creating a solver object per solve -- the reference's usage,
// Playground.cpp:199-202 -- then costs no allocation (two pinned allocations, a device one and a blocking memset were
// most of a 0.1 ms create + destroy).
struct KrylovRes {
  SolverState *d_st = nullptr, *h_st = nullptr;
  unsigned long long *h_ring = nullptr, *d_ring = nullptr;
  double *S = nullptr;  // the engine's scalar register file
  int S_cap = 0;
};

}  // namespace storm

struct storm_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;       // compute stream: every kernel runs here
  hipStream_t comm_stream = nullptr;  // halo pack + RCCL send/recv
  hipEvent_t ev_x_ready = nullptr, ev_halo_done = nullptr;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
  int num_cus = 0;
  std::string name;
  int64_t total_mem = 0;
  // reduction workspace
  double *d_partials = nullptr;       // [partials_capacity] per-block partial sums
  int64_t partials_capacity = 0;
  double *d_partials2 = nullptr;      // [kMaxMulti * kStage2] second-stage partials
  double *d_scalars = nullptr;        // [kMaxMulti] results of host-visible reductions
  unsigned long long lat_seq = 0;     // running sequence number of the cooperative Gram-Schmidt chains' all-reduces
  double *d_gmres = nullptr;          // GMRES' Hessenberg matrix, beta, cs, sn of the fused loop (grown on demand, kept)
  size_t gmres_capacity = 0;
  double *d_ticket_sums = nullptr;    // ... and the groups' sums (a buffer of their own: d_partials2 may hold a first pass a kernel is still reading)
  int *d_tickets = nullptr;           // ticket_device.hpp: self-re-arming counters of the in-kernel reductions
  char *d_lat_slots = nullptr;        // latency path: two 256-byte all-reduce slots per block (256 blocks)
  double *h_scalars = nullptr;        // pinned mirror
  // host-returning reductions on one rank: the kernel's last block stores the sums straight into pinned host memory
  // as self-validating words { low half | tag }, { high half | tag } and the host polls them -- no copy, no stream wait
  // (a ring of kResultRing requests: storm_hip_multi_dot_begin / _end keep several reductions in flight)
  unsigned long long *h_result_words = nullptr, *d_result_words = nullptr;  // [kResultRing][2 * 8]
  unsigned result_seq = 0;
  struct ResultSlot {
    unsigned tag = 0;     // 0: free
    int k = 0;
    bool ready = false;   // value[] holds the sums already (the ordinary road computed them in _begin)
    double value[storm::kMaxMulti] = {};
  } result_ring[storm::kResultRing];
  static constexpr int64_t opt_host_result = 1;        // 0: device scalars + hipMemcpyAsync + hipStreamSynchronize
  storm::SolverState *d_state = nullptr;
  storm::SolverState *h_state = nullptr;  // pinned staging copy of the state
  unsigned long long *h_done_ring = nullptr;  // pinned, written by the device's step kernels (solver_device.hpp advance())
  unsigned long long *d_done_ring = nullptr;  // device pointer to the same memory
  std::vector<hipEvent_t> ev_ring;            // (option poll_events = 1: a marker behind every iteration, the r02 form; created on first use)
  std::vector<storm::KrylovRes> krylov_free;   // krylov.hip: resources of destroyed engines, reused by the next create
  static constexpr int64_t opt_poll_events = 0;
  unsigned long long ring_gen = 0;            // generation of the current solve's ring words (state_init draws the next one)
  // options
  int64_t opt_ell_cap = 0;
  int64_t opt_spmv_dict = 4;         // dictionary records whenever an operator qualifies (lossless): 3 + paired rows, 2 values + column offsets, 1 values only, 0 never
  int64_t opt_spmv_spw = 0;          // slices per wave of the dictionary kernel: 1, 2 or 4 (0 = default)
  int64_t opt_spmv_xcd_remap = 8;    // 0 off; 1 one contiguous run per XCD (slower); G > 1: runs of G tiles per XCD (A/B knob, option spmv_xcd_remap: tools/tet_traffic_ab.py)
  // ... of the fp64-record kernel (spmv_sell_kernel: what any mesh gets): runs of 64 blocks = 16 384 rows per XCD.  Round 6,
  // profiles/r10f_tet_traffic_ab.jsonl, r10g_xcd_group_sweep.jsonl: a row of x that two XCDs gather is fetched into two L2s;
  // on the 12.6 M tetrahedra (Z-order numbering) runs of 8 blocks move 1.19 x the algorithmic bytes (0.779 of the peak),
  // of 32: 1.12 x, of 128: 1.075 x, ONE run per XCD 1.02 x but slower (eight distant streams); 64 is the fastest on both the
  // tetrahedra (0.811) and the 256^3 box in natural order (0.796 against 0.774; 16 and 32: 0.784; twice each,
  // r10i_sell_xcd_group_sweep_fine.jsonl).  On the box runs of 64 put every +-plane neighbour (256 blocks away) on ANOTHER XCD
  // -- the counter reads 1.09 x the algorithmic bytes where runs of 8 / 16 / 32 read 1.025 x (256 = 0 mod 8 G: the same XCD) --
  // and are still the fastest: those bytes are hits in the Infinity Cache, what costs time is how far apart in DRAM the eight
  // XCDs' streams run.  A chooser that counted cross-XCD columns per operator was written and removed: it picks 8 - 32 there.
  int64_t opt_spmv_xcd_remap_sell = 64;
  int64_t opt_nt = 1;
  static constexpr int64_t opt_sweep_alternate = 1;   // fused CG: consecutive kernels sweep the rows in opposite directions (2: and without non-temporal hints)
  int stream_reverse = 0;            // ... and the same for the next elementwise kernel
  int spmv_reverse = 0;              // set around a format-4 SpMV launch by the solver: deal the tiles out from the far end
  static constexpr int64_t opt_spmv_canon_groups = 2; // format-4 / 5 kernel: 128-row groups per wavefront (1 or 2)
  static constexpr int64_t opt_spmv_tile_lds_pad = 0;   // A/B knob: extra dynamic LDS per block of the tiled kernel (fewer resident tiles per CU)
  int64_t opt_spmv_canon_tile_min_rows = (int64_t)1 << 20;  // ... for operators of at least this many rows
  int64_t opt_spmv_canon_tile = 2;   // format 4 on a lattice (offsets -b,-a,-1,+1,+a,+b): tiles of 1024 rows x this many planes (2, or 4) with the +-a / +-1 neighbours from LDS and the +-b ones from registers; 0 = the plain kernel.  Measured at 256^3 (profiles/r03f, r03g): CG step 242 (2 planes) / 247 (4) us per iteration, BiCGStab 496 / 510
  static constexpr int64_t opt_vec_contiguous = 0;     // vectors in physically contiguous device memory (hipDeviceMallocContiguous)
  int64_t opt_mgs_steps = 4;          // throughput-path Gram-Schmidt: steps per pass over w (2: mgs_pair_kernel; 3, 4: mgs_multi_kernel)
  int64_t opt_resident_early = 1;    // resident CG: the residual's surface published under the all-reduce that yields beta (res_halo MODE 2; behind the block's own arrival at that all-reduce): bitwise the same solve, 7 - 12 % faster (128^3: 16.8 -> 15.4 us per iteration)
  int64_t opt_resident_apply_cache = 1;  // resident path: a pair of rows' coefficients stay in registers from plane to plane while the weight words do not change (0: decoded per plane; the same bits)
  int64_t opt_resident_halo_interleave = 1;  // resident CG, boxes of more than 2 planes: the second wave of every SIMD forms the halo of p' before the update of its own rows
  int64_t opt_coop_mgs_prefetch = 1; // ... the next group's basis vectors requested under the all-reduce (up to 4 row pairs per thread)
  int64_t opt_coop_mgs_lds_prefetch = 1; // ... eight row pairs per thread (128^3): the next group's vectors through LDS (LDS-DMA)
  int64_t opt_coop_mgs_rotate_early = 1;  // ... block 0 applies the column's earlier rotations under the norm's all-reduce (test_disable bit 512: off)
  int64_t opt_coop_mgs_alternate = 1;  // ... the chain's vector order alternates with k: ascending / descending (test_disable bit 256: off)
  int64_t opt_coop_mgs_xcd_runs = 1;  // ... the chain's blocks own ONE contiguous run of row chunks per XCD (test_disable bit 128: off)
  int64_t opt_coop_mgs_apply = 1;    // ... with the operator apply in front of it done by the chain kernel itself (format-4 lattice operators)
  static constexpr int64_t opt_coop_mgs_pairs = 1;    // cooperative Gram-Schmidt chain: two steps per synchronisation point
  static constexpr int64_t opt_coop_dense = 1;        // the multi-step Gram-Schmidt chain's all-reduce with dense value-major slots (0: the two-level form; 2: the resident kernels too)
  int64_t opt_coop_mgs_quad = 1;     // ... FOUR steps per synchronisation point (blocks of 512 threads; <= 2^21 rows)
  char *d_quad_slots = nullptr;      // ... its all-reduce slots (ten values each)
  int64_t opt_coop_mgs_lds = 1;      // ... with the next pair of basis vectors fetched by LDS-DMA into a ring (<= 2^21 rows)
  int64_t opt_spmv_mixed = 1;        // partitioned operators: format 4 for the groups that read no halo column, format 3 for the rest
  int64_t opt_profile_spmv = 0;
  std::vector<storm::LazyStmt> lazy_q;  // held-back statements (lazy.hip), in program order
  std::unordered_map<const void *, int> occupancy;  // latency.hip: blocks per CU of a cooperative kernel on THIS context's device
  int64_t opt_test_disable = 0;         // option test_disable (context.hip): a bit mask that switches single refinements OFF so that tests can compare a kernel with its plainer form, bit for bit
  int64_t opt_lazy = 0;                 // option lazy_statements
  int callback_depth = 0;               // > 0 while a solver is inside an operator / preconditioner callback (nothing waits there)
  int64_t n_lazy_fused_dots = 0, n_lazy_fused_pairs = 0, n_lazy_apply_dots = 0, n_lazy_cg_steps = 0;
  storm_hip_vec *lazy_spare = nullptr;  // where a fused CG step writes the new direction (lazy.hip: try_cg_step)
  int64_t opt_profile_comm = 0;       // RCCL transport: stamp kernels around the halo exchange and the all-reduces (comm.hip comm_profile_*)
  int64_t opt_blas1_nt = 1;  // non-temporal loads/stores in the streaming kernels: 0 never, 1 for vectors of at least blas1_nt_rows rows, 2 always
  static constexpr int64_t opt_blas1_nt_rows = (int64_t)6 << 20;  // (48 MiB per vector: beyond, a solver's vectors no longer stay in the 256 MiB Infinity Cache between kernels)
  static constexpr int64_t opt_graph = 0;     // replay CG / BiCGStab iterations from a captured hipGraph: measured slower than eager launches (profiles/r01_notes.md), off
  static constexpr int64_t opt_fuse_mgs = 1;  // GMRES/MGS on one rank, <= 2048 blocks: each step folds the previous step's partials itself (no final-reduction launch in between)
  static constexpr int64_t opt_coop_mgs_min_rows = 0;  // ... from this many rows on (0: always; with two steps per synchronisation point the chain is no slower than a launch per step even on small meshes)
  int64_t opt_coop_mgs = 1;             // GMRES: the Gram-Schmidt chain of an Arnoldi step as one cooperative kernel (latency.hip)
  int64_t opt_latency_publish = 1;      // ... its rows published with awaited atomic exchanges (0: write-through stores, ordered by their acknowledgement)
  int64_t opt_coop_plain = 1;           // the cooperative kernels by ordinary launches (latency.hip coop_launch; 0: hipLaunchCooperativeKernel)
  int64_t opt_coop_force_fail = 0;      // test hook: 1 = cooperative launches "fail", 2 = cooperative kernels "gave up" (once per solve)
  int coop_ran = 0;                     // a cooperative kernel of the current solve has run
  int coop_disabled = 0;                // set while a solve is re-run without cooperative kernels
  int coop_fallback = 0;                // what happened in the current solve (storm_hip_solver_result::path_fallback)
  int64_t coop_skip = 0, coop_backoff = 0;  // solves still to run without cooperative kernels after one gave up; the length of the last back-off
  int64_t opt_latency_path = 1;         // small operators: CG as one cooperative persistent kernel (latency.hip)
  // resident.hip: lattice operators as one persistent kernel per solve, every block owning a box of the lattice
  int64_t opt_resident_path = 1;
  static constexpr int64_t opt_resident_min_rows = 0;            // ... from this many rows on (below: the latency path, where it applies)
  int64_t opt_resident_max_rows = (int64_t)1 << 22;
  int64_t opt_resident_planes = 0;              // ... exactly this many planes per block (0: the fewest that cover the lattice with one block per CU)
  static constexpr int64_t opt_resident_max_planes = 12;         // ... with at most this many planes per block (registers)
  char *d_res_exch = nullptr;                   // its exchange buffer: one 16-byte granule per row (grown on demand)
  int64_t res_exch_rows = 0;
  int64_t opt_resident_profile = 0;             // the kernels time their phases (storm_hip_ctx_get_counter "resident_phase_max_k" / "_mean_k", ticks of 10 ns)
  long long *d_res_prof = nullptr;
  int res_prof_blocks = 0;
  char *d_res_slots = nullptr;                  // its all-reduce slots + the two sequence numbers it carries from solve to solve
  int opt_lin_fuse = 1;                 // engine: two consecutive vector statements go out as one pass
  int64_t opt_ticket_verify_inject = 0;  // test hook for the above
  int64_t opt_ticket_verify = 0;        // k > 0: every k-th iteration the fused loops recompute their ticketed reductions by the two-launch path and compare on the device (sticky flag -> the solve returns an error)
  int opt_ticket_reduce = 1;            // fused CG / BiCGStab: reductions finish inside the kernels that produce their partials
  int opt_fused_reduce = 1;             // engine: a reduction is ONE launch (its last block folds the partials and runs the scalar program)
  int opt_latency_cache = 1;            // ... with the wave's operator records held in registers where they fit
  int64_t opt_latency_rows = 1 << 19;   // ... up to this many rows (a compact copy of the operator is kept for it)
  int64_t opt_rccl_fused = 1;           // RCCL transport: the fused CG step on a partitioned lattice operator (boundary planes of the new direction packed by a small kernel, sent under the marching launch)
  int64_t opt_rccl_ticket = 1;          // ... with the LOCAL sums of <p,z> and <r,r> finished inside the kernels that produce them (tickets); the all-reduce and the scalar step stay launches
  int64_t opt_comm_wait_seconds = 120;  // RCCL transport, flag hand-offs: how long a one-thread waiter polls before it gives up (STORM_HIP_E_COMM from the next checked call; the first 4 exchanges of a communicator, inside which RCCL connects its peers: at least 180 s).  A cross-stream event waits for ever; a kernel must not, but a rank that builds an operator or reads a mesh between two solves may well be tens of seconds late
  int64_t opt_rccl_flag_wait = 1;       // RCCL: the boundary rows wait for a flag in device memory set behind the exchange, not for a cross-stream event (comm.hip)
  int64_t opt_rccl_early_halo = 1;      // RCCL, BiCGStab: the halo of s / p' leaves before the kernel that forms the vector runs (rows to send formed by a small kernel)
  static constexpr int64_t opt_ipc_bicg_ticket = 1;      // peer windows, BiCGStab: sums finished by tickets and exchanged by the finishing block (as CG does)
  static constexpr int64_t opt_ipc_fused = 1;            // peer-window transport: the interior launch sends, the boundary launch reads the window (0: stand-alone send / receive-copy kernels)
  static constexpr int64_t opt_ipc_streams = 2;          // peer-window halo exchange: 2 = on the comm stream beside the interior rows, 1 = on the compute stream around them
  int64_t opt_generic_solvers = 0;  // 1: storm_hip_krylov_solve never takes the fused CG / BiCGStab / GMRES loops (A/B knob)
  int64_t opt_cg_march_fill = 2048;    // ... fewer planes per block on smaller lattices, so that the grid holds about this many blocks (0: cg_march as given)
  int64_t opt_cg_march_alternate = 1;  // odd z-chunks of the marching step kernel march downwards (spmv.hip MarchArgs::alternate)
  int64_t opt_cg_march = 8;   // ... as blocks of 1024 rows marching through this many planes (0: tiles, spmv_canon_tile planes deep); 256^3, us per CG iteration: tiles 239, 8 planes 230, 16 234, 32 236, 64 237 (profiles/r03k)
  int64_t opt_cg_fuse = 1;   // fused CG, one rank, tiled format-4 operator: the SpMV kernel ends the previous iteration (x += alpha p, p = r + beta p) itself
  static constexpr int64_t opt_fold_pz = 1;   // CG, one rank, > 8192 SpMV partials: cg_r_kernel folds the first-pass partials of <p,z> itself (one launch fewer)
  static constexpr int64_t opt_fuse_dot = 1;  // 0: reductions after an SpMV run as separate kernels (A/B knob)
  // Vector storage released by vec_destroy, kept for the next vec_create of the same size: a solve
  // allocates its work vectors on entry and frees them on return (the reference re-assigns them in
  // every init, SolverCg.hpp:57-59); hipMalloc + hipFree of three 134 MB vectors cost ~7 ms per solve.
  // Reuse is ordered by the compute stream.  Bounded by opt_pool_bytes; freed with the context.
  std::vector<std::pair<size_t, double *>> pool;
  // Arenas: the vectors of one size are slots of ONE (physically contiguous) allocation, `pitch` bytes apart -- where a
  // solver's vectors lie relative to each other decides a few per cent of a multi-stream kernel's rate, and separate
  // allocations land wherever the driver puts them (context.hip: vec_create_impl; tools/placement_probe.py).
  struct VecArena {
    char *base = nullptr;
    size_t bytes = 0, pitch = 0;  // size class (= storm_hip_vec::bytes), distance between slots
    int slots = 0, used = 0;
  };
  std::vector<VecArena> arenas;
  int64_t opt_vec_arena = 1;            // 0: every vector an allocation of its own
  static constexpr int64_t opt_vec_arena_contiguous = 1; // arenas in physically contiguous memory (hipDeviceMallocContiguous)
  static constexpr int64_t opt_cg_roles = 8;             // solve_cg_body: permutation of the work vectors' roles over their arena slots (A/B knob; 24 permutations at 256^3: 4 505 - 4 570 it/s, profiles/r05z_roles.txt)
  static constexpr int64_t opt_vec_arena_slots = 8;
  static constexpr int64_t opt_vec_arena_max_bytes = (int64_t)64 << 30;  // all arenas of a context together; beyond: vectors allocated one by one
  static constexpr int64_t opt_vec_arena_skew_kib = 0;   // pitch = the vector rounded up to 2 MiB + this
  size_t pool_bytes = 0;
  int64_t opt_pool_bytes = (int64_t)16 << 30;
  std::vector<hipEvent_t> prof_events;  // pairs (start, stop), grown on demand
  size_t prof_used = 0;
  // Non-null while a solver is inside an operator / preconditioner callback: the device `done` flag of that
  // solve.  Public entry points predicate the kernels they enqueue on it (work past convergence is free).
  const int *api_done = nullptr;
  // diagnostics: which path the solves took (storm_hip_ctx_get_counter)
  int64_t n_resident_solves = 0, n_latency_solves = 0, n_throughput_solves = 0, n_engine_solves = 0, n_cg_fused_steps = 0;
  // communicator
  storm::Comm *comm = nullptr;
  int n_ranks = 1, rank = 0;
};

struct storm_hip_vec {
  storm_hip_ctx *ctx = nullptr;
  int64_t n_owned = 0, n_halo = 0;
  double *d = nullptr;     // element 0 (kVecGuard zero doubles sit in front of it, >= 4 behind the last row)
  double *base = nullptr;  // the allocation
  size_t bytes = 0;        // allocation size (pool key)
  bool exposed = false;    // storm_hip_vec_device_ptr has handed d out: the storage is never exchanged (lazy.hip)
};
constexpr int kVecGuard = 32;

namespace storm {

struct HaloPlan {
  int n_nbrs = 0;
  std::vector<int> nbr_rank;
  std::vector<int64_t> send_ptr, recv_ptr;  // [n_nbrs + 1]
  int *d_send_idx = nullptr;                // [send_ptr.back()]
  double *d_sendbuf = nullptr;
  int64_t n_send = 0;
};

}  // namespace storm

// Sliced ELL (slices of one wavefront = 64 rows, column-major inside a slice)
// plus a CSR tail for the entries of rows longer than the slice's width.
struct storm_hip_op {
  storm_hip_ctx *ctx = nullptr;
  int64_t n_rows = 0, n_halo = 0;
  int64_t n_slices = 0, n_interior_slices = 0;
  int64_t nnz = 0, ell_slots = 0, max_row_len = 0;
  int64_t *d_slice_off = nullptr;  // [n_slices + 1] byte offset of each slice record
  char *d_pack = nullptr;          // slice records: [ext 64 f64][col W*64 i32][val W*64 f64]
  double *d_dict = nullptr;        // value-dictionary records: the 256-entry table (spmv.hip)
  int dict_size = 0;               // > 0: records are [idx 64 u64][col W*64 i32]
  int *d_offs = nullptr;           // format 2: the 256-entry column-offset table
  int offs_size = 0;               // > 0: records are 64 x 16-byte words (values + offsets as byte indices)
  int xcd_group_sell = 0;          // spmv_sell_kernel: runs of this many blocks per XCD (option spmv_xcd_remap at build time)
  int pair = 0;                    // 1: format 3 -- 128-row groups of paired rows, n_slices counts those groups; 2: format 4 (common offset order)
  char *d_bnd_pack = nullptr;      // mixed operator: format-3 records of the boundary groups, in d_boundary order
  int bnd_width = 0;               // ... and their merged width
  int canon_k = 0, canon_m1 = -1;  // format 4: number of common offsets, slot of offset -1 (+1 follows)
  int canon_off[7] = {0, 0, 0, 0, 0, 0, 0};
  int64_t pack_bytes = 0;
  int64_t spw = 1;                 // slices per wavefront of the uniform-width dictionary kernel
  // tail
  int64_t tail_rows = 0, tail_nnz = 0;
  int *d_tail_row = nullptr;       // [tail_rows]
  int64_t *d_tail_ptr = nullptr;   // [tail_rows + 1]
  int *d_tail_col = nullptr;
  double *d_tail_val = nullptr;
  int uniform_width = 0;           // > 0 when every slice has this width
  // slices whose rows read no halo column (overlap the halo exchange) / the rest
  std::vector<int> h_interior, h_boundary;
  int *d_interior = nullptr, *d_boundary = nullptr;
  int64_t n_interior = 0, n_boundary = 0;
  int64_t int_plane0 = 0, int_plane1 = 0;  // mixed operator on a lattice: the interior groups are exactly these planes
  int64_t device_bytes = 0;
  // compact fp64 copy for the latency path (latency.hip); null when the operator does not qualify
  char *d_lat_pack = nullptr;
  int64_t *d_lat_off = nullptr;
  int64_t lat_bytes = 0;
  storm::HaloPlan halo;
};

namespace storm {

// ---- internal cross-TU API ----------------------------------------------------
// Where a kernel reads a scalar from: a host value or a slot of the device slab
// (optionally negated / transformed on the fly).
struct Scal {
  const double *p;  // device pointer or null
  double v;         // host value when p == null
  double sign;      // multiplies the loaded value
};
static inline Scal host_scal(double v) { return Scal{nullptr, v, 1.0}; }
static inline Scal dev_scal(const double *p, double sign = 1.0) { return Scal{p, 0.0, sign}; }

// Shape of every streaming (BLAS-1) kernel, from tools/stream_bench.hip on MI355X (1 GiB copy):
// a grid capped at 8 blocks/CU with a grid-stride loop reaches 4.96 TB/s; one trip per thread
// with 4 independent 16-byte accesses per stream in flight, non-temporal loads and stores,
// 6.47 TB/s.  So: blocks = ceil(n / 2048), each thread moves 4 double2 per stream.
constexpr int kUnroll = 4;
constexpr int kStreamBlockElems = kBlock * kUnroll * 2;  // doubles per block and stream
constexpr int kMaxStreamBlocks = 32768;  // beyond: grid-stride (keeps the partial arrays -- and the ticket groups -- bounded)
static inline int stream_blocks(int64_t n) {
  int64_t b = (n + kStreamBlockElems - 1) / kStreamBlockElems;
  if (b < 1) b = 1;
  if (b > kMaxStreamBlocks) b = kMaxStreamBlocks;
  return (int)b;
}

// Non-temporal accesses for a streaming kernel over n rows?  (blas1_device.hpp: nt_dispatch)
static inline int stream_nt(const storm_hip_ctx *c, int64_t n) {
  return (c->opt_blas1_nt == 2 || (c->opt_blas1_nt == 1 && n >= c->opt_blas1_nt_rows)) ? 1 : 0;
}

// blas1.hip -- all asynchronous on ctx->stream, owned rows only.
// `done` (nullable): device flag; kernels return immediately when it is set.
int k_fill(storm_hip_ctx *c, double *y, int64_t n, double v);
int k_copy(storm_hip_ctx *c, double *y, const double *x, int64_t n, const int *done);
int k_scale(storm_hip_ctx *c, double *y, int64_t n, Scal s, bool divide, const int *done);
// y = a*x + b*z
int k_axpbz(storm_hip_ctx *c, double *y, Scal a, const double *x, Scal b, const double *z,
            int64_t n, const int *done);
// p = r + beta*(p - omega*v)
int k_bicg_p(storm_hip_ctx *c, double *p, const double *r, Scal beta, Scal omega, const double *v,
             int64_t n, const int *done);
// out[j] = <a, bs[j]> for j < k, written to d_out (device); local sums only.
int k_multi_dot(storm_hip_ctx *c, const double *a, const double *const *bs, int k, int64_t n,
                double *d_out, const int *done);
int k_multi_dot_partials(storm_hip_ctx *c, const double *a, const double *const *bs, int k, int64_t n,
                         int *nb_out, const int *done);
// y += sum_j coef[j] * xs[j]; coefficients from device memory (d_coef, sign applied).
int k_multi_axpy(storm_hip_ctx *c, double *y, const double *d_coef, double sign,
                 const double *const *xs, int k, int64_t n, const int *done);
// Final pass over per-block partials: out[j] = sum_b partials[j * nblocks + b].
int k_dot_partials(storm_hip_ctx *c, const double *a, const double *b, int64_t n, double *partials, int nb,
                   const int *done);
int k_reduce_final(storm_hip_ctx *c, const double *partials, int nblocks, int k, double *d_out,
                   const int *done);

// context.hip: a work vector with only its guard, halo tail and padding zeroed (the solver writes the owned rows first)
int vec_create_work(const storm_hip_vec *like, storm_hip_vec **out);
// ... `count` of them, their edges zeroed by ONE launch (a memset is a launch of its own with ~10 us in front of it:
// eight of them per CG solve were a quarter of a K = 0 solve's 0.34 ms)
int vec_create_work_batch(const storm_hip_vec *like, int count, storm_hip_vec **out);
// context.hip: the device's SolverState for a new solve, written by one small kernel (no staged copy, no stream wait)
// ... and read back: one small kernel stores it into pinned host memory, then the stream wait every solve ends with (a
// staged device-to-host copy is a launch of its own with ~30 us of idle device in front of it)
int state_read(storm_hip_ctx *c, const SolverState *d_state, SolverState *h_pinned);
int state_init(storm_hip_ctx *c, SolverState *d_state, double abs_tol, double rel_tol, long long num_iterations, double *history,
               unsigned long long *d_ring);  // (draws the solve's generation: storm_hip_ctx::ring_gen)

// context.hip: the host's view of a solve's progress.  The device's step kernels post the verdict of iteration i
// (1-based) as ONE self-validating word (i << 1 | done) into a pinned ring (solver_device.hpp advance()); the host,
// `lag` iterations ahead, polls the word -- no marker in the stream: an event recorded behind every iteration is a
// barrier with a system-scope release between two kernels, 5.9 us per CG iteration at 256^3
// (profiles/r03t_event_gap.txt).  ring_post / ring_wait: option poll_events = 1 brings the markers back.
int ring_post(storm_hip_ctx *c, std::vector<hipEvent_t> &events, int64_t it);
int ring_wait(storm_hip_ctx *c, std::vector<hipEvent_t> &events, volatile unsigned long long *ring, int64_t it, bool *stop);  // (words of c->ring_gen only)

// spmv.hip
// y = beta*x + alpha*M x over slices [s0, s1); when dot_w != null also writes
// per-block partials of <dot_w, y> (and <y, y> when dot_yy) into partials.
struct SpmvDot {
  const double *w = nullptr;   // partial of <w, y>
  bool yy = false;             // also partial of <y, y>
  double *partials = nullptr;  // [2 * nblocks] layout: [<w,y> blocks..., <y,y> blocks...]
  int *nblocks_out = nullptr;  // host out: number of partial blocks written
  // Finish the reduction(s) inside the SpMV kernel (ticket_device.hpp): the sums go to out[0] (<w,y>) and out[1]
  // (<y,y>), no final-pass launch follows.  Honoured by the format-4 / 5 kernel in an unsplit launch;
  // *ticketed_out tells whether it was.
  double *out[2] = {nullptr, nullptr};
  int *ticketed_out = nullptr;
  // Fused CG step (spmv.hip, CgFuseArgs): the kernel first ends the previous iteration -- x += alpha p, p' = r + beta p
  // into p_out -- and applies the operator to p' (x = p is then the OLD direction; w must be x).  cg.x == null: off.
  struct {
    const long long *iteration = nullptr;
    long long my_iteration = 0;
    const double *ca = nullptr, *cb = nullptr;
    double *x = nullptr;
    const double *r = nullptr;
    double *p_out = nullptr;
    double ca_imm = 0.0, cb_imm = 0.0;  // with ca / cb null: the values themselves (unsplit marching launch only)
  } cg;
};
int spmv_launch(const storm_hip_op *op, Scal alpha, Scal beta, const double *x, double *y,
                const SpmvDot *dot, const int *done, bool accumulate = false);
int spmv_grid_blocks(const storm_hip_op *op);
bool spmv_can_fuse_cg(const storm_hip_op *op);
bool spmv_can_march(const storm_hip_op *op);  // the z-marching kernel applies to an unsplit launch of this operator
int op_upload_slice_lists(storm_hip_op *op);

// latency.hip
int op_make_latency_copy(storm_hip_op *op, int64_t n, int64_t n_halo, const std::vector<int64_t> &row_ptr,
                         const std::vector<int> &col, const std::vector<double> &val, const std::vector<double> &ext);
bool cg_latency_eligible(const storm_hip_op *op);
constexpr int kStatusCoopGaveUp = 1000;  // internal status of lat_check_gave_up: the caller re-runs the solve (below)
int lat_check_gave_up(storm_hip_ctx *c);
// Run a solve that may use cooperative kernels; when one of them gave up, restore x and run it again without them.
int coop_solve_with_fallback(storm_hip_ctx *c, storm_hip_vec *x, int (*run)(void *), void *arg, int *fallback_out);
// What the cooperative chain needs to finish an Arnoldi step itself (fused GMRES loop): the state, the Hessenberg and
// rotation arrays, and where sqrt(<w,w>) goes.
struct MgsGivens {
  SolverState *st;
  double *H, *beta, *cs, *sn, *hn_slot;
};
// The operator apply GMRES performs before its Gram-Schmidt chain (SolverGmres.hpp:155), for the chain kernel to do itself.
struct ChainApply {
  const storm_hip_op *op;
  double alpha, beta;  // y = beta x + alpha M(x)
  const double *x;     // = q[k]
};
int gmres_mgs_chain_coop(storm_hip_ctx *c, int64_t n, const int *done, double *w, const double *const *q, int k, int m,
                         double *H, double *norm2_out, bool normalise, bool *taken, const MgsGivens *givens,
                         const ChainApply *apply = nullptr, bool *applied = nullptr);
// *taken = false: no cooperative kernel ran (none fits, or the launch was refused) -- take the throughput path
int cg_latency_solve(const storm_hip_op *op, double alpha, double beta, const double *b, double *x, double *p,
                     double *r, SolverState *d_state, bool *taken);
int bicgstab_latency_solve(const storm_hip_op *op, double alpha, double beta, const double *b, double *x,
                           double *const work[4], SolverState *d_state, bool *taken);

// resident.hip
bool res_eligible(const storm_hip_op *op, bool bicgstab);
int res_solve(bool bicgstab, const storm_hip_op *op, double alpha, double beta, const double *b, double *x, double *rt,
              SolverState *d_state, bool *taken);

// lazy.hip
int lazy_flush(storm_hip_ctx *c);  // launch whatever is held back (a no-op when nothing is)
static inline int lazy_sync(storm_hip_ctx *c) { return (c != nullptr && !c->lazy_q.empty()) ? lazy_flush(c) : 0; }
int lazy_push_lin(storm_hip_ctx *c, storm_hip_vec *y, double c0, const double *v0, double c1, const double *v1, int nt, int64_t n);
int lazy_push_apply(const storm_hip_op *op, double alpha, double beta, const double *x, double *y);
bool lazy_try_dot(storm_hip_ctx *c, const double *a, const double *b, int64_t n, double *result, int *status);

// comm.hip
int comm_allreduce_sum(storm_hip_ctx *c, double *d_buf, int count);  // in place, on ctx->stream
int comm_halo_exchange_begin(const storm_hip_op *op, double *x);      // pack + send/recv on comm stream
int comm_halo_exchange_end(const storm_hip_op *op);                   // compute stream waits
bool comm_is_rccl(const storm_hip_ctx *c);
int comm_halo_exchange_begin_direction(const storm_hip_op *op, const double *p, const double *r, const double *cb, double *p_out);
// ... of BiCGStab's s (mode 0, target = r) / p' (mode 1, target = p), formed from the operands before the update kernel runs
void comm_forget_prebegun(storm_hip_ctx *c);  // at the ends of a solve
int comm_halo_exchange_begin_formed(const storm_hip_op *op, int mode, const double *r, const double *p, const double *v,
                                    const double *sa, const double *sb, double *target, double *alpha_seen = nullptr);
void comm_destroy(storm_hip_ctx *c);
int comm_check_error(storm_hip_ctx *c);  // a bounded wait of the peer-window transport gave up
struct IpcDev;                            // ipc_device.hpp
struct IpcSendPlan;
struct IpcRecvPlan;
bool comm_ipc_next(storm_hip_ctx *c, IpcDev *w);  // the window view for a kernel that all-reduces itself
bool comm_is_ipc(const storm_hip_ctx *c);
long long comm_ipc_stat(storm_hip_ctx *c, int k);
int comm_profile_reset(storm_hip_ctx *c);           // option profile_comm (RCCL transport)
long long comm_profile_read(storm_hip_ctx *c, int k);
int comm_ipc_exchange(const storm_hip_op *op, IpcDev *w, IpcSendPlan *sp, IpcRecvPlan *rp);
int comm_ipc_send(const storm_hip_op *op, const double *x, const IpcDev &w, const IpcSendPlan &sp);
int comm_ipc_recv_copy(const storm_hip_op *op, double *x, const IpcDev &w, const IpcRecvPlan &rp);

}  // namespace storm
