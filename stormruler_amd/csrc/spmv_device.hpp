// Shared by the translation units of the operator apply (spmv.hip and spmv_*.hip): record constants, kernel argument
// structs, the small device helpers every format's kernel uses, and the launch interface between the dispatch in
// spmv.hip and the per-format units.  DESIGN.md section 3 lists the formats, NOTES.md ("Source layout and dispatch of the apply") who owns which.
#pragma once
#include <hip/hip_ext.h>

#include "common.hpp"
#include "ticket_device.hpp"
#include "ipc_device.hpp"
#include "wave_device.hpp"

namespace storm {

__device__ __forceinline__ double ld_scal2(const Scal &s) { return s.p ? (*s.p) * s.sign : s.v; }

typedef int int2v __attribute__((ext_vector_type(2)));
typedef double double2v __attribute__((ext_vector_type(2)));

constexpr int kExtBytes = kWave * 8;      // 512
constexpr int kSlotBytes = kWave * 12;    // 768: one ELL slot of a slice (64 cols + 64 vals)

struct SellArgs {
  const char *__restrict__ pack;          // slice records
  const int64_t *__restrict__ slice_off;  // [n_slices + 1] byte offsets
  int64_t n_rows;
  int uniform_width;                      // > 0: every slice has this width, slice_off is not read
  int xcd_group;                          // tiles per XCD run (<= 1: one contiguous run per XCD)
  const double *__restrict__ dict;        // VARIANT 2: the 256-entry value dictionary
  int dict_size;
  const int *__restrict__ offs;           // format 2: the column-offset dictionary
  int offs_size;
  int accumulate;                         // y += alpha*M(x) (stormDivGrad's own form) instead of y = beta*x + alpha*M(x)
  int rec_by_pos = 0;                     // paired records stored in slice-LIST order (the boundary groups of a mixed operator)
};

constexpr int kDictSize = 256;
constexpr int kPairRecBytes = 2 * kWave * 8 + kWave * 8;  // format 3: 64 x (u64, u64) weights + 64 x u64 offsets per 128 rows
constexpr int kColSlotBytes = kWave * 4;  // 256: one slot of a value-dictionary record (columns only)

struct DotArgs {
  const double *w;   // partial of <w, y>, may be null
  double *partials;  // [<w,y> per block | <y,y> per block]
  int yy;
  int nblocks_total;  // stride between the two partial arrays
  int block_offset;   // where this launch's blocks start
  // tickets != null (format-4 / 5 kernel, unsplit launch): the reduction finishes in the kernel (ticket_device.hpp);
  // `partials` then holds one partial per BLOCK, part2 the groups' sums, and the totals go to out0 / out1
  int *tickets = nullptr;
  double *part2 = nullptr;
  double *out0 = nullptr, *out1 = nullptr;
};

// Peer-window transport, fused form (comm.hip, ipc_device.hpp): the interior launch SENDS this rank's rows (its first
// blocks store them into the neighbours' windows), the boundary launch READS the halo rows straight from this rank's
// window (polling each value's tag) and its last block acknowledges -- a partitioned apply is two launches on one
// stream, no pack / flag / receive / acknowledge kernels, no cross-stream events.
struct IpcFused {
  IpcDev w;
  IpcSendPlan sp;
  IpcRecvPlan rp;
};
struct IpcSendArgs {
  IpcDev w;
  IpcSendPlan sp;  // sp.n_blocks == 0: nothing to send
};
struct IpcRecvArgs {
  IpcDev w;
  IpcRecvPlan rp;
  int n_halo;      // halo rows of the operator (columns n_rows .. n_rows + n_halo)
};

// Blocks are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8), each with a private
// 4 MiB L2.  The remap gives every XCD one contiguous run of slices (neighbour rows of x then
// hit that XCD's L2).  Measured on the 256^3 problem it LOSES 7 %: eight XCDs walking eight
// distant regions means 8x the concurrent DRAM streams, and the x re-reads it avoids are served
// by the 256 MiB Infinity Cache anyway.  Kept as an option (spmv_xcd_remap), off by default.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int q = nb / kNumXcd, r = nb % kNumXcd;
  const int x = b % kNumXcd, j = b / kNumXcd;
  return x * q + (x < r ? x : r) + j;
}
// Grouped remap: XCD x takes runs of G consecutive tiles, the 8 XCDs' runs interleaved.  All XCDs
// then stream one shared window of 8 G tiles (few DRAM streams, like a plain copy) while most
// neighbour rows of a tile are processed by -- and cached in the L2 of -- the same XCD.
__device__ __forceinline__ int xcd_remap_grouped(int b, int nb, int G) {
  const int span = kNumXcd * G;
  if (b >= (nb / span) * span) return b;  // ragged tail: identity
  const int x = b % kNumXcd, j = b / kNumXcd;
  return ((j / G) * kNumXcd + x) * G + (j % G);
}

template <bool NT>
__device__ __forceinline__ int ld_i(const int *p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ double ld_d(const double *p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// Format 4 ("canonical" paired rows): a format-3 operator whose rows all list their neighbours in ONE common
// order of at most 7 offsets col - row (a structured box in its natural ordering: -nx*ny, -nx, -1, +1, +nx, +nx*ny).
// The offsets are kernel arguments (SGPRs) instead of per-lane bytes, so a 128-row group is 64 x (weights of row
// 2p : u64, of row 2p+1 : u64) = 8 B/row, and slot k means the same neighbour in every lane:
//   * the slots of offsets -1 and +1 (template M1, M1 + 1) need no load at all -- x[2p-1] is the left lane's
//     xi.y, x[2p+2] the right lane's xi.x (two DPP moves each); only lanes 0 and 63 load their outer neighbour,
//     one 8-byte load instruction with two active lanes;
//   * a row that lacks a neighbour carries weight 0 in that slot and gathers from a CLAMPED address (the value
//     is multiplied by 0; x finite is the precondition of storm_hip_op_apply).
// 8 + 8 + 8 = 24 B/row and 8 vector-memory instructions per row pair (format 3: 28 B/row and 10).  The sums run
// over the slots in the common order = every row's own face order, with exactly the bit patterns of the other
// formats: results are bit-identical (tests/test_gpu_formats.py).
constexpr int kCanonRecBytes = 2 * kWave * 8;
struct CanonArgs {
  int off[7];
  int max_gather;  // largest guard-relative index a 16-byte gather may start at
  int reverse;     // deal the tiles out from the far end (the solver's sweep-direction scheme; same tile per block index)
  int xcd_shift;   // >= 0: the XCD grouping with runs of 2^xcd_shift tiles, by shifts (no integer division per block)
  int xcd_full;    // ... applied to blocks below this index (a multiple of 8 * 2^xcd_shift), identity beyond
};
template <int CTRL>
__device__ __forceinline__ double dpp_shift(double v) {  // lanes without a source get 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// G: consecutive 128-row groups per wavefront (1 or 2).  With two, every load of both groups is in flight before the
// first use, the prologue (tile mapping, tables) and the fused-dot's wave reduction are paid once per 256 rows.

// ---- format 4, tiled --------------------------------------------------------------------------------------------
// What separates spmv_canon_kernel from a plain 2-read-1-write stream is its gathers: four 16-byte L2 -> L1
// transactions per row pair (offsets -b, -a, +a, +b), as many as the whole HBM stream.  When the common offsets are
// (-b, -a, -1, +1, +a, +b) -- a lattice: a = rows per line, b = rows per plane -- a block takes a TILE instead of 1024
// consecutive rows: kTileRun = 1024 consecutive rows of a plane, in TZ consecutive planes.
//   * the +-b neighbours of a row are the SAME LANE's own rows in the planes above and below: registers (only the two
//     outer planes of a tile are gathered: 2 / TZ per row);
//   * the +-a and +-1 neighbours come from an LDS copy of the tile's x, [TZ][a + 1024 + a] doubles: every wave writes
//     its own rows there, and the 2 a halo rows per plane are fetched once per tile by all 256 threads together
//     (a / 64 16-byte loads per thread);
// Per 1024-row line and wave: TZ x (2 own + 2 record) + 4 + a / 64 vector loads instead of TZ x 14.  Arithmetic,
// operand bit patterns and summation order per row are those of spmv_canon_kernel: y is bit-identical.
// XCD map: the tiles of a plane are dealt to the 8 XCDs in contiguous runs (tile yt -> XCD yt / (tiles per plane / 8)),
// chunk after chunk of planes, so that a tile's outer planes and lines were (or will be) some tile's OWN rows on the
// same XCD's L2.
constexpr int kTileRun = 4 * 4 * kWave;  // rows of a plane per tile: 4 waves x 2 groups x 128 rows
struct CanonTileArgs {
  int a, b;              // the lattice offsets (both even, 2 <= a <= 512, b >= 2 a)
  unsigned a_magic;      // ceil(2^32 / a): h / a = umulhi(h, a_magic) for h < 4096
  int tiles_per_plane;   // ceil(b / kTileRun)
  int per_xcd;           // tiles_per_plane / 8 when that divides, else 0 (plain order)
  int max_gather;        // largest guard-relative index a 16-byte gather may start at
  int reverse;
  int plane0, plane_end; // the planes this launch covers (a partitioned operator: those that read no halo column)
};
// FUSE (fused CG loop, one rank): the kernel first performs the END of the previous CG iteration on everything it loads,
//     x += alpha p,   p' = r + beta p                                  (SolverCg.hpp:98, :123)
// and then applies the operator to p' -- x and p are not streamed by a kernel of their own any more (56 instead of
// 24 + 40 B/row).  x of a row is its own lane's; p' of the tile's halo rows and outer planes is formed from THEIR r
// and p with the owner's expression (the same bits), which is why p' goes to a SECOND vector (F.p_out): another tile
// may still need this tile's old p.  Gated like cg_xp_kernel: on the iteration counter for x (the converging
// iteration's update must land), on `done` for p' and the apply.
struct CgFuseArgs {
  const long long *iteration;  // SolverState::iteration
  long long my_iteration;      // the update belongs to iteration my_iteration - 1: it ran iff *iteration >= my_iteration
  const double *ca, *cb;       // alpha, beta of that iteration (device slab)
  double *x;                   // null (marching kernel only): no x update -- the kernel forms p' = r + c p and applies the
                               // operator to it, nothing else (BiCGStab's s = r - alpha v; t = A s)
  const double *r;
  double *p_out;
  double ca_imm, cb_imm;       // marching kernel only: alpha, beta themselves where ca / cb are null (a HOST loop's step: lazy.hip)
};

// ---- the fused CG step, marching in z ------------------------------------------------------------------------------
// spmv_canon_tile_kernel<FUSE> forms p' = r + beta p for its tile's halo rows and outer planes from THEIR r and p: with
// tiles two planes deep that is one extra row of r and p per row, and with ~128 tiles per XCD in flight those rows no
// longer come from the L2 (PMC: 48 instead of 32 B/row fetched, profiles/r03i_pmc_summary.txt).  Here a block keeps
// its 1024 rows of the plane and MARCHES through `zc_planes` planes: p' of the planes below, at and above the one
// being applied sits in the lane's registers (each plane's p, r, x, record are loaded exactly once, prefetched one
// plane ahead), the +-a / +-1 neighbours come from an LDS copy of the current plane (three buffers in rotation, one
// barrier per plane), and only the two planes bounding the block's chunk are loaded for their p' alone.
//   reads  p, r, x, records (32 B/row) + the +-a halo lines (r, p; adjacent tiles of the same XCD march in step) + 2 / zc_planes planes
//   writes x, p', z (24 B/row)
// Arithmetic per row exactly spmv_canon_kernel's; x += alpha p and p' = r + beta p exactly cg_xp_kernel's.
struct MarchArgs {
  CanonTileArgs T;   // a, b, tiles_per_plane, per_xcd, max_gather, reverse, plane_end (= number of planes)
  int zc_planes;     // planes per block
  int apply_begin, apply_end;  // planes the operator is applied to (a partitioned operator: those that read no halo
                               // column -- the others get x and p' here and their z from the boundary launch)
  int alternate;     // odd chunks march DOWN: two z-adjacent chunks of a tile (co-resident on one XCD, 8 block slots apart)
                     // then touch the two planes they share at the same moment -- both at the start or both at the end of
                     // their marches -- instead of a whole march apart, and the second reader finds them in the L2 /
                     // Infinity Cache instead of HBM (the z-halo planes were most of the kernel's 8.7 % over-fetch)
};

// ---- the launch interface ------------------------------------------------------------------------------------------
// One kernel launch of the apply over all slices (slice_list == nullptr) or over one of a partitioned operator's
// two lists.  Filled by launch_range (spmv.hip), consumed by the per-format units.
struct RangeLaunch {
  const storm_hip_op *op;
  int nb;  // blocks (launch_range computes it with the geometry helpers below, so that partial slots match)
  Scal alpha, beta;
  const double *x;
  double *y;
  const int *slice_list;
  int64_t n_launch;
  DotArgs dot;
  bool want_dot;
  const int *done;
  hipEvent_t ev0, ev1;  // kernel-begin / kernel-end events of the profile option (or null)
  bool accumulate;
  const IpcFused *fused;      // peer-window transport: the interior launch sends, the boundary launch reads the window
  const CgFuseArgs *cg_fuse;  // the fused CG step (tiled form)
};
// A format's launcher returns false when the launch is not its to take (the dispatch table in spmv.hip tries them
// in order), true once the kernel is enqueued.
bool spmv_tile_run(const RangeLaunch &L);   // spmv_lattice.hip: format 4 on a lattice, tiles of 1024 rows x TZ planes
bool spmv_canon_run(const RangeLaunch &L);  // spmv_pair.hip: formats 4, 5 (common offsets as kernel arguments)
bool spmv_pair_run(const RangeLaunch &L);   // spmv_pair.hip: format 3 (paired rows, per-lane offsets)
bool spmv_dict_run(const RangeLaunch &L);   // spmv_dict.hip: formats 1, 2 of one uniform width
bool spmv_sell_run(const RangeLaunch &L);   // spmv_sell.hip: sliced-ELL records (fp64, or dictionary records of mixed width)
int spmv_tail_run(const storm_hip_op *op, Scal alpha, const double *x, double *y, const int *done);  // spmv_sell.hip: CSR tail
// spmv_lattice.hip: the z-marching fused CG step (its own grid: n_blocks marching blocks + the sending blocks of S)
int spmv_march_run(const storm_hip_op *op, const MarchArgs &M, int n_blocks, Scal alpha, Scal beta, const double *x, double *y,
                   const DotArgs &dot, const int *done, const CgFuseArgs &cgf, const IpcSendArgs &S);  // (cgf.x == null: no x update)
// spmv.hip: diagonal of beta I + alpha M; spmv_build.hip calls nothing of the kernels.

// ---- geometry shared by dispatch and launchers -----------------------------------------------------------------------
// 128-row groups per wave of the format-4 / 5 kernel.
static inline int canon_groups(const storm_hip_op *op) { return op->ctx->opt_spmv_canon_groups == 2 ? 2 : 1; }
// Slices per wave: SPW for the uniform-width value-dictionary kernel, 1 otherwise.
static inline int op_spw(const storm_hip_op *op) {
  if (op->pair) return 1;  // a "slice" of a format-3 operator is a 128-row group, one per wave
  return (op->dict_size > 0 && op->uniform_width > 0) ? (int)op->spw : 1;
}
static inline int blocks_for(const storm_hip_op *op, int64_t n_launch_slices, bool boundary_of_mixed = false) {
  const int64_t per_block = (kBlock / kWave) * ((op->pair >= 2 && !boundary_of_mixed) ? canon_groups(op) : op_spw(op));
  return (int)((n_launch_slices + per_block - 1) / per_block);
}
// the launch over the boundary groups of a mixed operator (format-3 records of their own, in list order)
static inline bool boundary_of_mixed(const RangeLaunch &L) {
  return L.op->d_bnd_pack != nullptr && L.slice_list != nullptr && L.slice_list == L.op->d_boundary;
}
// SellArgs of a paired-row launch (formats 3, 4, 5)
static inline SellArgs paired_args(const RangeLaunch &L, int *width) {
  const storm_hip_op *op = L.op;
  // the interior list of a partitioned operator is consecutive but for a few gaps: the XCD grouping still pays there
  const int group = (L.slice_list == nullptr || L.slice_list == op->d_interior) ? (int)op->ctx->opt_spmv_xcd_remap : 0;
  SellArgs A{op->d_pack, op->d_slice_off, op->n_rows, op->uniform_width, group, op->d_dict, op->dict_size,
             op->d_offs, op->offs_size, (int)L.accumulate};
  *width = op->uniform_width;
  if (boundary_of_mixed(L)) {  // the groups that read halo columns: format-3 records of their own, in list order
    A.pack = op->d_bnd_pack, A.rec_by_pos = 1;
    *width = op->bnd_width;
  }
  return A;
}
// spmv_lattice.hip
int canon_tile_planes(const storm_hip_op *op);
bool canon_tile_geometry(const storm_hip_op *op, CanonTileArgs *T, int *n_blocks, bool interior = false);
bool cg_march_geometry(const storm_hip_op *op, MarchArgs *M, int *n_blocks, bool partitioned = false);

}  // namespace storm
