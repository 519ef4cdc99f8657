// The resident path: CG and BiCGStab for LATTICE operators (format-4 records, spmv.hip) of up to ~3 M rows as ONE
// persistent kernel per solve in which every block OWNS a box of the lattice.
//
// BASELINE configs 4 and 5 (and every mid-size user) live at 128^3 = 2.1 M rows.  There the throughput path
// (solvers.hip) is three dependent launches of ~13 us each per CG iteration -- launch- and tail-bound although the
// whole working set sits in the Infinity Cache -- and the latency path (latency.hip), which keeps its vectors in
// registers, publishes EVERY row of r and p with returning atomics and gathers six neighbours per row through the
// L2s: 235 MB of atomic / L2 traffic per iteration at 128^3, slower than the launches beyond 100^3.  On a lattice
// (common offsets -b, -a, -1, +1, +a, +b: a = rows per line, b = rows per plane; nothing else about the mesh is
// assumed, absent neighbours carry weight 0) neither is necessary:
//
//   * a block owns the rows [s0, s0 + 1024) of TZ consecutive planes for the whole solve (SolverCg.hpp:54-126 /
//     SolverBiCgStab.hpp:60-167 inside Solver.hpp:116-147); one thread owns one PAIR of rows per plane and keeps
//     x, r, p (CG: + z and the weight words; BiCGStab: + v, rt, t) of its 2 TZ rows in REGISTERS -- 8 waves per CU,
//     256 registers per lane;
//   * the +-b neighbours of a row are the same thread's own registers, the +-a / +-1 ones come from an LDS copy of
//     the block's rows of the vector the operator is applied to, [TZ][a + 1024 + a] doubles (the tile kernel's layout);
//   * what crosses blocks is only the SURFACE of the box: the first and last a rows of a block's run in every plane,
//     and its first and last plane.  A block publishes those rows of p (BiCGStab: of p, then of s) as
//     self-validating 16-byte granules { low half | tag }, { high half | tag } (tag = the exchange's sequence number)
//     with one write-through store each, and polls its neighbours' granules until both tags of each are current: no
//     flag follows the data, no ordering between stores is relied on, a row that was never published never validates
//     (bounded wait -> the fallback of latency.hip).  128^3: 4 096 of a block's 8 192 rows, 64 KB out and in per
//     block and exchange, 16 MB chip-wide -- against the 168 MB a throughput iteration streams;
//   * a buffer row is overwritten only behind an all-reduce that every block enters after its last read of it;
//   * the reductions are the tagged-slot all-reduces of the latency path (each is its own grid barrier): two per CG
//     iteration, three per BiCGStab iteration.
//
// Arithmetic per row exactly spmv_canon_tile_kernel's (same operands, same order, the same two FMAs at the end): the
// operator values are bit-identical to the throughput kernels'; dot products group their terms differently
// (rounding-level differences, as on the latency path).
//
// Taken by storm_hip_solve_cg / storm_hip_solve_bicgstab when the operator qualifies (res_geometry below), the
// context has no communicator and option `resident_path` != 0.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.hpp"
#include "coop_device.hpp"
#include "solver_device.hpp"

namespace storm {

constexpr int kResThreads = 512;                // 8 waves: two per SIMD, 256 registers per lane
constexpr int kResWaves = kResThreads / kWave;
constexpr int kResRun = 2 * kResThreads;        // rows of a plane per block: one pair per thread
constexpr int kResMaxPlanes = 12;               // planes per block, at most (registers)
constexpr bool kResApplyCacheCg(int tz) { return tz >= 3 && tz <= 12; }  // (res_apply CACHE: where the registers allow it)
constexpr int kResMaxPlanesBicgEarly = 6;       // ... of its early-publish form
constexpr int kResMaxPlanesBicg = 8;            // ... of the BiCGStab kernel (r, p, v and the result of an apply: 227 registers at 8 planes)

typedef double double2r __attribute__((ext_vector_type(2)));
typedef unsigned long long u64x2r __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));

struct ResArgs {
  const char *pack;    // format-4 records: the 8-byte weight word of row i at pack + 8 i
  const double *dict;  // the value table (<= 32 entries; the words' bytes are byte offsets into it)
  int a, b;            // the lattice offsets
  int nplanes, nsec;   // planes of b rows; runs of kResRun rows per plane
  int64_t n_rows;
  double alpha, beta;  // A = beta I + alpha M
  const double *rhs;
  double *x;
  double *rt;          // BiCGStab: a work vector for the shadow residual
  char *exch;          // the exchange buffer: one 16-byte granule per row -- the even rows' granules, then (exch_half bytes on) the odd
                       // rows': a wave's pairs of rows go out, and come in, as two runs of 1 KiB (whole lines) instead of 64 half-filled ones
  size_t exch_half;
  size_t exch_stride;   // a second exchange buffer this many bytes on (BiCGStab's early publish: one per travelling vector)
  char *slots;         // all-reduce slots, kLatSlotStride bytes per (block, parity)
  char *dense;         // ... or (non-null) dense value-major granules: co_allreduce_dense, coop_device.hpp
  int *gave_up;        // the latency path's flag (lat_check_gave_up)
  long long *prof;     // option resident_profile: [gridDim.x][8] ticks of the 100 MHz counter per phase of the loop, summed over the solve
  unsigned long long *cnt;  // [0] all-reduce sequence number, [1] exchange sequence number: carried from solve to solve
  SolverState *st;
  int halo_interleave;  // CG, deeper boxes: half the waves form the halo of p' before the update of the own rows (option resident_halo_interleave)
  int apply_cache;    // res_apply CACHE on (option resident_apply_cache; 0: every plane decodes its coefficients -- the A/B of the tests)
  int early_publish;  // CG: the residual's surface before the second all-reduce (res_halo MODE 2); BiCGStab: res_bicgstab_early_kernel
};

// ---- all-reduce over the co-resident grid (latency.hip's scheme for 8 waves per block) --------------------------
// Two halves: ARRIVE (the block's sums folded and stored to its slot) and WAIT (every block's slot polled, the same tree in
// every block) -- a caller may put stores of its own between them (CG's early publish: behind the block's slot in the
// memory pipeline, not in front of it, or every block's arrival is late by the time those stores take to drain).
template <int NV>
__device__ __forceinline__ void res_allreduce_arrive(const double (&s)[NV], const ResArgs &A, unsigned long long seq, double *lds) {
  const unsigned tag = (unsigned)seq;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  double v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = lat_wave_sum(s[j]);
  __syncthreads();  // (lds may still be read by the previous call)
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) lds[j * kResWaves + wave] = v[j];
  }
  __syncthreads();
  if ((int)threadIdx.x < NV) {  // thread j folds and stores sum j
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kResWaves; ++w) t += lds[threadIdx.x * kResWaves + w];
    co_store_slot(A.slots + lat_slot_offset(blockIdx.x, seq) + 16 * threadIdx.x, tag, t);
  }
}
template <int NV>
__device__ __forceinline__ void res_allreduce_wait(double (&s)[NV], const ResArgs &A, unsigned long long seq, double *lds) {
  const unsigned tag = (unsigned)seq;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  double v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = 0.0;
  if (threadIdx.x < gridDim.x) {  // gridDim.x <= 256: thread t (waves 0 .. 3) watches block t
    const char *slot = A.slots + lat_slot_offset(threadIdx.x, seq);
    const long long t0 = wall_clock64();
    for (int spins = 0;; ++spins) {
      bool ok;
      if (NV == 1) ok = co_load_slot(slot, tag, &v[0]);
      else ok = co_load_slot2(slot, tag, &v[0], &v[NV - 1]);
      if (ok) break;
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 1023) == 1023 &&
          (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(A.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(A.gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = 0.0;
        break;
      }
    }
  }
  // slot order: lanes, then the four polling waves -- the same tree in every block
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = lat_wave_sum(v[j]);
  __syncthreads();
  if (lane == 0 && wave < 4) {
#pragma unroll
    for (int j = 0; j < NV; ++j) lds[j * kResWaves + wave] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j)
    s[j] = (lds[j * kResWaves] + lds[j * kResWaves + 1]) + (lds[j * kResWaves + 2] + lds[j * kResWaves + 3]);
}
template <int NV>
__device__ __forceinline__ void res_allreduce(double (&s)[NV], const ResArgs &A, unsigned long long seq, double *lds) {
  if (A.dense) {  // one 16-byte load per thread and poll, whole lines
    co_allreduce_dense<NV, kResWaves>(s, A.dense, A.gave_up, seq, lds);
    return;
  }
  res_allreduce_arrive<NV>(s, A, seq, lds);
  res_allreduce_wait<NV>(s, A, seq, lds);
}

// ---- the exchange: self-validating granules ---------------------------------------------------------------------
__device__ __forceinline__ u32x4r res_granule(double v, unsigned tag) {
  return u32x4r{(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
}
// (s_nop 1: a store of more than 8 bytes must be two wait states ahead of a VALU write to its data registers on gfx940+; the
//  compiler keeps that distance for its own stores but does not look inside an asm statement -- round 4 found
//  `v_mov_b64 v[8:9], -2` ONE wait state behind this store in res_bicgstab_early_kernel<1>: the granule went out with a
//  clobbered upper half, its tag never validated and the reader timed out.  The same pattern sat in res_cg_kernel<12>, and
//  it is the likely cause of the "hang that any change to a cold path cures" met in CG's early publish.)
__device__ __forceinline__ void res_store16(char *p, u32x4r w) {  // one write-through store, each 8-byte half single-copy atomic
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ bool res_tag_ok(u32x4r w, unsigned tag) { return w.y == tag && w.w == tag; }
__device__ __forceinline__ double res_value(u32x4r w) { return __hiloint2double((int)w.z, (int)w.x); }

// The pairs at rows ra and rb (even; a pair outside [0, n) reads as zeros) as published with `tag`: four coherent
// 16-byte loads in flight, repeated until every needed granule carries the tag.  Bounded like the all-reduce.
__device__ __forceinline__ void res_fetch2(const ResArgs &A, int64_t ra, int64_t rb, unsigned tag, double2r *va, double2r *vb) {
  const int64_t n = A.n_rows;
  const bool a0 = ra >= 0 && ra < n, a1 = ra >= 0 && ra + 1 < n, b0 = rb >= 0 && rb < n, b1 = rb >= 0 && rb + 1 < n;
  const unsigned oa = a0 ? (unsigned)ra << 3 : 0u, ob = b0 ? (unsigned)rb << 3 : 0u;  // (even rows: granule r / 2, 16 bytes each)
  const char *even = A.exch, *odd = A.exch + A.exch_half;
  *va = double2r{0.0, 0.0}, *vb = double2r{0.0, 0.0};
  if (!(a0 || b0)) return;
  const long long t0 = wall_clock64();
  for (int spins = 0;; ++spins) {
    u32x4r wa0, wa1, wb0, wb1;
    asm volatile(
        "s_nop 4\n\t"  // (the bases may come straight out of a v_readlane -- a spilt SGPR --: five wait states before a VMEM reads it, which the compiler does not count for an asm statement)
        "global_load_dwordx4 %0, %4, %6 sc1\n\tglobal_load_dwordx4 %1, %4, %7 sc1\n\t"
        "global_load_dwordx4 %2, %5, %6 sc1\n\tglobal_load_dwordx4 %3, %5, %7 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(wa0), "=&v"(wa1), "=&v"(wb0), "=&v"(wb1)
        : "v"(oa), "v"(ob), "s"(even), "s"(odd)
        : "memory");
    if ((!a0 || res_tag_ok(wa0, tag)) && (!a1 || res_tag_ok(wa1, tag)) && (!b0 || res_tag_ok(wb0, tag)) &&
        (!b1 || res_tag_ok(wb1, tag))) {
      va->x = a0 ? res_value(wa0) : 0.0, va->y = a1 ? res_value(wa1) : 0.0;
      vb->x = b0 ? res_value(wb0) : 0.0, vb->y = b1 ? res_value(wb1) : 0.0;
      return;
    }
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 1023) == 1023 &&
        (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(A.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      __hip_atomic_store(A.gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
}

// ... four pairs, eight loads in flight: ONE round trip for a thread's share of the surface in the common geometries.
__device__ __forceinline__ void res_fetch4(const ResArgs &A, const int64_t (&row)[4], unsigned tag, double2r (&v)[4]) {
  const int64_t n = A.n_rows;
  bool h0[4], h1[4], any = false;
  unsigned o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h0[i] = row[i] >= 0 && row[i] < n, h1[i] = row[i] >= 0 && row[i] + 1 < n;
    o[i] = h0[i] ? (unsigned)row[i] << 3 : 0u;
    v[i] = double2r{0.0, 0.0};
    any |= h0[i];
  }
  if (!any) return;
  const char *even = A.exch, *odd = A.exch + A.exch_half;
  const long long t0 = wall_clock64();
  for (int spins = 0;; ++spins) {
    u32x4r w0[4], w1[4];
    asm volatile(
        "s_nop 4\n\t"  // (as in res_fetch2)
        "global_load_dwordx4 %0, %8, %12 sc1\n\tglobal_load_dwordx4 %1, %8, %13 sc1\n\t"
        "global_load_dwordx4 %2, %9, %12 sc1\n\tglobal_load_dwordx4 %3, %9, %13 sc1\n\t"
        "global_load_dwordx4 %4, %10, %12 sc1\n\tglobal_load_dwordx4 %5, %10, %13 sc1\n\t"
        "global_load_dwordx4 %6, %11, %12 sc1\n\tglobal_load_dwordx4 %7, %11, %13 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(w0[0]), "=&v"(w1[0]), "=&v"(w0[1]), "=&v"(w1[1]), "=&v"(w0[2]), "=&v"(w1[2]), "=&v"(w0[3]), "=&v"(w1[3])
        : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "s"(even), "s"(odd)
        : "memory");
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) ok = ok && (!h0[i] || res_tag_ok(w0[i], tag)) && (!h1[i] || res_tag_ok(w1[i], tag));
    if (ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i].x = h0[i] ? res_value(w0[i]) : 0.0, v[i].y = h1[i] ? res_value(w1[i]) : 0.0;
      return;
    }
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 1023) == 1023 &&
        (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(A.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      __hip_atomic_store(A.gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
}

// What a block is, derived from blockIdx alone.
struct ResBox {
  int a, b, ldw;      // lattice offsets; doubles per plane of the LDS copy
  int s0, L;          // the block's run of each plane: rows [s0, s0 + L) of the plane (L even, <= kResRun)
  int z0, tzl;        // its planes [z0, z0 + tzl)
  int nplanes;
  unsigned g0;        // row A of this thread's pair in plane z0 (row B = g0 + 1); rows < 2^28 (a condition of the paired formats)
  int tid2;           // 2 * threadIdx.x: the pair's place in the run
  unsigned mask_a, mask_b;  // bit t: row A / row B of plane z0 + t exists
  bool edge_y;        // this thread's pair lies within a rows of an end of the run
};
template <int TZ>
__device__ __forceinline__ ResBox res_box(const ResArgs &A) {
  ResBox B;
  const int tid = threadIdx.x;
  B.a = A.a, B.b = A.b, B.ldw = kResRun + 2 * A.a, B.nplanes = A.nplanes;
  const int sec = (int)blockIdx.x % A.nsec, zc = (int)blockIdx.x / A.nsec;
  B.s0 = sec * kResRun;
  B.L = min(kResRun, A.b - B.s0);
  B.z0 = zc * TZ;
  B.tzl = min(TZ, A.nplanes - B.z0);
  B.g0 = (unsigned)B.z0 * (unsigned)A.b + (unsigned)(B.s0 + 2 * tid);
  B.tid2 = 2 * tid;
  B.mask_a = B.mask_b = 0u;
#pragma unroll
  for (int t = 0; t < TZ; ++t) {
    const int64_t g = (int64_t)B.g0 + (int64_t)t * A.b;
    const bool in = 2 * tid < B.L && t < B.tzl;
    B.mask_a |= (unsigned)(in && g < A.n_rows) << t;
    B.mask_b |= (unsigned)(in && g + 1 < A.n_rows) << t;
  }
  B.edge_y = 2 * tid < A.a || 2 * tid + A.a >= B.L;
  return B;
}
// The per-thread bases, opaque to the optimiser: called at the top of every pass of a solver's loop so that the 2 TZ
// global and TZ LDS addresses of the thread's rows are RE-FORMED where they are used (an add each) instead of being
// hoisted out of the loop into registers for the whole solve (88 of them at TZ = 8).
__device__ __forceinline__ ResBox res_fresh(ResBox B) {
  asm volatile("" : "+v"(B.g0), "+v"(B.tid2));
  return B;
}
// byte offset of the thread's pair in plane z0 + t within a vector of doubles (a 32-bit offset from a uniform base:
// one register per address)
__device__ __forceinline__ unsigned res_off8(const ResBox &B, int t) { return (B.g0 + (unsigned)t * (unsigned)B.b) << 3; }
__device__ __forceinline__ double2r res_ld_pair(const double *vec, unsigned off8) {
  return *reinterpret_cast<const double2r *>(reinterpret_cast<const char *>(vec) + off8);
}
__device__ __forceinline__ void res_st_pair(double *vec, unsigned off8, double2r v, bool has_a, bool has_b) {
  if (has_b) *reinterpret_cast<double2r *>(reinterpret_cast<char *>(vec) + off8) = v;
  else if (has_a) *reinterpret_cast<double *>(reinterpret_cast<char *>(vec) + off8) = v.x;
}

// Publish the pair of plane t if it lies on the box's surface (first / last a rows of the run, first / last plane).
__device__ __forceinline__ void res_publish_pair(const ResArgs &A, const ResBox &B, int t, double2r v, unsigned tag) {
  if (((B.mask_a >> t) & 1u) && (B.edge_y || t == 0 || t == B.tzl - 1)) {
    char *e = A.exch + (size_t)res_off8(B, t);  // (row g0 is even: its granule is number g0 / 2 of the even rows')
    res_store16(e, res_granule(v.x, tag));
    if ((B.mask_b >> t) & 1u) res_store16(e + A.exch_half, res_granule(v.y, tag));
  }
}

__device__ __forceinline__ double2r res_cg_direction(double2r r, double2r p, double beta) {
  return double2r{__builtin_fma(beta, p.x, r.x), __builtin_fma(beta, p.y, r.y)};  // r + beta p            SolverCg.hpp:123
}

// The halo of the LDS copy and the two planes bounding the box:
//   MODE 0: straight from a vector in memory that no block writes meanwhile (the start vector, at init);
//   MODE 1: the granules published with `tag`.
// (Round 4, first half: CG publishing the surface of its RESIDUAL before the all-reduce that yields beta showed run-to-run
//  differences at the 1e-10 level and was dropped "not understood".  Understood in the second half: the tag of an early
//  publish was only counted when the solve went on -- the LAST residual of a solve stayed in the buffer under a tag that
//  the NEXT solve's first exchange used again, and a fast reader took it for the new surface.  Every publish now takes a
//  fresh tag, consumed or not, and MODE 2 below is bitwise the late publish (tests/test_gpu_resident.py).  One more trap:
//  with the publish INSIDE the loop that updates r the build hung in every box with neighbours above and below (and
//  stopped hanging with any change to the polling loops' cold paths -- a debugging printf was enough); with a loop of
//  its own behind the update it does not (the likely cause, found later: res_store16's store-data hazard, see there).  Measured, us per iteration late / early: 16^3 5.7 / 5.3, 32^3 6.3 / 6.0,
//  64^3 9.8 / 9.5, 100^3 14.4 / 14.8, 128^3 16.7 / 17.9 -- since the granules travel as whole lines the wait it hides is
//  short, and forming the halo of p' costs more than it at the larger boxes: option resident_early, off.)
//   MODE 2 (CG): the granules hold the neighbours' RESIDUAL, published before the all-reduce that yields beta; the halo of
//   the direction is formed here, p' = r + beta p, from it and the halo of the old direction that the LDS copy / lo / hi
//   still hold -- the owner's own expression on the owner's own operands: the same bits.
template <int TZ, int MODE>
__device__ __forceinline__ void res_halo(const ResArgs &A, const ResBox &B, double *P, unsigned tag, const double *vec,
                                         double2r *lo, double2r *hi, double beta = 0.0) {
  const int tid = threadIdx.x;
  const int64_t n = A.n_rows;
  auto plain = [&](int64_t row) -> double2r {
    double2r v{0.0, 0.0};
    if (row >= 0 && row < n) v.x = vec[row];
    if (row >= 0 && row + 1 < n) v.y = vec[row + 1];
    return v;
  };
  // pair hh of the a pairs per plane (the a rows below the run and the a rows above it): its row and its place
  const int nh = B.tzl * B.a;
  auto place = [&](int hh, int64_t *row, int *at) {
    const int t = hh / B.a, j2 = 2 * (hh - t * B.a);
    const bool lower = j2 < B.a;
    const int jj = lower ? j2 : j2 - B.a;
    *at = hh < nh ? t * B.ldw + (lower ? jj : B.a + B.L + jj) : -1;
    *row = hh < nh ? (int64_t)(B.z0 + t) * B.b + B.s0 + (lower ? jj - B.a : B.L + jj) : -2;
  };
  auto put = [&](int at, double2r v) {
    if (at < 0) return;
    if (MODE == 2) v = res_cg_direction(v, *reinterpret_cast<const double2r *>(&P[at]), beta);
    *reinterpret_cast<double2r *>(&P[at]) = v;
  };
  // first batch: this thread's own column in the planes below and above, and its first two halo pairs
  const bool in = 2 * tid < B.L;
  int64_t row[4];
  int at[2];
  row[0] = (in && B.z0 > 0) ? (int64_t)B.g0 - B.b : -2;
  row[1] = (in && B.z0 + B.tzl < B.nplanes) ? (int64_t)B.g0 + (int64_t)B.tzl * B.b : -2;
  place(tid, &row[2], &at[0]);
  place(tid + kResThreads, &row[3], &at[1]);
  double2r v[4];
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = plain(row[i]);
  } else {
    res_fetch4(A, row, tag, v);
  }
  if (MODE == 2) *lo = res_cg_direction(v[0], *lo, beta), *hi = res_cg_direction(v[1], *hi, beta);
  else *lo = v[0], *hi = v[1];
  put(at[0], v[2]), put(at[1], v[3]);
  // deep boxes with long lines: the rest of the halo, two pairs at a time
#pragma unroll 1
  for (int h = tid + 2 * kResThreads; h < nh; h += 2 * kResThreads) {
    int64_t r2[2];
    int a2[2];
    place(h, &r2[0], &a2[0]);
    place(h + kResThreads, &r2[1], &a2[1]);
    double2r v0, v1;
    if (MODE == 0) v0 = plain(r2[0]), v1 = plain(r2[1]);
    else res_fetch2(A, r2[0], r2[1], tag, &v0, &v1);
    put(a2[0], v0), put(a2[1], v1);
  }
}

// out = beta v + alpha M v on the block's rows, v = the LDS copy (own rows and halo); lo / hi = this thread's pairs of v
// in the planes below / above the box.  spmv_canon_tile_kernel's arithmetic, plane by plane with the planes below /
// at / above in three register pairs.  (ONE instance per kernel: the solvers below run init and iterations through
// the same call -- an inlined copy per call site costs registers the rows need.)
// CACHE: the fourteen coefficients of a pair of rows stay in registers from plane to plane while the weight words do not
// change (wave-uniform test) -- in a box away from the lattice's top and bottom they are the same in every plane of a
// thread's column, and the byte-indexed look-ups are 14 of the 20 LDS reads a pair of rows costs: the apply is bound by the
// LDS read rate.  The same values either way: the same bits.
template <int TZ, bool CACHE = false>  // (CACHE: boxes of 3 - 12 planes, kResApplyCacheCg)
__device__ __forceinline__ void res_apply(const ResArgs &A, const ResBox &B, const double *P, const double *dict_sh,
                                          double2r lo, double2r hi, const u64x2r (&w)[TZ], double2r (&out)[TZ]) {
  const int at0 = B.a + B.tid2;
  const char *dsh = reinterpret_cast<const char *>(dict_sh);
  double2r prev = lo, cur = *reinterpret_cast<const double2r *>(&P[at0]);
  double ca[7], cb[7];  // (CACHE) the coefficients of the current weight words: ext, then the six neighbours'
#pragma unroll
  for (int t = 0; t < TZ; ++t) {
    const int at = t * B.ldw + at0;
    double2r next = hi;
    if (t + 1 < TZ) {
      const double2r up = *reinterpret_cast<const double2r *>(&P[at + B.ldw]);
      next = t + 1 < B.tzl ? up : hi;
    }
    // (the words are loop-invariant in the solver's loop: left alone, the compiler hoists the fourteen byte offsets
    //  of every pair of rows out of it -- 14 registers per plane)
    unsigned long long wa = w[t].x, wb = w[t].y;
    asm volatile("" : "+v"(wa), "+v"(wb));
    double2r xg[6];
    xg[0] = prev, xg[5] = next;
    xg[1] = *reinterpret_cast<const double2r *>(&P[at - B.a]);
    xg[4] = *reinterpret_cast<const double2r *>(&P[at + B.a]);
    xg[2].x = P[at - 1], xg[2].y = cur.x;
    xg[3].x = cur.y, xg[3].y = P[at + 2];
    double acc_a = 0.0, acc_b = 0.0;
    double ext_a, ext_b;
    if (CACHE) {
      bool fresh = true;
      if (t > 0) fresh = A.apply_cache == 0 || __builtin_amdgcn_ballot_w64(w[t].x != w[t - 1].x || w[t].y != w[t - 1].y) != 0ull;
      if (fresh) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          ca[k] = *reinterpret_cast<const double *>(dsh + ((unsigned)(wa >> (8 * k)) & 0xffu));
          cb[k] = *reinterpret_cast<const double *>(dsh + ((unsigned)(wb >> (8 * k)) & 0xffu));
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        acc_a += ca[k + 1] * (xg[k].x - cur.x);
        acc_b += cb[k + 1] * (xg[k].y - cur.y);
      }
      ext_a = ca[0], ext_b = cb[0];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const unsigned ba = (unsigned)(wa >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(wb >> (8 * (k + 1))) & 0xffu;
        acc_a += *reinterpret_cast<const double *>(dsh + ba) * (xg[k].x - cur.x);
        acc_b += *reinterpret_cast<const double *>(dsh + bb) * (xg[k].y - cur.y);
      }
      ext_a = *reinterpret_cast<const double *>(dsh + ((unsigned)wa & 0xffu));
      ext_b = *reinterpret_cast<const double *>(dsh + ((unsigned)wb & 0xffu));
    }
    const double ya = __builtin_fma(A.alpha, __builtin_fma(ext_a, cur.x, acc_a), A.beta * cur.x);
    const double yb = __builtin_fma(A.alpha, __builtin_fma(ext_b, cur.y, acc_b), A.beta * cur.y);
    out[t].x = ((B.mask_a >> t) & 1u) ? ya : 0.0;  // (rows that do not exist may have read anything: selected away)
    out[t].y = ((B.mask_b >> t) & 1u) ? yb : 0.0;
    prev = cur, cur = next;
    // plane by plane: the result is pinned HERE (left alone, the optimiser sinks a plane's arithmetic to the first use of
    // its result and keeps the twenty values it read from LDS until then), and no plane's LDS reads move up
    asm volatile("" : "+v"(out[t].x), "+v"(out[t].y));
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The own rows of `vec` (zeros where a row does not exist).
template <int TZ>
__device__ __forceinline__ void res_load_rows(const ResBox &B, const double *vec, double2r (&v)[TZ]) {
#pragma unroll
  for (int t = 0; t < TZ; ++t) {
    v[t] = double2r{0.0, 0.0};
    if ((B.mask_b >> t) & 1u) v[t] = res_ld_pair(vec, res_off8(B, t));
    else if ((B.mask_a >> t) & 1u) v[t].x = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(vec) + res_off8(B, t));
  }
}
template <int TZ>
__device__ __forceinline__ void res_load_weights(const ResArgs &A, const ResBox &B, u64x2r (&w)[TZ]) {
#pragma unroll
  for (int t = 0; t < TZ; ++t) {
    w[t] = u64x2r{0ull, 0ull};
    if ((B.mask_a >> t) & 1u) w[t] = *reinterpret_cast<const u64x2r *>(A.pack + res_off8(B, t));  // (records are whole 128-row groups)
  }
}
// One pair into the LDS copy.  (A pair that does not exist writes nothing: its place may be the run's upper halo.)
__device__ __forceinline__ void res_lds_pair(const ResBox &B, double *P, int t, double2r v) {
  if ((B.mask_a >> t) & 1u) *reinterpret_cast<double2r *>(&P[t * B.ldw + B.a + B.tid2]) = v;
}


// MODE 2 in two steps (CG's early publish): the first batch of the halo -- everything, but for deep boxes with long lines
// -- is fetched while the all-reduce that yields beta is still under way (the neighbours publish right behind their own
// arrival at it: nobody waits for this block to do so), and turned into the halo of p' = r + beta p once beta is there.
template <int TZ>
__device__ __forceinline__ void res_halo2_fetch(const ResArgs &A, const ResBox &B, unsigned tag, double2r (&v)[4], int (&at)[2]) {
  const int tid = threadIdx.x;
  const int nh = B.tzl * B.a;
  auto place = [&](int hh, int64_t *row, int *where) {
    const int t = hh / B.a, j2 = 2 * (hh - t * B.a);
    const bool lower = j2 < B.a;
    const int jj = lower ? j2 : j2 - B.a;
    *where = hh < nh ? t * B.ldw + (lower ? jj : B.a + B.L + jj) : -1;
    *row = hh < nh ? (int64_t)(B.z0 + t) * B.b + B.s0 + (lower ? jj - B.a : B.L + jj) : -2;
  };
  const bool in = 2 * tid < B.L;
  int64_t row[4];
  row[0] = (in && B.z0 > 0) ? (int64_t)B.g0 - B.b : -2;
  row[1] = (in && B.z0 + B.tzl < B.nplanes) ? (int64_t)B.g0 + (int64_t)B.tzl * B.b : -2;
  place(tid, &row[2], &at[0]);
  place(tid + kResThreads, &row[3], &at[1]);
  res_fetch4(A, row, tag, v);
}
template <int TZ>
__device__ __forceinline__ void res_halo2_finish(const ResArgs &A, const ResBox &B, double *P, unsigned tag, const double2r (&v)[4],
                                                 const int (&at)[2], double2r *lo, double2r *hi, double beta) {
  const int tid = threadIdx.x;
  const int nh = B.tzl * B.a;
  auto put = [&](int where, double2r val) {
    if (where < 0) return;
    *reinterpret_cast<double2r *>(&P[where]) = res_cg_direction(val, *reinterpret_cast<const double2r *>(&P[where]), beta);
  };
  *lo = res_cg_direction(v[0], *lo, beta), *hi = res_cg_direction(v[1], *hi, beta);
  put(at[0], v[2]), put(at[1], v[3]);
#pragma unroll 1
  for (int h = tid + 2 * kResThreads; h < nh; h += 2 * kResThreads) {  // deep boxes with long lines: as res_halo
    int64_t r2[2];
    int a2[2];
    for (int q = 0; q < 2; ++q) {
      const int hh = h + q * kResThreads;
      const int t = hh / B.a, j2 = 2 * (hh - t * B.a);
      const bool lower = j2 < B.a;
      const int jj = lower ? j2 : j2 - B.a;
      a2[q] = hh < nh ? t * B.ldw + (lower ? jj : B.a + B.L + jj) : -1;
      r2[q] = hh < nh ? (int64_t)(B.z0 + t) * B.b + B.s0 + (lower ? jj - B.a : B.L + jj) : -2;
    }
    double2r v0, v1;
    res_fetch2(A, r2[0], r2[1], tag, &v0, &v1);
    put(a2[0], v0), put(a2[1], v1);
  }
}

// ---- CG ------------------------------------------------------------------------------------------------------------
// Registers: r and the weight words of the own rows for the whole solve; z from the apply to `r -= alpha z`; x too
// (XREG) where the box is at most 8 planes deep -- deeper boxes keep x in memory (the XCD's L2 holds its block's rows):
// read under the second all-reduce into z's place, updated and stored once per iteration.  The direction p lives in
// the LDS copy alone; its surface is published behind the all-reduce that yields beta.
template <int TZ, bool XREG>
__global__ __launch_bounds__(kResThreads) void res_cg_kernel(ResArgs A) {
  extern __shared__ __attribute__((aligned(16))) double P[];  // [TZ][a + kResRun + a]
  __shared__ double dict_sh[32];
  __shared__ double red[2 * 256 + 16];  // (co_allreduce_dense: NV x 256 polled values + the results)
  const ResBox B0 = res_box<TZ>(A);
  ResBox B = B0;
  SolverState *st = A.st;
  if (threadIdx.x < 32) dict_sh[threadIdx.x] = A.dict[threadIdx.x];
  unsigned long long seq = A.cnt[0], xseq = A.cnt[1];
  const double abs_tol = st->abs_tol, rel_tol = st->rel_tol;
  const long long num_iterations = st->num_iterations;
  double *history = st->history;
  double2r r[TZ], z[TZ], x[XREG ? TZ : 1];
  u64x2r w[TZ];
  res_load_weights<TZ>(A, B, w);
  // the start vector into the LDS copy: the first pass of the loop applies the operator to it
  double2r lo, hi;
#pragma unroll
  for (int t = 0; t < TZ; ++t) {
    const double2r xt = ((B.mask_a >> t) & 1u) ? res_ld_pair(A.x, res_off8(B, t)) : double2r{0.0, 0.0};
    if (XREG) x[XREG ? t : 0] = xt;
    res_lds_pair(B, P, t, xt);
  }
  res_halo<TZ, 0>(A, B, P, 0u, A.x, &lo, &hi);  // (x is not written before every block is past the first all-reduce)

  double gamma = 0.0, initial_error = 0.0, abs_err = 0.0, rel_err = 0.0;
  bool converged = false, started = false;
  long long it = 0;
  // (option resident_profile: where the time of an iteration goes, per block, by thread 0's clock)
  long long tick[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = A.prof ? wall_clock64() : 0;
  auto lap = [&](int k) {
    if (A.prof) {
      const long long now = wall_clock64();
      tick[k] += now - t_mark, t_mark = now;
    }
  };
  for (;;) {
    B = res_fresh(B0);
    const int at0 = B.a + B.tid2;
    __syncthreads();
    lap(0);  // the halo of the new direction: a wait for the neighbours' surfaces
    res_apply<TZ, kResApplyCacheCg(TZ)>(A, B, P, dict_sh, lo, hi, w, z);
    lap(1);  // the apply
    double acc[1] = {0.0};
    if (!started) {
      // ---- init: r = b - A x; p = r; gamma = <r, r>                               SolverCg.hpp:54-84
      started = true;
      res_load_rows<TZ>(B, A.rhs, r);
#pragma unroll
      for (int t = 0; t < TZ; ++t) {
        r[t].x = ((B.mask_a >> t) & 1u) ? r[t].x - z[t].x : 0.0;
        r[t].y = ((B.mask_b >> t) & 1u) ? r[t].y - z[t].y : 0.0;
        acc[0] += r[t].x * r[t].x;
        acc[0] += r[t].y * r[t].y;
      }
      res_allreduce<1>(acc, A, ++seq, red);  // (its barriers: every thread is done with the copy of x)
      ++xseq;
#pragma unroll
      for (int t = 0; t < TZ; ++t) {  // p = r
        res_publish_pair(A, B, t, r[t], (unsigned)xseq);
        res_lds_pair(B, P, t, r[t]);
      }
      gamma = acc[0];
      initial_error = abs_err = sqrt(gamma);
      converged = abs_tol > 0.0 && initial_error < abs_tol;  // Solver.hpp:124-128
      if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[0] = initial_error;
      if (converged || num_iterations <= 0) break;
    } else {
      // ---- an iteration                                                           SolverCg.hpp:86-126
      // (the LDS copy holds the direction p; every block has read its halo of p once it has published its <p, z> partial)
#pragma unroll
      for (int t = 0; t < TZ; ++t) {
        const double2r pt = *reinterpret_cast<const double2r *>(&P[t * B.ldw + at0]);
        acc[0] += ((B.mask_a >> t) & 1u) ? pt.x * z[t].x : 0.0;
        acc[0] += ((B.mask_b >> t) & 1u) ? pt.y * z[t].y : 0.0;
      }
      lap(2);  // <p, z> partials
      res_allreduce<1>(acc, A, ++seq, red);
      lap(3);  // the first all-reduce
      const double alpha = safe_divide(gamma, acc[0]);
      acc[0] = 0.0;
      const bool early = A.early_publish != 0 && TZ <= 8;  // (twelve planes per box, 144^3: 28.8 against 26.3 us per iteration)
      if (early) ++xseq;  // (a fresh tag whether or not anybody will read the surface: see res_halo)
#pragma unroll
      for (int t = 0; t < TZ; ++t) {
        r[t].x -= alpha * z[t].x, r[t].y -= alpha * z[t].y;
        acc[0] += r[t].x * r[t].x;
        acc[0] += r[t].y * r[t].y;
      }
      // the surface of the new residual travels under the all-reduce below; the neighbours form p' = r + beta p on it --
      // BEHIND the block's own arrival at that all-reduce (in front of it every block arrives late by the time the
      // surface's stores take to drain: 2.7 -> 5.2 us for the all-reduce at 128^3)
      const bool split = early && A.dense == nullptr;
      if (split) res_allreduce_arrive<1>(acc, A, seq + 1, red);
      if (early) {
        const unsigned rtag = (unsigned)xseq;
#pragma unroll
        for (int t = 0; t < TZ; ++t) res_publish_pair(A, B, t, r[t], rtag);
      }
      if (!XREG) res_load_rows<TZ>(B, A.x, z);  // x, in z's place, travels under the all-reduce
      lap(4);  // r -= alpha z, <r, r> partials
      ++seq;
      double2r hv[4] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
      int hat[2] = {-1, -1};
      // (shallow boxes only: at 128^3, eight planes per box and 3 584 surface rows per block, polling for the neighbours'
      //  surfaces this early costs the all-reduce more than it saves behind it -- 16.9 against 15.4 us per iteration;
      //  64^3, one plane per box: 8.4 against 9.2)
      constexpr bool kFetchUnderAllreduce = TZ <= 2;
      if (split && kFetchUnderAllreduce) res_halo2_fetch<TZ>(A, B, (unsigned)xseq, hv, hat);
      if (split) res_allreduce_wait<1>(acc, A, seq, red);
      else res_allreduce<1>(acc, A, seq, red);
      lap(5);  // the second all-reduce
      const double gamma_bar = gamma;
      gamma = acc[0];
      const double beta = safe_divide(gamma, gamma_bar);
      abs_err = sqrt(gamma);
      rel_err = abs_err / initial_error;
      converged = (abs_tol > 0.0 && abs_err < abs_tol) || (rel_tol > 0.0 && rel_err < rel_tol);  // Solver.hpp:132-140
      ++it;
      if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[it] = abs_err;
      const bool go_on = !converged && it < num_iterations;
      if (go_on && !early) ++xseq;
      // (deeper boxes: the second wave of every SIMD fetches the neighbours' surfaces and forms the halo of p' BEFORE its
      //  part of the update below, the first one behind it -- while one waits for its granules the other computes; own rows
      //  and halo entries are disjoint)
      const bool halo_first = early && go_on && TZ > 2 && A.halo_interleave != 0 && ((threadIdx.x >> 8) & 1) != 0;
      if (halo_first) res_halo<TZ, 2>(A, B, P, (unsigned)xseq, nullptr, &lo, &hi, beta);
      // x += alpha p; p = r + beta p (own rows: nobody else reads them before the next barrier)     :98, :123
#pragma unroll
      for (int t = 0; t < TZ; ++t) {
        const double2r pt = *reinterpret_cast<const double2r *>(&P[t * B.ldw + at0]);
        if (XREG) {
          x[XREG ? t : 0].x += alpha * pt.x, x[XREG ? t : 0].y += alpha * pt.y;
        } else {
          z[t].x += alpha * pt.x, z[t].y += alpha * pt.y;
          res_st_pair(A.x, res_off8(B, t), z[t], (B.mask_a >> t) & 1u, (B.mask_b >> t) & 1u);
        }
        if (go_on) {
          const double2r pn = res_cg_direction(r[t], pt, beta);
          if (!early) res_publish_pair(A, B, t, pn, (unsigned)xseq);
          res_lds_pair(B, P, t, pn);
        }
      }
      lap(6);  // x += alpha p, p = r + beta p, the surface out
      if (!go_on) break;
      if (early) {
        if (__hip_atomic_load(A.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (split && TZ <= 2) res_halo2_finish<TZ>(A, B, P, (unsigned)xseq, hv, hat, &lo, &hi, beta);
        else if (!halo_first) res_halo<TZ, 2>(A, B, P, (unsigned)xseq, nullptr, &lo, &hi, beta);
        continue;
      }
    }
    if (__hip_atomic_load(A.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    res_halo<TZ, 1>(A, B, P, (unsigned)xseq, nullptr, &lo, &hi);
  }
  if (XREG) {
#pragma unroll
    for (int t = 0; t < TZ; ++t) res_st_pair(A.x, res_off8(B, t), x[XREG ? t : 0], (B.mask_a >> t) & 1u, (B.mask_b >> t) & 1u);
  }
  if (A.prof && threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) A.prof[blockIdx.x * 8 + k] = tick[k];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.cnt[0] = seq, A.cnt[1] = xseq;  // (every block holds the same numbers)
    st->initial_error = initial_error;
    st->absolute_error = abs_err;
    st->relative_error = rel_err;
    st->iteration = it;
    st->converged = converged ? 1 : 0;
    st->done = 1;
  }
}

// ---- BiCGStab ------------------------------------------------------------------------------------------------------
// SolverBiCgStab.hpp:60-167: three all-reduces -- <rt, v>, (<t, s>, <t, t>), (<r, r>, <rt, r>) -- and two exchanges
// (the surfaces of p and of s = r - alpha v) per iteration.  Registers: r (s), p, v and the weight words of the own
// rows, t from the second apply to `r = s - omega t`; x in LDS beside the copy where both fit (XLDS; 128^3: 80 + 64 KB),
// else in memory; the shadow residual rt in memory (read-only: the XCD's L2 holds the block's rows).
__device__ __forceinline__ double res_bicg_direction(double r, double p, double v, double beta, double omega) {
  return __builtin_fma(beta, __builtin_fma(-omega, v, p), r);  // r + beta (p - omega v)      SolverBiCgStab.hpp:119
}
template <int TZ, bool XLDS>
__global__ __launch_bounds__(kResThreads) void res_bicgstab_kernel(ResArgs A) {
  extern __shared__ __attribute__((aligned(16))) double P[];  // [TZ][a + kResRun + a] (+ XLDS: [TZ][kResRun], x of the own rows, private to its thread)
  __shared__ double dict_sh[32];
  __shared__ double red[2 * 256 + 16];  // (co_allreduce_dense: NV x 256 polled values + the results)
  const ResBox B0 = res_box<TZ>(A);
  ResBox B = B0;
  SolverState *st = A.st;
  if (threadIdx.x < 32) dict_sh[threadIdx.x] = A.dict[threadIdx.x];
  unsigned long long seq = A.cnt[0], xseq = A.cnt[1];
  const double abs_tol = st->abs_tol, rel_tol = st->rel_tol;
  const long long num_iterations = st->num_iterations;
  double *history = st->history;
  double2r r[TZ], p[TZ], v[TZ];
  u64x2r w[TZ];
  res_load_weights<TZ>(A, B, w);
  double2r lo, hi;
  double *Q = P + TZ * B0.ldw;
#pragma unroll
  for (int t = 0; t < TZ; ++t) {
    const double2r xt = ((B.mask_a >> t) & 1u) ? res_ld_pair(A.x, res_off8(B, t)) : double2r{0.0, 0.0};
    res_lds_pair(B, P, t, xt);
    if (XLDS) *reinterpret_cast<double2r *>(&Q[t * kResRun + B.tid2]) = xt;
    p[t] = v[t] = double2r{0.0, 0.0};
  }
  res_halo<TZ, 0>(A, B, P, 0u, A.x, &lo, &hi);
  __syncthreads();
  double rho, initial_error, abs_err, rel_err = 0.0, alpha = 0.0, beta = 0.0, omega = 0.0;
  {  // ---- init: r = b - A x; rt = r; rho = <rt, r>                              SolverBiCgStab.hpp:82-90
    double2r y[TZ];
    res_apply<TZ, kResApplyCacheCg(TZ)>(A, B, P, dict_sh, lo, hi, w, y);
    res_load_rows<TZ>(B, A.rhs, r);
    double a1[1] = {0.0};
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      r[t].x = ((B.mask_a >> t) & 1u) ? r[t].x - y[t].x : 0.0;
      r[t].y = ((B.mask_b >> t) & 1u) ? r[t].y - y[t].y : 0.0;
      res_st_pair(A.rt, res_off8(B, t), r[t], (B.mask_a >> t) & 1u, (B.mask_b >> t) & 1u);
      a1[0] += r[t].x * r[t].x;
      a1[0] += r[t].y * r[t].y;
    }
    res_allreduce<1>(a1, A, ++seq, red);
    rho = a1[0];
  }
  initial_error = abs_err = sqrt(rho);
  bool converged = abs_tol > 0.0 && initial_error < abs_tol;  // Solver.hpp:124-128
  if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[0] = initial_error;
  long long it = 0;
  // (option resident_profile, as in the CG kernel)
  long long tick[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = A.prof ? wall_clock64() : 0;
  auto lap = [&](int k) {
    if (A.prof) {
      const long long now = wall_clock64();
      tick[k] += now - t_mark, t_mark = now;
    }
  };
  while (!converged && it < num_iterations) {
    if (__hip_atomic_load(A.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    // p = r + beta (p - omega v) (first iteration: p = r); v = A p; alpha = rho / <rt, v>      :114-119, :137-139
    // (the exchange buffer's last readers -- of s -- and the LDS copy's are behind two all-reduces)
    B = res_fresh(B0);
    ++xseq;
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      p[t].x = res_bicg_direction(r[t].x, p[t].x, v[t].x, beta, omega);
      p[t].y = res_bicg_direction(r[t].y, p[t].y, v[t].y, beta, omega);
      res_publish_pair(A, B, t, p[t], (unsigned)xseq);
      res_lds_pair(B, P, t, p[t]);
    }
    lap(0);  // p = r + beta (p - omega v), its surface out
    res_halo<TZ, 1>(A, B, P, (unsigned)xseq, nullptr, &lo, &hi);
    __syncthreads();
    lap(1);  // the halo of p: a wait for the neighbours' surfaces
    res_apply<TZ, kResApplyCacheCg(TZ)>(A, B, P, dict_sh, lo, hi, w, v);
    lap(2);  // v = A p
    {
      double a1[1] = {0.0};
#pragma unroll
      for (int t = 0; t < TZ; ++t) {
        const double2r rt = ((B.mask_a >> t) & 1u) ? res_ld_pair(A.rt, res_off8(B, t)) : double2r{0.0, 0.0};
        a1[0] += rt.x * v[t].x, a1[0] += ((B.mask_b >> t) & 1u) ? rt.y * v[t].y : 0.0;
      }
      res_allreduce<1>(a1, A, ++seq, red);
      alpha = safe_divide(rho, a1[0]);
    }
    lap(3);  // <rt, v> (rt from memory) and its all-reduce
    // s = r - alpha v (kept in r); t = A s; omega = <t, s> / <t, t>                            :140-141, :158-160
    // (every block has read its halo of p: it is past the alpha all-reduce)
    B = res_fresh(B0);
    ++xseq;
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      r[t].x = __builtin_fma(-alpha, v[t].x, r[t].x);
      r[t].y = __builtin_fma(-alpha, v[t].y, r[t].y);
      res_publish_pair(A, B, t, r[t], (unsigned)xseq);
      res_lds_pair(B, P, t, r[t]);
    }
    res_halo<TZ, 1>(A, B, P, (unsigned)xseq, nullptr, &lo, &hi);
    __syncthreads();
    lap(4);  // s = r - alpha v, its surface out, the halo of s
    double2r y[TZ];
    res_apply<TZ, kResApplyCacheCg(TZ)>(A, B, P, dict_sh, lo, hi, w, y);
    lap(5);  // t = A s
    double acc[2] = {0.0, 0.0};
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      acc[0] += y[t].x * r[t].x, acc[0] += y[t].y * r[t].y;
      acc[1] += y[t].x * y[t].x, acc[1] += y[t].y * y[t].y;
    }
    res_allreduce<2>(acc, A, ++seq, red);
    lap(6);  // <t, s>, <t, t> and their all-reduce
    omega = safe_divide(acc[0], acc[1]);
    // x += alpha p + omega s; r = s - omega t; |r|, <rt, r>                                    :140, :161-164, :116
    acc[0] = acc[1] = 0.0;
    B = res_fresh(B0);
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      double2r xv = XLDS ? *reinterpret_cast<const double2r *>(&Q[t * kResRun + B.tid2])
                         : ((B.mask_a >> t) & 1u) ? res_ld_pair(A.x, res_off8(B, t)) : double2r{0.0, 0.0};
      const double2r rt = ((B.mask_a >> t) & 1u) ? res_ld_pair(A.rt, res_off8(B, t)) : double2r{0.0, 0.0};
      xv.x += alpha * p[t].x, xv.y += alpha * p[t].y;
      xv.x += omega * r[t].x, xv.y += omega * r[t].y;
      if (XLDS) *reinterpret_cast<double2r *>(&Q[t * kResRun + B.tid2]) = xv;
      else res_st_pair(A.x, res_off8(B, t), xv, (B.mask_a >> t) & 1u, (B.mask_b >> t) & 1u);
      r[t].x -= omega * y[t].x, r[t].y -= omega * y[t].y;
      acc[0] += r[t].x * r[t].x, acc[0] += r[t].y * r[t].y;
      acc[1] += rt.x * r[t].x, acc[1] += ((B.mask_b >> t) & 1u) ? rt.y * r[t].y : 0.0;
    }
    res_allreduce<2>(acc, A, ++seq, red);
    lap(7);  // x += alpha p + omega s (x, rt through memory), r = s - omega t, |r|, <rt, r> and their all-reduce
    const double rho_bar = rho;
    rho = acc[1];
    beta = safe_divide(alpha * rho, omega * rho_bar);  // :116-118, for the next iteration
    abs_err = sqrt(acc[0]);
    rel_err = abs_err / initial_error;
    converged = (abs_tol > 0.0 && abs_err < abs_tol) || (rel_tol > 0.0 && rel_err < rel_tol);  // Solver.hpp:132-140
    ++it;
    if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[it] = abs_err;
  }
  if (XLDS) {
    B = res_fresh(B0);
#pragma unroll
    for (int t = 0; t < TZ; ++t)
      res_st_pair(A.x, res_off8(B, t), *reinterpret_cast<const double2r *>(&Q[t * kResRun + B.tid2]), (B.mask_a >> t) & 1u, (B.mask_b >> t) & 1u);
  }
  if (A.prof && threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) A.prof[blockIdx.x * 8 + k] = tick[k];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.cnt[0] = seq, A.cnt[1] = xseq;
    st->initial_error = initial_error;
    st->absolute_error = abs_err;
    st->relative_error = rel_err;
    st->iteration = it;
    st->converged = converged ? 1 : 0;
    st->done = 1;
  }
}

// ---- BiCGStab, early publish (round 4) -----------------------------------------------------------------------------
// The loop above waits for its neighbours five times per iteration: three all-reduces and the halos of p and of s.  Both
// halos can be FORMED instead of fetched: p' = r + beta (p - omega v) and s = r - alpha v hold on the halo rows as on the
// own rows, with the owner's own expressions on the owner's own operands -- the same bits -- if the block keeps the halos
// of r, p and v.  What must travel is what an apply produced: the surface of v = A p, published right behind the block's
// arrival at the all-reduce of <rt, v> and fetched while that all-reduce is under way; and the surface of the new
// residual r = s - omega t, published and fetched under the all-reduce of |r|, <rt, r> (at init: under that of <r, r>).
// Three waits per iteration instead of five.  The two surfaces use two exchange buffers: a block that has passed the
// last all-reduce goes on to publish v without another synchronisation, while a neighbour may still be reading r.
// Halo sets (LDS, behind the copy): Hr, Hp, Hv, [tzl][2 a] doubles each, pair hh always handled by the same thread; the
// pairs of the planes below / above in registers.  Boxes of at most 6 planes: bitwise the exchanged halos (tests).
// (At 6 planes the two kernels first disagreed by an ulp in <t, t> of some iteration although every formed halo was verified
//  bitwise against the exchanged one inside the kernel: under -ffp-contract=fast the backend had fused a multiply and an add
//  of different statements in one kernel and not in the other.  This unit is compiled with -ffp-contract=on (Makefile).
//  8 planes: 11 registers spilt and x no longer in LDS beside the sets: 128^3 34.4 -> 38.5 us, not taken.)
template <int TZ>
__device__ __forceinline__ void res_halo_place(const ResBox &B, int hh, int64_t *row, int *at) {
  const int nh = B.tzl * B.a;
  const int t = hh / B.a, j2 = 2 * (hh - t * B.a);
  const bool lower = j2 < B.a;
  const int jj = lower ? j2 : j2 - B.a;
  *at = hh < nh ? t * B.ldw + (lower ? jj : B.a + B.L + jj) : -1;
  *row = hh < nh ? (int64_t)(B.z0 + t) * B.b + B.s0 + (lower ? jj - B.a : B.L + jj) : -2;
}
// the surface published with `tag` -> the halo set H and the pairs below / above
// (... and then the second half of the all-reduce `seq`, whose first half the caller has run)
template <int TZ, int NV>
__device__ __forceinline__ void res_halo_fetch_set(const ResArgs &A, const ResBox &B, unsigned tag, double *H, double2r *lo, double2r *hi,
                                                   double (&s)[NV], unsigned long long seq, double *lds) {
  const int tid = threadIdx.x;
  const int nh = B.tzl * B.a;
  const bool in = 2 * tid < B.L;
  int64_t row[4];
  int at[2];
  row[0] = (in && B.z0 > 0) ? (int64_t)B.g0 - B.b : -2;
  row[1] = (in && B.z0 + B.tzl < B.nplanes) ? (int64_t)B.g0 + (int64_t)B.tzl * B.b : -2;
  res_halo_place<TZ>(B, tid, &row[2], &at[0]);
  res_halo_place<TZ>(B, tid + kResThreads, &row[3], &at[1]);
  double2r v[4];
  // (one polling loop for the granules and the all-reduce's slots together -- all ten loads of a poll in flight at once -- was
  //  slower than the two in turn: 64^3 15.2 against 14.3 us per iteration)
  res_fetch4(A, row, tag, v);
  res_allreduce_wait<NV>(s, A, seq, lds);
  *lo = v[0], *hi = v[1];
  if (at[0] >= 0) *reinterpret_cast<double2r *>(&H[2 * tid]) = v[2];
  if (at[1] >= 0) *reinterpret_cast<double2r *>(&H[2 * (tid + kResThreads)]) = v[3];
#pragma unroll 1
  for (int h = tid + 2 * kResThreads; h < nh; h += 2 * kResThreads) {
    int64_t r2[2];
    int a2[2];
    res_halo_place<TZ>(B, h, &r2[0], &a2[0]);
    res_halo_place<TZ>(B, h + kResThreads, &r2[1], &a2[1]);
    double2r v0, v1;
    res_fetch2(A, r2[0], r2[1], tag, &v0, &v1);
    if (a2[0] >= 0) *reinterpret_cast<double2r *>(&H[2 * h]) = v0;
    if (a2[1] >= 0) *reinterpret_cast<double2r *>(&H[2 * (h + kResThreads)]) = v1;
  }
}
__device__ __forceinline__ void res_publish_surface_pair(char *exch, size_t exch_half, const ResBox &B, int t, double2r v, unsigned tag) {
  if (((B.mask_a >> t) & 1u) && (B.edge_y || t == 0 || t == B.tzl - 1)) {
    char *e = exch + (size_t)res_off8(B, t);
    res_store16(e, res_granule(v.x, tag));
    if ((B.mask_b >> t) & 1u) res_store16(e + exch_half, res_granule(v.y, tag));
  }
}

template <int TZ, bool XLDS>
__global__ __launch_bounds__(kResThreads) void res_bicgstab_early_kernel(ResArgs A) {
  extern __shared__ __attribute__((aligned(16))) double P[];  // the copy, [XLDS: x of the own rows,] Hr, Hp, Hv
  __shared__ double dict_sh[32];
  __shared__ double red[2 * 256 + 16];
  const ResBox B0 = res_box<TZ>(A);
  ResBox B = B0;
  SolverState *st = A.st;
  if (threadIdx.x < 32) dict_sh[threadIdx.x] = A.dict[threadIdx.x];
  unsigned long long seq = A.cnt[0], xseq = A.cnt[1];
  const double abs_tol = st->abs_tol, rel_tol = st->rel_tol;
  const long long num_iterations = st->num_iterations;
  double *history = st->history;
  char *const exch_v = A.exch, *const exch_r = A.exch + A.exch_stride;
  double2r r[TZ], p[TZ], v[TZ];
  u64x2r w[TZ];
  res_load_weights<TZ>(A, B, w);
  double2r lo, hi;  // of the vector in the copy
  double2r r_lo, r_hi, p_lo{0.0, 0.0}, p_hi{0.0, 0.0}, v_lo{0.0, 0.0}, v_hi{0.0, 0.0};
  double *Q = P + TZ * B0.ldw;
  double *Hr = Q + (XLDS ? TZ * kResRun : 0), *Hp = Hr + TZ * 2 * A.a, *Hv = Hp + TZ * 2 * A.a;
  const int nh = B0.tzl * B0.a;
  for (int h = threadIdx.x; h < nh; h += kResThreads)
    *reinterpret_cast<double2r *>(&Hp[2 * h]) = *reinterpret_cast<double2r *>(&Hv[2 * h]) = double2r{0.0, 0.0};
#pragma unroll
  for (int t = 0; t < TZ; ++t) {
    const double2r xt = ((B.mask_a >> t) & 1u) ? res_ld_pair(A.x, res_off8(B, t)) : double2r{0.0, 0.0};
    res_lds_pair(B, P, t, xt);
    if (XLDS) *reinterpret_cast<double2r *>(&Q[t * kResRun + B.tid2]) = xt;
    p[t] = v[t] = double2r{0.0, 0.0};
  }
  res_halo<TZ, 0>(A, B, P, 0u, A.x, &lo, &hi);
  __syncthreads();
  double rho, initial_error, abs_err, rel_err = 0.0, alpha = 0.0, beta = 0.0, omega = 0.0;
  {  // ---- init: r = b - A x; rt = r; rho = <rt, r>                              SolverBiCgStab.hpp:82-90
    double2r y[TZ];
    res_apply<TZ, kResApplyCacheCg(TZ)>(A, B, P, dict_sh, lo, hi, w, y);
    res_load_rows<TZ>(B, A.rhs, r);
    double a1[1] = {0.0};
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      r[t].x = ((B.mask_a >> t) & 1u) ? r[t].x - y[t].x : 0.0;
      r[t].y = ((B.mask_b >> t) & 1u) ? r[t].y - y[t].y : 0.0;
      res_st_pair(A.rt, res_off8(B, t), r[t], (B.mask_a >> t) & 1u, (B.mask_b >> t) & 1u);
      a1[0] += r[t].x * r[t].x;
      a1[0] += r[t].y * r[t].y;
    }
    res_allreduce_arrive<1>(a1, A, ++seq, red);
    ++xseq;
#pragma unroll
    for (int t = 0; t < TZ; ++t) res_publish_surface_pair(exch_r, A.exch_half, B, t, r[t], (unsigned)xseq);
    A.exch = exch_r;
    res_halo_fetch_set<TZ, 1>(A, B, (unsigned)xseq, Hr, &r_lo, &r_hi, a1, seq, red);
    rho = a1[0];
  }
  initial_error = abs_err = sqrt(rho);
  bool converged = abs_tol > 0.0 && initial_error < abs_tol;  // Solver.hpp:124-128
  if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[0] = initial_error;
  long long it = 0;
  long long tick[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = A.prof ? wall_clock64() : 0;
  auto lap = [&](int k) {
    if (A.prof) {
      const long long now = wall_clock64();
      tick[k] += now - t_mark, t_mark = now;
    }
  };
  while (!converged && it < num_iterations) {
    if (__hip_atomic_load(A.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    // p = r + beta (p - omega v) (first iteration: p = r) on the own rows AND on the halo; v = A p      :114-119, :137-139
    // (the copy's last readers -- of s -- are behind two all-reduces)
    B = res_fresh(B0);
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      p[t].x = res_bicg_direction(r[t].x, p[t].x, v[t].x, beta, omega);
      p[t].y = res_bicg_direction(r[t].y, p[t].y, v[t].y, beta, omega);
      res_lds_pair(B, P, t, p[t]);
    }
    p_lo.x = res_bicg_direction(r_lo.x, p_lo.x, v_lo.x, beta, omega), p_lo.y = res_bicg_direction(r_lo.y, p_lo.y, v_lo.y, beta, omega);
    p_hi.x = res_bicg_direction(r_hi.x, p_hi.x, v_hi.x, beta, omega), p_hi.y = res_bicg_direction(r_hi.y, p_hi.y, v_hi.y, beta, omega);
#pragma unroll 1
    for (int h = threadIdx.x; h < nh; h += kResThreads) {
      int64_t row;
      int at;
      res_halo_place<TZ>(B, h, &row, &at);
      const double2r hr = *reinterpret_cast<const double2r *>(&Hr[2 * h]), hv = *reinterpret_cast<const double2r *>(&Hv[2 * h]);
      double2r hp = *reinterpret_cast<const double2r *>(&Hp[2 * h]);
      hp.x = res_bicg_direction(hr.x, hp.x, hv.x, beta, omega), hp.y = res_bicg_direction(hr.y, hp.y, hv.y, beta, omega);
      *reinterpret_cast<double2r *>(&Hp[2 * h]) = hp;
      *reinterpret_cast<double2r *>(&P[at]) = hp;
    }
    __syncthreads();
    lap(0);  // p = r + beta (p - omega v), own rows and halo
    lap(1);
    res_apply<TZ, kResApplyCacheCg(TZ)>(A, B, P, dict_sh, p_lo, p_hi, w, v);
    lap(2);  // v = A p
    {
      double a1[1] = {0.0};
#pragma unroll
      for (int t = 0; t < TZ; ++t) {
        const double2r rt = ((B.mask_a >> t) & 1u) ? res_ld_pair(A.rt, res_off8(B, t)) : double2r{0.0, 0.0};
        a1[0] += rt.x * v[t].x, a1[0] += ((B.mask_b >> t) & 1u) ? rt.y * v[t].y : 0.0;
      }
      res_allreduce_arrive<1>(a1, A, ++seq, red);
      ++xseq;
#pragma unroll
      for (int t = 0; t < TZ; ++t) res_publish_surface_pair(exch_v, A.exch_half, B, t, v[t], (unsigned)xseq);
      A.exch = exch_v;
      res_halo_fetch_set<TZ, 1>(A, B, (unsigned)xseq, Hv, &v_lo, &v_hi, a1, seq, red);
      alpha = safe_divide(rho, a1[0]);
    }
    lap(3);  // <rt, v> (rt from memory), its all-reduce; under it the surface of v out and the neighbours' in
    // s = r - alpha v (kept in r) on the own rows and on the halo; t = A s; omega = <t, s> / <t, t>        :140-141, :158-160
    // (every block is past the alpha all-reduce: nobody reads the copy of p any more)
    B = res_fresh(B0);
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      r[t].x = __builtin_fma(-alpha, v[t].x, r[t].x);
      r[t].y = __builtin_fma(-alpha, v[t].y, r[t].y);
      res_lds_pair(B, P, t, r[t]);
    }
    lo = double2r{__builtin_fma(-alpha, v_lo.x, r_lo.x), __builtin_fma(-alpha, v_lo.y, r_lo.y)};
    hi = double2r{__builtin_fma(-alpha, v_hi.x, r_hi.x), __builtin_fma(-alpha, v_hi.y, r_hi.y)};
#pragma unroll 1
    for (int h = threadIdx.x; h < nh; h += kResThreads) {
      int64_t row;
      int at;
      res_halo_place<TZ>(B, h, &row, &at);
      const double2r hr = *reinterpret_cast<const double2r *>(&Hr[2 * h]), hv = *reinterpret_cast<const double2r *>(&Hv[2 * h]);
      *reinterpret_cast<double2r *>(&P[at]) = double2r{__builtin_fma(-alpha, hv.x, hr.x), __builtin_fma(-alpha, hv.y, hr.y)};
    }
    __syncthreads();
    lap(4);  // s = r - alpha v, own rows and halo
    double2r y[TZ];
    res_apply<TZ, kResApplyCacheCg(TZ)>(A, B, P, dict_sh, lo, hi, w, y);
    lap(5);  // t = A s
    double acc[2] = {0.0, 0.0};
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      acc[0] += y[t].x * r[t].x, acc[0] += y[t].y * r[t].y;
      acc[1] += y[t].x * y[t].x, acc[1] += y[t].y * y[t].y;
    }
    res_allreduce<2>(acc, A, ++seq, red);
    lap(6);  // <t, s>, <t, t> and their all-reduce
    omega = safe_divide(acc[0], acc[1]);
    // x += alpha p + omega s; r = s - omega t; |r|, <rt, r>                                    :140, :161-164, :116
    acc[0] = acc[1] = 0.0;
    B = res_fresh(B0);
#pragma unroll
    for (int t = 0; t < TZ; ++t) {
      double2r xv = XLDS ? *reinterpret_cast<const double2r *>(&Q[t * kResRun + B.tid2])
                         : ((B.mask_a >> t) & 1u) ? res_ld_pair(A.x, res_off8(B, t)) : double2r{0.0, 0.0};
      const double2r rt = ((B.mask_a >> t) & 1u) ? res_ld_pair(A.rt, res_off8(B, t)) : double2r{0.0, 0.0};
      xv.x += alpha * p[t].x, xv.y += alpha * p[t].y;
      xv.x += omega * r[t].x, xv.y += omega * r[t].y;
      if (XLDS) *reinterpret_cast<double2r *>(&Q[t * kResRun + B.tid2]) = xv;
      else res_st_pair(A.x, res_off8(B, t), xv, (B.mask_a >> t) & 1u, (B.mask_b >> t) & 1u);
      r[t].x -= omega * y[t].x, r[t].y -= omega * y[t].y;
      acc[0] += r[t].x * r[t].x, acc[0] += r[t].y * r[t].y;
      acc[1] += rt.x * r[t].x, acc[1] += ((B.mask_b >> t) & 1u) ? rt.y * r[t].y : 0.0;
    }
    res_allreduce_arrive<2>(acc, A, ++seq, red);
    ++xseq;  // (a fresh tag whether or not the solve goes on: see res_halo)
#pragma unroll
    for (int t = 0; t < TZ; ++t) res_publish_surface_pair(exch_r, A.exch_half, B, t, r[t], (unsigned)xseq);
    A.exch = exch_r;
    res_halo_fetch_set<TZ, 2>(A, B, (unsigned)xseq, Hr, &r_lo, &r_hi, acc, seq, red);
    lap(7);  // x += alpha p + omega s, r = s - omega t, |r|, <rt, r>, their all-reduce; under it the surface of r out and in
    const double rho_bar = rho;
    rho = acc[1];
    beta = safe_divide(alpha * rho, omega * rho_bar);  // :116-118, for the next iteration
    abs_err = sqrt(acc[0]);
    rel_err = abs_err / initial_error;
    converged = (abs_tol > 0.0 && abs_err < abs_tol) || (rel_tol > 0.0 && rel_err < rel_tol);  // Solver.hpp:132-140
    ++it;
    if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[it] = abs_err;
  }
  if (XLDS) {
    B = res_fresh(B0);
#pragma unroll
    for (int t = 0; t < TZ; ++t)
      res_st_pair(A.x, res_off8(B, t), *reinterpret_cast<const double2r *>(&Q[t * kResRun + B.tid2]), (B.mask_a >> t) & 1u, (B.mask_b >> t) & 1u);
  }
  if (A.prof && threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) A.prof[blockIdx.x * 8 + k] = tick[k];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.cnt[0] = seq, A.cnt[1] = xseq;
    st->initial_error = initial_error;
    st->absolute_error = abs_err;
    st->relative_error = rel_err;
    st->iteration = it;
    st->converged = converged ? 1 : 0;
    st->done = 1;
  }
}

// ---- host ----------------------------------------------------------------------------------------------------------
struct ResGeometry {
  int a, b, nplanes, nsec, tz, blocks;
  size_t lds_bytes;
  bool x_lds;  // BiCGStab: x of the own rows in LDS beside the copy
  bool early;  // BiCGStab: res_bicgstab_early_kernel (halo sets of r, p, v behind the copy)
};
// Does the operator run on the resident path, and how: the smallest number of planes per block with which one block
// per CU covers the lattice.
static bool res_geometry(const storm_hip_op *op, ResGeometry *G, bool bicgstab = false) {
  const storm_hip_ctx *c = op->ctx;
  if (c->opt_resident_path == 0 || c->opt_latency_path != 1 || c->coop_disabled != 0 || c->comm != nullptr || c->opt_profile_spmv != 0) return false;
  if (op->pair != 2 || op->canon_k != 6 || op->canon_m1 != 2 || op->n_halo != 0 || op->d_bnd_pack != nullptr) return false;
  if (op->dict_size <= 0 || op->dict_size > 32 || op->tail_rows != 0) return false;
  const int *o = op->canon_off;
  const int a = o[4], b = o[5];
  if (o[0] != -b || o[1] != -a || o[2] != -1 || o[3] != 1) return false;
  if (a < 2 || a > 512 || (a & 1) || (b & 1) || b < 2 * a) return false;
  const int64_t n = op->n_rows;
  if (n < c->opt_resident_min_rows || n > c->opt_resident_max_rows) return false;
  const int64_t nplanes = (n + b - 1) / b, nsec = (b + kResRun - 1) / kResRun;
  const int cus = std::min(c->num_cus, 256);
  if (nsec > cus) return false;
  const int variants[] = {1, 2, 3, 4, 6, 8, 12};
  for (int tz : variants) {
    if (c->opt_resident_planes > 0 && tz != c->opt_resident_planes) continue;  // (tests: a given depth, ragged last chunks)
    if (tz > kResMaxPlanes || tz > c->opt_resident_max_planes || (bicgstab && tz > kResMaxPlanesBicg)) break;
    const int64_t blocks = nsec * ((nplanes + tz - 1) / tz);
    const size_t lds = sizeof(double) * (size_t)tz * (size_t)(kResRun + 2 * a);
    if (blocks > cus || lds > (size_t)150 * 1024) continue;
    G->a = a, G->b = b, G->nplanes = (int)nplanes, G->nsec = (int)nsec, G->tz = tz, G->blocks = (int)blocks, G->lds_bytes = lds;
    G->x_lds = false, G->early = false;
    const size_t with_x = lds + sizeof(double) * (size_t)tz * kResRun;
    if (bicgstab && with_x <= (size_t)156 * 1024) G->x_lds = true, G->lds_bytes = with_x;
    const size_t sets = sizeof(double) * (size_t)3 * (size_t)tz * (size_t)(2 * a);
    if (bicgstab && c->opt_resident_early != 0 && tz <= kResMaxPlanesBicgEarly) {
      if (G->lds_bytes + sets <= (size_t)156 * 1024) G->early = true, G->lds_bytes += sets;
      else if (lds + sets <= (size_t)156 * 1024)  // (rather the halo sets than x in LDS)
        G->early = true, G->x_lds = false, G->lds_bytes = lds + sets;
    }
    return true;
  }
  return false;
}

bool res_eligible(const storm_hip_op *op, bool bicgstab) {
  ResGeometry G;
  return res_geometry(op, &G, bicgstab);
}

template <int TZ>
static const void *res_kernel(bool bicgstab, bool x_lds, bool early = false) {
  if constexpr (TZ <= kResMaxPlanesBicgEarly)
    if (bicgstab && early) return x_lds ? (const void *)res_bicgstab_early_kernel<TZ, true> : (const void *)res_bicgstab_early_kernel<TZ, false>;
  return bicgstab ? (x_lds ? (const void *)res_bicgstab_kernel<TZ, true> : (const void *)res_bicgstab_kernel<TZ, false>)
                  : (const void *)res_cg_kernel<TZ, (TZ <= 8)>;
}

// The whole solve; fills the SolverState on the device (the caller reads it back).  *taken = false: nothing ran (the
// operator does not qualify, or the kernel cannot be resident) -- the caller takes another path.
int res_solve(bool bicgstab, const storm_hip_op *op, double alpha, double beta, const double *b, double *x, double *rt,
              SolverState *d_state, bool *taken) {
  storm_hip_ctx *c = op->ctx;
  *taken = false;
  ResGeometry G;
  if (!res_geometry(op, &G, bicgstab)) return STORM_HIP_OK;
  if (c->opt_coop_force_fail == 1) {
    c->coop_fallback = 1;
    return STORM_HIP_OK;
  }
  const void *fn = nullptr;
  switch (G.tz) {
    case 1: fn = res_kernel<1>(bicgstab, G.x_lds, G.early); break;
    case 2: fn = res_kernel<2>(bicgstab, G.x_lds, G.early); break;
    case 3: fn = res_kernel<3>(bicgstab, G.x_lds, G.early); break;
    case 4: fn = res_kernel<4>(bicgstab, G.x_lds, G.early); break;
    case 6: fn = res_kernel<6>(bicgstab, G.x_lds, G.early); break;
    case 8: fn = res_kernel<8>(bicgstab, G.x_lds, G.early); break;
    case 12: fn = res_kernel<12>(bicgstab, G.x_lds); break;
    default: return STORM_HIP_OK;
  }
  // every block must be resident: one per CU with this much LDS, as the occupancy query sees it
  if (G.lds_bytes > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G.lds_bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      c->coop_fallback = 1;
      return STORM_HIP_OK;
    }
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kResThreads, G.lds_bytes) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    c->coop_fallback = 1;
    return STORM_HIP_OK;
  }
  // the exchange buffer (one granule per row) and the slots: zero-filled once, tags only ever grow
  const size_t exch_half = (size_t)16 * (size_t)((op->n_rows + 3) / 2 + 8);  // (the even rows' granules; 256-byte aligned start of the odd rows')
  const size_t exch_stride = (2 * exch_half + 256 + 255) / 256 * 256;  // (two buffers: BiCGStab's early publish)
  if (c->res_exch_rows < op->n_rows + 2) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_res_exch) (void)hipFree(c->d_res_exch);
    c->d_res_exch = nullptr, c->res_exch_rows = 0;
    HIP_TRY(hipMalloc((void **)&c->d_res_exch, 2 * exch_stride));
    HIP_TRY(hipMemsetAsync(c->d_res_exch, 0, 2 * exch_stride, c->stream));
    c->res_exch_rows = op->n_rows + 2;
  }
  if (c->d_res_slots == nullptr) {
    const size_t bytes = (size_t)2 * 256 * kLatSlotStride + 256 + (size_t)2 * kDenseMaxValues * 256 * 16;  // flat slots, counters, dense granules
    HIP_TRY(hipMalloc((void **)&c->d_res_slots, bytes));
    HIP_TRY(hipMemsetAsync(c->d_res_slots, 0, bytes, c->stream));
  }
  ResArgs A{};
  A.pack = op->d_pack, A.dict = op->d_dict, A.a = G.a, A.b = G.b, A.nplanes = G.nplanes, A.nsec = G.nsec, A.n_rows = op->n_rows;
  A.alpha = alpha, A.beta = beta, A.rhs = b, A.x = x, A.rt = rt;
  A.exch = c->d_res_exch, A.exch_half = (exch_half + 255) / 256 * 256, A.exch_stride = exch_stride, A.slots = c->d_res_slots;
  A.gave_up = reinterpret_cast<int *>(c->d_lat_slots + (size_t)2 * 256 * kLatSlotStride);
  A.cnt = reinterpret_cast<unsigned long long *>(c->d_res_slots + (size_t)2 * 256 * kLatSlotStride);
  // (measured A/B, CG us per iteration with the dense form / the 64-byte slots: 64^3 10.0 / 8.9, 128^3 16.9 / 15.8 -- with ONE
  //  or two values the extra barrier and the trip through LDS cost more than the fewer requests save; the Gram-Schmidt
  //  chains, six and ten values, are the dense form's case.  coop_dense = 2 forces it here, for that A/B.)
  A.dense = c->opt_coop_dense == 2 ? c->d_res_slots + (size_t)2 * 256 * kLatSlotStride + 256 : nullptr;
  A.early_publish = (int)(c->opt_resident_early != 0);
  A.apply_cache = (int)(c->opt_resident_apply_cache != 0);
  A.halo_interleave = (int)(c->opt_resident_halo_interleave != 0);
  A.st = d_state;
  A.prof = nullptr;
  if (c->opt_resident_profile != 0) {
    if (c->d_res_prof == nullptr) HIP_TRY(hipMalloc((void **)&c->d_res_prof, sizeof(long long) * 256 * 8));
    HIP_TRY(hipMemsetAsync(c->d_res_prof, 0, sizeof(long long) * 256 * 8, c->stream));
    A.prof = c->d_res_prof;
    c->res_prof_blocks = G.blocks;
  }
  void *args[] = {&A};
  // (launched like any kernel: the grid is one block per CU at most and the occupancy query accepts it -- see
  //  latency.hip coop_launch; option coop_plain = 0: through the runtime's cooperative queue)
  const hipError_t e = c->opt_coop_plain != 0
                           ? hipLaunchKernel(fn, dim3((unsigned)G.blocks), dim3(kResThreads), args, G.lds_bytes, c->stream)
                           : hipLaunchCooperativeKernel(fn, dim3((unsigned)G.blocks), dim3(kResThreads), args, (unsigned)G.lds_bytes, c->stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    c->coop_fallback = 1;
    return STORM_HIP_OK;
  }
  c->coop_ran = 1;
  *taken = true;
  return STORM_HIP_OK;
}

}  // namespace storm
