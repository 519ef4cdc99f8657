// The operator-apply hot path: stormDivGrad (source_apps/playground/Playground.cpp:115-131)
// as a gather-form SpMV on a sliced-ELL / CSR-tail hybrid.
//
// Layout in HBM (built once per operator on the host, slot order = face order):
//   slice s = rows [64 s, 64 s + 64): one wavefront.  Everything the slice streams is ONE
//   contiguous record at byte offset slice_off[s]:
//        [ ext : 64 x f64 ][ col : W x 64 x i32 ][ val : W x 64 x f64 ]      (512 + 768 W bytes)
//   Inside col / val the slots are stored in PAIRS, lane-major: lane l finds (slot 2p, slot 2p+1)
//   of its row as one int2 / one double2, so a wave reads 512 B of columns and 1 KiB of weights
//   per instruction (8- and 16-byte accesses per lane; an odd last slot follows column-major).  Packing ext/col/val of a slice
//   into one record leaves the kernel three DRAM streams (records, x, y) instead of five;
//   on MI355X the number of concurrent streams, not L2 locality, decided the rate (measured:
//   profiles/r01_notes.md).
//   W = min(longest row of the slice, ell_cap); what does not fit goes to a CSR tail handled
//   by a wave-per-row kernel with a __shfl_down reduction.  Padding slots carry weight 0 and
//   the row's own index as column.
// Value-dictionary records (VARIANT 2, chosen at build time when it is lossless): when ext and the
//   weights of the whole operator take at most 256 distinct fp64 values and no slice is wider than
//   7 slots -- any mesh made of a few repeated cell shapes, the 256^3 box of the benchmark being the
//   extreme case -- a slice stores, instead of 8 (W + 1) bytes of fp64 per row, ONE 8-byte word per
//   row holding W + 1 byte indices into a dictionary of the distinct values (bit patterns, so the
//   arithmetic is unchanged):
//        [ idx : 64 x u64 (byte 0 = ext, byte k+1 = slot k) ][ col : W x 64 x i32 ]   (512 + 256 W bytes)
//   The kernel keeps the dictionary in LDS (2 KiB per block).  At W = 6 a row then streams
//   24 + 8 + 8 + 8 = 48 bytes instead of 96.  Operators that do not qualify keep the fp64 records.
// Value + offset dictionaries (format 2): if, in addition, the column offsets col - row take at most
//   256 distinct values (a mesh numbered block-wise: the box has 7 -- 0, +-1, +-nx, +-nx ny -- a z-slab of
//   it with halo planes a few more), the columns become byte indices as well and a row is ONE 16-byte word
//        byte 0 = ext, bytes 1..7 = weight of slot 0..6, bytes 8..14 = offset of slot 0..6
//   so a slice is a 1 KiB record whatever its width, and a row streams 16 + 8 + 8 = 32 bytes.
// Paired rows (format 3, the default when it applies): with a third of the bytes the format-2 kernel is no
//   longer HBM-bound but bound by the number of vector-memory instructions a row costs (one 8-byte gather per
//   neighbour through the texture-address path; halving that count in a diagnostic build took 16 % off the
//   kernel).  Format 3 gives each LANE two consecutive rows (2p, 2p+1) whose neighbour lists are merged into
//   one list of column offsets (the shortest common supersequence of the two rows' offset lists, <= 7 long;
//   a row that lacks an offset of the merged list gets weight 0 there, which leaves its sum unchanged): one
//   16-byte load then fetches x[2p + off], x[2p + 1 + off] -- the neighbour of BOTH rows -- and x_i, y_i move
//   as 16-byte pairs too.  Record of a 128-row group: [64 x (weights of row 2p : u64, of row 2p+1 : u64)]
//   [64 x offsets : u64], index bytes pre-scaled (value index * 8, offset index * 4) so a bit-field extract is
//   the LDS byte address; 12 + 8 + 8 = 28 bytes per row and 10 instead of 18 vector-memory instructions per
//   row pair.  Needs <= 32 distinct values, <= 64 distinct offsets, < 2^28 columns, and every 16-byte gather in
//   bounds of [-kVecGuard, n + 3] (vectors carry a zero guard in front and zero padding behind).  A weight-0
//   slot still reads its partner's neighbour, so it contributes 0 * (x_nb - x_i): exact for finite x; a
//   non-finite x_nb reaches one more row than the face loop would carry it to.
// Arithmetic:  y_i = beta x_i + alpha ( sum_k w_ik (x[col_ik] - x_i) + ext_i x_i )
//   -- the difference form of the reference's flux  (c[out] - c[in]), which keeps the
//   cancellation behaviour of the face loop (no large diagonal * x_i term).
// Algorithmic bytes per apply: 8N (x) + 8N (y) + 8N (ext) + 12 nnz.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#include <hip/hip_ext.h>

#include "common.hpp"
#include "ticket_device.hpp"
#include "ipc_device.hpp"

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
  int nt_y = 1;                           // y stored non-temporally (0: it may stay in the Infinity Cache for the consumer)
  int rec_by_pos = 0;                     // paired records stored in slice-LIST order (the boundary groups of a mixed operator)
  const unsigned long long *types = nullptr;  // format 5: the (<= 32) distinct weight words of the operator's rows
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

// 64-lane sum with DPP moves (VALU only; __shfl_down compiles to ds_bpermute, which costs a
// trip through the LDS crossbar per step).  The total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_mov<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x114, 0xf>(v);  // row_shr:4
  v += dpp_mov<0x118, 0xf>(v);  // row_shr:8   -> lanes 12..15 of each row hold the row sum
  v += dpp_mov<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_mov<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return v;
}


// x[c], or the block's LDS copy of it when VARIANT == 1 and c lies in the block's own 256 rows.
template <int VARIANT>
__device__ __forceinline__ double gather_x(const double *__restrict__ x, int c, const double *xwin, int64_t row0) {
  if (VARIANT == 1) {
    const int64_t d = (int64_t)c - row0;
    return ((uint64_t)d < (uint64_t)kBlock) ? xwin[d] : x[c];
  }
  return x[c];
}

// sum_k w_k (x[col_k] - x_i) over slots [S0, S0 + W) of a record whose slice has `width` slots
// (W compile-time, S0 even).  Pairs are read as int2 / double2, an odd last slot unpaired.
template <bool NT, int VARIANT, int W>
__device__ __forceinline__ double row_sum(const char *rec, int width, int lane, const double *__restrict__ x,
                                          double xi, const double *xwin, int64_t row0, int s0 = 0) {
  constexpr int NP = W / 2;
  const int npair_total = width >> 1;
  const int2v *cp2 = reinterpret_cast<const int2v *>(rec + kExtBytes) + lane + (s0 >> 1) * kWave;
  const char *vbase = rec + kExtBytes + (int64_t)width * (kWave * 4);
  const double2v *vp2 = reinterpret_cast<const double2v *>(vbase) + lane + (s0 >> 1) * kWave;
  int2v c[NP > 0 ? NP : 1];
  double2v v[NP > 0 ? NP : 1];
  int ct = 0;
  double vt = 0.0;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    c[q] = NT ? __builtin_nontemporal_load(cp2 + q * kWave) : cp2[q * kWave];
    v[q] = NT ? __builtin_nontemporal_load(vp2 + q * kWave) : vp2[q * kWave];
  }
  if (W & 1) {  // the slice's unpaired last slot
    ct = ld_i<NT>(reinterpret_cast<const int *>(rec + kExtBytes + (int64_t)npair_total * (kWave * 8)) + lane);
    vt = ld_d<NT>(reinterpret_cast<const double *>(vbase + (int64_t)npair_total * (kWave * 16)) + lane);
  }
  double xg[W > 0 ? W : 1];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    xg[2 * q] = gather_x<VARIANT>(x, c[q].x, xwin, row0);
    xg[2 * q + 1] = gather_x<VARIANT>(x, c[q].y, xwin, row0);
  }
  if (W & 1) xg[W - 1] = gather_x<VARIANT>(x, ct, xwin, row0);
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    acc += v[q].x * (xg[2 * q] - xi);
    acc += v[q].y * (xg[2 * q + 1] - xi);
  }
  if (W & 1) acc += vt * (xg[W - 1] - xi);
  return acc;
}

// Rows wider than 8 slots: chunks of 8, then the remainder.
template <bool NT, int VARIANT>
__device__ __forceinline__ double row_sum_wide(const char *rec, int width, int lane, const double *__restrict__ x,
                                               double xi, const double *xwin, int64_t row0) {
  double acc = 0.0;
  int s0 = 0;
  for (; s0 + 8 <= width; s0 += 8) acc += row_sum<NT, VARIANT, 8>(rec, width, lane, x, xi, xwin, row0, s0);
  switch (width - s0) {
    case 1: acc += row_sum<NT, VARIANT, 1>(rec, width, lane, x, xi, xwin, row0, s0); break;
    case 2: acc += row_sum<NT, VARIANT, 2>(rec, width, lane, x, xi, xwin, row0, s0); break;
    case 3: acc += row_sum<NT, VARIANT, 3>(rec, width, lane, x, xi, xwin, row0, s0); break;
    case 4: acc += row_sum<NT, VARIANT, 4>(rec, width, lane, x, xi, xwin, row0, s0); break;
    case 5: acc += row_sum<NT, VARIANT, 5>(rec, width, lane, x, xi, xwin, row0, s0); break;
    case 6: acc += row_sum<NT, VARIANT, 6>(rec, width, lane, x, xi, xwin, row0, s0); break;
    case 7: acc += row_sum<NT, VARIANT, 7>(rec, width, lane, x, xi, xwin, row0, s0); break;
    default: break;
  }
  return acc;
}

// Value-dictionary record: columns as in row_sum, the weight of slot k is dict[byte k + 1 of iw] (LDS).
template <bool NT, int W>
__device__ __forceinline__ double row_sum_cv(const char *rec, int width, int lane, const double *__restrict__ x,
                                             double xi, uint64_t iw, const double *dict) {
  constexpr int NP = W / 2;
  const int npair_total = width >> 1;
  const int2v *cp2 = reinterpret_cast<const int2v *>(rec + kExtBytes) + lane;
  int2v c[NP > 0 ? NP : 1];
  int ct = 0;
#pragma unroll
  for (int q = 0; q < NP; ++q) c[q] = NT ? __builtin_nontemporal_load(cp2 + q * kWave) : cp2[q * kWave];
  if (W & 1) ct = ld_i<NT>(reinterpret_cast<const int *>(rec + kExtBytes + (int64_t)npair_total * (kWave * 8)) + lane);
  double xg[W > 0 ? W : 1];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    xg[2 * q] = x[c[q].x];
    xg[2 * q + 1] = x[c[q].y];
  }
  if (W & 1) xg[W - 1] = x[ct];
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < W; ++k) acc += dict[(unsigned)(iw >> (8 * (k + 1))) & 0xffu] * (xg[k] - xi);
  return acc;
}

// One wavefront per slice, one row per lane, 4 slices per 256-thread block.
//   NT      : record / y traffic marked non-temporal so it does not evict x from L2 (+15 %).
//   DOT     : epilogue writes per-block partials of <w, y> and <y, y> (fused reductions).
//   VARIANT : 0 gathers x straight from global memory (L1/L2/Infinity Cache serve the reuse);
//             1 stages the block's own 256 x rows in LDS and reads in-window neighbours there
//               (measured: no gain over 0 -- the +-1 neighbours already hit L1).
template <bool NT, bool DOT, int VARIANT, bool XCD>
__global__ __launch_bounds__(kBlock) void spmv_sell_kernel(SellArgs A, Scal alpha_s, Scal beta_s,
                                                           const double *__restrict__ x,
                                                           double *__restrict__ y,
                                                           const int *__restrict__ slice_list,
                                                           int64_t n_launch_slices, DotArgs dot,
                                                           const int *done) {
  // The `done` predicate is only needed before the first store: issue its (scalar) load now and
  // test it after the gathers, so it never sits at the head of a wave's dependency chain.
  const int done_flag = done ? *done : 0;
  __shared__ double xwin[VARIANT == 1 ? kBlock : 1];
  __shared__ double dict_s[VARIANT == 2 ? kDictSize : 1];
  if (VARIANT == 2) {
    static_assert(kDictSize == kBlock, "one dictionary entry per thread");
    dict_s[threadIdx.x] = A.dict[threadIdx.x];
    __syncthreads();
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: slice math runs on the SALU
  const int bidx = (int)blockIdx.x;
  const int lb = XCD ? (A.xcd_group > 1 ? xcd_remap_grouped(bidx, gridDim.x, A.xcd_group) : xcd_remap(bidx, gridDim.x))
                     : bidx;
  const int64_t sl = (int64_t)lb * (kBlock / kWave) + wave;
  const bool wave_active = sl < n_launch_slices;
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);

  double yi = 0.0, wi = 0.0;
  int64_t row = 0;
  bool valid = false;
  double xi = 0.0;
  int64_t slice = 0;
  if (wave_active) {
    slice = slice_list ? (int64_t)slice_list[sl] : sl;
    row = slice * kWave + lane;
    valid = row < A.n_rows;
    xi = valid ? x[row] : 0.0;
    if (DOT && dot.w) wi = (dot.w == x) ? xi : (valid ? dot.w[row] : 0.0);  // early: off the tail of the chain
  }
  int64_t row0 = 0;
  if (VARIANT == 1) {
    // Only meaningful when the block's 4 slices are consecutive (no slice list).
    row0 = (int64_t)lb * kBlock;
    xwin[threadIdx.x] = xi;
    __syncthreads();
  }
  if (wave_active) {
    constexpr int kSlot = (VARIANT == 2) ? kColSlotBytes : kSlotBytes;
    int64_t base;
    int width;
    if (A.uniform_width > 0) {
      width = A.uniform_width;
      base = slice * (int64_t)(kExtBytes + kSlot * width);
    } else {
      base = A.slice_off[slice];
      width = (int)((A.slice_off[slice + 1] - base - kExtBytes) / kSlot);
    }
    const char *rec = A.pack + base;
    double ext, acc;
    if (VARIANT == 2) {
      const uint64_t *ip = reinterpret_cast<const uint64_t *>(rec) + lane;
      const uint64_t iw = NT ? __builtin_nontemporal_load(ip) : *ip;
      ext = dict_s[(unsigned)iw & 0xffu];
      switch (width) {  // build_op guarantees width <= 7 for these records
        case 0: acc = 0.0; break;
        case 1: acc = row_sum_cv<NT, 1>(rec, 1, lane, x, xi, iw, dict_s); break;
        case 2: acc = row_sum_cv<NT, 2>(rec, 2, lane, x, xi, iw, dict_s); break;
        case 3: acc = row_sum_cv<NT, 3>(rec, 3, lane, x, xi, iw, dict_s); break;
        case 4: acc = row_sum_cv<NT, 4>(rec, 4, lane, x, xi, iw, dict_s); break;
        case 5: acc = row_sum_cv<NT, 5>(rec, 5, lane, x, xi, iw, dict_s); break;
        case 6: acc = row_sum_cv<NT, 6>(rec, 6, lane, x, xi, iw, dict_s); break;
        default: acc = row_sum_cv<NT, 7>(rec, 7, lane, x, xi, iw, dict_s); break;
      }
    } else {
    ext = ld_d<NT>(reinterpret_cast<const double *>(rec) + lane);
    // The width is wave-uniform: dispatch to a body with the width as a compile-time constant,
    // so all (col, val) loads of the row are issued back to back, then all gathers, then the
    // FMAs -- no branch (and no s_waitcnt) between the gathers of one row.
    switch (width) {
      case 0: acc = 0.0; break;
      case 1: acc = row_sum<NT, VARIANT, 1>(rec, 1, lane, x, xi, xwin, row0); break;
      case 2: acc = row_sum<NT, VARIANT, 2>(rec, 2, lane, x, xi, xwin, row0); break;
      case 3: acc = row_sum<NT, VARIANT, 3>(rec, 3, lane, x, xi, xwin, row0); break;
      case 4: acc = row_sum<NT, VARIANT, 4>(rec, 4, lane, x, xi, xwin, row0); break;
      case 5: acc = row_sum<NT, VARIANT, 5>(rec, 5, lane, x, xi, xwin, row0); break;
      case 6: acc = row_sum<NT, VARIANT, 6>(rec, 6, lane, x, xi, xwin, row0); break;
      case 7: acc = row_sum<NT, VARIANT, 7>(rec, 7, lane, x, xi, xwin, row0); break;
      case 8: acc = row_sum<NT, VARIANT, 8>(rec, 8, lane, x, xi, xwin, row0); break;
      default: acc = row_sum_wide<NT, VARIANT>(rec, width, lane, x, xi, xwin, row0); break;
    }
    }
    yi = (A.accumulate ? (valid ? y[row] : 0.0) : beta * xi) + alpha * (acc + ext * xi);
    if (valid && !done_flag) {
      if (NT) __builtin_nontemporal_store(yi, y + row);
      else y[row] = yi;
    }
    if (!valid) yi = 0.0;
  }
  if (done_flag) return;  // block-uniform
  if (DOT) {
    // One partial per WAVE (64-lane DPP tree, lane 63 stores): no LDS, no block barrier.
    // (A per-block partial with __syncthreads cost 6 % of the kernel: every wave of a block had
    // to outlive its slowest sibling.)  The 4x longer partial arrays are folded by the two-pass
    // final reduction in solvers.hip.
    double a = dot.w ? wi * yi : 0.0, b = dot.yy ? yi * yi : 0.0;
    a = wave_sum_to_lane63(a);
    if (dot.yy) b = wave_sum_to_lane63(b);
    if (lane == kWave - 1) {
      const int slot = dot.block_offset + (int)blockIdx.x * (kBlock / kWave) + wave;
      dot.partials[slot] = a;
      if (dot.yy) dot.partials[dot.nblocks_total + slot] = b;
    }
  }
}

// Value-dictionary records of one uniform width W: SPW consecutive slices per wavefront.
// Compared with the general kernel above: (i) every wave keeps its own copy of the dictionary in LDS
// (written and read by the same wave: no block barrier between the record loads and the lookups);
// (ii) all record loads of the wave's SPW slices are issued first, then all SPW x W gathers, then
// the lookups and FMAs -- with half the bytes per row the kernel is bound by memory-level
// parallelism per wave rather than by HBM, and a wave with one slice had too little in flight;
// (iii) one fused-dot partial per wave covers SPW slices.
// FMT 1: [idx u64][columns i32] records; FMT 2: one 16-byte word per row, columns = row + offs[byte].
// Branch-free but for the store: rows past the end (ragged last slice) and slices past the launch are
// redirected to valid memory (row n-1, slice 0) and masked at the store / in the partials.
template <bool DOT, int W, int SPW, int FMT>
__global__ __launch_bounds__(kBlock) void spmv_dict_kernel(SellArgs A, Scal alpha_s, Scal beta_s,
                                                           const double *__restrict__ x, double *__restrict__ y,
                                                           const int *__restrict__ slice_list,
                                                           int64_t n_launch_slices, DotArgs dot, const int *done) {
  constexpr int NP = W / 2;
  constexpr int64_t kRec = FMT == 2 ? (int64_t)kWave * 16 : kExtBytes + (int64_t)kColSlotBytes * W;
  const int done_flag = done ? *done : 0;
  __shared__ double dict_s[kBlock / kWave][kDictSize];
  __shared__ int offs_s[FMT == 2 ? kBlock / kWave : 1][FMT == 2 ? kDictSize : 1];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *dw = dict_s[wave];
  int *ow = offs_s[FMT == 2 ? wave : 0];
  const int bidx = (int)blockIdx.x;
  const int lb = A.xcd_group > 1 ? xcd_remap_grouped(bidx, gridDim.x, A.xcd_group)
                                 : (A.xcd_group == 1 ? xcd_remap(bidx, gridDim.x) : bidx);
  const int64_t sl0 = ((int64_t)lb * (kBlock / kWave) + wave) * SPW;
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const int64_t last_row = A.n_rows - 1;
  const bool w_is_x = DOT && dot.w == x;
  const bool w_load = DOT && dot.w != nullptr && !w_is_x;

  uint64_t iw[SPW], jw[SPW];
  int2v c[SPW][NP > 0 ? NP : 1];
  int ct[SPW];
  double xi[SPW], wi[SPW], yo[SPW];
  int64_t row[SPW];
  bool valid[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    const int64_t sl = sl0 + u;
    const bool active = sl < n_launch_slices;  // wave-uniform
    const int64_t slice = slice_list ? (int64_t)slice_list[active ? sl : 0] : (active ? sl : 0);
    const int64_t r = slice * kWave + lane;
    valid[u] = active && r <= last_row;
    row[u] = r <= last_row ? r : last_row;
    const char *rec = A.pack + slice * kRec;
    ct[u] = 0;
    jw[u] = 0;
    if (FMT == 2) {
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      const u64x2 word = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(rec) + lane);
      iw[u] = word.x, jw[u] = word.y;
    } else {
      iw[u] = __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(rec) + lane);
      const int2v *cp2 = reinterpret_cast<const int2v *>(rec + kExtBytes) + lane;
#pragma unroll
      for (int q = 0; q < NP; ++q) c[u][q] = __builtin_nontemporal_load(cp2 + q * kWave);
      if (W & 1)
        ct[u] = __builtin_nontemporal_load(reinterpret_cast<const int *>(rec + kExtBytes + NP * (kWave * 8)) + lane);
    }
    xi[u] = x[row[u]];
    wi[u] = 0.0;
    yo[u] = 0.0;
    if (A.accumulate) yo[u] = y[row[u]];
    if (w_load) wi[u] = dot.w[row[u]];  // (w == x, CG's <p, Ap>, reuses xi at the end: no copy here, a
  }                                      //  copy would wait for xi in the middle of the load issue)
  // The wave's own copy of the tables, requested AFTER the record loads (memory returns in order, so
  // the copy costs no extra round trip).  The tables are allocated with kDictSize entries: the first
  // 64 are copied unconditionally, the rest under a scalar branch that operators with a handful of
  // distinct values never take.
  {
    const int o0 = FMT == 2 ? A.offs[lane] : 0;
    const double d0 = A.dict[lane];
    if (FMT == 2) ow[lane] = o0;
    dw[lane] = d0;
    if (A.dict_size > kWave || A.offs_size > kWave) {
#pragma unroll
      for (int j = 1; j < kDictSize / kWave; ++j) {
        if (FMT == 2) ow[lane + j * kWave] = A.offs[lane + j * kWave];
        dw[lane + j * kWave] = A.dict[lane + j * kWave];
      }
    }
  }
  double xg[SPW][W > 0 ? W : 1];
  if (FMT == 2) __builtin_amdgcn_wave_barrier();  // the wave's offset table is complete (same-wave LDS order)
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    if (FMT == 2) {
      // padding slots (and every slot of a row past the end) carry offset 0 and weight 0
      const double *xr = x + row[u];
#pragma unroll
      for (int k = 0; k < W; ++k) xg[u][k] = xr[ow[(unsigned)(jw[u] >> (8 * k)) & 0xffu]];
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        xg[u][2 * q] = x[c[u][q].x];
        xg[u][2 * q + 1] = x[c[u][q].y];
      }
      if (W & 1) xg[u][W - 1] = x[ct[u]];
    }
  }
  __builtin_amdgcn_wave_barrier();  // the wave's dictionary copy is complete (same-wave LDS order)
  double da = 0.0, db = 0.0;
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < W; ++k) acc += dw[(unsigned)(iw[u] >> (8 * (k + 1))) & 0xffu] * (xg[u][k] - xi[u]);
    const double ext = dw[(unsigned)iw[u] & 0xffu];
    double yi = (A.accumulate ? yo[u] : beta * xi[u]) + alpha * (acc + ext * xi[u]);
    if (valid[u] && !done_flag) __builtin_nontemporal_store(yi, y + row[u]);
    yi = valid[u] ? yi : 0.0;
    if (DOT) {
      da += (w_is_x ? xi[u] : wi[u]) * yi;
      db += yi * yi;
    }
  }
  if (done_flag) return;
  if (DOT) {
    double a = dot.w ? da : 0.0;
    a = wave_sum_to_lane63(a);
    if (dot.yy) db = wave_sum_to_lane63(db);
    if (lane == kWave - 1) {
      const int slot = dot.block_offset + (int)blockIdx.x * (kBlock / kWave) + wave;
      dot.partials[slot] = a;
      if (dot.yy) dot.partials[dot.nblocks_total + slot] = db;
    }
  }
}

// Format 3: one lane = rows (2p, 2p + 1), one wave = 128 rows.  See the header comment.
// HALO: columns >= n_rows are not read from x's tail but from the peer window (each value polled until its tag is
// this exchange's); the kernel's last block acknowledges the planes.
template <bool DOT, int W, bool HALO>
__global__ __launch_bounds__(kBlock) void spmv_pair_kernel(SellArgs A, Scal alpha_s, Scal beta_s,
                                                           const double *__restrict__ x, double *__restrict__ y,
                                                           const int *__restrict__ slice_list,
                                                           int64_t n_launch_slices, DotArgs dot, const int *done,
                                                           IpcRecvArgs H) {
  const int done_flag = done ? *done : 0;
  __shared__ double dict_sh[32];
  __shared__ int offs_sh[64];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bidx = (int)blockIdx.x;
  const int lb = A.xcd_group > 1 ? xcd_remap_grouped(bidx, gridDim.x, A.xcd_group)
                                 : (A.xcd_group == 1 ? xcd_remap(bidx, gridDim.x) : bidx);
  const int64_t sl = (int64_t)lb * (kBlock / kWave) + wave;
  const bool active = sl < n_launch_slices;  // wave-uniform
  const uint32_t slice = (uint32_t)(slice_list ? slice_list[active ? sl : 0] : (active ? sl : 0));
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const uint32_t last_row = (uint32_t)(A.n_rows - 1);
  const bool w_is_x = DOT && dot.w == x;
  const bool w_load = DOT && dot.w != nullptr && !w_is_x;
  const char *xb = reinterpret_cast<const char *>(x);
  const char *xg_base = xb - (size_t)kVecGuard * 8;  // start of the zero guard in front of x
  char *yb = reinterpret_cast<char *>(y);

  const uint32_t r0 = slice * (2 * kWave) + 2 * lane;  // row A; row B = r0 + 1
  const bool valid_a = active && r0 <= last_row, valid_b = active && r0 + 1 <= last_row;
  const uint32_t rc = r0 <= last_row ? r0 : (last_row & ~1u);  // pairs past the end re-read the last pair
  const char *rec = A.pack + (size_t)(A.rec_by_pos ? (uint32_t)(active ? sl : 0) : slice) * kPairRecBytes;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 vw = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(rec) + lane);
  const uint64_t jw = __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(rec + 2 * kWave * 8) + lane);
  const double2v xi = *reinterpret_cast<const double2v *>(xb + (size_t)(rc << 3));
  double2v yo = {0.0, 0.0}, wi = {0.0, 0.0};
  if (A.accumulate) yo = *reinterpret_cast<const double2v *>(yb + (size_t)(rc << 3));
  if (w_load) wi = *reinterpret_cast<const double2v *>(reinterpret_cast<const char *>(dot.w) + (size_t)(rc << 3));
  {
    // tables: <= 64 entries each, one load per lane (the allocations hold kDictSize entries).  One copy per
    // block at a fixed LDS address; every wave stores the same words before it reads them: no barrier.
    const int o0 = A.offs[lane];
    const double d0 = A.dict[lane & 31];
    offs_sh[lane] = o0;
    if (lane < 32) dict_sh[lane] = d0;
  }
  __builtin_amdgcn_wave_barrier();  // this wave's copy of the tables is complete (same-wave LDS order)
  double2v xg[W > 0 ? W : 1];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const unsigned ob = (unsigned)(jw >> (8 * k)) & 0xffu;  // = offset index * 4
    const int off = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(offs_sh) + ob);
    // both rows' neighbour.  The index is biased by the guard so that it is never negative (the host checked
    // rc + off >= -kVecGuard): the address is a uniform base plus an UNSIGNED 32-bit byte offset.
    const int ca = (int)rc + off;  // column of row A's neighbour; row B's is ca + 1
    if (HALO && ca + 1 >= (int)A.n_rows) {
      // (an absent slot's column may point anywhere: beyond the halo rows it reads as 0, like x's zero padding)
      const int ha = ca - (int)A.n_rows, hb = ha + 1;
      double va, vb;
      ipc_halo_pair(H.w, H.rp, ha, hb, H.n_halo, &va, &vb);
      xg[k].x = ha < 0 ? x[ca] : va;
      xg[k].y = vb;
    } else {
      xg[k] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((rc + (uint32_t)(off + kVecGuard)) << 3));
    }
  }
  double acc_a = 0.0, acc_b = 0.0;
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const unsigned ba = (unsigned)(vw.x >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(vw.y >> (8 * (k + 1))) & 0xffu;
    acc_a += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ba) * (xg[k].x - xi.x);
    acc_b += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + bb) * (xg[k].y - xi.y);
  }
  const double ext_a = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw.x & 0xffu));
  const double ext_b = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw.y & 0xffu));
  double2v yi;
  yi.x = (A.accumulate ? yo.x : beta * xi.x) + alpha * (acc_a + ext_a * xi.x);
  yi.y = (A.accumulate ? yo.y : beta * xi.y) + alpha * (acc_b + ext_b * xi.y);
  if (!done_flag) {
    if (valid_b) __builtin_nontemporal_store(yi, reinterpret_cast<double2v *>(yb + (size_t)(rc << 3)));
    else if (valid_a) y[rc] = yi.x;  // the odd last row
  }
  if (HALO) ipc_halo_ack_last_block(H.w, H.rp);  // (every thread of every block gets here)
  if (done_flag) return;
  if (DOT) {
    yi.x = valid_a ? yi.x : 0.0;
    yi.y = valid_b ? yi.y : 0.0;
    double a = dot.w ? (w_is_x ? xi.x : wi.x) * yi.x + (w_is_x ? xi.y : wi.y) * yi.y : 0.0;
    double b = yi.x * yi.x + yi.y * yi.y;
    a = wave_sum_to_lane63(a);
    if (dot.yy) b = wave_sum_to_lane63(b);
    if (lane == kWave - 1) {
      const int slot = dot.block_offset + (int)blockIdx.x * (kBlock / kWave) + wave;
      dot.partials[slot] = a;
      if (dot.yy) dot.partials[dot.nblocks_total + slot] = b;
    }
  }
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
// TYPED (format 5): the rows' 8-byte weight words take at most 32 distinct values (a box with uniform spacing: the 27
// combinations of "which walls does the cell touch") -- a row stores ONE byte, the index of its word in a table held
// in LDS beside the value table: 1 + 8 + 8 = 17 B/row.
constexpr int kTypedRecBytes = 2 * kWave;
constexpr int kMaxRowTypes = 32;
// G: consecutive 128-row groups per wavefront (1 or 2).  With two, every load of both groups is in flight before the
// first use, the prologue (tile mapping, tables) and the fused-dot's wave reduction are paid once per 256 rows.
template <bool DOT, int K, int M1, bool TYPED, int G>
__global__ __launch_bounds__(kBlock) void spmv_canon_kernel(SellArgs A, CanonArgs C, Scal alpha_s, Scal beta_s,
                                                            const double *__restrict__ x, double *__restrict__ y,
                                                            const int *__restrict__ slice_list,
                                                            int64_t n_launch_slices, DotArgs dot, const int *done) {
  const int done_flag = done ? *done : 0;
  __shared__ double dict_sh[32];
  __shared__ unsigned long long types_sh[TYPED ? kMaxRowTypes : 1];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bidx = C.reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
  int lb = bidx;
  if (C.xcd_shift >= 0) {  // xcd_remap_grouped for a power-of-two run length
    if (bidx < C.xcd_full) {
      const int xcd = bidx & (kNumXcd - 1), j = bidx >> 3;
      lb = ((((j >> C.xcd_shift) << 3) + xcd) << C.xcd_shift) + (j & ((1 << C.xcd_shift) - 1));
    }
  } else if (A.xcd_group > 1) {
    lb = xcd_remap_grouped(bidx, gridDim.x, A.xcd_group);
  } else if (A.xcd_group == 1) {
    lb = xcd_remap(bidx, gridDim.x);
  }
  const int64_t sl0 = ((int64_t)lb * (kBlock / kWave) + wave) * G;
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const uint32_t last_row = (uint32_t)(A.n_rows - 1);
  const bool w_is_x = DOT && dot.w == x;
  const bool w_load = DOT && dot.w != nullptr && !w_is_x;
  const char *xb = reinterpret_cast<const char *>(x);
  const char *xg_base = xb - (size_t)kVecGuard * 8;  // start of the zero guard in front of x
  char *yb = reinterpret_cast<char *>(y);
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

  // Issue order matters (loads return in order; a wait for one load waits for every earlier one): the tables first
  // -- their LDS copies are needed before anything else can be consumed -- then per group the record, the own rows,
  // all gathers and the two outer neighbours back to back; nothing is consumed before the last load is in flight.
  const double dict_word = A.dict[lane & 31];
  unsigned long long type_word = 0ull;
  if (TYPED) type_word = A.types[lane & (kMaxRowTypes - 1)];
  bool valid_a[G], valid_b[G];
  uint32_t rc[G];
  u64x2 vw[G];
  unsigned type_pair[G];  // (type of row A) | (type of row B) << 8, both pre-scaled by 8
  double2v xi[G], yo[G], wi[G], xg[G][K];
  double e[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const bool active = sl0 + g < n_launch_slices;  // wave-uniform
    const uint32_t slice = (uint32_t)(slice_list ? slice_list[active ? sl0 + g : 0] : (active ? sl0 + g : 0));
    const uint32_t r0 = slice * (2 * kWave) + 2 * lane;  // row A; row B = r0 + 1
    valid_a[g] = active && r0 <= last_row, valid_b[g] = active && r0 + 1 <= last_row;
    rc[g] = r0 <= last_row ? r0 : (last_row & ~1u);  // pairs past the end re-read the last pair
    vw[g] = u64x2{0ull, 0ull};
    type_pair[g] = 0u;
    if (TYPED)
      type_pair[g] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(A.pack + (size_t)slice * kTypedRecBytes) + lane);
    else
      vw[g] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(A.pack + (size_t)slice * kCanonRecBytes) + lane);
    xi[g] = *reinterpret_cast<const double2v *>(xb + (size_t)(rc[g] << 3));
    yo[g] = double2v{0.0, 0.0}, wi[g] = double2v{0.0, 0.0};
    if (A.accumulate) yo[g] = *reinterpret_cast<const double2v *>(yb + (size_t)(rc[g] << 3));
    if (w_load) wi[g] = *reinterpret_cast<const double2v *>(reinterpret_cast<const char *>(dot.w) + (size_t)(rc[g] << 3));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (M1 >= 0 && (k == M1 || k == M1 + 1)) continue;
      int t = (int)rc[g] + C.off[k] + kVecGuard;  // guard-relative, clamped: an absent neighbour may point anywhere
      t = t < 0 ? 0 : t;
      t = t > C.max_gather ? C.max_gather : t;
      // (32-bit byte offset from a uniform base: n_rows + n_halo < 2^28 is a condition of the paired formats)
#if defined(STORM_CANON_EXPERIMENT) && STORM_CANON_EXPERIMENT >= 2  // (measurement only: no gathers either)
      xg[g][k] = xi[g] + (double)t;
#else
      xg[g][k] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((uint32_t)t << 3));
#endif
    }
    e[g] = 0.0;
    if (M1 >= 0 && (lane == 0 || lane == kWave - 1))  // x[rc - 1] of lane 0, x[rc + 2] of lane 63
      e[g] = *reinterpret_cast<const double *>(xg_base + (size_t)((rc[g] + (uint32_t)(kVecGuard + (lane == 0 ? -1 : 2))) << 3));
  }
  if (lane < 32) dict_sh[lane] = dict_word;  // one copy per block, every wave stores the same words: no barrier
  if (TYPED && lane < kMaxRowTypes) types_sh[lane] = type_word;
  __builtin_amdgcn_wave_barrier();  // this wave's copy of the tables is complete (same-wave LDS order)
  double dot_a = 0.0, dot_b = 0.0;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (M1 >= 0) {
      // x[rc - 1] and x[rc + 2] are the neighbouring lanes' own rows
      const double left = dpp_shift<0x138>(xi[g].y);   // wave_shr:1 -- lane i receives lane i - 1
      const double right = dpp_shift<0x130>(xi[g].x);  // wave_shl:1 -- lane i receives lane i + 1
      xg[g][M1 >= 0 ? M1 : 0].x = lane == 0 ? e[g] : left;
      xg[g][M1 >= 0 ? M1 : 0].y = xi[g].x;
      xg[g][M1 >= 0 ? M1 + 1 : 0].x = xi[g].y;
      xg[g][M1 >= 0 ? M1 + 1 : 0].y = lane == kWave - 1 ? e[g] : right;
    }
    if (TYPED) {
      vw[g].x = *reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(types_sh) + (type_pair[g] & 0xffu));
      vw[g].y = *reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(types_sh) + (type_pair[g] >> 8));
    }
    double acc_a = 0.0, acc_b = 0.0;
#ifdef STORM_CANON_EXPERIMENT  // (measurement only: the kernel's memory floor -- no table lookups, one add per neighbour)
#pragma unroll
    for (int k = 0; k < K; ++k) acc_a += xg[g][k].x, acc_b += xg[g][k].y;
    acc_a += __longlong_as_double((long long)vw[g].x), acc_b += __longlong_as_double((long long)vw[g].y);
#else
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned ba = (unsigned)(vw[g].x >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(vw[g].y >> (8 * (k + 1))) & 0xffu;
      acc_a += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ba) * (xg[g][k].x - xi[g].x);
      acc_b += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + bb) * (xg[g][k].y - xi[g].y);
    }
#endif
    const double ext_a = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[g].x & 0xffu));
    const double ext_b = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[g].y & 0xffu));
    double2v yi;
    yi.x = (A.accumulate ? yo[g].x : beta * xi[g].x) + alpha * (acc_a + ext_a * xi[g].x);
    yi.y = (A.accumulate ? yo[g].y : beta * xi[g].y) + alpha * (acc_b + ext_b * xi[g].y);
    if (!done_flag) {
      double2v *yp = reinterpret_cast<double2v *>(yb + (size_t)(rc[g] << 3));
      if (valid_b[g]) { if (A.nt_y) __builtin_nontemporal_store(yi, yp); else *yp = yi; }
      else if (valid_a[g]) y[rc[g]] = yi.x;  // the odd last row
    }
    if (DOT) {
      yi.x = valid_a[g] ? yi.x : 0.0;
      yi.y = valid_b[g] ? yi.y : 0.0;
      // (group by group, rows in order: with G == 1 exactly the sums of the one-group kernel)
      const double a = dot.w ? (w_is_x ? xi[g].x : wi[g].x) * yi.x + (w_is_x ? xi[g].y : wi[g].y) * yi.y : 0.0;
      const double b = yi.x * yi.x + yi.y * yi.y;
      dot_a = g == 0 ? a : dot_a + a;
      dot_b = g == 0 ? b : dot_b + b;
    }
  }
  if (done_flag) return;
  if (DOT) {
    dot_a = wave_sum_to_lane63(dot_a);
    if (dot.yy) dot_b = wave_sum_to_lane63(dot_b);
    if (dot.tickets == nullptr) {
      if (lane == kWave - 1) {
        const int slot = dot.block_offset + bidx * (kBlock / kWave) + wave;
        dot.partials[slot] = dot_a;
        if (dot.yy) dot.partials[dot.nblocks_total + slot] = dot_b;
      }
    } else {  // the reduction finishes here: block partial, then two levels of tickets
      __shared__ double wave_part[2 * (kBlock / kWave)];
      if (lane == kWave - 1) wave_part[wave] = dot_a, wave_part[kBlock / kWave + wave] = dot.yy ? dot_b : 0.0;
      __syncthreads();
      if (wave != 0) return;
      const double mine[2] = {(wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]),
                              (wave_part[4] + wave_part[5]) + (wave_part[6] + wave_part[7])};
      double total[2];
      const TicketArgs t{dot.tickets, dot.partials, dot.part2};
      if (ticket_reduce_wave0<2>(t, mine, dot.yy ? 2 : 1, (unsigned)bidx, gridDim.x, total) && lane == 0) {
        *dot.out0 = total[0];
        if (dot.yy) *dot.out1 = total[1];
      }
    }
  }
}

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
  double *x;
  const double *r;
  double *p_out;
};
template <bool DOT, bool WLOAD, int TZ, int HL, bool FUSE = false>
__global__ __launch_bounds__(kBlock) void spmv_canon_tile_kernel(SellArgs A, CanonTileArgs T, Scal alpha_s, Scal beta_s,
                                                                 const double *__restrict__ x, double *__restrict__ y,
                                                                 DotArgs dot, const int *done, IpcSendArgs S, CgFuseArgs F) {
  if (!FUSE && (int)blockIdx.x < S.sp.n_blocks) {  // the first blocks of a partitioned operator's interior launch send its rows
    ipc_halo_send_block(S.w, S.sp, x, (int)blockIdx.x);
    return;
  }
  if (FUSE && *F.iteration < F.my_iteration) return;  // enqueued past convergence: that iteration never ran
  const int done_flag = done ? *done : 0;
  const double cg_a = FUSE ? *F.ca : 0.0, cg_b = FUSE ? *F.cb : 0.0;
  const char *rb_ = reinterpret_cast<const char *>(F.r);
  extern __shared__ __attribute__((aligned(16))) double tile_sh[];  // [TZ][a + kTileRun + a]
  __shared__ double dict_sh[32];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_tiles = (int)gridDim.x - S.sp.n_blocks, tb = (int)blockIdx.x - S.sp.n_blocks;
  const int bidx = T.reverse ? n_tiles - 1 - tb : tb;
  int zc, yt;
  if (T.per_xcd > 0) {
    const int xcd = bidx & (kNumXcd - 1), j = bidx >> 3;
    zc = j / T.per_xcd;
    yt = xcd * T.per_xcd + (j - zc * T.per_xcd);
  } else {
    zc = bidx / T.tiles_per_plane;
    yt = bidx - zc * T.tiles_per_plane;
  }
  const int a = T.a, b = T.b;
  const int p0 = yt * kTileRun, z0 = T.plane0 + zc * TZ;
  const int ldw = kTileRun + 2 * a;  // doubles per plane of the LDS copy
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const uint32_t last_row = (uint32_t)(A.n_rows - 1);
  const bool w_is_x = DOT && dot.w == x;
  const char *xb = reinterpret_cast<const char *>(x);
  const char *xg_base = xb - (size_t)kVecGuard * 8;
  char *yb = reinterpret_cast<char *>(y);
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

  const double dict_word = A.dict[lane & 31];
  // ---- everything this wave reads from memory, issued back to back: own rows first (the LDS copy waits for them only)
  bool valid_a[TZ][2], valid_b[TZ][2];
  uint32_t rc[TZ][2];
  double2v xi[TZ][2];
#pragma unroll
  for (int t = 0; t < TZ; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int q = p0 + 256 * wave + 128 * g + 2 * lane;                 // row of the plane
      const int64_t row = (int64_t)(z0 + t) * b + q;
      const bool in_plane = q < b && z0 + t < T.plane_end;
      valid_a[t][g] = in_plane && row <= (int64_t)last_row, valid_b[t][g] = in_plane && row + 1 <= (int64_t)last_row;
      rc[t][g] = row <= (int64_t)last_row ? (uint32_t)row : (last_row & ~1u);  // pairs past the end re-read the last pair
      xi[t][g] = *reinterpret_cast<const double2v *>(xb + (size_t)(rc[t][g] << 3));
    }
  if (FUSE) {
    // x += alpha p (the OLD direction), then p' = r + beta p takes p's place in the registers
#pragma unroll
    for (int t = 0; t < TZ; ++t)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        double2v *xp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.x) + (size_t)(rc[t][g] << 3));
        const double2v xv = __builtin_nontemporal_load(xp_);
        const double2v rv = *reinterpret_cast<const double2v *>(rb_ + (size_t)(rc[t][g] << 3));
        double2v xn, pn;
        xn.x = __builtin_fma(cg_a, xi[t][g].x, xv.x), xn.y = __builtin_fma(cg_a, xi[t][g].y, xv.y);
        pn.x = __builtin_fma(cg_b, xi[t][g].x, rv.x), pn.y = __builtin_fma(cg_b, xi[t][g].y, rv.y);
        if (valid_b[t][g]) __builtin_nontemporal_store(xn, xp_);
        else if (valid_a[t][g]) F.x[rc[t][g]] = xn.x;
        xi[t][g] = pn;
        if (!done_flag) {
          double2v *pp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.p_out) + (size_t)(rc[t][g] << 3));
          if (valid_b[t][g]) __builtin_nontemporal_store(pn, pp_);
          else if (valid_a[t][g]) F.p_out[rc[t][g]] = pn.x;
        }
      }
    if (done_flag) return;  // converged in that iteration: x is final, no new direction, no apply
  }
  const char *rg_base = FUSE ? rb_ - (size_t)kVecGuard * 8 : nullptr;
  double2v halo[HL];
  int halo_at[HL];  // LDS index (doubles) of the pair, -1: none
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const unsigned h = threadIdx.x + (unsigned)kBlock * i;                // pair h of the tile's TZ * a halo pairs
    const unsigned t = __umulhi(h, T.a_magic), u = h - t * (unsigned)a;   // plane, pair within the plane's halo
    const bool on = t < (unsigned)TZ;
    const int jj = (int)(2 * u) < a ? (int)(2 * u) - a : kTileRun + (int)(2 * u) - a;  // tile-relative row: [-a, 0) or [1024, 1024 + a)
    int64_t gi = (int64_t)(z0 + (int)t) * b + p0 + jj + kVecGuard;        // guard-relative, clamped like every gather
    gi = gi < 0 ? 0 : gi;
    gi = gi > (int64_t)T.max_gather ? (int64_t)T.max_gather : gi;
    halo_at[i] = on ? (int)t * ldw + a + jj : -1;
    halo[i] = double2v{0.0, 0.0};
    if (on) halo[i] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((uint32_t)gi << 3));
    if (FUSE && on) {  // the halo row's new direction, with its owner's expression
      const double2v rv = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)gi << 3));
      halo[i].x = __builtin_fma(cg_b, halo[i].x, rv.x), halo[i].y = __builtin_fma(cg_b, halo[i].y, rv.y);
    }
  }
  u64x2 vw[TZ][2];
  double2v wi[WLOAD ? TZ : 1][2];
#pragma unroll
  for (int t = 0; t < TZ; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      vw[t][g] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(A.pack + (size_t)(rc[t][g] << 3)));
      if (WLOAD) wi[t][g] = *reinterpret_cast<const double2v *>(reinterpret_cast<const char *>(dot.w) + (size_t)(rc[t][g] << 3));
    }
  double2v xlo[2], xhi[2];  // the planes below the first and above the last one of the tile
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    int lo = (int)rc[0][g] - b + kVecGuard, hi = (int)rc[TZ - 1][g] + b + kVecGuard;
    lo = lo < 0 ? 0 : lo;
    hi = hi > T.max_gather ? T.max_gather : hi;
    xlo[g] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((uint32_t)lo << 3));
    xhi[g] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((uint32_t)hi << 3));
    if (FUSE) {
      const double2v rl = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)lo << 3));
      const double2v rh = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)hi << 3));
      xlo[g].x = __builtin_fma(cg_b, xlo[g].x, rl.x), xlo[g].y = __builtin_fma(cg_b, xlo[g].y, rl.y);
      xhi[g].x = __builtin_fma(cg_b, xhi[g].x, rh.x), xhi[g].y = __builtin_fma(cg_b, xhi[g].y, rh.y);
    }
  }
  // ---- the LDS copy of the tile's x (own rows + halo rows), one barrier
  if (lane < 32) dict_sh[lane] = dict_word;  // every wave stores the same words
#pragma unroll
  for (int t = 0; t < TZ; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g)
      *reinterpret_cast<double2v *>(&tile_sh[t * ldw + a + 256 * wave + 128 * g + 2 * lane]) = xi[t][g];
#pragma unroll
  for (int i = 0; i < HL; ++i)
    if (halo_at[i] >= 0) *reinterpret_cast<double2v *>(&tile_sh[halo_at[i]]) = halo[i];
  __syncthreads();
  double dot_a = 0.0, dot_b = 0.0;
#pragma unroll
  for (int t = 0; t < TZ; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int at = t * ldw + a + 256 * wave + 128 * g + 2 * lane;  // this pair in the LDS copy
      double2v xg[6];
      xg[0] = t == 0 ? xlo[g] : xi[t == 0 ? 0 : t - 1][g];
      xg[5] = t == TZ - 1 ? xhi[g] : xi[t == TZ - 1 ? t : t + 1][g];
      xg[1] = *reinterpret_cast<const double2v *>(&tile_sh[at - a]);
      xg[4] = *reinterpret_cast<const double2v *>(&tile_sh[at + a]);
      double el = 0.0;
      if (lane == 0) el = tile_sh[at - 1];
      if (lane == kWave - 1) el = tile_sh[at + 2];
      const double left = dpp_shift<0x138>(xi[t][g].y);   // wave_shr:1 -- lane i receives lane i - 1
      const double right = dpp_shift<0x130>(xi[t][g].x);  // wave_shl:1 -- lane i receives lane i + 1
      xg[2].x = lane == 0 ? el : left;
      xg[2].y = xi[t][g].x;
      xg[3].x = xi[t][g].y;
      xg[3].y = lane == kWave - 1 ? el : right;
      double acc_a = 0.0, acc_b = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const unsigned ba = (unsigned)(vw[t][g].x >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(vw[t][g].y >> (8 * (k + 1))) & 0xffu;
        acc_a += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ba) * (xg[k].x - xi[t][g].x);
        acc_b += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + bb) * (xg[k].y - xi[t][g].y);
      }
      const double ext_a = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[t][g].x & 0xffu));
      const double ext_b = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[t][g].y & 0xffu));
      // (spelled out: beta x rounded on its own, then the two FMAs the plain kernel's expression contracts to --
      //  `(accumulate ? y : beta x) + alpha (acc + ext x)` -- so that both kernels round alike)
      double2v yi;
      yi.x = __builtin_fma(alpha, __builtin_fma(ext_a, xi[t][g].x, acc_a), beta * xi[t][g].x);
      yi.y = __builtin_fma(alpha, __builtin_fma(ext_b, xi[t][g].y, acc_b), beta * xi[t][g].y);
      if (!done_flag) {
        double2v *yp = reinterpret_cast<double2v *>(yb + (size_t)(rc[t][g] << 3));
        if (valid_b[t][g]) { if (A.nt_y) __builtin_nontemporal_store(yi, yp); else *yp = yi; }
        else if (valid_a[t][g]) y[rc[t][g]] = yi.x;  // the odd last row
      }
      if (DOT) {
        yi.x = valid_a[t][g] ? yi.x : 0.0;
        yi.y = valid_b[t][g] ? yi.y : 0.0;
        const double2v wv = WLOAD ? wi[WLOAD ? t : 0][g] : xi[t][g];
        const double pa = dot.w ? wv.x * yi.x + wv.y * yi.y : 0.0;
        const double pb = yi.x * yi.x + yi.y * yi.y;
        dot_a = (t == 0 && g == 0) ? pa : dot_a + pa;
        dot_b = (t == 0 && g == 0) ? pb : dot_b + pb;
      }
    }
  (void)w_is_x;
  if (done_flag) return;
  if (DOT) {
    dot_a = wave_sum_to_lane63(dot_a);
    if (dot.yy) dot_b = wave_sum_to_lane63(dot_b);
    if (dot.tickets == nullptr) {
      if (lane == kWave - 1) {
        const int slot = dot.block_offset + bidx * (kBlock / kWave) + wave;
        dot.partials[slot] = dot_a;
        if (dot.yy) dot.partials[dot.nblocks_total + slot] = dot_b;
      }
    } else {  // the reduction finishes here: block partial, then two levels of tickets
      __shared__ double wave_part[2 * (kBlock / kWave)];
      if (lane == kWave - 1) wave_part[wave] = dot_a, wave_part[kBlock / kWave + wave] = dot.yy ? dot_b : 0.0;
      __syncthreads();
      if (wave != 0) return;
      const double mine[2] = {(wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]),
                              (wave_part[4] + wave_part[5]) + (wave_part[6] + wave_part[7])};
      double total[2];
      const TicketArgs tk{dot.tickets, dot.partials, dot.part2};
      if (ticket_reduce_wave0<2>(tk, mine, dot.yy ? 2 : 1, (unsigned)bidx, (unsigned)n_tiles, total) && lane == 0) {
        *dot.out0 = total[0];
        if (dot.yy) *dot.out1 = total[1];
      }
    }
  }
}

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
template <int HLP>  // halo pairs per thread and plane: ceil(a / 256)
__global__ __launch_bounds__(kBlock) void cg_step_march_kernel(SellArgs A, MarchArgs M, Scal alpha_s, Scal beta_s,
                                                               const double *__restrict__ p_in, double *__restrict__ z_out,
                                                               DotArgs dot, const int *done, CgFuseArgs F, IpcSendArgs S) {
  if ((int)blockIdx.x < S.sp.n_blocks) {
    // a partitioned operator: the first blocks send the NEW direction's boundary rows (whatever the iteration gate
    // below says: every rank enqueues the same exchanges, and the receivers poll for them)
    ipc_halo_send_block(S.w, S.sp, p_in, (int)blockIdx.x, F.r, *F.cb);
    return;
  }
  if (*F.iteration < F.my_iteration) return;  // enqueued past convergence: that iteration never ran
  const int done_flag = done ? *done : 0;
  const CanonTileArgs &T = M.T;
  extern __shared__ __attribute__((aligned(16))) double tile_sh[];  // [3][a + kTileRun + a]
  __shared__ double dict_sh[32];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_march = (int)gridDim.x - S.sp.n_blocks, mb = (int)blockIdx.x - S.sp.n_blocks;
  const int bidx = T.reverse ? n_march - 1 - mb : mb;
  int zc, yt;
  if (T.per_xcd > 0) {
    const int xcd = bidx & (kNumXcd - 1), j = bidx >> 3;
    zc = j / T.per_xcd;
    yt = xcd * T.per_xcd + (j - zc * T.per_xcd);
  } else {
    zc = bidx / T.tiles_per_plane;
    yt = bidx - zc * T.tiles_per_plane;
  }
  const int a = T.a, b = T.b;
  const int p0 = yt * kTileRun;
  const int z_begin = zc * M.zc_planes, z_end = min(z_begin + M.zc_planes, T.plane_end);
  const int ldw = kTileRun + 2 * a;
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const double cg_a = *F.ca, cg_b = *F.cb;
  const uint32_t last_row = (uint32_t)(A.n_rows - 1);
  const char *pb = reinterpret_cast<const char *>(p_in), *rb = reinterpret_cast<const char *>(F.r);
  const char *pg_base = pb - (size_t)kVecGuard * 8, *rg_base = rb - (size_t)kVecGuard * 8;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const double dict_word = A.dict[lane & 31];
  if (lane < 32) dict_sh[lane] = dict_word;  // every wave stores the same words; the first barrier below covers them

  // what is in flight for ONE plane: the own rows' p, r, x and record, and this thread's share of the halo lines
  struct Flight {
    double2v p[2], r[2], x[2], hp[HLP], hr[HLP];
    u64x2 w[2];
    uint32_t rc[2];
    bool va[2], vb[2];
    int hat[HLP];
  };
  auto issue = [&](int zp, bool own, Flight &f) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int q = p0 + 256 * wave + 128 * g + 2 * lane;
      const int64_t row = (int64_t)zp * b + q;
      const bool in_plane = q < b && own;
      f.va[g] = in_plane && row <= (int64_t)last_row, f.vb[g] = in_plane && row + 1 <= (int64_t)last_row;
      int64_t gi = row + kVecGuard;  // guard-relative, clamped: a plane below the first / above the last reads zeros or x's last rows, weight 0
      gi = gi < 0 ? 0 : gi;
      gi = gi > (int64_t)T.max_gather ? (int64_t)T.max_gather : gi;
      f.rc[g] = (row >= 0 && row <= (int64_t)last_row) ? (uint32_t)row : (last_row & ~1u);
      f.p[g] = *reinterpret_cast<const double2v *>(pg_base + (size_t)((uint32_t)gi << 3));
      f.r[g] = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)gi << 3));
      if (own) {
        f.x[g] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(reinterpret_cast<const char *>(F.x) + (size_t)(f.rc[g] << 3)));
        f.w[g] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(A.pack + (size_t)(f.rc[g] << 3)));
      }
    }
#pragma unroll
    for (int i = 0; i < HLP; ++i) {
      const int u = (int)threadIdx.x + kBlock * i;  // pair u of the plane's a halo pairs
      const bool on = own && u < a;
      const int jj = 2 * u < a ? 2 * u - a : kTileRun + 2 * u - a;
      int64_t gi = (int64_t)zp * b + p0 + jj + kVecGuard;
      gi = gi < 0 ? 0 : gi;
      gi = gi > (int64_t)T.max_gather ? (int64_t)T.max_gather : gi;
      f.hat[i] = on ? a + jj : -1;
      f.hp[i] = f.hr[i] = double2v{0.0, 0.0};
      if (on) {
        f.hp[i] = *reinterpret_cast<const double2v *>(pg_base + (size_t)((uint32_t)gi << 3));
        f.hr[i] = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)gi << 3));
      }
    }
  };
  // the plane has arrived: p' of the own rows (-> out), x and p' stored, the LDS copy of the plane filled
  auto consume = [&](bool own, const Flight &f, double2v (&out)[2], double *buf) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      double2v pn;
      pn.x = __builtin_fma(cg_b, f.p[g].x, f.r[g].x), pn.y = __builtin_fma(cg_b, f.p[g].y, f.r[g].y);
      out[g] = pn;
      if (own) {
        double2v xn;
        xn.x = __builtin_fma(cg_a, f.p[g].x, f.x[g].x), xn.y = __builtin_fma(cg_a, f.p[g].y, f.x[g].y);
        double2v *xp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.x) + (size_t)(f.rc[g] << 3));
        double2v *pp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.p_out) + (size_t)(f.rc[g] << 3));
        if (f.vb[g]) __builtin_nontemporal_store(xn, xp_), __builtin_nontemporal_store(pn, pp_);
        else if (f.va[g]) F.x[f.rc[g]] = xn.x, F.p_out[f.rc[g]] = pn.x;
        *reinterpret_cast<double2v *>(&buf[a + 256 * wave + 128 * g + 2 * lane]) = pn;
      }
    }
    if (own) {
#pragma unroll
      for (int i = 0; i < HLP; ++i)
        if (f.hat[i] >= 0) {
          double2v hn;
          hn.x = __builtin_fma(cg_b, f.hp[i].x, f.hr[i].x), hn.y = __builtin_fma(cg_b, f.hp[i].y, f.hr[i].y);
          *reinterpret_cast<double2v *>(&buf[f.hat[i]]) = hn;
        }
    }
  };

  if (done_flag) {  // converged in that iteration: only x += alpha p is left to do
    for (int zp = z_begin; zp < z_end; ++zp) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int q = p0 + 256 * wave + 128 * g + 2 * lane;
        const int64_t row = (int64_t)zp * b + q;
        if (q < b && row <= (int64_t)last_row) {
          double2v *xp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.x) + (size_t)((uint32_t)row << 3));
          if (row + 1 <= (int64_t)last_row) {
            const double2v pv = *reinterpret_cast<const double2v *>(pb + (size_t)((uint32_t)row << 3));
            double2v xv = *xp_;
            xv.x = __builtin_fma(cg_a, pv.x, xv.x), xv.y = __builtin_fma(cg_a, pv.y, xv.y);
            *xp_ = xv;
          } else {
            F.x[row] = __builtin_fma(cg_a, p_in[row], F.x[row]);
          }
        }
      }
    }
    return;
  }

  double2v pm[2], pc[2], pn[2];  // p' of the plane behind / at / ahead IN MARCHING ORDER
  u64x2 wc[2];
  uint32_t rcc[2];
  bool vac[2], vbc[2];
  Flight fl;
  const bool down = M.alternate != 0 && (zc & 1) != 0;  // (block-uniform)
  const int nz = z_end - z_begin;
  auto plane = [&](int s) { return down ? z_end - 1 - s : z_begin + s; };  // s = -1 and s = nz: the planes next to the chunk
  auto lds_of = [&](int zp) { return tile_sh + ((zp % 3 + 3) % 3) * ldw; };
  issue(plane(-1), false, fl);
  consume(false, fl, pm, nullptr);
  issue(plane(0), true, fl);
  consume(true, fl, pc, lds_of(plane(0)));
#pragma unroll
  for (int g = 0; g < 2; ++g) wc[g] = fl.w[g], rcc[g] = fl.rc[g], vac[g] = fl.va[g], vbc[g] = fl.vb[g];
  issue(plane(1), 1 < nz, fl);
  double dot_a = 0.0, dot_b = 0.0;
  for (int s = 0; s < nz; ++s) {
    const int zp = plane(s);
    const bool next_own = s + 1 < nz;
    u64x2 wn[2];
    uint32_t rcn[2];
    bool van[2], vbn[2];
    consume(next_own, fl, pn, lds_of(plane(s + 1)));
#pragma unroll
    for (int g = 0; g < 2; ++g) wn[g] = fl.w[g], rcn[g] = fl.rc[g], van[g] = fl.va[g], vbn[g] = fl.vb[g];
    if (s + 2 <= nz) issue(plane(s + 2), s + 2 < nz, fl);  // (one plane ahead of the one consumed next)
    __syncthreads();  // the LDS copy of plane zp is complete; the buffer two planes back is free again
    const double *buf = lds_of(zp);
    const bool applies = zp >= M.apply_begin && zp < M.apply_end;  // (block-uniform)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (!applies) break;
      const int at = a + 256 * wave + 128 * g + 2 * lane;
      double2v xg[6];
      xg[0] = down ? pn[g] : pm[g], xg[5] = down ? pm[g] : pn[g];  // the planes below / above, whichever way the block marches
      xg[1] = *reinterpret_cast<const double2v *>(&buf[at - a]);
      xg[4] = *reinterpret_cast<const double2v *>(&buf[at + a]);
      double el = 0.0;
      if (lane == 0) el = buf[at - 1];
      if (lane == kWave - 1) el = buf[at + 2];
      const double left = dpp_shift<0x138>(pc[g].y);
      const double right = dpp_shift<0x130>(pc[g].x);
      xg[2].x = lane == 0 ? el : left;
      xg[2].y = pc[g].x;
      xg[3].x = pc[g].y;
      xg[3].y = lane == kWave - 1 ? el : right;
      double acc_a = 0.0, acc_b = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const unsigned ba = (unsigned)(wc[g].x >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(wc[g].y >> (8 * (k + 1))) & 0xffu;
        acc_a += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ba) * (xg[k].x - pc[g].x);
        acc_b += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + bb) * (xg[k].y - pc[g].y);
      }
      const double ext_a = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)wc[g].x & 0xffu));
      const double ext_b = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)wc[g].y & 0xffu));
      double2v yi;
      yi.x = __builtin_fma(alpha, __builtin_fma(ext_a, pc[g].x, acc_a), beta * pc[g].x);
      yi.y = __builtin_fma(alpha, __builtin_fma(ext_b, pc[g].y, acc_b), beta * pc[g].y);
      double2v *yp = reinterpret_cast<double2v *>(reinterpret_cast<char *>(z_out) + (size_t)(rcc[g] << 3));
      if (vbc[g]) { if (A.nt_y) __builtin_nontemporal_store(yi, yp); else *yp = yi; }
      else if (vac[g]) z_out[rcc[g]] = yi.x;
      yi.x = vac[g] ? yi.x : 0.0;
      yi.y = vbc[g] ? yi.y : 0.0;
      dot_a += pc[g].x * yi.x + pc[g].y * yi.y;
      dot_b += yi.x * yi.x + yi.y * yi.y;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
      pm[g] = pc[g], pc[g] = pn[g], wc[g] = wn[g], rcc[g] = rcn[g], vac[g] = van[g], vbc[g] = vbn[g];
  }
  dot_a = wave_sum_to_lane63(dot_a);
  if (dot.yy) dot_b = wave_sum_to_lane63(dot_b);
  if (dot.tickets == nullptr) {
    if (lane == kWave - 1) {
      const int slot = dot.block_offset + bidx * (kBlock / kWave) + wave;
      dot.partials[slot] = dot_a;
      if (dot.yy) dot.partials[dot.nblocks_total + slot] = dot_b;
    }
    return;
  }
  // the reduction finishes here (one rank, unsplit): block partial, then two levels of tickets -- no final-pass launch
  __shared__ double wave_part[2 * (kBlock / kWave)];
  if (lane == kWave - 1) wave_part[wave] = dot_a, wave_part[kBlock / kWave + wave] = dot.yy ? dot_b : 0.0;
  __syncthreads();
  if (wave != 0) return;
  const double mine[2] = {(wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]),
                          (wave_part[4] + wave_part[5]) + (wave_part[6] + wave_part[7])};
  double total[2];
  const TicketArgs tk{dot.tickets, dot.partials, dot.part2};
  if (ticket_reduce_wave0<2>(tk, mine, dot.yy ? 2 : 1, (unsigned)bidx, (unsigned)n_march, total) && lane == 0) {
    *dot.out0 = total[0];
    if (dot.yy) *dot.out1 = total[1];
  }
}

// CSR tail: one wavefront per overflowing row; the lanes' partial products are folded
// with __shfl_down and lane 0 adds the row's remainder to y.
__global__ __launch_bounds__(kBlock) void spmv_tail_kernel(int64_t n_tail, const int *__restrict__ tail_row,
                                                           const int64_t *__restrict__ tail_ptr,
                                                           const int *__restrict__ tail_col,
                                                           const double *__restrict__ tail_val,
                                                           Scal alpha_s, const double *__restrict__ x,
                                                           double *__restrict__ y, const int *done) {
  if (done && *done) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t t = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (t >= n_tail) return;
  const double alpha = ld_scal2(alpha_s);
  const int r = tail_row[t];
  const double xi = x[r];
  double acc = 0.0;
  for (int64_t k = tail_ptr[t] + lane; k < tail_ptr[t + 1]; k += kWave)
    acc += tail_val[k] * (x[tail_col[k]] - xi);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, kWave);
  if (lane == 0) y[r] += alpha * acc;
}

template <bool NT, bool DOT, int VARIANT>
static void launch_sell(const storm_hip_op *op, int nb, Scal alpha, Scal beta, const double *x, double *y,
                        const int *slice_list, int64_t n_launch, DotArgs dot, const int *done, hipEvent_t ev0,
                        hipEvent_t ev1, bool accumulate) {
  SellArgs A{op->d_pack, op->d_slice_off, op->n_rows, op->uniform_width, (int)op->ctx->opt_spmv_xcd_remap, op->d_dict, op->dict_size,
             op->d_offs, op->offs_size, (int)accumulate};
  constexpr int LV = (VARIANT == 2) ? 2 : 0;  // listed slices: no LDS window, but the record format stays
  hipStream_t st = op->ctx->stream;
  if (slice_list == nullptr && op->ctx->opt_spmv_xcd_remap != 0) {
    hipExtLaunchKernelGGL((spmv_sell_kernel<NT, DOT, VARIANT, true>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha,
                       beta, x, y, slice_list, n_launch, dot, done);
  } else if (slice_list == nullptr) {
    hipExtLaunchKernelGGL((spmv_sell_kernel<NT, DOT, VARIANT, false>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha,
                       beta, x, y, slice_list, n_launch, dot, done);
  } else if (op->ctx->opt_spmv_xcd_remap != 0) {
    // listed slices (interior / boundary sets of a partitioned operator): the LDS window does not
    // apply, the XCD grouping still does -- the interior list is consecutive but for a few gaps
    hipExtLaunchKernelGGL((spmv_sell_kernel<NT, DOT, LV, true>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha, beta,
                       x, y, slice_list, n_launch, dot, done);
  } else {
    hipExtLaunchKernelGGL((spmv_sell_kernel<NT, DOT, LV, false>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha, beta,
                       x, y, slice_list, n_launch, dot, done);
  }
}

// 128-row groups per wave of the format-4 / 5 kernel.
static inline int canon_groups(const storm_hip_op *op) { return op->ctx->opt_spmv_canon_groups == 2 ? 2 : 1; }

// The tiled format-4 kernel applies to an UNSPLIT, non-accumulating launch of an operator whose common offsets are
// (-b, -a, -1, +1, +a, +b) with a, b even, a <= 512, b >= 2 a, and enough planes to fill tiles.
static inline int canon_tile_planes(const storm_hip_op *op) {
  const int64_t tz = op->ctx->opt_spmv_canon_tile;
  return tz == 4 ? 4 : 2;
}
// interior = true: the launch over a partitioned (mixed) operator's interior groups, which must be whole planes
// [int_plane0, int_plane1) (op_upload_slice_lists checks that).
static bool canon_tile_geometry(const storm_hip_op *op, CanonTileArgs *T, int *n_blocks, bool interior = false) {
  if (op->ctx->opt_spmv_canon_tile == 0 || op->pair != 2 || op->canon_k != 6 || op->canon_m1 != 2) return false;
  if (interior != (op->d_bnd_pack != nullptr)) return false;  // (a mixed operator always runs as its two lists)
  if (interior && op->int_plane1 <= op->int_plane0) return false;
  const int *o = op->canon_off;
  const int a = o[4], b = o[5];
  if (o[0] != -b || o[1] != -a || o[2] != -1 || o[3] != 1) return false;
  if (a < 2 || a > 512 || (a & 1) || (b & 1) || b < 2 * a) return false;
  const int tz = canon_tile_planes(op);
  if ((int64_t)sizeof(double) * tz * (kTileRun + 2 * a) > 60 * 1024) return false;  // the LDS copy of a tile (64 KiB per block)
  const int64_t plane0 = interior ? op->int_plane0 : 0, plane1 = interior ? op->int_plane1 : (op->n_rows + b - 1) / b;
  const int64_t planes = plane1 - plane0;
  if (planes < 2 * tz || op->n_rows < op->ctx->opt_spmv_canon_tile_min_rows) return false;  // small operators: the plain kernel (or the latency path)
  T->a = a, T->b = b;
  T->a_magic = (unsigned)((((uint64_t)1 << 32) + (uint64_t)a - 1) / (uint64_t)a);
  T->tiles_per_plane = (b + kTileRun - 1) / kTileRun;
  T->per_xcd = (T->tiles_per_plane % kNumXcd == 0 && op->ctx->opt_spmv_xcd_remap != 0) ? T->tiles_per_plane / kNumXcd : 0;
  T->max_gather = (int)(op->n_rows + op->n_halo) + kVecGuard + 2;
  T->reverse = op->ctx->spmv_reverse;
  T->plane0 = (int)plane0, T->plane_end = (int)plane1;
  *n_blocks = (int)(((planes + tz - 1) / tz) * T->tiles_per_plane);
  return true;
}
// Slices per wave: SPW for the uniform-width value-dictionary kernel, 1 otherwise.
static inline int op_spw(const storm_hip_op *op) {
  if (op->pair) return 1;  // a "slice" of a format-3 operator is a 128-row group, one per wave
  return (op->dict_size > 0 && op->uniform_width > 0) ? (int)op->spw : 1;
}

template <bool DOT>
static void launch_pair(const storm_hip_op *op, int nb, Scal alpha, Scal beta, const double *x, double *y,
                        const int *slice_list, int64_t n_launch, DotArgs dot, const int *done, hipEvent_t ev0,
                        hipEvent_t ev1, bool accumulate, const IpcFused *fused, const CgFuseArgs *cg_fuse = nullptr) {
  // the interior list of a partitioned operator is consecutive but for a few gaps: the XCD grouping still pays there
  const int group = (slice_list == nullptr || slice_list == op->d_interior) ? (int)op->ctx->opt_spmv_xcd_remap : 0;
  SellArgs A{op->d_pack, op->d_slice_off, op->n_rows, op->uniform_width, group, op->d_dict, op->dict_size,
             op->d_offs, op->offs_size, (int)accumulate};
  A.nt_y = (int)(op->ctx->opt_spmv_nt_y != 0);
  hipStream_t st = op->ctx->stream;
  int width = op->uniform_width;
  const bool boundary_of_mixed = op->d_bnd_pack != nullptr && slice_list != nullptr && slice_list == op->d_boundary;
  if (boundary_of_mixed) {  // the groups that read halo columns: format-3 records of their own, in list order
    A.pack = op->d_bnd_pack, A.rec_by_pos = 1;
    width = op->bnd_width;
  }
  CanonTileArgs T;
  int tile_blocks = 0;
  const bool interior_list = slice_list != nullptr && slice_list == op->d_interior;
  IpcSendArgs S{};
  if (fused != nullptr && interior_list) S.w = fused->w, S.sp = fused->sp;  // the interior launch sends
  if (op->pair == 2 && !boundary_of_mixed && (slice_list == nullptr || interior_list) && !accumulate &&
      canon_tile_geometry(op, &T, &tile_blocks, interior_list) && tile_blocks + S.sp.n_blocks == nb) {
    const int tz = canon_tile_planes(op);
    const int hl_need = (tz * T.a + kBlock - 1) / kBlock;
    const size_t lds = sizeof(double) * (size_t)tz * (size_t)(kTileRun + 2 * T.a) + (size_t)op->ctx->opt_spmv_tile_lds_pad;
    const bool wload = DOT && dot.w != nullptr && dot.w != x;
#define TILE_GO3(WL_, TZ_, HL_)                                                                                              \
  hipExtLaunchKernelGGL((spmv_canon_tile_kernel<DOT, WL_, TZ_, HL_>), dim3(nb), dim3(kBlock), lds, st, ev0, ev1, 0, A, T, alpha, \
                        beta, x, y, dot, done, S, CgFuseArgs{})
#define TILE_GO2(TZ_, HL_)                                                                                                   \
  do {                                                                                                                       \
    if (cg_fuse != nullptr) {                                                                                                \
      if constexpr (DOT)                                                                                                     \
        hipExtLaunchKernelGGL((spmv_canon_tile_kernel<true, false, TZ_, HL_, true>), dim3(nb), dim3(kBlock), lds, st, ev0, ev1, \
                              0, A, T, alpha, beta, x, y, dot, done, S, *cg_fuse);                                            \
    } else if (wload) TILE_GO3(true, TZ_, HL_);                                                                              \
    else TILE_GO3(false, TZ_, HL_);                                                                                          \
  } while (0)
#define TILE_GO(TZ_)                       \
  do {                                     \
    if (hl_need <= 1) TILE_GO2(TZ_, 1);     \
    else if (hl_need <= 2) TILE_GO2(TZ_, 2); \
    else if (hl_need <= 4) TILE_GO2(TZ_, 4); \
    else TILE_GO2(TZ_, 8);                  \
  } while (0)
    if (tz == 2) TILE_GO(2);
    else TILE_GO(4);
#undef TILE_GO
#undef TILE_GO2
#undef TILE_GO3
    return;
  }
  if (op->pair >= 2 && !boundary_of_mixed) {  // formats 4, 5: the common offsets travel as kernel arguments
    CanonArgs C;
    for (int k = 0; k < 7; ++k) C.off[k] = op->canon_off[k];
    C.max_gather = (int)(op->n_rows + op->n_halo) + kVecGuard + 2;
    C.reverse = op->ctx->spmv_reverse;
    C.xcd_shift = -1, C.xcd_full = 0;
    if (group > 1 && (group & (group - 1)) == 0) {
      while ((1 << (C.xcd_shift + 1)) <= group) ++C.xcd_shift;
      const int span = kNumXcd * group;
      C.xcd_full = (nb / span) * span;
    }
    A.types = op->d_types;
#define CANON_GO2(K_, M1_, T_, G_)                                                                                       \
  hipExtLaunchKernelGGL((spmv_canon_kernel<DOT, K_, M1_, T_, G_>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, C, alpha, \
                        beta, x, y, slice_list, n_launch, dot, done)
#define CANON_GO(K_, M1_)                                      \
  do {                                                         \
    const bool two = canon_groups(op) == 2;                    \
    if (op->pair == 3) {                                       \
      if (two) CANON_GO2(K_, M1_, true, 2);                    \
      else CANON_GO2(K_, M1_, true, 1);                        \
    } else {                                                   \
      if (two) CANON_GO2(K_, M1_, false, 2);                   \
      else CANON_GO2(K_, M1_, false, 1);                       \
    }                                                          \
  } while (0)
    if (op->canon_k == 6) CANON_GO(6, 2);
    else if (op->canon_k == 4) CANON_GO(4, 1);
    else CANON_GO(2, 0);
#undef CANON_GO
#undef CANON_GO2
    return;
  }
  IpcRecvArgs H{};
  const bool halo_reads = fused != nullptr && slice_list != nullptr && slice_list == op->d_boundary;
  if (halo_reads) H.w = fused->w, H.rp = fused->rp, H.n_halo = (int)op->n_halo;
#define PAIR_GO(W_)                                                                                                         \
  do {                                                                                                                      \
    if (halo_reads)                                                                                                         \
      hipExtLaunchKernelGGL((spmv_pair_kernel<DOT, W_, true>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha, beta, \
                            x, y, slice_list, n_launch, dot, done, H);                                                      \
    else                                                                                                                    \
      hipExtLaunchKernelGGL((spmv_pair_kernel<DOT, W_, false>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha, beta, \
                            x, y, slice_list, n_launch, dot, done, H);                                                      \
  } while (0)
  switch (width) {
    case 1: PAIR_GO(1); break;
    case 2: PAIR_GO(2); break;
    case 3: PAIR_GO(3); break;
    case 4: PAIR_GO(4); break;
    case 5: PAIR_GO(5); break;
    case 6: PAIR_GO(6); break;
    default: PAIR_GO(7); break;
  }
#undef PAIR_GO
}
static inline int blocks_for(const storm_hip_op *op, int64_t n_launch_slices, bool boundary_of_mixed = false) {
  const int64_t per_block = (kBlock / kWave) * ((op->pair >= 2 && !boundary_of_mixed) ? canon_groups(op) : op_spw(op));
  return (int)((n_launch_slices + per_block - 1) / per_block);
}

template <bool DOT, int SPW>
static void launch_dict(const storm_hip_op *op, int nb, Scal alpha, Scal beta, const double *x, double *y,
                        const int *slice_list, int64_t n_launch, DotArgs dot, const int *done, hipEvent_t ev0,
                        hipEvent_t ev1, bool accumulate) {
  // slice lists (interior / boundary sets) are not contiguous: no XCD grouping there
  const int group = slice_list ? 0 : (int)op->ctx->opt_spmv_xcd_remap;
  SellArgs A{op->d_pack, op->d_slice_off, op->n_rows, op->uniform_width, group, op->d_dict, op->dict_size,
             op->d_offs, op->offs_size, (int)accumulate};
  hipStream_t st = op->ctx->stream;
#define DICT_GO(W_)                                                                                                 \
  do {                                                                                                              \
    if (op->offs_size > 0)                                                                                          \
      hipExtLaunchKernelGGL((spmv_dict_kernel<DOT, W_, SPW, 2>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A,     \
                            alpha, beta, x, y, slice_list, n_launch, dot, done);                                    \
    else                                                                                                            \
      hipExtLaunchKernelGGL((spmv_dict_kernel<DOT, W_, SPW, 1>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A,     \
                            alpha, beta, x, y, slice_list, n_launch, dot, done);                                    \
  } while (0)
  switch (op->uniform_width) {
    case 1: DICT_GO(1); break;
    case 2: DICT_GO(2); break;
    case 3: DICT_GO(3); break;
    case 4: DICT_GO(4); break;
    case 5: DICT_GO(5); break;
    case 6: DICT_GO(6); break;
    default: DICT_GO(7); break;
  }
#undef DICT_GO
}

// fused (peer-window transport): the interior launch carries the send, the boundary launch reads the window.
static int interior_blocks(const storm_hip_op *op, bool accumulate, int n_send_blocks) {
  CanonTileArgs T;
  int nbt = 0;
  if (!accumulate && canon_tile_geometry(op, &T, &nbt, true)) return nbt + n_send_blocks;
  return -1;
}
static int launch_range(const storm_hip_op *op, Scal alpha, Scal beta, const double *x, double *y,
                        const int *slice_list, int64_t n_launch, DotArgs dot, bool want_dot,
                        const int *done, bool accumulate, const IpcFused *fused = nullptr, const CgFuseArgs *cg_fuse = nullptr) {
  if (n_launch <= 0) return STORM_HIP_OK;
  storm_hip_ctx *c = op->ctx;
  const bool prof = c->opt_profile_spmv != 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (prof) {
    while (c->prof_events.size() < c->prof_used + 2) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreate(&ev));
      c->prof_events.push_back(ev);
    }
    // the events are attached to the kernel dispatch itself (hipExtLaunchKernelGGL): they are
    // stamped at the kernel's begin and end, the quantity rocprofv3's kernel trace reports
    ev0 = c->prof_events[c->prof_used], ev1 = c->prof_events[c->prof_used + 1];
  }
  int nb = blocks_for(op, n_launch, op->d_bnd_pack != nullptr && slice_list != nullptr && slice_list == op->d_boundary);
  if (slice_list == nullptr && !accumulate && n_launch == op->n_slices) {
    CanonTileArgs T;
    int nbt = 0;
    if (canon_tile_geometry(op, &T, &nbt)) nb = nbt;  // the tiled format-4 kernel (launch_pair takes it on the same test)
  } else if (slice_list != nullptr && slice_list == op->d_interior) {
    const int nbi = interior_blocks(op, accumulate, fused ? fused->sp.n_blocks : 0);
    if (nbi >= 0) nb = nbi;  // ... over the interior planes of a partitioned operator, plus the sending blocks
    else if (fused != nullptr) STORM_TRY(comm_ipc_send(op, x, fused->w, fused->sp));  // a kernel that cannot send: stand-alone
  }
  const bool nt = c->opt_nt != 0;
  if (op->pair) {
    if (want_dot) launch_pair<true>(op, nb, alpha, beta, x, y, slice_list, n_launch, dot, done, ev0, ev1, accumulate, fused, cg_fuse);
    else launch_pair<false>(op, nb, alpha, beta, x, y, slice_list, n_launch, dot, done, ev0, ev1, accumulate, fused);
    HIP_TRY(hipGetLastError());
    if (prof) c->prof_used += 2;
    return STORM_HIP_OK;
  }
  if (op_spw(op) >= 1 && op->dict_size > 0 && op->uniform_width > 0) {
    switch (op_spw(op)) {
      case 1: if (want_dot) launch_dict<true, 1>(op, nb, alpha, beta, x, y, slice_list, n_launch, dot, done, ev0, ev1, accumulate);
              else launch_dict<false, 1>(op, nb, alpha, beta, x, y, slice_list, n_launch, dot, done, ev0, ev1, accumulate); break;
      case 2: if (want_dot) launch_dict<true, 2>(op, nb, alpha, beta, x, y, slice_list, n_launch, dot, done, ev0, ev1, accumulate);
              else launch_dict<false, 2>(op, nb, alpha, beta, x, y, slice_list, n_launch, dot, done, ev0, ev1, accumulate); break;
      default: if (want_dot) launch_dict<true, 4>(op, nb, alpha, beta, x, y, slice_list, n_launch, dot, done, ev0, ev1, accumulate);
               else launch_dict<false, 4>(op, nb, alpha, beta, x, y, slice_list, n_launch, dot, done, ev0, ev1, accumulate); break;
    }
    HIP_TRY(hipGetLastError());
    if (prof) c->prof_used += 2;
    return STORM_HIP_OK;
  }
#define SPMV_GO(NT_, DOT_, VAR_) \
  launch_sell<NT_, DOT_, VAR_>(op, nb, alpha, beta, x, y, slice_list, n_launch, dot, done, ev0, ev1, accumulate)
#define SPMV_VAR(VAR_)                                                                      \
  do {                                                                                      \
    if (nt) { if (want_dot) SPMV_GO(true, true, VAR_); else SPMV_GO(true, false, VAR_); }   \
    else    { if (want_dot) SPMV_GO(false, true, VAR_); else SPMV_GO(false, false, VAR_); } \
  } while (0)
  if (op->dict_size > 0) SPMV_VAR(2);
  else if (c->opt_spmv_variant == 1) SPMV_VAR(1);
  else SPMV_VAR(0);
#undef SPMV_VAR
#undef SPMV_GO
  HIP_TRY(hipGetLastError());
  if (prof) c->prof_used += 2;
  return STORM_HIP_OK;
}

// The z-marching form of the fused CG step: blocks of 1024 rows x opt_cg_march planes.
// partitioned: a mixed operator (interior planes on format 4, boundary groups on format 3): the march covers ALL owned
// planes for x and p', applies the operator to the interior ones.
static bool cg_march_geometry(const storm_hip_op *op, MarchArgs *M, int *n_blocks, bool partitioned = false) {
  const int64_t zc = op->ctx->opt_cg_march;
  int nbt = 0;
  if (zc < 2 || !canon_tile_geometry(op, &M->T, &nbt, partitioned)) return false;
  if ((int64_t)sizeof(double) * 3 * (kTileRun + 2 * M->T.a) > 60 * 1024) return false;
  if (partitioned && op->n_rows % M->T.b != 0) return false;  // (whole planes only)
  const int64_t planes = (op->n_rows + M->T.b - 1) / M->T.b;
  // (option cg_march is the chunk of a large lattice; a smaller one marches fewer planes per block, so that the grid
  //  still holds ~2 blocks per resident slot: 192^3 with 8-plane chunks is 864 blocks for 1 024 slots -- 124 us per CG
  //  iteration against 111 with 4-plane chunks)
  //  (option cg_march_fill: the block count aimed at; 0 = cg_march whatever the size)
  const int64_t want = op->ctx->opt_cg_march_fill;
  const int64_t fill = want > 0 ? planes * M->T.tiles_per_plane / want : zc;
  M->zc_planes = (int)std::min<int64_t>(std::min<int64_t>(zc, std::max<int64_t>(2, fill)), planes);
  M->alternate = (int)(op->ctx->opt_cg_march_alternate != 0);
  M->apply_begin = partitioned ? (int)op->int_plane0 : 0;
  M->apply_end = partitioned ? (int)op->int_plane1 : (int)planes;
  M->T.plane0 = 0, M->T.plane_end = (int)planes;
  const int64_t chunks = (planes + M->zc_planes - 1) / M->zc_planes;
  *n_blocks = (int)(chunks * M->T.tiles_per_plane);
  return true;
}

// The fused CG step (CgFuseArgs) applies to an operator whose unsplit apply runs the tiled format-4 kernel.
bool spmv_can_fuse_cg(const storm_hip_op *op) {
  CanonTileArgs T;
  int nb = 0;
  if (op->halo.n_nbrs == 0 && op->d_bnd_pack == nullptr && op->tail_rows == 0 && canon_tile_geometry(op, &T, &nb)) return true;
  // ... or a partitioned (mixed) operator on the peer-window transport, whose boundary launch reads the window itself
  MarchArgs M;
  // ... or on RCCL: the boundary planes of p' are packed by a small kernel and travel on the comm stream under the march
  return op->halo.n_nbrs > 0 && op->d_bnd_pack != nullptr && op->tail_rows == 0 &&
         ((comm_is_ipc(op->ctx) && op->ctx->opt_ipc_fused != 0) || (comm_is_rccl(op->ctx) && op->ctx->opt_rccl_fused != 0)) &&
         op->n_boundary > 0 && cg_march_geometry(op, &M, &nb, true);
}

int spmv_grid_blocks(const storm_hip_op *op) {
  CanonTileArgs T;
  int nb = 0;
  if (canon_tile_geometry(op, &T, &nb)) return nb;  // (an unsplit launch of a format-4 lattice operator)
  return blocks_for(op, op->n_slices);
}

int spmv_launch(const storm_hip_op *op, Scal alpha, Scal beta, const double *x, double *y,
                const SpmvDot *sd, const int *done, bool accumulate) {
  storm_hip_ctx *c = op->ctx;
  const bool fuse_dot = sd != nullptr && op->tail_rows == 0;
  STORM_REQUIRE(op->halo.n_nbrs == 0 || c->comm != nullptr,
                "operator has a halo plan but the context has no communicator (call storm_hip_ctx_comm_init)");
  // (a mixed operator -- format 4 inside, format 3 where rows read halo columns -- always runs as its two lists)
  const bool exchange = op->halo.n_nbrs > 0;
  const bool split = exchange || op->d_bnd_pack != nullptr;
  DotArgs dot{nullptr, nullptr, 0, 0, 0};
  // peer-window transport + paired records: the fused form of the exchange (IpcFused)
  const bool fuse_x = exchange && comm_is_ipc(c) && op->pair != 0 && c->opt_ipc_fused != 0 && op->n_boundary > 0;
  IpcFused fx;
  if (fuse_x) STORM_TRY(comm_ipc_exchange(op, &fx.w, &fx.sp, &fx.rp));
  const int nbi_tile = split ? interior_blocks(op, accumulate, 0) : -1;  // (send blocks carry no partials)
  const int nb_int = split ? (nbi_tile >= 0 ? nbi_tile : blocks_for(op, op->n_interior)) : spmv_grid_blocks(op);
  const int nb_bnd = split ? blocks_for(op, op->n_boundary, op->d_bnd_pack != nullptr) : 0;
  const int nb_total = nb_int + nb_bnd;
  if (fuse_dot) {
    STORM_REQUIRE(8 * (int64_t)nb_total <= c->partials_capacity,
                  "spmv: %d blocks exceed the partials workspace", nb_total);
    dot = DotArgs{sd->w, sd->partials, sd->yy ? 1 : 0, 4 * nb_total, 0};  // one partial per wave
  }
  if (sd && sd->nblocks_out) *sd->nblocks_out = fuse_dot ? 4 * nb_total : 0;
  if (sd && sd->ticketed_out) *sd->ticketed_out = 0;
  if (fuse_dot && !split && op->pair >= 2 && sd->out[0] != nullptr && c->opt_ticket_reduce != 0 && c->comm == nullptr &&
      nb_total <= kTicketGroup * kTicketMaxGroups && 2 * nb_total <= c->partials_capacity &&
      (!sd->yy || sd->out[1] != nullptr)) {
    dot.tickets = c->d_tickets, dot.part2 = c->d_ticket_sums, dot.out0 = sd->out[0], dot.out1 = sd->out[1];
    dot.nblocks_total = nb_total;
    if (sd->ticketed_out) *sd->ticketed_out = 1;
  }

  if (!split) {
    CgFuseArgs cgf{};
    const bool cg_fused = sd != nullptr && sd->cg.x != nullptr;
    if (cg_fused) {
      STORM_REQUIRE(spmv_can_fuse_cg(op) && fuse_dot && !accumulate && sd->w == x,
                    "spmv: the fused CG step needs the tiled format-4 kernel");
      dot.tickets = nullptr, dot.nblocks_total = 4 * nb_total;  // (the tiled form of the step leaves per-wave partials)
      if (sd->ticketed_out) *sd->ticketed_out = 0;
      cgf = CgFuseArgs{sd->cg.iteration, sd->cg.my_iteration, sd->cg.ca, sd->cg.cb, sd->cg.x, sd->cg.r, sd->cg.p_out};
    }
    MarchArgs M;
    int nb_march = 0;
    if (cg_fused && cg_march_geometry(op, &M, &nb_march)) {
      // the z-marching step kernel: its own grid, its own (fewer) partial slots -- or, where the caller takes the sum
      // from the slab (sd->out[0]), the reduction finished in the kernel by tickets
      dot.nblocks_total = 4 * nb_march;
      if (sd->nblocks_out) *sd->nblocks_out = 4 * nb_march;
      if (sd->out[0] != nullptr && c->opt_ticket_reduce != 0 && c->comm == nullptr && !sd->yy &&
          nb_march <= kTicketGroup * kTicketMaxGroups && 2 * (int64_t)nb_march <= c->partials_capacity) {
        dot.tickets = c->d_tickets, dot.part2 = c->d_ticket_sums, dot.out0 = sd->out[0], dot.out1 = nullptr;
        dot.nblocks_total = nb_march;
        if (sd->ticketed_out) *sd->ticketed_out = 1;
      }
      hipEvent_t ev0 = nullptr, ev1 = nullptr;
      if (c->opt_profile_spmv != 0) {
        while (c->prof_events.size() < c->prof_used + 2) {
          hipEvent_t ev;
          HIP_TRY(hipEventCreate(&ev));
          c->prof_events.push_back(ev);
        }
        ev0 = c->prof_events[c->prof_used], ev1 = c->prof_events[c->prof_used + 1];
        c->prof_used += 2;
      }
      SellArgs A{op->d_pack, op->d_slice_off, op->n_rows, op->uniform_width, 0, op->d_dict, op->dict_size, op->d_offs, op->offs_size, 0};
      A.nt_y = (int)(c->opt_spmv_nt_y != 0);
      const size_t lds = sizeof(double) * 3 * (size_t)(kTileRun + 2 * M.T.a);
      if (M.T.a <= kBlock)
        hipExtLaunchKernelGGL((cg_step_march_kernel<1>), dim3(nb_march), dim3(kBlock), lds, c->stream, ev0, ev1, 0, A, M, alpha, beta,
                              x, y, dot, done, cgf, IpcSendArgs{});
      else
        hipExtLaunchKernelGGL((cg_step_march_kernel<2>), dim3(nb_march), dim3(kBlock), lds, c->stream, ev0, ev1, 0, A, M, alpha, beta,
                              x, y, dot, done, cgf, IpcSendArgs{});
      HIP_TRY(hipGetLastError());
      return STORM_HIP_OK;
    }
    STORM_TRY(launch_range(op, alpha, beta, x, y, nullptr, op->n_slices, dot, fuse_dot, done, accumulate, nullptr,
                           cg_fused ? &cgf : nullptr));
  } else {
    const bool cg_fused_part = sd != nullptr && sd->cg.x != nullptr;
    if (cg_fused_part) {
      // The fused CG step on a partitioned operator (peer-window transport): ONE marching launch updates x and forms
      // p' on every owned plane, applies the operator to the interior planes and -- its first blocks -- sends p' of the
      // boundary rows; the boundary launch then reads p' (owned columns) and the window (halo columns).
      MarchArgs M;
      int nb_march = 0;
      const bool over_rccl = !fuse_x && comm_is_rccl(c);
      STORM_REQUIRE((fuse_x || over_rccl) && fuse_dot && !accumulate && sd->w == x && cg_march_geometry(op, &M, &nb_march, true),
                    "spmv: the fused CG step on a partitioned operator needs the peer-window or the RCCL transport and a lattice");
      // RCCL: p' = r + beta p of the rows to send is formed by a small kernel on the comm stream and travels while the march
      // below runs; the boundary launch waits for the planes (they land in p_out's halo tail)
      if (over_rccl) STORM_TRY(comm_halo_exchange_begin_direction(op, x, sd->cg.r, sd->cg.cb, sd->cg.p_out));
      const CgFuseArgs cgf{sd->cg.iteration, sd->cg.my_iteration, sd->cg.ca, sd->cg.cb, sd->cg.x, sd->cg.r, sd->cg.p_out};
      const int nb_b = blocks_for(op, op->n_boundary, true);
      STORM_REQUIRE(8 * (int64_t)(nb_march + nb_b) <= c->partials_capacity, "spmv: %d blocks exceed the partials workspace", nb_march + nb_b);
      dot = DotArgs{sd->cg.p_out, sd->partials, sd->yy ? 1 : 0, 4 * (nb_march + nb_b), 0};
      if (sd->nblocks_out) *sd->nblocks_out = 4 * (nb_march + nb_b);
      hipEvent_t ev0 = nullptr, ev1 = nullptr;
      if (c->opt_profile_spmv != 0) {
        while (c->prof_events.size() < c->prof_used + 2) {
          hipEvent_t ev;
          HIP_TRY(hipEventCreate(&ev));
          c->prof_events.push_back(ev);
        }
        ev0 = c->prof_events[c->prof_used], ev1 = c->prof_events[c->prof_used + 1];
        c->prof_used += 2;
      }
      SellArgs A{op->d_pack, op->d_slice_off, op->n_rows, op->uniform_width, 0, op->d_dict, op->dict_size, op->d_offs, op->offs_size, 0};
      A.nt_y = (int)(c->opt_spmv_nt_y != 0);
      const size_t lds = sizeof(double) * 3 * (size_t)(kTileRun + 2 * M.T.a);
      IpcSendArgs S{};
      if (fuse_x) S = IpcSendArgs{fx.w, fx.sp};  // (RCCL: no sending blocks in the march)
      // (the march kernel's own dots use w = p' from its registers; DotArgs::w only has to be non-null there)
      if (M.T.a <= kBlock)
        hipExtLaunchKernelGGL((cg_step_march_kernel<1>), dim3(nb_march + S.sp.n_blocks), dim3(kBlock), lds, c->stream, ev0, ev1, 0, A, M,
                              alpha, beta, x, y, dot, done, cgf, S);
      else
        hipExtLaunchKernelGGL((cg_step_march_kernel<2>), dim3(nb_march + S.sp.n_blocks), dim3(kBlock), lds, c->stream, ev0, ev1, 0, A, M,
                              alpha, beta, x, y, dot, done, cgf, S);
      HIP_TRY(hipGetLastError());
      dot.block_offset = 4 * nb_march;
      // the boundary groups: z = A p' from p_out and the window (RCCL: p_out's halo tail); <p', z> partials behind the march's
      if (over_rccl) STORM_TRY(comm_halo_exchange_end(op));
      STORM_TRY(launch_range(op, alpha, beta, sd->cg.p_out, y, op->d_boundary, op->n_boundary, dot, fuse_dot, done, accumulate,
                             fuse_x ? &fx : nullptr));
      return STORM_HIP_OK;
    }
    // interior rows overlap the halo exchange running on the comm stream
    if (exchange && !fuse_x) STORM_TRY(comm_halo_exchange_begin(op, const_cast<double *>(x)));
    if (fuse_x && op->n_interior == 0) STORM_TRY(comm_ipc_send(op, x, fx.w, fx.sp));
    STORM_TRY(launch_range(op, alpha, beta, x, y, op->d_interior, op->n_interior, dot, fuse_dot, done, accumulate,
                           fuse_x ? &fx : nullptr));
    if (exchange && !fuse_x) STORM_TRY(comm_halo_exchange_end(op));
    dot.block_offset = 4 * nb_int;
    STORM_TRY(launch_range(op, alpha, beta, x, y, op->d_boundary, op->n_boundary, dot, fuse_dot, done, accumulate,
                           fuse_x ? &fx : nullptr));
  }
  if (op->tail_rows > 0) {
    const int nb = (int)((op->tail_rows + 3) / 4);
    hipLaunchKernelGGL(spmv_tail_kernel, dim3(nb), dim3(kBlock), 0, c->stream, op->tail_rows,
                       op->d_tail_row, op->d_tail_ptr, op->d_tail_col, op->d_tail_val, alpha, x, y, done);
    HIP_TRY(hipGetLastError());
  }
  return STORM_HIP_OK;
}

// ---- diagonal of beta*I + alpha*M (for a Jacobi preconditioner) -----------------------------------
// In the difference form  (Mx)_i = sum_k w_ik (x_col - x_i) + ext_i x_i  the coefficient of x_i is
// ext_i - sum_k w_ik (no slot has col == i: build_op receives off-diagonal entries only, and padding
// slots carry w = 0).  One lane per row, same slot addressing as build_op; not a hot kernel.
__global__ __launch_bounds__(kBlock) void diag_sell_kernel(const char *__restrict__ pack,
                                                           const int64_t *__restrict__ slice_off, int64_t n_rows,
                                                           const double *__restrict__ dict, int fmt2, double alpha,
                                                           double beta, double *__restrict__ d,
                                                           const unsigned long long *__restrict__ types) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t s = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t r = s * kWave + lane;
  if (r >= n_rows) return;
  if (fmt2 >= 3) {  // paired rows: 128-row groups, weights of row r in word (r % 128), bytes pre-scaled by 8
    const uint64_t iw = fmt2 == 5 ? types[reinterpret_cast<const unsigned char *>(pack)[r] >> 3]
                                  : reinterpret_cast<const uint64_t *>(pack + (r >> 7) * (fmt2 == 4 ? kCanonRecBytes : kPairRecBytes))[r & 127];
    double sum = 0.0;
    for (int k = 0; k < 7; ++k) sum += dict[((unsigned)(iw >> (8 * (k + 1))) & 0xffu) >> 3];
    d[r] = beta + alpha * (dict[((unsigned)iw & 0xffu) >> 3] - sum);
    return;
  }
  const char *rec = pack + slice_off[s];
  if (dict) {  // value-dictionary record (format 2: 16-byte words, weights in bytes 1..7 of the first half)
    const int w = fmt2 ? 7 : (int)((slice_off[s + 1] - slice_off[s] - kExtBytes) / kColSlotBytes);
    const uint64_t iw = reinterpret_cast<const uint64_t *>(rec)[fmt2 ? 2 * lane : lane];
    double sum = 0.0;
    for (int k = 0; k < w; ++k) sum += dict[(unsigned)(iw >> (8 * (k + 1))) & 0xffu];
    d[r] = beta + alpha * (dict[(unsigned)iw & 0xffu] - sum);
    return;
  }
  const int w = (int)((slice_off[s + 1] - slice_off[s] - kExtBytes) / kSlotBytes);
  const double *val = reinterpret_cast<const double *>(rec + kExtBytes + (int64_t)w * (kWave * 4));
  const int np2 = w >> 1;
  double sum = 0.0;
  for (int k = 0; k < w; ++k) {
    const int at = (k < 2 * np2) ? ((k >> 1) * kWave + lane) * 2 + (k & 1) : np2 * 2 * kWave + lane;
    sum += val[at];
  }
  d[r] = beta + alpha * (reinterpret_cast<const double *>(rec)[lane] - sum);
}

__global__ void diag_tail_kernel(int64_t n_tail, const int *__restrict__ tail_row,
                                 const int64_t *__restrict__ tail_ptr, const double *__restrict__ tail_val,
                                 double alpha, double *__restrict__ d) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tail) return;
  double sum = 0.0;
  for (int64_t k = tail_ptr[t]; k < tail_ptr[t + 1]; ++k) sum += tail_val[k];
  d[tail_row[t]] -= alpha * sum;  // each overflowing row appears once in the tail
}

__global__ void safe_invert_kernel(int64_t n, double *__restrict__ d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = d[i] == 0.0 ? 0.0 : 1.0 / d[i];  // safe_inverse, Crow/MathUtils.hpp:54-58
}

// ---- host-side build ----------------------------------------------------------------------------

}  // namespace storm

namespace storm {

// Host threads for the operator build (record packing is ~10 passes over the rows / entries of the operator).
static int build_threads() {
  static const int n = [] {
    if (const char *e = getenv("STORM_HIP_BUILD_THREADS")) return std::max(1, atoi(e));
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hw ? hw : 1u, 16u));  // (a one-GPU share of a host is about 16 cores)
  }();
  return n;
}
// fn(t, begin, end) over [0, n) in contiguous chunks, chunk t on thread t (in index order: results that depend on
// "first occurrence" are merged in chunk order and come out as a serial pass would leave them).
template <class F>
static int parallel_chunks(int64_t n, int64_t min_chunk, F &&fn) {
  static const int64_t forced_chunk = getenv("STORM_HIP_BUILD_MIN_CHUNK") ? atoll(getenv("STORM_HIP_BUILD_MIN_CHUNK")) : 0;  // (tests: thread small inputs too)
  if (forced_chunk > 0) min_chunk = forced_chunk;
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(build_threads(), n / std::max<int64_t>(1, min_chunk)));
  const int64_t per = (n + T - 1) / T;
  if (T == 1) {
    fn(0, (int64_t)0, n);
    return 1;
  }
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back([&, t] { fn(t, std::min(n, t * per), std::min(n, (t + 1) * per)); });
  fn(0, (int64_t)0, std::min(n, per));
  for (auto &x : th) x.join();
  return T;
}
struct BuildTimer {  // STORM_HIP_BUILD_TIMING=1: stage times of the operator build on stderr
  bool on = getenv("STORM_HIP_BUILD_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void lap(const char *what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[storm_hip build] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

template <class T>
static int upload(T **dst, const std::vector<T> &src, int64_t *bytes) {
  const size_t nbytes = sizeof(T) * (src.size() ? src.size() : 1);
  hipError_t e = hipMalloc((void **)dst, nbytes);
  if (e != hipSuccess) STORM_FAIL(STORM_HIP_E_ALLOC, "hipMalloc(%zu) failed: %s", nbytes, hipGetErrorString(e));
  if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
  *bytes += (int64_t)nbytes;
  return STORM_HIP_OK;
}

// The distinct fp64 bit patterns of an operator, while there are at most 256 of them.
struct ValueDict {
  std::vector<uint64_t> values;               // index -> bit pattern
  std::vector<std::pair<uint64_t, int>> tab;  // open-addressing hash, 1024 buckets
  uint64_t last_bits = ~0ull;
  int last_idx = -1;
  ValueDict() : tab(1024, {0, -1}) {}
  static uint64_t bits(double v) {
    uint64_t b;
    memcpy(&b, &v, 8);
    return b;
  }
  int find(uint64_t b, bool insert) {
    if (b == last_bits) return last_idx;
    size_t h = (size_t)((b * 0x9E3779B97F4A7C15ull) >> 54);
    for (;; h = (h + 1) & 1023) {
      if (tab[h].second < 0) {
        if (!insert || values.size() >= (size_t)kDictSize) return -1;
        tab[h] = {b, (int)values.size()};
        values.push_back(b);
      }
      if (tab[h].first == b && tab[h].second >= 0) {
        last_bits = b, last_idx = tab[h].second;
        return last_idx;
      }
    }
  }
  bool add(double v) { return find(bits(v), true) >= 0; }
  int index(double v) { return find(bits(v), false); }
  // the same look-up without the one-entry cache: safe from several threads once the dictionary is complete
  int lookup(uint64_t b) const {
    size_t h = (size_t)((b * 0x9E3779B97F4A7C15ull) >> 54);
    for (;; h = (h + 1) & 1023) {
      if (tab[h].second < 0) return -1;
      if (tab[h].first == b) return tab[h].second;
    }
  }
  int lookup(double v) const { return lookup(bits(v)); }
  // Distinct values of keys(i), i in [0, n), in order of first occurrence (what a serial pass of add() would give),
  // gathered by the build threads; false when there are more than the dictionary holds.
  template <class K>
  bool add_all(int64_t n, K &&key) {
    std::vector<ValueDict> part((size_t)build_threads());
    std::vector<char> ok(part.size(), 1);
    const int T = parallel_chunks(n, 1 << 16, [&](int t, int64_t b, int64_t e) {
      ValueDict &d = part[(size_t)t];
      for (int64_t i = b; i < e; ++i)
        if (d.find(key(i), true) < 0) {
          ok[(size_t)t] = 0;
          return;
        }
    });
    for (int t = 0; t < T; ++t) {
      if (!ok[(size_t)t]) return false;
      for (uint64_t v : part[(size_t)t].values)
        if (find(v, true) < 0) return false;
    }
    return true;
  }
};

// Shortest common supersequence of two short offset lists (format 3: the merged neighbour list of a row pair).
// Returns its length (<= na + nb), the sequence in out[], and where each input element landed in pa[] / pb[].
static int merge_offsets(const int64_t *a, int na, const int64_t *b, int nb, int64_t *out, int *pa, int *pb) {
  int L[9][9];  // LCS of the suffixes a[i..], b[j..]
  for (int i = na; i >= 0; --i)
    for (int j = nb; j >= 0; --j)
      L[i][j] = (i == na || j == nb) ? 0 : (a[i] == b[j] ? 1 + L[i + 1][j + 1] : std::max(L[i + 1][j], L[i][j + 1]));
  int i = 0, j = 0, m = 0;
  while (i < na || j < nb) {
    if (i < na && j < nb && a[i] == b[j]) pa[i] = pb[j] = m, out[m++] = a[i], ++i, ++j;
    else if (j == nb || (i < na && L[i + 1][j] >= L[i][j + 1])) pa[i] = m, out[m++] = a[i], ++i;
    else pb[j] = m, out[m++] = b[j], ++j;
  }
  return m;
}

// Build from off-diagonal CSR rows (entries already in the order they must be summed).
static int build_op(storm_hip_ctx *c, int64_t n, int64_t n_halo, const std::vector<int64_t> &row_ptr,
                    const std::vector<int> &col, const std::vector<double> &val,
                    const std::vector<double> &ext, storm_hip_op **out) {
  HIP_TRY(hipSetDevice(c->device));
  auto *op = new storm_hip_op();
  op->ctx = c;
  op->n_rows = n;
  op->n_halo = n_halo;
  op->nnz = row_ptr[n];
  const int64_t n_slices = (n + kWave - 1) / kWave;
  op->n_slices = n_slices;
  BuildTimer timer;
  int64_t max_len = 0;
  {
    std::vector<int64_t> ml((size_t)build_threads(), 0);
    parallel_chunks(n, 1 << 16, [&](int t, int64_t b, int64_t e) {
      int64_t m = 0;
      for (int64_t i = b; i < e; ++i) m = std::max(m, row_ptr[i + 1] - row_ptr[i]);
      ml[(size_t)t] = m;
    });
    for (int64_t m : ml) max_len = std::max(max_len, m);
  }
  op->max_row_len = max_len;
  {
    const int st_lat = op_make_latency_copy(op, n, n_halo, row_ptr, col, val, ext);
    if (st_lat != STORM_HIP_OK) {
      storm_hip_op_destroy(op);
      return st_lat;
    }
  }
  int64_t cap = c->opt_ell_cap;
  if (cap <= 0) {
    const double mean = n > 0 ? (double)op->nnz / (double)n : 0.0;
    cap = std::max<int64_t>(8, (int64_t)std::ceil(2.0 * mean));
  }
  // Value dictionary (see the header comment): lossless, so taken whenever the operator qualifies.
  ValueDict vd;
  bool cv = c->opt_spmv_dict != 0 && std::min(max_len, cap) <= 7;
  timer.lap("latency copy, max row");
  if (cv) {
    cv = vd.add(0.0);  // padding slots
    cv = cv && vd.add_all(n, [&](int64_t i) { return ValueDict::bits(ext[(size_t)i]); });
    cv = cv && vd.add_all((int64_t)val.size(), [&](int64_t k) { return ValueDict::bits(val[(size_t)k]); });
  }
  timer.lap("value dictionary");
  // ... and the column offsets of the ELL part (format 2)
  const int64_t w_op = std::min(max_len, cap);
  ValueDict od;
  bool co = cv && c->opt_spmv_dict >= 2 && w_op > 0 && n + n_halo < (int64_t)INT32_MAX;
  if (co) {
    co = od.find(0, true) >= 0;  // padding slots point at their own row
    std::vector<ValueDict> part((size_t)build_threads());
    std::vector<char> ok(part.size(), 1);
    const int T = parallel_chunks(n, 1 << 14, [&](int t, int64_t rb, int64_t re) {
      ValueDict &d = part[(size_t)t];
      for (int64_t r = rb; r < re; ++r) {
        const int64_t e = std::min(row_ptr[r + 1], row_ptr[r] + w_op);
        for (int64_t k = row_ptr[r]; k < e; ++k)
          if (d.find((uint64_t)((int64_t)col[(size_t)k] - r), true) < 0) {
            ok[(size_t)t] = 0;
            return;
          }
      }
    });
    for (int t = 0; co && t < T; ++t) {
      co = ok[(size_t)t] != 0;
      for (size_t q = 0; co && q < part[(size_t)t].values.size(); ++q) co = od.find(part[(size_t)t].values[q], true) >= 0;
    }
  }
  timer.lap("offset dictionary");
  // ... and whether consecutive rows can share their gathers (format 3, see the header comment)
  bool pr = co && c->opt_spmv_dict >= 3 && max_len <= std::min<int64_t>(7, cap) && vd.values.size() <= 32 && od.values.size() <= 64 &&
            n + n_halo < ((int64_t)1 << 28);
  const int64_t n_groups = (n + 2 * kWave - 1) / (2 * kWave);
  std::vector<char> pair_pack;
  int pair_width = 0;
  if (pr) {
    pair_pack.assign((size_t)n_groups * kPairRecBytes, 0);
    const uint64_t zero_v = (uint64_t)vd.index(0.0) << 3, zero_o = (uint64_t)od.find(0, false) << 2;
    const int64_t n_total = n + n_halo;
    std::atomic<int> pr_ok{1};
    std::vector<int> widths((size_t)build_threads(), 0);
    parallel_chunks(n_groups * kWave, 1 << 13, [&](int t_, int64_t p_begin, int64_t p_end) {
    int pair_width = 0;  // (this thread's; folded below)
    for (int64_t p = p_begin; p < p_end && pr_ok.load(std::memory_order_relaxed); ++p) {
      const int64_t ra = 2 * p, rb = 2 * p + 1;
      int64_t oa[8], ob[8], merged[16];
      int pa[8], pb[8], na = 0, nb2 = 0;
      if (ra < n) for (int64_t k = row_ptr[ra]; k < row_ptr[ra + 1]; ++k) oa[na++] = (int64_t)col[(size_t)k] - ra;
      if (rb < n) for (int64_t k = row_ptr[rb]; k < row_ptr[rb + 1]; ++k) ob[nb2++] = (int64_t)col[(size_t)k] - rb;
      const int m = merge_offsets(oa, na, ob, nb2, merged, pa, pb);
      if (m > 7) { pr_ok = 0; break; }
      pair_width = std::max(pair_width, m);
      for (int k = 0; k < m; ++k)  // every 16-byte gather must stay inside [guard, padding]
        if (ra + merged[k] < -(int64_t)kVecGuard || rb + merged[k] > n_total + 3) pr_ok = 0;
      uint64_t wa = ra < n ? ((uint64_t)vd.lookup(ext[(size_t)ra]) << 3) : zero_v;
      uint64_t wb = rb < n ? ((uint64_t)vd.lookup(ext[(size_t)rb]) << 3) : zero_v;
      uint64_t jw = 0;
      for (int k = 0; k < 7; ++k) {
        wa |= zero_v << (8 * (k + 1)), wb |= zero_v << (8 * (k + 1));
        jw |= (k < m ? ((uint64_t)od.lookup((uint64_t)merged[k]) << 2) : zero_o) << (8 * k);
      }
      for (int k = 0; k < na; ++k) {
        wa &= ~(0xffull << (8 * (pa[k] + 1)));
        wa |= ((uint64_t)vd.lookup(val[(size_t)(row_ptr[ra] + k)]) << 3) << (8 * (pa[k] + 1));
      }
      for (int k = 0; k < nb2; ++k) {
        wb &= ~(0xffull << (8 * (pb[k] + 1)));
        wb |= ((uint64_t)vd.lookup(val[(size_t)(row_ptr[rb] + k)]) << 3) << (8 * (pb[k] + 1));
      }
      char *rec = pair_pack.data() + (p / kWave) * kPairRecBytes;
      const int l = (int)(p % kWave);
      reinterpret_cast<uint64_t *>(rec)[2 * l] = wa;
      reinterpret_cast<uint64_t *>(rec)[2 * l + 1] = wb;
      reinterpret_cast<uint64_t *>(rec + 2 * kWave * 8)[l] = jw;
    }
    widths[(size_t)t_] = pair_width;
    });
    for (int w_ : widths) pair_width = std::max(pair_width, w_);
    pr = pr_ok.load() != 0 && pair_width > 0;
  }
  timer.lap("paired records");
  // groups with a row that reads a halo column (they run behind the halo exchange)
  std::vector<char> grp_bnd;
  int64_t n_bnd_groups = 0;
  if (pr) {
    grp_bnd.assign((size_t)n_groups, 0);
    if (n_halo > 0) {
      std::vector<int64_t> cnt((size_t)build_threads(), 0);
      parallel_chunks(n_groups, 1 << 10, [&](int t, int64_t gb, int64_t ge) {
        for (int64_t g = gb; g < ge; ++g) {
          const int64_t r1 = std::min<int64_t>(n, (g + 1) * 2 * kWave);
          for (int64_t k = row_ptr[g * 2 * kWave]; k < row_ptr[r1] && !grp_bnd[(size_t)g]; ++k) grp_bnd[(size_t)g] = col[(size_t)k] >= n;
          cnt[(size_t)t] += grp_bnd[(size_t)g];
        }
      });
      for (int64_t v : cnt) n_bnd_groups += v;
    }
  }
  // ... and whether all rows list their neighbours in one common order of offsets (format 4, see spmv_canon_kernel).
  // A partitioned operator is MIXED: the common order is asked of the groups that read no halo column (a rank's
  // slab of a structured box but for its outer planes), the others keep their format-3 records.
  int64_t canon[16];
  int canon_len = 0, canon_m1 = -1;
  bool cn = pr && c->opt_spmv_dict >= 4 && 2 * n_bnd_groups <= n_groups && (n_bnd_groups == 0 || c->opt_spmv_mixed != 0);
  std::vector<char> bnd_pack;
  std::vector<unsigned long long> row_types;  // format 5
  if (cn) {
    // the distinct offsets and who precedes whom in some row; a common order = a linear extension of that relation
    int64_t dist[8];
    int nd = 0;
    bool before[8][8] = {};
    struct Local {
      int64_t dist[8];
      int nd = 0;
      bool before[8][8] = {};
      bool ok = true;
    };
    std::vector<Local> loc((size_t)build_threads());
    const int Tc = parallel_chunks(n, 1 << 14, [&](int t, int64_t rb_, int64_t re_) {
      Local &L = loc[(size_t)t];
      for (int64_t r = rb_; L.ok && r < re_; ++r) {
        if (grp_bnd[(size_t)(r / (2 * kWave))]) continue;
        int idx[8], no = 0;
        for (int64_t k = row_ptr[r]; L.ok && k < row_ptr[r + 1]; ++k) {
          const int64_t o = (int64_t)col[(size_t)k] - r;
          int q = 0;
          while (q < L.nd && L.dist[q] != o) ++q;
          if (q == L.nd) {
            if (L.nd == 7) { L.ok = false; break; }
            L.dist[L.nd++] = o;
          }
          idx[no++] = q;
        }
        for (int i = 0; L.ok && i < no; ++i)
          for (int j = i + 1; j < no; ++j) {
            if (idx[i] == idx[j]) L.ok = false;  // the same offset twice in one row
            L.before[idx[i]][idx[j]] = true;
          }
      }
    });
    for (int t = 0; cn && t < Tc; ++t) {  // union of the threads' offsets and of their "precedes" relations
      const Local &L = loc[(size_t)t];
      cn = L.ok;
      int map_[8];
      for (int q = 0; cn && q < L.nd; ++q) {
        int g = 0;
        while (g < nd && dist[g] != L.dist[q]) ++g;
        if (g == nd) {
          if (nd == 7) { cn = false; break; }
          dist[nd++] = L.dist[q];
        }
        map_[q] = g;
      }
      for (int i = 0; cn && i < L.nd; ++i)
        for (int j = 0; j < L.nd; ++j)
          if (L.before[i][j]) before[map_[i]][map_[j]] = true;
    }
    bool placed[8] = {};
    while (cn && canon_len < nd) {  // Kahn's algorithm; ties go to the smaller offset
      int pick = -1;
      for (int q = 0; q < nd; ++q) {
        if (placed[q]) continue;
        bool free_ = true;
        for (int q2 = 0; q2 < nd; ++q2) free_ = free_ && !(before[q2][q] && !placed[q2]);
        if (free_ && (pick < 0 || dist[q] < dist[pick])) pick = q;
      }
      if (pick < 0) { cn = false; break; }  // a cycle: rows disagree about the order
      placed[pick] = true;
      canon[canon_len++] = dist[pick];
    }
    for (int q = 0; cn && q + 1 < canon_len; ++q)
      if (canon[q] == -1 && canon[q + 1] == 1) canon_m1 = q;
    cn = cn && ((canon_len == 6 && canon_m1 == 2) || (canon_len == 4 && canon_m1 == 1) || (canon_len == 2 && canon_m1 == 0));
    for (int q = 0; cn && q < canon_len; ++q) cn = canon[q] > -(int64_t)INT32_MAX / 2 && canon[q] < (int64_t)INT32_MAX / 2;
  }
  if (cn) {
    for (int64_t g = 0; g < n_groups; ++g)  // the format-3 records of the boundary groups, in list order
      if (grp_bnd[(size_t)g])
        bnd_pack.insert(bnd_pack.end(), pair_pack.begin() + (size_t)g * kPairRecBytes, pair_pack.begin() + (size_t)(g + 1) * kPairRecBytes);
    pair_pack.assign((size_t)n_groups * kCanonRecBytes, 0);
    const uint64_t zero_v = (uint64_t)vd.index(0.0) << 3;
    parallel_chunks(n_groups * kWave, 1 << 13, [&](int, int64_t p_begin, int64_t p_end) {
    for (int64_t p = p_begin; p < p_end; ++p) {
      uint64_t w2[2];
      for (int half = 0; half < 2; ++half) {
        const int64_t r = 2 * p + half;
        uint64_t w = r < n ? ((uint64_t)vd.lookup(ext[(size_t)r]) << 3) : zero_v;
        for (int k = 0; k < 7; ++k) w |= zero_v << (8 * (k + 1));
        if (r < n) {
          int q = 0;
          const bool by_entry = grp_bnd[(size_t)(r / (2 * kWave))] != 0;  // never applied from here: the weights
          for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {         // only serve diag_sell_kernel
            if (by_entry) q = (int)(k - row_ptr[r]);
            else while (canon[q] != (int64_t)col[(size_t)k] - r) ++q;  // a subsequence of the common order
            w &= ~(0xffull << (8 * (q + 1)));
            w |= ((uint64_t)vd.lookup(val[(size_t)k]) << 3) << (8 * (q + 1));
          }
        }
        w2[half] = w;
      }
      uint64_t *rec = reinterpret_cast<uint64_t *>(pair_pack.data() + (p / kWave) * kCanonRecBytes);
      rec[2 * (p % kWave)] = w2[0], rec[2 * (p % kWave) + 1] = w2[1];
    }
    });
    timer.lap("canonical order + records");
    op->canon_k = canon_len, op->canon_m1 = canon_m1;
    for (int k = 0; k < 7; ++k) op->canon_off[k] = k < canon_len ? (int)canon[k] : 0;
    // ... and whether the rows' weight words take few distinct values (format 5): one byte per row
    if (c->opt_spmv_dict >= 5) {
      const uint64_t *words = reinterpret_cast<const uint64_t *>(pair_pack.data());
      const int64_t n_words = n_groups * 2 * kWave;
      std::vector<unsigned char> typed((size_t)n_words);
      bool ty = true;
      uint64_t last = ~0ull;
      int last_idx = -1;
      for (int64_t r = 0; ty && r < n_words; ++r) {
        const uint64_t w = words[r];
        int idx = (w == last) ? last_idx : -1;
        for (int t = 0; idx < 0 && t < (int)row_types.size(); ++t) idx = row_types[(size_t)t] == w ? t : -1;
        if (idx < 0) {
          if ((int)row_types.size() == kMaxRowTypes) ty = false;
          else idx = (int)row_types.size(), row_types.push_back(w);
        }
        last = w, last_idx = idx;
        typed[(size_t)r] = (unsigned char)(idx << 3);
      }
      if (ty) {
        pair_pack.assign(reinterpret_cast<const char *>(typed.data()), reinterpret_cast<const char *>(typed.data()) + typed.size());
        row_types.resize(kMaxRowTypes, 0ull);
      } else {
        row_types.clear();
      }
    }
  }
  if (pr) {
    // format 3 (or 4) it is: a "slice" of this operator is a 128-row group
    op->pair = cn ? (row_types.empty() ? 2 : 3) : 1;
    op->bnd_width = pair_width;
    if (cn) pair_width = canon_len;
    op->n_slices = n_groups;
    op->uniform_width = pair_width;
    op->ell_slots = n_groups * 2 * kWave * pair_width;
    std::vector<int64_t> goff((size_t)n_groups + 1);
    for (int64_t s = 0; s <= n_groups; ++s)
      goff[(size_t)s] = s * (cn ? (row_types.empty() ? kCanonRecBytes : kTypedRecBytes) : kPairRecBytes);
    for (int64_t s = 0; s < n_groups; ++s) (grp_bnd[(size_t)s] ? op->h_boundary : op->h_interior).push_back((int)s);
    op->n_interior_slices = (int64_t)op->h_interior.size();
    int st3 = STORM_HIP_OK;
    int64_t bytes3 = 0;
    std::vector<double> vtab((size_t)kDictSize, 0.0);
    for (size_t k = 0; k < vd.values.size(); ++k) memcpy(&vtab[k], &vd.values[k], 8);
    std::vector<int> otab((size_t)kDictSize, 0);
    for (size_t k = 0; k < od.values.size(); ++k) otab[k] = (int)(int64_t)od.values[k];
    op->dict_size = (int)vd.values.size();
    op->offs_size = (int)od.values.size();
    op->pack_bytes = (int64_t)pair_pack.size() + (int64_t)bnd_pack.size();
    op->spw = 1;
    std::vector<int> no_i;
    std::vector<int64_t> one_zero(1, 0);
    std::vector<double> no_d;
    if ((st3 = upload(&op->d_dict, vtab, &bytes3)) || (st3 = upload(&op->d_offs, otab, &bytes3)) ||
        (st3 = upload(&op->d_slice_off, goff, &bytes3)) || (st3 = upload(&op->d_pack, pair_pack, &bytes3)) ||
        (st3 = upload(&op->d_tail_row, no_i, &bytes3)) || (st3 = upload(&op->d_tail_ptr, one_zero, &bytes3)) ||
        (st3 = upload(&op->d_tail_col, no_i, &bytes3)) || (st3 = upload(&op->d_tail_val, no_d, &bytes3))) {
      storm_hip_op_destroy(op);
      return st3;
    }
    if (!row_types.empty() && (st3 = upload(&op->d_types, row_types, &bytes3))) {
      storm_hip_op_destroy(op);
      return st3;
    }
    if (!bnd_pack.empty() && ((st3 = upload(&op->d_bnd_pack, bnd_pack, &bytes3)) || (st3 = op_upload_slice_lists(op)))) {
      storm_hip_op_destroy(op);
      return st3;
    }
    timer.lap("upload");
    op->device_bytes += bytes3;
    const int64_t need3 = 8 * ((n_slices + 3) / 4) + 16 + 2 * kMaxMulti;
    if (need3 > c->partials_capacity) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      double *bigger = nullptr;
      HIP_TRY(hipMalloc(&bigger, sizeof(double) * (size_t)need3));
      (void)hipFree(c->d_partials);
      c->d_partials = bigger;
      c->partials_capacity = need3;
    }
    *out = op;
    return STORM_HIP_OK;
  }
  const int64_t slot_bytes = cv ? kColSlotBytes : kSlotBytes;
  std::vector<int64_t> slice_off(n_slices + 1, 0);  // bytes
  std::vector<int> width(n_slices, 0);
  bool uniform = true;
  for (int64_t s = 0; s < n_slices; ++s) {
    int64_t w = 0;
    const int64_t r1 = std::min<int64_t>(n, (s + 1) * kWave);
    for (int64_t r = s * kWave; r < r1; ++r) w = std::max(w, row_ptr[r + 1] - row_ptr[r]);
    w = std::min(w, cap);
    if (co) w = w_op;  // 16-byte words: every slice is padded to the operator's width
    width[s] = (int)w;
    slice_off[s + 1] = slice_off[s] + (co ? (int64_t)kWave * 16 : kExtBytes + w * slot_bytes);
    if (s > 0 && width[s] != width[0]) uniform = false;
    op->ell_slots += w * kWave;
  }
  op->uniform_width = (uniform && n_slices > 0 && width[0] > 0) ? width[0] : 0;
  std::vector<char> pack((size_t)slice_off[n_slices], 0);
  std::vector<int> tail_row, tail_col;
  std::vector<int64_t> tail_ptr(1, 0);
  std::vector<double> tail_val;
  for (int64_t s = 0; s < n_slices; ++s) {
    char *rec = pack.data() + slice_off[s];
    double *e_ = reinterpret_cast<double *>(rec);
    uint64_t *i_ = reinterpret_cast<uint64_t *>(rec);  // cv: the index words take the place of ext
    int *c_ = reinterpret_cast<int *>(rec + kExtBytes);
    double *v_ = reinterpret_cast<double *>(rec + kExtBytes + (int64_t)width[s] * (kWave * 4));
    bool touches_halo = false;
    for (int l = 0; l < kWave; ++l) {
      const int64_t r = s * kWave + l;
      const int64_t pad_col = r < n ? r : (n > 0 ? n - 1 : 0);
      const int64_t b = r < n ? row_ptr[r] : 0, e = r < n ? row_ptr[r + 1] : 0;
      uint64_t iw = 0, jw = 0;
      if (cv) iw = (uint64_t)vd.index(r < n ? ext[(size_t)r] : 0.0);
      else e_[l] = r < n ? ext[(size_t)r] : 0.0;
      if (co) {
        for (int k = 0; k < width[s]; ++k) {
          const bool real = b + k < e;
          iw |= (uint64_t)vd.index(real ? val[(size_t)(b + k)] : 0.0) << (8 * (k + 1));
          jw |= (uint64_t)od.find(real ? (uint64_t)((int64_t)col[(size_t)(b + k)] - r) : 0, false) << (8 * k);
          touches_halo |= real && col[(size_t)(b + k)] >= n;
        }
        i_[2 * l] = iw, i_[2 * l + 1] = jw;
      }
      const int np2 = width[s] >> 1;
      for (int k = 0; k < (co ? 0 : width[s]); ++k) {
        // slots are stored in pairs: lane l reads (slot 2p, slot 2p+1) as one 8-byte column pair and
        // one 16-byte weight pair; an odd last slot is stored column-major behind the pairs
        const int at = (k < 2 * np2) ? ((k >> 1) * kWave + l) * 2 + (k & 1) : np2 * 2 * kWave + l;
        const bool real = b + k < e;
        c_[at] = real ? col[(size_t)(b + k)] : (int)pad_col;
        const double w_k = real ? val[(size_t)(b + k)] : 0.0;
        if (cv) iw |= (uint64_t)vd.index(w_k) << (8 * (k + 1));
        else v_[at] = w_k;
        touches_halo |= real && c_[at] >= n;
      }
      if (cv && !co) i_[l] = iw;
      if (e - b > width[s]) {
        tail_row.push_back((int)r);
        for (int64_t k = b + width[s]; k < e; ++k) {
          tail_col.push_back(col[(size_t)k]);
          tail_val.push_back(val[(size_t)k]);
          touches_halo |= col[(size_t)k] >= n;
        }
        tail_ptr.push_back((int64_t)tail_col.size());
      }
    }
    (touches_halo ? op->h_boundary : op->h_interior).push_back((int)s);
  }
  op->tail_rows = (int64_t)tail_row.size();
  op->tail_nnz = (int64_t)tail_col.size();
  op->n_interior_slices = (int64_t)op->h_interior.size();

  int st = STORM_HIP_OK;
  int64_t bytes = 0;
  if (cv) {
    std::vector<double> table((size_t)kDictSize, 0.0);
    for (size_t k = 0; k < vd.values.size(); ++k) memcpy(&table[k], &vd.values[k], 8);
    op->dict_size = (int)vd.values.size();
    if ((st = upload(&op->d_dict, table, &bytes))) {
      storm_hip_op_destroy(op);
      return st;
    }
  }
  if (co) {
    std::vector<int> table((size_t)kDictSize, 0);
    for (size_t k = 0; k < od.values.size(); ++k) table[k] = (int)(int64_t)od.values[k];
    op->offs_size = (int)od.values.size();
    if ((st = upload(&op->d_offs, table, &bytes))) {
      storm_hip_op_destroy(op);
      return st;
    }
  }
  op->pack_bytes = (int64_t)pack.size();
  op->spw = (c->opt_spmv_spw == 1 || c->opt_spmv_spw == 2 || c->opt_spmv_spw == 4) ? c->opt_spmv_spw : 2;
  if ((st = upload(&op->d_slice_off, slice_off, &bytes)) || (st = upload(&op->d_pack, pack, &bytes)) ||
      (st = upload(&op->d_tail_row, tail_row, &bytes)) || (st = upload(&op->d_tail_ptr, tail_ptr, &bytes)) ||
      (st = upload(&op->d_tail_col, tail_col, &bytes)) || (st = upload(&op->d_tail_val, tail_val, &bytes))) {
    storm_hip_op_destroy(op);
    return st;
  }
  op->device_bytes = bytes;
  // fused-dot partials: two per SpMV block
  const int64_t need = 8 * ((n_slices + 3) / 4) + 16 + 2 * kMaxMulti;
  if (need > c->partials_capacity) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    double *bigger = nullptr;
    HIP_TRY(hipMalloc(&bigger, sizeof(double) * (size_t)need));
    (void)hipFree(c->d_partials);
    c->d_partials = bigger;
    c->partials_capacity = need;
  }
  *out = op;
  return STORM_HIP_OK;
}

// Called from op_set_halo (comm.hip): upload the interior / boundary slice lists.
int op_upload_slice_lists(storm_hip_op *op) {
  if (op->d_interior || op->d_boundary) return STORM_HIP_OK;
  int64_t bytes = 0;
  STORM_TRY(upload(&op->d_interior, op->h_interior, &bytes));
  STORM_TRY(upload(&op->d_boundary, op->h_boundary, &bytes));
  op->n_interior = (int64_t)op->h_interior.size();
  op->n_boundary = (int64_t)op->h_boundary.size();
  // a mixed operator whose interior groups are whole planes of its lattice (a slab of a box but for its outer planes)
  // runs them on the tiled kernel: planes [int_plane0, int_plane1)
  op->int_plane0 = op->int_plane1 = 0;
  if (op->pair == 2 && op->canon_k == 6 && op->n_interior > 0) {
    const int64_t b = op->canon_off[5], g0 = op->h_interior.front(), g1 = (int64_t)op->h_interior.back() + 1;
    const int64_t r0 = g0 * 2 * kWave, r1 = std::min<int64_t>(op->n_rows, g1 * 2 * kWave);
    if (b > 0 && g1 - g0 == op->n_interior && r0 % b == 0 && (r1 % b == 0 || r1 == op->n_rows))
      op->int_plane0 = r0 / b, op->int_plane1 = (r1 + b - 1) / b;
  }
  op->device_bytes += bytes;
  return STORM_HIP_OK;
}

}  // namespace storm

using namespace storm;

namespace storm {

// Rows of the operator from its faces, entries in FACE ORDER (== the order in which the reference's face loop
// accumulates into u[i]): entry (a -> b) of face f carries weight(f, false), entry (b -> a) weight(f, true).
// Threaded over CHUNKS OF FACES: a chunk counts its entries per row (one byte per row and chunk), a prefix over the
// chunks turns the counts into each chunk's first position inside every row, and the chunks then fill their entries
// -- two passes over the faces whatever the thread count, and the order inside a row does not depend on it.
// (A row that takes > 255 entries from one chunk: every thread scans all faces for its own range of rows instead.)
template <class W>
static void rows_from_faces(int64_t n_owned, int64_t n_faces, const int64_t *inner, const int64_t *outer, W &&weight,
                            std::vector<int64_t> &row_ptr, std::vector<int> &col, std::vector<double> &val) {
  row_ptr.assign((size_t)n_owned + 1, 0);
  const int64_t face_chunk = getenv("STORM_HIP_BUILD_MIN_CHUNK") ? std::max<int64_t>(1, atoll(getenv("STORM_HIP_BUILD_MIN_CHUNK"))) : (1 << 16);
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(build_threads(), n_faces / face_chunk));
  const int64_t per = (n_faces + T - 1) / T;
  std::vector<std::vector<unsigned char>> cnt((size_t)T);
  std::atomic<int> overflow{0};
  parallel_chunks(T, 1, [&](int, int64_t tb, int64_t te) {
    for (int64_t t = tb; t < te; ++t) {
      std::vector<unsigned char> &c_ = cnt[(size_t)t];
      c_.assign((size_t)n_owned, 0);
      for (int64_t f = t * per; f < std::min(n_faces, (t + 1) * per); ++f) {
        const int64_t a = inner[f], b = outer[f];
        if (a < n_owned && ++c_[(size_t)a] == 0) overflow = 1;
        if (b < n_owned && ++c_[(size_t)b] == 0) overflow = 1;
      }
    }
  });
  if (overflow.load()) {
    parallel_chunks(n_owned, 1 << 15, [&](int, int64_t r0, int64_t r1) {
      for (int64_t f = 0; f < n_faces; ++f) {
        const int64_t a = inner[f], b = outer[f];
        if (a >= r0 && a < r1) row_ptr[(size_t)a + 1]++;
        if (b >= r0 && b < r1) row_ptr[(size_t)b + 1]++;
      }
    });
    for (int64_t i = 0; i < n_owned; ++i) row_ptr[(size_t)i + 1] += row_ptr[(size_t)i];
    col.resize((size_t)row_ptr[(size_t)n_owned]), val.resize(col.size());
    std::vector<int64_t> fill(row_ptr.begin(), row_ptr.end() - 1);
    parallel_chunks(n_owned, 1 << 15, [&](int, int64_t r0, int64_t r1) {
      for (int64_t f = 0; f < n_faces; ++f) {
        const int64_t a = inner[f], b = outer[f];
        if (a >= r0 && a < r1) {
          const size_t at = (size_t)fill[(size_t)a]++;
          col[at] = (int)b, val[at] = weight(f, false);
        }
        if (b >= r0 && b < r1) {
          const size_t at = (size_t)fill[(size_t)b]++;
          col[at] = (int)a, val[at] = weight(f, true);
        }
      }
    });
    return;
  }
  // counts -> each chunk's offset inside the row (in place), row lengths -> row_ptr
  parallel_chunks(n_owned, 1 << 16, [&](int, int64_t r0, int64_t r1) {
    for (int64_t r = r0; r < r1; ++r) {
      int64_t run = 0;
      for (int t = 0; t < T; ++t) {
        const int64_t here = cnt[(size_t)t][(size_t)r];
        cnt[(size_t)t][(size_t)r] = (unsigned char)run;  // (a row of > 255 entries in all: the serial prefix below still holds
        run += here;                                     //  the truth; positions are taken modulo 256 only when run < 256)
      }
      row_ptr[(size_t)r + 1] = run;
    }
  });
  bool long_rows = false;
  for (int64_t i = 0; i < n_owned; ++i) {
    long_rows |= row_ptr[(size_t)i + 1] > 255;
    row_ptr[(size_t)i + 1] += row_ptr[(size_t)i];
  }
  col.resize((size_t)row_ptr[(size_t)n_owned]), val.resize(col.size());
  if (long_rows) {  // (offsets no longer fit a byte: one thread, plain fill)
    std::vector<int64_t> fill(row_ptr.begin(), row_ptr.end() - 1);
    for (int64_t f = 0; f < n_faces; ++f) {
      const int64_t a = inner[f], b = outer[f];
      if (a < n_owned) {
        const size_t at = (size_t)fill[(size_t)a]++;
        col[at] = (int)b, val[at] = weight(f, false);
      }
      if (b < n_owned) {
        const size_t at = (size_t)fill[(size_t)b]++;
        col[at] = (int)a, val[at] = weight(f, true);
      }
    }
    return;
  }
  parallel_chunks(T, 1, [&](int, int64_t tb, int64_t te) {
    for (int64_t t = tb; t < te; ++t) {
      std::vector<unsigned char> &o_ = cnt[(size_t)t];
      for (int64_t f = t * per; f < std::min(n_faces, (t + 1) * per); ++f) {
        const int64_t a = inner[f], b = outer[f];
        if (a < n_owned) {
          const size_t at = (size_t)(row_ptr[(size_t)a] + o_[(size_t)a]++);
          col[at] = (int)b, val[at] = weight(f, false);
        }
        if (b < n_owned) {
          const size_t at = (size_t)(row_ptr[(size_t)b] + o_[(size_t)b]++);
          col[at] = (int)a, val[at] = weight(f, true);
        }
      }
    }
  });
}

// inner / outer of every face inside [0, nt) and distinct; returns the first offending face or -1
static int64_t first_bad_face(int64_t n_faces, const int64_t *inner, const int64_t *outer, int64_t nt) {
  std::atomic<int64_t> bad{-1};
  parallel_chunks(n_faces, 1 << 16, [&](int, int64_t fb, int64_t fe) {
    for (int64_t f = fb; f < fe; ++f) {
      const int64_t a = inner[f], b = outer[f];
      if (!(a >= 0 && a < nt && b >= 0 && b < nt) || a == b) {
        int64_t cur = bad.load();
        while ((cur < 0 || f < cur) && !bad.compare_exchange_weak(cur, f)) {
        }
        return;
      }
    }
  });
  return bad.load();
}

}  // namespace storm

extern "C" {

int storm_hip_op_create_from_face_weights(storm_hip_ctx *c, int64_t n_owned, int64_t n_halo,
                                          int64_t n_faces, const int64_t *inner, const int64_t *outer,
                                          const double *w_inner, const double *w_outer,
                                          const double *diag_extra, storm_hip_op **out) {
  STORM_REQUIRE(c && out, "op_create: null argument");
  *out = nullptr;
  STORM_REQUIRE(n_owned >= 0 && n_halo >= 0 && n_faces >= 0, "op_create: negative size");
  STORM_REQUIRE(n_faces == 0 || (inner && outer && w_inner && w_outer), "op_create: null face array");
  const int64_t nt = n_owned + n_halo;
  STORM_REQUIRE(nt < (int64_t)INT32_MAX, "op_create: %lld cells exceed int32 indexing", (long long)nt);
  // Validate on the host once, instead of the reference's per-access STORM_ASSERT bounds checks
  // (Utils/Table.hpp:150-154, Feathers/Field.hpp:93-101): a bad index must never reach a kernel.
  BuildTimer timer;
  {
    const int64_t f = first_bad_face(n_faces, inner, outer, nt);
    if (f >= 0) {
      const int64_t a = inner[f], b = outer[f];
      STORM_REQUIRE(a >= 0 && a < nt && b >= 0 && b < nt, "op_create: face %lld joins cells (%lld, %lld) outside [0, %lld)",
                    (long long)f, (long long)a, (long long)b, (long long)nt);
      STORM_REQUIRE(a != b, "op_create: face %lld joins cell %lld to itself", (long long)f, (long long)a);
    }
  }
  std::vector<int64_t> row_ptr;
  std::vector<int> col;
  std::vector<double> val;
  rows_from_faces(n_owned, n_faces, inner, outer, [&](int64_t f, bool outer_side) { return outer_side ? w_outer[f] : w_inner[f]; },
                  row_ptr, col, val);
  timer.lap("rows from faces");
  std::vector<double> ext((size_t)n_owned, 0.0);
  if (diag_extra) std::copy(diag_extra, diag_extra + n_owned, ext.begin());
  return build_op(c, n_owned, n_halo, row_ptr, col, val, ext, out);
}

}  // extern "C"

// from_faces / from_mesh share everything but where a face's transmissibility A_f / d_f comes from
template <class Coef, class BCoef>
static int op_from_faces_impl(storm_hip_ctx *c, int64_t n_owned, int64_t n_halo, int64_t n_faces, const int64_t *inner,
                              const int64_t *outer, Coef &&coef, int64_t n_bfaces, const int64_t *b_cell, BCoef &&b_coef,
                              const double *volume, storm_hip_op **out, const char *who) {
  const int64_t nt = n_owned + n_halo;
  STORM_REQUIRE(nt < (int64_t)INT32_MAX, "%s: %lld cells exceed int32 indexing", who, (long long)nt);
  BuildTimer timer;
  for (int64_t i = 0; i < nt; ++i)
    STORM_REQUIRE(volume[i] > 0.0, "%s: cell %lld has volume %g", who, (long long)i, volume[i]);
  {
    const int64_t f = first_bad_face(n_faces, inner, outer, nt);
    if (f >= 0) {
      const int64_t a = inner[f], b = outer[f];
      STORM_REQUIRE(a >= 0 && a < nt && b >= 0 && b < nt, "%s: face %lld joins cells (%lld, %lld) outside [0, %lld)", who,
                    (long long)f, (long long)a, (long long)b, (long long)nt);
      STORM_REQUIRE(a != b, "%s: face %lld joins cell %lld to itself", who, (long long)f, (long long)a);
    }
  }
  std::vector<int64_t> row_ptr;
  std::vector<int> col;
  std::vector<double> val;
  // w_in = (A_f / d_f) / V_in, w_out = (A_f / d_f) / V_out      Playground.cpp:126-129
  rows_from_faces(n_owned, n_faces, inner, outer,
                  [&](int64_t f, bool outer_side) { return coef(f) / volume[outer_side ? outer[f] : inner[f]]; }, row_ptr, col, val);
  timer.lap("rows from faces");
  std::vector<double> ext((size_t)n_owned, 0.0);
  for (int64_t k = 0; k < n_bfaces; ++k) {  // flux to a zero ghost state at the wall
    const int64_t i = b_cell[k];
    STORM_REQUIRE(i >= 0 && i < n_owned, "%s: boundary face %lld on cell %lld outside [0, %lld)", who, (long long)k,
                  (long long)i, (long long)n_owned);
    ext[(size_t)i] -= b_coef(k) / volume[i];
  }
  return build_op(c, n_owned, n_halo, row_ptr, col, val, ext, out);
}

extern "C" {

int storm_hip_op_create_from_faces(storm_hip_ctx *c, int64_t n_owned, int64_t n_halo, int64_t n_faces,
                                   const int64_t *inner, const int64_t *outer, const double *coef,
                                   int64_t n_bfaces, const int64_t *b_cell, const double *b_coef,
                                   const double *volume, storm_hip_op **out) {
  STORM_REQUIRE(c && out, "op_create_from_faces: null argument");
  *out = nullptr;
  STORM_REQUIRE(n_owned >= 0 && n_halo >= 0 && n_faces >= 0 && n_bfaces >= 0, "op_create_from_faces: negative size");
  STORM_REQUIRE(volume && (n_faces == 0 || (inner && outer && coef)) && (n_bfaces == 0 || (b_cell && b_coef)),
                "op_create_from_faces: null array");
  return op_from_faces_impl(c, n_owned, n_halo, n_faces, inner, outer, [&](int64_t f) { return coef[f]; }, n_bfaces, b_cell,
                            [&](int64_t k) { return b_coef[k]; }, volume, out, "op_create_from_faces");
}

// length(a - b) as the reference forms it (MatrixAlgorithms.hpp:303-305 -> norm_2 :262-270): squares added left to
// right, one rounding per operation (no contraction: the coefficients must be the bits the host's numpy / the
// reference's scalar loop give).
static inline double center_distance(const double *a, const double *b, int dim) {
#pragma clang fp contract(off)
  double s = 0.0;
  for (int k = 0; k < dim; ++k) {
    const double d = a[k] - b[k];
    s = s + d * d;
  }
  return sqrt(s);
}

int storm_hip_op_create_from_mesh(storm_hip_ctx *c, int64_t n_owned, int64_t n_halo, int32_t dim, int64_t n_faces,
                                  const int64_t *inner, const int64_t *outer, const double *area, const double *center,
                                  int64_t n_bfaces, const int64_t *b_cell, const double *b_area, const double *b_center,
                                  const double *volume, storm_hip_op **out) {
  STORM_REQUIRE(c && out, "op_create_from_mesh: null argument");
  *out = nullptr;
  STORM_REQUIRE(n_owned >= 0 && n_halo >= 0 && n_faces >= 0 && n_bfaces >= 0 && dim >= 1 && dim <= 3,
                "op_create_from_mesh: bad size (dim = %d)", (int)dim);
  STORM_REQUIRE(volume && center && (n_faces == 0 || (inner && outer && area)) && (n_bfaces == 0 || (b_cell && b_area && b_center)),
                "op_create_from_mesh: null array");
  return op_from_faces_impl(
      c, n_owned, n_halo, n_faces, inner, outer,
      [&](int64_t f) { return area[f] / center_distance(center + outer[f] * dim, center + inner[f] * dim, dim); }, n_bfaces, b_cell,
      [&](int64_t k) { return b_area[k] / center_distance(b_center + k * dim, center + b_cell[k] * dim, dim); }, volume, out,
      "op_create_from_mesh");
}

int storm_hip_op_create_csr(storm_hip_ctx *c, int64_t n_rows, int64_t n_halo, const int64_t *row_ptr,
                            const int64_t *col, const double *val, storm_hip_op **out) {
  STORM_REQUIRE(c && out && row_ptr, "op_create_csr: null argument");
  *out = nullptr;
  STORM_REQUIRE(n_rows >= 0 && n_halo >= 0, "op_create_csr: negative size");
  const int64_t nt = n_rows + n_halo;
  STORM_REQUIRE(nt < (int64_t)INT32_MAX, "op_create_csr: %lld columns exceed int32 indexing", (long long)nt);
  STORM_REQUIRE(row_ptr[0] == 0, "op_create_csr: row_ptr[0] != 0");
  std::vector<int64_t> rp((size_t)n_rows + 1, 0);
  std::vector<int> oc;
  std::vector<double> ov;
  std::vector<double> ext((size_t)n_rows, 0.0);
  oc.reserve((size_t)row_ptr[n_rows]);
  ov.reserve((size_t)row_ptr[n_rows]);
  for (int64_t i = 0; i < n_rows; ++i) {
    STORM_REQUIRE(row_ptr[i + 1] >= row_ptr[i], "op_create_csr: row_ptr not monotone at row %lld", (long long)i);
    double rowsum = 0.0;  // M x = sum_j a_ij (x_j - x_i) + (sum_j a_ij) x_i
    for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
      STORM_REQUIRE(col[k] >= 0 && col[k] < nt, "op_create_csr: column %lld of row %lld outside [0, %lld)",
                    (long long)col[k], (long long)i, (long long)nt);
      rowsum += val[k];
      if (col[k] != i) {
        oc.push_back((int)col[k]);
        ov.push_back(val[k]);
      }
    }
    ext[(size_t)i] = rowsum;
    rp[(size_t)i + 1] = (int64_t)oc.size();
  }
  return build_op(c, n_rows, n_halo, rp, oc, ov, ext, out);
}

int storm_hip_op_apply(const storm_hip_op *op, double alpha, double beta, const storm_hip_vec *x,
                       storm_hip_vec *y) {
  STORM_REQUIRE(op && x && y, "op_apply: null argument");
  STORM_REQUIRE(x->ctx == op->ctx && y->ctx == op->ctx, "op_apply: context mismatch");
  STORM_REQUIRE(x != y && x->d != y->d, "op_apply: x and y must not alias");
  STORM_REQUIRE(x->n_owned == op->n_rows && y->n_owned == op->n_rows, "op_apply: operator has %lld rows, x %lld, y %lld",
                (long long)op->n_rows, (long long)x->n_owned, (long long)y->n_owned);
  STORM_REQUIRE(x->n_halo >= op->n_halo, "op_apply: x has %lld halo rows, operator needs %lld", (long long)x->n_halo,
                (long long)op->n_halo);
  return spmv_launch(op, host_scal(alpha), host_scal(beta), x->d, y->d, nullptr, op->ctx->api_done);
}

int storm_hip_op_apply_add(const storm_hip_op *op, double alpha, const storm_hip_vec *x, storm_hip_vec *y) {
  STORM_REQUIRE(op && x && y, "op_apply_add: null argument");
  STORM_REQUIRE(x->ctx == op->ctx && y->ctx == op->ctx, "op_apply_add: context mismatch");
  STORM_REQUIRE(x != y && x->d != y->d, "op_apply_add: x and y must not alias");
  STORM_REQUIRE(x->n_owned == op->n_rows && y->n_owned == op->n_rows, "op_apply_add: operator has %lld rows, x %lld, y %lld",
                (long long)op->n_rows, (long long)x->n_owned, (long long)y->n_owned);
  STORM_REQUIRE(x->n_halo >= op->n_halo, "op_apply_add: x has %lld halo rows, operator needs %lld", (long long)x->n_halo,
                (long long)op->n_halo);
  return spmv_launch(op, host_scal(alpha), host_scal(0.0), x->d, y->d, nullptr, op->ctx->api_done, true);
}

int storm_hip_op_get_diagonal(const storm_hip_op *op, double alpha, double beta, int invert, storm_hip_vec *d) {
  STORM_REQUIRE(op && d, "op_get_diagonal: null argument");
  STORM_REQUIRE(d->ctx == op->ctx, "op_get_diagonal: context mismatch");
  STORM_REQUIRE(d->n_owned == op->n_rows, "op_get_diagonal: operator has %lld rows, d %lld", (long long)op->n_rows,
                (long long)d->n_owned);
  if (op->n_rows == 0) return STORM_HIP_OK;
  storm_hip_ctx *c = op->ctx;
  HIP_TRY(hipSetDevice(c->device));
  const int64_t n64 = (op->n_rows + kWave - 1) / kWave;  // the kernel walks 64-row groups whatever the format
  const int nb = (int)((n64 + (kBlock / kWave) - 1) / (kBlock / kWave));
  hipLaunchKernelGGL(diag_sell_kernel, dim3(nb), dim3(kBlock), 0, c->stream, op->d_pack, op->d_slice_off, op->n_rows,
                     op->d_dict, op->pair >= 2 ? op->pair + 2 : op->pair ? 3 : (int)(op->offs_size > 0), alpha, beta, d->d,
                     op->d_types);
  if (op->tail_rows > 0)
    hipLaunchKernelGGL(diag_tail_kernel, dim3((int)((op->tail_rows + 255) / 256)), dim3(256), 0, c->stream,
                       op->tail_rows, op->d_tail_row, op->d_tail_ptr, op->d_tail_val, alpha, d->d);
  if (invert)
    hipLaunchKernelGGL(safe_invert_kernel, dim3((int)((op->n_rows + 255) / 256)), dim3(256), 0, c->stream, op->n_rows,
                       d->d);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

int storm_hip_op_get_stats(const storm_hip_op *op, storm_hip_op_stats *s) {
  STORM_REQUIRE(op && s, "op_get_stats: null argument");
  s->n_rows = op->n_rows;
  s->n_cols = op->n_rows + op->n_halo;
  s->nnz_offdiag = op->nnz;
  s->ell_slots = op->ell_slots;
  s->tail_nnz = op->tail_nnz;
  s->tail_rows = op->tail_rows;
  s->n_slices = op->n_slices;
  s->max_row_len = op->max_row_len;
  s->n_interior_slices = op->n_interior_slices;
  s->device_bytes = op->device_bytes;
  s->record_bytes = op->pack_bytes;
  s->value_dictionary_size = op->dict_size;
  s->offset_dictionary_size = op->offs_size;
  s->paired_rows = op->pair;
  CanonTileArgs T;
  int nbt = 0;
  s->tiled_planes = canon_tile_geometry(op, &T, &nbt, op->d_bnd_pack != nullptr) ? canon_tile_planes(op) : 0;  // (mixed operator: its interior planes)
  s->spmv_blocks = spmv_grid_blocks(op);
  return STORM_HIP_OK;
}

int storm_hip_op_destroy(storm_hip_op *op) {
  if (!op) return STORM_HIP_OK;
  if (op->ctx) (void)storm_hip_ctx_sync(op->ctx);
  (void)hipFree(op->d_interior);
  (void)hipFree(op->d_boundary);
  (void)hipFree(op->d_slice_off);
  (void)hipFree(op->d_pack);
  (void)hipFree(op->d_bnd_pack);
  (void)hipFree(op->d_types);
  (void)hipFree(op->d_dict);
  (void)hipFree(op->d_offs);
  (void)hipFree(op->d_tail_row);
  (void)hipFree(op->d_tail_ptr);
  (void)hipFree(op->d_tail_col);
  (void)hipFree(op->d_tail_val);
  (void)hipFree(op->d_lat_pack);
  (void)hipFree(op->d_lat_off);
  (void)hipFree(op->halo.d_send_idx);
  (void)hipFree(op->halo.d_sendbuf);
  delete op;
  return STORM_HIP_OK;
}

}  // extern "C"
