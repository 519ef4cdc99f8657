// The latency path: CG and BiCGStab for SMALL operators as ONE cooperative, persistent kernel per solve.
//
// The reference's own meshes have 6 000 .. 80 000 cells (tests/_data/mesh), BASELINE config 1 has 64^3 = 262 144:
// vectors of 50 KB .. 2 MB.  The throughput path (solvers.hip) spends such an iteration on launch latency -- 7 kernels
// of a few microseconds each.  Here a solve is one launch (SolverCg.hpp:54-126 inside Solver.hpp:116-147):
//
//   * every wavefront owns a fixed set of 64-row slices for the whole solve and keeps x, r, p, z of its rows in
//     REGISTERS; what other wavefronts need for their gathers is published once per iteration -- the rows of the
//     new r and of the CURRENT p -- and a gathering wave forms the neighbour's next direction itself,
//     p'[c] = r[c] + beta p[c], with the same expression (hence the same bits) as the owner does in registers.
//     That removes the third synchronisation point of a CG iteration ("p complete"): TWO grid barriers per
//     iteration remain, one behind each reduction;
//   * the operator is read from a compact fp64 sliced-ELL copy made when the operator was built
//     ([ext 64 f64][col W x 64 i32][val W x 64 f64] per slice, slot-major; small: it stays in L2 / Infinity Cache);
//   * a reduction IS the barrier: every block publishes its partial in its own slot as two self-validating 8-byte
//     words { half of the value, sequence number } (fire and forget: no ordering to rely on), and every block polls
//     all slots with single 16-byte coherent loads until both words of each carry the current sequence number, then
//     folds the values in slot order.  All
//     blocks hold bit-identical alpha, beta and the same convergence verdict (the exit condition is uniform), and
//     a synchronisation point costs about 2.5 memory round trips instead of the 6 of "partials, counter barrier,
//     read partials" (an iteration is a chain of ~0.8 us round trips; nothing else matters at this size);
//   * the records of a wave's slices are loaded into registers once (<= 8 slots per row, <= 2 slices per wave).
//   * rows are summed slot by slot exactly as the throughput kernels do (same expression, same contraction): the
//     SpMV values are bit-identical; dot products group their terms differently (rounding-level differences).
//
// Taken by storm_hip_solve_cg / storm_hip_solve_bicgstab when the operator has a latency copy (n_rows <= option `latency_rows`, no halo, no
// CSR tail), the context has no communicator, and option `latency_path` != 0.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "coop_device.hpp"
#include "spmv_device.hpp"
#include "solver_device.hpp"

namespace storm {

// Blocks of `fn` that fit a CU (0: the kernel cannot run with this much dynamic LDS), asked once PER CONTEXT: the answer --
// and the hipFuncAttributeMaxDynamicSharedMemorySize it needs -- belong to the device, a context is one device and one
// host thread (a process-wide static cache served a second device with the first one's answer and raced between threads).
static int occupancy_cached(storm_hip_ctx *c, const void *fn, int threads, size_t dyn_lds) {
  const auto it = c->occupancy.find(fn);
  if (it != c->occupancy.end()) return it->second;
  int res = 0;
  if (dyn_lds == 0 || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds) == hipSuccess)
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&res, fn, threads, dyn_lds);
  (void)hipGetLastError();
  c->occupancy[fn] = res;
  return res;
}


constexpr int kLatBlock = 1024;  // one block per CU: a synchronisation point costs per participating BLOCK
constexpr int kLatWaves = kLatBlock / kWave;

static bool coop_launch(storm_hip_ctx *c, const void *fn, unsigned blocks, void **args, size_t dyn_lds = 0, unsigned threads = kLatBlock);  // (below: a refused launch is a fallback, not an error)

struct LatArgs {
  const char *pack;          // compact records
  const int64_t *rec_off;    // [n_slices + 1] byte offsets
  int64_t n_rows, n_slices;
  double alpha, beta;        // A = beta I + alpha M
  const double *b;
  double *x;
  double *p, *r;             // published rows of the current direction and the new residual (see the header)
  double *v0, *v1;           // BiCGStab: published rows of v = A p of even / odd iterations
  char *slots;               // all-reduce slots, kLatSlotStride bytes per block, zeroed before the launch
  SolverState *st;
  int publish_xchg;          // rows are published with atomic exchanges whose return is awaited (option latency_publish)
};

// Sum over all blocks of `mine` (a per-thread partial), identical bits in every thread of every block.
// `publishes`: the block's waves have issued coherent stores (rows of r, p) that other blocks read once they are past
// this point -- every wave then drains its own store counter before the block's words go out.
__device__ __forceinline__ double lat_allreduce(double mine, char *slots, unsigned long long seq, double *lds,
                                                bool publishes = true, unsigned long long seen = 0ull) {
  const unsigned tag = (unsigned)seq;
  double v = lat_wave_sum(mine);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  asm volatile("" : : "v"(seen) : "memory");  // the exchanges that published this wave's rows have returned
  if (publishes) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // ... (store variant: acknowledged)
  __syncthreads();  // (lds may still be read by the previous call)
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kLatWaves; ++w) t += lds[w];
    co_store_slot(slots + lat_slot_offset(blockIdx.x, seq), tag, t);
  }
  v = 0.0;
  if (threadIdx.x < gridDim.x) {  // gridDim.x <= 256 <= blockDim.x: thread t watches block t
    const char *slot = slots + lat_slot_offset(threadIdx.x, seq);
    // Every wait is bounded: the grid is launched cooperatively (all blocks resident), but should a block never
    // arrive -- the device shared with another process's cooperative kernel, say -- the others give up after
    // kLatTimeoutTicks instead of spinning forever, raise the flag behind the slots and fall through every later
    // wait at once; the host turns the flag into an error.
    int *gave_up = reinterpret_cast<int *>(slots + (size_t)2 * 256 * kLatSlotStride);
    const long long t0 = wall_clock64();
    for (int spins = 0;; ++spins) {
      if (co_load_slot(slot, tag, &v)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 1023) == 1023 &&
          (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  // slot order: lanes, then the (up to four) polling waves -- the same tree in every block
  v = lat_wave_sum(v);
  __syncthreads();
  if (lane == 0 && wave < 4) lds[wave] = v;
  __syncthreads();
  return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// The same for TWO sums at once (BiCGStab's <t, s>, <t, t> and <r, r>, <rt, r>): the slot carries four words
// (co_load_slot2, coop_device.hpp).
__device__ __forceinline__ void lat_allreduce2(double &s0, double &s1, char *slots, unsigned long long seq, double *lds,
                                               bool publishes = true, unsigned long long seen = 0ull) {
  const unsigned tag = (unsigned)seq;
  double v0 = lat_wave_sum(s0), v1 = lat_wave_sum(s1);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  asm volatile("" : : "v"(seen) : "memory");
  if (publishes) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (lane == 0) lds[wave] = v0, lds[kLatWaves + wave] = v1;
  __syncthreads();
  if (threadIdx.x < 2) {  // thread 0 folds and stores the first sum, thread 1 the second
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kLatWaves; ++w) t += lds[threadIdx.x * kLatWaves + w];
    co_store_slot(slots + lat_slot_offset(blockIdx.x, seq) + 16 * threadIdx.x, tag, t);
  }
  v0 = v1 = 0.0;
  if (threadIdx.x < gridDim.x) {
    const char *slot = slots + lat_slot_offset(threadIdx.x, seq);
    int *gave_up = reinterpret_cast<int *>(slots + (size_t)2 * 256 * kLatSlotStride);
    const long long t0 = wall_clock64();
    for (int spins = 0;; ++spins) {
      if (co_load_slot2(slot, tag, &v0, &v1)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 1023) == 1023 &&
          (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v0 = v1 = 0.0;
        break;
      }
    }
  }
  v0 = lat_wave_sum(v0), v1 = lat_wave_sum(v1);
  __syncthreads();
  if (lane == 0 && wave < 4) lds[wave] = v0, lds[kLatWaves + wave] = v1;
  __syncthreads();
  s0 = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  s1 = (lds[kLatWaves] + lds[kLatWaves + 1]) + (lds[kLatWaves + 2] + lds[kLatWaves + 3]);
}

// ... and THREE (the paired Gram-Schmidt step: <w, q_i>, <w, q_i+1>, <q_i, q_i+1>): six words, 48 bytes of the slot.
__device__ __forceinline__ void lat_allreduce3(double &s0, double &s1, double &s2, char *slots, unsigned long long seq,
                                               double *lds /* [3 * kLatWaves] */) {
  const unsigned tag = (unsigned)seq;
  double v0 = lat_wave_sum(s0), v1 = lat_wave_sum(s1), v2 = lat_wave_sum(s2);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[wave] = v0, lds[kLatWaves + wave] = v1, lds[2 * kLatWaves + wave] = v2;
  __syncthreads();
  if (threadIdx.x < 3) {  // thread j folds and stores sum j
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kLatWaves; ++w) t += lds[threadIdx.x * kLatWaves + w];
    co_store_slot(slots + lat_slot_offset(blockIdx.x, seq) + 16 * threadIdx.x, tag, t);
  }
  v0 = v1 = v2 = 0.0;
  if (threadIdx.x < gridDim.x) {
    const char *slot = slots + lat_slot_offset(threadIdx.x, seq);
    int *gave_up = reinterpret_cast<int *>(slots + (size_t)2 * 256 * kLatSlotStride);
    const long long t0 = wall_clock64();
    for (int spins = 0;; ++spins) {
      if (co_load_slot3(slot, tag, &v0, &v1, &v2)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 1023) == 1023 &&
          (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v0 = v1 = v2 = 0.0;
        break;
      }
    }
  }
  v0 = lat_wave_sum(v0), v1 = lat_wave_sum(v1), v2 = lat_wave_sum(v2);
  __syncthreads();
  if (lane == 0 && wave < 4) lds[wave] = v0, lds[kLatWaves + wave] = v1, lds[2 * kLatWaves + wave] = v2;
  __syncthreads();
  s0 = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  s1 = (lds[kLatWaves] + lds[kLatWaves + 1]) + (lds[kLatWaves + 2] + lds[kLatWaves + 3]);
  s2 = (lds[2 * kLatWaves] + lds[2 * kLatWaves + 1]) + (lds[2 * kLatWaves + 2] + lds[2 * kLatWaves + 3]);
}

// Neighbour value of the vector an SpMV is applied to: plain x (init), or the direction p' = r + beta p formed
// from the published rows.
struct LatPlain {
  const double *v;
  __device__ __forceinline__ double operator()(int c) const { return v[c]; }
};
struct LatDirection {
  const double *r, *p;
  double beta;
  __device__ __forceinline__ double operator()(int c) const { return co_load(r + c) + beta * co_load(p + c); }
};

constexpr int kLatCacheWidth = 8;  // slots per row held in registers, at most
template <int S, int W>
struct LatRecords {  // the records of a wave's slices: W slots per row in registers, or (W == 0) re-read every time
  int col[W ? S : 1][W ? W : 1];
  double val[W ? S : 1][W ? W : 1];
  double ext[W ? S : 1];
};

// (M v)_row for one row of slice s: sum_k w_k (v[col_k] - v_i) + ext v_i, slots in order.
template <class Get>
__device__ __forceinline__ double lat_row(const LatArgs &a, int64_t s, int lane, const Get &get, double vi) {
  const int64_t o0 = a.rec_off[s];
  const int width = (int)((a.rec_off[s + 1] - o0 - kWave * 8) / (kWave * 12));
  const char *rec = a.pack + o0;
  const double ext = reinterpret_cast<const double *>(rec)[lane];
  const int *col = reinterpret_cast<const int *>(rec + kWave * 8) + lane;
  const double *val = reinterpret_cast<const double *>(rec + kWave * 8 + (int64_t)width * (kWave * 4)) + lane;
  double acc = 0.0;
  int k = 0;
  for (; k + 4 <= width; k += 4) {  // four neighbours in flight
    const int c0 = col[k * kWave], c1 = col[(k + 1) * kWave], c2 = col[(k + 2) * kWave], c3 = col[(k + 3) * kWave];
    const double w0 = val[k * kWave], w1 = val[(k + 1) * kWave], w2 = val[(k + 2) * kWave], w3 = val[(k + 3) * kWave];
    const double g0 = get(c0), g1 = get(c1), g2 = get(c2), g3 = get(c3);
    acc += w0 * (g0 - vi);
    acc += w1 * (g1 - vi);
    acc += w2 * (g2 - vi);
    acc += w3 * (g3 - vi);
  }
  for (; k < width; ++k) acc += val[k * kWave] * (get(col[k * kWave]) - vi);
  return a.beta * vi + a.alpha * (acc + ext * vi);
}
// The same from registers: W slots (4: triangle / quadrilateral meshes; 8), the ones past the row's width carry weight 0 and the row's own
// column (a term 0 * (v_i - v_i) leaves the sum as it is).
// (CHUNK neighbours in flight at a time: a neighbour costs one load with LatPlain, up to three with BiCGStab's.)
template <int S, int W, int CHUNK = W, class Get>
__device__ __forceinline__ double lat_row_cached(const LatArgs &a, const LatRecords<S, W> &rec, int q, const Get &get,
                                                 double vi) {
  double acc = 0.0;
#pragma unroll
  for (int k0 = 0; k0 < W; k0 += CHUNK) {
    double g[CHUNK];
#pragma unroll
    for (int k = 0; k < CHUNK; ++k) g[k] = get(rec.col[q][k0 + k]);
#pragma unroll
    for (int k = 0; k < CHUNK; ++k) acc += rec.val[q][k0 + k] * (g[k] - vi);
  }
  return a.beta * vi + a.alpha * (acc + rec.ext[q] * vi);
}

// The records of a wave's slices into registers (W > 0 variants).
template <int S, int W>
__device__ __forceinline__ void lat_load_records(const LatArgs &a, int64_t wave_id, int64_t n_waves, int lane,
                                                 LatRecords<S, W> &rec) {
  if (W > 0) {
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int64_t s = wave_id + q * n_waves, row = s * kWave + lane;
      const bool live = s < a.n_slices;
      const int64_t o0 = live ? a.rec_off[s] : 0;
      const int width = live ? (int)((a.rec_off[s + 1] - o0 - kWave * 8) / (kWave * 12)) : 0;
      const char *base = a.pack + o0;
      rec.ext[q] = live ? reinterpret_cast<const double *>(base)[lane] : 0.0;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const bool has = k < width;
        rec.col[q][k] = has ? (reinterpret_cast<const int *>(base + kWave * 8) + lane)[k * kWave]
                            : (int)(row < a.n_rows ? row : a.n_rows - 1);
        rec.val[q][k] = has ? (reinterpret_cast<const double *>(base + kWave * 8 + (int64_t)width * (kWave * 4)) + lane)[k * kWave]
                            : 0.0;
      }
    }
  }
}

template <int S, int W>
__global__ __launch_bounds__(kLatBlock) void cg_latency_kernel(LatArgs a) {
  __shared__ double lds[kLatWaves];
  SolverState *st = a.st;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave_id = (int64_t)blockIdx.x * kLatWaves + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * kLatWaves;
  unsigned long long seq = 0, seen = 0;  // seen: what the publishing exchanges returned (consumed at the all-reduces)
  double x[S], r[S], p[S], z[S];
  LatRecords<S, W> rec;
  lat_load_records<S, W>(a, wave_id, n_waves, lane, rec);
  auto apply_row = [&](int q, int64_t s, const auto &get, double vi) -> double {
    if constexpr (W > 0) return lat_row_cached<S, W>(a, rec, q, get, vi);
    else return lat_row(a, s, lane, get, vi);
  };

  // ---- init: r = b - A x; p = r; gamma = <r, r>                                   SolverCg.hpp:54-84
  // (a.p arrives zero-filled -- a fresh work vector -- so the first direction r + 0 * p is r)
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < S; ++q) {
    const int64_t s = wave_id + q * n_waves, row = s * kWave + lane;
    const bool valid = s < a.n_slices && row < a.n_rows;
    x[q] = valid ? a.x[row] : 0.0;
    r[q] = p[q] = z[q] = 0.0;
    if (s < a.n_slices) {
      const double ax = apply_row(q, s, LatPlain{a.x}, x[q]);  // x is not written before the kernel's end
      r[q] = valid ? a.b[row] - ax : 0.0;
      p[q] = r[q];
      if (valid) co_publish(a.r + row, r[q], a.publish_xchg, seen);
      acc += r[q] * r[q];
    }
  }
  double gamma = lat_allreduce(acc, a.slots, ++seq, lds, true, seen);
  const double initial_error = sqrt(gamma);
  const double abs_tol = st->abs_tol, rel_tol = st->rel_tol;
  const long long num_iterations = st->num_iterations;
  double *history = st->history;
  bool converged = abs_tol > 0.0 && initial_error < abs_tol;  // Solver.hpp:124-128
  double abs_err = initial_error, rel_err = 0.0, beta = 0.0;
  long long it = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[0] = initial_error;

  // ---- iterations                                                                 SolverCg.hpp:86-126
  // entering: registers hold x, r and the direction p of the own rows; memory holds r and the PREVIOUS direction,
  // from which a neighbour's current direction is r[c] + beta p_prev[c]
  while (!converged && it < num_iterations) {
    acc = 0.0;
    const LatDirection dir{a.r, a.p, beta};
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int64_t s = wave_id + q * n_waves;
      if (s < a.n_slices) {
        z[q] = apply_row(q, s, dir, p[q]);
        z[q] = (s * kWave + lane < a.n_rows) ? z[q] : 0.0;
        acc += p[q] * z[q];
      }
    }
    // every gather of this iteration is done once all blocks have published their <p, z> partial
    const double alpha = safe_divide(gamma, lat_allreduce(acc, a.slots, ++seq, lds));
    acc = 0.0;
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int64_t s = wave_id + q * n_waves, row = s * kWave + lane;
      x[q] += alpha * p[q];
      r[q] -= alpha * z[q];
      acc += r[q] * r[q];
      if (s < a.n_slices && row < a.n_rows)
        co_publish(a.r + row, r[q], a.publish_xchg, seen), co_publish(a.p + row, p[q], a.publish_xchg, seen);
    }
    // the new r and the current p are out (at the point of coherence before a block's tag is stored) with the <r, r> partials
    const double gamma_bar = gamma;
    gamma = lat_allreduce(acc, a.slots, ++seq, lds, true, seen);
    beta = safe_divide(gamma, gamma_bar);
    abs_err = sqrt(gamma);
    rel_err = abs_err / initial_error;
    converged = (abs_tol > 0.0 && abs_err < abs_tol) || (rel_tol > 0.0 && rel_err < rel_tol);  // Solver.hpp:132-140
    ++it;
    if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[it] = abs_err;
#pragma unroll
    for (int q = 0; q < S; ++q) p[q] = r[q] + beta * p[q];
  }
#pragma unroll
  for (int q = 0; q < S; ++q) {
    const int64_t s = wave_id + q * n_waves, row = s * kWave + lane;
    if (s < a.n_slices && row < a.n_rows) a.x[row] = x[q];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->initial_error = initial_error;
    st->absolute_error = abs_err;
    st->relative_error = rel_err;
    st->iteration = it;
    st->converged = converged ? 1 : 0;
    st->done = 1;
  }
}

// ---- BiCGStab on the latency path ---------------------------------------------------------------------------------
// SolverBiCgStab.hpp:60-167 inside Solver.hpp:116-147, THREE synchronisation points per iteration (the throughput path
// spends 13 launches on it).  Registers hold x, r, p, v, rt of the own rows.  What a neighbour needs is published as
// rows of r (the residual at the iteration's end), p, and v (two buffers, alternating by iteration); a gathering wave
// forms the vector the operator is applied to itself, by the expression the owner uses:
//   A p':  p'[c] = r[c] + beta (p[c] - omega v[c])            from r, p, v as of the END of the previous iteration
//   A s :  s[c]  = r[c] - alpha v'[c]                         from the same r and THIS iteration's v (other buffer)
// so neither "p complete" nor "s complete" is a barrier of its own: the all-reduces of <rt, v>, (<t, s>, <t, t>) and
// (<r, r>, <rt, r>) are the only ones.  A buffer is overwritten only after an all-reduce that every block enters
// after its last gather from it (r, p: behind the omega all-reduce; v of parity k: written in iteration k + 2, read
// last in iteration k + 1 before its first all-reduce).
__device__ __forceinline__ double bicg_direction(double r, double p, double v, double beta, double omega) {
  return __builtin_fma(beta, __builtin_fma(-omega, v, p), r);  // r + beta (p - omega v)      SolverBiCgStab.hpp:119
}
__device__ __forceinline__ double bicg_half_residual(double r, double v, double alpha) {
  return __builtin_fma(-alpha, v, r);  // r - alpha v                                         SolverBiCgStab.hpp:141
}
struct LatBicgDirection {
  const double *r, *p, *v;
  double beta, omega;
  __device__ __forceinline__ double operator()(int c) const {
    return bicg_direction(co_load(r + c), co_load(p + c), co_load(v + c), beta, omega);
  }
};
struct LatBicgHalf {
  const double *r, *v;
  double alpha;
  __device__ __forceinline__ double operator()(int c) const { return bicg_half_residual(co_load(r + c), co_load(v + c), alpha); }
};

template <int S, int W>
__global__ __launch_bounds__(kLatBlock) void bicgstab_latency_kernel(LatArgs a) {
  __shared__ double lds[2 * kLatWaves];
  SolverState *st = a.st;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave_id = (int64_t)blockIdx.x * kLatWaves + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * kLatWaves;
  unsigned long long seq = 0, seen = 0;  // seen: what the publishing exchanges returned (consumed at the all-reduces)
  double x[S], r[S], p[S], v[S], rt[S];
  LatRecords<S, W> rec;
  lat_load_records<S, W>(a, wave_id, n_waves, lane, rec);
  auto apply_row = [&](int q, int64_t s, const auto &get, double vi) -> double {
    if constexpr (W > 0) return lat_row_cached<S, W, 4>(a, rec, q, get, vi);
    else return lat_row(a, s, lane, get, vi);
  };

  // ---- init: r = b - A x; rt = r; rho = <rt, r>                                   SolverBiCgStab.hpp:82-90
  // (a.p, a.v0, a.v1 arrive zero-filled: the first direction r + 0 * (p - 0 * v) is r, :114)
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < S; ++q) {
    const int64_t s = wave_id + q * n_waves, row = s * kWave + lane;
    const bool valid = s < a.n_slices && row < a.n_rows;
    x[q] = valid ? a.x[row] : 0.0;
    r[q] = p[q] = v[q] = rt[q] = 0.0;
    if (s < a.n_slices) {
      const double ax = apply_row(q, s, LatPlain{a.x}, x[q]);
      r[q] = valid ? a.b[row] - ax : 0.0;
      rt[q] = r[q];
      if (valid) co_publish(a.r + row, r[q], a.publish_xchg, seen);
      acc += rt[q] * r[q];
    }
  }
  double rho = lat_allreduce(acc, a.slots, ++seq, lds, true, seen);
  const double initial_error = sqrt(rho);
  const double abs_tol = st->abs_tol, rel_tol = st->rel_tol;
  const long long num_iterations = st->num_iterations;
  double *history = st->history;
  bool converged = abs_tol > 0.0 && initial_error < abs_tol;  // Solver.hpp:124-128
  double abs_err = initial_error, rel_err = 0.0, alpha = 0.0, beta = 0.0, omega = 0.0;
  long long it = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[0] = initial_error;

  while (!converged && it < num_iterations) {
    double *v_prev = (it & 1) ? a.v0 : a.v1, *v_cur = (it & 1) ? a.v1 : a.v0;
    // p = r + beta (p - omega v) (own rows, registers); v = A p; <rt, v>              :114-119, :137-139
    acc = 0.0;
    const LatBicgDirection dir{a.r, a.p, v_prev, beta, omega};
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int64_t s = wave_id + q * n_waves, row = s * kWave + lane;
      p[q] = bicg_direction(r[q], p[q], v[q], beta, omega);
      if (s < a.n_slices) {
        v[q] = apply_row(q, s, dir, p[q]);
        v[q] = (row < a.n_rows) ? v[q] : 0.0;
        if (row < a.n_rows) co_publish(v_cur + row, v[q], a.publish_xchg, seen);
        acc += rt[q] * v[q];
      }
    }
    alpha = safe_divide(rho, lat_allreduce(acc, a.slots, ++seq, lds, true, seen));
    // s = r - alpha v (kept in r); t = A s; omega = <t, s> / <t, t>                   :140-141, :158-160
    double t[S];
    double acc_ts = 0.0, acc_tt = 0.0;
    const LatBicgHalf half{a.r, v_cur, alpha};
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int64_t s = wave_id + q * n_waves, row = s * kWave + lane;
      r[q] = bicg_half_residual(r[q], v[q], alpha);
      t[q] = 0.0;
      if (s < a.n_slices) {
        t[q] = apply_row(q, s, half, r[q]);
        t[q] = (row < a.n_rows) ? t[q] : 0.0;
        acc_ts += t[q] * r[q];
        acc_tt += t[q] * t[q];
      }
    }
    lat_allreduce2(acc_ts, acc_tt, a.slots, ++seq, lds, false);  // nothing published since the last one
    omega = safe_divide(acc_ts, acc_tt);
    // x += alpha p + omega s; r = s - omega t; |r|, <rt, r>                           :140, :161-164, :116
    double acc_rr = 0.0, acc_rho = 0.0;
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int64_t s = wave_id + q * n_waves, row = s * kWave + lane;
      x[q] += alpha * p[q];
      x[q] += omega * r[q];
      r[q] -= omega * t[q];
      acc_rr += r[q] * r[q];
      acc_rho += rt[q] * r[q];
      if (s < a.n_slices && row < a.n_rows)
        co_publish(a.r + row, r[q], a.publish_xchg, seen), co_publish(a.p + row, p[q], a.publish_xchg, seen);
    }
    lat_allreduce2(acc_rr, acc_rho, a.slots, ++seq, lds, true, seen);
    const double rho_bar = rho;
    rho = acc_rho;
    beta = safe_divide(alpha * rho, omega * rho_bar);  // :116-118, for the next iteration
    abs_err = sqrt(acc_rr);
    rel_err = abs_err / initial_error;
    converged = (abs_tol > 0.0 && abs_err < abs_tol) || (rel_tol > 0.0 && rel_err < rel_tol);  // Solver.hpp:132-140
    ++it;
    if (blockIdx.x == 0 && threadIdx.x == 0 && history) history[it] = abs_err;
  }
#pragma unroll
  for (int q = 0; q < S; ++q) {
    const int64_t s = wave_id + q * n_waves, row = s * kWave + lane;
    if (s < a.n_slices && row < a.n_rows) a.x[row] = x[q];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->initial_error = initial_error;
    st->absolute_error = abs_err;
    st->relative_error = rel_err;
    st->iteration = it;
    st->converged = converged ? 1 : 0;
    st->done = 1;
  }
}

// ---- modified Gram-Schmidt as ONE cooperative kernel -------------------------------------------------------------
// GMRES's Arnoldi step orthogonalises w = A q_k against q_0 .. q_k one after the other (SolverGmres.hpp:157-161); each
// step needs a global reduction before the next may start.  The throughput path runs a kernel per step that reads w,
// q_i and q_{i+1} and writes w (32 B/row/step, 13 us at 128^3: launch + HBM).  Here every wavefront keeps ITS rows of w
// in registers for the whole chain, streams the rows of q_i through once (prefetching q_{i+1} while the all-reduce of
// step i is in flight), and the k + 2 reductions are the tagged-slot all-reduces of the latency path: 8 B/row/step and
// ~3 us per step.  Same values in the same order (h_i = <w, q_i> with the updated w; w -= h_i q_i), the reduction
// trees differ in rounding only.  Finishes with h_{k+1,k}^2 = <w, w> and (optionally) q_{k+1} = w / sqrt of it.
constexpr int kMgsMaxVectors = 64;
struct MgsArgs {
  const double *q[kMgsMaxVectors];
  double *w;          // in: A q_k; out: q_{k+1} (normalised when `normalise`)
  double *H;          // column k of the (m + 1) x m row-major Hessenberg: H[i * m + k]
  double *norm2_out;  // <w, w> after the chain
  int64_t n_rows, n_slices;
  int k, m, normalise;
  int pairs;          // two Gram-Schmidt steps per synchronisation point (see the kernel)
  unsigned long long seq_base;  // tags of this launch: seq_base + 1 .. seq_base + k + 2 (bit 31 set: never a CG tag)
  char *slots;
  const int *done;
  MgsGivens givens;  // st == nullptr: the caller applies the rotations
  long long *prof;   // option resident_profile: [gridDim.x][8] ticks per phase of this launch (diagnostic)
  char *quad_slots;  // mgs_chain_quad_kernel: all-reduce slots of kQuadSlotStride bytes (two-level form) / dense granules
  int dense;         // ... the flat all-reduce with dense value-major slots instead of the two-level one
  int prefetch;      // ... with the next group's vectors requested between its halves (S <= 4)
  int xcd_runs;      // ... and the blocks' chunks of rows dealt out in ONE contiguous run per XCD (see the kernel)
  int descend;       // ... the basis vectors taken in the order k, k - 1, ..., 0 (odd k: see the kernel)
  int rotate_early;  // ... the k earlier rotations of the column under the norm's all-reduce (block 0)
  // mgs_chain_quad_kernel<S, T, true>: w is not read but FORMED -- w = beta x + alpha M(x), x = ap_x (the newest basis
  // vector), from the operator's format-4 records with spmv_canon_kernel's arithmetic (the same bits): the apply's
  // launch and the round trip of w through memory disappear (SolverGmres.hpp:155 inside the kernel that consumes it)
  const char *ap_pack;
  const double *ap_dict, *ap_x;
  int ap_off[6], ap_max_gather;
  double ap_alpha, ap_beta;
};
template <int S>
__global__ __launch_bounds__(kLatBlock) void mgs_chain_kernel(MgsArgs a) {
  if (a.done && *a.done) return;  // (uniform: every block reads the same flag before any of them synchronises)
  __shared__ double lds[3 * kLatWaves];
  // block 0 keeps column k of the Hessenberg and the earlier rotations in LDS: the Givens recurrence at the end is a
  // chain of k dependent steps -- ~1.5 us from LDS, ~9 us through memory
  __shared__ double hcol[kMgsMaxVectors + 1], cs_sh[kMgsMaxVectors], sn_sh[kMgsMaxVectors];
  const bool rotate = a.givens.st != nullptr && blockIdx.x == 0;
  if (rotate && (int)threadIdx.x < a.k) cs_sh[threadIdx.x] = a.givens.cs[threadIdx.x], sn_sh[threadIdx.x] = a.givens.sn[threadIdx.x];
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave_id = (int64_t)blockIdx.x * kLatWaves + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * kLatWaves;
  unsigned long long seq = a.seq_base;
  double w[S], qc[S], qn[S];
  int row[S];  // (-1: no such row; the chain takes at most 2^22 rows)
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t sl = wave_id + s * n_waves;
    row[s] = (sl < a.n_slices && sl * kWave + lane < a.n_rows) ? (int)(sl * kWave + lane) : -1;
    w[s] = row[s] >= 0 ? a.w[row[s]] : 0.0;
    qc[s] = row[s] >= 0 ? a.q[0][row[s]] : 0.0;
    qn[s] = 0.0;
  }
  int i0 = 0;
  if constexpr (S <= 8) if (a.pairs) {  // (16 slices per wavefront leave no registers for the pair's vectors)
    // Two steps per synchronisation point.  The reference's h_{i+1} = <w - h_i q_i, q_{i+1}> is, by bilinearity,
    // <w, q_{i+1}> - h_i <q_i, q_{i+1}>: the three dot products of the right-hand side need only the w BEFORE step i,
    // so they share one all-reduce (the same algorithm; the roundings of the dot products group differently, as
    // with any other summation order).  q_{i+2} travels while the reduction is in flight.
    // (both vectors of the NEXT pair travel while this pair's reduction is in flight: round 3 loaded the second one at
    //  the top of the next pass, 16.8 MB at 128^3 with nothing to hide behind -- 3.4 us per pair)
    double qd[S], qe[S];
    bool have_n = false;  // qn holds q_{i0+1} already
    for (; i0 + 1 <= a.k; i0 += 2) {
      if (!have_n) {
#pragma unroll
        for (int s = 0; s < S; ++s) qn[s] = row[s] >= 0 ? a.q[i0 + 1][row[s]] : 0.0;
      }
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int s = 0; s < S; ++s) s0 += w[s] * qc[s], s1 += w[s] * qn[s], s2 += qc[s] * qn[s];
      if (i0 + 2 <= a.k) {
#pragma unroll
        for (int s = 0; s < S; ++s) qd[s] = row[s] >= 0 ? a.q[i0 + 2][row[s]] : 0.0;
      }
      const bool have_next = i0 + 3 <= a.k;
      if (have_next) {
#pragma unroll
        for (int s = 0; s < S; ++s) qe[s] = row[s] >= 0 ? a.q[i0 + 3][row[s]] : 0.0;
      }
      lat_allreduce3(s0, s1, s2, a.slots, ++seq, lds);
      const double h0 = s0, h1 = s1 - h0 * s2;
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (rotate) hcol[i0] = h0, hcol[i0 + 1] = h1;
        else a.H[(int64_t)i0 * a.m + a.k] = h0, a.H[(int64_t)(i0 + 1) * a.m + a.k] = h1;
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        w[s] -= h0 * qc[s];
        w[s] -= h1 * qn[s];
        qc[s] = qd[s];
        if (have_next) qn[s] = qe[s];
      }
      have_n = have_next;
    }
  }
  for (int i = i0; i <= a.k; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) acc += w[s] * qc[s];
    if (i < a.k) {  // the next basis vector travels while the reduction is in flight
#pragma unroll
      for (int s = 0; s < S; ++s) qn[s] = row[s] >= 0 ? a.q[i + 1][row[s]] : 0.0;
    }
    const double h = lat_allreduce(acc, a.slots, ++seq, lds, false);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (rotate) hcol[i] = h;  // (written back rotated, below)
      else a.H[(int64_t)i * a.m + a.k] = h;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) w[s] -= h * qc[s], qc[s] = qn[s];
  }
  double acc = 0.0;
#pragma unroll
  for (int s = 0; s < S; ++s) acc += w[s] * w[s];
  const double norm2 = lat_allreduce(acc, a.slots, ++seq, lds, false);
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.norm2_out = norm2;
  const double hn = sqrt(norm2);
#pragma unroll
  for (int s = 0; s < S; ++s)
    if (row[s] >= 0) a.w[row[s]] = a.normalise ? w[s] / hn : w[s];
  // SolverGmres.hpp:161, :176-191 and Solver.hpp:132-140 by the thread that holds column k of the Hessenberg (the
  // arithmetic of gmres_givens_update, solver_device.hpp, on the LDS copies; cs_sh / sn_sh were filled before the
  // block's first barrier)
  if (rotate && threadIdx.x == 0) {
    const int k = a.k, m = a.m;
    *a.givens.hn_slot = hn;
    hcol[k + 1] = hn;
    for (int i = 0; i < k; ++i) {
      const double chi = cs_sh[i] * hcol[i] + sn_sh[i] * hcol[i + 1];
      hcol[i + 1] = -sn_sh[i] * hcol[i] + cs_sh[i] * hcol[i + 1];
      hcol[i] = chi;
    }
    const double ha = hcol[k], hb = hcol[k + 1];
    const double rr = hypot(ha, hb);
    double cs, sn;
    if (rr > 0.0) cs = ha / rr, sn = hb / rr;
    else cs = 1.0, sn = 0.0;
    a.givens.cs[k] = cs, a.givens.sn[k] = sn;
    hcol[k] = cs * ha + sn * hb;
    hcol[k + 1] = 0.0;
    for (int i = 0; i <= k + 1; ++i) a.givens.H[(int64_t)i * m + k] = hcol[i];
    const double bk = a.givens.beta[k];
    a.givens.beta[k + 1] = -sn * bk;
    a.givens.beta[k] = bk * cs;
    advance(a.givens.st, fabs(-sn * bk));
  }
}

// The three-value all-reduce for a kernel with LDS-DMA in flight: __syncthreads() carries a fence that waits for every
// outstanding vector-memory operation of the wave -- the DMAs included -- so the block's barriers here are bare
// `s_barrier`s behind `s_waitcnt lgkmcnt(0)` (the LDS writes they order), and only the POLLING waves, which the
// caller keeps free of DMAs until they are through, wait on the vector-memory counter.
__device__ __forceinline__ void raw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void lat_allreduce3_raw(double &s0, double &s1, double &s2, char *slots, unsigned long long seq,
                                                   double *lds /* [3 * kLatWaves] */) {
  const unsigned tag = (unsigned)seq;
  double v0 = lat_wave_sum(s0), v1 = lat_wave_sum(s1), v2 = lat_wave_sum(s2);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  raw_barrier();
  if (lane == 0) lds[wave] = v0, lds[kLatWaves + wave] = v1, lds[2 * kLatWaves + wave] = v2;
  raw_barrier();
  if (threadIdx.x < 3) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kLatWaves; ++w) t += lds[threadIdx.x * kLatWaves + w];
    co_store_slot(slots + lat_slot_offset(blockIdx.x, seq) + 16 * threadIdx.x, tag, t);
  }
  v0 = v1 = v2 = 0.0;
  if (threadIdx.x < gridDim.x) {
    const char *slot = slots + lat_slot_offset(threadIdx.x, seq);
    int *gave_up = reinterpret_cast<int *>(slots + (size_t)2 * 256 * kLatSlotStride);
    const long long t0 = wall_clock64();
    for (int spins = 0;; ++spins) {
      if (co_load_slot3(slot, tag, &v0, &v1, &v2)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 1023) == 1023 &&
          (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v0 = v1 = v2 = 0.0;
        break;
      }
    }
  }
  v0 = lat_wave_sum(v0), v1 = lat_wave_sum(v1), v2 = lat_wave_sum(v2);
  raw_barrier();
  if (lane == 0 && wave < 4) lds[wave] = v0, lds[kLatWaves + wave] = v1, lds[2 * kLatWaves + wave] = v2;
  raw_barrier();
  s0 = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  s1 = (lds[kLatWaves] + lds[kLatWaves + 1]) + (lds[kLatWaves + 2] + lds[kLatWaves + 3]);
  s2 = (lds[2 * kLatWaves] + lds[2 * kLatWaves + 1]) + (lds[2 * kLatWaves + 2] + lds[2 * kLatWaves + 3]);
}

// ---- ... with the basis vectors landing in LDS (round 4) ----------------------------------------------------------
// The chain above is bound by what it can keep in flight: w and the current pair of basis vectors fill the registers, so
// the next vectors' rows are requested only when a register array is free again and the HBM stream stops at every
// all-reduce (GMRES(30) at 128^3: 2.8 TB/s over the chain).  Here the NEXT pair of basis vectors is fetched by LDS-DMA
// (`global_load_lds_dwordx4`: no register destination) into a two-slot ring of the block's rows, 2 x SUB x 16 KiB,
// issued the moment the current pair has been read out of the ring: the stream runs through the reduction and the
// update of w.  A thread reads back exactly the 16 bytes its own DMA wrote (the ring is a per-thread landing zone, no
// barrier), behind `s_waitcnt vmcnt(0)`.  Rows of a block: [blockIdx * SUB * 2048, ...), pair 2 t + j * 2048 of thread t.
// Same steps, same values in the same order as the paired chain above; the block partials group the rows differently.
constexpr int kMgsSub = 2 * kLatBlock;  // rows per sub-chunk: one pair per thread
typedef double double2m __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst /* wave-uniform byte address */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
template <int SUB>
__global__ __launch_bounds__(kLatBlock) void mgs_chain_lds_kernel(MgsArgs a) {
  if (a.done && *a.done) return;  // (uniform: every block reads the same flag before any of them synchronises)
  extern __shared__ __attribute__((aligned(16))) double ring[];  // [2][SUB * kMgsSub]
  __shared__ double lds[3 * kLatWaves];
  __shared__ double hcol[kMgsMaxVectors + 1], cs_sh[kMgsMaxVectors], sn_sh[kMgsMaxVectors];
  const bool rotate = a.givens.st != nullptr && blockIdx.x == 0;
  if (rotate && (int)threadIdx.x < a.k) cs_sh[threadIdx.x] = a.givens.cs[threadIdx.x], sn_sh[threadIdx.x] = a.givens.sn[threadIdx.x];
  const int tid = threadIdx.x, wave = tid >> 6;
  unsigned long long seq = a.seq_base;
  const int64_t chunk0 = (int64_t)blockIdx.x * SUB * kMgsSub;
  int64_t row[SUB];
  bool va[SUB], vb[SUB];
  double2m w[SUB];
#pragma unroll
  for (int j = 0; j < SUB; ++j) {
    row[j] = chunk0 + (int64_t)j * kMgsSub + 2 * tid;
    va[j] = row[j] < a.n_rows, vb[j] = row[j] + 1 < a.n_rows;
    w[j] = double2m{0.0, 0.0};
    if (vb[j]) w[j] = *reinterpret_cast<const double2m *>(a.w + row[j]);
    else if (va[j]) w[j].x = a.w[row[j]];
  }
  const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)ring;  // the ring's LDS byte address
  // this block's rows of basis vector q into a slot of the ring (rows past the end: any valid address, masked below)
  auto issue = [&](int slot, const double *q) {
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(ring_base + (unsigned)(((slot * SUB + j) * kMgsSub + wave * 2 * kWave) * 8));
      glds16(q + (va[j] ? row[j] : 0), dst);
    }
  };
  auto take = [&](int slot, double2m (&v)[SUB]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's DMAs have landed
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
      const double2m t = *reinterpret_cast<const double2m *>(&ring[(slot * SUB + j) * kMgsSub + 2 * tid]);
      v[j].x = va[j] ? t.x : 0.0, v[j].y = vb[j] ? t.y : 0.0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and are in registers: the slot may be refilled
  };
  long long tick[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = a.prof ? wall_clock64() : 0;
  auto lap = [&](int p) {
    if (a.prof) {
      const long long now = wall_clock64();
      tick[p] += now - t_mark, t_mark = now;
    }
  };
  issue(0, a.q[0]);
  if (a.k >= 1) issue(1, a.q[1]);
  double2m qa[SUB], qb[SUB];
  int i = 0;
  for (; i + 1 <= a.k; i += 2) {
    lap(0);  // (the update of w, loop overhead)
    take(0, qa), take(1, qb);
    lap(1);  // waiting for the pair's rows
    // the next pair travels under the reduction and the update -- but for the waves that poll the other blocks' slots
    // (vector loads return in order: a poll behind a DMA would wait for it), which ask for theirs once they are through
    const bool polls = wave < 4;
    if (!polls) {
      if (i + 2 <= a.k) issue(0, a.q[i + 2]);
      if (i + 3 <= a.k) issue(1, a.q[i + 3]);
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
      s0 += w[j].x * qa[j].x, s1 += w[j].x * qb[j].x, s2 += qa[j].x * qb[j].x;
      s0 += w[j].y * qa[j].y, s1 += w[j].y * qb[j].y, s2 += qa[j].y * qb[j].y;
    }
    lap(2);  // DMA issue + dot products
    lat_allreduce3_raw(s0, s1, s2, a.slots, ++seq, lds);
    lap(3);  // the all-reduce
    if (polls) {
      if (i + 2 <= a.k) issue(0, a.q[i + 2]);
      if (i + 3 <= a.k) issue(1, a.q[i + 3]);
    }
    const double h0 = s0, h1 = s1 - h0 * s2;  // (mgs_chain_kernel: the reference's h_{i+1} by bilinearity)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (rotate) hcol[i] = h0, hcol[i + 1] = h1;
      else a.H[(int64_t)i * a.m + a.k] = h0, a.H[(int64_t)(i + 1) * a.m + a.k] = h1;
    }
#pragma unroll
    for (int j = 0; j < SUB; ++j) {
      w[j].x -= h0 * qa[j].x, w[j].y -= h0 * qa[j].y;
      w[j].x -= h1 * qb[j].x, w[j].y -= h1 * qb[j].y;
    }
  }
  if (i <= a.k) {  // an odd vector is left: it sits in slot 0
    take(0, qa);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < SUB; ++j) acc += w[j].x * qa[j].x, acc += w[j].y * qa[j].y;
    const double h = lat_allreduce(acc, a.slots, ++seq, lds, false);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (rotate) hcol[i] = h;
      else a.H[(int64_t)i * a.m + a.k] = h;
    }
#pragma unroll
    for (int j = 0; j < SUB; ++j) w[j].x -= h * qa[j].x, w[j].y -= h * qa[j].y;
  }
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < SUB; ++j) acc += w[j].x * w[j].x, acc += w[j].y * w[j].y;
  const double norm2 = lat_allreduce(acc, a.slots, ++seq, lds, false);
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.norm2_out = norm2;
  const double hn = sqrt(norm2);
#pragma unroll
  for (int j = 0; j < SUB; ++j) {
    const double2m o = a.normalise ? double2m{w[j].x / hn, w[j].y / hn} : w[j];
    if (vb[j]) *reinterpret_cast<double2m *>(a.w + row[j]) = o;
    else if (va[j]) a.w[row[j]] = o.x;
  }
  lap(4);  // the tail: odd vector, norm, store
  if (a.prof && threadIdx.x == 0)
    for (int p = 0; p < 8; ++p) a.prof[blockIdx.x * 8 + p] = tick[p];
  // SolverGmres.hpp:161, :176-191 and Solver.hpp:132-140, as in mgs_chain_kernel
  if (rotate && threadIdx.x == 0) {
    const int k = a.k, m = a.m;
    *a.givens.hn_slot = hn;
    hcol[k + 1] = hn;
    for (int t = 0; t < k; ++t) {
      const double chi = cs_sh[t] * hcol[t] + sn_sh[t] * hcol[t + 1];
      hcol[t + 1] = -sn_sh[t] * hcol[t] + cs_sh[t] * hcol[t + 1];
      hcol[t] = chi;
    }
    const double ha = hcol[k], hb = hcol[k + 1];
    const double rr = hypot(ha, hb);
    double cs, sn;
    if (rr > 0.0) cs = ha / rr, sn = hb / rr;
    else cs = 1.0, sn = 0.0;
    a.givens.cs[k] = cs, a.givens.sn[k] = sn;
    hcol[k] = cs * ha + sn * hb;
    hcol[k + 1] = 0.0;
    for (int t = 0; t <= k + 1; ++t) a.givens.H[(int64_t)t * m + k] = hcol[t];
    const double bk = a.givens.beta[k];
    a.givens.beta[k + 1] = -sn * bk;
    a.givens.beta[k] = bk * cs;
    advance(a.givens.st, fabs(-sn * bk));
  }
}

// ---- ... FOUR steps per synchronisation point (round 4) -----------------------------------------------------------
// Measured (option resident_profile, GMRES(30) at 128^3, the 30-vector chain): 134 of 167 us are the 15 all-reduces,
// 8.9 us each -- three times what the same all-reduce costs the resident CG kernel, because a poll queues behind the
// 33 MB of basis-vector rows the chain has just asked for (the ring above does not change that: the requests are FIFO).
// A chain is therefore (its bytes at the HBM rate) + (its synchronisation points x ~4.7 us), and what is left to take
// are the synchronisation points: FOUR Gram-Schmidt steps share one.  By bilinearity (as for the pairs above)
//   h_0 = <w, q_0>,   h_j = <w, q_j> - sum_{i < j} h_i <q_i, q_j>            (j = 1, 2, 3)
// are the reference's h_j = <w - h_0 q_0 - ... - h_{j-1} q_{j-1}, q_j> (SolverGmres.hpp:157-160); the ten dot products on
// the right need only the w before the group, and travel in one all-reduce.  Blocks of 512 threads (two wavefronts per
// SIMD: 256 registers per lane hold w and the group's four vectors of 2 S rows); the update w -= h_0 q_0; ... -= h_3 q_3
// runs in the reference's order.
constexpr int kQuadThreads = 512, kQuadWaves = kQuadThreads / kWave, kQuadSub = 2 * kQuadThreads;
constexpr int kQuadSlotStride = 256;  // ten values of 16 bytes
// LDSPF (S = 8, where no second set of vectors fits the registers): of the NEXT group's T vectors the first lands in
// registers and the others in LDS (LDS-DMA, 64 KiB per vector: `global_load_lds_dwordx4` has no register destination),
// all requested between the halves of the all-reduce; a thread reads back exactly the 16 bytes its own DMA wrote.
template <int S, int T, bool APPLY = false, bool LDSPF = false>  // T = 3 or 4 steps per synchronisation point
__global__ __launch_bounds__(kQuadThreads) void mgs_chain_quad_kernel(MgsArgs a) {
  if (a.done && *a.done) return;  // (uniform: every block reads the same flag before any of them synchronises)
  __shared__ double lds[10 * 256 + 16];  // co_allreduce_dense: NV x 256 polled values + the NV results
  __shared__ double dict_sh[32];
  extern __shared__ __attribute__((aligned(16))) double pf_ring[];  // LDSPF: [T - 1][S][kQuadSub] doubles
  __shared__ double hcol[kMgsMaxVectors + 1], cs_sh[kMgsMaxVectors], sn_sh[kMgsMaxVectors];
  const bool rotate = a.givens.st != nullptr && blockIdx.x == 0;
  if (rotate && (int)threadIdx.x < a.k) cs_sh[threadIdx.x] = a.givens.cs[threadIdx.x], sn_sh[threadIdx.x] = a.givens.sn[threadIdx.x];
  const int tid = threadIdx.x;
  unsigned long long seq = a.seq_base;
  int *gave_up = reinterpret_cast<int *>(a.slots + (size_t)2 * 256 * kLatSlotStride);  // (the latency path's flag)
  char *slots = a.quad_slots;
  // Which chunk of rows a block owns.  Blocks are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8); with the
  // apply in the kernel a chunk's +-b neighbours (the planes below and above: two chunks away at 128^3) are gathered from
  // rows that OTHER blocks load as their own -- given to blocks of the same XCD (one contiguous run of chunks per XCD)
  // those gathers meet the owner's load in that XCD's L2 instead of fetching the line a second and third time.  The
  // all-reduce slots stay indexed by blockIdx.x: the same sums in another, equally fixed order.
  const int64_t chunk0 = (int64_t)(a.xcd_runs ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * S * kQuadSub;
  unsigned off8[S];  // byte offset of the thread's pair j (rows < 2^22)
  bool va[S], vb[S];
  double2m w[S];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const int64_t row = chunk0 + (int64_t)j * kQuadSub + 2 * tid;
    va[j] = row < a.n_rows, vb[j] = row + 1 < a.n_rows;
    off8[j] = va[j] ? (unsigned)row << 3 : 0u;
    w[j] = double2m{0.0, 0.0};
    if (!APPLY && va[j]) w[j] = *reinterpret_cast<const double2m *>(reinterpret_cast<const char *>(a.w) + off8[j]);  // (>= 4 zero doubles behind the last row)
    w[j].y = vb[j] ? w[j].y : 0.0;
  }
  if constexpr (APPLY) {
    // w = beta x + alpha M(x) of the thread's row pairs: spmv_canon_kernel<false, 6, 2, false, G> (spmv_pair.hip) -- the
    // record word, the own pair, the four 16-byte gathers of offsets 0, 1, 4, 5, the +-1 neighbours from the adjacent
    // lanes (lanes 0 and 63 load theirs); two pairs' loads in flight at a time (registers)
    typedef unsigned long long u64x2m __attribute__((ext_vector_type(2)));
    const int lane = tid & (kWave - 1);
    if (lane < 32) dict_sh[lane] = a.ap_dict[lane];  // every wave stores the same words: no barrier (same-wave LDS order)
    __builtin_amdgcn_wave_barrier();
    const char *xb = reinterpret_cast<const char *>(a.ap_x);
    const char *xg_base = xb - (size_t)kVecGuard * 8;
    const double alpha = a.ap_alpha, beta = a.ap_beta;
    constexpr int JB = S >= 2 ? 2 : 1;  // (four pairs in flight at S = 8: 89.0 against 88.0 us per inner iteration at 128^3)
#pragma unroll
    for (int j0 = 0; j0 < S; j0 += JB) {
      u64x2m vw[JB];
      double2m xi[JB], xg[JB][6];
      double e[JB];
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        const unsigned rc = off8[j0 + jj] >> 3;  // (an absent pair re-reads pair 0: masked below)
        vw[jj] = __builtin_nontemporal_load(reinterpret_cast<const u64x2m *>(a.ap_pack + off8[j0 + jj]));
        xi[jj] = *reinterpret_cast<const double2m *>(xb + off8[j0 + jj]);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          if (k == 2 || k == 3) continue;
          int t = (int)rc + a.ap_off[k] + kVecGuard;  // guard-relative, clamped: an absent neighbour may point anywhere
          t = t < 0 ? 0 : t;
          t = t > a.ap_max_gather ? a.ap_max_gather : t;
          xg[jj][k] = *reinterpret_cast<const double2m *>(xg_base + (size_t)((unsigned)t << 3));
        }
        e[jj] = 0.0;
        if (lane == 0 || lane == kWave - 1)  // x[rc - 1] of lane 0, x[rc + 2] of lane 63
          e[jj] = *reinterpret_cast<const double *>(xg_base + (size_t)((rc + (unsigned)(kVecGuard + (lane == 0 ? -1 : 2))) << 3));
      }
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        const double left = dpp_shift<0x138>(xi[jj].y);   // wave_shr:1 -- lane i receives lane i - 1
        const double right = dpp_shift<0x130>(xi[jj].x);  // wave_shl:1 -- lane i receives lane i + 1
        xg[jj][2].x = lane == 0 ? e[jj] : left;
        xg[jj][2].y = xi[jj].x;
        xg[jj][3].x = xi[jj].y;
        xg[jj][3].y = lane == kWave - 1 ? e[jj] : right;
        double acc_a = 0.0, acc_b = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const unsigned ba = (unsigned)(vw[jj].x >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(vw[jj].y >> (8 * (k + 1))) & 0xffu;
          acc_a += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ba) * (xg[jj][k].x - xi[jj].x);
          acc_b += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + bb) * (xg[jj][k].y - xi[jj].y);
        }
        const double ext_a = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[jj].x & 0xffu));
        const double ext_b = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[jj].y & 0xffu));
        double2m yi;
        yi.x = __builtin_fma(alpha, __builtin_fma(ext_a, xi[jj].x, acc_a), beta * xi[jj].x);  // (spmv_canon_tile_kernel's form)
        yi.y = __builtin_fma(alpha, __builtin_fma(ext_b, xi[jj].y, acc_b), beta * xi[jj].y);
        w[j0 + jj].x = va[j0 + jj] ? yi.x : 0.0;
        w[j0 + jj].y = vb[j0 + jj] ? yi.y : 0.0;
        asm volatile("" : "+v"(w[j0 + jj].x), "+v"(w[j0 + jj].y));  // (the pair is finished HERE: nothing of it stays live)
      }
      // (group after group: with all S pairs' loads hoisted to the front the S = 8 kernel spilled 207 registers)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  constexpr int ND = T == 4 ? 10 : 6;
  long long tick[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = a.prof ? wall_clock64() : 0;
  auto lap = [&](int p) {
    if (a.prof) {
      const long long now = wall_clock64();
      tick[p] += now - t_mark, t_mark = now;
    }
  };
  // Where the registers allow it (S <= 4) the NEXT group's vectors are requested between the block's arrival at the
  // all-reduce and its wait for the others (co_allreduce_dense_arrive / _wait): their latency hides in the wait.
  constexpr bool kPrefetch = S <= 4;
  double2m qn[kPrefetch ? T : 1][(kPrefetch || LDSPF) ? S : 1];
  bool prefetched = false;
  const unsigned pf_base = LDSPF ? (unsigned)(size_t)(__attribute__((address_space(3))) void *)pf_ring : 0u;
  const int pf_wave = tid >> 6;
  // The ORDER in which w is orthogonalised against q_0 .. q_k alternates with k (round 6): ascending for even k, descending
  // for odd k.  A cycle's basis outgrows the 256 MB Infinity Cache from k = 15 on at 128^3 (16.8 MB per vector); read in the
  // same order every time, each vector has been evicted by the time it comes round again -- every read an HBM read.  Read
  // back and forth, an iteration starts with the vectors the previous one ended with: ~14 of them are still there.  The
  // reference's loop runs i = 0 .. k (SolverGmres.hpp:157-160); against an orthonormal basis the h_i of modified Gram-Schmidt
  // do not depend on the order but for their roundings (the fixed-K tests hold either order to 1e-10 / 1e-9), and the order
  // is a function of k alone: every run, every variant of this kernel takes the same one.
  const auto vidx = [&](int p) { return a.descend ? a.k - p : p; };  // position in the chain -> basis vector
  for (int i = 0; i <= a.k; i += T) {
    lap(0);  // the update of w
    double2m q[T][S];
    if (LDSPF && prefetched) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's DMAs (and register loads) have landed
#pragma unroll
      for (int j = 0; j < S; ++j) q[0][j] = qn[0][LDSPF ? j : 0];
#pragma unroll
      for (int v = 1; v < T; ++v) {
        const bool have = i + v <= a.k;
#pragma unroll
        for (int j = 0; j < S; ++j) {
          const double2m t = *reinterpret_cast<const double2m *>(&pf_ring[((v - 1) * S + j) * kQuadSub + 2 * tid]);
          q[v][j].x = (have && va[j]) ? t.x : 0.0, q[v][j].y = (have && vb[j]) ? t.y : 0.0;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and are in registers: the landing zone may be refilled
    } else if (kPrefetch && prefetched) {
#pragma unroll
      for (int v = 0; v < T; ++v)
#pragma unroll
        for (int j = 0; j < S; ++j) q[v][j] = qn[kPrefetch ? v : 0][kPrefetch ? j : 0];
    } else {
#pragma unroll
      for (int v = 0; v < T; ++v) {
        const bool have = i + v <= a.k;  // (uniform; a vector past the end reads as zeros: its h comes out 0)
        const char *src = reinterpret_cast<const char *>(a.q[vidx(have ? i + v : i)]);
#pragma unroll
        for (int j = 0; j < S; ++j) {
          q[v][j] = double2m{0.0, 0.0};
          if (have && va[j]) q[v][j] = *reinterpret_cast<const double2m *>(src + off8[j]);
          q[v][j].y = vb[j] ? q[v][j].y : 0.0;
        }
      }
    }
    prefetched = false;
    // T = 4: <w,q0..3>, <q0,q1>, <q0,q2>, <q0,q3>, <q1,q2>, <q1,q3>, <q2,q3>;  T = 3: <w,q0..2>, <q0,q1>, <q0,q2>, <q1,q2>
    double d[ND];
#pragma unroll
    for (int e = 0; e < ND; ++e) d[e] = 0.0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
#pragma unroll
      for (int v = 0; v < T; ++v) d[v] += w[j].x * q[v][j].x, d[v] += w[j].y * q[v][j].y;
      int e = T;
#pragma unroll
      for (int u = 0; u < T; ++u)
#pragma unroll
        for (int v = u + 1; v < T; ++v, ++e) d[e] += q[u][j].x * q[v][j].x, d[e] += q[u][j].y * q[v][j].y;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.prof) __syncthreads();  // (diagnostic: the whole block's rows have landed)
    lap(1);  // the group's rows (issue -> landed) and the dot products
    if (a.dense && (kPrefetch || LDSPF) && a.prefetch != 0) {
      ++seq;
      co_allreduce_dense_arrive<ND, kQuadWaves>(d, slots, seq, lds);
      if (i + T <= a.k) {
        if constexpr (LDSPF) {
          {  // the group's first vector: registers
            const char *src = reinterpret_cast<const char *>(a.q[vidx(i + T)]);
#pragma unroll
            for (int j = 0; j < S; ++j) {
              qn[0][j] = double2m{0.0, 0.0};
              if (va[j]) qn[0][j] = *reinterpret_cast<const double2m *>(src + off8[j]);
              qn[0][j].y = vb[j] ? qn[0][j].y : 0.0;
            }
          }
#pragma unroll
          for (int v = 1; v < T; ++v) {  // the others: LDS-DMA (rows past the end: any valid address, masked when read back)
            if (i + T + v <= a.k) {
              const char *src = reinterpret_cast<const char *>(a.q[vidx(i + T + v)]);
#pragma unroll
              for (int j = 0; j < S; ++j) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(pf_base + (unsigned)((((v - 1) * S + j) * kQuadSub + pf_wave * 2 * kWave) * 8));
                glds16(src + off8[j], dst);
              }
            }
          }
        }
        if constexpr (kPrefetch) {
#pragma unroll
          for (int v = 0; v < T; ++v) {
            const bool have = i + T + v <= a.k;
            const char *src = reinterpret_cast<const char *>(a.q[vidx(have ? i + T + v : i + T)]);
#pragma unroll
            for (int j = 0; j < S; ++j) {
              qn[v][j] = double2m{0.0, 0.0};
              if (have && va[j]) qn[v][j] = *reinterpret_cast<const double2m *>(src + off8[j]);
              qn[v][j].y = vb[j] ? qn[v][j].y : 0.0;
            }
          }
        }
        prefetched = true;
      }
      co_allreduce_dense_wait<ND, kQuadWaves>(d, slots, gave_up, seq, lds);
    } else if (a.dense) {
      co_allreduce_dense<ND, kQuadWaves>(d, slots, gave_up, ++seq, lds);
    } else {
      co_allreduce2_n<ND, kQuadWaves>(d, slots, kQuadSlotStride, gave_up, ++seq, lds);
    }
    lap(2);  // the all-reduce
    double h[T];
    {
      int e = T;  // h_v = <w, q_v> - sum_{u < v} h_u <q_u, q_v>, the pairs (u, v) in the order they were summed
      double g[T][T];
#pragma unroll
      for (int u = 0; u < T; ++u)
#pragma unroll
        for (int v = u + 1; v < T; ++v, ++e) g[u][v] = d[e];
#pragma unroll
      for (int v = 0; v < T; ++v) {
        h[v] = d[v];
#pragma unroll
        for (int u = 0; u < v; ++u) h[v] -= h[u] * g[u][v];
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      for (int v = 0; v < T && i + v <= a.k; ++v) {
        if (rotate) hcol[vidx(i + v)] = h[v];
        else a.H[(int64_t)vidx(i + v) * a.m + a.k] = h[v];
      }
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
#pragma unroll
      for (int v = 0; v < T; ++v) w[j].x -= h[v] * q[v][j].x, w[j].y -= h[v] * q[v][j].y;
    }
  }
  double acc[1] = {0.0};
#pragma unroll
  for (int j = 0; j < S; ++j) acc[0] += w[j].x * w[j].x, acc[0] += w[j].y * w[j].y;
  bool rotated = false;
  if (a.dense && a.rotate_early != 0) {
    // The k earlier rotations of column k (SolverGmres.hpp:176-180) need every h of the chain and not the norm: block 0's
    // first thread applies them between the block's arrival at the norm's all-reduce and its wait for the others -- ~1.5 us
    // of a dependent chain through LDS that used to run after everything else, with the whole chip waiting for the kernel
    // to end.
    ++seq;
    co_allreduce_dense_arrive<1, kQuadWaves>(acc, slots, seq, lds);
    if (rotate && threadIdx.x == 0) {
      for (int t = 0; t < a.k; ++t) {
        const double chi = cs_sh[t] * hcol[t] + sn_sh[t] * hcol[t + 1];
        hcol[t + 1] = -sn_sh[t] * hcol[t] + cs_sh[t] * hcol[t + 1];
        hcol[t] = chi;
      }
      rotated = true;
    }
    co_allreduce_dense_wait<1, kQuadWaves>(acc, slots, gave_up, seq, lds);
  } else if (a.dense) {
    co_allreduce_dense<1, kQuadWaves>(acc, slots, gave_up, ++seq, lds);
  } else {
    co_allreduce2_n<1, kQuadWaves>(acc, slots, kQuadSlotStride, gave_up, ++seq, lds);
  }
  const double norm2 = acc[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.norm2_out = norm2;
  const double hn = sqrt(norm2);
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const double2m o = a.normalise ? double2m{w[j].x / hn, w[j].y / hn} : w[j];
    if (vb[j]) *reinterpret_cast<double2m *>(reinterpret_cast<char *>(a.w) + off8[j]) = o;
    else if (va[j]) *reinterpret_cast<double *>(reinterpret_cast<char *>(a.w) + off8[j]) = o.x;
  }
  lap(3);  // the tail
  if (a.prof && threadIdx.x == 0)
    for (int p = 0; p < 8; ++p) a.prof[blockIdx.x * 8 + p] = tick[p];
  // SolverGmres.hpp:161, :176-191 and Solver.hpp:132-140, as in mgs_chain_kernel
  if (rotate && threadIdx.x == 0) {
    const int k = a.k, m = a.m;
    *a.givens.hn_slot = hn;
    hcol[k + 1] = hn;
    if (!rotated) {
      for (int t = 0; t < k; ++t) {
        const double chi = cs_sh[t] * hcol[t] + sn_sh[t] * hcol[t + 1];
        hcol[t + 1] = -sn_sh[t] * hcol[t] + cs_sh[t] * hcol[t + 1];
        hcol[t] = chi;
      }
    }
    const double ha = hcol[k], hb = hcol[k + 1];
    const double rr = hypot(ha, hb);
    double cs, sn;
    if (rr > 0.0) cs = ha / rr, sn = hb / rr;
    else cs = 1.0, sn = 0.0;
    a.givens.cs[k] = cs, a.givens.sn[k] = sn;
    hcol[k] = cs * ha + sn * hb;
    hcol[k + 1] = 0.0;
    for (int t = 0; t <= k + 1; ++t) a.givens.H[(int64_t)t * m + k] = hcol[t];
    const double bk = a.givens.beta[k];
    a.givens.beta[k + 1] = -sn * bk;
    a.givens.beta[k] = bk * cs;
    advance(a.givens.st, fabs(-sn * bk));
  }
}

// Returns STORM_HIP_OK with *taken = false when the chain does not qualify (too many rows / vectors, a communicator).
// apply (nullable): w has not been formed yet, w = beta x + alpha M(x) with x = q[k] (ChainApply, common.hpp): the quad
// kernels do it themselves, in front of any other variant it is launched here; *applied tells whether w exists when this
// returns -- if not (the chain did not qualify or could not be launched) the caller applies the operator itself.
int gmres_mgs_chain_coop(storm_hip_ctx *c, int64_t n, const int *done, double *w, const double *const *q, int k, int m,
                         double *H, double *norm2_out, bool normalise, bool *taken, const MgsGivens *givens,
                         const ChainApply *apply, bool *applied) {
  *taken = false;
  if (applied) *applied = false;
  bool with_apply = false;
  // (a step of the chain costs half an all-reduce, ~2.5 us, whatever the size; the kernel-per-step path costs a launch,
  //  ~3.5 us, or 32 B/row of HBM traffic, whichever is more -- measured, us per inner iteration, per-step vs chained:
  //  step.1 83..106 vs 74.5, 32^3 84..109 vs 72, 64^3 91..106 vs 93, 128^3 248 vs 147)
  if (c->opt_coop_mgs == 0 || c->coop_disabled != 0 || c->comm != nullptr || n < c->opt_coop_mgs_min_rows ||
      k + 1 > kMgsMaxVectors || c->opt_profile_spmv != 0)
    return STORM_HIP_OK;
  const int64_t n_slices = (n + kWave - 1) / kWave;
  int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(std::min(c->num_cus, 256), (n_slices + kLatWaves - 1) / kLatWaves));
  const int64_t waves = blocks * kLatWaves;
  const int64_t need = (n_slices + waves - 1) / waves;
  const void *fn = nullptr;
  size_t dyn_lds = 0;
  // the LDS-ring chain: two steps per synchronisation point, <= 4 sub-chunks of 2048 rows per block (2 x 64 KiB of ring)
  const int cus = std::min(c->num_cus, 256);
  const int64_t subs_total = (n + kMgsSub - 1) / kMgsSub;
  const int sub = (int)((subs_total + cus - 1) / cus);
  // (measured, GMRES(30) us per inner iteration, register pairs / LDS ring / triples with the two-level all-reduce:
  //  32^3 43.0 / 43.5 / 48.3, 64^3 60.5 / 51.1 / 55.2, 128^3 104.5 / 106.3 / 100.0 -- profiles/r04p_gmres_chain_ab.jsonl; with the
  //  dense flat all-reduce the triples / quadruples take 38.9 / 52.6 / 97.2 and are the default wherever they fit;
  //  options coop_mgs_quad: 0 off, else on; coop_mgs_lds: 0 never, 1 where the quadruples are off, 2 always)
  if ((c->opt_coop_mgs_lds == 2 || (c->opt_coop_mgs_lds == 1 && c->opt_coop_mgs_quad == 0 && n >= ((int64_t)1 << 17))) && c->opt_coop_mgs_pairs != 0 &&
      sub >= 1 && sub <= 4) {
    const int sv = sub <= 1 ? 1 : sub <= 2 ? 2 : 4;
    fn = sv == 1 ? (const void *)mgs_chain_lds_kernel<1> : sv == 2 ? (const void *)mgs_chain_lds_kernel<2> : (const void *)mgs_chain_lds_kernel<4>;
    dyn_lds = sizeof(double) * 2 * (size_t)sv * kMgsSub;
    const int res = occupancy_cached(c, fn, kLatBlock, dyn_lds);
    if (res >= 1) blocks = (subs_total + sv - 1) / sv;
    else fn = nullptr, dyn_lds = 0;
  }
  // four steps per synchronisation point (blocks of 512 threads, <= 8 pairs of rows per thread): the default
  unsigned threads = kLatBlock;
  const int64_t qsubs_total = (n + kQuadSub - 1) / kQuadSub;
  const int qsub = (int)((qsubs_total + cus - 1) / cus);
  if (c->opt_coop_mgs_quad != 0 && c->opt_coop_mgs_pairs != 0 && qsub >= 1 && qsub <= 8) {
    const int sv = qsub <= 1 ? 1 : qsub <= 2 ? 2 : qsub <= 4 ? 4 : 8;
    // (eight or sixteen rows per thread and FOUR vectors of them do not fit 256 registers beside the all-reduce: three there)
    // the operator applied inside the kernel: format-4 records with the six common offsets of a 3-D lattice numbering,
    // the dictionary within 32 values, no halo, no tail -- and the caller wanting the newest basis vector applied to
    const storm_hip_op *aop = apply ? apply->op : nullptr;
    with_apply = aop != nullptr && applied != nullptr && c->opt_coop_mgs_apply != 0 && aop->pair == 2 && aop->canon_k == 6 &&
                 aop->canon_m1 == 2 && aop->n_halo == 0 && aop->tail_rows == 0 && aop->d_bnd_pack == nullptr &&
                 aop->dict_size <= 32 && aop->n_rows == n && apply->x == q[k];
    const void *qf = sv == 1 ? (const void *)mgs_chain_quad_kernel<1, 4> : sv == 2 ? (const void *)mgs_chain_quad_kernel<2, 4>
                   : sv == 4 ? (const void *)mgs_chain_quad_kernel<4, 3> : (const void *)mgs_chain_quad_kernel<8, 3>;
    if (with_apply)
      qf = sv == 1 ? (const void *)mgs_chain_quad_kernel<1, 4, true> : sv == 2 ? (const void *)mgs_chain_quad_kernel<2, 4, true>
         : sv == 4 ? (const void *)mgs_chain_quad_kernel<4, 3, true> : (const void *)mgs_chain_quad_kernel<8, 3, true>;
    // eight row pairs per thread: the next group's vectors through LDS (mgs_chain_quad_kernel<8, 3, APPLY, true>)
    size_t quad_lds = 0;
    // (with the apply only: the kernel that reads w instead has no registers left for the first vector -- 65 spills)
    const bool lds_pf = sv == 8 && with_apply && c->opt_coop_dense != 0 && c->opt_coop_mgs_prefetch != 0 && c->opt_coop_mgs_lds_prefetch != 0;
    if (lds_pf) {
      qf = (const void *)mgs_chain_quad_kernel<8, 3, true, true>;
      quad_lds = sizeof(double) * 2 * 8 * (size_t)kQuadSub;
    }
    const int res = occupancy_cached(c, qf, kQuadThreads, quad_lds);
    if (res >= 1) {
      if (c->d_quad_slots == nullptr) {
        const size_t bytes = std::max((size_t)2 * (256 + 8) * kQuadSlotStride, (size_t)2 * kDenseMaxValues * 256 * 16);  // either form
        HIP_TRY(hipMalloc((void **)&c->d_quad_slots, bytes));
        HIP_TRY(hipMemsetAsync(c->d_quad_slots, 0, bytes, c->stream));
      }
      fn = qf, dyn_lds = quad_lds, threads = kQuadThreads, blocks = (qsubs_total + sv - 1) / sv;
    } else {
      with_apply = false;
    }
  } else {
    with_apply = false;
  }
  const bool lds_chain = fn != nullptr;
  if (!lds_chain)
    fn = need <= 1    ? (const void *)mgs_chain_kernel<1>
                   : need <= 2  ? (const void *)mgs_chain_kernel<2>
                   : need <= 4  ? (const void *)mgs_chain_kernel<4>
                   : need <= 8  ? (const void *)mgs_chain_kernel<8>
                   : need <= 16 ? (const void *)mgs_chain_kernel<16>
                                : nullptr;
  if (fn == nullptr) return STORM_HIP_OK;  // more than 16 slices per wavefront: registers cannot hold w
  if (!lds_chain) {
  if (occupancy_cached(c, fn, kLatBlock, 0) < 1) return STORM_HIP_OK;
  }
  MgsArgs a;
  for (int i = 0; i <= k; ++i) a.q[i] = q[i];
  for (int i = k + 1; i < kMgsMaxVectors; ++i) a.q[i] = q[0];
  a.w = w, a.H = H, a.norm2_out = norm2_out, a.n_rows = n, a.n_slices = n_slices, a.k = k, a.m = m;
  a.normalise = normalise ? 1 : 0;
  a.pairs = (c->opt_coop_mgs_pairs != 0 && need <= 8) ? 1 : 0;  // (16 slices per wave + a third basis vector: spills)
  a.seq_base = (1ull << 31) | (c->lat_seq & 0x7fffffffull);  // bit 31: never the tag of a CG solve (those count from 1)
  c->lat_seq += (unsigned long long)k + 2;
  a.slots = c->d_lat_slots, a.done = done;
  a.givens = (givens != nullptr && normalise && c->opt_coop_mgs != 2) ? *givens  // (coop_mgs = 2: A/B, rotations by the caller)
                                                                        : MgsGivens{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a.quad_slots = c->d_quad_slots;
  a.dense = (int)(c->opt_coop_dense != 0);
  a.prefetch = (int)(c->opt_coop_mgs_prefetch != 0);
  a.xcd_runs = (int)(c->opt_coop_mgs_xcd_runs != 0);
  a.descend = (int)(c->opt_coop_mgs_alternate != 0 && (k & 1) != 0);
  a.rotate_early = (int)(c->opt_coop_mgs_rotate_early != 0);
  a.ap_pack = nullptr, a.ap_dict = nullptr, a.ap_x = nullptr, a.ap_max_gather = 0, a.ap_alpha = 0.0, a.ap_beta = 0.0;
  for (int i = 0; i < 6; ++i) a.ap_off[i] = 0;
  if (with_apply) {
    const storm_hip_op *aop = apply->op;
    a.ap_pack = aop->d_pack, a.ap_dict = aop->d_dict, a.ap_x = apply->x;
    for (int i = 0; i < 6; ++i) a.ap_off[i] = aop->canon_off[i];
    a.ap_max_gather = (int)(aop->n_rows + aop->n_halo) + kVecGuard + 2;
    a.ap_alpha = apply->alpha, a.ap_beta = apply->beta;
  }
  a.prof = nullptr;
  if (c->opt_resident_profile != 0 && lds_chain && k == m - 1) {  // (diagnostic: the longest chain of a cycle)
    if (c->d_res_prof == nullptr) HIP_TRY(hipMalloc((void **)&c->d_res_prof, sizeof(long long) * 256 * 8));
    a.prof = c->d_res_prof, c->res_prof_blocks = (int)blocks;
  }
  bool formed = false;
  if (apply != nullptr && applied != nullptr && !with_apply) {  // a chain variant that READS w: the apply goes first
    STORM_TRY(spmv_launch(apply->op, host_scal(apply->alpha), host_scal(apply->beta), apply->x, w, nullptr, done));
    formed = true;
  }
  void *args[] = {&a};
  *taken = coop_launch(c, fn, (unsigned)blocks, args, dyn_lds, threads);
  if (!*taken) c->lat_seq -= (unsigned long long)k + 2;
  if (applied) *applied = formed || (*taken && with_apply);
  return STORM_HIP_OK;
}

// Compact fp64 copy of an operator for the latency path (called by build_op); absent when the operator is too
// large, partitioned, or has rows longer than its ELL cap.
int op_make_latency_copy(storm_hip_op *op, int64_t n, int64_t n_halo, const std::vector<int64_t> &row_ptr,
                         const std::vector<int> &col, const std::vector<double> &val, const std::vector<double> &ext) {
  storm_hip_ctx *c = op->ctx;
  if (c->opt_latency_path == 0 || n_halo != 0 || n <= 0 || n > c->opt_latency_rows) return STORM_HIP_OK;
  const int64_t n_slices = (n + kWave - 1) / kWave;
  std::vector<int64_t> off((size_t)n_slices + 1, 0);
  for (int64_t s = 0; s < n_slices; ++s) {
    int64_t w = 0;
    for (int64_t r = s * kWave; r < std::min(n, (s + 1) * kWave); ++r) w = std::max(w, row_ptr[r + 1] - row_ptr[r]);
    if (w > 64) return STORM_HIP_OK;  // a very long row: the throughput path's CSR tail handles those
    off[(size_t)s + 1] = off[(size_t)s] + kWave * 8 + w * (kWave * 12);
  }
  std::vector<char> pack((size_t)off[(size_t)n_slices], 0);
  for (int64_t s = 0; s < n_slices; ++s) {
    const int width = (int)((off[(size_t)s + 1] - off[(size_t)s] - kWave * 8) / (kWave * 12));
    char *rec = pack.data() + off[(size_t)s];
    double *e_ = reinterpret_cast<double *>(rec);
    int *c_ = reinterpret_cast<int *>(rec + kWave * 8);
    double *v_ = reinterpret_cast<double *>(rec + kWave * 8 + (int64_t)width * (kWave * 4));
    for (int l = 0; l < kWave; ++l) {
      const int64_t r = s * kWave + l;
      e_[l] = r < n ? ext[(size_t)r] : 0.0;
      const int64_t b0 = r < n ? row_ptr[r] : 0, e0 = r < n ? row_ptr[r + 1] : 0;
      for (int k = 0; k < width; ++k) {
        const bool real = b0 + k < e0;
        c_[k * kWave + l] = real ? col[(size_t)(b0 + k)] : (int)(r < n ? r : n - 1);
        v_[k * kWave + l] = real ? val[(size_t)(b0 + k)] : 0.0;
      }
    }
  }
  HIP_TRY(hipMalloc((void **)&op->d_lat_pack, pack.size() ? pack.size() : 1));
  HIP_TRY(hipMalloc((void **)&op->d_lat_off, sizeof(int64_t) * off.size()));
  HIP_TRY(hipMemcpy(op->d_lat_pack, pack.data(), pack.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(op->d_lat_off, off.data(), sizeof(int64_t) * off.size(), hipMemcpyHostToDevice));
  op->lat_bytes = (int64_t)pack.size();
  return STORM_HIP_OK;
}

// After a cooperative kernel has completed: did one of its waits give up?
int lat_check_gave_up(storm_hip_ctx *c) {
  if (!c->coop_ran) return STORM_HIP_OK;  // (no cooperative kernel since the last look: nothing to read back)
  int flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, c->d_lat_slots + (size_t)2 * 256 * kLatSlotStride, sizeof flag, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->opt_coop_force_fail == 2 && c->coop_ran && !c->coop_disabled) flag = 1;  // (test hook)
  c->coop_ran = 0;
  if (flag != 0) {
    (void)hipMemsetAsync(c->d_lat_slots + (size_t)2 * 256 * kLatSlotStride, 0, sizeof flag, c->stream);
    // The resident kernels carry their all-reduce / exchange sequence numbers from solve to solve in d_res_slots and
    // trust every block to leave with the same pair.  After a give-up that no longer holds (blocks left at different
    // checks, some never started): a later solve could find a slot or granule of the aborted one already at "its" number.
    // The stream is idle here: drop both buffers, the next resident solve allocates them zero-filled and restarts at 0.
    if (c->d_res_slots) (void)hipFree(c->d_res_slots);
    if (c->d_res_exch) (void)hipFree(c->d_res_exch);
    c->d_res_slots = nullptr, c->d_res_exch = nullptr, c->res_exch_rows = 0;
    set_error("cooperative kernel: a block waited 10 s for the others (is the device shared with another process's "
              "cooperative kernel?)");
    return kStatusCoopGaveUp;  // coop_solve_with_fallback re-runs the solve on the kernel-per-statement path
  }
  return STORM_HIP_OK;
}

// A cooperative launch that may be refused (too many blocks for what is resident, a device that does not take them):
// false = not launched, nothing ran, the error is cleared.
static bool coop_launch(storm_hip_ctx *c, const void *fn, unsigned blocks, void **args, size_t dyn_lds, unsigned threads) {
  if (c->opt_coop_force_fail == 1) {
    c->coop_fallback = 1;
    return false;
  }
  // coop_plain: an ordinary launch of the same kernel.  These kernels synchronise through memory (no grid.sync()); what
  // they need is every block resident, which the callers size the grid for (<= one block per CU, a variant that fits) and
  // which holds on a device this process has to itself once the kernel in front has drained -- the runtime's cooperative
  // launch adds no more than that check, but runs on a queue of its own: 12-13 us of idle device in front of the kernel
  // AND in front of the next ordinary one (kernel trace, GMRES(30) at 128^3: two such gaps per inner iteration of 160 us).
  const hipError_t e = c->opt_coop_plain != 0 ? hipLaunchKernel(fn, dim3(blocks), dim3(threads), args, dyn_lds, c->stream)
                                              : hipLaunchCooperativeKernel(fn, dim3(blocks), dim3(threads), args, (unsigned)dyn_lds, c->stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    c->coop_fallback = 1;
    return false;
  }
  c->coop_ran = 1;
  return true;
}

int coop_solve_with_fallback(storm_hip_ctx *c, storm_hip_vec *x, int (*run)(void *), void *arg, int *fallback_out) {
  c->coop_fallback = 0, c->coop_ran = 0;
  // A cooperative kernel of an earlier solve gave up for real (a grid that did not become resident: a device shared
  // with another tenant, a CU mask): the next solves run without them instead of paying the bounded wait again --
  // 16 solves after the first give-up, twice as many after every further one.
  const bool backing_off = c->coop_skip > 0;
  if (backing_off) --c->coop_skip, c->coop_disabled = 1;
  const int64_t n_total = x->n_owned + x->n_halo;
  storm_hip_vec *x0 = nullptr;  // the start vector, kept for the re-run (pooled storage: no allocation, no stream wait per solve)
  const bool keep = c->comm == nullptr && c->coop_disabled == 0 &&
                    (c->opt_latency_path != 0 || c->opt_coop_mgs != 0 || c->opt_resident_path != 0) &&
                    n_total > 0 && n_total <= ((int64_t)1 << 23);  // (no cooperative kernel takes more rows than that)
  if (keep) {
    STORM_TRY(vec_create_work_batch(x, 1, &x0));
    const hipError_t e = hipMemcpyAsync(x0->d, x->d, sizeof(double) * (size_t)n_total, hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) {
      (void)storm_hip_vec_destroy(x0);
      HIP_TRY(e);
    }
  }
  int st = run(arg);
  if (st == kStatusCoopGaveUp && keep) {
    (void)hipMemcpyAsync(x->d, x0->d, sizeof(double) * (size_t)n_total, hipMemcpyDeviceToDevice, c->stream);
    c->coop_disabled = 1;
    st = run(arg);
    c->coop_disabled = 0;
    c->coop_fallback = 2;
    if (c->opt_coop_force_fail != 2) {  // (the test hook gives up once per solve: no back-off)
      c->coop_backoff = c->coop_backoff == 0 ? 16 : std::min<int64_t>(2 * c->coop_backoff, (int64_t)1 << 30);
      c->coop_skip = c->coop_backoff;
    }
  }
  if (backing_off) c->coop_disabled = 0;
  if (st == kStatusCoopGaveUp) st = STORM_HIP_E_HIP;  // (the message of lat_check_gave_up stands)
  if (x0) (void)storm_hip_vec_destroy(x0);
  if (fallback_out) *fallback_out = c->coop_fallback;
  return st;
}

bool cg_latency_eligible(const storm_hip_op *op) {
  const storm_hip_ctx *c = op->ctx;
  return c->opt_latency_path != 0 && c->coop_disabled == 0 && c->comm == nullptr && op->d_lat_pack != nullptr &&
         c->opt_profile_spmv == 0;
}

// The whole solve; fills the SolverState on the device (the caller reads it back).  `bicgstab`: which of the two
// kernels; work vectors p, r (CG) and p, r, v0, v1 (BiCGStab) arrive zero-filled.
static int latency_solve(bool bicgstab, const storm_hip_op *op, LatArgs a, bool *taken) {
  storm_hip_ctx *c = op->ctx;
  *taken = false;
  const int64_t n_slices = (op->n_rows + kWave - 1) / kWave;
  // A co-resident grid (cooperative launch): one 1024-thread block per CU at most (<= 256 blocks: one polling
  // thread per block), at least one slice per wavefront; the smallest register variant that covers all slices.
  // registers can hold the records of a wave's slices when rows have <= kLatCacheWidth slots and S <= 2
  const int w = c->opt_latency_cache == 0 ? 0 : op->max_row_len <= 4 ? 4 : op->max_row_len <= kLatCacheWidth ? 8 : 0;
  auto pick = [w](const void *w0, const void *w4, const void *w8) { return w == 4 ? w4 : w == 8 ? w8 : w0; };
  const void *cg[4] = {
      pick((const void *)cg_latency_kernel<1, 0>, (const void *)cg_latency_kernel<1, 4>, (const void *)cg_latency_kernel<1, 8>),
      pick((const void *)cg_latency_kernel<2, 0>, (const void *)cg_latency_kernel<2, 4>, (const void *)cg_latency_kernel<2, 8>),
      (const void *)cg_latency_kernel<4, 0>, (const void *)cg_latency_kernel<8, 0>};
  const void *bi[4] = {  // (two slices of records AND five vectors do not fit the registers)
      pick((const void *)bicgstab_latency_kernel<1, 0>, (const void *)bicgstab_latency_kernel<1, 4>,
           (const void *)bicgstab_latency_kernel<1, 8>),
      (const void *)bicgstab_latency_kernel<2, 0>, (const void *)bicgstab_latency_kernel<4, 0>, (const void *)bicgstab_latency_kernel<8, 0>};
  const void *const *variants = bicgstab ? bi : cg;
  const int capacity[4] = {1, 2, 4, 8};
  const void *fn = nullptr;
  int64_t blocks = 0;
  for (int v = 0; v < 4 && fn == nullptr; ++v) {
    if (occupancy_cached(c, variants[v], kLatBlock, 0) < 1) continue;
    blocks = std::max<int64_t>(1, std::min<int64_t>(std::min(c->num_cus, 256), (n_slices + kLatWaves - 1) / kLatWaves));
    const int64_t waves = blocks * kLatWaves;
    if ((n_slices + waves - 1) / waves <= capacity[v]) fn = variants[v];
  }
  if (fn == nullptr) {  // no register variant holds this many rows per wavefront (latency_rows raised, few CUs): the
    c->coop_fallback = 1;  // throughput path takes the solve
    return STORM_HIP_OK;
  }
  HIP_TRY(hipMemsetAsync(c->d_lat_slots, 0, (size_t)2 * 256 * kLatSlotStride + 256, c->stream));  // tags restart at 1; flag down
  a.pack = op->d_lat_pack, a.rec_off = op->d_lat_off, a.n_rows = op->n_rows, a.n_slices = n_slices, a.slots = c->d_lat_slots;
  a.publish_xchg = (int)(c->opt_latency_publish != 0);
  void *args[] = {&a};
  *taken = coop_launch(c, fn, (unsigned)blocks, args);
  return STORM_HIP_OK;
}

int cg_latency_solve(const storm_hip_op *op, double alpha, double beta, const double *b, double *x, double *p,
                     double *r, SolverState *d_state, bool *taken) {
  LatArgs a{};
  a.alpha = alpha, a.beta = beta, a.b = b, a.x = x, a.p = p, a.r = r, a.st = d_state;
  return latency_solve(false, op, a, taken);
}

int bicgstab_latency_solve(const storm_hip_op *op, double alpha, double beta, const double *b, double *x,
                           double *const work[4], SolverState *d_state, bool *taken) {
  LatArgs a{};
  a.alpha = alpha, a.beta = beta, a.b = b, a.x = x, a.p = work[0], a.r = work[1], a.v0 = work[2], a.v1 = work[3], a.st = d_state;
  return latency_solve(true, op, a, taken);
}

}  // namespace storm
