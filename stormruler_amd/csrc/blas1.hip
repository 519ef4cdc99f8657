// BLAS-1 kernels: the element loops and reductions the Krylov bodies run on
// (reference: Bittern/MatrixAlgorithms.hpp:58-81 matrix_for_each, :162-205 reduce).
//
// All kernels are HBM-bound streams.  Shape (measured, tools/stream_bench.hip): one trip per
// thread, 4 independent 16-byte accesses per stream in flight per lane, non-temporal loads and
// stores, grid = ceil(n / 2048) blocks.  Reductions fold per-lane partials with a 64-wide
// __shfl_down tree, then across the 4 waves of a block through LDS, then a fixed-order final
// pass over the per-block partials (two passes when there are more than 4096 of them).  The
// summation tree depends only on n, so results are bitwise reproducible run to run.
#include <cstring>

#include "common.hpp"
#include "blas1_device.hpp"
#include "ticket_device.hpp"

namespace storm {


__device__ __forceinline__ double ld_scal(const Scal &s) { return s.p ? (*s.p) * s.sign : s.v; }

// ---- elementwise ---------------------------------------------------------------------
struct EwPtrs {
  double *y;
  const double *x0, *x1;
};

template <class F>
__global__ __launch_bounds__(kBlock) void ew_kernel(int64_t n, EwPtrs p, F f, const int *done, int nt, int reverse) {
  if (done && *done) return;
  const unsigned bx = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;  // a solver's sweep-direction scheme
  f.prepare();
  const int64_t n2 = n >> 1;
  double2v *__restrict__ y2 = reinterpret_cast<double2v *>(p.y);
  const double2v *__restrict__ a2 = reinterpret_cast<const double2v *>(p.x0);
  const double2v *__restrict__ b2 = reinterpret_cast<const double2v *>(p.x1);
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * kUnroll) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * kUnroll)) {
    double2v vy[kUnroll], va[kUnroll], vb[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        if (F::reads_y) vy[u] = ld2(y2 + i, nt);
        if (F::nin > 0) va[u] = ld2(a2 + i, nt);
        if (F::nin > 1) vb[u] = ld2(b2 + i, nt);
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        double2v o;
        o.x = f(F::reads_y ? vy[u].x : 0.0, F::nin > 0 ? va[u].x : 0.0, F::nin > 1 ? vb[u].x : 0.0);
        o.y = f(F::reads_y ? vy[u].y : 0.0, F::nin > 0 ? va[u].y : 0.0, F::nin > 1 ? vb[u].y : 0.0);
        st2(y2 + i, o, nt);
      }
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    const double vy = F::reads_y ? p.y[i] : 0.0;
    const double va = F::nin > 0 ? p.x0[i] : 0.0;
    const double vb = F::nin > 1 ? p.x1[i] : 0.0;
    p.y[i] = f(vy, va, vb);
  }
}

struct FillF {
  static constexpr bool reads_y = false;
  static constexpr int nin = 0;
  double v;
  __device__ void prepare() {}
  __device__ double operator()(double, double, double) const { return v; }
};
struct CopyF {
  static constexpr bool reads_y = false;
  static constexpr int nin = 1;
  __device__ void prepare() {}
  __device__ double operator()(double, double a, double) const { return a; }
};
struct ScaleF {
  static constexpr bool reads_y = true;
  static constexpr int nin = 0;
  Scal s;
  bool divide;
  double sv;
  __device__ void prepare() { sv = ld_scal(s); }
  __device__ double operator()(double y, double, double) const { return divide ? y / sv : y * sv; }
};
// y = a*x0 + b*x1 (either input may be y itself)
struct AxpbzF {
  static constexpr bool reads_y = false;
  static constexpr int nin = 2;
  Scal a, b;
  double av, bv;
  __device__ void prepare() { av = ld_scal(a), bv = ld_scal(b); }
  __device__ double operator()(double, double x0, double x1) const { return av * x0 + bv * x1; }
};
// p = r + beta*(p - omega*v)     SolverBiCgStab.hpp:119
struct BicgPF {
  static constexpr bool reads_y = true;
  static constexpr int nin = 2;
  Scal beta, omega;
  double bv, wv;
  __device__ void prepare() { bv = ld_scal(beta), wv = ld_scal(omega); }
  __device__ double operator()(double p, double r, double v) const { return r + bv * (p - wv * v); }
};

// y += s * (x0 .* x1)
struct VmulAddF {
  static constexpr bool reads_y = true;
  static constexpr int nin = 2;
  double s;
  __device__ void prepare() {}
  __device__ double operator()(double y, double a, double b) const { return y + s * (a * b); }
};

// y = x0 .* x1
struct VmulF {
  static constexpr bool reads_y = false;
  static constexpr int nin = 2;
  __device__ void prepare() {}
  __device__ double operator()(double, double a, double b) const { return a * b; }
};

// y = (s * x0) / x1  (x0 absent: y = s / x1): the elementwise quotients of Bittern/MatrixMath.hpp:261-265, :298-302
struct VdivF {
  double s;
  int has_a;
  static constexpr bool reads_y = false;
  static constexpr int nin = 2;
  __device__ void prepare() {}
  __device__ double operator()(double, double a, double b) const { return has_a ? (s * a) / b : s / b; }
};

// y = func(x0, x1, y) elementwise, func a short arithmetic program in reverse Polish notation (storm_hip_map): the
// element map `out <<= map(func, mats...)` of Bittern/MatrixMath.hpp:44-105 for callables made of exactly-rounded
// operations.  Every operation is a statement of its own, so nothing contracts (-ffp-contract=on fuses within a statement
// only): the value is what the host's scalar evaluation of the same expression gives, bit for bit.
//
// The program the kernel runs is the caller's after one peephole pass on the host (map_compile): a push that is
// followed by a binary operation disappears into it -- `top = top OP operand` --, so a
// left-deep expression like 2 c (c - 1)(2 c - 1) runs in 9 steps with 3 pushes instead of 13 with 7.  The operand stack
// lives in NAMED registers, as many as the program needs (2, 4 or 8: a push or a pop shifts them; no dynamically indexed
// array, no scratch), and the kernel streams only the vectors the program reads.  Round 6: the first version (always three
// input streams, eight registers shifted at every push, 13 steps) ran `map(dF_dc, c)` at 0.20 of the HBM peak.
constexpr int kMapMaxOps = 48, kMapMaxConsts = 16, kMapMaxDepth = 8;
enum { kMapFuse = 32 };  // internal opcodes kMapFuse + (OP - ADD): top = top OP src
struct MapProgram {
  int n_ops;
  int code[kMapMaxOps];  // opcode | source kind << 8 | constant index << 16
  double consts[kMapMaxConsts];
};
// One thread: kUnroll pairs of rows (E = 8 elements), ONE pass of the interpreter over all of them -- a step is decoded
// once (scalar: the program sits in the kernel's arguments) and applied to eight elements; the stack is DEPTH x E named
// registers.  The streaming shape is ew_kernel's.
template <int DEPTH, bool READS_Y, int NIN>
__global__ __launch_bounds__(kBlock) void map_kernel(int64_t n, EwPtrs p, MapProgram prog, const int *done, int nt, int reverse) {
  if (done && *done) return;
  constexpr int E = 2 * kUnroll;
  const unsigned bx = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const int64_t n2 = n >> 1;
  double2v *__restrict__ y2 = reinterpret_cast<double2v *>(p.y);
  const double2v *__restrict__ a2 = reinterpret_cast<const double2v *>(p.x0);
  const double2v *__restrict__ b2 = reinterpret_cast<const double2v *>(p.x1);
  auto run = [&](const double (&vy)[E], const double (&va)[E], const double (&vb)[E], double (&out)[E]) {
    double s[DEPTH][E];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int e = 0; e < E; ++e) s[d][e] = 0.0;
    for (int k = 0; k < prog.n_ops; ++k) {
      const int word = prog.code[k], op = word & 0xff, kind = (word >> 8) & 0xff;
      const double cst = prog.consts[(word >> 16) & 0xff];
      // the step's operand for element e (a uniform choice: one scalar branch per step, not per element)
      auto with_src = [&](auto &&fn) {
        if (kind == STORM_HIP_MAP_X0) {
#pragma unroll
          for (int e = 0; e < E; ++e) fn(e, va[e]);
        } else if (kind == STORM_HIP_MAP_X1) {
#pragma unroll
          for (int e = 0; e < E; ++e) fn(e, vb[e]);
        } else if (kind == STORM_HIP_MAP_Y) {
#pragma unroll
          for (int e = 0; e < E; ++e) fn(e, vy[e]);
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) fn(e, cst);
        }
      };
      auto with_binary = [&](int opc, auto &&apply) {  // apply(f) with f(l, r) the operation: chosen once per step
        if (opc == STORM_HIP_MAP_ADD) apply([](double l, double r) { return l + r; });
        else if (opc == STORM_HIP_MAP_SUB) apply([](double l, double r) { return l - r; });
        else if (opc == STORM_HIP_MAP_MUL) apply([](double l, double r) { return l * r; });
        else if (opc == STORM_HIP_MAP_DIV) apply([](double l, double r) { return l / r; });
        else if (opc == STORM_HIP_MAP_MIN) apply([](double l, double r) { return r < l ? r : l; });  // std::min(l, r)
        else apply([](double l, double r) { return l < r ? r : l; });                               // std::max(l, r)
      };
      if (op >= kMapFuse) {  // top = top OP operand
        with_binary(op - kMapFuse + STORM_HIP_MAP_ADD, [&](auto f) { with_src([&](int e, double v) { s[0][e] = f(s[0][e], v); }); });
      } else if (op < STORM_HIP_MAP_NEG) {  // push
#pragma unroll
        for (int d = DEPTH - 1; d >= 1; --d)
#pragma unroll
          for (int e = 0; e < E; ++e) s[d][e] = s[d - 1][e];
        with_src([&](int e, double v) { s[0][e] = v; });
      } else if (op < STORM_HIP_MAP_ADD) {
        if (op == STORM_HIP_MAP_NEG) {
#pragma unroll
          for (int e = 0; e < E; ++e) s[0][e] = -s[0][e];
        } else if (op == STORM_HIP_MAP_ABS) {
#pragma unroll
          for (int e = 0; e < E; ++e) s[0][e] = __builtin_fabs(s[0][e]);
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) s[0][e] = __builtin_sqrt(s[0][e]);
        }
      } else {  // second OP top, pop
        with_binary(op, [&](auto f) {
#pragma unroll
          for (int e = 0; e < E; ++e) s[0][e] = f(s[1 < DEPTH ? 1 : 0][e], s[0][e]);
        });
#pragma unroll
        for (int d = 1; d + 1 < DEPTH; ++d)
#pragma unroll
          for (int e = 0; e < E; ++e) s[d][e] = s[d + 1][e];
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] = s[0][e];
  };
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * kUnroll) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * kUnroll)) {
    double vy[E], va[E], vb[E], out[E];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      double2v ty{0.0, 0.0}, ta{0.0, 0.0}, tb{0.0, 0.0};
      if (i < n2) {
        if (READS_Y) ty = ld2(y2 + i, nt);
        if (NIN > 0) ta = ld2(a2 + i, nt);
        if (NIN > 1) tb = ld2(b2 + i, nt);
      }
      vy[2 * u] = ty.x, vy[2 * u + 1] = ty.y, va[2 * u] = ta.x, va[2 * u + 1] = ta.y, vb[2 * u] = tb.x, vb[2 * u + 1] = tb.y;
    }
    run(vy, va, vb, out);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) st2(y2 + i, double2v{out[2 * u], out[2 * u + 1]}, nt);
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {  // the odd last row: the same interpreter, seven idle elements
    const int64_t i = n - 1;
    double vy[E], va[E], vb[E], out[E];
#pragma unroll
    for (int e = 0; e < E; ++e) vy[e] = va[e] = vb[e] = 0.0;
    vy[0] = READS_Y ? p.y[i] : 0.0, va[0] = NIN > 0 ? p.x0[i] : 0.0, vb[0] = NIN > 1 ? p.x1[i] : 0.0;
    run(vy, va, vb, out);
    p.y[i] = out[0];
  }
}

// The caller's program (validated by storm_hip_map) -> the kernel's: pushes folded into the binary operation behind them.
// Returns the operand-stack depth the result needs; *reads: bit 0 x0, bit 1 x1, bit 2 y.
static int map_compile(const int32_t *program, int n_ops, int *code_out, int *n_out, int *reads) {
  int n = 0, depth = 0, max_depth = 0;
  *reads = 0;
  auto is_push = [](int op) { return op == STORM_HIP_MAP_X0 || op == STORM_HIP_MAP_X1 || op == STORM_HIP_MAP_Y || op == STORM_HIP_MAP_CONST; };
  auto is_binary = [](int op) { return op >= STORM_HIP_MAP_ADD && op <= STORM_HIP_MAP_MAX; };
  for (int k = 0; k < n_ops; ++k) {
    const int op = program[k] & 0xff, arg = program[k] >> 8;
    if (is_push(op)) {
      if (op != STORM_HIP_MAP_CONST) *reads |= 1 << op;
      const int src = (op << 8) | ((op == STORM_HIP_MAP_CONST ? arg : 0) << 16);
      // `... S OP`: the push of S and the operation in one step, top = top OP S (needs something under S: depth >= 1)
      if (k + 1 < n_ops && is_binary(program[k + 1] & 0xff) && depth >= 1) {
        code_out[n++] = (kMapFuse + (program[k + 1] & 0xff) - STORM_HIP_MAP_ADD) | src;
        ++k;
        continue;
      }
      code_out[n++] = op | src;  // (a plain push keeps its opcode; the source travels in the same fields)
      max_depth = std::max(max_depth, ++depth);
    } else if (is_binary(op)) {
      // `S <expression> OP` where the expression was just finished and S was pushed right under it cannot be seen from here
      // without a stack of provenances: only the operand-LAST form is folded (what a left-deep expression produces)
      code_out[n++] = op;
      --depth;
    } else {
      code_out[n++] = op;
    }
  }
  *n_out = n;
  return max_depth;
}

template <class F>
static int launch_ew(storm_hip_ctx *c, int64_t n, EwPtrs p, F f, const int *done) {
  if (n <= 0) return STORM_HIP_OK;
  hipLaunchKernelGGL(ew_kernel<F>, dim3(stream_blocks(n)), dim3(kBlock), 0, c->stream, n, p, f, done,
                     stream_nt(c, n), c->stream_reverse);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

int k_fill(storm_hip_ctx *c, double *y, int64_t n, double v) {
  return launch_ew(c, n, EwPtrs{y, nullptr, nullptr}, FillF{v}, nullptr);
}
int k_copy(storm_hip_ctx *c, double *y, const double *x, int64_t n, const int *done) {
  return launch_ew(c, n, EwPtrs{y, x, nullptr}, CopyF{}, done);
}
int k_scale(storm_hip_ctx *c, double *y, int64_t n, Scal s, bool divide, const int *done) {
  return launch_ew(c, n, EwPtrs{y, nullptr, nullptr}, ScaleF{s, divide, 0.0}, done);
}
int k_axpbz(storm_hip_ctx *c, double *y, Scal a, const double *x, Scal b, const double *z, int64_t n,
            const int *done) {
  return launch_ew(c, n, EwPtrs{y, x, z}, AxpbzF{a, b, 0.0, 0.0}, done);
}
int k_bicg_p(storm_hip_ctx *c, double *p, const double *r, Scal beta, Scal omega, const double *v,
             int64_t n, const int *done) {
  return launch_ew(c, n, EwPtrs{p, r, v}, BicgPF{beta, omega, 0.0, 0.0}, done);
}

// y = r + s*(a*x + b*z), evaluated in exactly this nesting (operands may alias y).
__global__ __launch_bounds__(kBlock) void lin3_kernel(int64_t n, double *y, const double *r, double s, double a,
                                                      const double *x, double b, const double *z, const int *done, int nt) {
  if (done && *done) return;
  const int64_t n2 = n >> 1;
  double2v *y2 = reinterpret_cast<double2v *>(y);
  const double2v *r2 = reinterpret_cast<const double2v *>(r), *x2 = reinterpret_cast<const double2v *>(x),
                 *z2 = reinterpret_cast<const double2v *>(z);
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)blockIdx.x * (kBlock * kUnroll) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * kUnroll)) {
    double2v vr[kUnroll], vx[kUnroll], vz[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) vr[u] = ld2(r2 + i, nt), vx[u] = ld2(x2 + i, nt), vz[u] = ld2(z2 + i, nt);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) st2(y2 + i, vr[u] + s * (a * vx[u] + b * vz[u]), nt);
    }
  }
  });
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = r[n - 1] + s * (a * x[n - 1] + b * z[n - 1]);
}

// ---- reductions -------------------------------------------------------------------------

// Sum over the 256 threads of a block, fixed order; result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *lds4) {
  v = wave_sum_down(v);  // (the __shfl_down tree's order and bits, without the LDS crossbar: wave_device.hpp)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  __syncthreads();  // lds4 may still be read by a previous call
  if (lane == 0) lds4[wave] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

template <int KB>
__global__ __launch_bounds__(kBlock) void multi_dot_kernel(int64_t n, const double *__restrict__ a,
                                                           DotPtrs bs, double *__restrict__ partials,
                                                           const int *done, int nt) {
  if (done && *done) return;
  __shared__ double lds[KB][4];
  double acc[KB], sums[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) acc[j] = 0.0;
  multi_dot_accumulate<KB>(n, a, bs, nt, acc);
  const unsigned bx = (nt & 2) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  block_sum_multi<KB>(acc, lds, sums);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < KB; ++j) partials[(int64_t)j * gridDim.x + bx] = sums[j];
  }
}

// The same with the reduction finished in the kernel (ticket_device.hpp): out[j] = <a, bs.b[j]>, one launch.
// host_words != null: the last block also stores the sums into pinned HOST memory, each as two self-validating words
// { tag | low half }, { tag | high half } (one atomic system-scope store each: no ordering between them and a flag to
// rely on) -- the host polls them instead of copying and waiting on the stream.
template <int KB>
__global__ __launch_bounds__(kBlock) void multi_dot_ticket_kernel(int64_t n, const double *__restrict__ a, DotPtrs bs,
                                                                  TicketArgs tickets, double *__restrict__ out,
                                                                  const int *done, int nt,
                                                                  unsigned long long *host_words, unsigned tag) {
  if (done && *done) return;
  __shared__ double lds[KB][4];
  double acc[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) acc[j] = 0.0;
  multi_dot_accumulate<KB>(n, a, bs, nt, acc);
  double mine[KB], total[KB];
  block_sum_multi<KB>(acc, lds, mine);
  if (threadIdx.x >= kWave) return;
  const unsigned bx = (nt & 2) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  if (ticket_reduce_wave0<KB>(tickets, mine, KB, bx, gridDim.x, total) && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      out[j] = total[j];
      if (host_words) {
        const unsigned long long t = (unsigned long long)tag << 32;
        __hip_atomic_store(host_words + 2 * j, t | (unsigned)__double2loint(total[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_words + 2 * j + 1, t | (unsigned)__double2hiint(total[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// out[j] = sum_b partials[j * nblocks + b]; one block per j, fixed order.
__global__ __launch_bounds__(kBlock) void reduce_final_kernel(const double *__restrict__ partials,
                                                              int nblocks, double *__restrict__ out,
                                                              const int *done) {
  if (done && *done) return;
  __shared__ double lds4[4];
  const double *p = partials + (int64_t)blockIdx.x * nblocks;
  double v = 0.0;
#pragma unroll 8
  for (int i = threadIdx.x; i < nblocks; i += kBlock) v += p[i];
  const double s = block_sum(v, lds4);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// First pass when a kernel left many partials: kStage2 blocks per array fold it to kStage2 values.
__global__ __launch_bounds__(kBlock) void reduce_stage1_plain_kernel(const double *__restrict__ partials,
                                                                     int nblocks, double *__restrict__ out,
                                                                     const int *done) {
  if (done && *done) return;
  __shared__ double lds4[4];
  const int j = blockIdx.y, g = blockIdx.x;
  const int chunk = (nblocks + gridDim.x - 1) / gridDim.x;
  const int i0 = g * chunk, i1 = min(i0 + chunk, nblocks);
  const double *p = partials + (int64_t)j * nblocks;
  double v = 0.0;
  for (int i = i0 + threadIdx.x; i < i1; i += kBlock) v += p[i];
  const double s = block_sum(v, lds4);
  if (threadIdx.x == 0) out[j * gridDim.x + g] = s;
}

int k_reduce_final(storm_hip_ctx *c, const double *partials, int nblocks, int k, double *d_out,
                   const int *done) {
  if (nblocks > kSinglePassPartials) {
    hipLaunchKernelGGL(reduce_stage1_plain_kernel, dim3(kStage2, k), dim3(kBlock), 0, c->stream, partials,
                       nblocks, c->d_partials2, done);
    HIP_TRY(hipGetLastError());
    partials = c->d_partials2;
    nblocks = kStage2;
  }
  hipLaunchKernelGGL(reduce_final_kernel, dim3(k), dim3(kBlock), 0, c->stream, partials, nblocks,
                     d_out, done);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

// Per-block partials of <a, bs[j]>, j < k, into c->d_partials[j * nb + block]; *nb_out = nb.
int k_multi_dot_partials(storm_hip_ctx *c, const double *a, const double *const *bs, int k, int64_t n,
                         int *nb_out, const int *done) {
  STORM_REQUIRE(k >= 1 && k <= kMaxMulti, "multi_dot: k = %d outside [1, %d]", k, kMaxMulti);
  int nb = stream_blocks(n);
  if ((int64_t)nb * k > c->partials_capacity) nb = (int)(c->partials_capacity / k);  // grid-stride covers the rest
  const int nt = stream_nt(c, n) | (c->stream_reverse << 1);
  for (int j0 = 0; j0 < k; j0 += kDotChunk) {
    const int kb = (k - j0) < kDotChunk ? (k - j0) : kDotChunk;
    DotPtrs ptrs;
    for (int j = 0; j < kDotChunk; ++j) ptrs.b[j] = bs[j0 + (j < kb ? j : 0)];
    double *part = c->d_partials + (int64_t)j0 * nb;
    const dim3 g(nb), b(kBlock);
    switch (kb) {
      case 1: hipLaunchKernelGGL(multi_dot_kernel<1>, g, b, 0, c->stream, n, a, ptrs, part, done, nt); break;
      case 2: hipLaunchKernelGGL(multi_dot_kernel<2>, g, b, 0, c->stream, n, a, ptrs, part, done, nt); break;
      case 3: hipLaunchKernelGGL(multi_dot_kernel<3>, g, b, 0, c->stream, n, a, ptrs, part, done, nt); break;
      case 4: hipLaunchKernelGGL(multi_dot_kernel<4>, g, b, 0, c->stream, n, a, ptrs, part, done, nt); break;
      case 5: hipLaunchKernelGGL(multi_dot_kernel<5>, g, b, 0, c->stream, n, a, ptrs, part, done, nt); break;
      case 6: hipLaunchKernelGGL(multi_dot_kernel<6>, g, b, 0, c->stream, n, a, ptrs, part, done, nt); break;
      case 7: hipLaunchKernelGGL(multi_dot_kernel<7>, g, b, 0, c->stream, n, a, ptrs, part, done, nt); break;
      default: hipLaunchKernelGGL(multi_dot_kernel<8>, g, b, 0, c->stream, n, a, ptrs, part, done, nt); break;
    }
    HIP_TRY(hipGetLastError());
  }
  *nb_out = nb;
  return STORM_HIP_OK;
}

static int k_multi_dot_host(storm_hip_ctx *c, const double *a, const double *const *bs, int k, int64_t n,
                            double *d_out, const int *done, unsigned long long *host_words, unsigned tag);
int k_multi_dot(storm_hip_ctx *c, const double *a, const double *const *bs, int k, int64_t n,
                double *d_out, const int *done) {
  return k_multi_dot_host(c, a, bs, k, n, d_out, done, nullptr, 0u);
}
static int k_multi_dot_host(storm_hip_ctx *c, const double *a, const double *const *bs, int k, int64_t n,
                            double *d_out, const int *done, unsigned long long *host_words, unsigned tag) {
  if (c->opt_ticket_reduce != 0 && k <= kDotChunk && n > 0) {  // one launch: partials, tickets, the sums
    int nbt = stream_blocks(n);
    if ((int64_t)nbt * k > c->partials_capacity) nbt = (int)(c->partials_capacity / k);
    const int nt = stream_nt(c, n) | (c->stream_reverse << 1);
    DotPtrs ptrs;
    for (int j = 0; j < kDotChunk; ++j) ptrs.b[j] = bs[j < k ? j : 0];
    const TicketArgs t{c->d_tickets, c->d_partials, c->d_ticket_sums};
    const dim3 g(nbt), b(kBlock);
#define TD_GO(K_) hipLaunchKernelGGL(multi_dot_ticket_kernel<K_>, g, b, 0, c->stream, n, a, ptrs, t, d_out, done, nt, host_words, tag)
    switch (k) {
      case 1: TD_GO(1); break;
      case 2: TD_GO(2); break;
      case 3: TD_GO(3); break;
      case 4: TD_GO(4); break;
      case 5: TD_GO(5); break;
      case 6: TD_GO(6); break;
      case 7: TD_GO(7); break;
      default: TD_GO(8); break;
    }
#undef TD_GO
    HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
  }
  int nb = 0;
  STORM_TRY(k_multi_dot_partials(c, a, bs, k, n, &nb, done));
  return k_reduce_final(c, c->d_partials, nb, k, d_out, done);
}

// Per-block partials of <a, b> without the final pass (the consumer folds them: solvers.hip, fused MGS).
int k_dot_partials(storm_hip_ctx *c, const double *a, const double *b, int64_t n, double *partials, int nb,
                   const int *done) {
  DotPtrs ptrs;
  for (int j = 0; j < kDotChunk; ++j) ptrs.b[j] = b;
  hipLaunchKernelGGL(multi_dot_kernel<1>, dim3(nb), dim3(kBlock), 0, c->stream, n, a, ptrs, partials, done,
                     stream_nt(c, n));
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

// ---- multi-axpy ------------------------------------------------------------------------------
constexpr int kAxpyChunk = 8;
struct AxpyArgs {
  const double *x[kAxpyChunk];
  double c[kAxpyChunk];
  const double *dc;  // device coefficients (override c[] when non-null)
  double sign;
};

template <int KB>
__global__ __launch_bounds__(kBlock) void multi_axpy_kernel(int64_t n, double *__restrict__ y,
                                                            AxpyArgs a, const int *done, int nt) {
  if (done && *done) return;
  double cf[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) cf[j] = a.dc ? a.dc[j] * a.sign : a.c[j];
  const int64_t n2 = n >> 1;
  double2v *__restrict__ y2 = reinterpret_cast<double2v *>(y);
  constexpr int U = KB <= 2 ? kUnroll : (KB <= 4 ? 2 : 1);
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)blockIdx.x * (kBlock * kUnroll) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * kUnroll)) {
#pragma unroll
    for (int u0 = 0; u0 < kUnroll; u0 += U) {
      double2v vy[U], vx[U][KB];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = base + (u0 + u) * kBlock;
        if (i < n2) {
          vy[u] = ld2(y2 + i, nt);
#pragma unroll
          for (int j = 0; j < KB; ++j) vx[u][j] = ld2(reinterpret_cast<const double2v *>(a.x[j]) + i, nt);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = base + (u0 + u) * kBlock;
        if (i < n2) {
#pragma unroll
          for (int j = 0; j < KB; ++j) {
            vy[u].x += cf[j] * vx[u][j].x;
            vy[u].y += cf[j] * vx[u][j].y;
          }
          st2(y2 + i, vy[u], nt);
        }
      }
    }
  }
  });
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double vy = y[n - 1];
#pragma unroll
    for (int j = 0; j < KB; ++j) vy += cf[j] * a.x[j][n - 1];
    y[n - 1] = vy;
  }
}

static int multi_axpy_impl(storm_hip_ctx *c, double *y, const double *h_coef, const double *d_coef,
                           double sign, const double *const *xs, int k, int64_t n, const int *done) {
  if (n <= 0 || k <= 0) return STORM_HIP_OK;
  const dim3 g(stream_blocks(n)), b(kBlock);
  const int nt = stream_nt(c, n);
  for (int j0 = 0; j0 < k; j0 += kAxpyChunk) {
    const int kb = (k - j0) < kAxpyChunk ? (k - j0) : kAxpyChunk;
    AxpyArgs a;
    for (int j = 0; j < kAxpyChunk; ++j) {
      a.x[j] = xs[j0 + (j < kb ? j : 0)];
      a.c[j] = (h_coef && j < kb) ? h_coef[j0 + j] : 0.0;
    }
    a.dc = d_coef ? d_coef + j0 : nullptr;
    a.sign = sign;
    switch (kb) {
      case 1: hipLaunchKernelGGL(multi_axpy_kernel<1>, g, b, 0, c->stream, n, y, a, done, nt); break;
      case 2: hipLaunchKernelGGL(multi_axpy_kernel<2>, g, b, 0, c->stream, n, y, a, done, nt); break;
      case 3: hipLaunchKernelGGL(multi_axpy_kernel<3>, g, b, 0, c->stream, n, y, a, done, nt); break;
      case 4: hipLaunchKernelGGL(multi_axpy_kernel<4>, g, b, 0, c->stream, n, y, a, done, nt); break;
      case 5: hipLaunchKernelGGL(multi_axpy_kernel<5>, g, b, 0, c->stream, n, y, a, done, nt); break;
      case 6: hipLaunchKernelGGL(multi_axpy_kernel<6>, g, b, 0, c->stream, n, y, a, done, nt); break;
      case 7: hipLaunchKernelGGL(multi_axpy_kernel<7>, g, b, 0, c->stream, n, y, a, done, nt); break;
      default: hipLaunchKernelGGL(multi_axpy_kernel<8>, g, b, 0, c->stream, n, y, a, done, nt); break;
    }
    HIP_TRY(hipGetLastError());
  }
  return STORM_HIP_OK;
}

int k_multi_axpy(storm_hip_ctx *c, double *y, const double *d_coef, double sign,
                 const double *const *xs, int k, int64_t n, const int *done) {
  return multi_axpy_impl(c, y, nullptr, d_coef, sign, xs, k, n, done);
}

}  // namespace storm

using namespace storm;

// ---- C ABI --------------------------------------------------------------------------------------
static int check_pair(const storm_hip_vec *a, const storm_hip_vec *b, const char *what) {
  STORM_REQUIRE(a && b, "%s: null vector", what);
  STORM_REQUIRE(a->ctx == b->ctx, "%s: vectors belong to different contexts", what);
  STORM_REQUIRE(a->n_owned == b->n_owned, "%s: size mismatch (%lld vs %lld owned rows)", what,
                (long long)a->n_owned, (long long)b->n_owned);
  return STORM_HIP_OK;
}

// Finish a host-visible reduction: sum over ranks, copy k scalars to the host.
static int finish_reduction(storm_hip_ctx *c, int k, double *out) {
  if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, c->d_scalars, k));
  HIP_TRY(hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(double) * (size_t)k, hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int j = 0; j < k; ++j) out[j] = c->h_scalars[j];
  return STORM_HIP_OK;
}

extern "C" {

// option lazy_statements (lazy.hip): linear statements wait for the call that needs them -- outside solver callbacks only
static inline bool lazy_on(const storm_hip_ctx *c) { return c->opt_lazy != 0 && c->callback_depth == 0 && c->api_done == nullptr; }

int storm_hip_fill(storm_hip_vec *y, double value) {
  STORM_REQUIRE(y, "fill: null vector");
  STORM_TRY(lazy_sync(y->ctx));
  return launch_ew(y->ctx, y->n_owned, EwPtrs{y->d, nullptr, nullptr}, FillF{value}, y->ctx->api_done);
}

int storm_hip_copy(storm_hip_vec *y, const storm_hip_vec *x) {
  STORM_TRY(check_pair(y, x, "copy"));
  if (lazy_on(y->ctx) && y->d != x->d) return lazy_push_lin(y->ctx, y, 1.0, x->d, 0.0, nullptr, 1, y->n_owned);
  STORM_TRY(lazy_sync(y->ctx));
  return k_copy(y->ctx, y->d, x->d, y->n_owned, y->ctx->api_done);
}

int storm_hip_scale(storm_hip_vec *y, double s) {
  STORM_REQUIRE(y, "scale: null vector");
  if (lazy_on(y->ctx)) return lazy_push_lin(y->ctx, y, s, y->d, 0.0, nullptr, 1, y->n_owned);
  STORM_TRY(lazy_sync(y->ctx));
  return k_scale(y->ctx, y->d, y->n_owned, host_scal(s), false, nullptr);
}

int storm_hip_div_scalar(storm_hip_vec *y, double s) {
  STORM_REQUIRE(y, "div_scalar: null vector");
  STORM_TRY(lazy_sync(y->ctx));
  return k_scale(y->ctx, y->d, y->n_owned, host_scal(s), true, nullptr);
}

int storm_hip_axpy(storm_hip_vec *y, double a, const storm_hip_vec *x) {
  STORM_TRY(check_pair(y, x, "axpy"));
  if (lazy_on(y->ctx)) return lazy_push_lin(y->ctx, y, a, x->d, 1.0, y->d, 2, y->n_owned);
  STORM_TRY(lazy_sync(y->ctx));
  return k_axpbz(y->ctx, y->d, host_scal(a), x->d, host_scal(1.0), y->d, y->n_owned, y->ctx->api_done);
}

int storm_hip_xpay(storm_hip_vec *y, const storm_hip_vec *x, double b) {
  STORM_TRY(check_pair(y, x, "xpay"));
  if (lazy_on(y->ctx)) return lazy_push_lin(y->ctx, y, 1.0, x->d, b, y->d, 2, y->n_owned);
  STORM_TRY(lazy_sync(y->ctx));
  return k_axpbz(y->ctx, y->d, host_scal(1.0), x->d, host_scal(b), y->d, y->n_owned, y->ctx->api_done);
}

int storm_hip_axpbz(storm_hip_vec *y, double a, const storm_hip_vec *x, double b, const storm_hip_vec *z) {
  STORM_TRY(check_pair(y, x, "axpbz"));
  STORM_TRY(check_pair(y, z, "axpbz"));
  if (lazy_on(y->ctx)) return lazy_push_lin(y->ctx, y, a, x->d, b, z->d, 2, y->n_owned);
  STORM_TRY(lazy_sync(y->ctx));
  return k_axpbz(y->ctx, y->d, host_scal(a), x->d, host_scal(b), z->d, y->n_owned, y->ctx->api_done);
}

int storm_hip_lin3(storm_hip_vec *y, const storm_hip_vec *r, double s, double a, const storm_hip_vec *x, double b,
                   const storm_hip_vec *z) {
  STORM_TRY(check_pair(y, r, "lin3"));
  STORM_TRY(check_pair(y, x, "lin3"));
  STORM_TRY(check_pair(y, z, "lin3"));
  STORM_TRY(lazy_sync(y->ctx));
  if (y->n_owned <= 0) return STORM_HIP_OK;
  storm_hip_ctx *c = y->ctx;
  hipLaunchKernelGGL(lin3_kernel, dim3(stream_blocks(y->n_owned)), dim3(kBlock), 0, c->stream, y->n_owned, y->d,
                     r->d, s, a, x->d, b, z->d, c->api_done, stream_nt(c, y->n_owned));
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

int storm_hip_vmul_add(storm_hip_vec *y, double s, const storm_hip_vec *a, const storm_hip_vec *b) {
  STORM_TRY(check_pair(y, a, "vmul_add"));
  STORM_TRY(check_pair(y, b, "vmul_add"));
  STORM_TRY(lazy_sync(y->ctx));
  if (y->n_owned <= 0) return STORM_HIP_OK;
  return launch_ew(y->ctx, y->n_owned, EwPtrs{y->d, a->d, b->d}, VmulAddF{s}, y->ctx->api_done);
}

int storm_hip_map(storm_hip_vec *y, const storm_hip_vec *x0, const storm_hip_vec *x1, const int32_t *program, int n_ops,
                  const double *constants, int n_constants) {
  STORM_REQUIRE(y && program, "map: null argument");
  STORM_REQUIRE(n_ops >= 1 && n_ops <= kMapMaxOps, "map: a program of %d operations (1 .. %d)", n_ops, kMapMaxOps);
  STORM_REQUIRE(n_constants >= 0 && n_constants <= kMapMaxConsts && (n_constants == 0 || constants),
                "map: %d constants (0 .. %d)", n_constants, kMapMaxConsts);
  if (x0) STORM_TRY(check_pair(y, x0, "map"));
  if (x1) STORM_TRY(check_pair(y, x1, "map"));
  int depth = 0;
  for (int k = 0; k < n_ops; ++k) {  // the program is checked here, once: the kernel trusts it
    const int op = program[k] & 0xff, arg = program[k] >> 8;
    if (op == STORM_HIP_MAP_X0 || op == STORM_HIP_MAP_X1 || op == STORM_HIP_MAP_Y || op == STORM_HIP_MAP_CONST) {
      STORM_REQUIRE(op != STORM_HIP_MAP_X0 || x0, "map: operation %d reads x0, which is null", k);
      STORM_REQUIRE(op != STORM_HIP_MAP_X1 || x1, "map: operation %d reads x1, which is null", k);
      STORM_REQUIRE(op != STORM_HIP_MAP_CONST || (arg >= 0 && arg < n_constants), "map: operation %d: constant %d of %d", k, arg,
                    n_constants);
      ++depth;
      STORM_REQUIRE(depth <= kMapMaxDepth, "map: the expression needs more than %d operands at once (operation %d)", kMapMaxDepth, k);
    } else if (op == STORM_HIP_MAP_NEG || op == STORM_HIP_MAP_ABS || op == STORM_HIP_MAP_SQRT) {
      STORM_REQUIRE(depth >= 1, "map: operation %d has no operand", k);
    } else if (op >= STORM_HIP_MAP_ADD && op <= STORM_HIP_MAP_MAX) {
      STORM_REQUIRE(depth >= 2, "map: operation %d has fewer than two operands", k);
      --depth;
    } else {
      STORM_FAIL(STORM_HIP_E_INVALID, "map: unknown operation code %d at %d", op, k);
    }
  }
  STORM_REQUIRE(depth == 1, "map: the program leaves %d values (it must leave one)", depth);
  int code[kMapMaxOps], n_code = 0, reads = 0;
  const int need = map_compile(program, n_ops, code, &n_code, &reads);
  STORM_TRY(lazy_sync(y->ctx));
  if (y->n_owned <= 0) return STORM_HIP_OK;
  storm_hip_ctx *c = y->ctx;
  const bool ry = (reads & 4) != 0;
  const int nin = (reads & 2) ? 2 : 1;  // (a program that reads no vector at all still streams one: the kernel's shape)
  const EwPtrs ptrs{y->d, x0 ? x0->d : y->d, x1 ? x1->d : y->d};
  MapProgram prog{};
  prog.n_ops = n_code;
  for (int k = 0; k < n_code; ++k) prog.code[k] = code[k];
  for (int k = 0; k < n_constants; ++k) prog.consts[k] = constants[k];
  const dim3 grid(stream_blocks(y->n_owned)), block(kBlock);
  const int nt = stream_nt(c, y->n_owned), rev = c->stream_reverse;
#define STORM_MAP_GO(D, RY, NI) \
  hipLaunchKernelGGL((map_kernel<D, RY, NI>), grid, block, 0, c->stream, y->n_owned, ptrs, prog, c->api_done, nt, rev)
#define STORM_MAP_D(D)                                               \
  do {                                                               \
    if (ry && nin == 2) STORM_MAP_GO(D, true, 2);                    \
    else if (ry) STORM_MAP_GO(D, true, 1);                           \
    else if (nin == 2) STORM_MAP_GO(D, false, 2);                    \
    else STORM_MAP_GO(D, false, 1);                                  \
  } while (0)
  if (need <= 2) STORM_MAP_D(2);
  else if (need <= 4) STORM_MAP_D(4);
  else STORM_MAP_D(8);
#undef STORM_MAP_D
#undef STORM_MAP_GO
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

int storm_hip_vmul(storm_hip_vec *y, const storm_hip_vec *a, const storm_hip_vec *b) {
  STORM_TRY(check_pair(y, a, "vmul"));
  STORM_TRY(check_pair(y, b, "vmul"));
  STORM_TRY(lazy_sync(y->ctx));
  if (y->n_owned <= 0) return STORM_HIP_OK;
  return launch_ew(y->ctx, y->n_owned, EwPtrs{y->d, a->d, b->d}, VmulF{}, y->ctx->api_done);
}

int storm_hip_vdiv(storm_hip_vec *y, double s, const storm_hip_vec *a, const storm_hip_vec *b) {
  STORM_TRY(check_pair(y, b, "vdiv"));
  if (a) STORM_TRY(check_pair(y, a, "vdiv"));
  STORM_TRY(lazy_sync(y->ctx));
  if (y->n_owned <= 0) return STORM_HIP_OK;
  return launch_ew(y->ctx, y->n_owned, EwPtrs{y->d, a ? a->d : b->d, b->d}, VdivF{s, a ? 1 : 0}, y->ctx->api_done);
}

int storm_hip_bicgstab_p(storm_hip_vec *p, const storm_hip_vec *r, double beta, double omega,
                         const storm_hip_vec *v) {
  STORM_TRY(check_pair(p, r, "bicgstab_p"));
  STORM_TRY(check_pair(p, v, "bicgstab_p"));
  STORM_TRY(lazy_sync(p->ctx));
  return k_bicg_p(p->ctx, p->d, r->d, host_scal(beta), host_scal(omega), v->d, p->n_owned, p->ctx->api_done);
}

int storm_hip_multi_dot_begin(const storm_hip_vec *a, const storm_hip_vec *const *bs, int k, int *request) {
  STORM_REQUIRE(a && bs && request, "multi_dot_begin: null argument");
  STORM_REQUIRE(k >= 1 && k <= kMaxMulti, "multi_dot: k = %d outside [1, %d]", k, kMaxMulti);
  const double *ptrs[kMaxMulti];
  for (int j = 0; j < k; ++j) {
    STORM_TRY(check_pair(a, bs[j], "multi_dot"));
    ptrs[j] = bs[j]->d;
  }
  storm_hip_ctx *c = a->ctx;
  STORM_TRY(lazy_sync(c));
  // any free slot (requests may be ended in any order: round 3 derived the slot from the tag, so that eight requests
  // begun and the third ended left "no" slot for the ninth); the request id names its slot
  int s = -1;
  for (int i = 0; i < kResultRing && s < 0; ++i)
    if (c->result_ring[i].tag == 0) s = i;
  STORM_REQUIRE(s >= 0, "multi_dot_begin: %d reductions in flight already (end one first)", kResultRing);
  const unsigned tag = ++c->result_seq ? c->result_seq : ++c->result_seq;  // never 0 (the words start zeroed)
  storm_hip_ctx::ResultSlot &slot = c->result_ring[s];
  slot.k = k;
  slot.ready = false;
  // one rank, one launch: the kernel's last block leaves the sums in pinned host memory; _end polls them
  const bool direct = a->n_owned > 0 && c->comm == nullptr && c->opt_host_result != 0 && c->opt_ticket_reduce != 0 &&
                      k <= kDotChunk && c->api_done == nullptr;
  if (direct) {
    STORM_TRY(k_multi_dot_host(c, a->d, ptrs, k, a->n_owned, c->d_scalars, nullptr, c->d_result_words + 16 * s, tag));
  } else {
    if (a->n_owned == 0) {
      HIP_TRY(hipMemsetAsync(c->d_scalars, 0, sizeof(double) * (size_t)k, c->stream));
    } else {
      STORM_TRY(k_multi_dot(c, a->d, ptrs, k, a->n_owned, c->d_scalars, c->api_done));
    }
    STORM_TRY(finish_reduction(c, k, slot.value));
    slot.ready = true;
  }
  slot.tag = tag;
  *request = (int)(((tag & 0x0fffffffu) << 3) | (unsigned)s);  // slot in the low three bits, the tag's low 28 above
  return STORM_HIP_OK;
}

int storm_hip_multi_dot_end(storm_hip_ctx *c, int request, double *out) {
  STORM_REQUIRE(c && out, "multi_dot_end: null argument");
  static_assert(kResultRing == 8, "a request id carries its slot in three bits");
  const int s = request & 7;
  storm_hip_ctx::ResultSlot &slot = c->result_ring[s];
  STORM_REQUIRE(request > 0 && slot.tag != 0 && (slot.tag & 0x0fffffffu) == (((unsigned)request >> 3) & 0x0fffffffu),
                "multi_dot_end: request %d is not in flight", request);
  const unsigned tag = slot.tag;
  const int k = slot.k;
  slot.tag = 0;
  if (slot.ready) {
    for (int j = 0; j < k; ++j) out[j] = slot.value[j];
    return STORM_HIP_OK;
  }
  volatile unsigned long long *w = c->h_result_words + 16 * s;
  auto arrived = [&]() {
    bool all = true;
    for (int j = 0; j < 2 * k; ++j) all &= (unsigned)(w[j] >> 32) == tag;
    return all;
  };
  for (long spin = 0; !arrived(); ++spin) {
    if ((spin & 0x3fff) == 0x3fff && hipStreamQuery(c->stream) == hipSuccess && !arrived()) {
      // the stream is idle and the words never came (a failed launch): the ordinary road reports it -- or, if the
      // kernel did run, returns the sums it left in device memory (only when no later reduction has replaced them)
      STORM_REQUIRE(c->result_seq == tag, "multi_dot_end: the result of request %d never arrived", request);
      return finish_reduction(c, k, out);
    }
  }
  for (int j = 0; j < k; ++j) {
    const unsigned long long lo = w[2 * j], hi = w[2 * j + 1];
    const unsigned long long bits = (hi << 32) | (lo & 0xffffffffull);
    memcpy(&out[j], &bits, sizeof(double));
  }
  return STORM_HIP_OK;
}

int storm_hip_multi_dot(const storm_hip_vec *a, const storm_hip_vec *const *bs, int k, double *out) {
  STORM_REQUIRE(out, "multi_dot: null argument");
  if (a && bs && k == 1 && bs[0] && a->ctx == bs[0]->ctx && a->n_owned == bs[0]->n_owned && !a->ctx->lazy_q.empty()) {
    // statements wait (option lazy_statements): the reduction rides in the kernel of the one that writes its operand
    int st = STORM_HIP_OK;
    if (lazy_try_dot(a->ctx, a->d, bs[0]->d, a->n_owned, out, &st)) return st;
    STORM_TRY(st);
  }
  int request = 0;
  STORM_TRY(storm_hip_multi_dot_begin(a, bs, k, &request));
  return storm_hip_multi_dot_end(a->ctx, request, out);
}

int storm_hip_dot(const storm_hip_vec *a, const storm_hip_vec *b, double *result) {
  STORM_REQUIRE(result, "dot: null result");
  return storm_hip_multi_dot(a, &b, 1, result);
}

int storm_hip_norm2(const storm_hip_vec *a, double *result) {
  STORM_REQUIRE(result, "norm2: null result");
  double s = 0.0;
  STORM_TRY(storm_hip_multi_dot(a, &a, 1, &s));
  *result = sqrt(s);
  return STORM_HIP_OK;
}

int storm_hip_multi_axpy(storm_hip_vec *y, const double *coefs, const storm_hip_vec *const *xs, int k) {
  STORM_REQUIRE(y && coefs && xs, "multi_axpy: null argument");
  STORM_REQUIRE(k >= 1 && k <= kMaxMulti, "multi_axpy: k = %d outside [1, %d]", k, kMaxMulti);
  const double *ptrs[kMaxMulti];
  for (int j = 0; j < k; ++j) {
    STORM_TRY(check_pair(y, xs[j], "multi_axpy"));
    ptrs[j] = xs[j]->d;
  }
  STORM_TRY(lazy_sync(y->ctx));
  return multi_axpy_impl(y->ctx, y->d, coefs, nullptr, 1.0, ptrs, k, y->n_owned, y->ctx->api_done);
}

}  // extern "C"
