// Storm.hpp -- the reference's Operator / Vector / Solver interface on top of the storm_hip C ABI.
//
// Driver code written against StormRuler's solver layer compiles against this header unchanged:
// the class templates below keep the reference's names, public members, defaults and virtual
// signatures (reference paths relative to its root):
//     Operator, FunctionalOperator, make_operator, make_symmetric_operator   Solvers/Operator.hpp:66-200
//     Preconditioner, IdentityPreconditioner, PreconditionerSide              Solvers/Preconditioner.hpp:39-97
//     Solver, IterativeSolver, InnerOuterIterativeSolver, solve<>, solve_non_uniform
//                                                                             Solvers/Solver.hpp:43-292
//     CgSolver / BiCgStabSolver / GmresSolver                                 Solvers/SolverCg.hpp, SolverBiCgStab.hpp, SolverGmres.hpp
// (three of those reference headers do not compile as shipped -- SURVEY.md headline fact 5 -- so
// they are restated here rather than included).
//
// `DeviceVector` plays the role of `Feathers::Field` (Feathers/Field.hpp:60-114).  Every vector
// statement a solver body executes is intercepted by an overload in this header and lowered to
// exactly ONE C-ABI call (one HIP kernel).  DeviceVector also models the reference's `legacy_vector_like`
// concept (Solvers/Operator.hpp:39-45: `shape()`, `operator()(row, col)`, `assign`), so the reference's own
// solver templates can be instantiated on it (tools/check_reference_binding.sh); its element
// access is a slow host proxy that exists for the concept only -- every statement of the overload census
//  has a non-template overload here for const and non-const operands alike, which overload
// resolution prefers to Bittern's generic element-loop templates.
//
// The solver classes of this header are knob holders: `solve` hands the whole solve to the library's
// device-resident loops (storm_hip_krylov_*, csrc/krylov.hip) for ANY operator -- a `HipStencilOperator`
// binds natively, a lambda through make_operator (as Playground.cpp:151-167 does) as a callback that only
// enqueues its kernels -- and no scalar of a recurrence ever visits the host.
//
// C++17, header-only, needs only <storm_hip.h> and libstorm_hip.so.
#pragma once

#include <storm_hip.h>

#include <array>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <exception>
#include <functional>
#include <initializer_list>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace Storm {

using real_t = double;  // Crow/Base/Types.hpp:38

namespace detail {
// Error convention of SURVEY 8b: a nonzero C-ABI status becomes std::runtime_error, like the
// reference's STORM_THROW_IO (Crow/Base/Exception.hpp:35-44); non-convergence is not an error.
inline void check(int status) {
  if (status != 0) throw std::runtime_error(std::string("storm_hip: ") + storm_hip_last_error());
}
}  // namespace detail

#ifndef STORM_HIP_NO_SOLVERS  // (defined: the reference's own Crow / Solvers headers supply what this block restates)
/// y == 0 ? 0 : x / y                                         (Crow/MathUtils.hpp:49-52)
inline real_t safe_divide(real_t x, real_t y) noexcept { return (y == 0.0) ? 0.0 : (x / y); }

/// Givens rotation (cs, sn, rr) with rr = hypot(a, b)         (Crow/MathUtils.hpp:164-179)
inline std::array<real_t, 3> sym_ortho(real_t a, real_t b) noexcept {
  const real_t rr = std::hypot(a, b);
  if (rr > 0.0) return {a / rr, b / rr, rr};
  return {1.0, 0.0, rr};
}

#endif  // STORM_HIP_NO_SOLVERS
// ---------------------------------------------------------------------------------------------
/// One GPU and its streams / workspaces.  One per process (rank).
class Context {
public:
  explicit Context(int device = 0) { detail::check(storm_hip_ctx_create(device, &_h)); }
  ~Context() { storm_hip_ctx_destroy(_h); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  storm_hip_ctx* handle() const noexcept { return _h; }
  void sync() const { detail::check(storm_hip_ctx_sync(_h)); }
  void set_option(const char* key, long long value) { detail::check(storm_hip_ctx_set_option(_h, key, value)); }
  /// Which path the context's solves took, what its host loops fused ... (`storm_hip_ctx_get_counter`).
  long long counter(const char* key) const {
    int64_t value = 0;
    detail::check(storm_hip_ctx_get_counter(_h, key, &value));
    return (long long)value;
  }
  void comm_init(const void* id128, int n_ranks, int rank) {
    detail::check(storm_hip_ctx_comm_init(_h, id128, n_ranks, rank));
  }

private:
  storm_hip_ctx* _h = nullptr;
};

// ---------------------------------------------------------------------------------------------
class DeviceVector;

namespace expr {
// The few expression shapes the solver bodies build (cf. the op census of SURVEY 8b).
struct Scaled { real_t a; const DeviceVector* v; };                          // a * v
struct Lin2 { real_t a; const DeviceVector* x; real_t b; const DeviceVector* z; };  // a x + b z
struct ScaledLin2 { real_t s; Lin2 e; };                                      // s (a x + b z)
struct Lin3 { const DeviceVector* r; real_t s; Lin2 e; };                     // r + s (a x + b z)
struct Quot { const DeviceVector* v; real_t s; };                             // v / s (true division)
}  // namespace expr

/// N doubles in HBM (+ halo rows in multi-GPU runs): the solver `Vector`.
class DeviceVector {
public:
  DeviceVector() = default;
  DeviceVector(const Context& ctx, std::size_t n_owned, std::size_t n_halo = 0) {
    detail::check(storm_hip_vec_create(ctx.handle(), (int64_t)n_owned, (int64_t)n_halo, &_h));
  }
  DeviceVector(DeviceVector&& o) noexcept : _h(std::exchange(o._h, nullptr)), _view(std::exchange(o._view, false)) {}
  DeviceVector& operator=(DeviceVector&& o) noexcept {
    if (this != &o) {
      release();
      _h = std::exchange(o._h, nullptr);
      _view = std::exchange(o._view, false);
    }
    return *this;
  }
  DeviceVector(const DeviceVector&) = delete;
  DeviceVector& operator=(const DeviceVector&) = delete;
  ~DeviceVector() { release(); }

  /// A non-owning view of a vector the library hands to an operator / preconditioner callback.
  static DeviceVector view_of(storm_hip_vec* handle) noexcept {
    DeviceVector v;
    v._h = handle, v._view = true;
    return v;
  }

  /// Field::assign(other, copy): a new zero-initialised vector shaped like `other`; the reference
  /// ignores `copy` (Feathers/Field.hpp:82-84) and so does this.
  void assign(const DeviceVector& other, bool /*copy*/ = true) {
    storm_hip_vec* fresh = nullptr;
    detail::check(storm_hip_vec_create_like(other._h, &fresh));
    release();
    _h = fresh;
  }

  /// Field::shape() = {N, NumVars}  (Field.hpp:77-79)
  std::array<std::size_t, 2> shape() const {
    int64_t n = 0;
    if (_h) detail::check(storm_hip_vec_size(_h, &n, nullptr));
    return {(std::size_t)n, 1};
  }
  std::size_t size() const { return shape()[0]; }

  /// Field::operator()(row, col) (Field.hpp:104-111; `col` ignored, NumVars == 1).  A HOST PROXY: one blocking
  /// 8-byte copy per call.  It exists so that DeviceVector models the reference's `matrix` concept
  /// (Bittern/Matrix.hpp:40-45: `std::apply(mat, mat.shape())`); no solver statement goes through it.
  real_t operator()(std::size_t row, std::size_t /*col*/ = 0) const {
    real_t value = 0.0;
    detail::check(storm_hip_vec_get(_h, (int64_t)row, &value));
    return value;
  }

  void upload(const real_t* host, std::size_t n) { detail::check(storm_hip_vec_upload(_h, host, (int64_t)n)); }
  void download(real_t* host, std::size_t n) const { detail::check(storm_hip_vec_download(_h, host, (int64_t)n)); }
  std::vector<real_t> to_host() const {
    std::vector<real_t> out(size());
    download(out.data(), out.size());
    return out;
  }
  storm_hip_vec* handle() const noexcept { return _h; }

  // TargetMatrixInterface (Bittern/MatrixTarget.hpp:96-119): one kernel each.
  DeviceVector& operator+=(const expr::Scaled& e) { detail::check(storm_hip_axpy(_h, e.a, e.v->_h)); return *this; }
  DeviceVector& operator-=(const expr::Scaled& e) { detail::check(storm_hip_axpy(_h, -e.a, e.v->_h)); return *this; }
  DeviceVector& operator+=(const DeviceVector& v) { detail::check(storm_hip_axpy(_h, 1.0, v._h)); return *this; }
  DeviceVector& operator-=(const DeviceVector& v) { detail::check(storm_hip_axpy(_h, -1.0, v._h)); return *this; }
  DeviceVector& operator*=(real_t s) { detail::check(storm_hip_scale(_h, s)); return *this; }
  DeviceVector& operator/=(real_t s) { detail::check(storm_hip_div_scalar(_h, s)); return *this; }

private:
  void release() noexcept {
    if (!_view) storm_hip_vec_destroy(_h);
    _h = nullptr, _view = false;
  }
  storm_hip_vec* _h = nullptr;
  bool _view = false;
};

// Expression builders (Bittern/MatrixMath.hpp:247-285 for this vector type).
//
// Bittern's own operators are templates on forwarding references (`operator*(scalar auto, matrix auto&&)` ...):
// for a NON-const DeviceVector lvalue they deduce `DeviceVector&`, an identity binding that would beat an
// overload taking `const DeviceVector&`.  So every first-level operator that has a raw DeviceVector operand is
// spelled out for both `DeviceVector&` and `const DeviceVector&` -- equal conversions, and then the
// non-template wins (tests/cpp/concept_check.cpp plays Bittern's side with templates of the same shape).
#define STORM_HIP_CV1(MACRO) MACRO(const DeviceVector&) MACRO(DeviceVector&)
#define STORM_HIP_CV2(MACRO)                                             \
  MACRO(const DeviceVector&, const DeviceVector&) MACRO(DeviceVector&, const DeviceVector&) \
  MACRO(const DeviceVector&, DeviceVector&) MACRO(DeviceVector&, DeviceVector&)

#define STORM_HIP_SCALED(V) \
  inline expr::Scaled operator*(real_t a, V v) { return {a, &v}; }
STORM_HIP_CV1(STORM_HIP_SCALED)
#define STORM_HIP_QUOT(V) \
  inline expr::Quot operator/(V v, real_t s) { return {&v, s}; }
STORM_HIP_CV1(STORM_HIP_QUOT)
#define STORM_HIP_PLUS_SCALED(V)                                                                    \
  inline expr::Lin2 operator+(V x, const expr::Scaled& s) { return {1.0, &x, s.a, s.v}; }           \
  inline expr::Lin2 operator-(V x, const expr::Scaled& s) { return {1.0, &x, -s.a, s.v}; }          \
  inline expr::Lin3 operator+(V r, const expr::ScaledLin2& e) { return {&r, e.s, e.e}; }
STORM_HIP_CV1(STORM_HIP_PLUS_SCALED)
#define STORM_HIP_SUM(X, Z)                                                      \
  inline expr::Lin2 operator+(X x, Z z) { return {1.0, &x, 1.0, &z}; }           \
  inline expr::Lin2 operator-(X x, Z z) { return {1.0, &x, -1.0, &z}; }
STORM_HIP_CV2(STORM_HIP_SUM)
namespace expr {  // (operands of namespace expr only: declared there, where argument-dependent lookup looks)
inline Lin2 operator+(const Scaled& a, const Scaled& b) { return {a.a, a.v, b.a, b.v}; }
inline ScaledLin2 operator*(real_t s, const Lin2& e) { return {s, e}; }
}  // namespace expr

// out <<= expr   (Bittern/MatrixAlgorithms.hpp:120-124)
#define STORM_HIP_COPY(V)                                              \
  inline DeviceVector& operator<<=(DeviceVector& out, V v) {           \
    detail::check(storm_hip_copy(out.handle(), v.handle()));           \
    return out;                                                        \
  }
STORM_HIP_CV1(STORM_HIP_COPY)
inline DeviceVector& operator<<=(DeviceVector& out, const expr::Scaled& e) {
  detail::check(storm_hip_axpbz(out.handle(), e.a, e.v->handle(), 0.0, e.v->handle()));
  return out;
}
inline DeviceVector& operator<<=(DeviceVector& out, const expr::Lin2& e) {
  detail::check(storm_hip_axpbz(out.handle(), e.a, e.x->handle(), e.b, e.z->handle()));
  return out;
}
inline DeviceVector& operator<<=(DeviceVector& out, const expr::Lin3& e) {
  // p <<= r + beta * (p - omega * v)  (SolverBiCgStab.hpp:119) and  p <<= u + beta * (q + beta * p)
  // (SolverCgs.hpp:122): one kernel that evaluates r + s * (a x + b z) in this nesting
  detail::check(storm_hip_lin3(out.handle(), e.r->handle(), e.s, e.e.a, e.e.x->handle(), e.e.b, e.e.z->handle()));
  return out;
}
inline DeviceVector& operator<<=(DeviceVector& out, const expr::ScaledLin2& e) {
  // z <<= delta_inverse * (z - w)  (SolverNewton.hpp:148): the inner sum is rounded before the
  // scaling, as the reference's expression tree evaluates it
  detail::check(storm_hip_axpbz(out.handle(), e.e.a, e.e.x->handle(), e.e.b, e.e.z->handle()));
  detail::check(storm_hip_scale(out.handle(), e.s));
  return out;
}
inline DeviceVector& operator<<=(DeviceVector& out, const expr::Quot& e) {  // p <<= r / phi, SolverIdrs.hpp:131
  if (e.v != &out) detail::check(storm_hip_copy(out.handle(), e.v->handle()));
  detail::check(storm_hip_div_scalar(out.handle(), e.s));
  return out;
}

// out <<= map(func, mats...)   (Bittern/MatrixMath.hpp:44-105, evaluated by MatrixAlgorithms.hpp:75-79, 120-124)
//
// `func` runs on the device, so it cannot be handed over as a host callable: it is TRACED.  Called once with operands of
// type expr::Sym, a generic lambda records the operations it performs, in the order it performs them, as a short program
// for storm_hip_map -- each operation rounded on its own, so the device's result equals the host's evaluation of the same
// expression bit for bit.  The playground's (Playground.cpp:142-148)
//     constexpr auto dF_dc = [](real_t c) noexcept { return 2.0 * c * (c - 1.0) * (2.0 * c - 1.0); };
//     f <<= map(dF_dc, c);
// needs ONE token changed: `real_t c` -> `auto c`.  Supported: + - * /, unary -, abs, sqrt, min, max, mixed with
// real_t constants; one or two vector operands, or three when one of them is the target itself (`y <<= map(f, y, a, b)`);
// no data-dependent branches (a traced value has no truth value).
namespace expr {
class Sym {
public:
  Sym(real_t constant) { _consts.push_back(constant), _code.push_back(STORM_HIP_MAP_CONST); }  // NOLINT: `2.0 * c`
  static Sym input(int slot) {  // (slot k is recorded as opcode k; `<<=` assigns the slots to x0 / x1 / y)
    Sym s;
    s._code.push_back(slot);
    return s;
  }
  static Sym unary(int op, const Sym& a) {
    Sym s = a;
    s._code.push_back(op);
    return s;
  }
  static Sym binary(int op, const Sym& a, const Sym& b) {
    Sym s = a;
    for (int32_t word : b._code) {
      if ((word & 0xff) == STORM_HIP_MAP_CONST) word = STORM_HIP_MAP_CONST | (s.constant(b._consts[(std::size_t)(word >> 8)]) << 8);
      s._code.push_back(word);
    }
    s._code.push_back(op);
    return s;
  }
  const std::vector<int32_t>& code() const noexcept { return _code; }
  const std::vector<real_t>& constants() const noexcept { return _consts; }

private:
  Sym() = default;
  int32_t constant(real_t value) {  // one slot per distinct bit pattern
    for (std::size_t k = 0; k < _consts.size(); ++k)
      if (std::memcmp(&_consts[k], &value, sizeof value) == 0) return (int32_t)k;
    _consts.push_back(value);
    return (int32_t)_consts.size() - 1;
  }
  std::vector<int32_t> _code;  // a lone constant's word carries index 0: Sym(real_t) pushes it first
  std::vector<real_t> _consts;
};
inline Sym operator+(const Sym& a, const Sym& b) { return Sym::binary(STORM_HIP_MAP_ADD, a, b); }
inline Sym operator-(const Sym& a, const Sym& b) { return Sym::binary(STORM_HIP_MAP_SUB, a, b); }
inline Sym operator*(const Sym& a, const Sym& b) { return Sym::binary(STORM_HIP_MAP_MUL, a, b); }
inline Sym operator/(const Sym& a, const Sym& b) { return Sym::binary(STORM_HIP_MAP_DIV, a, b); }
inline Sym operator-(const Sym& a) { return Sym::unary(STORM_HIP_MAP_NEG, a); }
inline Sym operator+(const Sym& a) { return a; }
inline Sym abs(const Sym& a) { return Sym::unary(STORM_HIP_MAP_ABS, a); }
inline Sym sqrt(const Sym& a) { return Sym::unary(STORM_HIP_MAP_SQRT, a); }
inline Sym min(const Sym& a, const Sym& b) { return Sym::binary(STORM_HIP_MAP_MIN, a, b); }
inline Sym max(const Sym& a, const Sym& b) { return Sym::binary(STORM_HIP_MAP_MAX, a, b); }

struct Mapped {  // map(func, mats...): the traced program and its operands (slot k of the program reads x[k])
  Sym program;
  const DeviceVector* x[3];
};
}  // namespace expr

#define STORM_HIP_MAP1(V)                                        \
  template<class Func>                                           \
  inline expr::Mapped map(Func func, V x0) {                     \
    return {func(expr::Sym::input(0)), {&x0, nullptr, nullptr}}; \
  }
STORM_HIP_CV1(STORM_HIP_MAP1)
#define STORM_HIP_MAP2(X, Z)                                                            \
  template<class Func>                                                                  \
  inline expr::Mapped map(Func func, X x0, Z x1) {                                      \
    return {func(expr::Sym::input(0), expr::Sym::input(1)), {&x0, &x1, nullptr}};       \
  }
STORM_HIP_CV2(STORM_HIP_MAP2)
/// Three operands (one of which must be the target of the `<<=` that consumes the node: the device kernel streams the
/// target and two more vectors).
#define STORM_HIP_MAP3(X, Y, Z)                                                                                \
  template<class Func>                                                                                         \
  inline expr::Mapped map(Func func, X x0, Y x1, Z x2) {                                                       \
    return {func(expr::Sym::input(0), expr::Sym::input(1), expr::Sym::input(2)), {&x0, &x1, &x2}};             \
  }
#define STORM_HIP_MAP3_Z(X, Y) STORM_HIP_MAP3(X, Y, const DeviceVector&) STORM_HIP_MAP3(X, Y, DeviceVector&)
STORM_HIP_CV2(STORM_HIP_MAP3_Z)
#undef STORM_HIP_MAP1
#undef STORM_HIP_MAP2
#undef STORM_HIP_MAP3
#undef STORM_HIP_MAP3_Z
inline DeviceVector& operator<<=(DeviceVector& out, const expr::Mapped& e) {
  // the program's slots -> the kernel's operands: the target itself is `y`, the others x0, x1 in their order
  int opcode_of[3] = {-1, -1, -1};
  const DeviceVector* xs[2] = {nullptr, nullptr};
  int used = 0;
  for (int k = 0; k < 3; ++k) {
    if (e.x[k] == nullptr) continue;
    if (e.x[k] == &out) {
      opcode_of[k] = STORM_HIP_MAP_Y;
    } else {
      if (used == 2) throw std::invalid_argument("map: three operands, none of which is the target of `<<=`");
      xs[used] = e.x[k];
      opcode_of[k] = used == 0 ? STORM_HIP_MAP_X0 : STORM_HIP_MAP_X1;
      ++used;
    }
  }
  std::vector<int32_t> code = e.program.code();
  for (int32_t& word : code)
    if ((word & 0xff) < STORM_HIP_MAP_CONST) word = opcode_of[word & 0xff];
  const auto& consts = e.program.constants();
  detail::check(storm_hip_map(out.handle(), xs[0] ? xs[0]->handle() : nullptr, xs[1] ? xs[1]->handle() : nullptr, code.data(),
                              (int)code.size(), consts.data(), (int)consts.size()));
  return out;
}

/// Bittern/MatrixAlgorithms.hpp:140-153: the reference's engine, distribution and sequence.
inline void fill_randomly(DeviceVector& a) { detail::check(storm_hip_fill_randomly(a.handle())); }

/// y = a .* b elementwise (a diagonal preconditioner's `mul`).
inline void vmul(DeviceVector& y, const DeviceVector& a, const DeviceVector& b) {
  detail::check(storm_hip_vmul(y.handle(), a.handle(), b.handle()));
}

/// Bittern/MatrixAlgorithms.hpp:310-317 (global sum over all ranks).
#define STORM_HIP_DOT(A, B)                                              \
  inline real_t dot_product(A a, B b) {                                  \
    real_t r = 0.0;                                                      \
    detail::check(storm_hip_dot(a.handle(), b.handle(), &r));            \
    return r;                                                            \
  }
STORM_HIP_CV2(STORM_HIP_DOT)
/// Bittern/MatrixAlgorithms.hpp:262-270.
#define STORM_HIP_NORM(A)                                   \
  inline real_t norm_2(A a) {                               \
    real_t r = 0.0;                                         \
    detail::check(storm_hip_norm2(a.handle(), &r));         \
    return r;                                               \
  }
STORM_HIP_CV1(STORM_HIP_NORM)
#undef STORM_HIP_SCALED
#undef STORM_HIP_QUOT
#undef STORM_HIP_PLUS_SCALED
#undef STORM_HIP_SUM
#undef STORM_HIP_COPY
#undef STORM_HIP_DOT
#undef STORM_HIP_NORM
#undef STORM_HIP_CV1
#undef STORM_HIP_CV2
/// ADL hook used at Solvers/Solver.hpp:281 and SolverBiCgStab.hpp:224.
inline void fill_with(DeviceVector& a, real_t value) { detail::check(storm_hip_fill(a.handle(), value)); }

// ---------------------------------------------------------------------------------------------
#ifndef STORM_HIP_NO_SOLVERS
/// Abstract operator y <- A(x).
template<class InVector, class OutVector = InVector>
class Operator {
public:
  virtual ~Operator() = default;

  virtual void mul(OutVector& y_vec, const InVector& x_vec) const = 0;

  /// z <- A(y <- B(x))
  template<class InOutVector = InVector>
  void mul(OutVector& z_vec, InOutVector& y_vec, const Operator<InVector, InOutVector>& other_op,
           const InVector& x_vec) const {
    other_op.mul(y_vec, x_vec);
    mul(z_vec, y_vec);
  }

  /// r <- b - A(x)
  void Residual(OutVector& r_vec, const OutVector& b_vec, const InVector& x_vec) const {
    mul(r_vec, x_vec);
    r_vec <<= b_vec - r_vec;
  }

  real_t ResidualNorm(const OutVector& b_vec, const InVector& x_vec) const {
    OutVector r_vec;
    r_vec.assign(b_vec, false);
    Residual(r_vec, b_vec, x_vec);
    return norm_2(r_vec);
  }

  virtual void conj_mul(InVector& /*x_vec*/, const OutVector& /*y_vec*/) const {
    throw std::runtime_error("`Operator::conj_mul` was not overriden");
  }
};

/// Operator given by callables.
template<class InVector, class OutVector = InVector>
class FunctionalOperator final : public Operator<InVector, OutVector> {
public:
  template<class MatVec>
  explicit FunctionalOperator(MatVec&& mat_vec) : _mat_vec{std::forward<MatVec>(mat_vec)} {}
  template<class MatVec, class ConjMatVec>
  FunctionalOperator(MatVec&& mat_vec, ConjMatVec&& conj_mat_vec)
      : _mat_vec{std::forward<MatVec>(mat_vec)}, _conj_mat_vec{std::forward<ConjMatVec>(conj_mat_vec)} {}

  void mul(OutVector& y_vec, const InVector& x_vec) const override { _mat_vec(y_vec, x_vec); }
  void conj_mul(InVector& x_vec, const OutVector& y_vec) const override {
    if (!_conj_mat_vec)
      throw std::runtime_error("`FunctionalOperator::conj_mul` conjugate product function was not set.");
    _conj_mat_vec(x_vec, y_vec);
  }

private:
  std::function<void(OutVector&, const InVector&)> _mat_vec;
  std::function<void(InVector&, const OutVector&)> _conj_mat_vec;
};

template<class InVector, class OutVector = InVector, class MatVec>
auto make_operator(MatVec&& mat_vec) {
  return std::make_unique<FunctionalOperator<InVector, OutVector>>(std::forward<MatVec>(mat_vec));
}
template<class InVector, class OutVector = InVector, class MatVec, class ConjMatVec>
auto make_operator(MatVec&& mat_vec, ConjMatVec&& conj_mat_vec) {
  return std::make_unique<FunctionalOperator<InVector, OutVector>>(std::forward<MatVec>(mat_vec),
                                                                    std::forward<ConjMatVec>(conj_mat_vec));
}
template<class Vector, class MatVec>
auto make_symmetric_operator(MatVec&& mat_vec) {
  return std::make_unique<FunctionalOperator<Vector>>(mat_vec, std::forward<MatVec>(mat_vec));
}

#endif  // STORM_HIP_NO_SOLVERS

// ---------------------------------------------------------------------------------------------
/// The face-graph operator M in HBM (sliced-ELL records + CSR tail); owns the handle.
class StencilMatrix {
public:
  StencilMatrix() = default;
  /// Diffusion stencil of stormDivGrad (Playground.cpp:115-131) from mesh quantities; see storm_hip.h.
  static StencilMatrix from_faces(const Context& ctx, std::size_t n_owned, std::size_t n_halo,
                                  const std::vector<int64_t>& inner, const std::vector<int64_t>& outer,
                                  const std::vector<real_t>& coef, const std::vector<int64_t>& b_cell,
                                  const std::vector<real_t>& b_coef, const std::vector<real_t>& volume) {
    StencilMatrix m;
    detail::check(storm_hip_op_create_from_faces(ctx.handle(), (int64_t)n_owned, (int64_t)n_halo,
                                                 (int64_t)inner.size(), inner.data(), outer.data(), coef.data(),
                                                 (int64_t)b_cell.size(), b_cell.data(), b_coef.data(),
                                                 volume.data(), &m._h));
    return m;
  }
  /// The same straight from the arrays the face loop reads (Playground.cpp:119-129): the library forms
  /// area / length(center(out) - center(in)) itself (storm_hip_op_create_from_mesh).  `center` = cells x dim, row-major.
  static StencilMatrix from_mesh(const Context& ctx, std::size_t n_owned, std::size_t n_halo, int dim,
                                 const std::vector<int64_t>& inner, const std::vector<int64_t>& outer,
                                 const std::vector<real_t>& area, const std::vector<real_t>& center,
                                 const std::vector<int64_t>& b_cell, const std::vector<real_t>& b_area,
                                 const std::vector<real_t>& b_center, const std::vector<real_t>& volume) {
    StencilMatrix m;
    detail::check(storm_hip_op_create_from_mesh(ctx.handle(), (int64_t)n_owned, (int64_t)n_halo, (int32_t)dim,
                                                (int64_t)inner.size(), inner.data(), outer.data(), area.data(), center.data(),
                                                (int64_t)b_cell.size(), b_cell.data(), b_area.data(), b_center.data(),
                                                volume.data(), &m._h));
    return m;
  }
  static StencilMatrix from_face_weights(const Context& ctx, std::size_t n_owned, std::size_t n_halo,
                                         const std::vector<int64_t>& inner, const std::vector<int64_t>& outer,
                                         const std::vector<real_t>& w_inner, const std::vector<real_t>& w_outer,
                                         const real_t* diag_extra = nullptr) {
    StencilMatrix m;
    detail::check(storm_hip_op_create_from_face_weights(ctx.handle(), (int64_t)n_owned, (int64_t)n_halo,
                                                        (int64_t)inner.size(), inner.data(), outer.data(),
                                                        w_inner.data(), w_outer.data(), diag_extra, &m._h));
    return m;
  }
  /// Takes ownership of an operator handle made through the C ABI.
  static StencilMatrix adopt(storm_hip_op* handle) noexcept {
    StencilMatrix m;
    m._h = handle;
    return m;
  }
  StencilMatrix(StencilMatrix&& o) noexcept : _h(std::exchange(o._h, nullptr)) {}
  StencilMatrix& operator=(StencilMatrix&& o) noexcept {
    if (this != &o) {
      storm_hip_op_destroy(_h);
      _h = std::exchange(o._h, nullptr);
    }
    return *this;
  }
  StencilMatrix(const StencilMatrix&) = delete;
  StencilMatrix& operator=(const StencilMatrix&) = delete;
  ~StencilMatrix() { storm_hip_op_destroy(_h); }

  /// y = beta x + alpha M(x)
  void apply(real_t alpha, real_t beta, const DeviceVector& x, DeviceVector& y) const {
    detail::check(storm_hip_op_apply(_h, alpha, beta, x.handle(), y.handle()));
  }
  /// y += alpha M(x)
  void apply_add(real_t alpha, const DeviceVector& x, DeviceVector& y) const {
    detail::check(storm_hip_op_apply_add(_h, alpha, x.handle(), y.handle()));
  }
  storm_hip_op* handle() const noexcept { return _h; }

private:
  storm_hip_op* _h = nullptr;
};

/// `stormDivGrad(mesh, u, dt, c)` (source_apps/playground/Playground.cpp:115-131): u += dt * div grad c,
/// with `matrix` holding what the face loop reads from `mesh`.  The playground's operator lambda
/// (:153-167) is then, statement for statement,
///     w_hat <<= f + sigma * (c_in - c);  stormDivGrad(L, w_hat, -Gamma, c_in);
///     c_hat <<= c_in;                    stormDivGrad(L, c_hat, -tau, w_hat);
inline void stormDivGrad(const StencilMatrix& matrix, DeviceVector& u, real_t dt, const DeviceVector& c) {
  matrix.apply_add(dt, c, u);
}

/// A mesh in host memory: exactly what stormDivGrad's face loop reads (`storm_hip_mesh`).  `read_mesh_from_tetgen`
/// (Mallard/IoTetgen.hpp:44-235) is `HostMesh::read_tetgen`; the operator over it `matrix()`.  As Playground.cpp:248-255:
///     auto mesh = HostMesh::read_tetgen("mesh/step.1.");   StencilMatrix L = mesh.matrix(ctx);
class HostMesh {
public:
  /// dim: mesh_dim_v<Mesh> the caller expects (2, 3; 0: what the node file says).  I/O errors throw std::runtime_error with
  /// the reference's message (STORM_THROW_IO, Crow/Base/Exception.hpp:35-44).
  static HostMesh read_tetgen(const std::string& prefix, int dim = 0) {
    HostMesh m;
    detail::check(storm_hip_mesh_read_tetgen(prefix.c_str(), (int32_t)dim, &m._h));
    return m;
  }
  HostMesh(HostMesh&& o) noexcept : _h(std::exchange(o._h, nullptr)) {}
  HostMesh& operator=(HostMesh&& o) noexcept {
    if (this != &o) {
      storm_hip_mesh_destroy(_h);
      _h = std::exchange(o._h, nullptr);
    }
    return *this;
  }
  HostMesh(const HostMesh&) = delete;
  HostMesh& operator=(const HostMesh&) = delete;
  ~HostMesh() { storm_hip_mesh_destroy(_h); }

  storm_hip_mesh_view view() const {
    storm_hip_mesh_view v;
    detail::check(storm_hip_mesh_get_view(_h, &v));
    return v;
  }
  std::size_t num_cells() const { return (std::size_t)view().n_cells; }
  std::size_t num_halo_cells() const { return (std::size_t)view().n_halo; }
  /// The gather-form operator of the face loop over this mesh (boundary faces skipped like `interior_faces()`,
  /// Playground.cpp:119, when `neumann`; otherwise they carry a homogeneous Dirichlet ghost value).
  StencilMatrix matrix(const Context& ctx, bool neumann = false) const {
    if (!neumann) return StencilMatrix::adopt(create(ctx, _h));
    const storm_hip_mesh_view v = view();
    storm_hip_mesh* interior = nullptr;  // the same mesh without its labelled faces
    detail::check(storm_hip_mesh_create(v.dim, v.n_cells, v.n_halo, v.n_faces, v.inner, v.outer, v.area, v.center, v.volume, 0,
                                        nullptr, nullptr, nullptr, v.global_id, v.halo_owner, &interior));
    storm_hip_op* op = nullptr;
    const int st = storm_hip_op_create_from_mesh_object(ctx.handle(), interior, &op);
    storm_hip_mesh_destroy(interior);
    detail::check(st);
    return StencilMatrix::adopt(op);
  }
  storm_hip_mesh* handle() const noexcept { return _h; }

private:
  HostMesh() = default;
  static storm_hip_op* create(const Context& ctx, const storm_hip_mesh* mesh) {
    storm_hip_op* op = nullptr;
    detail::check(storm_hip_op_create_from_mesh_object(ctx.handle(), mesh, &op));
    return op;
  }
  storm_hip_mesh* _h = nullptr;
};

/// A = beta I + alpha M as an Operator<DeviceVector> (the caller keeps `matrix` alive, as the
/// reference's operator lambdas capture the mesh by reference, Playground.cpp:152-167).
class HipStencilOperator final : public Operator<DeviceVector> {
public:
  HipStencilOperator(const StencilMatrix& matrix, real_t alpha, real_t beta)
      : _matrix{&matrix}, _alpha{alpha}, _beta{beta} {}
  void mul(DeviceVector& y_vec, const DeviceVector& x_vec) const override {
    _matrix->apply(_alpha, _beta, x_vec, y_vec);
  }
  const StencilMatrix& matrix() const noexcept { return *_matrix; }
  real_t alpha() const noexcept { return _alpha; }
  real_t beta() const noexcept { return _beta; }

private:
  const StencilMatrix* _matrix;
  real_t _alpha, _beta;
};

// ---------------------------------------------------------------------------------------------
#ifndef STORM_HIP_NO_SOLVERS
enum class PreconditionerSide { Left, Right, Symmetric };

template<class Vector>
class Preconditioner : public Operator<Vector> {
public:
  virtual void build(const Vector& /*x_vec*/, const Vector& /*b_vec*/, const Operator<Vector>& /*any_op*/) {}
  virtual void add_secant(const Vector& /*y_vec*/, const Vector& /*s_vec*/) {}
};

template<class Vector>
class IdentityPreconditioner final : public Preconditioner<Vector> {
  void mul(Vector& y_vec, const Vector& x_vec) const override { y_vec <<= x_vec; }
  void conj_mul(Vector& x_vec, const Vector& y_vec) const override { x_vec <<= y_vec; }
};

#endif  // STORM_HIP_NO_SOLVERS

/// Diagonal preconditioner P = diag(A)^-1 of a HipStencilOperator, entirely on the device: the
/// build's own addition behind the reference's pre_op hook (Solver.hpp:74-75).  `build`
/// (Preconditioner.hpp:70-72) reads the diagonal of the operator it is given.
class JacobiPreconditioner final : public Preconditioner<DeviceVector> {
public:
  void build(const DeviceVector& x_vec, const DeviceVector& /*b_vec*/,
             const Operator<DeviceVector>& any_op) override {
    const auto* hip_op = dynamic_cast<const HipStencilOperator*>(&any_op);
    if (hip_op == nullptr) throw std::runtime_error("JacobiPreconditioner needs a HipStencilOperator");
    _dinv.assign(x_vec, false);
    detail::check(storm_hip_op_get_diagonal(hip_op->matrix().handle(), hip_op->alpha(), hip_op->beta(), 1,
                                            _dinv.handle()));
  }
  void mul(DeviceVector& y_vec, const DeviceVector& x_vec) const override { vmul(y_vec, _dinv, x_vec); }
  void conj_mul(DeviceVector& x_vec, const DeviceVector& y_vec) const override { vmul(x_vec, _dinv, y_vec); }
  const DeviceVector& inverse_diagonal() const noexcept { return _dinv; }

private:
  DeviceVector _dinv;
};

#ifndef STORM_HIP_NO_SOLVERS
// ---------------------------------------------------------------------------------------------
template<class InVector, class OutVector = InVector>
class Solver {
public:
  virtual ~Solver() = default;
  virtual bool solve(InVector& x_vec, const OutVector& b_vec, const Operator<InVector, OutVector>& any_op) = 0;
};

namespace detail {
inline std::function<void(const std::string&)>& log_sink() {
  static std::function<void(const std::string&)> sink;  // empty: silent
  return sink;
}
inline void log_solve(std::size_t iteration, real_t absolute_error, real_t relative_error) {
  if (!log_sink()) return;
  char line[128];
  std::snprintf(line, sizeof line, "n_iter: %4zu, abs_err: %-12e, rel_err: %-12e", iteration, absolute_error,
                relative_error);
  log_sink()(line);
}

/// One `storm_hip_krylov` object -- the library's device-resident solver loops (csrc/krylov.hip) -- with the
/// caller's Operator / Preconditioner objects bound to it.  A HipStencilOperator binds natively; any other
/// operator (a lambda through make_operator, Playground.cpp:151-167) and any preconditioner but the Jacobi
/// one bind as callbacks, which only enqueue kernels.  An exception thrown inside a callback aborts the solve
/// and is rethrown from the call that ran it.
class Engine {
public:
  Engine() = default;
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  ~Engine() { storm_hip_krylov_destroy(_h); }

  void bind(int method, const DeviceVector& like, const Operator<DeviceVector>& any_op,
            const Preconditioner<DeviceVector>* pre_op, PreconditionerSide side) {
    storm_hip_ctx* ctx = nullptr;
    check(storm_hip_vec_context(like.handle(), &ctx));
    if (_h == nullptr || ctx != _ctx || method != _method) {
      storm_hip_krylov_destroy(_h);
      _h = nullptr;
      check(storm_hip_krylov_create(ctx, method, &_h));
      _ctx = ctx, _method = method;
    }
    _op = Bound{&any_op, &_thrown};
    if (const auto* hip_op = dynamic_cast<const HipStencilOperator*>(&any_op))
      check(storm_hip_krylov_set_operator(_h, hip_op->matrix().handle(), hip_op->alpha(), hip_op->beta()));
    else
      check(storm_hip_krylov_set_operator_fn(_h, &Engine::trampoline, &_op));
    const int c_side = side == PreconditionerSide::Left    ? STORM_HIP_LEFT
                       : side == PreconditionerSide::Right ? STORM_HIP_RIGHT
                                                           : STORM_HIP_SYMMETRIC;
    _pre = Bound{pre_op, &_thrown};
    if (pre_op == nullptr)
      check(storm_hip_krylov_set_preconditioner_diag(_h, nullptr, c_side));
    else if (const auto* jacobi = dynamic_cast<const JacobiPreconditioner*>(pre_op))
      check(storm_hip_krylov_set_preconditioner_diag(_h, jacobi->inverse_diagonal().handle(), c_side));
    else
      check(storm_hip_krylov_set_preconditioner_fn(_h, &Engine::trampoline, &_pre, c_side));
  }
  storm_hip_krylov* handle() const noexcept { return _h; }
  /// Status of a library call that may have run callbacks.
  void finish(int status) {
    if (_thrown) std::rethrow_exception(std::exchange(_thrown, nullptr));
    check(status);
  }

private:
  struct Bound {
    const Operator<DeviceVector>* op;
    std::exception_ptr* thrown;
  };
  static int trampoline(void* user, storm_hip_vec* y, const storm_hip_vec* x) noexcept {
    auto* bound = static_cast<Bound*>(user);
    try {
      DeviceVector y_view = DeviceVector::view_of(y);
      const DeviceVector x_view = DeviceVector::view_of(const_cast<storm_hip_vec*>(x));
      bound->op->mul(y_view, x_view);
      return 0;
    } catch (...) {
      *bound->thrown = std::current_exception();
      return 1;
    }
  }
  storm_hip_krylov* _h = nullptr;
  storm_hip_ctx* _ctx = nullptr;
  int _method = -1;
  Bound _op{nullptr, nullptr}, _pre{nullptr, nullptr};
  std::exception_ptr _thrown;
};
}  // namespace detail

/// The reference logs one line per solve through spdlog (`STORM_INFO("n_iter: ..., abs_err: ..., rel_err: ...")`,
/// Solvers/Solver.hpp:144-145).  This header has no logging dependency: install a sink to receive the same
/// line (e.g. `Storm::set_log_sink([](const std::string& s) { spdlog::info(s); })`); none installed = silent.
inline void set_log_sink(std::function<void(const std::string&)> sink) { detail::log_sink() = std::move(sink); }

namespace detail {
/// Option `lazy_statements` on the vector's context for the lifetime of the scope (leaving it launches what still waits).
struct LazyScope {
  storm_hip_ctx* ctx = nullptr;
  template<class V>
  LazyScope(const V& v, bool on) {
    if constexpr (std::is_same_v<V, DeviceVector>) {
      if (on && v.handle() != nullptr && storm_hip_vec_context(v.handle(), &ctx) == STORM_HIP_OK && ctx != nullptr)
        (void)storm_hip_ctx_set_option(ctx, "lazy_statements", 2);
    }
  }
  ~LazyScope() {
    if (ctx != nullptr) (void)storm_hip_ctx_set_option(ctx, "lazy_statements", 0);
  }
  LazyScope(const LazyScope&) = delete;
  LazyScope& operator=(const LazyScope&) = delete;
};
}  // namespace detail

/// Solvers/Solver.hpp:62-149: the public knobs (names, defaults), the protected stepping hooks and `solve`.
///
/// The solvers this header ships name a library method (`device_method()`); their `solve` is ONE call,
/// `storm_hip_krylov_solve`: the loop of Solver.hpp:132-140 and every scalar of the recurrence stay on the
/// device.  Their init / iterate / finalize hooks go through `storm_hip_krylov_init / _iterate / _finalize`,
/// and `device_loop = false` runs the host loop below over them instead (one host wait per iteration).  A
/// user-defined subclass that names no method gets that host loop over its own hooks.
template<class InVector, class OutVector = InVector>
class IterativeSolver : public Solver<InVector, OutVector> {
public:
  std::size_t iteration{0};
  std::size_t num_iterations{2000};
  real_t absolute_error{0.0};
  real_t relative_error{0.0};

  real_t absolute_error_tolerance{1.0e-6};
  real_t relative_error_tolerance{1.0e-6};

  PreconditionerSide pre_side{PreconditionerSide::Right};
  std::unique_ptr<Preconditioner<InVector>> pre_op{nullptr};
  std::string name;

  // extras of this build
  bool device_loop{true};
  /// The HOST loop (a user-defined solver, or `device_loop = false`) runs with the library's option `lazy_statements`:
  /// the linear vector statements and operator applies of an `iterate()` body wait for the call that needs their result,
  /// consecutive statements leave as one pass and a `dot_product` / `norm_2` over a vector the last waiting statement
  /// writes rides in that statement's kernel (`x += alpha * p; r -= alpha * z; dot_product(r, r)`: ONE kernel).  The
  /// linear statements and their reductions give the eager kernels' bits; two things do not: a dot product riding in an
  /// operator apply sums in the SpMV kernel's order, and (level 2, what this switch selects) `x += alpha p; p <<= r +
  /// beta p; z = A p; <p, z>` on a lattice operator is ONE launch of the device loop's fused step, whose updates round
  /// as fused multiply-adds -- equal to the eager statements to rounding, not to the bit (tests/test_gpu_lazy.py).
  /// Caveat: statements wait until a library call needs their result.  A kernel of YOUR OWN that reads a vector through a
  /// pointer from `storm_hip_vec_device_ptr` must be preceded by `storm_hip_ctx_sync` (or any reduction), which flushes
  /// what waits; level 2 never exchanges the storage of a vector whose address has been handed out.
  /// `false`: every statement is a launch of its own when it is called.
  bool lazy_statements{true};
  std::size_t num_applies{0}, num_pre_applies{0};  ///< of the last device-loop solve
  int path_fallback{0};  ///< storm_hip_solver_result::path_fallback of the last device-loop solve (0: the chosen path ran)

protected:
  virtual real_t init(const InVector& x_vec, const OutVector& b_vec, const Operator<InVector, OutVector>& any_op,
                      const Preconditioner<InVector>* pre_op) = 0;
  virtual real_t iterate(InVector& x_vec, const OutVector& b_vec, const Operator<InVector, OutVector>& any_op,
                         const Preconditioner<InVector>* pre_op) = 0;
  virtual void finalize(InVector& /*x_vec*/, const OutVector& /*b_vec*/,
                        const Operator<InVector, OutVector>& /*any_op*/, const Preconditioner<InVector>* /*pre_op*/) {}

  /// storm_hip_method of a shipped solver; -1: none (the host loop runs the hooks above).
  virtual int device_method() const noexcept { return -1; }
  virtual void fill_params(storm_hip_solver_params& /*p*/) const {}
  virtual void configure(storm_hip_krylov* /*k*/) const {}

  storm_hip_solver_params params() const {
    storm_hip_solver_params p;
    storm_hip_solver_params_default(&p);
    p.num_iterations = (int64_t)num_iterations;
    p.absolute_error_tolerance = absolute_error_tolerance;
    p.relative_error_tolerance = relative_error_tolerance;
    p.num_inner_iterations = 0;  // "the method's default"; InnerOuterIterativeSolver fills its knob in
    fill_params(p);
    return p;
  }
  detail::Engine _engine;

public:
  bool solve(InVector& x_vec, const OutVector& b_vec, const Operator<InVector, OutVector>& any_op) final {
    if (pre_op != nullptr) pre_op->build(x_vec, b_vec, any_op);
    if constexpr (std::is_same_v<InVector, DeviceVector> && std::is_same_v<OutVector, DeviceVector>) {
      if (device_method() >= 0 && device_loop) {
        _engine.bind(device_method(), x_vec, any_op, pre_op.get(), pre_side);
        configure(_engine.handle());
        const storm_hip_solver_params p = params();
        storm_hip_solver_result r{};
        int64_t n_pre = 0;
        _engine.finish(storm_hip_krylov_solve(_engine.handle(), b_vec.handle(), x_vec.handle(), &p, &r, nullptr, &n_pre));
        iteration = (std::size_t)r.iterations;
        absolute_error = r.absolute_error, relative_error = r.relative_error;
        num_applies = (std::size_t)r.num_applies, num_pre_applies = (std::size_t)n_pre;
        path_fallback = r.path_fallback;
        detail::log_solve(iteration, absolute_error, relative_error);
        return r.converged != 0;
      }
    }
    // Host loop with the reference's rule: stop on abs_tol > 0 && abs < abs_tol, or rel_tol > 0 && abs / initial < rel_tol.
    detail::LazyScope lazy(x_vec, lazy_statements);
    const real_t initial_error = init(x_vec, b_vec, any_op, pre_op.get());
    absolute_error = initial_error;
    const auto met = [this](real_t value, real_t tolerance) { return tolerance > 0.0 && value < tolerance; };
    bool converged = met(absolute_error, absolute_error_tolerance);
    const bool silent = converged;  // the early return of Solver.hpp:124-128 logs nothing
    iteration = 0;
    while (!converged && iteration < num_iterations) {
      absolute_error = iterate(x_vec, b_vec, any_op, pre_op.get());
      relative_error = absolute_error / initial_error;
      converged = met(absolute_error, absolute_error_tolerance) || met(relative_error, relative_error_tolerance);
      ++iteration;
    }
    finalize(x_vec, b_vec, any_op, pre_op.get());
    if (!silent) detail::log_solve(iteration, absolute_error, relative_error);
    return converged;
  }
};

/// Solvers/Solver.hpp:154-259: `num_inner_iterations` and the restart bookkeeping over five hooks.  (For the shipped
/// solvers that bookkeeping lives in the library; they implement only outer_init / inner_iterate / outer_finalize.)
template<class InVector, class OutVector = InVector>
class InnerOuterIterativeSolver : public IterativeSolver<InVector, OutVector> {
public:
  std::size_t inner_iteration{0};
  std::size_t num_inner_iterations{50};

protected:
  virtual real_t outer_init(const InVector& x_vec, const OutVector& b_vec,
                            const Operator<InVector, OutVector>& any_op, const Preconditioner<InVector>* pre_op) = 0;
  virtual void inner_init(const InVector&, const OutVector&, const Operator<InVector, OutVector>&,
                          const Preconditioner<InVector>*) {}
  virtual real_t inner_iterate(InVector& x_vec, const OutVector& b_vec,
                               const Operator<InVector, OutVector>& any_op, const Preconditioner<InVector>* pre_op) = 0;
  virtual void inner_finalize(InVector&, const OutVector&, const Operator<InVector, OutVector>&,
                              const Preconditioner<InVector>*) {}
  virtual void outer_finalize(InVector&, const OutVector&, const Operator<InVector, OutVector>&,
                              const Preconditioner<InVector>*) {}

  void fill_params(storm_hip_solver_params& p) const override { p.num_inner_iterations = (int64_t)num_inner_iterations; }

private:
  bool closes_a_cycle() const noexcept { return inner_iteration + 1 == num_inner_iterations; }

  real_t init(const InVector& x_vec, const OutVector& b_vec, const Operator<InVector, OutVector>& any_op,
              const Preconditioner<InVector>* pre_op) final {
    return outer_init(x_vec, b_vec, any_op, pre_op);
  }
  real_t iterate(InVector& x_vec, const OutVector& b_vec, const Operator<InVector, OutVector>& any_op,
                 const Preconditioner<InVector>* pre_op) final {
    inner_iteration = this->iteration % num_inner_iterations;
    if (inner_iteration == 0) inner_init(x_vec, b_vec, any_op, pre_op);
    const real_t residual_norm = inner_iterate(x_vec, b_vec, any_op, pre_op);
    if (closes_a_cycle()) inner_finalize(x_vec, b_vec, any_op, pre_op);
    return residual_norm;
  }
  void finalize(InVector& x_vec, const OutVector& b_vec, const Operator<InVector, OutVector>& any_op,
                const Preconditioner<InVector>* pre_op) final {
    if (!closes_a_cycle()) inner_finalize(x_vec, b_vec, any_op, pre_op);
    outer_finalize(x_vec, b_vec, any_op, pre_op);
  }
};

template<template<class> class SolverT, class Vector>
bool solve(Vector& x_vec, const Vector& b_vec, const Operator<Vector>& any_op) {
  SolverT<Vector> solver{};
  return solver.solve(x_vec, b_vec, any_op);
}

/// A(x) = b for a non-uniform operator (A(0) != 0): solve A(x) - A(0) = b - A(0).  Solver.hpp:271-292.
template<class Vector>
bool solve_non_uniform(Solver<Vector>& solver, Vector& x_vec, const Vector& b_vec, const Operator<Vector>& any_op) {
  Vector at_zero, rhs;
  at_zero.assign(x_vec, false);
  rhs.assign(b_vec, false);
  fill_with(rhs, 0.0);
  any_op.mul(at_zero, rhs);
  rhs <<= b_vec - at_zero;
  const auto shifted = make_operator<Vector>([&](Vector& y_vec, const Vector& in_vec) {
    any_op.mul(y_vec, in_vec);
    y_vec -= at_zero;
  });
  return solver.solve(x_vec, rhs, *shifted);
}

// ---------------------------------------------------------------------------------------------
// The shipped solvers.  Each names its library method; the loop bodies are csrc/krylov.hip (and, for a stencil
// operator without preconditioner, the fused kernels of csrc/solvers.hip).  They exist for DeviceVector only.
namespace detail {

/// Stepping hooks of a shipped plain solver over the library's stepping calls.
template<class Vector, int Method>
class DeviceIterativeSolver : public IterativeSolver<Vector> {
  static_assert(std::is_same_v<Vector, DeviceVector>, "the shipped solvers run on Storm::DeviceVector");

protected:
  int device_method() const noexcept final { return Method; }
  real_t init(const Vector& x_vec, const Vector& b_vec, const Operator<Vector>& any_op,
              const Preconditioner<Vector>* pre_op) final {
    this->_engine.bind(Method, x_vec, any_op, pre_op, this->pre_side);
    this->configure(this->_engine.handle());
    const storm_hip_solver_params p = this->params();
    real_t initial_error = 0.0;
    this->_engine.finish(storm_hip_krylov_init(this->_engine.handle(), b_vec.handle(), x_vec.handle(), &p, &initial_error));
    return initial_error;
  }
  real_t iterate(Vector&, const Vector&, const Operator<Vector>&, const Preconditioner<Vector>*) final {
    real_t residual_norm = 0.0;
    this->_engine.finish(storm_hip_krylov_iterate(this->_engine.handle(), &residual_norm));
    return residual_norm;
  }
  void finalize(Vector&, const Vector&, const Operator<Vector>&, const Preconditioner<Vector>*) final {
    this->_engine.finish(storm_hip_krylov_finalize(this->_engine.handle()));
  }
};

/// The same for a restarted solver: the library restarts / updates x inside its iterate and finalize.
template<class Vector, int Method, std::size_t DefaultInner>
class DeviceInnerOuterSolver : public InnerOuterIterativeSolver<Vector> {
  static_assert(std::is_same_v<Vector, DeviceVector>, "the shipped solvers run on Storm::DeviceVector");

public:
  DeviceInnerOuterSolver() { this->num_inner_iterations = DefaultInner; }

protected:
  int device_method() const noexcept final { return Method; }
  real_t outer_init(const Vector& x_vec, const Vector& b_vec, const Operator<Vector>& any_op,
                    const Preconditioner<Vector>* pre_op) final {
    this->_engine.bind(Method, x_vec, any_op, pre_op, this->pre_side);
    this->configure(this->_engine.handle());
    const storm_hip_solver_params p = this->params();
    real_t initial_error = 0.0;
    this->_engine.finish(storm_hip_krylov_init(this->_engine.handle(), b_vec.handle(), x_vec.handle(), &p, &initial_error));
    return initial_error;
  }
  real_t inner_iterate(Vector&, const Vector&, const Operator<Vector>&, const Preconditioner<Vector>*) final {
    real_t residual_norm = 0.0;
    this->_engine.finish(storm_hip_krylov_iterate(this->_engine.handle(), &residual_norm));
    return residual_norm;
  }
  void outer_finalize(Vector&, const Vector&, const Operator<Vector>&, const Preconditioner<Vector>*) final {
    this->_engine.finish(storm_hip_krylov_finalize(this->_engine.handle()));
  }
};

}  // namespace detail

/// SolverCg.hpp:47-128.
template<class Vector>
class CgSolver final : public detail::DeviceIterativeSolver<Vector, STORM_HIP_CG> {};
/// SolverBiCgStab.hpp:52-167.
template<class Vector>
class BiCgStabSolver final : public detail::DeviceIterativeSolver<Vector, STORM_HIP_BICGSTAB> {};
/// SolverCgs.hpp:50-176.
template<class Vector>
class CgsSolver final : public detail::DeviceIterativeSolver<Vector, STORM_HIP_CGS> {};
/// SolverTfqmr.hpp:227-240 / :252-265.
template<class Vector>
class TfqmrSolver final : public detail::DeviceIterativeSolver<Vector, STORM_HIP_TFQMR> {};
template<class Vector>
class Tfqmr1Solver final : public detail::DeviceIterativeSolver<Vector, STORM_HIP_TFQMR1> {};
/// SolverRichardson.hpp:41-98.
template<class Vector>
class RichardsonSolver final : public detail::DeviceIterativeSolver<Vector, STORM_HIP_RICHARDSON> {
public:
  real_t relaxation_factor{1.0e-4};

private:
  void configure(storm_hip_krylov* k) const override {
    detail::check(storm_hip_krylov_set_real(k, "relaxation_factor", relaxation_factor));
  }
};
/// SolverGmres.hpp:281-283 (default restart 50, Solver.hpp:159).  `gram_schmidt = 1`: classical Gram-Schmidt x2.
template<class Vector>
class GmresSolver final : public detail::DeviceInnerOuterSolver<Vector, STORM_HIP_GMRES, 50> {
public:
  int gram_schmidt{0};

private:
  void fill_params(storm_hip_solver_params& p) const override {
    p.num_inner_iterations = (int64_t)this->num_inner_iterations;
    p.gram_schmidt = gram_schmidt;
  }
};
/// SolverGmres.hpp:306-308: one preconditioned vector per inner iteration, so the preconditioner may vary.
template<class Vector>
class FgmresSolver final : public detail::DeviceInnerOuterSolver<Vector, STORM_HIP_FGMRES, 50> {};
/// SolverBiCgStab.hpp:184-383; `num_inner_iterations` is l (default 2, :379-381).
template<class Vector>
class BiCgStabLSolver final : public detail::DeviceInnerOuterSolver<Vector, STORM_HIP_BICGSTAB_L, 2> {};
/// SolverIdrs.hpp:52-291; `num_inner_iterations` is s (default 4, :287-289).
template<class Vector>
class IdrsSolver final : public detail::DeviceInnerOuterSolver<Vector, STORM_HIP_IDRS, 4> {};

/// SolverNewton.hpp:55-72: declared but unimplemented in the reference (STORM_ABORT); here the same
/// message arrives as an exception instead of std::abort().
template<class Vector>
class NewtonSolver : public IterativeSolver<Vector> {
  [[noreturn]] static void unimplemented() { throw std::runtime_error("Newton solver is not implemented yet!"); }
  real_t init(const Vector&, const Vector&, const Operator<Vector>&, const Preconditioner<Vector>*) override {
    unimplemented();
  }
  real_t iterate(Vector&, const Vector&, const Operator<Vector>&, const Preconditioner<Vector>*) override {
    unimplemented();
  }
};

/// SolverNewton.hpp:101-173: first-order Jacobian-free Newton-Krylov; `any_op` may be nonlinear.  A user-level
/// solver on the host loop: every Newton step solves J(x) t = r with a BiCgStabSolver at 1e-8 (:133-135) whose
/// operator is the finite-difference product (A(x + delta y) - A(x)) / delta, delta = mu / |y| (:136-148).
template<class Vector>
class JfnkSolver final : public IterativeSolver<Vector> {
public:
  std::size_t inner_iterations{0};

private:
  Vector _shifted, _step, _residual, _at_x;

  real_t residual_of(const Vector& x_vec, const Vector& b_vec, const Operator<Vector>& any_op) {
    any_op.mul(_at_x, x_vec);
    _residual <<= b_vec - _at_x;
    return norm_2(_residual);
  }
  real_t init(const Vector& x_vec, const Vector& b_vec, const Operator<Vector>& any_op,
              const Preconditioner<Vector>*) override {
    for (Vector* work : {&_shifted, &_step, &_residual, &_at_x}) work->assign(x_vec, false);
    inner_iterations = 0;
    return residual_of(x_vec, b_vec, any_op);
  }
  real_t iterate(Vector& x_vec, const Vector& b_vec, const Operator<Vector>& any_op,
                 const Preconditioner<Vector>*) override {
    const real_t mu = std::sqrt(std::numeric_limits<real_t>::epsilon()) * std::sqrt(1.0 + norm_2(x_vec));
    const auto jacobian = make_operator<Vector>([&](Vector& z_vec, const Vector& y_vec) {
      const real_t delta = safe_divide(mu, norm_2(y_vec));
      _shifted <<= x_vec + delta * y_vec;
      any_op.mul(z_vec, _shifted);
      z_vec <<= safe_divide(1.0, delta) * (z_vec - _at_x);
    });
    BiCgStabSolver<Vector> inner;
    inner.absolute_error_tolerance = inner.relative_error_tolerance = 1.0e-8;
    _step <<= _residual;
    inner.solve(_step, _residual, *jacobian);
    inner_iterations += inner.iteration;
    x_vec += _step;
    return residual_of(x_vec, b_vec, any_op);
  }
};

#endif  // STORM_HIP_NO_SOLVERS

}  // namespace Storm
