// Compile-time check (g++ -std=c++20 -fsyntax-only; built by __graft_entry__.build and tests/test_abi_symbols.py):
//
//  1. Storm::DeviceVector models what the reference's solver templates require of a `Vector`:
//     `legacy_vector_like` (Solvers/Operator.hpp:39-45) = `matrix` (Bittern/Matrix.hpp:40-45: `shape()` is
//     tuple-like and `std::apply(mat, mat.shape())` is a referenceable call) + `assign(v)` / `assign(v, bool)`.
//     The two concepts are restated below in this file's own words (the reference's headers need fmt/spdlog,
//     absent here; tools/check_reference_binding.sh runs the same check against the real headers).
//  2. Every statement of the overload census resolves to THIS header's overloads -- one kernel
//     each -- even when greedy generic templates of the shape Bittern declares (forwarding references
//     constrained on the matrix concept, Bittern/MatrixMath.hpp:247-285, MatrixAlgorithms.hpp:120-124,262-317)
//     are visible in the same namespace.  `Generic` marks a result that came from such a template.
#include <storm_hip/Storm.hpp>

#include <concepts>
#include <tuple>
#include <type_traits>

namespace probe {
template<class T>
concept referenceable = requires { typename std::add_lvalue_reference_t<T>; } && !std::is_void_v<T>;
template<class M>
concept matrix_like = requires(M& mat) {
  { std::tuple_size_v<std::remove_cvref_t<decltype(mat.shape())>> } -> std::convertible_to<std::size_t>;
  { std::apply(mat, mat.shape()) } -> referenceable;
};
template<class V>
concept vector_like = matrix_like<V> && requires(V& target, const V& source, bool copy) {
  { target.assign(source) };
  { target.assign(source, copy) };
};
struct Generic {};  // "a host element loop would have been instantiated"
}  // namespace probe

namespace Storm {  // the shape of Bittern's generic operators, declared where ADL finds them
template<class S, probe::matrix_like M>
  requires std::is_arithmetic_v<S>
probe::Generic operator*(S, M&&);
template<probe::matrix_like M, class S>
  requires std::is_arithmetic_v<S>
probe::Generic operator/(M&&, S);
template<probe::matrix_like A, probe::matrix_like B>
probe::Generic operator+(A&&, B&&);
template<probe::matrix_like A, probe::matrix_like B>
probe::Generic operator-(A&&, B&&);
template<probe::matrix_like Out, probe::matrix_like M>
probe::Generic operator<<=(Out&&, M&&);
template<probe::matrix_like A, probe::matrix_like B>
probe::Generic dot_product(A&&, B&&);
template<probe::matrix_like A>
probe::Generic norm_2(A&&);
template<probe::matrix_like A>
probe::Generic fill_randomly(A&&);
template<class Func, probe::matrix_like... Ms>  // Bittern/MatrixMath.hpp:100-105
probe::Generic map(Func, Ms&&...);
}  // namespace Storm

using Storm::DeviceVector;
using Storm::real_t;
namespace expr = Storm::expr;

static_assert(probe::matrix_like<DeviceVector>);
static_assert(probe::vector_like<DeviceVector>);
static_assert(std::tuple_size_v<decltype(std::declval<DeviceVector&>().shape())> == 2);  // {N, NumVars}, Field.hpp:77-79
static_assert(std::is_default_constructible_v<DeviceVector> && std::is_nothrow_move_constructible_v<DeviceVector> &&
              std::is_nothrow_move_assignable_v<DeviceVector> && std::is_swappable_v<DeviceVector>);  // std::swap, std::vector<Vector>

template<class T>
T& lvalue();  // a non-const lvalue: what a solver's own work vector (a member) is inside iterate()

#define SAME(expression, Type) static_assert(std::is_same_v<std::remove_cvref_t<decltype(expression)>, Type>)
#define V lvalue<DeviceVector>()
#define CV lvalue<const DeviceVector>()
const real_t a = 2.0;
// a * p                                             SolverCg.hpp:98-99
SAME(a * V, expr::Scaled);
SAME(a * CV, expr::Scaled);
// r + beta * p, b - r, a + b                        SolverCg.hpp:123, Operator.hpp:98, SolverCgs.hpp:142
SAME(V + a * V, expr::Lin2);
SAME(CV + a * CV, expr::Lin2);
SAME(V - a * CV, expr::Lin2);
SAME(CV - V, expr::Lin2);
SAME(V - CV, expr::Lin2);
SAME(V + V, expr::Lin2);
SAME(CV - CV, expr::Lin2);
// r + beta * (p - omega * v)                        SolverBiCgStab.hpp:119
SAME(V + a * (V - a * V), expr::Lin3);
SAME(CV + a * (CV + a * V), expr::Lin3);
// r / phi                                           SolverIdrs.hpp:131
SAME(V / a, expr::Quot);
SAME(CV / a, expr::Quot);
// out <<= ...                                       MatrixAlgorithms.hpp:120-124
SAME(V <<= V, DeviceVector);
SAME(V <<= CV, DeviceVector);
SAME(V <<= CV - V, DeviceVector);
SAME(V <<= V + a * (V - a * V), DeviceVector);
SAME(V <<= V / a, DeviceVector);
// += -= *= /=                                       MatrixTarget.hpp:96-119
SAME(V += a * V, DeviceVector);
SAME(V -= a * CV, DeviceVector);
SAME(V += CV, DeviceVector);
SAME(V -= V, DeviceVector);
SAME(V *= a, DeviceVector);
SAME(V /= a, DeviceVector);
// reductions                                        MatrixAlgorithms.hpp:262-270, 310-317
SAME(dot_product(V, V), real_t);
SAME(dot_product(CV, V), real_t);
SAME(dot_product(V, CV), real_t);
SAME(dot_product(CV, CV), real_t);
SAME(norm_2(V), real_t);
SAME(norm_2(CV), real_t);
SAME(fill_randomly(V), void);
SAME(fill_with(V, 0.0), void);
// f <<= map(dF_dc, c)                               Playground.cpp:142-148 (Bittern/MatrixMath.hpp:100-105): the traced map
constexpr auto dF_dc = [](auto c) noexcept { return 2.0 * c * (c - 1.0) * (2.0 * c - 1.0); };
SAME(map(dF_dc, V), expr::Mapped);
SAME(map(dF_dc, CV), expr::Mapped);
SAME(map([](auto x, auto y) { return x * y + 1.0; }, V, CV), expr::Mapped);
SAME(map([](auto x, auto y) { return x * y + 1.0; }, CV, V), expr::Mapped);
SAME(map([](auto y, auto x, auto z) { return y + x * z; }, V, V, CV), expr::Mapped);  // (three operands: one is the target)
SAME(map([](auto y, auto x, auto z) { return y + x * z; }, V, CV, V), expr::Mapped);
SAME(V <<= map(dF_dc, CV), DeviceVector);
SAME(dF_dc(expr::Sym::input(0)), expr::Sym);

int main() { return 0; }
