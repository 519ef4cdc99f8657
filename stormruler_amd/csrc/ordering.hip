// Cell orderings from geometry (host code; the role METIS plays in the north star -- the reference's hook is
// UnstructuredMesh::permute, Mallard/MeshUnstructured.hpp:443-459, 557-612; METIS itself is neither in this image nor
// used by the reference, CMakeLists.txt:373-384).
//
//   * LATTICE: when the cell centres form a tensor-product grid -- every coordinate of every cell falls into one of a
//     few well separated LEVELS per axis, and the level triples are a permutation of the grid's points -- the
//     lexicographic order of the levels is returned.  A renumbered structured (or mildly perturbed structured) mesh
//     gets its natural order back, and with it the lattice record formats and kernels of spmv.hip.  O(n), no sort.
//   * MORTON: otherwise the cells follow the Z-order curve of their centres quantised to 21 (3-D) / 31 (2-D) bits per
//     axis: a wavefront's 64 rows are a compact block of the mesh, their neighbours a few cache lines, column offsets
//     repeat (byte-indexed offsets often apply).  Threaded LSD radix sort of the keys.
//
// Measured on the seeded scramble of the 256^3 box (tools/ordering_probe.py, profiles/r04u_*): reverse Cuthill-McKee
// (scipy, 6.4 s) CG 2 720 it/s; Morton (here 0.5 s) 3 415 it/s; lattice (0.3 s) 4 675 it/s = the natural order's.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>
#include <stdexcept>
#include <thread>
#include <vector>

#include "common.hpp"

namespace storm {
namespace {

int order_threads() {
  const char *e = getenv("STORM_HIP_BUILD_THREADS");
  int t = e ? atoi(e) : 0;
  if (t <= 0) t = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  return t;
}
template <class F>
void par_for(int64_t n, int nt, F &&fn) {  // fn(thread, begin, end)
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, (n + 65535) / 65536));
  if (nt == 1) {
    fn(0, (int64_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  const int64_t chunk = (n + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { fn(t, std::min(n, t * chunk), std::min(n, (t + 1) * chunk)); });
  for (auto &x : th) x.join();
}

constexpr int kLevelBins = 1 << 20;

// Lattice detection: per axis the occupied bins of a 2^20-bin histogram of the coordinate form RUNS; a run is a level.
bool lattice_order(int dim, int64_t n, const double *c, const double *lo, const double *hi, int nt, int64_t *order) {
  std::vector<std::vector<int>> level_of_bin((size_t)dim);
  int64_t levels[3] = {1, 1, 1};
  double scale[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) {
    const double span = hi[d] - lo[d];
    scale[d] = span > 0 ? (kLevelBins - 1) / span : 0.0;
    std::vector<unsigned char> occ((size_t)kLevelBins, 0);
    par_for(n, nt, [&](int, int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i) occ[(size_t)((c[i * dim + d] - lo[d]) * scale[d] + 0.5)] = 1;  // (racing stores of the same byte)
    });
    auto &lv = level_of_bin[(size_t)d];
    lv.assign((size_t)kLevelBins, -1);
    int runs = 0;
    for (int b = 0; b < kLevelBins; ++b)
      if (occ[(size_t)b]) {
        if (b == 0 || !occ[(size_t)b - 1]) ++runs;
        lv[(size_t)b] = runs - 1;
      }
    levels[d] = runs;
  }
  // (a grid has far fewer levels per axis than cells, and exactly n points)
  long double prod = 1;
  for (int d = 0; d < dim; ++d) prod *= (long double)levels[d];
  if (prod != (long double)n) return false;
  std::vector<std::atomic<unsigned char>> seen((size_t)n);
  for (auto &s : seen) s.store(0, std::memory_order_relaxed);
  std::atomic<int> clash{0};
  par_for(n, nt, [&](int, int64_t b, int64_t e) {
    for (int64_t i = b; i < e && !clash.load(std::memory_order_relaxed); ++i) {
      int64_t key = 0;
      for (int d = dim - 1; d >= 0; --d)
        key = key * levels[d] + level_of_bin[(size_t)d][(size_t)((c[i * dim + d] - lo[d]) * scale[d] + 0.5)];
      if (seen[(size_t)key].exchange(1, std::memory_order_relaxed)) clash.store(1, std::memory_order_relaxed);
      else order[key] = i;
    }
  });
  return clash.load() == 0;
}

void morton_order(int dim, int64_t n, const double *c, const double *lo, const double *hi, int nt, int64_t *order, bool hilbert = false) {
  const int bits = dim >= 3 ? 21 : dim == 2 ? 31 : 62;
  double scale[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) scale[d] = hi[d] > lo[d] ? (double)(((uint64_t)1 << bits) - 1) / (hi[d] - lo[d]) : 0.0;
  struct KV {
    uint64_t key;
    int64_t idx;
  };
  std::vector<KV> a((size_t)n), b((size_t)n);
  auto spread = [&](uint64_t v) -> uint64_t {  // the bits of v, dim - 1 zero bits after each
    if (dim == 1) return v;
    uint64_t r = 0;
    for (int t = 0; t < bits; ++t) r |= ((v >> t) & 1ull) << (t * dim);
    return r;
  };
  par_for(n, nt, [&](int, int64_t s, int64_t e) {
    for (int64_t i = s; i < e; ++i) {
      uint64_t X[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) X[d] = (uint64_t)((c[i * dim + d] - lo[d]) * scale[d] + 0.5);
      uint64_t key = 0;
      if (hilbert && dim > 1) {
        // the Hilbert curve through the same quantised points (J. Skilling, "Programming the Hilbert curve", 2004:
        // axes -> transposed index): consecutive keys are always face-adjacent boxes -- no jumps across the domain
        const uint64_t M = 1ull << (bits - 1);
        for (uint64_t Q = M; Q > 1; Q >>= 1) {
          const uint64_t P = Q - 1;
          for (int d = 0; d < dim; ++d) {
            if (X[d] & Q) X[0] ^= P;
            else {
              const uint64_t t = (X[0] ^ X[d]) & P;
              X[0] ^= t, X[d] ^= t;
            }
          }
        }
        for (int d = 1; d < dim; ++d) X[d] ^= X[d - 1];
        uint64_t t = 0;
        for (uint64_t Q = M; Q > 1; Q >>= 1)
          if (X[dim - 1] & Q) t ^= Q - 1;
        for (int d = 0; d < dim; ++d) X[d] ^= t;
        for (int d = 0; d < dim; ++d) key |= spread(X[d]) << (dim - 1 - d);  // (axis 0 carries the leading bit of a group)
      } else {
        for (int d = 0; d < dim; ++d) key |= spread(X[d]) << d;
      }
      a[(size_t)i] = KV{key, i};
    }
  });
  // LSD radix sort, 8 bits a pass, stable; per pass: per-thread histograms, one prefix over (digit, thread), scatter
  const int passes = (bits * dim + 7) / 8;
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(nt, (n + 65535) / 65536));
  const int64_t chunk = (n + T - 1) / T;
  std::vector<int64_t> hist((size_t)T * 256);
  for (int p = 0; p < passes; ++p) {
    const int sh = 8 * p;
    std::fill(hist.begin(), hist.end(), 0);
    par_for(n, T, [&](int t, int64_t s, int64_t e) {
      int64_t *h = hist.data() + (size_t)t * 256;
      for (int64_t i = s; i < e; ++i) ++h[(a[(size_t)i].key >> sh) & 255];
    });
    int64_t run = 0;
    for (int dgt = 0; dgt < 256; ++dgt)
      for (int t = 0; t < T; ++t) {
        const int64_t cnt = hist[(size_t)t * 256 + dgt];
        hist[(size_t)t * 256 + dgt] = run;
        run += cnt;
      }
    {
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
          int64_t *h = hist.data() + (size_t)t * 256;
          const int64_t s = std::min(n, t * chunk), e = std::min(n, (t + 1) * chunk);
          for (int64_t i = s; i < e; ++i) b[(size_t)h[(a[(size_t)i].key >> sh) & 255]++] = a[(size_t)i];
        });
      for (auto &x : th) x.join();
    }
    a.swap(b);
  }
  par_for(n, nt, [&](int, int64_t s, int64_t e) {
    for (int64_t i = s; i < e; ++i) order[i] = a[(size_t)i].idx;
  });
}

}  // namespace
}  // namespace storm

using namespace storm;

static int order_cells_impl(int32_t dim, int64_t n_cells, const double *centers, int32_t mode, int64_t *order, int32_t *kind);

extern "C" int storm_hip_order_cells(int32_t dim, int64_t n_cells, const double *centers, int32_t mode, int64_t *order,
                                      int32_t *kind) {
  try {  // (vectors and threads: nothing may leave an extern "C" entry point)
    return order_cells_impl(dim, n_cells, centers, mode, order, kind);
  } catch (const std::bad_alloc &) {
    STORM_FAIL(STORM_HIP_E_ALLOC, "order_cells: out of host memory");
  } catch (const std::exception &e) {
    STORM_FAIL(STORM_HIP_E_INVALID, "order_cells: %s", e.what());
  }
}

static int order_cells_impl(int32_t dim, int64_t n_cells, const double *centers, int32_t mode, int64_t *order, int32_t *kind) {
  STORM_REQUIRE(dim >= 1 && dim <= 3 && n_cells >= 0 && (centers || n_cells == 0) && (order || n_cells == 0),
                "order_cells: bad argument");
  STORM_REQUIRE(mode >= 0 && mode <= 3, "order_cells: mode 0 (lattice, else Morton), 1 (Morton), 2 (lattice or fail), 3 (Hilbert)");
  if (kind) *kind = 0;
  if (n_cells == 0) return STORM_HIP_OK;
  const int nt = order_threads();
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) lo[d] = hi[d] = centers[d];
  {
    const int T = nt;
    std::vector<double> tlo((size_t)T * 3, 1e300), thi((size_t)T * 3, -1e300);
    std::atomic<int> bad{0};  // (min / max drop a NaN: every coordinate is looked at)
    par_for(n_cells, T, [&](int t, int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i)
        for (int d = 0; d < dim; ++d) {
          const double v = centers[i * dim + d];
          if (!std::isfinite(v)) bad.store(1, std::memory_order_relaxed);
          tlo[(size_t)t * 3 + d] = std::min(tlo[(size_t)t * 3 + d], v), thi[(size_t)t * 3 + d] = std::max(thi[(size_t)t * 3 + d], v);
        }
    });
    for (int t = 0; t < T; ++t)
      for (int d = 0; d < dim; ++d) lo[d] = std::min(lo[d], tlo[(size_t)t * 3 + d]), hi[d] = std::max(hi[d], thi[(size_t)t * 3 + d]);
    STORM_REQUIRE(bad.load() == 0, "order_cells: non-finite cell centre");
  }
  for (int d = 0; d < dim; ++d)
    STORM_REQUIRE(std::isfinite(lo[d]) && std::isfinite(hi[d]), "order_cells: non-finite cell centre");
  if (mode == 3) {
    morton_order(dim, n_cells, centers, lo, hi, nt, order, true);
    if (kind) *kind = 3;
    return STORM_HIP_OK;
  }
  if (mode != 1 && lattice_order(dim, n_cells, centers, lo, hi, nt, order)) {
    if (kind) *kind = 1;
    return STORM_HIP_OK;
  }
  STORM_REQUIRE(mode != 2, "order_cells: the cell centres do not form a lattice");
  morton_order(dim, n_cells, centers, lo, hi, nt, order);
  if (kind) *kind = 2;
  return STORM_HIP_OK;
}
