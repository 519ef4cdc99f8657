// Host-side meshes for the callers of the hot path (host code only; no device is touched):
//
//   * the Triangle / TetGen reader -- both branches of read_mesh_from_tetgen, Mallard/IoTetgen.hpp:44-235 -- and the
//     face graph the reference's insert() calls would build from it (MeshUnstructured.hpp:350-425, 509-554): what
//     stormDivGrad's face loop reads (Playground.cpp:119-129).  stormruler_amd/io_tetgen.py is the numpy restatement this
//     unit is checked against array for array (tests/test_tetgen_mesh.py); its docstring lists the rules and cites them.
//   * the entity permutation hook (UnstructuredMesh::permute, MeshUnstructured.hpp:443-459) for cells;
//   * the row partition of SURVEY.md 8e: recursive coordinate bisection / slabs, the local graph of a rank (owned
//     cells, then halo cells grouped by owner, each group in ascending global id) and its halo plan -- what
//     stormruler_amd/partition.py computes in numpy (the checker of this unit: tests/test_partition.py) -- so that a
//     C++ driver reaches storm_hip_op_set_halo without Python.  The reference has no partitioner (single process).
//
// Threaded where the work is per-entity (STORM_HIP_BUILD_THREADS, default min(16, cores)).
#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cstdarg>
#include <functional>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

namespace storm {
// std::vector without the zero-fill of resize(): the big arrays are first touched by the threads that fill them
template <class T>
struct default_init_alloc : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = default_init_alloc<U>;
  };
  template <class U>
  void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) {
    ::new (static_cast<void *>(p)) U;
  }
  template <class U, class... A>
  void construct(U *p, A &&...a) {
    ::new (static_cast<void *>(p)) U(std::forward<A>(a)...);
  }
};
template <class T>
using uvec = std::vector<T, default_init_alloc<T>>;
}  // namespace storm

struct storm_hip_mesh {
  int dim = 0;
  int64_t n_cells = 0, n_halo = 0;
  storm::uvec<int64_t> inner, outer;
  storm::uvec<double> area, center, volume;
  std::vector<int64_t> b_cell;
  std::vector<double> b_area, b_center;
  std::vector<int64_t> global_id;   // [n_cells + n_halo] or empty (single rank: the identity)
  std::vector<int32_t> halo_owner;  // [n_halo]
  // halo plan (storm_hip_mesh_partition)
  std::vector<int32_t> nbr_rank;
  std::vector<int64_t> send_ptr, send_idx, recv_ptr;
};

namespace storm {
namespace {

struct MeshError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
[[noreturn]] void fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void fail(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw MeshError(buf);
}

struct PhaseTimer {  // STORM_HIP_MESH_TIMING=1: seconds per phase on stderr
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  bool on = getenv("STORM_HIP_MESH_TIMING") != nullptr;
  void lap(const char *what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[mesh_host] %-28s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  }
};

int host_threads() {
  const char *e = getenv("STORM_HIP_BUILD_THREADS");
  int t = e ? atoi(e) : 0;
  if (t <= 0) t = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  return t;
}
template <class F>
void par_for(int64_t n, F &&fn, int64_t min_chunk = 32768) {  // fn(thread, begin, end)
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (n + min_chunk - 1) / min_chunk));
  if (nt == 1) {
    fn(0, (int64_t)0, n);
    return;
  }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)nt);
  const int64_t chunk = (n + nt - 1) / nt;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      try {
        fn(t, std::min(n, t * chunk), std::min(n, (t + 1) * chunk));
      } catch (...) {
        err[(size_t)t] = std::current_exception();
      }
    });
  for (auto &x : th) x.join();
  for (auto &e : err)
    if (e) std::rethrow_exception(e);
}

// ---- files ------------------------------------------------------------------------------------------------------
// A whole file with its comments blanked: '#' to the end of the line (FilteringStreambuf<'#', '\n'>, IoTetgen.hpp:61).
std::string slurp(const std::string &path, const char *what) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) fail("Cannot open the %s file '%s'!", what, path.c_str());  // STORM_THROW_IO, IoTetgen.hpp:56-58
  std::string s;
  if (fseek(f, 0, SEEK_END) == 0) {
    const long sz = ftell(f);
    if (sz > 0) s.resize((size_t)sz);
    rewind(f);
  }
  const size_t got = s.empty() ? 0 : fread(&s[0], 1, s.size(), f);
  fclose(f);
  if (got != s.size()) fail("Cannot read the %s file '%s'!", what, path.c_str());
  for (char *p = s.empty() ? nullptr : (char *)memchr(s.data(), '#', s.size()); p != nullptr;) {
    char *e = (char *)memchr(p, '\n', (size_t)(s.data() + s.size() - p));
    if (e == nullptr) e = &s[0] + s.size();
    memset(p, ' ', (size_t)(e - p));
    p = e < s.data() + s.size() ? (char *)memchr(e, '#', (size_t)(s.data() + s.size() - e)) : nullptr;
  }
  return s;
}

inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

// All whitespace-separated numbers of the text, in order (operator>> is newline-agnostic, and so is this).
template <class T>
std::vector<T> parse_numbers(const std::string &s, const char *what, const std::string &path) {
  const int64_t n = (int64_t)s.size();
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / (1 << 20)));
  std::vector<int64_t> cut((size_t)nt + 1, 0);
  cut[(size_t)nt] = n;
  for (int t = 1; t < nt; ++t) {  // chunk borders on whitespace
    int64_t p = n * t / nt;
    while (p < n && !is_space(s[(size_t)p])) ++p;
    cut[(size_t)t] = std::max(p, cut[(size_t)t - 1]);
  }
  std::vector<std::vector<T>> part((size_t)nt);
  std::atomic<int> bad{0};
  auto work = [&](int t) {
    const char *p = s.data() + cut[(size_t)t], *e = s.data() + cut[(size_t)t + 1];
    auto &out = part[(size_t)t];
    out.reserve((size_t)((e - p) / 6 + 16));
    while (p < e) {
      while (p < e && is_space(*p)) ++p;
      if (p >= e) break;
      const char *q = p;
      while (q < e && !is_space(*q)) ++q;
      T v{};
      const char *b = (*p == '+') ? p + 1 : p;  // (from_chars takes no leading plus sign; operator>> does)
      const auto r = std::from_chars(b, q, v);
      if (r.ec != std::errc() || r.ptr != q) {
        bad.store(1);
        return;
      }
      out.push_back(v);
      p = q;
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  if (bad.load()) fail("Cannot read the %s from file '%s'!", what, path.c_str());
  size_t total = 0;
  for (auto &v : part) total += v.size();
  std::vector<T> all;
  all.reserve(total);
  for (auto &v : part) all.insert(all.end(), v.begin(), v.end());
  return all;
}

struct Simplices {
  int dim = 0;
  int64_t n_nodes = 0, n_listed = 0, n_cells = 0;
  std::vector<double> pos;      // [n_nodes][dim]
  std::vector<int64_t> listed;  // [n_listed][dim]
  std::vector<int64_t> label;   // [n_listed] or empty
  std::vector<int64_t> cells;   // [n_cells][dim + 1]
};

Simplices read_files(std::string prefix, int want_dim) {
  if (!prefix.empty() && prefix.back() == '.') prefix.pop_back();
  Simplices S;
  PhaseTimer T;
  {
    const std::string path = prefix + ".node";
    const auto t = parse_numbers<double>(slurp(path, "node"), "nodes", path);
    if (t.size() < 4) fail("Cannot read the node file '%s' header!", path.c_str());
    // (header fields are counts: anything else -- a fraction, a NaN, 1e300 -- is not a header; the cast of such a double is undefined)
    for (int k = 0; k < 3; ++k)
      if (!(t[(size_t)k] >= 0.0 && t[(size_t)k] <= 9007199254740992.0 && std::floor(t[(size_t)k]) == t[(size_t)k]))
        fail("Cannot read the node file '%s' header!", path.c_str());
    const int64_t n = (int64_t)t[0], dim = (int64_t)t[1], n_attr = (int64_t)t[2], marker = t[3] != 0 ? 1 : 0;
    if ((dim != 2 && dim != 3) || (want_dim != 0 && dim != want_dim))
      fail("Unexpected number of the dimensions in node file '%s' header! Expected %s, got %lld.", path.c_str(),
           want_dim == 2 ? "2" : want_dim == 3 ? "3" : "2 or 3", (long long)dim);
    const int64_t stride = 1 + dim + n_attr + marker;
    if (n_attr > (int64_t)t.size() || n > ((int64_t)t.size() - 4) / stride) fail("Cannot read the nodes from file '%s'!", path.c_str());
    T.lap("read: .node parsed");
    S.dim = (int)dim, S.n_nodes = n;
    S.pos.resize((size_t)(n * dim));
    par_for(n, [&](int, int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i)
        for (int d = 0; d < dim; ++d) S.pos[(size_t)(i * dim + d)] = t[(size_t)(4 + i * stride + 1 + d)];
    });
  }
  const int dim = S.dim;
  auto sides = [&](const char *ext, const char *what, const char *whats, int npn, std::vector<int64_t> &nodes,
                   std::vector<int64_t> &label, int64_t &count) {
    const std::string path = prefix + ext;
    const auto t = parse_numbers<int64_t>(slurp(path, what), whats, path);
    if (t.size() < 2) fail("Cannot read the %s file '%s' header!", what, path.c_str());
    const int64_t n = t[0], marker = t[1] != 0 ? 1 : 0, stride = 1 + npn + marker;
    if (n < 0 || n > ((int64_t)t.size() - 2) / stride) fail("Cannot read the %s from file '%s'!", whats, path.c_str());
    count = n;
    nodes.resize((size_t)(n * npn));
    label.clear();
    if (marker) label.resize((size_t)n);
    par_for(n, [&](int, int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i) {
        for (int k = 0; k < npn; ++k) nodes[(size_t)(i * npn + k)] = t[(size_t)(2 + i * stride + 1 + k)];
        if (marker) label[(size_t)i] = t[(size_t)(2 + i * stride + 1 + npn)];
      }
    });
  };
  {  // the edges are read in both dimensions (IoTetgen.hpp:103-137); in 3-D they do not enter the face graph
    std::vector<int64_t> en, el;
    int64_t ne = 0;
    sides(".edge", "edge", "edges", 2, en, el, ne);
    for (int64_t v : en)
      if (v < 0 || v >= S.n_nodes) fail("node index out of range (files must be zero-based)");
    if (dim == 2) S.listed.swap(en), S.label.swap(el), S.n_listed = ne;
  }
  if (dim == 3) sides(".face", "face", "faces", 3, S.listed, S.label, S.n_listed);
  T.lap("read: .edge / .face");
  {
    const std::string path = prefix + ".ele";
    const auto t = parse_numbers<int64_t>(slurp(path, "cell"), "cells", path);
    if (t.size() < 3) fail("Cannot read the cell file '%s' header!", path.c_str());
    const int64_t n = t[0], npc = t[1], attr = t[2] != 0 ? 1 : 0, stride = 1 + npc + attr;
    if (npc != dim + 1)
      fail("Unexpected number of the nodes per cell in the cell file '%s' header! Expected %d, got %lld.", path.c_str(),
           dim + 1, (long long)npc);
    if (n < 0 || n > ((int64_t)t.size() - 3) / stride) fail("Cannot read the cells from file '%s'!", path.c_str());
    T.lap("read: .ele parsed");
    S.n_cells = n;
    S.cells.resize((size_t)(n * npc));
    par_for(n, [&](int, int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i)
        for (int k = 0; k < npc; ++k) S.cells[(size_t)(i * npc + k)] = t[(size_t)(3 + i * stride + 1 + k)];
    });
  }
  return S;
}

// ---- the face graph of a simplicial mesh ------------------------------------------------------------------------
// The sides of a cell in the order its insertion visits them: Triangle::edges() Shape.hpp:303-305,
// Tetrahedron::faces() :590-594.
constexpr int kPart2[3][2] = {{0, 1}, {1, 2}, {2, 0}};
constexpr int kPart3[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {2, 0, 3}};

struct Side {  // one appearance of a side: ascending node ids + where it appeared (listed sides first, then cell sides)
  uint32_t a, b, c, at;
};

void build_graph(const Simplices &S, storm_hip_mesh &M) {
  const int dim = S.dim, npc = dim + 1;
  const int64_t nl = S.n_listed, nc = S.n_cells, ncf = nc * npc, nall = nl + ncf;
  if (S.n_nodes >= ((int64_t)1 << 32) - 1 || nall >= ((int64_t)1 << 32) - 1) fail("mesh too large for 32-bit side records");
  auto check = [&](const std::vector<int64_t> &v) {
    std::atomic<int> bad{0};
    par_for((int64_t)v.size(), [&](int, int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i)
        if (v[(size_t)i] < 0 || v[(size_t)i] >= S.n_nodes) bad.store(1, std::memory_order_relaxed);
    });
    if (bad.load()) fail("node index out of range (files must be zero-based, `triangle -z` / `tetgen -z`)");
  };
  check(S.listed), check(S.cells);
  // the nodes of appearance `at`, in the order they were given
  auto nodes_of = [&](int64_t at, int64_t *out) {
    if (at < nl) {
      for (int k = 0; k < dim; ++k) out[k] = S.listed[(size_t)(at * dim + k)];
    } else {
      const int64_t cell = (at - nl) / npc, p = (at - nl) % npc;
      for (int k = 0; k < dim; ++k)
        out[k] = S.cells[(size_t)(cell * npc + (dim == 2 ? kPart2[p][k] : kPart3[p][k]))];
    }
  };
  // Order the appearances by (a, b, c, at): a stable parallel counting sort into coarse buckets of the smallest node
  // (contiguous node ranges), then every bucket sorted on its own.
  PhaseTimer T;
  std::vector<Side, default_init_alloc<Side>> app((size_t)nall), sorted((size_t)nall);
  const int64_t n_buckets = std::max<int64_t>(1, std::min<int64_t>(1 << 15, S.n_nodes / 16));
  auto bucket_of = [&](uint32_t a) { return (int64_t)(((unsigned __int128)a * (uint64_t)n_buckets) / (uint64_t)std::max<int64_t>(S.n_nodes, 1)); };
  const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (nall + 65535) / 65536));
  const int64_t chunk = (nall + nth - 1) / std::max(nth, 1);
  std::vector<int64_t> hist((size_t)nth * (size_t)n_buckets, 0);
  auto on_chunks = [&](auto &&fn) {  // fn(thread, begin, end) over the SAME chunks every time
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> ex((size_t)nth);
    for (int t = 0; t < nth; ++t)
      th.emplace_back([&, t] {
        try {
          fn(t, std::min(nall, t * chunk), std::min(nall, (t + 1) * chunk));
        } catch (...) {
          ex[(size_t)t] = std::current_exception();
        }
      });
    for (auto &x : th) x.join();
    for (auto &e : ex)
      if (e) std::rethrow_exception(e);
  };
  on_chunks([&](int t, int64_t b, int64_t e) {
    int64_t *h = hist.data() + (size_t)t * (size_t)n_buckets;
    for (int64_t i = b; i < e; ++i) {
      int64_t v[3] = {0, 0, 0};
      nodes_of(i, v);
      if (v[0] > v[1]) std::swap(v[0], v[1]);
      if (dim == 3) {
        if (v[1] > v[2]) std::swap(v[1], v[2]);
        if (v[0] > v[1]) std::swap(v[0], v[1]);
      }
      if (v[0] == v[1] || (dim == 3 && v[1] == v[2])) fail("a side with a repeated node");
      app[(size_t)i] = Side{(uint32_t)v[0], (uint32_t)v[1], dim == 3 ? (uint32_t)v[2] : 0u, (uint32_t)i};
      ++h[bucket_of((uint32_t)v[0])];
    }
  });
  std::vector<int64_t> head((size_t)n_buckets + 1, 0);  // start of every coarse bucket in `sorted`
  {
    int64_t run = 0;
    for (int64_t k = 0; k < n_buckets; ++k) {
      head[(size_t)k] = run;
      for (int t = 0; t < nth; ++t) {
        const int64_t cnt = hist[(size_t)t * (size_t)n_buckets + (size_t)k];
        hist[(size_t)t * (size_t)n_buckets + (size_t)k] = run;
        run += cnt;
      }
    }
    head[(size_t)n_buckets] = run;
  }
  on_chunks([&](int t, int64_t b, int64_t e) {
    int64_t *h = hist.data() + (size_t)t * (size_t)n_buckets;
    for (int64_t i = b; i < e; ++i) sorted[(size_t)h[bucket_of(app[(size_t)i].a)]++] = app[(size_t)i];
  });
  { decltype(app)().swap(app); }
  T.lap("graph: keys + counting sort");
  par_for(n_buckets, [&](int, int64_t b, int64_t e) {
    for (int64_t k = b; k < e; ++k)
      std::sort(sorted.begin() + head[(size_t)k], sorted.begin() + head[(size_t)k + 1], [](const Side &x, const Side &y) {
        return x.a != y.a ? x.a < y.a : x.b != y.b ? x.b < y.b : x.c != y.c ? x.c < y.c : x.at < y.at;
      });
  }, 64);
  T.lap("graph: bucket sorts");
  // a side = a run of equal (a, b, c); its id = the rank of its first appearance among all first appearances
  std::vector<unsigned char> is_first((size_t)nall, 0);
  uvec<uint32_t> id_at((size_t)nall);  // side id of a first appearance
  par_for(nall, [&](int, int64_t b, int64_t e) {
    for (int64_t i = b; i < e; ++i) {
      const Side &s = sorted[(size_t)i];
      const bool first = i == 0 || sorted[(size_t)i - 1].a != s.a || sorted[(size_t)i - 1].b != s.b || sorted[(size_t)i - 1].c != s.c;
      if (first) is_first[(size_t)s.at] = 1;
    }
  });
  int64_t n_sides = 0;
  for (int64_t i = 0; i < nall; ++i) {
    id_at[(size_t)i] = (uint32_t)n_sides;
    n_sides += is_first[(size_t)i];
  }
  for (int64_t i = 0; i < nl; ++i)
    if (!is_first[(size_t)i]) fail("a side is listed twice");
  T.lap("graph: side ids");
  uvec<int64_t> first((size_t)n_sides), second((size_t)n_sides), first_at((size_t)n_sides), label((size_t)n_sides);  // (every side is one run below: filled there)
  std::atomic<int> err{0};
  // (runs do not straddle buckets: a thread takes whole buckets)
  par_for(n_buckets, [&](int, int64_t vb, int64_t ve) {
    for (int64_t i = head[(size_t)vb]; i < head[(size_t)ve];) {
      int64_t j = i + 1;
      while (j < head[(size_t)ve] && sorted[(size_t)j].a == sorted[(size_t)i].a && sorted[(size_t)j].b == sorted[(size_t)i].b &&
             sorted[(size_t)j].c == sorted[(size_t)i].c)
        ++j;
      const int64_t sid = id_at[(size_t)sorted[(size_t)i].at];
      first_at[(size_t)sid] = sorted[(size_t)i].at;
      first[(size_t)sid] = second[(size_t)sid] = -1, label[(size_t)sid] = 0;
      int owners = 0, parity = 0;
      for (int64_t k = i; k < j; ++k) {
        const int64_t at = sorted[(size_t)k].at;
        if (at < nl) continue;  // (a second listed appearance was refused above)
        const int64_t cell = (at - nl) / npc;
        if (owners == 0) first[(size_t)sid] = cell;
        else second[(size_t)sid] = cell;
        ++owners;
        int64_t v[3] = {0, 0, 0};
        nodes_of(at, v);
        const int inv = dim == 2 ? (v[0] > v[1]) : ((v[0] > v[1]) + (v[0] > v[2]) + (v[1] > v[2]));
        parity += (inv & 1) ? -1 : 1;
      }
      if (owners == 0 || owners > 2) err.store(1, std::memory_order_relaxed);  // STORM_ABORT, MeshUnstructured.hpp:550
      else if (owners == 2 && parity != 0) err.store(2, std::memory_order_relaxed);  // STORM_ENSURE, :546-548
      i = j;
    }
  }, 64);
  if (err.load() == 1) fail("Invalid number of the face cells!");
  if (err.load() == 2) fail("Face has two adjacent cells, but the second cell cannot be the outer one!");
  T.lap("graph: owners");
  for (int64_t i = 0; i < nl && !S.label.empty(); ++i) label[(size_t)id_at[(size_t)i]] = S.label[(size_t)i];
  // interior sides (label 0) in side order, the others as boundary faces
  uvec<int64_t> slot((size_t)n_sides);
  int64_t nf = 0, nb = 0;
  for (int64_t s = 0; s < n_sides; ++s) {
    if (label[(size_t)s] == 0) {
      if (second[(size_t)s] < 0) fail("an unlabelled side has a single adjacent cell");
      slot[(size_t)s] = nf++;
    } else slot[(size_t)s] = nb++;
  }
  T.lap("graph: labels + slots");
  M.dim = dim, M.n_cells = nc, M.n_halo = 0;
  M.inner.resize((size_t)nf), M.outer.resize((size_t)nf), M.area.resize((size_t)nf);
  M.b_cell.resize((size_t)nb), M.b_area.resize((size_t)nb), M.b_center.resize((size_t)(nb * dim));
  M.center.resize((size_t)(nc * dim)), M.volume.resize((size_t)nc);
  const double *P = S.pos.data();
  par_for(n_sides, [&](int, int64_t b, int64_t e) {
    for (int64_t s = b; s < e; ++s) {
      int64_t v[3] = {0, 0, 0};
      nodes_of(first_at[(size_t)s], v);  // the nodes the side was inserted with
      double a, mid[3] = {0, 0, 0};
      if (dim == 2) {
        const double *q0 = P + v[0] * 2, *q1 = P + v[1] * 2;
        const double ex = q1[0] - q0[0], ey = q1[1] - q0[1];
        a = std::sqrt(0.0 + ex * ex + ey * ey);  // Shape.hpp:242-247
        mid[0] = 0.5 * (q0[0] + q1[0]), mid[1] = 0.5 * (q0[1] + q1[1]);
      } else {
        const double *q1 = P + v[0] * 3, *q2 = P + v[1] * 3, *q3 = P + v[2] * 3;
        const double u[3] = {q2[0] - q1[0], q2[1] - q1[1], q2[2] - q1[2]}, w[3] = {q3[0] - q1[0], q3[1] - q1[1], q3[2] - q1[2]};
        const double cx = u[1] * w[2] - u[2] * w[1], cy = u[2] * w[0] - u[0] * w[2], cz = u[0] * w[1] - u[1] * w[0];
        a = std::sqrt(((0.0 + cx * cx) + cy * cy) + cz * cz) / 2.0;  // Shape.hpp:321
        for (int d = 0; d < 3; ++d) mid[d] = ((q1[d] + q2[d]) + q3[d]) / 3.0;
      }
      const int64_t k = slot[(size_t)s];
      if (label[(size_t)s] == 0) M.inner[(size_t)k] = first[(size_t)s], M.outer[(size_t)k] = second[(size_t)s], M.area[(size_t)k] = a;
      else {
        M.b_cell[(size_t)k] = first[(size_t)s], M.b_area[(size_t)k] = a;
        for (int d = 0; d < dim; ++d) M.b_center[(size_t)(k * dim + d)] = mid[d];
      }
    }
  });
  par_for(nc, [&](int, int64_t b, int64_t e) {
    for (int64_t i = b; i < e; ++i) {
      const int64_t *cn = S.cells.data() + i * npc;
      if (dim == 2) {
        const double *p0 = P + cn[0] * 2, *p1 = P + cn[1] * 2, *p2 = P + cn[2] * 2;
        for (int d = 0; d < 2; ++d) M.center[(size_t)(i * 2 + d)] = ((p0[d] + p1[d]) + p2[d]) / 3.0;  // Shape.hpp:155-167
        const double d0x = p1[0] - p0[0], d0y = p1[1] - p0[1], d1x = p2[0] - p0[0], d1y = p2[1] - p0[1];
        M.volume[(size_t)i] = 0.5 * std::fabs(d0x * d1y - d0y * d1x);  // Shape.hpp:309-321
      } else {
        const double *p0 = P + cn[0] * 3, *p1 = P + cn[1] * 3, *p2 = P + cn[2] * 3, *p3 = P + cn[3] * 3;
        for (int d = 0; d < 3; ++d) M.center[(size_t)(i * 3 + d)] = (((p0[d] + p1[d]) + p2[d]) + p3[d]) / 4.0;  // Shape.hpp:601-606
        const double a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, bb[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]},
                     c[3] = {p3[0] - p0[0], p3[1] - p0[1], p3[2] - p0[2]};
        // (the reference has no volume(Tetrahedron): this build's completion, see io_tetgen.py)
        const double det = (a[0] * (bb[1] * c[2] - bb[2] * c[1]) - a[1] * (bb[0] * c[2] - bb[2] * c[0])) + a[2] * (bb[0] * c[1] - bb[1] * c[0]);
        M.volume[(size_t)i] = std::fabs(det) / 6.0;
      }
    }
  });
  for (int64_t i = 0; i < nc; ++i)
    if (!(M.volume[(size_t)i] > 0)) fail("cell %lld has no volume", (long long)i);
  T.lap("graph: geometry");
}

// ---- writer -----------------------------------------------------------------------------------------------------
void write_rows(const std::string &path, const std::string &header, int64_t n, int per_row_max,
                const std::function<char *(int64_t, char *)> &row) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) fail("Cannot open the file '%s' for writing!", path.c_str());
  bool ok = fwrite(header.data(), 1, header.size(), f) == header.size();
  const int64_t slab = 1 << 20;
  const int nt = host_threads();
  std::vector<std::string> buf((size_t)nt);
  for (int64_t s0 = 0; s0 < n && ok; s0 += slab * nt) {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
      const int64_t b = std::min(n, s0 + t * slab), e = std::min(n, b + slab);
      buf[(size_t)t].clear();
      if (b >= e) continue;
      th.emplace_back([&, t, b, e] {
        std::string &o = buf[(size_t)t];
        o.resize((size_t)((e - b) * per_row_max));
        char *p = &o[0];
        for (int64_t i = b; i < e; ++i) p = row(i, p);
        o.resize((size_t)(p - o.data()));
      });
    }
    for (auto &x : th) x.join();
    for (int t = 0; t < nt && ok; ++t) ok = fwrite(buf[(size_t)t].data(), 1, buf[(size_t)t].size(), f) == buf[(size_t)t].size();
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) fail("Cannot write the file '%s'!", path.c_str());
}
inline char *put_int(char *p, int64_t v) { return std::to_chars(p, p + 24, v).ptr; }
inline char *put_real(char *p, double v) { return std::to_chars(p, p + 32, v).ptr; }  // shortest form that round-trips

// ---- partition --------------------------------------------------------------------------------------------------
// Recursive coordinate bisection: the longest axis of the part's bounding box, cut at the k-th smallest of
// (coordinate, cell id), k proportional to the ranks on either side.  partition.py: rcb_partition.
void rcb(int dim, const double *c, std::vector<int64_t> &idx, int64_t b, int64_t e, int first, int count, int32_t *part) {
  if (count == 1) {
    for (int64_t i = b; i < e; ++i) part[idx[(size_t)i]] = first;
    return;
  }
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = b; i < e; ++i)
    for (int d = 0; d < dim; ++d) {
      const double v = c[idx[(size_t)i] * dim + d];
      lo[d] = std::min(lo[d], v), hi[d] = std::max(hi[d], v);
    }
  int axis = 0;
  for (int d = 1; d < dim; ++d)
    if (hi[d] - lo[d] > hi[axis] - lo[axis]) axis = d;
  const int left = count / 2;
  const int64_t k = ((e - b) * left) / count;
  auto less = [&](int64_t x, int64_t y) {
    const double vx = c[x * dim + axis], vy = c[y * dim + axis];
    return vx != vy ? vx < vy : x < y;
  };
  std::nth_element(idx.begin() + b, idx.begin() + b + k, idx.begin() + e, less);
  if (count >= 4 && e - b > (1 << 16)) {  // the two halves are independent
    std::thread t([&] { rcb(dim, c, idx, b, b + k, first, left, part); });
    rcb(dim, c, idx, b + k, e, first + left, count - left, part);
    t.join();
  } else {
    rcb(dim, c, idx, b, b + k, first, left, part);
    rcb(dim, c, idx, b + k, e, first + left, count - left, part);
  }
}

void local_graph(const storm_hip_mesh &G, const int32_t *part, int rank, storm_hip_mesh &L) {
  if (G.n_halo != 0) fail("mesh_partition: the mesh is a rank's local mesh already");
  const int dim = G.dim;
  const int64_t n = G.n_cells, nf = (int64_t)G.inner.size(), nb = (int64_t)G.b_cell.size();
  std::vector<int64_t> loc((size_t)n, -1);
  std::vector<int64_t> gid;
  for (int64_t i = 0; i < n; ++i)
    if (part[i] == rank) loc[(size_t)i] = (int64_t)gid.size(), gid.push_back(i);
  const int64_t n_own = (int64_t)gid.size();
  // faces touching an owned cell, in their order; the cells on their other side = the halo
  std::vector<int64_t> fsel, halo;
  for (int64_t f = 0; f < nf; ++f) {
    const int64_t i = G.inner[(size_t)f], o = G.outer[(size_t)f];
    const bool oi = part[i] == rank, oo = part[o] == rank;
    if (!oi && !oo) continue;
    fsel.push_back(f);
    if (!oi) halo.push_back(i);
    if (!oo) halo.push_back(o);
  }
  // by owner, then global id -- the id the send lists are sorted by too, i.e. through a permuted mesh's own map
  auto id_of = [&](int64_t x) { return G.global_id.empty() ? x : G.global_id[(size_t)x]; };
  std::sort(halo.begin(), halo.end(), [&](int64_t x, int64_t y) { return part[x] != part[y] ? part[x] < part[y] : id_of(x) < id_of(y); });
  halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
  for (int64_t h = 0; h < (int64_t)halo.size(); ++h) loc[(size_t)halo[(size_t)h]] = n_own + h;
  gid.insert(gid.end(), halo.begin(), halo.end());
  const int64_t nt = (int64_t)gid.size();
  L.dim = dim, L.n_cells = n_own, L.n_halo = (int64_t)halo.size();
  L.inner.resize(fsel.size()), L.outer.resize(fsel.size()), L.area.resize(fsel.size());
  for (size_t k = 0; k < fsel.size(); ++k) {
    const int64_t f = fsel[k];
    L.inner[k] = loc[(size_t)G.inner[(size_t)f]], L.outer[k] = loc[(size_t)G.outer[(size_t)f]], L.area[k] = G.area[(size_t)f];
  }
  L.center.resize((size_t)(nt * dim)), L.volume.resize((size_t)nt);
  for (int64_t i = 0; i < nt; ++i) {
    for (int d = 0; d < dim; ++d) L.center[(size_t)(i * dim + d)] = G.center[(size_t)(gid[(size_t)i] * dim + d)];
    L.volume[(size_t)i] = G.volume[(size_t)gid[(size_t)i]];
  }
  for (int64_t k = 0; k < nb; ++k)
    if (part[G.b_cell[(size_t)k]] == rank) {
      L.b_cell.push_back(loc[(size_t)G.b_cell[(size_t)k]]), L.b_area.push_back(G.b_area[(size_t)k]);
      for (int d = 0; d < dim; ++d) L.b_center.push_back(G.b_center[(size_t)(k * dim + d)]);
    }
  // global ids: through the global mesh's own map when it has one (a permuted mesh)
  L.global_id.resize((size_t)nt);
  for (int64_t i = 0; i < nt; ++i) L.global_id[(size_t)i] = G.global_id.empty() ? gid[(size_t)i] : G.global_id[(size_t)gid[(size_t)i]];
  L.halo_owner.resize(halo.size());
  for (size_t h = 0; h < halo.size(); ++h) L.halo_owner[h] = part[halo[h]];
}

// partition.py: halo_plan.  Neighbours ascending; a neighbour's halo group is the run of its cells in the halo tail;
// the send list towards it = the owned cells that share a face with one of its cells, in ascending global id.
void halo_plan(storm_hip_mesh &L, int rank) {
  const int64_t n = L.n_cells, nh = L.n_halo;
  L.nbr_rank.clear(), L.send_idx.clear();
  L.send_ptr.assign(1, 0), L.recv_ptr.assign(1, 0);
  if (nh == 0) return;
  for (int64_t h = 0; h < nh; ++h) {
    const int32_t o = L.halo_owner[(size_t)h];
    if (o < 0 || o == rank) fail("halo_plan: halo cell %lld has owner %d", (long long)h, (int)o);
    if (h > 0 && o < L.halo_owner[(size_t)h - 1]) fail("halo cells must be grouped by ascending owner rank");
    if (h > 0 && o == L.halo_owner[(size_t)h - 1] && !(L.global_id[(size_t)(n + h)] > L.global_id[(size_t)(n + h - 1)]))
      fail("halo group not in ascending global id");
    if (L.nbr_rank.empty() || L.nbr_rank.back() != o) {
      if (!L.nbr_rank.empty()) L.recv_ptr.push_back(h);
      L.nbr_rank.push_back(o);
    }
  }
  L.recv_ptr.push_back(nh);
  std::vector<std::vector<int64_t>> send(L.nbr_rank.size());
  auto q_of = [&](int32_t owner) { return (size_t)(std::lower_bound(L.nbr_rank.begin(), L.nbr_rank.end(), owner) - L.nbr_rank.begin()); };
  for (size_t f = 0; f < L.inner.size(); ++f) {
    const int64_t i = L.inner[f], o = L.outer[f];
    if (i < n && o >= n) send[q_of(L.halo_owner[(size_t)(o - n)])].push_back(i);
    else if (o < n && i >= n) send[q_of(L.halo_owner[(size_t)(i - n)])].push_back(o);
  }
  for (auto &s : send) {
    std::sort(s.begin(), s.end(), [&](int64_t x, int64_t y) {
      return L.global_id[(size_t)x] != L.global_id[(size_t)y] ? L.global_id[(size_t)x] < L.global_id[(size_t)y] : x < y;
    });
    s.erase(std::unique(s.begin(), s.end()), s.end());
    L.send_idx.insert(L.send_idx.end(), s.begin(), s.end());
    L.send_ptr.push_back((int64_t)L.send_idx.size());
  }
}

template <class F>
int guarded(F &&fn) {  // nothing may leave an extern "C" entry point
  try {
    return fn();
  } catch (const MeshError &e) {
    STORM_FAIL(STORM_HIP_E_INVALID, "%s", e.what());
  } catch (const std::bad_alloc &) {
    STORM_FAIL(STORM_HIP_E_ALLOC, "out of host memory");
  } catch (const std::exception &e) {
    STORM_FAIL(STORM_HIP_E_INVALID, "%s", e.what());
  }
}

}  // namespace
}  // namespace storm

using namespace storm;

extern "C" int storm_hip_mesh_create(int32_t dim, int64_t n_cells, int64_t n_halo, int64_t n_faces, const int64_t *inner,
                                      const int64_t *outer, const double *area, const double *center, const double *volume,
                                      int64_t n_bfaces, const int64_t *b_cell, const double *b_area, const double *b_center,
                                      const int64_t *global_id, const int32_t *halo_owner, storm_hip_mesh **out) {
  STORM_REQUIRE(out != nullptr && dim >= 1 && dim <= 3 && n_cells >= 0 && n_halo >= 0 && n_faces >= 0 && n_bfaces >= 0,
                "mesh_create: bad argument");
  STORM_REQUIRE((n_faces == 0 || (inner && outer && area)) && (n_cells + n_halo == 0 || (center && volume)) &&
                    (n_bfaces == 0 || (b_cell && b_area && b_center)) && (n_halo == 0 || (global_id && halo_owner)),
                "mesh_create: null array");
  return guarded([&]() -> int {
    const int64_t nt = n_cells + n_halo;
    for (int64_t f = 0; f < n_faces; ++f)
      STORM_REQUIRE(inner[f] >= 0 && inner[f] < nt && outer[f] >= 0 && outer[f] < nt && inner[f] != outer[f],
                    "mesh_create: face %lld joins cells %lld and %lld of %lld", (long long)f, (long long)inner[f], (long long)outer[f], (long long)nt);
    for (int64_t k = 0; k < n_bfaces; ++k)
      STORM_REQUIRE(b_cell[k] >= 0 && b_cell[k] < n_cells, "mesh_create: boundary face %lld on cell %lld", (long long)k, (long long)b_cell[k]);
    auto m = std::make_unique<storm_hip_mesh>();
    m->dim = dim, m->n_cells = n_cells, m->n_halo = n_halo;
    m->inner.assign(inner, inner + n_faces), m->outer.assign(outer, outer + n_faces), m->area.assign(area, area + n_faces);
    m->center.assign(center, center + nt * dim), m->volume.assign(volume, volume + nt);
    m->b_cell.assign(b_cell, b_cell + n_bfaces), m->b_area.assign(b_area, b_area + n_bfaces);
    m->b_center.assign(b_center, b_center + n_bfaces * dim);
    if (global_id) m->global_id.assign(global_id, global_id + nt);
    if (halo_owner) m->halo_owner.assign(halo_owner, halo_owner + n_halo);
    m->send_ptr.assign(1, 0), m->recv_ptr.assign(1, 0);
    *out = m.release();
    return (int)STORM_HIP_OK;
  });
}

extern "C" int storm_hip_mesh_from_simplices(int32_t dim, int64_t n_nodes, const double *pos, int64_t n_listed,
                                              const int64_t *listed, const int64_t *listed_label, int64_t n_cells,
                                              const int64_t *cells, storm_hip_mesh **out) {
  STORM_REQUIRE(out != nullptr && (dim == 2 || dim == 3) && n_nodes >= 0 && n_listed >= 0 && n_cells >= 0 &&
                    (pos || n_nodes == 0) && (listed || n_listed == 0) && (cells || n_cells == 0),
                "mesh_from_simplices: bad argument");
  return guarded([&]() -> int {
    Simplices S;
    S.dim = dim, S.n_nodes = n_nodes, S.n_listed = n_listed, S.n_cells = n_cells;
    S.pos.assign(pos, pos + n_nodes * dim), S.listed.assign(listed, listed + n_listed * dim);
    if (listed_label) S.label.assign(listed_label, listed_label + n_listed);
    S.cells.assign(cells, cells + n_cells * (dim + 1));
    auto m = std::make_unique<storm_hip_mesh>();
    build_graph(S, *m);
    m->send_ptr.assign(1, 0), m->recv_ptr.assign(1, 0);
    *out = m.release();
    return (int)STORM_HIP_OK;
  });
}

extern "C" int storm_hip_mesh_read_tetgen(const char *prefix, int32_t dim, storm_hip_mesh **out) {
  STORM_REQUIRE(prefix != nullptr && out != nullptr && (dim == 0 || dim == 2 || dim == 3), "mesh_read_tetgen: bad argument");
  return guarded([&]() -> int {
    const Simplices S = read_files(prefix, dim);
    auto m = std::make_unique<storm_hip_mesh>();
    build_graph(S, *m);
    m->send_ptr.assign(1, 0), m->recv_ptr.assign(1, 0);
    *out = m.release();
    return (int)STORM_HIP_OK;
  });
}

extern "C" int storm_hip_mesh_write_tetgen(const char *prefix, int32_t dim, int64_t n_nodes, const double *pos,
                                            int64_t n_listed, const int64_t *listed, const int64_t *listed_label,
                                            int64_t n_cells, const int64_t *cells) {
  STORM_REQUIRE(prefix != nullptr && (dim == 2 || dim == 3) && n_nodes >= 0 && n_listed >= 0 && n_cells >= 0 &&
                    (pos || n_nodes == 0) && (listed || n_listed == 0) && (cells || n_cells == 0),
                "mesh_write_tetgen: bad argument");
  return guarded([&]() -> int {
    std::string p(prefix);
    if (!p.empty() && p.back() == '.') p.pop_back();
    const std::string note = "# written by storm_hip_mesh_write_tetgen (zero-based ids)\n";
    write_rows(p + ".node", note + std::to_string(n_nodes) + " " + std::to_string(dim) + " 0 0\n", n_nodes, 24 + 33 * dim,
               [&](int64_t i, char *o) {
                 o = put_int(o, i);
                 for (int d = 0; d < dim; ++d) *o++ = ' ', o = put_real(o, pos[i * dim + d]);
                 *o++ = '\n';
                 return o;
               });
    auto side_rows = [&](const std::string &path, int64_t n) {
      write_rows(path, note + std::to_string(n) + " 1\n", n, 24 * (dim + 2), [&](int64_t i, char *o) {
        o = put_int(o, i);
        for (int k = 0; k < dim; ++k) *o++ = ' ', o = put_int(o, listed[i * dim + k]);
        *o++ = ' ', o = put_int(o, listed_label ? listed_label[i] : 0);
        *o++ = '\n';
        return o;
      });
    };
    if (dim == 2) side_rows(p + ".edge", n_listed);
    else {
      // TetGen "may not generate all the edges" (IoTetgen.hpp:219-221): none are listed
      write_rows(p + ".edge", note + "0 1\n", 0, 1, [](int64_t, char *o) { return o; });
      side_rows(p + ".face", n_listed);
    }
    write_rows(p + ".ele", note + std::to_string(n_cells) + " " + std::to_string(dim + 1) + " 0\n", n_cells, 24 * (dim + 2),
               [&](int64_t i, char *o) {
                 o = put_int(o, i);
                 for (int k = 0; k <= dim; ++k) *o++ = ' ', o = put_int(o, cells[i * (dim + 1) + k]);
                 *o++ = '\n';
                 return o;
               });
    return (int)STORM_HIP_OK;
  });
}

extern "C" int storm_hip_mesh_get_view(const storm_hip_mesh *m, storm_hip_mesh_view *v) {
  STORM_REQUIRE(m != nullptr && v != nullptr, "mesh_get_view: null argument");
  v->dim = m->dim, v->n_nbrs = (int32_t)m->nbr_rank.size();
  v->n_cells = m->n_cells, v->n_halo = m->n_halo, v->n_faces = (int64_t)m->inner.size(), v->n_bfaces = (int64_t)m->b_cell.size();
  v->inner = m->inner.data(), v->outer = m->outer.data(), v->area = m->area.data();
  v->center = m->center.data(), v->volume = m->volume.data();
  v->b_cell = m->b_cell.data(), v->b_area = m->b_area.data(), v->b_center = m->b_center.data();
  v->global_id = m->global_id.empty() ? nullptr : m->global_id.data();
  v->halo_owner = m->halo_owner.empty() ? nullptr : m->halo_owner.data();
  v->nbr_rank = m->nbr_rank.data(), v->send_ptr = m->send_ptr.data(), v->send_idx = m->send_idx.data(), v->recv_ptr = m->recv_ptr.data();
  return STORM_HIP_OK;
}

extern "C" int storm_hip_mesh_permute_cells(storm_hip_mesh *m, const int64_t *order) {
  STORM_REQUIRE(m != nullptr && (order != nullptr || m->n_cells == 0), "mesh_permute_cells: null argument");
  return guarded([&]() -> int {
    const int64_t n = m->n_cells, nt = n + m->n_halo;
    const int dim = m->dim;
    std::vector<int64_t> inv((size_t)nt, -1);
    for (int64_t i = 0; i < n; ++i) {
      STORM_REQUIRE(order[i] >= 0 && order[i] < n && inv[(size_t)order[i]] < 0, "mesh_permute_cells: not a permutation of the owned cells");
      inv[(size_t)order[i]] = i;
    }
    for (int64_t i = n; i < nt; ++i) inv[(size_t)i] = i;  // halo cells keep their slot
    par_for((int64_t)m->inner.size(), [&](int, int64_t b, int64_t e) {
      for (int64_t f = b; f < e; ++f) m->inner[(size_t)f] = inv[(size_t)m->inner[(size_t)f]], m->outer[(size_t)f] = inv[(size_t)m->outer[(size_t)f]];
    });
    for (auto &c : m->b_cell) c = inv[(size_t)c];
    for (auto &c : m->send_idx) c = inv[(size_t)c];
    uvec<double> center(m->center.size()), volume(m->volume.size());
    std::vector<int64_t> gid(m->global_id.size());
    par_for(nt, [&](int, int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i) {
        const int64_t o = i < n ? order[i] : i;
        for (int d = 0; d < dim; ++d) center[(size_t)(i * dim + d)] = m->center[(size_t)(o * dim + d)];
        volume[(size_t)i] = m->volume[(size_t)o];
        if (!gid.empty()) gid[(size_t)i] = m->global_id[(size_t)o];
      }
    });
    if (gid.empty()) {  // a permuted single-rank mesh remembers where its cells came from
      gid.resize((size_t)nt);
      for (int64_t i = 0; i < nt; ++i) gid[(size_t)i] = i < n ? order[i] : i;
    }
    m->center.swap(center), m->volume.swap(volume), m->global_id.swap(gid);
    return (int)STORM_HIP_OK;
  });
}

extern "C" int storm_hip_partition_rcb(int32_t dim, int64_t n_cells, const double *centers, int32_t n_parts, int32_t *part_out) {
  STORM_REQUIRE(dim >= 1 && dim <= 3 && n_cells >= 0 && n_parts >= 1 && (centers || n_cells == 0) && (part_out || n_cells == 0),
                "partition_rcb: bad argument");
  return guarded([&]() -> int {
    for (int64_t i = 0; i < n_cells * dim; ++i) STORM_REQUIRE(std::isfinite(centers[i]), "partition_rcb: non-finite cell centre");
    std::vector<int64_t> idx((size_t)n_cells);
    std::iota(idx.begin(), idx.end(), (int64_t)0);
    rcb(dim, centers, idx, 0, n_cells, 0, n_parts, part_out);
    return (int)STORM_HIP_OK;
  });
}

extern "C" int storm_hip_partition_slabs(int32_t dim, int64_t n_cells, const double *centers, int32_t axis, int32_t n_parts,
                                          int32_t *part_out) {
  STORM_REQUIRE(dim >= 1 && dim <= 3 && axis >= 0 && axis < dim && n_cells >= 0 && n_parts >= 1 && (centers || n_cells == 0) &&
                    (part_out || n_cells == 0),
                "partition_slabs: bad argument");
  return guarded([&]() -> int {
    // contiguous ranges of the cells ordered by (coordinate, cell id), as even as possible: rank r gets the cells of
    // positions [r n / P, (r + 1) n / P).  A structured box of nz = k P planes gets k whole planes per rank.
    for (int64_t i = 0; i < n_cells; ++i) STORM_REQUIRE(std::isfinite(centers[i * dim + axis]), "partition_slabs: non-finite cell centre");
    std::vector<int64_t> idx((size_t)n_cells);
    std::iota(idx.begin(), idx.end(), (int64_t)0);
    auto less = [&](int64_t x, int64_t y) {
      const double vx = centers[x * dim + axis], vy = centers[y * dim + axis];
      return vx != vy ? vx < vy : x < y;
    };
    int64_t b = 0;
    for (int r = 0; r < n_parts; ++r) {
      const int64_t e = (int64_t)(((__int128)n_cells * (r + 1)) / n_parts);
      if (r + 1 < n_parts && e > b && e < n_cells) std::nth_element(idx.begin() + b, idx.begin() + e, idx.end(), less);
      for (int64_t i = b; i < e; ++i) part_out[idx[(size_t)i]] = r;
      b = e;
    }
    return (int)STORM_HIP_OK;
  });
}

extern "C" int storm_hip_mesh_partition(const storm_hip_mesh *global, const int32_t *part, int32_t n_parts, int32_t rank,
                                         storm_hip_mesh **out) {
  STORM_REQUIRE(global != nullptr && out != nullptr && (part != nullptr || global->n_cells == 0) && n_parts >= 1 && rank >= 0 &&
                    rank < n_parts,
                "mesh_partition: bad argument");
  return guarded([&]() -> int {
    for (int64_t i = 0; i < global->n_cells; ++i)
      STORM_REQUIRE(part[i] >= 0 && part[i] < n_parts, "mesh_partition: cell %lld assigned to rank %d of %d", (long long)i, (int)part[i], (int)n_parts);
    auto m = std::make_unique<storm_hip_mesh>();
    local_graph(*global, part, rank, *m);
    halo_plan(*m, rank);
    *out = m.release();
    return (int)STORM_HIP_OK;
  });
}

extern "C" int storm_hip_mesh_halo_plan(storm_hip_mesh *m, int32_t rank) {
  STORM_REQUIRE(m != nullptr, "mesh_halo_plan: null mesh");
  STORM_REQUIRE(m->n_halo == 0 || ((int64_t)m->global_id.size() == m->n_cells + m->n_halo && (int64_t)m->halo_owner.size() == m->n_halo),
                "mesh_halo_plan: the mesh has halo cells but no global ids / owners");
  return guarded([&]() -> int {
    halo_plan(*m, rank);
    return (int)STORM_HIP_OK;
  });
}

extern "C" int storm_hip_op_create_from_mesh_object(storm_hip_ctx *ctx, const storm_hip_mesh *m, storm_hip_op **out) {
  STORM_REQUIRE(ctx != nullptr && m != nullptr && out != nullptr, "op_create_from_mesh_object: null argument");
  STORM_TRY(storm_hip_op_create_from_mesh(ctx, m->n_cells, m->n_halo, m->dim, (int64_t)m->inner.size(), m->inner.data(), m->outer.data(),
                                          m->area.data(), m->center.data(), (int64_t)m->b_cell.size(), m->b_cell.data(),
                                          m->b_area.data(), m->b_center.data(), m->volume.data(), out));
  if (!m->nbr_rank.empty()) {
    const int st = storm_hip_op_set_halo(*out, (int)m->nbr_rank.size(), m->nbr_rank.data(), m->send_ptr.data(), m->send_idx.data(),
                                         m->recv_ptr.data());
    if (st != STORM_HIP_OK) {
      (void)storm_hip_op_destroy(*out);
      *out = nullptr;
      return st;
    }
  }
  return STORM_HIP_OK;
}

extern "C" int storm_hip_mesh_destroy(storm_hip_mesh *m) {
  delete m;
  return STORM_HIP_OK;
}
