// Host-side build of the operator (records, dictionaries, slice lists) and the storm_hip_op_create_* entry points.
// Record layouts: the header of spmv.hip.
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
#include "spmv_device.hpp"

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

// (the records in a slot of their vectors' arena -- option pack_arena -- gave no gain: profiles/r05v_pack_arena.jsonl,
//  profiles/experiments/r08_pruned_experiments.patch)
static int upload_pack(storm_hip_op *op, const std::vector<char> &src, int64_t *bytes) { return upload(&op->d_pack, src, bytes); }

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
  op->xcd_group_sell = (int)c->opt_spmv_xcd_remap_sell;
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
  }
  if (pr) {
    // format 3 (or 4) it is: a "slice" of this operator is a 128-row group
    op->pair = cn ? 2 : 1;
    op->bnd_width = pair_width;
    if (cn) pair_width = canon_len;
    op->n_slices = n_groups;
    op->uniform_width = pair_width;
    op->ell_slots = n_groups * 2 * kWave * pair_width;
    std::vector<int64_t> goff((size_t)n_groups + 1);
    for (int64_t s = 0; s <= n_groups; ++s)
      goff[(size_t)s] = s * (cn ? kCanonRecBytes : kPairRecBytes);
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
        (st3 = upload(&op->d_slice_off, goff, &bytes3)) || (st3 = upload_pack(op, pair_pack, &bytes3)) ||
        (st3 = upload(&op->d_tail_row, no_i, &bytes3)) || (st3 = upload(&op->d_tail_ptr, one_zero, &bytes3)) ||
        (st3 = upload(&op->d_tail_col, no_i, &bytes3)) || (st3 = upload(&op->d_tail_val, no_d, &bytes3))) {
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

}  // extern "C"
