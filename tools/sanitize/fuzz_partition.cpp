// ASan / UBSan: permutations (valid and not), partitions with empty ranks, more ranks than cells, halo plans pairing up.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <map>
#include <algorithm>
#include <string>
#include "storm_hip.h"
#define CK(x) do { int s_ = (x); if (s_) { printf("%s -> %d: %s\n", #x, s_, storm_hip_last_error()); return 1; } } while (0)
int main(int argc, char **argv) {
  const std::string root = argc > 1 ? argv[1] : ".";
  std::mt19937_64 rng(7);
  storm_hip_mesh *m = nullptr;
  CK(storm_hip_mesh_read_tetgen((root + "/tests/golden/mesh/square_nb.1.").c_str(), 2, &m));
  storm_hip_mesh_view v;
  CK(storm_hip_mesh_get_view(m, &v));
  const int64_t n = v.n_cells;
  // invalid permutations
  std::vector<int64_t> p((size_t)n);
  for (int64_t i = 0; i < n; ++i) p[(size_t)i] = i;
  p[5] = p[6];
  printf("duplicate: %d (%s)\n", storm_hip_mesh_permute_cells(m, p.data()), storm_hip_last_error());
  p[5] = -1;
  printf("negative: %d (%s)\n", storm_hip_mesh_permute_cells(m, p.data()), storm_hip_last_error());
  p[5] = n;
  printf("too large: %d (%s)\n", storm_hip_mesh_permute_cells(m, p.data()), storm_hip_last_error());
  for (int trial = 0; trial < 40; ++trial) {
    for (int64_t i = 0; i < n; ++i) p[(size_t)i] = i;
    std::shuffle(p.begin(), p.end(), rng);
    if (trial % 3 == 0) CK(storm_hip_mesh_permute_cells(m, p.data()));
    CK(storm_hip_mesh_get_view(m, &v));
    const int P = trial < 6 ? (int)(1 + rng() % 3) : (int)(1 + rng() % 40);
    std::vector<int32_t> part((size_t)n);
    const int mode = trial % 4;
    for (int64_t i = 0; i < n; ++i)
      part[(size_t)i] = mode == 0 ? (int32_t)(rng() % P) : mode == 1 ? (int32_t)((rng() % P) / 2 * 2 % P) : mode == 2 ? (int32_t)(i * P / n) : (int32_t)(v.center[2 * i] > 1.0 ? P - 1 : 0);
    if (mode == 3 && trial % 8 == 3) CK(storm_hip_partition_rcb(2, n, v.center, P, part.data()));
    std::vector<storm_hip_mesh *> loc((size_t)P, nullptr);
    std::vector<storm_hip_mesh_view> lv((size_t)P);
    int64_t owned = 0;
    for (int r = 0; r < P; ++r) {
      CK(storm_hip_mesh_partition(m, part.data(), P, r, &loc[(size_t)r]));
      CK(storm_hip_mesh_get_view(loc[(size_t)r], &lv[(size_t)r]));
      owned += lv[(size_t)r].n_cells;
    }
    if (owned != n) { printf("owned %lld != %lld\n", (long long)owned, (long long)n); return 2; }
    // what r sends to q is what q expects from r
    for (int r = 0; r < P; ++r) {
      const auto &a = lv[(size_t)r];
      for (int qi = 0; qi < a.n_nbrs; ++qi) {
        const int q = a.nbr_rank[qi];
        const auto &b = lv[(size_t)q];
        int j = -1;
        for (int k = 0; k < b.n_nbrs; ++k) if (b.nbr_rank[k] == r) j = k;
        if (j < 0) { printf("rank %d sends to %d which does not expect it\n", r, q); return 3; }
        const int64_t ns = a.send_ptr[qi + 1] - a.send_ptr[qi], nr = b.recv_ptr[j + 1] - b.recv_ptr[j];
        if (ns != nr) { printf("count mismatch %d -> %d\n", r, q); return 4; }
        for (int64_t k = 0; k < ns; ++k)
          if (a.global_id[a.send_idx[a.send_ptr[qi] + k]] != b.global_id[b.n_cells + b.recv_ptr[j] + k]) { printf("order mismatch %d -> %d\n", r, q); return 5; }
      }
    }
    for (auto *l : loc) CK(storm_hip_mesh_destroy(l));
    // bad partitions
    part[0] = P;
    storm_hip_mesh *bad = nullptr;
    if (storm_hip_mesh_partition(m, part.data(), P, 0, &bad) == 0) { printf("accepted rank P\n"); return 6; }
    part[0] = -1;
    if (storm_hip_mesh_partition(m, part.data(), P, 0, &bad) == 0) { printf("accepted rank -1\n"); return 6; }
  }
  // more ranks than cells
  {
    std::vector<int32_t> part((size_t)n);
    CK(storm_hip_partition_rcb(2, 5, v.center, 9, part.data()));
    CK(storm_hip_partition_slabs(2, 5, v.center, 0, 9, part.data()));
    printf("rcb(5 cells, 9 ranks): %d %d %d %d %d\n", part[0], part[1], part[2], part[3], part[4]);
    printf("rcb 0 ranks: %d\n", storm_hip_partition_rcb(2, n, v.center, 0, part.data()));
    printf("slabs bad axis: %d\n", storm_hip_partition_slabs(2, n, v.center, 2, 3, part.data()));
  }
  CK(storm_hip_mesh_destroy(m));
  printf("ok\n");
  return 0;
}
