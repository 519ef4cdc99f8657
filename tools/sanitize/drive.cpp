// ASan / UBSan driver of the host mesh code: reader on the reference's 2-D mesh, a tetrahedral box through files, ordering, partition.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include "storm_hip.h"
#define CK(x) do { int s_ = (x); if (s_) { printf("%s -> %d: %s\n", #x, s_, storm_hip_last_error()); return 1; } } while (0)
int main(int argc, char **argv) {
  // argv: repo root, work directory, box edge
  const std::string root = argc > 1 ? argv[1] : ".", work = argc > 2 ? argv[2] : "/tmp";
  storm_hip_mesh *m = nullptr, *loc = nullptr;
  storm_hip_mesh_view v;
  CK(storm_hip_mesh_read_tetgen((root + "/tests/golden/mesh/square_nb.1.").c_str(), 2, &m));
  CK(storm_hip_mesh_get_view(m, &v));
  printf("2-D: %lld cells %lld faces %lld bfaces\n", (long long)v.n_cells, (long long)v.n_faces, (long long)v.n_bfaces);
  std::vector<int64_t> order((size_t)v.n_cells);
  std::vector<int32_t> part((size_t)v.n_cells);
  int32_t kind = 0;
  for (int mode : {0, 1, 3}) CK(storm_hip_order_cells(v.dim, v.n_cells, v.center, mode, order.data(), &kind));
  CK(storm_hip_mesh_permute_cells(m, order.data()));
  CK(storm_hip_mesh_get_view(m, &v));
  for (int P : {1, 2, 3, 7}) {
    CK(storm_hip_partition_rcb(v.dim, v.n_cells, v.center, P, part.data()));
    for (int r = 0; r < P; ++r) { CK(storm_hip_mesh_partition(m, part.data(), P, r, &loc)); CK(storm_hip_mesh_destroy(loc)); }
    CK(storm_hip_partition_slabs(v.dim, v.n_cells, v.center, 1, P, part.data()));
    for (int r = 0; r < P; ++r) { CK(storm_hip_mesh_partition(m, part.data(), P, r, &loc)); CK(storm_hip_mesh_destroy(loc)); }
  }
  CK(storm_hip_mesh_destroy(m));
  // a tetrahedral box: 5 tets per cube would not conform; use Kuhn's 6
  const int n = argc > 3 ? atoi(argv[3]) : 9, mm = n + 1;
  std::vector<double> pos;
  for (int k = 0; k < mm; ++k) for (int j = 0; j < mm; ++j) for (int i = 0; i < mm; ++i) { pos.push_back(i / (double)n); pos.push_back(j / (double)n); pos.push_back(k / (double)n); }
  std::vector<int64_t> cells, bf, lab;
  const int perms[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
  const bool odd[6] = {false, true, true, false, false, true};
  const int64_t step[3] = {1, mm, (int64_t)mm * mm};
  for (int k = 0; k < n; ++k) for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
    const int64_t base = ((int64_t)k * mm + j) * mm + i;
    for (int p = 0; p < 6; ++p) {
      int64_t a = base, b = a + step[perms[p][0]], c = b + step[perms[p][1]], d = c + step[perms[p][2]];
      if (odd[p]) std::swap(b, c);
      cells.insert(cells.end(), {a, b, c, d});
    }
  }
  // boundary faces: let the library tell us -- first build with none listed must fail
  int st = storm_hip_mesh_from_simplices(3, (int64_t)pos.size() / 3, pos.data(), 0, nullptr, nullptr, (int64_t)cells.size() / 4, cells.data(), &m);
  printf("no faces listed: status %d (%s)\n", st, storm_hip_last_error());
  auto coord = [&](int64_t id, int ax) { return ax == 0 ? id % mm : ax == 1 ? (id / mm) % mm : id / ((int64_t)mm * mm); };
  const int part3[4][3] = {{0,2,1},{0,1,3},{1,2,3},{2,0,3}};
  for (size_t c = 0; c < cells.size() / 4; ++c) for (int f = 0; f < 4; ++f) {
    int64_t t[3] = {cells[4*c+part3[f][0]], cells[4*c+part3[f][1]], cells[4*c+part3[f][2]]};
    bool wall = false;
    for (int ax = 0; ax < 3; ++ax) for (int w : {0, n}) wall |= coord(t[0], ax) == w && coord(t[1], ax) == w && coord(t[2], ax) == w;
    if (wall) { bf.insert(bf.end(), {t[0], t[1], t[2]}); lab.push_back(1); }
  }
  CK(storm_hip_mesh_write_tetgen((work + "/box.1").c_str(), 3, (int64_t)pos.size() / 3, pos.data(), (int64_t)lab.size(), bf.data(), lab.data(), (int64_t)cells.size() / 4, cells.data()));
  CK(storm_hip_mesh_read_tetgen((work + "/box.1.").c_str(), 3, &m));
  CK(storm_hip_mesh_get_view(m, &v));
  double vol = 0; for (int64_t i = 0; i < v.n_cells; ++i) vol += v.volume[i];
  printf("3-D: %lld cells %lld faces %lld bfaces, volume %.15f\n", (long long)v.n_cells, (long long)v.n_faces, (long long)v.n_bfaces, vol);
  order.resize((size_t)v.n_cells), part.resize((size_t)v.n_cells);
  CK(storm_hip_order_cells(3, v.n_cells, v.center, 3, order.data(), &kind));
  CK(storm_hip_mesh_permute_cells(m, order.data()));
  CK(storm_hip_mesh_get_view(m, &v));
  CK(storm_hip_partition_rcb(3, v.n_cells, v.center, 5, part.data()));
  for (int r = 0; r < 5; ++r) { CK(storm_hip_mesh_partition(m, part.data(), 5, r, &loc)); storm_hip_mesh_view lv; CK(storm_hip_mesh_get_view(loc, &lv)); if (r == 0) printf("rank 0: %lld owned %lld halo %d nbrs\n", (long long)lv.n_cells, (long long)lv.n_halo, lv.n_nbrs); CK(storm_hip_mesh_destroy(loc)); }
  CK(storm_hip_mesh_destroy(m));
  // errors
  printf("missing file: %d\n", storm_hip_mesh_read_tetgen((work + "/nope.1").c_str(), 0, &m));
  double nanc[6] = {0, 0, 0, NAN, 1, 1}; int64_t o2[2];
  printf("nan centre: %d (%s)\n", storm_hip_order_cells(3, 2, nanc, 1, o2, nullptr), storm_hip_last_error());
  return 0;
}
