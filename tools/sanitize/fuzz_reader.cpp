#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "storm_hip.h"
// Reads every mutated file set of <work>/fz/list.txt (tools/sanitize/make_fuzz_files.py): accepted or rejected, never a crash.
int main(int argc, char **argv) {
  const std::string work = argc > 1 ? argv[1] : "/tmp";
  FILE *f = fopen((work + "/fz/list.txt").c_str(), "r");
  if (!f) return 2;
  char path[512]; int dim, ok = 0, bad = 0;
  while (fscanf(f, "%500s %d", path, &dim) == 2) {
    storm_hip_mesh *m = nullptr;
    int st = storm_hip_mesh_read_tetgen(path, dim, &m);
    if (st == 0) {
      storm_hip_mesh_view v; storm_hip_mesh_get_view(m, &v);
      double s = 0; for (int64_t i = 0; i < v.n_cells; ++i) s += v.volume[i];
      ++ok; storm_hip_mesh_destroy(m);
    } else { ++bad; if (bad <= 8) printf("%s: %s\n", path, storm_hip_last_error()); }
  }
  printf("accepted %d rejected %d\n", ok, bad);
  return 0;
}
