/* Torch-free use of the deployment runtime: plain C, links libspk_hip.so only.
 *
 *   spk_run model.spkm system.bin [repeat] [skin]
 *
 * This is the call sequence a LAMMPS pair style (the role of interfaces/lammps/pair_schnetpack.cpp in the
 * reference: load in coeff() :128, forward in compute() :328) or any other MD code would use:
 * spk_potential_load once, spk_potential_compute_cell (or spk_potential_compute with the code's own neighbour
 * list) per step, spk_potential_free at the end.
 *
 * system.bin (little endian): int64 n_atoms, int64 n_mol, int64 has_cell; int64 z[n_atoms]; int64 idx_m[n_atoms];
 * float R[n_atoms*3]; then if has_cell: float cell[n_mol*9]; uint8 pbc[n_mol*3].
 * Output: "E <mol> <energy>" and "F <atom> <fx> <fy> <fz>" lines, then "ms_per_call <t>" over `repeat` calls.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "../../include/spk_hip.h"

static void die(const char* what) {
  fprintf(stderr, "spk_run: %s: %s\n", what, spk_last_error());
  exit(1);
}
static void rd(void* p, size_t sz, size_t n, FILE* f) {
  if (fread(p, sz, n, f) != n) { fprintf(stderr, "spk_run: short read\n"); exit(2); }
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: spk_run model.spkm system.bin [repeat] [skin]\n"); return 2; }
  const int repeat = argc > 3 ? atoi(argv[3]) : 1;
  const float skin = argc > 4 ? (float)atof(argv[4]) : 0.0f;
  spk_potential_t* pot = NULL;
  if (spk_potential_load(argv[1], &pot)) die("load");
  int32_t info[8]; float cutoff;
  spk_potential_info(pot, info, &cutoff);
  fprintf(stderr, "model: %s F=%d interactions=%d n_rbf=%d cutoff=%g\n", info[0] ? "PaiNN" : "SchNet", info[1], info[2], info[3], cutoff);

  FILE* f = fopen(argv[2], "rb");
  if (!f) { fprintf(stderr, "spk_run: cannot open %s\n", argv[2]); return 2; }
  int64_t hdr[3];
  rd(hdr, 8, 3, f);
  const int64_t n = hdr[0], m = hdr[1];
  int64_t* z = malloc(8 * n); int64_t* idx_m = malloc(8 * n);
  float* R = malloc(12 * n); float* cell = NULL; uint8_t* pbc = NULL;
  rd(z, 8, n, f); rd(idx_m, 8, n, f); rd(R, 4, 3 * n, f);
  if (hdr[2]) { cell = malloc(36 * m); pbc = malloc(3 * m); rd(cell, 4, 9 * m, f); rd(pbc, 1, 3 * m, f); }
  fclose(f);

  float* E = malloc(4 * m); float* F = malloc(12 * n);
  int64_t stats[2];
  if (spk_potential_compute_cell(pot, n, z, R, m, idx_m, cell, pbc, skin, E, F, stats)) die("compute");
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int k = 0; k < repeat; ++k)
    if (spk_potential_compute_cell(pot, n, z, R, m, idx_m, cell, pbc, skin, E, F, stats)) die("compute");
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (int64_t k = 0; k < m; ++k) printf("E %lld %.9g\n", (long long)k, E[k]);
  for (int64_t a = 0; a < n; ++a) printf("F %lld %.9g %.9g %.9g\n", (long long)a, F[3 * a], F[3 * a + 1], F[3 * a + 2]);
  printf("pairs %lld\n", (long long)stats[0]);
  printf("ms_per_call %.6f\n", ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / (repeat > 0 ? repeat : 1));
  spk_potential_free(pot);
  free(z); free(idx_m); free(R); free(cell); free(pbc); free(E); free(F);
  return 0;
}
