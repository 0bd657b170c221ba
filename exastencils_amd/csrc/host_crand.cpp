// Start values drawn from the C library's generator, for programs that initialise a field with
//   loop over F sequentially { F = native ( "((double)std::rand()/RAND_MAX)" ) }
// (Testing/Opts/base.exa4:166-170, Testing/Misc/inlining.exa4:199-203; the generated main() of an MPI program first calls
// std::srand(mpiRank), Compiler/src/exastencils/parallelization/api/mpi/MPI_IVs.scala:41-45).  The reference's results files were
// produced with glibc, whose rand() is the TYPE_3 additive feedback generator: 31 words seeded by the Lehmer sequence
// 16807 * x mod (2^31 - 1), r[i] = r[i-3] + r[i-31] (mod 2^32), the first 310 values discarded, result = r >> 1, RAND_MAX = 2^31 - 1.
// Restated here so that the values do not depend on the C library the host program happens to run with.  Host code: the values
// are written to a host buffer in the order they are drawn -- the loop order of the generated nest, x fastest --; the caller places
// and uploads them.
#include <stdint.h>

#include "examg_common.h"

extern "C" int examg_crand_seed(examg_crand_state_t *st, uint32_t seed) {
  if (!st) { examg::set_error("examg_crand_seed: null argument"); return 1; }
  int32_t r[344];
  r[0] = seed ? (int32_t)seed : 1;       // srand(0) seeds with 1
  for (int i = 1; i < 31; ++i) {
    const long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
    long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    r[i] = (int32_t)w;
  }
  uint32_t u[344];
  for (int i = 0; i < 31; ++i) u[i] = (uint32_t)r[i];
  for (int i = 31; i < 34; ++i) u[i] = u[i - 31];
  for (int i = 34; i < 344; ++i) u[i] = u[i - 31] + u[i - 3];
  for (int i = 0; i < 31; ++i) st->r[i] = u[344 - 31 + i];    // the last 31 words, oldest first
  st->k = 0;                                                  // index of the oldest word (= r[i-31] of the next value)
  return 0;
}

static inline uint32_t crand_next(examg_crand_state_t *st) {
  const int k = st->k;
  const uint32_t v = st->r[k] + st->r[(k + 28) % 31];         // r[i-31] + r[i-3]
  st->r[k] = v;
  st->k = (k + 1) % 31;
  return v >> 1;
}

extern "C" int examg_crand_draw_host(examg_crand_state_t *st, double *host_out, int64_t n) {
  if (!st || (!host_out && n > 0) || n < 0) { examg::set_error("examg_crand_draw_host: null argument or negative count"); return 1; }
  for (int64_t i = 0; i < n; ++i) host_out[i] = (double)crand_next(st) / 2147483647.0;
  return 0;
}
