/* Is  q' = fma(fma(-q, D, x), r, q)  with  q = RN(x r),  r = RN(1 / D)  the IEEE quotient x / D?
 * (div_cs in lettuce_amd/csrc/kernels.hpp: division by the constants 2 cs^2 and cs^2 of the
 * reference's QuadraticEquilibrium, quadratic_equilibrium.py:15-24, in three instructions.)
 *
 *   exact_division_check f32 <stride>   every stride-th fp32 bit pattern (stride 1 = exhaustive,
 *                                       ~45 s on 8 threads); quotients below 1e-30 in magnitude
 *                                       are skipped (the remainder underflows there)
 *   exact_division_check f32two <stride> the same for the two-instruction fp32 form  fma(x, r_hi, RN(x r_lo))  (round 3)
 *   exact_division_check f64 <count>    count random fp64 arguments with exponents in [-60, 20]
 * Prints "mismatches <n>"; exit status 0 iff n == 0.
 * Build: gcc -O2 -mfma -ffp-contract=off exact_division_check.c -lm -lpthread
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define THREADS 8
static uint64_t g_param;
static uint64_t g_bad[THREADS];

static void *run32(void *arg) {
  const int t = (int)(intptr_t)arg;
  const double cs2 = (1.0 / sqrt(3.0)) * (1.0 / sqrt(3.0));   /* python: (1 / sqrt(3)) ** 2 */
  uint64_t bad = 0;
  for (int w = 0; w < 2; ++w) {
    const float d = (float)(w == 0 ? 2.0 * cs2 : cs2);
    const float r = (float)(1.0 / (double)d);
    const uint64_t lo = ((uint64_t)1 << 32) / THREADS * t, hi = ((uint64_t)1 << 32) / THREADS * (t + 1);
    for (uint64_t i = lo; i < hi; i += g_param) {
      const uint32_t bits = (uint32_t)i;
      float x;
      memcpy(&x, &bits, 4);
      if (!(x == x) || isinf(x)) continue;
      volatile float q = x * r;
      const float rem = fmaf(-q, d, x);
      const float q2 = fmaf(rem, r, q);
      const float ref = x / d;
      if (fabsf(ref) < 1e-30f || isinf(ref)) continue;
      if (memcmp(&q2, &ref, 4) != 0) ++bad;
    }
  }
  g_bad[t] = bad;
  return 0;
}

/* fp32, round 3: the two-instruction form  fma(x, r_hi, RN(x r_lo)),  r_hi = RN(1 / D), r_lo = RN(1 / D - r_hi) */
static void *run32two(void *arg) {
  const int t = (int)(intptr_t)arg;
  const double cs2 = (1.0 / sqrt(3.0)) * (1.0 / sqrt(3.0));
  uint64_t bad = 0;
  for (int w = 0; w < 2; ++w) {
    const float d = (float)(w == 0 ? 2.0 * cs2 : cs2);
    const float rh = (float)(1.0 / (double)d);
    const float rl = (float)(1.0 / (double)d - (double)rh);
    const uint64_t lo = ((uint64_t)1 << 32) / THREADS * t, hi = ((uint64_t)1 << 32) / THREADS * (t + 1);
    for (uint64_t i = lo; i < hi; i += g_param) {
      const uint32_t bits = (uint32_t)i;
      float x;
      memcpy(&x, &bits, 4);
      if (!(x == x) || isinf(x)) continue;
      volatile float low = x * rl;
      const float q = fmaf(x, rh, low);
      const float ref = x / d;
      if (fabsf(ref) < 1e-30f || isinf(ref)) continue;
      if (memcmp(&q, &ref, 4) != 0) ++bad;
    }
  }
  g_bad[t] = bad;
  return 0;
}

static uint64_t next(uint64_t *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }

static void *run64(void *arg) {
  const int t = (int)(intptr_t)arg;
  const double cs2 = (1.0 / sqrt(3.0)) * (1.0 / sqrt(3.0));
  const double d[2] = {2.0 * cs2, cs2}, r[2] = {1.0 / (2.0 * cs2), 1.0 / cs2};
  uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1), bad = 0;
  for (uint64_t i = 0; i < g_param / THREADS; ++i) {
    const uint64_t m = next(&s);
    const int e = (int)(next(&s) % 81) - 60;
    const uint64_t bits = (m & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 + e) << 52);
    double x;
    memcpy(&x, &bits, 8);
    for (int w = 0; w < 2; ++w) {
      volatile double q = x * r[w];
      const double rem = fma(-q, d[w], x);
      const double q2 = fma(rem, r[w], q);
      if (q2 != x / d[w]) ++bad;
    }
  }
  g_bad[t] = bad;
  return 0;
}

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s f32 <stride> | f32two <stride> | f64 <count>\n", argv[0]); return 2; }
  g_param = strtoull(argv[2], 0, 10);
  if (g_param == 0) return 2;
  void *(*fn)(void *) = strcmp(argv[1], "f64") == 0 ? run64 : (strcmp(argv[1], "f32two") == 0 ? run32two : run32);
  pthread_t th[THREADS];
  for (int t = 0; t < THREADS; ++t) pthread_create(&th[t], 0, fn, (void *)(intptr_t)t);
  uint64_t bad = 0;
  for (int t = 0; t < THREADS; ++t) { pthread_join(th[t], 0); bad += g_bad[t]; }
  printf("mismatches %llu\n", (unsigned long long)bad);
  return bad == 0 ? 0 : 1;
}
