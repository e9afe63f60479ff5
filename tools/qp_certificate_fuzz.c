/*
 * qp_certificate_fuzz.c -- test infrastructure: the frozen-iterate certificate of include/mm_qp.h (mm_qp_frozen, the
 * MM_QP_CERTIFY build the HIP kernels run) against the literal interior-point loop, on the CPU.
 *
 *   gcc -O2 -ffp-contract=off -fopenmp -o qp_certificate_fuzz tools/qp_certificate_fuzz.c -lm
 *   ./qp_certificate_fuzz N            N random QPs (a, h0..h3, rows drawn around the feasibility boundary of the CBF row,
 *                                      the shapes the shield builds: cbf.py:288-322,374-422)
 *   ./qp_certificate_fuzz - < file     QPs from a file, one per line: a h0 h1 h2 h3 rows
 *
 * For every QP the certified loop must return the same (d, slack, status, iteration count) bits as the loop that never
 * asks the certificate.  Prints one JSON line; exit status 1 on any mismatch.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define MM_QP_CERTIFY
#include "../include/mm_qp.h"

static inline uint64_t rng(uint64_t *s) { uint64_t x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return *s = x; }
static inline double u01(uint64_t *s) { return (rng(s) >> 11) * 0x1p-53; }

typedef struct { double d, s; int status, iters, cert_at; } Ans;
/* certify = 0: the literal loop (a certificate that fires is ignored) */
static Ans solve(double a, double h0, double h1, double h2, double h3, int rows, int certify) {
  MMQpState q; MMQpRes r; Ans o = {0, 0, 0, 0, -1};
  if (mm_qp_start(&q, a, h0, h1, h2, h3, rows))
    for (;;) {
      int stop = mm_qp_top(&q, &r);
      if (stop == 3 && !certify) stop = 0;
      if (stop) { o.status = stop == 1; if (stop == 3) o.cert_at = q.iters; break; }
      if (!mm_qp_bottom(&q, &r)) break;
    }
  o.d = q.x0; o.s = q.x2; o.iters = o.cert_at >= 0 ? MM_QP_MAXITERS : q.iters;
  return o;
}
static void draw(long i, double *a, double *h0, double *h1, double *h2, double *h3, int *rows) {
  uint64_t st = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1) + 12345; rng(&st); rng(&st);
  const int mode = (int)(rng(&st) % 8);
  *a = (0.3 + 0.7 * u01(&st)) / 15.0;                       /* g.vx * dt */
  if (mode == 7) *a = (0.001 + u01(&st)) / 15.0;
  double u0 = 30 * u01(&st), vmin = fmax(0.0, u0 - 12.5 / 15 + (u01(&st) - 0.5) * 0.2), vmax = u0 + 6.0 / 15 + (u01(&st) - 0.5) * 0.2;
  if (mode == 6) vmin = u0 + 0.01 * u01(&st);               /* v_min - v_ll > 0 */
  *h1 = vmax - u0; *h2 = -vmin + u0;
  *rows = (rng(&st) % 10 == 0) ? 4 : 3;
  const double bnd = -*a * *h2;                             /* h0 below this: the CBF row needs the slack */
  double eps;
  switch (mode) {
    case 0: eps = 3 * u01(&st); break;
    case 1: eps = -0.5 * u01(&st); break;
    case 2: eps = ldexp(u01(&st), -(int)(rng(&st) % 60)); break;
    case 3: eps = -ldexp(u01(&st), -(int)(rng(&st) % 60)); break;
    case 4: eps = *a * *h2 + (u01(&st) - 0.5) * 0.2; break;
    default: eps = (u01(&st) - 0.7) * 2; break;
  }
  *h0 = bnd + eps; *h3 = 0;
  if (*rows == 4) {
    *h3 = bnd + ((rng(&st) & 1) ? eps : 3 * (u01(&st) - 0.3)) + (u01(&st) - 0.5) * ((rng(&st) & 1) ? 1.0 : 1e-6);
    if (rng(&st) % 3 == 0) { double t = *h0; *h0 = *h3 + 1.0; *h3 = t; }
  }
}
int main(int argc, char **argv) {
  const int from_file = argc > 1 && strcmp(argv[1], "-") == 0;
  long N = argc > 1 && !from_file ? atol(argv[1]) : 1000000;
  double *in = NULL;
  if (from_file) {
    long cap = 1 << 16; N = 0; in = (double *)malloc(cap * 6 * sizeof(double));
    double a, h0, h1, h2, h3; int rows;
    while (scanf("%lf %lf %lf %lf %lf %d", &a, &h0, &h1, &h2, &h3, &rows) == 6) {
      if (N == cap) { cap *= 2; in = (double *)realloc(in, cap * 6 * sizeof(double)); }
      double *p = in + N * 6; p[0] = a; p[1] = h0; p[2] = h1; p[3] = h2; p[4] = h3; p[5] = rows; N++;
    }
  }
  long capped = 0, certified = 0, uncert = 0, bad = 0, first = 1000, last = 0;
#pragma omp parallel for reduction(+:capped,certified,uncert,bad) reduction(min:first) reduction(max:last) schedule(dynamic, 1000)
  for (long i = 0; i < N; i++) {
    double a, h0, h1, h2, h3; int rows;
    if (from_file) { const double *p = in + i * 6; a = p[0]; h0 = p[1]; h1 = p[2]; h2 = p[3]; h3 = p[4]; rows = (int)p[5]; }
    else draw(i, &a, &h0, &h1, &h2, &h3, &rows);
    const Ans f = solve(a, h0, h1, h2, h3, rows, 0), c = solve(a, h0, h1, h2, h3, rows, 1);
    if (f.iters == MM_QP_MAXITERS) { capped++; if (c.cert_at < 0) uncert++; }
    if (c.cert_at >= 0) { certified++; if (c.cert_at < first) first = c.cert_at; if (c.cert_at > last) last = c.cert_at; }
    if (memcmp(&f.d, &c.d, 8) || memcmp(&f.s, &c.s, 8) || f.status != c.status || f.iters != c.iters) {
      bad++;
#pragma omp critical
      fprintf(stderr, "MISMATCH a=%a h=(%a,%a,%a,%a) rows=%d literal=(%a,%a,%d,%d) certified=(%a,%a,%d,%d at %d)\n", a, h0, h1, h2, h3, rows,
              f.d, f.s, f.status, f.iters, c.d, c.s, c.status, c.iters, c.cert_at);
    }
  }
  printf("{\"qps\": %ld, \"capped\": %ld, \"certified\": %ld, \"capped_not_certified\": %ld, \"mismatches\": %ld, \"first_certificate_iteration\": %ld, "
         "\"last_certificate_iteration\": %ld}\n", N, capped, certified, uncert, bad, certified ? first : -1, last);
  return bad != 0;
}
