/* CPU oracle, plain C, fp32: restatement of the digifil stage chain the reference selects
 * (process_vdif.py:156-182).  TEST INFRASTRUCTURE ONLY -- used by tests/ (validated against the
 * numpy fp64 oracle) and by bench.py's cpu_baseline leg (timed on the host cores); never linked or
 * loaded by the product.  PARITY UNPINNED against DSPSR itself (absent, see frb_oracle.py header).
 *
 *   unpack_2bit      A4  process_vdif.py:157,160 (-2); spif2file.sh:34
 *   filterbank_block A5+A6  -F C:R (process_vdif.py:162-171): forward real FFT of N = 2CR samples,
 *                    C slices of R bins, backward complex FFT of R per slice (unnormalised)
 *   detect           A7  -d1/-d3/-d4/-P (process_vdif.py:58-64,163-176)
 *   tscrunch         A8  -t (process_vdif.py:156-158)
 * The FFT is an iterative Stockham radix-4/2 autosort transform written for this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float re, im; } cpx;

static const float LEVELS[4] = {-3.3359f, -1.0f, 1.0f, 3.3359f};

/* u8[nbytes] -> pol0[2*nbytes], pol1[2*nbytes] */
void frbo_unpack_2bit(const uint8_t* payload, size_t nbytes, float* p0, float* p1) {
  for (size_t i = 0; i < nbytes; ++i) {
    const uint8_t b = payload[i];
    p0[2 * i] = LEVELS[b & 3];
    p1[2 * i] = LEVELS[(b >> 2) & 3];
    p0[2 * i + 1] = LEVELS[(b >> 4) & 3];
    p1[2 * i + 1] = LEVELS[(b >> 6) & 3];
  }
}

/* Stockham autosort FFT, n = power of two, sign = -1 forward / +1 backward, unnormalised.
 * x is overwritten with the result; y is scratch of the same size. tw[k] = exp(-2 pi i k / n). */
static void fft_stockham(cpx* x, cpx* y, size_t n, int sign, const cpx* tw) {
  size_t l = n, m = 1; /* l = remaining length, m = stride */
  cpx *a = x, *b = y;
  while (l >= 4) {
    const size_t l4 = l / 4;
    for (size_t j = 0; j < l4; ++j) {
      /* twiddles w^j, w^2j, w^3j with w = exp(sign 2 pi i / l) = tw[(n/l) * j] (conj for backward) */
      const size_t s = (n / l) * j;
      cpx w1 = tw[s % n], w2 = tw[(2 * s) % n], w3 = tw[(3 * s) % n];
      if (sign > 0) { w1.im = -w1.im; w2.im = -w2.im; w3.im = -w3.im; }
      for (size_t k = 0; k < m; ++k) {
        const cpx c0 = a[k + m * (j)], c1 = a[k + m * (j + l4)], c2 = a[k + m * (j + 2 * l4)], c3 = a[k + m * (j + 3 * l4)];
        const cpx t0 = {c0.re + c2.re, c0.im + c2.im}, t1 = {c0.re - c2.re, c0.im - c2.im};
        const cpx t2 = {c1.re + c3.re, c1.im + c3.im};
        cpx t3 = {c1.re - c3.re, c1.im - c3.im};
        /* multiply t3 by (sign) i:  forward: -i */
        cpx jt3;
        if (sign < 0) { jt3.re = t3.im; jt3.im = -t3.re; } else { jt3.re = -t3.im; jt3.im = t3.re; }
        const cpx u0 = {t0.re + t2.re, t0.im + t2.im};
        const cpx u1 = {t1.re + jt3.re, t1.im + jt3.im};
        const cpx u2 = {t0.re - t2.re, t0.im - t2.im};
        const cpx u3 = {t1.re - jt3.re, t1.im - jt3.im};
        cpx* o = b + k + m * (4 * j);
        o[0] = u0;
        o[m].re = u1.re * w1.re - u1.im * w1.im; o[m].im = u1.re * w1.im + u1.im * w1.re;
        o[2 * m].re = u2.re * w2.re - u2.im * w2.im; o[2 * m].im = u2.re * w2.im + u2.im * w2.re;
        o[3 * m].re = u3.re * w3.re - u3.im * w3.im; o[3 * m].im = u3.re * w3.im + u3.im * w3.re;
      }
    }
    l = l4; m *= 4;
    cpx* t = a; a = b; b = t;
  }
  if (l == 2) {
    for (size_t k = 0; k < m; ++k) {
      const cpx c0 = a[k], c1 = a[k + m];
      b[k].re = c0.re + c1.re; b[k].im = c0.im + c1.im;
      b[k + m].re = c0.re - c1.re; b[k + m].im = c0.im - c1.im;
    }
    cpx* t = a; a = b; b = t;
  }
  if (a != x) memcpy(x, a, n * sizeof(cpx));
}

typedef struct {
  size_t c, r, n;
  cpx *tw_half, *tw_r, *tw_n; /* exp(-2 pi i k / (n/2)), / r, / n (k < n/2) */
  cpx *buf, *scr, *spec;
  float *x0, *x1;
} frbo_plan;

static cpx* make_tw(size_t n, size_t count) {
  cpx* t = (cpx*)malloc(count * sizeof(cpx));
  for (size_t k = 0; k < count; ++k) {
    const double a = -2.0 * M_PI * (double)k / (double)n;
    t[k].re = (float)cos(a); t[k].im = (float)sin(a);
  }
  return t;
}

frbo_plan* frbo_plan_create(size_t nchan, size_t freq_res) {
  frbo_plan* p = (frbo_plan*)calloc(1, sizeof(frbo_plan));
  p->c = nchan; p->r = freq_res; p->n = 2 * nchan * freq_res;
  p->tw_half = make_tw(p->n / 2, p->n / 2);
  p->tw_r = make_tw(p->r, p->r);
  p->tw_n = make_tw(p->n, p->n / 2);
  p->buf = (cpx*)malloc(p->n / 2 * sizeof(cpx));
  p->scr = (cpx*)malloc(p->n / 2 * sizeof(cpx));
  p->spec = (cpx*)malloc(p->n / 2 * sizeof(cpx));
  p->x0 = (float*)malloc(p->n * sizeof(float));
  p->x1 = (float*)malloc(p->n * sizeof(float));
  return p;
}
void frbo_plan_destroy(frbo_plan* p) {
  if (!p) return;
  free(p->tw_half); free(p->tw_r); free(p->tw_n); free(p->buf); free(p->scr); free(p->spec); free(p->x0); free(p->x1);
  free(p);
}

/* forward real FFT of n samples -> spec[0 .. n/2) (Nyquist dropped), via an n/2-point complex FFT */
static void real_fft(frbo_plan* p, const float* x, cpx* spec) {
  const size_t h = p->n / 2;
  for (size_t i = 0; i < h; ++i) { p->buf[i].re = x[2 * i]; p->buf[i].im = x[2 * i + 1]; }
  fft_stockham(p->buf, p->scr, h, -1, p->tw_half);
  for (size_t k = 0; k < h; ++k) {
    const cpx zk = p->buf[k], zc = p->buf[(h - k) % h];
    const cpx e = {0.5f * (zk.re + zc.re), 0.5f * (zk.im - zc.im)};      /* (Z[k] + conj Z[h-k]) / 2 */
    const cpx o = {0.5f * (zk.im + zc.im), -0.5f * (zk.re - zc.re)};     /* (Z[k] - conj Z[h-k]) / 2i */
    const cpx w = p->tw_n[k];
    spec[k].re = e.re + (o.re * w.re - o.im * w.im);
    spec[k].im = e.im + (o.re * w.im + o.im * w.re);
  }
}

/* one block: payload bytes (n/2) -> power[nif][c][r/tscr]; pol_mode 0,1,2,3,4,5 */
void frbo_block_power(frbo_plan* p, const uint8_t* payload, int pol_mode, int tscr, float* power) {
  const size_t c = p->c, r = p->r, nt = r / (size_t)tscr;
  frbo_unpack_2bit(payload, p->n / 2, p->x0, p->x1);
  cpx* y0 = p->spec;                 /* reuse: spectra of pol0 in spec, pol1 processed per channel */
  real_fft(p, p->x0, y0);
  cpx* y1 = (cpx*)malloc(p->n / 2 * sizeof(cpx));
  real_fft(p, p->x1, y1);
  cpx* scr = p->scr;
  for (size_t k = 0; k < c; ++k) {
    cpx* a = y0 + k * r;
    cpx* b = y1 + k * r;
    fft_stockham(a, scr, r, +1, p->tw_r);
    fft_stockham(b, scr, r, +1, p->tw_r);
    for (size_t t = 0; t < nt; ++t) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int u = 0; u < tscr; ++u) {
        const cpx p0 = a[t * tscr + u], p1 = b[t * tscr + u];
        const float pp = p0.re * p0.re + p0.im * p0.im, qq = p1.re * p1.re + p1.im * p1.im;
        if (pol_mode == 0) s0 += pp;
        else if (pol_mode == 1) s0 += qq;
        else if (pol_mode == 2) s0 += pp + qq;
        else if (pol_mode == 3) s0 += (pp + qq) * (pp + qq);
        else {
          s0 += pp; s1 += qq;
          s2 += p0.re * p1.re + p0.im * p1.im;      /* Re(p0 conj p1) */
          s3 += p0.im * p1.re - p0.re * p1.im;      /* Im(p0 conj p1) */
        }
      }
      if (pol_mode == 5) {                          /* Stokes I, Q, U, V of circular feeds (frb_oracle.py: iquv) */
        power[(0 * c + k) * nt + t] = s0 + s1;
        power[(1 * c + k) * nt + t] = 2.0f * s2;
        power[(2 * c + k) * nt + t] = 2.0f * s3;
        power[(3 * c + k) * nt + t] = s0 - s1;
      } else {
      power[(0 * c + k) * nt + t] = s0;
      if (pol_mode == 4) {
        power[(1 * c + k) * nt + t] = s1;
        power[(2 * c + k) * nt + t] = s2;
        power[(3 * c + k) * nt + t] = s3;
      }
      }
    }
  }
  free(y1);
}
