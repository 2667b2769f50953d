/*
 * me_oracle.c -- plain-C many-chain restatement of the hot path.  TEST / BENCH INFRASTRUCTURE ONLY: the checker and
 * the strong CPU baseline; the product never links or loads it.
 *
 * Restates, for N independent chains, the reference's step_all with the identity proposal shape
 * (/root/reference/metropolisengine/metropolis_engine.py): draw_real_group :261-272 and draw_complex_group :274-302
 * with covariance = identity, metropolis_decision :319-338, update_*_sigma :429-456, for the diagonal quadratic
 * energy sum a_i x_i^2 + sum b_j |z_j|^2 (README.md:26-27, BASELINE configs 1-3).  float64 throughout, like the
 * reference.  Random streams: the counter-based Philox4x32-10 layout specified in oracle/philox.py, so a chain here
 * reproduces chain c of oracle/manychain.py (tests/test_oracle_c.py, 1e-12) and of the float64 HIP kernels.
 * Chains are spread over cores with OpenMP.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int32_t n_real, n_complex;
  int64_t n_chains;
  uint64_t seed, chain_offset, step_index, measure_count;
  double temp, target_acceptance, ratio;
  int32_t m;
  int64_t accepted, proposed;
  double *x;       /* [n_chains][D], D = n_real + 2 n_complex: real, Re z, Im z */
  double *energy;  /* [n_chains] */
  double *width;   /* [n_chains] */
  double *weight;  /* [D] = a_0.., b_0.., b_0.. */
} meo_state;

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

static double unit_open(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

static double energy_of(const meo_state *s, const double *x) {
  const int d = s->n_real + 2 * s->n_complex;
  double e = 0.0;
  for (int k = 0; k < d; ++k) e += s->weight[k] * x[k] * x[k];
  return e;
}

meo_state *meo_create(int32_t n_real, int32_t n_complex, int64_t n_chains, uint64_t seed, uint64_t chain_offset,
                      double temp, double target_acceptance, double sampling_width, double ratio,
                      const double *weights, const double *initial) {
  const int d = n_real + 2 * n_complex;
  meo_state *s = (meo_state *)calloc(1, sizeof(meo_state));
  s->n_real = n_real; s->n_complex = n_complex; s->n_chains = n_chains;
  s->seed = seed; s->chain_offset = chain_offset; s->measure_count = 1;
  s->temp = temp; s->target_acceptance = target_acceptance; s->ratio = ratio; s->m = n_real + n_complex;
  s->x = (double *)malloc(sizeof(double) * (size_t)n_chains * d);
  s->energy = (double *)malloc(sizeof(double) * (size_t)n_chains);
  s->width = (double *)malloc(sizeof(double) * (size_t)n_chains);
  s->weight = (double *)malloc(sizeof(double) * d);
  memcpy(s->weight, weights, sizeof(double) * d);
  for (int64_t c = 0; c < n_chains; ++c) {
    memcpy(s->x + c * d, initial, sizeof(double) * d);
    s->energy[c] = energy_of(s, initial);
    s->width[c] = sampling_width;
  }
  return s;
}

void meo_destroy(meo_state *s) {
  if (!s) return;
  free(s->x); free(s->energy); free(s->width); free(s->weight); free(s);
}

/* n_sweeps x step_all over every chain */
void meo_step(meo_state *s, int32_t n_sweeps) {
  const int nr = s->n_real, nc = s->n_complex, d = nr + 2 * nc;
  const int nw = 2 * ((d + 1) / 2);            /* words of the Box-Muller pairs; word nw = accept uniform */
  const int nblk = (nw + 1 + 3) / 4;
  const double damping = fmax((double)s->measure_count / (double)s->m, 200.0);   /* :430 */
  const double p = s->target_acceptance;
  int64_t accepted = 0;
#pragma omp parallel for schedule(static) reduction(+ : accepted)
  for (int64_t c = 0; c < s->n_chains; ++c) {
    double *x = s->x + c * d;
    double e = s->energy[c], w = s->width[c];
    const uint64_t gid = s->chain_offset + (uint64_t)c;
    uint32_t words[4 * 40];
    double g[2 * 80], xp[160];
    for (int sweep = 0; sweep < n_sweeps; ++sweep) {
      const uint64_t step = s->step_index + (uint64_t)sweep;
      for (int b = 0; b < nblk; ++b) {
        uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step,
                           (((uint32_t)(step >> 32) & 0xFFFFu) << 16) | (uint32_t)b};
        philox4x32_10(ctr, (uint32_t)s->seed, (uint32_t)(s->seed >> 32));
        memcpy(words + 4 * b, ctr, sizeof(ctr));
      }
      for (int q = 0; q < nw / 2; ++q) {
        const double r = sqrt(-2.0 * log(unit_open(words[2 * q])));
        const double theta = 6.283185307179586 * unit_open(words[2 * q + 1]);
        g[2 * q] = r * cos(theta);
        g[2 * q + 1] = r * sin(theta);
      }
      const double u = unit_open(words[nw]);
      for (int i = 0; i < nr; ++i) xp[i] = x[i] + w * g[i];                          /* :268 with C = I */
      for (int j = nr; j < d; ++j) xp[j] = x[j] + w * (g[j] * 0.70710678118654752440); /* :298: sigma^2/2 per part */
      const double e_new = energy_of(s, xp);
      const double diff = e_new - e;
      int accept = diff <= 0.0;                                                       /* :329-330 */
      if (!accept && s->temp > 0.0) accept = u <= exp(-diff / s->temp);               /* :333-336 */
      if (accept) {
        memcpy(x, xp, sizeof(double) * d);
        e = e_new;
        ++accepted;
      }
      const double scale = w * s->ratio;                                              /* :431-435 */
      w = accept ? w + scale * (1.0 - p) / damping : w - scale * p / damping;
    }
    s->energy[c] = e;
    s->width[c] = w;
  }
  s->accepted += accepted;
  s->proposed += s->n_chains * (int64_t)n_sweeps;
  s->step_index += (uint64_t)n_sweeps;
}

const double *meo_params(const meo_state *s) { return s->x; }
const double *meo_widths(const meo_state *s) { return s->width; }
const double *meo_energies(const meo_state *s) { return s->energy; }
int64_t meo_accepted(const meo_state *s) { return s->accepted; }
int64_t meo_proposed(const meo_state *s) { return s->proposed; }
