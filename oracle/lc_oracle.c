/* oracle/lc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Independent CPU statement of the transit / eclipse model behind
 * Observation.generate_lightcurves (reference: wayne/observation.py:293-357:
 * per wavelength element  pylc.transit('claret', ldcoeffs, Rp/Rs, ...) -
 * (1 - pylc.eclipse(depth, rp, ...)),  then planet_depths = 1 - models, :442-443).
 *
 * pylightcurve (>= 2.3.2, setup.py:33) is a third-party dependency that is not
 * in the reference tree and not installed here, so its published model is
 * restated: a star with the Claret four-coefficient law
 *     I(mu) = 1 - sum_{n=1..4} a_n (1 - mu^(n/2)),   mu = sqrt(1 - r^2)
 * occulted by an opaque disk of radius p at projected separation z, and a
 * uniform planet disk hidden by the star for the eclipse.  PARITY UNPINNED
 * against pylightcurve itself; pinned to closed forms by tests/test_lightcurve.py:
 * uniform disk (lens area), the r^2 law a = (0, 0, 0, a4) for a planet inside the
 * disk, and the small-planet limit of the quadratic law.
 *
 * Method -- deliberately NOT the product's (wayne_amd/csrc/k_lightcurve.h and
 * wayne_amd/lightcurve.py integrate I(r) r theta(r) dr over stellar radii with
 * the analytic arc theta): here the blocked flux is the plain 2-D integral of
 * the intensity over the PLANET's disk, in polar coordinates (rho, phi) about
 * the planet's centre,
 *     blocked = int_0^p rho d rho  2 int_{phi_a(rho)}^{pi} I(r(rho, phi)) d phi,
 *     r^2 = z^2 + rho^2 + 2 z rho cos(phi),
 * phi_a the angle at which the circle of radius rho leaves the stellar disk.
 * Both integrals use a double-exponential (tanh-sinh) rule in fp64, the outer
 * one split at rho = |1 - z| where the circle first touches the limb.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define LC_MAX_NODES 513

typedef struct {
  int n;
  double x[LC_MAX_NODES];   /* node in (0, 1) */
  double w[LC_MAX_NODES];   /* weight */
  double d[LC_MAX_NODES];   /* distance to the nearer end of (0, 1), accurate for tiny values */
} Rule;

static void make_rule(Rule* r, int n) {
  if (n < 9) n = 9;
  if (n > LC_MAX_NODES) n = LC_MAX_NODES;
  if (!(n & 1)) n += 1;
  const double t_max = 3.3;
  const double h = 2. * t_max / (n - 1);
  r->n = n;
  for (int i = 0; i < n; ++i) {
    const double t = -t_max + h * i;
    const double u = 0.5 * M_PI * sinh(t);
    r->x[i] = 0.5 * (1. + tanh(u));
    r->w[i] = h * 0.25 * M_PI * cosh(t) / (cosh(u) * cosh(u));
    r->d[i] = 0.5 * exp(-fabs(u)) / cosh(u);
  }
}

static double claret(const double* a, double mu) {
  const double s = sqrt(mu);
  return 1. - a[0] * (1. - s) - a[1] * (1. - mu) - a[2] * (1. - mu * s) - a[3] * (1. - mu * mu);
}

/* 2 int_{phi_a}^{pi} I(r(rho, phi)) d phi for the circle of radius rho about the planet's centre */
static double ring(const Rule* q, const double* a, double z, double rho) {
  if (rho <= 0.) return 2. * M_PI * (z < 1. ? claret(a, sqrt(fmax(1. - z * z, 0.))) : 0.);
  if (z <= 0.) return rho < 1. ? 2. * M_PI * claret(a, sqrt(1. - rho * rho)) : 0.;
  if (fabs(z - rho) >= 1.) return 0.;                 /* the circle lies outside the star */
  const int whole = (z + rho <= 1.);                  /* ... or wholly inside it */
  const double c_a = whole ? 1. : (1. - z * z - rho * rho) / (2. * z * rho);
  const double phi_a = whole ? 0. : acos(fmin(fmax(c_a, -1.), 1.));
  const double L = M_PI - phi_a;
  double sum = 0.;
  for (int i = 0; i < q->n; ++i) {
    const double lo = L * (q->x[i] < 0.5 ? q->d[i] : 1. - q->d[i]);    /* phi - phi_a */
    const double phi = phi_a + lo;
    double one_minus_r2;
    if (whole) {
      /* 1 - r^2 = (1 - (z + rho)^2) + 2 z rho (1 - cos phi),  1 - cos phi = 2 sin^2(phi / 2) */
      const double s = sin(0.5 * phi);
      one_minus_r2 = (1. - (z + rho)) * (1. + (z + rho)) + 4. * z * rho * s * s;
    } else {
      /* 1 - r^2 = 2 z rho (cos phi_a - cos phi) = 4 z rho sin((phi + phi_a) / 2) sin((phi - phi_a) / 2) */
      one_minus_r2 = 4. * z * rho * sin(0.5 * (phi + phi_a)) * sin(0.5 * lo);
    }
    sum += q->w[i] * claret(a, sqrt(fmax(one_minus_r2, 0.)));
  }
  return 2. * L * sum;
}

static double disk_segment(const Rule* q, const double* a, double z, double r0, double r1) {
  if (!(r1 > r0)) return 0.;
  const double L = r1 - r0;
  double sum = 0.;
  for (int i = 0; i < q->n; ++i) {
    const double rho = r0 + L * (q->x[i] < 0.5 ? q->d[i] : 1. - q->d[i]);
    sum += q->w[i] * rho * ring(q, a, z, rho);
  }
  return L * sum;
}

/* flux of the star hidden by the planet (units: the star's central intensity x stellar radius^2) */
static double blocked_flux(const Rule* q, const double* a, double z, double p) {
  if (!(p > 0.) || !(z < 1. + p)) return 0.;
  const double k = fabs(1. - z);
  if (k > 0. && k < p) return disk_segment(q, a, z, 0., k) + disk_segment(q, a, z, k, p);
  return disk_segment(q, a, z, 0., p);
}

static double star_flux(const double* a) {
  /* int_disk I dA = pi (1 - sum a_n n / (n + 4)) */
  double s = 0.;
  for (int n = 1; n <= 4; ++n) s += a[n - 1] * n / (n + 4.);
  return M_PI * (1. - s);
}

/* fraction of a disk of radius p at separation z that lies inside the unit disk (lens area / pi p^2) */
static double hidden_fraction(double z, double p) {
  if (!(p > 0.)) return 0.;
  if (z >= 1. + p) return 0.;
  if (z <= fabs(1. - p)) return p <= 1. ? 1. : 1. / (p * p);
  /* lens of two circles (radii 1 and p, centres z apart), by the two circular segments */
  const double x = (z * z + 1. - p * p) / (2. * z);        /* chord position from the star's centre */
  const double y = sqrt(fmax(1. - x * x, 0.));
  const double seg_star = atan2(y, x) - x * y;              /* segment of the unit circle beyond the chord */
  const double xp = z - x;                                  /* ... and of the planet's circle */
  const double seg_planet = p * p * atan2(y, xp) - xp * y;
  return (seg_star + seg_planet) / (M_PI * p * p);
}

/* 1 - transit flux for every (z[k], p[w]) */
void wayne_oracle_lc_deficit(int K, int W, const double* z, const double* p, const double* ld, int nodes,
                             double* out) {
  Rule* q = (Rule*)malloc(sizeof(Rule));
  make_rule(q, nodes);
  const double f0 = star_flux(ld);
  for (int k = 0; k < K; ++k)
    for (int w = 0; w < W; ++w) out[(size_t)k * W + w] = blocked_flux(q, ld, z[k], p[w]) / f0;
  free(q);
}

void wayne_oracle_lc_hidden(int n, const double* z, const double* p, double* out) {
  for (int i = 0; i < n; ++i) out[i] = hidden_fraction(z[i], p[i]);
}

/* planet_depths = 1 - (transit - (1 - eclipse)) per sub-sample and wavelength (observation.py:349-355,
 * 442-443): transit from (z_tr[k], sqrt(depth_w)); eclipse = (1 + f (1 - hidden_k)) / (1 + f) with the
 * planet-to-star flux ratio f = depth_w as the reference passes it (pylc.eclipse(spec_elem ** 2, ...)). */
void wayne_oracle_lc_depths(int K, int W, const double* z_tr, const double* hidden, const double* spectrum,
                            const double* ld, int nodes, double* out) {
  Rule* q = (Rule*)malloc(sizeof(Rule));
  make_rule(q, nodes);
  const double f0 = star_flux(ld);
  for (int k = 0; k < K; ++k)
    for (int w = 0; w < W; ++w) {
      const double f = spectrum[w];
      const double tr = 1. - blocked_flux(q, ld, z_tr[k], sqrt(f)) / f0;
      const double h = hidden ? hidden[k] : 0.;
      const double ecl = (1. + f * (1. - h)) / (1. + f);
      out[(size_t)k * W + w] = 1. - (tr - (1. - ecl));
    }
  free(q);
}
