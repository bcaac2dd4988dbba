/*
 * hitl_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See hitl_oracle.h.
 *
 * PARITY UNPINNED (no reference tests/goldens exist; reference unbuildable here) -- except orc_angle_mod and the rotation of
 * orc_odom_setup, held bit for bit to the reference's own shared/math/util.h compiled in place (oracle/_ref, tests/test_oracle_ref.py).
 *
 * What follows the reference, by section:
 *   jets            : forward-mode dual numbers = ceres::Jet<double,6> as used by
 *                     ceres::AutoDiffCostFunction at JointOptimization.cpp:553-557, 817-821,
 *                     994-1049.
 *   odometry        : JointOptimization.cpp:736-825 (setup), residual_functors.h:1081-1125.
 *   human rows      : JointOptimization.cpp:969-1054 (targets), residual_functors.h:1299-1415.
 *   STF glob factor : JointOptimization.cpp:539-559, residual_functors.h:786-830.
 *   matching        : JointOptimization.cpp:561-642, perception_tools/kdtree.cpp:37-197.
 *   solve           : the ceres::Solve call sites JointOptimization.cpp:1093, :1208 with the
 *                     options at :1066-1070 / :145-162; SetParams/CopyParams :380-402.
 * Compile with -ffp-contract=off so fp32 matching arithmetic is bit-reproducible.
 */
#include "hitl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define ORC_2PI 6.28318530717958647692 /* M_2PI, shared/math/util.h */

/* ===================================================================================== */
/* options                                                                                */
/* ===================================================================================== */
static void orc_options_default(hitl_options* o) {
  memset(o, 0, sizeof(*o));
  o->abi_version = HITL_ABI_VERSION;
  o->device = 0;
  o->laser_std_dev = 0.05f;
  o->point_correlation_factor = 1.0f / 40.0f;
  o->point_match_threshold = 0.15f;
  o->max_stf_angle_error = (float)(25.0 * M_PI / 180.0);
  o->max_correspondences_per_point = 6;
  o->num_skip_readings = 1;
  o->min_inter_pose_correspondence = 10;
  o->odom_radial_std_dev = 0.03f;
  o->odom_tangential_std_dev = 0.03f;
  o->odom_angular_std_dev = 0.01f;
  o->max_num_iterations = 100;
  o->jacobi_scaling = 1;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->linear_solver_tolerance = 1e-12;
  o->max_linear_iterations = 500;
  o->max_band_poses = 4;
  o->rank = 0;
  o->world_size = 1;
  o->timing_level = 1;
}

double orc_angle_mod(double a) { /* shared/math/util.h:433-439 */
  a -= ORC_2PI * rint(a / ORC_2PI);
  return a;
}

/* ===================================================================================== */
/* jets: value + 6 partials                                                               */
/* ===================================================================================== */
#define JN 6
typedef struct { double v; double d[JN]; } jet;

static jet jconst(double c) { jet r; r.v = c; for (int k = 0; k < JN; ++k) r.d[k] = 0; return r; }
static jet jvar(double v, int k) { jet r = jconst(v); if (k >= 0) r.d[k] = 1.0; return r; }
static jet jadd(jet a, jet b) { jet r; r.v = a.v + b.v; for (int k = 0; k < JN; ++k) r.d[k] = a.d[k] + b.d[k]; return r; }
static jet jsub(jet a, jet b) { jet r; r.v = a.v - b.v; for (int k = 0; k < JN; ++k) r.d[k] = a.d[k] - b.d[k]; return r; }
static jet jneg(jet a) { jet r; r.v = -a.v; for (int k = 0; k < JN; ++k) r.d[k] = -a.d[k]; return r; }
static jet jmul(jet a, jet b) { jet r; r.v = a.v * b.v; for (int k = 0; k < JN; ++k) r.d[k] = a.v * b.d[k] + a.d[k] * b.v; return r; }
static jet jdiv(jet a, jet b) {
  jet r; const double inv = 1.0 / b.v; r.v = a.v * inv;
  for (int k = 0; k < JN; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) * inv;
  return r;
}
static jet jsin(jet a) { jet r; const double c = cos(a.v); r.v = sin(a.v); for (int k = 0; k < JN; ++k) r.d[k] = c * a.d[k]; return r; }
static jet jcos(jet a) { jet r; const double s = -sin(a.v); r.v = cos(a.v); for (int k = 0; k < JN; ++k) r.d[k] = s * a.d[k]; return r; }
static jet jatan2(jet g, jet f) { /* atan2(g, f) */
  jet r; const double t = 1.0 / (f.v * f.v + g.v * g.v); r.v = atan2(g.v, f.v);
  for (int k = 0; k < JN; ++k) r.d[k] = t * (-g.v * f.d[k] + f.v * g.d[k]);
  return r;
}
static jet jsqrt(jet a) { jet r; r.v = sqrt(a.v); const double t = 1.0 / (2.0 * r.v); for (int k = 0; k < JN; ++k) r.d[k] = t * a.d[k]; return r; }

/* ===================================================================================== */
/* odometry factor                                                                        */
/* ===================================================================================== */
void orc_odom_setup(const float* pp, const float* pc, float A[4], float* rho, float* rot) {
  /* JointOptimization.cpp:748-769, fp32 like the reference (Vector2f / Rotation2Df). */
  const float kEpsilon = 1e-6f;
  const float tx = pc[0] - pp[0], ty = pc[1] - pp[1];
  float rx, ry;
  if (fabsf(tx) < kEpsilon && fabsf(ty) < kEpsilon) {
    rx = cosf(pc[2]); ry = sinf(pc[2]);
    *rho = 0.0f;
  } else {
    const float c = cosf(-pp[2]), s = sinf(-pp[2]);
    const float vx = c * tx - s * ty, vy = s * tx + c * ty;
    const float nrm = sqrtf(vx * vx + vy * vy);
    rx = vx / nrm; ry = vy / nrm;
    *rho = sqrtf(tx * tx + ty * ty);
  }
  /* tangential = Rotation2Df(M_PI_2) * radial */
  const float c90 = cosf((float)M_PI_2), s90 = sinf((float)M_PI_2);
  const float qx = c90 * rx - s90 * ry, qy = s90 * rx + c90 * ry;
  A[0] = rx; A[1] = ry; A[2] = qx; A[3] = qy;
  /* angle_mod<float>: evaluated in double (M_2PI is a double literal), stored to float */
  *rot = (float)orc_angle_mod((double)(pc[2] - pp[2]));
}

void orc_odom_eval(const double x1[3], const double x2[3], const float A[4], float rho, float rot,
                   const float sig[3], double r[3], double* J) {
  /* residual_functors.h:1081-1125, analytic Jacobian. */
  const double dx = x2[0] - x1[0], dy = x2[1] - x1[1];
  const double c = cos(-x1[2]), s = sin(-x1[2]); /* Rotation2D(-theta1) */
  const double ux = c * dx - s * dy, uy = s * dx + c * dy;
  const double a00 = A[0], a01 = A[1], a10 = A[2], a11 = A[3];
  const double wx = a00 * ux + a01 * uy, wy = a10 * ux + a11 * uy;
  const double sr = sig[0], st = sig[1], sa = sig[2];
  r[0] = (wx - (double)rho) / sr;
  r[1] = wy / st;
  const double phi = x2[2] - x1[2] - (double)rot;
  r[2] = atan2(sin(phi), cos(phi)) / sa;
  if (J) {
    /* du/dx1 = -R(-t1) col0, du/dy1 = -R(-t1) col1, du/dt1 = (uy, -ux), du/dx2 = +..., */
    const double R00 = c, R01 = -s, R10 = s, R11 = c; /* R(-t1) */
    const double dux[6] = {-R00, -R01, uy, R00, R01, 0.0};
    const double duy[6] = {-R10, -R11, -ux, R10, R11, 0.0};
    for (int k = 0; k < 6; ++k) {
      J[0 * 6 + k] = (a00 * dux[k] + a01 * duy[k]) / sr;
      J[1 * 6 + k] = (a10 * dux[k] + a11 * duy[k]) / st;
      J[2 * 6 + k] = 0.0;
    }
    J[2 * 6 + 2] = -1.0 / sa;
    J[2 * 6 + 5] = 1.0 / sa;
  }
}

void orc_odom_eval_jet(const double x1[3], const double x2[3], const float A[4], float rho,
                       float rot, const float sig[3], double r[3], double* J) {
  /* literal restatement of PoseConstraint::operator() with T = Jet */
  jet p1[3], p2[3];
  for (int k = 0; k < 3; ++k) { p1[k] = jvar(x1[k], k); p2[k] = jvar(x2[k], 3 + k); }
  jet tx = jsub(p2[0], p1[0]), ty = jsub(p2[1], p1[1]);
  jet ang = jneg(p1[2]);
  jet c = jcos(ang), s = jsin(ang);
  jet ux = jsub(jmul(c, tx), jmul(s, ty));
  jet uy = jadd(jmul(s, tx), jmul(c, ty));
  jet wx = jadd(jmul(jconst(A[0]), ux), jmul(jconst(A[1]), uy));
  jet wy = jadd(jmul(jconst(A[2]), ux), jmul(jconst(A[3]), uy));
  jet r0 = jdiv(jsub(wx, jconst(rho)), jconst(sig[0]));
  jet r1 = jdiv(wy, jconst(sig[1]));
  jet phi = jsub(jsub(p2[2], p1[2]), jconst(rot));
  jet r2 = jdiv(jatan2(jsin(phi), jcos(phi)), jconst(sig[2]));
  r[0] = r0.v; r[1] = r1.v; r[2] = r2.v;
  if (J) for (int k = 0; k < 6; ++k) { J[k] = r0.d[k]; J[6 + k] = r1.d[k]; J[12 + k] = r2.d[k]; }
}

/* ===================================================================================== */
/* human-constraint rows                                                                  */
/* ===================================================================================== */
void orc_hc_target(const hitl_human_constraint_t* row, const float* a, float tgt[4]) {
  /* JointOptimization.cpp:981-989 (types 2,4), :1023-1025 / :1037-1039 (types 5,6); fp32 */
  const float ang = a[2];
  const float pax = cosf(ang), pay = sinf(ang);
  const float pex = -pay, pey = pax;
  tgt[0] = (a[0] + row->delta_parallel * pax) + row->delta_perpendicular * pex;
  tgt[1] = (a[1] + row->delta_parallel * pay) + row->delta_perpendicular * pey;
  const float t_angle = ang + row->delta_angle;
  tgt[2] = atan2f(sinf(t_angle), cosf(t_angle));
  tgt[3] = ang + row->relative_penalty_dir;
}

int orc_hc_eval(uint32_t type, const float tgt[4], const double x[3], double r[3], double* J) {
  /* residual_functors.h:1305-1411; weights Ks = Ka = 1; no angle wrap */
  const double xt = tgt[0], yt = tgt[1], tt = tgt[2];
  if (type == HITL_HC_LINE_SEGMENT) {
    r[0] = xt - x[0]; r[1] = yt - x[1]; r[2] = tt - x[2];
    if (J) { memset(J, 0, 9 * sizeof(double)); J[0] = -1; J[4] = -1; J[8] = -1; }
    return 3;
  }
  if (type == HITL_HC_COLINEAR) {
    const double cp = cos((double)tgt[3]), sp = sin((double)tgt[3]);
    r[0] = cp * (xt - x[0]) + sp * (yt - x[1]);
    r[1] = tt - x[2];
    if (J) { J[0] = -cp; J[1] = -sp; J[2] = 0; J[3] = 0; J[4] = 0; J[5] = -1; }
    return 2;
  }
  if (type == HITL_HC_PERPENDICULAR || type == HITL_HC_PARALLEL) {
    r[0] = tt - x[2];
    if (J) { J[0] = 0; J[1] = 0; J[2] = -1; }
    return 1;
  }
  return 0;
}

int orc_hc_eval_jet(uint32_t type, const float tgt[4], const double x[3], double r[3], double* J) {
  jet p[3];
  for (int k = 0; k < 3; ++k) p[k] = jvar(x[k], k);
  jet res[3]; int n = 0;
  if (type == HITL_HC_LINE_SEGMENT) {
    res[0] = jsub(jconst(tgt[0]), p[0]); res[1] = jsub(jconst(tgt[1]), p[1]);
    res[2] = jsub(jconst(tgt[2]), p[2]); n = 3;
  } else if (type == HITL_HC_COLINEAR) {
    jet xd = jconst(cos((double)tgt[3])), yd = jconst(sin((double)tgt[3]));
    res[0] = jadd(jmul(xd, jsub(jconst(tgt[0]), p[0])), jmul(yd, jsub(jconst(tgt[1]), p[1])));
    res[1] = jsub(jconst(tgt[2]), p[2]); n = 2;
  } else if (type == HITL_HC_PERPENDICULAR || type == HITL_HC_PARALLEL) {
    res[0] = jsub(jconst(tgt[2]), p[2]); n = 1;
  }
  for (int i = 0; i < n; ++i) {
    r[i] = res[i].v;
    if (J) for (int k = 0; k < 3; ++k) J[i * 3 + k] = res[i].d[k];
  }
  return n;
}

/* ===================================================================================== */
/* STF "glob" factor                                                                      */
/* ===================================================================================== */
void orc_stf_eval(const double x0[3], const double x1[3], int64_t n, const float* p0x,
                  const float* p0y, const float* p1x, const float* p1y, const float* n0x,
                  const float* n0y, const float* n1x, const float* n1y, float std_dev,
                  float corr_factor, double r[2], double* J) {
  /* residual_functors.h:786-830; analytic derivative per SURVEY.md A.2 */
  const double c0 = cos(x0[2]), s0 = sin(x0[2]), c1 = cos(x1[2]), s1 = sin(x1[2]);
  const double corr = (double)corr_factor, sd = (double)std_dev;
  double S0 = 0, S1 = 0, G0[6] = {0, 0, 0, 0, 0, 0}, G1[6] = {0, 0, 0, 0, 0, 0};
#ifdef ORC_SIMD   /* cpu_baseline build only (make native): same arithmetic per match, vector-lane summation order */
#pragma omp simd reduction(+ : S0, S1, G0[:6], G1[:6])
#endif
  for (int64_t i = 0; i < n; ++i) {
    const double ax = p0x[i], ay = p0y[i], bx = p1x[i], by = p1y[i];
    const double rp0x = c0 * ax - s0 * ay, rp0y = s0 * ax + c0 * ay;   /* R0 p0 */
    const double rp1x = c1 * bx - s1 * by, rp1y = s1 * bx + c1 * by;   /* R1 p1 */
    const double m0x = c0 * (double)n0x[i] - s0 * (double)n0y[i], m0y = s0 * (double)n0x[i] + c0 * (double)n0y[i];
    const double m1x = c1 * (double)n1x[i] - s1 * (double)n1y[i], m1y = s1 * (double)n1x[i] + c1 * (double)n1y[i];
    const double dx = (rp1x + x1[0]) - (rp0x + x0[0]);
    const double dy = (rp1y + x1[1]) - (rp0y + x0[1]);
    const double e0 = (m0x * dx + m0y * dy) * corr / sd;
    const double e1 = (m1x * dx + m1y * dy) * corr / sd;
    S0 += e0 * e0; S1 += e1 * e1;
    if (J) {
      const double k = corr / sd;
      /* perp(v) = (-vy, vx) */
      const double de0[6] = {-k * m0x, -k * m0y,
                             k * ((-m0y * dx + m0x * dy) - (m0x * (-rp0y) + m0y * rp0x)),
                             k * m0x, k * m0y, k * (m0x * (-rp1y) + m0y * rp1x)};
      const double de1[6] = {-k * m1x, -k * m1y, -k * (m1x * (-rp0y) + m1y * rp0x),
                             k * m1x, k * m1y,
                             k * ((-m1y * dx + m1x * dy) + (m1x * (-rp1y) + m1y * rp1x))};
      for (int q = 0; q < 6; ++q) { G0[q] += e0 * de0[q]; G1[q] += e1 * de1[q]; }
    }
  }
  const double dn = (double)n;
  r[0] = (S0 != 0.0) ? sqrt(S0 / dn) : 0.0;
  r[1] = (S1 != 0.0) ? sqrt(S1 / dn) : 0.0;
  if (J) for (int q = 0; q < 6; ++q) {
    J[q] = (S0 != 0.0) ? G0[q] / (dn * r[0]) : 0.0;
    J[6 + q] = (S1 != 0.0) ? G1[q] / (dn * r[1]) : 0.0;
  }
}

void orc_stf_eval_jet(const double x0[3], const double x1[3], int64_t n, const float* p0x,
                      const float* p0y, const float* p1x, const float* p1y, const float* n0x,
                      const float* n0y, const float* n1x, const float* n1y, float std_dev,
                      float corr_factor, double r[2], double* J) {
  /* literal restatement of PointToPointGlobConstraint::operator() with T = Jet */
  jet q0[3], q1[3];
  for (int k = 0; k < 3; ++k) { q0[k] = jvar(x0[k], k); q1[k] = jvar(x1[k], 3 + k); }
  jet c0 = jcos(q0[2]), s0 = jsin(q0[2]), c1 = jcos(q1[2]), s1 = jsin(q1[2]);
  jet res0 = jconst(0), res1 = jconst(0);
  for (int64_t i = 0; i < n; ++i) {
    jet ax = jconst(p0x[i]), ay = jconst(p0y[i]), bx = jconst(p1x[i]), by = jconst(p1y[i]);
    jet p0gx = jadd(jsub(jmul(c0, ax), jmul(s0, ay)), q0[0]);
    jet p0gy = jadd(jadd(jmul(s0, ax), jmul(c0, ay)), q0[1]);
    jet p1gx = jadd(jsub(jmul(c1, bx), jmul(s1, by)), q1[0]);
    jet p1gy = jadd(jadd(jmul(s1, bx), jmul(c1, by)), q1[1]);
    jet m0x = jsub(jmul(c0, jconst(n0x[i])), jmul(s0, jconst(n0y[i])));
    jet m0y = jadd(jmul(s0, jconst(n0x[i])), jmul(c0, jconst(n0y[i])));
    jet m1x = jsub(jmul(c1, jconst(n1x[i])), jmul(s1, jconst(n1y[i])));
    jet m1y = jadd(jmul(s1, jconst(n1x[i])), jmul(c1, jconst(n1y[i])));
    jet dx = jsub(p1gx, p0gx), dy = jsub(p1gy, p0gy);
    jet e0 = jdiv(jmul(jadd(jmul(m0x, dx), jmul(m0y, dy)), jconst(corr_factor)), jconst(std_dev));
    jet e1 = jdiv(jmul(jadd(jmul(m1x, dx), jmul(m1y, dy)), jconst(corr_factor)), jconst(std_dev));
    res0 = jadd(res0, jmul(e0, e0));
    res1 = jadd(res1, jmul(e1, e1));
  }
  if (res0.v != 0.0) res0 = jsqrt(jdiv(res0, jconst((double)n)));
  if (res1.v != 0.0) res1 = jsqrt(jdiv(res1, jconst((double)n)));
  r[0] = res0.v; r[1] = res1.v;
  if (J) for (int k = 0; k < 6; ++k) { J[k] = res0.d[k]; J[6 + k] = res1.d[k]; }
}

/* ===================================================================================== */
/* solver object                                                                          */
/* ===================================================================================== */
struct orc_solver {
  hitl_options opt;
  int use_jets;
  int64_t n;            /* poses */
  int64_t* obs_offset;  /* n+1 */
  float *px, *py, *nx, *ny;
  float* obs4;          /* (px, py, nx, ny) per observation: one cache line per gathered point in gather_pair */
  float* cov9;
  int64_t n_hc;
  hitl_human_constraint_t* hc;
  int64_t n_pairs, n_matches;
  int32_t *pair_src, *pair_tgt;
  int64_t* match_offset;
  int32_t *m0, *m1;
  /* pose -> incident pair entries (2*pair + side), ascending pair index; built on first use */
  int64_t* inc_off;
  int64_t* inc_ent;
  /* wall-clock per phase of the last orc_solve (ms): 0 residual/Jacobian evaluation of the scan-to-scan pairs,
   * 1 accumulation of J^T J / gradient, 2 scaling + LM diagonal, 3 banded Cholesky, 4 step bookkeeping */
  double phase_ms[8];
};

static void free_pairs(orc_solver* s) {
  free(s->pair_src); free(s->pair_tgt); free(s->match_offset); free(s->m0); free(s->m1);
  s->pair_src = s->pair_tgt = NULL; s->match_offset = NULL; s->m0 = s->m1 = NULL;
  free(s->inc_off); free(s->inc_ent); s->inc_off = s->inc_ent = NULL;
  s->n_pairs = s->n_matches = 0;
}

int orc_get_phase_ms(orc_solver* s, double out[8]) {
  if (!s || !out) return HITL_ERR_INVALID_ARGUMENT;
  memcpy(out, s->phase_ms, sizeof(s->phase_ms));
  return HITL_OK;
}

int orc_create(const hitl_options* opt, orc_solver** out) {
  if (!out) return HITL_ERR_INVALID_ARGUMENT;
  orc_solver* s = (orc_solver*)calloc(1, sizeof(orc_solver));
  if (!s) return HITL_ERR_OUT_OF_MEMORY;
  if (opt) s->opt = *opt; else orc_options_default(&s->opt);
  *out = s;
  return HITL_OK;
}

void orc_destroy(orc_solver* s) {
  if (!s) return;
  free(s->obs_offset); free(s->px); free(s->py); free(s->nx); free(s->ny); free(s->cov9); free(s->obs4);
  free(s->hc); free_pairs(s); free(s);
}

int orc_set_num_threads(int n) {
#ifdef _OPENMP
  int prev = omp_get_max_threads();
  if (n > 0) omp_set_num_threads(n);
  return prev;
#else
  (void)n; return 1;
#endif
}

int orc_set_use_jets(orc_solver* s, int use) { s->use_jets = use; return HITL_OK; }

static void* dup_mem(const void* src, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (p && src && bytes) memcpy(p, src, bytes);
  return p;
}

int orc_set_map(orc_solver* s, int64_t n, const int64_t* off, const float* px, const float* py,
                const float* nx, const float* ny, const float* cov9) {
  if (!s || n < 1 || !off) return HITL_ERR_INVALID_ARGUMENT;
  free(s->obs_offset); free(s->px); free(s->py); free(s->nx); free(s->ny); free(s->cov9); free(s->obs4);
  s->cov9 = NULL; s->obs4 = NULL;
  s->n = n;
  const int64_t O = off[n];
  s->obs_offset = (int64_t*)dup_mem(off, (size_t)(n + 1) * sizeof(int64_t));
  s->px = (float*)dup_mem(px, (size_t)O * 4); s->py = (float*)dup_mem(py, (size_t)O * 4);
  s->nx = (float*)dup_mem(nx, (size_t)O * 4); s->ny = (float*)dup_mem(ny, (size_t)O * 4);
  if (cov9) s->cov9 = (float*)dup_mem(cov9, (size_t)n * 9 * 4);
  s->obs4 = (float*)malloc((size_t)(O ? O : 1) * 16);
  for (int64_t k = 0; k < O; ++k) { s->obs4[4 * k] = px[k]; s->obs4[4 * k + 1] = py[k]; s->obs4[4 * k + 2] = nx[k]; s->obs4[4 * k + 3] = ny[k]; }
  free_pairs(s);
  return HITL_OK;
}

int orc_set_human_constraints(orc_solver* s, int64_t n, const hitl_human_constraint_t* rows) {
  if (!s || n < 0) return HITL_ERR_INVALID_ARGUMENT;
  free(s->hc);
  s->hc = (hitl_human_constraint_t*)dup_mem(rows, (size_t)n * sizeof(*rows));
  s->n_hc = n;
  return HITL_OK;
}

int orc_set_correspondences(orc_solver* s, int64_t P, const int32_t* ps, const int32_t* pt,
                            const int64_t* mo, const int32_t* i0, const int32_t* i1) {
  if (!s || P < 0) return HITL_ERR_INVALID_ARGUMENT;
  free_pairs(s);
  const int64_t C = P ? mo[P] : 0;
  for (int64_t p = 0; p < P; ++p) {
    if (ps[p] < 0 || ps[p] >= s->n || pt[p] < 0 || pt[p] >= s->n || ps[p] == pt[p] ||
        mo[p + 1] <= mo[p])
      return HITL_ERR_INVALID_ARGUMENT; /* CHECK_GT(points0.size(), 0), residual_functors.h:783 */
  }
  s->n_pairs = P; s->n_matches = C;
  s->pair_src = (int32_t*)dup_mem(ps, (size_t)P * 4); s->pair_tgt = (int32_t*)dup_mem(pt, (size_t)P * 4);
  s->match_offset = (int64_t*)dup_mem(mo, (size_t)(P + 1) * 8);
  if (!P) s->match_offset[0] = 0;
  s->m0 = (int32_t*)dup_mem(i0, (size_t)C * 4); s->m1 = (int32_t*)dup_mem(i1, (size_t)C * 4);
  return HITL_OK;
}

int orc_get_correspondence_counts(orc_solver* s, int64_t* P, int64_t* C) {
  if (P) *P = s->n_pairs;
  if (C) *C = s->n_matches;
  return HITL_OK;
}

int orc_get_correspondences(orc_solver* s, int32_t* ps, int32_t* pt, int64_t* mo, int32_t* i0, int32_t* i1) {
  memcpy(ps, s->pair_src, (size_t)s->n_pairs * 4); memcpy(pt, s->pair_tgt, (size_t)s->n_pairs * 4);
  if (s->n_pairs) memcpy(mo, s->match_offset, (size_t)(s->n_pairs + 1) * 8); else mo[0] = 0;
  memcpy(i0, s->m0, (size_t)s->n_matches * 4); memcpy(i1, s->m1, (size_t)s->n_matches * 4);
  return HITL_OK;
}

/* ===================================================================================== */
/* correspondence search                                                                  */
/* ===================================================================================== */
/* Relative transform source i -> target j (JointOptimization.cpp:296-305, :601-605) in fp32, in the operation order of
 * the Eigen expression `target_tf.inverse(Eigen::Affine) * source_tf` (round 3; rounds 1-2 used R_j^T (t_i - t_j), which
 * is the same map but differs in the last bits -- up to ~1e-5 m at coordinates of 100 m, enough to flip near-ties):
 *   tf = Translation2Df(x, y) * Rotation2Df(theta): linear [[c, -s], [s, c]], translation (x, y);
 *   inverse(Affine): linear^-1 by the 2x2 cofactor formula with invdet = 1 / (m00 m11 - m10 m01), translation -linear^-1 t;
 *   Transform * Transform: linear = L1 L2, translation = L1 t2 + t1; every 2-term dot product is a*b + c*d. */
typedef struct { float m00, m01, m10, m11, tx, ty, cr, sr; } rel_tf;

static rel_tf make_rel_tf(const float* pi, const float* pj) {
  rel_tf t;
  const float ci = cosf(pi[2]), si = sinf(pi[2]), cj = cosf(pj[2]), sj = sinf(pj[2]);
  const float nsj = -sj, nsi = -si;
  const float det = cj * cj - sj * nsj;
  const float invdet = 1.0f / det;
  const float a00 = cj * invdet, a01 = (-nsj) * invdet, a10 = (-sj) * invdet, a11 = cj * invdet;     /* target linear^-1 */
  t.m00 = a00 * ci + a01 * si;   t.m01 = a00 * nsi + a01 * ci;
  t.m10 = a10 * ci + a11 * si;   t.m11 = a10 * nsi + a11 * ci;
  const float ix = (-a00) * pj[0] + (-a01) * pj[1], iy = (-a10) * pj[0] + (-a11) * pj[1];            /* -linear^-1 * t_j */
  t.tx = (a00 * pi[0] + a01 * pi[1]) + ix;
  t.ty = (a10 * pi[0] + a11 * pi[1]) + iy;
  const float dth = (float)((double)pj[2] - (double)pi[2]);
  t.cr = cosf(dth); t.sr = sinf(dth);
  return t;
}

/* ---- KD-tree restatement (kdtree.cpp:37-197) ---- */
typedef struct { float x, y, nx, ny; int32_t index; } kd_value;
typedef struct kd_node { kd_value v; int split; struct kd_node *left, *right; } kd_node;

static int kd_split_plane(const kd_value* vals, int64_t n) { /* kdtree.cpp:37-69, fp32 */
  float mx = 0, my = 0;
  for (int64_t i = 0; i < n; ++i) { mx = mx + vals[i].x; my = my + vals[i].y; }
  mx = mx / (float)n; my = my / (float)n;
  float sx = 0, sy = 0;
  for (int64_t i = 0; i < n; ++i) {
    sx = sx + (vals[i].x - mx) * (vals[i].x - mx);
    sy = sy + (vals[i].y - my) * (vals[i].y - my);
  }
  int plane = 0; float best = 0.0f;
  if (sx > best) { plane = 0; best = sx; }
  if (sy > best) { plane = 1; best = sy; }
  return plane;
}

static void kd_merge_sort(kd_value* a, kd_value* tmp, int64_t n, int dim) {
  /* std::sort at kdtree.cpp:115 is unstable with unspecified tie order; a stable sort is
   * one valid outcome of it. */
  if (n < 2) return;
  const int64_t h = n / 2;
  kd_merge_sort(a, tmp, h, dim); kd_merge_sort(a + h, tmp, n - h, dim);
  int64_t i = 0, j = h, k = 0;
  while (i < h && j < n) {
    const float vi = dim ? a[i].y : a[i].x, vj = dim ? a[j].y : a[j].x;
    tmp[k++] = (vj < vi) ? a[j++] : a[i++];
  }
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, (size_t)n * sizeof(kd_value));
}

static kd_node* kd_build(kd_value* vals, kd_value* tmp, int64_t n, kd_node** pool) {
  if (n <= 0) return NULL;
  kd_node* nd = (*pool)++;
  nd->split = kd_split_plane(vals, n);
  kd_merge_sort(vals, tmp, n, nd->split);
  const int64_t ind = n / 2;
  nd->v = vals[ind];
  nd->left = (ind > 0) ? kd_build(vals, tmp, ind, pool) : NULL;
  nd->right = (ind < n - 1) ? kd_build(vals + ind + 1, tmp, n - ind - 1, pool) : NULL;
  return nd;
}

static float kd_find_nearest_point_normal(const kd_node* nd, float qx, float qy, float thr, kd_value* out) {
  /* kdtree.cpp:141-197 */
  float best = FLT_MAX;
  {
    const float dx = nd->v.x - qx, dy = nd->v.y - qy;
    if (dx * dx + dy * dy < thr * thr) {
      *out = nd->v;
      best = fabsf(nd->v.nx * (qx - nd->v.x) + nd->v.ny * (qy - nd->v.y));
      if (best < FLT_MIN) return 0.0f;
    }
  }
  const float pd = (nd->split ? qy : qx) - (nd->split ? nd->v.y : nd->v.x);
  const kd_node* other = NULL;
  if (pd <= 0.0f && nd->left) {
    kd_value c; const float d = kd_find_nearest_point_normal(nd->left, qx, qy, thr, &c);
    if (d < best) { best = d; *out = c; }
    other = nd->right;
  }
  if (pd >= 0.0f && nd->right) {
    kd_value c; const float d = kd_find_nearest_point_normal(nd->right, qx, qy, thr, &c);
    if (d < best) { best = d; *out = c; }
    other = nd->left;
  }
  if (other && pd != 0.0f && fabsf(pd) < (best < thr ? best : thr)) {
    kd_value c; const float d = kd_find_nearest_point_normal(other, qx, qy, thr, &c);
    if (d < best) { best = d; *out = c; }
  }
  return best;
}

/* brute-force definition of the same query: among target points within the Euclidean gate,
 * the one with the smallest point-to-plane distance (first in index order on ties). */
static float bf_find_nearest_point_normal(const float* tx, const float* ty, const float* tnx,
                                          const float* tny, int64_t m, float qx, float qy,
                                          float thr, int32_t* out_idx) {
  float best = FLT_MAX; int32_t bi = -1;
  const float thr2 = thr * thr;
  for (int64_t t = 0; t < m; ++t) {
    const float dx = qx - tx[t], dy = qy - ty[t];
    const float dsq = dx * dx + dy * dy;
    if (dsq < thr2) {
      const float sc = fabsf(tnx[t] * dx + tny[t] * dy);
      if (sc < best) { best = sc; bi = (int32_t)t; }
    }
  }
  *out_idx = bi;
  return best;
}

typedef struct { int64_t cap, len; int32_t* a; int32_t* b; } ivec2;
static void ivec2_push(ivec2* v, int32_t a, int32_t b) {
  if (v->len == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 256;
    v->a = (int32_t*)realloc(v->a, (size_t)v->cap * 4); v->b = (int32_t*)realloc(v->b, (size_t)v->cap * 4);
  }
  v->a[v->len] = a; v->b[v->len] = b; v->len++;
}

int orc_find_correspondences(orc_solver* s, const float* poses, int64_t window, int method) {
  /* JointOptimization.cpp:561-642 */
  if (!s || !poses || !s->obs_offset) return HITL_ERR_INVALID_ARGUMENT;
  const int64_t n = s->n;
  if (window <= 0 || window > n) window = n;
  const float thr = s->opt.point_match_threshold;
  const float min_cos = cosf(s->opt.max_stf_angle_error);
  const int cap = s->opt.max_correspondences_per_point;
  const int skip = s->opt.num_skip_readings > 0 ? s->opt.num_skip_readings : 1;
  const int64_t min_corr = s->opt.min_inter_pose_correspondence;

  kd_node** roots = NULL; kd_node** pools = NULL;
  if (method == 1) { /* BuildKDTrees, JointOptimization.cpp:514-537 */
    roots = (kd_node**)calloc((size_t)n, sizeof(kd_node*));
    pools = (kd_node**)calloc((size_t)n, sizeof(kd_node*));
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; ++i) {
      const int64_t o = s->obs_offset[i], m = s->obs_offset[i + 1] - o;
      if (m <= 0) continue;
      kd_value* vals = (kd_value*)malloc((size_t)m * sizeof(kd_value));
      kd_value* tmp = (kd_value*)malloc((size_t)m * sizeof(kd_value));
      for (int64_t k = 0; k < m; ++k) {
        vals[k].x = s->px[o + k]; vals[k].y = s->py[o + k];
        vals[k].nx = s->nx[o + k]; vals[k].ny = s->ny[o + k]; vals[k].index = (int32_t)k;
      }
      pools[i] = (kd_node*)malloc((size_t)m * sizeof(kd_node));
      kd_node* pool = pools[i];
      roots[i] = kd_build(vals, tmp, m, &pool);
      free(vals); free(tmp);
    }
  }

  /* per-source-pose results, concatenated in pose order afterwards (:636-641) */
  ivec2* pose_pairs = (ivec2*)calloc((size_t)n, sizeof(ivec2));     /* (tgt, count) */
  ivec2* pose_matches = (ivec2*)calloc((size_t)n, sizeof(ivec2));   /* (k0, k1)    */
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t i = 0; i < n; ++i) {
    const int64_t oi = s->obs_offset[i], mi = s->obs_offset[i + 1] - oi;
    int* cnt = (int*)calloc((size_t)(mi ? mi : 1), sizeof(int));
    const int64_t j0 = (i - window < 0) ? 0 : i - window;
    const int64_t j1 = (i + window > n - 1) ? n - 1 : i + window;
    for (int64_t j = j0; j <= j1; ++j) {
      if (j == i) continue;
      const int64_t oj = s->obs_offset[j], mj = s->obs_offset[j + 1] - oj;
      const rel_tf tf = make_rel_tf(poses + 3 * i, poses + 3 * j);
      const int64_t start = pose_matches[i].len;
      for (int64_t k = 0; k < mi; k += skip) {
        if (cnt[k] >= cap) continue;
        const float ax = s->px[oi + k], ay = s->py[oi + k];
        const float qx = (tf.m00 * ax + tf.m01 * ay) + tf.tx;
        const float qy = (tf.m10 * ax + tf.m11 * ay) + tf.ty;
        const float nqx = tf.cr * s->nx[oi + k] - tf.sr * s->ny[oi + k];
        const float nqy = tf.sr * s->nx[oi + k] + tf.cr * s->ny[oi + k];
        float dist; int32_t bi = -1; float bnx = 0, bny = 0;
        if (method == 1) {
          kd_value nb; nb.index = 0; nb.nx = 0; nb.ny = 0; /* default-constructed KDNodeValue */
          dist = (mj > 0) ? kd_find_nearest_point_normal(roots[j], qx, qy, thr, &nb) : FLT_MAX;
          bi = nb.index; bnx = nb.nx; bny = nb.ny;
        } else {
          dist = bf_find_nearest_point_normal(s->px + oj, s->py + oj, s->nx + oj, s->ny + oj, mj, qx, qy, thr, &bi);
          if (bi >= 0) { bnx = s->nx[oj + bi]; bny = s->ny[oj + bi]; }
        }
        const float cosang = bnx * nqx + bny * nqy;
        if (dist < thr && cosang > min_cos) {
          ivec2_push(&pose_matches[i], (int32_t)k, bi);
          ++cnt[k];
        }
      }
      const int64_t got = pose_matches[i].len - start;
      if (got > min_corr) ivec2_push(&pose_pairs[i], (int32_t)j, (int32_t)got);
      else pose_matches[i].len = start; /* pair dropped (:632-634); cnt[] keeps its increments */
    }
    free(cnt);
  }
  int64_t P = 0, C = 0;
  for (int64_t i = 0; i < n; ++i) { P += pose_pairs[i].len; C += pose_matches[i].len; }
  free_pairs(s);
  s->n_pairs = P; s->n_matches = C;
  s->pair_src = (int32_t*)malloc((size_t)(P ? P : 1) * 4); s->pair_tgt = (int32_t*)malloc((size_t)(P ? P : 1) * 4);
  s->match_offset = (int64_t*)malloc((size_t)(P + 1) * 8);
  s->m0 = (int32_t*)malloc((size_t)(C ? C : 1) * 4); s->m1 = (int32_t*)malloc((size_t)(C ? C : 1) * 4);
  int64_t p = 0, c = 0;
  s->match_offset[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t local = 0;
    for (int64_t q = 0; q < pose_pairs[i].len; ++q) {
      s->pair_src[p] = (int32_t)i; s->pair_tgt[p] = pose_pairs[i].a[q];
      const int64_t cntq = pose_pairs[i].b[q];
      memcpy(s->m0 + c, pose_matches[i].a + local, (size_t)cntq * 4);
      memcpy(s->m1 + c, pose_matches[i].b + local, (size_t)cntq * 4);
      c += cntq; local += cntq; ++p;
      s->match_offset[p] = c;
    }
    free(pose_pairs[i].a); free(pose_pairs[i].b); free(pose_matches[i].a); free(pose_matches[i].b);
  }
  free(pose_pairs); free(pose_matches);
  if (roots) { for (int64_t i = 0; i < n; ++i) free(pools[i]); free(pools); free(roots); }
  return HITL_OK;
}

/* ===================================================================================== */
/* linearisation: cost, gradient, banded J^T J                                            */
/* ===================================================================================== */
typedef struct {
  int64_t n;         /* poses */
  int64_t dim;       /* 3n */
  int64_t bw;        /* scalar half bandwidth */
  double* H;         /* dim x (bw+1), lower band: H[i*(bw+1) + (j-i+bw)], i-bw <= j <= i */
  double* g;         /* dim */
  double cost;
  double* pair_r;    /* 2 per pair */
  double* pair_J;    /* 12 per pair */
  int num_residuals, num_hc_residuals;
} lin_sys;

static inline double* Hat(lin_sys* L, int64_t i, int64_t j) { /* i >= j */
  return &L->H[i * (L->bw + 1) + (j - i + L->bw)];
}

static void lin_alloc(lin_sys* L, int64_t n, int64_t band_poses, int64_t n_pairs) {
  L->n = n; L->dim = 3 * n; L->bw = 3 * (band_poses + 1) - 1;
  if (L->bw > L->dim - 1) L->bw = L->dim - 1;
  if (L->bw < 2) L->bw = 2;
  L->H = (double*)malloc((size_t)L->dim * (size_t)(L->bw + 1) * sizeof(double));
  L->g = (double*)malloc((size_t)L->dim * sizeof(double));
  L->pair_r = (double*)malloc((size_t)(n_pairs ? n_pairs : 1) * 2 * sizeof(double));
  L->pair_J = (double*)malloc((size_t)(n_pairs ? n_pairs : 1) * 12 * sizeof(double));
}
static void lin_free(lin_sys* L) { free(L->H); free(L->g); free(L->pair_r); free(L->pair_J); }

static double now_ms(void) {
#ifdef _OPENMP
  return omp_get_wtime() * 1e3;
#else
  return 0.0;
#endif
}

/* add J^T J and J^T r of one residual block touching poses a (cols 0..2) and b (cols 3..5);
 * b < 0 for unary blocks (J is nres x 3).  Pose 0 is constant: its columns are dropped.
 * only >= 0 restricts the update to the rows (columns p) of that pose: the per-pose gather below calls it once per
 * incident block, in the order a serial sweep over all blocks would reach that pose, so every entry of H and g sees the
 * same additions in the same order as the serial sweep (bit-identical, thread count does not matter). */
static void accumulate_block(lin_sys* L, int64_t a, int64_t b, int nres, const double* r, const double* J, int64_t only) {
  const int ncol = (b >= 0) ? 6 : 3;
  int64_t gi[6];
  for (int k = 0; k < 3; ++k) { gi[k] = 3 * a + k; if (b >= 0) gi[3 + k] = 3 * b + k; }
  for (int p = 0; p < ncol; ++p) {
    const int64_t ip = gi[p];
    if (ip < 3) continue;
    if (only >= 0 && ip / 3 != only) continue;
    double gs = 0;
    for (int q = 0; q < nres; ++q) gs += J[q * ncol + p] * r[q];
    L->g[ip] += gs;
    for (int q = 0; q < ncol; ++q) {
      const int64_t iq = gi[q];
      if (iq < 3 || iq > ip) continue;
      double hs = 0;
      for (int m = 0; m < nres; ++m) hs += J[m * ncol + p] * J[m * ncol + q];
      *Hat(L, ip, iq) += hs;
    }
  }
}

typedef struct { float A[4]; float rho, rot; } odom_meas;

static int64_t max_pair_distance(const orc_solver* s) {
  int64_t d = 1;
  for (int64_t p = 0; p < s->n_pairs; ++p) {
    int64_t q = llabs((long long)s->pair_src[p] - (long long)s->pair_tgt[p]);
    if (q > d) d = q;
  }
  return d;
}

static void gather_pair(const orc_solver* s, int64_t p, float** buf, int64_t* cap) {
  const int64_t m = s->match_offset[p + 1] - s->match_offset[p];
  if (m > *cap) { *cap = m * 2; *buf = (float*)realloc(*buf, (size_t)(*cap) * 8 * sizeof(float)); }
  float* b = *buf; const int64_t cp = *cap;
  const int64_t o0 = s->obs_offset[s->pair_src[p]], o1 = s->obs_offset[s->pair_tgt[p]];
  const int32_t* i0 = s->m0 + s->match_offset[p]; const int32_t* i1 = s->m1 + s->match_offset[p];
  for (int64_t q = 0; q < m; ++q) {
    const float* a = s->obs4 + 4 * (o0 + i0[q]); const float* c = s->obs4 + 4 * (o1 + i1[q]);
    b[0 * cp + q] = a[0]; b[1 * cp + q] = a[1];
    b[2 * cp + q] = c[0]; b[3 * cp + q] = c[1];
    b[4 * cp + q] = a[2]; b[5 * cp + q] = a[3];
    b[6 * cp + q] = c[2]; b[7 * cp + q] = c[3];
  }
}

/* pose -> (pair, side) incidence, ascending pair index within a pose; also human rows per pose in row order */
static void build_incidence(orc_solver* s) {
  if (s->inc_off) return;
  const int64_t n = s->n, P = s->n_pairs;
  s->inc_off = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
  s->inc_ent = (int64_t*)malloc((size_t)(2 * P + 1) * sizeof(int64_t));
  for (int64_t p = 0; p < P; ++p) { s->inc_off[s->pair_src[p] + 1]++; s->inc_off[s->pair_tgt[p] + 1]++; }
  for (int64_t i = 0; i < n; ++i) s->inc_off[i + 1] += s->inc_off[i];
  int64_t* fill = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
  memcpy(fill, s->inc_off, (size_t)(n + 1) * sizeof(int64_t));
  for (int64_t p = 0; p < P; ++p) { s->inc_ent[fill[s->pair_src[p]]++] = 2 * p; s->inc_ent[fill[s->pair_tgt[p]]++] = 2 * p + 1; }
  free(fill);
}

static void linearize(orc_solver* s, int mode, const odom_meas* om, const float* hc_tgt,
                      const double* x, lin_sys* L) {
  const int64_t n = s->n;
  const int with_stf = (mode == HITL_MODE_STF_ONLY || mode == HITL_MODE_JOINT) && s->n_pairs > 0;
  const int with_odom_hc = (mode == HITL_MODE_ODOM_HC || mode == HITL_MODE_JOINT);
  L->cost = 0; L->num_residuals = 0; L->num_hc_residuals = 0;
  double t0 = now_ms();

  if (with_stf) {
    /* per-pair residual blocks in parallel (the reference: OpenMP inside the functor,
     * residual_functors.h:801) */
#pragma omp parallel
    {
      float* buf = NULL; int64_t cap = 0;
#pragma omp for schedule(dynamic, 64)
      for (int64_t p = 0; p < s->n_pairs; ++p) {
        gather_pair(s, p, &buf, &cap);
        const int64_t m = s->match_offset[p + 1] - s->match_offset[p];
        const double* x0 = x + 3 * s->pair_src[p]; const double* x1 = x + 3 * s->pair_tgt[p];
        if (s->use_jets)
          orc_stf_eval_jet(x0, x1, m, buf, buf + cap, buf + 2 * cap, buf + 3 * cap, buf + 4 * cap, buf + 5 * cap,
                           buf + 6 * cap, buf + 7 * cap, s->opt.laser_std_dev, s->opt.point_correlation_factor,
                           L->pair_r + 2 * p, L->pair_J + 12 * p);
        else
          orc_stf_eval(x0, x1, m, buf, buf + cap, buf + 2 * cap, buf + 3 * cap, buf + 4 * cap, buf + 5 * cap,
                       buf + 6 * cap, buf + 7 * cap, s->opt.laser_std_dev, s->opt.point_correlation_factor,
                       L->pair_r + 2 * p, L->pair_J + 12 * p);
      }
      free(buf);
    }
    build_incidence(s);
  }
  double t1 = now_ms();
  s->phase_ms[0] += t1 - t0;

  /* odometry edges (AddOdometryConstraints :743-823) and human rows (AddHumanConstraints :975-1052), evaluated once */
  double *od_r = NULL, *od_J = NULL, *hc_r = NULL, *hc_J = NULL;
  int* hc_n = NULL; int64_t *hc_poff = NULL, *hc_pent = NULL;
  if (with_odom_hc) {
    const float sig[3] = {s->opt.odom_radial_std_dev, s->opt.odom_tangential_std_dev, s->opt.odom_angular_std_dev};
    od_r = (double*)malloc((size_t)n * 3 * sizeof(double)); od_J = (double*)malloc((size_t)n * 18 * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t i = 1; i < n; ++i) {
      if (s->use_jets) orc_odom_eval_jet(x + 3 * (i - 1), x + 3 * i, om[i].A, om[i].rho, om[i].rot, sig, od_r + 3 * i, od_J + 18 * i);
      else orc_odom_eval(x + 3 * (i - 1), x + 3 * i, om[i].A, om[i].rho, om[i].rot, sig, od_r + 3 * i, od_J + 18 * i);
    }
    const int64_t Hn = s->n_hc;
    hc_r = (double*)malloc((size_t)(Hn + 1) * 3 * sizeof(double)); hc_J = (double*)malloc((size_t)(Hn + 1) * 9 * sizeof(double));
    hc_n = (int*)malloc((size_t)(Hn + 1) * sizeof(int));
    hc_poff = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t)); hc_pent = (int64_t*)malloc((size_t)(Hn + 1) * sizeof(int64_t));
    for (int64_t h = 0; h < Hn; ++h) {
      const int64_t c = s->hc[h].constrained_pose_id;
      hc_n[h] = s->use_jets ? orc_hc_eval_jet(s->hc[h].type, hc_tgt + 4 * h, x + 3 * c, hc_r + 3 * h, hc_J + 9 * h)
                            : orc_hc_eval(s->hc[h].type, hc_tgt + 4 * h, x + 3 * c, hc_r + 3 * h, hc_J + 9 * h);
      hc_poff[c + 1]++;
    }
    for (int64_t i = 0; i < n; ++i) hc_poff[i + 1] += hc_poff[i];
    int64_t* fill = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
    memcpy(fill, hc_poff, (size_t)(n + 1) * sizeof(int64_t));
    for (int64_t h = 0; h < Hn; ++h) hc_pent[fill[s->hc[h].constrained_pose_id]++] = h;   /* row order within a pose */
    free(fill);
  }

  /* J^T J and gradient, gathered per pose: every pose's rows are written by one thread, blocks arrive in the order
   * of a serial sweep (pairs by index, edge (i-1,i), edge (i,i+1), human rows) */
  const int64_t w = L->bw + 1;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < n; ++i) {
    memset(L->H + (size_t)(3 * i) * (size_t)w, 0, (size_t)3 * (size_t)w * sizeof(double));
    L->g[3 * i] = L->g[3 * i + 1] = L->g[3 * i + 2] = 0.0;
    if (with_stf)
      for (int64_t e = s->inc_off[i]; e < s->inc_off[i + 1]; ++e) {
        const int64_t p = s->inc_ent[e] >> 1;
        accumulate_block(L, s->pair_src[p], s->pair_tgt[p], 2, L->pair_r + 2 * p, L->pair_J + 12 * p, i);
      }
    if (with_odom_hc) {
      if (i >= 1) accumulate_block(L, i - 1, i, 3, od_r + 3 * i, od_J + 18 * i, i);
      if (i + 1 < n) accumulate_block(L, i, i + 1, 3, od_r + 3 * (i + 1), od_J + 18 * (i + 1), i);
      for (int64_t e = hc_poff[i]; e < hc_poff[i + 1]; ++e) {
        const int64_t h = hc_pent[e];
        if (hc_n[h] > 0) accumulate_block(L, i, -1, hc_n[h], hc_r + 3 * h, hc_J + 9 * h, i);
      }
    }
  }
  /* cost = 1/2 sum r^2 in block order (serial: a few hundred thousand additions) */
  if (with_stf)
    for (int64_t p = 0; p < s->n_pairs; ++p) {
      for (int q = 0; q < 2; ++q) L->cost += 0.5 * L->pair_r[2 * p + q] * L->pair_r[2 * p + q];
      L->num_residuals += 2;
    }
  if (with_odom_hc) {
    for (int64_t i = 1; i < n; ++i) {
      for (int q = 0; q < 3; ++q) L->cost += 0.5 * od_r[3 * i + q] * od_r[3 * i + q];
      L->num_residuals += 3;
    }
    for (int64_t h = 0; h < s->n_hc; ++h) {
      for (int q = 0; q < hc_n[h]; ++q) L->cost += 0.5 * hc_r[3 * h + q] * hc_r[3 * h + q];
      L->num_residuals += hc_n[h]; L->num_hc_residuals += hc_n[h];
    }
    free(od_r); free(od_J); free(hc_r); free(hc_J); free(hc_n); free(hc_poff); free(hc_pent);
  }
  /* pose 0 constant: identity block so the banded factorisation stays well posed */
  for (int k = 0; k < 3; ++k) { *Hat(L, k, k) = 1.0; L->g[k] = 0.0; }
  s->phase_ms[1] += now_ms() - t1;
}

static int check_problem(const orc_solver* s, int mode) {
  if (!s || !s->obs_offset) return HITL_ERR_NOT_READY;
  if (mode < 0 || mode > 2) return HITL_ERR_INVALID_ARGUMENT;
  for (int64_t h = 0; h < s->n_hc; ++h) {
    if (s->hc[h].constrained_pose_id < 0 || s->hc[h].constrained_pose_id >= s->n ||
        s->hc[h].anchor_pose_id < 0 || s->hc[h].anchor_pose_id >= s->n)
      return HITL_ERR_INVALID_ARGUMENT;
  }
  return HITL_OK;
}

static void setup_factors(const orc_solver* s, const float* poses, odom_meas** om_out, float** tgt_out) {
  odom_meas* om = (odom_meas*)calloc((size_t)s->n, sizeof(odom_meas));
  for (int64_t i = 1; i < s->n; ++i) orc_odom_setup(poses + 3 * (i - 1), poses + 3 * i, om[i].A, &om[i].rho, &om[i].rot);
  float* tgt = (float*)calloc((size_t)(s->n_hc ? s->n_hc : 1) * 4, sizeof(float));
  for (int64_t h = 0; h < s->n_hc; ++h) orc_hc_target(&s->hc[h], poses + 3 * s->hc[h].anchor_pose_id, tgt + 4 * h);
  *om_out = om; *tgt_out = tgt;
}

int orc_linearize(orc_solver* s, int32_t mode, const float* poses_ref, const double* x, double* cost,
                  double* gradient, double* pair_r, double* pair_J) {
  int st = check_problem(s, mode); if (st) return st;
  odom_meas* om; float* tgt; setup_factors(s, poses_ref, &om, &tgt);
  lin_sys L; lin_alloc(&L, s->n, max_pair_distance(s), s->n_pairs);
  linearize(s, mode, om, tgt, x, &L);
  if (cost) *cost = L.cost;
  if (gradient) memcpy(gradient, L.g, (size_t)L.dim * sizeof(double));
  const int with_stf = (mode != HITL_MODE_ODOM_HC);
  if (pair_r && with_stf) memcpy(pair_r, L.pair_r, (size_t)s->n_pairs * 2 * sizeof(double));
  if (pair_J && with_stf) memcpy(pair_J, L.pair_J, (size_t)s->n_pairs * 12 * sizeof(double));
  lin_free(&L); free(om); free(tgt);
  return HITL_OK;
}

int orc_evaluate(orc_solver* s, int32_t mode, const float* poses, double* cost, double* gradient) {
  int st = check_problem(s, mode); if (st) return st;
  double* x = (double*)malloc((size_t)s->n * 3 * sizeof(double));
  for (int64_t i = 0; i < 3 * s->n; ++i) x[i] = (double)poses[i];
  st = orc_linearize(s, mode, poses, x, cost, gradient, NULL, NULL);
  free(x);
  return st;
}

int orc_assemble_dense(orc_solver* s, int32_t mode, const float* poses_ref, const double* x, double* H,
                       double* g, double* cost) {
  int st = check_problem(s, mode); if (st) return st;
  odom_meas* om; float* tgt; setup_factors(s, poses_ref, &om, &tgt);
  lin_sys L; lin_alloc(&L, s->n, max_pair_distance(s), s->n_pairs);
  linearize(s, mode, om, tgt, x, &L);
  const int64_t d = L.dim;
  if (H) {
    memset(H, 0, (size_t)d * (size_t)d * sizeof(double));
    for (int64_t i = 0; i < d; ++i)
      for (int64_t j = (i - L.bw < 0 ? 0 : i - L.bw); j <= i; ++j) { H[i * d + j] = *Hat(&L, i, j); H[j * d + i] = H[i * d + j]; }
  }
  if (g) memcpy(g, L.g, (size_t)d * sizeof(double));
  if (cost) *cost = L.cost;
  lin_free(&L); free(om); free(tgt);
  return HITL_OK;
}

/* ===================================================================================== */
/* exact banded Cholesky solve (what Ceres' sparse Cholesky delivers: an exact LM step)  */
/* ===================================================================================== */
static int band_cholesky_solve(int64_t dim, int64_t bw, double* A /* lower band, overwritten */, double* b /* rhs -> x */) {
  const int64_t w = bw + 1;
#define AB(i, j) A[(i) * w + ((j) - (i) + bw)]
  for (int64_t j = 0; j < dim; ++j) {
    const int64_t k0 = (j - bw < 0) ? 0 : j - bw;
    double sum = AB(j, j);
    for (int64_t k = k0; k < j; ++k) sum -= AB(j, k) * AB(j, k);
    if (!(sum > 0.0) || !isfinite(sum)) return 1;
    const double ljj = sqrt(sum);
    AB(j, j) = ljj;
    const int64_t i1 = (j + bw > dim - 1) ? dim - 1 : j + bw;
    for (int64_t i = j + 1; i <= i1; ++i) {
      const int64_t kk0 = (i - bw < 0) ? 0 : i - bw;
      double sacc = AB(i, j);
      for (int64_t k = kk0; k < j; ++k) sacc -= AB(i, k) * AB(j, k);
      AB(i, j) = sacc / ljj;
    }
  }
  for (int64_t i = 0; i < dim; ++i) { /* L y = b */
    const int64_t k0 = (i - bw < 0) ? 0 : i - bw;
    double sacc = b[i];
    for (int64_t k = k0; k < i; ++k) sacc -= AB(i, k) * b[k];
    b[i] = sacc / AB(i, i);
  }
  for (int64_t i = dim - 1; i >= 0; --i) { /* L^T x = y */
    const int64_t k1 = (i + bw > dim - 1) ? dim - 1 : i + bw;
    double sacc = b[i];
    for (int64_t k = i + 1; k <= k1; ++k) sacc -= AB(k, i) * b[k];
    b[i] = sacc / AB(i, i);
  }
#undef AB
  return 0;
}

/* y = H x for the symmetric lower-band matrix.  Row i walks its columns in ascending order (lower band, then the
 * transposed entries below the diagonal): the same order of additions per y[i] as a serial sweep over the lower
 * triangle, so the result does not depend on the thread count. */
static void band_symv(const lin_sys* L, const double* S /* column scale or NULL */, const double* x, double* y) {
  const int64_t d = L->dim, bw = L->bw, w = bw + 1;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < d; ++i) {
    double acc = 0;
    const int64_t j0 = (i - bw < 0) ? 0 : i - bw;
    for (int64_t j = j0; j <= i; ++j) {
      double h = L->H[i * w + (j - i + bw)];
      if (S) h *= S[i] * S[j];
      acc += h * x[j];
    }
    const int64_t j1 = (i + bw > d - 1) ? d - 1 : i + bw;
    for (int64_t j = i + 1; j <= j1; ++j) {
      double h = L->H[j * w + (i - j + bw)];
      if (S) h *= S[j] * S[i];
      acc += h * x[j];
    }
    y[i] = acc;
  }
}

/* ===================================================================================== */
/* Levenberg-Marquardt with Ceres trust-region semantics                 */
/* ===================================================================================== */
int orc_solve_f64(orc_solver* s, int32_t mode, const float* poses_in, double* x_out, hitl_summary* sum) {
  int st = check_problem(s, mode); if (st) return st;
  const double t_start = now_ms();
  const hitl_options* o = &s->opt;
  const int64_t n = s->n, d = 3 * n;
  hitl_summary S; memset(&S, 0, sizeof(S));
  memset(s->phase_ms, 0, sizeof(s->phase_ms));
  odom_meas* om; float* tgt; setup_factors(s, poses_in, &om, &tgt);
  const int64_t band = (mode == HITL_MODE_ODOM_HC) ? 1 : max_pair_distance(s);
  lin_sys La, Lb; lin_alloc(&La, n, band, s->n_pairs); lin_alloc(&Lb, n, band, s->n_pairs);
  lin_sys *cur = &La, *cand = &Lb;
  double* x = (double*)malloc((size_t)d * 8); double* xc = (double*)malloc((size_t)d * 8);
  double* scale = (double*)malloc((size_t)d * 8); double* dt = (double*)malloc((size_t)d * 8);
  double* tmp = (double*)malloc((size_t)d * 8);
  double* Hs = (double*)malloc((size_t)d * (size_t)(La.bw + 1) * 8);
  for (int64_t i = 0; i < d; ++i) x[i] = (double)poses_in[i]; /* SetParams :392-402 */

  double tl = now_ms();
  linearize(s, mode, om, tgt, x, cur);
  S.linearize_ms += now_ms() - tl; S.linearize_calls++;
  S.num_residuals = cur->num_residuals; S.num_hc_residuals = cur->num_hc_residuals;
  S.num_pairs = (mode == HITL_MODE_ODOM_HC) ? 0 : s->n_pairs;
  S.num_matches = (mode == HITL_MODE_ODOM_HC) ? 0 : s->n_matches;
  S.band_poses = (int32_t)band;
  S.initial_cost = cur->cost;
  const int64_t w = cur->bw + 1, bw = cur->bw;
  for (int64_t i = 0; i < d; ++i) { /* jacobi scaling from the initial Jacobian */
    const double cn2 = cur->H[i * w + bw];
    scale[i] = (o->jacobi_scaling && i >= 3) ? 1.0 / (1.0 + sqrt(cn2)) : 1.0;
  }
  double gmax = 0; for (int64_t i = 3; i < d; ++i) if (fabs(cur->g[i]) > gmax) gmax = fabs(cur->g[i]);
  double radius = o->initial_trust_region_radius, decrease_factor = 2.0;
  int iter = 0, invalid_run = 0;
  int term = HITL_NO_CONVERGENCE; int done = 0;
  if (gmax <= o->gradient_tolerance) { term = HITL_CONVERGENCE; done = 1; }
  while (!done) {
    if (iter >= o->max_num_iterations) { term = HITL_NO_CONVERGENCE; break; }
    ++iter;
    /* LevenbergMarquardtStrategy::ComputeStep in the Jacobi-scaled space */
    double tls = now_ms();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < d; ++i) {
      const int64_t j0 = (i - bw < 0) ? 0 : i - bw;
      for (int64_t j = j0; j <= i; ++j) Hs[i * w + (j - i + bw)] = cur->H[i * w + (j - i + bw)] * scale[i] * scale[j];
      if (i >= 3) {
        double dg = Hs[i * w + bw];
        if (dg < o->min_lm_diagonal) dg = o->min_lm_diagonal;
        if (dg > o->max_lm_diagonal) dg = o->max_lm_diagonal;
        Hs[i * w + bw] += dg / radius;
      }
      dt[i] = -scale[i] * cur->g[i];
    }
    const double tch = now_ms();
    s->phase_ms[2] += tch - tls;
    int bad = band_cholesky_solve(d, bw, Hs, dt);
    s->phase_ms[3] += now_ms() - tch;
    S.linear_solver_ms += now_ms() - tls; S.linear_iterations++;
    const double tbk = now_ms();
    double mcc = 0;
    if (!bad) {
      band_symv(cur, scale, dt, tmp);
      double a = 0, b = 0;
      for (int64_t i = 0; i < d; ++i) { a += dt[i] * scale[i] * cur->g[i]; b += dt[i] * tmp[i]; if (!isfinite(dt[i])) bad = 1; }
      mcc = -a - 0.5 * b;
      if (mcc < 0.0) bad = 1;
    }
    if (bad) { /* invalid step */
      ++S.unsuccessful_steps;
      if (++invalid_run >= 5) { term = HITL_FAILURE; break; }
      radius *= 0.5;   /* LevenbergMarquardtStrategy::StepIsInvalid: halve, decrease_factor untouched */
      continue;
    }
    invalid_run = 0;
    double step2 = 0, xn2 = 0;
    for (int64_t i = 0; i < d; ++i) {
      const double dl = (i >= 3) ? scale[i] * dt[i] : 0.0;
      xc[i] = x[i] + dl;
      if (i >= 3) { step2 += dl * dl; xn2 += x[i] * x[i]; }
    }
    s->phase_ms[4] += now_ms() - tbk;
    tl = now_ms();
    linearize(s, mode, om, tgt, xc, cand);
    S.linearize_ms += now_ms() - tl; S.linearize_calls++;
    const int late = o->tolerances_after_accept != 0;   /* hitl_options: test the two step tolerances after an accepted step instead */
    const int par_tol = sqrt(step2) <= o->parameter_tolerance * (sqrt(xn2) + o->parameter_tolerance);
    if (!late && par_tol) { term = HITL_CONVERGENCE; break; }
    const double cost_change = cur->cost - cand->cost;
    const int fun_tol = fabs(cost_change) <= o->function_tolerance * cur->cost;
    if (!late && fun_tol) { term = HITL_CONVERGENCE; break; }
    const double rho = cost_change / mcc;
    if (rho > o->min_relative_decrease) {
      ++S.successful_steps;
      double* t = x; x = xc; xc = t;
      lin_sys* tl2 = cur; cur = cand; cand = tl2;
      gmax = 0; for (int64_t i = 3; i < d; ++i) if (fabs(cur->g[i]) > gmax) gmax = fabs(cur->g[i]);
      if (gmax <= o->gradient_tolerance) { term = HITL_CONVERGENCE; break; }
      if (late && (par_tol || fun_tol)) { term = HITL_CONVERGENCE; break; }
      const double q = 2.0 * rho - 1.0;
      double f = 1.0 - q * q * q; if (f < 1.0 / 3.0) f = 1.0 / 3.0;
      radius = radius / f; if (radius > o->max_trust_region_radius) radius = o->max_trust_region_radius;
      decrease_factor = 2.0;
    } else {
      ++S.unsuccessful_steps;
      radius /= decrease_factor; decrease_factor *= 2.0;
      if (radius < o->min_trust_region_radius) { term = HITL_CONVERGENCE; break; }
    }
  }
  S.termination = term; S.iterations = iter;
  S.final_cost = cur->cost; S.final_gradient_max_norm = gmax; S.final_trust_region_radius = radius;
  memcpy(x_out, x, (size_t)d * 8);
  S.total_ms = now_ms() - t_start;
  if (sum) *sum = S;
  lin_free(&La); lin_free(&Lb); free(x); free(xc); free(scale); free(dt); free(tmp); free(Hs); free(om); free(tgt);
  return (term == HITL_FAILURE) ? HITL_ERR_SOLVER_FAILURE : HITL_OK;
}

int orc_solve(orc_solver* s, int32_t mode, const float* poses_in, float* poses_out, hitl_summary* sum) {
  if (!s || !poses_in || !poses_out) return HITL_ERR_INVALID_ARGUMENT;
  double* x = (double*)malloc((size_t)s->n * 3 * 8);
  int st = orc_solve_f64(s, mode, poses_in, x, sum);
  if (st == HITL_OK || st == HITL_ERR_SOLVER_FAILURE) {
    for (int64_t i = 0; i < s->n; ++i) { /* CopyParams :380-390 */
      poses_out[3 * i] = (float)x[3 * i]; poses_out[3 * i + 1] = (float)x[3 * i + 1];
      poses_out[3 * i + 2] = (float)orc_angle_mod(x[3 * i + 2]);
    }
  }
  free(x);
  return st;
}

/* ===================================================================================== */
/* observation scans either side of the solve (fp32, like the reference)                  */
/* ===================================================================================== */
static void world_point(const float* pose, float px, float py, float* wx, float* wy) {
  /* HitLSLAM.cpp:247-252: Affine2f pose_transform = translation * rotation; pose_transform * p */
  const float c = cosf(pose[2]), s = sinf(pose[2]);
  *wx = (c * px - s * py) + pose[0];
  *wy = (s * px + c * py) + pose[1];
}

int orc_transform_scans(orc_solver* s, const float* poses, float* wx, float* wy) {
  if (!s || !s->obs_offset) return HITL_ERR_NOT_READY;
#pragma omp parallel for schedule(static, 64)
  for (int64_t i = 0; i < s->n; ++i)
    for (int64_t k = s->obs_offset[i]; k < s->obs_offset[i + 1]; ++k) world_point(poses + 3 * i, s->px[k], s->py[k], &wx[k], &wy[k]);
  return HITL_OK;
}

int orc_verify_points(orc_solver* s, const float* poses, int32_t n_points, const float* pts, float thr, int32_t* seen) {
  /* HitLSLAM.cpp:222-235: (cloud point - selected point).norm() < local_select_thresh */
  if (!s || !s->obs_offset) return HITL_ERR_NOT_READY;
  for (int q = 0; q < n_points; ++q) seen[q] = 0;
  for (int64_t i = 0; i < s->n; ++i)
    for (int64_t k = s->obs_offset[i]; k < s->obs_offset[i + 1]; ++k) {
      float wx, wy; world_point(poses + 3 * i, s->px[k], s->py[k], &wx, &wy);
      for (int q = 0; q < n_points; ++q) {
        const float dx = wx - pts[2 * q], dy = wy - pts[2 * q + 1];
        if (sqrtf(dx * dx + dy * dy) < thr) seen[q] = 1;
      }
    }
  return HITL_OK;
}

static float dist_to_line_seg(const float* p1, const float* p2, float px, float py) {
  /* EMInput::distToLineSeg, EMinput.cpp:268-279 */
  const float ex = p2[0] - p1[0], ey = p2[1] - p1[1], dx = px - p1[0], dy = py - p1[1];
  const float t = (dx * ex + dy * ey) / (ex * ex + ey * ey);
  if (t < 0.0f) return sqrtf(dx * dx + dy * dy);
  if (t > 1.0f) { const float fx = px - p2[0], fy = py - p2[1]; return sqrtf(fx * fx + fy * fy); }
  const float qx = px - (p1[0] + t * ex), qy = py - (p1[1] + t * ey);
  return sqrtf(qx * qx + qy * qy);
}

int orc_observation_sets(orc_solver* s, const float* poses, const float* seg, double thr, int32_t* counts, uint8_t* mask) {
  /* EMInput::EstablishObservationSets, EMinput.cpp:281-323 (the > 5 filter is the caller's) */
  if (!s || !s->obs_offset) return HITL_ERR_NOT_READY;
#pragma omp parallel for schedule(static, 64)
  for (int64_t i = 0; i < s->n; ++i) {
    int c0 = 0, c1 = 0;
    for (int64_t k = s->obs_offset[i]; k < s->obs_offset[i + 1]; ++k) {
      float wx, wy; world_point(poses + 3 * i, s->px[k], s->py[k], &wx, &wy);
      const int in0 = (double)dist_to_line_seg(seg, seg + 2, wx, wy) < thr;
      const int in1 = (double)dist_to_line_seg(seg + 4, seg + 6, wx, wy) < thr;
      c0 += in0; c1 += in1;
      if (mask) mask[k] = (uint8_t)(in0 | (in1 << 1));
    }
    counts[2 * i] = c0; counts[2 * i + 1] = c1;
  }
  return HITL_OK;
}

/* ===================================================================================== */
/* the correction cycle around the solve: EMInput, AppExpCorrect,       */
/* Backprop and the angle normalisation of HitLSLAM::Run.  fp32 where the reference is    */
/* fp32; libm float overloads (acosf, atan2f, sinf, cosf) as a GCC >= 6 build resolves the */
/* unqualified calls on float arguments; no fp contraction.                               */
/* ===================================================================================== */

/* Eigen::DistanceToLineSegment, shared/math/eigen_helper.h:66-81.  NOTE the reference tests
 * t > 1 against the UNIT direction, i.e. "more than 1 m along the segment", not "past p1". */
static float distance_to_line_segment(const float* p0, const float* p1, float px, float py) {
  const float dx = p1[0] - p0[0], dy = p1[1] - p0[1];
  const float nrm = sqrtf(dx * dx + dy * dy);
  const float ux = dx / nrm, uy = dy / nrm;
  const float qx = px - p0[0], qy = py - p0[1];
  const float t = qx * ux + qy * uy;
  if (t < 0.0f) return sqrtf(qx * qx + qy * qy);
  if (t > 1.0f) { const float ex = px - p1[0], ey = py - p1[1]; return sqrtf(ex * ex + ey * ey); }
  return fabsf(-uy * qx + ux * qy);
}

/* segDistResidualEM, EMinput.cpp:107-150, evaluated on duals (d[0] = d/dtheta). */
static void seg_residual_em(double theta, double px, double py, double cmx, double cmy, double len, double* r, double* dr) {
  jet th = jvar(theta, 0);
  jet ax = jcos(th), ay = jsin(th);
  jet nrm = jsqrt(jadd(jmul(ax, ax), jmul(ay, ay)));   /* alpha.normalize() */
  ax = jdiv(ax, nrm); ay = jdiv(ay, nrm);
  jet p1x = jadd(jconst(cmx), jmul(jconst(len), ax)), p1y = jadd(jconst(cmy), jmul(jconst(len), ay));
  jet p2x = jsub(jconst(cmx), jmul(jconst(len), ax)), p2y = jsub(jconst(cmy), jmul(jconst(len), ay));
  jet ex = jsub(p2x, p1x), ey = jsub(p2y, p1y);
  jet qx = jsub(jconst(px), p1x), qy = jsub(jconst(py), p1y);
  jet t = jdiv(jadd(jmul(qx, ex), jmul(qy, ey)), jadd(jmul(ex, ex), jmul(ey, ey)));
  jet res;
  if (t.v < 0.0) res = jsqrt(jadd(jmul(qx, qx), jmul(qy, qy)));
  else if (t.v > 1.0) { jet fx = jsub(jconst(px), p2x), fy = jsub(jconst(py), p2y); res = jsqrt(jadd(jmul(fx, fx), jmul(fy, fy))); }
  else {
    jet gx = jsub(jconst(px), jadd(p1x, jmul(t, ex))), gy = jsub(jconst(py), jadd(p1y, jmul(t, ey)));
    res = jsqrt(jadd(jmul(gx, gx), jmul(gy, gy)));
  }
  *r = res.v; *dr = res.d[0];
}

static void seg_fit_eval(double theta, const double* data, int64_t n, double cmx, double cmy, double len, double* cost, double* g, double* h) {
  double c = 0, gg = 0, hh = 0;
  for (int64_t i = 0; i < n; ++i) {
    double r, dr; seg_residual_em(theta, data[2 * i], data[2 * i + 1], cmx, cmy, len, &r, &dr);
    c += r * r; gg += dr * r; hh += dr * dr;
  }
  *cost = 0.5 * c; *g = gg; *h = hh;
}

/* EMInput::SegFitEM, EMinput.cpp:153-191: ceres::Solve, DENSE_QR, 25 iterations, one parameter.
 * Same trust-region semantics as orc_solve_f64 above, specialised to a 1x1 system. */
static int seg_fit_em(const double p1[2], const double p2[2], const double* data, int64_t n, float ep[4], int* iterations) {
  const double icx = (p1[0] + p2[0]) / 2.0, icy = (p1[1] + p2[1]) / 2.0;
  const double hy = sqrt((p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]));
  const double ad = fabs(p1[0] - p2[0]);
  const double theta0 = acos(ad / hy);
  double theta = theta0;
  const double len = hy / 2.0;
  double cost, g, h;
  seg_fit_eval(theta, data, n, icx, icy, len, &cost, &g, &h);
  /* A residual that is exactly 0 (a clicked point that IS a scan point and an endpoint) has the derivative 0 * inf = NaN
   * under Jet sqrt: Ceres then reports FAILURE ("Residual and Jacobian evaluation failed"), leaves the user's
   * parameter untouched (Summary::IsSolutionUsable), and SegFitEM never looks at the summary.  Same for 5 invalid
   * steps in a row.  The candidate evaluation inside an iteration is cost-only, so only its cost has to be finite. */
  int failed = !(isfinite(cost) && isfinite(g) && isfinite(h));
  const double scale = 1.0 / (1.0 + sqrt(h));
  double radius = 1e4, decrease_factor = 2.0;
  int iter = 0, invalid_run = 0;
  int done = failed || fabs(g) <= 1e-10;
  while (!done) {
    if (iter >= 25) break;
    ++iter;
    const double hs = h * scale * scale;
    double dg = hs; if (dg < 1e-6) dg = 1e-6; if (dg > 1e32) dg = 1e32;
    const double dt = -scale * g / (hs + dg / radius);
    const double mcc = -dt * scale * g - 0.5 * dt * dt * hs;
    if (!isfinite(dt) || mcc < 0.0) {
      if (++invalid_run >= 5) { failed = 1; break; }
      radius /= decrease_factor; decrease_factor *= 2.0;
      continue;
    }
    invalid_run = 0;
    const double step = scale * dt, cand = theta + step;
    double cc, cg, ch;
    seg_fit_eval(cand, data, n, icx, icy, len, &cc, &cg, &ch);
    if (fabs(step) <= 1e-8 * (fabs(theta) + 1e-8)) break;
    if (!isfinite(cc)) cc = DBL_MAX;   /* "Step failed to evaluate. Treating it as a step with infinite cost" */
    const double cost_change = cost - cc;
    if (fabs(cost_change) <= 1e-6 * cost) break;
    const double rho = cost_change / mcc;
    if (rho > 1e-3) {
      theta = cand; cost = cc; g = cg; h = ch;
      if (!(isfinite(g) && isfinite(h))) { failed = 1; break; }
      if (fabs(g) <= 1e-10) break;
      const double q = 2.0 * rho - 1.0;
      double f = 1.0 - q * q * q; if (f < 1.0 / 3.0) f = 1.0 / 3.0;
      radius = radius / f; if (radius > 1e16) radius = 1e16;
      decrease_factor = 2.0;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0;
      if (radius < 1e-32) break;
    }
  }
  if (iterations) *iterations = iter;
  if (failed) theta = theta0;
  double ax = cos(theta), ay = sin(theta);
  const double nrm = sqrt(ax * ax + ay * ay); ax /= nrm; ay /= nrm;
  ep[0] = (float)(icx + len * ax); ep[1] = (float)(icy + len * ay);
  ep[2] = (float)(icx - len * ax); ep[3] = (float)(icy - len * ay);
  return 0;
}

int orc_fit_segments(orc_solver* s, const float* poses, int32_t n_seg, float* seg, int32_t* n_inliers, int32_t* passes, int32_t* lm_iterations) {
  /* EMInput::AutomaticEndpointAdjustment, EMinput.cpp:195-250 (world points re-derived from the poses as
   * HitLSLAM::transformPointCloudsToWorldFrame made them, HitLSLAM.cpp:245-254) */
  if (!s || !s->obs_offset) return HITL_ERR_NOT_READY;
  const int64_t total = s->obs_offset[s->n];
  float* wx = (float*)malloc((size_t)total * 4 + 4); float* wy = (float*)malloc((size_t)total * 4 + 4);
  double* data = (double*)malloc((size_t)total * 16 + 16);
  orc_transform_scans(s, poses, wx, wy);
  int st = HITL_OK;
  for (int k = 0; k < n_seg && st == HITL_OK; ++k) {
    float* sp = seg + 4 * k;
    const double thresh = 0.05;
    double adj1 = 2 * thresh, adj2 = 2 * thresh;
    int pass = 0, it_total = 0; int64_t m = 0;
    while (adj1 > thresh || adj2 > thresh) {
      if (sp[0] == sp[2] && sp[1] == sp[3]) { st = HITL_ERR_INVALID_ARGUMENT; break; }
      if (++pass > 64) { st = HITL_ERR_SOLVER_FAILURE; break; }
      m = 0;
      for (int64_t q = 0; q < total; ++q)
        if ((double)distance_to_line_segment(sp, sp + 2, wx[q], wy[q]) < 0.03) { data[2 * m] = wx[q]; data[2 * m + 1] = wy[q]; ++m; }
      const double P1[2] = {sp[0], sp[1]}, P2[2] = {sp[2], sp[3]};
      float ep[4]; int it = 0;
      if (seg_fit_em(P1, P2, data, m, ep, &it)) { st = HITL_ERR_SOLVER_FAILURE; break; }
      it_total += it;
      const float a0 = sp[0] - ep[0], a1 = sp[1] - ep[1], b0 = sp[2] - ep[2], b1 = sp[3] - ep[3];
      adj1 = sqrtf(a0 * a0 + a1 * a1); adj2 = sqrtf(b0 * b0 + b1 * b1);
      memcpy(sp, ep, sizeof(ep));
    }
    if (n_inliers) n_inliers[k] = (int32_t)m;
    if (passes) passes[k] = pass;
    if (lm_iterations) lm_iterations[k] = it_total;
  }
  free(wx); free(wy); free(data);
  return st;
}

/* EMInput::OrderAndFilterUserInput, EMinput.cpp:325-455.  counts = orc_observation_sets output. */
static int32_t erase_values(int32_t* v, int32_t n, const int32_t* drop, int32_t nd) {
  for (int32_t d = 0; d < nd; ++d) { int32_t w = 0; for (int32_t i = 0; i < n; ++i) if (v[i] != drop[d]) v[w++] = v[i]; n = w; }
  return n;
}

int orc_order_user_input(int64_t n_poses, const int32_t* counts, float* seg8, int32_t* corrected, int32_t* n_corrected,
                         int32_t* anchor, int32_t* n_anchor, int32_t bounds[2]) {
  int32_t* first = (int32_t*)malloc((size_t)n_poses * 4 + 4); int32_t* second = (int32_t*)malloc((size_t)n_poses * 4 + 4);
  int32_t* over = (int32_t*)malloc((size_t)n_poses * 4 + 4);
  int32_t nf = 0, ns = 0, no = 0;
  for (int64_t i = 0; i < n_poses; ++i) { if (counts[2 * i] > 5) first[nf++] = (int32_t)i; if (counts[2 * i + 1] > 5) second[ns++] = (int32_t)i; }
  for (int32_t i = 0; i < ns; ++i) for (int32_t j = 0; j < nf; ++j) if (second[i] == first[j]) over[no++] = first[j];
  int start = 0, end = 0, st = HITL_OK;
  *n_corrected = 0; *n_anchor = 0;
  if (no == nf && no == ns) { start = end = -1; }
  else if (no == nf) ns = erase_values(second, ns, over, no);
  else if (no == ns) nf = erase_values(first, nf, over, no);
  else if (no > 0) { nf = erase_values(first, nf, over, no); ns = erase_values(second, ns, over, no); }
  if (nf == 0 || ns == 0) { start = end = -1; st = HITL_ERR_INVALID_ARGUMENT; }   /* the reference indexes [0] of an empty vector here */
  else {
    const int fmin = first[0], fmax = first[nf - 1], smin = second[0], smax = second[ns - 1];
    if (fmin > smax) { /* "user was good" */
      memcpy(corrected, first, (size_t)nf * 4); *n_corrected = nf; memcpy(anchor, second, (size_t)ns * 4); *n_anchor = ns;
      start = smax + 1; end = fmin - 1;
    } else if (fmax < smin) { /* "bad user": swap the two selections */
      float t[4]; memcpy(t, seg8, 16); memcpy(seg8, seg8 + 4, 16); memcpy(seg8 + 4, t, 16);
      memcpy(corrected, second, (size_t)ns * 4); *n_corrected = ns; memcpy(anchor, first, (size_t)nf * 4); *n_anchor = nf;
      start = fmax + 1; end = smin - 1;
    } else { start = end = -1; }
  }
  bounds[0] = start; bounds[1] = end;
  free(first); free(second); free(over);
  return st;
}

/* AppExpCorrect::Run, ApplyExplicitCorrection.cpp:150-180, 229-494. */
static void normalized2(float x, float y, float* ox, float* oy) { const float n = sqrtf(x * x + y * y); *ox = x / n; *oy = y / n; }

int orc_explicit_correction(uint32_t type, const float* sp /* 4 points */, int64_t n_poses, float* poses, int32_t n_corr,
                            const int32_t* corr, int32_t n_anchor, const int32_t* anchor, float correction[3],
                            hitl_human_constraint_t* rows) {
  if (type != 2 && type != 4 && type != 5 && type != 6) return HITL_ERR_UNSUPPORTED;  /* "not currently supported" :325,334 */
  if (n_corr <= 0) return HITL_ERR_INVALID_ARGUMENT;
  float Ax, Ay, Bx, By;
  normalized2(sp[2] - sp[0], sp[3] - sp[1], &Ax, &Ay);
  normalized2(sp[6] - sp[4], sp[7] - sp[5], &Bx, &By);
  const float cmAx = (sp[2] + sp[0]) * 0.5f, cmAy = (sp[3] + sp[1]) * 0.5f;   /* /2.0 and 0.5* are both exact */
  const float cmBx = (sp[6] + sp[4]) * 0.5f, cmBy = (sp[7] + sp[5]) * 0.5f;
  const float dot = Ax * Bx + Ay * By, cross = Ax * By - Ay * Bx;
  float theta, pivx = cmAx, pivy = cmAy, dstx, dsty;
  if (type == 2) {            /* AddLineToLineHumanConstraint :150-180 */
    theta = acosf(dot); if (cross < 0.0f) theta = -theta;
    dstx = cmBx; dsty = cmBy;
  } else if (type == 4) {     /* AddColinearHumanConstraint :229-257 */
    theta = (cross >= 0.0f) ? acosf(dot) : -acosf(dot);
    const float alpha = (cmAx - cmBx) * Bx + (cmAy - cmBy) * By;
    dstx = cmBx + alpha * Bx; dsty = cmBy + alpha * By;
  } else if (type == 5) {     /* AddPerpendicularHumanConstraint :259-293 (theta is a double there) */
    double th = (cross < 0.0f) ? -(double)acosf(dot) : (double)acosf(dot);
    if (th == M_PI / 2.0 || th == -M_PI / 2.0) th = 0.0;
    else if (th > 0.0) th = -(-th + M_PI / 2.0);
    else th = -(-th - M_PI / 2.0);
    theta = (float)th; dstx = cmAx; dsty = cmAy;
  } else {                    /* AddParallelHumanConstraint :295-316 */
    theta = (cross >= 0.0f) ? acosf(dot) : -acosf(dot);
    dstx = cmAx; dsty = cmAy;
  }
  const float c = cosf(theta), s = sinf(theta);
  /* FindContiguousGroups :358-385 with min=0, max=n-1: group 0 = the first run of consecutive corrected ids */
  uint8_t* member = (uint8_t*)calloc((size_t)n_poses, 1);
  for (int32_t j = 0; j < n_corr; ++j) { if (corr[j] < 0 || corr[j] >= n_poses) { free(member); return HITL_ERR_INVALID_ARGUMENT; } member[corr[j]] = 1; }
  int64_t g0 = 0; while (!member[g0]) ++g0;
  int64_t g1 = g0; while (g1 + 1 < n_poses && member[g1 + 1]) ++g1;
  free(member);
  /* corrections are computed from the poses BEFORE anything moves (:171-178) */
  float c0[3] = {0, 0, 0}, clast[3] = {0, 0, 0};
  for (int64_t i = g0; i <= g1; ++i) {
    const float p0x = poses[3 * i], p0y = poses[3 * i + 1];
    const float qx = p0x - pivx, qy = p0y - pivy;
    const float p1x = dstx + (c * qx - s * qy), p1y = dsty + (s * qx + c * qy);
    const float T[3] = {p1x - p0x, p1y - p0y, theta};
    if (i == g0) memcpy(c0, T, sizeof(T));
    if (i == g1) memcpy(clast, T, sizeof(T));
    poses[3 * i] += T[0]; poses[3 * i + 1] += T[1]; poses[3 * i + 2] += T[2];   /* ApplyExplicitCorrections :391-398 */
  }
  { /* tail :404-413 (poses_[last_pose] has already moved) */
    const float lx = poses[3 * g1], ly = poses[3 * g1 + 1];
    const float cl = cosf(clast[2]), sl = sinf(clast[2]);
    for (int64_t k = g1 + 1; k < n_poses; ++k) {
      poses[3 * k + 2] += clast[2];
      const float abx = poses[3 * k] - lx, aby = poses[3 * k + 1] - ly;
      const float nx = cl * abx - sl * aby, ny = sl * abx + cl * aby;
      poses[3 * k] = (lx + nx) + clast[0]; poses[3 * k + 1] = (ly + ny) + clast[1];
    }
  }
  memcpy(correction, c0, sizeof(c0));
  /* calculateConstraintTargets :447-487 */
  const float dirx = sp[6] - sp[4], diry = sp[7] - sp[5];
  const float corr_angle = atan2f(diry, dirx);
  for (int32_t a = 0; a < n_anchor; ++a) {
    const float aa = poses[3 * anchor[a] + 2];
    const float rel = (float)((double)atan2f(sinf(corr_angle - aa), cosf(corr_angle - aa)) + M_PI / 2.0);
    const float alx = poses[3 * anchor[a]], aly = poses[3 * anchor[a] + 1];
    const float px = cosf(aa), py = sinf(aa);
    for (int32_t j = 0; j < n_corr; ++j) {
      const float rx = poses[3 * corr[j]] - alx, ry = poses[3 * corr[j] + 1] - aly;
      hitl_human_constraint_t* r = rows + (int64_t)a * n_corr + j;
      r->type = type; r->anchor_pose_id = anchor[a]; r->constrained_pose_id = corr[j];
      const float da = poses[3 * corr[j] + 2] - aa;
      r->delta_angle = atan2f(sinf(da), cosf(da));
      r->delta_parallel = px * rx + py * ry;
      r->delta_perpendicular = -py * rx + px * ry;
      r->relative_penalty_dir = rel;
    }
  }
  return HITL_OK;
}

/* Backprop::Run / BackPropagateError, Backprop.cpp:98-213 (the literal O(span^2) loops).  When x64 != NULL the same
 * loops also run in fp64 from the same inputs (3 doubles per pose) -- the yardstick for the fp32 rounding noise. */
int orc_backprop(int64_t n, float* poses, float* cov9, int32_t lo, int32_t hi, const float corr[3], double* x64) {
  if (!(lo < hi)) return HITL_OK;   /* "No poses between bounds. Cannot run backprop." */
  if (lo < 0 || hi >= n) return HITL_ERR_INVALID_ARGUMENT;
  const int64_t L = hi - lo;
  const float destx = poses[3 * hi] + corr[0], desty = poses[3 * hi + 1] + corr[1];
  const float drv = 0.0001f, dtv = 0.001f;
  float* rs = (float*)malloc((size_t)n * 4); float* ts = (float*)malloc((size_t)n * 4);
  for (int64_t i = 0; i < n; ++i) { rs[i] = cov9[9 * i + 8]; ts[i] = (float)((double)(cov9[9 * i] + cov9[9 * i + 4]) / 2.0); }
  float sum_r = 0.0f, sum_t = 0.0f;
  for (int64_t i = lo; i <= hi; ++i) { sum_r += rs[i]; sum_t += ts[i]; }
  sum_r += drv; sum_t += dtv;
  float* rw = (float*)malloc((size_t)(L + 1) * 4); float* tw = (float*)malloc((size_t)(L + 1) * 4);
  for (int64_t i = lo; i <= hi; ++i) { rw[i - lo] = rs[i] / sum_r; tw[i - lo] = ts[i] / sum_t; }
  const float rot_beta = 1 / (1 + (rs[hi - 1] / drv)), trans_beta = 1 / (1 + (ts[hi - 1] / dtv));
  for (int64_t i = lo; i < hi; ++i) {   /* :156-169, including the (0,2)-twice / (1,2)-never slip */
    float* c = cov9 + 9 * i;
    c[0] *= trans_beta; c[1] *= trans_beta; c[3] *= trans_beta; c[4] *= trans_beta;
    c[2] *= rot_beta; c[2] *= rot_beta; c[6] *= rot_beta; c[7] *= rot_beta; c[8] *= rot_beta;
  }
  if (x64) for (int64_t i = 0; i < 3 * n; ++i) x64[i] = poses[i];
  const float theta = corr[2];
  for (int64_t i = lo; i < hi; ++i) {
    const float dth = rw[i - lo] * theta;
    const float c = cosf(dth), s = sinf(dth);
    const float px = poses[3 * i], py = poses[3 * i + 1];
    const float tx = px + (c * -px + -s * -py), ty = py + (s * -px + c * -py);   /* T(p) * R * T(-p) */
    poses[3 * i + 2] += dth;
    for (int64_t k = i + 1; k <= hi; ++k) {
      poses[3 * k + 2] += dth;
      const float x = poses[3 * k], y = poses[3 * k + 1];
      poses[3 * k] = (c * x + -s * y) + tx; poses[3 * k + 1] = (s * x + c * y) + ty;
    }
    if (x64) {
      const double d = (double)rs[i] / ((double)sum_r) * (double)theta, cd = cos(d), sd = sin(d);
      const double qx = x64[3 * i], qy = x64[3 * i + 1];
      x64[3 * i + 2] += d;
      for (int64_t k = i + 1; k <= hi; ++k) {
        x64[3 * k + 2] += d;
        const double x = x64[3 * k] - qx, y = x64[3 * k + 1] - qy;
        x64[3 * k] = qx + (cd * x - sd * y); x64[3 * k + 1] = qy + (sd * x + cd * y);
      }
    }
  }
  const float trx = destx - poses[3 * hi], try_ = desty - poses[3 * hi + 1];
  for (int64_t i = lo; i < hi; ++i) {
    const float dx = tw[i - lo] * trx, dy = tw[i - lo] * try_;
    for (int64_t k = i + 1; k <= hi; ++k) { poses[3 * k] += dx; poses[3 * k + 1] += dy; }
  }
  if (x64) {
    const double tx = (double)destx - x64[3 * hi], ty = (double)desty - x64[3 * hi + 1];
    for (int64_t i = lo; i < hi; ++i) {
      const double w = (double)ts[i] / (double)sum_t;
      for (int64_t k = i + 1; k <= hi; ++k) { x64[3 * k] += w * tx; x64[3 * k + 1] += w * ty; }
    }
  }
  free(rs); free(ts); free(rw); free(tw);
  return HITL_OK;
}

/* HitLSLAM.cpp:442-447: angle = atan2(sin(angle), cos(angle)) for every pose before the solve */
void orc_normalize_angles(int64_t n, float* poses) {
  for (int64_t i = 0; i < n; ++i) { const float a = poses[3 * i + 2]; poses[3 * i + 2] = atan2f(sinf(a), cosf(a)); }
}

/* ===================================================================================== */
/* pose covariances: the ceres::Covariance block of                                       */
/* episodic_non_markov_localization/vector_mapping.cpp:2772-2812 (DENSE_SVD there; for a  */
/* full-rank J the pseudo-inverse of J^T J is its inverse), then                          */
/* one_cov = transform * one_cov * transform^T with transform = diag(Rotation2Df(-angle),1) */
/* ===================================================================================== */
int orc_pose_covariances(orc_solver* s, int32_t mode, const float* poses, double* cov9) {
  int st = check_problem(s, mode); if (st) return st;
  const int64_t n = s->n, d = 3 * n;
  if (n < 1 || n > 1200) return HITL_ERR_UNSUPPORTED;   /* dense O(d^3) checker */
  double* H = (double*)malloc((size_t)d * (size_t)d * 8);
  double* x = (double*)calloc((size_t)d, 8);
  double* y = (double*)malloc((size_t)d * 3 * 8);
  for (int64_t i = 0; i < d; ++i) x[i] = (double)poses[i];
  st = orc_assemble_dense(s, mode, poses, x, H, NULL, NULL);
  if (st == HITL_OK) { /* dense Cholesky, lower triangle in place */
    for (int64_t j = 0; j < d && st == HITL_OK; ++j) {
      double v = H[j * d + j];
      for (int64_t k = 0; k < j; ++k) v -= H[j * d + k] * H[j * d + k];
      if (!(v > 0.0)) { st = HITL_ERR_SOLVER_FAILURE; break; }
      const double l = sqrt(v); H[j * d + j] = l;
      for (int64_t i = j + 1; i < d; ++i) {   /* serial on purpose: d tiny parallel regions cost more than the work */
        double t = H[i * d + j];
        for (int64_t k = 0; k < j; ++k) t -= H[i * d + k] * H[j * d + k];
        H[i * d + j] = t / l;
      }
    }
  }
  for (int64_t e = 0; e < 9; ++e) cov9[e] = 0.0;   /* pose 0 is constant: the reference loop starts at min_poses + 1 */
  for (int64_t i = 1; i < n && st == HITL_OK; ++i) {
    /* Sigma_ab = (L^-1 e_a) . (L^-1 e_b); L^-1 e_a is zero above row a */
    const int64_t r0 = 3 * i;
    for (int c = 0; c < 3; ++c) {
      double* yc = y + (size_t)c * d;
      for (int64_t r = r0; r < d; ++r) {
        double t = (r == r0 + c) ? 1.0 : 0.0;
        for (int64_t k = r0; k < r; ++k) t -= H[r * d + k] * yc[k];
        yc[r] = t / H[r * d + r];
      }
    }
    double S[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
      double t = 0; for (int64_t r = r0; r < d; ++r) t += y[(size_t)a * d + r] * y[(size_t)b * d + r];
      S[a][b] = t;
    }
    const float ang = -poses[3 * i + 2];
    const double c = (double)cosf(ang), sn = (double)sinf(ang);
    const double T[3][3] = {{c, -sn, 0}, {sn, c, 0}, {0, 0, 1}};
    double M[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[a][b] = T[a][0] * S[0][b] + T[a][1] * S[1][b] + T[a][2] * S[2][b];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov9[9 * i + 3 * a + b] = M[a][0] * T[b][0] + M[a][1] * T[b][1] + M[a][2] * T[b][2];
  }
  free(H); free(x); free(y);
  return st;
}
