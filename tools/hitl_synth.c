/*
 * hitl_synth.c -- deterministic synthetic maps for tests and bench.py.
 *
 * Neither oracle nor product: a data generator.  Output is bit-reproducible on any box
 * (own counter-based splitmix64 -> Box-Muller; no libc rand, no numpy version dependence).
 *
 * World: "floors" -- closed corridor boxes of width 2.4 m with 0.3 m alcoves every 4 m on
 * alternating sides (longitudinal observability) and offset door walls every 20 m (bounded
 * sight lines), laid out on a compact 2-D grid so that fp32 coordinates keep < 1e-4 m
 * resolution even for 200k-pose maps.  The robot drives each floor once and "takes the
 * elevator" to the next one (a long odometry edge; no scan overlap across floors).
 * Scans: exactly M rays over [-135, +135] deg, analytic ray casting against the floor's
 * axis-aligned segments, range noise N(0, range_sigma), normals = wall normals facing the
 * sensor, everything stored robot-frame fp32 SoA like JointOpt's robot_frame_*_clouds_
 * (JointOptimization.h:62-68).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/hitl_gpu.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct synth_params {
  int64_t n_poses;
  int32_t obs_per_pose;       /* M */
  int32_t poses_per_floor;    /* 200 */
  uint64_t seed;
  double step;                /* 0.3 m */
  double range_sigma;         /* 0.01 m */
  double odom_sigma_t;        /* 0.003 m per step (scan-matched front end) */
  double odom_sigma_a;        /* 0.0005 rad per step */
} synth_params;

void synth_params_default(synth_params* p) {
  p->n_poses = 1000; p->obs_per_pose = 360; p->poses_per_floor = 200; p->seed = 20260102ull;
  p->step = 0.3; p->range_sigma = 0.01; p->odom_sigma_t = 0.003; p->odom_sigma_a = 0.0005;
}

/* ---- RNG ------------------------------------------------------------------------------ */
static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static double u01(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
  uint64_t h = splitmix64(seed ^ splitmix64(a * 0x100000001B3ull + splitmix64(b * 0x1000193ull + c)));
  return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
static double gauss(uint64_t seed, uint64_t a, uint64_t b) { /* Box-Muller */
  const double u1 = u01(seed, a, b, 1), u2 = u01(seed, a, b, 2);
  return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

/* ---- floor geometry --------------------------------------------------------------------- */
#define HALF_W 1.2
#define ALC_DEPTH 0.3
#define ALC_PERIOD 4.0
#define DOOR_PERIOD 20.0
#define DOOR_HALF 0.5
#define LANE_AMP 0.7
#define FLOOR_PITCH_Y 5.0
#define FLOOR_MARGIN_X 1.0

typedef struct { double x0, y0, x1, y1; } seg; /* axis aligned */

static double floor_length(const synth_params* p) { return p->step * p->poses_per_floor; }

static int build_floor(const synth_params* p, seg** out) {
  const double L = floor_length(p);
  const double xa = -FLOOR_MARGIN_X, xb = L + FLOOR_MARGIN_X;
  const int n_alc = (int)floor((L - 2.5) / ALC_PERIOD) + 1;
  seg* s = (seg*)malloc(sizeof(seg) * (size_t)(8 * (n_alc + 4) + 16));
  int n = 0;
  for (int side = -1; side <= 1; side += 2) {
    const double yw = side * HALF_W, ya = side * (HALF_W + ALC_DEPTH);
    double cur = xa;
    for (int k = 0; k < n_alc; ++k) {
      const int on_side = ((k % 2 == 0) ? 1 : -1) == side;
      if (!on_side) continue;
      const double a0 = ALC_PERIOD * k + 1.5, a1 = ALC_PERIOD * k + 2.5;
      if (a1 > L) break;
      s[n++] = (seg){cur, yw, a0, yw};
      s[n++] = (seg){a0, yw, a0, ya};
      s[n++] = (seg){a0, ya, a1, ya};
      s[n++] = (seg){a1, ya, a1, yw};
      cur = a1;
    }
    s[n++] = (seg){cur, yw, xb, yw};
  }
  s[n++] = (seg){xa, -HALF_W, xa, HALF_W};
  s[n++] = (seg){xb, -HALF_W, xb, HALF_W};
  for (int m = 1; DOOR_PERIOD * m < L - 1.0; ++m) {
    const double xd = DOOR_PERIOD * m;
    const double yd = -LANE_AMP * cos(M_PI * xd / DOOR_PERIOD);
    s[n++] = (seg){xd, -HALF_W, xd, yd - DOOR_HALF};
    s[n++] = (seg){xd, yd + DOOR_HALF, xd, HALF_W};
  }
  *out = s;
  return n;
}

static void floor_origin(const synth_params* p, int64_t f, double* ox, double* oy) {
  const int64_t nfl = (p->n_poses + p->poses_per_floor - 1) / p->poses_per_floor;
  const double pitch_x = floor_length(p) + 2 * FLOOR_MARGIN_X + 4.0;
  const double side = sqrt((double)nfl * pitch_x * FLOOR_PITCH_Y);
  int64_t ncols = (int64_t)floor(side / pitch_x + 0.5);
  if (ncols < 1) ncols = 1;
  const int64_t nrows = (nfl + ncols - 1) / ncols;
  const int64_t col = f % ncols, row = f / ncols;
  *ox = ((double)col - 0.5 * (double)(ncols - 1)) * pitch_x - 0.5 * floor_length(p);
  *oy = ((double)row - 0.5 * (double)(nrows - 1)) * FLOOR_PITCH_Y;
}

static void local_pose(const synth_params* p, int64_t k, double* x, double* y, double* th) {
  const double xl = 0.5 * p->step + p->step * (double)k;
  *x = xl;
  *y = -LANE_AMP * cos(M_PI * xl / DOOR_PERIOD);
  const double dydx = LANE_AMP * (M_PI / DOOR_PERIOD) * sin(M_PI * xl / DOOR_PERIOD);
  *th = atan(dydx) + 0.05 * sin(0.2 * p->step * (double)k);
}

/* Ground-truth poses (fp64), initial poses (fp32, per-floor integrated odometry noise) and
 * covariances (fp32 row-major 9) for ALL poses; any output may be NULL. */
int synth_poses(const synth_params* p, double* gt_xyt, float* init_xyt, float* cov9) {
  const int64_t n = p->n_poses, ppf = p->poses_per_floor;
  const int64_t nfl = (n + ppf - 1) / ppf;
#pragma omp parallel for schedule(static)
  for (int64_t f = 0; f < nfl; ++f) {
    double ox, oy; floor_origin(p, f, &ox, &oy);
    double ix = 0, iy = 0, ith = 0, px = 0, py = 0, pth = 0;
    for (int64_t k = 0; k < ppf; ++k) {
      const int64_t i = f * ppf + k;
      if (i >= n) break;
      double x, y, th; local_pose(p, k, &x, &y, &th);
      x += ox; y += oy;
      if (gt_xyt) { gt_xyt[3 * i] = x; gt_xyt[3 * i + 1] = y; gt_xyt[3 * i + 2] = th; }
      if (k == 0) { ix = x; iy = y; ith = th; }
      else {
        /* relative GT motion in the previous frame + per-step noise, composed onto init */
        const double c = cos(pth), s = sin(pth);
        const double rx = c * (x - px) + s * (y - py) + p->odom_sigma_t * gauss(p->seed, (uint64_t)i, 101);
        const double ry = -s * (x - px) + c * (y - py) + p->odom_sigma_t * gauss(p->seed, (uint64_t)i, 102);
        const double rt = (th - pth) + p->odom_sigma_a * gauss(p->seed, (uint64_t)i, 103);
        const double ci = cos(ith), si = sin(ith);
        ix += ci * rx - si * ry; iy += si * rx + ci * ry; ith += rt;
      }
      px = x; py = y; pth = th;
      if (init_xyt) { init_xyt[3 * i] = (float)ix; init_xyt[3 * i + 1] = (float)iy; init_xyt[3 * i + 2] = (float)ith; }
      if (cov9) {
        float* cv = cov9 + 9 * i; memset(cv, 0, 9 * sizeof(float));
        cv[0] = cv[4] = (float)(p->odom_sigma_t * p->odom_sigma_t * (double)k);
        cv[8] = (float)(p->odom_sigma_a * p->odom_sigma_a * (double)k);
      }
    }
  }
  return 0;
}

/* Scans of poses [pose_begin, pose_end): (pose_end - pose_begin) * M entries per array. */
int synth_scans(const synth_params* p, int64_t pose_begin, int64_t pose_end, float* opx, float* opy,
                float* onx, float* ony) {
  const int64_t ppf = p->poses_per_floor; const int M = p->obs_per_pose;
  seg* segs; const int ns = build_floor(p, &segs);
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = pose_begin; i < pose_end; ++i) {
    double x, y, th; local_pose(p, i % ppf, &x, &y, &th);
    const int64_t base = (i - pose_begin) * M;
    for (int k = 0; k < M; ++k) {
      const double phi = (-135.0 + 270.0 * (M > 1 ? (double)k / (double)(M - 1) : 0.5)) * M_PI / 180.0;
      const double dx = cos(th + phi), dy = sin(th + phi);
      double best = 1e30, bnx = 0, bny = 0;
      for (int q = 0; q < ns; ++q) {
        const seg* g = &segs[q];
        if (g->x0 == g->x1) { /* vertical */
          if (fabs(dx) < 1e-12) continue;
          const double t = (g->x0 - x) / dx;
          if (t <= 1e-9 || t >= best) continue;
          const double yy = y + t * dy;
          const double lo = g->y0 < g->y1 ? g->y0 : g->y1, hi = g->y0 < g->y1 ? g->y1 : g->y0;
          if (yy < lo || yy > hi) continue;
          best = t; bnx = dx > 0 ? -1.0 : 1.0; bny = 0;
        } else { /* horizontal */
          if (fabs(dy) < 1e-12) continue;
          const double t = (g->y0 - y) / dy;
          if (t <= 1e-9 || t >= best) continue;
          const double xx = x + t * dx;
          const double lo = g->x0 < g->x1 ? g->x0 : g->x1, hi = g->x0 < g->x1 ? g->x1 : g->x0;
          if (xx < lo || xx > hi) continue;
          best = t; bnx = 0; bny = dy > 0 ? -1.0 : 1.0;
        }
      }
      const double r = best + p->range_sigma * gauss(p->seed, (uint64_t)i, 1000 + (uint64_t)k);
      const double c = cos(th), s = sin(th);
      opx[base + k] = (float)(r * cos(phi)); opy[base + k] = (float)(r * sin(phi));
      onx[base + k] = (float)(c * bnx + s * bny); ony[base + k] = (float)(-s * bnx + c * bny);
    }
  }
  free(segs);
  return 0;
}

/* Human-constraint rows emitted directly (bypassing the GUI stages), following
 * AppExpCorrect::calculateConstraintTargets (ApplyExplicitCorrection.cpp:447-487) but with
 * the deltas taken from GROUND-TRUTH relative poses.  A group = 5 consecutive anchors x 5
 * consecutive corrected poses = 25 rows; anchors from the first third, corrected poses from
 * the last third of [pose_begin, pose_end), groups evenly spread.
 * kind: 2,4,5,6 = all rows of that CorrectionType; 0 = mixed round-robin 2,4,5,6.
 * Returns the number of rows written (25 * n_groups); rows must hold that many. */
int64_t synth_constraints(const synth_params* p, const double* gt_xyt, int64_t pose_begin, int64_t pose_end,
                          int32_t n_groups, int32_t kind, hitl_human_constraint_t* rows) {
  (void)p;
  const int64_t span = pose_end - pose_begin, third = span / 3;
  static const uint32_t mix[4] = {2, 4, 5, 6};
  int64_t w = 0;
  for (int32_t g = 0; g < n_groups; ++g) {
    const double frac = ((double)g + 0.5) / (double)n_groups;
    int64_t a0 = pose_begin + (int64_t)floor(frac * (double)(third - 5));
    int64_t c0 = pose_begin + 2 * third + (int64_t)floor(frac * (double)(third - 5));
    if (a0 < pose_begin + 1) a0 = pose_begin + 1;      /* never anchor/constrain pose 0's slot oddly */
    if (c0 + 5 > pose_end) c0 = pose_end - 5;
    const uint32_t type = kind ? (uint32_t)kind : mix[g % 4];
    for (int ai = 0; ai < 5; ++ai) {
      const int64_t a = a0 + ai;
      const float anchor_angle = (float)gt_xyt[3 * a + 2];
      const float correction_angle = 0.0f; /* walls run along +x */
      const float rel_pen_dir = atan2f(sinf(correction_angle - anchor_angle), cosf(correction_angle - anchor_angle)) + (float)(M_PI / 2.0);
      for (int ci = 0; ci < 5; ++ci) {
        const int64_t c = c0 + ci;
        const float rx = (float)(gt_xyt[3 * c] - gt_xyt[3 * a]), ry = (float)(gt_xyt[3 * c + 1] - gt_xyt[3 * a + 1]);
        const float px = cosf(anchor_angle), py = sinf(anchor_angle);
        hitl_human_constraint_t* r = &rows[w++];
        r->type = type;
        r->anchor_pose_id = (int32_t)a;
        r->constrained_pose_id = (int32_t)c;
        r->delta_parallel = px * rx + py * ry;
        r->delta_perpendicular = -py * rx + px * ry;
        const float da = (float)(gt_xyt[3 * c + 2] - gt_xyt[3 * a + 2]);
        r->delta_angle = atan2f(sinf(da), cosf(da));
        r->relative_penalty_dir = rel_pen_dir;
      }
    }
  }
  return w;
}
