// hitl_io.cpp -- file formats either side of the re-solve.
// Host-only C++ (no HIP): usable without a GPU.
//
//  * .stfs.covars reader  == loadPoseGraph            (HitLSLAM_main.cpp:192-300)
//  * .stfs.covars writer  == SaveStfsandCovars         (vector_mapping_main.cpp:1855-1901)
//  * results writer       == saveHitLResults           (HitLSLAM_main.cpp:572-581)
//  * input-log reader     == LoadLogFile               (HitLSLAM_main.cpp:676-764)
//  * input-log writer     == LogActivity               (HitLSLAM_main.cpp:776-822)
//
// Bug-compatibility: the reference loader (1) brings NORMALS into the robot
// frame with the POINT transform R(-theta)(n - t), leaving non-unit "normals" away from the
// origin, (2) starts a new pose on ANY bit difference of consecutive (x, y, theta) rows and
// keeps the covariance of the group's first row.  Both are reproduced by default;
// fix_normals != 0 rotates normals only.
// The log reader (3) recognises a parallel correction by the number 7, while the writer (and the enum,
// human_constraints.h:15) use 6: a logged parallel correction reads its 4 points but comes back with an
// uninitialised type.  Reproduced as type UNKNOWN unless fix_parallel != 0.

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hitl_gpu.h"

struct hitl_posegraph {
  std::string map_name;
  double timestamp = 0.0;
  std::vector<float> poses;        // n*3
  std::vector<float> cov9;         // n*9 row-major
  std::vector<int64_t> obs_offset; // n+1
  std::vector<float> px, py, nx, ny;   // robot frame
};

struct hitl_input_log {
  std::vector<uint32_t> type;
  std::vector<int32_t> undone;
  std::vector<int64_t> point_off;   // entries + 1
  std::vector<float> points;        // x, y interleaved
};

namespace {

void flush_group(hitl_posegraph* g, std::vector<float>& wx, std::vector<float>& wy, std::vector<float>& wnx,
                 std::vector<float>& wny, int fix_normals) {
  const size_t n = g->poses.size() / 3;
  const float tx = g->poses[3 * (n - 1)], ty = g->poses[3 * (n - 1) + 1], th = g->poses[3 * (n - 1) + 2];
  const float c = cosf(-th), s = sinf(-th);                      // Rotation2Df(-angle)
  for (size_t i = 0; i < wx.size(); ++i) {
    const float qx = wx[i] + (-tx), qy = wy[i] + (-ty);          // point + pose_location, pose_location = -t
    g->px.push_back(c * qx - s * qy);
    g->py.push_back(s * qx + c * qy);
    float mx = wnx[i], my = wny[i];
    if (!fix_normals) { mx = mx + (-tx); my = my + (-ty); }      // HitLSLAM_main.cpp:269, :293
    g->nx.push_back(c * mx - s * my);
    g->ny.push_back(s * mx + c * my);
  }
  g->obs_offset.push_back((int64_t)g->px.size());
  wx.clear(); wy.clear(); wnx.clear(); wny.clear();
}

}  // namespace

extern "C" {

int hitl_io_load_stfs_covars(const char* path, int fix_normals, hitl_posegraph** out) {
  if (!path || !out) return HITL_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  FILE* fid = fopen(path, "r");
  if (!fid) return HITL_ERR_INVALID_ARGUMENT;
  hitl_posegraph* g = new (std::nothrow) hitl_posegraph();
  if (!g) { fclose(fid); return HITL_ERR_OUT_OF_MEMORY; }
  char map_name[64];
  if (fscanf(fid, "%63s\n", map_name) != 1 || fscanf(fid, "%lf\n", &g->timestamp) != 1) {
    fclose(fid); delete g; return HITL_ERR_INVALID_ARGUMENT;
  }
  g->map_name = map_name;
  g->obs_offset.push_back(0);
  float p[3], pt[2], nm[2], cv[9];
  std::vector<float> wx, wy, wnx, wny;
  while (fscanf(fid, "%f,%f,%f,%f,%f, %f,%f,%f, %f, %f, %f, %f, %f, %f, %f, %f\n", &p[0], &p[1], &p[2], &pt[0], &pt[1],
                &nm[0], &nm[1], &cv[0], &cv[1], &cv[2], &cv[3], &cv[4], &cv[5], &cv[6], &cv[7], &cv[8]) == 16) {
    const size_t n = g->poses.size() / 3;
    const bool new_pose = (n == 0) || p[0] != g->poses[3 * (n - 1)] || p[1] != g->poses[3 * (n - 1) + 1] ||
                          p[2] != g->poses[3 * (n - 1) + 2];
    if (new_pose) {
      if (n > 0) flush_group(g, wx, wy, wnx, wny, fix_normals);
      g->poses.insert(g->poses.end(), p, p + 3);
      g->cov9.insert(g->cov9.end(), cv, cv + 9);               // covariance of the group's first row
    }
    wx.push_back(pt[0]); wy.push_back(pt[1]); wnx.push_back(nm[0]); wny.push_back(nm[1]);
  }
  if (!wx.empty()) flush_group(g, wx, wy, wnx, wny, fix_normals);
  fclose(fid);
  *out = g;
  return HITL_OK;
}

void hitl_posegraph_free(hitl_posegraph* g) { delete g; }
int64_t hitl_posegraph_num_poses(const hitl_posegraph* g) { return g ? (int64_t)(g->poses.size() / 3) : 0; }
int64_t hitl_posegraph_num_obs(const hitl_posegraph* g) { return g ? (int64_t)g->px.size() : 0; }
double hitl_posegraph_timestamp(const hitl_posegraph* g) { return g ? g->timestamp : 0.0; }
const char* hitl_posegraph_map_name(const hitl_posegraph* g) { return g ? g->map_name.c_str() : ""; }
const float* hitl_posegraph_poses(const hitl_posegraph* g) { return g ? g->poses.data() : nullptr; }
const float* hitl_posegraph_cov9(const hitl_posegraph* g) { return g ? g->cov9.data() : nullptr; }
const int64_t* hitl_posegraph_obs_offset(const hitl_posegraph* g) { return g ? g->obs_offset.data() : nullptr; }
const float* hitl_posegraph_px(const hitl_posegraph* g) { return g ? g->px.data() : nullptr; }
const float* hitl_posegraph_py(const hitl_posegraph* g) { return g ? g->py.data() : nullptr; }
const float* hitl_posegraph_nx(const hitl_posegraph* g) { return g ? g->nx.data() : nullptr; }
const float* hitl_posegraph_ny(const hitl_posegraph* g) { return g ? g->ny.data() : nullptr; }

int hitl_io_save_stfs_covars(const char* path, const char* map_name, double timestamp, int64_t n_poses,
                             const float* poses_xyt, const int64_t* obs_offset, const float* px, const float* py,
                             const float* nx, const float* ny, const float* cov9) {
  if (!path || !map_name || n_poses < 0 || !poses_xyt || !obs_offset) return HITL_ERR_INVALID_ARGUMENT;
  FILE* fid = fopen(path, "w");
  if (!fid) return HITL_ERR_INVALID_ARGUMENT;
  fprintf(fid, "%s\n", map_name);
  fprintf(fid, "%lf\n", timestamp);
  static const float zero9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = 0; i < n_poses; ++i) {
    const float tx = poses_xyt[3 * i], ty = poses_xyt[3 * i + 1], th = poses_xyt[3 * i + 2];
    const float c = cosf(th), s = sinf(th);
    const float* cv = cov9 ? cov9 + 9 * i : zero9;
    int64_t b = obs_offset[i], e = obs_offset[i + 1];
    float fx = 0.f, fy = 0.f, fnx = 1.f, fny = 0.f;             // empty scan: one fake point at the pose
    const bool empty = (e == b);                                  // (vector_mapping_main.cpp:1881-1886)
    if (empty) { b = 0; e = 1; }
    for (int64_t k = b; k < e; ++k) {
      const float lx = empty ? fx : px[k], ly = empty ? fy : py[k];
      const float lnx = empty ? fnx : nx[k], lny = empty ? fny : ny[k];
      const float wx = empty ? tx : (c * lx - s * ly) + tx, wy = empty ? ty : (s * lx + c * ly) + ty;
      const float wnx = empty ? cosf(th) : c * lnx - s * lny, wny = empty ? sinf(th) : s * lnx + c * lny;
      fprintf(fid, "%.4f,%.4f,%.4f,%.4f,%.4f, %.4f,%.4f,%f, %f, %f, %f, %f, %f, %f, %f, %f\n", tx, ty, th, wx, wy, wnx, wny,
              cv[0], cv[1], cv[2], cv[3], cv[4], cv[5], cv[6], cv[7], cv[8]);
    }
  }
  fclose(fid);
  return HITL_OK;
}

int hitl_io_save_results(const char* path, int64_t n_poses, const float* poses_xyt) {
  if (!path || n_poses < 0 || !poses_xyt) return HITL_ERR_INVALID_ARGUMENT;
  FILE* fid = fopen(path, "w");
  if (!fid) return HITL_ERR_INVALID_ARGUMENT;
  for (int64_t i = 0; i < n_poses; ++i) fprintf(fid, "%f %f %f\n", poses_xyt[3 * i], poses_xyt[3 * i + 1], poses_xyt[3 * i + 2]);
  fclose(fid);
  return HITL_OK;
}

// ---- input log ("recording") -------------------------------------------------------------------------------
// Line 1: number of entries.  Per entry: "type, undone", then one "x, y" line per selected point: 2 points for
// type 1, 8 for type 3, 4 for types 2, 4, 5, 6, none otherwise.  Reading stops quietly at end of file.
int hitl_io_load_input_log(const char* path, int fix_parallel, hitl_input_log** out) {
  if (!path || !out) return HITL_ERR_INVALID_ARGUMENT;
  FILE* fid = fopen(path, "r");
  if (!fid) return HITL_ERR_INVALID_ARGUMENT;
  hitl_input_log* g = new (std::nothrow) hitl_input_log();
  if (!g) { fclose(fid); return HITL_ERR_OUT_OF_MEMORY; }
  g->point_off.push_back(0);
  char line[512];
  int rc = HITL_OK;
  auto field = [](const char* s, int which, char* dst, size_t cap) {   // == getline(iss, line, ',') twice
    const char* b = s;
    for (int k = 0; k < which; ++k) { const char* c = strchr(b, ','); if (!c) { b = s + strlen(s); break; } b = c + 1; }
    size_t n = 0;
    while (b[n] && b[n] != ',' && b[n] != '\n' && n + 1 < cap) { dst[n] = b[n]; ++n; }
    dst[n] = 0;
  };
  auto to_int = [](const char* s, long* v) { char* e; *v = strtol(s, &e, 10); return e != s; };          // std::stoi
  auto to_float = [](const char* s, float* v) { char* e; *v = strtof(s, &e); return e != s; };          // std::stof
  long num_entries = 0;
  if (!fgets(line, sizeof(line), fid) || !to_int(line, &num_entries) || num_entries < 0) rc = HITL_ERR_INVALID_ARGUMENT;
  for (long i = 0; rc == HITL_OK && i < num_entries; ++i) {
    if (!fgets(line, sizeof(line), fid)) break;
    char tok[128];
    long ctype = 0, undone = 0;
    field(line, 0, tok, sizeof(tok));
    if (!to_int(tok, &ctype)) { rc = HITL_ERR_INVALID_ARGUMENT; break; }
    field(line, 1, tok, sizeof(tok));
    if (!to_int(tok, &undone)) { rc = HITL_ERR_INVALID_ARGUMENT; break; }
    int npts = 0;
    uint32_t t = HITL_HC_UNKNOWN;
    if (ctype == 1) { npts = 2; t = HITL_HC_POINT; }
    else if (ctype == 3) { npts = 8; t = HITL_HC_CORNER; }
    else if (ctype == 2 || ctype == 4 || ctype == 5 || ctype == 6) {
      npts = 4;
      if (ctype == 2) t = HITL_HC_LINE_SEGMENT;
      else if (ctype == 4) t = HITL_HC_COLINEAR;
      else if (ctype == 5) t = HITL_HC_PERPENDICULAR;
      else t = fix_parallel ? HITL_HC_PARALLEL : HITL_HC_UNKNOWN;     // the reference tests "== 7" here
    }
    for (int j = 0; j < npts && rc == HITL_OK; ++j) {
      float x = 0.f, y = 0.f;
      if (!fgets(line, sizeof(line), fid)) { line[0] = 0; }
      field(line, 0, tok, sizeof(tok));
      if (!to_float(tok, &x)) { rc = HITL_ERR_INVALID_ARGUMENT; break; }
      field(line, 1, tok, sizeof(tok));
      if (!to_float(tok, &y)) { rc = HITL_ERR_INVALID_ARGUMENT; break; }
      g->points.push_back(x); g->points.push_back(y);
    }
    if (rc != HITL_OK) break;
    g->type.push_back(t);
    g->undone.push_back((int32_t)undone);
    g->point_off.push_back((int64_t)(g->points.size() / 2));
  }
  fclose(fid);
  if (rc != HITL_OK) { delete g; return rc; }
  *out = g;
  return HITL_OK;
}

void hitl_input_log_free(hitl_input_log* g) { delete g; }
int32_t hitl_input_log_num_entries(const hitl_input_log* g) { return g ? (int32_t)g->type.size() : 0; }
int hitl_input_log_entry(const hitl_input_log* g, int32_t i, uint32_t* type, int32_t* undone, int32_t* n_points, const float** points_xy) {
  if (!g || i < 0 || i >= (int32_t)g->type.size()) return HITL_ERR_INVALID_ARGUMENT;
  if (type) *type = g->type[i];
  if (undone) *undone = g->undone[i];
  if (n_points) *n_points = (int32_t)(g->point_off[i + 1] - g->point_off[i]);
  if (points_xy) *points_xy = g->points.data() + 2 * g->point_off[i];
  return HITL_OK;
}

int hitl_io_save_input_log(const char* path, int32_t n_entries, const uint32_t* types, const int32_t* undone,
                           const int32_t* n_points, const float* points_xy) {
  if (!path || n_entries < 0 || (n_entries > 0 && (!types || !undone || !n_points))) return HITL_ERR_INVALID_ARGUMENT;
  FILE* fid = fopen(path, "w");
  if (!fid) return HITL_ERR_INVALID_ARGUMENT;
  fprintf(fid, "%d \n", n_entries);
  size_t at = 0;
  for (int32_t i = 0; i < n_entries; ++i) {
    fprintf(fid, "%d, %d\n", (int)types[i], undone[i]);
    for (int32_t j = 0; j < n_points[i]; ++j, ++at) fprintf(fid, "%.4f, %.4f\n", points_xy[2 * at], points_xy[2 * at + 1]);
  }
  fclose(fid);
  return HITL_OK;
}

}  // extern "C"
