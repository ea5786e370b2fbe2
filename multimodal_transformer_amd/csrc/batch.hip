// Batch construction: the per-(node, timestep) Python loop of get_node_timestep_data
// (model/dataset/preprocessing.py:79-123) as one gather + standardise kernel over a packed scene table.
//
// table: the state columns of every node of a scene, node after node: row_off[n] = first row of node n, which covers scene
// timesteps first[n] .. last[n] (Node.first_timestep / last_timestep, environment/node.py:77-139).  Element (i, j) of the batch
// is node node_idx[i] at scene time ts[i]:  x[i][j] = state at ts - max_ht + j (NaN outside the node's lifetime: Node.get
// padding, node.py:101-117), y[i][j] = prediction-state columns at ts + 1 + j, first_history_index = max(max_ht - (ts -
// first), 0) (preprocessing.py:105), x_st = (x - [x_now, y_now, 0...]) / scale, y_st = (y - [x_now, y_now]) / scale with NaN
// kept (Environment.standardize hard-codes the division by 80, environment.py:49-61; preprocessing.py:107-118).
// The table is fp64 like the reference's node arrays: the subtraction and the division run in fp64 and round once to fp32,
// so the results are bit-identical to torch.tensor(np_float64, dtype=torch.float).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/mmt_amd.h"

int mmt_set_error(int code, const char *fmt, ...);   // transformer.hip

namespace {

struct BatchArgs {
  const double *table;
  const int64_t *row_off;
  const int32_t *first, *last, *node_idx, *ts;
  int B, max_ht, max_ft, ncol, npred;
  int pred_col[8];
  double scale;
  int64_t *fhi;
  float *x, *y, *x_st, *y_st;
};

__global__ __launch_bounds__(256) void mmt_batch_gather_kernel(BatchArgs a) {
  const int Tx = a.max_ht + 1, per = Tx * a.ncol + a.max_ft * a.npred;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.B * per) return;
  const int i = (int)(idx / per), r = (int)(idx - (long)i * per);
  const int n = a.node_idx[i], t = a.ts[i], f = a.first[n], l = a.last[n];
  const double *rows = a.table + a.row_off[n] * a.ncol;
  const double nan = __builtin_nan("");
  if (r == 0) a.fhi[i] = max(a.max_ht - (t - f), 0);
  const bool now_ok = t >= f && t <= l;                       // (always true for nodes picked by present_nodes)
  if (r < Tx * a.ncol) {
    const int j = r / a.ncol, c = r - j * a.ncol, tt = t - a.max_ht + j;
    const bool ok = tt >= f && tt <= l;
    const double v = ok ? rows[(long)(tt - f) * a.ncol + c] : nan;
    const double rel = c < 2 ? (now_ok ? rows[(long)(t - f) * a.ncol + c] : nan) : 0.0;
    const long o = ((long)i * Tx + j) * a.ncol + c;
    a.x[o] = (float)v;
    a.x_st[o] = ok ? (float)((v - rel) / a.scale) : (float)nan;
  } else {
    const int q = r - Tx * a.ncol, j = q / a.npred, c = q - j * a.npred, tt = t + 1 + j;
    const bool ok = tt >= f && tt <= l;
    const double v = ok ? rows[(long)(tt - f) * a.ncol + a.pred_col[c]] : nan;
    const double rel = now_ok ? rows[(long)(t - f) * a.ncol + c] : nan;      // rel_state[0:2]: the current position
    const long o = ((long)i * a.max_ft + j) * a.npred + c;
    a.y[o] = (float)v;
    a.y_st[o] = ok ? (float)((v - rel) / a.scale) : (float)nan;
  }
}

// Lane batches (preprocessing.py:170-202): per batch element the lane points of the node's current timestep / 80, the lane mask
// (first two points differ) and, for every future position, the one-hot label of the nearest lane (point-to-line distance from
// the lane's first two points; absent lanes count as 1e10; np.argmin semantics: first minimum, a NaN wins).  All in fp64 like the
// reference's numpy code, without fused multiply-adds, rounded once to fp32.
struct LaneArgs {
  const double *table, *lanes;
  const int64_t *row_off;
  const int32_t *first, *last, *node_idx, *ts;
  int B, max_ft, ncol, nl, np;
  double scale;
  float *lane_input, *lane_label;
  uint8_t *lane_mask;
};

__global__ __launch_bounds__(256) void mmt_lane_gather_kernel(LaneArgs a) {
#pragma clang fp contract(off)
  const int npts = a.nl * a.np * 2, per = npts + a.max_ft;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.B * per) return;
  const int i = (int)(idx / per), r = (int)(idx - (long)i * per);
  const int n = a.node_idx[i], t = a.ts[i], f = a.first[n], l = a.last[n];
  const long row = a.row_off[n] + (t - f);                    // history_points_at(t) = t - first_timestep (node.py:93-99)
  const double *lp = a.lanes + row * npts;
  if (r < npts) {
    a.lane_input[(long)i * npts + r] = (float)(lp[r] / 80.0);
    return;
  }
  const int j = r - npts, tt = t + 1 + j;
  const double nan = __builtin_nan("");
  const double *rows = a.table + a.row_off[n] * a.ncol;
  const bool ok = tt >= f && tt <= l;
  const double px = ok ? (rows[(long)(tt - f) * a.ncol + 0] - rows[(long)(t - f) * a.ncol + 0]) / a.scale : nan;
  const double py = ok ? (rows[(long)(tt - f) * a.ncol + 1] - rows[(long)(t - f) * a.ncol + 1]) / a.scale : nan;
  int best = 0;
  double bestv = 0.0;
  for (int k = 0; k < a.nl; ++k) {
    const double x0 = lp[(k * a.np + 0) * 2] / 80.0, y0 = lp[(k * a.np + 0) * 2 + 1] / 80.0;
    const double x1 = lp[(k * a.np + 1) * 2] / 80.0, y1 = lp[(k * a.np + 1) * 2 + 1] / 80.0;
    const double mx = x0 - x1, my = y0 - y1;
    const bool present = sqrt(mx * mx + my * my) != 0.0;
    if (j == 0) a.lane_mask[(long)i * a.nl + k] = present ? 1 : 0;
    double dist = 1e10;
    if (present) {
      const double dx = x1 - x0, dy = y1 - y0;
      const double cr = dx * (py - y0) - dy * (px - x0);
      dist = fabs(cr) / fabs(sqrt(dx * dx + dy * dy));
    }
    if (k == 0) bestv = dist;
    else if (!isnan(bestv) && (isnan(dist) || dist < bestv)) {
      best = k;
      bestv = dist;
    }
  }
  for (int k = 0; k < a.nl; ++k) a.lane_label[((long)i * a.max_ft + j) * a.nl + k] = k == best ? 1.0f : 0.0f;
}

}  // namespace

extern "C" int mmt_batch_gather(const double *table, const int64_t *row_off, const int32_t *first, const int32_t *last,
                                const int32_t *node_idx, const int32_t *ts, int B, int max_ht, int max_ft, int ncol,
                                const int32_t *pred_col, int npred, double scale, int64_t *first_history_index, float *x, float *y,
                                float *x_st, float *y_st, void *stream) {
  if (!table || !row_off || !first || !last || !node_idx || !ts || !pred_col || !first_history_index || !x || !y || !x_st || !y_st)
    return mmt_set_error(MMT_EINVAL, "mmt_batch_gather: null pointer");
  if (B <= 0 || max_ht < 0 || max_ft < 1 || ncol < 2 || npred < 1 || npred > 2 || npred > ncol || !(scale != 0.0))
    return mmt_set_error(MMT_EINVAL, "mmt_batch_gather: need B > 0, max_ft >= 1, >= 2 state columns (position first), 1..2 "
                                     "prediction columns");
  BatchArgs a;
  a.table = table, a.row_off = row_off, a.first = first, a.last = last, a.node_idx = node_idx, a.ts = ts;
  a.B = B, a.max_ht = max_ht, a.max_ft = max_ft, a.ncol = ncol, a.npred = npred, a.scale = scale;
  for (int c = 0; c < 8; ++c) a.pred_col[c] = 0;
  for (int c = 0; c < npred; ++c) {       // pred_col is a HOST array
    if (pred_col[c] < 0 || pred_col[c] >= ncol) return mmt_set_error(MMT_EINVAL, "mmt_batch_gather: prediction column out of range");
    a.pred_col[c] = pred_col[c];
  }
  a.fhi = first_history_index, a.x = x, a.y = y, a.x_st = x_st, a.y_st = y_st;
  const long total = (long)B * ((max_ht + 1) * ncol + max_ft * npred);
  hipLaunchKernelGGL(mmt_batch_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : mmt_set_error(MMT_ELAUNCH, "mmt_batch_gather: %s", hipGetErrorString(e));
}

extern "C" int mmt_lane_gather(const double *table, const double *lanes, const int64_t *row_off, const int32_t *first,
                               const int32_t *last, const int32_t *node_idx, const int32_t *ts, int B, int max_ft, int ncol,
                               int n_lanes, int n_points, double scale, float *lane_input, float *lane_label, uint8_t *lane_mask,
                               void *stream) {
  if (!table || !lanes || !row_off || !first || !last || !node_idx || !ts || !lane_input || !lane_label || !lane_mask)
    return mmt_set_error(MMT_EINVAL, "mmt_lane_gather: null pointer");
  if (B <= 0 || max_ft < 1 || ncol < 2 || n_lanes < 1 || n_lanes > 16 || n_points < 2 || !(scale != 0.0))
    return mmt_set_error(MMT_EINVAL, "mmt_lane_gather: need B > 0, max_ft >= 1, >= 2 state columns, 1..16 lanes of >= 2 points");
  LaneArgs a;
  a.table = table, a.lanes = lanes, a.row_off = row_off, a.first = first, a.last = last, a.node_idx = node_idx, a.ts = ts;
  a.B = B, a.max_ft = max_ft, a.ncol = ncol, a.nl = n_lanes, a.np = n_points, a.scale = scale;
  a.lane_input = lane_input, a.lane_label = lane_label, a.lane_mask = lane_mask;
  const long total = (long)B * (n_lanes * n_points * 2 + max_ft);
  hipLaunchKernelGGL(mmt_lane_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : mmt_set_error(MMT_ELAUNCH, "mmt_lane_gather: %s", hipGetErrorString(e));
}
