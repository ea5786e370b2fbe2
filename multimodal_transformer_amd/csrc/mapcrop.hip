// Map patch extraction for gfx950: GeometricMap.get_cropped_maps_from_scene_map_batch (environment/map.py:93-173) with
// warp_affine_crop / grid_sample(bilinear, zeros, align_corners=True) (model/dataset/homography_warper.py:255-497) fused
// into one gather-rotate kernel that reads the uint8 scene raster and writes the (B, layers, rows, cols) fp32 patches
// the map encoders consume.  No padded copy of the map, no intermediate 2*pad crop, no
// 180x100 rotated image: an output pixel maps straight to its four source texels.
//
// The reference's chain, restated per output pixel (R, Cc) of the rotated dsize = (long, lat) image:
//   normalised destination grid   xn = xmin + Cc (xmax - xmin) / (lat - 1),  xmin/xmax = 2 (cx -/+ lat/2) / Wimg - 1
//                                 yn likewise with long / Himg                 (homography_warper.py:280-283, 204-216)
//   to pixels (align_corners)     u = (xn + 1)(Wimg - 1) / 2,  v = (yn + 1)(Himg - 1) / 2          (:293-304)
//   inverse rotation about the crop centre (cx, cy) = (pad_y, pad_x) by the heading angle          (:45-121, 332-353)
//   bilinear sample of crop/255 with zeros outside the crop [0, Wimg) x [0, Himg); the crop itself is the zero-padded
//   scene map window [px - pad_x, px + pad_x) x [py - pad_y, py + pad_y) around the agent's map point   (map.py:139-153)
// then the slice [long/2 - ps[1] : long/2 + ps[3], lat/2 - ps[0] : lat/2 + ps[2]]                     (map.py:169-173)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmt_amd.h"

int mmt_set_error(int code, const char *fmt, ...);   // transformer.hip

namespace {

struct CropGeom {
  int layers, lat, lng, pad_x, pad_y, row0, col0, rows, cols;   // row0/col0: slice origin inside the (lng x lat) image
};

__global__ __launch_bounds__(256) void mmt_map_crop_kernel(const uint8_t *const *__restrict__ maps, const int32_t *__restrict__ map_dims,
                                                           const int32_t *__restrict__ agent_map, const int32_t *__restrict__ centers,
                                                           const float *__restrict__ angles, CropGeom G, int B, float *__restrict__ out) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= G.rows * G.cols) return;
  const int r = pix / G.cols, c = pix - r * G.cols;
  const int mi = agent_map[b];
  const uint8_t *map = maps[mi];
  const int MX = map_dims[2 * mi], MY = map_dims[2 * mi + 1];     // data is [layers][x][y]
  const int px = centers[2 * b], py = centers[2 * b + 1];         // map point (long, truncated) before padding
  const float Himg = 2.0f * G.pad_x, Wimg = 2.0f * G.pad_y;       // crop: rows = x extent, cols = y extent
  const float cx = (float)G.pad_y, cy = (float)G.pad_x;           // centre (col, row)
  // destination grid, normalised, then back to pixels
  const float xmin = 2.0f * (cx - G.lat / 2.0f) / Wimg - 1.0f, xmax = 2.0f * (cx + G.lat / 2.0f) / Wimg - 1.0f;
  const float ymin = 2.0f * (cy - G.lng / 2.0f) / Himg - 1.0f, ymax = 2.0f * (cy + G.lng / 2.0f) / Himg - 1.0f;
  const float xn = xmin + (xmax - xmin) / (float)(G.lat - 1) * (float)(G.col0 + c);
  const float yn = ymin + (ymax - ymin) / (float)(G.lng - 1) * (float)(G.row0 + r);
  const float u = (xn + 1.0f) * (Wimg - 1.0f) * 0.5f, v = (yn + 1.0f) * (Himg - 1.0f) * 0.5f;
  // source = M^-1 (u, v): M rotates by +angle about (cx, cy) with [[a, b], [-b, a]]
  float a = 1.0f, bt = 0.0f;
  if (angles) {
    const float rad = angles[b] * 3.14159265358979323846f / 180.0f;
    a = cosf(rad);
    bt = sinf(rad);
  }
  const float du = u - cx, dv = v - cy;
  const float sx = a * du - bt * dv + cx, sy = bt * du + a * dv + cy;
  // bilinear, zeros padding, on the crop; crop texel (row, col) = map (px - pad_x + row, py - pad_y + col)
  const float fx = floorf(sx), fy = floorf(sy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
  const int Hi = 2 * G.pad_x, Wi = 2 * G.pad_y;
  float w[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
  int off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = y0 + (t >> 1), col = x0 + (t & 1);
    const int mx = px - G.pad_x + row, my = py - G.pad_y + col;
    const bool ok = row >= 0 && row < Hi && col >= 0 && col < Wi && mx >= 0 && mx < MX && my >= 0 && my < MY;
    off[t] = ok ? mx * MY + my : -1;
  }
  const size_t plane = (size_t)MX * MY;
  for (int l = 0; l < G.layers; ++l) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (off[t] >= 0) acc += (float)map[l * plane + off[t]] / 255.0f * w[t];
    out[(((size_t)b * G.layers + l) * G.rows + r) * G.cols + c] = acc;
  }
}

}  // namespace

extern "C" int mmt_map_crop(const uint8_t *const *maps, const int32_t *map_dims, int layers, const int32_t *agent_map,
                            const int32_t *centers, const float *angles_deg, const int32_t *patch_size, int B, float *out,
                            void *stream) {
  if (!maps || !map_dims || !agent_map || !centers || !patch_size || !out) return mmt_set_error(MMT_EINVAL, "mmt_map_crop: null pointer");
  if (B <= 0 || layers <= 0) return mmt_set_error(MMT_EINVAL, "mmt_map_crop: empty batch or no layers");
  const int p0 = patch_size[0], p1 = patch_size[1], p2 = patch_size[2], p3 = patch_size[3];
  if (p0 < 0 || p1 < 0 || p2 < 0 || p3 < 0 || p0 + p2 <= 0 || p1 + p3 <= 0)
    return mmt_set_error(MMT_EINVAL, "mmt_map_crop: bad patch_size [%d,%d,%d,%d]", p0, p1, p2, p3);
  CropGeom G;
  G.layers = layers;
  G.lat = 2 * (p0 > p2 ? p0 : p2);
  G.lng = 2 * (p1 > p3 ? p1 : p3);
  if (G.lat < 2 || G.lng < 2) return mmt_set_error(MMT_EINVAL, "mmt_map_crop: patch too small");
  G.pad_x = (int)ceil(sqrt(2.0) * G.lat);
  G.pad_y = (int)ceil(sqrt(2.0) * G.lng);
  G.row0 = G.lng / 2 - p1;
  G.col0 = G.lat / 2 - p0;
  G.rows = p1 + p3;
  G.cols = p0 + p2;
  hipLaunchKernelGGL(mmt_map_crop_kernel, dim3((G.rows * G.cols + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, maps, map_dims,
                     agent_map, centers, angles_deg, G, B, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt_set_error(MMT_ELAUNCH, "mmt_map_crop: %s", hipGetErrorString(e));
  return MMT_OK;
}
