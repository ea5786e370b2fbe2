"""``GeometricMap`` with the reference's interface (``Transformer/environment/map.py:7-230``) whose patch extraction runs
as ONE fused gather-rotate HIP kernel (``mmt_map_crop``) instead of pad -> crop -> normalise -> grid_sample on the CPU
inside ``collate`` (``model/dataset/preprocessing.py:32-40``), the step that feeds the map encoders.

The uint8 raster of a map is uploaded once and cached per device; a batch that mixes agents of several scenes passes
one raster pointer per distinct map."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from .._lib import MmtError


class Map(object):
    def __init__(self, data, homography, description=None):
        self.data = data
        self.homography = homography
        self.description = description

    def as_image(self):
        raise NotImplementedError

    def get_cropped_maps(self, world_pts, patch_size, rotation=None, device="cpu"):
        raise NotImplementedError

    def to_map_points(self, scene_pts):
        raise NotImplementedError


class GeometricMap(Map):
    """A Geometric Map is a uint8 array of shape [layers, x, y]; the homography maps scene to map coordinates."""

    def __init__(self, data, homography, description=None):
        super().__init__(data, homography, description=description)
        self._torch_map = {}

    def torch_map(self, device):
        key = str(device)
        if key not in self._torch_map:
            t = torch.as_tensor(np.ascontiguousarray(self.data)).to(torch.uint8)
            self._torch_map[key] = t.to(device).contiguous()
        return self._torch_map[key]

    def as_image(self):
        return (np.transpose(self.data, (2, 1, 0))).astype(np.uint)

    def to_map_points(self, scene_pts):
        """map.py:208-219"""
        org_shape = None
        if len(scene_pts.shape) > 2:
            org_shape = scene_pts.shape
            scene_pts = scene_pts.reshape((-1, 2))
        N, dims = scene_pts.shape
        points_with_one = np.ones((dims + 1, N))
        points_with_one[:dims] = scene_pts.T
        map_points = (self.homography @ points_with_one).T[..., :dims]
        if org_shape is not None:
            map_points = map_points.reshape(org_shape)
        return map_points

    @classmethod
    def get_cropped_maps_from_scene_map_batch(cls, maps, scene_pts, patch_size, rotation=None, device="cuda"):
        """map.py:93-173 -> (bs, layers, patch_size[1]+patch_size[3], patch_size[0]+patch_size[2]) fp32 in [0,1], on ``device``.

        :param maps: list of GeometricMap [bs]   :param scene_pts: [bs, 2] scene points (numpy or tensor)
        :param patch_size: [-x, -y, +x, +y]       :param rotation: degrees [bs] or None"""
        dev = torch.device(device)
        if dev.type != "cuda":
            raise MmtError("GeometricMap: patch extraction runs on the GPU only (no CPU fallback); pass device='cuda'")
        pts = scene_pts.detach().cpu().numpy() if isinstance(scene_pts, torch.Tensor) else np.asarray(scene_pts)
        bs = pts.shape[0]
        if bs == 0 or len(maps) != bs:
            raise MmtError(f"GeometricMap: {len(maps)} maps for {bs} scene points")
        lat_size = 2 * max(patch_size[0], patch_size[2])
        long_size = 2 * max(patch_size[1], patch_size[3])
        assert lat_size % 2 == 0, "Patch width must be divisible by 2"
        assert long_size % 2 == 0, "Patch length must be divisible by 2"
        # torch.tensor(np.array([to_map_points...]), dtype=torch.long): truncation towards zero (map.py:131-135)
        centers = np.trunc(np.array([m.to_map_points(pts[np.newaxis, i]) for i, m in enumerate(maps)])).astype(np.int64)
        centers = centers.reshape(bs, 2)
        distinct, index = [], np.empty(bs, dtype=np.int32)
        for i, m in enumerate(maps):
            for k, d in enumerate(distinct):
                if d is m:
                    index[i] = k
                    break
            else:
                distinct.append(m)
                index[i] = len(distinct) - 1
        layers = distinct[0].data.shape[0]
        rasters = [m.torch_map(dev) for m in distinct]
        for r in rasters:
            if r.dim() != 3 or r.shape[0] != layers:
                raise MmtError("GeometricMap: all maps of a batch must have the same number of layers")
        ptrs = torch.tensor([r.data_ptr() for r in rasters], dtype=torch.int64).to(dev)
        dims = torch.tensor([[r.shape[1], r.shape[2]] for r in rasters], dtype=torch.int32).to(dev)
        ctr = torch.from_numpy(centers.astype(np.int32)).to(dev)
        amap = torch.from_numpy(index).to(dev)
        ang = None
        if rotation is not None:
            ang = torch.as_tensor(rotation).detach().to(torch.float32).to(dev).contiguous()
            if ang.numel() != bs:
                raise MmtError("GeometricMap: rotation must have one angle per scene point")
        out = torch.empty(bs, layers, patch_size[1] + patch_size[3], patch_size[0] + patch_size[2], dtype=torch.float32,
                          device=dev)
        ps = (C.c_int32 * 4)(*[int(v) for v in patch_size])
        lib = _lib.load()
        _lib.check(lib.mmt_map_crop(C.c_void_p(ptrs.data_ptr()), C.c_void_p(dims.data_ptr()), layers,
                                    C.c_void_p(amap.data_ptr()), C.c_void_p(ctr.data_ptr()),
                                    C.c_void_p(ang.data_ptr()) if ang is not None else None, ps, bs,
                                    C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "mmt_map_crop")
        return out

    def get_cropped_maps(self, scene_pts, patch_size, rotation=None, device="cuda"):
        """map.py:175-206"""
        return self.get_cropped_maps_from_scene_map_batch([self] * scene_pts.shape[0], scene_pts, patch_size,
                                                          rotation=rotation, device=device)
