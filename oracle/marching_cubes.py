"""CPU restatement of saf_marching_cubes_* (TEST INFRASTRUCTURE; numpy, table-driven).

What it restates -- the mesh half of the reference's extract_mesh (clipfusion.py:723-739, clip_seem_fusion.py:824-842):
un-fused voxels (weight == 0) are NaN, marching cubes at level 0, faces touching a NaN vertex dropped, unused vertices
dropped and faces re-indexed.  The reference delegates the cube triangulation to scikit-image 0.22.0
(`skimage.measure.marching_cubes`, Lewiner's variant), which is absent from the build image, so PARITY WITH IT IS
UNPINNED: this file pins the HIP kernel, and is itself checked on analytic TSDFs (tests/test_marching_cubes.py: vertices
on a sphere within the linear-interpolation error, watertight and consistently oriented surface, Euler characteristic 2).
What is invariant under the choice of triangulation -- one vertex per grid edge whose fused end points lie on different
sides of the level, linearly interpolated -- is the vertex SET, which is what the reference samples features at.

The triangle table is the one the kernel uses (tools/gen_mc_table.py builds it from first principles; conventions there).
"""
from __future__ import annotations

import os
import sys

import numpy as np

_TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
if _TOOLS not in sys.path:
    sys.path.insert(0, _TOOLS)
import gen_mc_table  # noqa: E402

_TABLE = None


def table():
    global _TABLE
    if _TABLE is None:
        t = gen_mc_table.build()
        cnt = np.array([len(x) for x in t], dtype=np.int64)
        tris = np.full((256, 5, 3), 0, dtype=np.int64)
        for c, x in enumerate(t):
            for k, tri in enumerate(x):
                tris[c, k] = tri
        _TABLE = (cnt, tris)
    return _TABLE


def _edge_tables():
    ca = np.zeros(12, dtype=np.int64)
    cb = np.zeros(12, dtype=np.int64)
    owner = np.zeros((12, 3), dtype=np.int64)
    axis = np.zeros(12, dtype=np.int64)
    cid = lambda c: c[0] + 2 * c[1] + 4 * c[2]
    for e in range(12):
        a, b = gen_mc_table.EDGES[e]
        ca[e], cb[e] = cid(a), cid(b)
        owner[e] = a
        axis[e] = 0 if e < 4 else (1 if e < 8 else 2)
    return ca, cb, owner, axis


def marching_cubes(tsdf, weight, level=0.0):
    """tsdf [nx,ny,nz] f32, weight [nx,ny,nz] int -> (verts [V,3] f32 in voxel-index coordinates, faces [F,3] int64),
    vertices ordered by (owner voxel in raster order, axis), faces by cube in raster order then table order."""
    tsdf = np.asarray(tsdf, dtype=np.float32)
    fused = np.asarray(weight) > 0
    nx, ny, nz = tsdf.shape
    pos = fused & (tsdf > np.float32(level))
    cnt_t, tris_t = table()
    ca, cb, owner, axis = _edge_tables()
    sl = lambda d, n: slice(d, n - 1 + d)
    corner = lambda arr, c: arr[sl(c & 1, nx), sl((c >> 1) & 1, ny), sl(c >> 2, nz)]
    case = np.zeros((nx - 1, ny - 1, nz - 1), dtype=np.int64)
    fmask = np.zeros_like(case)
    for c in range(8):
        case |= corner(pos, c).astype(np.int64) << c
        fmask |= corner(fused, c).astype(np.int64) << c
    flat = lambda x, y, z: (x * ny + y) * nz + z
    cx, cy, cz = np.nonzero((case != 0) & (case != 255))
    cs = case[cx, cy, cz]
    fm = fmask[cx, cy, cz]
    face_edges = []  # global edge ids (owner voxel * 3 + axis), per kept triangle, in cube raster order
    order_key = []
    for k in range(5):
        has = cnt_t[cs] > k
        e = tris_t[cs, k]  # [n, 3]
        ok = has.copy()
        for j in range(3):
            ok &= ((fm >> ca[e[:, j]]) & 1).astype(bool) & ((fm >> cb[e[:, j]]) & 1).astype(bool)
        idx = np.nonzero(ok)[0]
        ge = np.zeros((len(idx), 3), dtype=np.int64)
        for j in range(3):
            ej = e[idx, j]
            ge[:, j] = flat(cx[idx] + owner[ej, 0], cy[idx] + owner[ej, 1], cz[idx] + owner[ej, 2]) * 3 + axis[ej]
        face_edges.append(ge)
        order_key.append(flat(cx[idx], cy[idx], cz[idx]) * 8 + k)
    ge = np.concatenate(face_edges) if face_edges else np.zeros((0, 3), dtype=np.int64)
    key = np.concatenate(order_key) if order_key else np.zeros(0, dtype=np.int64)
    ge = ge[np.argsort(key, kind="stable")]
    used = np.unique(ge)
    faces = np.searchsorted(used, ge).astype(np.int64)
    ov, ax = used // 3, used % 3
    x, y, z = ov // (ny * nz), (ov // nz) % ny, ov % nz
    step = np.array([ny * nz, nz, 1], dtype=np.int64)[ax]
    ft = tsdf.reshape(-1)
    va, vb = ft[ov], ft[ov + step]
    t = (np.float32(level) - va) / (vb - va)
    verts = np.stack([x, y, z], axis=1).astype(np.float32)
    verts[np.arange(len(used)), ax] += t.astype(np.float32)
    return verts, faces


def mesh_checks(verts, faces):
    """Topology report of a triangle mesh: directed-edge bookkeeping.  Returns a dict with the number of boundary
    edges (used by one face), non-manifold edges (more than two), orientation conflicts (an edge traversed twice in
    the same direction) and the Euler characteristic V - E + F."""
    f = np.asarray(faces, dtype=np.int64)
    de = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    und = np.sort(de, axis=1)
    keys, counts = np.unique(und[:, 0] * (len(verts) + 1) + und[:, 1], return_counts=True)
    dk, dcounts = np.unique(de[:, 0] * (len(verts) + 1) + de[:, 1], return_counts=True)
    return {"boundary_edges": int((counts == 1).sum()), "nonmanifold_edges": int((counts > 2).sum()),
            "orientation_conflicts": int((dcounts > 1).sum()),
            "euler": int(len(np.unique(f)) - len(keys) + len(f))}
