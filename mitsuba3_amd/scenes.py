"""Synthetic benchmark scenes of SURVEY.md 8(d) (no mesh assets exist in the reference tree)."""
import numpy as np

from .core import ScalarTransform4f, cornell_box


def bumpy_sphere(n_u=100, n_v=50, radius=0.08):
    """Deterministic tessellated "bumpy sphere": n_u x n_v quads = 2*n_u*n_v triangles, per-vertex
    normals, radial displacement 0.01*sin pattern (fixed formula, no RNG)."""
    u = np.arange(n_u + 1, dtype=np.float64) / n_u
    v = np.arange(n_v + 1, dtype=np.float64) / n_v
    U, Vv = np.meshgrid(u, v, indexing="xy")
    phi = 2 * np.pi * U; theta = np.pi * Vv
    r = radius * (1.0 + 0.125 * np.sin(7 * phi) * np.sin(5 * theta) ** 2)
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], -1)
    P = (r[..., None] * d).reshape(-1, 3)
    N = d.reshape(-1, 3)
    UV = np.stack([U, Vv], -1).reshape(-1, 2)
    idx = lambda i, j: j * (n_u + 1) + i
    F = []
    for j in range(n_v):
        for i in range(n_u):
            a, b, c, e = idx(i, j), idx(i + 1, j), idx(i, j + 1), idx(i + 1, j + 1)
            F.append((a, c, b)); F.append((b, c, e))
    return P.astype(np.float32), N.astype(np.float32), UV.astype(np.float32), np.asarray(F, np.uint32)


def checker_texture(tex_res=256):
    """C4's albedo bitmap: 0.5 + 0.25 * checker(8 x 8), raw RGB float32"""
    yy, xx = np.mgrid[0:tex_res, 0:tex_res]
    checker = (((xx * 8) // tex_res + (yy * 8) // tex_res) % 2).astype(np.float32)
    return np.repeat((0.5 + 0.25 * (checker - 0.5) * 2 * 0.5)[..., None], 3, -1).astype(np.float32)


def instanced_spheres_scene(width=512, height=512, spp=256, grid=10, n_u=100, n_v=50, flatten=False, max_depth=8, materials=False, textured=False, tex_res=256):
    """C3 of SURVEY.md 8(d): Cornell box + grid x grid instances of a 2*n_u*n_v-triangle bumpy
    sphere (10 x 10 x 10 000 = 1.0 M effective triangles).  flatten=True bakes every instance
    into unique triangles (true BVH-size stress).  textured=True: the `white` BSDF (floor, ceiling, back wall and -- instanced scene: all,
    flattened: every third -- spheres) takes the C4 checker bitmap as its reflectance, so that a PRB adjoint over this scene scatters its
    gradients into texels (bilinear taps, atomics under contention) instead of three constant-albedo slots."""
    T = ScalarTransform4f
    d = cornell_box()
    d['sensor']['film']['width'] = width; d['sensor']['film']['height'] = height
    d['sensor']['sampler']['sample_count'] = spp
    d['integrator']['max_depth'] = max_depth
    for k in ('small-box', 'large-box'):
        d.pop(k)
    if materials:       # every BSDF plugin of the hot path: rough plastic walls, GGX conductor / diffuse / glass spheres
        d['white'] = {'type': 'roughplastic', 'diffuse_reflectance': {'type': 'rgb', 'value': [0.885809, 0.698859, 0.666422]}, 'alpha': 0.2}
        d['green'] = {'type': 'twosided', 'm': {'type': 'roughconductor', 'distribution': 'ggx', 'alpha': 0.15, 'eta': [0.2, 0.92, 1.1], 'k': [3.9, 2.45, 2.14]}}
        d['glass'] = {'type': 'dielectric', 'int_ior': 1.5}
    if textured:
        d['white'] = {'type': 'diffuse', 'reflectance': {'type': 'bitmap', 'data': checker_texture(tex_res), 'raw': True}}
    P, N, UV, F = bumpy_sphere(n_u, n_v)
    mesh = {'type': 'mesh', 'positions': P, 'normals': N, 'texcoords': UV, 'faces': F, 'bsdf': {'type': 'ref', 'id': 'white'}}
    if not flatten:
        d['spheres'] = {'type': 'shapegroup', 'ball': mesh}
    k = 0
    for gy in range(grid):
        for gx in range(grid):
            x = -0.8 + 1.6 * gx / max(grid - 1, 1); y = -0.85 + 1.5 * gy / max(grid - 1, 1)
            z = -0.5 + 0.9 * ((gx * 7 + gy * 3) % grid) / max(grid - 1, 1)
            tf = T().translate([x, y, z]).rotate([0, 1, 0], 37.0 * k).scale(0.8 + 0.004 * k)
            if flatten:
                ids = ('white', 'green', 'red', 'glass') if materials else ('white', 'green', 'red')
                m = dict(mesh); m['to_world'] = tf; m['bsdf'] = {'type': 'ref', 'id': ids[k % len(ids)]}
                d['ball%03d' % k] = m
            else:
                d['inst%03d' % k] = {'type': 'instance', 'to_world': tf, 'group': {'type': 'ref', 'id': 'spheres'}}
            k += 1
    return d


def textured_cornell_box(res=256, tex_res=256, spp=256, max_depth=6):
    """C4 of SURVEY.md 8(d): Cornell box whose `white` reflectance is a raw bitmap
    0.5 + 0.25*checker(8x8), rendered with `prb`."""
    d = cornell_box()
    d['sensor']['film']['width'] = res; d['sensor']['film']['height'] = res
    d['sensor']['sampler']['sample_count'] = spp
    d['integrator'] = {'type': 'prb', 'max_depth': max_depth, 'rr_depth': 5}
    tex = checker_texture(tex_res)
    d['white'] = {'type': 'diffuse', 'reflectance': {'type': 'bitmap', 'data': tex, 'raw': True}}
    return d
