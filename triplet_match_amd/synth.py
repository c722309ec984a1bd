"""Seeded synthetic clouds for the triplet-match hot path.

The reference ships no data; every generator parameter here is ours. Surfel layout is the 48-byte
pcl::PointSurfel image used by the reference (include/common:62-70): 12 float32 per point,
[x y z 1 | nx ny nz 0 | rgba tx ty tz]; the tangent lives in the radius/confidence/curvature slots.

Model:  L-step prism (asymmetric, so its pose is unique under SE(3)), sampled area-uniformly; the
        points inside a thin band along the straight creases (~4 % of the area, never thinner than
        ~1.5 point spacings) carry the unit crease direction as tangent, all others a zero tangent.
Scene:  K planted rigid copies (independent resamplings of the same surface) + planar clutter
        (floor, two walls, with their own crease lines) + uniform outliers + Gaussian position noise.
"""
import numpy as np

EDGE_FRACTION = 0.04

# L-shaped cross-section in the x-z plane (unequal arms => no symmetry), extruded along y.
_POLY = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 0.4], [0.5, 0.4], [0.5, 0.8], [0.0, 0.8]])
_DEPTH = 0.7


def _rects_and_creases():
    """Returns (rects, creases): rect = (origin, a, b, normal); crease = (p0, p1, normal)."""
    rects, creases = [], []
    n = len(_POLY)
    for k in range(n):
        p, q = _POLY[k], _POLY[(k + 1) % n]
        o = np.array([p[0], 0.0, p[1]])
        a = np.array([q[0] - p[0], 0.0, q[1] - p[1]])
        b = np.array([0.0, _DEPTH, 0.0])
        nrm = np.cross(a, b)
        nrm /= np.linalg.norm(nrm)
        rects.append((o, a, b, nrm))
        # extruded polygon vertex -> crease along y
        creases.append((o, o + b, nrm))
        # cap boundary edges (both caps)
        for y, ny in ((0.0, -1.0), (_DEPTH, 1.0)):
            creases.append((np.array([p[0], y, p[1]]), np.array([q[0], y, q[1]]),
                            np.array([0.0, ny, 0.0])))
    # caps: the L splits into two rectangles in x-z
    for y, ny in ((0.0, -1.0), (_DEPTH, 1.0)):
        nrm = np.array([0.0, ny, 0.0])
        rects.append((np.array([0.0, y, 0.0]), np.array([1.0, 0, 0]), np.array([0, 0, 0.4]), nrm))
        rects.append((np.array([0.0, y, 0.4]), np.array([0.5, 0, 0]), np.array([0, 0, 0.4]), nrm))
    return rects, creases


def _pack(pos, nrm, tgt):
    n = pos.shape[0]
    s = np.zeros((n, 12), np.float32)
    s[:, 0:3] = pos
    s[:, 3] = 1.0
    s[:, 4:7] = nrm
    s[:, 9:12] = tgt
    return s


def _sample_rects(rng, rects, n):
    areas = np.array([np.linalg.norm(np.cross(a, b)) for (_, a, b, _) in rects])
    which = rng.choice(len(rects), size=n, p=areas / areas.sum())
    u = rng.random(n)
    v = rng.random(n)
    O = np.stack([r[0] for r in rects])[which]
    A = np.stack([r[1] for r in rects])[which]
    B = np.stack([r[2] for r in rects])[which]
    Nn = np.stack([r[3] for r in rects])[which]
    return O + u[:, None] * A + v[:, None] * B, Nn


def _crease_tangents(pos, creases, width):
    """Tangent of every point lying within `width` of a crease segment: the unit direction of the
    nearest crease (an edge detector marks a band of points, not a line); zero elsewhere."""
    n = pos.shape[0]
    best = np.full(n, np.inf)
    tgt = np.zeros((n, 3))
    for (p0, p1, _) in creases:
        d = p1 - p0
        L = np.linalg.norm(d)
        u = d / L
        t = np.clip((pos - p0) @ u, 0.0, L)
        dist = np.linalg.norm(pos - (p0 + t[:, None] * u), axis=1)
        closer = dist < best
        best = np.where(closer, dist, best)
        tgt[closer] = u
    tgt[best > width] = 0.0
    return tgt


def _band_width(rects, creases, edge_fraction, n, min_res_factor=1.5):
    """Band half-width giving ~edge_fraction of the area, but never thinner than ~1.5 x the expected
    point spacing (otherwise a band point's nearest model point is usually outside the band)."""
    area = sum(np.linalg.norm(np.cross(a, b)) for (_, a, b, _) in rects)
    length = sum(np.linalg.norm(c[1] - c[0]) for c in creases)
    w_area = edge_fraction * area / (2.0 * length)
    expected_res = 0.5 / np.sqrt(n / area)
    return max(w_area, min_res_factor * expected_res)


def _scaled(rects, creases, scale):
    sc = np.asarray(scale, dtype=np.float64)
    if np.all(sc == 1.0):
        return rects, creases
    def nrm(v):
        w = v / sc
        return w / np.linalg.norm(w)
    return ([(o * sc, a * sc, b * sc, nrm(n)) for (o, a, b, n) in rects],
            [(p0 * sc, p1 * sc, nrm(n)) for (p0, p1, n) in creases])


def synth_model(n, seed=1, edge_fraction=EDGE_FRACTION, scale=(1.0, 1.0, 1.0)):
    """n-point L-step model (optionally stretched per axis: a different model for multi-model
    searches). Returns (surfels[n,12] f32, tangent_subset u32 ascending)."""
    rng = np.random.default_rng(seed)
    rects, creases = _scaled(*_rects_and_creases(), scale)
    pos, nrm = _sample_rects(rng, rects, n)
    tgt = _crease_tangents(pos, creases, _band_width(rects, creases, edge_fraction, n))
    s = _pack(pos, nrm, tgt)
    subset = np.nonzero(np.linalg.norm(s[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
    return s, subset


def resolution(surfels):
    """Mean 1-NN distance (reference: include/impl/pointcloud.hpp:66-82), running mean in f32."""
    from scipy.spatial import cKDTree
    pts = np.ascontiguousarray(surfels[:, 0:3], dtype=np.float64)
    d, _ = cKDTree(pts).query(pts, k=2)
    return float(np.float32(d[:, 1].astype(np.float32).mean(dtype=np.float64)))


def diameter(surfels):
    p = surfels[:, 0:3]
    return float(np.linalg.norm(p.max(0) - p.min(0)))


def random_rotation(rng, upright=False):
    if upright:
        a = rng.random() * 2 * np.pi
        c, s = np.cos(a), np.sin(a)
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def synth_scene(n, model_n, seed=2, copies=4, outlier_fraction=0.05, noise_sigma_res=0.1,
                model_resolution=None, upright=False, edge_fraction=EDGE_FRACTION, shuffle=True,
                instance_points=None):
    """n-point scene containing `copies` planted instances of the model surface.

    Returns dict(surfels, tangent_mask i32, poses_model_to_scene [K,4,4] f64,
                 poses_scene_to_model [K,4,4] f64, instance_id i32 per point (-1 = clutter)).
    """
    rng = np.random.default_rng(seed)
    rects, creases = _rects_and_creases()
    diam = float(np.linalg.norm([1.0, _DEPTH, 0.8]))
    room = np.array([10.0, 10.0, 3.0]) * diam
    if instance_points is not None:
        model_n = int(instance_points)  # points sampled per planted instance
    if model_n * copies > n * 0.8:
        raise ValueError("scene too small for the requested copies")
    pos_l, nrm_l, tgt_l, inst_l = [], [], [], []
    m2s, s2m = [], []
    width = _band_width(rects, creases, edge_fraction, model_n)
    for k in range(copies):
        p, nn = _sample_rects(rng, rects, model_n)
        tt = _crease_tangents(p, creases, width)
        R = random_rotation(rng, upright)
        t = rng.random(3) * (room - diam) + 0.5 * diam * np.array([1, 1, 0.3])
        pos_l.append(p @ R.T + t)
        nrm_l.append(nn @ R.T)
        tgt_l.append(tt @ R.T)
        inst_l.append(np.full(model_n, k, np.int32))
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = t
        m2s.append(T)
        s2m.append(np.linalg.inv(T))
    used = copies * model_n
    n_out = int(round(n * outlier_fraction))
    n_clutter = n - used - n_out
    if n_clutter < 0:
        raise ValueError("scene too small for the requested copies")
    # clutter: floor + two walls, with crease lines where they meet and a few frame edges
    c_rects = [
        (np.zeros(3), np.array([room[0], 0, 0]), np.array([0, room[1], 0]), np.array([0, 0, 1.0])),
        (np.zeros(3), np.array([room[0], 0, 0]), np.array([0, 0, room[2]]), np.array([0, 1.0, 0])),
        (np.zeros(3), np.array([0, room[1], 0]), np.array([0, 0, room[2]]), np.array([1.0, 0, 0])),
    ]
    c_creases = [
        (np.zeros(3), np.array([room[0], 0, 0]), np.array([0, 0, 1.0])),
        (np.zeros(3), np.array([0, room[1], 0]), np.array([0, 0, 1.0])),
        (np.zeros(3), np.array([0, 0, room[2]]), np.array([1.0, 0, 0])),
    ]
    for _ in range(12):  # door/window-frame like segments on the walls
        a = rng.random(3) * room
        a[1 if rng.random() < 0.5 else 0] = 0.0
        d = np.zeros(3)
        d[2 if rng.random() < 0.5 else (0 if a[0] != 0.0 else 1)] = (0.5 + rng.random()) * diam
        nrm = np.array([0, 1.0, 0]) if a[1] == 0.0 else np.array([1.0, 0, 0])
        c_creases.append((a, a + d, nrm))
    pf, nf = _sample_rects(rng, c_rects, n_clutter)
    pos_l.append(pf)
    nrm_l.append(nf)
    tgt_l.append(_crease_tangents(pf, c_creases, _band_width(c_rects, c_creases, edge_fraction, max(n_clutter, 1))))
    inst_l.append(np.full(n_clutter, -1, np.int32))
    # outliers
    po = rng.random((n_out, 3)) * room
    no = rng.normal(size=(n_out, 3))
    no /= np.linalg.norm(no, axis=1, keepdims=True)
    pos_l.append(po)
    nrm_l.append(no)
    tgt_l.append(np.zeros_like(po))
    inst_l.append(np.full(n_out, -1, np.int32))
    pos = np.concatenate(pos_l)
    nrm = np.concatenate(nrm_l)
    tgt = np.concatenate(tgt_l)
    inst = np.concatenate(inst_l)
    if model_resolution is not None and noise_sigma_res > 0:
        pos = pos + rng.normal(size=pos.shape) * (noise_sigma_res * model_resolution)
    if shuffle:
        perm = rng.permutation(n)
        pos, nrm, tgt, inst = pos[perm], nrm[perm], tgt[perm], inst[perm]
    s = _pack(pos, nrm, tgt)
    tmask = (np.linalg.norm(s[:, 9:12], axis=1) > 0.7).astype(np.int32)
    return dict(surfels=s, tangent_mask=tmask, poses_model_to_scene=np.array(m2s),
                poses_scene_to_model=np.array(s2m), instance_id=inst)


def pose_colmajor(T):
    """4x4 (row-indexed numpy) -> 16 f32 column-major as the C ABI expects."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T.reshape(16))


def sample_scene_pairs(scene, model_diameter, n_first, per_first, seed=13,
                       min_diameter_factor=0.2, max_diameter_factor=1.0):
    """Seeded (i, j) scene sample pairs in RANSAC visit order: i from the tangent points, j from
    the tangent points within one model diameter of i (reference: include/impl/scene.hpp:131-144,
    273-297 — the distance window is applied here as well so that most pairs are productive)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    s = scene["surfels"]
    tidx = np.nonzero(scene["tangent_mask"])[0]
    pts = s[tidx, 0:3].astype(np.float64)
    tree = cKDTree(pts)
    firsts = rng.choice(len(tidx), size=min(n_first, len(tidx)), replace=False)
    out = []
    lo, hi = model_diameter * min_diameter_factor, model_diameter * max_diameter_factor
    for a in firsts:
        nb = np.array(tree.query_ball_point(pts[a], hi), dtype=np.int64)
        if nb.size == 0:
            continue
        d = np.linalg.norm(pts[nb] - pts[a], axis=1)
        nb = nb[(d >= lo) & (d <= hi)]
        if nb.size == 0:
            continue
        rng.shuffle(nb)
        for b in nb[:per_first]:
            out.append((tidx[a], tidx[b]))
    return np.array(out, dtype=np.uint32).reshape(-1, 2)
