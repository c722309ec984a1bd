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


# ---------------------------------------------------------------------------------------------
# configs 3 / 4 of BASELINE.json: scenes for the projector traits (cylinder, plane2)
# ---------------------------------------------------------------------------------------------
def _frame(axis):
    """right-handed orthonormal frame (x, y, axis) for a unit axis"""
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    h = np.array([1.0, 0, 0]) if abs(axis[0]) < 0.9 else np.array([0, 1.0, 0])
    x = np.cross(h, axis)
    x /= np.linalg.norm(x)
    return x, np.cross(axis, x), axis


def _pipe_pattern(theta, h, radius, spacing):
    """Tangent kind per point of a pipe section of height 2 (theta in [0, 2pi), h in [0, 2]):
    0 = none, 1 = axial seam, 2 = flange ring. Asymmetric, so the in-surface pose is unique."""
    w_arc = max(0.04, 1.5 * spacing)  # half width of a band, arc length / height units
    d1 = np.abs(np.angle(np.exp(1j * (theta - 1.0)))) * radius
    d2 = np.abs(np.angle(np.exp(1j * (theta - 4.0)))) * radius
    seam = (d1 < w_arc) | ((d2 < w_arc) & (h < 1.2))
    ring = (np.abs(h - 0.6) < w_arc) | (np.abs(h - 1.5) < 0.6 * w_arc)
    kind = np.zeros(theta.shape, np.int32)
    kind[ring] = 2
    kind[seam] = 1
    return kind


def _pipe_points(rng, n, radius, length, origin, axis, spacing, instances=(), decoy_seam=None):
    """n surfels on a pipe; `instances` = [(h0, theta0)] planted model sections of height 2."""
    x, y, z = _frame(axis)
    theta = rng.random(n) * 2 * np.pi
    h = rng.random(n) * length
    radial = np.cos(theta)[:, None] * x + np.sin(theta)[:, None] * y
    pos = np.asarray(origin) + radius * radial + h[:, None] * z
    nrm = -radial  # inward, as the cylinder projector's init_from_samples expects
    circ = -np.sin(theta)[:, None] * x + np.cos(theta)[:, None] * y
    tgt = np.zeros((n, 3))
    inst = np.full(n, -1, np.int32)
    for k, (h0, t0) in enumerate(instances):
        sel = (h >= h0) & (h <= h0 + 2.0)
        kind = _pipe_pattern(np.mod(theta[sel] - t0, 2 * np.pi), h[sel] - h0, radius, spacing)
        idx = np.nonzero(sel)[0]
        tgt[idx[kind == 1]] = z
        tgt[idx[kind == 2]] = circ[idx[kind == 2]]
        inst[idx] = k
    if decoy_seam is not None:
        d = np.abs(np.angle(np.exp(1j * (theta - decoy_seam)))) * radius
        sel = (d < max(0.04, 1.5 * spacing)) & (inst < 0)
        tgt[sel] = z
    return pos, nrm, tgt, inst, theta, h


def synth_pipe_section(n, seed=3, radius=0.5):
    """Model of config 3: a pipe section (axis +z through the origin, height 2) with two axial seams and
    two flange rings carrying tangents. Returns (surfels, tangent subset)."""
    rng = np.random.default_rng(seed)
    spacing = np.sqrt(2 * np.pi * radius * 2.0 / n)
    pos, nrm, tgt, _, _, _ = _pipe_points(rng, n, radius, 2.0, np.zeros(3), [0, 0, 1.0], spacing, instances=[(0.0, 0.0)])
    s = _pack(pos, nrm, tgt)
    return s, np.nonzero(np.linalg.norm(s[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)


def synth_pipe_plant(n, model_n, seed=4, radius=0.5, noise_sigma=0.0):
    """Scene of config 3: axis-aligned pipe runs of three radii; each run of the model's radius carries one
    planted section (known offset along and around the pipe); floor clutter + uniform outliers.

    Returns dict(surfels, tangent_mask, instance_id (planted section or -1), pipe_id (-1 = clutter),
    pipes = [(origin, axis, radius, length)], instances = [(pipe, h0, theta0)])."""
    rng = np.random.default_rng(seed)
    dens = model_n / (2 * np.pi * radius * 2.0)
    spacing = 1.0 / np.sqrt(dens)
    pipes = [  # origin, axis, radius, length, planted (h0, theta0) or None, decoy seam angle or None
        (np.array([0.0, 0.0, 1.0]), [1.0, 0, 0], radius, 4.0, (1.1, 0.7), 2.6),
        (np.array([5.0, 1.0, 1.5]), [0, 1.0, 0], radius, 4.0, (0.4, 3.9), None),
        (np.array([2.0, 6.0, 0.0]), [0, 0, 1.0], radius, 4.0, (1.7, 5.5), 0.3),
        (np.array([0.0, 3.0, 2.0]), [1.0, 0, 0], 0.6 * radius, 4.0, None, 1.0),
        (np.array([7.0, 0.0, 0.5]), [0, 1.0, 0], 1.6 * radius, 3.0, None, 4.4),
    ]
    areas = np.array([2 * np.pi * p[2] * p[3] for p in pipes])
    scale = min(1.0, 0.9 * n / (dens * areas.sum()))
    pos_l, nrm_l, tgt_l, inst_l, pipe_l, instances = [], [], [], [], [], []
    for k, (o, ax, r, L, planted, decoy) in enumerate(pipes):
        cnt = int(dens * scale * 2 * np.pi * r * L)
        inst_list = []
        if planted is not None:
            inst_list = [planted]
        p, nn, tt, ii, _, _ = _pipe_points(rng, cnt, r, L, o, ax, spacing / np.sqrt(scale), inst_list, decoy)
        if planted is not None:
            ii[ii == 0] = len(instances)
            instances.append((k, planted[0], planted[1]))
        pos_l.append(p)
        nrm_l.append(nn)
        tgt_l.append(tt)
        inst_l.append(ii)
        pipe_l.append(np.full(cnt, k, np.int32))
    used = sum(len(p) for p in pos_l)
    n_out = int(0.02 * n)
    n_floor = n - used - n_out
    pf = np.stack([rng.random(n_floor) * 9 - 1, rng.random(n_floor) * 8 - 1, np.full(n_floor, -0.8)], 1)
    pos_l.append(pf)
    nrm_l.append(np.tile([0, 0, 1.0], (n_floor, 1)))
    tgt_l.append(np.zeros((n_floor, 3)))
    po = rng.random((n_out, 3)) * [9, 8, 5] - [1, 1, 0.8]
    no = rng.normal(size=(n_out, 3))
    no /= np.linalg.norm(no, axis=1, keepdims=True)
    pos_l.append(po)
    nrm_l.append(no)
    tgt_l.append(np.zeros((n_out, 3)))
    for m in (n_floor, n_out):
        inst_l.append(np.full(m, -1, np.int32))
        pipe_l.append(np.full(m, -1, np.int32))
    pos, nrm, tgt = np.concatenate(pos_l), np.concatenate(nrm_l), np.concatenate(tgt_l)
    inst, pipe = np.concatenate(inst_l), np.concatenate(pipe_l)
    if noise_sigma > 0:
        pos = pos + rng.normal(size=pos.shape) * noise_sigma
    perm = rng.permutation(n)
    s = _pack(pos[perm], nrm[perm], tgt[perm])
    tmask = (np.linalg.norm(s[:, 9:12], axis=1) > 0.7).astype(np.int32)
    return dict(surfels=s, tangent_mask=tmask, instance_id=inst[perm], pipe_id=pipe[perm],
                pipes=[(p[0], np.asarray(p[1], np.float64), p[2], p[3]) for p in pipes], instances=instances)


def _panel_pattern(u, v, spacing):
    """Tangent kind of a 2.0 x 1.0 wall panel (window frame): 0 none, 1 edge along u, 2 edge along v."""
    w = max(0.02, 1.5 * spacing)
    along_u = (v < w) | (v > 1.0 - w) | (np.abs(v - 0.3) < 0.6 * w)
    along_v = (u < w) | (u > 2.0 - w) | (np.abs(u - 1.2) < w)
    kind = np.zeros(u.shape, np.int32)
    kind[along_u] = 1
    kind[along_v & ~along_u] = 2
    return kind


def synth_wall_panel(n, seed=5, z_sigma=0.002):
    """Model of config 4: a 2 x 1 planar panel in the xy-plane (normal +z) with frame, mullion and sill edges.
    A little depth noise keeps the voxel grid's w axis non-degenerate (a zero range would make every w
    land in the grid, i.e. the whole room would count under the 2-D metric)."""
    rng = np.random.default_rng(seed)
    spacing = np.sqrt(2.0 / n)
    u, v = rng.random(n) * 2.0, rng.random(n)
    kind = _panel_pattern(u, v, spacing)
    pos = np.stack([u, v, rng.normal(size=n) * z_sigma], 1)
    tgt = np.zeros((n, 3))
    tgt[kind == 1] = [1.0, 0, 0]
    tgt[kind == 2] = [0, 1.0, 0]
    s = _pack(pos, np.tile([0, 0, 1.0], (n, 1)), tgt)
    return s, np.nonzero(kind > 0)[0].astype(np.uint32)


def synth_indoor(n, model_n, seed=6, noise_sigma=0.0):
    """Scene of config 4: a room (floor + 4 walls, 8 x 6 x 3) sampled at the model's density; three walls
    carry one planted panel each (in-plane rotation and offset known); door-frame decoy edges elsewhere.

    Returns dict(surfels, tangent_mask, instance_id, wall_id, walls = [(origin, a, b, normal)],
    instances = [(wall, u0, v0, angle)])."""
    rng = np.random.default_rng(seed)
    dens = model_n / 2.0
    X, Y, Z = 8.0, 6.0, 3.0
    walls = [  # origin, a (u direction), b (v direction): normal = a x b
        (np.zeros(3), np.array([X, 0, 0]), np.array([0, 0, Z])),             # y = 0
        (np.array([X, 0, 0]), np.array([0, Y, 0]), np.array([0, 0, Z])),     # x = X
        (np.array([X, Y, 0]), np.array([-X, 0, 0]), np.array([0, 0, Z])),    # y = Y
        (np.array([0, Y, 0]), np.array([0, -Y, 0]), np.array([0, 0, Z])),    # x = 0
        (np.zeros(3), np.array([0, Y, 0]), np.array([X, 0, 0])),             # floor, normal +z... (a x b = -z)
    ]
    planted = {0: (2.5, 0.9, 0.0), 1: (2.5, 0.5, np.pi / 2), 2: (4.0, 0.6, 0.35)}  # u0, v0, in-plane angle
    areas = np.array([np.linalg.norm(np.cross(a, b)) for (_, a, b) in walls])
    scale = min(1.0, 0.97 * n / (dens * areas.sum()))
    spacing = 1.0 / np.sqrt(dens * scale)
    pos_l, nrm_l, tgt_l, inst_l, wall_l, instances, wall_out = [], [], [], [], [], [], []
    for k, (o, a, b) in enumerate(walls):
        cnt = int(dens * scale * areas[k])
        la, lb = np.linalg.norm(a), np.linalg.norm(b)
        ea, eb = a / la, b / lb
        nrm = np.cross(ea, eb)
        u, v = rng.random(cnt) * la, rng.random(cnt) * lb
        pos = o + u[:, None] * ea + v[:, None] * eb
        tgt = np.zeros((cnt, 3))
        inst = np.full(cnt, -1, np.int32)
        if k in planted:
            u0, v0, ang = planted[k]
            c, s_ = np.cos(ang), np.sin(ang)
            pu = c * (u - u0) + s_ * (v - v0)   # panel coordinates of the wall point
            pv = -s_ * (u - u0) + c * (v - v0)
            sel = (pu >= 0) & (pu <= 2.0) & (pv >= 0) & (pv <= 1.0)
            kind = _panel_pattern(pu[sel], pv[sel], spacing)
            idx = np.nonzero(sel)[0]
            du = c * ea + s_ * eb      # panel u / v directions in the world
            dv = -s_ * ea + c * eb
            tgt[idx[kind == 1]] = du
            tgt[idx[kind == 2]] = dv
            inst[idx] = len(instances)
            instances.append((k, u0, v0, ang))
        # decoy: a door frame (two verticals + lintel) on every wall, outside the panel
        if k < 4:
            w = max(0.02, 1.5 * spacing)
            du0 = 0.6 if k not in planted else la - 1.4
            door_v = ((np.abs(u - du0) < w) | (np.abs(u - du0 - 0.9) < w)) & (v < 2.1) & (inst < 0)
            door_h = (np.abs(v - 2.1) < w) & (u > du0) & (u < du0 + 0.9) & (inst < 0)
            tgt[door_v] = eb
            tgt[door_h] = ea
        pos_l.append(pos)
        nrm_l.append(np.tile(nrm, (cnt, 1)))
        tgt_l.append(tgt)
        inst_l.append(inst)
        wall_l.append(np.full(cnt, k, np.int32))
        wall_out.append((o, a, b, nrm))
    used = sum(len(p) for p in pos_l)
    n_out = n - used
    po = rng.random((n_out, 3)) * [X, Y, Z]
    no = rng.normal(size=(n_out, 3))
    no /= np.linalg.norm(no, axis=1, keepdims=True)
    pos_l.append(po)
    nrm_l.append(no)
    tgt_l.append(np.zeros((n_out, 3)))
    inst_l.append(np.full(n_out, -1, np.int32))
    wall_l.append(np.full(n_out, -1, np.int32))
    pos, nrm, tgt = np.concatenate(pos_l), np.concatenate(nrm_l), np.concatenate(tgt_l)
    inst, wall = np.concatenate(inst_l), np.concatenate(wall_l)
    if noise_sigma > 0:
        pos = pos + rng.normal(size=pos.shape) * noise_sigma
    perm = rng.permutation(n)
    s = _pack(pos[perm], nrm[perm], tgt[perm])
    tmask = (np.linalg.norm(s[:, 9:12], axis=1) > 0.7).astype(np.int32)
    return dict(surfels=s, tangent_mask=tmask, instance_id=inst[perm], wall_id=wall[perm], walls=wall_out,
                instances=instances)


def synth_multi_model_scene(n, n_models=32, seed=7, min_points=20_000, max_points=100_000):
    """Scene of config 5: one planted instance of each of `n_models` different models (scaled L-prisms,
    min_points..max_points points at one common sampling density) on a grid of cells, plus a floor with
    crease lines and outliers.

    Returns (models, scene): models = [(surfels, tangent subset, resolution)], scene = dict(surfels,
    tangent_mask, instance_id, poses_scene_to_model [n_models, 4, 4])."""
    rng = np.random.default_rng(seed)
    # one sampling density for every object (a scanner does not know the models): the point count of a
    # model follows its surface area, overall size 0.45..1.0 with a mild anisotropic jitter
    base = np.linspace(np.sqrt(min_points / max_points), 1.0, n_models)
    scales = base[:, None] * (0.85 + 0.3 * rng.random((n_models, 3)))
    rects, creases = _rects_and_creases()
    areas = np.array([sum(np.linalg.norm(np.cross(a, b)) for (_, a, b, _) in _scaled(rects, creases, tuple(sc))[0])
                      for sc in scales])
    sizes = np.maximum(1000, (max_points * areas / areas.max()).astype(int))
    models, parts, inst_l, s2m = [], [], [], []
    cols = int(np.ceil(np.sqrt(2 * n_models)))
    for k in range(n_models):
        sc = tuple(float(x) for x in scales[k])
        M, sub = synth_model(int(sizes[k]), seed=1 + k, scale=sc)
        models.append((M, sub, resolution(M)))
        inst, _ = synth_model(int(sizes[k]), seed=1000 + k, scale=sc)
        R = random_rotation(rng)
        t = np.array([3.0 * (k % cols) + 1.5, 3.0 * (k // cols) + 1.5, 1.0 + rng.random()])
        inst[:, 0:3] = inst[:, 0:3] @ R.T + t
        inst[:, 4:7] = inst[:, 4:7] @ R.T
        inst[:, 9:12] = inst[:, 9:12] @ R.T
        parts.append(inst.astype(np.float32))
        inst_l.append(np.full(len(inst), k, np.int32))
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = t
        s2m.append(np.linalg.inv(T))
    used = sum(len(p) for p in parts)
    if used > 0.9 * n:
        raise ValueError("scene too small for the requested models")
    rows = (n_models + cols - 1) // cols
    X, Y = 3.0 * cols, 3.0 * rows
    n_out = int(0.03 * n)
    n_floor = n - used - n_out
    pf = np.stack([rng.random(n_floor) * X, rng.random(n_floor) * Y, np.zeros(n_floor)], 1)
    spacing = np.sqrt(X * Y / n_floor)
    tf = np.zeros((n_floor, 3))
    w = max(0.01, 1.5 * spacing)
    gx = np.abs(np.mod(pf[:, 0], 3.0) - 1.5) > 1.5 - w   # grid lines between the cells
    gy = np.abs(np.mod(pf[:, 1], 3.0) - 1.5) > 1.5 - w
    tf[gx] = [0, 1.0, 0]
    tf[gy & ~gx] = [1.0, 0, 0]
    parts.append(_pack(pf, np.tile([0, 0, 1.0], (n_floor, 1)), tf))
    po = rng.random((n_out, 3)) * [X, Y, 3.0]
    no = rng.normal(size=(n_out, 3))
    no /= np.linalg.norm(no, axis=1, keepdims=True)
    parts.append(_pack(po, no, np.zeros((n_out, 3))))
    inst_l += [np.full(n_floor, -1, np.int32), np.full(n_out, -1, np.int32)]
    s = np.concatenate(parts)
    inst = np.concatenate(inst_l)
    perm = rng.permutation(len(s))
    s, inst = np.ascontiguousarray(s[perm]), inst[perm]
    tmask = (np.linalg.norm(s[:, 9:12], axis=1) > 0.7).astype(np.int32)
    return models, dict(surfels=s, tangent_mask=tmask, instance_id=inst, poses_scene_to_model=np.array(s2m))
