"""CPU-side checks of the product's C ABI: the library loads, exports every symbol include/tm_hip.h
declares, its host-side scalar functions agree bit for bit with the oracle, and without a GPU the
compute entry points fail loudly (there is no CPU fallback). No device compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import tmo
from triplet_match_amd import build as tm_build
from triplet_match_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    tm_build.build()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 50
    for n in names:
        assert hasattr(L, n), "libtm_hip.so does not export " + n
    assert set(names) == set(capi.SIGNATURES), set(names) ^ set(capi.SIGNATURES)
    assert b"gfx950" in capi.lib().tm_version()


def test_library_contains_gfx950_code_object():
    blob = open(capi.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    # single backend: no other GPU targets baked in
    assert b"amdgcn-amd-amdhsa--gfx942" not in blob and b"amdgcn-amd-amdhsa--gfx90a" not in blob


def test_host_scalar_functions_match_oracle_bitwise():
    rng = np.random.default_rng(7)
    for _ in range(200):
        w = rng.integers(0, 2**32, size=int(rng.integers(1, 10)), dtype=np.uint64).astype(np.uint32)
        assert capi.murmur3_32(w, 0) == tmo.murmur3_32(w, 0)
        assert capi.hash_key(w) == tmo.murmur_key(w)
    for _ in range(2000):
        v, mn, rg = (float(np.float32(x)) for x in rng.normal(size=3))
        st = int(rng.integers(1, 50))
        assert capi.discretize_range(v, mn, abs(rg), st) == tmo.discretize_range(v, mn, abs(rg), st)
        assert capi.discretize_step(abs(v), abs(rg) + 0.01) == tmo.discretize_step(abs(v), abs(rg) + 0.01)
        y, x = float(np.float32(rng.normal())), float(np.float32(rng.normal()))
        assert capi.atan2f(y, x) == tmo.atan2f(y, x)
    model, subset = synth.synth_model(500, seed=9)
    bmin = np.array([0.2, 0, 0, 0.2], np.float32)
    bmax = np.array([1.4, 1.6, 1.6, 1.4], np.float32)
    step = capi.default_angle_step()
    for _ in range(500):
        i, j = rng.choice(subset, size=2, replace=False)
        fa, fb = capi.feature(model[i], model[j]), tmo.feature(model[i], model[j])
        assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
        assert np.array_equal(capi.discretize_feature(fa, bmin, bmax, 20.0, step),
                              tmo.discretize_feature(fb, bmin, bmax, 20.0, step))
        assert capi.valid(fa, bmin, bmax) == tmo.valid(fb, bmin, bmax)
        k, l = rng.choice(subset, size=2, replace=False)
        Ta = capi.base_transform(model[i, 0:3], model[j, 0:3], model[i, 9:12], model[k, 0:3],
                                 model[l, 0:3], model[k, 9:12])
        Tb = tmo.base_transform(model[i, 0:3], model[j, 0:3], model[i, 9:12], model[k, 0:3],
                                model[l, 0:3], model[k, 9:12])
        assert np.array_equal(Ta.view(np.uint32), Tb.view(np.uint32))
    for _ in range(500):
        S = int(rng.integers(1, 3_000_000))
        t = int(rng.integers(1, S + 1))
        c = int(rng.integers(0, t + 1))
        assert capi.early_drop_upper(t, S, c) == tmo.early_drop_upper(t, S, c)


def test_appendix_c_kats_on_product_host_functions():
    assert capi.hash_key([3, 9, 9, 3]) == 0x29DA4D9D
    assert capi.hash_key([19, 0, 17, 19]) == 0xC8D595AD
    assert capi.discretize_step(float(np.float32(np.pi / 2)), float(np.float32(0.17453294))) == 9
    assert capi.early_drop_upper(100, 1000, 10) == 136


def test_traits_host_math_matches_oracle():
    rng = np.random.default_rng(11)
    ms = np.zeros(capi.STATE_FLOATS, np.float32)
    ms[1] = 0.3
    for kind, ns in ((capi.IDENTITY, 1), (capi.PLANE, 1), (capi.PLANE2, 3), (capi.CYLINDER, 2)):
        for _ in range(50):
            smp = np.zeros((ns, 12), np.float32)
            smp[:, 0:3] = rng.normal(size=(ns, 3))
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            if kind == capi.PLANE2:
                # coplanar samples with consistent normals
                e1 = np.cross(n, rng.normal(size=3))
                e1 /= np.linalg.norm(e1)
                e2 = np.cross(n, e1)
                smp[:, 0:3] = smp[0, 0:3] + rng.normal(size=(ns, 1)) * e1 + rng.normal(size=(ns, 1)) * e2
                smp[:, 4:7] = n
            else:
                smp[:, 4:7] = rng.normal(size=(ns, 3))
                smp[:, 4:7] /= np.linalg.norm(smp[:, 4:7], axis=1, keepdims=True)
            a = capi.traits_init_from_samples(kind, ms, smp.ravel())
            b = tmo.traits_init_from_samples(kind, ms, smp.ravel())
            assert (a is None) == (b is None)
            if a is None:
                continue
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            s = np.zeros(12, np.float32)
            s[0:3] = rng.normal(size=3)
            s[4:7] = n
            t = np.cross(n, rng.normal(size=3))
            s[9:12] = t / np.linalg.norm(t)
            pa, pb = capi.traits_project(kind, a, s[0:3]), tmo.traits_project(kind, b, s[0:3])
            assert (pa is None) == (pb is None)
            if pa is not None:
                assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
                ua, ub = capi.traits_unproject(kind, a, pa), tmo.traits_unproject(kind, b, pb)
                assert np.allclose(ua, ub, atol=1e-5)  # libm cos/sin on both sides; fp tolerance 1e-5
            assert np.array_equal(capi.traits_tangent(kind, a, s).view(np.uint32),
                                  tmo.traits_tangent(kind, b, s).view(np.uint32))
            assert np.array_equal(capi.traits_normal(kind, a, s).view(np.uint32),
                                  tmo.traits_normal(kind, b, s).view(np.uint32))
            p0, p1 = rng.normal(size=3), rng.normal(size=3)
            assert capi.traits_intrinsic_distance(kind, a, p0, p1) == \
                tmo.traits_intrinsic_distance(kind, b, p0, p1)


def test_best_select_is_deterministic():
    recs = []
    for hid, cnt, sc in ((5, 10, 0.5), (3, 12, 0.1), (9, 12, 0.4), (2, 12, 0.4), (0xFFFFFFFF, 99, 9.0)):
        r = capi.BestRecord()
        r.inlier_count, r.hyp_id, r.score = cnt, hid, sc
        recs.append(r)
    b = capi.best_select(recs)
    assert (b.inlier_count, b.hyp_id) == (12, 2)  # count desc, score desc, id asc; "none" ignored
    b2 = capi.best_select(recs[::-1])
    assert (b2.inlier_count, b2.hyp_id, b2.score) == (12, 2, 0.4)


def test_no_device_fails_loudly():
    """On a box without a GPU the compute path must error, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu tests")
    with pytest.raises(capi.TmError) as e:
        capi.Context(0)
    assert e.value.code == capi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_header_is_plain_c99_and_links_from_c(tmp_path):
    """include/tm_hip.h is consumed by a C99 translation unit (-pedantic); without a GPU the library answers
    TM_ERR_NO_DEVICE with its "no CPU fallback" message instead of computing anything."""
    import subprocess
    from triplet_match_amd import build as tm_build
    tm_build.build()
    exe = str(tmp_path / "abi_c99")
    libdir = os.path.join(ROOT, "triplet_match_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_c99.c"), "-o", exe, "-L" + libdir, "-ltm_hip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.check_output([exe]).decode()
    import torch
    if not torch.cuda.is_available():
        assert out.startswith("-6 ") and "no CPU fallback" in out
    else:
        assert out.startswith("0 ")


def test_product_does_not_touch_the_oracle():
    """Nothing under triplet_match_amd/ or include/ includes, imports or links the oracle; the shared library has
    no dependency on it."""
    import re
    import subprocess
    offenders = []
    for base in ("triplet_match_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if not f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    continue
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r'#include\s*[<"][^>"]*oracle|tm_oracle|libtm_oracle|from oracle|import oracle|tmo_[a-z]', text):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    from triplet_match_amd import build as tm_build
    lib = tm_build.build()
    deps = subprocess.check_output(["ldd", lib]).decode()
    assert "oracle" not in deps
    syms = subprocess.check_output(["nm", "-D", lib]).decode()
    assert "tmo_" not in syms


def test_committed_pmc_passes_belong_to_these_kernel_sources():
    """bench.py quotes roofline.traffic from the newest profiles/rNN/traffic.json whose hash matches the scoring kernels'
    sources (the marked region of csrc/tm_scene.hip + tm_score.h / tm_math.h / tm_table.h); without a match the line's
    `frac` is null. This fails when the kernels were edited after the last PMC passes: run tools/profile_round5.sh on a
    GPU box (stages `pmc folded ed`) and commit the refreshed profiles/ files."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    text = open(os.path.join(root, "triplet_match_amd", "csrc", "tm_scene.hip"), "rb").read()
    assert text.count(b"/* >>> scoring kernels") == 1 and text.count(b"/* <<< scoring kernels") == 1
    have = []
    for rnd in bench._profile_rounds():
        p = os.path.join(root, "profiles", rnd, "traffic.json")
        if os.path.exists(p):
            have.append(json.load(open(p)).get("source_hash"))
    assert bench.source_hash() in have, "scoring kernels changed since the committed PMC passes"
    ed = [json.load(open(os.path.join(root, "profiles", rnd, "early_drop_traffic.json"))).get("source_hash")
          for rnd in bench._profile_rounds() if os.path.exists(os.path.join(root, "profiles", rnd, "early_drop_traffic.json"))]
    assert bench.ed_source_hash() in ed, "early-drop kernels changed since the committed PMC pass"
