"""PCD reader/writer of the drop-in API (include/triplet_match/pcd_io.hpp; replaces the reference's
pcl::io::loadPCDFile call, include/impl/pointcloud.hpp:58-64). The files are produced here, from the
published format description, by an independent writer (including its own LZF compressor); the C++
reader must reproduce the 48-byte surfel records bit for bit. CPU only."""
import os
import subprocess

import numpy as np
import pytest

from test_cpp_dropin import _compile
from triplet_match_amd import synth

NAMES = ["x", "y", "z", "normal_x", "normal_y", "normal_z", "rgba", "radius", "confidence", "curvature"]
SLOT = [0, 1, 2, 4, 5, 6, 8, 9, 10, 11]


def lzf_compress(data: bytes) -> bytes:
    """Greedy LZF (literal runs <= 32, back references of 3..264 bytes within 8192)."""
    out = bytearray()
    n, ip, lit = len(data), 0, 0
    table = {}

    def flush(end):
        nonlocal lit
        while lit < end:
            run = min(32, end - lit)
            out.append(run - 1)
            out.extend(data[lit:lit + run])
            lit += run

    while ip + 2 < n:
        key = data[ip:ip + 3]
        ref = table.get(key)
        table[key] = ip
        if ref is not None and ip - ref <= 8192:
            ln = 3
            while ln < 264 and ip + ln < n and data[ref + ln] == data[ip + ln]:
                ln += 1
            flush(ip)
            dist, l = ip - ref - 1, ln - 2
            if l < 7:
                out.append((l << 5) | (dist >> 8))
            else:
                out.append((7 << 5) | (dist >> 8))
                out.append(l - 7)
            out.append(dist & 255)
            ip += ln
            lit = ip
        else:
            ip += 1
    flush(n)
    return bytes(out)


def header(fields, sizes, types, counts, n, mode, width=None, height=1):
    return ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\n"
            "WIDTH %d\nHEIGHT %d\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n"
            % (" ".join(fields), " ".join(map(str, sizes)), " ".join(types), " ".join(map(str, counts)),
               n if width is None else width, height, n, mode)).encode()


def surfels(n=3000):
    s, _ = synth.synth_model(n, seed=4)
    words = s.view(np.uint32).copy()
    words[:, 8] = 0xFF000000 | np.arange(n, dtype=np.uint32)  # rgba word
    return words.view(np.float32)


def expected(s):
    e = np.zeros_like(s)
    e[:, 3] = 1.0
    e.view(np.uint32)[:, SLOT] = s.view(np.uint32)[:, SLOT]
    return e


@pytest.fixture(scope="module")
def exe():
    return _compile("test_pcd")


def read_back(exe, path, tmp_path):
    raw = str(tmp_path / "out.raw")
    subprocess.check_call([exe, "dump", path, raw])
    return np.fromfile(raw, np.float32).reshape(-1, 12)


def test_binary_surfel_file(exe, tmp_path):
    s = surfels()
    p = str(tmp_path / "a.pcd")
    with open(p, "wb") as f:
        f.write(header(NAMES, [4] * 10, list("FFFFFFUFFF"), [1] * 10, len(s), "binary"))
        f.write(np.ascontiguousarray(s.view(np.uint32)[:, SLOT]).tobytes())
    got = read_back(exe, p, tmp_path)
    assert np.array_equal(got.view(np.uint32), expected(s).view(np.uint32))


def test_binary_with_padding_reordered_and_wide_fields(exe, tmp_path):
    """'_' padding with COUNT 4, an unknown field, double-precision coordinates, 16-bit intensity,
    organised cloud (WIDTH x HEIGHT), fields in a different order, no normals at all."""
    s = surfels(1200)
    n = len(s)
    rec = np.dtype([("curvature", "<f4"), ("_", "u1", (4,)), ("z", "<f8"), ("x", "<f8"), ("intensity", "<u2"),
                    ("y", "<f8"), ("rgb", "<f4")])
    a = np.zeros(n, rec)
    a["curvature"], a["x"], a["y"], a["z"] = s[:, 11], s[:, 0], s[:, 1], s[:, 2]
    a["rgb"] = s[:, 8]
    a["intensity"] = 7
    p = str(tmp_path / "b.pcd")
    with open(p, "wb") as f:
        f.write(header(["curvature", "_", "z", "x", "intensity", "y", "rgb"], [4, 1, 8, 8, 2, 8, 4],
                       list("FUFFUFF"), [1, 4, 1, 1, 1, 1, 1], n, "binary", width=n // 4, height=4))
        f.write(a.tobytes())
    got = read_back(exe, p, tmp_path)
    e = np.zeros_like(s)
    e[:, 3] = 1.0
    e[:, [0, 1, 2, 11]] = s[:, [0, 1, 2, 11]]
    e.view(np.uint32)[:, 8] = s.view(np.uint32)[:, 8]
    assert np.array_equal(got.view(np.uint32), e.view(np.uint32))


def test_ascii_file(exe, tmp_path):
    s = surfels(500)
    s[3, 0] = np.nan
    s[4, 1] = np.inf
    p = str(tmp_path / "c.pcd")
    with open(p, "wb") as f:
        f.write(header(NAMES, [4] * 10, list("FFFFFFUFFF"), [1] * 10, len(s), "ascii"))
        for r in s:
            w = r.view(np.uint32)
            toks = ["%.9g" % r[k] if k != 8 else "%d" % w[8] for k in SLOT]
            f.write((" ".join(toks) + "\r\n").encode())  # CRLF tolerated
    got = read_back(exe, p, tmp_path)
    assert np.array_equal(got.view(np.uint32), expected(s).view(np.uint32))


def test_binary_compressed_file(exe, tmp_path):
    s = surfels(4000)
    cols = np.ascontiguousarray(s.view(np.uint32)[:, SLOT].T)  # field-major
    raw = cols.tobytes()
    comp = lzf_compress(raw)
    assert len(comp) < 0.8 * len(raw)  # normals/tangents repeat: back references are exercised
    p = str(tmp_path / "d.pcd")
    with open(p, "wb") as f:
        f.write(header(NAMES, [4] * 10, list("FFFFFFUFFF"), [1] * 10, len(s), "binary_compressed"))
        f.write(np.array([len(comp), len(raw)], np.uint32).tobytes())
        f.write(comp)
    got = read_back(exe, p, tmp_path)
    assert np.array_equal(got.view(np.uint32), expected(s).view(np.uint32))


def test_writer_round_trips_and_errors(exe, tmp_path):
    out = subprocess.check_output([exe, "roundtrip", str(tmp_path)]).decode()
    assert out.strip() == "ok"
    # the compressed file the C++ writer produced really is compressed
    assert os.path.getsize(tmp_path / "rt_binary_compressed.pcd") < 0.8 * os.path.getsize(tmp_path / "rt_binary.pcd")
    # malformed inputs: pcd::read throws (exit 3), from_pcd yields an empty cloud like the reference
    bad = str(tmp_path / "bad.pcd")
    open(bad, "wb").write(header(NAMES, [4] * 10, list("FFFFFFUFFF"), [1] * 10, 100, "binary") + b"\0" * 40)
    assert subprocess.call([exe, "dump", bad, str(tmp_path / "x.raw")], stderr=subprocess.DEVNULL) == 3
    assert subprocess.call([exe, "dump", str(tmp_path / "missing.pcd"), str(tmp_path / "x.raw")],
                           stderr=subprocess.DEVNULL) == 3
    corrupt = str(tmp_path / "corrupt.pcd")
    open(corrupt, "wb").write(header(NAMES, [4] * 10, list("FFFFFFUFFF"), [1] * 10, 10, "binary_compressed")
                              + np.array([3, 400], np.uint32).tobytes() + bytes([0xE0, 0x00, 0x05]))
    assert subprocess.call([exe, "dump", corrupt, str(tmp_path / "x.raw")], stderr=subprocess.DEVNULL) == 3
    n = subprocess.check_output([exe, "from_pcd", str(tmp_path / "missing.pcd")], stderr=subprocess.DEVNULL)
    assert n.strip() == b"0"
    n = subprocess.check_output([exe, "from_pcd", str(tmp_path / "rt_ascii.pcd")])
    assert n.strip() == b"5000"


def test_reader_survives_mutated_files(exe, tmp_path):
    """Byte-flipped and truncated files end in a clean error (exit 3) or a successful read, never a crash."""
    import random
    subprocess.check_output([exe, "roundtrip", str(tmp_path)])
    rnd = random.Random(1)
    for mode in ("ascii", "binary", "binary_compressed"):
        data = open(tmp_path / ("rt_%s.pcd" % mode), "rb").read()[:40000]
        for _ in range(40):
            b = bytearray(data)
            for _ in range(rnd.choice([1, 1, 2, 5, 20])):
                b[rnd.randrange(0, min(len(b), 600 if rnd.random() < 0.6 else len(b)))] = rnd.randrange(256)
            if rnd.random() < 0.3:
                b = b[:rnd.randrange(1, len(b))]
            p = tmp_path / "fuzz.pcd"
            p.write_bytes(bytes(b))
            rc = subprocess.call([exe, "dump", str(p), str(tmp_path / "fuzz.raw")], stderr=subprocess.DEVNULL, timeout=60)
            assert rc in (0, 3), (mode, rc)
