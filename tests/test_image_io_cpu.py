"""HDRFilm::write equivalents (src/films/hdrfilm.cpp:414-560): EXR (OpenEXR 2, scanline, uncompressed, FLOAT) and PFM
written by the host library and parsed back by independent readers written against the public format specifications."""
import os
import struct

import numpy as np
import pytest


def read_exr(path):
    b = open(path, "rb").read(); p = 0
    magic, version = struct.unpack_from("<ii", b, p); p += 8
    assert magic == 20000630 and (version & 0xff) == 2 and not (version & 0x200), "single-part scanline file expected"
    attrs = {}
    while b[p] != 0:
        e = b.index(b"\0", p); name = b[p:e].decode(); p = e + 1
        e = b.index(b"\0", p); typ = b[p:e].decode(); p = e + 1
        (size,) = struct.unpack_from("<i", b, p); p += 4
        attrs[name] = (typ, b[p:p + size]); p += size
    p += 1
    chans = []; q = 0; cl = attrs["channels"][1]
    while cl[q] != 0:
        e = cl.index(b"\0", q); n = cl[q:e].decode(); q = e + 1
        ptype, plinear, xs, ys = struct.unpack_from("<iB3xii", cl, q); q += 16
        assert ptype == 2 and xs == 1 and ys == 1
        chans.append(n)
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1]); w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["displayWindow"][1] == attrs["dataWindow"][1] and chans == sorted(chans)
    offsets = struct.unpack_from("<%dQ" % h, b, p)
    img = {c: np.empty((h, w), np.float32) for c in chans}
    for y in range(h):
        o = offsets[y]; yy, size = struct.unpack_from("<ii", b, o); o += 8
        assert yy == y0 + y and size == len(chans) * w * 4
        for c in chans:
            img[c][y] = np.frombuffer(b, "<f4", w, o); o += 4 * w
    assert offsets[-1] + 8 + len(chans) * w * 4 == len(b)
    return img


def read_pfm(path):
    with open(path, "rb") as f:
        kind = f.readline().strip(); w, h = map(int, f.readline().split()); scale = float(f.readline())
        c = 3 if kind == b"PF" else 1
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, c)
    return data[::-1]


@pytest.mark.parametrize("c", [3, 4])
def test_exr_roundtrip(mi, tmp_path, c):
    rng = np.random.default_rng(c)
    img = rng.normal(size=(7, 13, c)).astype(np.float32); img[0, 0, 0] = np.inf; img[1, 2, 1] = 1e-30
    path = os.path.join(tmp_path, "a.exr")
    mi.Bitmap(img).write(path)
    got = read_exr(path)
    names = "RGBA"[:c]
    assert sorted(got) == sorted(names)
    for k, n in enumerate(names):
        assert np.array_equal(got[n], img[..., k])


def test_pfm_roundtrip_and_errors(mi, tmp_path):
    img = np.random.default_rng(0).random((5, 9, 3)).astype(np.float32)
    path = os.path.join(tmp_path, "a.pfm"); mi.write_bitmap(path, img)
    assert np.array_equal(read_pfm(path), img)
    mi.write_bitmap(os.path.join(tmp_path, "g.pfm"), img[..., 0])
    assert np.array_equal(read_pfm(os.path.join(tmp_path, "g.pfm"))[..., 0], img[..., 0])
    with pytest.raises(RuntimeError):
        mi.Bitmap(img).write(os.path.join(tmp_path, "a.png"))
    with pytest.raises(RuntimeError):
        mi.Bitmap(img[..., :2]).write(os.path.join(tmp_path, "b.exr"))
    with pytest.raises(RuntimeError):
        mi.Bitmap(img).write(os.path.join(tmp_path, "nodir", "b.exr"))


def test_bitmap_texture_sources(mi, tmp_path):
    """BitmapTexture (src/textures/bitmap.cpp:175-260): the same texels from `data`, a Bitmap object and an OpenEXR / PFM file; luminance
    images are replicated, alpha is dropped; exactly one source may be given"""
    import os
    import pytest
    tex = np.random.default_rng(0).random((8, 6, 3)).astype(np.float32)
    ref = mi.load_dict({"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "raw": True}}).texture
    for ext in ("exr", "pfm"):
        p = os.path.join(tmp_path, "t." + ext); mi.Bitmap(tex).write(p)
        assert np.array_equal(mi.load_dict({"type": "diffuse", "reflectance": {"type": "bitmap", "filename": p}}).texture, ref)
    assert np.array_equal(mi.load_dict({"type": "diffuse", "reflectance": {"type": "bitmap", "bitmap": mi.Bitmap(tex)}}).texture, ref)
    grey = mi.load_dict({"type": "diffuse", "reflectance": {"type": "bitmap", "bitmap": mi.Bitmap(tex[:, :, :1])}}).texture
    assert grey.shape == (8, 6, 3) and np.array_equal(grey[:, :, 2], tex[:, :, 0])
    rgba = np.concatenate([tex, np.ones((8, 6, 1), np.float32)], 2)
    assert np.array_equal(mi.load_dict({"type": "diffuse", "reflectance": {"type": "bitmap", "data": rgba}}).texture, ref)
    with pytest.raises(RuntimeError, match="exactly one"):
        mi.load_dict({"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "bitmap": mi.Bitmap(tex)}})
    with pytest.raises(RuntimeError, match="not found"):
        mi.load_dict({"type": "diffuse", "reflectance": {"type": "bitmap", "filename": os.path.join(tmp_path, "missing.exr")}})


def test_corrupt_files_are_errors_not_crashes(tmp_path):
    """the loaders must not trust file contents -- truncated / oversized headers give an error message, never a crash or a huge allocation"""
    import ctypes as C, struct
    import mitsuba3_amd as mi
    from mitsuba3_amd import _capi
    L = mi.lib()

    def read(path):
        img = _capi.HarImage() if hasattr(_capi, "HarImage") else None
        class Img(C.Structure):
            _fields_ = [("data", C.POINTER(C.c_float)), ("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32)]
        img = img or Img()
        rc = L.har_image_read(str(path).encode(), C.byref(img))
        msg = L.har_last_error().decode()
        if rc == 0:
            L.har_image_free(C.byref(img))
        return rc, msg

    good = tmp_path / "ok.exr"
    arr = np.random.default_rng(0).uniform(0, 1, (5, 7, 3)).astype(np.float32)
    mi.write_bitmap(str(good), arr) if hasattr(mi, "write_bitmap") else mi.Bitmap(arr).write(str(good))
    blob = good.read_bytes()
    assert read(good)[0] == 0
    # dataWindow attribute with a wrong size / an absurd window / int32 overflow in max - min
    i = blob.index(b"dataWindow\x00box2i\x00")
    j = i + len(b"dataWindow\x00box2i\x00")
    bad = [blob[:j] + struct.pack("<i", 4) + blob[j + 4:],                                                    # size 4 instead of 16
           blob[:j + 4] + struct.pack("<4i", 0, 0, 2 ** 31 - 1, 2 ** 31 - 1) + blob[j + 20:],                 # 2^31 x 2^31 pixels
           blob[:j + 4] + struct.pack("<4i", -2 ** 31, -2 ** 31, 2 ** 31 - 1, 2 ** 31 - 1) + blob[j + 20:],   # overflowing subtraction
           blob[:40], blob[: len(blob) // 2]]                                                                # truncated
    for k, b in enumerate(bad):
        f = tmp_path / ("bad%d.exr" % k); f.write_bytes(b)
        rc, msg = read(f)
        assert rc != 0 and "Error while loading" in msg, (k, rc, msg)
    # PFM: absurd dimensions, truncated payload
    for k, b in enumerate([b"PF\n999999999 999999999\n-1.0\n" + b"\x00" * 64, b"PF\n4 4\n-1.0\n" + b"\x00" * 16, b"PF\n-4 4\n-1.0\n"]):
        f = tmp_path / ("bad%d.pfm" % k); f.write_bytes(b)
        rc, msg = read(f)
        assert rc != 0, (k, msg)
