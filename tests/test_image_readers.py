"""PNG / PPM decoding of the host mirror (used for MTL textures, .scn textures and environment maps): every colour type and
bit depth stb_image handles without a lossy codec, every scanline filter, against the pixels the file was made from and,
where the compiled reference is present, against the reference's own load_image (stb_image)."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from pathtracer_amd import capi


def png_bytes(arr, ctype, depth, palette=None, filters=(0, 1, 2, 3, 4)):
    """arr: (H, W, channels) samples already in the file's value range.  Rows get the filters in turn."""
    H, W, ch = arr.shape
    bits = ch * depth
    rows = []
    for y in range(H):
        if depth == 8:
            rows.append(arr[y].astype(np.uint8).tobytes())
        else:
            v = arr[y].reshape(-1).astype(np.uint8)
            per = 8 // depth
            pad = (-len(v)) % per
            v = np.concatenate([v, np.zeros(pad, np.uint8)]).reshape(-1, per)
            byte = np.zeros(len(v), np.uint16)
            for k in range(per):
                byte = (byte << depth) | v[:, k]
            rows.append(byte.astype(np.uint8).tobytes())
    bpp = max(1, bits // 8)
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)]
        out = bytearray(len(row))
        for x in range(len(row)):
            a = row[x - bpp] if x >= bpp else 0
            b = prev[x]
            c = prev[x - bpp] if x >= bpp else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = a
            elif ft == 2: pred = b
            elif ft == 3: pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[x] = (row[x] - pred) & 255
        raw.append(ft); raw += out
        prev = row

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    z = zlib.compress(bytes(raw), 6)
    body = chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 0))
    if palette is not None:
        body += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    half = len(z) // 2                      # two IDAT chunks: the stream must be concatenated
    body += chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")
    return b"\x89PNG\r\n\x1a\n" + body


CASES = [(0, 8), (0, 4), (0, 2), (0, 1), (2, 8), (3, 8), (3, 4), (3, 2), (3, 1), (4, 8), (6, 8)]


def make_case(ctype, depth, rng):
    W, H = 37, 23
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    arr = rng.integers(0, 1 << depth, (H, W, ch))
    palette = rng.integers(0, 256, ((1 << depth), 3)) if ctype == 3 else None
    scale = {1: 255, 2: 85, 4: 17, 8: 1}[depth]
    if ctype == 0: rgb = np.repeat(arr[..., :1] * scale, 3, -1)
    elif ctype == 4: rgb = np.repeat(arr[..., :1], 3, -1)
    elif ctype == 3: rgb = palette[arr[..., 0]]
    else: rgb = arr[..., :3]
    return png_bytes(arr, ctype, depth, palette), rgb.astype(np.uint8)


def host_read(path):
    mipt, host = capi.load()
    W, H = C.c_int(0), C.c_int(0)
    buf = (C.c_ubyte * (1 << 22))()
    err = C.create_string_buffer(256)
    rc = host.mh_read_image(str(path).encode(), buf, len(buf), C.byref(W), C.byref(H), err, 256)
    if rc != 0:
        raise capi.MiptError(err.value.decode())
    return np.frombuffer(buf, np.uint8, W.value * H.value * 3).reshape(H.value, W.value, 3).copy()


@pytest.mark.parametrize("ctype,depth", CASES)
def test_png_decoding(tmp_path, ctype, depth):
    data, want = make_case(ctype, depth, np.random.default_rng(100 * ctype + depth))
    p = tmp_path / f"t_{ctype}_{depth}.png"
    p.write_bytes(data)
    assert np.array_equal(host_read(p), want)
    from oracle import binding
    if binding.ref_available():                      # the reference's own load_image (stb_image; it flips the rows)
        R = binding.Ref()
        W, H = C.c_int(0), C.c_int(0)
        buf = (C.c_ubyte * (1 << 22))()
        assert R.lib.ref_load_image(str(p).encode(), buf, len(buf), C.byref(W), C.byref(H)) == 0
        ref = np.frombuffer(buf, np.uint8, W.value * H.value * 3).reshape(H.value, W.value, 3)[::-1]
        assert np.array_equal(ref, want)


def test_undecodable_images_are_refused(tmp_path):
    data, _ = make_case(2, 8, np.random.default_rng(1))
    (tmp_path / "cut.png").write_bytes(data[: len(data) // 2])
    with pytest.raises(capi.MiptError):
        host_read(tmp_path / "cut.png")
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0" + bytes(64))          # a JPEG header with nothing behind it
    with pytest.raises(capi.MiptError):
        host_read(tmp_path / "x.jpg")
    (tmp_path / "x.tga").write_bytes(bytes(18) + bytes(64))
    with pytest.raises(capi.MiptError, match="JPEG, PNG"):
        host_read(tmp_path / "x.tga")
    ihdr16 = data.replace(struct.pack(">IIBB", 37, 23, 8, 2), struct.pack(">IIBB", 37, 23, 16, 2), 1)
    (tmp_path / "deep.png").write_bytes(ihdr16)
    with pytest.raises(capi.MiptError, match="16-bit"):
        host_read(tmp_path / "deep.png")


@pytest.mark.parametrize("bpp,topdown", [(24, False), (32, False), (24, True), (8, False)])
def test_bmp_decoding(tmp_path, bpp, topdown):
    rng = np.random.default_rng(bpp + topdown)
    W, H = 29, 17                                       # 29 * 3 bytes is not a multiple of 4: rows are padded
    if bpp == 8:
        pal = rng.integers(0, 256, (256, 3))
        idx = rng.integers(0, 256, (H, W))
        want = pal[idx].astype(np.uint8)
        pix = idx.astype(np.uint8)[..., None]
    else:
        want = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        pix = want[..., ::-1]
        if bpp == 32:
            pix = np.concatenate([pix, np.full((H, W, 1), 255, np.uint8)], -1)
    stride = ((W * bpp + 31) // 32) * 4
    rows = [pix[y].tobytes().ljust(stride, b"\0") for y in (range(H) if topdown else range(H - 1, -1, -1))]
    palbytes = b"".join(bytes([int(c[2]), int(c[1]), int(c[0]), 0]) for c in pal) if bpp == 8 else b""
    offset = 54 + len(palbytes)
    hdr = b"BM" + struct.pack("<IHHI", offset + stride * H, 0, 0, offset) + struct.pack("<IiiHHIIiiII", 40, W, -H if topdown else H, 1, bpp, 0, stride * H, 2835, 2835, 256 if bpp == 8 else 0, 0)
    p = tmp_path / "t.bmp"
    p.write_bytes(hdr + palbytes + b"".join(rows))
    assert np.array_equal(host_read(p), want)
    from oracle import binding
    if binding.ref_available():
        R = binding.Ref()
        Wc, Hc = C.c_int(0), C.c_int(0)
        buf = (C.c_ubyte * (1 << 22))()
        assert R.lib.ref_load_image(str(p).encode(), buf, len(buf), C.byref(Wc), C.byref(Hc)) == 0
        ref = np.frombuffer(buf, np.uint8, Wc.value * Hc.value * 3).reshape(Hc.value, Wc.value, 3)[::-1]
        assert np.array_equal(ref, want)


def test_jpeg_decoding_matches_the_reference_fixture(tmp_path):
    """Baseline and progressive JPEG (4:4:4 / 4:2:2 / 4:2:0, grey, restart markers, optimised tables, sizes that are not
    multiples of the MCU) against the pixels the reference's stb_image made of the same files
    (tests/golden/jpeg_cases.npz, tests/golden/make_golden.py --jpeg): bit for bit."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz"))
    for n in range(int(g["count"])):
        p = tmp_path / f"case{n}.jpg"
        p.write_bytes(g[f"file{n}"].tobytes())
        got = host_read(p)
        assert got.shape == g[f"rgb{n}"].shape and np.array_equal(got, g[f"rgb{n}"]), f"JPEG case {n}"


def test_jpeg_against_live_reference(tmp_path):
    from oracle import binding
    PIL = pytest.importorskip("PIL.Image")
    if not binding.ref_available():
        pytest.skip("compiled reference not present")
    rng = np.random.default_rng(4)
    R = binding.Ref()
    for h, w, sub, q, prog in ((50, 70, 2, 85, False), (50, 70, 1, 50, True), (23, 31, 0, 95, True), (64, 64, 2, 20, False)):
        img = rng.integers(0, 256, (h // 4 + 1, w // 4 + 1, 3), dtype=np.uint8).repeat(4, 0).repeat(4, 1)[:h, :w]
        p = tmp_path / "t.jpg"
        PIL.fromarray(img, "RGB").save(str(p), quality=q, subsampling=sub, progressive=prog)
        W, H = C.c_int(0), C.c_int(0)
        buf = (C.c_ubyte * (1 << 22))()
        assert R.lib.ref_load_image(str(p).encode(), buf, len(buf), C.byref(W), C.byref(H)) == 0
        ref = np.frombuffer(buf, np.uint8, W.value * H.value * 3).reshape(H.value, W.value, 3)[::-1]
        assert np.array_equal(host_read(p), ref)


def test_jpeg_kinds_outside_the_decoder_are_refused(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    p = tmp_path / "cmyk.jpg"
    PIL.fromarray(np.zeros((8, 8, 4), np.uint8), "CMYK").save(str(p))
    with pytest.raises(capi.MiptError, match="CMYK"):
        host_read(p)
    q = tmp_path / "cut.jpg"
    q.write_bytes(b"\xff\xd8\xff\xdb\x00")
    with pytest.raises(capi.MiptError):
        host_read(q)


def host_write(path, rgb):
    mipt, host = capi.load()
    err = C.create_string_buffer(256)
    rgb = np.ascontiguousarray(rgb, np.uint8)
    host.mh_save_image.restype = C.c_int
    rc = host.mh_save_image(str(path).encode(), rgb.ctypes.data_as(C.POINTER(C.c_ubyte)), rgb.shape[1], rgb.shape[0], err, 256)
    if rc != 0:
        raise capi.MiptError(err.value.decode())


@pytest.mark.parametrize("name", ["out.png", "OUT.PNG", "frame.bmp", "frame.ppm"])
def test_save_image_by_extension_round_trips(tmp_path, name):
    """save_image's rule (utils.cpp:178-234): the container follows the extension (lower-cased).  What the writer
    produces, the host mirror's own readers decode to the same pixels; a PNG is also a valid file for zlib + the PNG
    chunk layout (checked independently here)."""
    rng = np.random.default_rng(len(name))
    img = rng.integers(0, 256, (23, 37, 3), dtype=np.uint8)
    img[5:12, 3:30] = img[5:6, 3:4]                    # a flat patch and a gradient: exercises the row filters
    img[14:20] = (np.arange(37)[None, :, None] * 6 + np.arange(3)).astype(np.uint8)
    p = tmp_path / name
    host_write(p, img)
    assert np.array_equal(host_read(p), img)
    data = p.read_bytes()
    if name.lower().endswith(".png"):
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat, seen = 8, b"", []
        while pos < len(data):
            n, tag = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xffffffff
            seen.append(tag)
            if tag == b"IHDR":
                assert struct.unpack(">IIBBBBB", body) == (37, 23, 8, 2, 0, 0, 0)
            if tag == b"IDAT":
                idat += body
            pos += 12 + n
        assert seen[0] == b"IHDR" and seen[-1] == b"IEND"
        assert len(zlib.decompress(idat)) == 23 * (37 * 3 + 1)
    elif name.endswith(".bmp"):
        assert data[:2] == b"BM" and struct.unpack("<I", data[2:6])[0] == len(data)
    else:
        assert data.startswith(b"P6\n37 23\n255\n")


def test_save_image_tga_and_refused_extensions(tmp_path):
    img = np.random.default_rng(3).integers(0, 256, (5, 7, 3), dtype=np.uint8)
    host_write(tmp_path / "a.tga", img)
    d = (tmp_path / "a.tga").read_bytes()
    assert d[2] == 2 and struct.unpack("<HH", d[12:16]) == (7, 5) and d[16] == 24
    assert np.array_equal(np.frombuffer(d[18:], np.uint8).reshape(5, 7, 3)[..., ::-1], img)
    for bad, what in (("a.hdr", "float"), ("a.xyz", "extension"), ("noext", "extension")):       # .hdr takes float pixels (test_image_writers.py)
        with pytest.raises(capi.MiptError, match=what):
            host_write(tmp_path / bad, img)
        assert not (tmp_path / bad).exists()           # never another format under that name


def _jpeg_segments(data):
    """(marker, start, end) of the marker segments in front of the first scan."""
    pos, out = 2, []
    while pos + 4 <= len(data) and data[pos] == 0xff:
        m = data[pos + 1]
        n = struct.unpack(">H", data[pos + 2:pos + 4])[0]
        out.append((m, pos, pos + 2 + n))
        if m == 0xda:
            break
        pos += 2 + n
    return out


def test_malformed_jpeg_is_refused_not_trusted(tmp_path):
    """Crafted files: a scan naming a Huffman table no DHT defined, a DC table whose symbols are categories above 15,
    sampling factors that do not divide the largest one (3 under 4).  Each is an error, never a read out of bounds."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz"))
    done = set()
    for n in range(int(g["count"])):
        data = g[f"file{n}"].tobytes()
        segs = _jpeg_segments(data)
        # (1) no DHT at all
        stripped = bytearray(data[:2])
        for m, a, b in segs:
            if m != 0xc4:
                stripped += data[a:b]
        stripped += data[segs[-1][2]:]
        p = tmp_path / f"nodht{n}.jpg"
        p.write_bytes(bytes(stripped))
        with pytest.raises(capi.MiptError, match="never defined"):
            host_read(p)
        done.add("nodht")
        # (2) every symbol of the DC tables becomes category 31
        evil = bytearray(data)
        for m, a, b in segs:
            if m != 0xc4:
                continue
            q = a + 4
            while q < b:
                tc, cnt = evil[q] >> 4, sum(evil[q + 1:q + 17])
                if tc == 0:
                    for k in range(q + 17, q + 17 + cnt):
                        evil[k] = 31
                    done.add("dc31")
                q += 17 + cnt
        p = tmp_path / f"dc31_{n}.jpg"
        p.write_bytes(bytes(evil))
        with pytest.raises(capi.MiptError, match="huffman"):
            host_read(p)
        # (3) 3-component frames: component 0 sampled 4x1, component 1 sampled 3x1
        for m, a, b in segs:
            if m in (0xc0, 0xc2) and data[a + 9] == 3:
                bad = bytearray(data)
                bad[a + 11] = 0x41
                bad[a + 14] = 0x31
                p = tmp_path / f"samp{n}.jpg"
                p.write_bytes(bytes(bad))
                with pytest.raises(capi.MiptError, match="divide"):
                    host_read(p)
                done.add("samp")
    assert done == {"nodht", "dc31", "samp"}
