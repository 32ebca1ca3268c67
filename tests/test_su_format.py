"""Row f1: the SU reader against the FORMAT, not against its own writer.  The bytes below are laid out by hand
from the SEG-Y / SU trace-header definition (240-byte header; 1-based byte positions: tracl 1-4, fldr 9-12,
scalco 71-72, sx 73-76, sy 77-80, gx 81-84, gy 85-88, ns 115-116, dt 117-118 in microseconds; then ns IEEE
float32 samples), in both byte orders - DENISE writes native (little-endian) SU, which the reference re-links per
shot (models/networks.py:7669-7692)."""
import struct

import numpy as np
import pytest


def _trace(endian, tracl, fldr, sx, gx, ns, dt_us, samples, scalco=-100):
    h = bytearray(240)
    h[0:4] = struct.pack(endian + "i", tracl)
    h[8:12] = struct.pack(endian + "i", fldr)
    h[70:72] = struct.pack(endian + "h", scalco)
    h[72:76] = struct.pack(endian + "i", sx)
    h[76:80] = struct.pack(endian + "i", 4000)
    h[80:84] = struct.pack(endian + "i", gx)
    h[84:88] = struct.pack(endian + "i", 46000)
    h[114:116] = struct.pack(endian + "H", ns)
    h[116:118] = struct.pack(endian + "H", dt_us)
    body = b"".join(struct.pack(endian + "f", float(v)) for v in samples)
    return bytes(h) + body


@pytest.mark.parametrize("endian", ["<", ">"])
def test_read_su_follows_the_trace_header_layout(tmp_path, endian):
    from physicsbasedfwi2_amd.compat.pyapi_denise import read_su
    ns, dt_us = 2500, 2000                      # the reference's 5.0 s at 2 ms (networks.py:7331)
    rng = np.random.default_rng(0)
    data = rng.standard_normal((3, ns)).astype(np.float32)
    raw = b"".join(_trace(endian, k + 1, 7, 38000 + 8000 * k, 38000 + 2000 * k, ns, dt_us, data[k]) for k in range(3))
    assert len(raw) == 3 * (240 + 4 * ns)
    p = tmp_path / "DENISE_MARMOUSI_y.su.shot1"
    p.write_bytes(raw)
    out, dt, hdr = read_su(str(p), headers=True)                 # byte order found from the file length
    assert out.dtype == np.float32 and out.shape == (3, ns) and np.array_equal(out, data)
    assert dt == pytest.approx(0.002)
    assert list(hdr["tracl"]) == [1, 2, 3] and list(hdr["fldr"]) == [7, 7, 7] and list(hdr["scalco"]) == [-100] * 3
    assert list(hdr["sx"]) == [38000, 46000, 54000] and list(hdr["gx"]) == [38000, 40000, 42000]
    assert list(hdr["sy"]) == [4000] * 3 and list(hdr["gy"]) == [46000] * 3
    out2, dt2 = read_su(str(p), endian=endian)
    assert np.array_equal(out2, data) and dt2 == dt
    other = ">" if endian == "<" else "<"
    with pytest.raises(Exception):
        read_su(str(p), endian=other)                            # 2500 byte-swapped is 50185 samples: not this file


def test_read_su_rejects_truncated_and_ragged_files(tmp_path):
    from physicsbasedfwi2_amd.compat.pyapi_denise import read_su, write_su
    a = np.arange(40, dtype=np.float32).reshape(4, 10)
    raw = b"".join(_trace("<", k + 1, 1, 0, 0, 10, 4000, a[k]) for k in range(4))
    p = tmp_path / "t.su"
    p.write_bytes(raw[:-6])
    with pytest.raises(Exception):
        read_su(str(p))
    p.write_bytes(raw[:100])
    with pytest.raises(Exception):
        read_su(str(p))
    ragged = _trace("<", 1, 1, 0, 0, 10, 4000, a[0]) + _trace("<", 2, 1, 0, 0, 5, 4000, a[1][:5]) + bytes(20)
    assert len(ragged) % 280 == 0
    p.write_bytes(ragged)
    with pytest.raises(Exception):
        read_su(str(p))
    # the writer produces what the reader (and the layout above) expects
    write_su(str(p), a, 0.004)
    out, dt = read_su(str(p))
    assert np.array_equal(out, a) and dt == pytest.approx(0.004)
    b = p.read_bytes()
    assert struct.unpack_from("<H", b, 114)[0] == 10 and struct.unpack_from("<H", b, 116)[0] == 4000
    assert struct.unpack_from("<i", b, 280)[0] == 2              # tracl of the second trace
