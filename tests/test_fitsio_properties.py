"""Property tests of the FITS writer / reader pair (wayne_amd/fitsio.py) -- the data format either side of the path
(the reference writes its exposures with astropy.io.fits, exposure.py:133-214, and reads its
calibration files with it).  Round trips of arbitrary headers and images; the structure rules of the standard the
reference's files obey (2880-byte blocks, 80-character ASCII cards, big-endian payloads)."""
import os
import string

import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

from wayne_amd import fitsio

KEY = st.text(alphabet=string.ascii_uppercase + string.digits + "_-", min_size=1, max_size=8).filter(
    lambda k: k not in ("END", "SIMPLE", "XTENSION", "BITPIX", "EXTEND", "PCOUNT", "GCOUNT", "COMMENT", "HISTORY", "EXTNAME",
                        "BSCALE", "BZERO") and not k.startswith("NAXIS"))
TEXT = st.text(alphabet=string.ascii_letters + string.digits + " '/=.,:+-_()", max_size=40).map(lambda s: s.rstrip())
VALUE = st.one_of(st.booleans(), st.integers(min_value=-2 ** 62, max_value=2 ** 62),
                  st.floats(allow_nan=False, allow_infinity=False, width=64), TEXT)
COMMENT = st.text(alphabet=string.ascii_letters + " ", max_size=20).map(lambda s: s.strip())
CARDS = st.lists(st.tuples(KEY, VALUE, COMMENT), max_size=12, unique_by=lambda c: c[0])
DTYPE = st.sampled_from(["u1", "i2", "i4", "i8", "f4", "f8"])
SHAPE = st.lists(st.integers(min_value=1, max_value=9), min_size=1, max_size=3).map(tuple)


@st.composite
def image(draw):
    dt, shape = draw(DTYPE), draw(SHAPE)
    n = int(np.prod(shape))
    seed = draw(st.integers(min_value=0, max_value=2 ** 31 - 1))
    rs = np.random.RandomState(seed)
    if dt[0] == "f":
        a = (rs.standard_normal(n) * 10.0 ** rs.randint(-30, 30)).astype(dt)
    else:
        info = np.iinfo(dt)
        a = rs.randint(info.min, info.max, n, dtype=np.int64 if dt != "u1" else np.int64).astype(dt)
    return a.reshape(shape)


@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(primary_cards=CARDS, ext=st.lists(st.tuples(CARDS, st.one_of(st.none(), image())), max_size=4))
def test_files_round_trip_and_obey_the_block_structure(tmp_path, primary_cards, ext):
    path = os.path.join(str(tmp_path), "t.fits")
    hdus = [fitsio.HDU(fitsio.Header(primary_cards), None)]
    hdus += [fitsio.HDU(fitsio.Header(c), a, name="SCI") for c, a in ext]
    fitsio.write(path, hdus)
    raw = open(path, "rb").read()
    assert len(raw) % fitsio.BLOCK == 0 and raw.startswith(b"SIMPLE  =                    T")
    back = fitsio.read(path)
    assert len(back) == len(hdus)
    for h0, h1 in zip(hdus, back):
        for k, v, _ in h0.header.cards:
            got = h1.header[k]
            if isinstance(v, bool):
                assert got is v, (k, v, got)
            elif isinstance(v, float):
                assert isinstance(got, (int, float)) and float(got) == v, (k, v, got)     # (1e22 is written "1E+22")
            else:
                assert got == v and type(got) is type(v), (k, v, got)
        if h0.data is None:
            assert h1.data is None
        else:
            assert h1.data.shape == h0.data.shape and h1.data.dtype.kind == h0.data.dtype.kind
            assert h1.data.dtype.itemsize == h0.data.dtype.itemsize and h1.data.dtype.byteorder in (">", "|")
            np.testing.assert_array_equal(h1.data, h0.data)
    # the header area is ASCII cards of 80 characters, the last one of each header "END"
    pos = 0
    for h in back:
        end = raw.index(b"END" + b" " * 77, pos)
        assert (end - pos) % 80 == 0 and all(32 <= c < 127 for c in raw[pos:end])
        pos = (end + 80 + fitsio.BLOCK - 1) // fitsio.BLOCK * fitsio.BLOCK
        size = fitsio._data_size(h.header)
        pos += (size + fitsio.BLOCK - 1) // fitsio.BLOCK * fitsio.BLOCK
    assert pos == len(raw)


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(cards=CARDS, a=image())
def test_cached_header_blocks_equal_the_rendered_ones(tmp_path, cards, a):
    # Exposure.generate_fits writes pre-rendered header blocks + the payload in one writev (fitsio.cached_header_block /
    # write_pieces): the same bytes as the HDU-object writer
    p1, p2 = os.path.join(str(tmp_path), "a.fits"), os.path.join(str(tmp_path), "b.fits")
    fitsio.write(p1, [fitsio.HDU(fitsio.Header(cards), None), fitsio.HDU(fitsio.Header([]), a, name="SCI")])
    code = a.dtype.kind + str(a.dtype.itemsize)
    be = np.ascontiguousarray(a, dtype=a.dtype.newbyteorder(">"))
    pad = (-be.nbytes) % fitsio.BLOCK
    pieces = [fitsio.cached_header_block(object(), cards, primary=True),
              fitsio.cached_header_block(object(), [], data_shape=a.shape, dtype_code=code, name="SCI"),
              memoryview(be.reshape(-1)).cast("B"), b"\0" * pad]
    fitsio.write_pieces(p2, pieces)
    assert open(p1, "rb").read() == open(p2, "rb").read()
