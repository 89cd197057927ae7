"""A small FITS reader / writer (numpy only; astropy is not available here).

Covers what the exposure path touches:
  * reading the calibration files the reference opens with astropy.io.fits --
    primary / IMAGE HDUs of any BITPIX (flat cube grism.py:68-76, sky :417,
    pixel flat detector.py:202-203, linearity :58-62, super-darks :187-190,
    initial bias exposure_generator.py:457-458) and the BINTABLE of the grism
    sensitivity (grism.py:99-106);
  * writing the HST-style multi-extension files that Exposure.generate_fits
    produces (exposure.py:133-214): a header-only primary HDU followed by
    IMAGE extensions.

FITS is big-endian, 2880-byte blocks, 80-character header cards
(FITS Standard 4.0, sections 3-4 and 7).
"""
import numpy as np

BLOCK = 2880
_BITPIX_DTYPE = {8: ">u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}
_DTYPE_BITPIX = {"u1": 8, "i2": 16, "i4": 32, "i8": 64, "f4": -32, "f8": -64}
# BINTABLE TFORM letter -> (numpy type, bytes)
_TFORM = {"L": ("u1", 1), "B": ("u1", 1), "I": (">i2", 2), "J": (">i4", 4), "K": (">i8", 8),
          "E": (">f4", 4), "D": (">f8", 8), "A": ("S1", 1)}


class Header(object):
    """Ordered FITS header: list of (key, value, comment) cards."""

    def __init__(self, cards=None):
        self.cards = list(cards) if cards else []

    def __contains__(self, key):
        return any(k == key for k, _, _ in self.cards)

    def __getitem__(self, key):
        for k, v, _ in self.cards:
            if k == key:
                return v
        raise KeyError(key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __setitem__(self, key, value):
        comment = ""
        if isinstance(value, tuple):
            value, comment = value
        for i, (k, _, c) in enumerate(self.cards):
            if k == key:
                self.cards[i] = (key, value, comment or c)
                return
        self.cards.append((key, value, comment))

    def keys(self):
        return [k for k, _, _ in self.cards]

    def items(self):
        return [(k, v) for k, v, _ in self.cards]


def _parse_value(text):
    t = text.strip()
    if not t:
        return None
    if t[0] == "'":
        # string: closing quote is the last single quote not doubled
        end = 1
        out = []
        while end < len(t):
            if t[end] == "'":
                if end + 1 < len(t) and t[end + 1] == "'":
                    out.append("'")
                    end += 2
                    continue
                break
            out.append(t[end])
            end += 1
        return "".join(out).rstrip()
    t = t.split("/")[0].strip()
    if t == "T":
        return True
    if t == "F":
        return False
    try:
        return int(t)
    except ValueError:
        pass
    try:
        return float(t.replace("D", "E").replace("d", "e"))
    except ValueError:
        return t


def _read_header(buf, pos):
    cards = []
    while True:
        block = buf[pos:pos + BLOCK]
        if len(block) < BLOCK:
            raise ValueError("truncated FITS header")
        pos += BLOCK
        done = False
        for i in range(0, BLOCK, 80):
            card = block[i:i + 80].decode("ascii", "replace")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if card[8:10] == "= ":
                body = card[10:]
                value = _parse_value(body)
                comment = ""
                if "/" in body and not body.strip().startswith("'"):
                    comment = body.split("/", 1)[1].strip()
                cards.append((key, value, comment))
            elif key in ("COMMENT", "HISTORY"):
                cards.append((key, card[8:].rstrip(), ""))
        if done:
            return Header(cards), pos


class HDU(object):
    def __init__(self, header, data=None, name=None):
        self.header = header
        self.data = data
        self.name = name if name is not None else header.get("EXTNAME", "")


def _data_size(h):
    naxis = h.get("NAXIS", 0)
    if naxis == 0:
        return 0
    n = 1
    for i in range(1, naxis + 1):
        n *= h["NAXIS%d" % i]
    return abs(h["BITPIX"]) // 8 * h.get("GCOUNT", 1) * (h.get("PCOUNT", 0) + n)


def _table(h, raw):
    nrows, rowlen = h["NAXIS2"], h["NAXIS1"]
    fields, off = [], 0
    for i in range(1, h["TFIELDS"] + 1):
        form = str(h["TFORM%d" % i]).strip()
        digits = "".join(ch for ch in form if ch.isdigit())
        rep = int(digits) if digits else 1
        letter = form[len(digits)]
        if letter not in _TFORM:
            raise ValueError("unsupported TFORM %r" % form)
        dt, size = _TFORM[letter]
        name = str(h.get("TTYPE%d" % i, "col%d" % i)).strip()
        if letter == "A":
            fields.append((name, "S%d" % rep, off))
            off += rep
        else:
            fields.append((name, dt if rep == 1 else (dt, (rep,)), off))
            off += size * rep
    dtype = np.dtype({"names": [f[0] for f in fields], "formats": [f[1] for f in fields],
                      "offsets": [f[2] for f in fields], "itemsize": rowlen})
    return np.frombuffer(raw[:nrows * rowlen], dtype=dtype)


def read(path):
    """Read every HDU of a FITS file -> list of HDU (images scaled by BSCALE/BZERO)."""
    with open(path, "rb") as f:
        buf = f.read()
    hdus, pos = [], 0
    while pos < len(buf):
        if not buf[pos:pos + BLOCK].strip(b"\x00 "):
            break
        h, pos = _read_header(buf, pos)
        size = _data_size(h)
        raw = buf[pos:pos + size]
        pos += (size + BLOCK - 1) // BLOCK * BLOCK
        data = None
        xt = str(h.get("XTENSION", "IMAGE")).strip()
        if size:
            if xt == "BINTABLE":
                data = _table(h, raw)
            elif xt in ("IMAGE", ""):
                shape = tuple(h["NAXIS%d" % i] for i in range(h["NAXIS"], 0, -1))
                data = np.frombuffer(raw, dtype=_BITPIX_DTYPE[h["BITPIX"]]).reshape(shape)
                bscale, bzero = h.get("BSCALE", 1), h.get("BZERO", 0)
                if bscale != 1 or bzero != 0:
                    data = data * bscale + bzero
        hdus.append(HDU(h, data))
    return hdus


def scan(path):
    """The structure of a FITS file without its data: [(header, data bytes), ...] by reading the header blocks and
    seeking over the payloads, or None when the file is not a complete FITS file (a header that does not end, a payload
    the file is too short for, trailing bytes).  Cheap -- a few kilobytes read per HDU -- which is what a restarted
    visit needs to decide whether an exposure's file is whole (wayne_amd/observation.py, `resume`)."""
    import os
    try:
        total = os.path.getsize(path)
        out, pos = [], 0
        with open(path, "rb") as f:
            while pos < total:
                buf = b""
                while True:                                  # a header: blocks up to the one that holds END
                    block = f.read(BLOCK)
                    if len(block) < BLOCK:
                        return None
                    buf += block
                    if any(block[i:i + 8] == b"END     " for i in range(0, BLOCK, 80)):
                        break
                    if len(buf) > 64 * BLOCK:
                        return None
                h, used = _read_header(buf, 0)
                size = _data_size(h)
                padded = (size + BLOCK - 1) // BLOCK * BLOCK
                pos += used + padded
                if pos > total:
                    return None
                out.append((h, size))
                f.seek(pos)
        return out if pos == total and out else None
    except (OSError, KeyError, ValueError):
        return None


def _fmt_value(v):
    if isinstance(v, bool):
        return "%20s" % ("T" if v else "F")
    if isinstance(v, (int, np.integer)):
        return "%20d" % v
    if isinstance(v, (float, np.floating)):
        s = repr(float(v)).upper()
        if "E" not in s and "." not in s and "N" not in s:
            s += "."
        return "%20s" % s
    s = str(v).replace("'", "''")
    return "'%-8s'" % s[:66]


_CARD_CACHE = {}


def _card(key, value, comment=""):
    """One 80-character card.  A visit writes the same few hundred cards into every file, so rendered
    cards are memoised (keyed with the value's type: True == 1 == 1.0 as dictionary keys)."""
    try:
        k = (key, type(value), value, comment)
        hit = _CARD_CACHE.get(k)
    except TypeError:            # unhashable value
        return _render_card(key, value, comment)
    if hit is None:
        if len(_CARD_CACHE) > 20000:
            _CARD_CACHE.clear()
        hit = _CARD_CACHE[k] = _render_card(key, value, comment)
    return hit


def _render_card(key, value, comment=""):
    if key == "":                      # blank-keyword card: free text (section titles of the HST headers)
        return (" " * 8 + str(value))[:80].ljust(80)
    if key in ("COMMENT", "HISTORY"):
        return ("%-8s%s" % (key, value))[:80].ljust(80)
    body = "%-8s= %s" % (key[:8], _fmt_value(value))
    if comment:
        body += " / " + comment
    return body[:80].ljust(80)


def _header_bytes(cards):
    text = "".join(_card(*c) for c in cards) + "END".ljust(80)
    pad = (-len(text)) % BLOCK
    return (text + " " * pad).encode("ascii")


def _image_hdu_parts(data, extra_cards, primary, name=None):
    """[header bytes, payload (a big-endian array's buffer, or b""), zero padding] of one image HDU."""
    cards = [("SIMPLE", True, "conforms to FITS standard")] if primary else [("XTENSION", "IMAGE", "Image extension")]
    if data is None:
        cards += [("BITPIX", 8, ""), ("NAXIS", 0, "")]
        payload = b""
        nbytes = 0
    else:
        a = np.asarray(data)
        code = a.dtype.kind + str(a.dtype.itemsize)
        if code not in _DTYPE_BITPIX:
            a = a.astype(np.float64)
            code = "f8"
        cards += [("BITPIX", _DTYPE_BITPIX[code], ""), ("NAXIS", a.ndim, "")]
        for i, n in enumerate(a.shape[::-1], 1):
            cards.append(("NAXIS%d" % i, n, ""))
        # (the byte-swapped copy is written from its own buffer: no second copy into a bytes object, no third into
        # header + payload + padding)
        swapped = np.ascontiguousarray(a, dtype=a.dtype.newbyteorder(">"))
        payload = memoryview(swapped.reshape(-1)).cast("B") if swapped.size else b""
        nbytes = swapped.nbytes
    if primary:
        cards.append(("EXTEND", True, ""))
    else:
        cards += [("PCOUNT", 0, ""), ("GCOUNT", 1, "")]
        if name:
            cards.append(("EXTNAME", name, "extension name"))
    reserved = {"SIMPLE", "XTENSION", "BITPIX", "NAXIS", "EXTEND", "PCOUNT", "GCOUNT", "END"}
    for k, v, c in extra_cards:
        if k in reserved or k.startswith("NAXIS") or (k == "EXTNAME" and name):
            continue
        cards.append((k, v, c))
    return [_header_bytes(cards), payload, b"\x00" * ((-nbytes) % BLOCK)]


def _write_all(fd, pieces):
    """Write the byte pieces to the file descriptor with os.writev (one system call for a whole file, the interpreter
    lock released while it runs), going on after a short write -- a signal, a quota, a payload beyond 2 GiB -- until
    every byte is out.  (A raw FileIO.write may return early and Python does not retry it: a silently truncated file.)"""
    import os
    bufs = [memoryview(p_).cast("B") for p_ in pieces if len(p_)]
    IOV = 512
    while bufs:
        n = os.writev(fd, bufs[:IOV])
        while n > 0:
            if n >= len(bufs[0]):
                n -= len(bufs[0])
                bufs.pop(0)
            else:
                bufs[0] = bufs[0][n:]
                n = 0
        while bufs and not len(bufs[0]):
            bufs.pop(0)


PART_SUFFIX = ".part"


def write_pieces(path, pieces):
    """Write pre-rendered pieces (header blocks, big-endian payloads, padding) as one file.  The bytes go to
    `path + ".part"` and the finished file is renamed over `path` (atomic on one file system): a reader -- or a visit
    restarted after a crash -- never sees a file under its final name that is not whole.  An existing file is replaced,
    as the reference's remove + writeto does (exposure.py:211-213)."""
    import os
    part = path + PART_SUFFIX
    fd = os.open(part, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
    try:
        _write_all(fd, pieces)
    finally:
        os.close(fd)
    os.replace(part, path)


def write(path, hdus):
    """Write [HDU, ...]; the first becomes the primary HDU.  An existing file is replaced."""
    pieces = []
    for i, h in enumerate(hdus):
        cards = h.header.cards if isinstance(h.header, Header) else list(h.header or [])
        pieces += _image_hdu_parts(h.data, cards, primary=(i == 0), name=h.name or None)
    write_pieces(path, pieces)


_BLOCK_CACHE = {}


def cached_header_block(key, cards, data_shape=None, dtype_code="f8", name=None, primary=False):
    """The header bytes of an image HDU whose cards are a pure function of `key` (memoised): the five extensions of a
    read carry the same few cards in every file of a visit -- rendering them per file kept the writer threads holding
    the interpreter lock for a millisecond per 266 x 266 exposure.  `data_shape` None: a data-less HDU."""
    hit = _BLOCK_CACHE.get(key)
    if hit is None:
        if len(_BLOCK_CACHE) > 4096:
            _BLOCK_CACHE.clear()
        data = None if data_shape is None else np.empty((0,) * 0 + tuple(data_shape), dtype=dtype_code)[...]
        hit = _BLOCK_CACHE[key] = _image_hdu_parts(data, cards, primary=primary, name=name)[0]
    return hit
