"""TEST INFRASTRUCTURE: write a WFC3 calibration directory in the on-disk layout the REFERENCE's code opens -- derived
from that code, not from any data file (the real CALWF3 / aXe files are not in this container) -- with a FITS encoder
of its own, so that wayne_amd.calibration.CalibrationSet.from_directory is exercised on files that wayne_amd.fitsio did
not write and that do not follow its writer's habits.

What the reference opens (all under params._calb_dir, params.py:20-56):
  WFC3.IR.G141.flat.2.fits     flat cube: planes f0..f3 = the DATA of HDUs 0..3, WMIN / WMAX in the PRIMARY header
                               (grism.py:66-76);  WFC3.IR.G102.flat.2.fits likewise (:453-454)
  WFC3.IR.G141.sky.V1.0.fits   master sky = data of HDU 0 (grism.py:79-80, 411-423)
  WFC3.IR.G141.1st.sens.2.fits BINTABLE in HDU 1 with fields WAVELENGTH (angstrom), SENSITIVITY (grism.py:97-106)
  u4m1335mi_pfl.fits           pixel flat = data of HDU 1, 1024 x 1024, used as [5:-5, 5:-5] (detector.py:31, 200-209)
  u1k1727mi_lin.fits           c1..c4 = data of HDUs 1..4 (detector.py:56-67)
  <super-dark of the mode>     file name from the mode table (detector.py:69-100); for the read with NSAMP index n the
                               SCI frame is HDU -5 n and its error HDU -5 n + 1: 1 + 5 x 16 HDUs (SCI, ERR, DQ, SAMP,
                               TIME per read), LAST READ FIRST (detector.py:183-190)
STScI's conventions, as far as the reference depends on them: big-endian IEEE floats (BITPIX -32), no BSCALE / BZERO on
float images, int16 DQ / SAMP planes, 2880-byte blocks.  Deliberately varied here, because a reader must not depend on
them: the order of the optional cards, COMMENT / HISTORY / blank cards in between, EXTNAME / EXTVER values, D exponents,
a table with a third column and mixed 1E / 1D fields, and HDUs after the ones the reference reads.
"""
import os

import numpy as np

BLOCK = 2880


def _card(key, value=None, comment=""):
    if key in ("COMMENT", "HISTORY", ""):
        return ("%-8s%s" % (key, value or "")).ljust(80)[:80]
    if isinstance(value, bool):
        v = "%20s" % ("T" if value else "F")
    elif isinstance(value, int):
        v = "%20d" % value
    elif isinstance(value, float):
        v = "%20s" % ("%.10E" % value)
    elif isinstance(value, tuple) and value[0] == "raw":        # a value spelled by the caller (e.g. a D exponent)
        v = "%20s" % value[1]
    else:
        v = "'%-8s'" % str(value).replace("'", "''")
        v = "%-20s" % v
    s = "%-8s= %s" % (key, v)
    if comment:
        s += " / " + comment
    return s.ljust(80)[:80]


def _header_bytes(cards):
    text = "".join(_card(*c) for c in cards) + "END".ljust(80)
    text += " " * ((-len(text)) % BLOCK)
    return text.encode("ascii")


def _pad(b):
    return b + b"\x00" * ((-len(b)) % BLOCK)


_BITPIX = {"f4": -32, "f8": -64, "i2": 16, "i4": 32, "u1": 8}


def image_hdu(data, extra=(), primary=False, extname=None, extver=None, shuffle=0):
    """One image HDU (data None: header only).  `shuffle` rotates the optional cards: their order is not fixed."""
    cards = [("SIMPLE", True, "conforms to FITS standard")] if primary else [("XTENSION", "IMAGE", "Image extension")]
    if data is None:
        cards += [("BITPIX", 8, ""), ("NAXIS", 0, "")]
        payload = b""
    else:
        a = np.asarray(data)
        code = a.dtype.str[1:]
        cards += [("BITPIX", _BITPIX[code], ""), ("NAXIS", a.ndim, "")]
        cards += [("NAXIS%d" % (i + 1), int(n), "") for i, n in enumerate(a.shape[::-1])]
        payload = _pad(a.astype(">" + code).tobytes())
    if primary:
        cards.append(("EXTEND", True, "extensions may follow"))
    else:
        cards += [("PCOUNT", 0, ""), ("GCOUNT", 1, "")]
    opt = list(extra)
    if extname is not None:
        opt.append(("EXTNAME", extname, "extension name"))
    if extver is not None:
        opt.append(("EXTVER", extver, "extension version number"))
    opt += [("COMMENT", " written by tests/stsci_files.py, not by the product"), ("", ""), ("HISTORY", " layout after the reference's code")]
    if opt:
        k = shuffle % len(opt)
        opt = opt[k:] + opt[:k]
    return _header_bytes(cards + opt) + payload


def table_hdu(columns, extname="SENS", extra=()):
    """BINTABLE of (name, numpy array, TFORM letter, unit) columns of equal length."""
    n = len(columns[0][1])
    forms = {"E": ">f4", "D": ">f8", "J": ">i4"}
    dt = np.dtype([(c[0], forms[c[2]]) for c in columns])
    rec = np.zeros(n, dtype=dt)
    for name, arr, letter, unit in columns:
        rec[name] = arr
    cards = [("XTENSION", "BINTABLE", "binary table extension"), ("BITPIX", 8, ""), ("NAXIS", 2, ""),
             ("NAXIS1", dt.itemsize, "width of table in bytes"), ("NAXIS2", n, "number of rows"), ("PCOUNT", 0, ""),
             ("GCOUNT", 1, ""), ("TFIELDS", len(columns), "")]
    for i, (name, arr, letter, unit) in enumerate(columns, 1):
        cards += [("TTYPE%d" % i, name, ""), ("TFORM%d" % i, "1" + letter if i % 2 else letter, ""), ("TUNIT%d" % i, unit, "")]
    cards += [("EXTNAME", extname, "")] + list(extra)
    return _header_bytes(cards) + _pad(rec.tobytes())


def write_calibration_directory(path, src, detector, modes, grisms=("G141",)):
    """`src`: any object with .flat[g] (4, 1014, 1014), .flat_wl[g], .sky[g] (1014, 1014), .sens[g] = (wl_um, val),
    .pfl (1014, 1014: the light-sensitive part), .lin (4, 1024, 1024) and .super_dark_hdus(SUBARRAY, SAMPSEQ) -- e.g. a
    wayne_amd CalibrationSet.synthetic.  `modes`: [(SUBARRAY, SAMPSEQ), ...] whose super-darks are written."""
    from wayne_amd import calibration as C
    os.makedirs(path, exist_ok=True)
    rng = np.random.RandomState(5)
    for g in grisms:
        cube, (wmin, wmax) = src.flat[g], src.flat_wl[g]
        pieces = [image_hdu(cube[0], primary=True, shuffle=1,
                            extra=[("ORIGIN", "tests/stsci_files.py", ""), ("WMAX", float(wmax), "maximum wavelength (A)"),
                                   ("FILETYPE", "FLAT CUBE", ""), ("WMIN", ("raw", ("%.4E" % wmin).replace("E", "D")), "minimum wavelength (A)")])]
        for i in (1, 2, 3):
            pieces.append(image_hdu(cube[i], extname=None if i == 2 else "COEF%d" % i, extver=i if i != 1 else None, shuffle=i))
        pieces.append(image_hdu(np.zeros((3, 2), dtype=np.float32), extname="UNUSED"))          # a plane the reference never opens
        open(os.path.join(path, C.FLAT_FILES[g]), "wb").write(b"".join(pieces))
        open(os.path.join(path, C.SKY_FILES[g]), "wb").write(
            image_hdu(src.sky[g], primary=True, shuffle=2, extra=[("BUNIT", "ELECTRONS/S", ""), ("DATE", "2011-03-01", "")]))
        wl_um, val = src.sens[g]
        tbl = table_hdu([("WAVELENGTH", np.asarray(wl_um) * 1e4, "E", "ANGSTROM"),
                         ("SENSITIVITY", np.asarray(val), "D", "ELEC/S/(ERG/S/CM2/A)"),
                         ("ERROR", np.asarray(val) * 0.01, "E", "ELEC/S/(ERG/S/CM2/A)")])
        open(os.path.join(path, C.SENS_FILES[g]), "wb").write(image_hdu(None, primary=True, extra=[("NEXTEND", 1, "")]) + tbl)
    pfl_full = (1.0 + rng.normal(0, 0.01, (1024, 1024))).astype(np.float32)        # the border: reference pixels, never used
    pfl_full[5:-5, 5:-5] = src.pfl
    open(os.path.join(path, C.PFL_FILE), "wb").write(
        image_hdu(None, primary=True, extra=[("FILETYPE", "PIXEL-TO-PIXEL FLAT", ""), ("NEXTEND", 3, "")]) +
        image_hdu(pfl_full, extname="SCI", extver=1, shuffle=2) +
        image_hdu(np.full((1024, 1024), 0.001, dtype=np.float32), extname="ERR", extver=1) +
        image_hdu(np.zeros((1024, 1024), dtype=np.int16), extname="DQ", extver=1, shuffle=1))
    lin = [image_hdu(None, primary=True, extra=[("FILETYPE", "LINEARITY COEFFICIENTS", "")])]
    for i in range(4):
        lin.append(image_hdu(src.lin[i], extname="COEF", extver=i + 1, shuffle=i))
    lin.append(image_hdu(np.zeros((1024, 1024), dtype=np.float32), extname="ERR", extver=1))   # (the real file goes on: ERR, DQ, ...)
    open(os.path.join(path, C.LIN_FILE), "wb").write(b"".join(lin))
    for subarray, sampseq in modes:
        hdus = src.super_dark_hdus(subarray, sampseq, detector)
        S = min(subarray + 10, 1024)
        pieces = [image_hdu(None, primary=True, extra=[("SAMP_SEQ", sampseq, ""), ("SUBTYPE", "SQ%dSUB" % subarray if subarray < 1024 else "FULLIMAG", ""),
                                                        ("NEXTEND", len(hdus) - 1, "")])]
        n_reads = (len(hdus) - 1) // 5
        for r in range(n_reads):
            sci, err = hdus[1 + 5 * r], hdus[2 + 5 * r]
            ver = r + 1
            pieces.append(image_hdu(np.asarray(sci, dtype=np.float32), extname="SCI", extver=ver, shuffle=r,
                                    extra=[("SAMPNUM", n_reads - 1 - r, ""), ("BUNIT", "COUNTS", "")]))
            pieces.append(image_hdu(np.asarray(err, dtype=np.float32), extname="ERR", extver=ver))
            pieces.append(image_hdu(np.zeros((S, S), dtype=np.int16), extname="DQ", extver=ver, shuffle=1))
            pieces.append(image_hdu(np.ones((S, S), dtype=np.int16), extname="SAMP", extver=ver))
            pieces.append(image_hdu(np.full((S, S), 1.0, dtype=np.float32), extname="TIME", extver=ver, shuffle=2))
        open(os.path.join(path, detector.dark_file(subarray, sampseq)), "wb").write(b"".join(pieces))
    return path
