"""CPU: DESIGN.md is a design document, and it cannot drift from the measurements.

It is generated (scripts/make_design.py) from scripts/design_template.md and the JSON files of the round's collection under
profiles/<round>/: every number is a placeholder with one source file.  Here the template is rendered again and compared
with the committed file; the document is held to 400 lines and to the current state only (no bracketed figures of
earlier rounds: those live in HISTORY.md)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import make_design  # noqa: E402


def test_design_md_is_the_rendered_template():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"profiles/(r\d\d)/\*\.json", text.splitlines()[0])
    assert m, "DESIGN.md does not start with the generator's stamp"
    assert text == make_design.render(m.group(1)), "DESIGN.md is stale: run scripts/make_design.py"


def test_design_md_is_short_and_current():
    lines = open(os.path.join(ROOT, "DESIGN.md")).read().splitlines()
    assert len(lines) <= 400, len(lines)
    body = "\n".join(lines)
    assert not re.search(r"\(r0\d[:)]", body) and "round 4's figures in brackets" not in body
    assert "{{" not in body and "}}" not in body                        # every placeholder was filled
    for section in ("## 1. The path and its boundary", "## 3. Data layout in HBM", "## 4. Kernels", "## 5. Random numbers",
                    "## 6. Oracle and parity", "## 7. Measurement", "## 8. Multi-GPU"):
        assert section in body, section
    # the history is where the document says it is
    assert os.path.exists(os.path.join(ROOT, "HISTORY.md"))


def test_template_numbers_are_placeholders():
    # a number with a unit of measurement in the template's kernel table and measurement table must come from a file:
    # the two tables hold no literal exposures/s or microsecond figures
    t = open(os.path.join(ROOT, "scripts", "design_template.md")).read()
    table = t[t.index("| **`value`**"):t.index("`roofline` = algorithmic bytes")]
    rows = [r for r in table.splitlines() if r.startswith("|")]
    assert len(rows) >= 8
    for row in rows:
        assert re.search(r"\{\{b\.[a-z_0-9.]+(\*[0-9.e-]+)?\|[^}]*\}\}", row), row[:80]
        assert not re.search(r"\|\s*\**\d{3,}", re.sub(r"\{\{[^}]*\}\}", "", row)), row[:80]      # no literal rate beside them
