"""CPU: scripts/mutation_audit.py plants its defects by exact text; a source edit that moves one of those texts would turn
the audit into an error at its next run.  Every site must occur exactly once in the current sources, every mutant must
change something, and names are unique."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_every_mutation_site_occurs_exactly_once():
    import mutation_audit as m
    seen = set()
    groups = ((m.MUTANTS, os.path.join(ROOT, "wayne_amd", "csrc")), (m.PY_MUTANTS, ROOT), (m.CPU_MUTANTS, ROOT))
    n = 0
    for mutants, base in groups:
        for mu in mutants:
            assert mu["name"] not in seen, mu["name"]
            seen.add(mu["name"])
            assert mu["what"] and mu["stage"] and mu["edits"]
            for rel, old, new in mu["edits"]:
                text = open(os.path.join(base, rel)).read()
                assert text.count(old) == 1, "%s: %r occurs %d times in %s" % (mu["name"], old, text.count(old), rel)
                assert old != new
                n += 1
    assert n >= 60
