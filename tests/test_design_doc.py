"""DESIGN.md is the CURRENT design -- what ships, why, the measured figure and the profiles/ file it comes from: at most 400 lines of at
most 120 columns, and every profiles/ file it cites exists.  The history lives in HISTORY.md."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _design():
    return open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()


def test_design_is_short_enough_to_read():
    lines = _design().split("\n")
    assert len(lines) <= 400, len(lines)
    too_long = [(i + 1, len(l)) for i, l in enumerate(lines) if len(l) > 120]
    assert not too_long, too_long
    assert os.path.exists(os.path.join(ROOT, "HISTORY.md"))


def test_every_profile_design_cites_exists():
    text = _design()
    cited = set(re.findall(r"profiles/(r0\d_[A-Za-z0-9_.]+?\.(?:txt|jsonl|json|csv))", text))
    cited |= {m for m in re.findall(r"`(r0\d_[A-Za-z0-9_]+\.(?:txt|jsonl|json|csv))`", text)}
    assert len(cited) >= 10, cited
    missing = sorted(c for c in cited if not os.path.exists(os.path.join(ROOT, "profiles", c)))
    assert not missing, f"DESIGN.md cites profiles that do not exist: {missing}"


def test_readme_table_has_one_line_per_row():
    rows = [l for l in open(os.path.join(ROOT, "README.md"), encoding="utf-8").read().split("\n") if l.startswith("|")]
    assert rows and max(len(l) for l in rows) <= 200, max(len(l) for l in rows)
