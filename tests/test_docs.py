"""The documents the round-5 verdict asked for keep their promises: DESIGN.md describes what ships in at most 400 lines, and every evidence file that
profiles/REJECTED.md (the index of built-measured-rejected variants) or DESIGN.md cites exists in the tree."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cited_profile_files(text):
    out = set()
    for m in re.finditer(r"`((?:profiles/)?r0\d_[A-Za-z0-9_{},.*]+)`", text):
        name = m.group(1)
        name = name[len("profiles/"):] if name.startswith("profiles/") else name
        # brace / star patterns as the documents write them: r06_trace_lab_split4_{atrium,bust}.json, r04_vote_sim_*.txt
        alts = [name]
        b = re.search(r"\{([^}]*)\}", name)
        if b:
            alts = [name[:b.start()] + a + name[b.end():] for a in b.group(1).split(",")]
        out.update(alts)
    return out


def test_design_is_short_and_points_at_the_notebook_and_the_index():
    lines = open(os.path.join(ROOT, "DESIGN.md")).read().split("\n")
    assert len(lines) <= 400, len(lines)
    text = "\n".join(lines)
    assert "profiles/REJECTED.md" in text and "profiles/NOTEBOOK_r1_r5.md" in text
    assert os.path.exists(os.path.join(ROOT, "profiles", "NOTEBOOK_r1_r5.md"))


def test_every_cited_evidence_file_exists():
    missing = []
    for doc in ("DESIGN.md", os.path.join("profiles", "REJECTED.md")):
        for name in cited_profile_files(open(os.path.join(ROOT, doc)).read()):
            if not glob.glob(os.path.join(ROOT, "profiles", name)):
                missing.append((doc, name))
    assert not missing, missing
