"""DESIGN.md states the current design only (verdict r4, weak #11: the 88 KB lab diary is LABLOG.md now); its step-form table and the
library's per-form issue-slot figures (qoa_wave_stats.issue_slots, what callers price a batch with) must not drift apart."""
import os
import re

from conftest import ROOT


def test_design_is_the_current_state_and_short():
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert len(design.splitlines()) <= 300
    assert os.path.exists(os.path.join(ROOT, "LABLOG.md"))
    for heading in ("## 1. The path and its boundary", "## 3. Data layout in HBM", "### K1 ", "### K2u ", "## 5. Measurement", "## 6. Multi-GPU"):
        assert heading in design, heading


def test_step_form_table_matches_the_library():
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    capi = open(os.path.join(ROOT, "qoaudio_amd", "csrc", "qoa_capi.cpp")).read()
    m = re.search(r"kFormIssueSlots\[QOA_WAVE_FORMS\] = \{([^}]*)\}", capi)
    slots = [int(v) for v in m.group(1).split(",")]
    rows = {r.split("|")[1].strip(): r.split("|")[3].strip() for r in design.splitlines() if r.startswith("| ") and r.count("|") >= 7}
    want = {"cheap-quotient speculative": slots[0], "exact-quotient speculative": slots[1], "penalty": slots[2], "wide": slots[3],
            "full packed": slots[4], "32-bit block statement": slots[5]}
    for form, n in want.items():
        assert form in rows and rows[form].startswith(str(n)), (form, n, rows.get(form))
    hdr = open(os.path.join(ROOT, "include", "qoaudio_amd.h")).read()
    assert f"{slots[5]} x 4.06 x 1.10" in hdr and f"{slots[0]} x 4.06 x 1.10" in hdr
