"""One table of environment knobs (csrc/tor_knobs.hpp): every getenv of the library goes through
tor::knob(), every name it is asked for is in the table, and KNOBS.md says what the table says.  CPU only."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "trace-of-radiance_amd", "csrc")


def _sources():
    for pat in ("*.cpp", "*.hpp", "*.hip", "kernel/*.inc", "kernel/*.hpp"):
        for p in sorted(glob.glob(os.path.join(CSRC, pat))):
            yield p, open(p).read()


def test_every_getenv_goes_through_the_table(tor):
    table = {k["name"] for k in tor.knobs()}
    assert 20 <= len(table) == len(tor.knobs()) <= 28      # (round 5: pruned from 41)
    used = set()
    for path, text in _sources():
        if os.path.basename(path) == "tor_knobs.hpp":
            continue
        assert "getenv" not in text, f"{path}: getenv outside csrc/tor_knobs.hpp"
        used |= set(re.findall(r'(?:knob|env_ms)\("(TOR_[A-Z0-9_]+)"', text))
    assert used, "no knob() call found"
    assert used <= table, f"knobs read by the sources but missing from the table: {sorted(used - table)}"
    assert table <= used, f"knobs in the table that nothing reads: {sorted(table - used)}"
    for k in tor.knobs():
        assert k["when"] in ("call", "context", "upload") and k["what"] and k["default"] and k["range"], k


def test_knobs_md_is_generated_from_the_table():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_knob_doc.py"), "--check"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr


def test_an_unknown_name_reads_as_unset(tor, monkeypatch):
    # (the drop-in reads its defaults through the same table: a malformed value of a known knob is an error that names it)
    monkeypatch.setenv("TOR_DEFAULT_SEEDING", "nonsense")
    import pytest
    with pytest.raises(tor.TorError) as e:
        tor.render(tor.new_canvas(4, 4, 1), tor.camera(), tor.random_scene(0xFACADE).list(), 5)
    assert "TOR_DEFAULT_SEEDING" in str(e.value)
