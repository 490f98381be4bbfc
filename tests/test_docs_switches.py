"""Host-side hygiene: every STEMGNN_* environment switch the sources read is documented, and nothing documented is stale."""
import pathlib
import re

ROOT = pathlib.Path(__file__).resolve().parents[1]
NAME = re.compile(r"STEMGNN_[A-Z0-9_]+")


def _names(paths, pattern):
    found = set()
    for path in paths:
        for hit in pattern.findall(path.read_text(errors="ignore")):
            found.update(NAME.findall(hit))
    return found


def test_environment_switches_are_documented():
    sources = [p for p in (ROOT / "stemgnn_amd").rglob("*") if p.suffix in (".py", ".hip", ".h")]
    sources += [ROOT / "bench.py", ROOT / "__graft_entry__.py"]
    read = _names(sources, re.compile(r'(?:getenv\(|environ(?:\.get)?[\(\[])\s*"STEMGNN_[A-Z0-9_]+"'))
    docs = _names([ROOT / "DESIGN.md", ROOT / "INTEGRATION.md", ROOT / "README.md"], NAME)
    assert read, "no switches found: the scan is broken"
    assert not (read - docs), f"undocumented switches: {sorted(read - docs)}"
    assert not (docs - read), f"documented but not read anywhere: {sorted(docs - read)}"
