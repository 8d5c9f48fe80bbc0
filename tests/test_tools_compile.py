"""Every developer tool and benchmark script at least parses (they run on the GPU box only, so nothing else here would notice
a syntax error in them), and none of them reads the reference tree at run time (/root/reference does not exist on the GPU box)."""
import py_compile
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SCRIPTS = sorted(p for d in ("tools", "tools/probes", "benchmarks") for p in (ROOT / d).glob("*.py")) + [ROOT / "bench.py", ROOT / "__graft_entry__.py"]


@pytest.mark.parametrize("path", SCRIPTS, ids=lambda p: str(p.relative_to(ROOT)))
def test_script_parses_and_stays_off_the_reference_tree(path, tmp_path):
    py_compile.compile(str(path), cfile=str(tmp_path / "out.pyc"), doraise=True)
    assert "/root/reference" not in path.read_text()


def test_shell_probes_only_call_scripts_that_exist():
    import re
    for sh in sorted((ROOT / "tools").glob("*.sh")) + sorted((ROOT / "tools" / "probes").glob("*.sh")):
        for rel in re.findall(r"python3? +((?:tools|benchmarks)/[\w/]+\.py|bench\.py)", sh.read_text()):
            assert (ROOT / rel).exists(), (sh.name, rel)


def test_every_profile_of_this_round_is_indexed():
    """profiles/README.md names the command behind every committed r05_* summary (brace / star patterns allowed)."""
    import re
    text = (ROOT / "profiles" / "README.md").read_text()
    patterns = [re.escape(p).replace(r"\*", ".*").replace(r"\{fetch,write,dram\}", "(fetch|write|dram)")
                for p in re.findall(r"`([^`]*r05_[^`]*)`", text)]
    missing = [f.name for f in sorted((ROOT / "profiles").glob("r05_*"))
               if not any(re.fullmatch(p, f.name) or re.fullmatch(p, f.stem) for p in patterns)]
    assert not missing, missing
