"""The measurement tools are part of what the profiles and logs under profiles/ were made with: they must at least compile, and every one of
them must be listed in tools/README.md (a tool nobody can find is a tool nobody re-runs)."""
import glob
import os
import py_compile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tools_compile(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) >= 8
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


def test_every_tool_is_listed_in_the_tools_readme():
    readme = open(os.path.join(ROOT, "tools", "README.md")).read()
    missing = []
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.sh")) + glob.glob(os.path.join(ROOT, "tools", "ubench", "*.hip"))):
        name = os.path.basename(f)
        if name.startswith("_"):
            continue  # __init__.py, scratch scripts
        if name not in readme:
            missing.append(name)
    assert not missing, missing
