"""Every Python entry point of the repository byte-compiles, and bench.py's command line parses: a syntax slip in a script that only runs on
the GPU box (bench.py, tools/) must fail HERE, in the CPU suite, not at the end of a round."""
import glob
import os
import py_compile
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_script_byte_compiles(tmp_path):
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for pat in ("tools/*.py", "frizbee_amd/*.py", "oracle/*.py", "tests/*.py"):
        files += sorted(glob.glob(os.path.join(ROOT, pat)))
    assert len(files) > 20
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


def test_bench_command_line_parses():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout
