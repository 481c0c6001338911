"""DESIGN.md's "Numbers" block is generated from profiles/ (tools/design_numbers.py): the committed block must be what
the committed files give."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_numbers_block_matches_profiles():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
