"""The reference's own test/test-structs.c -- compiled where it lies, with the reference's headers, linked against
the product (oracle/build_ref_tests.sh) -- must run through its assertions: binary-level check that the product's
data model is a drop-in for container.c / frame.c / buffer.h users.  Built only where /root/reference exists; the
binary travels with the tree (oracle/_ref/), so the test also runs where the reference is absent."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_test_structs")


def test_reference_test_structs_passes_against_the_product():
    if not os.path.exists(EXE):
        if not os.path.isdir("/root/reference"):
            pytest.skip("oracle/_ref/ref_test_structs not built and no reference tree here")
        subprocess.run(["bash", os.path.join(ROOT, "oracle", "build_ref_tests.sh")], check=True)
    r = subprocess.run([EXE], cwd="/tmp", capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
