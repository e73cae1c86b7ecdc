"""CPU: generated sources are current, and the device's log10 restatement agrees with the host libm.

The hand-scheduled loops (chain_loop_gfx950.inc, feed_loop_gfx950.inc, wlod_loop_gfx950.inc) and the glibc log table
(glibc_log_data.inc) are committed generator output: regenerating them must be a no-op, so that what is
reviewed in tools/gen_*.py is what runs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "garlic_amd", "csrc")


@pytest.mark.parametrize("tool,inc", [("gen_chain_asm.py", "chain_loop_gfx950.inc"),
                                      ("gen_feed_asm.py", "feed_loop_gfx950.inc"),
                                      ("gen_wlod_asm.py", "wlod_loop_gfx950.inc"),
                                      ("gen_log_data.py", "glibc_log_data.inc")])
def test_regenerating_is_a_no_op(tmp_path, tool, inc):
    env = {k: v for k, v in os.environ.items() if not k.startswith("GARLIC_")}   # no experiment hooks
    env["GARLIC_GEN_OUT"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], capture_output=True, text=True, env=env)
    if tool == "gen_log_data.py" and r.returncode != 0 and "AssertionError" in r.stderr:
        pytest.skip("this host's libm.so.6 is not the glibc 2.35 build the table was read from")
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(tmp_path / inc).read() == open(os.path.join(CSRC, inc)).read(), f"{inc} is stale: run tools/{tool}"


def test_device_log10_restatement_matches_this_hosts_libm(tmp_path):
    """tests/host_unit/log10_unit.cpp: garlic_amd/csrc/tgls_math.hpp (what the TGLS kernels run) compiled
    for the host, against the host's log10 and the reference's lod() over specials, every branch border,
    every table cell and a few million random arguments.  Built without -mfma: fma() is then libm's,
    correctly rounded like the instruction."""
    exe = str(tmp_path / "log10_unit")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                         os.path.join(ROOT, "tests", "host_unit", "log10_unit.cpp"), "-lm"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    r = subprocess.run([exe, "400000"], capture_output=True, text=True)
    if r.returncode != 0 and "mismatches" in r.stdout:
        pytest.skip("the host's log10 is not glibc 2.35's FMA variant: the library falls back to host-computed terms here\n"
                    + r.stderr[-500:])
    assert r.returncode == 0 and "log10_unit ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
