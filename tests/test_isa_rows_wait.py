"""The second step engine waits for its gather in two parts (csrc/bs_grow_spec.hip, rows_wait): the registers of the
part waited for later are written by the hardware AFTER the inline-asm statement that names them, so the compiler must
not touch them in between -- an earlier cut of the loop made it copy them where two paths met (a fault on the GPU).
This compiles the file for gfx950 (no GPU needed, about a minute) and walks the generated ISA."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_instruction_touches_the_rows_before_their_wait():
    spec = importlib.util.spec_from_file_location("check_rows_wait", os.path.join(ROOT, "tools", "check_rows_wait.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    checked, bad = mod.check(mod.device_asm())
    assert checked >= 4  # hot loop + complete step, for k <= 16 and k <= 32
    assert not bad, "\n".join(bad)
