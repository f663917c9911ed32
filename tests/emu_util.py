"""Test helper: builds (once) and binds the CPU kernel-logic emulator build of the C ABI
(tools/hipemu/libleopard_amd_emu.so).  Only tests import this; leopard_amd never does."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(REPO, "tools", "hipemu", "libleopard_amd_emu.so")


def emu_ops():
    from leopard_amd import _lib
    from leopard_amd.ops import Ops
    srcs = [os.path.join(REPO, "leopard_amd", "csrc", f) for f in os.listdir(os.path.join(REPO, "leopard_amd", "csrc"))]
    srcs += [os.path.join(REPO, "tools", "hipemu", f) for f in ("hipemu.cpp", "hipemu.h")]
    srcs.append(os.path.join(REPO, "include", "leopard_amd.h"))
    stale = (not os.path.exists(EMU)) or any(os.path.getmtime(s) > os.path.getmtime(EMU) for s in srcs)
    if stale:
        r = subprocess.run(["make", "-C", REPO, "emu"], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("emulator build unavailable: " + r.stderr[-400:])
    return Ops(lib=_lib.bind(EMU), emulated=True)
