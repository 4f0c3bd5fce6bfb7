"""CPU: the C-ABI library builds, loads, exports every symbol include/ovvc_hip.h declares, and
fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import re
from pathlib import Path

import pytest

from openvvc_amd import capi

ROOT = Path(__file__).resolve().parent.parent


def test_exports_match_header(built_lib):
    hdr = (ROOT / "include" / "ovvc_hip.h").read_text()
    declared = set(re.findall(r"\b(ovhip_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(built_lib, name), f"{name} declared in include/ovvc_hip.h but not exported"
    assert declared == set(capi.EXPORTED_SYMBOLS)


def test_struct_sizes():
    assert C.sizeof(capi.TbCmd) == 32 and C.sizeof(capi.McUnit) == 32
    assert C.sizeof(capi.Pic) == 40


def test_no_cpu_fallback(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    r = built_lib.ovhip_ctx_create(C.byref(h), 0, None)
    assert r < 0 and not h.value, "engine must refuse to run without a HIP device"
    from openvvc_amd import engine
    with pytest.raises(engine.EngineError):
        engine.Context(0)


def test_bench_and_entry_scripts_parse():
    """bench.py / __graft_entry__.py compile, and bench.py's command line is the driver's contract (no GPU needed for --help)."""
    import py_compile, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for f in ("bench.py", "__graft_entry__.py"):
        py_compile.compile(str(root / f), doraise=True)
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--in-flight"):
        assert flag in out.stdout
