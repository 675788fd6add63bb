"""CPU: libhavatar_hip.so loads and exports every symbol include/havatar.h declares (no compute without a GPU),
and the product package never touches the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "havatar.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hav_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_all_declared_symbols():
    from havatar_amd import _lib, build
    build.build()
    names = _declared()
    assert len(names) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/havatar.h but not exported"
    L = _lib.lib()
    assert L.hav_abi_version() == _lib.ABI_VERSION
    assert L.hav_mlp_blob_bytes() > 4 * (176 * 128 + 128 * 128 + 68 * 128)


def test_out_size_helper_matches_reference_formula():
    from havatar_amd import _lib
    L = _lib.lib()
    oh, ow = ctypes.c_int(), ctypes.c_int()
    for (h, w, kh, kw, ux, uy, dx, dy, p) in [(128, 128, 4, 4, 1, 1, 1, 1, (2, 2, 2, 2)), (64, 48, 4, 4, 2, 2, 1, 1, (2, 1, 2, 1)),
                                               (513, 513, 4, 4, 1, 1, 2, 2, (1, 1, 1, 1)), (10, 13, 5, 3, 3, 2, 2, 3, (2, 3, 1, 4))]:
        assert L.hav_upfirdn2d_out_size(h, w, kh, kw, ux, uy, dx, dy, *p, ctypes.byref(oh), ctypes.byref(ow)) == 0
        assert oh.value == (h * uy + p[2] + p[3] - kh + dy) // dy
        assert ow.value == (w * ux + p[0] + p[1] - kw + dx) // dx
    assert L.hav_upfirdn2d_out_size(4, 4, 9, 9, 1, 1, 1, 1, 0, 0, 0, 0, ctypes.byref(oh), ctypes.byref(ow)) == -1


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors (product binding and the oracle's) against the C compiler's view of include/havatar.h:
    sizeof and the offset of every field."""
    import subprocess
    from havatar_amd import _lib
    from oracle import oracle
    structs = {"HavRenderParams": (_lib.HavRenderParams, oracle.HavRenderParams), "HavMlpWeights": (_lib.HavMlpWeights, oracle.HavMlpWeights),
               "HavRenderOut": (_lib.HavRenderOut,), "HavFieldParams": (_lib.HavFieldParams,)}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "havatar.h"', 'int main(void) {']
    for name, mirrors in structs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (name, name))
        for f, _ in mirrors[0]._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    seen = 0
    for ln in subprocess.check_output([str(exe)]).decode().split("\n"):
        if not ln:
            continue
        name, field, val = ln.split()
        for m in structs[name]:
            got = ctypes.sizeof(m) if field == "sizeof" else getattr(m, field).offset
            assert got == int(val), (name, field, m.__module__, got, int(val))
            seen += 1
    assert seen > 60
    assert ctypes.sizeof(_lib.HavRenderParams) == 160          # ABI 6: + grid_blocks, reserved0


def test_product_never_imports_oracle():
    """havatar_amd/ must not reference oracle/ (a product path through the oracle voids every parity claim)."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "havatar_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"(from|import)\s+oracle|hav_oracle|libhav_oracle|orc_", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch):
    from havatar_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libhavatar_hip.so")
    with pytest.raises(_lib.HavatarLibraryError):
        _lib.lib()


def test_cpu_tensors_are_refused_by_native_modules():
    import torch
    from havatar_amd.native import fused, upfirdn2d
    x = torch.zeros(1, 2, 4, 4)
    with pytest.raises(RuntimeError):
        fused.fused_bias_act(x, x.new_empty(0), x.new_empty(0), 3, 0, 0.2, 1.0)
    with pytest.raises(RuntimeError):
        upfirdn2d.upfirdn2d(x.reshape(2, 4, 4, 1), torch.ones(2, 2), 1, 1, 1, 1, 0, 0, 0, 0)


def test_build_info_records_the_validated_compiler():
    """havatar_amd/build.py records the hipcc identity next to the library it builds; another compiler is a warning and `tested: false`
    (an error only with HAVATAR_REQUIRE_TESTED_HIPCC=1), and an existing .so is never rebuilt or refused because of the installed
    compiler or a missing BUILD_INFO.json."""
    import warnings
    from havatar_amd import build as hb
    hb.build()
    info = hb.build_info()
    assert info is not None, "python -m havatar_amd.build writes lib/BUILD_INFO.json"
    assert tuple(info["hipcc"]) == hb.TESTED_HIPCC and info["tested"] is True
    import pytest as _pt
    fake = os.path.join(os.path.dirname(hb.BUILD_INFO), "_fake_hipcc.sh")
    with open(fake, "w") as f:
        f.write("#!/bin/sh\necho 'HIP version: 9.9.0'\necho 'AMD clang version 99'\n")
    os.chmod(fake, 0o755)
    saved = os.environ.get("HIPCC")
    try:
        os.environ.pop("HAVATAR_REQUIRE_TESTED_HIPCC", None)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ver, tested = hb.check_compiler(fake)
        assert ver[0] == "HIP version: 9.9.0" and tested is False and any("validated" in str(x.message) for x in w)
        os.environ["HAVATAR_REQUIRE_TESTED_HIPCC"] = "1"
        with _pt.raises(RuntimeError, match="differs from the compiler"):
            hb.check_compiler(fake)
        os.environ.pop("HAVATAR_REQUIRE_TESTED_HIPCC", None)
        # an up-to-date library is returned as it is: no compiler check, no rebuild, BUILD_INFO.json or not
        os.environ["HIPCC"] = fake
        moved = hb.BUILD_INFO + ".moved"
        os.rename(hb.BUILD_INFO, moved)
        try:
            assert hb.build() == hb.LIB
        finally:
            os.rename(moved, hb.BUILD_INFO)
    finally:
        os.environ.pop("HAVATAR_REQUIRE_TESTED_HIPCC", None)
        if saved is None:
            os.environ.pop("HIPCC", None)
        else:
            os.environ["HIPCC"] = saved
        os.remove(fake)
