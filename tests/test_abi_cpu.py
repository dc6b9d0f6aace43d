"""CPU: the C-ABI shared library loads without a GPU, exports every symbol include/ada_hip.h declares, and the ctypes
mirror of struct ada_igemm_args has exactly the C layout (checked by compiling the header with gcc)."""
import ctypes
import os
import re
import subprocess

import pytest

import hip_ext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ada_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ada_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = hip_ext.load()
    names = _declared_functions()
    assert set(names) == set(hip_ext.EXPORTS), (names, hip_ext.EXPORTS)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ada_hip.h but not exported"
    assert lib.ada_abi_version() == hip_ext.ABI_VERSION
    assert lib.ada_operand_dtype() in (hip_ext.DT_F16, hip_ext.DT_BF16)
    assert lib.ada_last_error() is not None


def test_bf16_variant_library_exports_the_same_abi():
    path = hip_ext.library_path(bf16=True)
    if not os.path.exists(path):
        pytest.skip("bf16 measurement variant not built")
    lib = ctypes.CDLL(path)
    for n in hip_ext.EXPORTS:
        assert hasattr(lib, n)
    lib.ada_operand_dtype.restype = ctypes.c_int
    assert lib.ada_operand_dtype() == hip_ext.DT_BF16


def test_igemm_args_struct_layout_matches_c(tmp_path):
    fields = [f[0] for f in hip_ext.IgemmArgs._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){",
            'printf("%zu\\n", sizeof(ada_igemm_args));']
    prog += [f'printf("%zu\\n", offsetof(ada_igemm_args, {f}));' for f in fields]
    prog += ["return 0;}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(c)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert vals[0] == ctypes.sizeof(hip_ext.IgemmArgs)
    for f, off in zip(fields, vals[1:]):
        assert getattr(hip_ext.IgemmArgs, f).offset == off, f


def test_layernorm_args_struct_layout_matches_c(tmp_path):
    names = [f[0] for f in hip_ext.LayerNormArgs._fields_]
    cnames = ["in" if f == "in_" else f for f in names]
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){", 'printf("%zu\\n", sizeof(ada_layernorm_args));']
    prog += [f'printf("%zu\\n", offsetof(ada_layernorm_args, {f}));' for f in cnames]
    prog += ["return 0;}"]
    c = tmp_path / "layout_ln.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "layout_ln"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(c)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert vals[0] == ctypes.sizeof(hip_ext.LayerNormArgs)
    for f, off in zip(names, vals[1:]):
        assert getattr(hip_ext.LayerNormArgs, f).offset == off, f


def test_constants_match_header():
    src = open(HEADER).read()
    consts = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(ADA_[A-Z0-9_]+)\s+\(?(-?(?:0x)?[0-9A-Fa-f]+)\)?\s", src)}
    for name, val in consts.items():
        py = name[4:]
        if hasattr(hip_ext, py):
            assert getattr(hip_ext, py) == val, name


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(hip_ext, "_lib", None)
    with pytest.raises(hip_ext.HipExtError, match="no CPU fallback"):
        hip_ext.load(str(tmp_path / "nope.so"))
    hip_ext.load()  # restore
