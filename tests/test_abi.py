"""CPU tests: the C-ABI library loads and exports every symbol include/pinot_gpu.h declares (no compute calls)."""
import os
import re

from pinot_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pinot_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(pg_[a-z_0-9]+)\s*\(", text))


def test_binding_covers_every_declared_symbol():
    assert _declared_symbols() == {name for name, _, _ in _abi.ABI_SYMBOLS}


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_abi.GPU_LIB_PATH), "run __graft_entry__.build() first"
    lib = _abi.load_gpu_library()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.pg_version()


def test_host_library_exports_every_symbol_its_header_declares():
    """include/pinot_host_c.h: the C entry points of the C++ host mirror (libpinot_host.so)."""
    import ctypes as C
    text = open(os.path.join(ROOT, "include", "pinot_host_c.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(ph_[a-z_0-9]+)\s*\(", text))
    assert len(declared) >= 25
    lib = C.CDLL(_abi.HOST_LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    # and nothing the library exports is missing from the header
    import subprocess
    exported = set(re.findall(r" T (ph_[a-z_0-9]+)", subprocess.check_output(["nm", "-D", "--defined-only", _abi.HOST_LIB_PATH]).decode()))
    assert exported == declared, exported ^ declared


def test_library_embeds_gfx950_code_object():
    data = open(_abi.GPU_LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"scan_agg_kernel" in data


def test_calls_fail_loudly_without_init_or_gpu():
    import ctypes as C
    lib = _abi.load_gpu_library()
    handle = C.c_void_p()
    desc = _abi.pg_segment_desc()
    st = lib.pg_segment_open(C.byref(desc), C.byref(handle))
    assert st in (_abi.PG_ERR_NOT_INITIALIZED, _abi.PG_ERR_DEVICE, _abi.PG_OK)
    if st != _abi.PG_OK:
        assert lib.pg_last_error()


def test_struct_layouts_match_the_header(tmp_path):
    """Compile a C probe against include/pinot_gpu.h and compare sizeof / offsetof with the ctypes mirror."""
    import ctypes as C
    import subprocess
    structs = ["pg_config", "pg_column_desc", "pg_segment_desc", "pg_predicate", "pg_filter_node", "pg_aggregation",
               "pg_query", "pg_agg_value", "pg_stats", "pg_result"]
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "pinot_gpu.h"', "int main(void) {"]
    for sname in structs:
        cls = getattr(_abi, sname)
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (sname, sname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (sname, fname, sname, fname))
    lines += ["return 0; }"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for sname in structs:
        cls = getattr(_abi, sname)
        assert int(out[sname]) == C.sizeof(cls), sname
        for fname, _ in cls._fields_:
            assert int(out["%s.%s" % (sname, fname)]) == getattr(cls, fname).offset, (sname, fname)
