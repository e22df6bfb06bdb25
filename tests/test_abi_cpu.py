"""CPU suite: the C-ABI library is built, loads, exports every symbol include/*.h
declare (sslam_frontend.h: the drop-in boundary; sslam_testing.h: test entry points), and refuses to run without a GPU (no CPU fallback)."""
import ctypes, os, re
import pytest
import pkg

ROOT = pkg.ROOT


def _declared_functions(header):
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(sslam_[a-z0-9_]+)\s*\(", txt)))


def _exported(lib_path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    return sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("sslam_"))


def test_library_exports_exactly_the_declared_symbols():
    """libsslam_frontend.so exports the entry points of include/sslam_frontend.h and nothing else (no self-tests, probes or stand-ins: those live in
    libsslam_frontend_testing.so = the same sources with -DSSLAM_TESTING, which exports the boundary plus include/sslam_testing.h)."""
    lib_path = pkg.builder().build(force=False, verbose=False)
    boundary = _declared_functions("sslam_frontend.h")
    testing = _declared_functions("sslam_testing.h")
    assert len(boundary) >= 80 and len(testing) >= 8 and not set(boundary) & set(testing)
    assert _exported(lib_path) == boundary
    assert _exported(pkg.builder().TEST_LIB) == sorted(boundary + testing)
    L = ctypes.CDLL(lib_path)
    assert L.sslam_abi_version() == 1


def test_struct_layouts_match_opencv():
    fe = pkg.frontend()
    assert fe.KP_DTYPE.itemsize == 28          # cv::KeyPoint
    assert fe.KL_DTYPE.itemsize == 68          # cv::line_descriptor::KeyLine
    assert fe.KL_DTYPE.names[:3] == ("angle", "class_id", "octave") and fe.KL_DTYPE.names[-1] == "numOfPixels"


def test_parameter_structs_match_the_header(tmp_path):
    """the ctypes mirrors of the C ABI's parameter structs (sslam_batch_match, sslam_frontend_params, sslam_record_header) have the layout the
    C compiler gives the header's declarations"""
    import subprocess
    fe = pkg.frontend()
    src = tmp_path / "layout.c"
    src.write_text("""#include <stdio.h>
#include <stddef.h>
#include "sslam_frontend.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sslam_batch_match), offsetof(sslam_batch_match, nnratio), offsetof(sslam_batch_match, bounds),
           offsetof(sslam_batch_match, line_gate_scale), offsetof(sslam_batch_match, line_ratio_mode), offsetof(sslam_batch_match, init_matches12),
           offsetof(sslam_batch_match, line_npairs), sizeof(sslam_frontend_params), sizeof(sslam_record_header));
    return 0;
}
""")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    B = fe.BatchMatch
    want = [ctypes.sizeof(B), B.nnratio.offset, B.bounds.offset, B.line_gate_scale.offset, B.line_ratio_mode.offset, B.init_matches12.offset, B.line_npairs.offset,
            ctypes.sizeof(fe.FrontendParams), 16]
    assert got == want, (got, want)


def test_fails_loudly_without_gpu():
    fe = pkg.frontend()
    try:
        ctx = fe.Context(0)
    except fe.SslamError as e:
        assert "no CPU fallback" in str(e)
        return
    ctx.close()
    pytest.skip("GPU present")


def test_graft_entry_build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    m.build()


def test_documents_name_only_declared_entry_points():
    """every sslam_* name in the documents is declared in include/sslam_frontend.h (prefixes like `sslam_vocab_*` and the C++ shim namespace aside)"""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set(re.findall(r"\b(sslam_[a-z0-9_]+)\b", open(os.path.join(root, "include", "sslam_frontend.h")).read() + open(os.path.join(root, "include", "sslam_testing.h")).read()))
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md", os.path.join("profiles", "README.md"), os.path.join("tools", "README.md")):
        names = set(re.findall(r"\b(sslam_[a-z0-9_]+)\b", open(os.path.join(root, doc)).read()))
        unknown = sorted(n for n in names if n not in declared and not n.startswith("sslam_shim") and not n.endswith("_") and n not in ("sslam_frontend", "sslam_testing"))      # (the two header names)
        assert not unknown, (doc, unknown)
