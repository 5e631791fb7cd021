"""The Go side of the boundary (go/, cmd/) cannot be compiled in this image (no Go toolchain).  tools/check_go.py checks,
statically, what a first compile would trip over: every C.bsg_* call against include/bloomgpu.h (name, arity, constants, struct
fields, layout of the Go mirrors), balanced delimiters and used imports, every reference identifier the overlay relies on against
/root/reference, and that go/overlay/engine_gpu.patch applies there with hooks that exist in gpu_engine.go AND its stub."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_go  # noqa: E402


def test_go_side_is_consistent_with_the_header_and_the_reference(capsys):
    rc = check_go.main(["--reference", "/root/reference"])
    out = capsys.readouterr().out
    assert rc == 0, out
    if os.path.isdir("/root/reference"):
        assert "engine_gpu.patch applies" in out


def test_the_checker_catches_what_it_claims_to(tmp_path):
    header = check_go.parse_header(os.path.join(ROOT, "include", "bloomgpu.h"))
    assert header[0]["bsg_query"] == 11 and header[0]["bsg_device_count"] == 0 and "BSG_OP_TERM" in header[1]
    assert [f[0] for f in header[2]["bsg_filter_desc"]] == ["word_off", "m", "k", "reserved"]
    bad = tmp_path / "bad.go"
    bad.write_text('package x\n/*\n#include "bloomgpu.h"\n*/\nimport "C"\nfunc f() {\n'
                   '\tC.bsg_sync(nil, 1)            // arity\n\tC.bsg_no_such(nil)           // name\n\t_ = C.BSG_NOPE                // constant\n'
                   '\tvar st C.bsg_ingest_stats\n\t_ = st.no_field              // field\n\t_ = "C.bsg_sync(1,2,3) in a string is ignored"\n}\n')
    problems = []
    check_go.check_cgo([str(bad)], header, problems)
    text = "\n".join(problems)
    assert "bsg_sync called with 2 arguments, the header declares 1" in text
    assert "C.bsg_no_such is not declared" in text and "C.BSG_NOPE is not defined" in text and "no such field" in text
    assert text.count("bsg_sync") == 1
    unbalanced = tmp_path / "u.go"
    unbalanced.write_text('package x\nimport "fmt"\nfunc f() { if true { }\n')
    problems = []
    check_go.check_syntax([str(unbalanced)], problems)
    assert any("never closed" in p for p in problems) and any("never used" in p for p in problems)
