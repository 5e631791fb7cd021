#!/usr/bin/env python3
"""Static checks of the Go side (go/, cmd/) that need no Go toolchain — there is none in the image the library is built in, so
these ~1 900 lines of Go have never met a compiler; this lowers the chance that the first Go box spends its time on typos.

    python tools/check_go.py [--reference /path/to/bloomsearch]

 1. every C.bsg_* call names a function include/bloomgpu.h declares, with the header's number of arguments; every C.BSG_*
    constant and every C.bsg_* type exists; fields read from C structs exist; the Go mirrors of bsg_filter_desc / bsg_term have
    the header's layout;
 2. delimiters balance in every Go file (outside strings, runes and comments) and every file has a package clause;
 3. every identifier the overlay uses that it does not declare itself is declared somewhere in the reference's non-test sources,
    in the cgo binding, or is a Go builtin / standard-library name from a short list (catches a misremembered newBloomEntrySets);
 4. go/overlay/engine_gpu.patch applies cleanly to the reference (patch --dry-run), and the names its hooks call exist in
    BOTH gpu_engine.go and gpu_engine_stub.go with the same parameter counts.
Checks 3 and 4 need the reference checkout (default /root/reference; skipped with a note when absent).
Exit status 0 = clean; every finding is printed.
"""
from __future__ import annotations

import argparse
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GO_KEYWORDS = set("break default func interface select case defer go map struct chan else goto package switch const fallthrough if range "
                  "type continue for import return var".split())
GO_PREDECLARED = set("bool byte complex64 complex128 error float32 float64 int int8 int16 int32 int64 rune string uint uint8 uint16 uint32 "
                     "uint64 uintptr any comparable true false iota nil append cap clear close complex copy delete imag len make max min new "
                     "panic print println real recover".split())
# standard-library / third-party selector names the overlay legitimately uses (methods and fields reached through a dot)
STD_NAMES = set("""
    Error Errorf Sprintf Printf Println Fprintf New Is As Join Unwrap Lock Unlock RLock RUnlock Add Done Wait Warn Debug Info Since Now Sub
    Duration Nanosecond Microsecond Millisecond Second Seconds Nanoseconds Milliseconds Search SearchInts Slice Ints IntsAreSorted Strings Split TrimSpace
    Atoi Itoa ParseInt ParseUint Getenv Setenv ReadFile WriteFile Open Create Close Read Write Seek ReadFull SeekStart ReadSeeker ReadSeekCloser Reader Writer Mutex RWMutex
    WaitGroup Pool Get Put Logger Handler DiscardHandler Context Background WithCancel WithTimeout Err TODO Fatal Fatalf Skip Skipf Helper Run Logf Log
    Errorf Name Cleanup TempDir Setenv B N T TB ResetTimer StopTimer StartTimer ReportMetric ReportAllocs Loop Pointer Sizeof Slice SliceData String StringData
    BloomFilter EstimateParameters FromWithM NewWithEstimates AddString TestString Cap K BitSet Bytes Equal WriteTo ReadFrom NumCPU GOMAXPROCS
    Marshal Unmarshal NewDecoder NewEncoder Decode Encode Scanner NewScanner Scan Text Buffer Bytes Len Reset Grow NewReader NewWriter Flush
    Intn Int63 Uint64 Uint32 Float64 Seed NewSource Rand Perm Shuffle Exit Args Stderr Stdout Stdin Arg Parse Int Bool Float64Var IntVar StringVar Var
    Contains HasPrefix HasSuffix Join Repeat Fields ToLower Index Builder WriteString WriteByte Sort Stable Sum Checksum MakeTable Castagnoli LittleEndian BigEndian
    PutUint32 PutUint64 Uint16 Count Fatal Errorf MaxInt MaxInt64 MaxUint32 Inf Ceil Log Pow Floor IsNaN
""".split())


def strip_go(src: str) -> str:
    """Go source with comments, strings and rune literals blanked out (same length, newlines kept)."""
    out = []
    i, n = 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join("\n" if ch == "\n" else " " for ch in src[i:j]))
            i = j
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('"' + " " * (j - i - 1) + '"')
            i = j + 1
        elif c == "`":
            j = src.find("`", i + 1)
            j = n if j < 0 else j
            out.append("`" + "".join("\n" if ch == "\n" else " " for ch in src[i + 1:j]) + "`")
            i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            out.append("'" + " " * (j - i - 1) + "'")
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def split_args(s: str):
    """Top-level comma split of an argument list (without the outer parentheses)."""
    args, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    last = "".join(cur).strip()
    if last:
        args.append(last)
    return args


def matching_paren(s: str, open_at: int) -> int:
    depth = 0
    for j in range(open_at, len(s)):
        if s[j] == "(":
            depth += 1
        elif s[j] == ")":
            depth -= 1
            if depth == 0:
                return j
    return -1


def parse_header(path: str):
    src = open(path).read()
    src_nc = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)), src, flags=re.S)
    funcs = {}
    for m in re.finditer(r"BSG_API\s+[\w\s\*]+?\b(bsg_\w+)\s*\(", src_nc):
        close = matching_paren(src_nc, m.end() - 1)
        args = split_args(src_nc[m.end():close])
        funcs[m.group(1)] = 0 if args == ["void"] else len(args)
    consts = set(re.findall(r"#define\s+(BSG_\w+)", src_nc))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src_nc, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fm = re.match(r"([\w\s]+?)\s*\*?\s*(\w+)\s*(\[\s*(\d+)\s*\])?$", decl)
            if fm:
                fields.append((fm.group(2), fm.group(1).strip(), int(fm.group(4)) if fm.group(4) else 1))
        structs[m.group(3)] = fields
    opaque = set(re.findall(r"typedef\s+struct\s+\w+\s+(\w+)\s*;", src_nc))
    return funcs, consts, structs, opaque


C_SIZES = {"uint64_t": 8, "int64_t": 8, "uint32_t": 4, "int32_t": 4, "double": 8, "float": 4, "uint8_t": 1}
GO_SIZES = {"uint64": 8, "int64": 8, "uint32": 4, "int32": 4, "float64": 8, "float32": 4, "uint8": 1, "byte": 1}


def check_cgo(files, header, problems):
    funcs, consts, structs, opaque = header
    for path in files:
        src = strip_go(open(path).read())
        rel = os.path.relpath(path, ROOT)
        if "bloomgpu_lab.h" in open(path).read():      # the lab switches are not the contract: a host binding includes bloomgpu.h only
            problems.append("%s: includes bloomgpu_lab.h (lab switches are not part of the drop-in contract)" % rel)
        for m in re.finditer(r"\bC\.(bsg_\w+)\s*\(", src):
            name = m.group(1)
            line = src.count("\n", 0, m.start()) + 1
            if name in structs or name in opaque:
                continue                                   # a conversion C.bsg_x(...)
            if name not in funcs:
                problems.append("%s:%d: C.%s is not declared in include/bloomgpu.h" % (rel, line, name))
                continue
            close = matching_paren(src, m.end() - 1)
            n = len(split_args(src[m.end():close]))
            if n != funcs[name]:
                problems.append("%s:%d: C.%s called with %d arguments, the header declares %d" % (rel, line, name, n, funcs[name]))
        for m in re.finditer(r"\bC\.(BSG_\w+)", src):
            if m.group(1) not in consts:
                problems.append("%s:%d: C.%s is not defined in include/bloomgpu.h" % (rel, src.count("\n", 0, m.start()) + 1, m.group(1)))
        for m in re.finditer(r"\bC\.(bsg_\w+)\b(?!\s*\()", src):
            if m.group(1) not in structs and m.group(1) not in opaque and m.group(1) not in funcs:
                problems.append("%s:%d: C.%s is not a type of include/bloomgpu.h" % (rel, src.count("\n", 0, m.start()) + 1, m.group(1)))
        # fields of C struct variables:  var st C.bsg_ingest_stats ... st.n_rows
        for m in re.finditer(r"\bvar\s+(\w+)\s+C\.(bsg_\w+)", src):
            var, typ = m.group(1), m.group(2)
            if typ not in structs:
                continue
            names = {f[0] for f in structs[typ]}
            for fm in re.finditer(r"\b%s\.(\w+)" % re.escape(var), src):
                if fm.group(1) not in names:
                    problems.append("%s:%d: %s.%s — bsg struct %s has no such field" % (rel, src.count("\n", 0, fm.start()) + 1, var, fm.group(1), typ))
    # the Go mirrors that are cast to C structs by pointer must have the header's layout
    binding = open(os.path.join(ROOT, "go", "bloomgpu", "bloomgpu.go")).read()
    for go_name, c_name in (("FilterDesc", "bsg_filter_desc"), ("Term", "bsg_term")):
        m = re.search(r"type\s+%s\s+struct\s*\{(.*?)\n\}" % go_name, binding, flags=re.S)
        if not m:
            problems.append("go/bloomgpu/bloomgpu.go: type %s not found" % go_name)
            continue
        go_sizes = []
        for line in strip_go(m.group(1)).splitlines():
            fm = re.match(r"\s*([\w, ]+?)\s+(\[(\d+)\])?(\w+)\s*$", line)
            if fm and fm.group(4) in GO_SIZES:
                go_sizes += [GO_SIZES[fm.group(4)] * (int(fm.group(3)) if fm.group(3) else 1)] * len(fm.group(1).split(","))
        c_sizes = [C_SIZES[t] * cnt for (_, t, cnt) in structs[c_name]]
        if go_sizes != c_sizes:
            problems.append("go/bloomgpu/bloomgpu.go: %s field sizes %s differ from %s's %s" % (go_name, go_sizes, c_name, c_sizes))


def check_syntax(files, problems):
    pairs = {")": "(", "]": "[", "}": "{"}
    for path in files:
        raw = open(path).read()
        src = strip_go(raw)
        rel = os.path.relpath(path, ROOT)
        if not re.search(r"^package\s+\w+", src, flags=re.M):
            problems.append("%s: no package clause" % rel)
        stack = []
        for i, ch in enumerate(src):
            if ch in "([{":
                stack.append((ch, i))
            elif ch in ")]}":
                if not stack or stack[-1][0] != pairs[ch]:
                    problems.append("%s:%d: unbalanced %r" % (rel, src.count("\n", 0, i) + 1, ch))
                    break
                stack.pop()
        else:
            if stack:
                problems.append("%s:%d: %r is never closed" % (rel, src.count("\n", 0, stack[-1][1]) + 1, stack[-1][0]))
        # imports that are never used / packages used without an import (the two commonest "first compile" errors)
        imports = {}
        im = re.search(r"^import\s*\((.*?)^\)", raw, flags=re.S | re.M)
        lines = im.group(1).splitlines() if im else re.findall(r'^import\s+(.*)$', raw, flags=re.M)
        for line in lines:
            m = re.match(r'\s*(\w+\s+)?"([^"]+)"', line)
            if m:
                alias = (m.group(1) or "").strip() or m.group(2).split("/")[-1]
                if alias == "v3":
                    alias = "bloom"
                imports[alias] = m.group(2)
        for alias, pkg in imports.items():
            if alias in ("_", "C"):
                continue
            if not re.search(r"\b%s\." % re.escape(alias), src.split("\n)", 1)[-1] if im else src):
                problems.append("%s: import %s is never used" % (rel, pkg))


def declared_identifiers(src_stripped: str):
    """Names a Go file declares: funcs, methods, types, struct fields, consts, vars, parameters, := and range targets, labels."""
    names = set()
    names.update(re.findall(r"\bfunc\s+(?:\([^)]*\)\s*)?(\w+)", src_stripped))
    names.update(re.findall(r"\btype\s+(\w+)", src_stripped))
    for m in re.finditer(r"\b(?:var|const)\s+(\w+(?:\s*,\s*\w+)*)", src_stripped):
        names.update(x.strip() for x in m.group(1).split(","))
    for m in re.finditer(r"\b(?:var|const)\s*\((.*?)\n\)", src_stripped, flags=re.S):
        for line in m.group(1).splitlines():
            lm = re.match(r"\s*(\w+(?:\s*,\s*\w+)*)", line)
            if lm:
                names.update(x.strip() for x in lm.group(1).split(","))
    for m in re.finditer(r"((?:\w+\s*,\s*)*\w+)\s*:=", src_stripped):
        names.update(x.strip() for x in m.group(1).split(","))
    for m in re.finditer(r"\bfor\s+((?:\w+\s*,\s*)?\w+)\s*:?=\s*range\b", src_stripped):
        names.update(x.strip() for x in m.group(1).split(","))
    # parameters and results: identifiers directly followed by a type inside func signatures
    for m in re.finditer(r"\bfunc\b[^{]*", src_stripped):
        for pm in re.finditer(r"[(,]\s*(\w+(?:\s*,\s*\w+)*)\s+(?:\.\.\.)?[\*\[\]\w\.]", m.group(0)):
            names.update(x.strip() for x in pm.group(1).split(","))
    # struct fields
    for m in re.finditer(r"\bstruct\s*\{(.*?)\n\s*\}", src_stripped, flags=re.S):
        for line in m.group(1).splitlines():
            lm = re.match(r"\s*(\w+(?:\s*,\s*\w+)*)\s+[\*\[\]\w\.]", line)
            if lm:
                names.update(x.strip() for x in lm.group(1).split(","))
    return names


def check_reference_names(overlay_files, binding_files, reference, problems):
    # the reference AS PATCHED: the overlay also relies on what engine_gpu.patch adds (GPUDevices, gpuRows, ...)
    work = tempfile.mkdtemp()
    try:
        for p in glob.glob(os.path.join(reference, "*.go")):
            shutil.copy(p, work)
        subprocess.run(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "go", "overlay", "engine_gpu.patch")], cwd=work, capture_output=True)
        ref_src = "\n".join(strip_go(open(p).read()) for p in sorted(glob.glob(os.path.join(work, "*.go"))) if not p.endswith("_test.go"))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    ref_test_src = "\n".join(strip_go(open(p).read()) for p in sorted(glob.glob(os.path.join(reference, "*_test.go"))))
    ref_idents = set(re.findall(r"\b[A-Za-z_]\w*\b", ref_src))
    ref_test_idents = set(re.findall(r"\b[A-Za-z_]\w*\b", ref_test_src))
    binding_idents = set()
    for p in binding_files:
        binding_idents |= set(re.findall(r"\b[A-Za-z_]\w*\b", strip_go(open(p).read())))
    overlay_src = {p: strip_go(open(p).read()) for p in overlay_files}
    local = set()
    for s in overlay_src.values():
        local |= declared_identifiers(s)
    for path, src in overlay_src.items():
        rel = os.path.relpath(path, ROOT)
        is_test = path.endswith("_test.go")
        body = re.sub(r"^import\s*\(.*?^\)", "", src, flags=re.S | re.M)
        body = re.sub(r"^package\s+\w+", "", body, flags=re.M)
        body = re.sub(r"^//go:build.*$", "", body, flags=re.M)
        pkgs = {"bloomgpu", "bloom", "errors", "fmt", "io", "slog", "sort", "sync", "time", "os", "strconv", "strings", "testing", "rand", "json", "bytes",
                "context", "runtime", "math", "bufio", "binary", "crc32", "unsafe", "flag", "filepath", "log"}
        seen = set()
        for m in re.finditer(r"\b[A-Za-z_]\w*\b", body):
            name = m.group(0)
            if name in seen:
                continue
            seen.add(name)
            if name in GO_KEYWORDS or name in GO_PREDECLARED or name in pkgs or name in local or name in STD_NAMES or name == "_":
                continue
            if name in ref_idents or name in binding_idents or (is_test and name in ref_test_idents):
                continue
            if len(name) == 1:
                continue
            problems.append("%s:%d: %s is declared neither in the overlay, nor in the reference's sources, nor in the binding"
                            % (rel, body.count("\n", 0, m.start()) + 1, name))
    # the names the overlay needs from the reference must be DECLARATIONS there, not mere mentions
    needed = {
        "func": ["newBloomEntrySets", "makeFieldTokenKey", "encodeFilterSection", "parseFilterSection", "readFullAt", "recordUnreadBlocks", "NewBlockRowScanner",
                 "isBasicWhitespaceLowerTokenizer", "BasicWhitespaceLowerTokenizer", "evalMatcherNode"],
        "method": ["indexRow", "buildFilters", "counts", "unionInto", "recordBlockError", "recordBlockStats", "matchRowBytes", "OnDiskSize", "Next"],
        "type": ["bloomEntrySets", "BloomFilters", "BloomEntryCounts", "BloomQuery", "BloomExpression", "partitionBuffer", "fileFilterJob", "blockScanCandidate",
                 "DataBlockMetadata", "BlockStats", "Results", "compiledRowMatcher", "rowMatchScratch", "matcherNode", "ValueTokenizerFunc", "BloomSearchEngineConfig"],
        "field": ["fields", "tokens", "fieldTokens", "entries", "filePointer", "RowDataOffset", "BloomFilterOffset", "BloomFilterSize", "Rows", "matchesAll",
                  "neverMatches", "fastTokens", "regexConds", "conditions", "root", "children", "cond", "kind", "field", "token", "index", "filterDuration",
                  "FieldBloomFilter", "TokenBloomFilter", "FieldTokenBloomFilter", "Fields", "Tokens", "FieldTokens", "Expression", "ExpressionType", "Condition",
                  "Children", "Type", "Field", "Token", "BloomFilterSkipped", "TotalRows", "TotalBytes", "Duration", "FilePointer", "BlockOffset", "ctx"],
        "const": ["BloomField", "BloomToken", "BloomFieldToken", "BloomExpressionCondition", "BloomExpressionAnd", "BloomExpressionOr", "rowCondField", "rowCondToken",
                  "rowCondFieldToken", "matcherNodeTrue", "matcherNodeCond", "matcherNodeAnd", "matcherNodeOr", "ErrInvalidHash", "ErrInvalidConfig"],
    }
    pats = {"func": r"^func\s+%s\s*\(", "method": r"^func\s+\([^)]*\)\s*%s\s*\(", "type": r"^type\s+%s\b", "field": r"^\s+%s\s+[\*\[\]\w\.]",
            "const": r"^\s*%s\b.*(=|\biota\b)|^\s+%s\s*$|^\s+%s\s+\w+\s*=|^var\s+%s\b|^\s+%s\s*=\s"}
    for kind, names in needed.items():
        for name in names:
            pat = pats[kind].replace("%s", re.escape(name))
            if not re.search(pat, ref_src, flags=re.M):
                problems.append("reference: no %s declaration of %s (the overlay relies on it)" % (kind, name))


def hook_signatures(path: str):
    """{name: parameter count} of the gpuEngine / gpuFlushFilters / gpuRowVerdicts methods and openGPUEngine in a file."""
    src = strip_go(open(path).read())
    out = {}
    for m in re.finditer(r"^func\s+(?:\(\s*\w*\s*\*?(gpuEngine|gpuFlushFilters|gpuRowVerdicts)\s*\)\s*)?(\w+)\s*\(", src, flags=re.M):
        if m.group(1) is None and m.group(2) != "openGPUEngine":
            continue
        close = matching_paren(src, m.end() - 1)
        params = split_args(src[m.end():close])
        n = 0
        for prm in params:                       # "a, b int" arrives as separate items: each item is one parameter
            n += 1
        out[(m.group(1) or "") + "." + m.group(2)] = n
    return out


def check_patch(reference, problems, notes):
    patch = os.path.join(ROOT, "go", "overlay", "engine_gpu.patch")
    work = tempfile.mkdtemp()
    try:
        for p in glob.glob(os.path.join(reference, "*.go")):
            shutil.copy(p, work)
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=work, capture_output=True, text=True)
        if r.returncode != 0:
            problems.append("go/overlay/engine_gpu.patch does not apply to %s:\n%s%s" % (reference, r.stdout, r.stderr))
            return
        subprocess.run(["patch", "-p1", "-s", "-i", patch], cwd=work, check=True)
        patched = [os.path.join(work, f) for f in ("engine.go", "ingest.go", "flush.go", "merge.go", "query_exec.go")]
        sub = []
        check_syntax(patched, sub)
        problems += [s.replace(work, "patched reference") for s in sub if "import" not in s]
        # every hook the patch calls must exist, with that many arguments, in the real engine file AND in the stub
        real = hook_signatures(os.path.join(ROOT, "go", "overlay", "gpu_engine.go"))
        stub = hook_signatures(os.path.join(ROOT, "go", "overlay", "gpu_engine_stub.go"))
        internal = {"gpuEngine.scope", "gpuEngine.release", "gpuEngine.done", "gpuEngine.arenaFor", "gpuEngine.loadArena", "gpuEngine.keepArena", "gpuEngine.dropLocked", "gpuEngine.evictLocked"}     # helpers the patch never calls
        if set(real) - internal != set(stub):
            problems.append("gpu_engine.go and gpu_engine_stub.go declare different hooks: %s" % sorted((set(real) - internal) ^ set(stub)))
        for name in set(real) & set(stub):
            if real[name] != stub[name] and name.split(".")[1] not in ("scope", "release", "done", "arenaFor", "dropLocked", "evictLocked"):
                problems.append("hook %s takes %d parameters in gpu_engine.go and %d in the stub" % (name, real[name], stub[name]))
        added = "\n".join(line[1:] for line in open(patch).read().splitlines() if line.startswith("+") and not line.startswith("+++"))
        added = strip_go(added)
        calls = {"b.gpu.": "gpuEngine", "gpuFilters.": "gpuFlushFilters", "gpuRows.": "gpuRowVerdicts"}
        n_calls = 0
        for prefix, typ in calls.items():
            for m in re.finditer(re.escape(prefix) + r"(\w+)\s*\(", added):
                close = matching_paren(added, m.end() - 1)
                n = len(split_args(added[m.end():close]))
                key = typ + "." + m.group(1)
                n_calls += 1
                if key not in real:
                    problems.append("engine_gpu.patch calls %s, which gpu_engine.go does not declare" % key)
                elif real[key] != n:
                    problems.append("engine_gpu.patch calls %s with %d arguments; it takes %d" % (key, n, real[key]))
        m = re.search(r"openGPUEngine\s*\(", added)
        if not m or len(split_args(added[m.end():matching_paren(added, m.end() - 1)])) != real.get(".openGPUEngine"):
            problems.append("engine_gpu.patch's openGPUEngine call does not match its declaration")
        notes.append("engine_gpu.patch applies to %s (%d hunks, %d hook calls checked against gpu_engine.go and the stub)"
                     % (reference, open(patch).read().count("\n@@"), n_calls + 1))
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args(argv)
    header = parse_header(os.path.join(ROOT, "include", "bloomgpu.h"))
    binding = sorted(glob.glob(os.path.join(ROOT, "go", "bloomgpu", "*.go")))
    overlay = sorted(glob.glob(os.path.join(ROOT, "go", "overlay", "*.go")))
    cmd = sorted(glob.glob(os.path.join(ROOT, "cmd", "*", "*.go")))
    problems, notes = [], []
    check_cgo(binding + overlay + cmd, header, problems)
    check_syntax(binding + overlay + cmd, problems)
    if os.path.isdir(args.reference) and glob.glob(os.path.join(args.reference, "*.go")):
        check_reference_names(overlay, binding, args.reference, problems)
        check_patch(args.reference, problems, notes)
    else:
        notes.append("no reference checkout at %s: identifier and patch checks skipped" % args.reference)
    for n in notes:
        print("note:", n)
    for p in problems:
        print("PROBLEM:", p)
    print("%d Go files, %d C-ABI functions in the header: %s" % (len(binding + overlay + cmd), len(header[0]),
                                                               "clean" if not problems else "%d problem(s)" % len(problems)))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
