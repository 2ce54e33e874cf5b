"""The bindings that cannot be compiled here (Go: no toolchain) or are only compiled in GPU tests (C++ mirror) are
checked against include/pbsgpu.h at the source level: every pbsgpu_* function they call is declared, with the same
number of arguments, and every status code / type they name exists. A maintainer wiring the cgo stub of INTEGRATION.md
into the fork starts from a binding that at least agrees with the header."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _split_args(s: str):
    """top-level comma split of an argument list (parentheses / brackets / braces balanced)"""
    args, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


def _calls(src: str, prefix: str):
    """(name, nargs) of every `<prefix>pbsgpu_xxx(...)` call in src, with balanced parentheses across lines"""
    out = []
    for m in re.finditer(re.escape(prefix) + r"(pbsgpu_[a-z0-9_]+)\s*\(", src):
        i, depth = m.end(), 1
        while depth and i < len(src):
            depth += src[i] == "("
            depth -= src[i] == ")"
            i += 1
        inner = src[m.end():i - 1]
        out.append((m.group(1), len(_split_args(inner)) if inner.strip() not in ("", "void") else 0))
    return out


def _header_functions():
    src = _strip_comments(open(os.path.join(ROOT, "include", "pbsgpu.h")).read())
    decls = {}
    for m in re.finditer(r"\b(pbsgpu_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        params = m.group(2).strip()
        decls[m.group(1)] = 0 if params in ("", "void") else len(_split_args(params))
    return decls, src


def test_header_parser_sees_the_whole_abi():
    decls, _ = _header_functions()
    assert len(decls) >= 55 and decls["pbsgpu_config_init"] == 3 and decls["pbsgpu_last_hip_error"] == 0, len(decls)
    assert decls["pbsgpu_submit_host_suggested"] == 8 and decls["pbsgpu_stream_begin_entry"] == 4


def test_go_binding_agrees_with_the_header():
    decls, hdr = _header_functions()
    go = _strip_comments(open(os.path.join(ROOT, "go", "pbsgpu", "pbsgpu.go")).read())
    calls = _calls(go, "C.")
    assert len(calls) >= 38, len(calls)
    for name, nargs in calls:
        assert name in decls, f"go binding calls undeclared {name}"
        assert nargs == decls[name], f"{name}: go passes {nargs} arguments, header declares {decls[name]}"
    for const in set(re.findall(r"C\.(PBSGPU_[A-Z0-9_]+)", go)):
        assert re.search(r"\b%s\b" % const, hdr), const
    for typ in set(re.findall(r"C\.(pbsgpu_[a-z0-9_]+)\b(?!\s*\()", go)):
        assert re.search(r"\b%s\b" % typ, hdr), typ
    # the entry points a commit / tape ingest needs are all bound
    bound = {n for n, _ in calls}
    for need in ("pbsgpu_stream_write", "pbsgpu_stream_reserve", "pbsgpu_stream_commit", "pbsgpu_stream_cut",
                 "pbsgpu_stream_begin_file", "pbsgpu_stream_end_file", "pbsgpu_stream_poll_files", "pbsgpu_stream_suggest",
                 "pbsgpu_submit_host", "pbsgpu_ticket_done", "pbsgpu_collect", "pbsgpu_reuse_lookup", "pbsgpu_reuse_should",
                 "pbsgpu_didx_encode", "pbsgpu_didx_decode", "pbsgpu_xxh3_many_host", "pbsgpu_sha256_many_host"):
        assert need in bound, need


def test_cpp_mirror_agrees_with_the_header():
    decls, _ = _header_functions()
    hpp = _strip_comments(open(os.path.join(ROOT, "include", "pbsgpu.hpp")).read())
    calls = [(n, k) for n, k in _calls(hpp, "") if n in decls or not n.endswith("_t")]
    assert len(calls) >= 20
    for name, nargs in calls:
        if name not in decls:      # a type used in a functional cast / constructor, e.g. pbsgpu_segment{...}: not a call
            continue
        assert nargs == decls[name], f"{name}: mirror passes {nargs} arguments, header declares {decls[name]}"


def test_python_binding_declares_every_function_it_calls():
    """_lib.py sets argtypes per symbol: the count must match the header (ctypes would accept a wrong call silently)."""
    decls, _ = _header_functions()
    from pbs_plus_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = _lib.lib()
    checked = 0
    for name, n in decls.items():
        fn = getattr(L, name, None)
        if fn is None or fn.argtypes is None:
            continue
        assert len(fn.argtypes) == n, f"{name}: ctypes argtypes has {len(fn.argtypes)} entries, header declares {n}"
        checked += 1
    assert checked >= 50, checked


def _go_exports(src: str):
    src = _strip_comments(src)
    funcs = set(re.findall(r"^func\s+([A-Z]\w*)\s*\(", src, flags=re.M))
    meths = set((m.group(1), m.group(2)) for m in re.finditer(r"^func\s+\(\w+\s+\*?(\w+)\)\s+([A-Z]\w*)\s*\(", src, flags=re.M))
    types = (set(re.findall(r"^type\s+([A-Z]\w*)\b", src, flags=re.M)) | set(re.findall(r"^\t([A-Z]\w*)\s+struct\b", src, flags=re.M))
             | set(re.findall(r"^\t(Ticket)\s+uint64\b", src, flags=re.M)))
    vars_ = set(re.findall(r"^(?:var\s+|\t)(Err[A-Z]\w*)\s*=", src, flags=re.M))
    return funcs, meths, types, vars_


def test_go_fallback_mirrors_the_whole_exported_surface():
    """CGO_ENABLED=0 builds (the project's default) must compile code written against the GPU binding: every exported
    function, method, type and error value of pbsgpu.go exists in fallback.go."""
    gf, gm, gt, gv = _go_exports(open(os.path.join(ROOT, "go", "pbsgpu", "pbsgpu.go")).read())
    ff, fm, ft, fv = _go_exports(open(os.path.join(ROOT, "go", "pbsgpu", "fallback.go")).read())
    assert len(gm) >= 40 and len(gf) >= 4, (len(gm), len(gf))
    assert gf <= ff, sorted(gf - ff)
    assert gm <= fm, sorted(gm - fm)
    assert gt <= ft, sorted(gt - ft)
    assert gv <= fv | {"ErrNotBuilt"}, sorted(gv - fv)
