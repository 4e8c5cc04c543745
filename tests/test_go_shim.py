"""The cgo shim under integration/go/ cannot be compiled here (no Go toolchain in the image).  It is still held to
the C ABI mechanically: every C.sdb_* call names a function include/semadb_amd.h declares and passes as many
arguments as that declaration has parameters, every C.SDB_* constant exists, every helper the package calls is
defined in it, and the exported surface SURVEY 8b lists is there with the reference's signatures.  Since round 6 also
a poor man's front end (names resolve, nothing unused, call arity inside the package, selectors on the package's own
structs, terminating statements): the classes of error an edit leaves behind and a compiler would have refused."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "semadb_amd.h")
GO = sorted(glob.glob(os.path.join(ROOT, "integration", "go", "*", "*.go")))


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _split_top_level(args):
    parts, depth, cur = [], 0, ""
    for ch in args:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def _balanced(text, start):
    """text[start] == '(' -> index just past its matching ')'"""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced call")


def header_functions():
    text = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"\b(sdb_[a-z0-9_]+)\s*\(", text):
        end = _balanced(text, m.end() - 1)
        params = text[m.end():end - 1].strip()
        out[m.group(1)] = 0 if params in ("", "void") else len(_split_top_level(params))
    return out


def header_param_types():
    """name -> list of parameter kinds: ("ptr", pointee C type) or ("val", C type)"""
    text = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"\b(sdb_[a-z0-9_]+)\s*\(", text):
        end = _balanced(text, m.end() - 1)
        params = text[m.end():end - 1].strip()
        kinds = []
        if params not in ("", "void"):
            for p in _split_top_level(params):
                p = re.sub(r"\bconst\b", "", p).strip()
                mm = re.match(r"^(.*?)(\**)\s*([A-Za-z_][A-Za-z0-9_]*)?(\[[^\]]*\])?$", p)
                base, stars = mm.group(1).strip(), mm.group(2)
                if p.endswith("]") or stars:
                    kinds.append(("ptr", base if len(stars) <= 1 and not (stars and p.endswith("]")) else base + "*"))
                else:
                    kinds.append(("val", base))
        out[m.group(1)] = kinds
    return out


def _arg_kind(expr):
    """what a Go argument expression visibly is: ("ptr", pointee or None), ("val", C type or None) or None (an
    identifier whose type this test cannot see)"""
    e = expr.strip()
    if e == "nil":
        return ("ptr", None)
    m = re.match(r"^\(\*C\.([A-Za-z0-9_]+)\)\(", e)
    if m:
        return ("ptr", m.group(1))
    m = re.match(r"^\(\*\*C\.([A-Za-z0-9_]+)\)\(", e)
    if m:
        return ("ptr", m.group(1) + "*")
    if e.startswith("unsafe.Pointer("):
        return ("ptr", "void")
    if e.startswith("&"):
        return ("ptr", None)
    m = re.match(r"^C\.([a-z][A-Za-z0-9_]*)\(", e)
    if m:
        return ("val", m.group(1))
    if re.match(r"^C\.SDB_[A-Z0-9_]+$", e) or re.match(r"^-?\d+$", e):
        return ("val", None)
    return None


def test_every_c_call_passes_arguments_of_the_declared_kind():
    """pointer where the header wants a pointer (and to the same C type when the cast names one), C.<type>(...) of the
    declared width where it wants a scalar -- the mistakes cgo would refuse to compile"""
    decl = header_param_types()
    widths = {"int": "int", "uint32_t": "uint32_t", "uint64_t": "uint64_t", "float": "float", "size_t": "size_t",
              "int32_t": "int32_t", "uint8_t": "uint8_t", "int64_t": "int64_t"}
    checked = 0
    for path, text in go_sources().items():
        code = _go_code(text)
        for m in re.finditer(r"\bC\.(sdb_[a-z0-9_]+)\s*\(", code):
            name = m.group(1)
            end = _balanced(code, m.end() - 1)
            inner = code[m.end():end - 1].strip()
            args = [] if inner == "" else _split_top_level(inner)
            for i, (arg, (kind, ctype)) in enumerate(zip(args, decl[name])):
                got = _arg_kind(arg)
                where = "%s: argument %d of C.%s (%s)" % (os.path.basename(path), i + 1, name, arg.strip())
                if got is None:
                    continue
                assert got[0] == kind, "%s is a %s, the header wants a %s" % (where, got[0], kind)
                if got[1] is None or ctype == "void":
                    continue
                if kind == "ptr" and got[1] != "void":
                    assert got[1] == ctype, "%s points at %s, the header says %s" % (where, got[1], ctype)
                if kind == "val":
                    assert widths.get(got[1], got[1]) == ctype, "%s is a %s, the header says %s" % (where, got[1], ctype)
                checked += 1
    assert checked >= 60


def header_constants():
    text = open(HEADER).read()
    names = set(re.findall(r"#define\s+(SDB_[A-Z0-9_]+)", text))
    names |= set(re.findall(r"\b(SDB_(?:OK|ERR_[A-Z_]+))\b", text))
    return names


def go_sources():
    assert GO, "integration/go/*/*.go is missing"
    return {p: open(p).read() for p in GO}


def _go_code(text):
    """Go source without // comments, /* */ comments (the cgo preamble included) and string literals"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
    return re.sub(r"//[^\n]*", "", text)


def test_every_c_call_matches_the_header():
    decl = header_functions()
    calls = 0
    for path, text in go_sources().items():
        code = _go_code(text)
        for m in re.finditer(r"\bC\.(sdb_[a-z0-9_]+)\s*\(", code):
            name = m.group(1)
            assert name in decl, "%s calls C.%s, which the header does not declare" % (os.path.basename(path), name)
            end = _balanced(code, m.end() - 1)
            inner = code[m.end():end - 1].strip()
            n = 0 if inner == "" else len(_split_top_level(inner))
            assert n == decl[name], "%s: C.%s called with %d arguments, declared with %d" % (
                os.path.basename(path), name, n, decl[name])
            calls += 1
    assert calls >= 25


def test_every_c_constant_and_type_exists():
    consts = header_constants()
    types = set(re.findall(r"typedef struct (sdb_[a-z_]+)", open(HEADER).read())) | {"sdb_index_params"}
    for path, text in go_sources().items():
        code = _go_code(text)
        for name in re.findall(r"\bC\.(SDB_[A-Z0-9_]+)\b", code):
            assert name in consts, "%s uses C.%s" % (os.path.basename(path), name)
        for name in re.findall(r"\bC\.(sdb_[a-z_]+)\b(?!\s*\()", code):
            assert name in types, "%s uses type C.%s" % (os.path.basename(path), name)


def test_helpers_are_defined_and_surface_is_complete():
    by_pkg = {}
    for path, text in go_sources().items():
        by_pkg.setdefault(os.path.basename(os.path.dirname(path)), []).append(_go_code(text))
    vam = "\n".join(by_pkg["vamana"])
    defined = set(re.findall(r"^func (?:\([a-z]+ \*?[A-Za-z]+\) )?([A-Za-z_][A-Za-z0-9_]*)\(", vam, flags=re.M))
    # every helper INTEGRATION.md used to name without writing
    for helper in ["newSearchBatcher", "submit", "packFilters", "flushToBucket", "exists", "deviceForShard",
                   "randomUnitVector", "exportVectors", "EdgeScan", "loadFromBucket", "fit", "flush", "stop", "loop"]:
        assert helper in defined, "vamana package does not define %s" % helper
    # calls to package-level or method helpers must resolve inside the package
    builtin = {"make", "append", "len", "copy", "close", "float32", "uint64", "uint32", "int", "int64", "uint8", "byte",
               "string", "panic", "new", "cap", "delete", "func", "go", "defer", "if", "for", "switch", "return", "select",
               "map", "chan", "range", "float64", "uint", "bool", "error"}
    for name in set(re.findall(r"(?<![\w.])([a-z][A-Za-z0-9_]*)\(", vam)):
        assert name in defined or name in builtin, "vamana package calls %s(), not defined in it" % name
    for name in set(re.findall(r"\b(?:v|b|ix)\.([a-z][A-Za-z0-9_]*)\(", vam)):
        assert name in defined or name in {"mu", "cond", "wg"}, "method %s() is not defined in the vamana package" % name
    for helper in ["takeFree", "rotateLocked", "flushSlab"]:
        assert helper in defined, "the batcher does not define %s" % helper
    assert "C.sdb_host_alloc" in vam and "C.sdb_host_free" in vam  # pinned slabs
    # the exported surface of the reference package, with its signatures
    for sig in [r"const STARTID = 1",
                r"type IndexVectorChange struct \{\s*Id\s+uint64\s*Vector \[\]float32\s*\}",
                r"func NewIndexVamana\(name string, params models\.IndexVectorVamanaParameters, bucket diskstore\.Bucket\) \(\*IndexVamana, error\)",
                r"func \(v \*IndexVamana\) Search\(ctx context\.Context, q models\.SearchVectorVamanaOptions, filter \*roaring64\.Bitmap\) \(\*roaring64\.Bitmap, \[\]models\.SearchResult, error\)",
                r"func \(v \*IndexVamana\) InsertUpdateDelete\(ctx context\.Context, points <-chan IndexVectorChange\) <-chan error",
                r"func \(v \*IndexVamana\) UpdateBucket\(bucket diskstore\.Bucket\)",
                r"func \(v \*IndexVamana\) SizeInMemory\(\) int64",
                r"func \(v \*IndexVamana\) EdgeScan\(deleteSet map\[uint64\]struct\{\}\) \(toPrune, toSave \[\]uint64, err error\)"]:
        assert re.search(sig, vam), "missing from the vamana package: %s" % sig
    # the distance package keeps the AVX2 assembly for pair-wise calls (distance_amd64.go:19-27): no override of the
    # package-level function variables with a kernel launch per pair, only a batched entry point
    dist = "\n".join(by_pkg["distance"])
    assert "dotProductImpl =" not in dist and "euclideanDistance =" not in dist
    assert re.search(r"func BatchDistance\(name string, dim int, queries, candidates \[\]float32, device int\) \(\[\]float32, error\)", dist)
    assert "C.sdb_distance_batch" in dist
    clu = "\n".join(by_pkg["cluster"])
    for fn in ["sdb_cluster_create_local", "sdb_cluster_search_batch", "sdb_cluster_destroy", "sdb_shard_limit",
               "sdb_cluster_next_ticket"]:
        assert "C." + fn in clu
    # one ticket per request, drawn under a lock, handed to every rank's call
    assert re.search(r"ticket := f\.nextTicket\(\)", clu) and re.search(r"C\.sdb_cluster_search_batch\(f\.ranks\[r\], f\.indexes\[r\], C\.uint64_t\(ticket\)", clu)


def test_imports_are_the_reference_modules():
    """import paths: the standard library or what the reference's go.mod (go 1.22; roaring v1.9.4) and its own tree
    provide -- a path from another major version would not resolve in the reference's module"""
    allowed = {"context", "fmt", "hash/fnv", "math", "math/rand/v2", "sync", "sync/atomic", "runtime", "time", "unsafe",
               "sort", "errors",
               "github.com/RoaringBitmap/roaring/roaring64", "github.com/semafind/semadb/conversion",
               "github.com/semafind/semadb/diskstore", "github.com/semafind/semadb/models",
               "github.com/semafind/semadb/shard/index/vamana"}
    for path, text in go_sources().items():
        block = re.search(r"\nimport \((.*?)\n\)", text, flags=re.S)
        if not block:
            continue
        for imp in re.findall(r'"([^"]+)"', block.group(1)):
            assert imp in allowed, "%s imports %s" % (os.path.basename(path), imp)


def test_flat_package_surface():
    """shard/index/flat's exported surface with the reference's signatures (flat.go:17,21,33,37,41,76), every helper
    it calls defined in the package, and the write path made of the calls the header documents for a flat index"""
    text = _go_code(open(os.path.join(ROOT, "integration", "go", "flat", "flat_mi355x.go")).read())
    for sig in [r"type IndexFlat struct",
                r"func NewIndexFlat\(params models\.IndexVectorFlatParameters, bucket diskstore\.Bucket\) \(inf IndexFlat, err error\)",
                r"func \(inf IndexFlat\) SizeInMemory\(\) int64",
                r"func \(inf IndexFlat\) UpdateBucket\(bucket diskstore\.Bucket\)",
                r"func \(inf IndexFlat\) InsertUpdateDelete\(ctx context\.Context, points <-chan vamana\.IndexVectorChange\) <-chan error",
                r"func \(inf IndexFlat\) Search\(ctx context\.Context, options models\.SearchVectorFlatOptions, filter \*roaring64\.Bitmap\) \(\*roaring64\.Bitmap, \[\]models\.SearchResult, error\)"]:
        assert re.search(sig, text), "missing from the flat package: %s" % sig
    defined = set(re.findall(r"^func (?:\([a-z]+ \*?[A-Za-z]+\) )?([A-Za-z_][A-Za-z0-9_]*)\(", text, flags=re.M))
    builtin = {"make", "append", "len", "copy", "close", "float32", "uint64", "uint32", "int", "int64", "string", "func",
               "go", "defer", "if", "for", "switch", "return", "range", "error"}
    for name in set(re.findall(r"(?<![\w.])([a-z][A-Za-z0-9_]*)\(", text)):
        assert name in defined or name in builtin, "flat package calls %s(), not defined in it" % name
    for name in set(re.findall(r"\b(?:s|inf\.s)\.([a-z][A-Za-z0-9_]*)\(", text)):
        assert name in defined, "method %s() is not defined in the flat package" % name
    for fn in ["sdb_index_begin_write", "sdb_index_set_vectors", "sdb_index_remove_vectors", "sdb_index_commit",
               "sdb_index_flat_search", "sdb_index_compact"]:
        assert "C." + fn in text


def test_every_file_has_the_build_tag_and_balanced_braces():
    for path, text in go_sources().items():
        assert text.startswith("//go:build mi355x\n"), path
        code = _go_code(text)
        for a, b in ["()", "{}", "[]"]:
            assert code.count(a) == code.count(b), "%s: unbalanced %s%s" % (os.path.basename(path), a, b)


GO_BUILTIN = set("""break default func interface select case defer go map struct chan else goto package switch const
fallthrough if range type continue for import return var append cap close complex copy delete imag len make new panic
print println real recover bool byte complex64 complex128 error float32 float64 int int8 int16 int32 int64 rune
string uint uint8 uint16 uint32 uint64 uintptr true false iota nil any min max clear _""".split())

def _go_strip(text):
    """comments -> nothing, string / rune literals -> "" / 0, one left-to-right scan (an apostrophe in a comment is not
    a rune, a // in a string is not a comment)"""
    out, i, n = [], 0, len(text)
    while i < n:
        two = text[i:i + 2]
        if two == "//":
            i = text.find("\n", i)
            i = n if i < 0 else i
        elif two == "/*":
            j = text.index("*/", i) + 2
            out.append("\n" * text.count("\n", i, j))
            i = j
        elif text[i] == "`":
            j = text.index("`", i + 1) + 1
            out.append('""' + "\n" * text.count("\n", i, j))
            i = j
        elif text[i] in "\"'":
            q, j = text[i], i + 1
            while text[j] != q:
                j += 2 if text[j] == "\\" else 1
            out.append('""' if q == '"' else "0")
            i = j + 1
        else:
            out.append(text[i])
            i += 1
    return "".join(out)

def _go_funcs(text):
    for m in re.finditer(r"^func\b", text, flags=re.M):
        i = text.index("{", m.end())
        # the body's brace: skip braces of interface{} / struct{} in the signature
        depth = 0
        j = m.end()
        while True:
            ch = text[j]
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            elif ch == "{" and depth == 0 and not re.search(r"(interface|struct)\s*$", text[:j]):
                break
            j += 1
        d, k = 0, j
        while True:
            if text[k] == "{": d += 1
            elif text[k] == "}":
                d -= 1
                if d == 0: break
            k += 1
        yield text[m.start():j], text[j:k + 1]

def _go_declared(sig, body):
    names = set()
    for m in re.finditer(r"([A-Za-z_]\w*(?:\s*,\s*[A-Za-z_]\w*)*)\s+(?:\.\.\.|\*|\[|<-|func\b|map\b|chan\b|[A-Za-z_])", sig):
        names.update(x.strip() for x in m.group(1).split(","))
    for m in re.finditer(r"([A-Za-z_][\w]*(?:\s*,\s*[A-Za-z_]\w*)*)\s*:=", body):
        names.update(x.strip() for x in m.group(1).split(","))
    for m in re.finditer(r"\b(?:var|const)\s+([A-Za-z_]\w*(?:\s*,\s*[A-Za-z_]\w*)*)", body):
        names.update(x.strip() for x in m.group(1).split(","))
    for m in re.finditer(r"\bfunc\s*\(([^)]*)\)", body):       # closures' parameters
        for p in m.group(1).split(","):
            p = p.strip().split()
            if p: names.add(p[0])
    for m in re.finditer(r"^\s*([A-Za-z_]\w*):\s*$", body, flags=re.M):   # labels
        names.add(m.group(1))
    return names

def _go_package_names(texts):
    names = set()
    for t in texts:
        for m in re.finditer(r"^func\s+(?:\([^)]*\)\s*)?([A-Za-z_]\w*)", t, flags=re.M): names.add(m.group(1))
        for m in re.finditer(r"^(?:type|var|const)\s+([A-Za-z_]\w*)", t, flags=re.M): names.add(m.group(1))
        for blk in re.finditer(r"^(?:var|const)\s*\((.*?)^\)", t, flags=re.M | re.S):
            for m in re.finditer(r"^\s*([A-Za-z_]\w*(?:\s*,\s*[A-Za-z_]\w*)*)", blk.group(1), flags=re.M):
                names.update(x.strip() for x in m.group(1).split(","))
    return names

def _go_imports(raw):
    out = set()
    blk = re.search(r"^import\s*\((.*?)^\)", raw, flags=re.M | re.S)
    lines = blk.group(1).splitlines() if blk else re.findall(r'^import\s+(.*)$', raw, flags=re.M)
    for ln in lines:
        m = re.match(r'\s*(?:([A-Za-z_]\w*)\s+)?"([^"]+)"', ln)
        if m:
            parts = m.group(2).split("/")
            if re.fullmatch(r"v\d+", parts[-1]) and len(parts) > 1:   # math/rand/v2 is package rand
                parts.pop()
            out.add(m.group(1) or parts[-1])
    return out

def _go_undeclared(pkgdir):
    raws = {p: open(p).read() for p in sorted(glob.glob(pkgdir + "/*.go"))}
    texts = {p: _go_strip(r) for p, r in raws.items()}
    pk = _go_package_names(texts.values())
    bad = []
    for p, t in texts.items():
        imp = _go_imports(raws[p]) | {"C"}
        for sig, body in _go_funcs(t):
            known = _go_declared(sig, body) | pk | imp | GO_BUILTIN
            for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)\b", body):
                name = m.group(1)
                if name in known: continue
                if re.match(r"\s*:[^=]", body[m.end():m.end() + 3]):   # key of a composite literal
                    continue
                bad.append((os.path.basename(p), sig.strip()[:70], name))
    return bad


def test_every_name_a_function_uses_is_declared():
    """No compiler here: a poor man's name resolution instead.  Every identifier in a function body (not a selector
    after a dot, not the key of a composite literal) must be a parameter, result, local, closure parameter or label of
    that function, a package-level name of the same package, an import of the file, or a Go builtin -- an edit that
    leaves a function using another function's local (round 6: `inSlab` in flush()) fails here."""
    bad = []
    for d in sorted(glob.glob(os.path.join(ROOT, "integration", "go", "*"))):
        bad += _go_undeclared(d)
    assert not bad, "undeclared names in the Go shim:\n" + "\n".join("%s: %s: %s" % b for b in bad)


def test_name_check_sees_a_planted_stray(tmp_path):
    pkg = tmp_path / "p"
    pkg.mkdir()
    (pkg / "a.go").write_text('package p\n\nimport "fmt"\n\nvar table = map[string]int{}\n\n'
                              'func a(x int) int {\n\t// it\'s a comment with "quotes\n\ty := x + table["k"]\n'
                              '\treturn y\n}\n\nfunc b(x int) {\n\tif y > 0 {\n\t\tfmt.Println(x)\n\t}\n}\n')
    assert [b[2] for b in _go_undeclared(str(pkg))] == ["y"]


def test_no_unused_locals_or_imports():
    """the two other things the Go compiler refuses and an edit here leaves behind easily"""
    bad = []
    for path, raw in go_sources().items():
        code = _go_strip(raw)
        for imp in _go_imports(raw) - {"C"}:
            if not re.search(r"(?<![\w.])%s\." % re.escape(imp), code):
                bad.append("%s: import %s is not used" % (os.path.basename(path), imp))
        for sig, body in _go_funcs(code):
            names = set()
            for m in re.finditer(r"([A-Za-z_]\w*(?:\s*,\s*[A-Za-z_]\w*)*)\s*:=", body):
                names.update(x.strip() for x in m.group(1).split(","))
            for m in re.finditer(r"\bvar\s+([A-Za-z_]\w*(?:\s*,\s*[A-Za-z_]\w*)*)", body):
                names.update(x.strip() for x in m.group(1).split(","))
            for name in names - {"_"}:
                if len(re.findall(r"(?<![\w.])%s\b" % re.escape(name), body)) < 2:
                    bad.append("%s: %s: %s declared and not used" % (os.path.basename(path), sig.strip()[:70], name))
    assert not bad, "\n".join(bad)


def _go_signature_arity(sig):
    """'func (recv) name(params) results' -> (name, is_method, fewest, most arguments)"""
    m = re.match(r"func\s*(\([^)]*\)\s*)?([A-Za-z_]\w*)\s*\(", sig)
    if not m:
        return None
    start = m.end() - 1
    plist = sig[start + 1:_balanced(sig, start) - 1].strip()
    if not plist:
        return m.group(2), bool(m.group(1)), 0, 0
    parts = _split_top_level(plist)  # "a, b int" are two parameters: one per comma-separated part
    variadic = "..." in parts[-1]
    return m.group(2), bool(m.group(1)), len(parts) - 1 if variadic else len(parts), 10 ** 6 if variadic else len(parts)


def _go_call_arity_mismatches(pkgdir):
    """calls of the package's own functions and methods (a method by its name, when every method of that name in the
    package takes the same number) whose argument count is not the declaration's; returns (mismatches, calls checked)"""
    texts = {p: _go_strip(open(p).read()) for p in sorted(glob.glob(pkgdir + "/*.go"))}
    funcs, methods = {}, {}
    for t in texts.values():
        for sig, _ in _go_funcs(t):
            r = _go_signature_arity(sig)
            if r:
                (methods if r[1] else funcs).setdefault(r[0], set()).add((r[2], r[3]))
    bad, checked = [], 0
    for p, t in texts.items():
        for sig, body in _go_funcs(t):
            for pattern, table in ((r"(?<![\w.])([a-z][A-Za-z0-9_]*)\(", funcs), (r"\.([a-z][A-Za-z0-9_]*)\(", methods)):
                for m in re.finditer(pattern, body):
                    name = m.group(1)
                    if name not in table or len(table[name]) != 1:
                        continue
                    args = body[m.end():_balanced(body, m.end() - 1) - 1].strip()
                    n = len(_split_top_level(args)) if args else 0
                    lo, hi = next(iter(table[name]))
                    checked += 1
                    if lo <= n <= hi or (n == 1 and re.match(r"^[\w.]+\(.*\)$", args, flags=re.S)):  # f(g()) spreads g's results
                        continue
                    bad.append("%s: %s: %s() called with %d arguments, declared with %s" %
                               (os.path.basename(p), sig.strip()[:60], name, n, lo if lo == hi else "%d or more" % lo))
    return bad, checked


def test_calls_inside_the_packages_pass_the_declared_number_of_arguments(tmp_path):
    total = 0
    for d in sorted(glob.glob(os.path.join(ROOT, "integration", "go", "*"))):
        bad, checked = _go_call_arity_mismatches(d)
        assert not bad, "\n".join(bad)
        total += checked
    assert total > 50, total  # the check does see the calls
    pkg = tmp_path / "p"
    pkg.mkdir()
    (pkg / "a.go").write_text("package p\n\ntype t struct{}\n\nfunc two(a, b int) int { return a + b }\n\n"
                              "func (x *t) one(a int, rest ...int) int { return a }\n\n"
                              "func use(x *t) int {\n\treturn two(1) + x.one() + two(1, 2) + x.one(1, 2, 3)\n}\n")
    bad, checked = _go_call_arity_mismatches(str(pkg))
    assert checked == 4 and len(bad) == 2 and "two() called with 1" in bad[0] and "one() called with 0" in bad[1]


def test_functions_with_results_end_in_a_terminating_statement():
    """'missing return' is a compile error: a function with results must end in return / panic, an endless `for {`
    or a `select {` / `switch` whose every branch ends that way (the last two are only recognised by their opener)."""
    bad = []
    for path, raw in go_sources().items():
        for sig, body in _go_funcs(_go_strip(raw)):
            m = re.match(r"func\s*(\([^)]*\)\s*)?([A-Za-z_]\w*)\s*\(", sig)
            if not m or not sig[_balanced(sig, m.end() - 1):].strip():
                continue  # no results
            depth, opener, last = 0, "", ""
            for line in body.strip()[1:-1].splitlines():
                s = line.strip()
                if not s:
                    continue
                after = depth + s.count("{") - s.count("}")
                if depth == 0 or after == 0:  # a statement of the function's own block, or the brace that closes one
                    last = s
                    if depth == 0 and s.endswith("{"):
                        opener = s
                depth = after
            ok = last.startswith(("return", "panic(")) or (last == "}" and re.match(r"^(for|select)\s*\{$", opener))
            if not ok:
                bad.append("%s: %s ends in %r" % (os.path.basename(path), sig.strip()[:70], last[:40]))
    assert not bad, "\n".join(bad)


def _go_structs_and_methods(texts):
    structs, methods = {}, {}
    for t in texts:
        for m in re.finditer(r"^type\s+([A-Za-z_]\w*)\s+struct\s*\{(.*?)^\}", t, flags=re.M | re.S):
            fields = set()
            for line in m.group(2).splitlines():
                line = line.strip()
                mm = re.match(r"^([A-Za-z_]\w*(?:\s*,\s*[A-Za-z_]\w*)*)\s+\S", line)
                if mm:
                    fields.update(x.strip() for x in mm.group(1).split(","))
                else:
                    mm = re.match(r"^\*?(?:\w+\.)?([A-Za-z_]\w*)$", line)  # an embedded type
                    if mm:
                        fields.add(mm.group(1))
            structs[m.group(1)] = fields
        for sig, _ in _go_funcs(t):
            mm = re.match(r"func\s*\(\s*\w+\s+\*?(\w+)\s*\)\s*(\w+)", sig)
            if mm:
                methods.setdefault(mm.group(1), set()).add(mm.group(2))
    return structs, methods


def _go_unknown_selectors(pkgdir):
    """x.name where x is a receiver, a parameter or a local made from a composite literal of one of the package's own
    struct types, and name is neither a field nor a method of that type; keys of the package's composite literals that
    are not fields.  Returns (findings, selectors checked)."""
    texts = {p: _go_strip(open(p).read()) for p in sorted(glob.glob(pkgdir + "/*.go"))}
    structs, methods = _go_structs_and_methods(texts.values())
    bad, checked = [], 0
    for p, t in texts.items():
        for sig, body in _go_funcs(t):
            typed = {}
            for m in re.finditer(r"([A-Za-z_]\w*(?:\s*,\s*[A-Za-z_]\w*)*)\s+\*?([A-Za-z_]\w*)\s*[,)]", sig):
                if m.group(2) in structs:
                    for name in m.group(1).split(","):
                        typed[name.strip()] = m.group(2)
            for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)\s*:=\s*&?([A-Za-z_]\w*)\{", body):
                if m.group(2) in structs:
                    typed[m.group(1)] = m.group(2)
            for var, typ in typed.items():
                if len(re.findall(r"(?<![\w.])%s\s*(?::=|,\s*\w+\s*:=)" % re.escape(var), body)) > (1 if re.search(
                        r"(?<![\w.])%s\s*:=\s*&?%s\{" % (re.escape(var), typ), body) else 0):
                    continue  # re-declared with something else somewhere: not followed
                for u in re.finditer(r"(?<![\w.])%s\.([A-Za-z_]\w*)" % re.escape(var), body):
                    checked += 1
                    if u.group(1) not in structs[typ] and u.group(1) not in methods.get(typ, set()):
                        bad.append("%s: %s: %s.%s is neither a field nor a method of %s" %
                                   (os.path.basename(p), sig.strip()[:50], var, u.group(1), typ))
            for m in re.finditer(r"(?<![\w.])&?([A-Za-z_]\w*)\{", body):
                if m.group(1) not in structs:
                    continue
                depth, i = 1, m.end()
                while depth:
                    depth += {"{": 1, "}": -1}.get(body[i], 0)
                    i += 1
                for part in _split_top_level(body[m.end():i - 1]):
                    km = re.match(r"^\s*([A-Za-z_]\w*)\s*:[^=]", part)
                    if km:
                        checked += 1
                        if km.group(1) not in structs[m.group(1)]:
                            bad.append("%s: %s{%s: ...}: no such field" % (os.path.basename(p), m.group(1), km.group(1)))
    return bad, checked


def test_selectors_on_the_packages_own_structs_exist(tmp_path):
    total = 0
    for d in sorted(glob.glob(os.path.join(ROOT, "integration", "go", "*"))):
        if os.path.isdir(d):
            bad, checked = _go_unknown_selectors(d)
            assert not bad, "\n".join(bad)
            total += checked
    assert total > 300, total
    pkg = tmp_path / "p"
    pkg.mkdir()
    (pkg / "a.go").write_text("package p\n\ntype t struct {\n\ta, b int\n\tname string\n}\n\nfunc (x *t) get() int { return x.a + x.c }\n\n"
                              "func mk(y *t) *t {\n\tz := &t{a: 1, d: 2}\n\tz.b = y.nam\n\treturn z\n}\n")
    bad, _ = _go_unknown_selectors(str(pkg))
    assert len(bad) == 3 and any("x.c is neither" in b for b in bad) and any("y.nam is neither" in b for b in bad) \
        and any("t{d: ...}: no such field" in b for b in bad), bad
