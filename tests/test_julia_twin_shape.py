"""julia/PenguinHIP.jl cannot be parsed here (no Julia in the image).  Two things can still be checked on the CPU: that its
blocks and brackets balance (a tokenizer that strips strings, characters and comments and pairs every block opener with an
`end`), and that every symbol it `ccall`s is declared in include/penguin_hip.h -- and, for the struct it passes by reference to
pg_solver_run, that it has as many fields as the C struct."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
JL = (ROOT / "julia" / "PenguinHIP.jl").read_text(encoding="utf-8")
HDR = (ROOT / "include" / "penguin_hip.h").read_text(encoding="utf-8")


def _strip(src: str) -> str:
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith('"""', i):
            j = src.find('"""', i + 3)
            i = n if j < 0 else j + 3
            out.append(' "" ')
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            i = j + 1
            out.append(' "" ')
        elif c == "#":
            if src.startswith("#=", i):
                j = src.find("=#", i + 2)
                i = n if j < 0 else j + 2
            else:
                j = src.find("\n", i)
                i = n if j < 0 else j
        elif c == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src.find("'", i + 2) in range(i + 3, i + 8))):
            i = src.find("'", i + 2) + 1
            out.append(" 'c' ")
        else:
            out.append(c)
            i += 1
    return "".join(out)


def test_blocks_and_brackets_balance():
    txt = _strip(JL)
    openers = {"function", "if", "for", "while", "let", "begin", "struct", "module", "try", "do", "quote", "macro"}
    sq = par = 0
    stack, prev, line = [], None, 1
    for m in re.finditer(r"[A-Za-z_ -￿][A-Za-z_0-9! -￿]*|[\[\]\(\)]|\n", txt):
        t = m.group(0)
        if t == "\n":
            line += 1
            continue
        if t == "[":
            sq += 1
        elif t == "]":
            sq -= 1
        elif t == "(":
            par += 1
        elif t == ")":
            par -= 1
        elif t in openers or (t == "type" and prev in ("abstract", "primitive")):
            if not (t in ("for", "if") and (sq > 0 or par > 0)):      # comprehensions / generators carry no `end`
                stack.append((t, line))
        elif t == "end" and sq == 0:                                   # a[end] is an index, not a block end
            assert stack, f"`end` without an opener at line {line}"
            stack.pop()
        assert sq >= 0 and par >= 0, f"closing bracket without an opener at line {line}"
        prev = t
    assert not stack, f"unclosed blocks: {stack[-5:]}"
    assert sq == 0 and par == 0


def test_every_ccall_binds_a_declared_symbol():
    declared = set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", HDR))
    called = set(re.findall(r"ccall\(\(:(pg_[a-z0-9_]+),\s*libpg\)", JL))
    assert len(called) >= 30
    assert not (called - declared), f"ccall'ed but not declared in include/penguin_hip.h: {sorted(called - declared)}"


def _c_fields(name):
    body = re.search(r"typedef struct \{([^}]*)\} %s;" % name, HDR, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for typ, decl in re.findall(r"\b(const double\s*\*|int64_t|int32_t|double)\s*([^;]+);", body):
        for nm in decl.split(","):
            nm = nm.strip()
            m = re.match(r"(\w+)\s*(?:\[(\d+)\])?$", nm)
            assert m, (name, nm)
            out.append((m.group(1), typ.replace(" ", ""), int(m.group(2)) if m.group(2) else 0))   # (name, C type, array length)
    return out


def _jl_fields(name):
    m = re.search(r"struct %s\b(.*?)\bend\b" % name, JL, re.S)
    assert m, name
    body = m.group(1)
    body = body.split("%s(" % name)[0]            # (an inner constructor follows the fields)
    return re.findall(r"(\w+)::((?:NTuple\{\d+,\s*\w+\})|(?:Ptr\{\w+\})|\w+)", body)


_JL_OF_C = {"int64_t": "Int64", "int32_t": "Int32", "double": "Float64", "constdouble*": "Ptr{Float64}"}


@pytest.mark.parametrize("name", ["pg_bc_desc", "pg_border_desc", "pg_jump_desc", "pg_krylov_opts", "pg_step_info", "pg_run_info",
                                  "pg_motion_desc"])
def test_structs_passed_across_the_abi_match_field_for_field(name):
    """Same field names, order and types as the C struct (arrays as NTuple of the element type)."""
    c, j = _c_fields(name), _jl_fields(name)
    assert [f[0] for f in c] == [f[0] for f in j], (c, j)
    for (cn, ct, cl), (jn, jt) in zip(c, j):
        want = _JL_OF_C[ct]
        if cl:
            assert re.fullmatch(r"NTuple\{%d,\s*%s\}" % (cl, want), jt), (name, cn, jt)
        else:
            assert jt == want, (name, cn, ct, jt)


def _top_level_split(txt):
    parts, depth, cur = [], 0, []
    for ch in txt:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    last = "".join(cur).strip()
    if last:
        parts.append(last)
    return parts


def _balanced(txt, start):
    """txt[start] == '(' -> index just past its partner."""
    depth = 0
    for k in range(start, len(txt)):
        if txt[k] == "(":
            depth += 1
        elif txt[k] == ")":
            depth -= 1
            if depth == 0:
                return k + 1
    raise AssertionError("unbalanced")


def test_every_ccall_passes_as_many_arguments_as_the_c_prototype_takes():
    hdr = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint32_t\s+(pg_[a-z0-9_]+)\s*\(", hdr):
        end = _balanced(hdr, m.end() - 1)
        args = hdr[m.end():end - 1].strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_top_level_split(args))
    txt = _strip(JL)
    seen = 0
    for m in re.finditer(r"ccall\(\(:(pg_[a-z0-9_]+),\s*libpg\),\s*(\w+),\s*\(", txt):
        name = m.group(1)
        tend = _balanced(txt, m.end() - 1)
        types = _top_level_split(txt[m.end():tend - 1])
        call_end = _balanced(txt, m.start() + len("ccall"))
        rest = txt[tend:call_end - 1].lstrip()
        assert rest.startswith(","), (name, rest[:40])
        values = _top_level_split(rest[1:])
        assert len(types) == protos[name], (name, "types", len(types), "C parameters", protos[name])
        assert len(values) == len(types), (name, "values", len(values), "types", len(types))
        assert m.group(2) == "Int32", (name, m.group(2))
        seen += 1
    assert seen >= 45


_J_OF_C = {"int32_t": "Int32", "int64_t": "Int64", "double": "Float64", "size_t": "Csize_t", "char": "UInt8", "unsigned char": "UInt8",
           "uint8_t": "UInt8", "void": "Cvoid", "uint64_t": "UInt64", "unsigned long long": "UInt64"}
_ABI_STRUCTS = {"pg_bc_desc", "pg_border_desc", "pg_jump_desc", "pg_krylov_opts", "pg_step_info", "pg_run_info", "pg_motion_desc",
                "pg_system_info"}


def _c_param(arg):
    """(base type, pointer level) of one C parameter declaration."""
    a = re.sub(r"\s+", " ", arg.replace("const", " ").strip())
    m = re.match(r"(.*?)(\w+)\s*(\[\d*\])?$", a)
    base, arr = m.group(1).strip(), m.group(3)
    return base.replace("*", "").strip(), base.count("*") + (1 if arr else 0)


def _jl_param(ty):
    lvl = 0
    while True:
        m = re.fullmatch(r"(?:Ptr|Ref)\{(.*)\}", ty.strip())
        if not m:
            return ty.strip(), lvl
        ty, lvl = m.group(1), lvl + 1


def test_every_ccall_argument_type_matches_the_c_parameter():
    """Base type and pointer level of every argument: Int32 / Int64 / Float64 / Csize_t by value, Ptr or Ref of them for
    pointers, Ptr{Cvoid} for the opaque handles (Ref{Ptr{Cvoid}} for handle outputs), the ABI structs by their own name."""
    hdr = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint32_t\s+(pg_[a-z0-9_]+)\s*\(", hdr):
        end = _balanced(hdr, m.end() - 1)
        args = hdr[m.end():end - 1].strip()
        protos[m.group(1)] = [] if args in ("", "void") else _top_level_split(args)
    txt = _strip(JL)
    bad = []
    for m in re.finditer(r"ccall\(\(:(pg_[a-z0-9_]+),\s*libpg\),\s*(\w+),\s*\(", txt):
        name = m.group(1)
        tend = _balanced(txt, m.end() - 1)
        for k, (ca, jt) in enumerate(zip(protos[name], _top_level_split(txt[m.end():tend - 1]))):
            cb, cl = _c_param(ca)
            want = ("Cvoid", cl) if (cb.startswith("pg_") and cb not in _ABI_STRUCTS) else (_J_OF_C.get(cb, cb), cl)
            if _jl_param(jt) != want:
                bad.append((name, k, ca, jt))
    assert not bad, bad
