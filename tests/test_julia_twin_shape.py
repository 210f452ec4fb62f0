"""julia/PenguinHIP.jl cannot be parsed here (no Julia in the image).  Two things can still be checked on the CPU: that its
blocks and brackets balance (a tokenizer that strips strings, characters and comments and pairs every block opener with an
`end`), and that every symbol it `ccall`s is declared in include/penguin_hip.h -- and, for the struct it passes by reference to
pg_solver_run, that it has as many fields as the C struct."""
import re
from pathlib import Path

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


def test_run_info_struct_has_the_c_fields():
    c = re.search(r"typedef struct \{([^}]*)\} pg_run_info;", HDR, re.S).group(1)
    c = re.sub(r"/\*.*?\*/", "", c, flags=re.S)
    c_fields = re.findall(r"\b(?:int64_t|int32_t|double)\s+(\w+)\s*;", c)
    j = re.search(r"mutable struct pg_run_info\n(.*?)\n\s*pg_run_info\(\)", JL, re.S).group(1)
    j_fields = re.findall(r"(\w+)::(?:Int64|Int32|Float64)", j)
    assert j_fields == c_fields, (j_fields, c_fields)
