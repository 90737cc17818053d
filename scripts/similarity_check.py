"""Function-body similarity of this repo's Python against the reference (build container only:
needs /root/reference).  Method of VERDICT r3: for every function in ``rlpyt_amd/``, ``ast.unparse``
with docstrings removed, tokenise, and compare with every SAME-NAMED function of the reference
package by ``difflib.SequenceMatcher`` over the token lists.  Prints the functions at >= 0.8 and the
share of product Python tokens they hold, per file and in total.

usage: python scripts/similarity_check.py [--min 0.8] [--all]"""
import ast
import difflib
import io
import os
import sys
import tokenize
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/rlpyt"


def strip_docstrings(node):
    for n in ast.walk(node):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef, ast.Module)):
            b = n.body
            if b and isinstance(b[0], ast.Expr) and isinstance(getattr(b[0], "value", None), ast.Constant) \
                    and isinstance(b[0].value.value, str):
                n.body = b[1:] or [ast.Pass()]
    return node


def tokens(src):
    out = []
    try:
        for t in tokenize.generate_tokens(io.StringIO(src).readline):
            if t.type in (tokenize.NEWLINE, tokenize.NL, tokenize.INDENT, tokenize.DEDENT,
                          tokenize.COMMENT, tokenize.ENDMARKER):
                continue
            out.append(t.string)
    except tokenize.TokenError:
        pass
    return out


def functions(path):
    try:
        tree = ast.parse(open(path).read())
    except SyntaxError:
        return []
    out = []
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef)):
            body = strip_docstrings(n)
            out.append((n.name, n.lineno, tokens(ast.unparse(body))))
    return out


def py_files(root):
    for d, _, fs in os.walk(root):
        if "__pycache__" in d:
            continue
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def main():
    thr = 0.8
    if "--min" in sys.argv:
        thr = float(sys.argv[sys.argv.index("--min") + 1])
    ref = defaultdict(list)
    for p in py_files(REF):
        for name, line, tk in functions(p):
            ref[name].append((os.path.relpath(p, "/root/reference"), line, tk))
    total = 0
    per_file = defaultdict(lambda: [0, 0])
    hits = []
    for p in sorted(py_files(os.path.join(ROOT, "rlpyt_amd"))):
        rel = os.path.relpath(p, ROOT)
        for name, line, tk in functions(p):
            total += len(tk)
            per_file[rel][1] += len(tk)
            best, where = 0., None
            for rp, rl, rtk in ref.get(name, ()):
                r = difflib.SequenceMatcher(None, tk, rtk, autojunk=False).ratio()
                if r > best:
                    best, where = r, f"{rp}:{rl}"
            if best >= thr:
                per_file[rel][0] += len(tk)
                hits.append((rel, line, name, len(tk), best, where))
    sim = sum(h[3] for h in hits)
    for h in sorted(hits, key=lambda x: (-x[3])):
        print(f"{h[4]:.2f} {h[3]:5d} tok  {h[0]}:{h[1]} {h[2]}  <- {h[5]}")
    print()
    for f, (s, t) in sorted(per_file.items(), key=lambda kv: -kv[1][0]):
        if s:
            print(f"{s:6d} / {t:6d} = {s / max(t, 1):.2f}  {f}")
    print(f"\nTOTAL: {sim} of {total} function-body tokens at >= {thr}: {100. * sim / max(total, 1):.1f} %")


if __name__ == "__main__":
    main()
