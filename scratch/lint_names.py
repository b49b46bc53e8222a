"""Undefined-name check for the package's Python modules (no pyflakes in the image): every Name that is loaded must be bound somewhere in
the module (assignment, def, class, import, argument, comprehension / with / except / for target) or be a builtin.  Coarse (scopes are
not modelled), but it catches a helper that an edit deleted -- which otherwise shows up as a NameError on the GPU box, minutes later.
    python scratch/lint_names.py            -> exit code 1 and the offenders if any"""
import ast
import builtins
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
for pat in ("tf-faster-rcnn_amd/frcnn_hip/*.py", "tf-faster-rcnn_amd/lib/*/*.py", "tf-faster-rcnn_amd/tools/*.py", "tests/*.py", "oracle/*.py"):
    files += sorted(glob.glob(os.path.join(ROOT, pat)))
bad = 0
for f in files:
    tree = ast.parse(open(f).read(), f)
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(n.name)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for x in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                bound.add(x.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                bound.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            bound.update(n.names)
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound:
            print("%s:%d: undefined name %r" % (os.path.relpath(f, ROOT), n.lineno, n.id))
            bad += 1
    # `self` / `cls` loaded inside a method that does not receive it (a @staticmethod edited as if it were a method: round 5 lost 8 GPU-minutes
    # to exactly that); nested functions and lambdas see the enclosing method's arguments
    def scan(fn, seen):
        a = fn.args
        seen = seen | {x.arg for x in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else [])}
        out = []
        for ch in ast.iter_child_nodes(fn):
            stack = [ch]
            while stack:
                n = stack.pop()
                if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                    out += scan(n, seen)
                    continue
                if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id in ("self", "cls") and n.id not in seen:
                    out.append((n.lineno, n.id))
                stack.extend(ast.iter_child_nodes(n))
        return out
    for c in ast.walk(tree):
        if isinstance(c, ast.ClassDef):
            for m in c.body:
                if isinstance(m, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    for ln, name in scan(m, set()):
                        print("%s:%d: %r used in %s.%s, which does not receive it" % (os.path.relpath(f, ROOT), ln, name, c.name, m.name))
                        bad += 1
sys.exit(1 if bad else 0)
