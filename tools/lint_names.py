"""Undefined-name check for the test files (no pyflakes in the image): python tools/lint_names.py FILE...   Flags names that are loaded somewhere in a module but
never bound anywhere in it (imports, assignments, defs, arguments, comprehension targets) and are not builtins -- what a GPU-only test would otherwise reveal
one gpurun call later."""
import ast, builtins, sys
bad = 0
for path in sys.argv[1:]:
    tree = ast.parse(open(path).read(), path)
    bound = set(dir(builtins)) | {'__file__', '__name__', '__doc__'}
    for n in ast.walk(tree):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            bound.update((a.asname or a.name).split('.')[0] for a in n.names)
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(n.name)
            if not isinstance(n, ast.ClassDef):
                a = n.args
                bound.update(x.arg for x in a.args + a.kwonlyargs + a.posonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []))
        elif isinstance(n, ast.Lambda):
            a = n.args
            bound.update(x.arg for x in a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []))
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound:
            print(f'{path}:{n.lineno}: undefined name {n.id!r}')
            bad += 1
sys.exit(1 if bad else 0)
