"""Names a function reads as globals that the module never binds (the tests that only a GPU box runs are
otherwise first executed at the end of a round): python scripts/undefined_names.py [files...]"""
import builtins
import glob
import symtable
import sys


def check(path):
    top = symtable.symtable(open(path).read(), path, "exec")
    bound = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    bound |= set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    bad = []

    def walk(t):
        for s in t.get_symbols():
            if s.is_referenced() and s.is_global() and not s.is_assigned() and s.get_name() not in bound:
                bad.append((t.get_name(), t.get_lineno(), s.get_name()))
        for c in t.get_children():
            walk(c)

    for c in top.get_children():
        walk(c)
    return bad


if __name__ == "__main__":
    files = sys.argv[1:] or (glob.glob("tests/*.py") + glob.glob("warpx_amd/*.py") + glob.glob("scripts/*.py") +
                             ["bench.py", "__graft_entry__.py"])
    n = 0
    for f in files:
        for fn, line, name in check(f):
            print(f"{f}:{line}: {fn}() reads the global '{name}', which the module never binds")
            n += 1
    sys.exit(1 if n else 0)
