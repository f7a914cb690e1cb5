"""bench.py at N > 1: every call that holds a collective -- a training step (its backward issues the gradient all-reduces), a fence
(barrier), a precision-mode leg, an explicit torch.distributed call -- must be reached by EVERY rank.  A rank-0-only guard around one of
them leaves rank 0 waiting for peers that never come (round 4 found exactly that in the per-kernel event pass, which only a run with
more than one rank and the roofline leg on could have shown).  The multi-rank run itself needs GPUs; this is the static half: no such
call may sit lexically inside an ``if`` whose condition mentions ``rank``."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLLECTIVE_NAMES = {"step", "fence", "timed_mode", "ddp_step", "stock_ddp_leg"}                    # bare-name calls inside bench.main()
COLLECTIVE_ATTRS = {("dist", "all_reduce"), ("dist", "barrier"), ("dist", "broadcast"), ("dist", "all_gather"), ("sync", "finish")}


def _is_collective(call: ast.Call) -> bool:
    f = call.func
    if isinstance(f, ast.Name):
        return f.id in COLLECTIVE_NAMES
    if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name):
        return (f.value.id, f.attr) in COLLECTIVE_ATTRS
    return False


def _mentions_rank(test: ast.AST) -> bool:
    return any(isinstance(n, ast.Name) and n.id in ("rank", "local_rank") for n in ast.walk(test))


def _offenders(fn: ast.FunctionDef):
    bad = []

    def walk(node, guarded):
        for child in ast.iter_child_nodes(node):
            if isinstance(child, ast.If) and _mentions_rank(child.test):
                for sub in child.body + child.orelse:
                    walk_stmt(sub, True)
                continue
            walk_stmt(child, guarded)

    def walk_stmt(node, guarded):
        if guarded:
            for n in ast.walk(node):
                if isinstance(n, ast.Call) and _is_collective(n):
                    bad.append((n.lineno, ast.unparse(n.func)))
            return
        walk(node, guarded)
    walk(fn, False)
    return bad


def test_no_collective_under_a_rank_guard():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    # the analysis sees the calls it is meant to see ...
    seen = [n for n in ast.walk(main) if isinstance(n, ast.Call) and _is_collective(n)]
    assert len(seen) >= 10, len(seen)
    # ... and none of them under a rank condition
    assert _offenders(main) == []


def test_the_check_catches_a_rank_zero_only_step():
    bad = ast.parse("def main():\n    if rank == 0 and not skip:\n        for _ in range(2):\n            step()\n    fence()\n").body[0]
    assert _offenders(bad) == [(4, "step")]
    good = ast.parse("def main():\n    if not skip:\n        step()\n    if rank == 0:\n        print(1)\n").body[0]
    assert _offenders(good) == []
