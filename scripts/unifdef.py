#!/usr/bin/env python3
"""Resolve preprocessor conditionals on a given set of macros and leave every other line alone (a small `unifdef`).

  python scripts/unifdef.py -DSOME_FORK=1 -DOTHER_FORK=0 [-UBF_CHECK] file.h > file.resolved.h      (round 4 used it with -DBF_FAST_EXTEND=1 -DBF_REFILL=0; those forks are gone)

Used in round 4 to collapse the experiment forks of bt_best.h once the GPU had decided them.  A conditional whose
expression mentions a macro that is not given stays as it is (its body is still processed); `#ifndef X / #define X v /
#endif` default blocks of a given macro are dropped.  Expressions may use defined(), !, &&, ||, comparisons, integers.
"""
import re
import sys


def parse_args(argv):
    defs, undefs, files = {}, set(), []
    for a in argv:
        if a.startswith("-D"):
            k, _, v = a[2:].partition("=")
            defs[k] = v if v != "" else "1"
        elif a.startswith("-U"):
            undefs.add(a[2:])
        else:
            files.append(a)
    return defs, undefs, files


IDENT = re.compile(r"[A-Za-z_][A-Za-z_0-9]*")


def evaluate(expr, defs, undefs):
    """-> True / False, or None if the expression depends on a macro that is not given"""
    e = re.sub(r"/\*.*?\*/", "", expr).strip()
    e = re.sub(r"//.*$", "", e).strip()

    def repl_defined(m):
        name = m.group(1) or m.group(2)
        if name in defs:
            return " 1 "
        if name in undefs:
            return " 0 "
        return " __UNKNOWN__ "
    e = re.sub(r"defined\s*\(\s*([A-Za-z_][A-Za-z_0-9]*)\s*\)|defined\s+([A-Za-z_][A-Za-z_0-9]*)", repl_defined, e)

    def repl_ident(m):
        name = m.group(0)
        if name in defs:
            return "(" + defs[name] + ")"
        if name in undefs:
            return "0"
        return "__UNKNOWN__"
    e = IDENT.sub(repl_ident, e)
    if "__UNKNOWN__" in e:
        # partial knowledge could still decide (0 && x), but keeping the conditional is always safe
        return None
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    e = re.sub(r"(\d+)[uUlL]+", r"\1", e)
    try:
        return bool(eval(e, {"__builtins__": {}}, {}))
    except Exception:
        return None


def unifdef(lines, defs, undefs):
    out = []
    # stack entries: dict(kind= "known" | "kept", emitting=bool, taken=bool, parent_emitting=bool)
    stack = []

    def emitting():
        return all(s["emitting"] for s in stack)

    i = 0
    n = len(lines)
    while i < n:
        line = lines[i]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", line)
        if not m:
            if emitting():
                out.append(line)
            i += 1
            continue
        kw, rest = m.group(1), m.group(2)
        if kw in ("if", "ifdef", "ifndef"):
            if kw == "ifdef":
                expr = "defined(%s)" % rest.strip().split()[0]
            elif kw == "ifndef":
                name = rest.strip().split()[0]
                expr = "!defined(%s)" % name
                # default block of a given macro: #ifndef X / #define X ... / #endif -> gone
                if name in defs and i + 2 < n and re.match(r"\s*#\s*define\s+%s\b" % re.escape(name), lines[i + 1]) \
                        and re.match(r"\s*#\s*endif\b", lines[i + 2]):
                    i += 3
                    continue
            else:
                expr = rest
            v = evaluate(expr, defs, undefs) if emitting() else False
            if not emitting():
                stack.append(dict(kind="known", emitting=False, taken=True))
            elif v is None:
                out.append(line)
                stack.append(dict(kind="kept", emitting=True, taken=False))
            else:
                stack.append(dict(kind="known", emitting=v, taken=v))
        elif kw == "elif":
            s = stack[-1]
            if s["kind"] == "kept":
                out.append(line)
            else:
                outer = all(t["emitting"] for t in stack[:-1])
                if s["taken"] or not outer:
                    s["emitting"] = False
                else:
                    v = evaluate(rest, defs, undefs)
                    if v is None:
                        raise SystemExit("line %d: #elif on an unknown macro after a resolved #if: resolve by hand" % (i + 1))
                    s["emitting"] = v
                    s["taken"] = v
        elif kw == "else":
            s = stack[-1]
            if s["kind"] == "kept":
                out.append(line)
            else:
                outer = all(t["emitting"] for t in stack[:-1])
                s["emitting"] = outer and not s["taken"]
                s["taken"] = True
        else:  # endif
            s = stack.pop()
            if s["kind"] == "kept":
                out.append(line)
        i += 1
    if stack:
        raise SystemExit("unbalanced conditionals")
    return out


def main():
    defs, undefs, files = parse_args(sys.argv[1:])
    if len(files) != 1:
        raise SystemExit(__doc__)
    with open(files[0]) as f:
        lines = f.read().split("\n")
    sys.stdout.write("\n".join(unifdef(lines, defs, undefs)))


if __name__ == "__main__":
    main()
