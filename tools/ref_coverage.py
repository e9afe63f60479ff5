#!/usr/bin/env python3
"""Which lines of the reference's hot path do the golden tapes execute?  (build container only)

Runs tools/gen_golden.py's whole fixture generation (into a scratch directory, the committed fixtures are not touched)
under a line tracer restricted to the reference's env / vehicle / road / safety modules and reports, per function of
SURVEY 8a, the executable lines no tape reached -- the branches of the reference that the parity chain does not pin.

    python tools/ref_coverage.py  ->  profiles/r02/reference_coverage.json
"""
import json
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
os.environ["MM_GOLDEN_OUT"] = os.environ.get("MM_COV_OUT", "/tmp/golden_cov")
REF = "/root/reference/highway_env/"
WATCH = ("envs/common/abstract.py", "envs/common/action.py", "envs/common/observation.py", "envs/merge_env_v1.py",
         "road/road.py", "road/lane.py", "vehicle/kinematics.py", "vehicle/controller.py", "vehicle/safe_controller.py",
         "vehicle/behavior.py", "vehicle/safety/cbf.py", "vehicle/safety/decentral_layer.py", "utils.py")
hits = {}


def tracer(frame, event, arg):
    fn = frame.f_code.co_filename
    if not fn.startswith(REF):
        return None
    rel = fn[len(REF):]
    if rel not in WATCH:
        return None
    s = hits.setdefault(rel, set())

    def local(frame, event, arg):
        if event == "line":
            s.add(frame.f_lineno)
        return local
    s.add(frame.f_lineno)
    return local


def executable_lines(path):
    """Line numbers that carry code, per function (qualified name -> sorted lines), from the compiled code objects."""
    src = open(path).read()
    top = compile(src, path, "exec")
    out = {}

    def walk(code, prefix):
        name = prefix + code.co_name if code.co_name != "<module>" else ""
        if name:
            lines = sorted({l for _, _, l in code.co_lines() if l is not None and l != code.co_firstlineno})
            out[name] = lines
        for c in code.co_consts:
            if hasattr(c, "co_code"):
                walk(c, (name + ".") if name else "")
    walk(top, "")
    return out


def main():
    sys.argv = [sys.argv[0]]
    sys.path.insert(0, HERE)
    threading.settrace(tracer)
    sys.settrace(tracer)
    import gen_golden
    gen_golden.main()
    sys.settrace(None)
    report = {}
    for rel in WATCH:
        ex = executable_lines(REF + rel)
        got = hits.get(rel, set())
        funcs = {}
        for fn, lines in ex.items():
            if not lines:
                continue
            miss = [l for l in lines if l not in got]
            if any(l in got for l in lines) or fn.split(".")[-1] in ("step", "act"):  # functions the path enters at all
                funcs[fn] = {"lines": len(lines), "missed": miss}
        report[rel] = funcs
    out = os.path.join(REPO, "profiles", "r02", "reference_coverage.json")
    json.dump(report, open(out, "w"), indent=1)
    tot = sum(f["lines"] for r in report.values() for f in r.values())
    mis = sum(len(f["missed"]) for r in report.values() for f in r.values())
    print("entered functions: %d executable lines, %d never executed (%.1f %%) -> %s" % (tot, mis, 100.0 * mis / max(tot, 1), out))


if __name__ == "__main__":
    main()
