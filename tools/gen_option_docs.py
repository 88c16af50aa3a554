#!/usr/bin/env python3
"""The option table of include/ccmp.h, generated from the library's own table (ccmp_ctx_option_info: csrc/ccmp_policy.cpp), so that
the header cannot state a default the code does not have.

    python tools/gen_option_docs.py            print the block
    python tools/gen_option_docs.py --write    rewrite it in include/ccmp.h between its BEGIN / END markers

tests/test_host_cabi.py::test_header_option_table_is_the_librarys compares the two on every run of the CPU suite."""
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HEADER = os.path.join(ROOT, "include", "ccmp.h")
BEGIN = "/* BEGIN OPTION TABLE (generated from csrc/ccmp_policy.cpp: python tools/gen_option_docs.py --write)"
END = " * END OPTION TABLE */"
LONG_MAX = (1 << 63) - 1


def block():
    from closed_chain_motion_planner_amd import option_table

    lines = [BEGIN, " *   name                            default   range         meaning"]
    for o in option_table():
        hi = "max" if o["hi"] >= LONG_MAX else str(o["hi"])
        head = " *   %-31s %-9d %-13s " % ('"%s"' % o["name"], o["default"], "%d..%s" % (o["lo"], hi))
        body = textwrap.wrap(o["doc"], 150 - len(head)) or [""]
        lines.append(head + body[0])
        lines += [" *   " + " " * (len(head) - 5) + b for b in body[1:]]
    lines.append(END)
    return "\n".join(lines) + "\n"


def main():
    text = block()
    if "--write" not in sys.argv:
        sys.stdout.write(text)
        return
    src = open(HEADER).read()
    a, b = src.index(BEGIN), src.index(END) + len(END) + 1
    open(HEADER, "w").write(src[:a] + text + src[b:])
    print("rewrote the option table of", HEADER)


if __name__ == "__main__":
    main()
