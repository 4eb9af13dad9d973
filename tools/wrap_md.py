"""Re-flow the prose of a Markdown file to a column limit ("split DESIGN.md lines at 160 columns — a design nobody can diff is not reviewed").

    python tools/wrap_md.py DESIGN.md [--width 160] [--check]

Only paragraphs / list items (`* `, `- `, `1. `, `3k. `, with any indent) that hold an over-long line are touched: their lines are joined and wrapped again with a hanging
indent under the item's text, so the rendered document does not change.  Table rows, fenced / indented code, headings, HTML comments and the generated `<!-- numbers:* -->` blocks are left alone (a table row cannot be
split; `--check` lists those that stay over the limit and exits 1 if a wrappable line is over it)."""
import argparse
import re
import sys
import textwrap


def wrap_line_unused(line, width):
    m = re.match(r"^(\s*)((?:[*+-]|\d+[.)]|\d+[a-z]'?\.)\s+)?(.*)$", line)
    indent, bullet, text = m.group(1), m.group(2) or "", m.group(3)
    first = indent + bullet
    rest = indent + " " * len(bullet)
    out = textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)
    return out or [line]


BULLET = re.compile(r"^(\s*)((?:[*+-]|\d+[.)]|\d+[a-z]'?\.)\s+)")


def process(lines, width):
    """re-flow paragraph by paragraph: consecutive prose lines of one paragraph / list item are joined and wrapped again when any of them is over the limit"""
    out, stuck, fenced, numbers = [], [], False, False
    para = []                                                  # (first-line prefix, rest prefix, [text pieces]) of the paragraph being collected

    def flush():
        nonlocal para
        if not para:
            return
        first, rest, pieces, raw = para
        if all(len(r) <= width for r in raw):
            out.extend(raw)                                    # nothing over the limit: leave the author's line breaks alone
        else:
            out.extend(textwrap.wrap(" ".join(pieces), width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False))
        para = []

    prev_blank = True
    for i, l in enumerate(lines):
        s = l.rstrip("\n")
        st = s.lstrip()
        if st.startswith("```"):
            flush(); fenced = not fenced; out.append(s); prev_blank = False; continue
        if "<!-- numbers:" in s:
            numbers = True
        if fenced or numbers or st.startswith("|") or s.startswith("#") or st.startswith("<!--") or st == "" or (prev_blank and s.startswith("    ") and not BULLET.match(s)):
            flush()
            if "<!-- /numbers:" in s:
                numbers = False
            if len(s) > width and not fenced:
                stuck.append(i + 1)
            out.append(s); prev_blank = st == ""; continue
        prev_blank = False
        m = BULLET.match(s)
        if m:                                                  # a new list item
            flush()
            para = [m.group(1) + m.group(2), m.group(1) + " " * len(m.group(2)), [s[m.end():].strip()], [s]]
        elif para:
            para[2].append(st); para[3].append(s)
        else:
            ind = s[:len(s) - len(st)]
            para = [ind, ind, [st], [s]]
    flush()
    return out, stuck


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    bad = 0
    for f in a.files:
        lines = open(f).read().split("\n")
        new, stuck = process(lines, a.width)
        if new != lines:
            if a.check:
                bad += 1
                print("%s: wrappable lines over %d columns" % (f, a.width))
            else:
                open(f, "w").write("\n".join(new))
        if stuck:
            print("%s: %d table / generated lines stay over %d columns (first: line %d)" % (f, len(stuck), a.width, stuck[0]))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
